//! `HipBackend`: the reference's backend trait stack over `libferrum_hip.so` (MI355X / gfx950).
//!
//! NOT COMPILED in the repository that ships the library (its image has no Rust toolchain): `ffi.rs` is generated from
//! `include/ferrum_hip.h` and verified against the library's exports; the impls below follow the reference's trait
//! signatures (`crates/ferrum-kernels/src/backend/traits.rs`, `capabilities.rs`, `ferrum-models/src/common/llm.rs`,
//! `ferrum-interfaces/src/model_executor.rs`) and are to be built inside the reference workspace.  `TRAIT_MAP` lists
//! every trait method next to the C entry point that implements it; the CPU test suite (tests/test_abi_symbols.py)
//! checks that each named entry point exists in the library AND — parsing the reference's trait sources — that every
//! required method of `Backend`, `BackendTimer`, `DecoderOnlyLLM` and `ModelExecutor` has an `fn` of the same name and
//! the same number of parameters here.
pub mod ffi;
mod backend;
mod collective;
mod executor;
mod moe;
mod paged;
mod quant;

pub use backend::{HipBackend, HipBuf, HipCtx, HipEventTimer};
pub use executor::{HipDecoderModel, HipModelExecutor};
pub use paged::paged_pool_bytes;

use ferrum_types::{FerrumError, Result};
use std::ffi::CStr;
use std::os::raw::c_int;

pub(crate) fn last_error() -> String {
    unsafe { CStr::from_ptr(ffi::ferrum_hip_last_error()).to_string_lossy().into_owned() }
}

/// rc 3 is the trait's "fall back" signal (`FerrumError::unsupported`, traits.rs:1095, capabilities.rs:147).
pub(crate) fn check(rc: c_int) -> Result<()> {
    match rc {
        ffi::FERRUM_HIP_OK => Ok(()),
        ffi::FERRUM_HIP_UNSUPPORTED => Err(FerrumError::unsupported(last_error())),
        _ => Err(FerrumError::backend(last_error())),
    }
}

/// Infallible core ops panic on misuse like the CUDA lane (traits.rs:202 ff.).
pub(crate) fn must(rc: c_int, what: &str) {
    if rc != ffi::FERRUM_HIP_OK {
        panic!("{what}: {}", last_error());
    }
}

/// (trait, method, reference file:line, C entry point) — one row per method the backend implements.
pub const TRAIT_MAP: &[(&str, &str, &str, &str)] = &[
    ("Backend", "alloc", "traits.rs:118", "ferrum_hip_alloc"),
    ("Backend", "alloc_typed", "traits.rs:98", "ferrum_hip_alloc"),
    ("Backend", "from_slice_typed", "traits.rs:104", "ferrum_hip_memcpy_h2d"),
    ("Backend", "write_typed", "traits.rs:110", "ferrum_hip_memcpy_h2d"),
    ("Backend", "from_slice", "traits.rs:130", "ferrum_hip_memcpy_h2d"),
    ("Backend", "to_vec", "traits.rs:124", "ferrum_hip_memcpy_d2h"),
    ("Backend", "zero_buffer", "traits.rs:140", "ferrum_hip_memset_zero"),
    ("Backend", "make_timer", "traits.rs:52", "ferrum_hip_event_create"),
    ("BackendTimer", "record_start", "timer.rs:98", "ferrum_hip_event_record"),
    ("BackendTimer", "elapsed_ms", "timer.rs:108", "ferrum_hip_event_elapsed_ms"),
    ("Backend", "split_qkv", "traits.rs:823", "ferrum_hip_split_qkv_f16"),
    ("Backend", "qk_norm_rope", "traits.rs:1238", "ferrum_hip_qk_norm_rope_f16"),
    ("Backend", "transpose_token_to_head", "traits.rs:1294", "ferrum_hip_transpose_token_to_head_f16"),
    ("Backend", "scaled_add_inplace", "traits.rs:1322", "ferrum_hip_scaled_add_inplace_f16"),
    ("Backend", "add_bias", "traits.rs:1359", "ferrum_hip_add_bias_f16"),
    ("Backend", "layer_norm", "traits.rs:1370", "ferrum_hip_layer_norm_f16"),
    ("Backend", "gelu", "traits.rs:1384", "ferrum_hip_gelu_f16"),
    ("Backend", "argmax_rows_f16_sparse_repetition_penalty", "traits.rs:1571", "ferrum_hip_apply_repetition_penalties_sparse_f16"),
    ("Backend", "sync", "traits.rs:88", "ferrum_hip_stream_synchronize"),
    ("Backend", "gemm", "traits.rs:190", "ferrum_hip_gemm_f16"),
    ("Backend", "rms_norm", "traits.rs:202", "ferrum_hip_rms_norm_f16"),
    ("Backend", "fused_add_rms_norm", "traits.rs:212", "ferrum_hip_fused_add_rms_norm_f16"),
    ("Backend", "flash_attention", "traits.rs:225", "ferrum_hip_flash_attention_f16"),
    ("Backend", "copy_slice", "traits.rs:798", "ferrum_hip_copy_slice_f16"),
    ("Backend", "embedding_lookup", "traits.rs:809", "ferrum_hip_embedding_lookup_f16"),
    ("Backend", "fused_silu_mul_split", "traits.rs:863", "ferrum_hip_fused_silu_mul_split_f16"),
    ("Backend", "fused_gelu_tanh_mul_split", "traits.rs:875", "ferrum_hip_fused_gelu_tanh_mul_split_f16"),
    ("Backend", "scale_inplace", "traits.rs:889", "ferrum_hip_scale_inplace_f16"),
    ("Backend", "kv_cache_append_head_major", "traits.rs:1266", "ferrum_hip_kv_cache_append_head_major_f16"),
    ("Backend", "transpose_head_to_token", "traits.rs:1281", "ferrum_hip_transpose_head_to_token_f16"),
    ("Backend", "add_inplace", "traits.rs:1308", "ferrum_hip_add_inplace_f16"),
    ("Backend", "argmax_rows_f16", "traits.rs:1534", "ferrum_hip_argmax_rows_f16"),
    ("BackendPagedKv", "split_qkv_norm_rope_into_paged_cache_varlen", "traits.rs:1764", "ferrum_hip_split_qkv_norm_rope_into_paged_cache_varlen_f16"),
    ("BackendPagedKv", "paged_decode_attention", "traits.rs:1719", "ferrum_hip_paged_decode_attention_f16"),
    ("BackendPagedKv", "paged_varlen_attention", "traits.rs:1813", "ferrum_hip_paged_varlen_attention_f16"),
    ("BackendPagedKv", "paged_batched_decode_attention", "traits.rs:1885", "ferrum_hip_paged_batched_decode_attention_f16"),
    ("BackendQuantMarlin", "load_gptq", "capabilities.rs:136", "ferrum_hip_gptq_load"),
    ("BackendQuantMarlin", "load_gptq_stacked", "capabilities.rs:170", "ferrum_hip_gptq_load_stacked"),
    ("Linear", "forward", "linear.rs:109", "ferrum_hip_gptq_linear_forward_f16"),
    ("MarlinExpertStack", "gemm_phase_batched", "marlin_expert_stack.rs:63", "ferrum_hip_moe_gemm_phase_batched_f16"),
    ("MarlinExpertStack", "gemm_phase_vllm", "marlin_expert_stack.rs:86", "ferrum_hip_moe_gemm_phase_f16"),
    ("BackendMoeFused", "route_topk_softmax", "capabilities.rs:334", "ferrum_hip_moe_route_topk_softmax_f16"),
    ("BackendMoeFused", "moe_build_pairs_by_token", "capabilities.rs:410", "ferrum_hip_moe_build_pairs_by_token"),
    ("BackendMoeFused", "moe_align_block_size", "capabilities.rs:429", "ferrum_hip_moe_align_block_size_packed_rows"),
    ("BackendMoeFused", "moe_align_block_size_pair_ids", "capabilities.rs:449", "ferrum_hip_moe_align_block_size"),
    ("BackendMoeFused", "weighted_sum_batched", "capabilities.rs:560", "ferrum_hip_weighted_sum_batched_f16"),
    ("BackendMoeFused", "weighted_sum_batched_offset", "capabilities.rs:580", "ferrum_hip_weighted_sum_batched_f16"),
    ("BackendMoeFused", "moe_combine", "capabilities.rs:684", "ferrum_hip_moe_combine_pairs_f16"),
    ("BackendGraph", "begin_graph_capture", "capabilities.rs:45", "ferrum_hip_graph_begin_capture"),
    ("BackendGraph", "end_graph_capture", "capabilities.rs:52", "ferrum_hip_graph_end_capture"),
    ("BackendGraph", "replay_graph", "capabilities.rs:58", "ferrum_hip_graph_replay"),
    ("BackendGraph", "reset_graph", "capabilities.rs:66", "ferrum_hip_graph_destroy"),
    ("BackendCollective", "world_size", "capabilities.rs:86", "ferrum_hip_comm_world_size"),
    ("BackendCollective", "rank", "capabilities.rs:89", "ferrum_hip_comm_rank"),
    ("BackendCollective", "all_reduce", "capabilities.rs:92", "ferrum_hip_all_reduce_f16"),
    ("BackendCollective", "all_gather", "capabilities.rs:95", "ferrum_hip_all_gather_f16"),
    ("BackendCollective", "broadcast", "capabilities.rs:104", "ferrum_hip_broadcast_f16"),
    ("DecoderOnlyLLM", "reserve_kv_slots", "common/llm.rs:133", "ferrum_hip_model_reserve_kv_slots"),
    ("DecoderOnlyLLM", "kv_slot_capacity_snapshot", "common/llm.rs:141", "ferrum_hip_model_kv_capacity_snapshot"),
    ("DecoderOnlyLLM", "unified_forward_with_logits_policy", "common/llm.rs:260", "ferrum_hip_model_unified_forward_ex"),
    ("DecoderOnlyLLM", "release", "common/llm.rs:278", "ferrum_hip_model_release"),
    ("ModelExecutor", "unified_decode", "model_executor.rs:567", "ferrum_hip_model_unified_forward_ex"),
];
