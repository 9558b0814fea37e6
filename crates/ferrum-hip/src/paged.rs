//! `BackendPagedKv` (traits.rs:1640-1900): paged KV write + attention over the native block layout.
use crate::{backend::{HipBackend, HipBuf, HipCtx}, check, ffi};
use ferrum_kernels::backend::BackendPagedKv;
use ferrum_types::Result;
use std::os::raw::c_int;

impl BackendPagedKv for HipBackend {
    fn supports_paged_kv() -> bool { true }
    fn supports_varlen_qkv() -> bool { true }

    // traits.rs:1719 — q_len == 1: one token per sequence; q_len > 1: single-sequence causal prefill, head-major q / out
    fn paged_decode_attention(ctx: &mut HipCtx, q: &HipBuf, k_pool: &HipBuf, v_pool: &HipBuf, out: &mut HipBuf, block_tables: &HipBuf,
                              context_lens: &HipBuf, num_seqs: usize, num_heads: usize, num_kv_heads: usize, head_dim: usize, block_size: usize,
                              max_num_blocks_per_seq: usize, q_len: usize) -> Result<()> {
        check(unsafe {
            ffi::ferrum_hip_paged_decode_attention_f16(q.ptr, k_pool.ptr, v_pool.ptr, out.ptr, block_tables.ptr as *const i32,
                context_lens.ptr as *const u32, num_seqs as c_int, num_heads as c_int, num_kv_heads as c_int, head_dim as c_int,
                block_size as c_int, max_num_blocks_per_seq as c_int, q_len as c_int, ctx.ws, ctx.stream)
        })
    }
    // traits.rs:1813
    fn paged_varlen_attention(ctx: &mut HipCtx, q: &HipBuf, k_pool: &HipBuf, v_pool: &HipBuf, out: &mut HipBuf, cu_seqlens_q: &HipBuf,
                              pos_offsets: &HipBuf, block_tables: &HipBuf, num_seqs: usize, total_q_tokens: usize, max_kv_len: usize,
                              num_heads: usize, num_kv_heads: usize, head_dim: usize, sliding_window: usize, block_size: usize,
                              max_num_blocks_per_seq: usize) -> Result<()> {
        check(unsafe {
            ffi::ferrum_hip_paged_varlen_attention_f16(q.ptr, k_pool.ptr, v_pool.ptr, out.ptr, cu_seqlens_q.ptr as *const u32,
                pos_offsets.ptr as *const u32, block_tables.ptr as *const i32, num_seqs as c_int, total_q_tokens as c_int, max_kv_len as c_int,
                num_heads as c_int, num_kv_heads as c_int, head_dim as c_int, sliding_window as c_int, block_size as c_int,
                max_num_blocks_per_seq as c_int, 0, ctx.ws, ctx.stream)
        })
    }
    // traits.rs:1885
    fn paged_batched_decode_attention(ctx: &mut HipCtx, q: &HipBuf, k_pool: &HipBuf, v_pool: &HipBuf, out: &mut HipBuf, block_tables: &HipBuf,
                                      valid_kv_lens: &HipBuf, num_seqs: usize, max_kv_len: usize, num_heads: usize, num_kv_heads: usize,
                                      head_dim: usize, block_size: usize, max_num_blocks_per_seq: usize) -> Result<()> {
        check(unsafe {
            ffi::ferrum_hip_paged_batched_decode_attention_f16(q.ptr, k_pool.ptr, v_pool.ptr, out.ptr, block_tables.ptr as *const i32,
                valid_kv_lens.ptr as *const u32, num_seqs as c_int, max_kv_len as c_int, num_heads as c_int, num_kv_heads as c_int,
                head_dim as c_int, block_size as c_int, max_num_blocks_per_seq as c_int, ctx.ws, ctx.stream)
        })
    }
    // traits.rs:1764 — split [m_total, q_dim + 2·kv_dim] rows, per-head QK-norm + RoPE (qk_mode), Q token-major, K/V into the pool
    fn split_qkv_norm_rope_into_paged_cache_varlen(ctx: &mut HipCtx, qkv: &HipBuf, q_norm_w: &HipBuf, k_norm_w: &HipBuf, cos: &HipBuf, sin: &HipBuf,
                                                   q_out: &mut HipBuf, cache_k: &mut HipBuf, cache_v: &mut HipBuf, cu_seqlens_q: &HipBuf,
                                                   pos_offsets: &HipBuf, block_tables: &HipBuf, num_seqs: usize, m_total: usize, q_heads: usize,
                                                   kv_heads: usize, head_dim: usize, eps: f32, qk_mode: i32, block_size: usize,
                                                   max_blocks_per_seq: usize) -> Result<()> {
        // (f32 RoPE tables: the shadow Backend::from_slice keeps — see HipBuf)
        if cos.f32_shadow.is_null() || sin.f32_shadow.is_null() {
            return Err(ferrum_types::FerrumError::backend("split_qkv_norm_rope_into_paged_cache_varlen: cos / sin must come from Backend::from_slice"));
        }
        check(unsafe {
            ffi::ferrum_hip_split_qkv_norm_rope_into_paged_cache_varlen_f16(qkv.ptr, q_norm_w.ptr, k_norm_w.ptr, cos.f32_shadow as *const f32,
                sin.f32_shadow as *const f32, q_out.ptr, cache_k.ptr, cache_v.ptr, cu_seqlens_q.ptr as *const u32, pos_offsets.ptr as *const u32,
                block_tables.ptr as *const i32, num_seqs as c_int, m_total as c_int, q_heads as c_int, kv_heads as c_int, head_dim as c_int, eps,
                qk_mode as c_int, block_size as c_int, max_blocks_per_seq as c_int, ctx.stream)
        })
    }
}

/// Bytes of a K (or V) pool of `num_blocks` blocks in the native tile layout (kv_layout.h): what the model allocates with
/// `B::alloc_typed(Dtype::F16, bytes / 2)` before the first paged write.
pub fn paged_pool_bytes(num_blocks: usize, kv_heads: usize, head_dim: usize) -> usize {
    unsafe { ffi::ferrum_hip_paged_pool_bytes(num_blocks as c_int, kv_heads as c_int, head_dim as c_int) }
}
