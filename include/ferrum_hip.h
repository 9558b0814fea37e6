/*
 * ferrum_hip.h — C ABI of the MI355X (gfx950) decode hot path for ferrum-infer-rs.
 *
 * This is the drop-in boundary: the entry points a Rust FFI crate (`ferrum-hip`) binds to implement
 * the reference's backend traits
 *     Backend + BackendGraph + BackendPagedKv + BackendQuantMarlin + BackendMoeFused (+ BackendCollective)
 *     = MoeLlmBackend          (crates/ferrum-kernels/src/backend/traits.rs:2196-2208)
 * in place of the CUDA lane of `ferrum-kernels`.  INTEGRATION.md shows the binding.
 *
 * Conventions (same as the reference's own C entry points: FA2 shim
 * crates/ferrum-kernels/src/backend/cuda/fa2_ffi.rs:16-37, Marlin cuda/marlin.rs:293-387):
 *   - every function returns 0 on success, non-zero on failure; the message is available from
 *     ferrum_hip_last_error() (thread-local) — capability ops that the reference returns
 *     `FerrumError::unsupported` for return FERRUM_HIP_UNSUPPORTED (3);
 *   - buffers are raw DEVICE pointers owned by the caller (never freed here); dims are plain ints;
 *     `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - activations are fp16, accumulation fp32; index tensors are i32/u32 exactly as in the traits;
 *   - no call allocates device memory after ferrum_hip_*_create / *_load, so every compute entry
 *     point may be captured into a hipGraph (capabilities.rs:205-214 contract).
 * All paths in comments are relative to the reference's `crates/` directory.
 */
#ifndef FERRUM_HIP_H
#define FERRUM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden */
#endif

#define FERRUM_HIP_OK 0
#define FERRUM_HIP_ERROR 1
#define FERRUM_HIP_INVALID 2
#define FERRUM_HIP_UNSUPPORTED 3

/* ── native-operator artifact contract ─────────────────────────────────────────────────────────
 * ferrum-native-ops/src/abi.rs:3-13: required exports of a native operator artifact and the
 * #[repr(C)] descriptor; the resolver checks them with `nm -g` (resolver.rs:136-318). */
typedef struct {
    uint32_t abi_version;             /* FERRUM_NATIVE_ABI_VERSION = 1 */
    const char* operator_name;        /* "ferrum_hip_decode" */
    const char* operator_abi_version; /* "1" */
} FerrumNativeOperatorDescriptor;
int ferrum_native_op_init(void);
const FerrumNativeOperatorDescriptor* ferrum_native_op_descriptor(void);

/* ── context / memory: Backend::{new_context, sync, alloc, from_slice, to_vec, zero_buffer,
 *    copy_slice} (ferrum-kernels/src/backend/traits.rs:69,90,159,170-182,798,1388-1390) ──────── */
const char* ferrum_hip_last_error(void);
int ferrum_hip_device_count(int* count);
int ferrum_hip_set_device(int ordinal);
int ferrum_hip_stream_create(void** stream);
int ferrum_hip_stream_destroy(void* stream);
int ferrum_hip_stream_synchronize(void* stream);
int ferrum_hip_alloc(void** dev_ptr, size_t bytes);           /* zero-filled, like B::alloc */
int ferrum_hip_free(void* dev_ptr);
int ferrum_hip_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);
int ferrum_hip_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);
int ferrum_hip_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes, void* stream);
int ferrum_hip_memset_zero(void* dev_ptr, size_t bytes, void* stream);

/* Scratch the split-K GEMMs and the split-KV attention use.  One per stream; sized once. */
typedef struct FerrumHipWorkspace FerrumHipWorkspace;
int ferrum_hip_workspace_create(FerrumHipWorkspace** ws, size_t bytes);
int ferrum_hip_workspace_destroy(FerrumHipWorkspace* ws);

/* ── norms, embedding, elementwise ─────────────────────────────────────────────────────────────
 * Backend::rms_norm (traits.rs:202), fused_add_rms_norm (:212), embedding_lookup (:809),
 * fused_silu_mul_split (:863), fused_gelu_tanh_mul_split (:875), scale_inplace (:889),
 * add_inplace (:1308), add_bias (:1359).  CPU forms: backend/cpu.rs:495-538,1632-1704,2042. */
int ferrum_hip_rms_norm_f16(const void* x, const void* w, float eps, void* out, int tokens, int dim, void* stream);
int ferrum_hip_fused_add_rms_norm_f16(void* residual, const void* x, const void* w, float eps, void* out,
                                      int tokens, int dim, void* stream);
int ferrum_hip_embedding_lookup_f16(const void* table, const uint32_t* ids_dev, void* out, int n_ids, int dim,
                                    void* stream);
int ferrum_hip_fused_silu_mul_split_f16(const void* gate_up, void* out, int tokens, int intermediate, void* stream);
int ferrum_hip_fused_gelu_tanh_mul_split_f16(const void* gate_up, void* out, int tokens, int intermediate,
                                             void* stream);
int ferrum_hip_scale_inplace_f16(void* buf, float scale, size_t len, void* stream);
int ferrum_hip_add_inplace_f16(void* residual, const void* x, size_t len, void* stream);
int ferrum_hip_add_bias_f16(void* data, const void* bias, int rows, int cols, void* stream);
/* Backend::layer_norm / Backend::gelu (traits.rs, required core ops; CPU cpu.rs:2081-2122 — exact-form GELU with the
 * reference's own erf polynomial, cpu.rs:2263-2273).  Not on the LLM decode path; present so that the binding returns
 * "unsupported" for no required method. */
int ferrum_hip_layer_norm_f16(const void* x, const void* gamma, const void* beta, float eps, void* out, int tokens, int dim, void* stream);
int ferrum_hip_gelu_f16(const void* x, void* out, size_t len, void* stream);
/* Backend::Timer (backend/timer.rs:88-109): device events recorded on a context's stream; elapsed_ms waits for `end`. */
int ferrum_hip_event_create(void** event);
int ferrum_hip_event_destroy(void* event);
int ferrum_hip_event_record(void* event, void* stream);
int ferrum_hip_event_elapsed_ms(void* start, void* end, float* ms);

/* ── dense fp16 GEMM: Backend::gemm (traits.rs:190; cpu.rs:438-493): out[m,n] = a[m,k]·b[n,k]ᵀ.
 *    Used for the MoE router and the unquantised lm_head.  `_f32out` keeps fp32 logits. ─────────── */
int ferrum_hip_gemm_f16(const void* a, const void* b, void* out, int m, int n, int k, FerrumHipWorkspace* ws,
                        void* stream);
int ferrum_hip_gemm_f16_f32out(const void* a, const void* b, float* out, int m, int n, int k,
                               FerrumHipWorkspace* ws, void* stream);

/* Dense fp16 weights streamed once per step (lm_head, MoE router) can be re-laid into MFMA-fragment-major
 * tiles ("f16t": [ceil(N/16)][K/32][64 lanes][8 halves]) so every wave load is one contiguous 1-KiB burst —
 * the `DenseLinear` load-time transform of this backend (weights are owned by the backend after load,
 * capabilities.rs:136-193 ownership rule).  K % 32 == 0. */
size_t ferrum_hip_dense_f16t_bytes(int n, int k);
int ferrum_hip_dense_repack_f16t(const void* w_rowmajor_dev, void* out_tiled_dev, int n, int k, void* stream);
int ferrum_hip_gemm_f16t(const void* a, const void* b_tiled, void* out, int m, int n, int k, FerrumHipWorkspace* ws,
                         void* stream);
int ferrum_hip_gemm_f16t_f32out(const void* a, const void* b_tiled, float* out, int m, int n, int k,
                                FerrumHipWorkspace* ws, void* stream);

/* ── GPTQ-INT4 linear: BackendQuantMarlin::load_gptq / load_gptq_stacked (capabilities.rs:136-193;
 *    CPU cpu.rs:2283-2379) and Linear<B>::forward (linear.rs:109-129), MarlinExpertStack::
 *    gemm_phase_vllm (marlin_expert_stack.rs:86).  Weights are handed over ONCE as HOST slices
 *    (qweight [K/8,N] i32, scales [K/g,N] f32, qzeros [K/g,N/8] i32, optional g_idx [K], optional
 *    bias [N] f32); the handle owns the repacked device copy thereafter.  bits must be 4,
 *    K % 128 == 0, group_size % 128 == 0 (else FERRUM_HIP_UNSUPPORTED, like Marlin's K,N % 128). */
typedef struct FerrumHipGptq FerrumHipGptq;
int ferrum_hip_gptq_load(FerrumHipGptq** handle, const int32_t* qweight, const float* scales,
                         const int32_t* qzeros, const int32_t* g_idx, const float* bias, int bits,
                         int group_size, int k, int n);
/* Stacked experts, contiguous per expert.  fuse_gate_up != 0 declares the N axis as [gate(I)|up(I)]
 * and enables the fused silu·mul epilogue of ferrum_hip_moe_gemm_phase (output width N/2).
 * Zero points may be asymmetric (the reference's vLLM-Marlin branch, cuda/quant.rs:795-839);
 * g_idx, when it describes act-order, is ONE [K] array for the whole stack (cuda/quant.rs:862 ff.
 * samples expert 0's): rows are packed in sorted-group order and every phase entry point gathers
 * its input columns to match.  The one-launch gate_up→down pair refuses an act-order DOWN stack. */
int ferrum_hip_gptq_load_stacked(FerrumHipGptq** handle, const int32_t* const* qweights,
                                 const float* const* scales, const int32_t* const* qzeros,
                                 const int32_t* g_idx, int bits, int group_size, int k, int n_per_expert,
                                 int num_experts, int fuse_gate_up);
int ferrum_hip_gptq_free(FerrumHipGptq* handle);
int ferrum_hip_gptq_info(const FerrumHipGptq* handle, int* k, int* n, int* num_experts, int* symmetric);
/* out[m,N] = in[m,K]·Wᵀ (+bias) */
int ferrum_hip_gptq_linear_forward_f16(const FerrumHipGptq* handle, const void* in, void* out, int m,
                                       FerrumHipWorkspace* ws, void* stream);
/* One launch over the align-block routing arrays (moe_block_size 16: one-wave decode kernel; 32 or 64: LDS-tiled prefill
 * kernel; 96 or 128: the tall-tile prefill kernel — K must be a multiple of 256 — whose fp16 weights are (q − zero)·scale
 * rounded once, like the reference's own dequantisation; anything else → FERRUM_HIP_UNSUPPORTED): for every valid sorted id p,
 * out[p] = in[p / top_k]·W[block expert]ᵀ; with fused_silu_mul the stack must have been loaded with
 * fuse_gate_up and out[p] = silu(gate)·up ([T·k, N/2]).  prob_m = number of valid pair ids.  sorted_token_ids and expert_ids hold
 * max_blocks·moe_block_size and max_blocks entries (vLLM's max_num_tokens_padded sizing): a launched block reads its ids before
 * it compares against num_tokens_past_padded. */
int ferrum_hip_moe_gemm_phase_f16(const FerrumHipGptq* stack, const void* input, const int32_t* sorted_token_ids,
                                  const int32_t* expert_ids, const int32_t* num_tokens_past_padded, void* output,
                                  int prob_m, int moe_block_size, int top_k, int max_blocks, int fused_silu_mul,
                                  void* stream);

/* Same GEMM with the align-block-size step computed inside the kernel from the raw router output
 * (expert_ids_per_pair [prob_m] i32; prob_m ≤ 1024, i.e. decode-sized batches): identical blocks and
 * row order as ferrum_hip_moe_align_block_size, without the separate launch. */
int ferrum_hip_moe_gemm_phase_inline_align_f16(const FerrumHipGptq* stack, const void* input,
                                               const int32_t* expert_ids_per_pair, void* output, int prob_m,
                                               int num_experts, int top_k, int max_blocks, int fused_silu_mul,
                                               void* stream);

/* Same GEMM again, expert-major: the grid is (column tile, expert), every wave requests its expert's weights at once and
 * finds the expert's pairs (ascending pair id, the order ferrum_hip_moe_align_block_size produces) meanwhile — no align
 * arrays at all.  Same outputs bit for bit; for decode-sized batches (prob_m ≤ 1024) in which most experts are routed to
 * (an expert without pairs costs one wasted 4-KiB weight request per column tile). */
int ferrum_hip_moe_gemm_phase_expert_major_f16(const FerrumHipGptq* stack, const void* input,
                                               const int32_t* expert_ids_per_pair, void* output, int prob_m,
                                               int num_experts, int top_k, int fused_silu_mul, void* stream);

/* gate_up (+ silu·mul) AND down of a decode batch in ONE expert-major launch (MarlinExpertStack::gemm_phase_vllm twice,
 * marlin_expert_stack.rs:86; layer order qwen3_moe_forward_unified_layer.rs:300-420): the down tiles of an expert wait inside
 * the launch for that expert's gate_up tiles, with their first weights already requested.  gate_up_stack must have been loaded
 * with fuse_gate_up; act_out [prob_m, N_gate_up/2] receives the gated activations (the hand-off buffer), output [prob_m, N_down]
 * the expert outputs — the same bits the two ferrum_hip_moe_gemm_phase_expert_major_f16 calls produce.  Shapes the merged form
 * does not take (mixed symmetric / asymmetric stacks, K_down < 256, widths not multiples of 8) → FERRUM_HIP_UNSUPPORTED. */
int ferrum_hip_moe_gemm_phase_expert_major_pair_f16(FerrumHipGptq* gate_up_stack, const FerrumHipGptq* down_stack, const void* input,
                                                    const int32_t* expert_ids_per_pair, void* act_out, void* output, int prob_m,
                                                    int num_experts, int top_k, void* stream);

/* The same pair of phases for a SMALL batch (prob_m ≤ 64 pairs: decode at c ≤ 8) as one block-major launch: (gate_up tiles +
 * down tiles) × 16-row blocks of the align order, four waves per tile splitting K with the whole tile in flight, the down tiles
 * of a block waiting in the launch for that block's gate_up tiles.  max_blocks ≥ the number of 16-row blocks the routing needs
 * (≤ num_experts).  Outputs: the bits of ferrum_hip_moe_gemm_phase_inline_align_f16 ×2 for prob_m ≤ 16 (the same K split);
 * within fp32 summation order of them beyond (those forms do not split K there). */
int ferrum_hip_moe_gemm_phase_block_major_pair_f16(FerrumHipGptq* gate_up_stack, const FerrumHipGptq* down_stack, const void* input,
                                                   const int32_t* expert_ids_per_pair, void* act_out, void* output, int prob_m,
                                                   int num_experts, int top_k, int max_blocks, void* stream);

/* Number of in-launch waits of the stack's merged launches that gave up so far (0 = every hand-off completed; a non-zero
 * count means the outputs of the affected call are invalid — the waits are bounded so that a missing producer never hangs). */
int ferrum_hip_moe_pair_status(const FerrumHipGptq* gate_up_stack, unsigned* timeouts);

/* Sandwich-norm residual update on an fp32 residual stream (Gemma 3): `rms_norm_activation_add_to_f32` +
 * `rms_norm_f32_to_activation` of the reference's device path (llama_family.rs:3381-3421) in one launch:
 * residual_f32 += rms_norm(branch, w_branch);  norm_out = f16(rms_norm(residual_f32, w_next))  (w_next NULL → skipped). */
int ferrum_hip_sandwich_add_rms_norm_f32(const void* branch_f16, const void* w_branch, float* residual_f32,
                                         const void* w_next, float eps, void* norm_out_f16, int tokens, int dim, void* stream);
/* out_f16[i] = rms_norm(residual_f32[row_idx ? row_idx[i] : i], w) — final norm / first input norm of the fp32 stream */
int ferrum_hip_rms_norm_f32_to_f16(const float* x_f32, const int32_t* row_idx, const void* w, float eps, void* out_f16,
                                   int n_rows, int dim, void* stream);

/* ── contiguous-KV lane of the core `Backend` trait (the non-paged path, kv_layer.rs:370-513): split_qkv (traits.rs:850) →
 *    qk_norm_rope (traits.rs:897; token-major [T,heads,hd] → head-major [heads,T,hd]; mode 0 transpose only, 1 per-head
 *    RMSNorm + half-split RoPE, 2 half-split RoPE, 3 interleaved RoPE; position = pos_offset + token) →
 *    kv_cache_append_head_major (traits.rs:1266; cache [nkv, capacity, hd]) → flash_attention (traits.rs:225 with AttnConfig
 *    {num_heads, num_kv_heads, head_dim, causal, scale, kv_seq_stride, sliding_window}; q/out [nq, q_len, hd]; batch must be 1;
 *    attends keys [max(0, end − window), end) with end = min(pos_offset + i + 1, kv_len) when causal, cpu.rs:2179-2259) →
 *    transpose_head_to_token (traits.rs:1281).  copy_slice (traits.rs:798) and scaled_add_inplace (traits.rs:1318) complete
 *    the set.  Simple bandwidth-shaped kernels — the paged lane below is the fast path. ── */
int ferrum_hip_split_qkv_f16(const void* qkv, void* q, void* k, void* v, int tokens, int q_dim, int kv_dim, void* stream);
int ferrum_hip_qk_norm_rope_f16(const void* input, const void* norm_w, const float* cos_tab, const float* sin_tab, void* output,
                                int tokens, int heads, int head_dim, int pos_offset, float eps, int mode, void* stream);
int ferrum_hip_kv_cache_append_head_major_f16(void* cache_k, void* cache_v, int cache_len, int cache_capacity, const void* new_k,
                                              const void* new_v, int new_tokens, int nkv, int head_dim, void* stream);
int ferrum_hip_transpose_head_to_token_f16(const void* src, void* dst, int tokens, int heads, int dim, void* stream);
int ferrum_hip_transpose_token_to_head_f16(const void* src, void* dst, int tokens, int heads, int dim, void* stream);
int ferrum_hip_copy_slice_f16(const void* src, size_t src_offset, void* dst, size_t dst_offset, size_t len, void* stream);
int ferrum_hip_scaled_add_inplace_f16(void* dst, const void* src, float scale, size_t len, void* stream);
int ferrum_hip_flash_attention_f16(const void* q, const void* k, const void* v, void* out, int batch, int q_len, int kv_len,
                                   int pos_offset, int num_heads, int num_kv_heads, int head_dim, int causal, float scale,
                                   int kv_seq_stride, int sliding_window, void* stream);

/* ── paged KV: BackendPagedKv (traits.rs:1622-1904).  Block tables and block ids are the
 *    reference's (ferrum-models/src/common/paged_pool.rs); the bytes inside a block use the native
 *    MFMA-shaped tile layout (csrc/kv_layout.h).  Pools: [num_blocks][kv_heads][16·head_dim] fp16,
 *    must be zero-initialised (ferrum_hip_alloc does).  block_size must be 16. ──────────────────── */
size_t ferrum_hip_paged_pool_bytes(int num_blocks, int kv_heads, int head_dim);
/* split_qkv_norm_rope_into_paged_cache_varlen (traits.rs:1764): qk_mode 0 copy, 1 QK-norm + half-split
 * RoPE, 2 half-split RoPE, 3 interleaved RoPE.  cos/sin are f32 [max_seq, head_dim/2]
 * (llama_family.rs:5220-5237 values). */
int ferrum_hip_split_qkv_norm_rope_into_paged_cache_varlen_f16(
    const void* qkv, const void* q_norm_w, const void* k_norm_w, const float* cos_tab, const float* sin_tab,
    void* q_out, void* cache_k, void* cache_v, const uint32_t* cu_seqlens_q, const uint32_t* pos_offsets,
    const int32_t* block_tables, int num_seqs, int m_total, int q_heads, int kv_heads, int head_dim, float eps,
    int qk_mode, int block_size, int max_blocks_per_seq, void* stream);
/* paged_varlen_attention (traits.rs:1813): causal GQA over the pool, mixed q_len, optional sliding
 * window.  max_q_len is an extra hint (0 = derive from total_q_tokens). */
int ferrum_hip_paged_varlen_attention_f16(const void* q, const void* k_pool, const void* v_pool, void* out,
                                          const uint32_t* cu_seqlens_q, const uint32_t* pos_offsets,
                                          const int32_t* block_tables, int num_seqs, int total_q_tokens,
                                          int max_kv_len, int num_heads, int num_kv_heads, int head_dim,
                                          int sliding_window, int block_size, int max_num_blocks_per_seq,
                                          int max_q_len, FerrumHipWorkspace* ws, void* stream);
/* paged_batched_decode_attention (traits.rs:1885): one token per sequence; valid_kv_lens[s] is the
 * sequence's kv length INCLUDING the token being decoded. */
int ferrum_hip_paged_batched_decode_attention_f16(const void* q, const void* k_pool, const void* v_pool, void* out,
                                                  const int32_t* block_tables, const uint32_t* valid_kv_lens,
                                                  int num_seqs, int max_kv_len, int num_heads, int num_kv_heads,
                                                  int head_dim, int block_size, int max_num_blocks_per_seq,
                                                  FerrumHipWorkspace* ws, void* stream);
/* BackendPagedKv::paged_decode_attention (traits.rs:1719): q_len == 1 → one token per sequence, q/out [num_seqs, nq, hd];
 * q_len > 1 → ONE sequence in causal prefill, q/out head-major [nq, q_len, hd], context_lens[0] = final kv length. */
int ferrum_hip_paged_decode_attention_f16(const void* q, const void* k_pool, const void* v_pool, void* out,
                                          const int32_t* block_tables, const uint32_t* context_lens, int num_seqs,
                                          int num_heads, int num_kv_heads, int head_dim, int block_size,
                                          int max_num_blocks_per_seq, int q_len, FerrumHipWorkspace* ws, void* stream);
/* Decode step of one layer in ONE launch: split_qkv_norm_rope_into_paged_cache_varlen with one token per sequence
 * at position valid_kv_lens[s] − 1, then paged_batched_decode_attention.  Bit-identical to the two-op chain
 * (same float operations per row); q is never materialised.  GQA group ≤ 14.  sliding_window > 0 restricts every
 * sequence to its last `sliding_window` keys (the varlen op's window rule). */
int ferrum_hip_paged_decode_attention_fused_qkv_f16(const void* qkv, const void* q_norm_w, const void* k_norm_w,
                                                    const float* cos_tab, const float* sin_tab, float eps, int qk_mode,
                                                    void* k_pool, void* v_pool, void* out, const int32_t* block_tables,
                                                    const uint32_t* valid_kv_lens, int num_seqs, int max_kv_len,
                                                    int num_heads, int num_kv_heads, int head_dim, int sliding_window,
                                                    int block_size, int max_num_blocks_per_seq, FerrumHipWorkspace* ws,
                                                    void* stream);
/* Gather one sequence's K/V to token-major [kv_len, kv_heads, head_dim] (ferrum-kv read_kv order,
 * ferrum-kv/src/managers/paged.rs:528-561).  Off the hot path (tests, prefix export). */
int ferrum_hip_paged_kv_read_f16(const void* cache_k, const void* cache_v, const int32_t* block_table, int kv_len,
                                 int kv_heads, int head_dim, int block_size, void* k_out, void* v_out,
                                 void* stream);

/* ── MoE routing: BackendMoeFused (capabilities.rs:305-724) ─────────────────────────────────────
 * route_topk_softmax (:334; CPU ferrum-models/src/moe/router.rs:113-195), moe_align_block_size_pair_ids
 * (:449), moe_combine / weighted_sum_batched (:684,:560). */
int ferrum_hip_moe_route_topk_softmax_f16(const void* logits, int32_t* expert_ids, float* expert_weights,
                                          int tokens, int num_experts, int top_k, int norm_topk_prob, void* stream);
int ferrum_hip_moe_route_topk_softmax_f32(const float* logits, int32_t* expert_ids, float* expert_weights,
                                          int tokens, int num_experts, int top_k, int norm_topk_prob, void* stream);
/* sorted_max = batch_x_topk + num_experts·block_size (ferrum-models/src/moe/dispatch.rs:1865);
 * block_ids holds sorted_max/block_size entries. */
int ferrum_hip_moe_align_block_size(const int32_t* expert_ids_per_pair, int32_t* sorted_token_ids,
                                    int32_t* block_ids, int32_t* total_tokens_post_pad, int batch_x_topk,
                                    int num_experts, int block_size, int sorted_max, void* stream);
/* out[b] = Σ_k weights[b,k]·down[b·top_k+k]; accumulate != 0 adds into out (fused residual add). */
int ferrum_hip_moe_combine_f16(const void* down, const float* weights, void* out, int tokens, int top_k,
                               int hidden, int accumulate, void* stream);
/* BackendMoeFused::moe_align_block_size (capabilities.rs:429; kernels/moe_align_block_size.cu): the variant whose
 * sorted_token_ids hold UNPADDED PACKED ROWS (expert e's region = expert_offsets[e] + 0, 1, …; padding = batch_x_topk). */
int ferrum_hip_moe_align_block_size_packed_rows(const int32_t* expert_ids_per_pair, int32_t* sorted_token_ids, int32_t* block_ids,
                                                int32_t* total_tokens_post_pad, int batch_x_topk, int num_experts, int block_size,
                                                int sorted_max_size, void* stream);
/* BackendMoeFused::moe_build_pairs_by_token (capabilities.rs:410; kernels/moe_build_pairs.cu): stable counting sort of the
 * (token, slot) pairs by expert = MoeBucketPlan::rebuild_into (moe/dispatch.rs:1408-1461), bit for bit: pairs_by_token[p] =
 * packed row of pair p (−1 for ids outside [0, E)), packed_token_idx[row] = p / top_k, expert_offsets[E + 1]. */
int ferrum_hip_moe_build_pairs_by_token(const int32_t* expert_ids, int32_t* pairs_by_token, int32_t* packed_token_idx,
                                        int32_t* expert_offsets, int batch_x_topk, int num_experts, int top_k, void* stream);
/* BackendMoeFused::moe_combine with the trait's own arguments (capabilities.rs:684): rows of the expert-bucketed packed_down
 * are found through pairs_by_token (−1 = skipped). */
int ferrum_hip_moe_combine_pairs_f16(const void* packed_down, const int32_t* pairs_by_token, const float* pair_weights, void* out,
                                     int batch, int hidden, int top_k, int total_pairs, void* stream);
/* BackendMoeFused::weighted_sum_batched / weighted_sum_batched_offset (capabilities.rs:560,580): offsets in elements. */
int ferrum_hip_weighted_sum_batched_f16(const void* slots, const float* weights, size_t weights_offset, void* out,
                                        size_t out_offset, int batch, int top_k, int hidden, void* stream);
/* MarlinExpertStack::gemm_phase_batched (marlin_expert_stack.rs:63): dispatches = num_dispatches × (expert, in_row_offset,
 * out_row_offset, m) int32 on the HOST (like the reference's slice); one grouped launch. */
int ferrum_hip_moe_gemm_phase_batched_f16(FerrumHipGptq* stack, const void* input, const int32_t* dispatches, int num_dispatches,
                                          void* output, int k, int fused_silu_mul, void* stream);

/* Fused forms of op chains the reference issues back to back (qwen3_moe_forward_unified_layer.rs:380-451);
 * same per-element arithmetic as the unfused entry points, one launch each:
 *   fused_add_rms_norm + router gemm + route_topk_softmax  (router_w_tiled: [E,H] in the f16t layout from
 *     ferrum_hip_dense_repack_f16t; logits_out optional fp32 [T,E])
 *   moe_combine + add_inplace (+ the NEXT layer's rms_norm when next_norm_w != NULL) */
int ferrum_hip_fused_add_rms_norm_route_f16(void* residual, const void* x, const void* w, float eps, void* norm_out,
                                            const void* router_w_tiled, int num_experts, int top_k, int norm_topk_prob,
                                            int32_t* expert_ids, float* expert_weights, float* logits_out,
                                            int tokens, int hidden, void* stream);
int ferrum_hip_moe_combine_add_rms_norm_f16(const void* down, const float* weights, void* residual,
                                            const void* next_norm_w, float eps, void* norm_out, int tokens,
                                            int top_k, int hidden, void* stream);

/* Decode-path form of the same chain with one token's router split over Q workgroups (a single CU can pull only
 * ≈70 GB/s from L2): every part re-does the token's add+norm, scores E/Q experts and writes its best ≤8
 * candidates {f32 logit, i32 id} (sorted) to cand[T][Q][8] and (max, Σexp) to stats[T][Q][2].  residual_in and
 * residual_out must differ when Q > 1.  x may be given as S fp32 split-K slabs (x_slabs, stride slab_stride,
 * row stride ld_slab; x_f16 ignored) — reduced in slab order.  num_experts = 0 → add + norm only.
 * ferrum_hip_moe_gemm_phase_merge_route_f16 is the gate_up GEMM that merges the lists (tokens ≤ 64, tokens·Q ≤ 128),
 * derives its align blocks and publishes expert_ids/weights [T·k] and the three align arrays. */
int ferrum_hip_fused_add_rms_norm_route_parts_f16(const void* residual_in, void* residual_out, const void* x_f16,
                                                  const float* x_slabs, int num_slabs, long slab_stride, int ld_slab,
                                                  const void* w, float eps, void* norm_out, const void* router_w_tiled,
                                                  int num_experts, int top_k, int num_parts, void* cand, float* stats,
                                                  float* logits_out, int tokens, int hidden, void* stream);
/* Same kernel, merge done INSIDE the launch: `arrive` is one zeroed uint32 per token (the kernel leaves it zeroed);
 * the last part of a token to arrive merges the lists and writes expert_ids / expert_weights [tokens, top_k]
 * exactly as ferrum_hip_fused_add_rms_norm_route_f16 does.  num_parts ≤ 8. */
int ferrum_hip_fused_add_rms_norm_route_split_f16(const void* residual_in, void* residual_out, const void* x_f16,
                                                  const float* x_slabs, int num_slabs, long slab_stride, int ld_slab,
                                                  const void* w, float eps, void* norm_out, const void* router_w_tiled,
                                                  int num_experts, int top_k, int norm_topk_prob, int num_parts,
                                                  void* cand, float* stats, uint32_t* arrive, int32_t* expert_ids,
                                                  float* expert_weights, float* logits_out, int tokens, int hidden,
                                                  void* stream);
int ferrum_hip_moe_gemm_phase_merge_route_f16(const FerrumHipGptq* stack, const void* input, const void* cand,
                                              const float* stats, void* output, int tokens, int num_parts, int top_k,
                                              int norm_topk_prob, int num_experts, int max_blocks, int fused_silu_mul,
                                              int32_t* expert_ids_out, float* expert_weights_out,
                                              int32_t* sorted_token_ids_out, int32_t* block_ids_out,
                                              int32_t* total_post_pad_out, void* stream);

/* ── device sampling: Backend::argmax_rows_f16[_masked|_sparse_repetition_penalty]
 *    (traits.rs:1534-1591).  First maximum wins.  valid_token_mask may be NULL. ───────────────── */
int ferrum_hip_argmax_rows_f16(const void* logits, uint32_t* out_ids_dev, const uint8_t* valid_token_mask,
                               int mask_len, int m, int n, void* stream);
int ferrum_hip_argmax_rows_f32(const float* logits, uint32_t* out_ids_dev, const uint8_t* valid_token_mask,
                               int mask_len, int m, int n, void* stream);
/* Same result with a workspace (≥ m·1 KiB): vocabulary-wide rows are split over up to 128 workgroups per row and
 * merged by a second launch (a single workgroup per row is bound by one CU's fetch rate).  ws = NULL → one launch. */
int ferrum_hip_argmax_rows_f16_ws(const void* logits, uint32_t* out_ids_dev, const uint8_t* valid_token_mask,
                                  int mask_len, int m, int n, FerrumHipWorkspace* ws, void* stream);
int ferrum_hip_argmax_rows_f32_ws(const float* logits, uint32_t* out_ids_dev, const uint8_t* valid_token_mask,
                                  int mask_len, int m, int n, FerrumHipWorkspace* ws, void* stream);
int ferrum_hip_apply_repetition_penalties_sparse_f16(void* logits, const uint32_t* row_offsets,
                                                     const uint32_t* token_ids, const float* penalties, int m,
                                                     int n, void* stream);
int ferrum_hip_apply_repetition_penalties_sparse_f32(float* logits, const uint32_t* row_offsets,
                                                     const uint32_t* token_ids, const float* penalties, int m,
                                                     int n, void* stream);

/* ── host-side sampling chain (callers that take FullLogits): the C++ mirror of the reference's logits processors and
 *    samplers (ferrum-interfaces/src/sampler.rs:186-467), operating on host f32 logits in place.  Order of
 *    ferrum_hip_sampler_sample = the reference's priorities: repetition penalty (High), top-k, top-p (Normal), temperature
 *    (Low — last), then GreedySampler (LAST maximum, `max_by`; the device argmax keeps the FIRST, traits.rs:1547) or
 *    MultinomialSampler (threshold = random_u32 / u32::MAX in f32, first index whose running f32 sum reaches it). ── */
typedef struct {
    float temperature;            /* ≤ 0 or 1 → untouched */
    int32_t top_k;                /* ≤ 0 → off */
    float top_p;                  /* outside (0,1) → off */
    float repetition_penalty;     /* 1 → off */
    const uint32_t* previous_tokens;
    int32_t num_previous_tokens;
    int32_t greedy;               /* 1 → GreedySampler, 0 → MultinomialSampler */
    uint32_t random_u32;          /* RngCore::next_u32() of the caller's generator */
    uint32_t _pad;
} FerrumHipSamplingParams;
int ferrum_hip_sampler_apply_temperature(float* logits, int n, float temperature);
int ferrum_hip_sampler_apply_top_k(float* logits, int n, int k);
int ferrum_hip_sampler_apply_top_p(float* logits, int n, float p);
int ferrum_hip_sampler_apply_repetition_penalty(float* logits, int n, const uint32_t* previous_tokens, int num_previous,
                                                float penalty);
int ferrum_hip_sampler_greedy(const float* logits, int n, uint32_t* token);
int ferrum_hip_sampler_multinomial(const float* logits, int n, uint32_t random_u32, uint32_t* token);
int ferrum_hip_sampler_sample(float* logits, int n, const FerrumHipSamplingParams* params, uint32_t* token);

/* ── host-side KV block bookkeeping: BlockAllocator (ferrum-models/src/common/paged_pool.rs:106-365).
 *    Pure host code; reproduces block ids bit-exactly (ids from 0, LIFO, prefer-unhashed). ─────── */
typedef struct FerrumHipBlockAllocator FerrumHipBlockAllocator;
int ferrum_hip_block_allocator_create(FerrumHipBlockAllocator** a, uint32_t num_blocks);
int ferrum_hip_block_allocator_destroy(FerrumHipBlockAllocator* a);
int ferrum_hip_block_allocator_allocate(FerrumHipBlockAllocator* a, uint32_t* block);       /* INVALID when exhausted */
int ferrum_hip_block_allocator_allocate_n(FerrumHipBlockAllocator* a, uint32_t n, uint32_t* blocks); /* atomic */
int ferrum_hip_block_allocator_free(FerrumHipBlockAllocator* a, const uint32_t* blocks, uint32_t n);
int ferrum_hip_block_allocator_acquire(FerrumHipBlockAllocator* a, uint32_t block);
int ferrum_hip_block_allocator_register_hash(FerrumHipBlockAllocator* a, uint32_t block, uint64_t hash);
int ferrum_hip_block_allocator_try_acquire_by_hash(FerrumHipBlockAllocator* a, uint64_t hash, int64_t* block); /* -1 miss */
uint32_t ferrum_hip_block_allocator_free_count(const FerrumHipBlockAllocator* a);
uint32_t ferrum_hip_block_allocator_ref_count(const FerrumHipBlockAllocator* a, uint32_t block);
uint32_t ferrum_hip_block_allocator_peak_in_use(const FerrumHipBlockAllocator* a);
uint32_t ferrum_hip_block_allocator_hash_table_size(const FerrumHipBlockAllocator* a);

/* ── decoder runner: the C++ mirror of DecoderOnlyLLM / LlamaFamilyModel / Qwen3MoeModel
 *    unified forward (ferrum-models/src/common/llm.rs:45-293,
 *    models/llama_family_forward_batched.rs:1676-2420, models/qwen3_moe_forward_unified.rs:174-445)
 *    behind ModelExecutor::{reserve_kv_slots, unified_decode, release}
 *    (ferrum-interfaces/src/model_executor.rs:456-651). ────────────────────────────────────────── */
typedef struct FerrumHipModel FerrumHipModel;

typedef struct {
    int32_t num_layers, hidden, num_heads, num_kv_heads, head_dim, intermediate, vocab;
    int32_t max_seq_len;          /* RoPE table length / per-sequence KV capacity */
    int32_t has_qk_norm;          /* qk_mode 1 else 2 (llama_family.rs llama_qk_mode) */
    int32_t activation;           /* 0 silu, 1 gelu_tanh */
    int32_t num_experts;          /* 0 = dense MLP */
    int32_t top_k, expert_inter, norm_topk_prob;
    int32_t rope_scaling_kind;    /* 0 none, 1 linear, 2 llama3 */
    int32_t sliding_window;       /* 0 = full attention */
    int32_t group_size;           /* GPTQ group size (128) */
    int32_t kv_num_blocks;        /* physical KV blocks in the pool (block size 16) */
    int32_t max_seqs;             /* max sequences per unified batch */
    int32_t max_tokens;           /* max query tokens per unified batch */
    float rms_eps;
    float _pad;
    double rope_theta;
    double rope_p0, rope_p1, rope_p2, rope_p3;
    int32_t tp_rank, tp_world;    /* tensor-parallel shard of this process (world 1 = none) */
    /* Gemma-3 layer semantics (llama_family.rs:520-552).  The load-time folds stay with the loader: RMSNorm `x̂·(1+w)`
     * (+1 on every norm weight) and 1/√query_pre_attn_scalar (× √(head_dim/scalar) on q_norm), llama_family.rs:891-967. */
    int32_t sliding_window_pattern; /* N > 0: layer (idx+1) % N == 0 is global (full attention, main RoPE table); the others
                                       are local (sliding_window, local RoPE table).  0 = every layer uses sliding_window */
    int32_t sandwich_norms;       /* post_attn / post_ffn norms applied to the branch BEFORE its residual add; the residual
                                     stream is then kept in fp32 on the device (the reference's F32 residual shadow) */
    float embed_scale;            /* 0 = none; else embedding rows × this (Gemma: bf16-rounded √hidden) */
    float _pad2;
    double rope_local_theta;      /* 0 = one table; else θ of the unscaled table the local layers use */
    /* Beyond the reference (its MoE config is not sharded, README.md:242, is_ep = 0), with tp_world > 1:
     * expert_parallel 1 = experts sharded over the ranks (rank r owns experts [r·E/N, (r+1)·E/N); router replicated; each rank
     * runs the grouped GEMMs of its own experts and the partial MoE outputs meet in the same [T, H] all-reduce a dense MLP
     * uses) on top of tensor-parallel attention (num_heads / num_kv_heads are per-rank values, all-reduce after o_proj);
     * 2 = experts sharded, attention replicated (full head counts on every rank, no all-reduce after o_proj — for world
     * sizes the kv heads do not divide).  vocab_parallel 1 = lm_head rows sharded; per-rank argmax pairs are gathered. */
    int32_t expert_parallel;
    int32_t vocab_parallel;
} FerrumHipModelConfig;

/* One item of a UnifiedBatch (model_executor.rs:354-386). */
typedef struct {
    uint64_t seq_id;              /* cache id */
    const uint32_t* q_tokens;     /* host pointer */
    int32_t num_q_tokens;
    int32_t pos_offset;           /* kv position of the first q token */
    int32_t is_final_chunk;       /* produce logits / a token for the last q token */
    int32_t _pad;
} FerrumHipBatchItem;

/* KvSlotRequest / KvSlotReservation (model_executor.rs:24-64). */
typedef struct {
    uint64_t seq_id;
    int32_t target_len;
    int32_t _pad;
} FerrumHipKvSlotRequest;
typedef struct {
    int32_t block_size, total_blocks, free_blocks_before, free_blocks_after;
} FerrumHipKvSlotReservation;

int ferrum_hip_model_create(FerrumHipModel** model, const FerrumHipModelConfig* cfg);
int ferrum_hip_model_destroy(FerrumHipModel* model);
/* Weight hand-over (host slices, copied/repacked to device).  which:
 *   global dense: 0 embed [V,H], 1 lm_head [V,H] (omit → tied to embed), 2 final_norm [H]
 *   layer dense : 0 input_ln [H], 1 post_ln [H] (sandwich: pre_feedforward_layernorm), 2 q_norm [hd], 3 k_norm [hd],
 *                 4 router [E,H], 5 post_attn_ln [H], 6 post_ffn_ln [H] (sandwich norms),
 *                 7 qkv_bias [nq·hd + 2·nkv·hd] (fused q|k|v projection bias, Qwen2 family)
 *   gptq        : 0 qkv, 1 o, 2 gate_up, 3 down, 4 expert gate_up, 5 expert down               */
int ferrum_hip_model_set_global_f32(FerrumHipModel* model, int which, const float* data);
int ferrum_hip_model_set_layer_dense_f32(FerrumHipModel* model, int layer, int which, const float* data);
int ferrum_hip_model_set_gptq(FerrumHipModel* model, int layer, int which, int expert, const int32_t* qweight,
                              const float* scales, const int32_t* qzeros, const int32_t* g_idx, int k, int n);
/* Unquantised projection (DenseLinear next to GptqLinear, ferrum-kernels/src/linear.rs:109-129): weight [n, k] row-major
 * f32 on the host, kept as fp16 on the device; which = 0 qkv, 1 o, 2 gate_up, 3 down (dense-MLP models). */
int ferrum_hip_model_set_dense_f32(FerrumHipModel* model, int layer, int which, const float* weight, int k, int n);
/* Deterministic synthetic weights generated on the device (bench: no checkpoints offline). */
int ferrum_hip_model_init_synthetic(FerrumHipModel* model, uint64_t seed);
int ferrum_hip_model_finalize(FerrumHipModel* model);

/* ── block-level prefix cache (paged_pool.rs:60-98, models/qwen3_moe/prefix_cache.rs:66-235, prefill_decode.rs:10-70).
 *    block_hash = SipHash-1-3, zero key (Rust `DefaultHasher::new()`), over parent u64 ‖ token u32s, little endian; one hash
 *    per full 16-token block, chained.  acquire: after reserve_kv_slots, before the first forward of a sequence — splices
 *    the longest cached block prefix into the sequence (fresh blocks in those slots are freed), sets its kv length and
 *    returns the number of prompt tokens that need no prefill; a full hit is rolled back by one block so the forward still
 *    yields logits.  The caller then forwards tokens[cached:] at pos_offset = cached and calls register with the FULL prompt. */
uint64_t ferrum_hip_siphash(int c_rounds, int d_rounds, uint64_t k0, uint64_t k1, const uint8_t* data, size_t len);
int ferrum_hip_block_hash_chain(const uint32_t* tokens, int n, int block_size, uint64_t* out, int capacity, int* count);
int ferrum_hip_model_prefix_cache_acquire(FerrumHipModel* model, uint64_t seq_id, const uint32_t* tokens, int n,
                                          int* cached_tokens);
int ferrum_hip_model_prefix_cache_register(FerrumHipModel* model, uint64_t seq_id, const uint32_t* all_tokens, int n,
                                           int prior_cached_tokens);
int ferrum_hip_model_prefix_cache_stats(const FerrumHipModel* model, uint64_t* hits, uint64_t* misses,
                                        uint64_t* saved_prefill_tokens, uint64_t* entries);

/* ── checkpoint reader (host only): safetensors shards + HF config.json + GPTQ quantize config, with the reference
 *    loader's semantics — shard index (ferrum-quantization/src/native_safetensors.rs:142-195), f32/f16/bf16 → f32 and raw
 *    i32 reads (:197-330), fused GPTQ linears q|k|v → qkv and gate|up → gate_up (:887-1000) with symmetric-4-bit qzeros
 *    canonicalised to 0x77777777 (:1242-1246) and g_idx validation (:1288-1324), quantize_config.json or config.json
 *    "quantization_config" (:1475-1530), config.json mapping (ferrum-models/src/definition.rs:225-375,
 *    models/llama_family.rs:596-680,733-810, moe_config.rs:91-130), tensor names (llama_family.rs:900-945,
 *    qwen3_moe/load.rs:178-260).  Architectures: Llama, Mistral, Qwen2 (fused q|k|v bias), Qwen3, Qwen3-MoE, Gemma-3 (norm folds of
 *    llama_family.rs:891-967 applied at load); others → FERRUM_HIP_UNSUPPORTED. ── */
typedef struct FerrumHipCheckpoint FerrumHipCheckpoint;
int ferrum_hip_checkpoint_open(FerrumHipCheckpoint** ck, const char* model_dir);
int ferrum_hip_checkpoint_close(FerrumHipCheckpoint* ck);
int ferrum_hip_checkpoint_num_tensors(const FerrumHipCheckpoint* ck);
/* dtype: 0 F32, 1 F16, 2 BF16, 3 I32, 4 I64, 5 other; shape4 padded with 1 */
int ferrum_hip_checkpoint_tensor_info(const FerrumHipCheckpoint* ck, const char* name, int* dtype, int* ndim, int64_t* shape4);
int ferrum_hip_checkpoint_read_f32(const FerrumHipCheckpoint* ck, const char* name, float* out, size_t capacity);
int ferrum_hip_checkpoint_read_i32(const FerrumHipCheckpoint* ck, const char* name, int32_t* out, size_t capacity);
int ferrum_hip_checkpoint_quant_config(const FerrumHipCheckpoint* ck, int* is_gptq, int* bits, int* group_size, int* desc_act,
                                       int* sym);
/* Fused GPTQ read of tensor-name stems (e.g. "model.layers.0.self_attn.q_proj", "…k_proj", "…v_proj").  Call with null
 * buffers for k / n / has_g_idx, then with qweight [k/8·n], scales [k/group·n], qzeros [k/group·n/8], g_idx [k]. */
int ferrum_hip_checkpoint_read_gptq_fused(const FerrumHipCheckpoint* ck, const char* const* parts, int num_parts,
                                          int32_t* qweight, float* scales, int32_t* qzeros, int32_t* g_idx, int* k, int* n,
                                          int* has_g_idx);
/* config.json → architecture fields of the runner config (kv_num_blocks / max_seqs / max_tokens stay 0 for the caller);
 * max_seq_len = min(max_position_embeddings, cap) (cap 0 = none); tied_lm_head = no lm_head.weight tensor. */
int ferrum_hip_checkpoint_model_config(const FerrumHipCheckpoint* ck, int max_seq_len_cap, FerrumHipModelConfig* cfg,
                                       char* arch_out, size_t arch_cap, int* tied_lm_head);
/* Hand every weight to a created, not yet finalized model (dimensions from ferrum_hip_checkpoint_model_config). */
int ferrum_hip_model_load_checkpoint(FerrumHipModel* model, const FerrumHipCheckpoint* ck);

/* ModelExecutor::reserve_kv_slots (model_executor.rs:484): allocate physical blocks so every
 * requested sequence can hold target_len tokens; atomic — on exhaustion nothing is taken and
 * FERRUM_HIP_INVALID is returned. */
int ferrum_hip_model_reserve_kv_slots(FerrumHipModel* model, const FerrumHipKvSlotRequest* reqs, int n,
                                      FerrumHipKvSlotReservation* out);
int ferrum_hip_model_kv_capacity_snapshot(const FerrumHipModel* model, FerrumHipKvSlotReservation* out);
int ferrum_hip_model_release(FerrumHipModel* model, uint64_t seq_id);
/* Block table of a sequence (u32 physical ids), for "bit-exact KV-block indexing" checks. */
int ferrum_hip_model_block_table(const FerrumHipModel* model, uint64_t seq_id, uint32_t* blocks, int capacity,
                                 int* num_blocks, int* kv_len);
/* Read a sequence's K or V of one layer back as fp32 [kv_len, kv_heads, head_dim]. */
int ferrum_hip_model_read_kv_f32(FerrumHipModel* model, uint64_t seq_id, int layer, int is_v, float* out_host);

/* ModelExecutor::unified_decode (model_executor.rs; llm.rs:260): mixed prefill+decode forward.
 * For every item with is_final_chunk, in item order:
 *   - greedy != 0 : out_tokens[j] = device argmax (LogitsReturnPolicy::GreedyArgmax, model_executor.rs:109)
 *   - logits_out != NULL : logits_out[j·vocab ..] = fp32 logits (LogitsReturnPolicy::FullLogits)
 * Blocks are reserved on demand (as unified_forward_internal does via ensure_paged_kv_capacity);
 * returns FERRUM_HIP_INVALID if the pool is exhausted. */
int ferrum_hip_model_unified_forward(FerrumHipModel* model, const FerrumHipBatchItem* items, int num_items,
                                     int greedy, uint32_t* out_tokens, float* logits_out);
/* LogitsReturnPolicy::GreedyArgmax { token_mask, repetition_penalty } (model_executor.rs:109-150) for the whole batch, host
 * pointers: valid_token_mask[mask_len] (0 = forbidden; ids ≥ mask_len forbidden; NULL = none) is shared by every sampled row
 * like the reference's single device mask (llama_family_forward_batched.rs:2392-2410); the sparse repetition penalty follows
 * argmax_rows_f16_sparse_repetition_penalty (traits.rs:1571-1591): sampled row r owns penalty_token_ids[penalty_row_offsets[r]
 * .. penalty_row_offsets[r+1]) (de-duplicated by the caller) and penalties[r]; logit v → v / p if v > 0 else v · p.
 * NULL arrays = no penalty.  Logits returned through logits_out are the penalised ones. */
typedef struct {
    const uint8_t* valid_token_mask;
    int32_t mask_len;
    int32_t _pad;
    const uint32_t* penalty_row_offsets;   /* [num_sampled + 1] */
    const uint32_t* penalty_token_ids;
    const float* penalties;                /* [num_sampled] */
} FerrumHipGreedyOptions;
/* With cfg.vocab_parallel, logits_out receives THIS RANK's vocabulary slice, [num_sampled, ferrum_hip_model_local_vocab()] packed
 * (rows vocab_start … vocab_start + count of the full matrix); out_tokens are global ids, identical on every rank. */
int ferrum_hip_model_local_vocab(const FerrumHipModel* model, int* vocab_start, int* vocab_count);
int ferrum_hip_model_unified_forward_ex(FerrumHipModel* model, const FerrumHipBatchItem* items, int num_items, int greedy,
                                        const FerrumHipGreedyOptions* opts, uint32_t* out_tokens, float* logits_out);
/* Steady-state decode: every listed sequence advances by one token (its last sampled token),
 * `steps` times, with device-side greedy sampling and hipGraph replay when available.
 * out_tokens [steps, n] (may be NULL). */
int ferrum_hip_model_decode_steps(FerrumHipModel* model, const uint64_t* seq_ids, const uint32_t* first_tokens,
                                  int n, int steps, uint32_t* out_tokens);
/* Tap for parity tests: hidden state after each layer of the LAST forward, fp32 [layers, T, H]. */
int ferrum_hip_model_enable_taps(FerrumHipModel* model, int enable);
int ferrum_hip_model_read_taps(FerrumHipModel* model, float* out_host, int max_tokens);
int ferrum_hip_model_stream(FerrumHipModel* model, void** stream);
/* Bench instrumentation: mean device time (µs, HIP events on the model stream) of one launch of a hot
 * kernel, cycled over every layer's weights with the index/routing state of the last forward.
 * which: 0 MoE gate_up(+silu·mul), 1 MoE down, 2 paged decode attention, 3 qkv GEMM, 4 o GEMM, 5 lm_head. */
int ferrum_hip_model_time_kernel(FerrumHipModel* model, int which, int n_seqs, int max_kv_len, int reps,
                                 float* avg_us, int* moe_blocks);
/* ── Tensor-parallel communicator: BackendCollective (ferrum-kernels/src/backend/capabilities.rs:84-109; the CUDA lane's
 * NcclRank, nccl_comm.rs:21-49).  One communicator per rank; `ferrum_hip_all_reduce_f16` is the in-place fp16 sum the
 * decode runner issues after o_proj and down_proj (cuda/tp_decode.rs:350-372).  Transports: RCCL over xGMI (any size), and a
 * hand-written one-shot peer reduce for decode-sized messages (rank-ordered fp32 sum, identical bits on every rank) over
 * buffers of the same process (`create_local_group`) or hipIpc-imported ones (`oneshot_export` / `oneshot_attach`).
 * Both are stream-ordered device work, so a decode step with its all-reduces is captured in one hipGraph. ── */
typedef struct FerrumHipComm FerrumHipComm;
/* 128-byte RCCL unique id created on rank 0 and broadcast by the host. */
int ferrum_hip_comm_unique_id(uint8_t id[128]);
int ferrum_hip_tp_unique_id(uint8_t id[128]);                       /* same, older name */
int ferrum_hip_comm_create_rccl(FerrumHipComm** comm, int world, int rank, const uint8_t id[128]);
/* A rank with no transport yet (one-shot buffers attached afterwards); all_reduce fails until they are. */
int ferrum_hip_comm_create_bare(FerrumHipComm** comm, int world, int rank);
/* `world` ranks inside one process (threads of a test on one GPU; devices[] = NULL puts every buffer on the current device). */
int ferrum_hip_comm_create_local_group(FerrumHipComm** comms, int world, size_t max_message_bytes, const int* devices);
int ferrum_hip_comm_oneshot_export(FerrumHipComm* comm, size_t max_message_bytes, uint8_t handle[64]);
int ferrum_hip_comm_oneshot_attach(FerrumHipComm* comm, const uint8_t* handles /* world x 64 bytes, rank order */, int world);
int ferrum_hip_comm_oneshot_status(FerrumHipComm* comm, unsigned* epoch, unsigned* timeouts);
int ferrum_hip_comm_destroy(FerrumHipComm* comm);
int ferrum_hip_comm_world_size(const FerrumHipComm* comm);          /* BackendCollective::world_size (NULL comm: 1) */
int ferrum_hip_comm_rank(const FerrumHipComm* comm);                /* BackendCollective::rank */
int ferrum_hip_all_reduce_f16(FerrumHipComm* comm, void* buf, size_t count, void* stream);   /* ReduceOp::Sum, in place */
/* The all-reduce folded into its consumer (cuda/tp_decode.rs:350-372 runs all_reduce, then the layer's residual add + norm):
 * residual[rows, dim] += Σ_ranks x; norm_out = rms_norm(residual)·w, ONE launch where the one-shot transport carries the
 * message (rows ≤ 64, dim ≤ 8192) — bit for bit ferrum_hip_all_reduce_f16 followed by ferrum_hip_fused_add_rms_norm_f16.
 * *fused = 0: nothing was done (RCCL transport, message too large, …) and the caller runs the two calls. */
int ferrum_hip_all_reduce_add_rms_norm_f16(FerrumHipComm* comm, const void* x, void* residual, const void* w, float eps,
                                           void* norm_out, int rows, int dim, int* fused, void* stream);
int ferrum_hip_all_gather_f16(FerrumHipComm* comm, const void* local, void* global, size_t local_count, void* stream);
int ferrum_hip_broadcast_f16(FerrumHipComm* comm, void* buf, size_t count, int src_rank, void* stream);
/* Runner side: an RCCL rank owned by the model, or a communicator of the caller's. */
int ferrum_hip_model_tp_init(FerrumHipModel* model, const uint8_t id[128]);
int ferrum_hip_model_set_comm(FerrumHipModel* model, FerrumHipComm* comm);
/* In-process stand-in for the communicator with HOST barriers (tests, one GPU, eager launches only): the ranks of a
 * tensor-parallel group are runner models driven by threads of one process; each all-reduce is two host barriers around a
 * device-side rank-ordered sum. */
typedef struct FerrumHipTpLoopback FerrumHipTpLoopback;
int ferrum_hip_tp_loopback_create(FerrumHipTpLoopback** lb, int world);
int ferrum_hip_tp_loopback_destroy(FerrumHipTpLoopback* lb);
int ferrum_hip_model_tp_attach_loopback(FerrumHipModel* model, FerrumHipTpLoopback* lb);
/* 1-rank RCCL round trip on the current device (fp16 sum all-reduce, in place), eagerly and from a captured + replayed
 * hipGraph: checks the dlopen'ed entry points, enum values, by-value ncclUniqueId passing and stream capture of the
 * tensor-parallel path without needing a second GPU. */
int ferrum_hip_tp_selftest(int count);

/* ── BackendGraph (capabilities.rs:35-70): stream capture / replay of whatever the caller enqueues between begin and end.
 * No entry point allocates inside a capture window as long as workspaces and handles were created before it. ── */
typedef struct FerrumHipGraph FerrumHipGraph;
int ferrum_hip_graph_begin_capture(void* stream);
int ferrum_hip_graph_end_capture(void* stream, FerrumHipGraph** graph);
int ferrum_hip_graph_replay(FerrumHipGraph* graph, void* stream);
int ferrum_hip_graph_destroy(FerrumHipGraph* graph);

/* ── Debug / test support: which kernel forms the launchers chose, and re-reading the FERRUM_HIP_* development knobs
 * (they are read once at library load; no launch path calls getenv). ── */
int ferrum_hip_debug_form_count(void);
const char* ferrum_hip_debug_form_name(int form);
int ferrum_hip_debug_form_hits(uint64_t* hits, int capacity);
int ferrum_hip_debug_form_reset(void);
int ferrum_hip_debug_reload_knobs(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FERRUM_HIP_H */
