// Internal C++ interface between the kernel translation units and the C-ABI / runner.
#pragma once
#include <algorithm>
#include <vector>

#include "common.h"

namespace fh {

// one router candidate (fused.hip part kernel → w4_gemm.hip merge)
struct RouteCand { float logit; int id; };

// ── GPTQ-INT4 (w4_gemm.hip) ──────────────────────────────────────────────────
struct W4HostPacked {
    int k = 0, n = 0, n64 = 0, G = 0;
    bool symmetric = true;
    std::vector<uint32_t> qw;
    std::vector<uint16_t> sc, zp;
    std::vector<int32_t> perm;   // act-order input gather (empty = identity)
};

struct W4Device {
    int k = 0, n = 0, n64 = 0, G = 0;
    int num_experts = 1;         // stacked experts share one allocation
    uint32_t* qw = nullptr;
    __half* sc = nullptr;
    __half* zp = nullptr;        // null when symmetric
    int32_t* perm = nullptr;     // device act-order permutation or null: the GEMM's input is x'[j] = x[perm[j]]
    int32_t* inv_perm = nullptr; // its inverse (x'[inv_perm[c]] = x[c]) for producers that scatter (decode attention → o_proj)
    bool perm_folded = false;    // the producer ALWAYS writes the permuted row (down: gate_up's columns were packed in perm order)
    __half* bias = nullptr;      // optional [n]
    bool fused_gate_up = false;  // columns permuted for the fused silu·mul epilogue
    __half* f16t = nullptr;      // UNQUANTISED projection (DenseLinear, linear.rs:109-129): fp16 weights [n, k] in f16t tiles;
                                 // qw / sc are then null and w4_gemm_dense routes to the fp16 GEMM
};

int w4_repack_host(const int32_t* qweight, const float* scales, const int32_t* qzeros,
                   const int32_t* g_idx, const int32_t* col_perm, int group_size, int k, int n,
                   W4HostPacked* out);
// Optional prologue of the ≤ 4-row dense projection: the input rows are the MoE combine + residual add + RMSNorm of the
// previous layer's tail (fused.hip kernel A) computed inside the GEMM launch; `x` is then ignored.
struct FusedCombineNorm {
    const __half* down; const float* weights; const __half* residual_in; __half* residual_out; const __half* ln_w;
    float eps; int top_k;
};
bool w4_gemm_dense_can_fuse_combine_norm(const W4Device& w, int m);
int w4_gemm_dense(const W4Device& w, const __half* x, __half* out, int m, float* workspace,
                  size_t workspace_bytes, hipStream_t stream, const FusedCombineNorm* fa = nullptr);
int w4_gemm_moe_tile(const W4Device& w, const __half* x, __half* out, const int32_t* sorted_token_ids, const int32_t* block_ids,
                     const int32_t* total_post_pad, int num_valid_pairs, int max_blocks, int block_rows, int top_k, int fused_silu,
                     hipStream_t stream);
int w4_gemm_dense_slabs_tile(const W4Device& w, const __half* x, float* slabs, size_t slab_bytes, int m, int* S_out,
                             int* rows_pad_out, int* n_pad_out, hipStream_t stream);
int w4_gemm_dense_lds_splits(const W4Device& w, int m);
int w4_gemm_dense_slabs_lds(const W4Device& w, const __half* x, float* slabs, size_t slab_bytes, int m, int* S_inout,
                            int* rows_pad_out, int* n_pad_out, hipStream_t stream);

int w4_gemm_moe(const W4Device& w, const __half* x, __half* out, const int32_t* sorted_token_ids,
                const int32_t* block_ids, const int32_t* total_post_pad, int num_valid_pairs,
                int max_blocks, int top_k, int fused_silu, hipStream_t stream);
int w4_gemm_moe_expert_major(const W4Device& w, const __half* x, __half* out, const int32_t* pair_expert_ids, int num_experts,
                             int num_valid_pairs, int top_k, int fused_silu, hipStream_t stream);
// words between the per-expert arrival counters of the merged launch (one 256-byte line each): `arrive` / `arrive_next` hold
// num_experts · MOE_PAIR_COUNTER_STRIDE words
constexpr int MOE_PAIR_COUNTER_STRIDE = 64;
// routing handed over as the router's Q per-part candidate lists of every token (fused.hip part kernel / chain.hip role B
// with defer_merge) instead of merged pair ids: the grouped GEMM merges them in its prologue and publishes ids + combine weights
struct MoeRouteLists {
    const RouteCand* cand = nullptr; const float* stats = nullptr;
    int T = 0, Q = 0, norm_topk = 0;
    int32_t* pub_expert_ids = nullptr; float* pub_expert_w = nullptr;
};
bool w4_gemm_moe_expert_major_pair_supports(const W4Device& gu, const W4Device& dn, int num_experts, int num_valid_pairs,
                                            const MoeRouteLists* route);
int w4_gemm_moe_expert_major_pair(const W4Device& gu, const W4Device& dn, const __half* x, __half* h, __half* out,
                                  const int32_t* pair_expert_ids, int num_experts, int num_valid_pairs, int top_k,
                                  unsigned* arrive, unsigned* arrive_next, unsigned* timeout, int* took, hipStream_t stream,
                                  const MoeRouteLists* route = nullptr);
bool w4_gemm_moe_block_major_pair_supports(const W4Device& gu, const W4Device& dn, int num_experts, int num_valid_pairs, int max_blocks,
                                           const MoeRouteLists* route);
int w4_gemm_moe_block_major_pair(const W4Device& gu, const W4Device& dn, const __half* x, __half* h, __half* out,
                                 const int32_t* pair_expert_ids, int num_experts, int num_valid_pairs, int max_blocks, int top_k,
                                 unsigned* arrive, unsigned* arrive_next, unsigned* timeout, int* took, hipStream_t stream,
                                 const MoeRouteLists* route = nullptr);
int w4_gemm_moe_inline_align(const W4Device& w, const __half* x, __half* out, const int32_t* pair_expert_ids,
                             int num_experts, int num_valid_pairs, int max_blocks, int top_k, int fused_silu,
                             int32_t* pub_sorted, int32_t* pub_block_ids, int32_t* pub_total, hipStream_t stream,
                             bool few_pairs_per_expert = false);
int w4_gemm_moe_merge_route(const W4Device& w, const __half* x, __half* out, const RouteCand* cand, const float* stats,
                            int tokens, int Q, int top_k, int norm_topk, int num_experts, int max_blocks, int fused_silu,
                            int32_t* pub_expert_ids, float* pub_expert_w, int32_t* pub_sorted, int32_t* pub_block_ids,
                            int32_t* pub_total, hipStream_t stream);
int w4_gemm_dense_slabs(const W4Device& w, const __half* x, float* slabs, size_t slab_bytes, int m, int S,
                        int* rows_pad_out, int* n_pad_out, hipStream_t stream);
int f16_gemm(const __half* x, const __half* w, __half* out, int m, int n, int k, float* workspace,
             size_t workspace_bytes, hipStream_t stream);
int f16_gemm_f32out(const __half* x, const __half* w, float* out, int m, int n, int k, float* workspace,
                    size_t workspace_bytes, hipStream_t stream);

// fp16 weights in MFMA-fragment-major tiles (w4_gemm.hip "f16t"): [N/16][K/32][64 lanes][8]
int f16t_repack(const __half* w_rowmajor, __half* out_tiled, int n, int k, hipStream_t stream);
size_t f16t_elems(int n, int k);
int f16t_gemm(const __half* x, const __half* wt, __half* out, int m, int n, int k, float* workspace,
              size_t workspace_bytes, hipStream_t stream);
int f16t_gemm_f32out(const __half* x, const __half* wt, float* out, int m, int n, int k, float* workspace,
                     size_t workspace_bytes, hipStream_t stream);

// ── norms / elementwise (norm.hip) ───────────────────────────────────────────
int rms_norm_f16(const __half* x, const __half* w, float eps, __half* out, int tokens, int dim, hipStream_t s);
int embed_rms_norm_f16(const __half* table, const uint32_t* token_ids, float embed_scale, __half* residual, float* residual_f32,
                       const __half* w, float eps, __half* norm_out, int tokens, int dim, unsigned* zero_words, int n_zero, hipStream_t s,
                       const int32_t* out_perm = nullptr);   // out_perm: norm_out[row][j] = y[out_perm[j]] (act-order consumer)
int gather_rms_norm_f16(const __half* x, const int32_t* row_idx, const __half* w, float eps, __half* out, int rows, int dim, hipStream_t s);
int fused_add_rms_norm_f16(__half* residual, const __half* x, const __half* w, float eps, __half* out,
                           int tokens, int dim, hipStream_t s, const int32_t* out_perm = nullptr);
int embedding_lookup_f16(const __half* table, const uint32_t* ids, __half* out, int n_ids, int dim, hipStream_t s);
int fused_silu_mul_split_f16(const __half* gate_up, __half* out, int tokens, int im, hipStream_t s);
int fused_gelu_tanh_mul_split_f16(const __half* gate_up, __half* out, int tokens, int im, hipStream_t s);
int add_inplace_f16(__half* residual, const __half* x, long len, hipStream_t s);
int sandwich_add_rms_norm_f32(const __half* branch, const __half* w_branch, float* residual, const __half* w_next, float eps,
                              __half* norm_out, int tokens, int dim, hipStream_t s, const int32_t* out_perm = nullptr);
int sandwich_add_rms_norm_f32_slabs(const float* slabs, int S, long slab_stride, int ld_slab, const __half* w_branch,
                                    float* residual, const __half* w_next, float eps, __half* norm_out, int tokens, int dim,
                                    hipStream_t s, const int32_t* out_perm = nullptr);
int fused_gated_act_slabs_f16(const float* slabs, int S, long slab_stride, int ld, __half* out, int tokens, int im, int gelu,
                              hipStream_t s);
int rms_norm_f32_to_f16(const float* x, const int32_t* row_idx, const __half* w, float eps, __half* out, int n_rows, int dim,
                        hipStream_t s);
// contiguous-KV lane (contig_ops.hip)
int split_qkv_f16(const __half* qkv, __half* q, __half* k, __half* v, int tokens, int q_dim, int kv_dim, hipStream_t s);
int qk_norm_rope_f16(const __half* input, const __half* norm_w, const float* cos_t, const float* sin_t, __half* output,
                     int tokens, int heads, int head_dim, int pos_offset, float eps, int mode, hipStream_t s);
int kv_cache_append_head_major_f16(__half* cache_k, __half* cache_v, int cache_len, int cache_capacity, const __half* new_k,
                                   const __half* new_v, int new_tokens, int nkv, int hd, hipStream_t s);
int transpose_head_to_token_f16(const __half* src, __half* dst, int tokens, int heads, int dim, hipStream_t s);
int transpose_token_to_head_f16(const __half* src, __half* dst, int tokens, int heads, int dim, hipStream_t s);
int copy_slice_f16(const __half* src, long src_offset, __half* dst, long dst_offset, long len, hipStream_t s);
int scaled_add_inplace_f16(__half* dst, const __half* src, float scale, long len, hipStream_t s);
int flash_attention_contig_f16(const __half* q, const __half* k, const __half* v, __half* out, int q_len, int kv_len, int causal,
                               int pos_offset, int num_heads, int num_kv_heads, int head_dim, float scale, int kv_seq_stride,
                               int sliding_window, hipStream_t s);
int scale_inplace_f16(__half* buf, float scale, long len, hipStream_t s);
int add_bias_f16(__half* data, const __half* bias, int rows, int cols, hipStream_t s);
int layer_norm_f16(const __half* x, const __half* gamma, const __half* beta, float eps, __half* out, int tokens, int dim, hipStream_t s);
int gelu_f16(const __half* x, __half* out, long len, hipStream_t s);
int gather_columns_f16(const __half* in, const int32_t* perm, __half* out, int rows, int cols, hipStream_t s);
int gather_rows_f16(const __half* in, const int32_t* row_idx, __half* out, int n_rows, int dim, hipStream_t s);

// ── paged KV (rope_kv.hip, attention.hip) ────────────────────────────────────
// Native pool layout, per (block, kv_head) 4 KiB tiles for head_dim 128 / block 16:
//   K : [hd/8 chunks][16 keys][8 dims]                — MFMA A-fragment shaped
//   V : [hd/32][4 key-quads][16 d][2 d-sub][4 keys]    — MFMA B-fragment shaped
// pool = [num_blocks][kv_heads][tile].
int split_qkv_norm_rope_into_paged_cache_varlen_f16(
    const __half* qkv, const __half* q_norm_w, const __half* k_norm_w, const float* cos_t, const float* sin_t,
    __half* q_out, __half* cache_k, __half* cache_v, const uint32_t* cu_seqlens_q, const uint32_t* pos_offsets,
    const int32_t* block_tables, int num_seqs, int m_total, int q_heads, int kv_heads, int head_dim, float eps,
    int qk_mode, int block_size, int max_blocks_per_seq, hipStream_t s);
int paged_kv_read_f16(const __half* cache_k, const __half* cache_v, const int32_t* block_table, int kv_len,
                      int kv_heads, int head_dim, int block_size, __half* k_out, __half* v_out, hipStream_t s);
int paged_varlen_attention_f16(const __half* q, const __half* k_pool, const __half* v_pool, __half* out,
                               const uint32_t* cu_seqlens_q, const uint32_t* pos_offsets,
                               const int32_t* block_tables, int num_seqs, int total_q_tokens, int max_q_len,
                               int max_kv_len, int num_heads, int num_kv_heads, int head_dim,
                               int sliding_window, int block_size, int max_blocks_per_seq, float* workspace,
                               size_t workspace_bytes, hipStream_t s);
int paged_decode_attention_fused_qkv_f16(const __half* qkv, const __half* q_norm_w, const __half* k_norm_w,
                                         const float* cos_t, const float* sin_t, float eps, int qk_mode, __half* k_pool,
                                         __half* v_pool, __half* out, const int32_t* block_tables,
                                         const uint32_t* valid_kv_lens, int num_seqs, int max_kv_len, int num_heads,
                                         int num_kv_heads, int head_dim, int sliding_window, int block_size,
                                         int max_blocks_per_seq, float* workspace, size_t workspace_bytes, hipStream_t s,
                                         const int32_t* out_scatter = nullptr);   // act-order o_proj: out[row][out_scatter[c]] = y[c]
int paged_batched_decode_attention_f16(const __half* q, const __half* k_pool, const __half* v_pool, __half* out,
                                       const int32_t* block_tables, const uint32_t* valid_kv_lens, int num_seqs,
                                       int max_kv_len, int num_heads, int num_kv_heads, int head_dim, int block_size,
                                       int max_blocks_per_seq, float* workspace, size_t workspace_bytes,
                                       hipStream_t s, const int32_t* out_scatter = nullptr);
size_t paged_attention_workspace_bytes(int total_q_tokens, int num_heads, int head_dim, int max_kv_len);

// ── MoE routing (moe.hip) ────────────────────────────────────────────────────
int moe_route_topk_softmax_f16(const __half* logits, int32_t* expert_ids, float* expert_weights, int tokens,
                               int num_experts, int top_k, int norm_topk_prob, hipStream_t s);
// prefill router in one launch (E = 128): logits on the matrix cores, softmax + top-k in the same workgroup (moe.hip)
bool moe_route_gemm_topk_supports(int num_experts, int hidden, int top_k);
int moe_route_gemm_topk_f16(const __half* x, const __half* router_f16t, int32_t* expert_ids, float* expert_weights, int tokens,
                            int num_experts, int hidden, int top_k, int norm_topk_prob, hipStream_t s);
int moe_route_topk_softmax_f32(const float* logits, int32_t* expert_ids, float* expert_weights, int tokens,
                               int num_experts, int top_k, int norm_topk_prob, hipStream_t s);
int moe_align_block_size(const int32_t* expert_ids, int32_t* sorted_token_ids, int32_t* block_ids,
                         int32_t* total_post_pad, int batch_x_topk, int num_experts, int block_size,
                         int sorted_max, hipStream_t s);
int moe_align_block_size_packed_rows(const int32_t* expert_ids, int32_t* sorted_token_ids, int32_t* block_ids,
                                     int32_t* total_post_pad, int batch_x_topk, int num_experts, int block_size,
                                     int sorted_max, hipStream_t s);
int moe_build_pairs_by_token(const int32_t* expert_ids, int32_t* pairs_by_token, int32_t* packed_token_idx,
                             int32_t* expert_offsets, int batch_x_topk, int num_experts, int top_k, hipStream_t s);
int moe_remap_expert_ids(const int32_t* ids, int32_t* local, int n, int e0, int e_local, hipStream_t s);
int moe_combine_local_f16(const __half* down, const float* weights, const int32_t* pair_ids, __half* out, int tokens, int top_k,
                          int hidden, hipStream_t s);
int moe_combine_pairs_f16(const __half* packed_down, const int32_t* pairs_by_token, const float* pair_weights, __half* out,
                          int batch, int hidden, int top_k, int total_pairs, hipStream_t s);
int moe_combine_f16(const __half* down, const float* weights, __half* out, int tokens, int top_k, int hidden,
                    int accumulate_into_residual, hipStream_t s);

// ── launch-count fusions (fused.hip) ─────────────────────────────────────────
int fused_add_rms_norm_route_f16(__half* residual, const __half* x, const __half* w, float eps, __half* norm_out,
                                 const __half* router_w, int num_experts, int top_k, int norm_topk_prob,
                                 int32_t* expert_ids, float* expert_weights, float* logits_out, int tokens, int H,
                                 hipStream_t s);
int fused_add_rms_norm_route_slabs_f16(__half* residual, const __half* x, const float* x_slabs, int S, long slab_stride,
                                       int ld_slab, const __half* w, float eps, __half* norm_out, const __half* router_w,
                                       int num_experts, int top_k, int norm_topk_prob, int32_t* expert_ids,
                                       float* expert_weights, float* logits_out, int tokens, int H, hipStream_t s,
                                       const int32_t* out_perm = nullptr);
int moe_combine_add_rms_norm_f16(const __half* down, const float* weights, const __half* residual,
                                 __half* residual_out, const __half* next_w, float eps, __half* norm_out, int tokens,
                                 int top_k, int H, hipStream_t s, const int32_t* out_perm = nullptr);
// B split over Q expert parts per token (+ optional fp32 split-K slabs of the o projection); the Q sorted
// candidate lists are merged by the gate_up grouped GEMM (w4_gemm_moe_merge_route).
int fused_add_rms_norm_route_split_f16(const __half* residual_in, __half* residual_out, const __half* x,
                                       const float* x_slabs, int S, long slab_stride, int ld_slab, const __half* w,
                                       float eps, __half* norm_out, const __half* router_w, int num_experts, int top_k,
                                       int Q, RouteCand* cand, float* stats, unsigned* arrive, int norm_topk_prob,
                                       int32_t* expert_ids, float* expert_weights, float* logits_out, int tokens, int H,
                                       hipStream_t s);
int fused_add_rms_norm_route_parts_f16(const __half* residual_in, __half* residual_out, const __half* x,
                                       const float* x_slabs, int S, long slab_stride, int ld_slab, const __half* w,
                                       float eps, __half* norm_out, const __half* router_w, int num_experts, int top_k,
                                       int Q, RouteCand* cand, float* stats, float* logits_out, int tokens, int H,
                                       hipStream_t s);

// ── the attention half of a MoE decode layer as one launch (chain.hip) ───────────────────────────────────────────────
// [combine + add + norm of the previous layer's tail] → q|k|v → QK-norm / RoPE / KV write / attention → o_proj →
// add + norm + router + top-k, as roles of one grid with in-launch hand-offs.  Buffers as the stand-alone kernels use them.
constexpr int CHAIN_MAX_ATTN_WGS = 1024;      // (sequence, kv head, KV split) workgroups whose states the split buffer holds
struct DecodeChainDesc {
    int T = 0, H = 0, nq = 0, nkv = 0, head_dim = 0;
    bool has_a = false;          // the previous layer's tail runs as the first role (else norm1 / res_a come from an earlier launch)
    int top_k = 0;               // role A: expert rows per token
    const __half* down = nullptr; const float* comb_w = nullptr; const __half* res_in = nullptr; const __half* ln_in = nullptr;
    // dense model (E == 0): the tail is residual + Σ_slabs of the MLP's down projection [a_S][rows][a_ld] fp32, then the norm
    const float* a_slabs = nullptr; int a_S = 0; long a_slab_stride = 0; int a_ld = 0;
    const __half* a_x = nullptr;  // … or, without slabs, the MLP output as fp16 rows [T, H]
    float eps = 0.f;
    __half* res_a = nullptr;     // residual after the tail (role A's output, role B's input)
    __half* norm1 = nullptr;     // [T, H] input rows of the q|k|v projection
    const W4Device* qkv = nullptr; __half* qkv_out = nullptr;
    __half* k_pool = nullptr; __half* v_pool = nullptr; const int32_t* block_tables = nullptr; const uint32_t* kv_lens = nullptr;
    const __half* q_norm_w = nullptr; const __half* k_norm_w = nullptr; const float* cos_t = nullptr; const float* sin_t = nullptr;
    int qk_mode = 0, max_blocks = 0, sliding_window = 0;
    __half* attn_out = nullptr;
    const W4Device* o = nullptr; __half* o_out = nullptr;
    __half* res_b_out = nullptr; const __half* post_ln = nullptr; __half* norm2 = nullptr; const __half* router_w = nullptr;
    int E = 0, r_top_k = 0, Q = 0, norm_topk = 0;
    bool defer_merge = false;    // role B stops at the per-part candidate lists (cand / stats); the grouped GEMM that follows merges them
    RouteCand* cand = nullptr; float* stats = nullptr; unsigned* route_arrive = nullptr; int32_t* ids = nullptr; float* weights = nullptr;
    float* o_part = nullptr;       // [2][T][H] fp32: o_proj's two K parts (one workgroup each per block; role B adds them)
    int attn_splits = 1;           // KV ranges per (sequence, kv head) in the attention role (≤ 16); > 1 needs the two buffers below
    float* attn_partial = nullptr; // [T·nkv·attn_splits][16][head_dim + 4] fp32
    unsigned* attn_tickets = nullptr;   // [T·nkv] zeroed words (self-resetting)
    unsigned* cnt = nullptr;       // decode_chain_counter_words() words, zero on entry
    unsigned* cnt_next = nullptr;  // the other half of the double buffer: zeroed by this launch
    unsigned* timeout = nullptr;   // host-visible word, bumped by a bounded wait that gave up
};
int decode_chain_counter_words();
bool decode_chain_supports(const DecodeChainDesc& d);
int decode_chain_f16(const DecodeChainDesc& d, hipStream_t stream);

// ── sampling (sampling.hip) ──────────────────────────────────────────────────
int argmax_rows_f16_ws(const __half* logits, uint32_t* out_ids, const uint8_t* valid_mask, int mask_len, int m, int n,
                       float* workspace, size_t workspace_bytes, hipStream_t s);
// Device-side bookkeeping of a decode step, done by the last stage of the greedy argmax (one launch fewer per step):
// the sampled id of row i becomes the next input token, positions / kv lengths advance, the id is recorded in the history.
struct DecodeAdvance {
    uint32_t* tokens = nullptr; uint32_t* pos_offsets = nullptr; uint32_t* kv_lens = nullptr; uint32_t* history = nullptr;
    int32_t* step_counter = nullptr; unsigned* ticket = nullptr; int n = 0;
};
// *fused = 1 when the advance ran inside the argmax launches (two-stage form), 0 when the caller still has to run it
int argmax_pairs_f32(const float* logits, const uint32_t* local_ids, void* pairs, int rows, int n_local, int v0, const uint8_t* mask, int mask_len, hipStream_t s);
int argmax_merge_ranks(const void* gathered, uint32_t* out, int rows, int world, const DecodeAdvance* adv, hipStream_t s);
int apply_repetition_penalties_sparse_f32_shard(float* logits, const uint32_t* row_offsets, const uint32_t* token_ids,
                                                const float* penalties, int m, int n_local, int v0, hipStream_t s);
int argmax_rows_f32_ws_advance(const float* logits, uint32_t* out_ids, const uint8_t* valid_mask, int mask_len, int m, int n,
                               float* workspace, size_t workspace_bytes, const DecodeAdvance* adv, int* fused, hipStream_t s);
int argmax_rows_f32_ws(const float* logits, uint32_t* out_ids, const uint8_t* valid_mask, int mask_len, int m, int n,
                       float* workspace, size_t workspace_bytes, hipStream_t s);
int argmax_rows_f16(const __half* logits, uint32_t* out_ids, const uint8_t* valid_mask, int mask_len, int m, int n,
                    hipStream_t s);
int argmax_rows_f32(const float* logits, uint32_t* out_ids, const uint8_t* valid_mask, int mask_len, int m, int n,
                    hipStream_t s);
int apply_repetition_penalties_sparse_f32(float* logits, const uint32_t* row_offsets, const uint32_t* token_ids,
                                          const float* penalties, int m, int n, hipStream_t s);
int apply_repetition_penalties_sparse_f16(__half* logits, const uint32_t* row_offsets, const uint32_t* token_ids,
                                          const float* penalties, int m, int n, hipStream_t s);

}  // namespace fh
