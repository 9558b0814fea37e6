// Decoder runner: the C++ host side above the C-ABI kernels.  Mirrors the reference's
// `DecoderOnlyLLM` / `LlamaFamilyModel` / `Qwen3MoeModel` unified forward and the executor-level
// KV admission contract (see include/ferrum_hip.h for file:line citations).
#pragma once
#include <memory>
#include <unordered_map>

#include "../../include/ferrum_hip.h"
#include "block_allocator.h"
#include "kernels.h"
#include "kv_layout.h"

namespace fh {

struct LayerWeights {
    __half* input_ln = nullptr;
    __half* post_ln = nullptr;
    __half* q_norm = nullptr;
    __half* k_norm = nullptr;
    __half* router = nullptr;      // [E, H] fp16
    __half* post_attn_ln = nullptr;   // sandwich norms (Gemma 3); post_ln is then the pre-MLP norm
    __half* post_ffn_ln = nullptr;
    __half* qkv_bias = nullptr;       // staged until finalize (then owned by qkv.bias)
    W4Device qkv, o, gate_up, down;
    // act-order (desc_act) MLP: down's row permutation is folded into the column order of gate_up at repack, so the gated
    // activation leaves already permuted.  Whichever of the two arrives first waits here for the other.
    struct PendingGptq { std::vector<int32_t> qweight, qzeros, g_idx; std::vector<float> scales; int k = 0, n = 0; bool has_g_idx = false; };
    std::unique_ptr<PendingGptq> pending_gate_up;
    std::vector<int32_t> down_perm_host;   // empty: down has no permutation (or has not arrived)
    bool down_set = false;
    W4Device exp_gate_up, exp_down;   // stacked experts
    std::vector<uint8_t> exp_loaded;   // per expert: bit0 gate_up, bit1 down
    std::vector<int32_t> exp_gate_up_perm_host, exp_down_perm_host;   // act-order stacks: the row permutation every expert of the stack must share
    __half* k_pool = nullptr;
    __half* v_pool = nullptr;
};

struct SeqState {
    std::vector<uint32_t> blocks;
    int len = 0;
};

// Layout of the per-forward index block (host pinned mirror → device).
struct IndexLayout {
    size_t tokens, cu_seqlens, pos_offsets, kv_lens, sampled_idx, block_tables, total;
};

}  // namespace fh

struct FerrumHipModel {
    FerrumHipModelConfig cfg{};
    hipStream_t stream = nullptr;
    bool finalized = false;
    int max_blocks_per_seq = 0;

    // weights
    __half* embed = nullptr;
    __half* lm_head = nullptr;     // row-major upload; null → tied.  Freed once repacked.
    __half* lm_head_t = nullptr;   // f16t tiles used by the forward
    __half* final_norm = nullptr;
    float* cos_t = nullptr;
    float* sin_t = nullptr;
    float* cos_local = nullptr;    // RoPE table of the local-attention layers (rope_local_theta), else null
    float* sin_local = nullptr;
    float* residual_f32 = nullptr; // fp32 residual stream of sandwich-norm models
    __half* gather_scratch = nullptr;   // [max_tokens, max K] input column gather for act-order (desc_act) weights
    std::vector<fh::LayerWeights> layers;

    // KV
    std::unique_ptr<fh::BlockAllocator> alloc;
    std::unordered_map<uint64_t, fh::SeqState> seqs;
    uint64_t prefix_hits = 0, prefix_misses = 0, prefix_saved_tokens = 0;   // block-level prefix cache probes

    // scratch
    __half *residual = nullptr, *norm_out = nullptr, *qkv_out = nullptr, *q_out = nullptr, *attn_out = nullptr,
           *o_out = nullptr, *gate_up_out = nullptr, *act_out = nullptr, *mlp_out = nullptr, *sampled_hidden = nullptr,
           *moe_act = nullptr, *moe_down = nullptr, *moe_gather_x = nullptr, *moe_gather_h = nullptr;
    __half* residual2 = nullptr;          // ping-pong partner of `residual` for the Q-part route kernel
    float* chain_o_part = nullptr;        // decode chain: o_proj's two K parts, [2][128][H] fp32
    float* chain_attn_partial = nullptr;  // decode chain, attention role with KV splits: [≤ 1024 (unit, split)][16][head_dim + 4] states
    unsigned* chain_attn_tickets = nullptr;   // [4096] one self-resetting ticket per (sequence, kv head)
    fh::RouteCand* route_cand = nullptr;  // [T ≤ 64][Q][8]
    float* route_stats = nullptr;         // [T][Q][2]
    unsigned* route_arrive = nullptr;     // [T] arrival counters of the split route kernel (zero between launches)
    unsigned* em2_arrive = nullptr;       // [2][E] per-expert arrival counters of the merged gate_up → down launch (double buffer: each launch zeroes the other half) + [1] give-up count
    size_t arrive_half_words = 0;         // words per half: pair counters + chain counters
    int chain_parity = 0;                 // … and of the decode chain launch (chain.hip)
    int em2_parity = 0;                   // which half the next merged launch counts in (enqueue order = stream order)
    bool em2_failed = false;              // a bounded in-launch wait gave up once: the two-launch form from then on
    unsigned* inlaunch_timeouts = nullptr; // pinned host word (device-visible): bumped by any in-launch wait that gave up; read after every host sync
    unsigned inlaunch_timeouts_seen = 0;
    int route_parts = 4;                  // expert parts per token in the decode route kernel (1 = single-workgroup kernel)
    bool fuse_rope_attn = true;           // decode: QK-norm + RoPE + KV write inside the attention launch
    int fuse_tail_max_rows = 1;           // MoE decode at ≤ this many rows (≤ 4): a layer's combine + add + norm runs as the prologue of the next
                                          // q|k|v GEMM (0 = never).  Measured: c=1 430 → 437 tok/s; at 4 rows every workgroup repeating the 4 × 36 KB
                                          // of L2 reads costs more than the launch saved (c=4 1455 → 1335 tok/s), so only single rows fuse
    int route_gemm_min_tokens = 512;          // from this many tokens the router runs as a GEMM + top-k (3 launches)
    int moe_tile_min_pairs_per_expert = 32;   // average pairs per expert from which MoE GEMMs use 64-row LDS tiles
    int moe_tile96_min_pairs_per_expert = 32;         // … from which (even group counts) they use 96-row blocks through w4_gemm_big_kernel<6>
    int moe_tile128_min_pairs_per_expert = 1 << 30;   // … from which they would use 128-row blocks through w4_gemm_big_kernel: off —
                                                      // at K = 2048 / 768 (16 / 6 groups per tile, one workgroup per CU) padding and the
                                                      // un-overlapped prologue / epilogue cost more than the schedule gains (§3)
    int moe_tile32_min_pairs_per_expert = 8;  // … from which (below the 64-row threshold) they use 32-row LDS tiles
    int attn_flash_min_rows = 512;            // query rows (tokens × GQA group) per prompt from which attention takes the LDS-shared K/V form
    // decode (P ≤ 1024): average pairs per expert, in eighths, from which the grouped GEMMs run expert-major (0 = never).  The
    // one-launch gate_up → down pair moved the break-even down from 2 pairs per expert: c=20 4.08 → 3.84 ms per step, c=24 4.21 →
    // 3.92, c=28 4.38 → 3.99 (c=16, one pair per expert: 3.62 vs 3.64 — the block-major launches keep it)
    int moe_em_min_pairs_x8 = 9;
    bool dense_slabs = true;              // dense MLP block at 17–32 rows: slab GEMMs reduced by their consumers
    int o_slabs = 8;                      // split-K slabs of the o projection on the decode path (0 = direct)
    // expert parallelism (cfg.expert_parallel, tp_world > 1): this rank owns experts [ep_e0, ep_e0 + ep_E) of num_experts
    int ep_e0 = 0, ep_E = 0;
    int32_t* expert_ids_local = nullptr;  // [T·k] expert id − ep_e0 for the rank's own experts, −1 for the others
    float* ones = nullptr;                // [max_tokens] 1.0f: the all-reduced MoE output enters the residual like one more "expert row"
    // vocabulary-parallel lm_head (cfg.vocab_parallel, tp_world > 1): this rank scores vocabulary rows [vp_v0, vp_v0 + vp_n)
    int vp_v0 = 0, vp_n = 0;
    void* vp_pairs = nullptr;             // [max_seqs] (local winner's logit, global id)
    void* vp_gathered = nullptr;          // [world][max_seqs]
    float* router_logits = nullptr;
    int32_t *expert_ids = nullptr, *sorted_ids = nullptr, *block_ids = nullptr, *total_post_pad = nullptr;
    float* expert_w = nullptr;
    float* logits = nullptr;       // [max_seqs, V] fp32
    uint32_t* out_tokens = nullptr;   // [max_seqs]
    float* workspace = nullptr;
    size_t workspace_bytes = 0;
    float* taps = nullptr;         // [L, max_tokens, H] fp32 when enabled
    bool taps_enabled = false;
    int taps_tokens = 0;

    // index block
    fh::IndexLayout il{};
    uint8_t* idx_host = nullptr;   // pinned
    uint8_t* idx_dev = nullptr;

    // decode-steps state
    uint32_t* history = nullptr;   // [max_steps, max_seqs] device
    int history_cap = 0;
    int32_t* step_counter = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_n = 0, graph_max_kv = 0;
    bool graph_refused = false;    // a tensor-parallel step the runtime would not capture: eager from then on

    // tensor parallel (RCCL, resolved lazily by dlopen)
    uint8_t* greedy_opts_dev = nullptr;    // token mask + sparse repetition-penalty arrays of the current forward
    size_t greedy_opts_bytes = 0;
    struct FerrumHipComm* comm = nullptr;   // RCCL rank and/or one-shot peer group (tp_comm.hip)
    bool comm_owned = false;
    struct FerrumHipTpLoopback* tp_loopback = nullptr;   // in-process test stand-in for the communicator
    __half* tp_tmp = nullptr;
    size_t tp_tmp_elems = 0;
};
