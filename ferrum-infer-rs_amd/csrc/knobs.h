// Development knobs and kernel-form accounting.
//
// Knobs: every FERRUM_HIP_* tuning variable a launcher consults is read ONCE (library load) into this struct —
// no getenv on a launch path.  Tests that steer a kernel form set the variable and call
// ferrum_hip_debug_reload_knobs(); product code never does.
//
// Forms: each launcher records which kernel form it chose (one relaxed atomic add at enqueue time), so a test
// that names a form can assert that this form — not a neighbour picked by a drifted heuristic — produced the
// result it checked (ferrum_hip_debug_form_hits).  A hipGraph replay does not re-count its captured launches;
// FORM_GRAPH_REPLAY counts replays.
#pragma once
#include <stdint.h>

namespace fh {

struct Knobs {
    // paged attention (attention.hip)
    bool attn_no_flash = false;
    bool attn_flash_min_rows_set = false;
    long attn_flash_min_rows = 512;
    int attn_splits = 0;              // > 0: forced KV split count
    long attn_rs_min_wgs = 512;
    bool attn_no_rs = false;
    bool attn_narrow = false;
    bool attn_no_resident = false;    // never take the resident-K/V short-prompt prefill form
    long attn_resident_min_wgs = 128; // … and only from this many workgroups (tests lower it)
    bool attn_flash32 = false;        // the flash form with 32-key steps (paged_prefill_attn_kernel) for head_dim 128 too
    // INT4 GEMMs (w4_gemm.hip)
    int moe_kw_pairs = 16;
    int decode_chain = 1;             // MoE decode at ≤ 32 rows: tail + q|k|v + attention + o_proj + route as ONE launch (0 = five launches)
    int chain_max_rows = 128;         // decode chain up to this many rows (≤ 128)
    int sandwich_wide = 1;            // sandwich add + norms of ≤ 64 rows: 1024 threads per row, loads ahead of the reductions (0 = 256 threads)
    int chain_qkv_half = 1;           // decode chain at ≤ 16 rows: q|k|v in 32-column blocks (0 = 64)
    int chain_o_half = 1;             // … and o_proj
    int chain_split_keys = 256;       // decode chain attention: target keys per KV range (≤ 16 ranges per (sequence, kv head), see runner.hip); 0 = never split
    int chain_qkv_wide = -1;          // decode chain q|k|v blocks of 128 columns: -1 = where the 64-column blocks + attention exceed chain_slots
    int chain_slots = 256;            // workgroups of the chain kernel the chip holds at once (one per CU)
    int route_gemm_topk = 1;          // prefill router GEMM + top-k as one launch (E = 128)
    int chain_attn_splits = 0;        // > 0: that many KV ranges whatever the context (experiments)
    int chain_max_keys = 4096;        // decode chain only up to this many keys per attention workgroup (× T·nkv / 128)
    int dense_chain = 1;              // dense models at 17–32 rows: the attention half of the layer as the chain launch too (0 = five launches)
    int moe_em2 = 1;                  // decode: gate_up → down as one expert-major launch (0 = two launches)
    int moe_bm2 = 0;                  // decode at ≤ 64 pairs: gate_up → down as one block-major launch — measured slower than the two launches (profiles/r03_moe_bm2_timeline.txt): off
    int moe_deferred_merge = 1;       // decode chain: the grouped GEMM's prologue merges the router's candidate lists (0 = role B does)
    int w4_tile_min_m = 0;
    int w4_tile_wgs = 256;
    int w4_ldsa = 1;
    int w4_ldsa_nw = 0;
    int w4_ldsa_s = 0;
    int w4_ldsw = 0;             // EXPERIMENTS builds: 17–32-row dense GEMM through w4_gemm_ldsw_kernel (3 or 6 = ring slots)
    int w4_big = 0;                   // 0 auto, −1 never, 8 / 16 forces w4_gemm_big_kernel's 128- / 256-row tiles
    int w4_ldsk = 0;                  // EXPERIMENTS=1 builds only: nw·100 + kw·10 + d forces a w4_gemm_ldsk_kernel form (+1000: ares)
    int w4_nt = 0;
    int w4_w = 0;
    int lds_min_wgs = 128;
    int lds_min_groups = 8;
    // runner
    bool no_graph = false;
    bool trace_launches = false;
    bool time_same_layer = false;
    int tp_fused_norm = 1;            // tensor parallel, one-shot transport: all-reduce + residual add + norm as one launch (0 = two)
    int tp_oneshot = -1;              // -1 auto, 0 never, 1 always (tensor-parallel all-reduce form)
};

const Knobs& knobs();
void reload_knobs();

enum Form : int {
    FORM_ATTN_FLASH = 0,        // paged_prefill_attn_kernel (LDS-shared K/V, long prompts)
    FORM_ATTN_ROW_SPLIT,        // paged_attn_kernel<…, RS> (four row tiles per workgroup)
    FORM_ATTN_KV_WIDE,          // paged_attn_kernel 8 waves (decode, ≥ 8 block pairs per split)
    FORM_ATTN_KV_NARROW,        // paged_attn_kernel 4 waves
    FORM_ATTN_FUSED_QKV_WIDE,   // decode with QK-norm + RoPE + KV write in the prologue, 8 waves
    FORM_ATTN_FUSED_QKV_NARROW,
    FORM_ATTN_SPLIT_REDUCE,     // KV-split partials + reduce launch
    FORM_W4_WGSPLIT,            // ≤ 16 rows (and small projections): K split over the waves of a workgroup
    FORM_W4_LDSA,               // 17–32 rows: LDS-shared activations
    FORM_W4_TILEP,              // ≥ 33/64 rows: pipelined 64-row tiles
    FORM_W4_SLABS,              // fp32 split-K slabs from the skinny kernel
    FORM_W4_SLABS_LDS,          // … from the LDS-shared-activation kernel
    FORM_W4_SLABS_TILE,         // … from the pipelined tile kernel
    FORM_W4_ROWSUM,             // prefill tile kernel with activation row sums taken at LDS staging
    FORM_MOE_EXPERT_MAJOR,
    FORM_MOE_INLINE_ALIGN,
    FORM_MOE_BLOCK16,
    FORM_MOE_TILE64,
    FORM_MOE_TILE32,
    FORM_MOE_TILE_BIG,          // prefill: 96- or 128-pair blocks through w4_gemm_big_kernel
    FORM_MOE_MERGE_ROUTE,
    FORM_ROUTE_SPLIT,           // add + norm + router split over Q parts per token, merge in the launch
    FORM_ROUTE_FUSED,           // add + norm + router + top-k, one workgroup per token
    FORM_ROUTE_GEMM,            // router as a GEMM over all tokens + top-k kernel
    FORM_DENSE_SLAB_CHAIN,      // dense MLP block: slab GEMMs reduced by their consumers
    FORM_GRAPH_CAPTURE,
    FORM_GRAPH_REPLAY,
    FORM_TP_ALLREDUCE_RCCL,
    FORM_TP_ALLREDUCE_LOOPBACK,
    FORM_TP_ALLREDUCE_ONESHOT,
    FORM_F16_DENSE_LINEAR,      // unquantised projection (DenseLinear)
    FORM_W4_FUSED_TAIL,         // ≤ 4-row q|k|v GEMM with the previous layer's combine + add + norm as its prologue
    FORM_ATTN_RESIDENT,         // short-prompt prefill: the sequence's K/V resident in LDS
    FORM_W4_BIG,                // prefill: 128- / 256-row tiles, scale folded into the fp16 B operand
    FORM_W4_LDSK,               // 17–64 rows: LDS-shared activations, K split over the waves of a workgroup
    FORM_GATHER_COLUMNS,        // act-order input gather as a launch of its own (no producer wrote the permuted row)
    FORM_PERM_PRODUCER,         // act-order: the producing kernel (norm, gated activation via gate_up's column order, decode attention) wrote the permuted row
    FORM_MOE_EXPERT_MAJOR_PAIR, // gate_up → down in one expert-major launch (in-launch hand-off per expert)
    FORM_DECODE_CHAIN,          // the attention half of a MoE decode layer as one launch (chain.hip)
    FORM_MOE_DEFERRED_MERGE,    // the merged grouped GEMM took the routing as per-part candidate lists and merged them itself
    FORM_DENSE_CHAIN,           // dense model: tail + q|k|v + attention + o_proj + add/norm as the one chain launch (chain.hip)
    FORM_TP_ALLREDUCE_NORM_FUSED, // tensor parallel: one-shot all-reduce + residual add + norm as one launch
    FORM_MOE_BLOCK_MAJOR_PAIR,  // ≤ 64 pairs: gate_up → down in one block-major launch (in-launch hand-off per 16-row block)
    FORM_CHAIN_ATTN_KV_SPLITS,  // decode chain with several KV ranges per (sequence, kv head) in its attention role (ticket merge)
    FORM_CHAIN_QKV_WIDE,        // decode chain with 128-column q|k|v blocks (the narrow ones + attention exceed the resident workgroups)
    FORM_ROUTE_GEMM_TOPK,       // prefill router: logits (MFMA) + softmax + top-k in one launch (moe_route_gemm_topk_kernel)
    FORM_COUNT
};

void form_hit(Form f);
const char* form_name(int f);

}  // namespace fh
