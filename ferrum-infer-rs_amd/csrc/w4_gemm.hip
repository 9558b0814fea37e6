// GPTQ-INT4 dequant-GEMM for gfx950 (skinny-M / decode regime) and the MoE grouped form.
//
// Replaces, on the reference side: `GptqLinear::forward` → `B::load_gptq` + Marlin
// (ferrum-quantization/src/gptq.rs:42-112, ferrum-kernels/src/backend/cuda/marlin.rs:488-633)
// and `MarlinExpertStack::gemm_phase_vllm` (ferrum-kernels/src/marlin_expert_stack.rs:86).
// Arithmetic contract (cpu.rs:2283-2315): W[n,k] = (q − (z+1))·s[k/g, n].
//
// Design (HBM-bound at T ≤ 64):
//  * load-time repack GPTQ [K/8,N] → "w4t" tiles: one wave-instruction (64 lanes × 16 B = 1 KiB,
//    fully coalesced) fetches a 16-column × 128-row (= one quant group) tile whose per-lane dwords
//    ARE the B-operand fragments of four v_mfma_f32_16x16x32_f16 k-steps.
//  * weights go HBM → VGPR directly (no LDS round trip: each weight byte is used once);
//    nibbles are expanded to exact fp16 integers with the 0x6400 magic-number trick at 5 VALU per
//    8 weights (the kernel is VALU-issue-bound, not HBM-bound, above that); the 1024/zero-point
//    offsets are removed by four extra MFMAs per group instead of per-weight subtracts, and the group
//    scale is applied to the fp32 MFMA chain result, so the only rounding is fp32 accumulation.
//  * activations are the A operand (rows = tokens, padded to 16); a wave owns a 64-column
//    "supertile" so every A fragment feeds four MFMAs.
//  * split-K writes fp32 slabs that a small epilogue kernel sums in fixed order (deterministic).
#include "common.h"
#include "kernels.h"
#include "knobs.h"
#include "route_merge.h"
#include "w4_device.h"

namespace fh {

// ───────────────────────────── repacked layout ─────────────────────────────
// qw : u32  [n64][G][4 nt][64 lanes][4 dwords]          (G = K/128, n64 = ceil(N/64))
//      lane l=(a=l>>4, b=l&15), column n = st·64 + nt·16 + b.  Dwords come in pairs: (d0,d1) carry
//      k-steps 0 ("lo") and 1 ("hi") of the group, (d2,d3) k-steps 2 and 3.  Element j (0..7) of k-step s
//      is k = g·128 + 32s + 8a + j.  Inside dword d_{2p+h} (h = j>>2, jj = j&3) the LO k-step sits in
//      nibbles {0,4,2,6}[jj] and the HI k-step in nibbles {1,5,3,7}[jj]: a lo nibble masked in place is
//      the fp16 1024+q, a hi nibble masked in place is 1024+16q — 5 VALU ops expand a dword (see below).
// sc : f16  [n64][G][16 b][4 nt]                         group scale of column st·64+nt·16+b
// zp : f16  [n64][G][16 b][4 nt]  (only when asymmetric)  value (z+1) as fp16

// nibble slot of (hi k-step?, jj) inside a dword
static inline int nibble_slot(int hi, int jj) {
    static const int lo_slots[4] = {0, 4, 2, 6}, hi_slots[4] = {1, 5, 3, 7};
    return hi ? hi_slots[jj] : lo_slots[jj];
}

static inline uint16_t f32_to_f16_bits(float f) {
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

int w4_repack_host(const int32_t* qweight, const float* scales, const int32_t* qzeros,
                   const int32_t* g_idx, const int32_t* col_perm, int group_size, int k, int n,
                   W4HostPacked* out) {
    FH_REQUIRE(k % 128 == 0, "w4 repack: K=%d must be a multiple of 128", k);
    FH_REQUIRE(n % 8 == 0, "w4 repack: N=%d must be a multiple of 8", n);
    FH_REQUIRE(group_size > 0 && group_size % 128 == 0 && k % group_size == 0,
               "w4 repack: group_size=%d must be a multiple of 128 dividing K=%d", group_size, k);
    const int G = k / 128, n64 = (n + 63) / 64;
    const int num_groups = k / group_size;

    // act-order: rows sorted by g_idx so that quant groups become contiguous
    // (same transform the reference applies before Marlin: cuda/quant.rs:434 + gather_columns).
    std::vector<int32_t> perm;
    bool use_perm = false;
    if (g_idx) {
        bool sequential = true;
        for (int i = 0; i < k; i++)
            if (g_idx[i] != i / group_size) { sequential = false; break; }
        if (!sequential) {
            use_perm = true;
            perm.resize(k);
            for (int i = 0; i < k; i++) perm[i] = i;
            std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) { return g_idx[x] < g_idx[y]; });
            std::vector<int> counts(num_groups, 0);
            for (int i = 0; i < k; i++) {
                FH_REQUIRE(g_idx[i] >= 0 && g_idx[i] < num_groups, "w4 repack: g_idx[%d]=%d out of range", i, g_idx[i]);
                counts[g_idx[i]]++;
            }
            for (int g = 0; g < num_groups; g++)
                FH_REQUIRE(counts[g] == group_size, "w4 repack: act-order groups must be balanced (group %d has %d rows)", g, counts[g]);
        }
    }
    // symmetric ⇔ every zero nibble is 7 (zero point 8), the `sym=true` canonical form
    // (ferrum-quantization/src/native_safetensors.rs:1242).
    bool symmetric = true;
    for (long i = 0; i < (long)num_groups * (n / 8); i++)
        if ((uint32_t)qzeros[i] != 0x77777777u) { symmetric = false; break; }

    out->k = k; out->n = n; out->n64 = n64; out->G = G; out->symmetric = symmetric;
    out->qw.assign((size_t)n64 * G * 4 * 64 * 4, 0);
    out->sc.assign((size_t)n64 * G * 16 * 4, 0);
    if (!symmetric) out->zp.assign((size_t)n64 * G * 16 * 4, f32_to_f16_bits(8.0f));
    else out->zp.clear();
    out->perm = perm;

    auto src_k = [&](int kk) { return use_perm ? perm[kk] : kk; };
    for (int st = 0; st < n64; st++)
        for (int g = 0; g < G; g++) {
            for (int nt = 0; nt < 4; nt++)
                for (int lane = 0; lane < 64; lane++) {
                    int a = lane >> 4, b = lane & 15;
                    int np = st * 64 + nt * 16 + b;          // packed column
                    int col = np < n ? (col_perm ? col_perm[np] : np) : -1;
                    uint32_t dw[4] = {0, 0, 0, 0};
                    for (int s = 0; s < 4; s++)
                        for (int j = 0; j < 8; j++) {
                            int kk = src_k(g * 128 + 32 * s + 8 * a + j);
                            // padded columns decode to q − zero = 0 (zero point 8 when symmetric)
                            uint32_t q = col < 0 ? 8u
                                                 : (((uint32_t)qweight[(long)(kk / 8) * n + col] >> (4 * (kk % 8))) & 0xFu);
                            int d = 2 * (s >> 1) + (j >> 2);
                            dw[d] |= q << (4 * nibble_slot(s & 1, j & 3));
                        }
                    for (int d = 0; d < 4; d++)
                        out->qw[((((size_t)st * G + g) * 4 + nt) * 64 + lane) * 4 + d] = dw[d];
                }
            for (int b = 0; b < 16; b++)
                for (int nt = 0; nt < 4; nt++) {
                    int np = st * 64 + nt * 16 + b;
                    int col = np < n ? (col_perm ? col_perm[np] : np) : -1;
                    size_t idx = (((size_t)st * G + g) * 16 + b) * 4 + nt;
                    if (col < 0) { out->sc[idx] = 0; continue; }
                    int kk0 = src_k(g * 128);
                    int grp = g_idx ? g_idx[kk0] : kk0 / group_size;
                    out->sc[idx] = f32_to_f16_bits(scales[(long)grp * n + col]);
                    if (!symmetric) {
                        uint32_t zw = (uint32_t)qzeros[(long)grp * (n / 8) + col / 8];
                        int zero = (int)((zw >> (4 * (col % 8))) & 0xF) + 1;
                        out->zp[idx] = f32_to_f16_bits((float)zero);
                    }
                }
        }
    return 0;
}

// ─────────────────────────────── device code ───────────────────────────────

struct W4Args {
    const uint32_t* qw;   // repacked weights (expert 0)
    const __half* sc;
    const __half* zp;     // null when symmetric
    long expert_stride_qw;   // dwords between experts (MoE), 0 for dense
    long expert_stride_sc;   // halves between experts
    const __half* x;      // activations [rows, K]
    __half* out;          // fp16 output [rows, ldo] (non-split, or fused-act)
    float* partial;       // fp32 slabs [S][rows_pad][n_pad] when split-K
    const __half* bias;   // optional [N] (wgsplit kernel)
    int M;                // rows (dense) / number of valid pair ids (MoE: T·k)
    int K, N, G, n64;
    int ldo;              // output row stride (elements)
    int S;                // K splits
    // MoE routing (null for dense)
    const int32_t* sorted_token_ids;
    const int32_t* block_ids;
    const int32_t* total_post_pad;
    int top_k;            // input row = id / top_k
    int few_pairs_per_expert;   // the caller's promise that no expert holds more than 16 of the pairs (≤ 16 tokens, distinct experts per token)
    int rows_pad;         // slab row count
    int n_pad;            // slab column count (= n64·64)
    // inline align (decode-sized batches): raw per-pair expert ids instead of the three arrays above
    const int32_t* pair_expert_ids;
    int num_experts;
    // when set, the inline align publishes its result so the NEXT grouped GEMM (down) can read it
    int32_t* pub_sorted_token_ids;
    int32_t* pub_block_ids;
    int32_t* pub_total_post_pad;
    // route merge (decode): Q sorted candidate lists per token from add_rmsnorm_route_part_kernel
    const RouteCand* cand;
    const float* stats;
    int route_T, route_Q, route_K, norm_topk;
    int32_t* pub_expert_ids;   // [T·K] merged ids  (published by one workgroup)
    float* pub_expert_w;       // [T·K] combine weights
    // fused prologue of the ≤ 4-row skinny GEMM (w4_gemm_wgsplit_kernel<…, FUSE_A>): the input rows are not read from `x` but
    // computed — MoE combine + residual add + RMSNorm of the PREVIOUS layer's tail (fused.hip kernel A)
    const __half* fa_down;     // [M·top_k, K] expert outputs
    const float* fa_weights;   // [M·top_k] combine weights
    const __half* fa_res_in;   // [M, K] residual before the add
    __half* fa_res_out;        // [M, K] residual after the add (≠ fa_res_in: every workgroup reads, one writes)
    const __half* fa_ln;       // [K] norm weights
    float fa_eps;
    int fa_top_k;
#ifdef FERRUM_HIP_EXPERIMENTS
    unsigned long long* tl;    // development: per-wave wall-clock stamps of the LDS-shared-activation kernel (tools/exp_timeline.py)
#endif
};
#ifdef FERRUM_HIP_EXPERIMENTS
static unsigned long long* g_timeline = nullptr;
static int g_timeline_mode = 0;      // expert-major grouped GEMM: 0 every launch, 1 down, 2 gate_up
extern "C" __attribute__((visibility("default"))) void ferrum_hip_debug_set_timeline(void* p) { g_timeline = (unsigned long long*)p; }
extern "C" __attribute__((visibility("default"))) void ferrum_hip_debug_set_timeline_mode(int m) { g_timeline_mode = m; }
#define FH_TL(i)                                                                                                        \
    do {                                                                                                                \
        if (p.tl && lane == 0)                                                                                          \
            p.tl[((((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4 + (i)] = wall_clock64(); \
    } while (0)
#else
#define FH_TL(i) do {} while (0)
#endif

// Align-block-size computed INSIDE the grouped GEMM (P ≤ 1024 pairs): every workgroup derives its
// 16-row block (expert + ascending pair ids) from the raw router output with an LDS histogram, a wave
// scan of the padded counts and a ballot compaction — identical result to moe_align_block_size
// (ascending pair id inside an expert), but no separate ≈8 µs launch on the decode critical path.
// Merge the Q per-part candidate lists of every token (each sorted by logit desc, id asc) into the token's
// top-K, in the same order route_into picks them (ferrum-models/src/moe/router.rs:159-178: descending
// probability, ties → lower expert id), and derive the combine weights (softmax over all experts from the
// parts' (max, Σexp) statistics, optional renormalisation over the K picked — router.rs:141-157,180-193).
// One wave; lane t ↔ token t (T ≤ 64).  Results: s_ids[t·K+k] in LDS; one workgroup also publishes them.
__device__ __forceinline__ void merge_route_candidates(const RouteCand* __restrict__ cand, const float* __restrict__ stats,
                                                       int T, int Q, int K, int norm_topk, RouteCand* s_cand, int* s_ids,
                                                       bool publish, int32_t* pub_ids, float* pub_w) {
    const int lane = threadIdx.x & 63;
    const int n = T * Q * 8;
    for (int i = lane; i < n; i += 64) s_cand[i] = cand[i];
    __syncthreads();
    if (lane < T) {
        const int t = lane;
        float M = -INFINITY;
        for (int q = 0; q < Q; q++) M = fmaxf(M, stats[(t * Q + q) * 2]);
        float S = 0.f;
        for (int q = 0; q < Q; q++) S += stats[(t * Q + q) * 2 + 1] * expf(stats[(t * Q + q) * 2] - M);
        const float inv_sum = 1.0f / S;
        uint32_t heads = 0;                         // 4 bits per part: next unread candidate
        float pk[8];
        int idk[8];
        float sel_sum = 0.f;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            pk[k] = 0.f;
            idk[k] = 0;
            if (k < K) {
                float best_l = -INFINITY;
                int best_id = 0x7fffffff, best_q = 0;
                for (int q = 0; q < Q; q++) {
                    const int h = (heads >> (4 * q)) & 15;
                    if (h < 8) {
                        const RouteCand c = s_cand[(t * Q + q) * 8 + h];
                        if (c.logit > best_l || (c.logit == best_l && c.id < best_id)) { best_l = c.logit; best_id = c.id; best_q = q; }
                    }
                }
                heads += 1u << (4 * best_q);
                if (best_id == 0x7fffffff) best_id = 0;
                const float p = expf(best_l - M) * inv_sum;
                pk[k] = p;
                idk[k] = best_id;
                sel_sum += p;
                s_ids[t * K + k] = best_id;
            }
        }
        if (publish) {
            const float scale = norm_topk ? (sel_sum > 0.f ? 1.0f / sel_sum : 0.f) : 1.0f;
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k < K) {
                    pub_ids[t * K + k] = idk[k];
                    pub_w[t * K + k] = (norm_topk && !(sel_sum > 0.f)) ? 1.0f / (float)K : pk[k] * scale;
                }
        }
    }
    __syncthreads();
}

// LDS hand-over between the lanes of ONE wave (the other waves of the workgroup are elsewhere) or of the whole workgroup
template <bool ONE_WAVE>
__device__ __forceinline__ void lds_sync() {
    if constexpr (ONE_WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
        __syncthreads();
    }
}

// The same merge for a handful of tokens (T ≤ 8: decode at c ≤ 8), lane-parallel and without LDS staging: the lists of all
// tokens are requested at once (one 8-byte load per lane and token pair), every pair of tokens then takes route_merge_pair's
// 4-probe searches — short enough to sit in the prologue of every workgroup of the gate_up launch (role B of the decode chain
// then stops at the lists: chain.hip, defer_merge).  Only the publishing workgroup derives the combine weights.  One wave;
// results in s_ids[t·K + k].  Q ∈ {1, 2, 4}.
template <bool ONE_WAVE = false>
__device__ __forceinline__ void merge_route_lists_fast(const RouteCand* __restrict__ cand, const float* __restrict__ stats, int T,
                                                       int Q, int K, int norm_topk, int* s_ids, bool publish, int32_t* pub_ids,
                                                       float* pub_w) {
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, li = lane & 31, ncand = Q * 8;
    const unsigned long long* cg = reinterpret_cast<const unsigned long long*>(cand);
    const unsigned long long none = (0x7fffffffull << 32) | 0xff800000ull;
    unsigned long long c[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int t = 2 * r + half;
        c[r] = (t < T && li < ncand) ? cg[t * ncand + li] : none;           // RouteCand {logit, id}: low word logit, high word id
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (2 * r < T) {
            const int t = 2 * r + half;
            const bool valid = t < T && li < ncand;
            int id;
            const int rank = route_merge_pair(c[r], Q, &id);
            if (id == 0x7fffffff) id = 0;                                     // (fewer than K experts in all: as merge_route_candidates)
            const bool keep = valid && rank < K;
            if (keep) s_ids[t * K + rank] = id;
            if (publish) {
                const float ww = route_merge_pair_weight(c[r], rank, valid, stats + (long)(t < T ? t : 0) * Q * 2, Q, K, norm_topk);
                if (keep) { pub_ids[t * K + rank] = id; pub_w[t * K + rank] = ww; }
            }
        }
    }
    lds_sync<ONE_WAVE>();
}

// Align-block-size computed INSIDE the grouped GEMM (P ≤ 1024 pairs) from the per-pair expert ids staged in
// LDS: LDS histogram, wave scan of the padded counts, ballot compaction — identical result to
// moe_align_block_size (ascending pair id inside an expert) without the separate ≈8 µs launch.
// One wave per workgroup.
template <bool ONE_WAVE = false>
__device__ __forceinline__ bool inline_align_block(const int* s_ids, int P, int E, int rb, int* s_cnt, int* s_rows,
                                                   int* expert_out, int* total_blocks_out) {
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < E; i += 64) s_cnt[i] = 0;
    if (lane < 16) s_rows[lane] = P;      // sentinel
    lds_sync<ONE_WAVE>();
    for (int p = lane; p < P; p += 64) {
        int e = s_ids[p];
        if (e >= 0 && e < E) atomicAdd(&s_cnt[e], 1);
    }
    lds_sync<ONE_WAVE>();
    // lane owns experts [8·lane, 8·lane+8): blocks per expert, then an exclusive wave scan
    int nb[8], local = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int e = lane * 8 + i;
        nb[i] = e < E ? (s_cnt[e] + 15) >> 4 : 0;
        local += nb[i];
    }
    int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    const int total = __shfl(incl, 63, 64);
    *total_blocks_out = total;
    if (rb >= total) return false;
    const int excl = incl - local;
    int found = -1;          // (expert << 16) | block-within-expert, on the owning lane
    if (rb >= excl && rb < incl) {
        int base = excl;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (rb >= base && rb < base + nb[i]) found = ((lane * 8 + i) << 16) | (rb - base);
            base += nb[i];
        }
    }
    const unsigned long long owner = __ballot(found >= 0);
    found = __shfl(found, __ffsll((long long)owner) - 1, 64);
    const int e_star = found >> 16, j = found & 0xffff;
    int base = 0;
    for (int p0 = 0; p0 < P; p0 += 64) {
        int p = p0 + lane;
        bool mine = p < P && s_ids[p] == e_star;
        unsigned long long bal = __ballot(mine);
        int r = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (mine && (r >> 4) == j) s_rows[r & 15] = p;
        base += __popcll(bal);
    }
    lds_sync<ONE_WAVE>();
    *expert_out = e_star;
    return true;
}

// The align for a few tokens (P ≤ 128 pairs, E ≤ 128 experts, every expert at most once per token and T ≤ 16 tokens — so one
// 16-row block per active expert): no histogram, no scan.  The active experts are two 64-bit words of presence bits; block rb
// is the rb-th set bit; its rows are the pairs holding that expert, compacted in ascending pair order — the arrays
// moe_align_block_size would produce.  One wave, two pairs per lane; ≈ 0.3 µs instead of inline_align_block's ≈ 1.5.
template <bool ONE_WAVE = false>
__device__ __forceinline__ bool align_block_few_pairs(const int* s_ids, int P, int rb, unsigned* s_bits, int* s_rows, int* expert_out,
                                                      int* total_blocks_out) {
    const int lane = threadIdx.x & 63;
    if (lane < 4) s_bits[lane] = 0u;
    if (lane < 16) s_rows[lane] = P;      // sentinel
    lds_sync<ONE_WAVE>();
    const int e0 = lane < P ? s_ids[lane] : -1, e1 = 64 + lane < P ? s_ids[64 + lane] : -1;
    if (e0 >= 0 && e0 < 128) atomicOr(&s_bits[e0 >> 5], 1u << (e0 & 31));
    if (e1 >= 0 && e1 < 128) atomicOr(&s_bits[e1 >> 5], 1u << (e1 & 31));
    lds_sync<ONE_WAVE>();
    const unsigned long long lo = ((unsigned long long)s_bits[1] << 32) | s_bits[0], hi = ((unsigned long long)s_bits[3] << 32) | s_bits[2];
    const int n_lo = __popcll(lo), total = n_lo + __popcll(hi);
    *total_blocks_out = total;
    if (rb >= total) return false;
    // lane l ↔ bit l of the word that holds the rb-th set bit
    const unsigned long long w = rb < n_lo ? lo : hi;
    const int want = rb < n_lo ? rb : rb - n_lo;
    const bool hit = ((w >> lane) & 1ull) && __popcll(w & ((1ull << lane) - 1ull)) == want;
    const int e_star = (__ffsll((long long)__ballot(hit)) - 1) + (rb < n_lo ? 0 : 64);
    const bool m0 = e0 == e_star, m1 = e1 == e_star;
    const unsigned long long b0 = __ballot(m0), b1 = __ballot(m1);
    if (m0) s_rows[__popcll(b0 & ((1ull << lane) - 1ull)) & 15] = lane;
    if (m1) s_rows[(__popcll(b0) + __popcll(b1 & ((1ull << lane) - 1ull))) & 15] = 64 + lane;
    lds_sync<ONE_WAVE>();
    *expert_out = e_star;
    return true;
}

// MODE: 0 dense, 1 MoE (plain output), 2 MoE gate_up with fused silu·mul epilogue
// KW (MoE only): waves per workgroup that split K.  Small batches (≤ 64 pairs) launch too few one-wave workgroups to
// cover the chip and each streams its 64 KiB at one group per memory round trip; with KW = 4 every wave takes a quarter
// of K (all its groups in flight) and the partial sums meet in LDS — c=1 gate_up 11.8 → ≈6 µs.  Each wave runs the
// (cheap) routing prologue on its own LDS slice, so no cross-wave protocol is needed before the reduce.
// RT (MoE only): the routing prologue (candidate merge / align) runs inside the launch and owns ≈ 12 KiB of LDS per
// workgroup; with explicit align arrays (RT = false) no LDS is allocated, so 25 one-wave workgroups per CU stay resident
// instead of 13 (prefill-side batches of a few hundred tokens launch ≈ 6500 of them).
template <int MT, bool HAS_ZP, int MODE, int KW = 1, bool RT = true>
__global__ __launch_bounds__(MODE == 0 ? 256 : 64 * KW) void w4_gemm_kernel(W4Args p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int a = lane >> 4, b = lane & 15;
    // dense: 4 waves per workgroup, one 64-column supertile each.  MoE: ONE wave per workgroup — the
    // grouped GEMM is bound by the per-CU fetch rate (≈10 B/clk/CU), so its time is the byte count of the
    // most loaded CU; wave-granular workgroups balance 64-KiB streams over the 256 CUs to ±1.
    const int st = MODE == 0 ? blockIdx.x * 4 + wave : blockIdx.x;
    if (st >= p.n64) return;
    const int rb = blockIdx.y;                     // row block (16·MT rows)
    const int z = blockIdx.z;                      // K split

    const uint32_t* qw = p.qw;
    const __half* sc = p.sc;
    const __half* zp = p.zp;
    int row_in[MT], row_out[MT];
    bool row_ok[MT];
    if (MODE == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            int r = rb * 16 * MT + mt * 16 + b;
            row_ok[mt] = r < p.M;
            row_in[mt] = row_ok[mt] ? r : p.M - 1;
            row_out[mt] = r;
        }
    } else {
        // MoE: one 16-row block of sorted pair ids, all of one expert.
        // LDS: raw[0..8K) holds the route candidates during the merge, then (first 2 KB) the histogram
        int e, id;
        if constexpr (RT) {
            __shared__ __attribute__((aligned(16))) unsigned char s_raw_all[KW][KW == 1 ? 8192 : 2048];
            __shared__ int s_ids_all[KW][KW == 1 ? 1024 : 64], s_rows_all[KW][16];
            unsigned char* s_raw = s_raw_all[KW == 1 ? 0 : wave];
            int* s_ids = s_ids_all[KW == 1 ? 0 : wave];
            int* s_rows = s_rows_all[KW == 1 ? 0 : wave];
            int* s_cnt = reinterpret_cast<int*>(s_raw);
            const bool publisher = blockIdx.x == 0 && rb == 0 && wave == 0;
            if (p.cand && p.route_T <= 8 && (p.route_Q == 1 || p.route_Q == 2 || p.route_Q == 4)) {
                merge_route_lists_fast(p.cand, p.stats, p.route_T, p.route_Q, p.route_K, p.norm_topk, s_ids, publisher,
                                       p.pub_expert_ids, p.pub_expert_w);
            } else if (p.cand) {      // (KW = 4: every wave merges on its own slice — ≤ 256 candidates, ≤ 64 pairs; launch_w4 checks)
                merge_route_candidates(p.cand, p.stats, p.route_T, p.route_Q, p.route_K, p.norm_topk,
                                       reinterpret_cast<RouteCand*>(s_raw), s_ids, publisher, p.pub_expert_ids, p.pub_expert_w);
            } else {
                for (int i = lane; i < p.M; i += 64) s_ids[i] = p.pair_expert_ids[i];
                __syncthreads();
            }
            int total_blocks;
            if (((p.cand && p.route_T <= 8) || p.few_pairs_per_expert) && p.M <= (KW == 1 ? 128 : 64) && p.num_experts <= 128) {
                if (!align_block_few_pairs(s_ids, p.M, rb, reinterpret_cast<unsigned*>(s_cnt), s_rows, &e, &total_blocks)) return;
            } else if (!inline_align_block(s_ids, p.M, p.num_experts, rb, s_cnt, s_rows, &e, &total_blocks)) return;
            id = s_rows[b];
            if (p.pub_sorted_token_ids && blockIdx.x == 0 && wave == 0) {
                if (threadIdx.x < 16) p.pub_sorted_token_ids[rb * 16 + threadIdx.x] = s_rows[threadIdx.x];
                if (threadIdx.x == 0) {
                    p.pub_block_ids[rb] = e;
                    if (rb == 0) *p.pub_total_post_pad = total_blocks * 16;
                }
            }
        } else {
            const int total = *p.total_post_pad;
            if (rb * 16 >= total) return;
            e = p.block_ids[rb];
            id = p.sorted_token_ids[rb * 16 + b];
        }
        qw += (long)e * p.expert_stride_qw;
        sc += (long)e * p.expert_stride_sc;
        if (HAS_ZP) zp += (long)e * p.expert_stride_sc;
        row_ok[0] = id < p.M;
        row_out[0] = id;
        row_in[0] = row_ok[0] ? id / p.top_k : 0;
    }

    int g0 = (int)((long)p.G * z / p.S), g1 = (int)((long)p.G * (z + 1) / p.S);
    if (MODE != 0 && KW > 1) {   // this wave's quarter of K
        const int span = g1 - g0, base = g0;
        g0 = base + span * wave / KW;
        g1 = base + span * (wave + 1) / KW;
    }
    // per-lane bases
    typedef uint32_t u32x4g __attribute__((ext_vector_type(4)));
    const u32x4g* qw_lane = reinterpret_cast<const u32x4g*>(qw) + ((long)st * p.G * 4) * 64 + lane;
    const uint2* sc_lane = reinterpret_cast<const uint2*>(sc) + ((long)st * p.G) * 16 + b;
    const uint2* zp_lane = HAS_ZP ? reinterpret_cast<const uint2*>(zp) + ((long)st * p.G) * 16 + b : nullptr;
    const __half* xrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) xrow[mt] = p.x + (long)row_in[mt] * p.K + 8 * a;

    float4v acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) acc[mt][nt] = (float4v){0.f, 0.f, 0.f, 0.f};

    // register double buffer: group g+1 is in flight while group g is consumed
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 wq[2][4];
    uint2 scv[2], zpv[2];
    half8 af[2][MT][4];
    auto issue = [&](int buf, int g) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) wq[buf][nt] = __builtin_nontemporal_load(qw_lane + ((long)g * 4 + nt) * 64);
        scv[buf] = sc_lane[(long)g * 16];
        if (HAS_ZP) zpv[buf] = zp_lane[(long)g * 16];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int s = 0; s < 4; s++)
                af[buf][mt][s] = *reinterpret_cast<const half8*>(xrow[mt] + g * 128 + 32 * s);
    };
    auto consume = [&](int buf) {
        const unsigned long long sb = ((unsigned long long)scv[buf].y << 32) | scv[buf].x;
        const unsigned long long zb = HAS_ZP ? (((unsigned long long)zpv[buf].y << 32) | zpv[buf].x) : 0ull;
        w4_consume_group<MT, 4, HAS_ZP>(wq[buf], sb, zb, 0, af[buf], acc);
    };

    // sched_barrier(0) pins "issue next group" ABOVE "consume this group": without it hipcc sinks the
    // loads below the MFMAs (to shorten live ranges) and every group pays its full HBM latency.
#define FH_PIN() __builtin_amdgcn_sched_barrier(0)
    if (g0 < g1) {
        issue(0, g0);
        FH_PIN();
        int g = g0;
        for (; g + 2 <= g1 - 1; g += 2) {   // two groups per trip keeps buffer indices static
            issue(1, g + 1);
            FH_PIN();
            consume(0);
            FH_PIN();
            issue(0, g + 2);
            FH_PIN();
            consume(1);
            FH_PIN();
        }
        if (g + 1 < g1) {
            issue(1, g + 1);
            FH_PIN();
            consume(0);
            consume(1);
        } else {
            consume(0);
        }
    }
#undef FH_PIN
    if (MODE != 0 && KW > 1) {
        // partial sums of the KW K-quarters meet in LDS; wave 0 adds them in wave order and owns the epilogue
        __shared__ float kred[KW][16][64];
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) kred[wave][nt * 4 + r][lane] = acc[0][nt][r];
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float s = kred[0][nt * 4 + r][lane];
#pragma unroll
                for (int k = 1; k < KW; k++) s += kred[k][nt * 4 + r][lane];
                acc[0][nt][r] = s;
            }
    }

    // accumulator map (v_mfma_f32_16x16x32): column = lane&15 → n, row = 4·(lane>>4)+r → token.
    // Lanes exchange nothing: every lane owns token rows 4a..4a+3 of column b.  The A operand
    // was loaded with lane b ↔ token b, so token t = 4a'+r lives in lanes with (lane>>4)==a'.
    // (D row index is the A row index, i.e. the token slot 0..15 of this block.)
    int out_rows[MT][4];
    bool out_ok[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            // the token slot 4a+r's routing lives in lane (·, b=4a+r): fetch it by shuffle
            int src_lane = 4 * a + r;
            out_rows[mt][r] = __shfl(row_out[mt], src_lane, 64);
            out_ok[mt][r] = __shfl((int)row_ok[mt], src_lane, 64) != 0;
        }

    if (MODE == 2) {
        // supertile = [16 gate | 16 gate | 16 up | 16 up] (column-permuted at repack):
        // act[row][st·32 + j·16 + b] = silu(gate)·up   (cpu.rs:1666-1680)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (!out_ok[0][r]) continue;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                float gt = acc[0][j][r], up = acc[0][2 + j][r];
                float v = (gt / (1.0f + __expf(-gt))) * up;
                int col = st * 32 + j * 16 + b;
                if (col < p.ldo) p.out[(long)out_rows[0][r] * p.ldo + col] = __float2half(v);
            }
        }
        return;
    }
    if (p.S > 1 || (MODE == 0 && p.partial)) {
        float* slab = p.partial + (long)z * p.rows_pad * p.n_pad;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int row = MODE == 0 ? rb * 16 * MT + mt * 16 + 4 * a + r : rb * 16 + 4 * a + r;
#pragma unroll
                for (int nt = 0; nt < 4; nt++)
                    slab[(long)row * p.n_pad + st * 64 + nt * 16 + b] = acc[mt][nt][r];
            }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (!out_ok[mt][r]) continue;
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                int col = st * 64 + nt * 16 + b;
                if (col < p.N) p.out[(long)out_rows[mt][r] * p.ldo + col] = __float2half(acc[mt][nt][r]);
            }
        }
}

// ── MoE grouped GEMM, expert-major (decode batches where most experts are routed to) ───────────────────────────
// w4_gemm_kernel's MoE modes are block-major: a wave must finish the align (LDS histogram, scan, compaction ≈ 2 µs) before
// it knows WHICH expert's weights to stream, and every wave of the launch pays that with nothing in flight.  Here the grid
// is (64-column supertile, expert): the wave requests its expert's first weight group at once, finds the expert's pairs
// meanwhile (ballot compaction of pair_expert_ids == e, ascending pair id — the order moe_align_block_size produces), and
// leaves if there are none.  More than 16 pairs of one expert (rare at P ≤ 8·E) take further passes over the weights (L2).
// Same per-row arithmetic as the block-major kernel (w4_consume_group, one 16-row tile): bit-identical outputs.
// (138 VGPRs + 16 AGPRs: three waves per SIMD, 3072 resident at once — all of gate_up's, three quarters of down's 4096, whose
// last quarter starts when the first waves leave, tools/exp_timeline_moe.py.  Forcing four per SIMD spills into the loop
// (32 → 52 µs), a non-interleaved consume at four per SIMD still spills (down 17.8 → 21.3 µs), four-wave workgroups change nothing
// (the bound is registers, not workgroup slots), and two supertiles per wave for down — 2048 waves, one round — land on the same
// 17.8 µs: the second round is not what holds down at 4.9 TB/s.)
template <bool HAS_ZP, int MODE>
__global__ __launch_bounds__(64) void w4_gemm_moe_em_kernel(W4Args p) {
    static_assert(MODE == 1 || MODE == 2, "grouped-GEMM modes only");
    const int lane = threadIdx.x;
    const int a = lane >> 4, b = lane & 15;
    const int st = blockIdx.x, e = blockIdx.y;
    __shared__ int s_rows[1024];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* qw_lane = reinterpret_cast<const u32x4*>(p.qw + (long)e * p.expert_stride_qw) + ((long)st * p.G * 4) * 64 + lane;
    const uint2* sc_lane = reinterpret_cast<const uint2*>(p.sc + (long)e * p.expert_stride_sc) + ((long)st * p.G) * 16 + b;
    const uint2* zp_lane = HAS_ZP ? reinterpret_cast<const uint2*>(p.zp + (long)e * p.expert_stride_sc) + ((long)st * p.G) * 16 + b : nullptr;
    u32x4 wq[2][4];
    uint2 scv[2], zpv[2];
    half8 af[2][1][4];
    auto issue_w = [&](int buf, int g) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) wq[buf][nt] = __builtin_nontemporal_load(qw_lane + ((long)g * 4 + nt) * 64);
        scv[buf] = sc_lane[(long)g * 16];
        if (HAS_ZP) zpv[buf] = zp_lane[(long)g * 16];
    };
    FH_TL(0);
    // the routing and the first weight group are requested together
    int ids[16];
    const int P = p.M, chunks = (P + 63) >> 6;
#pragma unroll
    for (int i = 0; i < 16; i++) ids[i] = (i < chunks && i * 64 + lane < P) ? p.pair_expert_ids[i * 64 + lane] : -1;
    issue_w(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    int n_e = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (i < chunks) {
            const bool mine = ids[i] == e;
            const unsigned long long bal = __ballot(mine);
            if (mine) s_rows[n_e + __popcll(bal & ((1ull << lane) - 1ull))] = i * 64 + lane;
            n_e += __popcll(bal);
        }
    }
    FH_TL(1);
    if (n_e == 0) return;
    __syncthreads();
    for (int j = 0; j * 16 < n_e; j++) {
        const int id = j * 16 + b < n_e ? s_rows[j * 16 + b] : P;
        const bool row_ok = id < P;
        int orow[4];          // D row 4a + r ↔ block row 4a + r: its pair id, read now so the epilogue waits for nothing
#pragma unroll
        for (int r = 0; r < 4; r++) orow[r] = j * 16 + 4 * a + r < n_e ? s_rows[j * 16 + 4 * a + r] : P;
        const __half* xrow = p.x + (long)(row_ok ? id / p.top_k : 0) * p.K + 8 * a;
        auto issue_a = [&](int buf, int g) {
#pragma unroll
            for (int s = 0; s < 4; s++) af[buf][0][s] = *reinterpret_cast<const half8*>(xrow + g * 128 + 32 * s);
        };
        float4v acc[1][4];
#pragma unroll
        for (int nt = 0; nt < 4; nt++) acc[0][nt] = (float4v){0.f, 0.f, 0.f, 0.f};
        auto consume = [&](int buf) {
            const unsigned long long sb = ((unsigned long long)scv[buf].y << 32) | scv[buf].x;
            const unsigned long long zb = HAS_ZP ? (((unsigned long long)zpv[buf].y << 32) | zpv[buf].x) : 0ull;
            w4_consume_group<1, 4, HAS_ZP>(wq[buf], sb, zb, 0, af[buf], acc);
        };
#define FH_PIN() __builtin_amdgcn_sched_barrier(0)
        if (j > 0) issue_w(0, 0);                    // further passes re-stream the expert (L2)
        issue_a(0, 0);
        FH_PIN();
        int g = 0;
        const int g1 = p.G;
        for (; g + 2 <= g1 - 1; g += 2) {            // two groups per trip keeps buffer indices static
            issue_w(1, g + 1); issue_a(1, g + 1);
            FH_PIN();
            consume(0);
            FH_PIN();
            issue_w(0, g + 2); issue_a(0, g + 2);
            FH_PIN();
            consume(1);
            FH_PIN();
        }
        if (g + 1 < g1) {
            issue_w(1, g + 1); issue_a(1, g + 1);
            FH_PIN();
            consume(0);
            consume(1);
        } else {
            consume(0);
        }
#undef FH_PIN
        FH_TL(2);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (orow[r] >= P) continue;
            if (MODE == 2) {
#pragma unroll
                for (int jj = 0; jj < 2; jj++) {
                    const float gt = acc[0][jj][r], up = acc[0][2 + jj][r];
                    const float v = (gt / (1.0f + __expf(-gt))) * up;
                    const int col = st * 32 + jj * 16 + b;
                    if (col < p.ldo) p.out[(long)orow[r] * p.ldo + col] = __float2half(v);
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
                    const int col = st * 64 + nt * 16 + b;
                    if (col < p.N) p.out[(long)orow[r] * p.ldo + col] = __float2half(acc[0][nt][r]);
                }
            }
        }
#ifdef FERRUM_HIP_EXPERIMENTS
        if (p.tl) { __builtin_amdgcn_s_waitcnt(0); FH_TL(3); }
#endif
    }
}

// ── gate_up → down of a decode batch in ONE expert-major launch ────────────────────────────────────────────────────
// The two grouped GEMMs of a layer as one grid: blocks [0, n64_gu·E) are gate_up tiles (column supertile fastest, expert
// next — the em kernel's own order), blocks behind them down tiles.  All gate_up waves are resident at once (three per
// SIMD), so a down wave only ever gets a slot a gate_up wave has left; it asks for its expert's first two weight groups at
// once, finds the expert's pairs, and then waits until the expert's gate_up tiles have all arrived (one counter per
// expert, n64_gu arrivals) before it reads the gated activations.  What that buys over two launches: no kernel boundary,
// gate_up's tail (the last few hundred 64-KiB streams cannot fill the memory pipe) runs beside down's weight requests, and
// down's first-data latency is gone.  Hand-off (MI355X_MICROARCH.md § visibility, valid forms, first row): the gated
// activations are stored write-through (8-byte sc1 stores — the transposed accumulator, w4_consume_group_t), every storing
// wave drains its stores and then adds to its expert's counter; the consumer polls that counter with relaxed sc1 loads
// (one lane, s_sleep between polls) and reads the activations with sc1 buffer loads only.  No wave waits before it has
// produced everything it will ever produce, every wait is bounded (a give-up bumps `timeout`, which the host reads at its
// next synchronisation), and nothing depends on where a block runs.  The counters of the NEXT launch (the other half of a
// double buffer) are zeroed here, so no memset node sits in front of the launch.
struct W4Em2Args {
    const uint32_t* gu_qw; const __half* gu_sc; const __half* gu_zp; long gu_stride_qw, gu_stride_sc; int gu_G, gu_n64;
    const uint32_t* dn_qw; const __half* dn_sc; const __half* dn_zp; long dn_stride_qw, dn_stride_sc; int dn_G, dn_n64;
    const __half* x;                 // [T, K] normalised rows (input row of pair p = p / top_k)
    __half* h;                       // [P, I] gated activations: written and read inside the launch
    __half* out;                     // [P, H] expert outputs
    const int32_t* pair_expert_ids;  // [P]
    // … or the router's Q candidate lists per token (role B of the decode chain, chain.hip): every workgroup merges them in its
    // prologue, under its first weight loads; workgroup 0 publishes the merged ids / combine weights for the layer's tail
    const RouteCand* cand; const float* stats; int route_T, route_Q, norm_topk;
    int32_t* pub_expert_ids; float* pub_expert_w;
    int P, top_k, E, K, I, H;
    unsigned* arrive;                // [E · EM2_STRIDE] one counter per expert, each on a 256-byte line of its own; zero on entry
    unsigned* arrive_next;           // the same for the launch after this one: zeroed here
    unsigned* timeout;
#ifdef FERRUM_HIP_EXPERIMENTS
    unsigned long long* tl;          // development: per-wave wall-clock stamps (tools/exp_timeline_moe.py)
#endif
};
// Thousands of waves poll 128 counters: packed into 512 bytes they would all sit behind one or two memory channels (polls
// and arrivals queue up behind each other there — the first build of this kernel took 108 µs instead of 50); one counter per
// 256-byte line spreads them over the channels.
constexpr int EM2_STRIDE = MOE_PAIR_COUNTER_STRIDE;
#ifdef FERRUM_HIP_EXPERIMENTS
#define FH_TL2(i) do { if (p.tl && lane == 0) p.tl[(long)blockIdx.x * 4 + (i)] = wall_clock64(); } while (0)
#else
#define FH_TL2(i) do {} while (0)
#endif

template <bool HAS_ZP, bool IS_GU, bool CAND>
__device__ __forceinline__ void w4_em2_role(const W4Em2Args& p, int tile, int* s_rows, _Float16* s_out) {
    const int lane = threadIdx.x;
    const int a = lane >> 4, b = lane & 15;
    const int n64 = IS_GU ? p.gu_n64 : p.dn_n64, G = IS_GU ? p.gu_G : p.dn_G;
    const int st = tile % n64, e = tile / n64;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t* qw0 = IS_GU ? p.gu_qw : p.dn_qw;
    const __half* sc0 = IS_GU ? p.gu_sc : p.dn_sc;
    const __half* zp0 = IS_GU ? p.gu_zp : p.dn_zp;
    const long sqw = IS_GU ? p.gu_stride_qw : p.dn_stride_qw, ssc = IS_GU ? p.gu_stride_sc : p.dn_stride_sc;
    const u32x4* qw_lane = reinterpret_cast<const u32x4*>(qw0 + (long)e * sqw) + ((long)st * G * 4) * 64 + lane;
    const uint2* sc_lane = reinterpret_cast<const uint2*>(sc0 + (long)e * ssc) + ((long)st * G) * 16 + b;
    const uint2* zp_lane = HAS_ZP ? reinterpret_cast<const uint2*>(zp0 + (long)e * ssc) + ((long)st * G) * 16 + b : nullptr;
    u32x4 wq[2][4];
    uint2 scv[2], zpv[2];
    half8 af[2][1][4];
    auto issue_w = [&](int buf, int g) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) wq[buf][nt] = __builtin_nontemporal_load(qw_lane + ((long)g * 4 + nt) * 64);
        scv[buf] = sc_lane[(long)g * 16];
        if (HAS_ZP) zpv[buf] = zp_lane[(long)g * 16];
    };
    if (IS_GU && st == 0 && lane == 0) p.arrive_next[e * EM2_STRIDE] = 0u;
    FH_TL2(0);
    int ids[16];
    const int P = p.P, chunks = (P + 63) >> 6;
    if constexpr (!CAND) {
#pragma unroll
        for (int i = 0; i < 16; i++) ids[i] = (i < chunks && i * 64 + lane < P) ? p.pair_expert_ids[i * 64 + lane] : -1;
    }
    issue_w(0, 0);
    if (!IS_GU) issue_w(1, 1);         // (down: G ≥ 2, checked by the launcher) both ring slots are on their way before the wait
    __builtin_amdgcn_sched_barrier(0);
    int n_e = 0;
    if constexpr (!CAND) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (i < chunks) {
                const bool mine = ids[i] == e;
                const unsigned long long bal = __ballot(mine);
                if (mine) s_rows[n_e + __popcll(bal & ((1ull << lane) - 1ull))] = i * 64 + lane;
                n_e += __popcll(bal);
            }
        }
    } else {
        // Candidate mode: lane t looks at token t's Q ≤ 4 lists of 8 (logit, id) pairs.  Is expert e among the token's top-K, and
        // at which rank?  The merged order is (logit descending, ties → lower id) (merge_route_candidates / router.rs:159-178):
        // rank = the number of the token's candidates that come before e's.  Pair id = t·K + rank — ascending with the lane, the
        // order moe_align_block_size produces.  The list of e's own part first (64 bytes per lane); the other parts only for the
        // lanes that found e there (every workgroup of the launch reads these few KiB: the fewer bytes the better).
        const int tiles = (p.E + 15) >> 4, EQ = (tiles / p.route_Q) * 16, qe = e / EQ;
        const bool tok = lane < p.route_T;
        const u32x4* cg = reinterpret_cast<const u32x4*>(p.cand) + (long)(tok ? lane : 0) * p.route_Q * 4;
        u32x4 own[4];
#pragma unroll
        for (int j = 0; j < 4; j++) own[j] = cg[qe * 4 + j];
        float le = 0.f;
        int rank = -1;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int h = 0; h < 2; h++)
                if ((int)own[j][2 * h + 1] == e) { rank = 2 * j + h; le = __uint_as_float(own[j][2 * h]); }      // (a sorted list: the ones before it beat it)
        const bool found = tok && rank >= 0;
        if (found) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (q < p.route_Q && q != qe) {
                    u32x4 o[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) o[j] = cg[q * 4 + j];
#pragma unroll
                    for (int j = 0; j < 4; j++)
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const float l = __uint_as_float(o[j][2 * h]);
                            const int id = (int)o[j][2 * h + 1];
                            rank += (l > le || (l == le && id < e)) ? 1 : 0;
                        }
                }
            }
        }
        const bool mine = found && rank < p.top_k;
        const unsigned long long bal = __ballot(mine);
        if (mine) s_rows[__popcll(bal & ((1ull << lane) - 1ull))] = lane * p.top_k + rank;
        n_e = __popcll(bal);
    }
    FH_TL2(1);
    if (n_e == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // (one wave per workgroup: its own LDS writes, in order)
    __builtin_amdgcn_wave_barrier();
    float4v acc[1][4];
    auto consume = [&](int buf) {
        const unsigned long long sb = ((unsigned long long)scv[buf].y << 32) | scv[buf].x;
        const unsigned long long zb = HAS_ZP ? (((unsigned long long)zpv[buf].y << 32) | zpv[buf].x) : 0ull;
        w4_consume_group<1, 4, HAS_ZP>(wq[buf], sb, zb, 0, af[buf], acc);
    };
#define FH_PIN() __builtin_amdgcn_sched_barrier(0)
    if constexpr (IS_GU) {
        for (int j = 0; j * 16 < n_e; j++) {
            const int id = j * 16 + b < n_e ? s_rows[j * 16 + b] : P;
            const __half* xrow = p.x + (long)(id < P ? id / p.top_k : 0) * p.K + 8 * a;
            auto issue_a = [&](int buf, int g) {
#pragma unroll
                for (int s = 0; s < 4; s++) af[buf][0][s] = *reinterpret_cast<const half8*>(xrow + g * 128 + 32 * s);
            };
#pragma unroll
            for (int nt = 0; nt < 4; nt++) acc[0][nt] = (float4v){0.f, 0.f, 0.f, 0.f};
            if (j > 0) issue_w(0, 0);                    // further passes re-stream the expert (L2)
            issue_a(0, 0);
            FH_PIN();
            int g = 0;
            for (; g + 2 <= G - 1; g += 2) {
                issue_w(1, g + 1); issue_a(1, g + 1);
                FH_PIN();
                consume(0);
                FH_PIN();
                issue_w(0, g + 2); issue_a(0, g + 2);
                FH_PIN();
                consume(1);
                FH_PIN();
            }
            if (g + 1 < G) {
                issue_w(1, g + 1); issue_a(1, g + 1);
                FH_PIN();
                consume(0);
                consume(1);
            } else {
                consume(0);
            }
            // silu(gate)·up → the 16 × 32 output block in LDS → one 16-byte write-through store per lane (row lane/4, 8 columns)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int jj = 0; jj < 2; jj++) {
                    const float gt = acc[0][jj][r], up = acc[0][2 + jj][r];
                    s_out[(4 * a + r) * 32 + jj * 16 + b] = (_Float16)((gt / (1.0f + __expf(-gt))) * up);
                }
            __builtin_amdgcn_wave_barrier();
            {
                const int row = lane >> 2, c8 = (lane & 3) * 8;
                const u32x4 v = *reinterpret_cast<const u32x4*>(&s_out[row * 32 + c8]);
                const int rid = j * 16 + row < n_e ? s_rows[j * 16 + row] : P;
                const int col = st * 32 + c8;
                if (rid < P && col + 7 < p.I) {
                    const __amdgpu_buffer_rsrc_t hw = __builtin_amdgcn_make_buffer_rsrc(p.h, 0, 0x7fffffff, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(v, hw, (rid * p.I + col) * 2, 0, 16);      // aux 16 = sc1
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // every store of this wave has left before the expert's counter moves
        FH_TL2(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(p.arrive + e * EM2_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        FH_TL2(3);
    } else {
        // down: wait for the expert's gate_up tiles (bounded: ≈ 20 ms of the 100 MHz wall clock)
        {
            const unsigned need = (unsigned)p.gu_n64;
            const unsigned long long t0 = wall_clock64();
            unsigned spins = 0;
            for (;;) {
                unsigned c = lane == 0 ? __hip_atomic_load(p.arrive + e * EM2_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                c = __builtin_amdgcn_readfirstlane(c);
                if (c >= need) break;
                // the expert's tiles arrive over ≈ 15 µs: long naps (≈ 1.7 µs) while most are missing, short ones for the last few
                if (c + 4 < need) __builtin_amdgcn_s_sleep(64);
                else __builtin_amdgcn_s_sleep(8);
                if ((++spins & 63u) == 0u && wall_clock64() - t0 > 2000000ull) {
                    if (lane == 0) __hip_atomic_fetch_add(p.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     // no instruction: keeps the loads below behind the poll
        }
        FH_TL2(2);
        const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.h, 0, P * p.I * 2, 0x00020000);
        for (int j = 0; j * 16 < n_e; j++) {
            const int id = j * 16 + b < n_e ? s_rows[j * 16 + b] : P;
            // (a lane without a row reads past the buffer's end: zeros, and no memory request — every byte read here comes from
            // the memory side, 16 lanes of row 0 per wave would be most of the launch's activation traffic)
            const int hoff = id < P ? (id * p.I + 8 * a) * 2 : 0x7ffffff0 - 4096;      // bytes; P·I·2 < 2 GiB
            auto issue_a = [&](int buf, int g) {
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, hoff + (g * 128 + 32 * s) * 2, 0, 16);   // aux 16 = sc1
                    af[buf][0][s] = __builtin_bit_cast(half8, v);
                }
            };
#pragma unroll
            for (int nt = 0; nt < 4; nt++) acc[0][nt] = (float4v){0.f, 0.f, 0.f, 0.f};
            if (j > 0) { issue_w(0, 0); issue_w(1, 1); }
            issue_a(0, 0); issue_a(1, 1);
            FH_PIN();
            int g = 0;
            for (; g + 3 < G; g += 2) {
                consume(0);
                FH_PIN();
                issue_w(0, g + 2); issue_a(0, g + 2);
                FH_PIN();
                consume(1);
                FH_PIN();
                issue_w(1, g + 3); issue_a(1, g + 3);
                FH_PIN();
            }
            consume(0);
            if (g + 2 < G) {
                FH_PIN();
                issue_w(0, g + 2); issue_a(0, g + 2);
                FH_PIN();
                consume(1);
                consume(0);
            } else {
                consume(1);
            }
            // the 16 × 64 output block through LDS → two 16-byte stores per lane (rows lane/8 and 8 + lane/8)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
                    // (the f32 sum is rounded to f32 first, like every other form: left alone, the asymmetric instantiation folds the
                    // last group's fma and this conversion into one v_fma_mixlo_f16 — a single rounding, one fp16 ulp apart on
                    // about one output in 2^13)
                    float v = acc[0][nt][r];
                    asm volatile("" : "+v"(v));
                    s_out[(4 * a + r) * 64 + nt * 16 + b] = (_Float16)v;
                }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int row = hh * 8 + (lane >> 3), c8 = (lane & 7) * 8;
                const u32x4 v = *reinterpret_cast<const u32x4*>(&s_out[row * 64 + c8]);
                const int rid = j * 16 + row < n_e ? s_rows[j * 16 + row] : P;
                const int col = st * 64 + c8;
                if (rid < P && col + 7 < p.H) *reinterpret_cast<u32x4*>(p.out + (long)rid * p.H + col) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
#ifdef FERRUM_HIP_EXPERIMENTS
        if (p.tl) { __builtin_amdgcn_s_waitcnt(0); FH_TL2(3); }
#endif
    }
#undef FH_PIN
}

// (three waves per SIMD are what keeps all gate_up tiles resident: without the attribute the two inlined roles take 154 + 36
// registers — two waves per SIMD; with it 168, and one 8-byte spill in the prologue, outside every loop)
template <bool HAS_ZP, bool CAND>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void w4_gemm_moe_em2_kernel(W4Em2Args p) {
    __shared__ __attribute__((aligned(16))) int s_rows[1024];
    __shared__ __attribute__((aligned(16))) _Float16 s_out[16 * 64];     // the tile's output rows, for 16-byte stores
    const int n_gu = p.gu_n64 * p.E, n_dn = p.dn_n64 * p.E;
    if ((int)blockIdx.x < n_gu) { w4_em2_role<HAS_ZP, true, CAND>(p, blockIdx.x, s_rows, s_out); return; }
    if ((int)blockIdx.x < n_gu + n_dn) { w4_em2_role<HAS_ZP, false, CAND>(p, blockIdx.x - n_gu, s_rows, s_out); return; }
    if constexpr (CAND) {
        // behind every tile: the merged ids and combine weights of 16 tokens per workgroup, for the layer's tail (the next
        // launch) — the full merge with the softmax statistics, which the tiles above did not need
        const int t0 = ((int)blockIdx.x - n_gu - n_dn) * 16, tn = p.route_T - t0 < 16 ? p.route_T - t0 : 16;
        if (tn <= 0) return;
        merge_route_candidates(p.cand + (long)t0 * p.route_Q * 8, p.stats + (long)t0 * p.route_Q * 2, tn, p.route_Q, p.top_k, p.norm_topk,
                               reinterpret_cast<RouteCand*>(s_rows), reinterpret_cast<int*>(s_out), true,
                               p.pub_expert_ids + t0 * p.top_k, p.pub_expert_w + t0 * p.top_k);
    }
}

// ── MoE gate_up (+ silu·mul) → down of a SMALL decode batch (≤ 64 pairs: c ≤ 8) as ONE block-major launch ───────────────
// Few pairs means few active experts (8 at c = 1): the expert-major grid above would launch 7168 workgroups to find 448 with
// work, and the two block-major launches it replaces (w4_gemm_kernel MODE 2 / MODE 1) each pay a launch boundary, their own
// routing prologue and one quant group per memory round trip.  Here the grid is (gate_up tiles + down tiles) × 16-row blocks,
// block index slowest, so every down tile is dispatched after the gate_up tiles of its block:
//   * wave 0 of a workgroup derives the routing (the router's candidate lists merged in place — merge_route_lists_fast — or the
//     pair ids), runs the align, and hands expert + rows to the other three waves through LDS;
//   * the four waves split K (the same split as w4_gemm_kernel's KW = 4 form: identical partial sums, identical bits) and every
//     wave has ALL its quant groups in flight at once — the whole 64-KiB tile of a gate_up workgroup — instead of two;
//   * gate_up tiles write the gated activations with write-through stores, drain, and count an arrival for their block; down
//     tiles request their weights first, wait for the block's gu_n64 arrivals, then read the activations with sc1 loads
//     (the hand-off of w4_gemm_moe_em2_kernel, one counter per block on a 256-byte line of its own).
struct W4Bm2Args {
    const uint32_t* gu_qw; const __half* gu_sc; const __half* gu_zp; long gu_stride_qw, gu_stride_sc; int gu_G, gu_n64;
    const uint32_t* dn_qw; const __half* dn_sc; const __half* dn_zp; long dn_stride_qw, dn_stride_sc; int dn_G, dn_n64;
    const __half* x; __half* h; __half* out;
    const int32_t* pair_expert_ids;
    const RouteCand* cand; const float* stats; int route_T, route_Q, norm_topk;
    int32_t* pub_expert_ids; float* pub_expert_w;
    int P, top_k, E, K, I, H;
    unsigned* arrive; unsigned* arrive_next; unsigned* timeout;
#ifdef FERRUM_HIP_EXPERIMENTS
    unsigned long long* tl;
#endif
};
#ifdef FERRUM_HIP_EXPERIMENTS
#define FH_TL3(i) do { if (p.tl && threadIdx.x == 0) p.tl[((long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (i)] = wall_clock64(); } while (0)
#else
#define FH_TL3(i) do {} while (0)
#endif

template <bool HAS_ZP>
__global__ __launch_bounds__(256) void w4_gemm_moe_bm2_kernel(W4Bm2Args p) {
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[2048];   // align histogram (E ≤ 512 counters)
    __shared__ int s_ids[64], s_rows[16], s_hdr[2];
    __shared__ float kred[4][16][64];
    __shared__ __attribute__((aligned(16))) _Float16 s_out[16 * 64];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int a = lane >> 4, b = lane & 15;
    const int rb = blockIdx.y;
    const bool is_gu = (int)blockIdx.x < p.gu_n64;
    const int st = is_gu ? blockIdx.x : blockIdx.x - p.gu_n64;
    if (blockIdx.x == 0 && threadIdx.x == 0) p.arrive_next[rb * EM2_STRIDE] = 0u;
    FH_TL3(0);
    const int P = p.P;
    if (wave == 0) {
        // routing and align by ONE wave (wave-level LDS hand-overs); the other three meet it at the barrier below
        const bool publisher = (int)blockIdx.x == p.gu_n64 && rb == 0;      // a down tile: it waits for its block anyway
        if (p.cand) {
            merge_route_lists_fast<true>(p.cand, p.stats, p.route_T, p.route_Q, p.top_k, p.norm_topk, s_ids, publisher,
                                         p.pub_expert_ids, p.pub_expert_w);
        } else {
            if (lane < P) s_ids[lane] = p.pair_expert_ids[lane];
            lds_sync<true>();
        }
        int e0 = 0, total = 0;
        const bool ok = (p.cand && p.E <= 128)
                            ? align_block_few_pairs<true>(s_ids, P, rb, reinterpret_cast<unsigned*>(s_raw), s_rows, &e0, &total)
                            : inline_align_block<true>(s_ids, P, p.E, rb, reinterpret_cast<int*>(s_raw), s_rows, &e0, &total);
        if (lane == 0) { s_hdr[0] = ok ? e0 : -1; s_hdr[1] = total; }
    }
    __syncthreads();
    const int e = s_hdr[0];
    FH_TL3(1);
    if (e < 0) return;
    const int id = s_rows[b];
    const bool row_ok = id < P;

    const int G = is_gu ? p.gu_G : p.dn_G;
    const int g0 = G * wave / 4, g1 = G * (wave + 1) / 4, ng = g1 - g0;                    // this wave's quarter of K (≤ 4 groups)
    const uint32_t* qw0 = is_gu ? p.gu_qw : p.dn_qw;
    const __half* sc0 = is_gu ? p.gu_sc : p.dn_sc;
    const __half* zp0 = is_gu ? p.gu_zp : p.dn_zp;
    const long sqw = is_gu ? p.gu_stride_qw : p.dn_stride_qw, ssc = is_gu ? p.gu_stride_sc : p.dn_stride_sc;
    const u32x4* qw_lane = reinterpret_cast<const u32x4*>(qw0 + (long)e * sqw) + ((long)st * G * 4) * 64 + lane;
    const uint2* sc_lane = reinterpret_cast<const uint2*>(sc0 + (long)e * ssc) + ((long)st * G) * 16 + b;
    const uint2* zp_lane = HAS_ZP ? reinterpret_cast<const uint2*>(zp0 + (long)e * ssc) + ((long)st * G) * 16 + b : nullptr;
    u32x4 wq[4][4];
    uint2 scv[4], zpv[4];
    half8 af[4][1][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i < ng) {
            const int g = g0 + i;
#pragma unroll
            for (int nt = 0; nt < 4; nt++) wq[i][nt] = __builtin_nontemporal_load(qw_lane + ((long)g * 4 + nt) * 64);
            scv[i] = sc_lane[(long)g * 16];
            if (HAS_ZP) zpv[i] = zp_lane[(long)g * 16];
        }
    }
    float4v acc[1][4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) acc[0][nt] = (float4v){0.f, 0.f, 0.f, 0.f};
    if (is_gu) {
        const __half* xrow = p.x + (long)(row_ok ? id / p.top_k : 0) * p.K + 8 * a;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (i < ng) {
#pragma unroll
                for (int s = 0; s < 4; s++) af[i][0][s] = *reinterpret_cast<const half8*>(xrow + (g0 + i) * 128 + 32 * s);
            }
    } else {
        // down: the weights are on their way; now the block's gate_up tiles (bounded wait, ≈ 20 ms of the 100 MHz clock)
        __builtin_amdgcn_sched_barrier(0);
        const unsigned need = (unsigned)p.gu_n64;
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
            unsigned c = lane == 0 ? __hip_atomic_load(p.arrive + rb * EM2_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            c = __builtin_amdgcn_readfirstlane(c);
            if (c >= need) break;
            __builtin_amdgcn_s_sleep(4);
            if ((++spins & 255u) == 0u && wall_clock64() - t0 > 2000000ull) {
                if (lane == 0) __hip_atomic_fetch_add(p.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        FH_TL3(2);
        const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.h, 0, P * p.I * 2, 0x00020000);
        const int hoff = row_ok ? (id * p.I + 8 * a) * 2 : 0x7ffffff0 - 8192;      // (no row: past the end — zeros, no request)
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (i < ng) {
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, hoff + ((g0 + i) * 128 + 32 * s) * 2, 0, 16);   // sc1
                    af[i][0][s] = __builtin_bit_cast(half8, v);
                }
            }
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (i < ng) {
            const unsigned long long sb = ((unsigned long long)scv[i].y << 32) | scv[i].x;
            const unsigned long long zb = HAS_ZP ? (((unsigned long long)zpv[i].y << 32) | zpv[i].x) : 0ull;
            w4_consume_group<1, 4, HAS_ZP>(wq[i], sb, zb, 0, af[i], acc);
        }
    // the four K-quarters meet in LDS; wave 0 adds them in wave order (as w4_gemm_kernel's KW = 4 form) and owns the epilogue
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++) kred[wave][nt * 4 + r][lane] = acc[0][nt][r];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float s = kred[0][nt * 4 + r][lane];
#pragma unroll
            for (int k = 1; k < 4; k++) s += kred[k][nt * 4 + r][lane];
            acc[0][nt][r] = s;
        }
    if (is_gu) {
        // silu(gate)·up → the 16 × 32 block in LDS → one 16-byte write-through store per lane (row lane/4, 8 columns)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                const float gt = acc[0][jj][r], up = acc[0][2 + jj][r];
                s_out[(4 * a + r) * 32 + jj * 16 + b] = (_Float16)((gt / (1.0f + __expf(-gt))) * up);
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        {
            const int row = lane >> 2, c8 = (lane & 3) * 8;
            const u32x4 v = *reinterpret_cast<const u32x4*>(&s_out[row * 32 + c8]);
            const int rid = s_rows[row];
            const int col = st * 32 + c8;
            if (rid < P && col + 7 < p.I) {
                const __amdgpu_buffer_rsrc_t hw = __builtin_amdgcn_make_buffer_rsrc(p.h, 0, 0x7fffffff, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(v, hw, (rid * p.I + col) * 2, 0, 16);      // aux 16 = sc1
            }
        }
        FH_TL3(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(p.arrive + rb * EM2_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        FH_TL3(3);
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                float v = acc[0][nt][r];
                asm volatile("" : "+v"(v));           // (f32 → f16 as its own rounding, like every other form; see the em2 epilogue)
                s_out[(4 * a + r) * 64 + nt * 16 + b] = (_Float16)v;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int row = hh * 8 + (lane >> 3), c8 = (lane & 7) * 8;
            const u32x4 v = *reinterpret_cast<const u32x4*>(&s_out[row * 64 + c8]);
            const int rid = s_rows[row];
            const int col = st * 64 + c8;
            if (rid < P && col + 7 < p.H) *reinterpret_cast<u32x4*>(p.out + (long)rid * p.H + col) = v;
        }
#ifdef FERRUM_HIP_EXPERIMENTS
        if (p.tl) { __builtin_amdgcn_s_waitcnt(0); FH_TL3(3); }
#endif
    }
}

// Sum S fp32 slabs in fixed order → fp16 [M, N] (optionally gathering padded MoE rows).
template <typename OutT>
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, OutT* __restrict__ out, int S,
                                     int M, int N, int rows_pad, int n_pad, int ldo) {
    int col = blockIdx.x * blockDim.x + threadIdx.x;
    int row = blockIdx.y;
    if (col >= N || row >= M) return;
    float s = 0.f;
    for (int z = 0; z < S; z++) s += partial[((long)z * rows_pad + row) * n_pad + col];
    out[(long)row * ldo + col] = (OutT)s;
}

__global__ void splitk_reduce_bias_kernel(const float* __restrict__ partial, __half* __restrict__ out,
                                          const __half* __restrict__ bias, int S, int M, int N, int rows_pad, int n_pad,
                                          int ldo) {
    int col = blockIdx.x * blockDim.x + threadIdx.x;
    int row = blockIdx.y;
    if (col >= N || row >= M) return;
    float s = 0.f;
    for (int z = 0; z < S; z++) s += partial[((long)z * rows_pad + row) * n_pad + col];
    if (bias) s += __half2float(bias[col]);
    out[(long)row * ldo + col] = __float2half(s);
}

// ── dense skinny GEMM, K split ACROSS THE WAVES of one workgroup ─────────────────────────────────
// The small projections of the decode layer (qkv, o) are latency-bound: a slab split-K needs a second
// launch (≈5 µs) to reduce.  Here the W waves of a workgroup take K-slices of the same NT column
// tiles and reduce their fp32 accumulators through LDS, so the GEMM is ONE launch with fp16 output.
// FUSE_A (MT = 1, ≤ 4 rows — decode at c ≤ 4, where every dependent launch costs ≈ 4 µs whatever it does): the activation rows
// are produced by the workgroup itself — residual' = residual + Σ_k w_k·down_k (k ascending, fp32, one fp16 rounding: the
// arithmetic of moe_combine_add_rmsnorm_kernel), x = rms_norm(residual')·ln — into LDS, after the first weight group has been
// requested; workgroup 0 also stores residual'.  Every workgroup repeats the ≈ 36 KB per row of L2 reads: cheap at ≤ 4 rows,
// prohibitive at 32, which is why only the small batches fuse.
template <int MT, int NT, bool HAS_ZP, bool FUSE_A = false>
__global__ __launch_bounds__(MT == 1 ? 1024 : (MT == 2 ? (NT == 1 ? 1024 : 512) : 256)) void w4_gemm_wgsplit_kernel(W4Args p) {
    static_assert(!FUSE_A || MT == 1, "the fused prologue serves one 16-row tile");
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int W = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int a = lane >> 4, b = lane & 15;
    const int tile0 = blockIdx.x * NT;            // first 16-column tile of this workgroup
    const int st = tile0 >> 2, nt0 = tile0 & 3;   // NT divides 4 → all tiles in one supertile
    const int rb = blockIdx.y;
    constexpr int V = MT * NT * 4;

    const __half* xrow[MT];
    const int KP = p.K + 8;                                   // LDS row pitch of the fused rows (16 B of padding: rows on different banks)
    __half* xs = reinterpret_cast<__half*>(red + (size_t)W * V * 64);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        int r = rb * 16 * MT + mt * 16 + b;
        if (FUSE_A) xrow[mt] = xs + (long)(r < p.M ? r : p.M - 1) * KP + 8 * a;
        else xrow[mt] = p.x + (long)(r < p.M ? r : p.M - 1) * p.K + 8 * a;
    }
    typedef uint32_t u32x4g __attribute__((ext_vector_type(4)));
    const u32x4g* qw_lane = reinterpret_cast<const u32x4g*>(p.qw) + ((long)st * p.G * 4 + nt0) * 64 + lane;
    const uint2* sc_lane = reinterpret_cast<const uint2*>(p.sc) + ((long)st * p.G) * 16 + b;
    const uint2* zp_lane = HAS_ZP ? reinterpret_cast<const uint2*>(p.zp) + ((long)st * p.G) * 16 + b : nullptr;

    float4v acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[mt][nt] = (float4v){0.f, 0.f, 0.f, 0.f};

    const int g0 = (int)((long)p.G * wave / W), g1 = (int)((long)p.G * (wave + 1) / W);
    u32x4g wq[2][NT];
    uint2 scv[2], zpv[2];
    half8 af[2][MT][4];
    auto issue_w = [&](int buf, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) wq[buf][nt] = __builtin_nontemporal_load(qw_lane + ((long)g * 4 + nt) * 64);
        scv[buf] = sc_lane[(long)g * 16];
        if (HAS_ZP) zpv[buf] = zp_lane[(long)g * 16];
    };
    auto issue_a = [&](int buf, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int s = 0; s < 4; s++) af[buf][mt][s] = *reinterpret_cast<const half8*>(xrow[mt] + g * 128 + 32 * s);
    };
    auto issue = [&](int buf, int g) __attribute__((always_inline)) { issue_w(buf, g); issue_a(buf, g); };
    auto consume = [&](int buf) {
        const unsigned long long sb = ((unsigned long long)scv[buf].y << 32) | scv[buf].x;
        const unsigned long long zb = HAS_ZP ? (((unsigned long long)zpv[buf].y << 32) | zpv[buf].x) : 0ull;
        w4_consume_group<MT, NT, HAS_ZP, false>(wq[buf], sb, zb, nt0, af[buf], acc);
    };
#define FH_PIN() __builtin_amdgcn_sched_barrier(0)
    if (FUSE_A) {
        if (g0 < g1) issue_w(0, g0);                          // the first weight group travels while the rows are made
        FH_PIN();
        float* wred = reinterpret_cast<float*>(xs + (size_t)p.M * KP);     // [W] partial sums of squares
        const int nvec = p.K >> 3, tk = p.fa_top_k;
        const bool writer = blockIdx.x == 0 && blockIdx.y == 0;
        for (int t = 0; t < p.M; t++) {
            float ss = 0.f;
            for (int i = threadIdx.x; i < nvec; i += blockDim.x) {
                float accv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                float wk8[8];
                half8 d8[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {                  // the first 8 expert rows are requested together
                    const int kc = k < tk ? k : tk - 1;
                    wk8[k] = p.fa_weights[t * tk + kc];
                    d8[k] = *reinterpret_cast<const half8*>(p.fa_down + ((long)t * tk + kc) * p.K + i * 8);
                }
                half8 rv = *reinterpret_cast<const half8*>(p.fa_res_in + (long)t * p.K + i * 8);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (k < tk) {
#pragma unroll
                        for (int j = 0; j < 8; j++) accv[j] += wk8[k] * (float)d8[k][j];
                    }
                for (int k = 8; k < tk; k++) {
                    const float wk = p.fa_weights[t * tk + k];
                    const half8 d = *reinterpret_cast<const half8*>(p.fa_down + ((long)t * tk + k) * p.K + i * 8);
#pragma unroll
                    for (int j = 0; j < 8; j++) accv[j] += wk * (float)d[j];
                }
#pragma unroll
                for (int j = 0; j < 8; j++) rv[j] = (_Float16)((float)rv[j] + accv[j]);
                if (writer) *reinterpret_cast<half8*>(p.fa_res_out + (long)t * p.K + i * 8) = rv;
                *reinterpret_cast<half8*>(xs + (long)t * KP + i * 8) = rv;           // un-normalised for now
#pragma unroll
                for (int j = 0; j < 8; j++) ss += (float)rv[j] * (float)rv[j];
            }
            ss = wave_reduce_sum(ss);
            if (lane == 0) wred[wave] = ss;
            __syncthreads();
            float total = 0.f;
            for (int w = 0; w < W; w++) total += wred[w];
            const float inv = 1.0f / sqrtf(total / (float)p.K + p.fa_eps);
            for (int i = threadIdx.x; i < nvec; i += blockDim.x) {
                half8 rv = *reinterpret_cast<const half8*>(xs + (long)t * KP + i * 8);       // this thread's own store
                const half8 wv = *reinterpret_cast<const half8*>(p.fa_ln + i * 8);
#pragma unroll
                for (int j = 0; j < 8; j++) rv[j] = (_Float16)((float)rv[j] * inv * (float)wv[j]);
                *reinterpret_cast<half8*>(xs + (long)t * KP + i * 8) = rv;
            }
            __syncthreads();
        }
        if (g0 < g1) issue_a(0, g0);
        FH_PIN();
    }
    if (g0 < g1) {
        if (!FUSE_A) issue(0, g0);
        FH_PIN();
        int g = g0;
        for (; g + 2 <= g1 - 1; g += 2) {
            issue(1, g + 1);
            FH_PIN();
            consume(0);
            FH_PIN();
            issue(0, g + 2);
            FH_PIN();
            consume(1);
            FH_PIN();
        }
        if (g + 1 < g1) {
            issue(1, g + 1);
            FH_PIN();
            consume(0);
            consume(1);
        } else {
            consume(0);
        }
    }
#undef FH_PIN
    // cross-wave reduction through LDS: red[wave][v][lane]
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[((wave * V) + (mt * NT + nt) * 4 + r) * 64 + lane] = acc[mt][nt][r];
    __syncthreads();
    for (int v = wave; v < V; v += W) {
        float sum = 0.f;
        for (int w = 0; w < W; w++) sum += red[(w * V + v) * 64 + lane];
        const int r = v & 3, nt = (v >> 2) % NT, mt = (v >> 2) / NT;
        const int row = rb * 16 * MT + mt * 16 + 4 * a + r;
        const int col = (tile0 + nt) * 16 + b;
        if (row < p.M && col < p.N) {
            if (p.bias) sum += __half2float(p.bias[col]);
            p.out[(long)row * p.ldo + col] = __float2half(sum);
        }
    }
}

// ── dense GEMM for 17–64 rows: activations staged ONCE per workgroup in LDS ───────────────────────
// With MT ≥ 2 row tiles a wave that fetches its own A fragments pulls 2–4× more activation bytes (from L2)
// than weight bytes (from HBM) and the kernel runs at the activation rate (gate_up 4096→28672: 12.6 µs at
// m = 1, 25.7 µs at m = 32).  Here the NW waves of a workgroup own NW neighbouring 64-column supertiles over
// the SAME K range: each 128-k group's activations (MT·4 KiB, fragment-major) are loaded cooperatively into
// a double-buffered LDS tile and every wave reads its MFMA A fragments with conflict-free ds_read_b128.
// One barrier per group.  K may be split over grid.z into fp32 slabs (reduced by the consumer kernel).
template <int MT, int NW, bool HAS_ZP>
__global__ __launch_bounds__(NW * 64) void w4_gemm_ldsa_kernel(W4Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half8* lds_a = reinterpret_cast<half8*>(lds_raw);          // [2][MT·4][64] half8
    constexpr int FR = MT * 256;                                // 16-byte fragments per group
    constexpr int NT_ = NW * 64;
    constexpr int ALD = (FR + NT_ - 1) / NT_;                   // fragment loads per thread per group
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int a = lane >> 4, b = lane & 15;
    const int st_raw = blockIdx.x * NW + wave;
    const bool st_ok = st_raw < p.n64;
    const int st = st_ok ? st_raw : p.n64 - 1;
    const int rb = blockIdx.y, z = blockIdx.z;
    const int g0 = (int)((long)p.G * z / p.S), g1 = (int)((long)p.G * (z + 1) / p.S);
    FH_TL(0);

    // cooperative A loads: fragment idx ↔ (mt, k-step s, lane l): row mt·16 + (l & 15), k = 32 s + 8 (l >> 4)
    const __half* asrc[ALD];
#pragma unroll
    for (int i = 0; i < ALD; i++) {
        const int idx = min(threadIdx.x + i * NT_, FR - 1);
        const int mt = idx >> 8, s = (idx >> 6) & 3, l = idx & 63;
        const int r = rb * 16 * MT + mt * 16 + (l & 15);
        asrc[i] = p.x + (long)(r < p.M ? r : p.M - 1) * p.K + 32 * s + 8 * (l >> 4);
    }
    typedef uint32_t u32x4g __attribute__((ext_vector_type(4)));
    const u32x4g* qw_lane = reinterpret_cast<const u32x4g*>(p.qw) + ((long)st * p.G * 4) * 64 + lane;
    const uint2* sc_lane = reinterpret_cast<const uint2*>(p.sc) + ((long)st * p.G) * 16 + b;
    const uint2* zp_lane = HAS_ZP ? reinterpret_cast<const uint2*>(p.zp) + ((long)st * p.G) * 16 + b : nullptr;

    float4v acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) acc[mt][nt] = (float4v){0.f, 0.f, 0.f, 0.f};

    // Rings of depth 4 (static slots through a 4× unrolled loop): weights of group g+3 and activations of
    // group g+3 are requested while group g is consumed, so one workgroup alone keeps ≈3 groups × NW × 4 KiB
    // in flight (a CU needs ≈48 KB in flight to hold its share of HBM at ≈2 µs latency).
    constexpr int D = 4;
    u32x4g wq[D][4];
    uint2 scv[D], zpv[D];
    half8 areg[D][ALD];
    auto issue_w = [&](int slot, int g) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) wq[slot][nt] = __builtin_nontemporal_load(qw_lane + ((long)g * 4 + nt) * 64);
        scv[slot] = sc_lane[(long)g * 16];
        if (HAS_ZP) zpv[slot] = zp_lane[(long)g * 16];
    };
    auto issue_a = [&](int slot, int g) {
#pragma unroll
        for (int i = 0; i < ALD; i++) areg[slot][i] = *reinterpret_cast<const half8*>(asrc[i] + (long)g * 128);
    };
    auto store_a = [&](int slot, int buf) {
#pragma unroll
        for (int i = 0; i < ALD; i++) {
            const int idx = threadIdx.x + i * NT_;
            if (idx < FR) lds_a[buf * FR + idx] = areg[slot][i];
        }
    };
    auto consume = [&](int slot, int abuf) {
        half8 af[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int s = 0; s < 4; s++) af[mt][s] = lds_a[abuf * FR + (mt * 4 + s) * 64 + lane];
        const unsigned long long sb = ((unsigned long long)scv[slot].y << 32) | scv[slot].x;
        const unsigned long long zb = HAS_ZP ? (((unsigned long long)zpv[slot].y << 32) | zpv[slot].x) : 0ull;
        w4_consume_group<MT, 4, HAS_ZP>(wq[slot], sb, zb, 0, af, acc);
    };
#define FH_PIN() __builtin_amdgcn_sched_barrier(0)
    // No load sits under a runtime condition in the steady loop (a conditional load makes hipcc wait vmcnt(0) at
    // the join and every group then pays the full memory latency): prefetch indices are CLAMPED to the last group
    // instead — the few repeated requests at the tail hit L2.
    if (g0 < g1) {
        const int gl = g1 - 1;
#pragma unroll
        for (int d = 0; d < D - 1; d++) { issue_w(d, min(g0 + d, gl)); issue_a(d, min(g0 + d, gl)); }
        FH_PIN();
        store_a(0, 0);                                  // A(g0) → LDS buffer 0
        __syncthreads();
        FH_TL(1);
        int gb = g0;
        for (; gb + D <= g1; gb += D) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                const int g = gb + d;
                issue_w((d + D - 1) % D, min(g + D - 1, gl));
                issue_a((d + D - 1) % D, min(g + D - 1, gl));
                store_a((d + 1) % D, (d + 1) & 1);      // A(g+1), requested two groups ago
                FH_PIN();
                consume(d, d & 1);
                FH_PIN();
                __syncthreads();
            }
        }
        // tail: ≤ 3 groups, already requested (slots 0..2 relative to gb)
#pragma unroll
        for (int d = 0; d < D - 1; d++) {
            if (gb + d < g1) {
                if (d + 1 < D - 1) store_a(d + 1, (d + 1) & 1);
                consume(d, d & 1);
                __syncthreads();
            }
        }
    }
#undef FH_PIN
    FH_TL(2);
    if (!st_ok) return;
    if (p.partial) {
        float* slab = p.partial + (long)z * p.rows_pad * p.n_pad;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = rb * 16 * MT + mt * 16 + 4 * a + r;
#pragma unroll
                for (int nt = 0; nt < 4; nt++) slab[(long)row * p.n_pad + st * 64 + nt * 16 + b] = acc[mt][nt][r];
            }
#ifdef FERRUM_HIP_EXPERIMENTS
        if (p.tl) { __builtin_amdgcn_s_waitcnt(0); FH_TL(3); }
#endif
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = rb * 16 * MT + mt * 16 + 4 * a + r;
            if (row >= p.M) continue;
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const int col = st * 64 + nt * 16 + b;
                if (col < p.N) {
                    float v = acc[mt][nt][r];
                    if (p.bias) v += __half2float(p.bias[col]);
                    p.out[(long)row * p.ldo + col] = __float2half(v);
                }
            }
        }
}

#ifdef FERRUM_HIP_EXPERIMENTS
#include "experiments_w4_decode_forms.inc"
#endif

// ── MoE prefill grouped GEMM: 64-row tiles, activations through LDS, 4 waves × 64 columns ──────────────────────
// For blocks of 64 sorted pairs the skinny kernels re-fetch their A fragments from L2 for every 64-column supertile and
// spend half their time on it.  Here a workgroup owns 64 rows × 256 columns: every 128-k group's 64×128 activation tile is
// gathered once (pair rows through sorted_token_ids) into a double-buffered fragment-major LDS tile, each wave streams
// the INT4 group of ITS 64-column supertile straight to registers, expands it once and runs it against all four 16-row
// tiles (80 MFMAs per 4 KiB of weights instead of 20).  MODE 1 plain, 2 gate_up with the fused silu·mul epilogue.  Same
// arithmetic as w4_consume_group (exact integer-valued products, fp32 scale), restructured k-pair-major to hold 250 VGPRs.
// (The dense M ≥ 64 GEMM uses the hand-pipelined w4_gemm_tilep_kernel below.)
// MTN = 16-row tiles per block: 4 (64-pair align blocks, ≥ 32 pairs per expert) or 2 (32-pair blocks: a few hundred
// tokens, e.g. a fresh prompt riding along with the decode batch — half the MFMA padding and the LDS traffic per block).
template <bool HAS_ZP, int MODE, int MTN = 4>
__global__ __launch_bounds__(256, 2) void w4_gemm_tile_kernel(W4Args p) {
    constexpr int ROWS = 16 * MTN;
    static_assert(MODE == 1 || MODE == 2, "grouped-GEMM modes only");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half8* lds_a = reinterpret_cast<half8*>(lds_raw);          // [2][MTN mt][4 s][64] half8 = 2 × MTN·4 KiB
    constexpr int FR = MTN * 256;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int a = lane >> 4, b = lane & 15;
    // XCD-aware tile order: workgroups are dealt to the 8 XCDs round-robin by launch index; here each XCD walks a contiguous
    // range of (column tile, row block) with the row block fastest, so the row blocks of one expert meet its weight slice in
    // ONE XCD's L2 instead of four
    const int nwg = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, virt = xcd * (nwg >> 3) + min(xcd, nwg & 7) + (lin >> 3);      // XCD x owns nwg/8 (+1 if x < nwg%8) tiles
    const int st_raw = (virt / (int)gridDim.y) * 4 + wave;
    const bool st_ok = st_raw < p.n64;
    const int st = st_ok ? st_raw : p.n64 - 1;
    const int rb = virt % (int)gridDim.y;

    const uint32_t* qw = p.qw;
    const __half* sc = p.sc;
    const __half* zp = p.zp;
    // rows of this tile: fragment loads are split by wave (k-step s = wave), so every thread needs the 4 rows
    // mt·16 + b of its lane; the epilogue needs rows mt·16 + 4a + r
    int row_in[MTN], row_out_l[MTN];
    {
        // (independent reads, in bounds for every launched row block: requested together — one round trip, not three)
        const int total = *p.total_post_pad;
        const int e = p.block_ids[rb];
        int ids[MTN];
#pragma unroll
        for (int mt = 0; mt < MTN; mt++) ids[mt] = p.sorted_token_ids[rb * ROWS + mt * 16 + b];
        if (rb * ROWS >= total) return;
        qw += (long)e * p.expert_stride_qw;
        sc += (long)e * p.expert_stride_sc;
        if (HAS_ZP) zp += (long)e * p.expert_stride_sc;
#pragma unroll
        for (int mt = 0; mt < MTN; mt++) {
            const int id = ids[mt];
            row_out_l[mt] = id < p.M ? id : -1;
            row_in[mt] = id < p.M ? id / p.top_k : 0;
        }
    }
    const __half* asrc[MTN];
#pragma unroll
    for (int mt = 0; mt < MTN; mt++) asrc[mt] = p.x + (long)row_in[mt] * p.K + 32 * wave + 8 * a;

    typedef uint32_t u32x4g __attribute__((ext_vector_type(4)));
    const u32x4g* qw_lane = reinterpret_cast<const u32x4g*>(qw) + ((long)st * p.G * 4) * 64 + lane;
    const uint2* sc_lane = reinterpret_cast<const uint2*>(sc) + ((long)st * p.G) * 16 + b;
    const uint2* zp_lane = HAS_ZP ? reinterpret_cast<const uint2*>(zp) + ((long)st * p.G) * 16 + b : nullptr;

    float4v acc[MTN][4];
#pragma unroll
    for (int mt = 0; mt < MTN; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) acc[mt][nt] = (float4v){0.f, 0.f, 0.f, 0.f};

    u32x4g wq[2][4];
    uint2 scv[2], zpv[2];
    half8 areg[MTN];
    auto issue_w = [&](int slot, int g) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) wq[slot][nt] = __builtin_nontemporal_load(qw_lane + ((long)g * 4 + nt) * 64);
        scv[slot] = sc_lane[(long)g * 16];
        if (HAS_ZP) zpv[slot] = zp_lane[(long)g * 16];
    };
    auto issue_a = [&](int g) {
#pragma unroll
        for (int mt = 0; mt < MTN; mt++) areg[mt] = *reinterpret_cast<const half8*>(asrc[mt] + (long)g * 128);
    };
    auto store_a = [&](int buf) {   // fragment (mt, s = wave, lane)
#pragma unroll
        for (int mt = 0; mt < MTN; mt++) lds_a[buf * FR + (mt * 4 + wave) * 64 + lane] = areg[mt];
    };
    // low nibbles land in mantissa bits 0-3 of 0x6400 (1024 + n), high nibbles in bits 4-7 of 0x5400 (64 + n): both
    // expansions are integer-valued, so no operand needs rescaling (the vector issue port is what bounds this kernel)
    const uint32_t magic = opaque_vgpr(0x64006400u), magic_hi = opaque_vgpr(0x54005400u);
    const uint32_t m_lo = opaque_sgpr(0x000F000Fu), m_hi = opaque_sgpr(0x00F000F0u);
    const half8 b_lo = splat_half8(HAS_ZP ? -1024.0f : -1032.0f), b_hi = splat_half8(HAS_ZP ? -64.0f : -72.0f);
    const half8 ones = splat_half8(1.0f);
    auto half_at = [](unsigned long long bits, int i) {
        union { uint16_t u; _Float16 h; } c;
        c.u = (uint16_t)(bits >> (16 * i));
        return c.h;
    };
    auto consume = [&](int slot, int abuf) {
        const half8* at = lds_a + abuf * FR;
        // expand the group once (16 B-operand fragments), then one 16-row tile at a time: 4 offset MFMAs seed the
        // tile's 4 chains, 16 MFMAs extend them, the fp32 scale folds them into acc
        half8 lo[2][4], hi[2][4];
#pragma unroll
        for (int pr = 0; pr < 2; pr++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const uint32_t d0 = wq[slot][nt][2 * pr], d1 = wq[slot][nt][2 * pr + 1];
                const uint32_t d0s = d0 >> 8, d1s = d1 >> 8;
                lo[pr][nt] = pack_half8(and_or(d0, m_lo, magic), and_or(d0s, m_lo, magic), and_or(d1, m_lo, magic), and_or(d1s, m_lo, magic));
                hi[pr][nt] = pack_half8(and_or(d0, m_hi, magic_hi), and_or(d0s, m_hi, magic_hi), and_or(d1, m_hi, magic_hi), and_or(d1s, m_hi, magic_hi));
            }
        const unsigned long long sb = ((unsigned long long)scv[slot].y << 32) | scv[slot].x;
        const unsigned long long zb = HAS_ZP ? (((unsigned long long)zpv[slot].y << 32) | zpv[slot].x) : 0ull;
#pragma unroll
        for (int mt = 0; mt < MTN; mt++) {
            half8 af[4];
#pragma unroll
            for (int s = 0; s < 4; s++) af[s] = at[(mt * 4 + s) * 64 + lane];
            float4v t = {0.f, 0.f, 0.f, 0.f}, u = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; s++) {
                t = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[s], (s & 1) ? b_hi : b_lo, t, 0, 0, 0);
                if (HAS_ZP) u = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[s], ones, u, 0, 0, 0);
            }
            float4v tmp[4];
#pragma unroll
            for (int nt = 0; nt < 4; nt++) tmp[nt] = t;
#pragma unroll
            for (int pr = 0; pr < 2; pr++) {
#pragma unroll
                for (int nt = 0; nt < 4; nt++) tmp[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[2 * pr], lo[pr][nt], tmp[nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < 4; nt++) tmp[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[2 * pr + 1], hi[pr][nt], tmp[nt], 0, 0, 0);
            }
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const float s_f = (float)half_at(sb, nt);                  // packed fp32 FMAs: two accumulators per issue slot
                const float4v s4 = {s_f, s_f, s_f, s_f};
                if (HAS_ZP) {
                    const float z_f = (float)half_at(zb, nt);
                    const float4v nz4 = {-z_f, -z_f, -z_f, -z_f};
                    acc[mt][nt] = __builtin_elementwise_fma(s4, __builtin_elementwise_fma(nz4, u, tmp[nt]), acc[mt][nt]);
                } else {
                    acc[mt][nt] = __builtin_elementwise_fma(s4, tmp[nt], acc[mt][nt]);
                }
            }
        }
    };
#define FH_PIN() __builtin_amdgcn_sched_barrier(0)
    const int gz0 = 0, gz1 = p.G;
    {
        const int gl = gz1 - 1;
        issue_w(0, gz0);
        issue_a(gz0);
        FH_PIN();
        store_a(0);
        issue_a(min(gz0 + 1, gl));
        __syncthreads();
        int g = gz0;
        for (; g + 2 <= gz1; g += 2) {              // two groups per trip: static buffer indices, no conditional loads
            issue_w(1, min(g + 1, gl));
            store_a(1);                              // A(g+1), requested one group ago
            issue_a(min(g + 2, gl));
            FH_PIN();
            consume(0, 0);
            FH_PIN();
            __syncthreads();
            issue_w(0, min(g + 2, gl));
            store_a(0);                              // A(g+2)
            issue_a(min(g + 3, gl));
            FH_PIN();
            consume(1, 1);
            FH_PIN();
            __syncthreads();
        }
        if (g < gz1) consume(0, 0);                  // odd group count: the last group is already staged
    }
#undef FH_PIN
    if (!st_ok) return;
    // epilogue: D row 4a + r of tile mt ↔ tile row mt·16 + 4a + r, whose routing lives in lane (·, b = 4a + r)
#pragma unroll
    for (int mt = 0; mt < MTN; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int orow = __shfl(row_out_l[mt], 4 * a + r, 64);
            if (orow < 0) continue;
            if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const float gt = acc[mt][j][r], up = acc[mt][2 + j][r];
                    const float v = (gt / (1.0f + __expf(-gt))) * up;
                    const int col = st * 32 + j * 16 + b;
                    if (col < p.ldo) p.out[(long)orow * p.ldo + col] = __float2half(v);
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
                    const int col = st * 64 + nt * 16 + b;
                    if (col < p.N) {
                        float v = acc[mt][nt][r];
                        p.out[(long)orow * p.ldo + col] = __float2half(v);
                    }
                }
            }
        }
}

// ── dense prefill GEMM (M ≥ 64): 64-row tiles, software-pipelined by hand ──────────────────────────────────────────
// Same tile as the MoE kernel above (64 rows × 256 columns per workgroup, activations through a double-buffered
// fragment-major LDS tile, each wave streaming the INT4 group of its own 64-column supertile), but every phase of a wave is
// interleaved with its MFMA stream explicitly instead of running as separate phases (SQ_VALU_MFMA_BUSY_CYCLES of the phased
// kernel: ≈ 45 % of wall):
//   * a quant group is consumed as two k-halves; while the MFMAs of one half run, the OTHER half's B operands are expanded,
//     two VALU behind each MFMA (an MFMA holds the SIMD's vector issue for 8 of its 16 cycles);
//   * one "step" = one 16-row tile × one k-half (2 offset + 8 product MFMAs).  The A fragments of step i+1 are read from LDS
//     during step i; the chain result of step i−1 is folded into the accumulators (fp32 FMA with the group scale) just
//     before step i overwrites the chain registers;
//   * activation staging (global → registers → LDS, one group ahead) and the weight loads (two groups ahead) are issued one
//     instruction per step; every global address is a scalar base + a 32-bit lane offset (no 64-bit address registers —
//     a spill inside this loop costs a vmcnt(0) drain of the prefetches).
// Same arithmetic as w4_consume_group except that the fp32 scale is applied per k-half (two roundings per group instead of
// one).  Measured against the phased kernel (tools/exp_prefill_gemm.py): 4096→28672 at M = 8192 745 → 937 TFLOP/s,
// 4096→6144 at M = 2048 746 → 806, 2048→5120 at M = 2048 636 → 676; M = 64…512 +3…8 %.  A 128-row variant (one wave per
// SIMD, 128 accumulators) was tried and rejected: the fp32 fold needs the accumulators in arch VGPRs, the compiler shuttles
// them through AGPRs (≈ 840 v_accvgpr moves per two groups) and it ran at 264–471 TFLOP/s.  The MoE modes gain nothing
// from this schedule at K = 2048 / 768 (158.6 → 157.1 µs, 101.2 → 103.4 µs at 2048 tokens: padding and the short K bound
// them), so the grouped GEMM keeps the phased kernel.
template <bool HAS_ZP>
__global__ __launch_bounds__(256, 2) void w4_gemm_tilep_kernel(W4Args p) {
    constexpr int MTN = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half8* lds_a = reinterpret_cast<half8*>(lds_raw);          // [2][MTN mt][4 s][64] half8 = 2 × MTN·4 KiB
    constexpr int FR = MTN * 256, ROWS = MTN * 16;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave-uniform: scalar bases
    const int a = lane >> 4, b = lane & 15;
    const int st_raw = blockIdx.x * 4 + wave;
    const bool st_ok = st_raw < p.n64;
    const int st = st_ok ? st_raw : p.n64 - 1;
    const int rb = blockIdx.y;

    const uint32_t* qw = p.qw;
    const __half* sc = p.sc;
    const __half* zp = p.zp;
    uint32_t aoff[MTN];                                        // BYTE offset of this thread's staging rows (mt·16 + b): scalar base + 32-bit lane offset
#pragma unroll
    for (int mt = 0; mt < MTN; mt++) {
        const int r = rb * ROWS + mt * 16 + b;
        aoff[mt] = ((uint32_t)(r < p.M ? r : p.M - 1) * (uint32_t)p.K + 32 * wave + 8 * a) * 2u;
    }
    typedef uint32_t u32x4g __attribute__((ext_vector_type(4)));
    const char* qw_wave = reinterpret_cast<const char*>(qw) + ((long)st * p.G * 4) * 64 * 16;     // scalar
    const char* sc_wave = reinterpret_cast<const char*>(sc) + ((long)st * p.G) * 16 * 8;
    const char* zp_wave = HAS_ZP ? reinterpret_cast<const char*>(zp) + ((long)st * p.G) * 16 * 8 : nullptr;
    const char* x_base = reinterpret_cast<const char*>(p.x);
    const uint32_t lane16 = lane * 16, b8 = b * 8;

    float4v acc[MTN][4];
#pragma unroll
    for (int mt = 0; mt < MTN; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) acc[mt][nt] = (float4v){0.f, 0.f, 0.f, 0.f};
    u32x4g wq[2][4];                 // raw words, groups of parity 0 / 1
    uint2 scv[2], zpv[2];
    half8 areg[MTN];                  // activation staging
    u32x4g bw[2][4][2];              // expanded B operands [k-half][column tile][lo, hi]
    half8 af[2][2];                  // A fragments of the current / next step
    float4v tmp[4], usum[2];         // chain results: folded into acc just before the next step overwrites them
    float sf[2][4], zf[2][4];

    const uint32_t magic = opaque_vgpr(0x64006400u), magic_hi = opaque_vgpr(0x54005400u);
    const uint32_t m_lo = opaque_sgpr(0x000F000Fu), m_hi = opaque_sgpr(0x00F000F0u);
    const half8 b_lo = splat_half8(HAS_ZP ? -1024.0f : -1032.0f), b_hi = splat_half8(HAS_ZP ? -64.0f : -72.0f);
    const half8 ones = splat_half8(1.0f);
    const float4v zero4 = {0.f, 0.f, 0.f, 0.f};
    auto half_at = [](uint2 v, int i) {
        union { uint16_t u; _Float16 h; } c;
        c.u = (uint16_t)((i < 2 ? v.x : v.y) >> (16 * (i & 1)));
        return c.h;
    };
    const int gz0 = (int)((long)p.G * blockIdx.z / p.S), gz1 = (int)((long)p.G * (blockIdx.z + 1) / p.S);
    const int gl = gz1 - 1;
    auto issue_w = [&](int sl, int nt, int g) __attribute__((always_inline)) {
        wq[sl][nt] = *reinterpret_cast<const u32x4g*>(qw_wave + ((uint32_t)g * 4096u + lane16) + nt * 1024);   // scalar base + 32-bit offset
    };
    auto issue_s = [&](int sl, int g) __attribute__((always_inline)) {
        scv[sl] = *reinterpret_cast<const uint2*>(sc_wave + ((uint32_t)g * 128u + b8));
        if (HAS_ZP) zpv[sl] = *reinterpret_cast<const uint2*>(zp_wave + ((uint32_t)g * 128u + b8));
    };
    auto issue_a = [&](int mt, int g) __attribute__((always_inline)) {
        areg[mt] = *reinterpret_cast<const half8*>(x_base + (aoff[mt] + (uint32_t)g * 256u));
    };
    // expansion of column tile nt of k-half pr_t from words (2·pr_t, 2·pr_t + 1) of wq[sl_src][nt]: lo (6 VALU) or hi (6)
    auto expand = [&](int pr_t, int sl_src, int nt, int part, int piece) __attribute__((always_inline)) {
        const uint32_t d0 = wq[sl_src][nt][2 * pr_t], d1 = wq[sl_src][nt][2 * pr_t + 1];
        const uint32_t ml = part == 0 ? m_lo : m_hi, mg = part == 0 ? magic : magic_hi;
        u32x4g& dst = bw[pr_t][nt][part];
        if (piece == 0) { dst[0] = and_or(d0, ml, mg); dst[1] = and_or(d0 >> 8, ml, mg); }
        else { dst[2] = and_or(d1, ml, mg); dst[3] = and_or(d1 >> 8, ml, mg); }
    };
#define FH_PIN() __builtin_amdgcn_sched_barrier(0)
    // fold the chain result of the previous step (row tile pm, scales of group parity psl) into the accumulators
    auto fold = [&](int pp, int pm, int psl, int nt) __attribute__((always_inline)) {   // pp: parity of usum
        const float s_f = sf[psl][nt];
        const float4v s4 = {s_f, s_f, s_f, s_f};
        if (HAS_ZP) {
            const float z_f = zf[psl][nt];
            const float4v nz4 = {-z_f, -z_f, -z_f, -z_f};
            acc[pm][nt] = __builtin_elementwise_fma(s4, __builtin_elementwise_fma(nz4, usum[pp], tmp[nt]), acc[pm][nt]);
        } else {
            acc[pm][nt] = __builtin_elementwise_fma(s4, tmp[nt], acc[pm][nt]);
        }
    };
    // one step: row tile mt × k-half pr of group g (register/LDS parity sl)
    auto step = [&](int pr, int mt, int sl, int g) __attribute__((always_inline)) {
        const int cp = mt & 1, np = cp ^ 1;
        const half8* at = lds_a + sl * FR;
        // next step's A fragments (after the group's barrier when they belong to the next group)
        if (pr == 1 && mt == MTN - 1) {
            __syncthreads();
            const half8* atn = lds_a + (sl ^ 1) * FR;
            af[np][0] = atn[lane];
            af[np][1] = atn[64 + lane];
        } else {
            const int nmt = (mt + 1) % MTN, npr = mt == MTN - 1 ? 1 : pr;
            af[np][0] = at[(nmt * 4 + 2 * npr) * 64 + lane];
            af[np][1] = at[(nmt * 4 + 2 * npr + 1) * 64 + lane];
        }
        if (pr == 0) {                       // activations: A(g+1) → LDS, A(g+2) requested
            lds_a[(sl ^ 1) * FR + (mt * 4 + wave) * 64 + lane] = areg[mt];
            issue_a(mt, min(g + 2, gl));
        } else if (mt < 4) {                 // weights of group g+2 into the words this group has consumed
            issue_w(sl, mt, min(g + 2, gl));
        }
        if (pr == 0 && mt == 0) {            // this group's scales; their words are then free for group g+2
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                sf[sl][nt] = (float)half_at(scv[sl], nt);
                if (HAS_ZP) zf[sl][nt] = (float)half_at(zpv[sl], nt);
            }
        }
        if (pr == 0 && mt == 1) issue_s(sl, min(g + 2, gl));
        float4v t = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cp][0], b_lo, zero4, 0, 0, 0);
        FH_PIN();
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cp][1], b_hi, t, 0, 0, 0);
        FH_PIN();
        if (HAS_ZP) {
            float4v u1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cp][0], ones, zero4, 0, 0, 0);
            usum[cp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cp][1], ones, u1, 0, 0, 0);
            FH_PIN();
        }
        const int pm = (mt + MTN - 1) % MTN, psl = (pr == 0 && mt == 0) ? (sl ^ 1) : sl;
        const int xpr = pr ^ 1, xsl = pr == 0 ? sl : (sl ^ 1);
        constexpr int PPS = 16 / MTN;          // expansion pieces (2 VALU each) per step
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            fold(np, pm, psl, nt);
            tmp[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cp][0], __builtin_bit_cast(half8, bw[pr][nt][0]), t, 0, 0, 0);
            FH_PIN();
        }
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            if (nt < PPS) { const int pc = mt * PPS + nt; expand(xpr, xsl, pc >> 2, (pc >> 1) & 1, pc & 1); }
            tmp[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cp][1], __builtin_bit_cast(half8, bw[pr][nt][1]), tmp[nt], 0, 0, 0);
            FH_PIN();
        }
    };
    auto group = [&](int sl, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int pr = 0; pr < 2; pr++)
#pragma unroll
            for (int mt = 0; mt < MTN; mt++) step(pr, mt, sl, g);
    };
    // prologue: weights of the first two groups, A(gz0) staged, B operands of (gz0, k-half 0), an empty "previous step"
#pragma unroll
    for (int nt = 0; nt < 4; nt++) { issue_w(0, nt, gz0); issue_w(1, nt, min(gz0 + 1, gl)); }
    issue_s(0, gz0);
    issue_s(1, min(gz0 + 1, gl));
#pragma unroll
    for (int mt = 0; mt < MTN; mt++) issue_a(mt, gz0);
#pragma unroll
    for (int mt = 0; mt < MTN; mt++) lds_a[(mt * 4 + wave) * 64 + lane] = areg[mt];
#pragma unroll
    for (int mt = 0; mt < MTN; mt++) issue_a(mt, min(gz0 + 1, gl));
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
        expand(0, 0, nt, 0, 0); expand(0, 0, nt, 0, 1); expand(0, 0, nt, 1, 0); expand(0, 0, nt, 1, 1);
        tmp[nt] = zero4;
        sf[1][nt] = 0.f;
        zf[1][nt] = 0.f;
    }
    usum[1] = zero4;
    __syncthreads();
    af[0][0] = lds_a[lane];
    af[0][1] = lds_a[64 + lane];
    FH_PIN();
    int g = gz0;
    for (; g + 2 <= gz1; g += 2) {
        group(0, g);
        group(1, g + 1);
    }
    int lsl = 1;
    if (g < gz1) { group(0, g); lsl = 0; }
#undef FH_PIN
    // the last step's chain result is still pending
    if (lsl == 0) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) fold(1, MTN - 1, 0, nt);
    } else {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) fold(1, MTN - 1, 1, nt);
    }
    if (!st_ok) return;
    if (p.partial) {                                 // split-K: fp32 slab, summed by the reduce launch
        float* slab = p.partial + (long)blockIdx.z * p.rows_pad * p.n_pad;
#pragma unroll
        for (int mt = 0; mt < MTN; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = rb * ROWS + mt * 16 + 4 * a + r;
#pragma unroll
                for (int nt = 0; nt < 4; nt++) slab[(long)row * p.n_pad + st * 64 + nt * 16 + b] = acc[mt][nt][r];
            }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MTN; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = rb * ROWS + mt * 16 + 4 * a + r;
            if (row >= p.M) continue;
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const int col = st * 64 + nt * 16 + b;
                if (col < p.N) {
                    float v = acc[mt][nt][r];
                    if (p.bias) v += __half2float(p.bias[col]);
                    p.out[(long)row * p.ldo + col] = __float2half(v);
                }
            }
        }
}

// ── prefill GEMM, 128- or 256-row tiles: group scale and zero folded into the B operand ───────────────────────────────
// The 64-row kernels above are bound by vector issue, not by the matrix pipe: every 4-KiB weight group costs a wave ≈ 80
// VALU of nibble expansion plus one fp32 FMA per accumulator (the per-group scale is applied to a chain result), and the
// offset removal adds 2 MFMAs to every 8 — all of it amortised over only four row tiles.  Here the expanded nibbles are
// finished as fp16 weights in registers — (1024 + q) − (1024 + zero) is an exact fp16 integer (v_pk_add_f16), × the group
// scale is one rounding (v_pk_mul_f16): the fp16 weight the reference's own dequantisation produces — so the MFMAs accumulate
// straight into their final registers: no chain results, no fold, no offset MFMAs, and the MT·16 accumulators of a wave
// can live in AGPRs.  With MT = 8 (128 rows) a group costs a wave ≈ 210 VALU against 128 MFMAs, with MT = 16 against 256:
// the matrix pipe (16 cycles per MFMA, 8 of them holding the issue port) is the bound.  One wave per SIMD; the next
// k-half's operands are expanded between the MFMAs of the current one; activations global → registers → LDS one group ahead,
// weights two groups ahead.
// MODE 0 dense; 1 / 2 the MoE grouped GEMM over MT·16-row align blocks (rows gathered through sorted_token_ids, the block's
// expert from block_ids; 2 = gate_up with the silu·mul epilogue), tiles dealt to the XCDs like w4_gemm_tile_kernel's.
template <int MT, bool HAS_ZP, int MODE = 0>
__global__ __launch_bounds__(256, MT <= 6 ? 2 : 1) void w4_gemm_big_kernel(W4Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    half8* lds_a = reinterpret_cast<half8*>(lds_raw);          // [2][MT][4 s][64] half8
    constexpr int FR = MT * 256, ROWS = MT * 16;
    typedef _Float16 half2v_ __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4g __attribute__((ext_vector_type(4)));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int a = lane >> 4, b = lane & 15;
    int ct = blockIdx.x, rb = blockIdx.y;
    if (MODE != 0) {        // each XCD walks a contiguous range of (column tile, row block), row block fastest: one expert's blocks meet its weights in one L2
        const int nwg = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * blockIdx.y;
        const int xcd = lin & 7, virt = xcd * (nwg >> 3) + min(xcd, nwg & 7) + (lin >> 3);
        ct = virt / (int)gridDim.y;
        rb = virt % (int)gridDim.y;
    }
    const int st_raw = ct * 4 + wave;
    const bool st_ok = st_raw < p.n64;
    const int st = st_ok ? st_raw : p.n64 - 1;
    const uint32_t* qw = p.qw;
    const __half* sc = p.sc;
    const __half* zp = p.zp;
    uint32_t aoff[MT];                                          // byte offset of this thread's staging rows: scalar base + 32-bit offset
    int row_out_l[MODE != 0 ? MT : 1];                          // MoE: output row (pair id) of tile row mt·16 + b, −1 = padding
    if (MODE != 0) {
        // the three reads a tile starts with are independent and in bounds for every launched row block (padding blocks hold
        // sentinels): request them together, then decide — one memory round trip instead of three before the first A row
        const int total = *p.total_post_pad;
        const int e = p.block_ids[rb];
        int ids[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) ids[mt] = p.sorted_token_ids[rb * ROWS + mt * 16 + b];
        if (rb * ROWS >= total) return;
        qw += (long)e * p.expert_stride_qw;
        sc += (long)e * p.expert_stride_sc;
        if (HAS_ZP) zp += (long)e * p.expert_stride_sc;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int id = ids[mt];
            row_out_l[mt] = id < p.M ? id : -1;
            aoff[mt] = ((uint32_t)(id < p.M ? id / p.top_k : 0) * (uint32_t)p.K + 32 * wave + 8 * a) * 2u;
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int r = rb * ROWS + mt * 16 + b;
            aoff[mt] = ((uint32_t)(r < p.M ? r : p.M - 1) * (uint32_t)p.K + 32 * wave + 8 * a) * 2u;
        }
    }
    // MoE: the valid rows of an align block are a prefix; 16-row tiles that hold only padding are skipped (their MFMAs, not the
    // side work riding behind them): at ≈ 512 pairs per expert 96-row blocks carry ≈ 19 % padding rows, 16-row granularity ≈ 2 %
    int mt_valid = MT;
    if (MODE != 0) {
        mt_valid = 0;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) mt_valid += __ballot(row_out_l[mt] >= 0) != 0 ? 1 : 0;
        mt_valid = __builtin_amdgcn_readfirstlane(mt_valid);
    }
    const char* qw_wave = reinterpret_cast<const char*>(qw) + ((long)st * p.G * 4) * 64 * 16;
    const char* sc_wave = reinterpret_cast<const char*>(sc) + ((long)st * p.G) * 16 * 8;
    const char* zp_wave = HAS_ZP ? reinterpret_cast<const char*>(zp) + ((long)st * p.G) * 16 * 8 : nullptr;
    const char* x_base = reinterpret_cast<const char*>(p.x);
    const uint32_t lane16 = lane * 16, b8 = b * 8;

    float4v acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) acc[mt][nt] = (float4v){0.f, 0.f, 0.f, 0.f};
    u32x4g wq[2][4];
    uint2 scv[2], zpv[2];
    half8 areg[MT];
    u32x4g bq[2][4];                                            // finished fp16 B operands [k-step parity][column tile]
    const uint32_t magic = opaque_vgpr(0x64006400u), magic_hi = opaque_vgpr(0x54005400u);
    const uint32_t m_lo = opaque_sgpr(0x000F000Fu), m_hi = opaque_sgpr(0x00F000F0u);
    const int gz0 = (int)((long)p.G * blockIdx.z / p.S), gz1 = (int)((long)p.G * (blockIdx.z + 1) / p.S);
    const int gl = gz1 - 1;
    auto issue_w = [&](int sl, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
            wq[sl][nt] = *reinterpret_cast<const u32x4g*>(qw_wave + ((uint32_t)g * 4096u + lane16) + nt * 1024);
        scv[sl] = *reinterpret_cast<const uint2*>(sc_wave + ((uint32_t)g * 128u + b8));
        if (HAS_ZP) zpv[sl] = *reinterpret_cast<const uint2*>(zp_wave + ((uint32_t)g * 128u + b8));
    };
    auto issue_a = [&](int mt, int g) __attribute__((always_inline)) {
        areg[mt] = *reinterpret_cast<const half8*>(x_base + (aoff[mt] + (uint32_t)g * 256u));
    };
    auto h2 = [](uint2 v, int i) __attribute__((always_inline)) {   // packed half i of four → both lanes of a half2
        const uint32_t w = i < 2 ? v.x : v.y;
        const uint32_t h = (i & 1) ? (w >> 16) : (w & 0xFFFFu);
        return __builtin_bit_cast(half2v_, h | (h << 16));
    };
    // B operands of column tile nt, k-half pr of the group in slot sl — part 0: the lo nibbles (k-step 2·pr), part 1: the hi
    // nibbles (k-step 2·pr + 1): expand (5 VALU per 8 nibbles), remove the offset (exact), scale (one rounding).  Cut into eight
    // micro-steps of ≤ 2 VALU so that each rides behind one MFMA (an MFMA holds the issue port for 8 of its 16 cycles).
    half2v_ s2v[2][4], zlo[2][4], zhi[2][4];
    const uint32_t off_lo = opaque_vgpr(0xE408E408u), off_hi = opaque_vgpr(0xD480D480u);   // −1032, −72 as half2 in VGPRs (an SGPR operand
                                                                                           // with op_sel costs a wait state before the multiply)
    auto prep_scales = [&](int sl) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            s2v[sl][nt] = h2(scv[sl], nt);
            if (HAS_ZP) {
                const half2v_ z2 = h2(zpv[sl], nt);
                zlo[sl][nt] = (half2v_){(_Float16)-1024.0f, (_Float16)-1024.0f} - z2;      // exact: integers ≤ 2048
                zhi[sl][nt] = (half2v_){(_Float16)-64.0f, (_Float16)-64.0f} - z2;
            }
        }
    };
    auto micro = [&](int pr, int sl, int nt, int part, int step, uint32_t (&tt)[4]) __attribute__((always_inline)) {
        const uint32_t d0 = wq[sl][nt][2 * pr], d1 = wq[sl][nt][2 * pr + 1];
        const half2v_ s2 = s2v[sl][nt];
        half2v_ off;
        if (HAS_ZP) off = part ? zhi[sl][nt] : zlo[sl][nt];
        else off = __builtin_bit_cast(half2v_, part ? off_hi : off_lo);
        const uint32_t ml = part ? m_hi : m_lo, mg = part ? magic_hi : magic;
        auto addo = [&](uint32_t v) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2v_, v) + off); };
        auto muls = [&](uint32_t v) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2v_, v) * s2); };
        u32x4g& dst = bq[part][nt];             // k-step 2·pr + part: parity = part
        switch (step) {
            case 0: tt[0] = and_or(d0, ml, mg); tt[1] = d0 >> 8; break;
            case 1: tt[0] = addo(tt[0]); tt[1] = and_or(tt[1], ml, mg); break;
            case 2: dst[0] = muls(tt[0]); tt[1] = addo(tt[1]); break;
            case 3: dst[1] = muls(tt[1]); tt[2] = and_or(d1, ml, mg); break;
            case 4: tt[2] = addo(tt[2]); tt[3] = d1 >> 8; break;
            case 5: dst[2] = muls(tt[2]); tt[3] = and_or(tt[3], ml, mg); break;
            case 6: tt[3] = addo(tt[3]); break;
            default: dst[3] = muls(tt[3]); break;
        }
    };
    auto finish = [&](int pr, int sl, int nt, int part) __attribute__((always_inline)) {
        uint32_t tt[4];
#pragma unroll
        for (int s = 0; s < 8; s++) micro(pr, sl, nt, part, s, tt);
    };
#define FH_PIN() __builtin_amdgcn_sched_barrier(0)
#define FH_MFMA(ACC, A, B, T) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
    // one k-step (32 k) of one group: MT × 4 MFMAs, in place in AGPRs (hipcc otherwise picks three-address MFMAs whose results
    // drift through the register file and permutes all MT·16 accumulators back at the loop edge: ≈ 90 v_accvgpr moves per group).
    // Dependent MFMAs are MT·4 issue slots apart, B operands were finished one k-step earlier, A fragments come from ds_read.
    // Side work of k-step s: the four operand parts of k-step s + 1 (32 micro-steps spread over the MT·4 MFMAs); k-steps 0 / 1 also
    // move A(g+1) from its staging registers to LDS and request A(g+2).
    uint32_t tt[4];
    half8 af[2];
    // AT = row tiles that hold real rows (MoE: the valid rows of an align block are a prefix; at ≈ 512 pairs per expert 96-row
    // blocks carry ≈ 19 % padding rows — a block whose tail tiles are all padding runs the 2- or 4-tile body instead)
    auto k_step = [&](auto at_c, int ks, int sl, int g) __attribute__((always_inline)) {
        constexpr int AT = decltype(at_c)::value;
        static_assert(AT % 2 == 0 && AT <= MT, "the A-fragment ping-pong returns to slot 0 at every k-step");
        const half8* at = lds_a + sl * FR;
        constexpr int NM = AT * 4;                   // MFMAs per k-step
        const int nks = (ks + 1) & 3, npr = nks >> 1, nsl = ks == 3 ? (sl ^ 1) : sl;     // the k-step whose operands are built now
        if (ks == 0) af[0] = at[lane];               // (the group's barrier has just been passed)
#pragma unroll
        for (int mt = 0; mt < AT; mt++) {
            const int cp = mt & 1;
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const int q = mt * 4 + nt;
                FH_MFMA(acc[mt][nt], af[cp], bq[ks & 1][nt], tt);
                if (nt == 0) {
                    if (mt + 1 < AT) af[cp ^ 1] = at[((mt + 1) * 4 + ks) * 64 + lane];
                    else if (ks < 3) af[cp ^ 1] = at[(ks + 1) * 64 + lane];          // first fragment of the next k-step
                    if (ks < 2) {                    // staging rows mt' = ks·AT/2 … of this k-step's share
                        constexpr int HM = (AT + 1) / 2;
                        const int smt = ks * HM + mt;
                        if (mt < HM && smt < AT) {
                            lds_a[(sl ^ 1) * FR + (smt * 4 + wave) * 64 + lane] = areg[smt];   // A(g+1) → LDS
                            issue_a(smt, min(g + 2, gl));                                      // A(g+2) requested
                        }
                    }
                }
#pragma unroll
                for (int ms = q * 32 / NM; ms < (q + 1) * 32 / NM; ms++)
                    micro(npr, nsl, ms >> 3, nks & 1, ms & 7, tt);      // part ms/8 = column tile, of k-step nks (lo / hi nibbles)
                FH_PIN();
            }
        }
    };
    auto group = [&](auto at_c, int sl, int g) __attribute__((always_inline)) {
        k_step(at_c, 0, sl, g);
        k_step(at_c, 1, sl, g);
        k_step(at_c, 2, sl, g);
        prep_scales(sl ^ 1);
        issue_w(sl, min(g + 2, gl));            // this slot's words are all expanded: group g+2
        FH_PIN();
        k_step(at_c, 3, sl, g);
        __syncthreads();                        // A(g+1) is in LDS; every wave is done with A(g)
    };
    // prologue
    issue_w(0, gz0);
    issue_w(1, min(gz0 + 1, gl));
#pragma unroll
    for (int mt = 0; mt < MT; mt++) issue_a(mt, gz0);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) lds_a[(mt * 4 + wave) * 64 + lane] = areg[mt];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) issue_a(mt, min(gz0 + 1, gl));
    prep_scales(0);
#pragma unroll
    for (int nt = 0; nt < 4; nt++) finish(0, 0, nt, 0);            // operands of k-step 0 of the first group
    __syncthreads();
    FH_PIN();
    auto run = [&](auto at_c) __attribute__((always_inline)) {
        for (int g = gz0; g < gz1; g += 2) {   // an even number of groups per split (the launcher checks)
            group(at_c, 0, g);
            group(at_c, 1, g + 1);
        }
    };
    run(std::integral_constant<int, MT>{});
#undef FH_PIN
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results before any VALU reads them
    if (!st_ok) return;
    // epilogue: every accumulator leaves its AGPR through an explicit read at its point of use (left to the allocator, all
    // MT·16 of them move to VGPRs at the loop exit and the 256-row form spills 250 registers)
    auto acc_get = [&](int mt, int nt) __attribute__((always_inline)) {
        float4v v;
        asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                     : "a"(acc[mt][nt][0]), "a"(acc[mt][nt][1]), "a"(acc[mt][nt][2]), "a"(acc[mt][nt][3]));
        return v;
    };
    if (MODE != 0) {        // D row 4a + r of tile mt ↔ block row mt·16 + 4a + r, whose pair id lives in lane (·, b = 4a + r)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            int orow[4];
#pragma unroll
            for (int r = 0; r < 4; r++) orow[r] = __shfl(row_out_l[mt], 4 * a + r, 64);
            if (MODE == 2) {
#pragma unroll
                for (int jj = 0; jj < 2; jj++) {
                    const float4v gt = acc_get(mt, jj), up = acc_get(mt, 2 + jj);
                    const int col = st * 32 + jj * 16 + b;
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (orow[r] >= 0 && col < p.ldo)
                            p.out[(long)orow[r] * p.ldo + col] = __float2half((gt[r] / (1.0f + __expf(-gt[r]))) * up[r]);
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
                    const float4v v = acc_get(mt, nt);
                    const int col = st * 64 + nt * 16 + b;
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (orow[r] >= 0 && col < p.N) p.out[(long)orow[r] * p.ldo + col] = __float2half(v[r]);
                }
            }
        }
        return;
    }
    if (p.partial) {
        float* slab = p.partial + (long)blockIdx.z * p.rows_pad * p.n_pad + (long)(rb * ROWS + 4 * a) * p.n_pad + st * 64 + b;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const float4v v = acc_get(mt, nt);
#pragma unroll
                for (int r = 0; r < 4; r++) slab[(long)(mt * 16 + r) * p.n_pad + nt * 16] = v[r];
            }
        return;
    }
    __half* out_lane = p.out + (long)(rb * ROWS + 4 * a) * p.ldo + st * 64 + b;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const float4v v = acc_get(mt, nt);
            const int col = st * 64 + nt * 16 + b;
            const float bias = (p.bias && col < p.N) ? __half2float(p.bias[col]) : 0.f;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = rb * ROWS + mt * 16 + 4 * a + r;
                if (row < p.M && col < p.N) out_lane[(long)(mt * 16 + r) * p.ldo + nt * 16] = __float2half(v[r] + bias);
            }
        }
}

template <int MT, int MODE = 0>
static int launch_big(const W4Args& a, bool has_zp, dim3 grid, hipStream_t stream) {
    const size_t lds = (size_t)2 * MT * 256 * 16;
    if (has_zp) hipLaunchKernelGGL((w4_gemm_big_kernel<MT, true, MODE>), grid, dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((w4_gemm_big_kernel<MT, false, MODE>), grid, dim3(256), lds, stream, a);
    FH_CHECK_LAUNCH();
    return 0;
}

static int launch_tilep(const W4Args& a, bool has_zp, dim3 grid, hipStream_t stream) {
    const size_t lds = (size_t)2 * 1024 * 16;
    if (has_zp) hipLaunchKernelGGL((w4_gemm_tilep_kernel<true>), grid, dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((w4_gemm_tilep_kernel<false>), grid, dim3(256), lds, stream, a);
    FH_CHECK_LAUNCH();
    return 0;
}

template <int MODE, int MTN>
static int launch_tile(const W4Args& a, bool has_zp, dim3 grid, hipStream_t stream) {
    const size_t lds = (size_t)2 * MTN * 256 * 16;
    if (has_zp) hipLaunchKernelGGL((w4_gemm_tile_kernel<true, MODE, MTN>), grid, dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((w4_gemm_tile_kernel<false, MODE, MTN>), grid, dim3(256), lds, stream, a);
    FH_CHECK_LAUNCH();
    return 0;
}

#ifdef FERRUM_HIP_EXPERIMENTS
static int launch_ldsw(const W4Args& a_in, bool has_zp, dim3 grid, hipStream_t stream);
#endif
template <int MT, int NW>
static int launch_ldsa(const W4Args& a_in, bool has_zp, dim3 grid, hipStream_t stream) {
    const size_t lds = (size_t)2 * MT * 256 * 16;
#ifdef FERRUM_HIP_EXPERIMENTS
    W4Args a = a_in;
    a.tl = g_timeline;
#else
    const W4Args& a = a_in;
#endif
#ifdef FERRUM_HIP_EXPERIMENTS
    if constexpr (MT == 2 && NW == 4)
        if (knobs().w4_ldsw && grid.y == 1) return launch_ldsw(a_in, has_zp, grid, stream);
#endif
    if (has_zp) hipLaunchKernelGGL((w4_gemm_ldsa_kernel<MT, NW, true>), grid, dim3(NW * 64), lds, stream, a);
    else hipLaunchKernelGGL((w4_gemm_ldsa_kernel<MT, NW, false>), grid, dim3(NW * 64), lds, stream, a);
    FH_CHECK_LAUNCH();
    return 0;
}

#ifdef FERRUM_HIP_EXPERIMENTS
#include "experiments_w4_decode_launchers.inc"
#endif

template <int MT, int NT>
static int launch_wgsplit(const W4Args& a, bool has_zp, dim3 grid, int W, hipStream_t stream) {
    size_t lds = (size_t)W * MT * NT * 4 * 64 * sizeof(float);
    if constexpr (MT == 1) {
        if (a.fa_down) {                                       // fused combine + add + norm prologue (≤ 4 rows)
            lds += (size_t)a.M * (a.K + 8) * sizeof(__half) + 64 * sizeof(float);
            FH_REQUIRE(lds <= 64 * 1024, "w4_gemm_dense: fused prologue needs %zu bytes of LDS", lds);
            form_hit(FORM_W4_FUSED_TAIL);
            if (has_zp) hipLaunchKernelGGL((w4_gemm_wgsplit_kernel<1, NT, true, true>), grid, dim3(W * 64), lds, stream, a);
            else hipLaunchKernelGGL((w4_gemm_wgsplit_kernel<1, NT, false, true>), grid, dim3(W * 64), lds, stream, a);
            FH_CHECK_LAUNCH();
            return 0;
        }
    }
    if (has_zp) hipLaunchKernelGGL((w4_gemm_wgsplit_kernel<MT, NT, true>), grid, dim3(W * 64), lds, stream, a);
    else hipLaunchKernelGGL((w4_gemm_wgsplit_kernel<MT, NT, false>), grid, dim3(W * 64), lds, stream, a);
    FH_CHECK_LAUNCH();
    return 0;
}

// ─────────────────────────────── host launchers ────────────────────────────

template <int MODE>
static int launch_w4(const W4Args& a, int mt, bool has_zp, dim3 grid, hipStream_t stream) {
#define FH_W4_CASE(MTV, ZPV)                                                       \
    hipLaunchKernelGGL((w4_gemm_kernel<MTV, ZPV, MODE>), grid, dim3(MODE == 0 ? 256 : 64), 0, stream, a)
    if constexpr (MODE != 0) {
        // few pairs (decode at c ≤ 8): 4 waves per workgroup split K (see KW)
        // default 16 pairs, measured on Qwen3-30B-A3B: c=1 + 10 %, c=2 + 5.5 % (756 → 798 tok/s), c=3 ±0, c=4 − 2.7 %, c=8 − 6.5 %.
        // Never beyond 64 pairs: the K-split form keeps its per-wave pair list in 64 LDS slots.
        const int kw_pairs = std::min(knobs().moe_kw_pairs, 64);
        const bool rt = a.cand != nullptr || a.pair_expert_ids != nullptr;      // routing prologue inside the launch (owns LDS)
        const bool kw4 = a.M <= kw_pairs && (a.cand == nullptr || a.route_T * a.route_Q * 8 <= 256) && a.G >= 4;
#define FH_W4_MOE(ZPV, KWV, RTV) hipLaunchKernelGGL((w4_gemm_kernel<1, ZPV, MODE, KWV, RTV>), grid, dim3(64 * KWV), 0, stream, a)
#define FH_W4_MOE_ZP(ZPV)                                                                    \
        if (kw4) { if (rt) FH_W4_MOE(ZPV, 4, true); else FH_W4_MOE(ZPV, 4, false); }         \
        else { if (rt) FH_W4_MOE(ZPV, 1, true); else FH_W4_MOE(ZPV, 1, false); }
        if (has_zp) { FH_W4_MOE_ZP(true) } else { FH_W4_MOE_ZP(false) }
#undef FH_W4_MOE_ZP
#undef FH_W4_MOE
    } else {
        if (mt == 1) { if (has_zp) FH_W4_CASE(1, true); else FH_W4_CASE(1, false); }
        else if (mt == 2) { if (has_zp) FH_W4_CASE(2, true); else FH_W4_CASE(2, false); }
        else { if (has_zp) FH_W4_CASE(4, true); else FH_W4_CASE(4, false); }
    }
#undef FH_W4_CASE
    FH_CHECK_LAUNCH();
    return 0;
}

static int launch_w4_slabs(const W4Args& a, int mt, bool has_zp, dim3 grid, hipStream_t stream) {
    return launch_w4<0>(a, mt, has_zp, grid, stream);
}

// Dense y[M,N] = x[M,K]·Wᵀ (+bias).  One launch: K is split across the waves of each workgroup.
bool w4_gemm_dense_can_fuse_combine_norm(const W4Device& w, int m) {
    return m >= 1 && m <= 4 && w.qw != nullptr && w.f16t == nullptr && w.perm == nullptr && w.k % 8 == 0 && w.k <= 8192;
}

int w4_gemm_dense(const W4Device& w, const __half* x, __half* out, int m, float* workspace,
                  size_t workspace_bytes, hipStream_t stream, const FusedCombineNorm* fa) {
    if (m <= 0) return 0;
    // (act-order weights: the packed rows are in sorted-g_idx order and every caller hands over activations whose columns were
    // gathered with the same permutation — dense_linear, ferrum_hip_gptq_linear_forward_f16 — so every kernel form applies)
    FH_REQUIRE(!fa || (w4_gemm_dense_can_fuse_combine_norm(w, m) && fa->residual_in != fa->residual_out && fa->top_k >= 1),
               "w4_gemm_dense: the fused combine + norm prologue needs <= 4 rows of a plain INT4 projection and ping-pong residual buffers");
    if (w.f16t) {                                     // DenseLinear: B::gemm on fp16 weights (linear.rs:109-129)
        form_hit(FORM_F16_DENSE_LINEAR);
        if (int rc = f16t_gemm(x, w.f16t, out, m, w.n, w.k, workspace, workspace_bytes, stream)) return rc;
        return w.bias ? add_bias_f16(out, w.bias, m, w.n, stream) : 0;
    }
    W4Args a{};
    a.qw = w.qw; a.sc = w.sc; a.zp = w.zp; a.bias = w.bias;
    a.x = x; a.out = out; a.M = m; a.K = w.k; a.N = w.n; a.G = w.G; a.n64 = w.n64; a.ldo = w.n; a.S = 1;
    if (fa) {
        a.fa_down = fa->down; a.fa_weights = fa->weights; a.fa_res_in = fa->residual_in; a.fa_res_out = fa->residual_out;
        a.fa_ln = fa->ln_w; a.fa_eps = fa->eps; a.fa_top_k = fa->top_k;
    }
    const int mt = m <= 16 ? 1 : (m <= 32 ? 2 : 4);
    const int row_blocks = cdiv(m, 16 * mt);
    // LDS-shared activations (w4_gemm_ldsa_kernel) + fp32 split-K slabs + one reduce launch.  Measured at m = 32 against
    // the one-launch intra-workgroup split below (tools/exp_dense.py): down 14336→4096 31.3 → 18.4 µs (S = 8),
    // gate_up 4096→28672 25.7 → 23.2 µs (S = 2); the small projections (qkv, o) tie, so they keep the single launch.
    // 64-row pipelined tiles (w4_gemm_tilep_kernel) for prefill — and for 33–63 rows on the larger projections, where the
    // skinny kernels' four row tiles re-fetch every activation fragment per 16–64 columns (c=48 decode, Llama-3.1-8B 5.18 →
    // 4.02 ms per step, Gemma-3-27B 15.1 → 11.5; the 2048→5120 / 4096→2048 projections of Qwen3-30B-A3B are faster skinny)
    const Knobs& kn = knobs();
    // ≥ 96 rows: 96- / 128- / 256-row tiles with the group scale folded into the fp16 B operand (w4_gemm_big_kernel) when they
    // fill the chip.  Deep K (≥ 24 groups): the tallest tile whose last round of 256 workgroups is ≥ 70 % full — 4096→28672 at
    // M = 8192: 64-row tiles 905, 96-row 1204, 128-row 1171, 256-row 1277 TFLOP/s (tools/exp_prefill_gemm.py, a throttled
    // back-to-back loop; one prefill of Llama-3.1-8B: 133 → 101 ms).  Shallow K: the 96-row form, two workgroups per CU, whose
    // prologues and epilogues overlap (2048→5120 at M = 8192: 852 / 924 TFLOP/s for 128 / 256 rows, 962 for 96).
    if (kn.w4_big >= 0 && m >= 96 && w.G % 2 == 0) {
        const int cols = cdiv(w.n64, 4);
        const long wgs6 = (long)cols * cdiv(m, 96), wgs8 = (long)cols * cdiv(m, 128), wgs16 = (long)cols * cdiv(m, 256);
        auto fill = [](long wgs) { return (double)wgs / (double)(cdiv(wgs, 256) * 256); };
        int MTv = 0;
        if (kn.w4_big == 6 || kn.w4_big == 8 || kn.w4_big == 16) MTv = m >= 16 * kn.w4_big ? kn.w4_big : 0;       // development: forced
        else if (w.G >= 24 && m >= 256 && wgs16 >= 192 && fill(wgs16) >= 0.7) MTv = 16;
        else if (w.G >= 24 && m >= 128 && wgs8 >= 192 && fill(wgs8) >= 0.7) MTv = 8;
        else if (wgs6 >= 128) MTv = 6;
        if (MTv) {
            a.S = 1; a.partial = nullptr;
            form_hit(FORM_W4_BIG);
            const dim3 grid(cols, cdiv(m, 16 * MTv), 1);
            if (MTv == 6) return launch_big<6>(a, w.zp != nullptr, grid, stream);
            return MTv == 16 ? launch_big<16>(a, w.zp != nullptr, grid, stream) : launch_big<8>(a, w.zp != nullptr, grid, stream);
        }
    }
#ifdef FERRUM_HIP_EXPERIMENTS
    if (kn.w4_ldsk && mt >= 2 && w.perm == nullptr) {
        int S = std::max(1, std::min(kn.w4_ldsa_s, w.G));
        const int rows_pad = row_blocks * 16 * mt, n_pad = w.n64 * 64;
        FH_REQUIRE(S == 1 || (workspace && (size_t)S * rows_pad * n_pad * sizeof(float) <= workspace_bytes), "ldsk: workspace");
        a.S = S; a.rows_pad = rows_pad; a.n_pad = n_pad;
        a.partial = S > 1 ? workspace : nullptr;
        const int rc = launch_ldsk_code(mt, kn.w4_ldsk, a, w.zp != nullptr, w.n64, row_blocks, stream);
        FH_REQUIRE(rc >= 0, "FERRUM_HIP_W4_LDSK=%d is not an instantiated form for %d row tiles", kn.w4_ldsk, mt);
        form_hit(FORM_W4_LDSK);
        if (rc || S == 1) return rc;
        hipLaunchKernelGGL(splitk_reduce_bias_kernel, dim3(cdiv(w.n, 256), m), dim3(256), 0, stream, workspace, out, w.bias, S, m,
                           w.n, rows_pad, n_pad, w.n);
        FH_CHECK_LAUNCH();
        return 0;
    }
#else
    FH_REQUIRE(kn.w4_ldsk == 0 && kn.w4_ldsw == 0, "FERRUM_HIP_W4_LDSK / _LDSW: the experimental decode GEMM forms are not compiled in (make EXPERIMENTS=1)");
#endif
    const int tile_min_env = kn.w4_tile_min_m;
    const int tile_min_m = tile_min_env > 0 ? tile_min_env : ((long)w.k * w.n >= (12L << 20) ? 33 : 64);
    if (m >= tile_min_m) {
        // too few tiles to cover the chip (narrow N or few rows): split K over grid.z into fp32 slabs + one reduce launch,
        // keeping ≥ 8 quant groups per split
        const int tile_wgs = kn.w4_tile_wgs;
        const int cols = cdiv(w.n64, 4), rts = cdiv(m, 64);
        int S = 1;
        while ((long)cols * rts * S < tile_wgs && w.G / (S * 2) >= 8) S *= 2;
        const int rows_pad = rts * 64, n_pad = w.n64 * 64;
        if (S > 1 && (workspace == nullptr || (size_t)S * rows_pad * n_pad * sizeof(float) > workspace_bytes)) S = 1;
        a.S = S;
        a.rows_pad = rows_pad; a.n_pad = n_pad;
        a.partial = S > 1 ? workspace : nullptr;
        if (int rc = launch_tilep(a, w.zp != nullptr, dim3(cols, rts, S), stream)) return rc;
        form_hit(FORM_W4_TILEP);
        if (S == 1) return 0;
        hipLaunchKernelGGL(splitk_reduce_bias_kernel, dim3(cdiv(w.n, 256), m), dim3(256), 0, stream, workspace, out, w.bias, S, m,
                           w.n, rows_pad, n_pad, w.n);
        FH_CHECK_LAUNCH();
        return 0;
    }
    const int lds_mode = kn.w4_ldsa;
    const bool lds_shape = (w.G >= 64 && w.n64 >= 32) || w.n64 >= 256;
    if (mt >= 2 && (lds_mode == 2 || (lds_mode == 1 && lds_shape && mt == 2))) {
        int nw = 4;
        if (kn.w4_ldsa_nw) nw = kn.w4_ldsa_nw == 8 && mt == 2 ? 8 : 4;
        const int cols = cdiv(w.n64, nw);
        int S = 1;
        while ((long)cols * row_blocks * S < 128 && w.G / (S * 2) >= 8) S *= 2;
        while (w.n64 <= 128 && (long)cols * row_blocks * S < 256 && w.G / (S * 2) >= 16) S *= 2;
        if (kn.w4_ldsa_s) S = std::max(1, std::min(kn.w4_ldsa_s, w.G));
        const int rows_pad = row_blocks * 16 * mt, n_pad = w.n64 * 64;
        if (S > 1 && (workspace == nullptr || (size_t)S * rows_pad * n_pad * sizeof(float) > workspace_bytes)) S = 1;
        a.S = S;
        a.rows_pad = rows_pad; a.n_pad = n_pad;
        a.partial = S > 1 ? workspace : nullptr;
        dim3 grid(cols, row_blocks, S);
        const bool zp = w.zp != nullptr;
        int rc;
        if (mt == 2) rc = nw == 8 ? launch_ldsa<2, 8>(a, zp, grid, stream) : launch_ldsa<2, 4>(a, zp, grid, stream);
        else rc = launch_ldsa<4, 4>(a, zp, grid, stream);      // MT = 4 with 8 waves would exceed 256 VGPRs per lane
        form_hit(FORM_W4_LDSA);
        if (rc || S == 1) return rc;
        hipLaunchKernelGGL(splitk_reduce_bias_kernel, dim3(cdiv(w.n, 256), m), dim3(256), 0, stream, workspace, out, w.bias, S, m,
                           w.n, rows_pad, n_pad, w.n);
        FH_CHECK_LAUNCH();
        return 0;
    }
    const int n16 = w.n64 * 4;
    // column tiles per workgroup: as wide as possible (A-fragment reuse) while ≥ ~384 workgroups exist
    int nt = 4;
    while (nt > 1 && (long)cdiv(n16, nt) * row_blocks < 384) nt >>= 1;
    // with ≥ 2 row tiles the activation fragments dominate the load traffic: share each across two
    // column tiles as long as ≥ 128 workgroups remain (measured: qkv 2048→5120 at T=32 10.4 → 7.2 µs)
    if (mt >= 2 && nt == 1 && (long)cdiv(n16, 2) * row_blocks >= 128) nt = 2;
    // waves per workgroup (= K slices): enough waves to cover the chip, ≥ 1 group each, LDS ≤ 128 KiB
    const long wgs = (long)cdiv(n16, nt) * row_blocks;
    const int w_cap = mt == 1 ? 16 : (mt == 2 ? (nt == 1 ? 16 : 8) : 4);    // matches the kernel's __launch_bounds__
    int W = 4;
    while (wgs * W < 2048 && W * 2 <= std::min(w_cap, w.G) && W * 2 * mt * nt * 4 <= 512) W <<= 1;
    while (W > w.G && W > 1) W >>= 1;
    if (kn.w4_nt) nt = kn.w4_nt;        // tuning overrides (development)
    if (kn.w4_w) W = kn.w4_w;
    dim3 grid(cdiv(n16, nt), row_blocks, 1);
    const bool zp = w.zp != nullptr;
    form_hit(FORM_W4_WGSPLIT);
#define FH_WG(MTV, NTV) return launch_wgsplit<MTV, NTV>(a, zp, grid, W, stream)
    if (mt == 1) { if (nt == 4) FH_WG(1, 4); if (nt == 2) FH_WG(1, 2); FH_WG(1, 1); }
    if (mt == 2) { if (nt == 4) FH_WG(2, 4); if (nt == 2) FH_WG(2, 2); FH_WG(2, 1); }
    if (nt == 4) FH_WG(4, 4);
    if (nt == 2) FH_WG(4, 2);
    FH_WG(4, 1);
#undef FH_WG
}

// MoE grouped GEMM over align-block routing arrays (block size 16).
// out row = pair id, in row = pair id / top_k (vLLM marlin_moe convention, ops.cu:942).
int w4_gemm_moe_inline_align(const W4Device& w, const __half* x, __half* out, const int32_t* pair_expert_ids,
                             int num_experts, int num_valid_pairs, int max_blocks, int top_k, int fused_silu,
                             int32_t* pub_sorted, int32_t* pub_block_ids, int32_t* pub_total, hipStream_t stream,
                             bool few_pairs_per_expert) {
    if (num_valid_pairs <= 0 || max_blocks <= 0) return 0;
    FH_REQUIRE(num_valid_pairs <= 1024 && num_experts <= 512, "inline align: pairs=%d experts=%d out of range", num_valid_pairs, num_experts);
    W4Args a{};
    a.qw = w.qw; a.sc = w.sc; a.zp = w.zp;
    a.expert_stride_qw = (long)w.n64 * w.G * 4 * 64 * 4;
    a.expert_stride_sc = (long)w.n64 * w.G * 16 * 4;
    a.x = x; a.out = out; a.M = num_valid_pairs; a.K = w.k; a.N = w.n; a.G = w.G; a.n64 = w.n64;
    a.ldo = fused_silu ? w.n / 2 : w.n;
    a.S = 1;
    a.pair_expert_ids = pair_expert_ids; a.num_experts = num_experts;
    a.pub_sorted_token_ids = pub_sorted; a.pub_block_ids = pub_block_ids; a.pub_total_post_pad = pub_total;
    a.top_k = top_k;
    a.few_pairs_per_expert = few_pairs_per_expert ? 1 : 0;
    dim3 grid(w.n64, max_blocks, 1);
    form_hit(FORM_MOE_INLINE_ALIGN);
    if (fused_silu) return launch_w4<2>(a, 1, w.zp != nullptr, grid, stream);
    return launch_w4<1>(a, 1, w.zp != nullptr, grid, stream);
}

// Expert-major grouped GEMM straight from the per-pair expert ids (no align arrays at all): out row = pair id,
// in row = pair id / top_k.  For decode batches with P ≤ 1024 pairs where most experts are active.
int w4_gemm_moe_expert_major(const W4Device& w, const __half* x, __half* out, const int32_t* pair_expert_ids, int num_experts,
                             int num_valid_pairs, int top_k, int fused_silu, hipStream_t stream) {
    if (num_valid_pairs <= 0) return 0;
    FH_REQUIRE(num_valid_pairs <= 1024 && num_experts <= 65535, "expert-major grouped GEMM: pairs=%d experts=%d out of range",
               num_valid_pairs, num_experts);
    W4Args a{};
    a.qw = w.qw; a.sc = w.sc; a.zp = w.zp;
    a.expert_stride_qw = (long)w.n64 * w.G * 4 * 64 * 4;
    a.expert_stride_sc = (long)w.n64 * w.G * 16 * 4;
    a.x = x; a.out = out; a.M = num_valid_pairs; a.K = w.k; a.N = w.n; a.G = w.G; a.n64 = w.n64;
    a.ldo = fused_silu ? w.n / 2 : w.n;
    a.S = 1;
    a.pair_expert_ids = pair_expert_ids; a.num_experts = num_experts;
    a.top_k = top_k;
    const dim3 grid(w.n64, num_experts, 1);
    const bool zp = w.zp != nullptr;
#ifdef FERRUM_HIP_EXPERIMENTS
    a.tl = (g_timeline_mode == 0 || g_timeline_mode == (fused_silu ? 2 : 1)) ? g_timeline : nullptr;
#endif
    form_hit(FORM_MOE_EXPERT_MAJOR);
    if (fused_silu) {
        if (zp) hipLaunchKernelGGL((w4_gemm_moe_em_kernel<true, 2>), grid, dim3(64), 0, stream, a);
        else hipLaunchKernelGGL((w4_gemm_moe_em_kernel<false, 2>), grid, dim3(64), 0, stream, a);
    } else {
        if (zp) hipLaunchKernelGGL((w4_gemm_moe_em_kernel<true, 1>), grid, dim3(64), 0, stream, a);
        else hipLaunchKernelGGL((w4_gemm_moe_em_kernel<false, 1>), grid, dim3(64), 0, stream, a);
    }
    FH_CHECK_LAUNCH();
    return 0;
}

// gate_up (+ silu·mul) and down of a decode batch as ONE launch (w4_gemm_moe_em2_kernel).  `arrive` must be zero on entry;
// the launch zeroes `arrive_next` (the caller alternates two counter arrays).  *took = 0 when the shapes do not fit the
// merged form (the caller then runs the two expert-major launches).
bool w4_gemm_moe_expert_major_pair_supports(const W4Device& gu, const W4Device& dn, int num_experts, int num_valid_pairs,
                                            const MoeRouteLists* route) {
    if (num_valid_pairs <= 0) return false;
    const bool zp = gu.zp != nullptr;
    if (zp != (dn.zp != nullptr) || dn.G < 2 || !gu.fused_gate_up || gu.n / 2 != dn.k || (gu.n / 2) % 8 != 0 || dn.n % 8 != 0 ||
        num_valid_pairs > 1024 || num_experts > 65535 || (long)num_valid_pairs * (gu.n / 2) * 2 >= (1L << 31))
        return false;
    if (route && (route->T < 1 || route->T > 64 || route->Q < 1 || route->Q > 4 || num_valid_pairs > 512 ||
                  !route->cand || !route->stats || !route->pub_expert_ids || !route->pub_expert_w))
        return false;
    return true;
}

int w4_gemm_moe_expert_major_pair(const W4Device& gu, const W4Device& dn, const __half* x, __half* h, __half* out,
                                  const int32_t* pair_expert_ids, int num_experts, int num_valid_pairs, int top_k,
                                  unsigned* arrive, unsigned* arrive_next, unsigned* timeout, int* took, hipStream_t stream,
                                  const MoeRouteLists* route) {
    *took = 0;
    if (num_valid_pairs <= 0) return 0;
    const bool zp = gu.zp != nullptr;
    if (!w4_gemm_moe_expert_major_pair_supports(gu, dn, num_experts, num_valid_pairs, route) || !arrive || !arrive_next || !timeout ||
        (!route && !pair_expert_ids))
        return 0;
    W4Em2Args a{};
    a.gu_qw = gu.qw; a.gu_sc = gu.sc; a.gu_zp = gu.zp; a.gu_G = gu.G; a.gu_n64 = gu.n64;
    a.gu_stride_qw = (long)gu.n64 * gu.G * 4 * 64 * 4; a.gu_stride_sc = (long)gu.n64 * gu.G * 16 * 4;
    a.dn_qw = dn.qw; a.dn_sc = dn.sc; a.dn_zp = dn.zp; a.dn_G = dn.G; a.dn_n64 = dn.n64;
    a.dn_stride_qw = (long)dn.n64 * dn.G * 4 * 64 * 4; a.dn_stride_sc = (long)dn.n64 * dn.G * 16 * 4;
    a.x = x; a.h = h; a.out = out; a.pair_expert_ids = pair_expert_ids;
    if (route) {
        a.cand = route->cand; a.stats = route->stats; a.route_T = route->T; a.route_Q = route->Q; a.norm_topk = route->norm_topk;
        a.pub_expert_ids = route->pub_expert_ids; a.pub_expert_w = route->pub_expert_w;
    }
    a.P = num_valid_pairs; a.top_k = top_k; a.E = num_experts; a.K = gu.k; a.I = gu.n / 2; a.H = dn.n;
    a.arrive = arrive; a.arrive_next = arrive_next; a.timeout = timeout;
#ifdef FERRUM_HIP_EXPERIMENTS
    a.tl = g_timeline;
#endif
    const dim3 grid((unsigned)((gu.n64 + dn.n64) * num_experts + (route ? (route->T + 15) / 16 : 0)), 1, 1);
    form_hit(FORM_MOE_EXPERT_MAJOR_PAIR);
    if (route) {
        if (zp) hipLaunchKernelGGL((w4_gemm_moe_em2_kernel<true, true>), grid, dim3(64), 0, stream, a);
        else hipLaunchKernelGGL((w4_gemm_moe_em2_kernel<false, true>), grid, dim3(64), 0, stream, a);
    } else {
        if (zp) hipLaunchKernelGGL((w4_gemm_moe_em2_kernel<true, false>), grid, dim3(64), 0, stream, a);
        else hipLaunchKernelGGL((w4_gemm_moe_em2_kernel<false, false>), grid, dim3(64), 0, stream, a);
    }
    FH_CHECK_LAUNCH();
    *took = 1;
    return 0;
}

// gate_up (+ silu·mul) and down of a small decode batch (≤ 64 pairs) as ONE block-major launch (w4_gemm_moe_bm2_kernel).
// `arrive` (max_blocks counters, MOE_PAIR_COUNTER_STRIDE words apart) must be zero on entry; the launch zeroes `arrive_next`.
// `route` hands the routing over as the router's per-part candidate lists (≤ 8 tokens) instead of pair ids.
bool w4_gemm_moe_block_major_pair_supports(const W4Device& gu, const W4Device& dn, int num_experts, int num_valid_pairs, int max_blocks,
                                           const MoeRouteLists* route) {
    if (num_valid_pairs <= 0 || num_valid_pairs > 64 || max_blocks <= 0 || max_blocks > num_experts || num_experts > 512) return false;
    if ((gu.zp != nullptr) != (dn.zp != nullptr) || !gu.fused_gate_up || gu.n / 2 != dn.k || (gu.n / 2) % 8 != 0 || dn.n % 8 != 0 ||
        gu.G < 1 || gu.G > 16 || dn.G < 1 || dn.G > 16 || (long)num_valid_pairs * (gu.n / 2) * 2 >= (1L << 31))
        return false;
    if (route && (route->T < 1 || route->T > 8 || !(route->Q == 1 || route->Q == 2 || route->Q == 4) || !route->cand || !route->stats ||
                  !route->pub_expert_ids || !route->pub_expert_w))
        return false;
    return true;
}

int w4_gemm_moe_block_major_pair(const W4Device& gu, const W4Device& dn, const __half* x, __half* h, __half* out,
                                 const int32_t* pair_expert_ids, int num_experts, int num_valid_pairs, int max_blocks, int top_k,
                                 unsigned* arrive, unsigned* arrive_next, unsigned* timeout, int* took, hipStream_t stream,
                                 const MoeRouteLists* route) {
    *took = 0;
    if (num_valid_pairs <= 0) return 0;
    if (!w4_gemm_moe_block_major_pair_supports(gu, dn, num_experts, num_valid_pairs, max_blocks, route) || !arrive || !arrive_next ||
        !timeout || (!route && !pair_expert_ids))
        return 0;
    W4Bm2Args a{};
    a.gu_qw = gu.qw; a.gu_sc = gu.sc; a.gu_zp = gu.zp; a.gu_G = gu.G; a.gu_n64 = gu.n64;
    a.gu_stride_qw = (long)gu.n64 * gu.G * 4 * 64 * 4; a.gu_stride_sc = (long)gu.n64 * gu.G * 16 * 4;
    a.dn_qw = dn.qw; a.dn_sc = dn.sc; a.dn_zp = dn.zp; a.dn_G = dn.G; a.dn_n64 = dn.n64;
    a.dn_stride_qw = (long)dn.n64 * dn.G * 4 * 64 * 4; a.dn_stride_sc = (long)dn.n64 * dn.G * 16 * 4;
    a.x = x; a.h = h; a.out = out; a.pair_expert_ids = pair_expert_ids;
    if (route) {
        a.cand = route->cand; a.stats = route->stats; a.route_T = route->T; a.route_Q = route->Q; a.norm_topk = route->norm_topk;
        a.pub_expert_ids = route->pub_expert_ids; a.pub_expert_w = route->pub_expert_w;
    }
    a.P = num_valid_pairs; a.top_k = top_k; a.E = num_experts; a.K = gu.k; a.I = gu.n / 2; a.H = dn.n;
    a.arrive = arrive; a.arrive_next = arrive_next; a.timeout = timeout;
#ifdef FERRUM_HIP_EXPERIMENTS
    a.tl = g_timeline;
#endif
    const dim3 grid((unsigned)(gu.n64 + dn.n64), (unsigned)max_blocks, 1);
    form_hit(FORM_MOE_BLOCK_MAJOR_PAIR);
    if (gu.zp) hipLaunchKernelGGL((w4_gemm_moe_bm2_kernel<true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((w4_gemm_moe_bm2_kernel<false>), grid, dim3(256), 0, stream, a);
    FH_CHECK_LAUNCH();
    *took = 1;
    return 0;
}

// gate_up phase of a decode step: routing is taken from the Q candidate lists written by
// fused_add_rms_norm_route_parts_f16; the merged ids / weights and the align arrays are published.
int w4_gemm_moe_merge_route(const W4Device& w, const __half* x, __half* out, const RouteCand* cand, const float* stats,
                            int tokens, int Q, int top_k, int norm_topk, int num_experts, int max_blocks, int fused_silu,
                            int32_t* pub_expert_ids, float* pub_expert_w, int32_t* pub_sorted, int32_t* pub_block_ids,
                            int32_t* pub_total, hipStream_t stream) {
    if (tokens <= 0 || max_blocks <= 0) return 0;
    FH_REQUIRE(tokens <= 64 && tokens * Q <= 128 && tokens * top_k <= 1024 && top_k <= 8 && num_experts <= 512,
               "merge_route: tokens=%d Q=%d top_k=%d experts=%d out of range", tokens, Q, top_k, num_experts);
    W4Args a{};
    a.qw = w.qw; a.sc = w.sc; a.zp = w.zp;
    a.expert_stride_qw = (long)w.n64 * w.G * 4 * 64 * 4;
    a.expert_stride_sc = (long)w.n64 * w.G * 16 * 4;
    a.x = x; a.out = out; a.M = tokens * top_k; a.K = w.k; a.N = w.n; a.G = w.G; a.n64 = w.n64;
    a.ldo = fused_silu ? w.n / 2 : w.n;
    a.S = 1;
    a.num_experts = num_experts;
    a.cand = cand; a.stats = stats; a.route_T = tokens; a.route_Q = Q; a.route_K = top_k; a.norm_topk = norm_topk;
    a.pub_expert_ids = pub_expert_ids; a.pub_expert_w = pub_expert_w;
    a.pub_sorted_token_ids = pub_sorted; a.pub_block_ids = pub_block_ids; a.pub_total_post_pad = pub_total;
    a.top_k = top_k;
    dim3 grid(w.n64, max_blocks, 1);
    form_hit(FORM_MOE_MERGE_ROUTE);
    if (fused_silu) return launch_w4<2>(a, 1, w.zp != nullptr, grid, stream);
    return launch_w4<1>(a, 1, w.zp != nullptr, grid, stream);
}

// Dense projection as S fp32 split-K slabs [S][rows_pad][n_pad] (no reduction here: the consumer kernel sums
// them in slab order).  Used where the consumer is a fused kernel anyway (o_proj → add+norm+route).
int w4_gemm_dense_slabs(const W4Device& w, const __half* x, float* slabs, size_t slab_bytes, int m, int S,
                        int* rows_pad_out, int* n_pad_out, hipStream_t stream) {
    if (m <= 0) return 0;
    // (act-order weights: the caller hands over rows its producer already wrote in the packed-row order, like w4_gemm_dense)
    FH_REQUIRE(w.bias == nullptr, "w4_gemm_dense_slabs: bias weights use the direct path");
    W4Args a{};
    a.qw = w.qw; a.sc = w.sc; a.zp = w.zp;
    a.x = x; a.M = m; a.K = w.k; a.N = w.n; a.G = w.G; a.n64 = w.n64; a.ldo = w.n;
    const int mt = m <= 16 ? 1 : (m <= 32 ? 2 : 4);
    const int row_blocks = cdiv(m, 16 * mt);
    a.rows_pad = row_blocks * 16 * mt;
    a.n_pad = w.n64 * 64;
    S = std::max(1, std::min(S, w.G));
    FH_REQUIRE((size_t)S * a.rows_pad * a.n_pad * sizeof(float) <= slab_bytes, "w4_gemm_dense_slabs: workspace too small");
    a.S = S;
    a.partial = slabs;
    *rows_pad_out = a.rows_pad;
    *n_pad_out = a.n_pad;
    // S == 1 still writes a slab (the MODE-0 kernel writes fp16 `out` only when S == 1 → force the slab path)
    form_hit(FORM_W4_SLABS);
    return launch_w4_slabs(a, mt, w.zp != nullptr, dim3(cdiv(w.n64, 4), row_blocks, S), stream);
}

// Dense projection for ≥ 64 rows as S fp32 slabs through the pipelined tile kernel (S as w4_gemm_dense would pick it); the
// consumer kernel sums the slabs in order, so the reduce launch disappears (short prefills: o_proj → add + norm + route).
int w4_gemm_dense_slabs_tile(const W4Device& w, const __half* x, float* slabs, size_t slab_bytes, int m, int* S_out,
                             int* rows_pad_out, int* n_pad_out, hipStream_t stream) {
    if (m <= 0) return 0;
    FH_REQUIRE(m >= 64 && w.perm == nullptr && w.bias == nullptr, "w4_gemm_dense_slabs_tile: m=%d / act-order / bias use the direct path", m);
    W4Args a{};
    a.qw = w.qw; a.sc = w.sc; a.zp = w.zp;
    a.x = x; a.M = m; a.K = w.k; a.N = w.n; a.G = w.G; a.n64 = w.n64; a.ldo = w.n;
    const int cols = cdiv(w.n64, 4), rts = cdiv(m, 64);
    int S = 1;
    while ((long)cols * rts * S < 256 && w.G / (S * 2) >= 8) S *= 2;
    a.rows_pad = rts * 64;
    a.n_pad = w.n64 * 64;
    while (S > 1 && (size_t)S * a.rows_pad * a.n_pad * sizeof(float) > slab_bytes) S /= 2;
    FH_REQUIRE((size_t)S * a.rows_pad * a.n_pad * sizeof(float) <= slab_bytes, "w4_gemm_dense_slabs_tile: workspace too small");
    a.S = S;
    a.partial = slabs;
    *S_out = S;
    *rows_pad_out = a.rows_pad;
    *n_pad_out = a.n_pad;
    form_hit(FORM_W4_SLABS_TILE);
    return launch_tilep(a, w.zp != nullptr, dim3(cols, rts, S), stream);
}

// Split count the LDS-shared-activation kernel wants for this shape (tools/exp_dense.py sweeps): enough workgroups to
// cover the chip (≥ 128–256 of 4 waves) while every split keeps ≥ 8 quant groups.
int w4_gemm_dense_lds_splits(const W4Device& w, int m) {
    const int mt = m <= 16 ? 1 : (m <= 32 ? 2 : 4);
    const int row_blocks = cdiv(m, 16 * mt), cols = cdiv(w.n64, 4);
    int S = 1;
    const int min_wgs = knobs().lds_min_wgs, min_groups = knobs().lds_min_groups;
    while ((long)cols * row_blocks * S < min_wgs && w.G / (S * 2) >= min_groups) S *= 2;
    // deep K (≥ 16 groups per split left): go on to one workgroup per CU — Llama-70B down 28672→8192 ran on 128 of 256 CUs
    // (narrow N only: for wide N the extra fp32 slabs cost more than the idle CUs — Gemma-3 gate_up 5376→43008 got slower)
    while (w.n64 <= 128 && (long)cols * row_blocks * S < 2 * min_wgs && w.G / (S * 2) >= 2 * min_groups) S *= 2;
    return S;
}

// Dense projection for 17–32 rows as S fp32 slabs through w4_gemm_ldsa_kernel (activations staged once per workgroup in
// LDS); the consumer kernel sums the slabs in order.  S ≤ 0 → w4_gemm_dense_lds_splits.
int w4_gemm_dense_slabs_lds(const W4Device& w, const __half* x, float* slabs, size_t slab_bytes, int m, int* S_inout,
                            int* rows_pad_out, int* n_pad_out, hipStream_t stream) {
    if (m <= 0) return 0;
    FH_REQUIRE(w.bias == nullptr, "w4_gemm_dense_slabs_lds: bias weights use the direct path");
    FH_REQUIRE(m > 16 && m <= 32, "w4_gemm_dense_slabs_lds: m=%d (17..32 rows)", m);
    int S = *S_inout > 0 ? *S_inout : w4_gemm_dense_lds_splits(w, m);
    S = std::max(1, std::min(S, w.G));
    W4Args a{};
    a.qw = w.qw; a.sc = w.sc; a.zp = w.zp;
    a.x = x; a.M = m; a.K = w.k; a.N = w.n; a.G = w.G; a.n64 = w.n64; a.ldo = w.n;
    a.rows_pad = 32;
    a.n_pad = w.n64 * 64;
    FH_REQUIRE((size_t)S * a.rows_pad * a.n_pad * sizeof(float) <= slab_bytes, "w4_gemm_dense_slabs_lds: workspace too small");
    a.S = S;
    a.partial = slabs;
    *S_inout = S;
    *rows_pad_out = a.rows_pad;
    *n_pad_out = a.n_pad;
    form_hit(FORM_W4_SLABS_LDS);
    return launch_ldsa<2, 4>(a, w.zp != nullptr, dim3(cdiv(w.n64, 4), 1, S), stream);
}

int w4_gemm_moe(const W4Device& w, const __half* x, __half* out, const int32_t* sorted_token_ids,
                const int32_t* block_ids, const int32_t* total_post_pad, int num_valid_pairs, int max_blocks,
                int top_k, int fused_silu, hipStream_t stream) {
    if (num_valid_pairs <= 0 || max_blocks <= 0) return 0;
    W4Args a{};
    a.qw = w.qw; a.sc = w.sc; a.zp = w.zp;
    a.expert_stride_qw = (long)w.n64 * w.G * 4 * 64 * 4;
    a.expert_stride_sc = (long)w.n64 * w.G * 16 * 4;
    a.x = x; a.out = out; a.M = num_valid_pairs; a.K = w.k; a.N = w.n; a.G = w.G; a.n64 = w.n64;
    a.ldo = fused_silu ? w.n / 2 : w.n;
    a.S = 1;
    a.sorted_token_ids = sorted_token_ids; a.block_ids = block_ids; a.total_post_pad = total_post_pad;
    a.top_k = top_k;
    dim3 grid(w.n64, max_blocks, 1);
    form_hit(FORM_MOE_BLOCK16);
    if (fused_silu) return launch_w4<2>(a, 1, w.zp != nullptr, grid, stream);
    return launch_w4<1>(a, 1, w.zp != nullptr, grid, stream);
}

// MoE grouped GEMM over 64-row blocks (moe_align_block_size with block 64): prefill-sized batches, where an expert sees
// tens to hundreds of pairs and the 16-row kernel would re-stream its weights once per 16 pairs.
int w4_gemm_moe_tile(const W4Device& w, const __half* x, __half* out, const int32_t* sorted_token_ids, const int32_t* block_ids,
                     const int32_t* total_post_pad, int num_valid_pairs, int max_blocks, int block_rows, int top_k, int fused_silu,
                     hipStream_t stream) {
    if (num_valid_pairs <= 0 || max_blocks <= 0) return 0;
    FH_REQUIRE(block_rows == 128 || block_rows == 96 || block_rows == 64 || block_rows == 32, "w4_gemm_moe_tile: block_rows=%d (128, 96, 64 or 32)", block_rows);
    FH_REQUIRE(block_rows < 96 || w.G % 2 == 0, "w4_gemm_moe_tile: 96- / 128-row blocks need an even number of quant groups (K=%d)", w.k);
    W4Args a{};
    a.qw = w.qw; a.sc = w.sc; a.zp = w.zp;
    a.expert_stride_qw = (long)w.n64 * w.G * 4 * 64 * 4;
    a.expert_stride_sc = (long)w.n64 * w.G * 16 * 4;
    a.x = x; a.out = out; a.M = num_valid_pairs; a.K = w.k; a.N = w.n; a.G = w.G; a.n64 = w.n64;
    a.ldo = fused_silu ? w.n / 2 : w.n;
    a.S = 1;
    a.sorted_token_ids = sorted_token_ids; a.block_ids = block_ids; a.total_post_pad = total_post_pad;
    a.top_k = top_k;
    dim3 grid(cdiv(w.n64, 4), max_blocks, 1);
    const bool zp = w.zp != nullptr;
    form_hit(block_rows >= 96 ? FORM_MOE_TILE_BIG : (block_rows == 64 ? FORM_MOE_TILE64 : FORM_MOE_TILE32));
    if (block_rows == 128) return fused_silu ? launch_big<8, 2>(a, zp, grid, stream) : launch_big<8, 1>(a, zp, grid, stream);
    if (block_rows == 96) return fused_silu ? launch_big<6, 2>(a, zp, grid, stream) : launch_big<6, 1>(a, zp, grid, stream);
    if (block_rows == 64) return fused_silu ? launch_tile<2, 4>(a, zp, grid, stream) : launch_tile<1, 4>(a, zp, grid, stream);
    return fused_silu ? launch_tile<2, 2>(a, zp, grid, stream) : launch_tile<1, 2>(a, zp, grid, stream);
}

// ───────────────────────── fp16 skinny GEMM (router, lm_head) ───────────────
// out[M,N] = x[M,K]·Wᵀ with W [N,K] fp16 row-major (B::gemm, traits.rs:190; cuBLAS hgemm on the
// reference's CUDA lane).  Same operand roles as the INT4 kernel: W rows are B-operand columns.
template <int MT, typename OutT>
__global__ __launch_bounds__(256) void f16_gemm_kernel(const __half* __restrict__ x, const __half* __restrict__ w,
                                                       OutT* __restrict__ out, float* __restrict__ partial,
                                                       int M, int N, int K, int S, int rows_pad, int n_pad) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int a = lane >> 4, b = lane & 15;
    const int ntile2 = blockIdx.x * 4 + wave;       // pair of 16-column tiles
    const int n0 = ntile2 * 32;
    if (n0 >= N) return;
    const int rb = blockIdx.y, z = blockIdx.z;
    const int ksteps = K / 32;
    const int s0 = (int)((long)ksteps * z / S), s1 = (int)((long)ksteps * (z + 1) / S);
    const __half* xrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        int r = rb * 16 * MT + mt * 16 + b;
        xrow[mt] = x + (long)(r < M ? r : M - 1) * K + 8 * a;
    }
    const __half* wrow[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        int n = n0 + j * 16 + b;
        wrow[j] = w + (long)(n < N ? n : N - 1) * K + 8 * a;
    }
    float4v acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[mt][j] = (float4v){0.f, 0.f, 0.f, 0.f};

    constexpr int U = 4;   // k-steps in flight per trip
    int s = s0;
    for (; s + U <= s1; s += U) {
        half8 bw[U][2], ax[U][MT];
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int j = 0; j < 2; j++)
                bw[u][j] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(wrow[j] + (s + u) * 32));
#pragma unroll
            for (int mt = 0; mt < MT; mt++) ax[u][mt] = *reinterpret_cast<const half8*>(xrow[mt] + (s + u) * 32);
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
                    acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ax[u][mt], bw[u][j], acc[mt][j], 0, 0, 0);
    }
    for (; s < s1; s++) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            half8 bwv = *reinterpret_cast<const half8*>(wrow[j] + s * 32);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                half8 axv = *reinterpret_cast<const half8*>(xrow[mt] + s * 32);
                acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axv, bwv, acc[mt][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int row = rb * 16 * MT + mt * 16 + 4 * a + r;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                int col = n0 + j * 16 + b;
                if (S > 1) {
                    partial[((long)z * rows_pad + row) * n_pad + col] = acc[mt][j][r];
                } else if (row < M && col < N) {
                    out[(long)row * N + col] = (OutT)acc[mt][j][r];
                }
            }
        }
}

template <typename OutT>
static int f16_gemm_impl(const __half* x, const __half* w, OutT* out, int m, int n, int k, float* workspace,
                         size_t workspace_bytes, hipStream_t stream) {
    if (m <= 0) return 0;
    FH_REQUIRE(k % 32 == 0, "f16_gemm: K=%d must be a multiple of 32", k);
    int mt = m <= 16 ? 1 : (m <= 32 ? 2 : 4);
    int row_blocks = cdiv(m, 16 * mt);
    int ntile2 = cdiv(n, 32);
    int ksteps = k / 32;
    long tasks = (long)ntile2 * row_blocks;
    int S = 1;
    while (tasks * S < 2048 && S * 2 <= ksteps / 8) S *= 2;
    int rows_pad = row_blocks * 16 * mt, n_pad = ntile2 * 32;
    if (S > 1 && (size_t)S * rows_pad * n_pad * sizeof(float) > workspace_bytes) S = 1;
    dim3 grid(cdiv(ntile2, 4), row_blocks, S);
    if (mt == 1) hipLaunchKernelGGL((f16_gemm_kernel<1, OutT>), grid, dim3(256), 0, stream, x, w, out, workspace, m, n, k, S, rows_pad, n_pad);
    else if (mt == 2) hipLaunchKernelGGL((f16_gemm_kernel<2, OutT>), grid, dim3(256), 0, stream, x, w, out, workspace, m, n, k, S, rows_pad, n_pad);
    else hipLaunchKernelGGL((f16_gemm_kernel<4, OutT>), grid, dim3(256), 0, stream, x, w, out, workspace, m, n, k, S, rows_pad, n_pad);
    FH_CHECK_LAUNCH();
    if (S > 1) {
        hipLaunchKernelGGL(splitk_reduce_kernel<OutT>, dim3(cdiv(n, 256), m), dim3(256), 0, stream, workspace, out, S, m, n,
                           rows_pad, n_pad, n);
        FH_CHECK_LAUNCH();
    }
    return 0;
}

int f16_gemm(const __half* x, const __half* w, __half* out, int m, int n, int k, float* workspace,
             size_t workspace_bytes, hipStream_t stream) {
    return f16_gemm_impl<__half>(x, w, out, m, n, k, workspace, workspace_bytes, stream);
}
int f16_gemm_f32out(const __half* x, const __half* w, float* out, int m, int n, int k, float* workspace,
                    size_t workspace_bytes, hipStream_t stream) {
    return f16_gemm_impl<float>(x, w, out, m, n, k, workspace, workspace_bytes, stream);
}

// ───────────── fp16 weights in MFMA-fragment-major tiles ("f16t") ─────────────
// Dense fp16 weights that are streamed once per step (lm_head, MoE router) are re-laid at load from
// row-major [N,K] to [N/16 tiles][K/32 k-steps][64 lanes][8 halves]: lane (a,b) of k-step s holds
// W[16·tile + b][32s + 8a .. +8], i.e. exactly its B-operand fragment, so every wave load is one
// contiguous 1-KiB line burst (the row-major form makes each lane of a load touch a different 4-KiB-
// strided row — 64 separate 16-byte segments per instruction).
__global__ void f16t_repack_kernel(const __half* __restrict__ w, __half* __restrict__ out, int N, int K) {
    const long chunk = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one 16-byte chunk per thread
    const int ksteps = K >> 5;
    const long total = (long)((N + 15) >> 4) * ksteps * 64;
    if (chunk >= total) return;
    const int lane = (int)(chunk & 63);
    const long ts = chunk >> 6;
    const int s = (int)(ts % ksteps);
    const long tile = ts / ksteps;
    const int a = lane >> 4, b = lane & 15;
    const long n = tile * 16 + b;
    half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (n < N) v = *reinterpret_cast<const half8*>(w + n * K + 32 * s + 8 * a);
    *reinterpret_cast<half8*>(out + chunk * 8) = v;
}

int f16t_repack(const __half* w_rowmajor, __half* out_tiled, int n, int k, hipStream_t stream) {
    FH_REQUIRE(k % 32 == 0, "f16t repack: K=%d must be a multiple of 32", k);
    long total = (long)cdiv(n, 16) * (k / 32) * 64;
    hipLaunchKernelGGL(f16t_repack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, w_rowmajor, out_tiled, n, k);
    FH_CHECK_LAUNCH();
    return 0;
}
size_t f16t_elems(int n, int k) { return (size_t)cdiv(n, 16) * 16 * k; }

// J = 16-column tiles per wave: 2, or 4 for two row tiles against a wide N (the lm_head at 17–32 rows: with 2 the activation
// fragments a wave re-reads from L2 are as many bytes as the weights it streams from HBM)
template <int MT, typename OutT, int J = 2>
__global__ __launch_bounds__(256) void f16t_gemm_kernel(const __half* __restrict__ x, const __half* __restrict__ wt,
                                                        OutT* __restrict__ out, float* __restrict__ partial, int M, int N,
                                                        int K, int S, int rows_pad, int n_pad) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int a = lane >> 4, b = lane & 15;
    const int ntile2 = blockIdx.x * 4 + wave;       // group of J 16-column tiles
    const int n0 = ntile2 * 16 * J;
    if (n0 >= N) return;
    const int rb = blockIdx.y, z = blockIdx.z;
    const int ksteps = K >> 5;
    const int tiles = (N + 15) >> 4;
    const int s0 = (int)((long)ksteps * z / S), s1 = (int)((long)ksteps * (z + 1) / S);
    const __half* xrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        int r = rb * 16 * MT + mt * 16 + b;
        xrow[mt] = x + (long)(r < M ? r : M - 1) * K + 8 * a;
    }
    const __half* wtile[J];
#pragma unroll
    for (int j = 0; j < J; j++) {
        int t = ntile2 * J + j;
        wtile[j] = wt + ((long)(t < tiles ? t : tiles - 1) * ksteps * 64 + lane) * 8;
    }
    float4v acc[MT][J];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int j = 0; j < J; j++) acc[mt][j] = (float4v){0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;
    int s = s0;
    for (; s + U <= s1; s += U) {
        half8 bw[U][J], ax[U][MT];
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int j = 0; j < J; j++)
                bw[u][j] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(wtile[j] + (long)(s + u) * 512));
#pragma unroll
            for (int mt = 0; mt < MT; mt++) ax[u][mt] = *reinterpret_cast<const half8*>(xrow[mt] + (s + u) * 32);
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int j = 0; j < J; j++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
                    acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ax[u][mt], bw[u][j], acc[mt][j], 0, 0, 0);
    }
    for (; s < s1; s++) {
#pragma unroll
        for (int j = 0; j < J; j++) {
            half8 bwv = *reinterpret_cast<const half8*>(wtile[j] + (long)s * 512);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                half8 axv = *reinterpret_cast<const half8*>(xrow[mt] + s * 32);
                acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axv, bwv, acc[mt][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int row = rb * 16 * MT + mt * 16 + 4 * a + r;
#pragma unroll
            for (int j = 0; j < J; j++) {
                int col = n0 + j * 16 + b;
                if (S > 1) {
                    partial[((long)z * rows_pad + row) * n_pad + col] = acc[mt][j][r];
                } else if (row < M && col < N) {
                    out[(long)row * N + col] = (OutT)acc[mt][j][r];
                }
            }
        }
}

template <typename OutT>
static int f16t_gemm_impl(const __half* x, const __half* wt, OutT* out, int m, int n, int k, float* workspace,
                          size_t workspace_bytes, hipStream_t stream) {
    if (m <= 0) return 0;
    FH_REQUIRE(k % 32 == 0, "f16t_gemm: K=%d must be a multiple of 32", k);
    int mt = m <= 16 ? 1 : (m <= 32 ? 2 : 4);
    int row_blocks = cdiv(m, 16 * mt);
    int ntile2 = cdiv(n, 32);
    int ksteps = k / 32;
    long tasks = (long)ntile2 * row_blocks;
    int S = 1;
    while (tasks * S < 2048 && S * 2 <= ksteps / 8) S *= 2;
    int rows_pad = row_blocks * 16 * mt, n_pad = ntile2 * 32;
    if (S > 1 && (size_t)S * rows_pad * n_pad * sizeof(float) > workspace_bytes) S = 1;
    if (mt == 2 && S == 1 && n >= 32768 && n % 64 == 0) {       // wide N at 17–32 rows (the lm_head): 64 columns per wave
        hipLaunchKernelGGL((f16t_gemm_kernel<2, OutT, 4>), dim3(cdiv(cdiv(n, 64), 4), row_blocks, 1), dim3(256), 0, stream, x, wt, out, workspace, m,
                           n, k, 1, rows_pad, n_pad);
        FH_CHECK_LAUNCH();
        return 0;
    }
    dim3 grid(cdiv(ntile2, 4), row_blocks, S);
    if (mt == 1) hipLaunchKernelGGL((f16t_gemm_kernel<1, OutT>), grid, dim3(256), 0, stream, x, wt, out, workspace, m, n, k, S, rows_pad, n_pad);
    else if (mt == 2) hipLaunchKernelGGL((f16t_gemm_kernel<2, OutT>), grid, dim3(256), 0, stream, x, wt, out, workspace, m, n, k, S, rows_pad, n_pad);
    else hipLaunchKernelGGL((f16t_gemm_kernel<4, OutT>), grid, dim3(256), 0, stream, x, wt, out, workspace, m, n, k, S, rows_pad, n_pad);
    FH_CHECK_LAUNCH();
    if (S > 1) {
        hipLaunchKernelGGL(splitk_reduce_kernel<OutT>, dim3(cdiv(n, 256), m), dim3(256), 0, stream, workspace, out, S, m, n,
                           rows_pad, n_pad, n);
        FH_CHECK_LAUNCH();
    }
    return 0;
}
int f16t_gemm(const __half* x, const __half* wt, __half* out, int m, int n, int k, float* workspace,
              size_t workspace_bytes, hipStream_t stream) {
    return f16t_gemm_impl<__half>(x, wt, out, m, n, k, workspace, workspace_bytes, stream);
}
int f16t_gemm_f32out(const __half* x, const __half* wt, float* out, int m, int n, int k, float* workspace,
                     size_t workspace_bytes, hipStream_t stream) {
    return f16t_gemm_impl<float>(x, wt, out, m, n, k, workspace, workspace_bytes, stream);
}

}  // namespace fh
