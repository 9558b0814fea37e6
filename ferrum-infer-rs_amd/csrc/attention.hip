// Paged-KV attention for gfx950: one kernel family for decode (q_len = 1), chunked prefill and
// mixed batches.
//
// Reference ops: `paged_varlen_attention` / `paged_batched_decode_attention` /
// `paged_decode_attention` (ferrum-kernels/src/backend/traits.rs:1813,1885,1719; CUDA launch sites
// backend/cuda/paged.rs:942,1309,1160).  Maths = causal GQA softmax(QKᵀ/√hd)·V with optional sliding
// window (CPU forms: backend/cpu.rs:2179-2259, ferrum-kv/src/attention.rs:30-114).
//
// Design (HBM-bound in decode):
//  * a workgroup owns (sequence, kv_head, 16 query rows) where a "row" is a (token, q-head of the
//    GQA group) pair — at decode the 16 rows are the nq/nkv heads of the single new token, so K/V
//    of a kv head are read ONCE for the whole group (the reference's CUDA lane reads them once per
//    q head, kernels/paged_decode_attention.cu:177-179);
//  * K/V tiles are stored MFMA-fragment shaped (kv_layout.h): every wave load is 1 KiB contiguous,
//    goes HBM → VGPR with no LDS staging, and is directly the A operand of v_mfma_f32_16x16x32_f16;
//  * Sᵀ = K·Qᵀ puts one query row per lane column, so softmax statistics are lane-local plus two
//    xor-shuffles; Oᵀ = Vᵀ·Pᵀ keeps the same column ↔ row map, so no cross-lane traffic for the
//    rescale and P feeds the second MFMA straight from the accumulator registers;
//  * the 4 waves of a workgroup split the KV range, and grid.z splits it further for long
//    contexts (flash-decode); partial (m, l, O) are merged in fixed order (deterministic);
//  * prefill (RS, ≥ 4 row tiles per sequence): the waves of a workgroup take 4 CONSECUTIVE row tiles instead and each
//    walks its whole KV range — no cross-wave merge, a loop long enough to amortise the prologue (at 256-token prompts
//    the KV-split form gives a wave 1–2 block pairs), and the four waves request the same K/V lines at about the same
//    time (L1 hits instead of 4× the L2 traffic).
#include "common.h"
#include "kernels.h"
#include "knobs.h"
#include "kv_layout.h"
#include "rope_rows.h"

namespace fh {

struct AttnArgs {
    const __half* q;        // [total_q, nq, hd]
    const __half* k_pool;   // [blocks, nkv, tile]
    const __half* v_pool;
    __half* out;            // [total_q, nq, hd]
    const uint32_t* cu_seqlens_q;   // [num_seqs+1] or null (decode: token i ↔ seq i)
    const uint32_t* pos_offsets;    // [num_seqs] kv position of each seq's first q token
    const uint32_t* kv_lens;        // decode form: [num_seqs] kv length incl. the new token (pos = len-1)
    const int32_t* block_tables;    // [num_seqs, max_blocks]
    float* partial;         // [tiles][nsplit][16][hd+2] when nsplit > 1
    int num_seqs, nq, nkv, tiles_per_seq, max_blocks, sliding_window, nsplit;
    int compact;            // grid.x enumerates real work units of a ragged batch (see the kernel), not num_seqs × tiles_per_seq
    float scale;
    // fused decode form (FUSED_QKV): q / new K / new V come straight from the qkv projection output
    const __half* qkv;      // [num_seqs, (nq + 2·nkv)·hd]
    const __half* q_norm_w;
    const __half* k_norm_w;
    const float* cos_t;
    const float* sin_t;
    __half* k_pool_w;
    __half* v_pool_w;
    float eps;
    int qk_mode;
    // decode forms: the consumer is an act-order o_proj — channel c of a row is stored at out_scatter[c] (inverse of the
    // projection's row permutation), so no gather launch sits between attention and the GEMM
    const int32_t* out_scatter;
};

// FUSED_QKV (decode, q_len = 1): the workgroup of (sequence, kv head) first does what
// split_qkv_norm_rope_into_paged_cache_varlen does for ITS rows of the new token — the G query heads of the
// group, the kv head's key (both per-head RMSNorm + RoPE per qk_mode) and value — into LDS, with the same
// float operations in the same order (bit-identical fp16 results).  The split that owns the last KV block
// writes the new K/V into the paged cache and every wave patches the new token's slot into the fragments it
// loaded from that block, so nothing depends on the visibility of the cache store inside the launch.
template <int HD, bool FUSED_QKV, int NW, bool RS = false>
__global__ __launch_bounds__(NW * 64) void paged_attn_kernel(AttnArgs p) {
    static_assert(!(RS && FUSED_QKV), "row-split is the prefill form");
    constexpr int DT = HD / 16;      // output d-tiles
    constexpr int KS = HD / 32;      // QKᵀ k-steps (= 1-KiB loads per K tile = per V tile)
    constexpr int OSTRIDE = HD + 4;  // padded LDS row (floats)
    __shared__ __attribute__((aligned(16))) float lds_o[NW * 16 * OSTRIDE];
    __shared__ float lds_m[NW * 16], lds_l[NW * 16];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int a = lane >> 4, b = lane & 15;
    const int kvh = blockIdx.y, z = RS ? 0 : blockIdx.z;
    const int G = p.nq / p.nkv;
    int seq, unit;                   // unit: row tile (KV-split form) or group of NW row tiles (row-split form) in the sequence
    if (p.compact) {
        // ragged batches (31 decode tokens + one 256-token prompt): a grid of num_seqs × max tiles launches thousands of
        // workgroups that only find out they are empty (≈ 15 ns each, 70 µs per layer); here grid.x is an upper bound of
        // the REAL work units and every wave finds its (sequence, unit) by a scan of cu_seqlens_q, 64 sequences per pass
        const int w = blockIdx.x;
        int base = 0;
        seq = -1;
        unit = 0;
        for (int s0 = 0; s0 < p.num_seqs; s0 += 64) {
            const int s = s0 + lane;
            int units = 0;
            if (s < p.num_seqs) {
                const int tiles = ((int)(p.cu_seqlens_q[s + 1] - p.cu_seqlens_q[s]) * G + 15) >> 4;
                units = RS ? (tiles <= 1 ? tiles : (tiles + NW - 1) / NW) : tiles;
            }
            int incl = units;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            const int total = __shfl(incl, 63, 64);
            if (w < base + total) {
                const unsigned long long hit = __ballot(w < base + incl);      // first lane whose inclusive sum passes w
                const int src = __ffsll((long long)hit) - 1;
                seq = s0 + src;
                unit = w - (base + __shfl(incl - units, src, 64));
                break;
            }
            base += total;
        }
        if (seq < 0) return;
    } else {
        const int tpw = RS ? (p.tiles_per_seq + NW - 1) / NW : p.tiles_per_seq;      // workgroups per sequence
        seq = blockIdx.x / tpw;
        unit = blockIdx.x % tpw;
    }
    const int tok0 = p.cu_seqlens_q ? (int)p.cu_seqlens_q[seq] : seq;
    const int q_len = p.cu_seqlens_q ? (int)p.cu_seqlens_q[seq + 1] - tok0 : 1;
    const int rows_total = q_len * G;
    // a row-split launch still serves its single-tile sequences (the decode tokens of a mixed batch) KV-split: one wave
    // alone would walk the whole context
    const bool rs = RS && rows_total > 16;
    const int tile = rs ? unit * NW + wave : unit;
    if (tile * 16 >= rows_total) return;
    const int pos0 = p.kv_lens ? (int)p.kv_lens[seq] - 1 : (int)p.pos_offsets[seq];

    // this lane's query row (column b of every MFMA)
    const int rho = tile * 16 + b;
    const bool row_ok = rho < rows_total;
    const int t_local = row_ok ? rho / G : 0;
    const int g = row_ok ? rho % G : 0;
    const int row_pos = pos0 + t_local;                 // attends keys [win_lo, row_pos]
    const int win_lo = p.sliding_window > 0 ? max(0, row_pos + 1 - p.sliding_window) : 0;
    const long q_off = ((long)(tok0 + t_local) * p.nq + kvh * G + g) * HD;

    // tile-wide key range
    const int t_first = (tile * 16) / G;
    const int t_last = min(q_len - 1, (tile * 16 + 15) / G);
    const int kv_end = pos0 + t_last + 1;
    const int kv_begin = p.sliding_window > 0 ? max(0, pos0 + t_first + 1 - p.sliding_window) : 0;
    const int pair_lo = (kv_begin / KV_BLOCK) / 2;
    const int pair_hi = (cdiv_dev(kv_end, KV_BLOCK) + 1) / 2;       // exclusive
    const int npairs = pair_hi - pair_lo;
    const int per_split = RS ? npairs : (npairs + p.nsplit - 1) / p.nsplit;      // (a row-split launch has nsplit = 1)
    const int my_lo = pair_lo + z * per_split;
    const int my_hi = min(pair_hi, my_lo + per_split);
    const int STEP = rs ? 1 : NW;                       // row-split: every wave walks the whole range of ITS tile
    const int first = my_lo + (rs ? 0 : wave);
    const int nblocks = cdiv_dev(kv_end, KV_BLOCK);

    // Block-table entries of this split are fetched once (64 per wave, re-fetched every 32 pairs) and the K/V
    // fragments of a wave's NEXT pair are requested before the current pair is consumed, so a wave exposes one
    // memory round trip, not one (table) + one (tiles) per pair.
    const int32_t* bt = p.block_tables + (long)seq * p.max_blocks;
    const long tile_elems = kv_tile_elems(HD);
    int bt_lo = 2 * my_lo;
    int btv = bt[min(bt_lo + lane, p.max_blocks - 1)];
    struct Frags { half8 k0[KS], k1[KS], v0[KS], v1[KS]; };
    auto issue = [&](int pr, Frags& f) {
        const int blk0 = 2 * pr, blk1 = 2 * pr + 1;
        const bool has1 = blk1 < nblocks;
        if (blk1 - bt_lo >= 64) {                        // wave-uniform: next chunk of the table
            bt_lo = blk0;
            btv = bt[min(bt_lo + lane, p.max_blocks - 1)];
        }
        const long phys0 = __builtin_amdgcn_readlane(btv, blk0 - bt_lo);
        const long phys1 = has1 ? __builtin_amdgcn_readlane(btv, blk1 - bt_lo) : phys0;
        const __half* k0 = p.k_pool + (phys0 * p.nkv + kvh) * tile_elems + lane * 8;
        const __half* k1 = p.k_pool + (phys1 * p.nkv + kvh) * tile_elems + lane * 8;
        const __half* v0 = p.v_pool + (phys0 * p.nkv + kvh) * tile_elems + lane * 8;
        const __half* v1 = p.v_pool + (phys1 * p.nkv + kvh) * tile_elems + lane * 8;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            f.k0[s] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(k0 + s * 512));
            f.k1[s] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(k1 + s * 512));
        }
#pragma unroll
        for (int s = 0; s < KS; s++) {
            f.v0[s] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(v0 + s * 512));
            f.v1[s] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(v1 + s * 512));
        }
    };
    Frags cur, nxt;
    if (first < my_hi) issue(first, cur);

    half8 qf[KS];
    __shared__ __attribute__((aligned(16))) __half lds_q[FUSED_QKV ? 16 * HD : 8];
    __shared__ __attribute__((aligned(16))) __half lds_kv[FUSED_QKV ? 2 * HD : 8];
    const int last_blk = pos0 / KV_BLOCK, slot_new = pos0 % KV_BLOCK;
    if (FUSED_QKV) {
        // 16 quarter-wave rows cover the G query heads, the key (row G), the value (row G + 1); rows ≥ G of lds_q are zero
        constexpr int HALF = HD / 2;
        const int q_dim = p.nq * HD, kv_dim = p.nkv * HD;
        const __half* qrow = p.qkv + (long)seq * (q_dim + 2 * kv_dim);
        const int r16 = threadIdx.x >> 4, q16 = threadIdx.x & 15;
        const int rc = r16 < G + 2 ? r16 : G + 1;     // clamped: loads stay unconditional (r16 < 16·NW/4)
        const __half* src = rc < G ? qrow + (kvh * G + rc) * HD
                                   : rc == G ? qrow + q_dim + kvh * HD : qrow + q_dim + kv_dim + kvh * HD;
        const RopeRow<HD> rr = rope_row16<HD>(src, rc < G ? p.q_norm_w : p.k_norm_w, p.cos_t + (long)pos0 * HALF,
                                              p.sin_t + (long)pos0 * HALF, rc <= G ? p.qk_mode : 0, p.qk_mode == 1,
                                              p.qk_mode != 0, p.eps, q16);
        using hv = typename RopeRow<HD>::hv;
        hv z0, z1;
#pragma unroll
        for (int k = 0; k < HD / 32; k++) { z0[k] = (_Float16)0.f; z1[k] = (_Float16)0.f; }
        _Float16* lq = reinterpret_cast<_Float16*>(lds_q) + r16 * HD;
        if (r16 < 16) {
            *reinterpret_cast<hv*>(lq + rr.off0) = r16 < G ? rr.out0 : z0;
            *reinterpret_cast<hv*>(lq + rr.off1) = r16 < G ? rr.out1 : z1;
        }
        if (r16 == G || r16 == G + 1) {
            _Float16* lk = reinterpret_cast<_Float16*>(lds_kv) + (r16 - G) * HD;
            *reinterpret_cast<hv*>(lk + rr.off0) = rr.out0;
            *reinterpret_cast<hv*>(lk + rr.off1) = rr.out1;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < KS; s++) qf[s] = *reinterpret_cast<const half8*>(lds_q + b * HD + 32 * s + 8 * a);
        // the split that owns the last block stores the new token's K (wave 0) and V (wave 1) into the cache
        const int last_pair = last_blk >> 1;
        if (last_pair >= my_lo && last_pair < my_hi && wave < 2) {
            const long phys = p.block_tables[(long)seq * p.max_blocks + last_blk];
            const long toff = (phys * p.nkv + kvh) * kv_tile_elems(HD);
            for (int d = lane; d < HD; d += 64) {
                if (wave == 0) p.k_pool_w[toff + k_tile_off(slot_new, d)] = lds_kv[d];
                else p.v_pool_w[toff + v_tile_off(slot_new, d)] = lds_kv[HD + d];
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < KS; s++) qf[s] = *reinterpret_cast<const half8*>(p.q + q_off + 32 * s + 8 * a);
    }

    float m_run = -INFINITY, l_run = 0.f;
    float4v o_acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; dt++) o_acc[dt] = (float4v){0.f, 0.f, 0.f, 0.f};


    for (int pr = first; pr < my_hi; pr += STEP) {
        const int blk0 = 2 * pr, blk1 = 2 * pr + 1;
        const bool has1 = blk1 < nblocks;
        if (pr + STEP < my_hi) issue(pr + STEP, nxt);
        __builtin_amdgcn_sched_barrier(0);               // keep the prefetch ahead of this pair's MFMAs
        half8 (&kf0)[KS] = cur.k0, (&kf1)[KS] = cur.k1, (&vf0)[KS] = cur.v0, (&vf1)[KS] = cur.v1;
        if (FUSED_QKV && (blk0 == last_blk || blk1 == last_blk)) {
            // patch the new token's slot (key slot_new of block last_blk) into the loaded fragments
            const bool in1 = blk1 == last_blk;
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const half8 nk = *reinterpret_cast<const half8*>(lds_kv + 32 * s + 8 * a);
                if (b == slot_new) { if (in1) kf1[s] = nk; else kf0[s] = nk; }
            }
            const bool a_hit = a == (slot_new >> 2);
            const int jn = slot_new & 3;
#pragma unroll
            for (int ld = 0; ld < KS; ld++) {
#pragma unroll
                for (int sub = 0; sub < 2; sub++) {
                    const _Float16 nv = (_Float16)lds_kv[HD + 32 * ld + 16 * sub + b];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (a_hit && j == jn) { if (in1) vf1[ld][4 * sub + j] = nv; else vf0[ld][4 * sub + j] = nv; }
                    }
                }
            }
        }
        // Sᵀ[key][row] for the two blocks: lane (a,b) ← keys 4a+r of each block, query row b
        float4v s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; s++) {
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf0[s], qf[s], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf1[s], qf[s], s1, 0, 0, 0);
        }
        float sc[8];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int kp0 = blk0 * KV_BLOCK + 4 * a + r, kp1 = blk1 * KV_BLOCK + 4 * a + r;
            bool ok0 = row_ok && kp0 <= row_pos && kp0 >= win_lo;
            bool ok1 = row_ok && has1 && kp1 <= row_pos && kp1 >= win_lo;
            sc[r] = ok0 ? s0[r] * p.scale : -INFINITY;
            sc[4 + r] = ok1 ? s1[r] * p.scale : -INFINITY;
            mx = fmaxf(mx, fmaxf(sc[r], sc[4 + r]));
        }
        mx = rows_reduce_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float m_safe = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __expf(m_run - m_safe);      // m_run = -inf → 0
        float psum = 0.f;
        half8 pf;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float pv = __expf(sc[j] - m_safe);            // masked → 0
            psum += pv;
            pf[j] = (_Float16)pv;
        }
        psum = rows_reduce_sum(psum);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        // Oᵀ[d][row] += Vᵀ[d][key]·Pᵀ[key][row]; contraction index (a,j): j<4 block0 key 4a+j, else block1
        typedef uint32_t u32x4a __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int dt = 0; dt < DT; dt++) {
            // 8-byte half `dt & 1` of the lane's V fragment of block 0, then of block 1 (whole dwords: register selection, no
            // element moves); a missing block 1 is a copy of block 0 whose P is 0
            const int ld = dt >> 1, sub = (dt & 1) * 2;
            const u32x4a w0 = __builtin_bit_cast(u32x4a, vf0[ld]), w1 = __builtin_bit_cast(u32x4a, vf1[ld]);
            const half8 vfrag = __builtin_bit_cast(half8, (u32x4a){w0[sub], w0[sub + 1], w1[sub], w1[sub + 1]});
#pragma unroll
            for (int r = 0; r < 4; r++) o_acc[dt][r] *= alpha;
            o_acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfrag, pf, o_acc[dt], 0, 0, 0);
        }
        cur = nxt;
    }

    if (rs) {
        // wave-private transposition through LDS (Oᵀ[d][row] → rows of 64-byte pieces); LDS ops of one wave execute in order
        float* lo = lds_o + wave * 16 * OSTRIDE;
#pragma unroll
        for (int dt = 0; dt < DT; dt++) *reinterpret_cast<float4v*>(&lo[b * OSTRIDE + dt * 16 + 4 * a]) = o_acc[dt];
        const int row = lane >> 2, dl = lane & 3;
        const float l_row = __shfl(l_run, row, 64);       // lane `row` (a = 0, b = row) holds that row's sum
        const int rho_o = tile * 16 + row;
        if (rho_o >= rows_total) return;
        const float inv = l_row > 0.f ? 1.0f / l_row : 0.f;
        constexpr int DPL = HD / 4;
        __half* o = p.out + ((long)(tok0 + rho_o / G) * p.nq + kvh * G + rho_o % G) * HD + dl * DPL;
#pragma unroll
        for (int i = 0; i < DPL; i += 8) {
            half8 h;
#pragma unroll
            for (int j = 0; j < 8; j++) h[j] = (_Float16)(lo[row * OSTRIDE + dl * DPL + i + j] * inv);
            *reinterpret_cast<half8*>(o + i) = h;
        }
        return;
    }
    // merge the NW waves' partial states through LDS
    if (a == 0) {
        lds_m[wave * 16 + b] = m_run;
        lds_l[wave * 16 + b] = l_run;
    }
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
        *reinterpret_cast<float4v*>(&lds_o[(wave * 16 + b) * OSTRIDE + dt * 16 + 4 * a]) = o_acc[dt];
    __syncthreads();

    // thread → (row = tid / TPR, DPT dims)
    constexpr int TPR = NW * 4;
    const int row = threadIdx.x / TPR, dl = threadIdx.x % TPR;
    float mw[NW], M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; w++) { mw[w] = lds_m[w * 16 + row]; M = fmaxf(M, mw[w]); }
    const float Ms = M == -INFINITY ? 0.f : M;
    float fw[NW], L = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) { fw[w] = __expf(mw[w] - Ms); L += lds_l[w * 16 + row] * fw[w]; }

    const int rho_o = tile * 16 + row;
    const bool ok_o = rho_o < rows_total;
    const int t_o = ok_o ? rho_o / G : 0, g_o = ok_o ? rho_o % G : 0;
    constexpr int DPT = HD / TPR;    // dims per thread
    float ov[DPT];
#pragma unroll
    for (int i = 0; i < DPT; i++) {
        int d = dl * DPT + i;
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) acc += lds_o[(w * 16 + row) * OSTRIDE + d] * fw[w];
        ov[i] = acc;
    }
    if (p.nsplit > 1) {
        float* dst = p.partial + (((long)blockIdx.x * p.nkv + kvh) * p.nsplit + z) * 16 * (HD + 2) + row * (HD + 2);
#pragma unroll
        for (int i = 0; i < DPT; i++) dst[dl * DPT + i] = ov[i];
        if (dl == 0) { dst[HD] = M; dst[HD + 1] = L; }
        return;
    }
    if (!ok_o) return;
    const float inv = L > 0.f ? 1.0f / L : 0.f;
    if (p.out_scatter) {
        __half* orow = p.out + (long)(tok0 + t_o) * p.nq * HD;
        const int32_t* sc = p.out_scatter + (kvh * G + g_o) * HD + dl * DPT;
#pragma unroll
        for (int i = 0; i < DPT; i++) orow[sc[i]] = __float2half(ov[i] * inv);
        return;
    }
    __half* o = p.out + ((long)(tok0 + t_o) * p.nq + kvh * G + g_o) * HD + dl * DPT;
#pragma unroll
    for (int i = 0; i < DPT; i++) o[i] = __float2half(ov[i] * inv);
}

// ── prefill attention with the K/V tiles of a block pair staged ONCE per workgroup in LDS (flash form) ───────────────────
// The forms above pull every K/V fragment from L2 once per 16-row tile: at long prompts that is the whole cost (one 8192-token
// prompt: 2.6 ms per layer, ≈ 210 TFLOP/s).  Here a workgroup owns NW·MT = 8 (HD ≤ 128) consecutive row tiles of one sequence
// and one kv head — 128 query rows, i.e. 16 tokens × 8 heads of the GQA group — and walks their common key range pair by pair:
// wave w copies tile w of the pair (K block 0, K block 1, V block 0, V block 1: one 16·HD-element tile each, the same
// fragment-shaped bytes the other forms load) into a double-buffered LDS stage while the previous pair is consumed; every wave
// then reads the fragments with ds_read_b128 and runs them against ITS MT row tiles (the per-row arithmetic is the KV-split
// form's: Sᵀ = K·Qᵀ, lane-local online softmax, Oᵀ += Vᵀ·Pᵀ).  One barrier per pair; L2 traffic per query row ÷ 8.
template <int HD, int MT>
__global__ __launch_bounds__(256, HD <= 128 ? 2 : 1) void paged_prefill_attn_kernel(AttnArgs p) {
    constexpr int NW = 4, DT = HD / 16, KS = HD / 32, TILE = 16 * HD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __half* lds_kv = reinterpret_cast<__half*>(smem);          // [2 stages][K0, K1, V0, V1][TILE]; reused by the epilogue
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int a = lane >> 4, b = lane & 15;
    const int kvh = blockIdx.y;
    const int G = p.nq / p.nkv;
    // (sequence, unit of NW·MT row tiles) of this workgroup: scan of cu_seqlens_q, 64 sequences per pass
    int seq = -1, unit = 0;
    {
        const int w = blockIdx.x;
        int base = 0;
        for (int s0 = 0; s0 < p.num_seqs; s0 += 64) {
            const int s = s0 + lane;
            int units = 0;
            if (s < p.num_seqs) {
                const int tiles = ((int)(p.cu_seqlens_q[s + 1] - p.cu_seqlens_q[s]) * G + 15) >> 4;
                units = (tiles + NW * MT - 1) / (NW * MT);
            }
            int incl = units;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            const int total = __shfl(incl, 63, 64);
            if (w < base + total) {
                const unsigned long long hit = __ballot(w < base + incl);
                const int src = __ffsll((long long)hit) - 1;
                seq = s0 + src;
                unit = w - (base + __shfl(incl - units, src, 64));
                break;
            }
            base += total;
        }
        if (seq < 0) return;                       // whole workgroup (the scan is wave-uniform and identical in every wave)
    }
    const int tok0 = (int)p.cu_seqlens_q[seq];
    const int q_len = (int)p.cu_seqlens_q[seq + 1] - tok0;
    const int rows_total = q_len * G, tiles_s = (rows_total + 15) >> 4;
    const int pos0 = (int)p.pos_offsets[seq];

    // this wave's MT row tiles; lane column b ↔ query row
    int row_pos[MT], win_lo[MT];
    bool row_ok[MT];
    int vis_hi[MT], vis_lo[MT];      // keys in [vis_lo, vis_hi] are visible to EVERY row of the tile (−1: never take the fast path)
    half8 qf[MT][KS];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        const int tile = (unit * NW + wave) * MT + mt;
        const bool full = tile * 16 + 15 < rows_total;
        const int tmin = pos0 + (tile * 16) / G, tmax = pos0 + (tile * 16 + 15) / G;     // first / last token position of the tile
        vis_hi[mt] = full ? tmin : -1;
        vis_lo[mt] = p.sliding_window > 0 ? max(0, tmax + 1 - p.sliding_window) : 0;
        const int rho = tile * 16 + b;
        row_ok[mt] = rho < rows_total;
        const int t_local = row_ok[mt] ? rho / G : 0, g = row_ok[mt] ? rho % G : 0;
        row_pos[mt] = pos0 + t_local;
        win_lo[mt] = p.sliding_window > 0 ? max(0, row_pos[mt] + 1 - p.sliding_window) : 0;
        const long q_off = ((long)(tok0 + t_local) * p.nq + kvh * G + g) * HD;
#pragma unroll
        for (int s = 0; s < KS; s++) qf[mt][s] = *reinterpret_cast<const half8*>(p.q + q_off + 32 * s + 8 * a);
    }
    // key range common to the workgroup's rows
    const int tile_first = unit * NW * MT, tile_last = min(tile_first + NW * MT, tiles_s) - 1;
    const int t_first = (tile_first * 16) / G, t_last = min(q_len - 1, (tile_last * 16 + 15) / G);
    const int kv_end = pos0 + t_last + 1;
    const int kv_begin = p.sliding_window > 0 ? max(0, pos0 + t_first + 1 - p.sliding_window) : 0;
    const int pair_lo = (kv_begin / KV_BLOCK) / 2, pair_hi = (cdiv_dev(kv_end, KV_BLOCK) + 1) / 2;
    const int nblocks = cdiv_dev(kv_end, KV_BLOCK);

    // staging: wave w moves tile w of a pair (0: K block 0, 1: K block 1, 2: V block 0, 3: V block 1)
    const int32_t* bt = p.block_tables + (long)seq * p.max_blocks;
    int bt_lo = 2 * pair_lo;
    int btv = bt[min(bt_lo + lane, p.max_blocks - 1)];
    const __half* pool = wave < 2 ? p.k_pool : p.v_pool;
    half8 st[KS];
    auto issue = [&](int pr) {
        int blk = 2 * pr + (wave & 1);
        if (blk >= nblocks) blk = 2 * pr;                // odd tail: a copy of block 0, masked below
        if (2 * pr + 1 - bt_lo >= 64) {                  // wave-uniform: next chunk of the table
            bt_lo = 2 * pr;
            btv = bt[min(bt_lo + lane, p.max_blocks - 1)];
        }
        const long phys = __builtin_amdgcn_readlane(btv, blk - bt_lo);
        const __half* src = pool + (phys * p.nkv + kvh) * (long)TILE + lane * 8;
#pragma unroll
        for (int s = 0; s < KS; s++) st[s] = *reinterpret_cast<const half8*>(src + s * 512);
    };
    auto stash = [&](int stage) {
        __half* dst = lds_kv + (stage * 4 + wave) * TILE + lane * 8;
#pragma unroll
        for (int s = 0; s < KS; s++) *reinterpret_cast<half8*>(dst + s * 512) = st[s];
    };

    const float sl2 = p.scale * 1.4426950408889634f;
    float m_run[MT], l_run[MT];            // running maximum (log2 units) and sum per row
    float4v o_acc[MT][DT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        m_run[mt] = -INFINITY;
        l_run[mt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; dt++) o_acc[mt][dt] = (float4v){0.f, 0.f, 0.f, 0.f};
    }
    if (pair_lo < pair_hi) {
        issue(pair_lo);
        stash(0);
    }
    __syncthreads();
    for (int pr = pair_lo; pr < pair_hi; pr++) {
        const int cur = (pr - pair_lo) & 1;
        const bool more = pr + 1 < pair_hi;
        if (more) issue(pr + 1);
        __builtin_amdgcn_sched_barrier(0);               // keep the global requests ahead of this pair's work
        const __half* kb = lds_kv + (cur * 4) * TILE + lane * 8;
        const int blk0 = 2 * pr, blk1 = 2 * pr + 1;
        const bool has1 = blk1 < nblocks;
        // Sᵀ[key][row] of both blocks for every row tile; K fragments are read once per k-step
        float4v s0[MT], s1[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) { s0[mt] = (float4v){0.f, 0.f, 0.f, 0.f}; s1[mt] = (float4v){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const half8 k0 = *reinterpret_cast<const half8*>(kb + s * 512);
            const half8 k1 = *reinterpret_cast<const half8*>(kb + TILE + s * 512);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                s0[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, qf[mt][s], s0[mt], 0, 0, 0);
                s1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, qf[mt][s], s1[mt], 0, 0, 0);
            }
        }
        half8 pf[MT];
        float alpha[MT];
        bool rescale[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            // probabilities as exp2(s·scale·log₂e − m): one FMA per score feeds the native exp2
            const bool clear = blk1 * KV_BLOCK + KV_BLOCK - 1 <= vis_hi[mt] && blk0 * KV_BLOCK >= vis_lo[mt];
            float sc[8];
            float mx = -INFINITY;
            if (clear) {
                // both blocks lie below the causal diagonal (and inside the window) of every row of the tile: no masks
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    sc[r] = s0[mt][r];
                    sc[4 + r] = s1[mt][r];
                    mx = fmaxf(mx, fmaxf(sc[r], sc[4 + r]));
                }
                mx *= sl2;                                   // sl2 > 0: the maximum commutes with the scaling
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int kp0 = blk0 * KV_BLOCK + 4 * a + r, kp1 = blk1 * KV_BLOCK + 4 * a + r;
                    const bool ok0 = row_ok[mt] && kp0 <= row_pos[mt] && kp0 >= win_lo[mt];
                    const bool ok1 = row_ok[mt] && has1 && kp1 <= row_pos[mt] && kp1 >= win_lo[mt];
                    sc[r] = ok0 ? s0[mt][r] : -INFINITY;
                    sc[4 + r] = ok1 ? s1[mt][r] : -INFINITY;
                    mx = fmaxf(mx, fmaxf(sc[r], sc[4 + r]));
                }
                mx *= sl2;
            }
            mx = rows_reduce_max(mx);
            const float m_new = fmaxf(m_run[mt], mx);
            const float m_safe = m_new == -INFINITY ? 0.f : m_new;
            alpha[mt] = __builtin_amdgcn_exp2f(m_run[mt] - m_safe);
            rescale[mt] = __ballot(m_new != m_run[mt]) != 0;     // wave-uniform: some row's maximum moved
            float psum = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[j], sl2, -m_safe));      // masked: −inf·sl2 − m = −inf → 0
                psum += pv;
                pf[mt][j] = (_Float16)pv;
            }
            psum = rows_reduce_sum(psum);
            l_run[mt] = l_run[mt] * alpha[mt] + psum;
            m_run[mt] = m_new;
        }
        // Oᵀ[d][row] += Vᵀ[d][key]·Pᵀ[key][row]; V fragments are read once per d-tile pair
        // (the A operand of d-tile 2·ld + sub is the 8-byte half `sub` of the lane's V fragment of block 0 followed by the same
        // half of block 1: two ds_read_b64 straight into the operand registers, no element moves; a missing block 1 has P = 0)
        typedef uint32_t u32x4a __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int ld = 0; ld < KS; ld++) {
#pragma unroll
            for (int sub = 0; sub < 2; sub++) {
                const uint2 v0 = *reinterpret_cast<const uint2*>(kb + 2 * TILE + ld * 512 + 4 * sub);
                const uint2 v1 = *reinterpret_cast<const uint2*>(kb + 3 * TILE + ld * 512 + 4 * sub);
                const half8 vfrag = __builtin_bit_cast(half8, (u32x4a){v0.x, v0.y, v1.x, v1.y});
                const int dt = 2 * ld + sub;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    if (rescale[mt]) {
#pragma unroll
                        for (int r = 0; r < 4; r++) o_acc[mt][dt][r] *= alpha[mt];
                    }
                    o_acc[mt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfrag, pf[mt], o_acc[mt][dt], 0, 0, 0);
                }
            }
        }
        if (more) stash(cur ^ 1);
        __syncthreads();
    }
    // epilogue: wave-private transposition through the (now idle) stage memory — HD floats per row, 16 rows per wave
    float* lo = reinterpret_cast<float*>(smem) + wave * 16 * HD;
    const int row = lane >> 2, dl = lane & 3;
    constexpr int DPL = HD / 4;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
#pragma unroll
        for (int dt = 0; dt < DT; dt++) *reinterpret_cast<float4v*>(&lo[b * HD + dt * 16 + 4 * a]) = o_acc[mt][dt];
        const float l_row = __shfl(l_run[mt], row, 64);
        const int rho_o = ((unit * NW + wave) * MT + mt) * 16 + row;
        if (rho_o < rows_total) {
            const float inv = l_row > 0.f ? 1.0f / l_row : 0.f;
            __half* o = p.out + ((long)(tok0 + rho_o / G) * p.nq + kvh * G + rho_o % G) * HD + dl * DPL;
#pragma unroll
            for (int i = 0; i < DPL; i += 8) {
                half8 h;
#pragma unroll
                for (int j = 0; j < 8; j++) h[j] = (_Float16)(lo[row * HD + dl * DPL + i + j] * inv);
                *reinterpret_cast<half8*>(o + i) = h;
            }
        }
    }
}

// ── flash form, 64 keys per step ─────────────────────────────────────────────────────────────────────────────────────
// paged_prefill_attn_kernel above pays its softmax bookkeeping (row maximum and its cross-lane reduce, α, the rescale test, the
// row sum and ITS reduce, one barrier) once per 32 keys, and that vector work — not its 32 MFMAs — sets its pace.  Here a step is
// TWO block pairs (64 keys): one maximum / α / rescale per 64 keys, the row sums come out of the matrix pipe (an all-ones A
// operand against Pᵀ: every lane's accumulator then holds Σ_keys P of its query row, rescaled by α like O), and steps whose 64
// keys are visible to every row of a tile (all but the diagonal ones) run a mask-free body.  Same staging, tile layout and
// per-row arithmetic otherwise (Sᵀ = K·Qᵀ, lane-local exp2, Oᵀ += Vᵀ·Pᵀ); l is the sum of the fp16 P the numerator uses.
template <int HD>
__global__ __launch_bounds__(256, 2) void paged_prefill_attn64_kernel(AttnArgs p) {
    constexpr int NW = 4, MT = 2, DT = HD / 16, KS = HD / 32, TILE = 16 * HD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __half* lds_kv = reinterpret_cast<__half*>(smem);          // [2 stages][2 pairs][K0, K1, V0, V1][TILE]; reused by the epilogue
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int a = lane >> 4, b = lane & 15;
    const int kvh = blockIdx.y;
    const int G = p.nq / p.nkv;
    int seq = -1, unit = 0;
    {
        const int w = gridDim.x - 1 - blockIdx.x;       // launch order reversed: the longest key ranges (last units of a causal prompt) start first
        int base = 0;
        for (int s0 = 0; s0 < p.num_seqs; s0 += 64) {
            const int s = s0 + lane;
            int units = 0;
            if (s < p.num_seqs) {
                const int tiles = ((int)(p.cu_seqlens_q[s + 1] - p.cu_seqlens_q[s]) * G + 15) >> 4;
                units = (tiles + NW * MT - 1) / (NW * MT);
            }
            int incl = units;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            const int total = __shfl(incl, 63, 64);
            if (w < base + total) {
                const unsigned long long hit = __ballot(w < base + incl);
                const int src = __ffsll((long long)hit) - 1;
                seq = s0 + src;
                unit = w - (base + __shfl(incl - units, src, 64));
                break;
            }
            base += total;
        }
        if (seq < 0) return;
        seq = __builtin_amdgcn_readfirstlane(seq);          // wave-uniform by construction: scalar table reads below
        unit = __builtin_amdgcn_readfirstlane(unit);
    }
    const int tok0 = (int)p.cu_seqlens_q[seq];
    const int q_len = (int)p.cu_seqlens_q[seq + 1] - tok0;
    const int rows_total = q_len * G, tiles_s = (rows_total + 15) >> 4;
    const int pos0 = (int)p.pos_offsets[seq];

    int row_pos[MT], win_lo[MT];
    bool row_ok[MT];
    int vis_hi[MT], vis_lo[MT];
    half8 qf[MT][KS];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        const int tile = (unit * NW + wave) * MT + mt;
        const bool full = tile * 16 + 15 < rows_total;
        const int tmin = pos0 + (tile * 16) / G, tmax = pos0 + (tile * 16 + 15) / G;
        vis_hi[mt] = full ? tmin : -1;
        vis_lo[mt] = p.sliding_window > 0 ? max(0, tmax + 1 - p.sliding_window) : 0;
        const int rho = tile * 16 + b;
        row_ok[mt] = rho < rows_total;
        const int t_local = row_ok[mt] ? rho / G : 0, g = row_ok[mt] ? rho % G : 0;
        row_pos[mt] = pos0 + t_local;
        win_lo[mt] = p.sliding_window > 0 ? max(0, row_pos[mt] + 1 - p.sliding_window) : 0;
        const long q_off = ((long)(tok0 + t_local) * p.nq + kvh * G + g) * HD;
#pragma unroll
        for (int s = 0; s < KS; s++) qf[mt][s] = *reinterpret_cast<const half8*>(p.q + q_off + 32 * s + 8 * a);
    }
    const int tile_first = unit * NW * MT, tile_last = min(tile_first + NW * MT, tiles_s) - 1;
    const int t_first = (tile_first * 16) / G, t_last = min(q_len - 1, (tile_last * 16 + 15) / G);
    const int kv_end = pos0 + t_last + 1;
    const int kv_begin = p.sliding_window > 0 ? max(0, pos0 + t_first + 1 - p.sliding_window) : 0;
    const int nblocks = cdiv_dev(kv_end, KV_BLOCK);
    const int step_lo = (kv_begin / KV_BLOCK) / 4, step_hi = (nblocks + 3) / 4;       // steps of four blocks

    // staging: wave w moves tile w (0: K block 0, 1: K block 1, 2: V block 0, 3: V block 1) of both pairs of a step.  No load
    // sits under a condition (a conditional load makes hipcc drain vmcnt at the next use of ANY loaded register — here the Q
    // fragments, i.e. a full memory round trip inside every step): the prefetch index is clamped to the last step instead, and
    // the block ids are scalar loads (uniform address: lgkmcnt, not vmcnt).
    const int32_t* bt = p.block_tables + (long)seq * p.max_blocks;
    const __half* pool = wave < 2 ? p.k_pool : p.v_pool;
    half8 st[2][KS];
    int phys_next[2];                                    // physical blocks of the step to be requested next (read one step ahead)
    auto lookup = [&](int sp) {
#pragma unroll
        for (int j = 0; j < 2; j++)
            phys_next[j] = bt[min(4 * sp + 2 * j + (wave & 1), nblocks - 1)];    // past the end: a copy of the last block (finite data), masked below
    };
    auto issue = [&]() {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const __half* src = pool + ((long)phys_next[j] * p.nkv + kvh) * (long)TILE + lane * 8;
#pragma unroll
            for (int s = 0; s < KS; s++) st[j][s] = *reinterpret_cast<const half8*>(src + s * 512);
        }
    };
    // K tiles keep their fragment order; the two V tiles of a pair are interleaved so that the PV A operand of d-tile 2·ld + sub —
    // the 8-byte half `sub` of the lane's fragment of block 0 followed by the same half of block 1 — is ONE conflict-free
    // ds_read_b128: 16-byte slot ((sub·KS + ld)·64 + lane) of the pair's V region, block i in its half i
    auto stash = [&](int stage) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if (wave < 2) {
                __half* dst = lds_kv + ((stage * 2 + j) * 4 + wave) * TILE + lane * 8;
#pragma unroll
                for (int s = 0; s < KS; s++) *reinterpret_cast<half8*>(dst + s * 512) = st[j][s];
            } else {
                __half* dst = lds_kv + ((stage * 2 + j) * 4 + 2) * TILE + lane * 8 + (wave & 1) * 4;
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    const uint4 w4 = __builtin_bit_cast(uint4, st[j][s]);
                    *reinterpret_cast<uint2*>(dst + s * 512) = make_uint2(w4.x, w4.y);                 // sub 0
                    *reinterpret_cast<uint2*>(dst + (KS + s) * 512) = make_uint2(w4.z, w4.w);          // sub 1
                }
            }
        }
    };

    const float sl2 = p.scale * 1.4426950408889634f;
    float m_run[MT];
    float4v o_acc[MT][DT], l_acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        m_run[mt] = -INFINITY;
        l_acc[mt] = (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < DT; dt++) o_acc[mt][dt] = (float4v){0.f, 0.f, 0.f, 0.f};
    }
    const half8 ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
    if (step_lo < step_hi) {
        lookup(step_lo);
        issue();
        lookup(min(step_lo + 1, step_hi - 1));
        stash(0);
    }
    __syncthreads();
    for (int sp = step_lo; sp < step_hi; sp++) {
        const int cur = (sp - step_lo) & 1;
        issue();                                         // step sp + 1 (clamped), whose block ids arrived during the previous step
        lookup(min(sp + 2, step_hi - 1));
        __builtin_amdgcn_sched_barrier(0);
        const __half* kb = lds_kv + (cur * 8) * TILE + lane * 8;
        // Sᵀ[key][row]: s[mt][2·j + i] = block i of pair j
        float4v sv[MT][4];
        const float4v zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; s++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const half8 kf = *reinterpret_cast<const half8*>(kb + ((q >> 1) * 4 + (q & 1)) * TILE + s * 512);
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
                    sv[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[mt][s], s == 0 ? zero4 : sv[mt][q], 0, 0, 0);
            }
        }
        half8 pf[MT][2];
        const int key0 = sp * 4 * KV_BLOCK;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const bool clear = key0 + 4 * KV_BLOCK - 1 <= vis_hi[mt] && key0 >= vis_lo[mt];      // wave-uniform
            if (!clear) {
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int kp = key0 + q * KV_BLOCK + 4 * a + r;
                        const bool ok = row_ok[mt] && kp <= row_pos[mt] && kp >= win_lo[mt];
                        sv[mt][q][r] = ok ? sv[mt][q][r] : -INFINITY;
                    }
            }
            float mx = fmaxf(fmaxf(sv[mt][0][0], sv[mt][0][1]), fmaxf(sv[mt][0][2], sv[mt][0][3]));
#pragma unroll
            for (int q = 1; q < 4; q++) mx = fmaxf(mx, fmaxf(fmaxf(sv[mt][q][0], sv[mt][q][1]), fmaxf(sv[mt][q][2], sv[mt][q][3])));
            mx = rows_reduce_max(mx * sl2);                       // sl2 > 0: the maximum commutes with the scaling
            const float m_new = fmaxf(m_run[mt], mx);
            const float m_safe = m_new == -INFINITY ? 0.f : m_new;
            if (__ballot(m_new != m_run[mt]) != 0) {              // wave-uniform: some row's maximum moved
                const float alpha = __builtin_amdgcn_exp2f(m_run[mt] - m_safe);
#pragma unroll
                for (int dt = 0; dt < DT; dt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) o_acc[mt][dt][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 4; r++) l_acc[mt][r] *= alpha;
                m_run[mt] = m_new;
            }
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        pf[mt][j][4 * i + r] = (_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(sv[mt][2 * j + i][r], sl2, -m_safe));
        }
        // Oᵀ[d][row] += Vᵀ[d][key]·Pᵀ[key][row]; the row sums ride along as one more "d tile" whose A operand is all ones
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const __half* vb = kb + (j * 4 + 2) * TILE;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) l_acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf[mt][j], l_acc[mt], 0, 0, 0);
#pragma unroll
            for (int ld = 0; ld < KS; ld++) {
#pragma unroll
                for (int sub = 0; sub < 2; sub++) {
                    const half8 vfrag = *reinterpret_cast<const half8*>(vb + (sub * KS + ld) * 512);
                    const int dt = 2 * ld + sub;
#pragma unroll
                    for (int mt = 0; mt < MT; mt++)
                        o_acc[mt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfrag, pf[mt][j], o_acc[mt][dt], 0, 0, 0);
                }
            }
        }
        stash(cur ^ 1);
        __syncthreads();
    }
    float* lo = reinterpret_cast<float*>(smem) + wave * 16 * HD;
    const int row = lane >> 2, dl = lane & 3;
    constexpr int DPL = HD / 4;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
#pragma unroll
        for (int dt = 0; dt < DT; dt++) *reinterpret_cast<float4v*>(&lo[b * HD + dt * 16 + 4 * a]) = o_acc[mt][dt];
        const float l_row = __shfl(l_acc[mt][0], row, 64);
        const int rho_o = ((unit * NW + wave) * MT + mt) * 16 + row;
        if (rho_o < rows_total) {
            const float inv = l_row > 0.f ? 1.0f / l_row : 0.f;
            __half* o = p.out + ((long)(tok0 + rho_o / G) * p.nq + kvh * G + rho_o % G) * HD + dl * DPL;
#pragma unroll
            for (int i = 0; i < DPL; i += 8) {
                half8 h;
#pragma unroll
                for (int j = 0; j < 8; j++) h[j] = (_Float16)(lo[row * HD + dl * DPL + i + j] * inv);
                *reinterpret_cast<half8*>(o + i) = h;
            }
        }
    }
}

// ── short prompts: the sequence's whole K/V resident in LDS ─────────────────────────────────────────────────────────────
// A 256-token prompt is ≤ 4 steps of 64 keys per workgroup of the flash form, most of them on the diagonal: its prologue (unit
// scan, Q, first stage, barrier) and epilogue are most of its time (32 × 256 tokens: 79 µs for 17 GFLOP).  When a sequence's
// keys fit — kv ≤ 256 at head_dim 128: 2 × 64 KiB — ONE copy of its K/V tiles is staged per workgroup (pairs of V tiles
// interleaved as in the 64-key form), one barrier, and then the 8 waves walk the sequence's row tiles two at a time with
// no further synchronisation: wave w of workgroup j ∈ {0, 1} takes tile pairs 2·(w + 8·i) + j … (interleaved, so the causal
// work is balanced).  PV runs with the operands swapped — A = P (the lane's row b, keys (a, j)), B = the V fragment (the very
// registers that serve as A for Oᵀ) — so the result is O[row 4a + r][d = b]: output rows are stored straight from the
// accumulators, no transposition through LDS (there is none left).  Row maxima live in the S layout (lane column b = row); α
// reaches the O layout (row 4a + r) by four lane broadcasts, only when a maximum moved.
template <int HD>
__global__ __launch_bounds__(512, 1) void paged_prefill_attn_resident_kernel(AttnArgs p) {
    constexpr int NW = 8, MT = 2, DT = HD / 16, KS = HD / 32, TILE = 16 * HD, MAXB = 256 / KV_BLOCK;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __half* lds_k = reinterpret_cast<__half*>(smem);            // [MAXB][TILE]
    __half* lds_v = lds_k + MAXB * TILE;                        // [MAXB / 2 pairs][2·TILE], interleaved
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int a = lane >> 4, b = lane & 15;
    const int seq = blockIdx.x >> 1, jpar = blockIdx.x & 1, kvh = blockIdx.y;
    const int G = p.nq / p.nkv;
    // row → (token, head of the group): a shift when the GQA group is a power of two (an emulated integer division costs ≈ 20
    // VALU, and the pass set-up and the row stores do ten of them per pass)
    const int gsh = (G & (G - 1)) == 0 ? __builtin_ctz(G) : -1;
    auto divmod_g = [&](int rho, int& t, int& g) {
        if (gsh >= 0) { t = rho >> gsh; g = rho & (G - 1); }
        else { t = rho / G; g = rho - t * G; }
    };
    const int tok0 = (int)p.cu_seqlens_q[seq];
    const int q_len = (int)p.cu_seqlens_q[seq + 1] - tok0;
    const int rows_total = q_len * G, tiles_s = (rows_total + 15) >> 4;
    const int pos0 = (int)p.pos_offsets[seq];
    const int kv_end = pos0 + q_len;
    const int nblocks = cdiv_dev(kv_end, KV_BLOCK);             // ≤ MAXB (the launcher checks)
    // stage K and V of every block: tile t of the pool → LDS (V: 8-byte halves of a pair's two blocks side by side)
    {
        // (every load of the wave requested before the first LDS write: one memory round trip, not one per tile)
        const int32_t* bt = p.block_tables + (long)seq * p.max_blocks;
        constexpr int NT = 2 * MAXB / NW;                       // tile copies per wave (K and V of MAXB blocks over NW waves)
        long phys[NT];
#pragma unroll
        for (int i = 0; i < NT; i++) phys[i] = bt[min((wave + NW * i) >> 1, nblocks - 1)];
        half8 st[NT][KS];
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int t = wave + NW * i, isv = t & 1;
            const __half* src = (isv ? p.v_pool : p.k_pool) + (phys[i] * p.nkv + kvh) * (long)TILE + lane * 8;
#pragma unroll
            for (int s = 0; s < KS; s++) st[i][s] = *reinterpret_cast<const half8*>(src + s * 512);
        }
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int t = wave + NW * i, blk = t >> 1, isv = t & 1;
            if (blk < nblocks) {
                if (!isv) {
                    __half* dst = lds_k + blk * TILE + lane * 8;
#pragma unroll
                    for (int s = 0; s < KS; s++) *reinterpret_cast<half8*>(dst + s * 512) = st[i][s];
                } else {
                    __half* dst = lds_v + (blk >> 1) * 2 * TILE + lane * 8 + (blk & 1) * 4;
#pragma unroll
                    for (int s = 0; s < KS; s++) {
                        const uint4 w4 = __builtin_bit_cast(uint4, st[i][s]);
                        *reinterpret_cast<uint2*>(dst + s * 512) = make_uint2(w4.x, w4.y);
                        *reinterpret_cast<uint2*>(dst + (KS + s) * 512) = make_uint2(w4.z, w4.w);
                    }
                }
            }
        }
        // a step spans four blocks: V of the blocks between the context's end and the step boundary must read as zeros (their
        // P is 0, but 0 × a stale Inf / NaN is NaN); stale K only yields scores that the masks discard
        for (int blk = nblocks + wave; blk < min(MAXB, (nblocks + 3) & ~3); blk += NW) {
            __half* dst = lds_v + (blk >> 1) * 2 * TILE + lane * 8 + (blk & 1) * 4;
#pragma unroll
            for (int s = 0; s < 2 * KS; s++) *reinterpret_cast<uint2*>(dst + s * 512) = make_uint2(0u, 0u);
        }
    }
    __syncthreads();
    const float sl2 = p.scale * 1.4426950408889634f;
    const half8 ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
    const float4v zero4 = {0.f, 0.f, 0.f, 0.f};
    // this wave's passes: tile pair index tp = jpar + 2·(wave + NW·i) → tiles 2·tp, 2·tp + 1
    for (int tp = jpar + 2 * wave; 2 * tp < tiles_s; tp += 2 * NW) {
        int row_pos[MT], win_lo[MT], vis_hi[MT], vis_lo[MT];
        bool row_ok[MT];
        half8 qf[MT][KS];
        int t_last = 0;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int tile = 2 * tp + mt;
            const bool full = tile * 16 + 15 < rows_total;
            int tq0, tq1, gq_;
            divmod_g(tile * 16, tq0, gq_);
            divmod_g(min(tile * 16 + 15, rows_total - 1), tq1, gq_);
            const int tmin = pos0 + tq0, tmax = pos0 + tq1;
            vis_hi[mt] = full ? tmin : -1;
            vis_lo[mt] = p.sliding_window > 0 ? max(0, tmax + 1 - p.sliding_window) : 0;
            const int rho = tile * 16 + b;
            row_ok[mt] = rho < rows_total;
            int t_local, g;
            divmod_g(row_ok[mt] ? rho : 0, t_local, g);
            row_pos[mt] = pos0 + t_local;
            win_lo[mt] = p.sliding_window > 0 ? max(0, row_pos[mt] + 1 - p.sliding_window) : 0;
            const long q_off = ((long)(tok0 + t_local) * p.nq + kvh * G + g) * HD;
#pragma unroll
            for (int s = 0; s < KS; s++) qf[mt][s] = *reinterpret_cast<const half8*>(p.q + q_off + 32 * s + 8 * a);
            if (tile * 16 < rows_total) t_last = max(t_last, tmax);
        }
        const int kv_hi = t_last + 1;                                           // keys [0, kv_hi) matter to this pass
        int tw0, gw_;
        divmod_g(2 * tp * 16, tw0, gw_);
        const int kv_lo = p.sliding_window > 0 ? max(0, pos0 + tw0 + 1 - p.sliding_window) : 0;
        const int step_lo = (kv_lo / KV_BLOCK) / 4, step_hi = (cdiv_dev(kv_hi, KV_BLOCK) + 3) / 4;
        float m_run[MT];
        float4v o_acc[MT][DT], l_acc[MT];                                       // O[row 4a + r][d = dt·16 + b], Σ P in the same layout
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            m_run[mt] = -INFINITY;
            l_acc[mt] = zero4;
#pragma unroll
            for (int dt = 0; dt < DT; dt++) o_acc[mt][dt] = zero4;
        }
        for (int sp = step_lo; sp < step_hi; sp++) {
            // (the element offset of the step is made opaque so that the 16 K reads use 16-bit immediates off ONE per-step base;
            // left alone, hipcc keeps a loop-wide base and spends a VALU add per ds_read on offsets beyond 64 KiB)
            int koff = (sp * 4) * TILE + lane * 8;
            asm volatile("" : "+v"(koff));
            const __half* kb = lds_k + koff;
            float4v sv[MT][4];
#pragma unroll
            for (int s = 0; s < KS; s++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const half8 kf = *reinterpret_cast<const half8*>(kb + q * TILE + s * 512);     // (blocks past the end: stale LDS, masked below)
#pragma unroll
                    for (int mt = 0; mt < MT; mt++)
                        sv[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[mt][s], s == 0 ? zero4 : sv[mt][q], 0, 0, 0);
                }
            }
            half8 pf[MT][2];
            const int key0 = sp * 4 * KV_BLOCK;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const bool clear = key0 + 4 * KV_BLOCK - 1 <= vis_hi[mt] && key0 >= vis_lo[mt];      // wave-uniform
                if (!clear) {
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int kp = key0 + q * KV_BLOCK + 4 * a + r;
                            const bool ok = row_ok[mt] && kp <= row_pos[mt] && kp >= win_lo[mt];
                            sv[mt][q][r] = ok ? sv[mt][q][r] : -INFINITY;
                        }
                }
                float mx = fmaxf(fmaxf(sv[mt][0][0], sv[mt][0][1]), fmaxf(sv[mt][0][2], sv[mt][0][3]));
#pragma unroll
                for (int q = 1; q < 4; q++) mx = fmaxf(mx, fmaxf(fmaxf(sv[mt][q][0], sv[mt][q][1]), fmaxf(sv[mt][q][2], sv[mt][q][3])));
                mx = rows_reduce_max(mx * sl2);
                const float m_new = fmaxf(m_run[mt], mx);
                const float m_safe = m_new == -INFINITY ? 0.f : m_new;
                if (__ballot(m_new != m_run[mt]) != 0) {                        // wave-uniform: some row's maximum moved
                    const float alpha = __builtin_amdgcn_exp2f(m_run[mt] - m_safe);      // lane column b ↔ row b
                    float al[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) al[r] = __shfl(alpha, 4 * a + r, 64);   // O layout: row 4a + r
#pragma unroll
                    for (int dt = 0; dt < DT; dt++)
#pragma unroll
                        for (int r = 0; r < 4; r++) o_acc[mt][dt][r] *= al[r];
#pragma unroll
                    for (int r = 0; r < 4; r++) l_acc[mt][r] *= al[r];
                    m_run[mt] = m_new;
                }
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            pf[mt][j][4 * i + r] = (_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(sv[mt][2 * j + i][r], sl2, -m_safe));
            }
            // O[row][d] += P[row][key]·V[key][d]: A = P fragment, B = V fragment (operands swapped against the Oᵀ forms)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                int voff = (sp * 2 + j) * 2 * TILE + lane * 8;
                asm volatile("" : "+v"(voff));
                const __half* vb = lds_v + voff;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) l_acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf[mt][j], ones, l_acc[mt], 0, 0, 0);
#pragma unroll
                for (int ld = 0; ld < KS; ld++)
#pragma unroll
                    for (int sub = 0; sub < 2; sub++) {
                        const half8 vfrag = *reinterpret_cast<const half8*>(vb + (sub * KS + ld) * 512);
                        const int dt = 2 * ld + sub;
#pragma unroll
                        for (int mt = 0; mt < MT; mt++)
                            o_acc[mt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf[mt][j], vfrag, o_acc[mt][dt], 0, 0, 0);
                    }
            }
        }
        // rows straight from the accumulators: lane (a, b) holds O[row 4a + r][dt·16 + b]
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int rho_o = (2 * tp + mt) * 16 + 4 * a + r;
                if (rho_o < rows_total) {
                    const float l_row = l_acc[mt][r];
                    const float inv = l_row > 0.f ? 1.0f / l_row : 0.f;
                    int to, go;
                    divmod_g(rho_o, to, go);
                    __half* o = p.out + ((long)(tok0 + to) * p.nq + kvh * G + go) * HD + b;
#pragma unroll
                    for (int dt = 0; dt < DT; dt++) o[dt * 16] = __float2half(o_acc[mt][dt][r] * inv);
                }
            }
    }
}

// merge grid.z partials: one thread per (row, dim)
template <int HD>
__global__ void paged_attn_reduce_kernel(AttnArgs p) {
    const int seq = blockIdx.x / p.tiles_per_seq, tile = blockIdx.x % p.tiles_per_seq;
    const int kvh = blockIdx.y;
    const int G = p.nq / p.nkv;
    const int tok0 = p.cu_seqlens_q ? (int)p.cu_seqlens_q[seq] : seq;
    const int q_len = p.cu_seqlens_q ? (int)p.cu_seqlens_q[seq + 1] - tok0 : 1;
    const int rows_total = q_len * G;
    const int row = threadIdx.x / (HD / 8), dl = threadIdx.x % (HD / 8);
    const int rho = tile * 16 + row;
    if (rho >= rows_total) return;
    const float* base = p.partial + ((long)blockIdx.x * p.nkv + kvh) * p.nsplit * 16 * (HD + 2) + row * (HD + 2);
    float M = -INFINITY;
    for (int z = 0; z < p.nsplit; z++) M = fmaxf(M, base[(long)z * 16 * (HD + 2) + HD]);
    const float Ms = M == -INFINITY ? 0.f : M;
    float L = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < p.nsplit; z++) {
        const float* src = base + (long)z * 16 * (HD + 2);
        float f = __expf(src[HD] - Ms);
        L += src[HD + 1] * f;
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] += src[dl * 8 + i] * f;
    }
    const float inv = L > 0.f ? 1.0f / L : 0.f;
    const int t_o = rho / G, g_o = rho % G;
    if (p.out_scatter) {
        __half* orow = p.out + (long)(tok0 + t_o) * p.nq * HD;
        const int32_t* sc = p.out_scatter + (kvh * G + g_o) * HD + dl * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) orow[sc[i]] = __float2half(acc[i] * inv);
        return;
    }
    __half* o = p.out + ((long)(tok0 + t_o) * p.nq + kvh * G + g_o) * HD + dl * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = __float2half(acc[i] * inv);
}

static int choose_splits(int num_tiles_total, int nkv, int max_kv_len) {
    // enough workgroups to cover the 256 CUs; at least 4 block pairs (128 keys) per wave per split
    long wgs = (long)num_tiles_total * nkv;
    int pairs = cdiv(cdiv(max_kv_len, KV_BLOCK), 2);
    int s = 1;
    // a split costs a second (reduce) launch ≈ 5 µs: only worth it for ≥ 512 keys per split
    while (wgs * s < 256 && pairs / (s * 2) >= 16 && s < 32) s *= 2;
    return s;
}

size_t paged_attention_workspace_bytes(int total_q_tokens, int num_heads, int head_dim, int max_kv_len) {
    // worst case: every (token, head) row in its own tile slot, 32 splits
    long rows = (long)total_q_tokens * num_heads + 16L * 4096;
    return (size_t)rows * 32 * (head_dim + 2) * sizeof(float);
}

struct FusedQkv {
    const __half* qkv; const __half* q_norm_w; const __half* k_norm_w; const float* cos_t; const float* sin_t;
    __half* k_pool_w; __half* v_pool_w; float eps; int qk_mode;
};

static int paged_attention_launch(const __half* q, const __half* k_pool, const __half* v_pool, __half* out,
                               const uint32_t* cu_seqlens_q, const uint32_t* pos_offsets, const uint32_t* kv_lens,
                               const int32_t* block_tables, int num_seqs, int total_q_tokens, int max_q_len,
                               int max_kv_len, int num_heads, int num_kv_heads, int head_dim,
                               int sliding_window, int block_size, int max_blocks_per_seq, float* workspace,
                               size_t workspace_bytes, hipStream_t s, const FusedQkv* fq = nullptr,
                               const int32_t* out_scatter = nullptr) {
    if (num_seqs <= 0 || total_q_tokens <= 0) return 0;
    FH_REQUIRE(!out_scatter || (!cu_seqlens_q && max_q_len == 1), "paged attention: scattered output is a decode-form feature");
    FH_REQUIRE(block_size == KV_BLOCK, "paged attention: block_size=%d unsupported (native layout uses 16)", block_size);
    FH_REQUIRE(num_kv_heads > 0 && num_heads % num_kv_heads == 0, "paged attention: nq=%d not a multiple of nkv=%d",
               num_heads, num_kv_heads);
    FH_REQUIRE(head_dim == 128 || head_dim == 64 || head_dim == 256, "paged attention: head_dim=%d unsupported", head_dim);
    const int G = num_heads / num_kv_heads;
    if (max_q_len <= 0) max_q_len = total_q_tokens - (num_seqs - 1);
    AttnArgs a{};
    a.q = q; a.k_pool = k_pool; a.v_pool = v_pool; a.out = out;
    a.cu_seqlens_q = cu_seqlens_q; a.pos_offsets = pos_offsets; a.kv_lens = kv_lens; a.block_tables = block_tables;
    a.num_seqs = num_seqs; a.nq = num_heads; a.nkv = num_kv_heads;
    a.tiles_per_seq = cdiv((long)max_q_len * G, 16);
    a.max_blocks = max_blocks_per_seq; a.sliding_window = sliding_window;
    a.scale = 1.0f / sqrtf((float)head_dim);
    a.out_scatter = out_scatter;
    // prefill-like batches (most sequences bring many rows): LDS-shared K/V form.  The token count bounds the work units.
    const Knobs& kn = knobs();                                   // read once at library load (knobs.h), never per launch
    const bool flash_off = kn.attn_no_flash;
    const long flash_min_rows = kn.attn_flash_min_rows;          // tests lower it
    // … and only when its workgroups (128 rows × one kv head each) cover the chip: a lone 256-token prompt has 64 of them and
    // is faster KV-split over 512 workgroups (TTFT 6.7 vs 7.1 ms)
    const long flash_min_wgs = kn.attn_flash_min_rows_set ? 1 : 256;
    // short prompts whose whole context fits one LDS image (kv ≤ 256 at head_dim 128): resident-K/V form, two workgroups per
    // (sequence, kv head).  max_kv_len bounds every sequence; needs ≥ 4 row tiles per sequence on average to pay off.
    if (!fq && cu_seqlens_q && head_dim == 128 && !kn.attn_no_resident && max_kv_len <= 256 && max_q_len >= 32 &&
        (long)total_q_tokens * G >= (long)num_seqs * 64 && (long)num_seqs * 2 * num_kv_heads >= kn.attn_resident_min_wgs) {
        const size_t lds_res = (size_t)2 * (256 / KV_BLOCK) * 16 * 128 * 2;       // K + V images: 128 KiB
        static bool attr_res = false;
        if (!attr_res) {
            FH_CHECK_HIP(hipFuncSetAttribute((const void*)paged_prefill_attn_resident_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_res));
            attr_res = true;
        }
        hipLaunchKernelGGL((paged_prefill_attn_resident_kernel<128>), dim3(num_seqs * 2, num_kv_heads, 1), dim3(512), lds_res, s, a);
        FH_CHECK_LAUNCH();
        form_hit(FORM_ATTN_RESIDENT);
        return 0;
    }
    if (!fq && cu_seqlens_q && !flash_off && (long)max_q_len * G >= flash_min_rows &&
        (long)total_q_tokens * 2 >= (long)num_seqs * max_q_len &&
        ((long)total_q_tokens * G / (head_dim == 256 ? 64 : 128)) * num_kv_heads >= flash_min_wgs) {
        const int mt = head_dim == 256 ? 1 : 2, unit_tiles = 4 * mt;
        const long tiles_bound = ((long)total_q_tokens * G) / 16 + num_seqs;
        const long units = cdiv(tiles_bound, (long)unit_tiles) + num_seqs;
        const dim3 grid((unsigned)units, num_kv_heads, 1);
        const size_t lds = (size_t)2 * 4 * 16 * head_dim * 2;      // two stages × four tiles
#define FH_FLASH(HDV, MTV)                                                                                             \
        {                                                                                                              \
            static bool attr_set = false;                                                                              \
            if (!attr_set && lds > 65536) {                                                                            \
                FH_CHECK_HIP(hipFuncSetAttribute((const void*)paged_prefill_attn_kernel<HDV, MTV>,                     \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
                attr_set = true;                                                                                       \
            }                                                                                                          \
            hipLaunchKernelGGL((paged_prefill_attn_kernel<HDV, MTV>), grid, dim3(256), lds, s, a);                     \
        }
        if (head_dim == 128 && !kn.attn_flash32) {           // 64 keys per step: two pairs per stage
            const size_t lds64 = 2 * lds;
            static bool attr64 = false;
            if (!attr64) {
                FH_CHECK_HIP(hipFuncSetAttribute((const void*)paged_prefill_attn64_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64));
                attr64 = true;
            }
            hipLaunchKernelGGL((paged_prefill_attn64_kernel<128>), grid, dim3(256), lds64, s, a);
        } else if (head_dim == 128) FH_FLASH(128, 2) else if (head_dim == 64) FH_FLASH(64, 2) else FH_FLASH(256, 1)
#undef FH_FLASH
        FH_CHECK_LAUNCH();
        form_hit(FORM_ATTN_FLASH);
        return 0;
    }
    const int tiles = num_seqs * a.tiles_per_seq;
    int nsplit = choose_splits(tiles, num_kv_heads, max_kv_len);
    if (kn.attn_splits > 0) nsplit = kn.attn_splits;             // tuning override (development)
    size_t need = (size_t)tiles * num_kv_heads * nsplit * 16 * (head_dim + 2) * sizeof(float);
    if (nsplit > 1 && (workspace == nullptr || need > workspace_bytes)) nsplit = 1;
    a.nsplit = nsplit;
    a.partial = workspace;
    if (fq) {
        a.qkv = fq->qkv; a.q_norm_w = fq->q_norm_w; a.k_norm_w = fq->k_norm_w; a.cos_t = fq->cos_t; a.sin_t = fq->sin_t;
        a.k_pool_w = fq->k_pool_w; a.v_pool_w = fq->v_pool_w; a.eps = fq->eps; a.qk_mode = fq->qk_mode;
    }
    dim3 grid(tiles, num_kv_heads, nsplit);
    // prefill (≥ 4 row tiles per sequence, no KV split): the waves of a workgroup take consecutive row tiles
    // (only when that still leaves ≥ 2 workgroups per CU: a single 256-token prompt is faster KV-split, 7.5 vs 7.85 ms TTFT)
    const long rs_min_wgs = kn.attn_rs_min_wgs;                  // tests lower it
    const bool rs = !fq && nsplit == 1 && a.tiles_per_seq >= 4 && (long)num_seqs * cdiv(a.tiles_per_seq, 4) * num_kv_heads >= rs_min_wgs &&
                    !kn.attn_no_rs;
    if (rs) grid = dim3(num_seqs * cdiv(a.tiles_per_seq, 4), num_kv_heads, 1);
    // ragged batch without a KV split: enumerate only the real work units (upper bound from the token count)
    if (cu_seqlens_q && nsplit == 1 && a.tiles_per_seq > 1) {
        const long tiles_bound = ((long)total_q_tokens * G) / 16 + num_seqs;
        const long units = rs ? cdiv(tiles_bound, 4) + num_seqs : tiles_bound;
        if (units < (long)grid.x) {
            a.compact = 1;
            grid.x = (unsigned)units;
        }
    }
    // decode with ≥ 8 block pairs per split: 8 waves per workgroup (more loads in flight per CU)
    const bool wide = max_q_len == 1 && cdiv(cdiv(max_kv_len, KV_BLOCK), 2) / nsplit >= 8 && !kn.attn_narrow;
#define FH_ATTN(HDV)                                                                              \
    if (fq && wide) hipLaunchKernelGGL((paged_attn_kernel<HDV, true, 8>), grid, dim3(512), 0, s, a);       \
    else if (fq) hipLaunchKernelGGL((paged_attn_kernel<HDV, true, 4>), grid, dim3(256), 0, s, a);          \
    else if (rs) hipLaunchKernelGGL((paged_attn_kernel<HDV, false, 4, true>), grid, dim3(256), 0, s, a);   \
    else if (wide) hipLaunchKernelGGL((paged_attn_kernel<HDV, false, 8>), grid, dim3(512), 0, s, a);       \
    else hipLaunchKernelGGL((paged_attn_kernel<HDV, false, 4>), grid, dim3(256), 0, s, a);                 \
    FH_CHECK_LAUNCH();                                                                            \
    if (nsplit > 1) {                                                                             \
        hipLaunchKernelGGL(paged_attn_reduce_kernel<HDV>, dim3(tiles, num_kv_heads), dim3(16 * HDV / 8), 0, s, a); \
        FH_CHECK_LAUNCH();                                                                        \
    }
    if (head_dim == 128) { FH_ATTN(128) } else if (head_dim == 64) { FH_ATTN(64) } else { FH_ATTN(256) }
#undef FH_ATTN
    form_hit(fq ? (wide ? FORM_ATTN_FUSED_QKV_WIDE : FORM_ATTN_FUSED_QKV_NARROW)
                : rs ? FORM_ATTN_ROW_SPLIT : (wide ? FORM_ATTN_KV_WIDE : FORM_ATTN_KV_NARROW));
    if (nsplit > 1) form_hit(FORM_ATTN_SPLIT_REDUCE);
    return 0;
}

int paged_varlen_attention_f16(const __half* q, const __half* k_pool, const __half* v_pool, __half* out,
                               const uint32_t* cu_seqlens_q, const uint32_t* pos_offsets,
                               const int32_t* block_tables, int num_seqs, int total_q_tokens, int max_q_len,
                               int max_kv_len, int num_heads, int num_kv_heads, int head_dim,
                               int sliding_window, int block_size, int max_blocks_per_seq, float* workspace,
                               size_t workspace_bytes, hipStream_t s) {
    return paged_attention_launch(q, k_pool, v_pool, out, cu_seqlens_q, pos_offsets, nullptr, block_tables, num_seqs,
                                  total_q_tokens, max_q_len, max_kv_len, num_heads, num_kv_heads, head_dim,
                                  sliding_window, block_size, max_blocks_per_seq, workspace, workspace_bytes, s);
}

// paged_batched_decode_attention (traits.rs:1885): q_len = 1 per sequence, kv length given directly.
int paged_batched_decode_attention_f16(const __half* q, const __half* k_pool, const __half* v_pool, __half* out,
                                       const int32_t* block_tables, const uint32_t* valid_kv_lens, int num_seqs,
                                       int max_kv_len, int num_heads, int num_kv_heads, int head_dim, int block_size,
                                       int max_blocks_per_seq, float* workspace, size_t workspace_bytes,
                                       hipStream_t s, const int32_t* out_scatter) {
    return paged_attention_launch(q, k_pool, v_pool, out, nullptr, nullptr, valid_kv_lens, block_tables, num_seqs,
                                  num_seqs, 1, max_kv_len, num_heads, num_kv_heads, head_dim, 0, block_size,
                                  max_blocks_per_seq, workspace, workspace_bytes, s, nullptr, out_scatter);
}

// Decode step of one layer in ONE launch: split_qkv_norm_rope_into_paged_cache_varlen (q_len = 1 per sequence,
// position = valid_kv_lens[seq] − 1) + paged_batched_decode_attention.  Same results as the two-op chain.
int paged_decode_attention_fused_qkv_f16(const __half* qkv, const __half* q_norm_w, const __half* k_norm_w,
                                         const float* cos_t, const float* sin_t, float eps, int qk_mode, __half* k_pool,
                                         __half* v_pool, __half* out, const int32_t* block_tables,
                                         const uint32_t* valid_kv_lens, int num_seqs, int max_kv_len, int num_heads,
                                         int num_kv_heads, int head_dim, int sliding_window, int block_size,
                                         int max_blocks_per_seq, float* workspace, size_t workspace_bytes, hipStream_t s,
                                         const int32_t* out_scatter) {
    FH_REQUIRE(num_kv_heads > 0 && num_heads % num_kv_heads == 0 && num_heads / num_kv_heads <= 14,
               "fused decode attention: GQA group %d/%d must be <= 14", num_heads, num_kv_heads);
    FH_REQUIRE(qk_mode >= 0 && qk_mode <= 3, "fused decode attention: qk_mode=%d out of range", qk_mode);
    FusedQkv fq{qkv, q_norm_w, k_norm_w, cos_t, sin_t, k_pool, v_pool, eps, qk_mode};
    return paged_attention_launch(nullptr, k_pool, v_pool, out, nullptr, nullptr, valid_kv_lens, block_tables, num_seqs,
                                  num_seqs, 1, max_kv_len, num_heads, num_kv_heads, head_dim, sliding_window, block_size,
                                  max_blocks_per_seq, workspace, workspace_bytes, s, &fq, out_scatter);
}

}  // namespace fh
