// Merge of a token's Q per-part router candidate lists, one wave, lane-parallel: shared by role B of the decode chain
// (chain.hip, the parts meet inside the launch) and by the grouped GEMMs that take the lists themselves (w4_gemm.hip).
// Selection order (ferrum-models/src/moe/router.rs:159-178): logit descending, ties → lower expert id; combine weights
// (router.rs:141-157,180-193): softmax over ALL experts from the parts' (max, Σexp) statistics, optionally renormalised over
// the K picked, summed in pick order.
#pragma once
#include "common.h"

namespace fh {

// Lanes [0, ncand) hold one candidate each, c = (expert id << 32) | logit bits (an unused slot: id 0x7fffffff, logit −inf),
// lanes [0, Q) the statistics of part `lane`.  Returns the lane's rank in the merged order; `weight` (only when
// want_weights, a wave-uniform flag) is the lane's combine weight if rank < top_k.
__device__ __forceinline__ int route_merge_token(unsigned long long c, int ncand, float pmx, float psum, int Q, int top_k,
                                                 int norm_topk, bool want_weights, float* weight) {
    const int lane = threadIdx.x & 63;
    const float my_l = __uint_as_float((unsigned)c);
    const int my_id = (int)(c >> 32);
    int rank = 0;
    for (int j = 0; j < ncand; j++) {
        const float lj = __uint_as_float(__builtin_amdgcn_readlane((int)(unsigned)c, j));
        const int ij = __builtin_amdgcn_readlane((int)(c >> 32), j);
        rank += (lj > my_l || (lj == my_l && ij < my_id)) ? 1 : 0;
    }
    if (want_weights) {
        const float gmax = wave_reduce_max(lane < Q ? pmx : -INFINITY);
        const float gsum = wave_reduce_sum(lane < Q ? psum * expf(pmx - gmax) : 0.f);
        const float pr = lane < ncand ? expf(my_l - gmax) * (1.0f / gsum) : 0.f;
        float sel_sum = 0.f;
        for (int k = 0; k < top_k; k++) sel_sum += wave_reduce_sum((lane < ncand && rank == k) ? pr : 0.f);
        float ww = pr;
        if (norm_topk) ww = sel_sum > 0.f ? ww * (1.0f / sel_sum) : 1.0f / (float)top_k;
        *weight = ww;
    }
    return rank;
}

// Two tokens per pass, one per half of the wave (lane = 32·half + 8·part + slot), for the grouped GEMM prologues that merge
// the lists of a few tokens themselves: a candidate's rank is its slot in its own (sorted) list plus, for every other part, the
// length of that list's prefix that comes before it — a 4-probe binary search through ds_bpermute instead of route_merge_token's
// 32 readlane steps (≈ 1 µs per token, too long for a prologue that every workgroup of the launch runs).  Same ranks, and —
// for the publishing workgroup — the same combine weights bit for bit (Q ≤ 4: the statistics are summed as the wave
// reduction sums them, (s0 + s1) + (s2 + s3)).
__device__ __forceinline__ int route_merge_pair(unsigned long long c, int Q, int* id_out) {
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, li = lane & 31, q = li >> 3, j = li & 7;
    const float my_l = __uint_as_float((unsigned)c);
    const int my_id = (int)(c >> 32);
    *id_out = my_id;
    // the searches through the other parts' lists are independent: their probes advance together (three ds_bpermute pairs in
    // flight per step instead of a chain of twelve)
    int lo[3] = {0, 0, 0}, hi[3] = {8, 8, 8}, base[3];
#pragma unroll
    for (int d = 0; d < 3; d++) base[d] = half * 32 + ((q + d + 1) % Q) * 8;
#pragma unroll
    for (int it = 0; it < 4; it++) {
        float lj[3];
        int ij[3], mid[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            mid[d] = (lo[d] + hi[d]) >> 1;
            const int src = base[d] + (mid[d] < 7 ? mid[d] : 7);
            lj[d] = __uint_as_float((unsigned)__shfl((int)(unsigned)c, src, 64));
            ij[d] = __shfl((int)(c >> 32), src, 64);
        }
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const bool act = lo[d] < hi[d];
            const bool beats = lj[d] > my_l || (lj[d] == my_l && ij[d] < my_id);
            if (act && beats) lo[d] = mid[d] + 1;
            if (act && !beats) hi[d] = mid[d];
        }
    }
    int rank = j;
#pragma unroll
    for (int d = 0; d < 3; d++) rank += d + 1 < Q ? lo[d] : 0;
    return rank;
}

// combine weight of the lane's candidate (valid where rank < top_k); st = the token's (max, Σexp) statistics, Q pairs
__device__ __forceinline__ float route_merge_pair_weight(unsigned long long c, int rank, bool valid, const float* __restrict__ st, int Q,
                                                         int top_k, int norm_topk) {
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    float mx[4], sm[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { mx[i] = i < Q ? st[2 * i] : -INFINITY; sm[i] = i < Q ? st[2 * i + 1] : 0.f; }
    const float gmax = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]));
    float e[4];
#pragma unroll
    for (int i = 0; i < 4; i++) e[i] = i < Q ? sm[i] * expf(mx[i] - gmax) : 0.f;
    const float gsum = (e[0] + e[1]) + (e[2] + e[3]);
    const float pr = valid ? expf(__uint_as_float((unsigned)c) - gmax) * (1.0f / gsum) : 0.f;
    float sel_sum = 0.f;
    for (int k = 0; k < top_k; k++) {
        const unsigned long long bal = __ballot(valid && rank == k);
        const unsigned mine = (unsigned)(bal >> (32 * half));
        const int src = 32 * half + (mine ? __ffs((int)mine) - 1 : 0);
        const float pk = __shfl(pr, src, 64);
        sel_sum += mine ? pk : 0.f;
    }
    float ww = pr;
    if (norm_topk) ww = sel_sum > 0.f ? ww * (1.0f / sel_sum) : 1.0f / (float)top_k;
    return ww;
}

}  // namespace fh
