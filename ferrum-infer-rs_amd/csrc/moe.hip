// MoE routing on device: router top-k softmax, align-block-size, weighted combine.
//
// Reference: `BackendMoeFused::route_topk_softmax` (ferrum-kernels/src/backend/capabilities.rs:334;
// CUDA kernels/moe_router.cu:32; CPU ferrum-models/src/moe/router.rs:113-195),
// `moe_align_block_size_pair_ids` (capabilities.rs:449; kernels/moe_align_block_size_pair_ids.cu:13),
// `moe_combine` / `weighted_sum_batched` (capabilities.rs:684,560; kernels/moe_combine.cu:29,62).
// Integer outputs are bit-exact with the host plan (dispatch.rs:1408-1461): slots inside an expert
// are in ascending pair id (the CUDA lane's atomicAdd order is unspecified; ours is not).
#include "common.h"
#include "kernels.h"

namespace fh {

// One wave per token.  E ≤ 512.  f32 softmax, k argmax-mask passes, ties → lowest index.
template <typename T>
__global__ __launch_bounds__(64) void moe_route_kernel(const T* __restrict__ logits, int32_t* __restrict__ ids,
                                                       float* __restrict__ weights, int num_experts, int top_k,
                                                       int norm_topk_prob) {
    constexpr int PER = 8;   // experts per lane (E ≤ 512)
    const int tok = blockIdx.x, lane = threadIdx.x;
    const T* row = logits + (long)tok * num_experts;
    float v[PER];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        int e = lane + i * 64;
        v[i] = e < num_experts ? (float)row[e] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_reduce_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        int e = lane + i * 64;
        v[i] = e < num_experts ? expf(v[i] - mx) : 0.f;
        sum += v[i];
    }
    sum = wave_reduce_sum(sum);
    const float inv_sum = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        int e = lane + i * 64;
        v[i] = e < num_experts ? v[i] * inv_sum : -INFINITY;
    }
    float sel_sum = 0.f;
    float my_w = 0.f;   // lane k keeps the k-th selected weight
    int my_id = 0;
    for (int k = 0; k < top_k; k++) {
        float best = -INFINITY;
        int best_idx = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            int e = lane + i * 64;
            if (v[i] > best) { best = v[i]; best_idx = e; }   // ascending e within a lane → first max
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            float ob = __shfl_xor(best, off, 64);
            int oi = __shfl_xor(best_idx, off, 64);
            if (ob > best || (ob == best && oi < best_idx)) { best = ob; best_idx = oi; }
        }
        if (best_idx == 0x7fffffff) best_idx = 0;   // all -inf (router.rs keeps best_idx = 0)
        sel_sum += best;
        if (lane == (k & 63)) { my_w = best; my_id = best_idx; }
#pragma unroll
        for (int i = 0; i < PER; i++)
            if (lane + i * 64 == best_idx) v[i] = -INFINITY;
        if (top_k > 64 && (k & 63) == 63) {}  // top_k ≤ 64 enforced on host
    }
    if (lane < top_k) {
        float w = my_w;
        if (norm_topk_prob) w = sel_sum > 0.f ? w * (1.0f / sel_sum) : 1.0f / (float)top_k;
        ids[(long)tok * top_k + lane] = my_id;
        weights[(long)tok * top_k + lane] = w;
    }
}

int moe_route_topk_softmax_f16(const __half* logits, int32_t* expert_ids, float* expert_weights, int tokens,
                               int num_experts, int top_k, int norm_topk_prob, hipStream_t s) {
    if (tokens <= 0) return 0;
    FH_REQUIRE(num_experts > 0 && num_experts <= 512, "moe route: num_experts=%d must be in [1,512]", num_experts);
    FH_REQUIRE(top_k > 0 && top_k <= num_experts && top_k <= 64, "moe route: top_k=%d invalid for %d experts", top_k, num_experts);
    hipLaunchKernelGGL(moe_route_kernel<__half>, dim3(tokens), dim3(64), 0, s, logits, expert_ids, expert_weights,
                       num_experts, top_k, norm_topk_prob);
    FH_CHECK_LAUNCH();
    return 0;
}
int moe_route_topk_softmax_f32(const float* logits, int32_t* expert_ids, float* expert_weights, int tokens,
                               int num_experts, int top_k, int norm_topk_prob, hipStream_t s) {
    if (tokens <= 0) return 0;
    FH_REQUIRE(num_experts > 0 && num_experts <= 512, "moe route: num_experts=%d must be in [1,512]", num_experts);
    FH_REQUIRE(top_k > 0 && top_k <= num_experts && top_k <= 64, "moe route: top_k=%d invalid for %d experts", top_k, num_experts);
    hipLaunchKernelGGL(moe_route_kernel<float>, dim3(tokens), dim3(64), 0, s, logits, expert_ids, expert_weights,
                       num_experts, top_k, norm_topk_prob);
    FH_CHECK_LAUNCH();
    return 0;
}

// ── prefill router in ONE launch: logits = x · Wrᵀ on the matrix cores, then softmax + top-k, 32 tokens per workgroup ──────────
// Replaces f16t GEMM (split-K 4) → reduce → one-wave-per-token top-k (27 + 7 + 16 µs per layer at 8192 tokens): the logits never
// leave the CU.  E = 128 (eight 16-expert tiles of the f16t layout, w4_gemm.hip), K split over the four waves (each wave: 32
// tokens × 128 experts × K/4), partial sums meet in LDS in wave order; then 16 lanes per token (8 experts each, e = l + 16·i), four
// tokens per wave pass: f32 softmax over all experts, k argmax-mask passes (strict > within a lane's ascending experts, ties →
// lower id across lanes), renormalised by the selected sum — router.rs:113-195 / moe_router.cu:32 as moe_route_kernel above.
constexpr int RG_TOK = 32, RG_LD = 132;
template <int CTRL>
__device__ __forceinline__ void rg_argmax_step(float& best, int& idx) {      // (larger value, then smaller index): a total order, any pairing
    const float ob = dpp_move<CTRL>(best);
    const int oi = __builtin_amdgcn_update_dpp(0, idx, CTRL, 0xF, 0xF, true);
    if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
}
__global__ __launch_bounds__(256) void moe_route_gemm_topk_kernel(const __half* __restrict__ x, const __half* __restrict__ wt,
                                                                   int32_t* __restrict__ ids, float* __restrict__ weights, int T,
                                                                   int K, int top_k, int norm_topk_prob) {
    extern __shared__ __attribute__((aligned(16))) float rg_lds[];     // [4][32][RG_LD] partials; the logits replace wave 0's
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int a = lane >> 4, b = lane & 15;
    const int t0 = blockIdx.x * RG_TOK;
    const int ksteps = K >> 5, s0 = ksteps * wave / 4, s1 = ksteps * (wave + 1) / 4;
    const __half* xrow[2];
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        const int r = t0 + mt * 16 + b;
        xrow[mt] = x + (long)(r < T ? r : T - 1) * K + 8 * a;
    }
    const __half* wl = wt + (long)lane * 8;
    float4v acc[2][8];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 8; nt++) acc[mt][nt] = (float4v){0.f, 0.f, 0.f, 0.f};
    // two k-steps per ring slot, two slots: 2–4 k-steps (20–40 KiB per wave, the router from L2) in flight
    half8 bw[2][2][8], ax[2][2][2];
    auto issue = [&](int buf, int s) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int su = s + u < s1 ? s + u : s1 - 1;
#pragma unroll
            for (int mt = 0; mt < 2; mt++) ax[buf][u][mt] = *reinterpret_cast<const half8*>(xrow[mt] + su * 32);
#pragma unroll
            for (int nt = 0; nt < 8; nt++) bw[buf][u][nt] = *reinterpret_cast<const half8*>(wl + ((long)nt * ksteps + su) * 512);
        }
    };
    auto consume = [&](int buf, int s) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (s + u < s1) {
#pragma unroll
                for (int nt = 0; nt < 8; nt++)
#pragma unroll
                    for (int mt = 0; mt < 2; mt++)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ax[buf][u][mt], bw[buf][u][nt], acc[mt][nt], 0, 0, 0);
            }
        }
    };
    if (s0 < s1) issue(0, s0);
    for (int s = s0; s < s1; s += 4) {
        if (s + 2 < s1) issue(1, s + 2);
        consume(0, s);
        if (s + 4 < s1) issue(0, s + 4);
        if (s + 2 < s1) consume(1, s + 2);
    }
    float* part = rg_lds + (long)wave * RG_TOK * RG_LD;
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 8; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) part[(mt * 16 + 4 * a + r) * RG_LD + nt * 16 + b] = acc[mt][nt][r];
    __syncthreads();
    float* logit = rg_lds;                 // (in place of wave 0's partials: every thread reads its own 16 sums' inputs first)
    {
        const int tok = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 16;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            float v = rg_lds[tok * RG_LD + c0 + j];
#pragma unroll
            for (int w = 1; w < 4; w++) v += rg_lds[(w * RG_TOK + tok) * RG_LD + c0 + j];
            logit[tok * RG_LD + c0 + j] = v;
        }
    }
    __syncthreads();
    const int grp = lane >> 4, l16 = lane & 15;
    for (int pass = 0; pass < 2; pass++) {
        const int tl = wave * 8 + pass * 4 + grp, tok = t0 + tl;
        float v[8];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 8; i++) { v[i] = logit[tl * RG_LD + l16 + 16 * i]; mx = fmaxf(mx, v[i]); }
        // (reductions inside the token's 16-lane row by DPP — quad permutes, row_half_mirror, row_mirror — not ds_bpermute)
        mx = fmaxf(mx, dpp_move<0xB1>(mx)); mx = fmaxf(mx, dpp_move<0x4E>(mx)); mx = fmaxf(mx, dpp_move<0x141>(mx)); mx = fmaxf(mx, dpp_move<0x140>(mx));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; i++) { v[i] = expf(v[i] - mx); sum += v[i]; }
        sum += dpp_move<0xB1>(sum); sum += dpp_move<0x4E>(sum); sum += dpp_move<0x141>(sum); sum += dpp_move<0x140>(sum);
        const float inv_sum = 1.0f / sum;
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] *= inv_sum;
        float sel_sum = 0.f, my_w = 0.f;
        int my_id = 0;
        for (int k = 0; k < top_k; k++) {
            float best = -INFINITY;
            int best_idx = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (v[i] > best) { best = v[i]; best_idx = l16 + 16 * i; }
            rg_argmax_step<0xB1>(best, best_idx); rg_argmax_step<0x4E>(best, best_idx);
            rg_argmax_step<0x141>(best, best_idx); rg_argmax_step<0x140>(best, best_idx);
            if (best_idx == 0x7fffffff) best_idx = 0;
            sel_sum += best;
            if (l16 == k) { my_w = best; my_id = best_idx; }
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (l16 + 16 * i == best_idx) v[i] = -INFINITY;
        }
        if (l16 < top_k && tok < T) {
            float w = my_w;
            if (norm_topk_prob) w = sel_sum > 0.f ? w * (1.0f / sel_sum) : 1.0f / (float)top_k;
            ids[(long)tok * top_k + l16] = my_id;
            weights[(long)tok * top_k + l16] = w;
        }
    }
}

bool moe_route_gemm_topk_supports(int num_experts, int hidden, int top_k) {
    return num_experts == 128 && hidden % 128 == 0 && top_k >= 1 && top_k <= 16;
}
int moe_route_gemm_topk_f16(const __half* x, const __half* router_f16t, int32_t* expert_ids, float* expert_weights, int tokens,
                            int num_experts, int hidden, int top_k, int norm_topk_prob, hipStream_t s) {
    if (tokens <= 0) return 0;
    FH_REQUIRE(moe_route_gemm_topk_supports(num_experts, hidden, top_k), "route gemm+top-k: E=%d H=%d k=%d unsupported", num_experts, hidden, top_k);
    const size_t lds = (size_t)4 * RG_TOK * RG_LD * sizeof(float);
    static bool attr = false;
    if (!attr) { FH_CHECK_HIP(hipFuncSetAttribute((const void*)moe_route_gemm_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
    hipLaunchKernelGGL(moe_route_gemm_topk_kernel, dim3(cdiv(tokens, RG_TOK)), dim3(256), lds, s, x, router_f16t, expert_ids, expert_weights,
                       tokens, hidden, top_k, norm_topk_prob);
    FH_CHECK_LAUNCH();
    return 0;
}

// One workgroup per expert: histogram of every pair (LDS atomics, order-free) → prefixes →
// ordered compaction of this expert's pair ids (ballot prefix keeps ascending pair id).  Three outputs share the walk:
//   MODE 0  moe_align_block_size_pair_ids (capabilities.rs:449): sorted_token_ids holds pair ids p = token·top_k + slot
//   MODE 1  moe_align_block_size (capabilities.rs:429, kernels/moe_align_block_size.cu:1-30): sorted_token_ids holds the
//           UNPADDED packed row of each slot (expert e's region = unpadded_offset[e] + 0, 1, …) — no compaction needed
//   MODE 2  moe_build_pairs_by_token (capabilities.rs:410, kernels/moe_build_pairs.cu): pairs_by_token[p] = packed row of
//           pair p, packed_token_idx[row] = p / top_k, expert_offsets[E + 1] — the stable counting sort of
//           MoeBucketPlan::rebuild_into (moe/dispatch.rs:1408-1461); ids outside [0, E) get pairs_by_token = −1
constexpr int MAX_EXPERTS = 512;
constexpr int ALIGN_THREADS = 1024;
template <int MODE>
__global__ __launch_bounds__(ALIGN_THREADS) void moe_align_kernel(const int32_t* __restrict__ expert_ids,
                                                                  int32_t* __restrict__ sorted_token_ids,
                                                                  int32_t* __restrict__ block_ids,
                                                                  int32_t* __restrict__ total_post_pad, int n_pairs,
                                                                  int num_experts, int block_size, int sorted_max,
                                                                  int32_t* __restrict__ pairs_by_token,
                                                                  int32_t* __restrict__ packed_token_idx,
                                                                  int32_t* __restrict__ expert_offsets, int top_k) {
    // One workgroup per expert.  Every thread keeps CH = 64 CONSECUTIVE pair ids in registers (all 16 loads of a trip in flight
    // at once; a trip covers 65 536 pairs — an 8192-token prefill of a top-8 model is one trip), so the histogram and the ordered
    // compaction (ascending pair id = thread order, then element order) both run from registers with ONE workgroup scan per
    // trip.  (Before: 64 serialised load → LDS-atomic round trips for the histogram and 16 scan trips of two barriers each:
    // 52 µs per layer at 8192 tokens.)
    __shared__ int counts[MAX_EXPERTS];
    __shared__ int wave_cnt[ALIGN_THREADS / 64];
    __shared__ int s_offset, s_total, s_unpadded;
    constexpr int NT = ALIGN_THREADS, NWV = ALIGN_THREADS / 64, CH = 64, TRIP = NT * CH;
    const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < num_experts; i += NT) counts[i] = 0;
    __syncthreads();
    const bool vec_ok = (reinterpret_cast<uintptr_t>(expert_ids) & 15) == 0;
    int ids[CH];
    auto load_trip = [&](int trip0) {
        const int p = trip0 + tid * CH;
        if (vec_ok && p + CH <= n_pairs) {
#pragma unroll
            for (int q = 0; q < CH / 4; q++) {
                const int4 v = *reinterpret_cast<const int4*>(expert_ids + p + 4 * q);
                ids[4 * q] = v.x; ids[4 * q + 1] = v.y; ids[4 * q + 2] = v.z; ids[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int jj = 0; jj < CH; jj++) ids[jj] = p + jj < n_pairs ? expert_ids[p + jj] : -2;      // −2: beyond the input
        }
    };
    const int ntrips = (n_pairs + TRIP - 1) / TRIP;
    for (int t = 0; t < ntrips; t++) {
        load_trip(t * TRIP);
        const int p = t * TRIP + tid * CH;
#pragma unroll
        for (int jj = 0; jj < CH; jj++) {
            const int x = ids[jj];
            if (x >= 0 && x < num_experts) atomicAdd(&counts[x], 1);
            else if (MODE == 2 && e == 0 && x != -2) pairs_by_token[p + jj] = -1;
        }
    }
    __syncthreads();
    if (wave == 0) {        // exclusive scans over the experts (padded and unpadded), lanes take contiguous runs of experts
        const int per = (num_experts + 63) / 64, i0 = lane * per;
        int pad_sum = 0, raw_sum = 0;
        for (int i = i0; i < min(i0 + per, num_experts); i++) {
            pad_sum += ((counts[i] + block_size - 1) / block_size) * block_size;
            raw_sum += counts[i];
        }
        int pad_incl = pad_sum, raw_incl = raw_sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int tp = __shfl_up(pad_incl, off, 64), tr = __shfl_up(raw_incl, off, 64);
            if (lane >= off) { pad_incl += tp; raw_incl += tr; }
        }
        int pad_acc = pad_incl - pad_sum, raw_acc = raw_incl - raw_sum;
        for (int i = i0; i < min(i0 + per, num_experts); i++) {
            if (i == e) { s_offset = pad_acc; s_unpadded = raw_acc; }
            if (MODE == 2 && e == 0) expert_offsets[i] = raw_acc;
            pad_acc += ((counts[i] + block_size - 1) / block_size) * block_size;
            raw_acc += counts[i];
        }
        if (lane == 63) {
            s_total = pad_incl;
            if (MODE != 2 && e == 0) total_post_pad[0] = pad_incl;
            if (MODE == 2 && e == 0) expert_offsets[num_experts] = raw_incl;
        }
    }
    __syncthreads();
    const int offset = MODE == 2 ? s_unpadded : s_offset, total = s_total;
    const int cnt = counts[e];
    if (MODE != 2) {
        const int padded = ((cnt + block_size - 1) / block_size) * block_size;
        // sentinel for the padding tail of this expert and (striped) the unused end of the array
        for (int i = cnt + tid; i < padded; i += NT) sorted_token_ids[offset + i] = n_pairs;
        for (int i = total + e * NT + tid; i < sorted_max; i += NT * gridDim.x) sorted_token_ids[i] = n_pairs;
        for (int b = tid; b < padded / block_size; b += NT) block_ids[offset / block_size + b] = e;
    }
    if (MODE == 1) {
        for (int i = tid; i < cnt; i += NT) sorted_token_ids[offset + i] = s_unpadded + i;
        return;
    }
    // ordered compaction: one workgroup scan per trip
    int base = 0;
    for (int t = 0; t < ntrips; t++) {
        if (ntrips > 1) load_trip(t * TRIP);        // (a single trip is still in registers)
        const int p = t * TRIP + tid * CH;
        int c = 0;
#pragma unroll
        for (int jj = 0; jj < CH; jj++) c += ids[jj] == e ? 1 : 0;
        int incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int tv = __shfl_up(incl, off, 64);
            if (lane >= off) incl += tv;
        }
        if (lane == 63) wave_cnt[wave] = incl;
        __syncthreads();
        int before = 0, chunk_total = 0;
#pragma unroll
        for (int w = 0; w < NWV; w++) {
            const int wc = wave_cnt[w];
            if (w < wave) before += wc;
            chunk_total += wc;
        }
        int pos = offset + base + before + incl - c;
        if (c > 0) {
#pragma unroll
            for (int jj = 0; jj < CH; jj++)
                if (ids[jj] == e) {
                    if (MODE == 2) {
                        pairs_by_token[p + jj] = pos;
                        packed_token_idx[pos] = (p + jj) / top_k;
                        pos++;
                    } else {
                        sorted_token_ids[pos++] = p + jj;
                    }
                }
        }
        base += chunk_total;
        __syncthreads();
    }
}

int moe_align_block_size(const int32_t* expert_ids, int32_t* sorted_token_ids, int32_t* block_ids,
                         int32_t* total_post_pad, int batch_x_topk, int num_experts, int block_size,
                         int sorted_max, hipStream_t s) {
    FH_REQUIRE(num_experts > 0 && num_experts <= MAX_EXPERTS, "moe align: num_experts=%d must be in [1,%d]", num_experts, MAX_EXPERTS);
    FH_REQUIRE(block_size > 0, "moe align: block_size=%d", block_size);
    FH_REQUIRE(sorted_max >= batch_x_topk, "moe align: sorted_max=%d < pairs=%d", sorted_max, batch_x_topk);
    hipLaunchKernelGGL(moe_align_kernel<0>, dim3(num_experts), dim3(ALIGN_THREADS), 0, s, expert_ids, sorted_token_ids, block_ids,
                       total_post_pad, batch_x_topk, num_experts, block_size, sorted_max, nullptr, nullptr, nullptr, 1);
    FH_CHECK_LAUNCH();
    return 0;
}

int moe_align_block_size_packed_rows(const int32_t* expert_ids, int32_t* sorted_token_ids, int32_t* block_ids,
                                     int32_t* total_post_pad, int batch_x_topk, int num_experts, int block_size,
                                     int sorted_max, hipStream_t s) {
    FH_REQUIRE(num_experts > 0 && num_experts <= MAX_EXPERTS, "moe align: num_experts=%d must be in [1,%d]", num_experts, MAX_EXPERTS);
    FH_REQUIRE(block_size > 0, "moe align: block_size=%d", block_size);
    FH_REQUIRE(sorted_max >= batch_x_topk, "moe align: sorted_max=%d < pairs=%d", sorted_max, batch_x_topk);
    hipLaunchKernelGGL(moe_align_kernel<1>, dim3(num_experts), dim3(ALIGN_THREADS), 0, s, expert_ids, sorted_token_ids, block_ids,
                       total_post_pad, batch_x_topk, num_experts, block_size, sorted_max, nullptr, nullptr, nullptr, 1);
    FH_CHECK_LAUNCH();
    return 0;
}

int moe_build_pairs_by_token(const int32_t* expert_ids, int32_t* pairs_by_token, int32_t* packed_token_idx,
                             int32_t* expert_offsets, int batch_x_topk, int num_experts, int top_k, hipStream_t s) {
    FH_REQUIRE(num_experts > 0 && num_experts <= MAX_EXPERTS, "moe build pairs: num_experts=%d must be in [1,%d]", num_experts, MAX_EXPERTS);
    FH_REQUIRE(top_k > 0 && batch_x_topk >= 0, "moe build pairs: top_k=%d pairs=%d", top_k, batch_x_topk);
    hipLaunchKernelGGL(moe_align_kernel<2>, dim3(num_experts), dim3(ALIGN_THREADS), 0, s, expert_ids, nullptr, nullptr, nullptr,
                       batch_x_topk, num_experts, 1, 0, pairs_by_token, packed_token_idx, expert_offsets, top_k);
    FH_CHECK_LAUNCH();
    return 0;
}

// moe_combine with the trait's own signature (capabilities.rs:684-724; kernels/moe_combine.cu:29): rows of the
// expert-bucketed `packed_down` are found through pairs_by_token (−1 = skipped slot); k ascending, fp32 accumulate.
__global__ void moe_combine_pairs_kernel(const __half* __restrict__ packed_down, const int32_t* __restrict__ pairs_by_token,
                                         const float* __restrict__ weights, __half* __restrict__ out, int top_k, int hidden,
                                         int total_pairs) {
    const long b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (hidden >> 3)) return;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < top_k; k++) {
        const int row = pairs_by_token[b * top_k + k];
        if (row < 0 || row >= total_pairs) continue;
        const float w = weights[b * top_k + k];
        const half8 d = *reinterpret_cast<const half8*>(packed_down + (long)row * hidden + i * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] += w * (float)d[j];
    }
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = (_Float16)acc[j];
    *reinterpret_cast<half8*>(out + b * hidden + i * 8) = o;
}

int moe_combine_pairs_f16(const __half* packed_down, const int32_t* pairs_by_token, const float* pair_weights, __half* out,
                          int batch, int hidden, int top_k, int total_pairs, hipStream_t s) {
    if (batch <= 0) return 0;
    FH_REQUIRE(hidden % 8 == 0 && top_k > 0, "moe combine: hidden=%d must be a multiple of 8, top_k=%d", hidden, top_k);
    hipLaunchKernelGGL(moe_combine_pairs_kernel, dim3(cdiv(hidden / 8, 256), batch), dim3(256), 0, s, packed_down, pairs_by_token,
                       pair_weights, out, top_k, hidden, total_pairs);
    FH_CHECK_LAUNCH();
    return 0;
}

// out[b] (+)= Σ_k w[b,k]·down[b·top_k + k]   (k ascending, fp32 accumulate; moe_forward_cpu order,
// ferrum-models/src/moe/dispatch.rs:2277-2283).  accumulate=1 folds the residual add
// (qwen3_moe_forward_unified_layer.rs:451) into the same pass.
// Expert parallelism: global expert id → the rank's local id (id − e0), −1 for experts owned by other ranks; every
// grouped-GEMM form then simply finds no work for those pairs (ids outside [0, E_local) are ignored by the align / the
// expert-major match), and the combine below skips them.
__global__ void moe_remap_expert_ids_kernel(const int32_t* __restrict__ ids, int32_t* __restrict__ local, int n, int e0, int e_local) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int v = ids[i] - e0;
        local[i] = (v >= 0 && v < e_local) ? v : -1;
    }
}
int moe_remap_expert_ids(const int32_t* ids, int32_t* local, int n, int e0, int e_local, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(moe_remap_expert_ids_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, ids, local, n, e0, e_local);
    FH_CHECK_LAUNCH();
    return 0;
}

__global__ void moe_combine_kernel(const __half* __restrict__ down, const float* __restrict__ weights,
                                   __half* __restrict__ out, int top_k, int hidden, int accumulate,
                                   const int32_t* __restrict__ pair_ids) {
    const long b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (hidden >> 3)) return;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // the first 8 expert rows are requested together (a runtime-bound loop pays one memory round trip per expert)
    float w8[8];
    half8 d8[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int kc = k < top_k ? k : top_k - 1;
        w8[k] = weights[b * top_k + kc];
        d8[k] = *reinterpret_cast<const half8*>(down + (b * top_k + kc) * hidden + i * 8);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k < top_k && (!pair_ids || pair_ids[b * top_k + k] >= 0)) {      // pair_ids: rows of other ranks' experts were never written
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] += w8[k] * (float)d8[k][j];
        }
    }
    for (int k = 8; k < top_k; k++) {
        if (pair_ids && pair_ids[b * top_k + k] < 0) continue;
        float w = weights[b * top_k + k];
        half8 d = *reinterpret_cast<const half8*>(down + (b * top_k + k) * hidden + i * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] += w * (float)d[j];
    }
    half8 o;
    if (accumulate) {
        half8 r = *reinterpret_cast<const half8*>(out + b * hidden + i * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (_Float16)((float)r[j] + acc[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (_Float16)acc[j];
    }
    *reinterpret_cast<half8*>(out + b * hidden + i * 8) = o;
}

int moe_combine_f16(const __half* down, const float* weights, __half* out, int tokens, int top_k, int hidden,
                    int accumulate_into_residual, hipStream_t s) {
    if (tokens <= 0) return 0;
    FH_REQUIRE(hidden % 8 == 0, "moe combine: hidden=%d must be a multiple of 8", hidden);
    hipLaunchKernelGGL(moe_combine_kernel, dim3(cdiv(hidden / 8, 256), tokens), dim3(256), 0, s, down, weights, out,
                       top_k, hidden, accumulate_into_residual, (const int32_t*)nullptr);
    FH_CHECK_LAUNCH();
    return 0;
}

// the rank's PARTIAL MoE output under expert parallelism: Σ over the pairs whose expert is local (pair_ids ≥ 0), k ascending
int moe_combine_local_f16(const __half* down, const float* weights, const int32_t* pair_ids, __half* out, int tokens, int top_k,
                          int hidden, hipStream_t s) {
    if (tokens <= 0) return 0;
    FH_REQUIRE(hidden % 8 == 0 && pair_ids, "moe combine (local): hidden=%d must be a multiple of 8", hidden);
    hipLaunchKernelGGL(moe_combine_kernel, dim3(cdiv(hidden / 8, 256), tokens), dim3(256), 0, s, down, weights, out,
                       top_k, hidden, 0, pair_ids);
    FH_CHECK_LAUNCH();
    return 0;
}

}  // namespace fh
