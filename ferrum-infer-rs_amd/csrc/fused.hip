// Launch-count fusions of the MoE decode layer (gfx950).
//
// At c ≤ 32 every small op of the layer is latency-bound (≈5 µs per dependent launch on MI355X while
// the arithmetic is nanoseconds), so the runner merges the chains the reference issues one op at a time
// (qwen3_moe_forward_unified_layer.rs:380-451):
//   B  fused_add_rms_norm → router GEMM → route_topk_softmax            → ONE launch per token row
//      (router weights in the f16t tile layout of w4_gemm.hip, padded to a multiple of 16 experts)
//   A  moe_combine (weighted_sum) → add_inplace → next layer's rms_norm  → ONE launch per token row
// The arithmetic per element is unchanged: same rounding points as the unfused HIP ops (residual stored
// fp16, norm taken from the rounded residual, router logits fp32 from fp16 operands, top-k by k
// argmax-mask passes with lowest-index ties).
#include "common.h"
#include "kernels.h"

namespace fh {

__device__ __forceinline__ float block_sum_256(float v, float* smem) {
    v = wave_reduce_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    float t = smem[0] + smem[1] + smem[2] + smem[3];
    __syncthreads();
    return t;
}

// ── B: residual += x; norm_out = rms(residual)·w; logits = norm_out·routerᵀ; top-k softmax ────────
// One 1024-thread workgroup per token.  H ≤ 8192, E ≤ 512, top_k ≤ 64.
// The router GEMV runs on the matrix cores: the token row is row 0 of a 16-row A operand (other rows
// zero), router rows are the B operand (16 experts per tile, fetched as 1-KiB fragment loads), and the
// 16 waves split (expert tile × K slice) so ≥ 8 independent loads per lane are in flight — the naive
// lane-per-K GEMV was latency-bound at 70 µs per launch.
__device__ __forceinline__ float block_sum_1024(float v, float* smem) {
    v = wave_reduce_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; w++) t += smem[w];
    __syncthreads();
    return t;
}



template <bool SLABS>
__global__ __launch_bounds__(1024) void add_rmsnorm_route_kernel(
    __half* __restrict__ residual, const __half* __restrict__ x, const float* __restrict__ x_slabs, int S,
    long slab_stride, int ld_slab, const __half* __restrict__ w, float eps,
    __half* __restrict__ norm_out, const __half* __restrict__ router_w, int num_experts, int top_k,
    int norm_topk_prob, int32_t* __restrict__ ids, float* __restrict__ weights, float* __restrict__ logits_out,
    int H, const int32_t* __restrict__ out_perm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __half* xs = reinterpret_cast<__half*>(smem_raw);                      // normalised row [H]
    const int tiles = (num_experts + 15) >> 4;
    const int ksplit = tiles >= 16 ? 1 : 16 / (tiles > 0 ? tiles : 1);
    float* part = reinterpret_cast<float*>(smem_raw + (size_t)H * 2);      // partial logits [ksplit][tiles·16]
    __shared__ float red[16];
    const long row = blockIdx.x;
    const int nvec = H >> 3;
    const int i = threadIdx.x;
    half8 v;
    float ss = 0.f;
    if (i < nvec) {
        float o[8];
        if (SLABS) {     // o_proj arrives as S fp32 split-K slabs: reduce in slab order, round like the fp16 op output
            reduce_slabs8(x_slabs + row * ld_slab + i * 8, slab_stride, S, o);
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (float)(_Float16)o[j];
        } else {
            half8 xv = *reinterpret_cast<const half8*>(x + row * H + i * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (float)xv[j];
        }
        half8 rv = *reinterpret_cast<const half8*>(residual + row * H + i * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) rv[j] = (_Float16)((float)rv[j] + o[j]);
        *reinterpret_cast<half8*>(residual + row * H + i * 8) = rv;
        v = rv;
#pragma unroll
        for (int j = 0; j < 8; j++) ss += (float)rv[j] * (float)rv[j];
    }
    const float total = block_sum_1024(ss, red);
    const float inv = 1.0f / sqrtf(total / (float)H + eps);
    if (i < nvec) {
        half8 wv = *reinterpret_cast<const half8*>(w + i * 8);
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (_Float16)((float)v[j] * inv * (float)wv[j]);
        if (!out_perm) *reinterpret_cast<half8*>(norm_out + row * H + i * 8) = o;
        *reinterpret_cast<half8*>(xs + i * 8) = o;
    }
    __syncthreads();
    // act-order consumer: the row leaves in the packed-row order of its weights (the router below reads the natural row)
    if (out_perm && i < nvec)
        *reinterpret_cast<half8*>(norm_out + row * H + i * 8) = lds_gather8(reinterpret_cast<const _Float16*>(xs), out_perm, i * 8);
    if (num_experts <= 0) return;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int a = lane >> 4, b = lane & 15;
    const int ksteps = H >> 5;
    for (int u = wave; u < tiles * ksplit; u += 16) {
        const int tile = u / ksplit, ks = u % ksplit;
        const int s0 = ksteps * ks / ksplit, s1 = ksteps * (ks + 1) / ksplit;
        // router weights are in f16t tiles: [tile][k-step][64 lanes][8] → contiguous 1-KiB wave loads
        const __half* wrow = router_w + ((long)tile * ksteps * 64 + lane) * 8;
        float4v acc = {0.f, 0.f, 0.f, 0.f};
        constexpr int U = 16;
        int s = s0;
        for (; s + U <= s1; s += U) {
            half8 bw[U];
#pragma unroll
            for (int q = 0; q < U; q++) bw[q] = *reinterpret_cast<const half8*>(wrow + (long)(s + q) * 512);
#pragma unroll
            for (int q = 0; q < U; q++) {
                half8 av = {0, 0, 0, 0, 0, 0, 0, 0};
                if (b == 0) av = *reinterpret_cast<const half8*>(xs + (s + q) * 32 + 8 * a);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bw[q], acc, 0, 0, 0);
            }
        }
        for (; s < s1; s++) {
            half8 bwv = *reinterpret_cast<const half8*>(wrow + (long)s * 512);
            half8 av = {0, 0, 0, 0, 0, 0, 0, 0};
            if (b == 0) av = *reinterpret_cast<const half8*>(xs + s * 32 + 8 * a);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bwv, acc, 0, 0, 0);
        }
        // D[row 4a+r][col b]: the token is row 0 → lanes with a == 0, register 0
        if (a == 0) part[ks * tiles * 16 + tile * 16 + b] = acc[0];
    }
    __syncthreads();
    // softmax + top-k (ferrum-models/src/moe/router.rs:113-195 semantics).  Selection is by RANK —
    // rank(e) = #{j : p_j > p_e or (p_j == p_e and j < e)} — computed with broadcast LDS reads, which
    // gives the same ids as k argmax-mask passes with lowest-index ties but without 8 × 12 dependent
    // cross-lane shuffles on the critical path.
    float* prob = part + ksplit * tiles * 16;          // [E]
    float* sel_w = prob + tiles * 16;                  // [top_k]
    int* sel_id = reinterpret_cast<int*>(sel_w + 64);  // [top_k]
    const int t = threadIdx.x;
    float l = -INFINITY;
    if (t < num_experts) {
        l = 0.f;
        for (int ks = 0; ks < ksplit; ks++) l += part[ks * tiles * 16 + t];
        if (logits_out) logits_out[row * num_experts + t] = l;
    }
    // max and sum over ≤ 512 experts: the first 8 waves hold them, reduce through LDS
    float mx = wave_reduce_max(l);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
    __syncthreads();
    float ex = t < num_experts ? expf(l - mx) : 0.f;
    float sm = wave_reduce_sum(ex);
    if (lane == 0) red[wave] = sm;
    __syncthreads();
    // sequential-ish order (wave partials 0..7) — fp32, within the 1e-6 weight tolerance of router.rs:219-222
    sm = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
    const float p_mine = ex * (1.0f / sm);
    if (t < tiles * 16) prob[t] = t < num_experts ? p_mine : -INFINITY;
    __syncthreads();
    if (t < num_experts) {
        int rank = 0;
        for (int j = 0; j < num_experts; j++) {
            float pj = prob[j];
            rank += (pj > p_mine || (pj == p_mine && j < t)) ? 1 : 0;
        }
        if (rank < top_k) { sel_w[rank] = p_mine; sel_id[rank] = t; }
    }
    __syncthreads();
    if (t < top_k) {
        float sel_sum = 0.f;
        for (int k = 0; k < top_k; k++) sel_sum += sel_w[k];          // descending order, as route_into adds them
        float ww = sel_w[t];
        if (norm_topk_prob) ww = sel_sum > 0.f ? ww * (1.0f / sel_sum) : 1.0f / (float)top_k;
        ids[row * top_k + t] = sel_id[t];
        weights[row * top_k + t] = ww;
    }
}

int fused_add_rms_norm_route_slabs_f16(__half* residual, const __half* x, const float* x_slabs, int S, long slab_stride,
                                       int ld_slab, const __half* w, float eps, __half* norm_out, const __half* router_w,
                                       int num_experts, int top_k, int norm_topk_prob, int32_t* expert_ids,
                                       float* expert_weights, float* logits_out, int tokens, int H, hipStream_t s,
                                       const int32_t* out_perm);

int fused_add_rms_norm_route_f16(__half* residual, const __half* x, const __half* w, float eps, __half* norm_out,
                                 const __half* router_w, int num_experts, int top_k, int norm_topk_prob,
                                 int32_t* expert_ids, float* expert_weights, float* logits_out, int tokens, int H,
                                 hipStream_t s) {
    return fused_add_rms_norm_route_slabs_f16(residual, x, nullptr, 0, 0, 0, w, eps, norm_out, router_w, num_experts, top_k,
                                              norm_topk_prob, expert_ids, expert_weights, logits_out, tokens, H, s, nullptr);
}

int fused_add_rms_norm_route_slabs_f16(__half* residual, const __half* x, const float* x_slabs, int S, long slab_stride,
                                       int ld_slab, const __half* w, float eps, __half* norm_out, const __half* router_w,
                                       int num_experts, int top_k, int norm_topk_prob, int32_t* expert_ids,
                                       float* expert_weights, float* logits_out, int tokens, int H, hipStream_t s,
                                       const int32_t* out_perm) {
    if (tokens <= 0) return 0;
    FH_REQUIRE(H % 32 == 0 && H <= 8192, "fused_add_rms_norm_route: hidden=%d must be a multiple of 32, <= 8192", H);
    FH_REQUIRE(num_experts <= 512 && top_k <= 64 && (num_experts == 0 || (top_k > 0 && top_k <= num_experts)),
               "fused_add_rms_norm_route: experts=%d top_k=%d", num_experts, top_k);
    const int tiles = (num_experts + 15) / 16;
    const int ksplit = tiles >= 16 ? 1 : 16 / std::max(tiles, 1);
    const size_t lds = (size_t)H * 2 + ((size_t)std::max(tiles, 1) * 16 * (ksplit + 1) + 128) * 4;
    if (x_slabs)
        hipLaunchKernelGGL(add_rmsnorm_route_kernel<true>, dim3(tokens), dim3(1024), lds, s, residual, x, x_slabs, S, slab_stride,
                           ld_slab, w, eps, norm_out, router_w, num_experts, top_k, norm_topk_prob, expert_ids, expert_weights,
                           logits_out, H, out_perm);
    else
        hipLaunchKernelGGL(add_rmsnorm_route_kernel<false>, dim3(tokens), dim3(1024), lds, s, residual, x, x_slabs, S, slab_stride,
                           ld_slab, w, eps, norm_out, router_w, num_experts, top_k, norm_topk_prob, expert_ids, expert_weights,
                           logits_out, H, out_perm);
    FH_CHECK_LAUNCH();
    return 0;
}

// ── B, split over expert quarters: grid (tokens, Q) ───────────────────────────────────────────────
// One token's router GEMV is bound by the ≈70 GB/s a single CU can pull from L2 (512 KB → 7 µs), so the
// experts are split over Q workgroups per token.  Every part repeats the (tiny) add + norm of its token,
// computes the logits of its E/Q experts on the matrix cores and emits its best min(8, E/Q) candidates
// (sorted by logit, ties → lower id) plus its softmax statistics (max, Σexp).  The consumer — the prologue
// of the gate_up grouped GEMM — merges the Q sorted lists per token (merge_route_candidates in w4_gemm.hip).
// The o-projection may arrive as S fp32 split-K slabs, reduced here in slab order (deterministic).
template <bool SLABS>
__global__ __launch_bounds__(512) void add_rmsnorm_route_part_kernel(
    const __half* __restrict__ residual, __half* __restrict__ residual_out, const __half* __restrict__ x,
    const float* __restrict__ x_slabs, int S,
    long slab_stride, int ld_slab, const __half* __restrict__ w, float eps, __half* __restrict__ norm_out,
    const __half* __restrict__ router_w, int num_experts, int top_k, RouteCand* cand, float* stats,
    float* __restrict__ logits_out, int H, unsigned* arrive, int norm_topk_prob, int32_t* __restrict__ ids,
    float* __restrict__ weights) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __half* xs = reinterpret_cast<__half*>(smem_raw);
    float* part = reinterpret_cast<float*>(smem_raw + (size_t)H * 2);
    __shared__ float red[8];
    const long row = blockIdx.x;
    const int q = blockIdx.y, Q = gridDim.y;
    const int nvec = H >> 3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int CH = 2;                        // H ≤ 8192 with 512 threads
    half8 v[CH];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int i = threadIdx.x + c * 512;
        if (i < nvec) {
            float o[8];
            if (SLABS) {
                reduce_slabs8(x_slabs + row * ld_slab + i * 8, slab_stride, S, o);
#pragma unroll
                for (int j = 0; j < 8; j++) o[j] = (float)(_Float16)o[j];      // the fp16 o_proj output the unfused op stores
            } else {
                half8 xv = *reinterpret_cast<const half8*>(x + row * H + i * 8);
#pragma unroll
                for (int j = 0; j < 8; j++) o[j] = (float)xv[j];
            }
            half8 rv = *reinterpret_cast<const half8*>(residual + row * H + i * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) rv[j] = (_Float16)((float)rv[j] + o[j]);
            v[c] = rv;
#pragma unroll
            for (int j = 0; j < 8; j++) ss += (float)rv[j] * (float)rv[j];
        }
    }
    // requested now, used after the two barriers below: the norm weights and the router weights of this wave's first unit
    // (16 k-steps × 1 KiB per wave) travel while the row statistics are reduced and the row is normalised into LDS
    half8 wv_pre[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int i = threadIdx.x + c * 512;
        wv_pre[c] = *reinterpret_cast<const half8*>(w + (i < nvec ? i : 0) * 8);
    }
    constexpr int U = 16;
    const int tiles = (num_experts + 15) >> 4;
    const int tiles_q = num_experts > 0 ? tiles / Q : 0;                      // launcher guarantees divisibility
    const int ksplit = tiles_q >= 8 || tiles_q == 0 ? 1 : 8 / tiles_q;
    const int ksteps = H >> 5;
    half8 bw_pre[U];
    bool pre = false;
    if (wave < tiles_q * ksplit) {
        const int tl = wave / ksplit, ks = wave % ksplit;
        const int s0 = ksteps * ks / ksplit, s1 = ksteps * (ks + 1) / ksplit;
        if (s0 + U <= s1) {
            const __half* wrow = router_w + ((long)(q * tiles_q + tl) * ksteps * 64 + lane) * 8;
#pragma unroll
            for (int kk = 0; kk < U; kk++) bw_pre[kk] = *reinterpret_cast<const half8*>(wrow + (long)(s0 + kk) * 512);
            pre = true;
        }
    }
    ss = wave_reduce_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float total = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
    const float inv = 1.0f / sqrtf(total / (float)H + eps);
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int i = threadIdx.x + c * 512;
        if (i < nvec) {
            const half8 wv = wv_pre[c];
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (_Float16)((float)v[c][j] * inv * (float)wv[j]);
            *reinterpret_cast<half8*>(xs + i * 8) = o;
            if (q == 0) {     // part 0 owns the stores (all parts computed identical values)
                *reinterpret_cast<half8*>(norm_out + row * H + i * 8) = o;
            }
        }
    }
    // Parts of one token run concurrently and all read `residual`, so the updated residual goes to a
    // SEPARATE buffer (the runner ping-pongs two residual buffers between this kernel and kernel A).
    if (q == 0) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int i = threadIdx.x + c * 512;
            if (i < nvec) *reinterpret_cast<half8*>(residual_out + row * H + i * 8) = v[c];
        }
    }
    __syncthreads();
    if (num_experts <= 0) return;
    const int a = lane >> 4, b = lane & 15;
    for (int u = wave; u < tiles_q * ksplit; u += 8) {
        const int tl = u / ksplit, ks = u % ksplit;
        const int tile = q * tiles_q + tl;
        const int s0 = ksteps * ks / ksplit, s1 = ksteps * (ks + 1) / ksplit;
        const __half* wrow = router_w + ((long)tile * ksteps * 64 + lane) * 8;
        float4v acc = {0.f, 0.f, 0.f, 0.f};
        int s = s0;
        if (u == wave && pre) {                      // the chunk requested at the top of the kernel
#pragma unroll
            for (int kk = 0; kk < U; kk++) {
                half8 av = {0, 0, 0, 0, 0, 0, 0, 0};
                if (b == 0) av = *reinterpret_cast<const half8*>(xs + (s + kk) * 32 + 8 * a);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bw_pre[kk], acc, 0, 0, 0);
            }
            s += U;
        }
        for (; s + U <= s1; s += U) {
            half8 bw[U];
#pragma unroll
            for (int k = 0; k < U; k++) bw[k] = *reinterpret_cast<const half8*>(wrow + (long)(s + k) * 512);
#pragma unroll
            for (int k = 0; k < U; k++) {
                half8 av = {0, 0, 0, 0, 0, 0, 0, 0};
                if (b == 0) av = *reinterpret_cast<const half8*>(xs + (s + k) * 32 + 8 * a);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bw[k], acc, 0, 0, 0);
            }
        }
        for (; s < s1; s++) {
            half8 bwv = *reinterpret_cast<const half8*>(wrow + (long)s * 512);
            half8 av = {0, 0, 0, 0, 0, 0, 0, 0};
            if (b == 0) av = *reinterpret_cast<const half8*>(xs + s * 32 + 8 * a);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bwv, acc, 0, 0, 0);
        }
        if (a == 0) part[ks * tiles_q * 16 + tl * 16 + b] = acc[0];
    }
    __syncthreads();
    const int EQ = tiles_q * 16;                         // experts of this part (≤ 128)
    float* lgs = part + ksplit * EQ;                     // [EQ] summed logits
    __shared__ unsigned long long cand_s[8];             // this part's sorted candidates {logit bits, id << 32}
    const int t = threadIdx.x;
    const int e_glob = q * EQ + t;
    float l = -INFINITY;
    if (t < EQ && e_glob < num_experts) {
        l = 0.f;
        for (int ks = 0; ks < ksplit; ks++) l += part[ks * EQ + t];
        if (logits_out) logits_out[row * num_experts + e_glob] = l;
    }
    if (t < EQ) lgs[t] = l;
    // part statistics (EQ ≤ 128 → waves 0,1)
    float mx = wave_reduce_max(l);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(red[0], red[1]);
    __syncthreads();
    float ex = (t < EQ && e_glob < num_experts) ? expf(l - mx) : 0.f;
    float sm = wave_reduce_sum(ex);
    if (lane == 0) red[wave] = sm;
    const int keep = top_k < EQ ? top_k : EQ;
    if (t < EQ) {
        int rank = 0;
        for (int j = 0; j < EQ; j++) {
            float lj = lgs[j];
            rank += (lj > l || (lj == l && j < t)) ? 1 : 0;
        }
        if (rank < keep) cand_s[rank] = ((unsigned long long)(unsigned)e_glob << 32) | __float_as_uint(l);
    }
    if (t >= keep && t < 8) cand_s[t] = (0x7fffffffull << 32) | __float_as_uint(-INFINITY);   // unused slots of a short list
    __syncthreads();
    if (wave != 0) return;
    // Publish (wave 0 only): write-through 8-byte stores, drained before the arrival count — the hand-off form
    // of cdna_hip_programming.md Guideline 16 R1.  Consumers: the gate_up prologue (next launch) or, with
    // `arrive`, the last-arriving part of this token below.
    unsigned long long* cand_g = reinterpret_cast<unsigned long long*>(cand) + (row * Q + q) * 8;
    unsigned long long* stats_g = reinterpret_cast<unsigned long long*>(stats) + (row * Q + q);
    if (lane < 8) __hip_atomic_store(cand_g + lane, cand_s[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 8)
        __hip_atomic_store(stats_g, ((unsigned long long)__float_as_uint(red[0] + red[1]) << 32) | __float_as_uint(mx),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (arrive == nullptr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(arrive + row, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != (unsigned)(Q - 1)) return;
    // Last part of this token to arrive: merge the Q sorted lists (≤ 64 candidates, one per lane).  Every
    // handed-off byte was stored write-through and drained before its part's ticket; the loads below are sc1
    // (L1-bypassing) behind an agent acquire (dropping the acquire measured no faster, so it stays).
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(arrive + row, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
    const int ncand = Q * 8;
    unsigned long long c = (0x7fffffffull << 32) | __float_as_uint(-INFINITY);
    if (lane < ncand)
        c = __hip_atomic_load(reinterpret_cast<unsigned long long*>(cand) + row * Q * 8 + lane, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_AGENT);
    float pmx = -INFINITY, psum = 0.f;
    if (lane < Q) {
        unsigned long long st = __hip_atomic_load(reinterpret_cast<unsigned long long*>(stats) + row * Q + lane,
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pmx = __uint_as_float((unsigned)st);
        psum = __uint_as_float((unsigned)(st >> 32));
    }
    const float gmax = wave_reduce_max(pmx);
    const float gsum = wave_reduce_sum(lane < Q ? psum * expf(pmx - gmax) : 0.f);
    const float my_l = __uint_as_float((unsigned)c);
    const int my_id = (int)(c >> 32);
    int rank = 0;
    for (int j = 0; j < ncand; j++) {
        const float lj = __uint_as_float(__builtin_amdgcn_readlane((int)(unsigned)c, j));
        const int ij = __builtin_amdgcn_readlane((int)(c >> 32), j);
        rank += (lj > my_l || (lj == my_l && ij < my_id)) ? 1 : 0;
    }
    const float p = expf(my_l - gmax) * (1.0f / gsum);
    float sel_sum = 0.f;                                  // Σ of the winners in descending order, as route_into adds them
    for (int k = 0; k < top_k; k++) sel_sum += wave_reduce_sum(rank == k ? p : 0.f);
    if (rank < top_k) {
        float ww = p;
        if (norm_topk_prob) ww = sel_sum > 0.f ? ww * (1.0f / sel_sum) : 1.0f / (float)top_k;
        ids[row * top_k + rank] = my_id;
        weights[row * top_k + rank] = ww;
    }
}

int fused_add_rms_norm_route_parts_f16(const __half* residual_in, __half* residual_out, const __half* x,
                                       const float* x_slabs, int S, long slab_stride, int ld_slab, const __half* w,
                                       float eps, __half* norm_out, const __half* router_w, int num_experts, int top_k,
                                       int Q, RouteCand* cand, float* stats, float* logits_out, int tokens, int H,
                                       hipStream_t s) {
    return fused_add_rms_norm_route_split_f16(residual_in, residual_out, x, x_slabs, S, slab_stride, ld_slab, w, eps, norm_out,
                                              router_w, num_experts, top_k, Q, cand, stats, nullptr, 0, nullptr, nullptr,
                                              logits_out, tokens, H, s);
}

// With `arrive` (one zeroed counter per token; left zeroed), the last part of each token to arrive merges the
// lists inside the launch and writes expert_ids / expert_weights [tokens, top_k] — same result as the single-
// workgroup kernel.  Without it the candidate lists are the output (merged by the gate_up prologue).
int fused_add_rms_norm_route_split_f16(const __half* residual_in, __half* residual_out, const __half* x,
                                       const float* x_slabs, int S, long slab_stride, int ld_slab, const __half* w,
                                       float eps, __half* norm_out, const __half* router_w, int num_experts, int top_k,
                                       int Q, RouteCand* cand, float* stats, unsigned* arrive, int norm_topk_prob,
                                       int32_t* expert_ids, float* expert_weights, float* logits_out, int tokens, int H,
                                       hipStream_t s) {
    if (tokens <= 0) return 0;
    FH_REQUIRE(arrive == nullptr || (expert_ids && expert_weights && Q <= 8),
               "route_split: in-launch merge needs ids/weights outputs and Q <= 8 (got %d)", Q);
    FH_REQUIRE(H % 32 == 0 && H <= 8192, "route_parts: hidden=%d must be a multiple of 32, <= 8192", H);
    FH_REQUIRE(residual_in != residual_out || (Q == 1), "route_parts: in-place residual needs Q == 1");
    const int tiles = std::max(1, (num_experts + 15) / 16);
    if (num_experts > 0) {
        FH_REQUIRE(Q >= 1 && tiles % Q == 0 && tiles / Q <= 8, "route_parts: %d expert tiles not divisible into %d parts of <= 8", tiles, Q);
        FH_REQUIRE(top_k >= 1 && top_k <= 8, "route_parts: top_k=%d must be in [1,8]", top_k);
    } else {
        Q = 1;
    }
    const int tiles_q = std::max(1, tiles / Q), ksplit = tiles_q >= 8 ? 1 : 8 / tiles_q;
    const size_t lds = (size_t)H * 2 + (size_t)tiles_q * 16 * (ksplit + 1) * 4;
    dim3 grid(tokens, Q);
    if (x_slabs)
        hipLaunchKernelGGL(add_rmsnorm_route_part_kernel<true>, grid, dim3(512), lds, s, residual_in, residual_out, x,
                           x_slabs, S, slab_stride, ld_slab, w, eps, norm_out, router_w, num_experts, top_k, cand, stats,
                           logits_out, H, arrive, norm_topk_prob, expert_ids, expert_weights);
    else
        hipLaunchKernelGGL(add_rmsnorm_route_part_kernel<false>, grid, dim3(512), lds, s, residual_in, residual_out, x,
                           x_slabs, S, slab_stride, ld_slab, w, eps, norm_out, router_w, num_experts, top_k, cand, stats,
                           logits_out, H, arrive, norm_topk_prob, expert_ids, expert_weights);
    FH_CHECK_LAUNCH();
    return 0;
}

// ── A: residual += Σ_k w[b,k]·down[b·K+k];  norm_out = rms(residual)·next_w (optional) ───────────
template <int CHUNKS>
__global__ __launch_bounds__(256) void moe_combine_add_rmsnorm_kernel(
    const __half* __restrict__ down, const float* __restrict__ weights, const __half* __restrict__ residual,
    __half* __restrict__ residual_out, const __half* __restrict__ next_w, float eps, __half* __restrict__ norm_out,
    int top_k, int H, const int32_t* __restrict__ out_perm) {
    __shared__ float red[4];
    extern __shared__ _Float16 stage[];            // H halves when out_perm (act-order q|k|v of the next layer: the row leaves permuted)
    const long row = blockIdx.x;
    const int nvec = H >> 3;
    half8 v[CHUNKS];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
        int i = threadIdx.x + c * 256;
        if (i < nvec) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // the first 8 expert rows, the residual row (and below the norm weights) are requested together: a runtime-bound
            // loop keeps one load in flight and pays a memory round trip per expert
            float wk8[8];
            half8 d8[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int kc = k < top_k ? k : top_k - 1;
                wk8[k] = weights[row * top_k + kc];
                d8[k] = *reinterpret_cast<const half8*>(down + (row * top_k + kc) * H + i * 8);
            }
            half8 rv = *reinterpret_cast<const half8*>(residual + row * H + i * 8);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (k < top_k) {
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[j] += wk8[k] * (float)d8[k][j];
                }
            }
            for (int k = 8; k < top_k; k++) {
                float wk = weights[row * top_k + k];
                half8 d = *reinterpret_cast<const half8*>(down + (row * top_k + k) * H + i * 8);
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] += wk * (float)d[j];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) rv[j] = (_Float16)((float)rv[j] + acc[j]);
            *reinterpret_cast<half8*>(residual_out + row * H + i * 8) = rv;
            v[c] = rv;
#pragma unroll
            for (int j = 0; j < 8; j++) ss += (float)rv[j] * (float)rv[j];
        }
    }
    if (!next_w) return;
    const float total = block_sum_256(ss, red);
    const float inv = 1.0f / sqrtf(total / (float)H + eps);
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
        int i = threadIdx.x + c * 256;
        if (i < nvec) {
            half8 wv = *reinterpret_cast<const half8*>(next_w + i * 8);
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (_Float16)((float)v[c][j] * inv * (float)wv[j]);
            if (out_perm) *reinterpret_cast<half8*>(stage + i * 8) = o;
            else *reinterpret_cast<half8*>(norm_out + row * H + i * 8) = o;
        }
    }
    if (out_perm) {
        __syncthreads();
        for (int i = threadIdx.x; i < nvec; i += 256) *reinterpret_cast<half8*>(norm_out + row * H + i * 8) = lds_gather8(stage, out_perm, i * 8);
    }
}

int moe_combine_add_rms_norm_f16(const __half* down, const float* weights, const __half* residual,
                                 __half* residual_out, const __half* next_w, float eps, __half* norm_out, int tokens,
                                 int top_k, int H, hipStream_t s, const int32_t* out_perm) {
    if (tokens <= 0) return 0;
    FH_REQUIRE(H % 8 == 0 && H <= 8 * 256 * 4, "moe_combine_add_rms_norm: hidden=%d must be a multiple of 8, <= 8192", H);
    const int chunks = cdiv(H / 8, 256);
    const size_t lds = (out_perm && next_w) ? (size_t)H * 2 : 0;
#define FH_A(C) hipLaunchKernelGGL((moe_combine_add_rmsnorm_kernel<C>), dim3(tokens), dim3(256), lds, s, down, weights, \
                                   residual, residual_out, next_w, eps, norm_out, top_k, H, next_w ? out_perm : nullptr)
    if (chunks <= 1) FH_A(1); else if (chunks <= 2) FH_A(2); else FH_A(4);
#undef FH_A
    FH_CHECK_LAUNCH();
    return 0;
}

}  // namespace fh
