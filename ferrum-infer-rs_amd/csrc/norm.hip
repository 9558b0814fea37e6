// RMSNorm family, embedding gather and the elementwise ops of the decode layer (gfx950).
//
// Reference contracts (ferrum-kernels/src/backend/traits.rs; CPU forms in backend/cpu.rs):
//   rms_norm            traits.rs:202  / cpu.rs:495-513   y = x·rsqrt(mean(x²)+eps)·w
//   fused_add_rms_norm  traits.rs:212  / cpu.rs:515-538   res += x; y = rms(res)·w
//   embedding_lookup    traits.rs:809  / cpu.rs:1632
//   fused_silu_mul_split / fused_gelu_tanh_mul_split  traits.rs:863,875 / cpu.rs:1666-1698
//   add_inplace, scale_inplace, add_bias              traits.rs:1308,889,1359
//   gather_columns (act-order)                        kernels/gather_columns.cu:15
// All are HBM/latency-bound: 16-byte vector accesses (8 × fp16 per lane), fp32 math, one pass
// over the row with the row held in registers between the reduction and the scale.
#include "common.h"
#include "kernels.h"
#include "knobs.h"

namespace fh {

__device__ __forceinline__ float block_reduce_sum_256(float v, float* smem) {
    v = wave_reduce_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    float t = smem[0] + smem[1] + smem[2] + smem[3];
    __syncthreads();
    return t;
}

// every thread has written its vectors of the row to `stage`; 256 threads, vector i of thread t = t + c·256
__device__ __forceinline__ void store_row_gathered_256(__half* out_row, const _Float16* stage, const int32_t* __restrict__ perm, int nvec) {
    __syncthreads();
    for (int i = threadIdx.x; i < nvec; i += 256) *reinterpret_cast<half8*>(out_row + i * 8) = lds_gather8(stage, perm, i * 8);
}

// One 256-thread workgroup per row; each thread keeps up to CHUNKS 16-byte chunks in registers.
template <bool FUSED_ADD, int CHUNKS>
__global__ __launch_bounds__(256) void rms_norm_kernel(const __half* __restrict__ x, __half* __restrict__ residual,
                                                       const __half* __restrict__ w, float eps,
                                                       __half* __restrict__ out, int dim,
                                                       const int32_t* __restrict__ out_perm) {
    __shared__ float smem[4];
    extern __shared__ _Float16 stage[];            // dim halves when out_perm
    const long row = blockIdx.x;
    const int nvec = dim >> 3;
    half8 v[CHUNKS];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
        int i = threadIdx.x + c * 256;
        if (i < nvec) {
            half8 xv = *reinterpret_cast<const half8*>(x + row * dim + i * 8);
            if (FUSED_ADD) {
                half8 rv = *reinterpret_cast<const half8*>(residual + row * dim + i * 8);
#pragma unroll
                for (int j = 0; j < 8; j++) rv[j] = (_Float16)((float)rv[j] + (float)xv[j]);
                // the updated residual is stored as fp16 and the variance is taken from the
                // rounded values (same rounding point as the reference's fp16 lane,
                // kernels/fused_add_rms_norm.cu:88-101)
                *reinterpret_cast<half8*>(residual + row * dim + i * 8) = rv;
                xv = rv;
            }
            v[c] = xv;
#pragma unroll
            for (int j = 0; j < 8; j++) ss += (float)xv[j] * (float)xv[j];
        }
    }
    float total = block_reduce_sum_256(ss, smem);
    float inv = 1.0f / sqrtf(total / (float)dim + eps);
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
        int i = threadIdx.x + c * 256;
        if (i < nvec) {
            half8 wv = *reinterpret_cast<const half8*>(w + i * 8);
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (_Float16)((float)v[c][j] * inv * (float)wv[j]);
            if (out_perm) *reinterpret_cast<half8*>(stage + i * 8) = o;
            else *reinterpret_cast<half8*>(out + row * dim + i * 8) = o;
        }
    }
    if (out_perm) store_row_gathered_256(out + row * dim, stage, out_perm, nvec);
}

template <bool FUSED>
static int launch_rms(const __half* x, __half* residual, const __half* w, float eps, __half* out, int tokens,
                      int dim, const int32_t* out_perm, hipStream_t s) {
    if (tokens <= 0) return 0;
    FH_REQUIRE(dim % 8 == 0 && dim <= 8 * 256 * 8, "rms_norm: dim=%d must be a multiple of 8 and <= 16384", dim);
    int chunks = cdiv(dim / 8, 256);
    dim3 grid(tokens), block(256);
    const size_t lds = out_perm ? (size_t)dim * 2 : 0;
    if (chunks <= 1) hipLaunchKernelGGL((rms_norm_kernel<FUSED, 1>), grid, block, lds, s, x, residual, w, eps, out, dim, out_perm);
    else if (chunks <= 2) hipLaunchKernelGGL((rms_norm_kernel<FUSED, 2>), grid, block, lds, s, x, residual, w, eps, out, dim, out_perm);
    else if (chunks <= 4) hipLaunchKernelGGL((rms_norm_kernel<FUSED, 4>), grid, block, lds, s, x, residual, w, eps, out, dim, out_perm);
    else hipLaunchKernelGGL((rms_norm_kernel<FUSED, 8>), grid, block, lds, s, x, residual, w, eps, out, dim, out_perm);
    FH_CHECK_LAUNCH();
    return 0;
}

int rms_norm_f16(const __half* x, const __half* w, float eps, __half* out, int tokens, int dim, hipStream_t s) {
    return launch_rms<false>(x, nullptr, w, eps, out, tokens, dim, nullptr, s);
}
int fused_add_rms_norm_f16(__half* residual, const __half* x, const __half* w, float eps, __half* out, int tokens,
                           int dim, hipStream_t s, const int32_t* out_perm) {
    return launch_rms<true>(x, residual, w, eps, out, tokens, dim, out_perm, s);
}

// Head / tail of a forward in ONE launch each (a dependent launch costs ≈ 4 µs even when trivial; a decode step had nine of
// them outside the layers): gather a row (embedding row by token id, or residual row by sampled index) → optional scale
// (fp16 rounding, like scale_inplace) → optional copies of the gathered row (fp16 residual, fp32 residual stream) →
// rms_norm·w → out.  Same per-element arithmetic as embedding_lookup → scale_inplace → rms_norm.  Block 0 also zeroes
// `zero_words` (the split router's arrival counters, re-armed per forward).
template <typename IdxT, int CHUNKS>
__global__ __launch_bounds__(256) void gather_rms_norm_kernel(const __half* __restrict__ table, const IdxT* __restrict__ ids, float scale,
                                                              __half* __restrict__ copy_f16, float* __restrict__ copy_f32,
                                                              const __half* __restrict__ w, float eps, __half* __restrict__ out, int dim,
                                                              unsigned* __restrict__ zero_words, int n_zero,
                                                              const int32_t* __restrict__ out_perm) {
    __shared__ float smem[4];
    extern __shared__ _Float16 stage[];            // dim halves when out_perm
    const long row = blockIdx.x;
    const long src = (long)ids[row];
    if (blockIdx.x == 0 && zero_words)
        for (int i = threadIdx.x; i < n_zero; i += 256) zero_words[i] = 0u;
    const int nvec = dim >> 3;
    half8 v[CHUNKS];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
        const int i = threadIdx.x + c * 256;
        if (i < nvec) {
            half8 xv = *reinterpret_cast<const half8*>(table + src * dim + i * 8);
            if (scale != 0.0f) {
#pragma unroll
                for (int j = 0; j < 8; j++) xv[j] = (_Float16)((float)xv[j] * scale);
            }
            if (copy_f16) *reinterpret_cast<half8*>(copy_f16 + row * dim + i * 8) = xv;
            if (copy_f32) {
                float4v lo, hi;
#pragma unroll
                for (int j = 0; j < 4; j++) { lo[j] = (float)xv[j]; hi[j] = (float)xv[4 + j]; }
                float4v* d = reinterpret_cast<float4v*>(copy_f32 + row * dim + i * 8);
                d[0] = lo;
                d[1] = hi;
            }
            v[c] = xv;
#pragma unroll
            for (int j = 0; j < 8; j++) ss += (float)xv[j] * (float)xv[j];
        }
    }
    const float total = block_reduce_sum_256(ss, smem);
    const float inv = 1.0f / sqrtf(total / (float)dim + eps);
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
        const int i = threadIdx.x + c * 256;
        if (i < nvec) {
            const half8 wv = *reinterpret_cast<const half8*>(w + i * 8);
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (_Float16)((float)v[c][j] * inv * (float)wv[j]);
            if (out_perm) *reinterpret_cast<half8*>(stage + i * 8) = o;
            else *reinterpret_cast<half8*>(out + row * dim + i * 8) = o;
        }
    }
    if (out_perm) store_row_gathered_256(out + row * dim, stage, out_perm, nvec);
}

template <typename IdxT>
static int launch_gather_rms(const __half* table, const IdxT* ids, float scale, __half* copy_f16, float* copy_f32, const __half* w, float eps,
                             __half* out, int rows, int dim, unsigned* zero_words, int n_zero, const int32_t* out_perm, hipStream_t s) {
    if (rows <= 0) return 0;
    FH_REQUIRE(dim % 8 == 0 && dim <= 8 * 256 * 8, "gather_rms_norm: dim=%d must be a multiple of 8 and <= 16384", dim);
    const int chunks = cdiv(dim / 8, 256);
    dim3 grid(rows), block(256);
    const size_t lds = out_perm ? (size_t)dim * 2 : 0;
#define FH_GRN(C) hipLaunchKernelGGL((gather_rms_norm_kernel<IdxT, C>), grid, block, lds, s, table, ids, scale, copy_f16, copy_f32, w, eps, out, dim, zero_words, n_zero, out_perm)
    if (chunks <= 1) FH_GRN(1); else if (chunks <= 2) FH_GRN(2); else if (chunks <= 4) FH_GRN(4); else FH_GRN(8);
#undef FH_GRN
    FH_CHECK_LAUNCH();
    return 0;
}
// embedding_lookup (+ scale_inplace) + rms_norm of layer 0, the residual stream written on the way (fp16, and fp32 for
// sandwich-norm models)
int embed_rms_norm_f16(const __half* table, const uint32_t* token_ids, float embed_scale, __half* residual, float* residual_f32,
                       const __half* w, float eps, __half* norm_out, int tokens, int dim, unsigned* zero_words, int n_zero, hipStream_t s,
                       const int32_t* out_perm) {
    return launch_gather_rms<uint32_t>(table, token_ids, embed_scale, residual, residual_f32, w, eps, norm_out, tokens, dim, zero_words, n_zero, out_perm, s);
}
// gather_rows + rms_norm on the sampled rows (final norm)
int gather_rms_norm_f16(const __half* x, const int32_t* row_idx, const __half* w, float eps, __half* out, int rows, int dim, hipStream_t s) {
    return launch_gather_rms<int32_t>(x, row_idx, 0.0f, nullptr, nullptr, w, eps, out, rows, dim, nullptr, 0, nullptr, s);
}

// ── row gathers ──────────────────────────────────────────────────────────────
template <typename IdxT>
__global__ void gather_rows_kernel(const __half* __restrict__ table, const IdxT* __restrict__ ids,
                                   __half* __restrict__ out, int dim) {
    const long row = blockIdx.y;
    const long src = (long)ids[row];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (dim >> 3))
        *reinterpret_cast<half8*>(out + row * dim + i * 8) = *reinterpret_cast<const half8*>(table + src * dim + i * 8);
}

int embedding_lookup_f16(const __half* table, const uint32_t* ids, __half* out, int n_ids, int dim, hipStream_t s) {
    if (n_ids <= 0) return 0;
    FH_REQUIRE(dim % 8 == 0, "embedding_lookup: dim=%d must be a multiple of 8", dim);
    hipLaunchKernelGGL(gather_rows_kernel<uint32_t>, dim3(cdiv(dim / 8, 256), n_ids), dim3(256), 0, s, table, ids, out, dim);
    FH_CHECK_LAUNCH();
    return 0;
}

// ── fp32 residual stream (Gemma-3 sandwich norms) ────────────────────────────────────────────────
// residual += rms_norm(branch)·w_branch (all f32, branch read as fp16);  norm_out = f16(rms_norm(residual)·w_next).
// One 256-thread workgroup per token, the row held in registers between the two reductions (H ≤ 8192).
template <bool SLABS>
__global__ __launch_bounds__(256) void sandwich_add_norm_f32_kernel(const __half* __restrict__ branch,
                                                                     const float* __restrict__ slabs, int S, long slab_stride,
                                                                     int ld_slab, const __half* __restrict__ w_branch,
                                                                     float* __restrict__ residual,
                                                                     const __half* __restrict__ w_next, float eps,
                                                                     __half* __restrict__ norm_out, int H,
                                                                     const int32_t* __restrict__ out_perm) {
    __shared__ float red[4];
    extern __shared__ _Float16 stage[];            // H halves when out_perm
    const long row = blockIdx.x;
    const int nvec = H >> 3;
    constexpr int CH = 4;
    float x[CH][8], r[CH][8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int i = threadIdx.x + c * 256;
        if (i < nvec) {
            if (SLABS) {   // branch = S fp32 split-K slabs, summed in slab order and rounded like the fp16 GEMM output
                float acc[8];
                reduce_slabs8(slabs + row * ld_slab + i * 8, slab_stride, S, acc);     // all slab loads in flight at once
#pragma unroll
                for (int j = 0; j < 8; j++) { x[c][j] = (float)(_Float16)acc[j]; ss += x[c][j] * x[c][j]; }
            } else {
                const half8 v = *reinterpret_cast<const half8*>(branch + row * H + i * 8);
#pragma unroll
                for (int j = 0; j < 8; j++) { x[c][j] = (float)v[j]; ss += x[c][j] * x[c][j]; }
            }
        }
    }
    auto block_sum = [&](float v) {
        v = wave_reduce_sum(v);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        const float t = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
        return t;
    };
    const float inv1 = 1.0f / sqrtf(block_sum(ss) / (float)H + eps);
    float ss2 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int i = threadIdx.x + c * 256;
        if (i < nvec) {
            const half8 wv = *reinterpret_cast<const half8*>(w_branch + i * 8);
            float4v* rp = reinterpret_cast<float4v*>(residual + row * H + i * 8);
            float4v r0 = rp[0], r1 = rp[1];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float b = x[c][j] * inv1 * (float)wv[j];
                r[c][j] = (j < 4 ? r0[j] : r1[j - 4]) + b;
                ss2 += r[c][j] * r[c][j];
            }
            rp[0] = (float4v){r[c][0], r[c][1], r[c][2], r[c][3]};
            rp[1] = (float4v){r[c][4], r[c][5], r[c][6], r[c][7]};
        }
    }
    if (w_next == nullptr) return;
    const float inv2 = 1.0f / sqrtf(block_sum(ss2) / (float)H + eps);
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int i = threadIdx.x + c * 256;
        if (i < nvec) {
            const half8 wv = *reinterpret_cast<const half8*>(w_next + i * 8);
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (_Float16)(r[c][j] * inv2 * (float)wv[j]);
            if (out_perm) *reinterpret_cast<half8*>(stage + i * 8) = o;
            else *reinterpret_cast<half8*>(norm_out + row * H + i * 8) = o;
        }
    }
    if (out_perm) store_row_gathered_256(norm_out + row * H, stage, out_perm, nvec);
}

// The same for a decode batch (≤ 64 rows): 1024 threads per row, ONE 16-byte vector per thread, and everything that does not
// depend on a reduction — the branch row or its S slabs, the residual row, both norm weights — requested before the first
// block sum.  The 256-thread form above walks three chunks per thread at hidden 5376 with the slab loads of each behind the
// previous one and the residual / weight loads behind the reductions: 9.8 µs per launch at 32 rows (two per Gemma-3 layer,
// 14 % of its decode step); the row is latency, not bytes.  (Another summation order of the row sums: results equal within
// fp32 rounding of the norm scale.)
template <bool SLABS>
__global__ __launch_bounds__(1024) void sandwich_add_norm_f32_wide_kernel(const __half* __restrict__ branch,
                                                                           const float* __restrict__ slabs, int S, long slab_stride,
                                                                           int ld_slab, const __half* __restrict__ w_branch,
                                                                           float* __restrict__ residual,
                                                                           const __half* __restrict__ w_next, float eps,
                                                                           __half* __restrict__ norm_out, int H,
                                                                           const int32_t* __restrict__ out_perm) {
    __shared__ float red[16];
    extern __shared__ _Float16 stage[];            // H halves when out_perm
    const long row = blockIdx.x;
    const int nvec = H >> 3, i = threadIdx.x;
    const bool on = i < nvec;
    const int ic = on ? i : 0;
    float x[8];
    if (SLABS) {
        reduce_slabs8(slabs + row * ld_slab + ic * 8, slab_stride, S, x);
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = (float)(_Float16)x[j];
    } else {
        const half8 v = *reinterpret_cast<const half8*>(branch + row * H + ic * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = (float)v[j];
    }
    const half8 wb = *reinterpret_cast<const half8*>(w_branch + ic * 8);
    const half8 wn = *reinterpret_cast<const half8*>((w_next ? w_next : w_branch) + ic * 8);
    float4v* rp = reinterpret_cast<float4v*>(residual + row * H + ic * 8);
    const float4v r0 = rp[0], r1 = rp[1];
    auto block_sum = [&](float v) {
        v = wave_reduce_sum(v);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; w += 4) t += (red[w] + red[w + 1]) + (red[w + 2] + red[w + 3]);
        __syncthreads();
        return t;
    };
    float ss = 0.f;
    if (on) {
#pragma unroll
        for (int j = 0; j < 8; j++) ss += x[j] * x[j];
    }
    const float inv1 = 1.0f / sqrtf(block_sum(ss) / (float)H + eps);
    float r[8], ss2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        r[j] = (j < 4 ? r0[j] : r1[j - 4]) + x[j] * inv1 * (float)wb[j];
        if (on) ss2 += r[j] * r[j];
    }
    if (on) {
        rp[0] = (float4v){r[0], r[1], r[2], r[3]};
        rp[1] = (float4v){r[4], r[5], r[6], r[7]};
    }
    if (w_next == nullptr) return;
    const float inv2 = 1.0f / sqrtf(block_sum(ss2) / (float)H + eps);
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = (_Float16)(r[j] * inv2 * (float)wn[j]);
    if (!out_perm) {
        if (on) *reinterpret_cast<half8*>(norm_out + row * H + i * 8) = o;
        return;
    }
    if (on) *reinterpret_cast<half8*>(stage + i * 8) = o;
    __syncthreads();
    if (on) *reinterpret_cast<half8*>(norm_out + row * H + i * 8) = lds_gather8(stage, out_perm, i * 8);
}

int sandwich_add_rms_norm_f32(const __half* branch, const __half* w_branch, float* residual, const __half* w_next, float eps,
                              __half* norm_out, int tokens, int dim, hipStream_t s, const int32_t* out_perm) {
    if (tokens <= 0) return 0;
    FH_REQUIRE(dim % 8 == 0 && dim <= 8192, "sandwich_add_rms_norm_f32: dim=%d must be a multiple of 8, <= 8192", dim);
    if (tokens <= 64 && knobs().sandwich_wide) {
        hipLaunchKernelGGL(sandwich_add_norm_f32_wide_kernel<false>, dim3(tokens), dim3(1024), out_perm ? (size_t)dim * 2 : 0, s, branch, nullptr, 0,
                           0L, 0, w_branch, residual, w_next, eps, norm_out, dim, out_perm);
        FH_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(sandwich_add_norm_f32_kernel<false>, dim3(tokens), dim3(256), out_perm ? (size_t)dim * 2 : 0, s, branch, nullptr, 0, 0L, 0,
                       w_branch, residual, w_next, eps, norm_out, dim, out_perm);
    FH_CHECK_LAUNCH();
    return 0;
}
int sandwich_add_rms_norm_f32_slabs(const float* slabs, int S, long slab_stride, int ld_slab, const __half* w_branch,
                                    float* residual, const __half* w_next, float eps, __half* norm_out, int tokens, int dim,
                                    hipStream_t s, const int32_t* out_perm) {
    if (tokens <= 0) return 0;
    FH_REQUIRE(dim % 8 == 0 && dim <= 8192 && S >= 1, "sandwich_add_rms_norm_f32_slabs: dim=%d S=%d", dim, S);
    if (tokens <= 64 && knobs().sandwich_wide) {
        hipLaunchKernelGGL(sandwich_add_norm_f32_wide_kernel<true>, dim3(tokens), dim3(1024), out_perm ? (size_t)dim * 2 : 0, s, nullptr, slabs, S,
                           slab_stride, ld_slab, w_branch, residual, w_next, eps, norm_out, dim, out_perm);
        FH_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(sandwich_add_norm_f32_kernel<true>, dim3(tokens), dim3(256), out_perm ? (size_t)dim * 2 : 0, s, nullptr, slabs, S, slab_stride,
                       ld_slab, w_branch, residual, w_next, eps, norm_out, dim, out_perm);
    FH_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(256) void rms_norm_f32_to_f16_kernel(const float* __restrict__ x, const int32_t* __restrict__ row_idx,
                                                                   const __half* __restrict__ w, float eps,
                                                                   __half* __restrict__ out, int H) {
    __shared__ float red[4];
    const long src = row_idx ? row_idx[blockIdx.x] : blockIdx.x;
    const long dst = blockIdx.x;
    const int nvec = H >> 3;
    constexpr int CH = 4;
    float v[CH][8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int i = threadIdx.x + c * 256;
        if (i < nvec) {
            const float4v* p = reinterpret_cast<const float4v*>(x + src * H + i * 8);
            const float4v a = p[0], b = p[1];
#pragma unroll
            for (int j = 0; j < 8; j++) { v[c][j] = j < 4 ? a[j] : b[j - 4]; ss += v[c][j] * v[c][j]; }
        }
    }
    ss = wave_reduce_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float inv = 1.0f / sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)H + eps);
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int i = threadIdx.x + c * 256;
        if (i < nvec) {
            const half8 wv = *reinterpret_cast<const half8*>(w + i * 8);
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (_Float16)(v[c][j] * inv * (float)wv[j]);
            *reinterpret_cast<half8*>(out + dst * H + i * 8) = o;
        }
    }
}

int rms_norm_f32_to_f16(const float* x, const int32_t* row_idx, const __half* w, float eps, __half* out, int n_rows, int dim,
                        hipStream_t s) {
    if (n_rows <= 0) return 0;
    FH_REQUIRE(dim % 8 == 0 && dim <= 8192, "rms_norm_f32_to_f16: dim=%d must be a multiple of 8, <= 8192", dim);
    hipLaunchKernelGGL(rms_norm_f32_to_f16_kernel, dim3(n_rows), dim3(256), 0, s, x, row_idx, w, eps, out, dim);
    FH_CHECK_LAUNCH();
    return 0;
}

int gather_rows_f16(const __half* in, const int32_t* row_idx, __half* out, int n_rows, int dim, hipStream_t s) {
    if (n_rows <= 0) return 0;
    FH_REQUIRE(dim % 8 == 0, "gather_rows: dim=%d must be a multiple of 8", dim);
    hipLaunchKernelGGL(gather_rows_kernel<int32_t>, dim3(cdiv(dim / 8, 256), n_rows), dim3(256), 0, s, in, row_idx, out, dim);
    FH_CHECK_LAUNCH();
    return 0;
}

// ── gated activations: [T, 2I] (gate columns, then up columns) → [T, I] ──────
template <bool GELU>
__global__ void gated_act_kernel(const __half* __restrict__ gate_up, __half* __restrict__ out, int im) {
    const long t = blockIdx.y;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (im >> 3)) return;
    half8 g = *reinterpret_cast<const half8*>(gate_up + t * 2 * im + i * 8);
    half8 u = *reinterpret_cast<const half8*>(gate_up + t * 2 * im + im + i * 8);
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        float gf = (float)g[j], uf = (float)u[j], a;
        if (GELU) {
            float inner = 0.79788456f * (gf + 0.044715f * gf * gf * gf);
            a = 0.5f * gf * (1.0f + tanhf(inner));
        } else {
            a = gf / (1.0f + __expf(-gf));
        }
        o[j] = (_Float16)(a * uf);
    }
    *reinterpret_cast<half8*>(out + t * im + i * 8) = o;
}

// Same, with the gate_up projection arriving as S fp32 split-K slabs [S][rows_pad][ld]: summed in slab order and rounded
// to fp16 first, so the result is bit-identical to reduce → fused_*_mul_split (one launch instead of two).
template <bool GELU>
__global__ void gated_act_slabs_kernel(const float* __restrict__ slabs, int S, long slab_stride, int ld,
                                       __half* __restrict__ out, int im) {
    const long t = blockIdx.y;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (im >> 3)) return;
    float g[8], u[8];
    reduce_slabs8(slabs + t * ld + i * 8, slab_stride, S, g);          // all slab loads in flight at once (a runtime-bound
    reduce_slabs8(slabs + t * ld + im + i * 8, slab_stride, S, u);     // loop pays one memory round trip per slab)
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float gf = (float)(_Float16)g[j], uf = (float)(_Float16)u[j];
        float a;
        if (GELU) {
            float inner = 0.79788456f * (gf + 0.044715f * gf * gf * gf);
            a = 0.5f * gf * (1.0f + tanhf(inner));
        } else {
            a = gf / (1.0f + __expf(-gf));
        }
        o[j] = (_Float16)(a * uf);
    }
    *reinterpret_cast<half8*>(out + t * im + i * 8) = o;
}

int fused_gated_act_slabs_f16(const float* slabs, int S, long slab_stride, int ld, __half* out, int tokens, int im, int gelu,
                              hipStream_t s) {
    if (tokens <= 0) return 0;
    FH_REQUIRE(im % 8 == 0 && S >= 1 && ld >= 2 * im, "fused_gated_act_slabs: intermediate=%d S=%d ld=%d", im, S, ld);
    dim3 grid(cdiv(im / 8, 256), tokens);
    if (gelu) hipLaunchKernelGGL(gated_act_slabs_kernel<true>, grid, dim3(256), 0, s, slabs, S, slab_stride, ld, out, im);
    else hipLaunchKernelGGL(gated_act_slabs_kernel<false>, grid, dim3(256), 0, s, slabs, S, slab_stride, ld, out, im);
    FH_CHECK_LAUNCH();
    return 0;
}

int fused_silu_mul_split_f16(const __half* gate_up, __half* out, int tokens, int im, hipStream_t s) {
    if (tokens <= 0) return 0;
    FH_REQUIRE(im % 8 == 0, "fused_silu_mul_split: intermediate=%d must be a multiple of 8", im);
    hipLaunchKernelGGL(gated_act_kernel<false>, dim3(cdiv(im / 8, 256), tokens), dim3(256), 0, s, gate_up, out, im);
    FH_CHECK_LAUNCH();
    return 0;
}
int fused_gelu_tanh_mul_split_f16(const __half* gate_up, __half* out, int tokens, int im, hipStream_t s) {
    if (tokens <= 0) return 0;
    FH_REQUIRE(im % 8 == 0, "fused_gelu_tanh_mul_split: intermediate=%d must be a multiple of 8", im);
    hipLaunchKernelGGL(gated_act_kernel<true>, dim3(cdiv(im / 8, 256), tokens), dim3(256), 0, s, gate_up, out, im);
    FH_CHECK_LAUNCH();
    return 0;
}

// ── flat elementwise ─────────────────────────────────────────────────────────
__global__ void add_inplace_kernel(__half* __restrict__ r, const __half* __restrict__ x, long len) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= len) {
        half8 a = *reinterpret_cast<half8*>(r + i), b = *reinterpret_cast<const half8*>(x + i);
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = (_Float16)((float)a[j] + (float)b[j]);
        *reinterpret_cast<half8*>(r + i) = a;
    } else {
        for (long j = i; j < len; j++) r[j] = __float2half(__half2float(r[j]) + __half2float(x[j]));
    }
}
int add_inplace_f16(__half* residual, const __half* x, long len, hipStream_t s) {
    if (len <= 0) return 0;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(cdiv(cdiv(len, 8), 256)), dim3(256), 0, s, residual, x, len);
    FH_CHECK_LAUNCH();
    return 0;
}

__global__ void scale_inplace_kernel(__half* __restrict__ buf, float scale, long len) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) buf[i] = __float2half(__half2float(buf[i]) * scale);
}
int scale_inplace_f16(__half* buf, float scale, long len, hipStream_t s) {
    if (len <= 0) return 0;
    hipLaunchKernelGGL(scale_inplace_kernel, dim3(cdiv(len, 256)), dim3(256), 0, s, buf, scale, len);
    FH_CHECK_LAUNCH();
    return 0;
}

__global__ void add_bias_kernel(__half* __restrict__ data, const __half* __restrict__ bias, int cols) {
    long r = blockIdx.y;
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < cols) data[r * cols + c] = __float2half(__half2float(data[r * cols + c]) + __half2float(bias[c]));
}
int add_bias_f16(__half* data, const __half* bias, int rows, int cols, hipStream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(add_bias_kernel, dim3(cdiv(cols, 256), rows), dim3(256), 0, s, data, bias, cols);
    FH_CHECK_LAUNCH();
    return 0;
}

// Backend::layer_norm (traits.rs; CPU cpu.rs:2081-2113): out = (x − mean)·(1/√(var + eps))·γ + β over `dim`, one workgroup per
// token row (mean and variance in two passes over the row held in registers, fp32 — the CPU path accumulates them in f64).
__global__ __launch_bounds__(256) void layer_norm_kernel(const __half* __restrict__ x, const __half* __restrict__ gamma,
                                                        const __half* __restrict__ beta, float eps, __half* __restrict__ out, int dim) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float v[32];                                       // dim ≤ 8192
    float s = 0.f;
    int n = 0;
    for (int i = threadIdx.x; i < dim; i += 256) { v[n] = __half2float(x[row * dim + i]); s += v[n]; n++; }
    s = wave_reduce_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)dim;
    __syncthreads();
    float q = 0.f;
    for (int j = 0; j < n; j++) { const float d = v[j] - mean; q += d * d; }
    q = wave_reduce_sum(q);
    if (lane == 0) red[wave] = q;
    __syncthreads();
    const float inv = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)dim + eps);
    n = 0;
    for (int i = threadIdx.x; i < dim; i += 256) {
        out[row * dim + i] = __float2half((v[n] - mean) * inv * __half2float(gamma[i]) + __half2float(beta[i]));
        n++;
    }
}
int layer_norm_f16(const __half* x, const __half* gamma, const __half* beta, float eps, __half* out, int tokens, int dim, hipStream_t s) {
    if (tokens <= 0) return 0;
    FH_REQUIRE(dim > 0 && dim <= 8192, "layer_norm: dim=%d must be in [1, 8192]", dim);
    hipLaunchKernelGGL(layer_norm_kernel, dim3(tokens), dim3(256), 0, s, x, gamma, beta, eps, out, dim);
    FH_CHECK_LAUNCH();
    return 0;
}

// Backend::gelu (CPU cpu.rs:2115-2122): exact-form GELU 0.5·x·(1 + erf(x/√2)) with the reference's own erf — the
// Abramowitz–Stegun 7.1.26 polynomial of cpu.rs:2263-2273 — so both sides round the same function.
__global__ void gelu_kernel(const __half* __restrict__ x, __half* __restrict__ out, long len) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const float xi = __half2float(x[i]);
    const float z = xi / 1.41421356237309504880f;
    const float sign = z < 0.f ? -1.0f : 1.0f, az = fabsf(z);
    const float t = 1.0f / (1.0f + 0.3275911f * az);
    const float y = 1.0f - (((((1.0614054f * t - 1.4531521f) * t) + 1.4214138f) * t - 0.28449672f) * t + 0.2548296f) * t * expf(-az * az);
    out[i] = __float2half(0.5f * xi * (1.0f + sign * y));
}
int gelu_f16(const __half* x, __half* out, long len, hipStream_t s) {
    if (len <= 0) return 0;
    hipLaunchKernelGGL(gelu_kernel, dim3(cdiv(len, 256)), dim3(256), 0, s, x, out, len);
    FH_CHECK_LAUNCH();
    return 0;
}

// act-order input gather A'[m, j] = A[m, perm[j]] (kernels/gather_columns.cu:15).
__global__ void gather_columns_kernel(const __half* __restrict__ in, const int32_t* __restrict__ perm,
                                      __half* __restrict__ out, int cols) {
    long r = blockIdx.y;
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < cols) out[r * cols + c] = in[r * cols + perm[c]];
}
// Rows that fit LDS (≤ 32768 columns): the row is read once with 16-byte loads, gathered from LDS and written with 16-byte
// stores (2-byte global gathers took 150 µs for 8192 × 4096; this form runs at the copy rate).
__global__ __launch_bounds__(256) void gather_columns_lds_kernel(const __half* __restrict__ in, const int32_t* __restrict__ perm,
                                                                 __half* __restrict__ out, int cols) {
    extern __shared__ _Float16 stage[];
    const long r = blockIdx.x;
    const int nvec = cols >> 3;
    for (int i = threadIdx.x; i < nvec; i += 256)
        *reinterpret_cast<half8*>(stage + i * 8) = *reinterpret_cast<const half8*>(in + r * cols + i * 8);
    store_row_gathered_256(out + r * cols, stage, perm, nvec);
}
int gather_columns_f16(const __half* in, const int32_t* perm, __half* out, int rows, int cols, hipStream_t s) {
    if (rows <= 0) return 0;
    form_hit(FORM_GATHER_COLUMNS);
    if (cols % 8 == 0 && cols <= 32768) {
        hipLaunchKernelGGL(gather_columns_lds_kernel, dim3(rows), dim3(256), (size_t)cols * 2, s, in, perm, out, cols);
        FH_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(gather_columns_kernel, dim3(cdiv(cols, 256), rows), dim3(256), 0, s, in, perm, out, cols);
    FH_CHECK_LAUNCH();
    return 0;
}

}  // namespace fh
