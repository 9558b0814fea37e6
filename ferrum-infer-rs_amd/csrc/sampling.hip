// Device-side greedy sampling: row argmax (+ token mask) and sparse repetition penalty.
//
// Reference: `Backend::argmax_rows_f16[_masked|_sparse_repetition_penalty]`
// (ferrum-kernels/src/backend/traits.rs:1534-1591; CUDA kernels/argmax_rows.cu:16,56,123).
// Tie-break is the trait default's: strict `>` ⇒ FIRST maximum (traits.rs:1547).
#include "common.h"
#include "kernels.h"

namespace fh {

template <typename T>
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const T* __restrict__ logits, uint32_t* __restrict__ out,
                                                           const uint8_t* __restrict__ mask, int mask_len, int n) {
    __shared__ float s_val[16];
    __shared__ int s_idx[16];
    const long row = blockIdx.x;
    const T* p = logits + row * n;
    float best = -INFINITY;
    int best_idx = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        if (mask && (i >= mask_len || mask[i] == 0)) continue;
        float v = (float)p[i];
        if (v > best) { best = v; best_idx = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float ob = __shfl_xor(best, off, 64);
        int oi = __shfl_xor(best_idx, off, 64);
        if (ob > best || (ob == best && oi < best_idx)) { best = ob; best_idx = oi; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { s_val[wave] = best; s_idx[wave] = best_idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int nw = blockDim.x >> 6;
        for (int w = 1; w < nw; w++)
            if (s_val[w] > best || (s_val[w] == best && s_idx[w] < best_idx)) { best = s_val[w]; best_idx = s_idx[w]; }
        out[row] = best_idx == 0x7fffffff ? 0u : (uint32_t)best_idx;   // all -inf/NaN row → 0 (max_idx init)
    }
}

// Two-stage form for wide rows (a vocabulary-sized row on ONE workgroup is bound by what one CU can pull, ≈25 GB/s):
// stage 1 grid (rows, C chunks) → (value, index) partials in the workspace; stage 2 one wave per row.  Same
// tie-break: first maximum.
template <typename T>
__global__ __launch_bounds__(256) void argmax_partial_kernel(const T* __restrict__ logits, float* __restrict__ pval,
                                                             int* __restrict__ pidx, const uint8_t* __restrict__ mask,
                                                             int mask_len, int n, int chunk) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    const long row = blockIdx.x;
    const int c = blockIdx.y, C = gridDim.y;
    const int lo = c * chunk, hi = min(n, lo + chunk);
    const T* p = logits + row * n;
    float best = -INFINITY;
    int best_idx = 0x7fffffff;
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        if (mask && (i >= mask_len || mask[i] == 0)) continue;
        float v = (float)p[i];
        if (v > best) { best = v; best_idx = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float ob = __shfl_xor(best, off, 64);
        int oi = __shfl_xor(best_idx, off, 64);
        if (ob > best || (ob == best && oi < best_idx)) { best = ob; best_idx = oi; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { s_val[wave] = best; s_idx[wave] = best_idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++)
            if (s_val[w] > best || (s_val[w] == best && s_idx[w] < best_idx)) { best = s_val[w]; best_idx = s_idx[w]; }
        pval[row * C + c] = best;
        pidx[row * C + c] = best_idx;
    }
}

__global__ __launch_bounds__(64) void argmax_final_kernel(const float* __restrict__ pval, const int* __restrict__ pidx,
                                                          uint32_t* __restrict__ out, int C) {
    const long row = blockIdx.x;
    float best = -INFINITY;
    int best_idx = 0x7fffffff;
    for (int c = threadIdx.x; c < C; c += 64) {
        float v = pval[row * C + c];
        int i = pidx[row * C + c];
        if (v > best || (v == best && i < best_idx)) { best = v; best_idx = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float ob = __shfl_xor(best, off, 64);
        int oi = __shfl_xor(best_idx, off, 64);
        if (ob > best || (ob == best && oi < best_idx)) { best = ob; best_idx = oi; }
    }
    if (threadIdx.x == 0) out[row] = best_idx == 0x7fffffff ? 0u : (uint32_t)best_idx;
}

template <typename T>
static int argmax_rows_ws(const T* logits, uint32_t* out_ids, const uint8_t* valid_mask, int mask_len, int m, int n,
                          float* workspace, size_t workspace_bytes, hipStream_t s) {
    if (m <= 0) return 0;
    int C = std::min(128, std::max(1, n / 2048));
    if (workspace == nullptr || workspace_bytes < (size_t)m * C * 8 || C < 2) {
        hipLaunchKernelGGL(argmax_rows_kernel<T>, dim3(m), dim3(1024), 0, s, logits, out_ids, valid_mask, mask_len, n);
        FH_CHECK_LAUNCH();
        return 0;
    }
    const int chunk = cdiv(n, C);
    float* pval = workspace;
    int* pidx = reinterpret_cast<int*>(workspace + (size_t)m * C);
    hipLaunchKernelGGL(argmax_partial_kernel<T>, dim3(m, C), dim3(256), 0, s, logits, pval, pidx, valid_mask, mask_len, n, chunk);
    FH_CHECK_LAUNCH();
    hipLaunchKernelGGL(argmax_final_kernel, dim3(m), dim3(64), 0, s, pval, pidx, out_ids, C);
    FH_CHECK_LAUNCH();
    return 0;
}
int argmax_rows_f16_ws(const __half* logits, uint32_t* out_ids, const uint8_t* valid_mask, int mask_len, int m, int n,
                       float* workspace, size_t workspace_bytes, hipStream_t s) {
    return argmax_rows_ws(logits, out_ids, valid_mask, mask_len, m, n, workspace, workspace_bytes, s);
}
int argmax_rows_f32_ws(const float* logits, uint32_t* out_ids, const uint8_t* valid_mask, int mask_len, int m, int n,
                       float* workspace, size_t workspace_bytes, hipStream_t s) {
    return argmax_rows_ws(logits, out_ids, valid_mask, mask_len, m, n, workspace, workspace_bytes, s);
}

int argmax_rows_f16(const __half* logits, uint32_t* out_ids, const uint8_t* valid_mask, int mask_len, int m, int n,
                    hipStream_t s) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(argmax_rows_kernel<__half>, dim3(m), dim3(1024), 0, s, logits, out_ids, valid_mask, mask_len, n);
    FH_CHECK_LAUNCH();
    return 0;
}
int argmax_rows_f32(const float* logits, uint32_t* out_ids, const uint8_t* valid_mask, int mask_len, int m, int n,
                    hipStream_t s) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(argmax_rows_kernel<float>, dim3(m), dim3(1024), 0, s, logits, out_ids, valid_mask, mask_len, n);
    FH_CHECK_LAUNCH();
    return 0;
}

// logits[row][id] = v > 0 ? v / p : v · p for every id listed for the row (caller de-duplicates,
// as the reference host does: ferrum-interfaces/src/sampler.rs:327-345).
template <typename T>
__global__ void rep_penalty_kernel(T* __restrict__ logits, const uint32_t* __restrict__ row_offsets,
                                   const uint32_t* __restrict__ token_ids, const float* __restrict__ penalties, int n) {
    const int row = blockIdx.x;
    const uint32_t lo = row_offsets[row], hi = row_offsets[row + 1];
    const float pen = penalties[row];
    if (pen == 1.0f) return;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        uint32_t id = token_ids[i];
        if (id >= (uint32_t)n) continue;
        float v = (float)logits[(long)row * n + id];
        logits[(long)row * n + id] = (T)(v > 0.f ? v / pen : v * pen);
    }
}

int apply_repetition_penalties_sparse_f16(__half* logits, const uint32_t* row_offsets, const uint32_t* token_ids,
                                          const float* penalties, int m, int n, hipStream_t s) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(rep_penalty_kernel<__half>, dim3(m), dim3(256), 0, s, logits, row_offsets, token_ids, penalties, n);
    FH_CHECK_LAUNCH();
    return 0;
}
int apply_repetition_penalties_sparse_f32(float* logits, const uint32_t* row_offsets, const uint32_t* token_ids,
                                          const float* penalties, int m, int n, hipStream_t s) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(rep_penalty_kernel<float>, dim3(m), dim3(256), 0, s, logits, row_offsets, token_ids, penalties, n);
    FH_CHECK_LAUNCH();
    return 0;
}

}  // namespace fh
