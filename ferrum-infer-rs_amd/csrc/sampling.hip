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
