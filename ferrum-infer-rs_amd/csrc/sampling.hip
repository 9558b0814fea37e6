// Device-side greedy sampling: row argmax (+ token mask) and sparse repetition penalty.
//
// Reference: `Backend::argmax_rows_f16[_masked|_sparse_repetition_penalty]`
// (ferrum-kernels/src/backend/traits.rs:1534-1591; CUDA kernels/argmax_rows.cu:16,56,123).
// Tie-break is the trait default's: strict `>` ⇒ FIRST maximum (traits.rs:1547).
#include "common.h"
#include "kernels.h"

namespace fh {

// (value, index) wave reduction under the total order "larger value, then smaller index" — exact whatever the pairing, so
// the DPP / permlane form (no LDS round trips, see common.h) gives the same answer as any other tree.
template <int CTRL>
__device__ __forceinline__ void argmax_dpp_step(float& best, int& idx) {
    float ob = dpp_move<CTRL>(best);
    int oi = __builtin_amdgcn_update_dpp(0, idx, CTRL, 0xF, 0xF, true);
    if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
}
__device__ __forceinline__ void argmax_pick(float& best, int& idx, float v0, int i0, float v1, int i1) {
    const bool one = v1 > v0 || (v1 == v0 && i1 < i0);
    best = one ? v1 : v0;
    idx = one ? i1 : i0;
}
__device__ __forceinline__ void wave_argmax(float& best, int& idx) {
    argmax_dpp_step<0xB1>(best, idx);
    argmax_dpp_step<0x4E>(best, idx);
    argmax_dpp_step<0x141>(best, idx);
    argmax_dpp_step<0x140>(best, idx);
    auto rv = __builtin_amdgcn_permlane16_swap(__float_as_uint(best), __float_as_uint(best), false, false);
    auto ri = __builtin_amdgcn_permlane16_swap((unsigned)idx, (unsigned)idx, false, false);
    argmax_pick(best, idx, __uint_as_float(rv[0]), (int)ri[0], __uint_as_float(rv[1]), (int)ri[1]);
    rv = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
    ri = __builtin_amdgcn_permlane32_swap((unsigned)idx, (unsigned)idx, false, false);
    argmax_pick(best, idx, __uint_as_float(rv[0]), (int)ri[0], __uint_as_float(rv[1]), (int)ri[1]);
}

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
    wave_argmax(best, best_idx);
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
template <typename T, bool VEC>
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
    if (VEC) {
        // chunk == 2048 and n % 8 == 0: one 8-element request per thread (plus the 8 mask bytes), a single round trip per
        // workgroup instead of eight 2-byte ones.
        const int i0 = lo + threadIdx.x * 8;
        if (i0 < hi) {
            T v8[8];
            if (sizeof(T) == 2) {
                *reinterpret_cast<uint4*>(v8) = *reinterpret_cast<const uint4*>(p + i0);
            } else {
                reinterpret_cast<uint4*>(v8)[0] = reinterpret_cast<const uint4*>(p + i0)[0];
                reinterpret_cast<uint4*>(v8)[1] = reinterpret_cast<const uint4*>(p + i0)[1];
            }
            unsigned long long mk = ~0ull;
            if (mask) {
                if (i0 + 8 <= mask_len && (reinterpret_cast<uintptr_t>(mask) & 7) == 0) {
                    mk = *reinterpret_cast<const unsigned long long*>(mask + i0);
                } else {
                    mk = 0;
                    for (int j = 0; j < 8; j++)
                        if (i0 + j < mask_len && mask[i0 + j]) mk |= 0xffull << (8 * j);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float v = (float)v8[j];
                if (((mk >> (8 * j)) & 0xff) && v > best) { best = v; best_idx = i0 + j; }
            }
        }
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += 256) {
            if (mask && (i >= mask_len || mask[i] == 0)) continue;
            float v = (float)p[i];
            if (v > best) { best = v; best_idx = i; }
        }
    }
    wave_argmax(best, best_idx);
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
                                                          uint32_t* __restrict__ out, int C, DecodeAdvance adv) {
    const long row = blockIdx.x;
    // decode loop: every row reads the step index before any row can advance it (the LAST row to finish does)
    const int step = adv.tokens ? *adv.step_counter : 0;
    float best = -INFINITY;
    int best_idx = 0x7fffffff;
    for (int c = threadIdx.x; c < C; c += 64) {
        float v = pval[row * C + c];
        int i = pidx[row * C + c];
        if (v > best || (v == best && i < best_idx)) { best = v; best_idx = i; }
    }
    wave_argmax(best, best_idx);
    if (threadIdx.x == 0) {
        const uint32_t id = best_idx == 0x7fffffff ? 0u : (uint32_t)best_idx;
        out[row] = id;
        if (adv.tokens && row < adv.n) {
            adv.tokens[row] = id;
            adv.pos_offsets[row] += 1;
            adv.kv_lens[row] += 1;
            adv.history[(long)step * adv.n + row] = id;
            const unsigned t = __hip_atomic_fetch_add(adv.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == gridDim.x - 1) {
                __hip_atomic_store(adv.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *adv.step_counter = step + 1;
            }
        }
    }
}

template <typename T>
static int argmax_rows_ws(const T* logits, uint32_t* out_ids, const uint8_t* valid_mask, int mask_len, int m, int n,
                          float* workspace, size_t workspace_bytes, hipStream_t s, const DecodeAdvance* adv = nullptr, int* fused = nullptr) {
    if (fused) *fused = 0;
    if (m <= 0) return 0;
    // Vector form: 2048-element chunks (one 16/32-byte request per thread) when rows stay 16-byte aligned.
    const bool vec = n % 8 == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 && n >= 4096 &&
                     workspace != nullptr && workspace_bytes >= (size_t)m * cdiv(n, 2048) * 8;
    int C = vec ? cdiv(n, 2048) : std::min(128, std::max(1, n / 2048));
    if (workspace == nullptr || workspace_bytes < (size_t)m * C * 8 || C < 2) {
        hipLaunchKernelGGL(argmax_rows_kernel<T>, dim3(m), dim3(1024), 0, s, logits, out_ids, valid_mask, mask_len, n);
        FH_CHECK_LAUNCH();
        return 0;
    }
    const int chunk = vec ? 2048 : cdiv(n, C);
    float* pval = workspace;
    int* pidx = reinterpret_cast<int*>(workspace + (size_t)m * C);
    if (vec)
        hipLaunchKernelGGL((argmax_partial_kernel<T, true>), dim3(m, C), dim3(256), 0, s, logits, pval, pidx, valid_mask, mask_len, n, chunk);
    else
        hipLaunchKernelGGL((argmax_partial_kernel<T, false>), dim3(m, C), dim3(256), 0, s, logits, pval, pidx, valid_mask, mask_len, n, chunk);
    FH_CHECK_LAUNCH();
    const bool adv_ok = adv && adv->tokens && adv->n == m;
    hipLaunchKernelGGL(argmax_final_kernel, dim3(m), dim3(64), 0, s, pval, pidx, out_ids, C, adv_ok ? *adv : DecodeAdvance());
    FH_CHECK_LAUNCH();
    if (fused && adv_ok) *fused = 1;
    return 0;
}
int argmax_rows_f32_ws_advance(const float* logits, uint32_t* out_ids, const uint8_t* valid_mask, int mask_len, int m, int n,
                               float* workspace, size_t workspace_bytes, const DecodeAdvance* adv, int* fused, hipStream_t s) {
    return argmax_rows_ws(logits, out_ids, valid_mask, mask_len, m, n, workspace, workspace_bytes, s, adv, fused);
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

// ── vocabulary-parallel greedy sampling ─────────────────────────────────────
// Every rank holds the logits of its own vocabulary rows [v0, v0 + n_local) and has taken its local argmax (first maximum).
// pairs[row] = (local winner's logit, GLOBAL id); after the all-gather the global winner is the first maximum over the ranks
// in rank order — rank r's ids all lie below rank r+1's, so "strictly greater wins" keeps the lowest id among equal logits,
// exactly the single-GPU tie-break (traits.rs:1547).
// A rank whose vocabulary slice holds NO valid id (a sparse token mask, or a mask that ends below the slice) must not compete:
// its local argmax fell back to local id 0, whose raw logit could beat the other ranks' valid maxima.  Such a row publishes
// (−inf, 0xffffffff); the merge ignores it and falls back to id 0 only when every rank reports none (the single-GPU masked
// argmax's own fallback, traits.rs:1571-1591).
__global__ void argmax_pairs_kernel(const float* __restrict__ logits, const uint32_t* __restrict__ local_ids, float2* __restrict__ pairs,
                                    int rows, int n_local, int v0, const uint8_t* __restrict__ mask, int mask_len) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) {
        const uint32_t id = local_ids[r];
        const bool valid = !mask || ((int)id < mask_len && mask[id] != 0);
        pairs[r] = valid ? make_float2(logits[(long)r * n_local + id], __uint_as_float(id + (uint32_t)v0))
                         : make_float2(-INFINITY, __uint_as_float(0xffffffffu));
    }
}
__global__ void argmax_merge_ranks_kernel(const float2* __restrict__ gathered, uint32_t* __restrict__ out, int rows, int world, DecodeAdvance adv) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int step = adv.tokens ? *adv.step_counter : 0;
    if (r < rows) {
        float best = -INFINITY;
        uint32_t best_id = 0xffffffffu;
        for (int p = 0; p < world; p++) {                  // first maximum in rank order = the lowest id among equal logits
            const float2 v = gathered[(long)p * rows + r];
            const uint32_t id = __float_as_uint(v.y);
            if (id != 0xffffffffu && (best_id == 0xffffffffu || v.x > best)) { best = v.x; best_id = id; }
        }
        if (best_id == 0xffffffffu) best_id = 0;           // no valid id on any rank
        out[r] = best_id;
        if (adv.tokens && r < adv.n) {
            adv.tokens[r] = best_id;
            adv.pos_offsets[r] += 1;
            adv.kv_lens[r] += 1;
            adv.history[(long)step * adv.n + r] = best_id;
        }
    }
    __syncthreads();                                   // one block (rows ≤ 1024): every row has read `step`
    if (adv.tokens && threadIdx.x == 0 && blockIdx.x == 0) *adv.step_counter = step + 1;
}
int argmax_pairs_f32(const float* logits, const uint32_t* local_ids, void* pairs, int rows, int n_local, int v0, const uint8_t* mask, int mask_len, hipStream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(argmax_pairs_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, s, logits, local_ids, (float2*)pairs, rows, n_local, v0, mask, mask_len);
    FH_CHECK_LAUNCH();
    return 0;
}
int argmax_merge_ranks(const void* gathered, uint32_t* out, int rows, int world, const DecodeAdvance* adv, hipStream_t s) {
    if (rows <= 0) return 0;
    FH_REQUIRE(rows <= 1024, "argmax_merge_ranks: %d rows > 1024", rows);
    hipLaunchKernelGGL(argmax_merge_ranks_kernel, dim3(1), dim3(1024), 0, s, (const float2*)gathered, out, rows, world, adv ? *adv : DecodeAdvance());
    FH_CHECK_LAUNCH();
    return 0;
}

// logits[row][id] = v > 0 ? v / p : v · p for every id listed for the row (caller de-duplicates,
// as the reference host does: ferrum-interfaces/src/sampler.rs:327-345).
template <typename T>
__global__ void rep_penalty_kernel(T* __restrict__ logits, const uint32_t* __restrict__ row_offsets,
                                   const uint32_t* __restrict__ token_ids, const float* __restrict__ penalties, int n, uint32_t id_offset = 0) {
    const int row = blockIdx.x;
    const uint32_t lo = row_offsets[row], hi = row_offsets[row + 1];
    const float pen = penalties[row];
    if (pen == 1.0f) return;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        uint32_t id = token_ids[i] - id_offset;               // vocabulary shard: ids below the shard wrap to huge values and are skipped
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
int apply_repetition_penalties_sparse_f32_shard(float* logits, const uint32_t* row_offsets, const uint32_t* token_ids,
                                                const float* penalties, int m, int n_local, int v0, hipStream_t s) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(rep_penalty_kernel<float>, dim3(m), dim3(256), 0, s, logits, row_offsets, token_ids, penalties, n_local, (uint32_t)v0);
    FH_CHECK_LAUNCH();
    return 0;
}

}  // namespace fh
