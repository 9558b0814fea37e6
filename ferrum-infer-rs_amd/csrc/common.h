// Shared helpers for the gfx950 kernels and the C-ABI layer.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>

namespace fh {

// Thread-local error text returned by ferrum_hip_last_error().
void set_error(const char* fmt, ...);
const char* last_error();

#define FH_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            fh::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

#define FH_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            fh::set_error(__VA_ARGS__);       \
            return 2;                         \
        }                                     \
    } while (0)

#define FH_CHECK_LAUNCH()                                                            \
    do {                                                                             \
        hipError_t _e = hipGetLastError();                                           \
        if (_e != hipSuccess) {                                                      \
            fh::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

// Reductions over the four 16-lane rows of a wave (lanes L, L^16, L^32, L^48 — the MFMA 16×16 accumulator's key groups) with the
// gfx950 row/half swaps: v_permlane16_swap / v_permlane32_swap are VALU instructions, where __shfl_xor(·, 16 | 32) is a
// ds_bpermute round trip through the LDS crossbar.  Same pairing order as shfl_xor 16 then 32, so sums are bit-identical.
__device__ __forceinline__ float rows_reduce_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float rows_reduce_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// Whole-wave reductions (every lane gets the result) without LDS round trips: inside a 16-lane row by DPP (quad permutes for
// lane^1 and lane^2, then row_half_mirror and row_mirror), across the four rows by the permlane swaps above.  __shfl_xor is a
// ds_bpermute (≈ 100+ cycles each, six per reduction) on the critical path of the latency-bound decode kernels.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// Act-order consumers (desc_act GPTQ: the packed weight rows are sorted by quant group, w4_repack_host) read their input as
// x'[j] = x[perm[j]] (kernels/gather_columns.cu:15).  A producer that owns a whole row stages it in LDS and writes the gathered
// row itself — 16-byte stores, the permutation from L2 — instead of leaving a gather launch in front of the GEMM.
__device__ __forceinline__ half8 lds_gather8(const _Float16* row, const int32_t* __restrict__ perm, int j0) {
    const int4 p0 = *reinterpret_cast<const int4*>(perm + j0), p1 = *reinterpret_cast<const int4*>(perm + j0 + 4);
    half8 o;
    o[0] = row[p0.x]; o[1] = row[p0.y]; o[2] = row[p0.z]; o[3] = row[p0.w];
    o[4] = row[p1.x]; o[5] = row[p1.y]; o[6] = row[p1.z]; o[7] = row[p1.w];
    return o;
}
__device__ __forceinline__ float wave_reduce_sum(float v) {
    v += dpp_move<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);       // row_half_mirror
    v += dpp_move<0x140>(v);       // row_mirror
    return rows_reduce_sum(v);
}
__device__ __forceinline__ float wave_reduce_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    v = fmaxf(v, dpp_move<0x4E>(v));
    v = fmaxf(v, dpp_move<0x141>(v));
    v = fmaxf(v, dpp_move<0x140>(v));
    return rows_reduce_max(v);
}

// Sum S ≤ 8 fp32 split-K slabs of one 8-element span in slab order.  All 16 loads are issued up front (index
// clamped, not branched: a load under a runtime condition costs one L2 round trip per slab) and masked in the add.
__device__ __forceinline__ void reduce_slabs8(const float* __restrict__ base, long slab_stride, int S, float (&o)[8]) {
    float4v sv[8][2];
#pragma unroll
    for (int z = 0; z < 8; z++) {
        const float4v* sp = reinterpret_cast<const float4v*>(base + (long)(z < S ? z : S - 1) * slab_stride);
        sv[z][0] = sp[0];
        sv[z][1] = sp[1];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = 0.f;
#pragma unroll
    for (int z = 0; z < 8; z++) {
        if (z < S) {
#pragma unroll
            for (int j = 0; j < 4; j++) { o[j] += sv[z][0][j]; o[4 + j] += sv[z][1][j]; }
        }
    }
    for (int z = 8; z < S; z++) {          // S > 8: rare, serial
        const float4v* sp = reinterpret_cast<const float4v*>(base + (long)z * slab_stride);
        float4v s0 = sp[0], s1 = sp[1];
#pragma unroll
        for (int j = 0; j < 4; j++) { o[j] += s0[j]; o[4 + j] += s1[j]; }
    }
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
__host__ __device__ inline int cdiv_dev(int a, int b) { return (a + b - 1) / b; }

}  // namespace fh
