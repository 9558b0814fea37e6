// Per-head QK-RMSNorm + RoPE of ONE head row by a quarter wave (16 lanes), shared by the stand-alone
// split_qkv_norm_rope_into_paged_cache kernel and the fused decode-attention prologue so that both produce the
// same fp16 bits.  Maths: ferrum-kernels/src/backend/cpu.rs:1645-1783 (split_qkv + qk_norm_rope); modes as in
// traits.rs:1764 — 0 copy, 1 norm + half-split RoPE, 2 half-split RoPE, 3 interleaved RoPE.
//
// Lane q16 of the quarter wave owns PPL = HD/32 rotation pairs i = q16·PPL + k.  Its 2·PPL inputs are fetched as
// two contiguous PPL-element vectors (half-split: dims [i..] and [HD/2 + i..]; interleaved: dims [2i.., 2i + PPL..]),
// every load of the row is issued before the first use (one memory round trip), and contraction is off so the
// rotation rounds like the reference's separate multiplies and adds.
#pragma once
#include "common.h"

namespace fh {

template <int HD>
struct RopeRow {
    static constexpr int PPL = HD / 32;
    using hv = _Float16 __attribute__((ext_vector_type(PPL)));
    using fv = float __attribute__((ext_vector_type(PPL)));
    hv out0, out1;     // results, to be stored at element offsets off0 / off1 of the destination row
    int off0, off1;
};

// SC1: the source row was written by another workgroup of THIS launch (write-through stores behind a counter): it is read with
// L1-bypassing loads (relaxed agent-scope 8-byte loads; head_dim 128 only).
template <int HD, bool SC1 = false>
__device__ __forceinline__ RopeRow<HD> rope_row16(const __half* src, const __half* nw, const float* cs, const float* sn,
                                                  int mode, bool have_nw, bool have_rope, float eps, int q16) {
#pragma clang fp contract(off)
    using R = RopeRow<HD>;
    constexpr int PPL = R::PPL, HALF = HD / 2;
    const int base = q16 * PPL;
    const bool m3 = mode == 3;
    R r;
    r.off0 = m3 ? 2 * base : base;
    r.off1 = m3 ? 2 * base + PPL : base + HALF;
    typename R::hv c0, c1;
    if constexpr (SC1) {
        static_assert(!SC1 || PPL == 4, "the in-launch hand-off form reads 8-byte pieces");
        const unsigned long long u0 = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(src + r.off0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long u1 = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(src + r.off1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c0 = __builtin_bit_cast(typename R::hv, u0);
        c1 = __builtin_bit_cast(typename R::hv, u1);
    } else {
        c0 = *reinterpret_cast<const typename R::hv*>(src + r.off0);
        c1 = *reinterpret_cast<const typename R::hv*>(src + r.off1);
    }
    typename R::hv w0, w1;
    typename R::fv cv, sv;
#pragma unroll
    for (int k = 0; k < PPL; k++) { w0[k] = (_Float16)1.f; w1[k] = (_Float16)1.f; cv[k] = 1.f; sv[k] = 0.f; }
    if (have_nw) {
        w0 = *reinterpret_cast<const typename R::hv*>(nw + r.off0);
        w1 = *reinterpret_cast<const typename R::hv*>(nw + r.off1);
    }
    if (have_rope) {
        cv = *reinterpret_cast<const typename R::fv*>(cs + base);
        sv = *reinterpret_cast<const typename R::fv*>(sn + base);
    }
    float c[2 * PPL], w[2 * PPL];
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        c[k] = (float)c0[k]; c[PPL + k] = (float)c1[k];
        w[k] = (float)w0[k]; w[PPL + k] = (float)w1[k];
    }
    float scale = 1.0f;
    {
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < 2 * PPL; k++) ss += c[k] * c[k];
        // sum over the 16 lanes of the row by DPP (no LDS round trips)
        ss += dpp_move<0xB1>(ss);
        ss += dpp_move<0x4E>(ss);
        ss += dpp_move<0x141>(ss);
        ss += dpp_move<0x140>(ss);
        if (mode == 1) scale = 1.0f / sqrtf(ss / (float)HD + eps);
    }
    float r0[PPL], r1[PPL];
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        float x0 = m3 ? c[2 * k] : c[k], x1 = m3 ? c[2 * k + 1] : c[PPL + k];
        if (mode == 1) {
            x0 = x0 * scale * (m3 ? w[2 * k] : w[k]);
            x1 = x1 * scale * (m3 ? w[2 * k + 1] : w[PPL + k]);
        }
        r0[k] = x0;
        r1[k] = x1;
        if (mode != 0) {
            r0[k] = x0 * cv[k] - x1 * sv[k];
            r1[k] = x1 * cv[k] + x0 * sv[k];
        }
    }
    float o[2 * PPL];          // back to the load order: compile-time slots, per-lane select on the pairing
#pragma unroll
    for (int j = 0; j < 2 * PPL; j++) {
        const float inter = (j & 1) ? r1[j >> 1] : r0[j >> 1];
        const float split = j < PPL ? r0[j % PPL] : r1[j % PPL];
        o[j] = m3 ? inter : split;
    }
#pragma unroll
    for (int k = 0; k < PPL; k++) { r.out0[k] = (_Float16)o[k]; r.out1[k] = (_Float16)o[PPL + k]; }
    return r;
}

}  // namespace fh
