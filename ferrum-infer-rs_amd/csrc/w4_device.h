// Device-side pieces of the INT4 (w4t) GEMMs shared by the kernel translation units (w4_gemm.hip, chain.hip):
// in-place nibble expansion and the one-quant-group MFMA step.  Layout and arithmetic: see w4_gemm.hip's header comment.
#pragma once
#include "common.h"

namespace fh {

__device__ __forceinline__ half2v u32_as_half2(uint32_t u) {
    union { uint32_t u; half2v h; } c;
    c.u = u;
    return c.h;
}

// (w & mask) | magic in ONE VALU op (v_and_or_b32).  With literal constants hipcc emits v_and_b32 +
// v_or_b32 (VOP3 has no literal operands on gfx9-family encodings), so the constants are made opaque
// register values (mask in an SGPR, magic in a VGPR) and the plain C expression then selects
// v_and_or_b32.  No inline-asm instruction is involved, so the compiler still tracks the
// VALU-write → MFMA-read hazard itself (an asm v_and_or feeding an MFMA directly read stale operands).
__device__ __forceinline__ uint32_t opaque_sgpr(uint32_t v) {
    v = __builtin_amdgcn_readfirstlane(v);
    asm volatile("" : "+s"(v));
    return v;
}
__device__ __forceinline__ uint32_t opaque_vgpr(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ uint32_t and_or(uint32_t w, uint32_t mask, uint32_t magic) { return (w & mask) | magic; }

__device__ __forceinline__ half8 pack_half8(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3) {
    union { uint32_t u[4]; half8 h; } c;
    c.u[0] = r0; c.u[1] = r1; c.u[2] = r2; c.u[3] = r3;
    return c.h;
}
__device__ __forceinline__ half8 splat_half8(float v) {
    _Float16 h = (_Float16)v;
    return (half8){h, h, h, h, h, h, h, h};
}

// One quant group (128 k) of NT column tiles × MT row tiles.
//   wq[nt]  : the lane's 4 packed dwords of tile nt          af[mt][s] : activation fragments, k-steps 0..3
//   acc     : fp32 accumulators                              sbits/zbits: 4 packed fp16 scales / zeros
// Nibble expansion costs 5 VALU per dword (1 shift + 4 v_and_or): a lo nibble masked in place under 0x6400 is the
// exact fp16 1024+q, a hi nibble (mantissa bits 4-7) under 0x5400 is the exact fp16 64+q, so
//   Σ_lo (1024+q)x + Σ_hi (64+q)x = Σ q·x + 1024·S_lo + 64·S_hi.  The offset (and the
// zero point: −zero·(S_lo+S_hi)) is removed with NO per-weight work: four extra MFMAs per group against a
// constant B operand produce −(1024+zero)·S_lo − (64+zero)·S_hi per token row, and that is used as the
// starting accumulator of every tile's MFMA chain.  The group scale multiplies the fp32 chain result.
template <int MT, int NT, bool HAS_ZP, bool ILV = true, typename WQ>
__device__ __forceinline__ void w4_consume_group(const WQ (&wq)[NT], unsigned long long sbits,
                                                 unsigned long long zbits, int nt0, half8 (&af)[MT][4],
                                                 float4v (&acc)[MT][NT]) {
    const uint32_t magic = opaque_vgpr(0x64006400u), magic_hi = opaque_vgpr(0x54005400u);
    const uint32_t m_lo = opaque_sgpr(0x000F000Fu), m_hi = opaque_sgpr(0x00F000F0u);
    auto half_at = [](unsigned long long bits, int i) {
        union { uint16_t u; _Float16 h; } c;
        c.u = (uint16_t)(bits >> (16 * i));
        return c.h;
    };
    float4v neg_off[MT], s_sum[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        // symmetric: −1032·S_lo − 72·S_hi;  asymmetric: −1024·S_lo − 64·S_hi
        const half8 b_lo = splat_half8(HAS_ZP ? -1024.0f : -1032.0f), b_hi = splat_half8(HAS_ZP ? -64.0f : -72.0f);
        float4v t = {0.f, 0.f, 0.f, 0.f};
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][0], b_lo, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][1], b_hi, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][2], b_lo, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][3], b_hi, t, 0, 0, 0);
        neg_off[mt] = t;
        if (HAS_ZP) {   // Σ x over the group
            const half8 o_lo = splat_half8(1.0f), o_hi = o_lo;
            float4v u = {0.f, 0.f, 0.f, 0.f};
            u = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][0], o_lo, u, 0, 0, 0);
            u = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][1], o_hi, u, 0, 0, 0);
            u = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][2], o_lo, u, 0, 0, 0);
            u = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][3], o_hi, u, 0, 0, 0);
            s_sum[mt] = u;
        }
    }
    // All NT×MT accumulator chains advance together (k-step by k-step) so that consecutive MFMAs are independent:
    // with one tile at a time the chain tmp ← mfma(·,·,tmp) serialises on the MFMA latency whenever a SIMD holds a
    // single wave (the dense LDS kernel) — same operations per chain, same order, same bits.
    // (ILV = false: one tile at a time — fewer live registers, for the 1024-thread intra-workgroup split kernel.)
    if constexpr (!ILV) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            float4v t1[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) t1[mt] = neg_off[mt];
#pragma unroll
            for (int pr = 0; pr < 2; pr++) {
                const uint32_t d0 = wq[nt][2 * pr], d1 = wq[nt][2 * pr + 1];
                const uint32_t d0s = d0 >> 8, d1s = d1 >> 8;
                const half8 lo = pack_half8(and_or(d0, m_lo, magic), and_or(d0s, m_lo, magic), and_or(d1, m_lo, magic),
                                            and_or(d1s, m_lo, magic));
                const half8 hi = pack_half8(and_or(d0, m_hi, magic_hi), and_or(d0s, m_hi, magic_hi), and_or(d1, m_hi, magic_hi),
                                            and_or(d1s, m_hi, magic_hi));
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    t1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][2 * pr], lo, t1[mt], 0, 0, 0);
                    t1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][2 * pr + 1], hi, t1[mt], 0, 0, 0);
                }
            }
            const float s_f = (float)half_at(sbits, nt0 + nt);
            const float z_f = HAS_ZP ? (float)half_at(zbits, nt0 + nt) : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = HAS_ZP ? __builtin_fmaf(-z_f, s_sum[mt][r], t1[mt][r]) : t1[mt][r];
                    acc[mt][nt][r] = __builtin_fmaf(s_f, v, acc[mt][nt][r]);      // (explicit: left to contraction, one inlining context fused it and another did not)
                }
        }
        return;
    }
    float4v tmp[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) tmp[mt][nt] = neg_off[mt];
#pragma unroll
    for (int pr = 0; pr < 2; pr++) {
        half8 lo[NT], hi[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const uint32_t d0 = wq[nt][2 * pr], d1 = wq[nt][2 * pr + 1];
            const uint32_t d0s = d0 >> 8, d1s = d1 >> 8;
            lo[nt] = pack_half8(and_or(d0, m_lo, magic), and_or(d0s, m_lo, magic), and_or(d1, m_lo, magic), and_or(d1s, m_lo, magic));
            hi[nt] = pack_half8(and_or(d0, m_hi, magic_hi), and_or(d0s, m_hi, magic_hi), and_or(d1, m_hi, magic_hi), and_or(d1s, m_hi, magic_hi));
        }
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) tmp[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][2 * pr], lo[nt], tmp[mt][nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) tmp[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][2 * pr + 1], hi[nt], tmp[mt][nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const float s_f = (float)half_at(sbits, nt0 + nt);
        const float z_f = HAS_ZP ? (float)half_at(zbits, nt0 + nt) : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float v = HAS_ZP ? __builtin_fmaf(-z_f, s_sum[mt][r], tmp[mt][nt][r]) : tmp[mt][nt][r];
                acc[mt][nt][r] = __builtin_fmaf(s_f, v, acc[mt][nt][r]);
            }
    }
}

}  // namespace fh
