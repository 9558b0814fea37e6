// Fused split-QKV + per-head QK-RMSNorm + RoPE + paged-KV write (varlen), and the KV read-back.
//
// Reference: `BackendPagedKv::split_qkv_norm_rope_into_paged_cache_varlen`
// (ferrum-kernels/src/backend/traits.rs:1764; CUDA kernels/split_qkv_norm_rope_into_paged_cache.cu:170;
// CPU maths split_qkv + qk_norm_rope, backend/cpu.rs:1645-1783).  qk_mode: 0 copy, 1 norm + half-split
// RoPE, 2 half-split RoPE, 3 interleaved RoPE; V is always copied; pos = pos_offsets[seq] + local index;
// K/V land in pool block block_tables[seq][pos/16], slot pos%16; Q goes out token-major.
#include "common.h"
#include "kernels.h"
#include "kv_layout.h"
#include "rope_rows.h"

namespace fh {

// One quarter wave per (token, head): a 64-thread workgroup covers 4 consecutive heads of one token
// (q heads, then k heads, then v heads).  Row maths in rope_rows.h.
template <int HD>
__global__ __launch_bounds__(64) void split_qkv_norm_rope_paged_kernel(
    const __half* __restrict__ qkv, const __half* __restrict__ q_norm_w, const __half* __restrict__ k_norm_w,
    const float* __restrict__ cos_t, const float* __restrict__ sin_t, __half* __restrict__ q_out,
    __half* __restrict__ cache_k, __half* __restrict__ cache_v, const uint32_t* __restrict__ cu_seqlens_q,
    const uint32_t* __restrict__ pos_offsets, const int32_t* __restrict__ block_tables, int num_seqs,
    int q_heads, int kv_heads, float eps, int qk_mode, int max_blocks_per_seq) {
    constexpr int PPL = HD / 32, HALF = HD / 2;
    const int tok = blockIdx.x, lane = threadIdx.x, q16 = lane & 15;
    const int heads = q_heads + 2 * kv_heads;
    const int head_raw = blockIdx.y * 4 + (lane >> 4);
    const bool act = head_raw < heads;
    const int head = act ? head_raw : heads - 1;          // clamped: loads stay unconditional
    const int q_dim = q_heads * HD, kv_dim = kv_heads * HD;
    const __half* row = qkv + (long)tok * (q_dim + 2 * kv_dim);

    // locate the sequence: the last s with cu_seqlens_q[s] ≤ tok (what the reference kernel's linear scan returns), found
    // 64 sequences per memory round trip — a serial scan costs the tokens of the 32nd prompt 31 dependent loads
    int seq = 0;
    for (int s0 = 0; s0 < num_seqs; s0 += 64) {
        const int s = s0 + lane;
        const unsigned long long le = __ballot(s < num_seqs && cu_seqlens_q[s] <= (uint32_t)tok);
        if (le == 0) break;
        seq = s0 + 63 - __clzll((long long)le);
        if (le != ~0ull) break;
    }
    const int pos = (int)pos_offsets[seq] + (tok - (int)cu_seqlens_q[seq]);

    const bool is_q = head < q_heads;
    const bool is_k = !is_q && head < q_heads + kv_heads;
    const int local_head = is_q ? head : is_k ? head - q_heads : head - q_heads - kv_heads;
    const __half* src = row + (is_q ? 0 : is_k ? q_dim : q_dim + kv_dim) + local_head * HD;
    const int mode = (is_q || is_k) ? qk_mode : 0;
    const RopeRow<HD> rr = rope_row16<HD>(src, is_q ? q_norm_w : k_norm_w, cos_t + (long)pos * HALF, sin_t + (long)pos * HALF,
                                          mode, qk_mode == 1, qk_mode != 0, eps, q16);
    if (!act) return;
    using hv = typename RopeRow<HD>::hv;
    if (is_q) {
        __half* dst = q_out + ((long)tok * q_heads + local_head) * HD;
        *reinterpret_cast<hv*>(dst + rr.off0) = rr.out0;
        *reinterpret_cast<hv*>(dst + rr.off1) = rr.out1;
        return;
    }
    const int logical = pos / KV_BLOCK, slot = pos % KV_BLOCK;
    const long physical = block_tables[(long)seq * max_blocks_per_seq + logical];
    __half* tile = (is_k ? cache_k : cache_v) + (physical * kv_heads + local_head) * kv_tile_elems(HD);
    _Float16* t16 = reinterpret_cast<_Float16*>(tile);
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        if (is_k) {
            t16[k_tile_off(slot, rr.off0 + k)] = rr.out0[k];
            t16[k_tile_off(slot, rr.off1 + k)] = rr.out1[k];
        } else {
            t16[v_tile_off(slot, rr.off0 + k)] = rr.out0[k];
            t16[v_tile_off(slot, rr.off1 + k)] = rr.out1[k];
        }
    }
}

int split_qkv_norm_rope_into_paged_cache_varlen_f16(
    const __half* qkv, const __half* q_norm_w, const __half* k_norm_w, const float* cos_t, const float* sin_t,
    __half* q_out, __half* cache_k, __half* cache_v, const uint32_t* cu_seqlens_q, const uint32_t* pos_offsets,
    const int32_t* block_tables, int num_seqs, int m_total, int q_heads, int kv_heads, int head_dim, float eps,
    int qk_mode, int block_size, int max_blocks_per_seq, hipStream_t s) {
    if (m_total <= 0) return 0;
    FH_REQUIRE(block_size == KV_BLOCK, "paged KV: block_size=%d unsupported (native layout uses 16)", block_size);
    FH_REQUIRE(head_dim == 64 || head_dim == 128 || head_dim == 256, "paged KV: head_dim=%d must be 64, 128 or 256", head_dim);
    FH_REQUIRE(qk_mode >= 0 && qk_mode <= 3, "paged KV: qk_mode=%d out of range", qk_mode);
    const dim3 grid(m_total, cdiv(q_heads + 2 * kv_heads, 4));
#define FH_ROPE(HDV)                                                                                                   \
    hipLaunchKernelGGL(split_qkv_norm_rope_paged_kernel<HDV>, grid, dim3(64), 0, s, qkv, q_norm_w, k_norm_w, cos_t, sin_t, \
                       q_out, cache_k, cache_v, cu_seqlens_q, pos_offsets, block_tables, num_seqs, q_heads, kv_heads, eps, \
                       qk_mode, max_blocks_per_seq)
    if (head_dim == 64) FH_ROPE(64); else if (head_dim == 128) FH_ROPE(128); else FH_ROPE(256);
#undef FH_ROPE
    FH_CHECK_LAUNCH();
    return 0;
}

// Gather one sequence's K/V back to token-major [kv_len, kv_heads, hd] (the order ferrum-kv's
// read_kv returns, ferrum-kv/src/managers/paged.rs:528-561).  Used by parity tests and by
// prefix-cache style consumers; not on the decode hot path.
__global__ void paged_kv_read_kernel(const __half* __restrict__ cache_k, const __half* __restrict__ cache_v,
                                     const int32_t* __restrict__ block_table, int kv_heads, int hd,
                                     __half* __restrict__ k_out, __half* __restrict__ v_out) {
    const int pos = blockIdx.x, head = blockIdx.y;
    const long physical = block_table[pos / KV_BLOCK];
    const int slot = pos % KV_BLOCK;
    const long tile = (physical * kv_heads + head) * kv_tile_elems(hd);
    for (int d = threadIdx.x; d < hd; d += blockDim.x) {
        long o = ((long)pos * kv_heads + head) * hd + d;
        k_out[o] = cache_k[tile + k_tile_off(slot, d)];
        v_out[o] = cache_v[tile + v_tile_off(slot, d)];
    }
}

int paged_kv_read_f16(const __half* cache_k, const __half* cache_v, const int32_t* block_table, int kv_len,
                      int kv_heads, int head_dim, int block_size, __half* k_out, __half* v_out, hipStream_t s) {
    if (kv_len <= 0) return 0;
    FH_REQUIRE(block_size == KV_BLOCK, "paged KV: block_size=%d unsupported (native layout uses 16)", block_size);
    hipLaunchKernelGGL(paged_kv_read_kernel, dim3(kv_len, kv_heads), dim3(64), 0, s, cache_k, cache_v, block_table,
                       kv_heads, head_dim, k_out, v_out);
    FH_CHECK_LAUNCH();
    return 0;
}

}  // namespace fh
