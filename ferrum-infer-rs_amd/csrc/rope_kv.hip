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

namespace fh {

// One wave per (token, head).  hd ≤ 256.
__global__ __launch_bounds__(64) void split_qkv_norm_rope_paged_kernel(
    const __half* __restrict__ qkv, const __half* __restrict__ q_norm_w, const __half* __restrict__ k_norm_w,
    const float* __restrict__ cos_t, const float* __restrict__ sin_t, __half* __restrict__ q_out,
    __half* __restrict__ cache_k, __half* __restrict__ cache_v, const uint32_t* __restrict__ cu_seqlens_q,
    const uint32_t* __restrict__ pos_offsets, const int32_t* __restrict__ block_tables, int num_seqs,
    int q_heads, int kv_heads, int hd, float eps, int qk_mode, int max_blocks_per_seq) {
    const int tok = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
    const int half_d = hd >> 1;
    const int q_dim = q_heads * hd, kv_dim = kv_heads * hd;
    const __half* row = qkv + (long)tok * (q_dim + 2 * kv_dim);

    // locate the sequence (cu_seqlens_q is a short prefix sum; same scan as the reference kernel)
    int seq = 0;
    while (seq + 1 < num_seqs && (uint32_t)tok >= cu_seqlens_q[seq + 1]) seq++;
    const int pos = (int)pos_offsets[seq] + (tok - (int)cu_seqlens_q[seq]);

    const bool is_q = head < q_heads;
    const bool is_k = !is_q && head < q_heads + kv_heads;
    int local_head, mode;
    const __half* src;
    const __half* nw = nullptr;
    if (is_q) { local_head = head; src = row + local_head * hd; mode = qk_mode; nw = q_norm_w; }
    else if (is_k) { local_head = head - q_heads; src = row + q_dim + local_head * hd; mode = qk_mode; nw = k_norm_w; }
    else { local_head = head - q_heads - kv_heads; src = row + q_dim + kv_dim + local_head * hd; mode = 0; }

    __half* dst_q = nullptr;
    __half* tile = nullptr;
    int slot = 0;
    if (is_q) {
        dst_q = q_out + ((long)tok * q_heads + local_head) * hd;
    } else {
        const int logical = pos / KV_BLOCK;
        slot = pos % KV_BLOCK;
        const long physical = block_tables[(long)seq * max_blocks_per_seq + logical];
        tile = (is_k ? cache_k : cache_v) + (physical * kv_heads + local_head) * kv_tile_elems(hd);
    }
    auto store = [&](int d, float v) {
        __half h = __float2half(v);
        if (is_q) dst_q[d] = h;
        else if (is_k) tile[k_tile_off(slot, d)] = h;
        else tile[v_tile_off(slot, d)] = h;
    };

    if (mode == 0) {
        for (int i = lane; i < hd; i += 64) store(i, __half2float(src[i]));
        return;
    }
    float scale = 1.0f;
    if (mode == 1) {
        float ss = 0.f;
        for (int i = lane; i < hd; i += 64) { float x = __half2float(src[i]); ss += x * x; }
        ss = wave_reduce_sum(ss);
        scale = 1.0f / sqrtf(ss / (float)hd + eps);
    }
    const float* cs = cos_t + (long)pos * half_d;
    const float* sn = sin_t + (long)pos * half_d;
    for (int i = lane; i < half_d; i += 64) {
        int i0 = mode == 3 ? 2 * i : i, i1 = mode == 3 ? 2 * i + 1 : i + half_d;
        float x0 = __half2float(src[i0]), x1 = __half2float(src[i1]);
        if (mode == 1) {
            x0 = x0 * scale * __half2float(nw[i0]);
            x1 = x1 * scale * __half2float(nw[i1]);
        }
        float c = cs[i], s = sn[i];
        store(i0, x0 * c - x1 * s);
        store(i1, x1 * c + x0 * s);
    }
}

int split_qkv_norm_rope_into_paged_cache_varlen_f16(
    const __half* qkv, const __half* q_norm_w, const __half* k_norm_w, const float* cos_t, const float* sin_t,
    __half* q_out, __half* cache_k, __half* cache_v, const uint32_t* cu_seqlens_q, const uint32_t* pos_offsets,
    const int32_t* block_tables, int num_seqs, int m_total, int q_heads, int kv_heads, int head_dim, float eps,
    int qk_mode, int block_size, int max_blocks_per_seq, hipStream_t s) {
    if (m_total <= 0) return 0;
    FH_REQUIRE(block_size == KV_BLOCK, "paged KV: block_size=%d unsupported (native layout uses 16)", block_size);
    FH_REQUIRE(head_dim % 32 == 0 && head_dim <= 256, "paged KV: head_dim=%d must be a multiple of 32, <= 256", head_dim);
    FH_REQUIRE(qk_mode >= 0 && qk_mode <= 3, "paged KV: qk_mode=%d out of range", qk_mode);
    hipLaunchKernelGGL(split_qkv_norm_rope_paged_kernel, dim3(m_total, q_heads + 2 * kv_heads), dim3(64), 0, s, qkv,
                       q_norm_w, k_norm_w, cos_t, sin_t, q_out, cache_k, cache_v, cu_seqlens_q, pos_offsets,
                       block_tables, num_seqs, q_heads, kv_heads, head_dim, eps, qk_mode, max_blocks_per_seq);
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
