// The contiguous-KV lane of the Backend trait: the ops the reference's non-paged path composes
// (ferrum-kernels/src/backend/kv_layer.rs:370-513): split_qkv → qk_norm_rope (token-major → head-major) →
// kv_cache_append_head_major → flash_attention over the head-major cache → transpose_head_to_token, plus copy_slice /
// scaled_add_inplace / transpose_token_to_head.  Signatures: traits.rs:225 (flash_attention), :798 (copy_slice),
// :850 (split_qkv), :897 (qk_norm_rope), :1266 (kv_cache_append_head_major), :1281,1296 (transposes), :1318
// (scaled_add_inplace); maths cpu.rs:1645-1783, 1993-2060, 2179-2259.
// The paged lane (rope_kv.hip / attention.hip) is the fast path; these exist so that a backend built on this library
// implements the WHOLE core trait, and they are simple, bandwidth-shaped kernels rather than tuned ones.
#include "common.h"
#include "kernels.h"
#include "rope_rows.h"

namespace fh {

__global__ void split_qkv_kernel(const __half* __restrict__ qkv, __half* __restrict__ q, __half* __restrict__ k,
                                 __half* __restrict__ v, int q_dim, int kv_dim) {
    const long t = blockIdx.y;
    const int row = q_dim + 2 * kv_dim;
    for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * 8; i < row; i += gridDim.x * blockDim.x * 8) {
        const half8 x = *reinterpret_cast<const half8*>(qkv + t * row + i);
        if (i < q_dim) *reinterpret_cast<half8*>(q + t * q_dim + i) = x;
        else if (i < q_dim + kv_dim) *reinterpret_cast<half8*>(k + t * kv_dim + (i - q_dim)) = x;
        else *reinterpret_cast<half8*>(v + t * kv_dim + (i - q_dim - kv_dim)) = x;
    }
}

int split_qkv_f16(const __half* qkv, __half* q, __half* k, __half* v, int tokens, int q_dim, int kv_dim, hipStream_t s) {
    if (tokens <= 0) return 0;
    FH_REQUIRE(q_dim % 8 == 0 && kv_dim % 8 == 0, "split_qkv: q_dim=%d kv_dim=%d must be multiples of 8", q_dim, kv_dim);
    const int vecs = (q_dim + 2 * kv_dim) / 8;
    hipLaunchKernelGGL(split_qkv_kernel, dim3(cdiv(vecs, 256), tokens), dim3(256), 0, s, qkv, q, k, v, q_dim, kv_dim);
    FH_CHECK_LAUNCH();
    return 0;
}

// one quarter wave per (token, head); output head-major [heads, tokens, hd]
template <int HD>
__global__ __launch_bounds__(64) void qk_norm_rope_kernel(const __half* __restrict__ in, const __half* __restrict__ norm_w,
                                                          const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                          __half* __restrict__ out, int tokens, int heads, int pos_offset,
                                                          float eps, int mode) {
    const int tok = blockIdx.x, lane = threadIdx.x, q16 = lane & 15;
    const int head_raw = blockIdx.y * 4 + (lane >> 4);
    const bool act = head_raw < heads;
    const int head = act ? head_raw : heads - 1;
    const __half* src = in + ((long)tok * heads + head) * HD;
    const long pos = pos_offset + tok;
    const RopeRow<HD> rr = rope_row16<HD>(src, norm_w, cos_t + pos * (HD / 2), sin_t + pos * (HD / 2), mode, mode == 1, mode != 0,
                                          eps, q16);
    if (!act) return;
    using hv = typename RopeRow<HD>::hv;
    __half* dst = out + ((long)head * tokens + tok) * HD;
    *reinterpret_cast<hv*>(dst + rr.off0) = rr.out0;
    *reinterpret_cast<hv*>(dst + rr.off1) = rr.out1;
}

int qk_norm_rope_f16(const __half* input, const __half* norm_w, const float* cos_t, const float* sin_t, __half* output,
                     int tokens, int heads, int head_dim, int pos_offset, float eps, int mode, hipStream_t s) {
    if (tokens <= 0 || heads <= 0) return 0;
    FH_REQUIRE(head_dim == 64 || head_dim == 128 || head_dim == 256, "qk_norm_rope: head_dim=%d must be 64, 128 or 256", head_dim);
    FH_REQUIRE(mode >= 0 && mode <= 3, "qk_norm_rope: mode=%d out of range", mode);
    const dim3 grid(tokens, cdiv(heads, 4));
#define FH_QNR(HDV) hipLaunchKernelGGL(qk_norm_rope_kernel<HDV>, grid, dim3(64), 0, s, input, norm_w, cos_t, sin_t, output, tokens, \
                                       heads, pos_offset, eps, mode)
    if (head_dim == 64) FH_QNR(64); else if (head_dim == 128) FH_QNR(128); else FH_QNR(256);
#undef FH_QNR
    FH_CHECK_LAUNCH();
    return 0;
}

// cache [nkv, capacity, hd] ← new [nkv, new_tokens, hd] at slot cache_len
__global__ void kv_append_hm_kernel(__half* __restrict__ cache_k, __half* __restrict__ cache_v, const __half* __restrict__ nk,
                                    const __half* __restrict__ nv, int cache_len, int capacity, int new_tokens, int hd) {
    const int h = blockIdx.z, t = blockIdx.y;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= hd) return;
    const long src = ((long)h * new_tokens + t) * hd + i, dst = ((long)h * capacity + cache_len + t) * hd + i;
    *reinterpret_cast<half8*>(cache_k + dst) = *reinterpret_cast<const half8*>(nk + src);
    *reinterpret_cast<half8*>(cache_v + dst) = *reinterpret_cast<const half8*>(nv + src);
}

int kv_cache_append_head_major_f16(__half* cache_k, __half* cache_v, int cache_len, int cache_capacity, const __half* new_k,
                                   const __half* new_v, int new_tokens, int nkv, int hd, hipStream_t s) {
    if (new_tokens <= 0) return 0;
    FH_REQUIRE(hd % 8 == 0 && cache_len >= 0 && cache_len + new_tokens <= cache_capacity,
               "kv_cache_append: len %d + %d exceeds capacity %d (or hd %d not a multiple of 8)", cache_len, new_tokens, cache_capacity, hd);
    hipLaunchKernelGGL(kv_append_hm_kernel, dim3(cdiv(hd / 8, 64), new_tokens, nkv), dim3(64), 0, s, cache_k, cache_v, new_k, new_v,
                       cache_len, cache_capacity, new_tokens, hd);
    FH_CHECK_LAUNCH();
    return 0;
}

// [A, B, dim] → [B, A, dim]
__global__ void transpose_ab_kernel(const __half* __restrict__ src, __half* __restrict__ dst, int A, int Bn, int dim) {
    const int a = blockIdx.z, b = blockIdx.y;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= dim) return;
    *reinterpret_cast<half8*>(dst + ((long)b * A + a) * dim + i) = *reinterpret_cast<const half8*>(src + ((long)a * Bn + b) * dim + i);
}

int transpose_head_to_token_f16(const __half* src, __half* dst, int tokens, int heads, int dim, hipStream_t s) {
    if (tokens <= 0 || heads <= 0) return 0;
    FH_REQUIRE(dim % 8 == 0, "transpose: dim=%d must be a multiple of 8", dim);
    hipLaunchKernelGGL(transpose_ab_kernel, dim3(cdiv(dim / 8, 64), tokens, heads), dim3(64), 0, s, src, dst, heads, tokens, dim);
    FH_CHECK_LAUNCH();
    return 0;
}
int transpose_token_to_head_f16(const __half* src, __half* dst, int tokens, int heads, int dim, hipStream_t s) {
    if (tokens <= 0 || heads <= 0) return 0;
    FH_REQUIRE(dim % 8 == 0, "transpose: dim=%d must be a multiple of 8", dim);
    hipLaunchKernelGGL(transpose_ab_kernel, dim3(cdiv(dim / 8, 64), heads, tokens), dim3(64), 0, s, src, dst, tokens, heads, dim);
    FH_CHECK_LAUNCH();
    return 0;
}

__global__ void copy_slice_kernel(const __half* __restrict__ src, __half* __restrict__ dst, long len) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) dst[i] = src[i];
}
int copy_slice_f16(const __half* src, long src_offset, __half* dst, long dst_offset, long len, hipStream_t s) {
    if (len <= 0) return 0;
    hipLaunchKernelGGL(copy_slice_kernel, dim3(cdiv(len, 256)), dim3(256), 0, s, src + src_offset, dst + dst_offset, len);
    FH_CHECK_LAUNCH();
    return 0;
}

__global__ void scaled_add_kernel(__half* __restrict__ dst, const __half* __restrict__ src, float scale, long len) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) {
#pragma clang fp contract(off)                         // the reference's f32 loop rounds the product before the add
        const float prod = scale * __half2float(src[i]);
        dst[i] = __float2half(__half2float(dst[i]) + prod);
    }
}
int scaled_add_inplace_f16(__half* dst, const __half* src, float scale, long len, hipStream_t s) {
    if (len <= 0) return 0;
    hipLaunchKernelGGL(scaled_add_kernel, dim3(cdiv(len, 256)), dim3(256), 0, s, dst, src, scale, len);
    FH_CHECK_LAUNCH();
    return 0;
}

// flash_attention over a contiguous head-major cache (cpu.rs:2179-2259): q/out [nq, q_len, hd], k/v [nkv, kv_stride, hd].
// One 256-thread workgroup per (head, query token): scores of the attended range into LDS (key-parallel), block max and
// Σexp, then out[d] = Σ p·v[d] with one dimension per thread (coalesced V rows).  kv ranges up to 16 000 keys.
__global__ __launch_bounds__(256) void contig_attention_kernel(const __half* __restrict__ q, const __half* __restrict__ k,
                                                               const __half* __restrict__ v, __half* __restrict__ out, int q_len,
                                                               int kv_len, int causal, int pos_offset, int nq, int nkv, int hd,
                                                               float scale, long kv_stride, int window) {
    extern __shared__ float sc[];                      // [kv range] scores, then probabilities
    __shared__ float red[4];
    __shared__ __attribute__((aligned(16))) __half qs[256];
    const int h = blockIdx.y, qi = blockIdx.x;
    const int kvh = h / (nq / nkv);
    int end = kv_len;
    if (causal) end = min(pos_offset + qi + 1, kv_len);
    int start = 0;
    if (causal && window > 0) start = end > window ? end - window : 0;
    const int n = end - start;
    const __half* qrow = q + ((long)h * q_len + qi) * hd;
    __half* orow = out + ((long)h * q_len + qi) * hd;
    if (n <= 0) return;                                // nothing visible: the reference leaves `out` untouched
    for (int d = threadIdx.x; d < hd; d += 256) qs[d] = qrow[d];
    __syncthreads();
    const __half* kb = k + (long)kvh * kv_stride * hd;
    const __half* vb = v + (long)kvh * kv_stride * hd;
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < n; j += 256) {
        const __half* kr = kb + (long)(start + j) * hd;
        float dot = 0.f;
        for (int d = 0; d < hd; d += 8) {
            const half8 kk = *reinterpret_cast<const half8*>(kr + d);
            const half8 qq = *reinterpret_cast<const half8*>(qs + d);
#pragma unroll
            for (int e = 0; e < 8; e++) dot += (float)qq[e] * (float)kk[e];
        }
        const float sv = dot * scale;
        sc[j] = sv;
        mx = fmaxf(mx, sv);
    }
    auto block_reduce = [&](float val, bool is_max) {
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(val, off, 64);
            val = is_max ? fmaxf(val, o) : val + o;
        }
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = val;
        __syncthreads();
        const float r = is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
        return r;
    };
    mx = block_reduce(mx, true);
    float sum = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) {
        const float p = expf(sc[j] - mx);
        sc[j] = p;
        sum += p;
    }
    sum = block_reduce(sum, false);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    for (int d = threadIdx.x; d < hd; d += 256) {
        float acc = 0.f;
        for (int j = 0; j < n; j++) acc += sc[j] * __half2float(vb[(long)(start + j) * hd + d]);
        orow[d] = __float2half(acc * inv);
    }
}

int flash_attention_contig_f16(const __half* q, const __half* k, const __half* v, __half* out, int q_len, int kv_len, int causal,
                               int pos_offset, int num_heads, int num_kv_heads, int head_dim, float scale, int kv_seq_stride,
                               int sliding_window, hipStream_t s) {
    if (q_len <= 0 || kv_len <= 0) return 0;
    FH_REQUIRE(num_kv_heads > 0 && num_heads % num_kv_heads == 0 && head_dim % 8 == 0 && head_dim <= 256,
               "flash_attention: heads %d/%d head_dim %d", num_heads, num_kv_heads, head_dim);
    FH_REQUIRE(kv_len <= 16000, "flash_attention (contiguous lane): kv_len=%d > 16000 (scores live in 64 KB of LDS); use the paged lane", kv_len);
    const long stride = kv_seq_stride > 0 ? kv_seq_stride : kv_len;
    hipLaunchKernelGGL(contig_attention_kernel, dim3(q_len, num_heads), dim3(256), (size_t)kv_len * sizeof(float), s, q, k, v, out,
                       q_len, kv_len, causal, pos_offset, num_heads, num_kv_heads, head_dim, scale, stride, sliding_window);
    FH_CHECK_LAUNCH();
    return 0;
}

}  // namespace fh
