/*
 * ferrum_oracle.c — CPU restatement of the reference's decode hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 * load this library, and only as the checker / reported CPU baseline.  The
 * product path (ferrum-infer-rs_amd/csrc, libferrum_hip.so) never links,
 * imports or falls back to anything in here.
 *
 * Every function restates ONE reference function in plain C (f32 arithmetic,
 * same loop order, same rounding points) and cites the file:line it follows.
 * Paths are relative to /root/reference/crates/.  Build with
 * -ffp-contract=off so no a*b+c is fused (Rust never fuses f32 mul+add).
 *
 * Parity pinning: the reference's Rust cannot be built in this image (no
 * rustc/cargo), so the oracle is pinned by the reference's own known-answer
 * tests re-derived in tests/test_oracle_golden.py (SURVEY.md §8c list):
 * GPTQ LCG self-check, paged-attention hand cases, allocator sequences,
 * router tie-breaks, gelu/scale/bf16 KATs.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define FO_API __attribute__((visibility("default")))

/* Worker threads for the row/head/token loops below.  Every output element is still computed by ONE thread with the
 * reference's own loop order, so results are bit-identical for any thread count; 1 (the default) is the reference's
 * single-threaded CPU path (cpu.rs:483-491) and is what bench.py's cpu_baseline times.  Parity tests at the BASELINE
 * configs' real dimensions raise it so that a 256-token prompt through a full-size layer finishes in seconds. */
static int fo_threads = 1;
FO_API void fo_set_threads(int n) { fo_threads = n < 1 ? 1 : n; }
FO_API int fo_get_threads(void) { return fo_threads; }

/* ─────────────────────────── deterministic inputs ─────────────────────────
 * ferrum-quantization/tests/gptq_parity_test.rs:28-38 (rnd_u32 / rnd_f32). */
FO_API uint32_t fo_lcg_u32(uint64_t *state) {
    *state = *state * 6364136223846793005ULL + 1442695040888963407ULL;
    return (uint32_t)(*state >> 33);
}

FO_API float fo_lcg_f32(uint64_t *state, float lo, float hi) {
    float u = (float)(fo_lcg_u32(state) & 0x00FFFFFFu) / 16777216.0f;
    return lo + u * (hi - lo);
}

/* gptq_parity_test.rs:61-93 make_synthetic; :95-103 symmetric variant.
 * qweight [K/8,N] i32, scales [K/g,N] f32, qzeros [K/g,N/8] i32.
 * Returns the advanced LCG state so callers can chain tensors. */
FO_API uint64_t fo_make_synthetic_gptq(int k, int n, int group, uint64_t seed,
                                       int symmetric, int32_t *qweight,
                                       float *scales, int32_t *qzeros) {
    uint64_t rs = seed;
    int groups = k / group;
    for (long i = 0; i < (long)(k / 8) * n; i++) qweight[i] = (int32_t)fo_lcg_u32(&rs);
    for (long i = 0; i < (long)groups * n; i++) scales[i] = fo_lcg_f32(&rs, 0.01f, 0.1f);
    for (long i = 0; i < (long)groups * (n / 8); i++) {
        uint32_t word = 0;
        for (int bi = 0; bi < 8; bi++) word |= (fo_lcg_u32(&rs) & 0xFu) << (bi * 4);
        qzeros[i] = symmetric ? (int32_t)0x77777777u : (int32_t)word;
    }
    return rs;
}

/* gptq_parity_test.rs:50-59 make_desc_act_g_idx. */
FO_API void fo_make_desc_act_g_idx(int k, int group, int32_t *g_idx) {
    int groups = k / group;
    for (int i = 0; i < k; i++) g_idx[i] = i % groups;
}

/* ─────────────────────────────── dense ops ─────────────────────────────── */

/* ferrum-kernels/src/backend/cpu.rs:2137-2150 dot_product (Linux branch:
 * plain f32 iterator sum, left to right). */
static float fo_dot(const float *a, const float *b, long n) {
    float s = 0.0f;
    for (long i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}

/* cpu.rs:438-493 CpuBackend::gemm, non-macOS branch: out[i,j] =
 * (f32) Σ_p (f64)a[i,p]·(f64)b[j,p];  b is [n,k] row-major. */
FO_API void fo_gemm(const float *a, const float *b, float *out, int m, int n, int k) {
#pragma omp parallel for collapse(2) schedule(static) num_threads(fo_threads) if (fo_threads > 1 && (long)m * n * k >= (1L << 18))
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double sum = 0.0;
            const float *ar = a + (long)i * k, *br = b + (long)j * k;
            for (int p = 0; p < k; p++) sum += (double)ar[p] * (double)br[p];
            out[(long)i * n + j] = (float)sum;
        }
}

/* cpu.rs:495-513 rms_norm. */
FO_API void fo_rms_norm(const float *x, const float *w, float eps, float *out,
                        int tokens, int dim) {
    for (int t = 0; t < tokens; t++) {
        const float *row = x + (long)t * dim;
        float *o = out + (long)t * dim;
        float sum_sq = fo_dot(row, row, dim);
        float inv = 1.0f / sqrtf(sum_sq / (float)dim + eps);
        for (int i = 0; i < dim; i++) o[i] = row[i] * inv * w[i];
    }
}

/* cpu.rs:2081-2113 layer_norm: mean and variance over `dim` in f64, then (x − mean)·inv·γ + β in f32. */
FO_API void fo_layer_norm(const float *x, const float *gamma, const float *beta, float eps, float *out, int tokens, int dim) {
    for (int t = 0; t < tokens; t++) {
        long off = (long)t * dim;
        double mean = 0.0;
        for (int i = 0; i < dim; i++) mean += (double)x[off + i];
        mean /= (double)dim;
        double var = 0.0;
        for (int i = 0; i < dim; i++) { double d = (double)x[off + i] - mean; var += d * d; }
        var /= (double)dim;
        float inv = 1.0f / sqrtf((float)var + eps);
        float mean_f32 = (float)mean;
        for (int i = 0; i < dim; i++) out[off + i] = (x[off + i] - mean_f32) * inv * gamma[i] + beta[i];
    }
}

/* cpu.rs:2263-2273 libm_erf (Abramowitz–Stegun 7.1.26) and cpu.rs:2115-2122 gelu = 0.5·x·(1 + erf(x/√2)). */
static float fo_libm_erf(float x) {
    float sign = x < 0.0f ? -1.0f : 1.0f;
    x = fabsf(x);
    float t = 1.0f / (1.0f + 0.3275911f * x);
    float y = 1.0f - (((((1.0614054f * t - 1.4531521f) * t) + 1.4214138f) * t - 0.28449672f) * t + 0.2548296f) * t * expf(-x * x);
    return sign * y;
}
FO_API void fo_gelu(const float *x, float *out, long len) {
    for (long i = 0; i < len; i++) out[i] = 0.5f * x[i] * (1.0f + fo_libm_erf(x[i] / 1.41421356237309504880f));
}

/* cpu.rs:515-538 fused_add_rms_norm: residual += x; out = rms(residual)·w. */
FO_API void fo_fused_add_rms_norm(float *residual, const float *x, const float *w,
                                  float eps, float *out, int tokens, int dim) {
    for (int t = 0; t < tokens; t++) {
        long off = (long)t * dim;
        for (int i = 0; i < dim; i++) residual[off + i] += x[off + i];
        const float *row = residual + off;
        float sum_sq = fo_dot(row, row, dim);
        float inv = 1.0f / sqrtf(sum_sq / (float)dim + eps);
        for (int i = 0; i < dim; i++) out[off + i] = row[i] * inv * w[i];
    }
}

/* cpu.rs:1632-1643 embedding_lookup. */
FO_API void fo_embedding_lookup(const float *table, const uint32_t *ids, int n_ids,
                                float *out, int dim) {
    for (int i = 0; i < n_ids; i++)
        memcpy(out + (long)i * dim, table + (long)ids[i] * dim, sizeof(float) * dim);
}

/* cpu.rs:1645-1664 split_qkv. */
FO_API void fo_split_qkv(const float *qkv, float *q, float *k, float *v, int tokens,
                         int q_dim, int kv_dim) {
    int qkv_dim = q_dim + 2 * kv_dim;
    for (int t = 0; t < tokens; t++) {
        const float *base = qkv + (long)t * qkv_dim;
        memcpy(q + (long)t * q_dim, base, sizeof(float) * q_dim);
        memcpy(k + (long)t * kv_dim, base + q_dim, sizeof(float) * kv_dim);
        memcpy(v + (long)t * kv_dim, base + q_dim + kv_dim, sizeof(float) * kv_dim);
    }
}

/* cpu.rs:1666-1680 fused_silu_mul_split: [T,2I] (gate cols, then up) → [T,I]. */
FO_API void fo_fused_silu_mul_split(const float *gate_up, float *out, int tokens, int im) {
    for (int t = 0; t < tokens; t++)
        for (int i = 0; i < im; i++) {
            float g = gate_up[(long)t * 2 * im + i];
            float u = gate_up[(long)t * 2 * im + im + i];
            out[(long)t * im + i] = (g / (1.0f + expf(-g))) * u;
        }
}

/* cpu.rs:1682-1698 fused_gelu_tanh_mul_split. */
FO_API void fo_fused_gelu_tanh_mul_split(const float *gate_up, float *out, int tokens, int im) {
    const float SQRT_2_OVER_PI = 0.79788456f;
    for (int t = 0; t < tokens; t++)
        for (int i = 0; i < im; i++) {
            float g = gate_up[(long)t * 2 * im + i];
            float u = gate_up[(long)t * 2 * im + im + i];
            float inner = SQRT_2_OVER_PI * (g + 0.044715f * g * g * g);
            out[(long)t * im + i] = 0.5f * g * (1.0f + tanhf(inner)) * u;
        }
}

/* cpu.rs:1700-1704 scale_inplace. */
FO_API void fo_scale_inplace(float *buf, float scale, long len) {
    for (long i = 0; i < len; i++) buf[i] *= scale;
}

/* cpu.rs:2042-2051 add_inplace. */
FO_API void fo_add_inplace(float *residual, const float *x, long len) {
    for (long i = 0; i < len; i++) residual[i] += x[i];
}

/* cpu.rs:1706-1783 qk_norm_rope.  input [T,heads,hd] token-major → output
 * [heads,T,hd] head-major.  mode 0 transpose, 1 norm+half-split rope,
 * 2 half-split rope, 3 interleaved rope.  pos = pos_offset + t. */
FO_API void fo_qk_norm_rope(const float *input, const float *norm_w, const float *cos_t,
                            const float *sin_t, float *output, int tokens, int heads,
                            int head_dim, int pos_offset, float eps, int mode) {
    int half = head_dim / 2;
    for (int t = 0; t < tokens; t++) {
        long pos = (long)pos_offset + t;
        for (int h = 0; h < heads; h++) {
            const float *src = input + ((long)t * heads + h) * head_dim;
            float *dst = output + ((long)h * tokens + t) * head_dim;
            if (mode == 0) {
                for (int i = 0; i < head_dim; i++) dst[i] = src[i];
                continue;
            }
            float scale = 1.0f;
            if (mode == 1) {
                float sum_sq = 0.0f;
                for (int i = 0; i < head_dim; i++) sum_sq += src[i] * src[i];
                scale = 1.0f / sqrtf(sum_sq / (float)head_dim + eps);
            }
            if (mode == 3) {
                for (int i = 0; i < half; i++) {
                    int j = 2 * i;
                    float x0 = src[j], x1 = src[j + 1];
                    float c = cos_t[pos * half + i], s = sin_t[pos * half + i];
                    dst[j] = x0 * c - x1 * s;
                    dst[j + 1] = x1 * c + x0 * s;
                }
            } else {
                for (int i = 0; i < half; i++) {
                    float x0 = src[i], x1 = src[i + half];
                    if (mode == 1) {
                        x0 = x0 * scale * norm_w[i];
                        x1 = x1 * scale * norm_w[i + half];
                    }
                    float c = cos_t[pos * half + i], s = sin_t[pos * half + i];
                    dst[i] = x0 * c - x1 * s;
                    dst[i + half] = x1 * c + x0 * s;
                }
            }
        }
    }
}

/* cpu.rs:1993-2023 kv_cache_append_head_major: cache [nkv,cap,hd]. */
FO_API void fo_kv_cache_append_head_major(float *cache_k, float *cache_v, int cache_len,
                                          int cache_capacity, const float *new_k,
                                          const float *new_v, int new_tokens, int nkv, int hd) {
    for (int h = 0; h < nkv; h++) {
        long dst = (long)h * cache_capacity * hd + (long)cache_len * hd;
        long src = (long)h * new_tokens * hd;
        memcpy(cache_k + dst, new_k + src, sizeof(float) * (long)new_tokens * hd);
        memcpy(cache_v + dst, new_v + src, sizeof(float) * (long)new_tokens * hd);
    }
}

/* cpu.rs:2025-2040 transpose_head_to_token: [heads,T,d] → [T,heads,d]. */
FO_API void fo_transpose_head_to_token(const float *src, float *dst, int tokens, int heads,
                                       int dim) {
    for (int h = 0; h < heads; h++)
        for (int t = 0; t < tokens; t++)
            memcpy(dst + ((long)t * heads + h) * dim, src + ((long)h * tokens + t) * dim,
                   sizeof(float) * dim);
}

/* cpu.rs:2179-2259 cpu_attention (batch = 1).  q/out [nh,q_len,d] head-major,
 * k/v [nkv,kv_stride,d]; online softmax one key at a time; causal uses
 * attend_end = min(pos_offset+qi+1, kv_len); sliding window trims the start. */
FO_API void fo_cpu_attention(const float *q, const float *k, const float *v, float *out,
                             int q_len, int kv_len, int causal, int pos_offset, int nh,
                             int nkv, int d, float scale, int kv_seq_stride,
                             int sliding_window) {
    int n_rep = nh / nkv;
    long kv_stride = kv_seq_stride > 0 ? kv_seq_stride : kv_len;
    if (d > 512) return;   /* the per-thread accumulator below; the reference's models use 64/128/256 */
#pragma omp parallel for schedule(dynamic) num_threads(fo_threads) if (fo_threads > 1 && (long)nh * q_len * kv_len > (1L << 16))
    for (int h = 0; h < nh; h++) {
        float acc[512];   /* d ≤ 512 (head_dim 64/128/256) */
        int kv_h = h / n_rep;
        long q_off = (long)h * q_len * d;
        long k_off = (long)kv_h * kv_stride * d;
        for (int qi = 0; qi < q_len; qi++) {
            int attend_end = kv_len;
            if (causal) {
                attend_end = pos_offset + qi + 1;
                if (attend_end > kv_len) attend_end = kv_len;
            }
            int attend_start = 0;
            if (causal && sliding_window > 0)
                attend_start = attend_end > sliding_window ? attend_end - sliding_window : 0;
            float max_score = -INFINITY, sum_exp = 0.0f;
            for (int di = 0; di < d; di++) acc[di] = 0.0f;
            for (int ki = attend_start; ki < attend_end; ki++) {
                float dot = 0.0f;
                for (int di = 0; di < d; di++)
                    dot += q[q_off + (long)qi * d + di] * k[k_off + (long)ki * d + di];
                float score = dot * scale;
                if (score > max_score) {
                    float correction = expf(max_score - score);
                    for (int di = 0; di < d; di++) acc[di] *= correction;
                    sum_exp *= correction;
                    max_score = score;
                }
                float w = expf(score - max_score);
                sum_exp += w;
                for (int di = 0; di < d; di++) acc[di] += w * v[k_off + (long)ki * d + di];
            }
            if (sum_exp > 0.0f) {
                float inv = 1.0f / sum_exp;
                for (int di = 0; di < d; di++) out[q_off + (long)qi * d + di] = acc[di] * inv;
            }
        }
    }
}

/* ferrum-kv/src/attention.rs:30-114 paged_attention over a paged pool.
 * pool_k/pool_v: [num_blocks][block_size][nkv][hd] (blocks/storage.rs:38-46,
 * `[slot][head][dim]` per block); block_table[logical] → physical
 * (managers/paged.rs:498-561 read_kv).  query/out [q_tokens,nh,hd] token-major.
 * Three-pass softmax (scores → exp/sum → divide → weighted V), causal mask
 * max_visible = kv_len - q_tokens + qt, scale 1/sqrt(hd). */
FO_API int fo_paged_attention(const float *query, int q_tokens, int nh, int nkv, int hd,
                              const float *pool_k, const float *pool_v,
                              const int32_t *block_table, int block_size, int kv_len,
                              float *output) {
    if (kv_len <= 0) return -1;
    int heads_per_kv = nh / nkv;
    float scale = 1.0f / sqrtf((float)hd);
    float *scores = (float *)malloc(sizeof(float) * kv_len);
    memset(output, 0, sizeof(float) * (long)q_tokens * nh * hd);
    for (int qt = 0; qt < q_tokens; qt++)
        for (int h = 0; h < nh; h++) {
            int kv_h = h / heads_per_kv;
            const float *q = query + ((long)qt * nh + h) * hd;
            for (int p = 0; p < kv_len; p++) {
                long blk = block_table[p / block_size];
                const float *kk = pool_k + ((blk * block_size + p % block_size) * nkv + kv_h) * hd;
                float dot = 0.0f;
                for (int i = 0; i < hd; i++) dot += q[i] * kk[i];
                scores[p] = dot * scale;
            }
            int max_visible = kv_len - q_tokens + qt;
            for (int p = max_visible + 1; p < kv_len; p++) scores[p] = -INFINITY;
            float max_score = -INFINITY;
            for (int p = 0; p < kv_len; p++) max_score = fmaxf(max_score, scores[p]);
            float sum = 0.0f;
            for (int p = 0; p < kv_len; p++) {
                scores[p] = expf(scores[p] - max_score);
                sum += scores[p];
            }
            if (sum > 0.0f)
                for (int p = 0; p < kv_len; p++) scores[p] /= sum;
            float *o = output + ((long)qt * nh + h) * hd;
            for (int p = 0; p < kv_len; p++) {
                long blk = block_table[p / block_size];
                const float *vv = pool_v + ((blk * block_size + p % block_size) * nkv + kv_h) * hd;
                float w = scores[p];
                for (int i = 0; i < hd; i++) o[i] += w * vv[i];
            }
        }
    free(scores);
    return 0;
}

/* ───────────────────────────────── GPTQ ────────────────────────────────── */

/* cpu.rs:2283-2315 cpu_dequant_gptq → w[n,k] f32;
 * q = (qweight[k/8,n] >> 4(k%8)) & 15; zero = nibble(qzeros[k/g, n/8]) + 1.
 * g_idx == NULL ⇒ group = k/g (cpu.rs ignores g_idx); with g_idx the group
 * comes from g_idx[k] (gptq_parity_test.rs:168-189). */
FO_API int fo_dequant_gptq(const int32_t *qweight, const float *scales, const int32_t *qzeros,
                           const int32_t *g_idx, int bits, int group_size, int k, int n,
                           float *w) {
    if (bits != 4) return -1;
    int packed_rows = k / 8;
    for (int pr = 0; pr < packed_rows; pr++)
        for (int col = 0; col < n; col++) {
            uint32_t packed = (uint32_t)qweight[(long)pr * n + col];
            for (int bi = 0; bi < 8; bi++) {
                int ki = pr * 8 + bi;
                int q = (int)((packed >> (bi * 4)) & 0xF);
                int grp = g_idx ? g_idx[ki] : ki / group_size;
                float scale = scales[(long)grp * n + col];
                uint32_t z_packed = (uint32_t)qzeros[(long)grp * (n / 8) + col / 8];
                int zero = (int)((z_packed >> ((col % 8) * 4)) & 0xF) + 1;
                w[(long)col * k + ki] = (float)(q - zero) * scale;
            }
        }
    return 0;
}

/* ────────────────────────────────── MoE ────────────────────────────────── */

/* ferrum-models/src/moe/router.rs:113-195 route_into. */
FO_API void fo_route_topk(const float *logits, int batch, int num_experts, int top_k,
                          int norm_topk_prob, uint32_t *expert_ids, float *expert_weights) {
    float *probs = (float *)malloc(sizeof(float) * num_experts);
    for (int b = 0; b < batch; b++) {
        const float *row = logits + (long)b * num_experts;
        float max = -INFINITY;
        for (int i = 0; i < num_experts; i++)
            if (row[i] > max) max = row[i];
        float sum = 0.0f;
        for (int i = 0; i < num_experts; i++) {
            float e = expf(row[i] - max);
            probs[i] = e;
            sum += e;
        }
        float inv_sum = 1.0f / sum;
        for (int i = 0; i < num_experts; i++) probs[i] *= inv_sum;
        float sel_sum = 0.0f;
        long lo = (long)b * top_k;
        for (int k = 0; k < top_k; k++) {
            float best = -INFINITY;
            int best_idx = 0;
            for (int i = 0; i < num_experts; i++)
                if (probs[i] > best) {
                    best = probs[i];
                    best_idx = i;
                }
            expert_ids[lo + k] = (uint32_t)best_idx;
            expert_weights[lo + k] = best;
            sel_sum += best;
            probs[best_idx] = -INFINITY;
        }
        if (norm_topk_prob) {
            if (sel_sum > 0.0f) {
                float scale = 1.0f / sel_sum;
                for (int k = 0; k < top_k; k++) expert_weights[lo + k] *= scale;
            } else {
                float uniform = 1.0f / (float)top_k;
                for (int k = 0; k < top_k; k++) expert_weights[lo + k] = uniform;
            }
        }
    }
    free(probs);
}

/* ferrum-models/src/moe/dispatch.rs:1408-1461 MoeBucketPlan::rebuild_into:
 * stable counting sort of (b,k) pairs by expert.  expert_offsets[E+1],
 * packed_token_idx[T·k] (slot → token), pairs_by_token[T·k] (pair → slot). */
FO_API void fo_bucket_plan(const uint32_t *expert_ids, int batch, int num_experts, int top_k,
                           uint32_t *expert_offsets, uint32_t *packed_token_idx,
                           int32_t *pairs_by_token) {
    int total = batch * top_k;
    for (int e = 0; e <= num_experts; e++) expert_offsets[e] = 0;
    for (int i = 0; i < total; i++) {
        packed_token_idx[i] = 0;
        pairs_by_token[i] = -1;
    }
    for (int i = 0; i < total; i++) expert_offsets[expert_ids[i] + 1] += 1;
    for (int e = 0; e < num_experts; e++) expert_offsets[e + 1] += expert_offsets[e];
    uint32_t *cursors = (uint32_t *)malloc(sizeof(uint32_t) * num_experts);
    memcpy(cursors, expert_offsets, sizeof(uint32_t) * num_experts);
    for (int b = 0; b < batch; b++)
        for (int k = 0; k < top_k; k++) {
            int pair = b * top_k + k;
            uint32_t eid = expert_ids[pair];
            uint32_t slot = cursors[eid]++;
            packed_token_idx[slot] = (uint32_t)b;
            pairs_by_token[pair] = (int32_t)slot;
        }
    free(cursors);
}

/* ferrum-kernels/kernels/moe_align_block_size_pair_ids.cu:13-95 (the vLLM-
 * native variant the bucketed path uses, dispatch.rs:1919-1941): counts are
 * padded to block_size per expert; sorted_token_ids[slot] = flattened pair id
 * p = token·top_k + k, sentinel = batch·top_k elsewhere; block_ids[b] = expert
 * of rows [b·block, (b+1)·block); total_post_pad.  The CUDA kernel claims
 * slots with atomicAdd (order inside an expert unspecified); this restatement
 * uses ascending pair id, the order the host plan (dispatch.rs:1446-1455)
 * defines.  sorted_max = T·k + E·block (dispatch.rs:1865). */
FO_API int fo_moe_align_block_size(const int32_t *expert_ids_per_pair, int batch_x_topk,
                                   int num_experts, int block_size, int sorted_max,
                                   int32_t *sorted_token_ids, int32_t *block_ids,
                                   int32_t *total_post_pad) {
    int *counts = (int *)calloc(num_experts, sizeof(int));
    int *offsets = (int *)calloc(num_experts + 1, sizeof(int));
    int *cursors = (int *)calloc(num_experts, sizeof(int));
    for (int i = 0; i < sorted_max; i++) sorted_token_ids[i] = batch_x_topk;
    for (int p = 0; p < batch_x_topk; p++) {
        int e = expert_ids_per_pair[p];
        if (e >= 0 && e < num_experts) counts[e]++;
    }
    int acc = 0;
    for (int e = 0; e < num_experts; e++) {
        offsets[e] = acc;
        acc += ((counts[e] + block_size - 1) / block_size) * block_size;
    }
    offsets[num_experts] = acc;
    *total_post_pad = acc;
    for (int e = 0; e < num_experts; e++) cursors[e] = offsets[e];
    for (int p = 0; p < batch_x_topk; p++) {
        int e = expert_ids_per_pair[p];
        if (e >= 0 && e < num_experts) sorted_token_ids[cursors[e]++] = p;
    }
    int total_blocks = acc / block_size;
    for (int b = 0; b < total_blocks; b++) {
        int row = b * block_size, e = 0;
        for (int ei = 0; ei < num_experts; ei++)
            if (offsets[ei] <= row && row < offsets[ei + 1]) {
                e = ei;
                break;
            }
        block_ids[b] = e;
    }
    free(counts);
    free(offsets);
    free(cursors);
    return total_blocks;
}

/* ferrum-models/src/moe/dispatch.rs:2208-2288 moe_forward_cpu.  Expert weights
 * are the CPU backend's dequantised stacks (cpu.rs:2340-2379): gate_up_w
 * [E][2I][H], down_w [E][H][I], both f32.  Per (b,k): gu = x_b·W_gu[e]ᵀ,
 * h = silu(gu[:I])·gu[I:], d = h·W_d[e]ᵀ, out[b] += weight·d (k ascending). */
FO_API void fo_moe_forward_cpu(const float *x, int batch, int hidden, int inter, int top_k,
                               const uint32_t *expert_ids, const float *expert_weights,
                               const float *gate_up_w, const float *down_w, float *out) {
    memset(out, 0, sizeof(float) * (long)batch * hidden);
    /* tokens are independent (each b owns out[b]); the inner 1-row gemms stay serial inside a token's thread */
#pragma omp parallel num_threads(fo_threads) if (fo_threads > 1 && batch > 1)
    {
    float *gu = (float *)malloc(sizeof(float) * 2 * inter);
    float *act = (float *)malloc(sizeof(float) * inter);
    float *dn = (float *)malloc(sizeof(float) * hidden);
#pragma omp for schedule(dynamic)
    for (int b = 0; b < batch; b++) {
        const float *xb = x + (long)b * hidden;
        for (int k = 0; k < top_k; k++) {
            int pair = b * top_k + k;
            long e = expert_ids[pair];
            float weight = expert_weights[pair];
            fo_gemm(xb, gate_up_w + e * 2 * inter * hidden, gu, 1, 2 * inter, hidden);
            fo_fused_silu_mul_split(gu, act, 1, inter);
            fo_gemm(act, down_w + e * (long)hidden * inter, dn, 1, hidden, inter);
            float *o = out + (long)b * hidden;
            for (int i = 0; i < hidden; i++) o[i] += weight * dn[i];
        }
    }
    free(gu);
    free(act);
    free(dn);
    }
}

/* ──────────────────────────────── sampling ─────────────────────────────── */

/* ferrum-kernels/src/backend/traits.rs:1534-1555 argmax_rows_f16 default:
 * strict `>` ⇒ FIRST maximum wins. */
FO_API void fo_argmax_rows(const float *logits, int m, int n, uint32_t *out) {
    for (int r = 0; r < m; r++) {
        const float *row = logits + (long)r * n;
        int max_idx = 0;
        float max_val = -INFINITY;
        for (int i = 0; i < n; i++)
            if (row[i] > max_val) {
                max_val = row[i];
                max_idx = i;
            }
        out[r] = (uint32_t)max_idx;
    }
}

/* ferrum-interfaces/src/sampler.rs:359-378 GreedySampler: Iterator::max_by
 * returns the LAST maximum on ties (partial_cmp Equal keeps the later one). */
FO_API uint32_t fo_greedy_sample(const float *logits, int n) {
    int best = 0;
    for (int i = 1; i < n; i++)
        if (!(logits[i] < logits[best])) best = i; /* a<=b (or unordered) → take b */
    return (uint32_t)best;
}

/* sampler.rs:327-345 RepetitionPenaltyProcessor (first occurrence of each id
 * only; ids ≥ vocab skipped): v>0 ? v/p : v·p. */
FO_API void fo_repetition_penalty(float *logits, int n, const uint32_t *prev, int n_prev,
                                  float penalty) {
    if (penalty == 1.0f) return;
    unsigned char *seen = (unsigned char *)calloc(n, 1);
    for (int i = 0; i < n_prev; i++) {
        uint32_t id = prev[i];
        if (id >= (uint32_t)n || seen[id]) continue;
        seen[id] = 1;
        float cur = logits[id];
        logits[id] = cur > 0.0f ? cur / penalty : cur * penalty;
    }
    free(seen);
}

/* sampler.rs:196-204 TemperatureProcessor. */
FO_API void fo_temperature(float *logits, int n, float temperature) {
    if (temperature > 0.0f && temperature != 1.0f)
        for (int i = 0; i < n; i++) logits[i] /= temperature;
}

/* stable descending argsort (Rust sort_by is a stable merge sort). */
static void fo_stable_argsort_desc(const float *v, int n, int *idx) {
    int *tmp = (int *)malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++) idx[i] = i;
    for (int width = 1; width < n; width *= 2) {
        for (int lo = 0; lo < n; lo += 2 * width) {
            int mid = lo + width < n ? lo + width : n;
            int hi = lo + 2 * width < n ? lo + 2 * width : n;
            int a = lo, b = mid, o = lo;
            while (a < mid && b < hi) {
                /* take from the right run only when strictly greater */
                if (v[idx[b]] > v[idx[a]]) tmp[o++] = idx[b++];
                else tmp[o++] = idx[a++];
            }
            while (a < mid) tmp[o++] = idx[a++];
            while (b < hi) tmp[o++] = idx[b++];
        }
        memcpy(idx, tmp, sizeof(int) * n);
    }
    free(tmp);
}

/* sampler.rs:226-247 TopKProcessor: threshold = k-th largest; mask < threshold. */
FO_API void fo_top_k(float *logits, int n, int k) {
    if (k <= 0 || k >= n) return;
    int *idx = (int *)malloc(sizeof(int) * n);
    fo_stable_argsort_desc(logits, n, idx);
    float threshold = logits[idx[k - 1]];
    for (int i = 0; i < n; i++)
        if (logits[i] < threshold) logits[i] = -INFINITY;
    free(idx);
}

/* sampler.rs:265-309 TopPProcessor. */
FO_API void fo_top_p(float *logits, int n, float p) {
    if (!(p < 1.0f && p > 0.0f)) return;
    float max_logit = -INFINITY;
    for (int i = 0; i < n; i++) max_logit = fmaxf(max_logit, logits[i]);
    float *probs = (float *)malloc(sizeof(float) * n);
    float sum = 0.0f;
    for (int i = 0; i < n; i++) probs[i] = expf(logits[i] - max_logit);
    for (int i = 0; i < n; i++) sum += probs[i];
    for (int i = 0; i < n; i++) probs[i] /= sum;
    int *idx = (int *)malloc(sizeof(int) * n);
    fo_stable_argsort_desc(probs, n, idx);
    float cum = 0.0f;
    int cutoff = n;
    for (int i = 0; i < n; i++) {
        cum += probs[idx[i]];
        if (cum > p) {
            cutoff = i + 1;
            break;
        }
    }
    for (int i = cutoff; i < n; i++) logits[idx[i]] = -INFINITY;
    free(probs);
    free(idx);
}

/* sampler.rs:385-425 MultinomialSampler with threshold = u32 / u32::MAX. */
FO_API uint32_t fo_multinomial(const float *logits, int n, uint32_t rng_u32) {
    float max_logit = -INFINITY;
    for (int i = 0; i < n; i++) max_logit = fmaxf(max_logit, logits[i]);
    float *probs = (float *)malloc(sizeof(float) * n);
    float sum = 0.0f;
    for (int i = 0; i < n; i++)
        probs[i] = (isfinite(logits[i])) ? expf(logits[i] - max_logit) : 0.0f;
    for (int i = 0; i < n; i++) sum += probs[i];
    uint32_t result = (uint32_t)(n - 1);
    if (sum > 0.0f) {
        for (int i = 0; i < n; i++) probs[i] /= sum;
        float threshold = (float)rng_u32 / (float)UINT32_MAX;
        float cumulative = 0.0f;
        for (int i = 0; i < n; i++) {
            cumulative += probs[i];
            if (cumulative >= threshold) {
                result = (uint32_t)i;
                break;
            }
        }
    }
    free(probs);
    return result;
}

/* ─────────────────────────────── RoPE tables ───────────────────────────── */

/* ferrum-models/src/models/llama_family.rs:5262-5282 scale_llama3_rope_freq. */
FO_API double fo_scale_llama3_rope_freq(double freq, double factor, double low_freq_factor,
                                        double high_freq_factor, double original_max_pos) {
    double wavelen = 2.0 * M_PI / freq;
    double low_freq_wavelen = original_max_pos / low_freq_factor;
    double high_freq_wavelen = original_max_pos / high_freq_factor;
    if (wavelen < high_freq_wavelen) return freq;
    if (wavelen > low_freq_wavelen) return freq / factor;
    double smooth = (original_max_pos / wavelen - low_freq_factor) /
                    (high_freq_factor - low_freq_factor);
    return (1.0 - smooth) * freq / factor + smooth * freq;
}

/* llama_family.rs:5239-5260 rope_freq.  scaling_kind 0 none, 1 linear (p0 =
 * factor), 2 llama3 (p0 factor, p1 low, p2 high, p3 original_max_pos). */
FO_API double fo_rope_freq(double theta, int head_dim, int pair_idx, int scaling_kind,
                           double p0, double p1, double p2, double p3) {
    double base = 1.0 / pow(theta, (double)(2 * pair_idx) / (double)head_dim);
    if (scaling_kind == 1) return base / p0;
    if (scaling_kind == 2) return fo_scale_llama3_rope_freq(base, p0, p1, p2, p3);
    return base;
}

/* llama_family.rs:5220-5237 build_rope_cache: angle in f64, stored f32. */
FO_API void fo_build_rope_cache(double theta, int head_dim, int max_seq, int scaling_kind,
                                double p0, double p1, double p2, double p3, float *cos_t,
                                float *sin_t) {
    int half = head_dim / 2;
    for (int pos = 0; pos < max_seq; pos++)
        for (int i = 0; i < half; i++) {
            double freq = fo_rope_freq(theta, head_dim, i, scaling_kind, p0, p1, p2, p3);
            double angle = (double)pos * freq;
            cos_t[(long)pos * half + i] = (float)cos(angle);
            sin_t[(long)pos * half + i] = (float)sin(angle);
        }
}

/* llama_family.rs bf16_round (tests :5951-5959): round-to-nearest-even to
 * bf16, returned as f32 with the low 16 bits cleared. */
FO_API float fo_bf16_round(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    uint32_t r = u + 0x7FFFu + ((u >> 16) & 1u);
    r &= 0xFFFF0000u;
    float y;
    memcpy(&y, &r, 4);
    return y;
}

/* ───────────────────────────── block allocator ─────────────────────────── */
/* ferrum-models/src/common/paged_pool.rs:106-365 BlockAllocator: ids from 0,
 * LIFO free list, allocate prefers an un-hashed free block via
 * rposition + swap_remove (:182-193), ref counts, hash table. */
typedef struct {
    uint32_t capacity;
    uint32_t *free_list;
    uint32_t free_len;
    uint16_t *ref_counts;
    uint8_t *has_hash;
    uint64_t *block_hash;
    /* hash_table as a flat list of (hash → block); capacity entries max. */
    uint64_t *ht_hash;
    uint32_t *ht_block;
    uint32_t ht_len;
    uint32_t peak_in_use;
} fo_allocator;

/* ── KV block content hashes (ferrum-models/src/common/paged_pool.rs:60-98) ──────────────────────────
 * `block_hash` feeds Rust's `DefaultHasher::new()` — SipHash-1-3 with a zero key (a third-party algorithm as far as
 * /root/reference is concerned: it lives in Rust's std, version pinned by the toolchain, not vendored) — the parent
 * hash as a u64 and every token id as a u32, native endian.  The routine below is the published SipHash-c-d
 * (Aumasson & Bernstein 2012), pinned by the paper's SipHash-2-4 vectors in tests/test_oracle_golden.py; the 1-3
 * VALUES have no golden in the reference (its tests check chain properties only) → value parity with Rust unpinned. */
#define FO_ROTL64(x, b) (((x) << (b)) | ((x) >> (64 - (b))))
#define FO_SIPROUND do { \
    v0 += v1; v1 = FO_ROTL64(v1, 13); v1 ^= v0; v0 = FO_ROTL64(v0, 32); \
    v2 += v3; v3 = FO_ROTL64(v3, 16); v3 ^= v2; \
    v0 += v3; v3 = FO_ROTL64(v3, 21); v3 ^= v0; \
    v2 += v1; v1 = FO_ROTL64(v1, 17); v1 ^= v2; v2 = FO_ROTL64(v2, 32); } while (0)
FO_API uint64_t fo_siphash(int c, int d, uint64_t k0, uint64_t k1, const uint8_t *in, size_t len) {
    uint64_t v0 = 0x736f6d6570736575ULL ^ k0, v1 = 0x646f72616e646f6dULL ^ k1;
    uint64_t v2 = 0x6c7967656e657261ULL ^ k0, v3 = 0x7465646279746573ULL ^ k1;
    const uint8_t *end = in + (len - (len % 8));
    for (; in != end; in += 8) {
        uint64_t m = 0;
        for (int j = 7; j >= 0; j--) m = (m << 8) | in[j];
        v3 ^= m;
        for (int i = 0; i < c; i++) FO_SIPROUND;
        v0 ^= m;
    }
    uint64_t b = ((uint64_t)len) << 56;
    for (int j = (int)(len % 8) - 1; j >= 0; j--) b |= ((uint64_t)in[j]) << (8 * j);
    v3 ^= b;
    for (int i = 0; i < c; i++) FO_SIPROUND;
    v0 ^= b;
    v2 ^= 0xff;
    for (int i = 0; i < d; i++) FO_SIPROUND;
    return v0 ^ v1 ^ v2 ^ v3;
}
/* block_hash_chain (paged_pool.rs:89-98): out gets one hash per full block; returns the count. */
FO_API int fo_block_hash_chain(const uint32_t *tokens, int n, int block_size, uint64_t *out) {
    uint64_t parent = 0;
    int count = 0;
    uint8_t *msg = (uint8_t *)malloc(8 + 4 * (size_t)block_size);
    for (int i = 0; i + block_size <= n; i += block_size) {
        memcpy(msg, &parent, 8);                                   /* little-endian host, like the reference's targets */
        memcpy(msg + 8, tokens + i, 4 * (size_t)block_size);
        parent = fo_siphash(1, 3, 0, 0, msg, 8 + 4 * (size_t)block_size);
        out[count++] = parent;
    }
    free(msg);
    return count;
}

FO_API fo_allocator *fo_alloc_new(uint32_t num_blocks) {
    fo_allocator *a = (fo_allocator *)calloc(1, sizeof(fo_allocator));
    a->capacity = num_blocks;
    a->free_list = (uint32_t *)malloc(sizeof(uint32_t) * (num_blocks ? num_blocks : 1));
    for (uint32_t i = 0; i < num_blocks; i++) a->free_list[i] = num_blocks - 1 - i;
    a->free_len = num_blocks;
    a->ref_counts = (uint16_t *)calloc(num_blocks ? num_blocks : 1, sizeof(uint16_t));
    a->has_hash = (uint8_t *)calloc(num_blocks ? num_blocks : 1, 1);
    a->block_hash = (uint64_t *)calloc(num_blocks ? num_blocks : 1, sizeof(uint64_t));
    a->ht_hash = (uint64_t *)calloc(num_blocks ? num_blocks : 1, sizeof(uint64_t));
    a->ht_block = (uint32_t *)calloc(num_blocks ? num_blocks : 1, sizeof(uint32_t));
    return a;
}

FO_API void fo_alloc_free_obj(fo_allocator *a) {
    free(a->free_list);
    free(a->ref_counts);
    free(a->has_hash);
    free(a->block_hash);
    free(a->ht_hash);
    free(a->ht_block);
    free(a);
}

static int fo_ht_find(fo_allocator *a, uint64_t h) {
    for (uint32_t i = 0; i < a->ht_len; i++)
        if (a->ht_hash[i] == h) return (int)i;
    return -1;
}
static void fo_ht_remove_at(fo_allocator *a, int i) {
    a->ht_len--;
    a->ht_hash[i] = a->ht_hash[a->ht_len];
    a->ht_block[i] = a->ht_block[a->ht_len];
}
static void fo_ht_insert(fo_allocator *a, uint64_t h, uint32_t block) {
    int i = fo_ht_find(a, h);
    if (i >= 0) {
        a->ht_block[i] = block;
        return;
    }
    a->ht_hash[a->ht_len] = h;
    a->ht_block[a->ht_len] = block;
    a->ht_len++;
}
/* swap_remove on the free list */
static uint32_t fo_free_swap_remove(fo_allocator *a, uint32_t pos) {
    uint32_t b = a->free_list[pos];
    a->free_len--;
    a->free_list[pos] = a->free_list[a->free_len];
    return b;
}
/* paged_pool.rs:182-193 */
static int fo_pop_free_preferring_unhashed(fo_allocator *a, uint32_t *out) {
    if (a->free_len == 0) return 0;
    uint32_t pos = a->free_len - 1;
    for (int64_t i = (int64_t)a->free_len - 1; i >= 0; i--)
        if (!a->has_hash[a->free_list[i]]) {
            pos = (uint32_t)i;
            break;
        }
    *out = fo_free_swap_remove(a, pos);
    return 1;
}
/* paged_pool.rs:197-209 */
static void fo_evict_hash_if_any(fo_allocator *a, uint32_t block) {
    if (!a->has_hash[block]) return;
    a->has_hash[block] = 0;
    int i = fo_ht_find(a, a->block_hash[block]);
    if (i >= 0 && a->ht_block[i] == block) fo_ht_remove_at(a, i);
}
static void fo_track_peak(fo_allocator *a) {
    uint32_t in_use = a->capacity - a->free_len;
    if (in_use > a->peak_in_use) a->peak_in_use = in_use;
}

/* paged_pool.rs:159-180; returns -1 when exhausted. */
FO_API int64_t fo_alloc_allocate(fo_allocator *a) {
    uint32_t b;
    if (!fo_pop_free_preferring_unhashed(a, &b)) return -1;
    fo_evict_hash_if_any(a, b);
    a->ref_counts[b] = 1;
    fo_track_peak(a);
    return (int64_t)b;
}

/* paged_pool.rs:213-236 allocate_n: atomic. */
FO_API int fo_alloc_allocate_n(fo_allocator *a, uint32_t n, uint32_t *out) {
    if (a->free_len < n) return -1;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t b;
        fo_pop_free_preferring_unhashed(a, &b);
        fo_evict_hash_if_any(a, b);
        a->ref_counts[b] = 1;
        out[i] = b;
    }
    fo_track_peak(a);
    return 0;
}

/* paged_pool.rs:333-345 free. */
FO_API void fo_alloc_free(fo_allocator *a, const uint32_t *blocks, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) {
        uint32_t b = blocks[i];
        a->ref_counts[b] -= 1;
        if (a->ref_counts[b] == 0) a->free_list[a->free_len++] = b;
    }
}

/* paged_pool.rs:307-316 acquire. */
FO_API void fo_alloc_acquire(fo_allocator *a, uint32_t block) { a->ref_counts[block] += 1; }

/* paged_pool.rs:270-292 register_block_hash. */
FO_API void fo_alloc_register_hash(fo_allocator *a, uint32_t block, uint64_t hash) {
    if (a->has_hash[block]) {
        uint64_t old = a->block_hash[block];
        if (old == hash) return;
        int i = fo_ht_find(a, old);
        if (i >= 0 && a->ht_block[i] == block) fo_ht_remove_at(a, i);
    }
    a->has_hash[block] = 1;
    a->block_hash[block] = hash;
    fo_ht_insert(a, hash, block);
}

/* paged_pool.rs:243-265 try_acquire_by_hash; -1 on miss. */
FO_API int64_t fo_alloc_try_acquire_by_hash(fo_allocator *a, uint64_t hash) {
    int i = fo_ht_find(a, hash);
    if (i < 0) return -1;
    uint32_t block = a->ht_block[i];
    if (a->ref_counts[block] == 0) {
        int64_t pos = -1;
        for (int64_t j = (int64_t)a->free_len - 1; j >= 0; j--)
            if (a->free_list[j] == block) {
                pos = j;
                break;
            }
        if (pos < 0) return -1;
        fo_free_swap_remove(a, (uint32_t)pos);
        a->ref_counts[block] = 1;
        fo_track_peak(a);
    } else {
        a->ref_counts[block] += 1;
    }
    return (int64_t)block;
}

FO_API uint32_t fo_alloc_free_count(fo_allocator *a) { return a->free_len; }
FO_API uint32_t fo_alloc_ref_count(fo_allocator *a, uint32_t b) { return a->ref_counts[b]; }
FO_API uint32_t fo_alloc_peak_in_use(fo_allocator *a) { return a->peak_in_use; }
FO_API uint32_t fo_alloc_hash_table_size(fo_allocator *a) { return a->ht_len; }

/* ferrum-kernels/src/moe_host.rs:22-56 compute_ids_tpe: tpe[e] = #pairs of
 * expert e; ids[e·mpe + slot] = pair index in (b,k) order; mpe = max(1, max
 * count).  `ids` must hold num_experts·batch·top_k entries (upper bound); it
 * is written with row stride mpe, the value returned. */
FO_API int fo_compute_ids_tpe(const uint32_t *selected, int num_experts, int batch, int top_k,
                              int32_t *tpe, int32_t *ids) {
    int n = batch * top_k;
    for (int e = 0; e < num_experts; e++) tpe[e] = 0;
    for (int i = 0; i < n; i++)
        if (selected[i] < (uint32_t)num_experts) tpe[selected[i]]++;
    int mpe = 1;
    for (int e = 0; e < num_experts; e++)
        if (tpe[e] > mpe) mpe = tpe[e];
    for (long i = 0; i < (long)num_experts * mpe; i++) ids[i] = 0;
    int *fill = (int *)calloc(num_experts, sizeof(int));
    for (int i = 0; i < n; i++) {
        uint32_t e = selected[i];
        if (e < (uint32_t)num_experts) ids[(long)e * mpe + fill[e]++] = i;
    }
    free(fill);
    return mpe;
}
