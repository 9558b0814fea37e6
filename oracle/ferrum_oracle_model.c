/*
 * ferrum_oracle_model.c — model-level CPU restatement (TEST INFRASTRUCTURE ONLY;
 * see the header of ferrum_oracle.c for who may load it).
 *
 * Follows the reference's CPU path B:
 *   LlamaFamilyModel<CpuBackend>::prefill_internal / decode_internal
 *     (ferrum-models/src/models/llama_family.rs:3772,4112)
 *   → forward_layer_with_residual_shadow (llama_family.rs:2793-3187): rms_norm →
 *     qkv_proj → contig_write (ferrum-kernels/src/backend/kv_layer.rs:370-495:
 *     split_qkv + 3× qk_norm_rope + kv_cache_append_head_major) → set_len →
 *     flash_attention(causal, scale 1/sqrt(hd), kv_seq_stride = capacity)
 *   → forward_layer_post_attn (llama_family.rs:3207-3606): transpose_head_to_token
 *     (tokens>1) → o_proj → fused_add_rms_norm → gate_up → silu/gelu·mul → down →
 *     add_inplace
 *   Qwen3-MoE tail (qwen3_moe_forward_unified_layer.rs:385-451 / moe_forward_cpu
 *     ferrum-models/src/moe/dispatch.rs:2208-2288): fused_add_rms_norm → router
 *     gemm → route_into → per-(b,k) expert MLP → weighted sum → add_inplace.
 *   final rms_norm + lm_head (tied to embed when absent, llama_family.rs:969-1001).
 * All activations f32; every GEMM is the f64-accumulating triple loop
 * (cpu.rs:482-492).  GPTQ linears are dequantised at load to [n,k] f32
 * (cpu.rs:2283-2339), exactly as the reference CPU backend does.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FO_API __attribute__((visibility("default")))

/* from ferrum_oracle.c */
void fo_gemm(const float *, const float *, float *, int, int, int);
void fo_rms_norm(const float *, const float *, float, float *, int, int);
void fo_fused_add_rms_norm(float *, const float *, const float *, float, float *, int, int);
void fo_embedding_lookup(const float *, const uint32_t *, int, float *, int);
void fo_split_qkv(const float *, float *, float *, float *, int, int, int);
void fo_fused_silu_mul_split(const float *, float *, int, int);
void fo_fused_gelu_tanh_mul_split(const float *, float *, int, int);
void fo_add_inplace(float *, const float *, long);
void fo_scale_inplace(float *, float, long);
void fo_qk_norm_rope(const float *, const float *, const float *, const float *, float *, int,
                     int, int, int, float, int);
void fo_kv_cache_append_head_major(float *, float *, int, int, const float *, const float *,
                                   int, int, int);
void fo_transpose_head_to_token(const float *, float *, int, int, int);
void fo_cpu_attention(const float *, const float *, const float *, float *, int, int, int, int,
                      int, int, int, float, int, int);
int fo_dequant_gptq(const int32_t *, const float *, const int32_t *, const int32_t *, int, int,
                    int, int, float *);
void fo_route_topk(const float *, int, int, int, int, uint32_t *, float *);
void fo_moe_forward_cpu(const float *, int, int, int, int, const uint32_t *, const float *,
                        const float *, const float *, float *);
void fo_build_rope_cache(double, int, int, int, double, double, double, double, float *,
                         float *);

typedef struct {
    int32_t num_layers, hidden, num_heads, num_kv_heads, head_dim, intermediate, vocab;
    int32_t max_seq_len;
    int32_t has_qk_norm;      /* qk_mode 1 (norm + half-split rope) else 2 */
    int32_t activation;       /* 0 silu, 1 gelu_tanh */
    int32_t num_experts;      /* 0 → dense MLP */
    int32_t top_k, expert_inter, norm_topk_prob;
    int32_t rope_scaling_kind; /* 0 none, 1 linear, 2 llama3 */
    int32_t sliding_window;    /* 0 = full attention on every layer */
    float rms_eps;
    float _pad;
    double rope_theta;
    double rope_p0, rope_p1, rope_p2, rope_p3;
    /* Gemma-3 layer semantics (llama_family.rs:520-552): */
    int32_t sliding_window_pattern; /* N > 0: layer (idx+1) % N == 0 is global (full attention, main rope table), the rest
                                       are local (sliding_window, local rope table); 0 = uniform */
    int32_t sandwich_norms;         /* post_attn_ln / post_ffn_ln applied to the branch BEFORE the residual add */
    float embed_scale;              /* 0 = none; else embedding output × this */
    float _pad2;
    double rope_local_theta;        /* 0 = one table; else θ of the unscaled table local layers use */
} fo_model_cfg;

typedef struct {
    float *input_ln, *post_ln, *q_norm, *k_norm;
    float *post_attn_ln, *post_ffn_ln;   /* sandwich norms (Gemma 3); post_ln is then pre_feedforward_layernorm */
    float *qkv_bias;                     /* [qkv_dim] fused projection bias (Qwen2 family) or NULL */
    float *qkv_w, *o_w;          /* [n,k] f32 */
    float *gate_up_w, *down_w;   /* dense MLP */
    float *router_w;             /* [E,H] */
    float *exp_gate_up_w;        /* [E][2I][H] */
    float *exp_down_w;           /* [E][H][I] */
} fo_layer;

typedef struct {
    float *k, *v; /* [layers][nkv][cap][hd] */
    int len;
    int used;
} fo_kv;

#define FO_MAX_CACHES 64

typedef struct {
    fo_model_cfg cfg;
    float *embed;    /* [V,H] */
    float *lm_head;  /* [V,H] or NULL → tied */
    float *final_norm;
    fo_layer *layers;
    float *cos_t, *sin_t;
    float *cos_local, *sin_local;   /* second rope table (local layers), NULL when rope_local_theta == 0 */
    fo_kv caches[FO_MAX_CACHES];
    /* debug taps: last call's hidden state after each layer [L][T][H] (optional) */
    float *tap_hidden;
    int tap_tokens;
    /* test aid: smallest k-th / (k+1)-th router-logit gap over the layers and tokens of the last call — lets a
     * parity test tell a routing near-tie (a legitimately different expert pick) from an arithmetic error */
    float last_route_gap;
    /* … and the same gap relative to that token's router-logit spread (max − min): the scale fp16 storage noise of the router
     * input has to be compared with */
    float last_route_gap_rel;
} fo_model;

static float *fo_dup(const float *src, long n) {
    float *p = (float *)malloc(sizeof(float) * n);
    memcpy(p, src, sizeof(float) * n);
    return p;
}

FO_API fo_model *fo_model_new(const fo_model_cfg *cfg) {
    fo_model *m = (fo_model *)calloc(1, sizeof(fo_model));
    m->cfg = *cfg;
    m->layers = (fo_layer *)calloc(cfg->num_layers, sizeof(fo_layer));
    int half = cfg->head_dim / 2;
    m->cos_t = (float *)malloc(sizeof(float) * (long)cfg->max_seq_len * half);
    m->sin_t = (float *)malloc(sizeof(float) * (long)cfg->max_seq_len * half);
    fo_build_rope_cache(cfg->rope_theta, cfg->head_dim, cfg->max_seq_len,
                        cfg->rope_scaling_kind, cfg->rope_p0, cfg->rope_p1, cfg->rope_p2,
                        cfg->rope_p3, m->cos_t, m->sin_t);
    if (cfg->rope_local_theta > 0.0) {   /* llama_family.rs:1843: own θ, never rope-scaled */
        m->cos_local = (float *)malloc(sizeof(float) * (long)cfg->max_seq_len * half);
        m->sin_local = (float *)malloc(sizeof(float) * (long)cfg->max_seq_len * half);
        fo_build_rope_cache(cfg->rope_local_theta, cfg->head_dim, cfg->max_seq_len, 0, 0, 0, 0, 0, m->cos_local, m->sin_local);
    }
    return m;
}

FO_API void fo_model_free(fo_model *m) {
    for (int l = 0; l < m->cfg.num_layers; l++) {
        fo_layer *L = &m->layers[l];
        free(L->input_ln); free(L->post_ln); free(L->q_norm); free(L->k_norm);
        free(L->post_attn_ln); free(L->post_ffn_ln); free(L->qkv_bias);
        free(L->qkv_w); free(L->o_w); free(L->gate_up_w); free(L->down_w);
        free(L->router_w); free(L->exp_gate_up_w); free(L->exp_down_w);
    }
    for (int c = 0; c < FO_MAX_CACHES; c++) { free(m->caches[c].k); free(m->caches[c].v); }
    free(m->layers); free(m->embed); free(m->lm_head); free(m->final_norm);
    free(m->cos_t); free(m->sin_t); free(m->cos_local); free(m->sin_local); free(m->tap_hidden);
    free(m);
}

/* which: 0 embed [V,H], 1 lm_head [V,H], 2 final_norm [H] */
FO_API void fo_model_set_global(fo_model *m, int which, const float *data) {
    long vh = (long)m->cfg.vocab * m->cfg.hidden;
    if (which == 0) { free(m->embed); m->embed = fo_dup(data, vh); }
    else if (which == 1) { free(m->lm_head); m->lm_head = fo_dup(data, vh); }
    else { free(m->final_norm); m->final_norm = fo_dup(data, m->cfg.hidden); }
}

/* which: 0 input_ln [H], 1 post_ln [H] (sandwich: pre_feedforward_layernorm), 2 q_norm [hd], 3 k_norm [hd], 4 router [E,H],
 *        5 post_attn_ln [H], 6 post_ffn_ln [H] (sandwich norms), 7 qkv_bias [qkv_dim] (GptqLinear bias, gptq.rs:56,
 *        added like Backend::add_bias, cpu.rs:2065-2078) */
FO_API void fo_model_set_layer_dense(fo_model *m, int layer, int which, const float *data) {
    fo_layer *L = &m->layers[layer];
    int H = m->cfg.hidden, hd = m->cfg.head_dim;
    switch (which) {
    case 0: free(L->input_ln); L->input_ln = fo_dup(data, H); break;
    case 1: free(L->post_ln); L->post_ln = fo_dup(data, H); break;
    case 2: free(L->q_norm); L->q_norm = fo_dup(data, hd); break;
    case 3: free(L->k_norm); L->k_norm = fo_dup(data, hd); break;
    case 4: free(L->router_w); L->router_w = fo_dup(data, (long)m->cfg.num_experts * H); break;
    case 5: free(L->post_attn_ln); L->post_attn_ln = fo_dup(data, H); break;
    case 6: free(L->post_ffn_ln); L->post_ffn_ln = fo_dup(data, H); break;
    case 7: free(L->qkv_bias);
        L->qkv_bias = fo_dup(data, (long)(m->cfg.num_heads + 2 * m->cfg.num_kv_heads) * hd); break;
    }
}

/* which: 0 qkv (H → nq·hd+2·nkv·hd), 1 o (nq·hd → H), 2 gate_up (H → 2I), 3 down (I → H),
 *        4 expert gate_up (H → 2·Ie), 5 expert down (Ie → H).  expert ignored for 0-3.
 * GPTQ tensors as in ferrum-quantization/src/gptq.rs:42-112; dequantised here like
 * CpuBackend::load_gptq (cpu.rs:2318-2339). */
FO_API int fo_model_set_gptq(fo_model *m, int layer, int which, int expert,
                             const int32_t *qweight, const float *scales,
                             const int32_t *qzeros, const int32_t *g_idx, int group, int k,
                             int n) {
    fo_layer *L = &m->layers[layer];
    float **slot = NULL;
    long off = 0;
    const fo_model_cfg *c = &m->cfg;
    switch (which) {
    case 0: slot = &L->qkv_w; break;
    case 1: slot = &L->o_w; break;
    case 2: slot = &L->gate_up_w; break;
    case 3: slot = &L->down_w; break;
    case 4:
        if (!L->exp_gate_up_w)
            L->exp_gate_up_w = (float *)calloc((long)c->num_experts * 2 * c->expert_inter * c->hidden, sizeof(float));
        slot = &L->exp_gate_up_w; off = (long)expert * 2 * c->expert_inter * c->hidden; break;
    case 5:
        if (!L->exp_down_w)
            L->exp_down_w = (float *)calloc((long)c->num_experts * c->hidden * c->expert_inter, sizeof(float));
        slot = &L->exp_down_w; off = (long)expert * c->hidden * c->expert_inter; break;
    default: return -1;
    }
    if (which < 4) { free(*slot); *slot = (float *)malloc(sizeof(float) * (long)n * k); }
    return fo_dequant_gptq(qweight, scales, qzeros, g_idx, 4, group, k, n, *slot + off);
}

/* Unquantised projection (DenseLinear, ferrum-kernels/src/linear.rs:109-129; weights upcast to f32 on the CPU backend):
 * which 0 qkv, 1 o, 2 gate_up, 3 down; weight [n, k] row-major — the layout fo_gemm's b operand takes. */
FO_API int fo_model_set_dense(fo_model *m, int layer, int which, const float *weight, int k, int n) {
    fo_layer *L = &m->layers[layer];
    float **slot = which == 0 ? &L->qkv_w : which == 1 ? &L->o_w : which == 2 ? &L->gate_up_w : which == 3 ? &L->down_w : NULL;
    if (!slot) return -1;
    free(*slot);
    *slot = fo_dup(weight, (long)n * k);
    return 0;
}

static fo_kv *fo_get_cache(fo_model *m, int cache_id) {
    fo_kv *c = &m->caches[cache_id];
    if (!c->used) {
        const fo_model_cfg *g = &m->cfg;
        long per = (long)g->num_layers * g->num_kv_heads * g->max_seq_len * g->head_dim;
        c->k = (float *)calloc(per, sizeof(float));
        c->v = (float *)calloc(per, sizeof(float));
        c->len = 0;
        c->used = 1;
    }
    return c;
}

FO_API void fo_model_release_cache(fo_model *m, int cache_id) {
    if (cache_id < 0 || cache_id >= FO_MAX_CACHES) return;
    fo_kv *c = &m->caches[cache_id];
    free(c->k); free(c->v);
    memset(c, 0, sizeof(*c));
}

FO_API int fo_model_cache_len(fo_model *m, int cache_id) {
    return cache_id < 0 || cache_id >= FO_MAX_CACHES ? -1 : m->caches[cache_id].len;
}

/* Copy layer `layer`'s cached K or V for positions [0,len) out as
 * [len][nkv][hd] token-major (the order ferrum-kv read_kv returns). */
FO_API void fo_model_read_kv(fo_model *m, int cache_id, int layer, int is_v, float *out) {
    const fo_model_cfg *g = &m->cfg;
    fo_kv *c = &m->caches[cache_id];
    const float *base = (is_v ? c->v : c->k) +
                        (long)layer * g->num_kv_heads * g->max_seq_len * g->head_dim;
    for (int p = 0; p < c->len; p++)
        for (int h = 0; h < g->num_kv_heads; h++)
            memcpy(out + ((long)p * g->num_kv_heads + h) * g->head_dim,
                   base + ((long)h * g->max_seq_len + p) * g->head_dim,
                   sizeof(float) * g->head_dim);
}

FO_API void fo_model_enable_taps(fo_model *m, int max_tokens) {
    free(m->tap_hidden);
    m->tap_hidden = (float *)calloc((long)m->cfg.num_layers * max_tokens * m->cfg.hidden, sizeof(float));
    m->tap_tokens = max_tokens;
}
FO_API const float *fo_model_taps(fo_model *m) { return m->tap_hidden; }
FO_API float fo_model_last_route_gap(const fo_model *m) { return m->last_route_gap; }
FO_API float fo_model_last_route_gap_rel(const fo_model *m) { return m->last_route_gap_rel; }

/* gap between the k-th and (k+1)-th largest of n logits (n > k) */
static float fo_topk_gap(const float *l, int n, int k) {
    float *tmp = (float *)malloc(sizeof(float) * n);
    memcpy(tmp, l, sizeof(float) * n);
    for (int i = 0; i <= k && i < n; i++) {           /* partial selection sort, descending */
        int best = i;
        for (int j = i + 1; j < n; j++) if (tmp[j] > tmp[best]) best = j;
        float t = tmp[i]; tmp[i] = tmp[best]; tmp[best] = t;
    }
    float gap = k < n ? tmp[k - 1] - tmp[k] : INFINITY;
    free(tmp);
    return gap;
}

/* One forward over `tokens` for sequence `cache_id` starting at pos_offset
 * (must equal the cache length).  logits_out [vocab] receives the logits of
 * the LAST token; if all_logits != NULL it receives [n_tokens, vocab]. */
FO_API int fo_model_forward(fo_model *m, int cache_id, const uint32_t *tokens, int n_tokens,
                            int pos_offset, float *logits_out, float *all_logits) {
    const fo_model_cfg *g = &m->cfg;
    if (cache_id < 0 || cache_id >= FO_MAX_CACHES) return -4;
    fo_kv *cache = fo_get_cache(m, cache_id);
    if (cache->len != pos_offset) return -2;
    if (pos_offset + n_tokens > g->max_seq_len) return -3;
    m->last_route_gap = INFINITY;
    m->last_route_gap_rel = INFINITY;
    int T = n_tokens, H = g->hidden, nh = g->num_heads, nkv = g->num_kv_heads, hd = g->head_dim;
    int q_dim = nh * hd, kv_dim = nkv * hd, qkv_dim = q_dim + 2 * kv_dim;
    int I = g->intermediate;
    int qk_mode = g->has_qk_norm ? 1 : 2;
    float scale = 1.0f / sqrtf((float)hd);

    float *residual = (float *)malloc(sizeof(float) * (long)T * H);
    float *norm_out = (float *)malloc(sizeof(float) * (long)T * H);
    float *qkv = (float *)malloc(sizeof(float) * (long)T * qkv_dim);
    float *q_buf = (float *)malloc(sizeof(float) * (long)T * q_dim);
    float *k_buf = (float *)malloc(sizeof(float) * (long)T * kv_dim);
    float *v_buf = (float *)malloc(sizeof(float) * (long)T * kv_dim);
    float *q_hm = (float *)malloc(sizeof(float) * (long)T * q_dim);
    float *k_hm = (float *)malloc(sizeof(float) * (long)T * kv_dim);
    float *v_hm = (float *)malloc(sizeof(float) * (long)T * kv_dim);
    float *attn_hm = (float *)calloc((long)T * q_dim, sizeof(float));
    float *attn_tm = (float *)malloc(sizeof(float) * (long)T * q_dim);
    float *o_out = (float *)malloc(sizeof(float) * (long)T * H);
    float *mlp_out = (float *)malloc(sizeof(float) * (long)T * H);
    int Imax = g->num_experts > 0 ? g->expert_inter : I;
    float *gate_up = (float *)malloc(sizeof(float) * (long)T * 2 * (Imax > 0 ? Imax : 1));
    float *act = (float *)malloc(sizeof(float) * (long)T * (Imax > 0 ? Imax : 1));
    float *router_logits = NULL; uint32_t *eids = NULL; float *ew = NULL;
    if (g->num_experts > 0) {
        router_logits = (float *)malloc(sizeof(float) * (long)T * g->num_experts);
        eids = (uint32_t *)malloc(sizeof(uint32_t) * (long)T * g->top_k);
        ew = (float *)malloc(sizeof(float) * (long)T * g->top_k);
    }

    fo_embedding_lookup(m->embed, tokens, T, residual, H);
    if (g->embed_scale != 0.0f) fo_scale_inplace(residual, g->embed_scale, (long)T * H);   /* llama_family.rs:3656 */
    float *branch = g->sandwich_norms ? (float *)malloc(sizeof(float) * (long)T * H) : NULL;

    for (int li = 0; li < g->num_layers; li++) {
        fo_layer *L = &m->layers[li];
        long layer_kv = (long)li * nkv * g->max_seq_len * hd;
        const float *dummy = L->input_ln;
        const float *qn = L->q_norm ? L->q_norm : dummy;
        const float *kn = L->k_norm ? L->k_norm : dummy;

        /* per-layer attention schedule (llama_layer_attention_schedule, llama_family.rs:1028-1045) */
        const int pattern = g->sliding_window_pattern;
        const int is_global = pattern == 0 || (li + 1) % pattern == 0;
        const int layer_window = pattern == 0 ? g->sliding_window : (is_global ? 0 : g->sliding_window);
        const float *cos_l = (!is_global && m->cos_local) ? m->cos_local : m->cos_t;
        const float *sin_l = (!is_global && m->sin_local) ? m->sin_local : m->sin_t;

        fo_rms_norm(residual, L->input_ln, g->rms_eps, norm_out, T, H);
        fo_gemm(norm_out, L->qkv_w, qkv, T, qkv_dim, H);
        if (L->qkv_bias)
            for (long t = 0; t < T; t++)
                for (int j = 0; j < qkv_dim; j++) qkv[t * qkv_dim + j] += L->qkv_bias[j];
        /* kv_layer.rs:437-495 unfused contig_write chain */
        fo_split_qkv(qkv, q_buf, k_buf, v_buf, T, q_dim, kv_dim);
        fo_qk_norm_rope(q_buf, qn, cos_l, sin_l, q_hm, T, nh, hd, pos_offset, g->rms_eps, qk_mode);
        fo_qk_norm_rope(k_buf, kn, cos_l, sin_l, k_hm, T, nkv, hd, pos_offset, g->rms_eps, qk_mode);
        fo_qk_norm_rope(v_buf, qn, cos_l, sin_l, v_hm, T, nkv, hd, pos_offset, g->rms_eps, 0);
        fo_kv_cache_append_head_major(cache->k + layer_kv, cache->v + layer_kv, cache->len,
                                      g->max_seq_len, k_hm, v_hm, T, nkv, hd);
        int new_len = cache->len + T;
        memset(attn_hm, 0, sizeof(float) * (long)T * q_dim);
        fo_cpu_attention(q_hm, cache->k + layer_kv, cache->v + layer_kv, attn_hm, T, new_len,
                         1, pos_offset, nh, nkv, hd, scale, g->max_seq_len, layer_window);
        const float *attn_in = attn_hm;
        if (T > 1) {
            fo_transpose_head_to_token(attn_hm, attn_tm, T, nh, hd);
            attn_in = attn_tm;
        }
        fo_gemm(attn_in, L->o_w, o_out, T, H, q_dim);
        if (g->sandwich_norms) {
            /* llama_family.rs:3357-3436 (host path): norm the attention output FIRST, add, then the pre-MLP norm */
            fo_rms_norm(o_out, L->post_attn_ln, g->rms_eps, branch, T, H);
            fo_add_inplace(residual, branch, (long)T * H);
            fo_rms_norm(residual, L->post_ln, g->rms_eps, norm_out, T, H);
        } else {
            fo_fused_add_rms_norm(residual, o_out, L->post_ln, g->rms_eps, norm_out, T, H);
        }

        if (g->num_experts > 0) {
            fo_gemm(norm_out, L->router_w, router_logits, T, g->num_experts, H);
            fo_route_topk(router_logits, T, g->num_experts, g->top_k, g->norm_topk_prob, eids, ew);
            for (int t = 0; t < T; t++) {
                float gap = fo_topk_gap(router_logits + (long)t * g->num_experts, g->num_experts, g->top_k);
                if (gap < m->last_route_gap) m->last_route_gap = gap;
                const float *rl = router_logits + (long)t * g->num_experts;
                float lo = rl[0], hi = rl[0];
                for (int e = 1; e < g->num_experts; e++) { if (rl[e] < lo) lo = rl[e]; if (rl[e] > hi) hi = rl[e]; }
                float rel = hi > lo ? gap / (hi - lo) : INFINITY;
                if (rel < m->last_route_gap_rel) m->last_route_gap_rel = rel;
            }
            fo_moe_forward_cpu(norm_out, T, H, g->expert_inter, g->top_k, eids, ew,
                               L->exp_gate_up_w, L->exp_down_w, mlp_out);
        } else {
            fo_gemm(norm_out, L->gate_up_w, gate_up, T, 2 * I, H);
            if (g->activation == 1) fo_fused_gelu_tanh_mul_split(gate_up, act, T, I);
            else fo_fused_silu_mul_split(gate_up, act, T, I);
            fo_gemm(act, L->down_w, mlp_out, T, H, I);
        }
        if (g->sandwich_norms) {   /* post_feedforward_layernorm wraps the MLP output before its residual add */
            fo_rms_norm(mlp_out, L->post_ffn_ln, g->rms_eps, branch, T, H);
            fo_add_inplace(residual, branch, (long)T * H);
        } else {
            fo_add_inplace(residual, mlp_out, (long)T * H);
        }
        if (m->tap_hidden && T <= m->tap_tokens)
            memcpy(m->tap_hidden + (long)li * m->tap_tokens * H, residual, sizeof(float) * (long)T * H);
    }
    cache->len += T;

    fo_rms_norm(residual, m->final_norm, g->rms_eps, norm_out, T, H);
    const float *head = m->lm_head ? m->lm_head : m->embed;
    if (all_logits) fo_gemm(norm_out, head, all_logits, T, g->vocab, H);
    if (logits_out) fo_gemm(norm_out + (long)(T - 1) * H, head, logits_out, 1, g->vocab, H);

    free(residual); free(norm_out); free(qkv); free(q_buf); free(k_buf); free(v_buf);
    free(q_hm); free(k_hm); free(v_hm); free(attn_hm); free(attn_tm); free(o_out);
    free(mlp_out); free(gate_up); free(act); free(router_logits); free(eids); free(ew); free(branch);
    return 0;
}
