// Decoder runner (C-ABI `ferrum_hip_model_*`): weights, paged-KV admission, the unified
// prefill+decode forward and the steady-state decode loop with hipGraph replay.
//
// Op order per layer follows the reference's unified forward
//   LlamaFamilyModel::unified_forward_layer (ferrum-models/src/models/llama_family_forward_batched.rs:2625-3090)
//   Qwen3MoeModel::unified_forward_layer    (models/qwen3_moe_forward_unified_layer.rs:46-455)
//   moe_forward_bucketed                    (moe/dispatch.rs:1558-2192)
// and the drivers unified_forward_internal (models/qwen3_moe_forward_unified.rs:174-445):
//   embed → L × { rms_norm → qkv → split/qk-norm/rope/paged-write → paged attention → o_proj →
//   fused_add_rms_norm → [gate_up → act·mul → down | router → top-k → align → grouped gate_up(+silu·mul)
//   → grouped down → combine] → residual add } → final rms_norm (sampled rows) → lm_head → argmax.
#include <math.h>

#include "runner.h"
#include "knobs.h"
#include "tp_comm.h"

#include <condition_variable>
#include <mutex>

using namespace fh;

namespace {

int upload_f32_as_f16(const float* src, size_t n, __half** dst) {
    std::vector<uint16_t> tmp(n);
    for (size_t i = 0; i < n; i++) {
        _Float16 h = (_Float16)src[i];
        memcpy(&tmp[i], &h, 2);
    }
    if (!*dst) FH_CHECK_HIP(hipMalloc((void**)dst, n * 2));
    FH_CHECK_HIP(hipMemcpy(*dst, tmp.data(), n * 2, hipMemcpyHostToDevice));
    return 0;
}

template <typename T>
int dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (!count) return 0;
    FH_CHECK_HIP(hipMalloc((void**)p, count * sizeof(T)));
    FH_CHECK_HIP(hipMemset(*p, 0, count * sizeof(T)));
    return 0;
}

void free_w4(W4Device& w) {
    if (w.qw) (void)hipFree(w.qw);
    if (w.sc) (void)hipFree(w.sc);
    if (w.zp) (void)hipFree(w.zp);
    if (w.perm) (void)hipFree(w.perm);
    if (w.inv_perm) (void)hipFree(w.inv_perm);
    if (w.bias) (void)hipFree(w.bias);
    if (w.f16t) (void)hipFree(w.f16t);
    w = W4Device();
}

int upload_perm(const std::vector<int32_t>& perm, W4Device* w);
int upload_w4(const W4HostPacked& hp, W4Device* w) {
    free_w4(*w);
    w->k = hp.k; w->n = hp.n; w->n64 = hp.n64; w->G = hp.G; w->num_experts = 1;
    FH_CHECK_HIP(hipMalloc((void**)&w->qw, hp.qw.size() * 4));
    FH_CHECK_HIP(hipMemcpy(w->qw, hp.qw.data(), hp.qw.size() * 4, hipMemcpyHostToDevice));
    FH_CHECK_HIP(hipMalloc((void**)&w->sc, hp.sc.size() * 2));
    FH_CHECK_HIP(hipMemcpy(w->sc, hp.sc.data(), hp.sc.size() * 2, hipMemcpyHostToDevice));
    if (!hp.symmetric) {
        FH_CHECK_HIP(hipMalloc((void**)&w->zp, hp.zp.size() * 2));
        FH_CHECK_HIP(hipMemcpy(w->zp, hp.zp.data(), hp.zp.size() * 2, hipMemcpyHostToDevice));
    }
    return hp.perm.empty() ? 0 : upload_perm(hp.perm, w);
}

// input-column permutation of an act-order projection and its inverse (producers that scatter use the inverse)
int upload_perm(const std::vector<int32_t>& perm, W4Device* w) {
    std::vector<int32_t> inv(perm.size());
    for (size_t j = 0; j < perm.size(); j++) inv[perm[j]] = (int32_t)j;
    FH_CHECK_HIP(hipMalloc((void**)&w->perm, perm.size() * 4));
    FH_CHECK_HIP(hipMemcpy(w->perm, perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
    FH_CHECK_HIP(hipMalloc((void**)&w->inv_perm, inv.size() * 4));
    FH_CHECK_HIP(hipMemcpy(w->inv_perm, inv.data(), inv.size() * 4, hipMemcpyHostToDevice));
    return 0;
}

// fused gate_up column order: supertile st = [gate 32st.. | up I+32st..]
std::vector<int32_t> gate_up_col_perm(int n) {
    std::vector<int32_t> p(n);
    const int I = n / 2;
    for (int st = 0; st < n / 64; st++)
        for (int c = 0; c < 64; c++) p[st * 64 + c] = c < 32 ? st * 32 + c : I + st * 32 + (c - 32);
    return p;
}

// RoPE table (llama_family.rs:5220-5282): angle in f64, stored f32.
double rope_freq(const FerrumHipModelConfig& c, int i) {
    double base = 1.0 / pow(c.rope_theta, (double)(2 * i) / (double)c.head_dim);
    if (c.rope_scaling_kind == 1) return base / c.rope_p0;
    if (c.rope_scaling_kind == 2) {
        double wavelen = 2.0 * M_PI / base;
        double low_wl = c.rope_p3 / c.rope_p1, high_wl = c.rope_p3 / c.rope_p2;
        if (wavelen < high_wl) return base;
        if (wavelen > low_wl) return base / c.rope_p0;
        double smooth = (c.rope_p3 / wavelen - c.rope_p1) / (c.rope_p2 - c.rope_p1);
        return (1.0 - smooth) * base / c.rope_p0 + smooth * base;
    }
    return base;
}

// ── tiny device helpers of the decode loop ──────────────────────────────────
// After a decode step: record the sampled tokens, feed them back as the next inputs, advance the
// per-sequence positions.  Everything the next step needs lives in device buffers, so one captured
// step can be replayed (the reference keeps per-iter state in buffers for the same reason,
// kernels/split_qkv_norm_rope_into_paged_cache.cu:9-16).
__global__ void decode_advance_kernel(const uint32_t* __restrict__ sampled, uint32_t* __restrict__ tokens,
                                      uint32_t* __restrict__ pos_offsets, uint32_t* __restrict__ kv_lens,
                                      uint32_t* __restrict__ history, int32_t* __restrict__ step_counter, int n) {
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    int step = *step_counter;
    if (i < n) {
        uint32_t t = sampled[i];
        tokens[i] = t;
        pos_offsets[i] += 1;
        kv_lens[i] += 1;
        history[(long)step * n + i] = t;
    }
    __syncthreads();
    if (i == 0) *step_counter = step + 1;
}

__global__ void f16_to_f32_kernel(const __half* __restrict__ src, float* __restrict__ dst, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = __half2float(src[i]);
}

// synthetic weights: counter-based hash RNG (splitmix64), reproducible for a given seed
__device__ __forceinline__ uint64_t splitmix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// packed INT4 words with zero-mean codes: nibble 0 is remapped to 8, so q − 8 ∈ [−7, 7] has mean 0
// (a non-zero mean would give every GEMM output a token-independent common mode and collapse the
// synthetic router onto a handful of experts — not the traffic pattern of a trained model).
__global__ void synth_u32_kernel(uint32_t* p, long n, uint64_t seed) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w = (uint32_t)(splitmix(seed + (uint64_t)i) >> 16);
    uint32_t zero_nibbles = ~(w | (w >> 1) | (w >> 2) | (w >> 3)) & 0x11111111u;   // 1 where a nibble is 0
    p[i] = w | (zero_nibbles << 3);
}
__global__ void synth_f16_uniform_kernel(__half* p, long n, uint64_t seed, float lo, float hi) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float u = (float)(splitmix(seed + (uint64_t)i) >> 40) / 16777216.0f;
        p[i] = __float2half(lo + u * (hi - lo));
    }
}
__global__ void synth_f16_normal_kernel(__half* p, long n, uint64_t seed, float std) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        uint64_t r = splitmix(seed + (uint64_t)i);
        float u1 = ((float)(r >> 40) + 0.5f) / 16777216.0f, u2 = (float)((r >> 16) & 0xFFFFFF) / 16777216.0f;
        p[i] = __float2half(std * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2));
    }
}
__global__ void fill_f16_kernel(__half* p, long n, float v) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = __float2half(v);
}

int check_shape(const char* what, int k, int n, int ek, int en) {
    FH_REQUIRE(k == ek && n == en, "model_set_gptq(%s): got K=%d N=%d, config expects K=%d N=%d", what, k, n, ek, en);
    return 0;
}

}  // namespace

namespace {
// Σ over ranks in rank order, fp32 accumulate, one rounding — every rank computes the same bits
__global__ void loopback_sum_kernel(const __half* const* bufs, int world, __half* out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int r = 0; r < world; r++) s += __half2float(bufs[r][i]);
    out[i] = __float2half(s);
}
}  // namespace

// In-process stand-in for the RCCL communicator (tests): the ranks of one tensor-parallel group are runner models driven
// by threads of ONE process on one GPU; an all-reduce is two host barriers around a device-side sum.  It validates the
// runner's sharded forward (column-parallel qkv / gate_up, row-parallel o / down, kv heads split) end to end where only
// one GPU exists; production uses ferrum_hip_model_tp_init (RCCL over xGMI).
struct FerrumHipTpLoopback {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    const __half* bufs[8] = {};
    const __half** bufs_dev = nullptr;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const long gen = generation;
        if (++arrived == world) { arrived = 0; generation++; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};

namespace {
int loopback_all_reduce(FerrumHipModel* m, __half* buf, size_t count) {
    FerrumHipTpLoopback* lb = m->tp_loopback;
    if (m->tp_tmp_elems < count) {
        if (m->tp_tmp) (void)hipFree(m->tp_tmp);
        FH_CHECK_HIP(hipMalloc((void**)&m->tp_tmp, count * 2));
        m->tp_tmp_elems = count;
    }
    FH_CHECK_HIP(hipStreamSynchronize(m->stream));             // this rank's partial is complete
    { std::lock_guard<std::mutex> lk(lb->mu); lb->bufs[m->cfg.tp_rank] = buf; }
    lb->barrier();                                             // every partial is complete and published
    if (m->cfg.tp_rank == 0)
        FH_CHECK_HIP(hipMemcpy((void*)lb->bufs_dev, lb->bufs, sizeof(void*) * lb->world, hipMemcpyHostToDevice));
    lb->barrier();
    hipLaunchKernelGGL(loopback_sum_kernel, dim3(cdiv((long)count, 256)), dim3(256), 0, m->stream, lb->bufs_dev, lb->world, m->tp_tmp,
                       (long)count);
    FH_CHECK_LAUNCH();
    FH_CHECK_HIP(hipStreamSynchronize(m->stream));
    lb->barrier();                                             // everyone has read every partial
    FH_CHECK_HIP(hipMemcpyAsync(buf, m->tp_tmp, count * 2, hipMemcpyDeviceToDevice, m->stream));
    return 0;
}

// small all-gather through the host-barrier loopback (tests): out[world][bytes] in rank order
int loopback_all_gather(FerrumHipModel* m, const void* in, void* out, size_t bytes) {
    FerrumHipTpLoopback* lb = m->tp_loopback;
    FH_CHECK_HIP(hipStreamSynchronize(m->stream));
    { std::lock_guard<std::mutex> lk(lb->mu); lb->bufs[m->cfg.tp_rank] = reinterpret_cast<const __half*>(in); }
    lb->barrier();
    for (int p = 0; p < lb->world; p++)
        FH_CHECK_HIP(hipMemcpyAsync((uint8_t*)out + (size_t)p * bytes, lb->bufs[p], bytes, hipMemcpyDeviceToDevice, m->stream));
    FH_CHECK_HIP(hipStreamSynchronize(m->stream));
    lb->barrier();
    return 0;
}
int tp_all_gather(FerrumHipModel* m, const void* in, void* out, size_t bytes) {
    if (m->tp_loopback) { form_hit(FORM_TP_ALLREDUCE_LOOPBACK); return loopback_all_gather(m, in, out, bytes); }
    FH_REQUIRE(m->comm, "vocabulary parallel: no communicator (ferrum_hip_model_tp_init / ferrum_hip_model_set_comm)");
    return comm_all_gather_bytes(m->comm, in, out, bytes, m->stream);
}

int tp_all_reduce(FerrumHipModel* m, __half* buf, size_t count) {
    if (m->cfg.tp_world <= 1) return 0;
    if (m->tp_loopback) { form_hit(FORM_TP_ALLREDUCE_LOOPBACK); return loopback_all_reduce(m, buf, count); }
    FH_REQUIRE(m->comm, "tensor parallel: no communicator (ferrum_hip_model_tp_init / ferrum_hip_model_set_comm)");
    return comm_all_reduce_f16(m->comm, buf, count, m->stream);
}

// residual += all_reduce(x); out = rms_norm(residual)·w (tp_decode.rs:350-372 + the layer's add + norm): ONE launch where the
// one-shot transport carries the message, else the all-reduce and fused_add_rms_norm_f16 — the same bits either way.
int tp_all_reduce_add_rms_norm(FerrumHipModel* m, __half* x, __half* residual, const __half* w, float eps, __half* out, int T, int H,
                               hipStream_t s, const int32_t* out_perm) {
    if (m->cfg.tp_world > 1 && !m->tp_loopback && m->comm && !out_perm) {
        int fused = 0;
        if (int rc = comm_all_reduce_add_rms_norm_f16(m->comm, x, residual, w, eps, out, T, H, &fused, s)) return rc;
        if (fused) return 0;
    }
    if (int rc = tp_all_reduce(m, x, (size_t)T * H)) return rc;
    return fused_add_rms_norm_f16(residual, x, w, eps, out, T, H, s, out_perm);
}
}  // namespace

static void drop_graph(FerrumHipModel* m) {
    if (m->graph_exec) { (void)hipGraphExecDestroy(m->graph_exec); m->graph_exec = nullptr; }
    if (m->graph) { (void)hipGraphDestroy(m->graph); m->graph = nullptr; }
}

// Merged launches hand data over between workgroups and bound every wait; a wait that gave up has produced garbage.  Called
// after every host synchronisation of a forward: the call fails, the captured graph is dropped and the merged forms stay off.
static int check_inlaunch_waits(FerrumHipModel* m) {
    // tensor / expert / vocabulary parallel: a one-shot all-reduce or all-gather whose wait for a peer gave up left the rank's
    // un-reduced partial in the output — ids sampled from it must not be handed back as a success
    if (const unsigned n = comm_take_timeouts(m->comm)) {
        drop_graph(m);
        fh::set_error("%u one-shot collective call(s) gave up waiting for a peer: the results of this call are invalid; the one-shot "
                      "transport is off from now on (RCCL, where the communicator has a rank)", n);
        return 3;
    }
    if (!m->inlaunch_timeouts) return 0;
    const unsigned now = *reinterpret_cast<volatile unsigned*>(m->inlaunch_timeouts);
    if (now == m->inlaunch_timeouts_seen) return 0;
    const unsigned n = now - m->inlaunch_timeouts_seen;
    m->inlaunch_timeouts_seen = now;
    m->em2_failed = true;
    drop_graph(m);
    fh::set_error("%u in-launch wait(s) of a merged kernel gave up (gate_up → down hand-off): the results of this call are invalid; "
                  "the two-launch form is used from now on", n);
    return 3;
}

static int q_dim(const FerrumHipModelConfig& c) { return c.num_heads * c.head_dim; }
static int kv_dim(const FerrumHipModelConfig& c) { return c.num_kv_heads * c.head_dim; }
static int qkv_dim(const FerrumHipModelConfig& c) { return q_dim(c) + 2 * kv_dim(c); }

extern "C" {

int ferrum_hip_model_create(FerrumHipModel** model, const FerrumHipModelConfig* cfg) {
    FH_REQUIRE(model && cfg, "model_create: null argument");
    FH_REQUIRE(cfg->num_layers > 0 && cfg->hidden > 0 && cfg->hidden % 128 == 0, "model_create: hidden=%d must be a multiple of 128", cfg->hidden);
    FH_REQUIRE(cfg->num_kv_heads > 0 && cfg->num_heads % cfg->num_kv_heads == 0, "model_create: heads %d/%d", cfg->num_heads, cfg->num_kv_heads);
    FH_REQUIRE(cfg->num_heads / cfg->num_kv_heads <= 16, "model_create: GQA group %d > 16 unsupported", cfg->num_heads / cfg->num_kv_heads);
    FH_REQUIRE(cfg->head_dim == 64 || cfg->head_dim == 128 || cfg->head_dim == 256, "model_create: head_dim=%d unsupported", cfg->head_dim);
    FH_REQUIRE(cfg->max_seqs > 0 && cfg->max_tokens >= cfg->max_seqs && cfg->kv_num_blocks > 0 && cfg->max_seq_len > 0,
               "model_create: max_seqs/max_tokens/kv_num_blocks/max_seq_len must be positive");
    FH_REQUIRE(cfg->group_size > 0 && cfg->group_size % 128 == 0, "model_create: group_size=%d must be a multiple of 128", cfg->group_size);
    if (cfg->num_experts > 0) {
        FH_REQUIRE(cfg->top_k > 0 && cfg->top_k <= cfg->num_experts && cfg->num_experts <= 512, "model_create: experts=%d top_k=%d", cfg->num_experts, cfg->top_k);
        FH_REQUIRE(cfg->expert_inter % 128 == 0, "model_create: expert_inter=%d must be a multiple of 128", cfg->expert_inter);
    } else {
        FH_REQUIRE(cfg->intermediate > 0 && cfg->intermediate % 128 == 0, "model_create: intermediate=%d must be a multiple of 128", cfg->intermediate);
    }
    FH_REQUIRE(!cfg->sandwich_norms || cfg->num_experts == 0, "model_create: sandwich norms are implemented for dense MLP models");
    FH_REQUIRE(cfg->sliding_window_pattern >= 0 && cfg->rope_local_theta >= 0.0, "model_create: bad local-attention schedule");
    FH_REQUIRE(cfg->expert_parallel >= 0 && cfg->expert_parallel <= 2 && cfg->vocab_parallel >= 0 && cfg->vocab_parallel <= 1,
               "model_create: expert_parallel=%d vocab_parallel=%d", cfg->expert_parallel, cfg->vocab_parallel);
    if (cfg->expert_parallel && cfg->tp_world > 1) {
        FH_REQUIRE(cfg->num_experts > 0 && cfg->num_experts % cfg->tp_world == 0, "model_create: expert_parallel needs num_experts (%d) divisible by the world size (%d)",
                   cfg->num_experts, cfg->tp_world);
    }
    auto* m = new FerrumHipModel();
    m->cfg = *cfg;
    if (m->cfg.tp_world < 1) { m->cfg.tp_world = 1; m->cfg.tp_rank = 0; }
    if (m->cfg.tp_world == 1) { m->cfg.expert_parallel = 0; m->cfg.vocab_parallel = 0; }
    m->vp_n = m->cfg.vocab;
    if (m->cfg.vocab_parallel) {
        const int per = (cdiv(m->cfg.vocab, m->cfg.tp_world) + 15) / 16 * 16;       // whole 16-row tiles per rank
        m->vp_v0 = std::min(m->cfg.vocab, m->cfg.tp_rank * per);
        m->vp_n = std::min(m->cfg.vocab, m->vp_v0 + per) - m->vp_v0;
        FH_REQUIRE(m->vp_n > 0, "model_create: vocab_parallel leaves rank %d of %d without vocabulary rows (V=%d)", m->cfg.tp_rank, m->cfg.tp_world, m->cfg.vocab);
    }
    m->ep_E = m->cfg.num_experts;
    if (m->cfg.expert_parallel) {
        m->ep_E = m->cfg.num_experts / m->cfg.tp_world;
        m->ep_e0 = m->cfg.tp_rank * m->ep_E;
    }
    m->layers.resize(cfg->num_layers);
    m->max_blocks_per_seq = cdiv(cfg->max_seq_len, KV_BLOCK);
    hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete m; fh::set_error("model_create: hipStreamCreate: %s", hipGetErrorString(e)); return 1; }
    // RoPE tables: the main one (θ, optional scaling) and, for Gemma-3 style models, the unscaled local-layer table
    const int half = cfg->head_dim / 2;
    auto build_table = [&](bool local, float** cos_d, float** sin_d) -> bool {
        std::vector<float> cs((size_t)cfg->max_seq_len * half), sn((size_t)cfg->max_seq_len * half);
        std::vector<double> freq(half);
        for (int i = 0; i < half; i++)
            freq[i] = local ? 1.0 / pow(cfg->rope_local_theta, (double)(2 * i) / (double)cfg->head_dim) : rope_freq(*cfg, i);
        for (int pos = 0; pos < cfg->max_seq_len; pos++)
            for (int i = 0; i < half; i++) {
                double ang = (double)pos * freq[i];
                cs[(size_t)pos * half + i] = (float)cos(ang);
                sn[(size_t)pos * half + i] = (float)sin(ang);
            }
        if (hipMalloc((void**)cos_d, cs.size() * 4) != hipSuccess || hipMalloc((void**)sin_d, sn.size() * 4) != hipSuccess) return false;
        (void)hipMemcpy(*cos_d, cs.data(), cs.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(*sin_d, sn.data(), sn.size() * 4, hipMemcpyHostToDevice);
        return true;
    };
    if (!build_table(false, &m->cos_t, &m->sin_t) || (cfg->rope_local_theta > 0.0 && !build_table(true, &m->cos_local, &m->sin_local))) {
        fh::set_error("model_create: rope table allocation failed");
        ferrum_hip_model_destroy(m);
        return 1;
    }
    m->alloc.reset(new BlockAllocator((uint32_t)cfg->kv_num_blocks));
    if (const char* e = getenv("FERRUM_HIP_ROUTE_PARTS")) m->route_parts = std::max(1, atoi(e));
    if (const char* e = getenv("FERRUM_HIP_FUSE_TAIL_ROWS")) m->fuse_tail_max_rows = std::max(0, std::min(4, atoi(e)));
    if (const char* e = getenv("FERRUM_HIP_FUSE_ROPE")) m->fuse_rope_attn = atoi(e) != 0;
    if (const char* e = getenv("FERRUM_HIP_O_SLABS")) m->o_slabs = std::max(0, atoi(e));
    if (const char* e = getenv("FERRUM_HIP_DENSE_SLABS")) m->dense_slabs = atoi(e) != 0;
    if (const char* e = getenv("FERRUM_HIP_ROUTE_GEMM_TOKENS")) m->route_gemm_min_tokens = std::max(1, atoi(e));
    if (const char* e = getenv("FERRUM_HIP_MOE_TILE_PAIRS")) m->moe_tile_min_pairs_per_expert = std::max(1, atoi(e));
    if (const char* e = getenv("FERRUM_HIP_MOE_EM_PAIRS")) m->moe_em_min_pairs_x8 = 8 * std::max(0, atoi(e));
    if (const char* e = getenv("FERRUM_HIP_MOE_EM_PAIRS_X8")) m->moe_em_min_pairs_x8 = std::max(0, atoi(e));
    if (knobs().attn_flash_min_rows_set) m->attn_flash_min_rows = std::max(1, (int)knobs().attn_flash_min_rows);   // as the launcher reads it
    if (const char* e = getenv("FERRUM_HIP_MOE_TILE96_PAIRS")) m->moe_tile96_min_pairs_per_expert = std::max(1, atoi(e));
    if (const char* e = getenv("FERRUM_HIP_MOE_TILE128_PAIRS")) m->moe_tile128_min_pairs_per_expert = std::max(1, atoi(e));
    if (const char* e = getenv("FERRUM_HIP_MOE_TILE32_PAIRS")) m->moe_tile32_min_pairs_per_expert = std::max(1, atoi(e));
    *model = m;
    return 0;
}

int ferrum_hip_model_destroy(FerrumHipModel* m) {
    if (!m) return 0;
    drop_graph(m);
    if (m->comm && m->comm_owned) ferrum_hip_comm_destroy(m->comm);
    for (auto& L : m->layers) {
        for (__half* p : {L.input_ln, L.post_ln, L.q_norm, L.k_norm, L.router, L.k_pool, L.v_pool, L.post_attn_ln, L.post_ffn_ln, L.qkv_bias})
            if (p) (void)hipFree(p);
        free_w4(L.qkv); free_w4(L.o); free_w4(L.gate_up); free_w4(L.down); free_w4(L.exp_gate_up); free_w4(L.exp_down);
    }
    for (void* p : {(void*)m->embed, (void*)m->lm_head, (void*)m->lm_head_t, (void*)m->final_norm, (void*)m->cos_t, (void*)m->sin_t,
                    (void*)m->residual, (void*)m->norm_out, (void*)m->qkv_out, (void*)m->q_out, (void*)m->attn_out,
                    (void*)m->o_out, (void*)m->gate_up_out, (void*)m->act_out, (void*)m->mlp_out,
                    (void*)m->sampled_hidden, (void*)m->moe_act, (void*)m->moe_down, (void*)m->moe_gather_x, (void*)m->moe_gather_h, (void*)m->router_logits,
                    (void*)m->expert_ids, (void*)m->sorted_ids, (void*)m->block_ids, (void*)m->total_post_pad,
                    (void*)m->expert_w, (void*)m->logits, (void*)m->out_tokens, (void*)m->workspace, (void*)m->taps,
                    (void*)m->idx_dev, (void*)m->history, (void*)m->step_counter, (void*)m->residual2, (void*)m->chain_o_part, (void*)m->chain_attn_partial, (void*)m->chain_attn_tickets,
                    (void*)m->route_cand, (void*)m->route_stats, (void*)m->route_arrive, (void*)m->cos_local,
                    (void*)m->sin_local, (void*)m->residual_f32, (void*)m->gather_scratch, (void*)m->greedy_opts_dev, (void*)m->tp_tmp,
                    (void*)m->expert_ids_local, (void*)m->ones, m->vp_pairs, m->vp_gathered})
        if (p) (void)hipFree(p);
    if (m->idx_host) (void)hipHostFree(m->idx_host);
    if (m->inlaunch_timeouts) (void)hipHostFree(m->inlaunch_timeouts);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
    return 0;
}

int ferrum_hip_model_set_global_f32(FerrumHipModel* m, int which, const float* data) {
    FH_REQUIRE(m && data, "model_set_global: null argument");
    const size_t vh = (size_t)m->cfg.vocab * m->cfg.hidden;
    switch (which) {
    case 0: return upload_f32_as_f16(data, vh, &m->embed);
    case 1: return upload_f32_as_f16(data, vh, &m->lm_head);
    case 2: return upload_f32_as_f16(data, m->cfg.hidden, &m->final_norm);
    }
    fh::set_error("model_set_global: which=%d", which);
    return FERRUM_HIP_INVALID;
}

int ferrum_hip_model_set_layer_dense_f32(FerrumHipModel* m, int layer, int which, const float* data) {
    FH_REQUIRE(m && data && layer >= 0 && layer < m->cfg.num_layers, "model_set_layer_dense: bad argument");
    LayerWeights& L = m->layers[layer];
    switch (which) {
    case 0: return upload_f32_as_f16(data, m->cfg.hidden, &L.input_ln);
    case 1: return upload_f32_as_f16(data, m->cfg.hidden, &L.post_ln);
    case 2: return upload_f32_as_f16(data, m->cfg.head_dim, &L.q_norm);
    case 3: return upload_f32_as_f16(data, m->cfg.head_dim, &L.k_norm);
    case 4:
        FH_REQUIRE(m->cfg.num_experts > 0, "model_set_layer_dense: router on a dense model");
        return upload_f32_as_f16(data, (size_t)m->cfg.num_experts * m->cfg.hidden, &L.router);
    case 7:   // fused q|k|v projection bias (Qwen2 family, gptq.rs:56): attached to the qkv GEMM at finalize
        return upload_f32_as_f16(data, (size_t)qkv_dim(m->cfg), &L.qkv_bias);
    case 5: return upload_f32_as_f16(data, m->cfg.hidden, &L.post_attn_ln);
    case 6: return upload_f32_as_f16(data, m->cfg.hidden, &L.post_ffn_ln);
    }
    fh::set_error("model_set_layer_dense: which=%d", which);
    return FERRUM_HIP_INVALID;
}

static int ensure_expert_stack(W4Device* w, int k, int n, int E, bool fused) {
    if (w->qw) return 0;
    w->k = k; w->n = n; w->n64 = (n + 63) / 64; w->G = k / 128; w->num_experts = E; w->fused_gate_up = fused;
    size_t qw = (size_t)w->n64 * w->G * 4 * 64 * 4, sc = (size_t)w->n64 * w->G * 16 * 4;
    FH_CHECK_HIP(hipMalloc((void**)&w->qw, qw * 4 * E));
    FH_CHECK_HIP(hipMalloc((void**)&w->sc, sc * 2 * E));
    return 0;
}

// Dense gate_up: when down is act-order (perm P: down reads act'[j] = act[P[j]]), pack column j of the gate half from the
// checkpoint's gate column P[j] and column I + j from up column I + P[j]: the gated activation of the packed output IS act'.
static int pack_gate_up(FerrumHipModel* m, LayerWeights& L, const int32_t* qweight, const float* scales, const int32_t* qzeros,
                        const int32_t* g_idx, int k, int n) {
    W4HostPacked hp;
    std::vector<int32_t> cols;
    const std::vector<int32_t>& P = L.down_perm_host;
    const int I = n / 2;
    const bool fold = !P.empty() && (int)P.size() == I && L.down.qw != nullptr;
    if (fold) {
        cols.resize(n);
        for (int j = 0; j < I; j++) { cols[j] = P[j]; cols[I + j] = I + P[j]; }
    }
    if (int r = w4_repack_host(qweight, scales, qzeros, g_idx, fold ? cols.data() : nullptr, m->cfg.group_size, k, n, &hp)) return r;
    if (int r = upload_w4(hp, &L.gate_up)) return r;
    L.down.perm_folded = fold;
    return 0;
}

int ferrum_hip_model_set_gptq(FerrumHipModel* m, int layer, int which, int expert, const int32_t* qweight,
                              const float* scales, const int32_t* qzeros, const int32_t* g_idx, int k, int n) {
    FH_REQUIRE(m && qweight && scales && qzeros && layer >= 0 && layer < m->cfg.num_layers, "model_set_gptq: bad argument");
    const FerrumHipModelConfig& c = m->cfg;
    LayerWeights& L = m->layers[layer];
    W4HostPacked hp;
    const int H = c.hidden;
    if (which <= 3) {
        int rc = 0;
        W4Device* dst = nullptr;
        switch (which) {
        case 0: rc = check_shape("qkv", k, n, H, qkv_dim(c)); dst = &L.qkv; break;
        case 1: rc = check_shape("o", k, n, q_dim(c), H); dst = &L.o; break;
        case 2: rc = check_shape("gate_up", k, n, H, 2 * c.intermediate); dst = &L.gate_up; break;
        case 3: rc = check_shape("down", k, n, c.intermediate, H); dst = &L.down; break;
        }
        if (rc) return rc;
        if (which == 2 && !L.down_set) {
            // down has not arrived: if it turns out to be act-order, its row permutation becomes the column order of gate_up
            // (the gated activation then leaves already permuted — no gather in front of down) — wait for it
            auto p = std::make_unique<LayerWeights::PendingGptq>();
            const int groups = k / c.group_size;
            p->qweight.assign(qweight, qweight + (size_t)(k / 8) * n);
            p->scales.assign(scales, scales + (size_t)groups * n);
            p->qzeros.assign(qzeros, qzeros + (size_t)groups * (n / 8));
            p->has_g_idx = g_idx != nullptr;
            if (g_idx) p->g_idx.assign(g_idx, g_idx + k);
            p->k = k; p->n = n;
            L.pending_gate_up = std::move(p);
            return 0;
        }
        if (which == 2) return pack_gate_up(m, L, qweight, scales, qzeros, g_idx, k, n);
        if (int r = w4_repack_host(qweight, scales, qzeros, g_idx, nullptr, c.group_size, k, n, &hp)) return r;
        if (int r = upload_w4(hp, dst)) return r;
        if (which == 3) {
            L.down_set = true;
            L.down_perm_host = hp.perm;
            if (L.pending_gate_up) {
                std::unique_ptr<LayerWeights::PendingGptq> p = std::move(L.pending_gate_up);
                return pack_gate_up(m, L, p->qweight.data(), p->scales.data(), p->qzeros.data(), p->has_g_idx ? p->g_idx.data() : nullptr, p->k, p->n);
            }
        }
        return 0;
    }
    FH_REQUIRE(which == 4 || which == 5, "model_set_gptq: which=%d", which);
    FH_REQUIRE(c.num_experts > 0 && expert >= 0 && expert < c.num_experts, "model_set_gptq: expert=%d of %d", expert, c.num_experts);
    if (expert < m->ep_e0 || expert >= m->ep_e0 + m->ep_E) return 0;     // expert parallel: another rank's expert (a loader may offer all of them)
    expert -= m->ep_e0;
    const bool gu = which == 4;
    if (int rc = gu ? check_shape("expert gate_up", k, n, H, 2 * c.expert_inter) : check_shape("expert down", k, n, c.expert_inter, H)) return rc;
    std::vector<int32_t> perm;
    if (gu) perm = gate_up_col_perm(n);
    // act-order (desc_act) expert stacks (cuda/quant.rs:862 ff.): the packed rows are sorted by quant group, and — as the reference
    // takes ONE g_idx for a stack (capabilities.rs:180-189: the experts share their K-axis quantisation) — every expert must
    // bring the same row permutation; the grouped GEMMs then read gathered input rows (moe_gemm_inputs below).
    if (int r = w4_repack_host(qweight, scales, qzeros, g_idx, gu ? perm.data() : nullptr, c.group_size, k, n, &hp)) return r;
    W4Device* w = gu ? &L.exp_gate_up : &L.exp_down;
    std::vector<int32_t>& perm_host = gu ? L.exp_gate_up_perm_host : L.exp_down_perm_host;
    const bool first = L.exp_loaded.empty() || std::none_of(L.exp_loaded.begin(), L.exp_loaded.end(), [&](uint8_t b) { return (b & (gu ? 1 : 2)) != 0; });
    if (int rc = ensure_expert_stack(w, k, n, m->ep_E, gu)) return rc;
    if (first) {
        perm_host = hp.perm;
        if (!hp.perm.empty())
            if (int rc = upload_perm(hp.perm, w)) return rc;
    } else {
        FH_REQUIRE(perm_host == hp.perm, "model_set_gptq: layer %d expert %d %s brings a different g_idx than the experts before it (a stack shares one)",
                   layer, expert + m->ep_e0, gu ? "gate_up" : "down");
    }
    // explicit zero points (asymmetric packs; cuda/quant.rs:795-839 sends those to the vLLM MoE lane): the stack gets a zero-point
    // array as soon as one expert needs it; symmetric experts of such a stack carry the implicit zero 8
    const size_t sc_elems = hp.sc.size();
    if (!hp.symmetric && !w->zp) {
        FH_CHECK_HIP(hipMalloc((void**)&w->zp, sc_elems * 2 * (size_t)m->ep_E));
        std::vector<uint16_t> eight(sc_elems * (size_t)m->ep_E, (uint16_t)0x4800);          // fp16 8.0
        FH_CHECK_HIP(hipMemcpy(w->zp, eight.data(), eight.size() * 2, hipMemcpyHostToDevice));
    }
    FH_CHECK_HIP(hipMemcpy(w->qw + (size_t)expert * hp.qw.size(), hp.qw.data(), hp.qw.size() * 4, hipMemcpyHostToDevice));
    FH_CHECK_HIP(hipMemcpy(w->sc + (size_t)expert * hp.sc.size(), hp.sc.data(), hp.sc.size() * 2, hipMemcpyHostToDevice));
    if (w->zp && !hp.symmetric)
        FH_CHECK_HIP(hipMemcpy(reinterpret_cast<uint16_t*>(w->zp) + (size_t)expert * sc_elems, hp.zp.data(), sc_elems * 2, hipMemcpyHostToDevice));
    if (L.exp_loaded.empty()) L.exp_loaded.assign(m->ep_E, 0);
    L.exp_loaded[expert] |= gu ? 1 : 2;
    return 0;
}

// Unquantised projection (DenseLinear, ferrum-kernels/src/linear.rs:109-129): W [n, k] row-major f32 on the host → fp16
// f16t tiles on the device; the forward then runs it through the fp16 GEMM (B::gemm).  which: 0 qkv, 1 o, 2 gate_up, 3 down.
int ferrum_hip_model_set_dense_f32(FerrumHipModel* m, int layer, int which, const float* weight, int k, int n) {
    FH_REQUIRE(m && weight && !m->finalized && layer >= 0 && layer < m->cfg.num_layers, "model_set_dense: bad argument");
    const FerrumHipModelConfig& c = m->cfg;
    LayerWeights& L = m->layers[layer];
    const int H = c.hidden;
    int rc = 0;
    W4Device* dst = nullptr;
    switch (which) {
    case 0: rc = check_shape("qkv", k, n, H, qkv_dim(c)); dst = &L.qkv; break;
    case 1: rc = check_shape("o", k, n, q_dim(c), H); dst = &L.o; break;
    case 2: rc = check_shape("gate_up", k, n, H, 2 * c.intermediate); dst = &L.gate_up; break;
    case 3: rc = check_shape("down", k, n, c.intermediate, H); dst = &L.down; break;
    default: fh::set_error("model_set_dense: which=%d", which); return FERRUM_HIP_INVALID;
    }
    if (rc) return rc;
    FH_REQUIRE(k % 32 == 0, "model_set_dense: K=%d must be a multiple of 32", k);
    free_w4(*dst);
    dst->k = k; dst->n = n; dst->n64 = (n + 63) / 64; dst->G = k / 128; dst->num_experts = 1;
    __half* row = nullptr;
    if (int r = upload_f32_as_f16(weight, (size_t)n * k, &row)) return r;
    hipError_t e = hipMalloc((void**)&dst->f16t, f16t_elems(n, k) * 2);
    if (e != hipSuccess) { (void)hipFree(row); fh::set_error("model_set_dense: hipMalloc: %s", hipGetErrorString(e)); return 1; }
    rc = f16t_repack(row, dst->f16t, n, k, m->stream);
    (void)hipStreamSynchronize(m->stream);
    (void)hipFree(row);
    return rc;
}

static void launch1d(void (*k)(__half*, long, uint64_t, float, float), __half* p, long n, uint64_t seed, float a, float b, hipStream_t s) {
    hipLaunchKernelGGL(k, dim3(cdiv(n, 256)), dim3(256), 0, s, p, n, seed, a, b);
}

// Synthetic weights generated on the device (bench only: no checkpoints are available offline).
// qweight nibbles uniform, scales uniform [0.01,0.1)·f with f = 1/(0.25·sqrt(K)) so that
// dequantised W has std ≈ 1/sqrt(K) and activations stay O(1) through all layers,
// zero point 8 (sym), norm weights 1, embeddings / lm_head / router N(0, 0.02).
static int synth_w4(W4Device* w, int k, int n, int E, bool fused, uint64_t seed, hipStream_t s) {
    free_w4(*w);
    w->k = k; w->n = n; w->n64 = (n + 63) / 64; w->G = k / 128; w->num_experts = E; w->fused_gate_up = fused;
    long qw = (long)w->n64 * w->G * 4 * 64 * 4 * E, sc = (long)w->n64 * w->G * 16 * 4 * E;
    FH_CHECK_HIP(hipMalloc((void**)&w->qw, qw * 4));
    FH_CHECK_HIP(hipMalloc((void**)&w->sc, sc * 2));
    hipLaunchKernelGGL(synth_u32_kernel, dim3(cdiv(qw, 256)), dim3(256), 0, s, w->qw, qw, seed);
    float f = 1.0f / (0.25f * sqrtf((float)k));   // std(q-8)·mean(scale) ≈ 4.5·0.055
    launch1d(synth_f16_uniform_kernel, w->sc, sc, seed ^ 0x5ca1e5ull, 0.01f * f, 0.1f * f, s);
    FH_CHECK_LAUNCH();
    return 0;
}

int ferrum_hip_model_init_synthetic(FerrumHipModel* m, uint64_t seed) {
    FH_REQUIRE(m && !m->finalized, "model_init_synthetic: bad state");
    const FerrumHipModelConfig& c = m->cfg;
    hipStream_t s = m->stream;
    auto alloc_fill = [&](__half** p, long n, float v) -> int {
        if (!*p) FH_CHECK_HIP(hipMalloc((void**)p, n * 2));
        hipLaunchKernelGGL(fill_f16_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, *p, n, v);
        return 0;
    };
    auto alloc_normal = [&](__half** p, long n, uint64_t sd, float std) -> int {
        if (!*p) FH_CHECK_HIP(hipMalloc((void**)p, n * 2));
        hipLaunchKernelGGL(synth_f16_normal_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, *p, n, sd, std);
        return 0;
    };
    const long vh = (long)c.vocab * c.hidden;
    if (int rc = alloc_normal(&m->embed, vh, seed ^ 0xe0bedull, 0.02f)) return rc;
    if (int rc = alloc_normal(&m->lm_head, vh, seed ^ 0x1a4eadull, 0.02f)) return rc;
    if (int rc = alloc_fill(&m->final_norm, c.hidden, 1.0f)) return rc;
    for (int li = 0; li < c.num_layers; li++) {
        LayerWeights& L = m->layers[li];
        // replicated tensors (router, norms) come from the layer seed; the sharded projections of a tensor-parallel rank
        // additionally mix in the rank, so every rank holds its own slice-shaped weights while embed / lm_head / norms /
        // router agree across ranks (all ranks then sample the same token from identical all-reduced activations)
        const uint64_t ls0 = seed + 0x1000003ull * (uint64_t)(li + 1);
        const uint64_t ls = ls0 + 0x51ed270bull * (uint64_t)(c.tp_world > 1 ? c.tp_rank : 0);
        if (int rc = alloc_fill(&L.input_ln, c.hidden, 1.0f)) return rc;
        if (int rc = alloc_fill(&L.post_ln, c.hidden, 1.0f)) return rc;
        if (c.sandwich_norms) {
            if (int rc = alloc_fill(&L.post_attn_ln, c.hidden, 1.0f)) return rc;
            if (int rc = alloc_fill(&L.post_ffn_ln, c.hidden, 1.0f)) return rc;
        }
        if (c.has_qk_norm) {
            if (int rc = alloc_fill(&L.q_norm, c.head_dim, 1.0f)) return rc;
            if (int rc = alloc_fill(&L.k_norm, c.head_dim, 1.0f)) return rc;
        }
        if (int rc = synth_w4(&L.qkv, c.hidden, qkv_dim(c), 1, false, ls ^ 0x11, s)) return rc;
        if (int rc = synth_w4(&L.o, q_dim(c), c.hidden, 1, false, ls ^ 0x22, s)) return rc;
        if (c.num_experts > 0) {
            if (int rc = alloc_normal(&L.router, (long)c.num_experts * c.hidden, ls0 ^ 0x33, 0.02f)) return rc;
            if (int rc = synth_w4(&L.exp_gate_up, c.hidden, 2 * c.expert_inter, m->ep_E, true, ls ^ 0x44, s)) return rc;
            if (int rc = synth_w4(&L.exp_down, c.expert_inter, c.hidden, m->ep_E, false, ls ^ 0x55, s)) return rc;
            L.exp_loaded.assign(m->ep_E, 3);
        } else {
            if (int rc = synth_w4(&L.gate_up, c.hidden, 2 * c.intermediate, 1, false, ls ^ 0x44, s)) return rc;
            if (int rc = synth_w4(&L.down, c.intermediate, c.hidden, 1, false, ls ^ 0x55, s)) return rc;
        }
    }
    // development (FERRUM_HIP_SYNTH_DESC_ACT=1): give every dense projection an activation-order permutation, as a desc_act GPTQ
    // checkpoint has (BASELINE configs[3]) — the weights are random anyway; what this reproduces is the cost of serving the
    // permuted inputs (producers write them permuted; prefill o_proj keeps one gather launch)
    if (const char* e = getenv("FERRUM_HIP_SYNTH_DESC_ACT"); e && atoi(e)) {
        for (int li = 0; li < c.num_layers; li++) {
            LayerWeights& L = m->layers[li];
            for (W4Device* w : {&L.qkv, &L.o, &L.gate_up, &L.down}) {
                if (!w->qw || w->perm) continue;
                std::vector<int32_t> perm(w->k);
                for (int i = 0; i < w->k; i++) perm[i] = i;
                uint64_t st = seed ^ (0x9e3779b97f4a7c15ull * (uint64_t)(li * 8 + (w - &L.qkv) + 1));
                for (int i = w->k - 1; i > 0; i--) {          // Fisher–Yates with the reference's LCG
                    st = st * 6364136223846793005ull + 1442695040888963407ull;
                    std::swap(perm[i], perm[(int)((st >> 33) % (uint64_t)(i + 1))]);
                }
                if (int rc = upload_perm(perm, w)) return rc;
            }
            // (random weights: a gate_up whose columns were packed in down's row order is just another random matrix)
            L.down.perm_folded = L.down.perm && L.gate_up.qw;
        }
    }
    FH_CHECK_HIP(hipStreamSynchronize(s));
    return 0;
}

int ferrum_hip_model_finalize(FerrumHipModel* m) {
    FH_REQUIRE(m, "model_finalize: null");
    if (m->finalized) return 0;
    const FerrumHipModelConfig& c = m->cfg;
    FH_REQUIRE(m->embed && m->final_norm, "model_finalize: embed / final_norm missing");
    for (LayerWeights& L : m->layers)
        if (L.pending_gate_up) {      // down never arrived as GPTQ (unquantised, or missing: reported below): nothing to fold
            std::unique_ptr<LayerWeights::PendingGptq> p = std::move(L.pending_gate_up);
            if (int rc = pack_gate_up(m, L, p->qweight.data(), p->scales.data(), p->qzeros.data(), p->has_g_idx ? p->g_idx.data() : nullptr, p->k, p->n))
                return rc;
        }
    for (int li = 0; li < c.num_layers; li++) {
        const LayerWeights& L = m->layers[li];
        auto have = [](const W4Device& w) { return w.qw != nullptr || w.f16t != nullptr; };
        FH_REQUIRE(L.input_ln && L.post_ln && have(L.qkv) && have(L.o), "model_finalize: layer %d attention weights missing", li);
        FH_REQUIRE(!c.has_qk_norm || (L.q_norm && L.k_norm), "model_finalize: layer %d q/k norm missing", li);
        FH_REQUIRE(!c.sandwich_norms || (L.post_attn_ln && L.post_ffn_ln), "model_finalize: layer %d sandwich norms missing", li);
        if (m->layers[li].qkv_bias) {    // the GEMM epilogue / split-K reduce adds it (W4Device::bias owns it from here)
            m->layers[li].qkv.bias = m->layers[li].qkv_bias;
            m->layers[li].qkv_bias = nullptr;
        }
        if (c.num_experts > 0) {
            FH_REQUIRE(L.router && L.exp_gate_up.qw && L.exp_down.qw, "model_finalize: layer %d MoE weights missing", li);
            for (int e = 0; e < m->ep_E; e++)
                FH_REQUIRE(L.exp_loaded[e] == 3, "model_finalize: layer %d expert %d incomplete", li, e + m->ep_e0);
        } else {
            FH_REQUIRE(have(L.gate_up) && have(L.down), "model_finalize: layer %d MLP weights missing", li);
        }
    }
    const size_t T = c.max_tokens, S = c.max_seqs, H = c.hidden;
    // dense fp16 weights that are streamed every step → fragment-major tiles (f16t)
    {
        const __half* head = m->lm_head ? m->lm_head : m->embed;     // tied lm_head (llama_family.rs:969-1001)
        // vocabulary parallel: only this rank's rows [vp_v0, vp_v0 + vp_n) become tiles
        FH_CHECK_HIP(hipMalloc((void**)&m->lm_head_t, f16t_elems(m->vp_n, c.hidden) * 2));
        if (int rc = f16t_repack(head + (size_t)m->vp_v0 * c.hidden, m->lm_head_t, m->vp_n, c.hidden, m->stream)) return rc;
        if (m->lm_head) {
            FH_CHECK_HIP(hipStreamSynchronize(m->stream));
            (void)hipFree(m->lm_head);
            m->lm_head = nullptr;
        }
        for (auto& L : m->layers)
            if (L.router) {
                __half* t = nullptr;
                FH_CHECK_HIP(hipMalloc((void**)&t, f16t_elems(c.num_experts, c.hidden) * 2));
                if (int rc = f16t_repack(L.router, t, c.num_experts, c.hidden, m->stream)) return rc;
                FH_CHECK_HIP(hipStreamSynchronize(m->stream));
                (void)hipFree(L.router);
                L.router = t;
            }
    }
    // KV pools, zero-initialised
    for (auto& L : m->layers) {
        size_t elems = (size_t)c.kv_num_blocks * c.num_kv_heads * kv_tile_elems(c.head_dim);
        if (int rc = dev_alloc(&L.k_pool, elems)) return rc;
        if (int rc = dev_alloc(&L.v_pool, elems)) return rc;
    }
    int rc = 0;
    rc |= dev_alloc(&m->residual, T * H);
    if (c.sandwich_norms) rc |= dev_alloc(&m->residual_f32, T * H);
    {
        size_t kmax = 0;     // act-order input gather scratch
        for (const LayerWeights& L : m->layers)
            for (const W4Device* w : {&L.qkv, &L.o, &L.gate_up, &L.down})
                if (w->perm && !w->perm_folded) kmax = std::max(kmax, (size_t)w->k);
        if (kmax) rc |= dev_alloc(&m->gather_scratch, T * kmax);
    }
    rc |= dev_alloc(&m->norm_out, T * H);
    rc |= dev_alloc(&m->qkv_out, T * qkv_dim(c));
    rc |= dev_alloc(&m->q_out, T * q_dim(c));
    rc |= dev_alloc(&m->attn_out, T * q_dim(c));
    rc |= dev_alloc(&m->o_out, T * H);
    rc |= dev_alloc(&m->mlp_out, T * H);
    rc |= dev_alloc(&m->sampled_hidden, S * H);
    rc |= dev_alloc(&m->logits, S * (size_t)m->vp_n);
    if (c.vocab_parallel) {
        FH_CHECK_HIP(hipMalloc(&m->vp_pairs, S * 8));
        FH_CHECK_HIP(hipMalloc(&m->vp_gathered, S * 8 * (size_t)c.tp_world));
    }
    rc |= dev_alloc(&m->out_tokens, S);
    if (c.num_experts > 0) {
        const size_t P = T * c.top_k, sorted_max = P + (size_t)c.num_experts * 128;  // room for 128-row blocks (prefill)
        rc |= dev_alloc(&m->router_logits, T * c.num_experts);
        rc |= dev_alloc(&m->expert_ids, P);
        if (c.expert_parallel) {
            rc |= dev_alloc(&m->expert_ids_local, P);
            if (!rc) {
                FH_CHECK_HIP(hipMalloc((void**)&m->ones, T * sizeof(float)));
                std::vector<float> one(T, 1.0f);
                FH_CHECK_HIP(hipMemcpy(m->ones, one.data(), T * sizeof(float), hipMemcpyHostToDevice));
            }
        }
        rc |= dev_alloc(&m->expert_w, P);
        rc |= dev_alloc(&m->sorted_ids, sorted_max);
        rc |= dev_alloc(&m->block_ids, sorted_max / 16 + 1);
        rc |= dev_alloc(&m->total_post_pad, (size_t)4);
        rc |= dev_alloc(&m->residual2, (size_t)128 * H);
        rc |= dev_alloc(&m->route_cand, (size_t)512 * 8);     // [T ≤ 64][Q ≤ 8][8]
        rc |= dev_alloc(&m->route_stats, (size_t)512 * 2);
        // [64] arrival counters of the split route kernel (zeroed here; the kernel re-arms them) + [2][E] per-expert counters of the
        // merged gate_up → down launch + its give-up count; the head kernel of every forward zeroes the route counters and half 0
        // layout: route [64] | pair half 0 | chain half 0 | pair half 1 | chain half 1
        m->arrive_half_words = (size_t)c.num_experts * MOE_PAIR_COUNTER_STRIDE + decode_chain_counter_words();
        rc |= dev_alloc(&m->route_arrive, (size_t)64 + 2 * m->arrive_half_words + 4);
        if (!rc) m->em2_arrive = m->route_arrive + 64;
        if (!rc) {
            FH_CHECK_HIP(hipHostMalloc((void**)&m->inlaunch_timeouts, 64, hipHostMallocDefault));
            *m->inlaunch_timeouts = 0u;
        }
        rc |= dev_alloc(&m->moe_act, P * c.expert_inter);
        rc |= dev_alloc(&m->moe_down, P * H);
        bool gx_perm = false, dn_perm = false;                  // act-order expert stacks: gathered GEMM inputs
        for (const LayerWeights& L : m->layers) { gx_perm |= L.exp_gate_up.perm != nullptr; dn_perm |= L.exp_down.perm != nullptr; }
        if (gx_perm) rc |= dev_alloc(&m->moe_gather_x, T * H);
        if (dn_perm) rc |= dev_alloc(&m->moe_gather_h, P * c.expert_inter);
    } else {
        rc |= dev_alloc(&m->gate_up_out, T * 2 * c.intermediate);
        rc |= dev_alloc(&m->act_out, T * (size_t)c.intermediate);
        // the one-launch attention half of the decode layer (chain.hip) for dense models: ping-pong residual, its two counter halves
        // (layout as above without the per-expert part) and the give-up word
        rc |= dev_alloc(&m->residual2, (size_t)128 * H);
        m->arrive_half_words = (size_t)decode_chain_counter_words();
        rc |= dev_alloc(&m->route_arrive, (size_t)64 + 2 * m->arrive_half_words + 4);
        if (!rc) m->em2_arrive = m->route_arrive + 64;
        if (!rc) {
            FH_CHECK_HIP(hipHostMalloc((void**)&m->inlaunch_timeouts, 64, hipHostMallocDefault));
            *m->inlaunch_timeouts = 0u;
        }
    }
    if (!rc && m->em2_arrive) rc |= dev_alloc(&m->chain_o_part, (size_t)4 * 128 * H);
    if (!rc && m->em2_arrive && c.head_dim == 128) {
        rc |= dev_alloc(&m->chain_attn_partial, (size_t)CHAIN_MAX_ATTN_WGS * 16 * (c.head_dim + 4));
        rc |= dev_alloc(&m->chain_attn_tickets, (size_t)4096);
    }
    if (rc) return rc;
    // workspace: split-K slabs (≤ 64 rows) and split-KV attention partials
    size_t widest = std::max<size_t>({(size_t)qkv_dim(c), (size_t)H, (size_t)2 * std::max(c.intermediate, 1), (size_t)c.vocab});
    size_t gemm_ws = (size_t)64 * 64 * ((widest + 63) / 64 * 64) * sizeof(float);
    size_t attn_ws = (size_t)32 * S * c.num_kv_heads * 16 * (c.head_dim + 2) * sizeof(float) * 2;
    m->workspace_bytes = std::max(gemm_ws, attn_ws);
    FH_CHECK_HIP(hipMalloc((void**)&m->workspace, m->workspace_bytes));
    // index block
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    m->il.tokens = take(T * 4);
    m->il.cu_seqlens = take((S + 1) * 4);
    m->il.pos_offsets = take(S * 4);
    m->il.kv_lens = take(S * 4);
    m->il.sampled_idx = take(S * 4);
    m->il.block_tables = take(S * m->max_blocks_per_seq * 4);
    m->il.total = off;
    FH_CHECK_HIP(hipHostMalloc((void**)&m->idx_host, off, hipHostMallocDefault));
    FH_CHECK_HIP(hipMalloc((void**)&m->idx_dev, off));
    FH_CHECK_HIP(hipMemset(m->idx_dev, 0, off));
    if (int r = dev_alloc(&m->step_counter, (size_t)4)) return r;
    m->finalized = true;
    return 0;
}

int ferrum_hip_model_kv_capacity_snapshot(const FerrumHipModel* m, FerrumHipKvSlotReservation* out) {
    FH_REQUIRE(m && out, "kv_capacity_snapshot: null");
    out->block_size = KV_BLOCK;
    out->total_blocks = (int)m->alloc->capacity();
    out->free_blocks_before = out->free_blocks_after = (int)m->alloc->free_count();
    return 0;
}

// reserve_paged_kv_slots (llama_family.rs:2255) + PagedSeqState::ensure_capacity (paged_pool.rs:416-442)
static int reserve(FerrumHipModel* m, const FerrumHipKvSlotRequest* reqs, int n, FerrumHipKvSlotReservation* out) {
    const int before = (int)m->alloc->free_count();
    long need = 0;
    for (int i = 0; i < n; i++) {
        int blocks = cdiv(reqs[i].target_len, KV_BLOCK);
        if (blocks > m->max_blocks_per_seq) {
            fh::set_error("paged KV: target_len=%d would need %d blocks, exceeds max_blocks_per_seq=%d", reqs[i].target_len, blocks, m->max_blocks_per_seq);
            return FERRUM_HIP_INVALID;
        }
        auto it = m->seqs.find(reqs[i].seq_id);
        int have = it == m->seqs.end() ? 0 : (int)it->second.blocks.size();
        if (blocks > have) need += blocks - have;
    }
    if (need > before) {
        fh::set_error("paged KV pool exhausted: need %ld blocks but only %d free", need, before);
        return FERRUM_HIP_INVALID;
    }
    for (int i = 0; i < n; i++) {
        SeqState& st = m->seqs[reqs[i].seq_id];
        int blocks = cdiv(reqs[i].target_len, KV_BLOCK);
        while ((int)st.blocks.size() < blocks) {
            uint32_t b;
            m->alloc->allocate(&b);
            st.blocks.push_back(b);
        }
    }
    if (out) {
        out->block_size = KV_BLOCK;
        out->total_blocks = (int)m->alloc->capacity();
        out->free_blocks_before = before;
        out->free_blocks_after = (int)m->alloc->free_count();
    }
    return 0;
}

int ferrum_hip_model_reserve_kv_slots(FerrumHipModel* m, const FerrumHipKvSlotRequest* reqs, int n,
                                      FerrumHipKvSlotReservation* out) {
    FH_REQUIRE(m && (n == 0 || reqs), "reserve_kv_slots: null");
    return reserve(m, reqs, n, out);
}

int ferrum_hip_model_release(FerrumHipModel* m, uint64_t seq_id) {
    FH_REQUIRE(m, "model_release: null");
    auto it = m->seqs.find(seq_id);
    if (it == m->seqs.end()) return 0;
    m->alloc->free(it->second.blocks.data(), (uint32_t)it->second.blocks.size());
    m->seqs.erase(it);
    return 0;
}

// ── block-level prefix cache (ferrum-models/src/models/qwen3_moe/prefix_cache.rs:66-235, prefill_decode.rs:10-70) ──
int ferrum_hip_block_hash_chain(const uint32_t* tokens, int n, int block_size, uint64_t* out, int capacity, int* count) {
    FH_REQUIRE((tokens || n == 0) && block_size > 0 && count, "block_hash_chain: bad argument");
    std::vector<uint64_t> h = fh::block_hash_chain(tokens, n, block_size);
    *count = (int)h.size();
    for (int i = 0; i < (int)h.size() && i < capacity; i++) out[i] = h[i];
    return 0;
}
uint64_t ferrum_hip_siphash(int c_rounds, int d_rounds, uint64_t k0, uint64_t k1, const uint8_t* data, size_t len) {
    return fh::siphash(c_rounds, d_rounds, k0, k1, data, len);
}

int ferrum_hip_model_prefix_cache_acquire(FerrumHipModel* m, uint64_t seq_id, const uint32_t* tokens, int n, int* cached_tokens) {
    FH_REQUIRE(m && tokens && n > 0 && cached_tokens, "prefix_cache_acquire: bad argument");
    *cached_tokens = 0;
    auto it = m->seqs.find(seq_id);
    FH_REQUIRE(it != m->seqs.end(), "prefix_cache_acquire: reserve KV slots for sequence %llu first", (unsigned long long)seq_id);
    SeqState& st = it->second;
    FH_REQUIRE(st.len == 0, "prefix_cache_acquire: sequence %llu already holds %d tokens", (unsigned long long)seq_id, st.len);
    const std::vector<uint64_t> hashes = fh::block_hash_chain(tokens, n, KV_BLOCK);
    // longest contiguous prefix: stop at the first miss
    std::vector<uint32_t> matched;
    for (uint64_t h : hashes) {
        if (matched.size() >= st.blocks.size()) break;
        int64_t b = m->alloc->try_acquire_by_hash(h);
        if (b < 0) break;
        matched.push_back((uint32_t)b);
    }
    if (matched.empty()) {
        m->prefix_misses++;
        return 0;
    }
    // the freshly reserved blocks in the matched slots go back to the pool; the cached ids take their place
    m->alloc->free(st.blocks.data(), (uint32_t)matched.size());
    for (size_t i = 0; i < matched.size(); i++) st.blocks[i] = matched[i];
    int cached = (int)matched.size() * KV_BLOCK;
    // a full hit keeps one block's worth to re-run so the forward still yields final logits (prefill_decode.rs:34-48);
    // the suffix then rewrites the shared block with identical content
    if (cached >= n) cached = std::min(std::max(cached - KV_BLOCK, 0), n - 1);
    st.len = cached;
    if (cached > 0) { m->prefix_hits++; m->prefix_saved_tokens += (uint64_t)cached; } else { m->prefix_misses++; }
    *cached_tokens = cached;
    return 0;
}

int ferrum_hip_model_prefix_cache_register(FerrumHipModel* m, uint64_t seq_id, const uint32_t* all_tokens, int n,
                                           int prior_cached_tokens) {
    FH_REQUIRE(m && all_tokens && n >= 0 && prior_cached_tokens >= 0, "prefix_cache_register: bad argument");
    auto it = m->seqs.find(seq_id);
    FH_REQUIRE(it != m->seqs.end(), "prefix_cache_register: unknown sequence %llu", (unsigned long long)seq_id);
    const SeqState& st = it->second;
    const std::vector<uint64_t> hashes = fh::block_hash_chain(all_tokens, n, KV_BLOCK);
    // only blocks whose 16 slots were all written: (i + 1)·16 ≤ kv length
    for (size_t i = (size_t)prior_cached_tokens / KV_BLOCK; i < hashes.size() && i < st.blocks.size(); i++) {
        if ((int)(i + 1) * KV_BLOCK > st.len) break;
        m->alloc->register_block_hash(st.blocks[i], hashes[i]);
    }
    return 0;
}

int ferrum_hip_model_prefix_cache_stats(const FerrumHipModel* m, uint64_t* hits, uint64_t* misses, uint64_t* saved_prefill_tokens,
                                        uint64_t* entries) {
    FH_REQUIRE(m, "prefix_cache_stats: null");
    if (hits) *hits = m->prefix_hits;
    if (misses) *misses = m->prefix_misses;
    if (saved_prefill_tokens) *saved_prefill_tokens = m->prefix_saved_tokens;
    if (entries) *entries = m->alloc ? m->alloc->hash_table_size() : 0;
    return 0;
}

int ferrum_hip_model_block_table(const FerrumHipModel* m, uint64_t seq_id, uint32_t* blocks, int capacity,
                                 int* num_blocks, int* kv_len) {
    FH_REQUIRE(m, "model_block_table: null");
    auto it = m->seqs.find(seq_id);
    FH_REQUIRE(it != m->seqs.end(), "model_block_table: unknown sequence %llu", (unsigned long long)seq_id);
    int nb = (int)it->second.blocks.size();
    if (num_blocks) *num_blocks = nb;
    if (kv_len) *kv_len = it->second.len;
    for (int i = 0; i < nb && i < capacity; i++) blocks[i] = it->second.blocks[i];
    return 0;
}

int ferrum_hip_model_read_kv_f32(FerrumHipModel* m, uint64_t seq_id, int layer, int is_v, float* out_host) {
    FH_REQUIRE(m && m->finalized && out_host && layer >= 0 && layer < m->cfg.num_layers, "model_read_kv: bad argument");
    auto it = m->seqs.find(seq_id);
    FH_REQUIRE(it != m->seqs.end(), "model_read_kv: unknown sequence");
    const SeqState& st = it->second;
    if (st.len == 0) return 0;
    const FerrumHipModelConfig& c = m->cfg;
    size_t n = (size_t)st.len * c.num_kv_heads * c.head_dim;
    __half *kd = nullptr, *vd = nullptr;
    float* fd = nullptr;
    int32_t* bt = nullptr;
    FH_CHECK_HIP(hipMalloc((void**)&kd, n * 2));
    FH_CHECK_HIP(hipMalloc((void**)&vd, n * 2));
    FH_CHECK_HIP(hipMalloc((void**)&fd, n * 4));
    FH_CHECK_HIP(hipMalloc((void**)&bt, st.blocks.size() * 4));
    FH_CHECK_HIP(hipMemcpy(bt, st.blocks.data(), st.blocks.size() * 4, hipMemcpyHostToDevice));
    int rc = paged_kv_read_f16(m->layers[layer].k_pool, m->layers[layer].v_pool, bt, st.len, c.num_kv_heads, c.head_dim,
                               KV_BLOCK, kd, vd, m->stream);
    if (!rc) {
        hipLaunchKernelGGL(f16_to_f32_kernel, dim3(cdiv((long)n, 256)), dim3(256), 0, m->stream, is_v ? vd : kd, fd, (long)n);
        (void)hipMemcpyAsync(out_host, fd, n * 4, hipMemcpyDeviceToHost, m->stream);
        (void)hipStreamSynchronize(m->stream);
    }
    (void)hipFree(kd); (void)hipFree(vd); (void)hipFree(fd); (void)hipFree(bt);
    return rc;
}

int ferrum_hip_model_enable_taps(FerrumHipModel* m, int enable) {
    FH_REQUIRE(m && m->finalized, "model_enable_taps: bad state");
    m->taps_enabled = enable != 0;
    if (m->taps_enabled && !m->taps)
        return dev_alloc(&m->taps, (size_t)m->cfg.num_layers * m->cfg.max_tokens * m->cfg.hidden);
    return 0;
}
int ferrum_hip_model_read_taps(FerrumHipModel* m, float* out_host, int max_tokens) {
    FH_REQUIRE(m && m->taps && out_host, "model_read_taps: taps not enabled");
    const int T = std::min(max_tokens, m->taps_tokens);
    for (int li = 0; li < m->cfg.num_layers; li++)
        FH_CHECK_HIP(hipMemcpy(out_host + (size_t)li * max_tokens * m->cfg.hidden,
                               m->taps + (size_t)li * m->cfg.max_tokens * m->cfg.hidden,
                               (size_t)T * m->cfg.hidden * 4, hipMemcpyDeviceToHost));
    return 0;
}
int ferrum_hip_model_stream(FerrumHipModel* m, void** stream) {
    FH_REQUIRE(m && stream, "model_stream: null");
    *stream = m->stream;
    return 0;
}

}  // extern "C"

// ── the forward itself ──────────────────────────────────────────────────────
namespace {

struct StepShape {
    int m_total, num_seqs, max_q_len, max_kv_len, num_sampled;
    bool pure_decode;        // every item is one token AND the model has no uniform sliding window (full-attention decode)
    bool all_single_token;   // every item is one token (windowed layers included): the fused decode attention applies
    int single_prefix;       // leading items that bring exactly one token (the decode rows of a mixed batch) …
    int rest_min_q;          // … and the fewest tokens any item after them brings (0 when there is none)
};

template <typename T>
T* idx(FerrumHipModel* m, size_t off) { return reinterpret_cast<T*>(m->idx_dev + off); }

// Dense projection of the runner: act-order (desc_act) weights were repacked with their rows sorted by group, so the
// input columns are gathered with the same permutation first (like the op-level ferrum_hip_gptq_linear_forward_f16).
// gate_up (+silu·mul) and down grouped GEMMs of a decode-sized batch (P ≤ 1024 pairs) straight from expert_ids:
// expert-major when most experts are routed to (no align at all), else block-major with the align computed inside gate_up.
// Under expert parallelism the kernels see the rank's own experts only: local ids (−1 = another rank's expert, ignored by
// every form) and the local expert count.
static const int32_t* moe_ids(const FerrumHipModel* m) { return m->cfg.expert_parallel ? m->expert_ids_local : m->expert_ids; }

// Inputs of the two grouped GEMMs.  Act-order (desc_act) expert stacks read x'[j] = x[perm[j]] (the packed rows are in sorted-g_idx
// order): the normalised rows / the gated activations are gathered by a launch in front of the GEMM (one per act-order stack and
// layer: kernels/gather_columns.cu:15 in the reference sits in front of every Marlin call).  Natural-order stacks: no launch.
static int moe_gate_up_input(FerrumHipModel* m, LayerWeights& L, int T, hipStream_t s, const __half** x) {
    *x = m->norm_out;
    if (!L.exp_gate_up.perm) return 0;
    FH_REQUIRE(m->moe_gather_x, "MoE act-order gate_up stack without gather scratch");
    if (int rc = gather_columns_f16(m->norm_out, L.exp_gate_up.perm, m->moe_gather_x, T, L.exp_gate_up.k, s)) return rc;
    *x = m->moe_gather_x;
    return 0;
}
static int moe_down_input(FerrumHipModel* m, LayerWeights& L, int P, hipStream_t s, const __half** h) {
    *h = m->moe_act;
    if (!L.exp_down.perm) return 0;
    FH_REQUIRE(m->moe_gather_h, "MoE act-order down stack without gather scratch");
    if (int rc = gather_columns_f16(m->moe_act, L.exp_down.perm, m->moe_gather_h, P, L.exp_down.k, s)) return rc;
    *h = m->moe_gather_h;
    return 0;
}

// `route` (decode chain with a deferred merge): the routing arrives as per-part candidate lists; only forms that merge them in
// their own prologue may run (moe_deferred_merge_ok decides that BEFORE the chain launch).
static bool moe_pair_form(const FerrumHipModel* m, const LayerWeights& L, int P) {
    return m->moe_em_min_pairs_x8 > 0 && 8L * P >= (long)m->moe_em_min_pairs_x8 * m->cfg.num_experts && knobs().moe_em2 &&
           m->em2_arrive && !m->em2_failed && !L.exp_down.perm;
}
static bool moe_deferred_merge_ok(const FerrumHipModel* m, const LayerWeights& L, int T, int Q) {
    const int mode = knobs().moe_deferred_merge;
    if (!mode || m->cfg.expert_parallel || Q < 2 || Q > 4 || m->cfg.top_k > 8) return false;
    const int P = T * m->cfg.top_k;
    // few pairs (block-major grid): every workgroup runs the full merge + align on a few hundred bytes of lists
    // (only where the gate_up launch is the K-split form — ≤ 16 pairs, c ≤ 2: c = 1 1.99 → 1.88 ms per step, c = 2 2.16 → 2.03.
    // In the one-wave-per-tile form every one of the ≈ 770 workgroups at c = 4 pays the merge: its launch 13.5 → 17.3 µs against
    // 2.9 µs saved in the chain; at c = 8 2.89 → 3.03 ms per step)
    if (P <= std::min(knobs().moe_kw_pairs, 64) && T <= 4 && !moe_pair_form(m, L, P)) return true;
    // expert-major merged launch: each tile only checks its own expert against the lists; measured at c = 32 the ≈ 8 KiB of lists
    // read by all 7168 workgroups cost the launch what role B's merge cost the chain (47.4 → 50.3 µs vs ≈ −3 µs): off unless asked for
    if (mode < 2) return false;
    MoeRouteLists r;
    r.cand = m->route_cand; r.stats = m->route_stats; r.T = T; r.Q = Q; r.pub_expert_ids = m->expert_ids; r.pub_expert_w = m->expert_w;
    return moe_pair_form(m, L, P) && !L.exp_gate_up.perm && w4_gemm_moe_expert_major_pair_supports(L.exp_gate_up, L.exp_down, m->ep_E, P, &r);
}

static int moe_decode_gemms(FerrumHipModel* m, LayerWeights& L, int P, int max_blocks, hipStream_t s, const MoeRouteLists* route = nullptr) {
    const FerrumHipModelConfig& c = m->cfg;
    const int E = m->ep_E, K = c.top_k;
    const int32_t* ids = moe_ids(m);
    const __half *gx = nullptr, *hx = nullptr;
    if (int rc = moe_gate_up_input(m, L, P / K, s, &gx)) return rc;
    // (the expert-major threshold compares pairs per expert: P pairs over num_experts, whatever share of them is local)
    if (m->moe_em_min_pairs_x8 > 0 && 8L * P >= (long)m->moe_em_min_pairs_x8 * c.num_experts) {
        if (moe_pair_form(m, L, P)) {
            // one launch: down tiles wait for their expert's gate_up tiles inside it (w4_gemm_moe_em2_kernel)
            unsigned* cur = m->em2_arrive + (size_t)m->em2_parity * m->arrive_half_words;
            unsigned* nxt = m->em2_arrive + (size_t)(m->em2_parity ^ 1) * m->arrive_half_words;
            int took = 0;
            if (int rc = w4_gemm_moe_expert_major_pair(L.exp_gate_up, L.exp_down, gx, m->moe_act, m->moe_down, ids, E, P, K, cur, nxt,
                                                       m->inlaunch_timeouts, &took, s, route)) return rc;
            if (took) { m->em2_parity ^= 1; if (route) form_hit(FORM_MOE_DEFERRED_MERGE); return 0; }
        }
        FH_REQUIRE(!route, "MoE decode: the routing was left as candidate lists but the merged launch did not take the shapes");
        if (int rc = w4_gemm_moe_expert_major(L.exp_gate_up, gx, m->moe_act, ids, E, P, K, 1, s)) return rc;
        if (int rc = moe_down_input(m, L, P, s, &hx)) return rc;
        return w4_gemm_moe_expert_major(L.exp_down, hx, m->moe_down, ids, E, P, 1, 0, s);
    }
    if (knobs().moe_bm2 && m->em2_arrive && !m->em2_failed && !L.exp_down.perm && max_blocks <= E &&
        w4_gemm_moe_block_major_pair_supports(L.exp_gate_up, L.exp_down, E, P, max_blocks, route)) {
        // ≤ 64 pairs: one block-major launch, down tiles wait for their block's gate_up tiles inside it (w4_gemm_moe_bm2_kernel)
        unsigned* cur = m->em2_arrive + (size_t)m->em2_parity * m->arrive_half_words;
        unsigned* nxt = m->em2_arrive + (size_t)(m->em2_parity ^ 1) * m->arrive_half_words;
        int took = 0;
        if (int rc = w4_gemm_moe_block_major_pair(L.exp_gate_up, L.exp_down, gx, m->moe_act, m->moe_down, ids, E, P, max_blocks, K, cur, nxt,
                                                  m->inlaunch_timeouts, &took, s, route)) return rc;
        if (took) { m->em2_parity ^= 1; if (route) form_hit(FORM_MOE_DEFERRED_MERGE); return 0; }
    }
    if (route) {
        // gate_up merges the lists, aligns, and publishes ids / weights / align arrays; down reads the align arrays
        if (int rc = w4_gemm_moe_merge_route(L.exp_gate_up, gx, m->moe_act, route->cand, route->stats, route->T, route->Q, K, route->norm_topk,
                                             E, max_blocks, 1, route->pub_expert_ids, route->pub_expert_w, m->sorted_ids, m->block_ids,
                                             m->total_post_pad, s)) return rc;
        form_hit(FORM_MOE_DEFERRED_MERGE);
        if (int rc = moe_down_input(m, L, P, s, &hx)) return rc;
        return w4_gemm_moe(L.exp_down, hx, m->moe_down, m->sorted_ids, m->block_ids, m->total_post_pad, P, max_blocks, 1, 0, s);
    }
    // (the router picks K distinct experts per token: with ≤ 16 tokens no expert holds more than 16 pairs — the ballot align)
    if (int rc = w4_gemm_moe_inline_align(L.exp_gate_up, gx, m->moe_act, ids, E, P, max_blocks, K, 1,
                                          m->sorted_ids, m->block_ids, m->total_post_pad, s, P / K <= 16 && !c.expert_parallel)) return rc;
    if (int rc = moe_down_input(m, L, P, s, &hx)) return rc;
    return w4_gemm_moe(L.exp_down, hx, m->moe_down, m->sorted_ids, m->block_ids, m->total_post_pad, P, max_blocks, 1, 0, s);
}

// gate_up (+silu·mul) and down grouped GEMMs for any batch size, straight from expert_ids: decode-sized batches (≤ 1024
// pairs) without a separate align launch, else align + the block shape that fits the pairs per expert.
#define FH_TRY(call) do { if (int rc_ = (call)) return rc_; } while (0)
static int moe_batch_gemms(FerrumHipModel* m, LayerWeights& L, int P, int sorted_max, int max_blocks, hipStream_t s) {
    const FerrumHipModelConfig& c = m->cfg;
    const int E = m->ep_E, K = c.top_k, Eg = c.num_experts;     // block shapes are chosen by pairs per expert over ALL experts
    const int32_t* ids = moe_ids(m);
    const __half *gx = nullptr, *hx = nullptr;
    if (P > 1024) FH_TRY(moe_gate_up_input(m, L, P / K, s, &gx));            // (decode-sized batches: moe_decode_gemms gathers for itself)
    if (P <= 1024) {
        FH_TRY(moe_decode_gemms(m, L, P, max_blocks, s));
    } else if (P >= (long)m->moe_tile96_min_pairs_per_expert * Eg && L.exp_gate_up.G % 2 == 0 && L.exp_down.G % 2 == 0) {
        // long prefill: 96-row blocks through w4_gemm_big_kernel<6> (two workgroups per CU; group scale folded into the fp16 B operand)
        const int sorted_max96 = P + E * 96, max_blocks96 = std::min(sorted_max96 / 96, P / 96 + std::min(P, E));
        FH_TRY(moe_align_block_size(ids, m->sorted_ids, m->block_ids, m->total_post_pad, P, E, 96, sorted_max96, s));
        {
            FH_TRY(w4_gemm_moe_tile(L.exp_gate_up, gx, m->moe_act, m->sorted_ids, m->block_ids, m->total_post_pad, P,
                                 max_blocks96, 96, K, 1, s));
        }
        FH_TRY(moe_down_input(m, L, P, s, &hx));
        FH_TRY(w4_gemm_moe_tile(L.exp_down, hx, m->moe_down, m->sorted_ids, m->block_ids, m->total_post_pad, P,
                             max_blocks96, 96, 1, 0, s));
    } else if (P >= (long)m->moe_tile128_min_pairs_per_expert * Eg && L.exp_gate_up.G % 2 == 0 && L.exp_down.G % 2 == 0) {
        // long prefill: 128-row blocks through w4_gemm_big_kernel (group scale folded into the fp16 B operand: the matrix pipe, not
        // vector issue, bounds it; twice the padding of 64-row blocks, hence only from a few hundred pairs per expert)
        const int sorted_max128 = P + E * 128, max_blocks128 = std::min(sorted_max128 / 128, P / 128 + std::min(P, E));
        FH_TRY(moe_align_block_size(ids, m->sorted_ids, m->block_ids, m->total_post_pad, P, E, 128, sorted_max128, s));
        FH_TRY(w4_gemm_moe_tile(L.exp_gate_up, gx, m->moe_act, m->sorted_ids, m->block_ids, m->total_post_pad, P,
                             max_blocks128, 128, K, 1, s));
        FH_TRY(moe_down_input(m, L, P, s, &hx));
        FH_TRY(w4_gemm_moe_tile(L.exp_down, hx, m->moe_down, m->sorted_ids, m->block_ids, m->total_post_pad, P,
                             max_blocks128, 128, 1, 0, s));
    } else if (P >= m->moe_tile_min_pairs_per_expert * Eg) {
        // prefill: ≥ 32 pairs per expert on average → 64-row blocks through the LDS-tiled kernel
        const int sorted_max64 = P + E * 64, max_blocks64 = std::min(sorted_max64 / 64, P / 64 + std::min(P, E));
        FH_TRY(moe_align_block_size(ids, m->sorted_ids, m->block_ids, m->total_post_pad, P, E, 64, sorted_max64, s));
        FH_TRY(w4_gemm_moe_tile(L.exp_gate_up, gx, m->moe_act, m->sorted_ids, m->block_ids, m->total_post_pad, P,
                             max_blocks64, 64, K, 1, s));
        FH_TRY(moe_down_input(m, L, P, s, &hx));
        FH_TRY(w4_gemm_moe_tile(L.exp_down, hx, m->moe_down, m->sorted_ids, m->block_ids, m->total_post_pad, P,
                             max_blocks64, 64, 1, 0, s));
    } else if (P >= m->moe_tile32_min_pairs_per_expert * Eg) {
        // a few hundred tokens (a fresh prompt riding along with the decode batch, a lone short prefill: 8–31 pairs
        // per expert): 32-row blocks through the LDS-tiled kernel — every expert's weights about once instead of
        // once per 16 pairs
        const int sorted_max32 = P + E * 32, max_blocks32 = std::min(sorted_max32 / 32, P / 32 + std::min(P, E));
        FH_TRY(moe_align_block_size(ids, m->sorted_ids, m->block_ids, m->total_post_pad, P, E, 32, sorted_max32, s));
        FH_TRY(w4_gemm_moe_tile(L.exp_gate_up, gx, m->moe_act, m->sorted_ids, m->block_ids, m->total_post_pad, P,
                             max_blocks32, 32, K, 1, s));
        FH_TRY(moe_down_input(m, L, P, s, &hx));
        FH_TRY(w4_gemm_moe_tile(L.exp_down, hx, m->moe_down, m->sorted_ids, m->block_ids, m->total_post_pad, P,
                             max_blocks32, 32, 1, 0, s));
    } else {
        FH_TRY(moe_align_block_size(ids, m->sorted_ids, m->block_ids, m->total_post_pad, P, E, 16, sorted_max, s));
        FH_TRY(w4_gemm_moe(L.exp_gate_up, gx, m->moe_act, m->sorted_ids, m->block_ids, m->total_post_pad, P,
                        max_blocks, K, 1, s));
        FH_TRY(moe_down_input(m, L, P, s, &hx));
        FH_TRY(w4_gemm_moe(L.exp_down, hx, m->moe_down, m->sorted_ids, m->block_ids, m->total_post_pad, P,
                        max_blocks, 1, 0, s));
    }
    return 0;
}
#undef FH_TRY

int dense_linear(FerrumHipModel* m, const W4Device& w, const __half* x, __half* out, int T, hipStream_t s, bool x_permuted = false) {
    if (w.perm && (x_permuted || w.perm_folded)) form_hit(FORM_PERM_PRODUCER);
    else if (w.perm) {
        FH_REQUIRE(m->gather_scratch, "dense_linear: act-order weights but no gather scratch");
        if (int rc = gather_columns_f16(x, w.perm, m->gather_scratch, T, w.k, s)) return rc;
        x = m->gather_scratch;
    }
    return w4_gemm_dense(w, x, out, T, m->workspace, m->workspace_bytes, s);
}

// Enqueue every kernel of one forward on m->stream.  Index tensors are already on the device.
struct GreedyDeviceOpts {          // device pointers, already uploaded on m->stream
    const uint8_t* mask = nullptr; int mask_len = 0;
    const uint32_t* row_offsets = nullptr; const uint32_t* token_ids = nullptr; const float* penalties = nullptr;
};
int enqueue_forward(FerrumHipModel* m, const StepShape& sh, bool greedy, const GreedyDeviceOpts* gopts = nullptr,
                    const DecodeAdvance* advance = nullptr, int* advance_fused = nullptr) {
    if (advance_fused) *advance_fused = 0;
    const FerrumHipModelConfig& c = m->cfg;
    hipStream_t s = m->stream;
    const int T = sh.m_total, H = c.hidden, nq = c.num_heads, nkv = c.num_kv_heads, hd = c.head_dim;
    const int qk_mode = c.has_qk_norm ? 1 : 2;
    const uint32_t* tokens = idx<uint32_t>(m, m->il.tokens);
    const uint32_t* cu = idx<uint32_t>(m, m->il.cu_seqlens);
    const uint32_t* pos = idx<uint32_t>(m, m->il.pos_offsets);
    const uint32_t* kvl = idx<uint32_t>(m, m->il.kv_lens);
    const int32_t* sampled_idx = idx<int32_t>(m, m->il.sampled_idx);
    const int32_t* bt = idx<int32_t>(m, m->il.block_tables);
    int rc;
    // FERRUM_HIP_TRACE_LAUNCHES=1 (with FERRUM_HIP_NO_GRAPH=1: a stream cannot be synchronised inside a capture): every step of
    // the forward is named on stderr and waited for, so a faulting kernel is the last line printed
    const bool trace = knobs().trace_launches;
#define RUN(x)                                                                       \
    do {                                                                             \
        if (trace) fprintf(stderr, "[ferrum_hip] T=%d %.70s\n", T, #x);              \
        if ((rc = (x))) return rc;                                                   \
        if (trace) FH_CHECK_HIP(hipStreamSynchronize(s));                            \
    } while (0)
    // arrival counters of the split route kernel: re-armed by the kernel itself, but zeroed per forward as well (by the head
    // kernel below) so that an aborted launch can never poison the next one
    const bool sandwich = c.sandwich_norms != 0;
    // head, one launch: embedding_lookup (+ the Gemma embedding scale, llama_family.rs:3656) → residual stream (fp16; for
    // sandwich-norm models also the fp32 shadow, llama_family.rs:3664-3674) → layer 0's input norm.  Later layers get their
    // input norm fused into the previous layer's tail.
    // act-order (desc_act) projections read x'[j] = x[perm[j]]: the kernel that PRODUCES the row writes it permuted wherever one
    // workgroup owns the row (every norm of the dense layer) or the rows are few (decode attention scatters); the gated
    // activation is permuted by gate_up's column order (pack_gate_up).  Only the prefill o_proj keeps a gather launch.
    RUN(embed_rms_norm_f16(m->embed, tokens, c.embed_scale, m->residual, sandwich ? m->residual_f32 : nullptr, m->layers[0].input_ln,
                           c.rms_eps, m->norm_out, T, H, m->route_arrive, m->route_arrive ? 64 + (int)m->arrive_half_words : 0, s, m->layers[0].qkv.perm));
    m->em2_parity = 0;          // the head kernel zeroed half 0 of the merged launches' counters; every such launch zeroes the other half
    m->chain_parity = 0;
    bool qkv_in_perm = m->layers[0].qkv.perm != nullptr;      // norm_out holds the row in L.qkv's packed order
    // decode at ≤ 4 rows (MoE models): the tail of layer l — combine + residual add + next input norm — runs as the prologue of
    // layer l+1's q|k|v GEMM instead of as a launch of its own (every dependent launch costs ≈ 4 µs there)
    bool pending_tail = false;
    FusedCombineNorm tail{};
    bool pending_chain_tail = false;
    auto chain_desc = [&](int li) {
        LayerWeights& L = m->layers[li];
        DecodeChainDesc d;
        d.T = T; d.H = H; d.nq = nq; d.nkv = nkv; d.head_dim = hd;
        d.top_k = c.top_k; d.down = m->moe_down; d.comb_w = m->expert_w; d.res_in = m->residual2; d.ln_in = L.input_ln; d.eps = c.rms_eps;
        d.res_a = m->residual; d.norm1 = m->norm_out;
        d.qkv = &L.qkv; d.qkv_out = m->qkv_out;
        d.k_pool = L.k_pool; d.v_pool = L.v_pool; d.block_tables = bt; d.kv_lens = kvl;
        d.q_norm_w = L.q_norm ? L.q_norm : L.input_ln; d.k_norm_w = L.k_norm ? L.k_norm : L.input_ln;
        const int pattern = c.sliding_window_pattern;
        const bool is_global = pattern == 0 || (li + 1) % pattern == 0;
        d.sliding_window = pattern == 0 ? c.sliding_window : (is_global ? 0 : c.sliding_window);
        d.cos_t = (!is_global && m->cos_local) ? m->cos_local : m->cos_t;
        d.sin_t = (!is_global && m->sin_local) ? m->sin_local : m->sin_t;
        d.qk_mode = qk_mode; d.max_blocks = m->max_blocks_per_seq;
        d.attn_out = m->attn_out; d.o = &L.o; d.o_out = m->o_out; d.o_part = m->chain_o_part;
        d.res_b_out = m->residual2; d.post_ln = L.post_ln; d.norm2 = m->norm_out; d.router_w = L.router;
        d.E = c.num_experts; d.r_top_k = c.top_k; d.norm_topk = c.norm_topk_prob;
        int Q = m->route_parts;
        const int tiles = (c.num_experts + 15) / 16;
        while (Q > 1 && (tiles % Q != 0 || Q > 8)) Q >>= 1;
        d.Q = c.num_experts > 0 ? Q : 1;
        d.cand = m->route_cand; d.stats = m->route_stats; d.route_arrive = m->route_arrive; d.ids = m->expert_ids; d.weights = m->expert_w;
        d.timeout = m->inlaunch_timeouts;
        // KV ranges per (sequence, kv head) in the attention role: ≈ chain_split_keys (256) keys each while the role stays within
        // W workgroups — W = 32 … 256 growing with the (sequence, kv head) count: the chain holds one workgroup per CU, and ranges
        // beyond that push the o_proj / router roles (which prefetch their weights while they wait) out of residence — but never
        // more than ≈ 1024 keys per range while 256 workgroups allow (profiles/r03_chain_kv_splits.txt: c × context × ranges)
        const long keys = d.sliding_window > 0 ? std::min<long>(sh.max_kv_len, d.sliding_window) : sh.max_kv_len;
        const long units = (long)T * nkv, sk = std::max(1, knobs().chain_split_keys);
        const long cap_w = std::min<long>(256, std::max<long>(4 * units, std::min<long>(8 * units, 64)));
        long ns_l = std::min<long>({16, std::max<long>(1, cap_w / units), (keys + sk - 1) / sk});
        ns_l = std::max<long>(ns_l, std::min<long>({16, std::max<long>(1, 256 / units), (keys + 4 * sk - 1) / (4 * sk)}));
        int ns = keys <= 2 * sk ? 1 : (int)std::max<long>(1, ns_l);        // (≤ 512 keys: one range — c=4 at 256+: 2.31 vs 2.36 ms split in two)
        if (knobs().chain_attn_splits > 0) ns = std::min(16, knobs().chain_attn_splits);
        while (ns > 1 && units * ns > CHAIN_MAX_ATTN_WGS) ns--;
        if (!m->chain_attn_partial || units > 4096 || knobs().chain_split_keys <= 0) ns = 1;
        d.attn_splits = ns; d.attn_partial = m->chain_attn_partial; d.attn_tickets = m->chain_attn_tickets;
        return d;
    };
    // The chain stays the layer's form while one attention workgroup has ≤ chain_max_keys (4096) keys to stream (more where T·nkv
    // workgroups already cover the chip twice); with its KV ranges it beat the stand-alone split-KV launches at every point measured
    // (c=1 at 32 k keys: 3.06 vs 3.52 ms per step; c=4 at 16 k: 3.62 vs 3.96; c=32 at 8 k: 7.85 vs 8.24) — beyond that is unmeasured
    auto chain_kv_ok = [&](int li) {
        const int pattern = c.sliding_window_pattern;
        const bool is_global = pattern == 0 || (li + 1) % pattern == 0;
        const int window = pattern == 0 ? c.sliding_window : (is_global ? 0 : c.sliding_window);
        const long keys = window > 0 ? std::min<long>(sh.max_kv_len, window) : sh.max_kv_len;
        return keys <= (long)knobs().chain_max_keys * chain_desc(li).attn_splits * std::max<long>(1, (long)T * nkv / 128);
    };
    auto chain_ok = [&](int li) {
        if (c.num_experts <= 0 || !knobs().decode_chain || m->em2_failed || !m->em2_arrive || !m->inlaunch_timeouts) return false;
        if (!sh.all_single_token || !m->fuse_rope_attn || T > knobs().chain_max_rows || m->taps_enabled || sandwich) return false;
        if (c.tp_world > 1 && c.expert_parallel != 2) return false;             // (tensor-parallel attention: an all-reduce sits behind o_proj)
        if (T * c.top_k > 1024 || c.top_k > 8 || T > 128 || !chain_kv_ok(li)) return false;
        return decode_chain_supports(chain_desc(li));
    };
    // dense models at 17–32 rows: the same launch for the attention half (tail of the previous layer's MLP — its down slabs +
    // residual + norm — q|k|v, attention, o_proj, add + norm), then gate_up slabs → gated activation → down slabs
    bool pending_dense_tail = false;
    int tail_S = 0, tail_rows_pad = 0, tail_n_pad = 0;
    auto dense_chain_ok = [&](int li) {
        const LayerWeights& L = m->layers[li];
        if (c.num_experts > 0 || !knobs().decode_chain || !knobs().dense_chain || m->em2_failed || !m->em2_arrive || !m->inlaunch_timeouts || !m->residual2) return false;
        if (!sh.all_single_token || !m->fuse_rope_attn || T > std::min(64, knobs().chain_max_rows) || m->taps_enabled || sandwich || c.tp_world > 1) return false;      // (beyond 64 rows the 64-row GEMM tiles win: Llama-3.1-8B c=96 5.56 vs 5.61 ms)
        if (!L.o.qw || !L.gate_up.qw || !L.down.qw || L.gate_up.perm || L.down.perm || L.qkv.perm || L.o.perm) return false;
        if (L.qkv.bias || L.o.bias || !chain_kv_ok(li)) return false;
        DecodeChainDesc d = chain_desc(li);
        d.has_a = false;
        return decode_chain_supports(d);
    };
    for (int li = 0; li < c.num_layers; li++) {
        LayerWeights& L = m->layers[li];
        const __half* dummy = L.input_ln;
        const __half* next_ln = li + 1 < c.num_layers ? m->layers[li + 1].input_ln : nullptr;
        const int32_t* next_qkv_perm = li + 1 < c.num_layers ? m->layers[li + 1].qkv.perm : nullptr;
        const int32_t* gu_perm = L.gate_up.perm;
        // MoE decode at ≤ 32 rows: [the previous layer's tail] + q|k|v + attention + o_proj + add/norm/route as ONE launch
        // (chain.hip), then the merged gate_up → down launch: two launches per layer instead of seven
        if (dense_chain_ok(li)) {
            DecodeChainDesc d = chain_desc(li);
            d.has_a = pending_dense_tail;
            d.a_slabs = m->workspace; d.a_S = tail_S; d.a_slab_stride = (long)tail_rows_pad * tail_n_pad; d.a_ld = tail_n_pad;
            d.a_x = tail_S == 0 ? m->mlp_out : nullptr;
            d.cnt = m->em2_arrive + (size_t)m->chain_parity * m->arrive_half_words;
            d.cnt_next = m->em2_arrive + (size_t)(m->chain_parity ^ 1) * m->arrive_half_words;
            RUN(decode_chain_f16(d, s));
            m->chain_parity ^= 1;
            pending_dense_tail = false;
            form_hit(FORM_DENSE_CHAIN);
            const int I = c.intermediate;
            int S = 0, rows_pad = 0, n_pad = 0;
            const bool slabs = m->dense_slabs && T > 16 && T <= 32;      // 17–32 rows: split-K slabs summed by their consumers
            if (slabs) {
                RUN(w4_gemm_dense_slabs_lds(L.gate_up, m->norm_out, m->workspace, m->workspace_bytes, T, &S, &rows_pad, &n_pad, s));
                RUN(fused_gated_act_slabs_f16(m->workspace, S, (long)rows_pad * n_pad, n_pad, m->act_out, T, I, c.activation == 1, s));
                S = 0;
                RUN(w4_gemm_dense_slabs_lds(L.down, m->act_out, m->workspace, m->workspace_bytes, T, &S, &rows_pad, &n_pad, s));
            } else {
                RUN(dense_linear(m, L.gate_up, m->norm_out, m->gate_up_out, T, s, true));
                if (c.activation == 1) { RUN(fused_gelu_tanh_mul_split_f16(m->gate_up_out, m->act_out, T, I, s)); }
                else { RUN(fused_silu_mul_split_f16(m->gate_up_out, m->act_out, T, I, s)); }
                RUN(dense_linear(m, L.down, m->act_out, m->mlp_out, T, s));
            }
            if (li + 1 < c.num_layers && dense_chain_ok(li + 1)) {
                pending_dense_tail = true;          // MLP output + residual + next input norm: the first role of the next layer's launch
                tail_S = S; tail_rows_pad = rows_pad; tail_n_pad = n_pad;
            } else if (!slabs) {
                if (next_ln) { RUN(fused_add_rms_norm_f16(m->residual2, m->mlp_out, next_ln, c.rms_eps, m->norm_out, T, H, s, next_qkv_perm)); }
                else { RUN(add_inplace_f16(m->residual2, m->mlp_out, (long)T * H, s)); }
                FH_CHECK_HIP(hipMemcpyAsync(m->residual, m->residual2, (size_t)T * H * sizeof(__half), hipMemcpyDeviceToDevice, s));
                qkv_in_perm = next_ln && next_qkv_perm != nullptr;
            } else {
                // (the residual of this layer sits in the ping-pong buffer: the stand-alone tail works on it in place, then it moves back)
                RUN(fused_add_rms_norm_route_slabs_f16(m->residual2, nullptr, m->workspace, S, (long)rows_pad * n_pad, n_pad,
                                                       next_ln ? next_ln : L.input_ln, c.rms_eps, m->norm_out, nullptr, 0, 0, 0, nullptr, nullptr,
                                                       nullptr, T, H, s, next_qkv_perm));
                FH_CHECK_HIP(hipMemcpyAsync(m->residual, m->residual2, (size_t)T * H * sizeof(__half), hipMemcpyDeviceToDevice, s));
                qkv_in_perm = next_ln && next_qkv_perm != nullptr;
            }
            continue;
        }
        if (chain_ok(li)) {
            const int E = c.num_experts, K = c.top_k, P = T * K, sorted_max = P + E * 16;
            const int max_blocks = std::min(sorted_max / 16, P / 16 + std::min(P, E));
            DecodeChainDesc d = chain_desc(li);
            d.has_a = pending_chain_tail;
            d.cnt = m->em2_arrive + (size_t)m->chain_parity * m->arrive_half_words + (size_t)c.num_experts * MOE_PAIR_COUNTER_STRIDE;
            d.cnt_next = m->em2_arrive + (size_t)(m->chain_parity ^ 1) * m->arrive_half_words + (size_t)c.num_experts * MOE_PAIR_COUNTER_STRIDE;
            // the Q candidate lists of a token are merged by the grouped GEMM's prologue (under its first weight loads) where that
            // form runs, instead of by role B's first part at the end of the chain (≈ 3.7 µs of every layer)
            d.defer_merge = d.Q > 1 && moe_deferred_merge_ok(m, L, T, d.Q);
            MoeRouteLists lists;
            lists.cand = m->route_cand; lists.stats = m->route_stats; lists.T = T; lists.Q = d.Q; lists.norm_topk = c.norm_topk_prob;
            lists.pub_expert_ids = m->expert_ids; lists.pub_expert_w = m->expert_w;
            RUN(decode_chain_f16(d, s));
            m->chain_parity ^= 1;
            pending_chain_tail = false;
            form_hit(d.Q > 1 ? FORM_ROUTE_SPLIT : FORM_ROUTE_FUSED);
            if (c.expert_parallel) RUN(moe_remap_expert_ids(m->expert_ids, m->expert_ids_local, P, m->ep_e0, m->ep_E, s));
            RUN(moe_decode_gemms(m, L, P, max_blocks, s, d.defer_merge ? &lists : nullptr));
            if (!c.expert_parallel && li + 1 < c.num_layers && chain_ok(li + 1)) {
                pending_chain_tail = true;          // combine + add + next input norm: the first role of the next layer's launch
            } else if (!c.expert_parallel) {
                RUN(moe_combine_add_rms_norm_f16(m->moe_down, m->expert_w, m->residual2, m->residual, next_ln, c.rms_eps, m->norm_out, T, K, H, s,
                                                 next_qkv_perm));
                qkv_in_perm = next_ln && next_qkv_perm != nullptr;
            } else {
                RUN(moe_combine_local_f16(m->moe_down, m->expert_w, m->expert_ids_local, m->mlp_out, T, K, H, s));
                RUN(tp_all_reduce(m, m->mlp_out, (size_t)T * H));
                RUN(moe_combine_add_rms_norm_f16(m->mlp_out, m->ones, m->residual2, m->residual, next_ln, c.rms_eps, m->norm_out, T, 1, H, s,
                                                 next_qkv_perm));
                qkv_in_perm = next_ln && next_qkv_perm != nullptr;
            }
            continue;
        }
        if (pending_tail) {
            RUN(w4_gemm_dense(L.qkv, nullptr, m->qkv_out, T, m->workspace, m->workspace_bytes, s, &tail));
            pending_tail = false;
        } else {
            RUN(dense_linear(m, L.qkv, m->norm_out, m->qkv_out, T, s, qkv_in_perm));
        }
        qkv_in_perm = false;
        bool attn_perm = false;          // attn_out holds the row in L.o's packed order
        // per-layer attention schedule (llama_layer_attention_schedule, llama_family.rs:1028-1045)
        const int pattern = c.sliding_window_pattern;
        const bool is_global = pattern == 0 || (li + 1) % pattern == 0;
        const int layer_window = pattern == 0 ? c.sliding_window : (is_global ? 0 : c.sliding_window);
        const float* cos_l = (!is_global && m->cos_local) ? m->cos_local : m->cos_t;
        const float* sin_l = (!is_global && m->sin_local) ? m->sin_local : m->sin_t;
        const bool layer_decode = sh.pure_decode && layer_window == 0;
        if (sh.all_single_token && m->fuse_rope_attn && nq / nkv <= 14) {
            // decode: QK-norm + RoPE + KV write happen in the attention kernel's prologue (one launch fewer)
            RUN(paged_decode_attention_fused_qkv_f16(m->qkv_out, L.q_norm ? L.q_norm : dummy, L.k_norm ? L.k_norm : dummy,
                                                     cos_l, sin_l, c.rms_eps, qk_mode, L.k_pool, L.v_pool, m->attn_out,
                                                     bt, kvl, sh.num_seqs, sh.max_kv_len, nq, nkv, hd, layer_window, KV_BLOCK,
                                                     m->max_blocks_per_seq, m->workspace, m->workspace_bytes, s, L.o.inv_perm));
            attn_perm = L.o.perm != nullptr;
        } else {
            RUN(split_qkv_norm_rope_into_paged_cache_varlen_f16(m->qkv_out, L.q_norm ? L.q_norm : dummy,
                                                                L.k_norm ? L.k_norm : dummy, cos_l, sin_l, m->q_out,
                                                                L.k_pool, L.v_pool, cu, pos, bt, sh.num_seqs, T, nq, nkv, hd,
                                                                c.rms_eps, qk_mode, KV_BLOCK, m->max_blocks_per_seq, s));
            if (layer_decode) {
                RUN(paged_batched_decode_attention_f16(m->q_out, L.k_pool, L.v_pool, m->attn_out, bt, kvl, sh.num_seqs,
                                                       sh.max_kv_len, nq, nkv, hd, KV_BLOCK, m->max_blocks_per_seq,
                                                       m->workspace, m->workspace_bytes, s, L.o.inv_perm));
                attn_perm = L.o.perm != nullptr;
            } else if (sh.single_prefix > 0 && sh.single_prefix < sh.num_seqs && (long)sh.rest_min_q * (nq / nkv) >= 16L * m->attn_flash_min_rows) {
                // decode rows first, LONG prompts after them (≥ 1024 tokens at a GQA group of 8; for a 256-token prompt the
                // second launch costs more than the form saves: 9.7 → 10.2 ms per iteration): two launches, so that the
                // prompts take the LDS-shared K/V form and the decode rows the KV-split form (q/out are indexed by the
                // absolute token offsets in cu_seqlens_q, so sub-ranges of the sequence arrays are enough)
                const int n1 = sh.single_prefix;
                RUN(paged_varlen_attention_f16(m->q_out, L.k_pool, L.v_pool, m->attn_out, cu, pos, bt, n1, n1, 1, sh.max_kv_len, nq,
                                               nkv, hd, layer_window, KV_BLOCK, m->max_blocks_per_seq, m->workspace,
                                               m->workspace_bytes, s));
                RUN(paged_varlen_attention_f16(m->q_out, L.k_pool, L.v_pool, m->attn_out, cu + n1, pos + n1,
                                               bt + (size_t)n1 * m->max_blocks_per_seq, sh.num_seqs - n1, T - n1, sh.max_q_len,
                                               sh.max_kv_len, nq, nkv, hd, layer_window, KV_BLOCK, m->max_blocks_per_seq,
                                               m->workspace, m->workspace_bytes, s));
            } else {
                RUN(paged_varlen_attention_f16(m->q_out, L.k_pool, L.v_pool, m->attn_out, cu, pos, bt, sh.num_seqs, T,
                                               sh.max_q_len, sh.max_kv_len, nq, nkv, hd, layer_window, KV_BLOCK,
                                               m->max_blocks_per_seq, m->workspace, m->workspace_bytes, s));
            }
        }
        if (c.num_experts > 0) {
            const int E = c.num_experts, K = c.top_k, P = T * K, sorted_max = P + E * 16;
            // Σ_e ceil(cnt_e/16) ≤ P/16 + min(P, E): the grid covers every block that can exist
            const int max_blocks = std::min(sorted_max / 16, P / 16 + std::min(P, E));
            const int tiles = (E + 15) / 16;
            int Q = m->route_parts;
            while (Q > 1 && (tiles % Q != 0 || Q > 8)) Q >>= 1;
            // attention heads sharded, o_proj row-parallel: all-reduce before the add + norm + route kernel (expert_parallel 2 keeps
            // attention replicated: full heads on every rank, nothing to reduce there)
            const bool tp = c.tp_world > 1 && c.expert_parallel != 2;
            const bool ep = c.expert_parallel != 0;
            // expert parallel: after routing, the pair ids become local ids (−1 = another rank's expert)
            auto after_route = [&]() -> int {
                return ep ? moe_remap_expert_ids(m->expert_ids, m->expert_ids_local, P, m->ep_e0, m->ep_E, s) : 0;
            };
            // the layer tail: residual_out = residual_in + Σ_k w_k·down_k, then the next input norm.  Expert parallel: the sum over
            // the rank's own experts is a partial [T, H] that meets the other ranks' in the all-reduce a dense MLP uses, and the
            // reduced row then enters the residual like a single expert row of weight 1 (the same kernel, top_k = 1)
            auto moe_tail = [&](const __half* res_in, __half* res_out) -> int {
                // (an act-order q|k|v of the next layer gets its rows permuted by this kernel)
                qkv_in_perm = next_ln && next_qkv_perm != nullptr;
                if (!ep)
                    return moe_combine_add_rms_norm_f16(m->moe_down, m->expert_w, res_in, res_out, next_ln, c.rms_eps, m->norm_out, T, K, H, s, next_qkv_perm);
                if (int rc_ = moe_combine_local_f16(m->moe_down, m->expert_w, m->expert_ids_local, m->mlp_out, T, K, H, s)) return rc_;
                if (int rc_ = tp_all_reduce(m, m->mlp_out, (size_t)T * H)) return rc_;
                return moe_combine_add_rms_norm_f16(m->mlp_out, m->ones, res_in, res_out, next_ln, c.rms_eps, m->norm_out, T, 1, H, s, next_qkv_perm);
            };
            const bool decode_fast = T <= 64 && P <= 1024 && K <= 8 && Q >= 1 && tiles / Q <= 8 && (!L.o.perm || attn_perm);
            const bool o_quant = L.o.qw != nullptr;           // the slab forms are INT4 kernels; an unquantised o_proj takes the direct GEMM
            if (decode_fast) {
                // o_proj as fp32 split-K slabs, reduced inside the next kernel (no reduce launch)
                const float* slabs = nullptr;
                int S = m->o_slabs, rows_pad = 0, n_pad = 0;
                if (tp || !o_quant) {
                    RUN(dense_linear(m, L.o, m->attn_out, m->o_out, T, s, attn_perm));
                    if (tp) RUN(tp_all_reduce(m, m->o_out, (size_t)T * H));
                    S = 0;
                } else if (S > 0 && T > 16 && T <= 32 && !L.o.bias) {
                    if (L.o.perm) form_hit(FORM_PERM_PRODUCER);          // (attn_perm: the decode attention wrote the permuted rows)
                    // 17–32 rows: activations staged once per workgroup in LDS (a wave fetching its own fragments pulls 2× the
                    // weight bytes from L2); decode c=32 4.60 → 4.57 ms per step
                    RUN(w4_gemm_dense_slabs_lds(L.o, m->attn_out, m->workspace, m->workspace_bytes, T, &S, &rows_pad, &n_pad, s));
                    slabs = m->workspace;
                } else if (S > 0) {
                    if (L.o.perm) form_hit(FORM_PERM_PRODUCER);
                    RUN(w4_gemm_dense_slabs(L.o, m->attn_out, m->workspace, m->workspace_bytes, T, S, &rows_pad, &n_pad, s));
                    slabs = m->workspace;
                    S = std::min(S, L.o.G);
                } else {
                    RUN(dense_linear(m, L.o, m->attn_out, m->o_out, T, s, attn_perm));
                }
                form_hit(Q > 1 ? FORM_ROUTE_SPLIT : FORM_ROUTE_FUSED);
                if (Q > 1) {
                    // B over Q expert parts per token (one CU pulls the 512-KB router at ≈70 GB/s; Q CUs share it):
                    // residual → residual2 (ping-pong), norm_out; each token's last-arriving part merges the Q
                    // candidate lists inside the launch and writes expert_ids / expert_w.
                    RUN(fused_add_rms_norm_route_split_f16(m->residual, m->residual2, m->o_out, slabs, S, (long)rows_pad * n_pad,
                                                           n_pad, L.post_ln, c.rms_eps, m->norm_out, L.router, E, K, Q,
                                                           m->route_cand, m->route_stats, m->route_arrive, c.norm_topk_prob,
                                                           m->expert_ids, m->expert_w, nullptr, T, H, s));
                    RUN(after_route());
                    RUN(moe_decode_gemms(m, L, P, max_blocks, s));
                    if (T <= m->fuse_tail_max_rows && next_ln && !m->taps_enabled && !ep &&
                        w4_gemm_dense_can_fuse_combine_norm(m->layers[li + 1].qkv, T)) {
                        tail = FusedCombineNorm{m->moe_down, m->expert_w, m->residual2, m->residual, next_ln, c.rms_eps, K};
                        pending_tail = true;
                    } else {
                        RUN(moe_tail(m->residual2, m->residual));
                    }
                } else {
                    RUN(fused_add_rms_norm_route_slabs_f16(m->residual, m->o_out, slabs, S, (long)rows_pad * n_pad, n_pad, L.post_ln,
                                                           c.rms_eps, m->norm_out, L.router, E, K, c.norm_topk_prob, m->expert_ids,
                                                           m->expert_w, nullptr, T, H, s));
                    RUN(after_route());
                    RUN(moe_decode_gemms(m, L, P, max_blocks, s));
                    RUN(moe_tail(m->residual, m->residual));
                }
            } else if (T >= 64 && T < m->route_gemm_min_tokens && !tp && o_quant && !L.o.perm && !L.o.bias && m->o_slabs > 0) {
                // short prefill / a prompt riding along with the decode batch: o_proj as fp32 split-K slabs straight into the
                // add + norm + route kernel (no reduce launch)
                int S = 1, rows_pad = 0, n_pad = 0;
                RUN(w4_gemm_dense_slabs_tile(L.o, m->attn_out, m->workspace, m->workspace_bytes, T, &S, &rows_pad, &n_pad, s));
                RUN(fused_add_rms_norm_route_slabs_f16(m->residual, m->o_out, m->workspace, S, (long)rows_pad * n_pad, n_pad, L.post_ln,
                                                       c.rms_eps, m->norm_out, L.router, E, K, c.norm_topk_prob, m->expert_ids,
                                                       m->expert_w, nullptr, T, H, s));
                RUN(after_route());
                RUN(moe_batch_gemms(m, L, P, sorted_max, max_blocks, s));
                RUN(moe_tail(m->residual, m->residual));
            } else {
                RUN(dense_linear(m, L.o, m->attn_out, m->o_out, T, s, attn_perm));
                if (tp) RUN(tp_all_reduce(m, m->o_out, (size_t)T * H));
                form_hit(T >= m->route_gemm_min_tokens ? FORM_ROUTE_GEMM : FORM_ROUTE_FUSED);
                if (T >= m->route_gemm_min_tokens) {
                    // prefill: one workgroup per token would pull the whole router (E·H·2 B) from L2 per token (2048 tokens:
                    // 1 GB, 95 µs) — run the router as a GEMM over all tokens instead, then the top-k kernel
                    RUN(fused_add_rms_norm_f16(m->residual, m->o_out, L.post_ln, c.rms_eps, m->norm_out, T, H, s));
                    if (knobs().route_gemm_topk && !m->taps_enabled && moe_route_gemm_topk_supports(E, H, K)) {
                        // … and the top-k in the same workgroups: the logits stay on the CU (8192 tokens: 27 + 7 + 16 µs → one launch)
                        form_hit(FORM_ROUTE_GEMM_TOPK);
                        RUN(moe_route_gemm_topk_f16(m->norm_out, L.router, m->expert_ids, m->expert_w, T, E, H, K, c.norm_topk_prob, s));
                    } else {
                        RUN(f16t_gemm_f32out(m->norm_out, L.router, m->router_logits, T, E, H, m->workspace, m->workspace_bytes, s));
                        RUN(moe_route_topk_softmax_f32(m->router_logits, m->expert_ids, m->expert_w, T, E, K, c.norm_topk_prob, s));
                    }
                } else {
                    // residual += o; post-attention norm; router logits; top-k — one launch (fused.hip B)
                    RUN(fused_add_rms_norm_route_f16(m->residual, m->o_out, L.post_ln, c.rms_eps, m->norm_out, L.router, E, K,
                                                     c.norm_topk_prob, m->expert_ids, m->expert_w, nullptr, T, H, s));
                }
                RUN(after_route());
                RUN(moe_batch_gemms(m, L, P, sorted_max, max_blocks, s));
                RUN(moe_tail(m->residual, m->residual));
            }
        } else {
            const int I = c.intermediate;
            // 17–32 rows, one GPU: every projection of the MLP block writes fp32 split-K slabs (LDS-shared activations) and
            // the consumer that exists anyway — add+norm, gated activation — sums them: no reduce launches.
            // Tensor parallel (o / down row-parallel): the partial sums of the ranks meet in an all-reduce, so those two
            // projections produce fp16 partials (GEMM + reduce), all-reduce, then add + norm; the column-parallel gate_up keeps
            // its slabs → gated-activation form.  Everything stays stream-ordered device work, so the step is still one graph.
            const bool tp = c.tp_world > 1;
            // (act-order projections: every input of the chain arrives permuted from its producer — decode attention, the norms
            // below, gate_up's column order)
            const bool slab_chain = m->dense_slabs && T > 16 && T <= 32 && (!L.o.perm || attn_perm) && (!L.down.perm || L.down.perm_folded) &&
                                    L.o.qw && L.gate_up.qw && L.down.qw;   // INT4 slab kernels; unquantised projections take the op chain
            int S = 0, rows_pad = 0, n_pad = 0;
            if (slab_chain) {
                form_hit(FORM_DENSE_SLAB_CHAIN);
                if (tp) {
                    RUN(dense_linear(m, L.o, m->attn_out, m->o_out, T, s, attn_perm));
                    if (sandwich) {
                        RUN(tp_all_reduce(m, m->o_out, (size_t)T * H));
                        RUN(sandwich_add_rms_norm_f32(m->o_out, L.post_attn_ln, m->residual_f32, L.post_ln, c.rms_eps, m->norm_out, T, H, s, gu_perm));
                    } else {
                        RUN(tp_all_reduce_add_rms_norm(m, m->o_out, m->residual, L.post_ln, c.rms_eps, m->norm_out, T, H, s, gu_perm));
                    }
                } else {
                if (L.o.perm) form_hit(FORM_PERM_PRODUCER);
                RUN(w4_gemm_dense_slabs_lds(L.o, m->attn_out, m->workspace, m->workspace_bytes, T, &S, &rows_pad, &n_pad, s));
                if (sandwich) {
                    RUN(sandwich_add_rms_norm_f32_slabs(m->workspace, S, (long)rows_pad * n_pad, n_pad, L.post_attn_ln, m->residual_f32,
                                                        L.post_ln, c.rms_eps, m->norm_out, T, H, s, gu_perm));
                } else {
                    RUN(fused_add_rms_norm_route_slabs_f16(m->residual, nullptr, m->workspace, S, (long)rows_pad * n_pad, n_pad, L.post_ln,
                                                           c.rms_eps, m->norm_out, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, T, H, s, gu_perm));
                }
                }
                if (gu_perm) form_hit(FORM_PERM_PRODUCER);
                if (L.down.perm) form_hit(FORM_PERM_PRODUCER);
                S = 0;
                RUN(w4_gemm_dense_slabs_lds(L.gate_up, m->norm_out, m->workspace, m->workspace_bytes, T, &S, &rows_pad, &n_pad, s));
                RUN(fused_gated_act_slabs_f16(m->workspace, S, (long)rows_pad * n_pad, n_pad, m->act_out, T, I, c.activation == 1, s));
                S = 0;
                if (tp) {
                    RUN(dense_linear(m, L.down, m->act_out, m->mlp_out, T, s));
                    if (sandwich) {
                        RUN(tp_all_reduce(m, m->mlp_out, (size_t)T * H));
                        RUN(sandwich_add_rms_norm_f32(m->mlp_out, L.post_ffn_ln, m->residual_f32, next_ln, c.rms_eps, m->norm_out, T, H, s, next_qkv_perm));
                    } else if (next_ln) {
                        RUN(tp_all_reduce_add_rms_norm(m, m->mlp_out, m->residual, next_ln, c.rms_eps, m->norm_out, T, H, s, next_qkv_perm));
                    } else {
                        RUN(tp_all_reduce(m, m->mlp_out, (size_t)T * H));
                        RUN(add_inplace_f16(m->residual, m->mlp_out, (long)T * H, s));
                    }
                } else {
                RUN(w4_gemm_dense_slabs_lds(L.down, m->act_out, m->workspace, m->workspace_bytes, T, &S, &rows_pad, &n_pad, s));
                if (sandwich) {
                    RUN(sandwich_add_rms_norm_f32_slabs(m->workspace, S, (long)rows_pad * n_pad, n_pad, L.post_ffn_ln, m->residual_f32,
                                                        next_ln, c.rms_eps, m->norm_out, T, H, s, next_qkv_perm));
                } else {
                    // last layer: the norm output is unused (the final norm runs on the sampled rows)
                    RUN(fused_add_rms_norm_route_slabs_f16(m->residual, nullptr, m->workspace, S, (long)rows_pad * n_pad, n_pad,
                                                           next_ln ? next_ln : L.input_ln, c.rms_eps, m->norm_out, nullptr, 0, 0, 0,
                                                           nullptr, nullptr, nullptr, T, H, s, next_qkv_perm));
                }
                }
                qkv_in_perm = next_qkv_perm != nullptr;
            } else {
            RUN(dense_linear(m, L.o, m->attn_out, m->o_out, T, s, attn_perm));
            if (sandwich) {
                RUN(tp_all_reduce(m, m->o_out, (size_t)T * H));
                // Gemma 3 (llama_family.rs:3357-3421): residual += norm(o, post_attention_layernorm); pre-MLP norm
                RUN(sandwich_add_rms_norm_f32(m->o_out, L.post_attn_ln, m->residual_f32, L.post_ln, c.rms_eps, m->norm_out, T, H, s, gu_perm));
            } else {
                RUN(tp_all_reduce_add_rms_norm(m, m->o_out, m->residual, L.post_ln, c.rms_eps, m->norm_out, T, H, s, gu_perm));
            }
            RUN(dense_linear(m, L.gate_up, m->norm_out, m->gate_up_out, T, s, true));
            if (c.activation == 1) {
                RUN(fused_gelu_tanh_mul_split_f16(m->gate_up_out, m->act_out, T, I, s));
            } else {
                RUN(fused_silu_mul_split_f16(m->gate_up_out, m->act_out, T, I, s));
            }
            RUN(dense_linear(m, L.down, m->act_out, m->mlp_out, T, s));
            if (sandwich) {
                RUN(tp_all_reduce(m, m->mlp_out, (size_t)T * H));
                // residual += norm(mlp_out, post_feedforward_layernorm); next layer's input norm rides along
                RUN(sandwich_add_rms_norm_f32(m->mlp_out, L.post_ffn_ln, m->residual_f32, next_ln, c.rms_eps, m->norm_out, T, H, s, next_qkv_perm));
            } else if (next_ln) {
                RUN(tp_all_reduce_add_rms_norm(m, m->mlp_out, m->residual, next_ln, c.rms_eps, m->norm_out, T, H, s, next_qkv_perm));
            } else {
                RUN(tp_all_reduce(m, m->mlp_out, (size_t)T * H));
                RUN(add_inplace_f16(m->residual, m->mlp_out, (long)T * H, s));
            }
            qkv_in_perm = next_ln && next_qkv_perm != nullptr;
            }
        }
        if (m->taps_enabled && m->taps) {
            if (sandwich)
                FH_CHECK_HIP(hipMemcpyAsync(m->taps + (size_t)li * c.max_tokens * H, m->residual_f32, (size_t)T * H * 4,
                                            hipMemcpyDeviceToDevice, s));
            else
                hipLaunchKernelGGL(f16_to_f32_kernel, dim3(cdiv((long)T * H, 256)), dim3(256), 0, s, m->residual,
                                   m->taps + (size_t)li * c.max_tokens * H, (long)T * H);
        }
    }
    m->taps_tokens = T;
    if (sh.num_sampled > 0) {
        // final norm + lm_head on the sampled rows only (same rows the reference packs,
        // qwen3_moe_forward_unified.rs:365-392)
        if (sandwich) {
            RUN(rms_norm_f32_to_f16(m->residual_f32, sampled_idx, m->final_norm, c.rms_eps, m->sampled_hidden, sh.num_sampled, H, s));
        } else {
            RUN(gather_rms_norm_f16(m->residual, sampled_idx, m->final_norm, c.rms_eps, m->sampled_hidden, sh.num_sampled, H, s));
        }
        const int Vn = m->vp_n, v0 = m->vp_v0;                  // this rank's vocabulary rows (all of them unless vocabulary parallel)
        RUN(f16t_gemm_f32out(m->sampled_hidden, m->lm_head_t, m->logits, sh.num_sampled, Vn, H, m->workspace,
                             m->workspace_bytes, s));
        if (greedy && !c.vocab_parallel) {
            // LogitsReturnPolicy::GreedyArgmax { token_mask, repetition_penalty } (model_executor.rs:109-150): the sparse
            // penalty rewrites the listed logits in place, then raw or masked argmax (traits.rs:1571-1591)
            if (gopts && gopts->row_offsets)
                RUN(apply_repetition_penalties_sparse_f32(m->logits, gopts->row_offsets, gopts->token_ids, gopts->penalties,
                                                          sh.num_sampled, c.vocab, s));
            RUN(argmax_rows_f32_ws_advance(m->logits, m->out_tokens, gopts ? gopts->mask : nullptr, gopts ? gopts->mask_len : 0,
                                           sh.num_sampled, c.vocab, m->workspace, m->workspace_bytes, advance, advance_fused, s));
        } else if (greedy) {
            // vocabulary parallel: penalty and mask act on the rank's slice (ids / mask bytes shifted by v0), the local first
            // maximum becomes a (logit, global id) pair, the pairs of all ranks are gathered and the first maximum in rank
            // order wins — the lowest id among equal logits, as on one GPU.  The merge also advances the decode state.
            if (gopts && gopts->row_offsets)
                RUN(apply_repetition_penalties_sparse_f32_shard(m->logits, gopts->row_offsets, gopts->token_ids, gopts->penalties,
                                                                sh.num_sampled, Vn, v0, s));
            const uint8_t* mask = gopts && gopts->mask ? gopts->mask + std::min(v0, gopts->mask_len) : nullptr;
            const int mask_len = gopts && gopts->mask ? std::max(0, gopts->mask_len - v0) : 0;
            RUN(argmax_rows_f32_ws(m->logits, m->out_tokens, mask, mask_len, sh.num_sampled, Vn, m->workspace, m->workspace_bytes, s));
            RUN(argmax_pairs_f32(m->logits, m->out_tokens, m->vp_pairs, sh.num_sampled, Vn, v0, mask, mask_len, s));
            RUN(tp_all_gather(m, m->vp_pairs, m->vp_gathered, (size_t)sh.num_sampled * 8));
            RUN(argmax_merge_ranks(m->vp_gathered, m->out_tokens, sh.num_sampled, c.tp_world, advance, s));
            if (advance && advance_fused) *advance_fused = 1;
        }
    }
#undef RUN
    return 0;
}

}  // namespace

extern "C" {

int ferrum_hip_model_local_vocab(const FerrumHipModel* m, int* vocab_start, int* vocab_count) {
    FH_REQUIRE(m, "model_local_vocab: null");
    if (vocab_start) *vocab_start = m->vp_v0;
    if (vocab_count) *vocab_count = m->vp_n;
    return 0;
}

int ferrum_hip_model_unified_forward(FerrumHipModel* m, const FerrumHipBatchItem* items, int num_items, int greedy,
                                     uint32_t* out_tokens, float* logits_out) {
    return ferrum_hip_model_unified_forward_ex(m, items, num_items, greedy, nullptr, out_tokens, logits_out);
}

int ferrum_hip_model_unified_forward_ex(FerrumHipModel* m, const FerrumHipBatchItem* items, int num_items, int greedy,
                                        const FerrumHipGreedyOptions* opts, uint32_t* out_tokens, float* logits_out) {
    FH_REQUIRE(m && m->finalized, "unified_forward: model not finalized");
    if (num_items <= 0) return 0;
    FH_REQUIRE(items, "unified_forward: null items");
    const FerrumHipModelConfig& c = m->cfg;
    FH_REQUIRE(num_items <= c.max_seqs, "unified_forward: %d items > max_seqs %d", num_items, c.max_seqs);
    // shape + admission
    StepShape sh{};
    sh.num_seqs = num_items;
    sh.pure_decode = true;
    std::vector<FerrumHipKvSlotRequest> reqs(num_items);
    for (int i = 0; i < num_items; i++) {
        const FerrumHipBatchItem& it = items[i];
        FH_REQUIRE(it.num_q_tokens > 0 && it.q_tokens, "unified_forward: item %d has no tokens", i);
        auto f = m->seqs.find(it.seq_id);
        int have = f == m->seqs.end() ? 0 : f->second.len;
        FH_REQUIRE(it.pos_offset == have, "unified_forward: item %d pos_offset=%d but cache holds %d tokens", i, it.pos_offset, have);
        FH_REQUIRE(it.pos_offset + it.num_q_tokens <= c.max_seq_len, "unified_forward: item %d exceeds max_seq_len %d", i, c.max_seq_len);
        for (int t = 0; t < it.num_q_tokens; t++)
            FH_REQUIRE(it.q_tokens[t] < (uint32_t)c.vocab, "unified_forward: item %d token %u out of vocab", i, it.q_tokens[t]);
        sh.m_total += it.num_q_tokens;
        sh.max_q_len = std::max(sh.max_q_len, it.num_q_tokens);
        sh.max_kv_len = std::max(sh.max_kv_len, it.pos_offset + it.num_q_tokens);
        if (it.num_q_tokens != 1) sh.pure_decode = false;
        if (it.is_final_chunk) sh.num_sampled++;
        reqs[i] = {it.seq_id, it.pos_offset + it.num_q_tokens, 0};
    }
    FH_REQUIRE(sh.m_total <= c.max_tokens, "unified_forward: %d tokens > max_tokens %d", sh.m_total, c.max_tokens);
    sh.all_single_token = sh.pure_decode;
    if (c.sliding_window > 0 && c.sliding_window_pattern == 0) sh.pure_decode = false;   // uniform window: no full-attention layer
    if (int rc = reserve(m, reqs.data(), num_items, nullptr)) return rc;

    // index block
    uint32_t* h_tok = reinterpret_cast<uint32_t*>(m->idx_host + m->il.tokens);
    uint32_t* h_cu = reinterpret_cast<uint32_t*>(m->idx_host + m->il.cu_seqlens);
    uint32_t* h_pos = reinterpret_cast<uint32_t*>(m->idx_host + m->il.pos_offsets);
    uint32_t* h_kvl = reinterpret_cast<uint32_t*>(m->idx_host + m->il.kv_lens);
    int32_t* h_samp = reinterpret_cast<int32_t*>(m->idx_host + m->il.sampled_idx);
    int32_t* h_bt = reinterpret_cast<int32_t*>(m->idx_host + m->il.block_tables);
    int t = 0, j = 0;
    for (int i = 0; i < num_items; i++) {
        const FerrumHipBatchItem& it = items[i];
        h_cu[i] = (uint32_t)t;
        memcpy(h_tok + t, it.q_tokens, (size_t)it.num_q_tokens * 4);
        t += it.num_q_tokens;
        h_pos[i] = (uint32_t)it.pos_offset;
        h_kvl[i] = (uint32_t)(it.pos_offset + it.num_q_tokens);
        if (it.is_final_chunk) h_samp[j++] = t - 1;
        const SeqState& st = m->seqs[it.seq_id];
        int32_t* row = h_bt + (size_t)i * m->max_blocks_per_seq;
        for (int b = 0; b < m->max_blocks_per_seq; b++) row[b] = b < (int)st.blocks.size() ? (int32_t)st.blocks[b] : 0;   // padded with 0 (paged_pool.rs:438-440)
    }
    h_cu[num_items] = (uint32_t)t;
    sh.single_prefix = 0;
    while (sh.single_prefix < num_items && items[sh.single_prefix].num_q_tokens == 1) sh.single_prefix++;
    sh.rest_min_q = 0;
    for (int i = sh.single_prefix; i < num_items; i++)
        sh.rest_min_q = i == sh.single_prefix ? items[i].num_q_tokens : std::min(sh.rest_min_q, items[i].num_q_tokens);
    // only the used prefix of the block-table region needs to travel
    size_t used = m->il.block_tables + (size_t)num_items * m->max_blocks_per_seq * 4;
    FH_CHECK_HIP(hipMemcpyAsync(m->idx_dev, m->idx_host, used, hipMemcpyHostToDevice, m->stream));
    GreedyDeviceOpts g{};
    if (greedy && opts && sh.num_sampled > 0) {
        // device copies of the greedy options live in a growable side buffer: [mask | row_offsets | token_ids | penalties]
        size_t total_ids = 0;
        if (opts->penalty_row_offsets) {
            FH_REQUIRE(opts->penalty_token_ids && opts->penalties, "unified_forward: incomplete repetition-penalty arrays");
            FH_REQUIRE(opts->penalty_row_offsets[0] == 0, "unified_forward: penalty_row_offsets must start at 0");
            for (int i = 0; i < sh.num_sampled; i++)
                FH_REQUIRE(opts->penalty_row_offsets[i + 1] >= opts->penalty_row_offsets[i], "unified_forward: penalty_row_offsets not monotone");
            total_ids = opts->penalty_row_offsets[sh.num_sampled];
        }
        const size_t mask_bytes = opts->valid_token_mask ? ((size_t)opts->mask_len + 15) / 16 * 16 : 0;
        const size_t off_ro = mask_bytes, off_ids = off_ro + ((size_t)sh.num_sampled + 1 + 3) / 4 * 16;
        const size_t off_pen = off_ids + (total_ids + 3) / 4 * 16, need = off_pen + ((size_t)sh.num_sampled + 3) / 4 * 16;
        if (m->greedy_opts_bytes < need) {
            if (m->greedy_opts_dev) (void)hipFree(m->greedy_opts_dev);
            m->greedy_opts_dev = nullptr;
            FH_CHECK_HIP(hipMalloc((void**)&m->greedy_opts_dev, need));
            m->greedy_opts_bytes = need;
        }
        uint8_t* base = m->greedy_opts_dev;
        if (opts->valid_token_mask) {
            FH_REQUIRE(opts->mask_len > 0, "unified_forward: mask_len=%d", opts->mask_len);
            FH_CHECK_HIP(hipMemcpyAsync(base, opts->valid_token_mask, (size_t)opts->mask_len, hipMemcpyHostToDevice, m->stream));
            g.mask = base;
            g.mask_len = opts->mask_len;
        }
        if (opts->penalty_row_offsets) {
            FH_CHECK_HIP(hipMemcpyAsync(base + off_ro, opts->penalty_row_offsets, ((size_t)sh.num_sampled + 1) * 4, hipMemcpyHostToDevice, m->stream));
            if (total_ids)
                FH_CHECK_HIP(hipMemcpyAsync(base + off_ids, opts->penalty_token_ids, total_ids * 4, hipMemcpyHostToDevice, m->stream));
            FH_CHECK_HIP(hipMemcpyAsync(base + off_pen, opts->penalties, (size_t)sh.num_sampled * 4, hipMemcpyHostToDevice, m->stream));
            g.row_offsets = reinterpret_cast<const uint32_t*>(base + off_ro);
            g.token_ids = reinterpret_cast<const uint32_t*>(base + off_ids);
            g.penalties = reinterpret_cast<const float*>(base + off_pen);
        }
    }
    if (int rc = enqueue_forward(m, sh, greedy != 0, &g)) return rc;
    if (sh.num_sampled > 0) {
        if (greedy && out_tokens)
            FH_CHECK_HIP(hipMemcpyAsync(out_tokens, m->out_tokens, (size_t)sh.num_sampled * 4, hipMemcpyDeviceToHost, m->stream));
        if (logits_out)
            FH_CHECK_HIP(hipMemcpyAsync(logits_out, m->logits, (size_t)sh.num_sampled * m->vp_n * 4, hipMemcpyDeviceToHost, m->stream));
    }
    FH_CHECK_HIP(hipStreamSynchronize(m->stream));
    for (int i = 0; i < num_items; i++) m->seqs[items[i].seq_id].len += items[i].num_q_tokens;
    return check_inlaunch_waits(m);
}

int ferrum_hip_model_decode_steps(FerrumHipModel* m, const uint64_t* seq_ids, const uint32_t* first_tokens, int n,
                                  int steps, uint32_t* out_tokens) {
    FH_REQUIRE(m && m->finalized && seq_ids && first_tokens, "decode_steps: bad argument");
    if (n <= 0 || steps <= 0) return 0;
    const FerrumHipModelConfig& c = m->cfg;
    FH_REQUIRE(n <= c.max_seqs && n <= 1024, "decode_steps: %d sequences > max_seqs %d (or 1024)", n, c.max_seqs);
    // admission for the whole run up front: block tables are then constant across the steps
    std::vector<FerrumHipKvSlotRequest> reqs(n);
    StepShape sh{};
    sh.m_total = sh.num_seqs = sh.num_sampled = n;
    sh.max_q_len = 1;
    // uniform sliding window: every layer takes the windowed varlen path (cu / pos live on the device like kv_lens);
    // a local/global pattern decides per layer inside enqueue_forward
    sh.pure_decode = !(c.sliding_window > 0 && c.sliding_window_pattern == 0);
    sh.all_single_token = true;
    for (int i = 0; i < n; i++) {
        auto f = m->seqs.find(seq_ids[i]);
        FH_REQUIRE(f != m->seqs.end() && f->second.len > 0, "decode_steps: sequence %llu has no prefilled context", (unsigned long long)seq_ids[i]);
        FH_REQUIRE(f->second.len + steps <= c.max_seq_len, "decode_steps: sequence %llu would exceed max_seq_len", (unsigned long long)seq_ids[i]);
        FH_REQUIRE(first_tokens[i] < (uint32_t)c.vocab, "decode_steps: token out of vocab");
        reqs[i] = {seq_ids[i], f->second.len + steps, 0};
        sh.max_kv_len = std::max(sh.max_kv_len, f->second.len + steps);
    }
    if (int rc = reserve(m, reqs.data(), n, nullptr)) return rc;
    if (m->history_cap < steps * n) {
        // the captured step holds the history pointer as a kernel argument: a graph recorded against the old buffer must
        // not be replayed (it would write the sampled ids into freed memory)
        drop_graph(m);
        FH_CHECK_HIP(hipStreamSynchronize(m->stream));
        if (m->history) (void)hipFree(m->history);
        m->history = nullptr;
        m->history_cap = 0;
        const int cap = std::max(steps * n, 4096);             // rarely regrown: 4096 ids cover 128 steps of 32 sequences
        FH_CHECK_HIP(hipMalloc((void**)&m->history, (size_t)cap * 4));
        m->history_cap = cap;
    }
    uint32_t* h_tok = reinterpret_cast<uint32_t*>(m->idx_host + m->il.tokens);
    uint32_t* h_cu = reinterpret_cast<uint32_t*>(m->idx_host + m->il.cu_seqlens);
    uint32_t* h_pos = reinterpret_cast<uint32_t*>(m->idx_host + m->il.pos_offsets);
    uint32_t* h_kvl = reinterpret_cast<uint32_t*>(m->idx_host + m->il.kv_lens);
    int32_t* h_samp = reinterpret_cast<int32_t*>(m->idx_host + m->il.sampled_idx);
    int32_t* h_bt = reinterpret_cast<int32_t*>(m->idx_host + m->il.block_tables);
    for (int i = 0; i < n; i++) {
        const SeqState& st = m->seqs[seq_ids[i]];
        h_tok[i] = first_tokens[i];
        h_cu[i] = (uint32_t)i;
        h_pos[i] = (uint32_t)st.len;
        h_kvl[i] = (uint32_t)st.len + 1;
        h_samp[i] = i;
        int32_t* row = h_bt + (size_t)i * m->max_blocks_per_seq;
        for (int b = 0; b < m->max_blocks_per_seq; b++) row[b] = b < (int)st.blocks.size() ? (int32_t)st.blocks[b] : 0;
    }
    h_cu[n] = (uint32_t)n;
    size_t used = m->il.block_tables + (size_t)n * m->max_blocks_per_seq * 4;
    FH_CHECK_HIP(hipMemcpyAsync(m->idx_dev, m->idx_host, used, hipMemcpyHostToDevice, m->stream));
    FH_CHECK_HIP(hipMemsetAsync(m->step_counter, 0, 8, m->stream));    // step index + the advance ticket

    auto enqueue_step = [&]() -> int {
        // the step's bookkeeping (sampled id → next input, positions, history) rides in the last stage of the argmax
        DecodeAdvance adv;
        adv.tokens = idx<uint32_t>(m, m->il.tokens); adv.pos_offsets = idx<uint32_t>(m, m->il.pos_offsets);
        adv.kv_lens = idx<uint32_t>(m, m->il.kv_lens); adv.history = m->history; adv.step_counter = m->step_counter;
        adv.ticket = reinterpret_cast<unsigned*>(m->step_counter + 1); adv.n = n;
        int fused = 0;
        if (int rc = enqueue_forward(m, sh, true, nullptr, &adv, &fused)) return rc;
        if (!fused) {                                            // narrow vocabularies take the single-kernel argmax
            hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(1024), 0, m->stream, m->out_tokens, adv.tokens, adv.pos_offsets,
                               adv.kv_lens, m->history, m->step_counter, n);
            FH_CHECK_LAUNCH();
        }
        return 0;
    };

    // One decode step is captured into a hipGraph (BackendGraph, capabilities.rs:35-70) and
    // replayed: every per-step quantity lives in device buffers.  A graph is reusable while
    // (n, grid-shaping max_kv_len bucket) stay the same.
    // Tensor parallel: RCCL and the one-shot peer reduce are stream-ordered device work and are captured with the step;
    // only the host-barrier loopback of the tests cannot be.
    const bool use_graph = !(c.tp_world > 1 && m->tp_loopback) && !m->taps_enabled && !knobs().no_graph;
    int kv_bucket = cdiv(sh.max_kv_len, 256) * 256;
    sh.max_kv_len = kv_bucket;
    bool graph_ok = use_graph && !m->graph_refused;
    if (graph_ok && (!m->graph_exec || m->graph_n != n || m->graph_max_kv != kv_bucket)) {
        drop_graph(m);
        FH_CHECK_HIP(hipStreamSynchronize(m->stream));
        form_hit(FORM_GRAPH_CAPTURE);
        // (relaxed mode under tensor parallelism: the collective library may touch the allocator inside the capture window)
        FH_CHECK_HIP(hipStreamBeginCapture(m->stream, c.tp_world > 1 ? hipStreamCaptureModeRelaxed : hipStreamCaptureModeThreadLocal));
        int rc = enqueue_step();
        hipError_t e = hipStreamEndCapture(m->stream, &m->graph);
        if (!rc && e == hipSuccess) e = hipGraphInstantiate(&m->graph_exec, m->graph, nullptr, nullptr, 0);
        if (rc || e != hipSuccess) {
            // A step with a collective inside that the runtime will not capture (tensor parallel only): run it eagerly from now
            // on rather than fail — the capture attempt enqueued nothing, so the device state is untouched.
            (void)hipGetLastError();
            drop_graph(m);
            if (c.tp_world <= 1) {
                if (rc) return rc;
                FH_CHECK_HIP(e);
            }
            fprintf(stderr, "[ferrum_hip] decode step could not be captured under tensor parallelism (%s): eager launches from here on\n",
                    rc ? fh::last_error() : hipGetErrorString(e));
            m->graph_refused = true;
            graph_ok = false;
        } else {
            m->graph_n = n;
            m->graph_max_kv = kv_bucket;
        }
    }
    if (graph_ok) {
        for (int st = 0; st < steps; st++) { FH_CHECK_HIP(hipGraphLaunch(m->graph_exec, m->stream)); form_hit(FORM_GRAPH_REPLAY); }
    } else {
        for (int st = 0; st < steps; st++)
            if (int rc = enqueue_step()) return rc;
    }
    if (out_tokens)
        FH_CHECK_HIP(hipMemcpyAsync(out_tokens, m->history, (size_t)steps * n * 4, hipMemcpyDeviceToHost, m->stream));
    FH_CHECK_HIP(hipStreamSynchronize(m->stream));
    for (int i = 0; i < n; i++) m->seqs[seq_ids[i]].len += steps;
    return check_inlaunch_waits(m);
}

// Bench instrumentation: average device time of ONE launch of a hot kernel, measured with HIP events
// on the model's stream.  The launch is repeated over every layer's weights (L distinct weight sets,
// far larger than the 256 MiB Infinity Cache) using the routing / index state the last forward left
// in the scratch buffers, `reps` rounds.  which: 0 MoE gate_up (+silu·mul), 1 MoE down,
// 2 paged decode attention, 3 qkv GEMM, 4 o GEMM, 5 lm_head GEMM, 8 MoE gate_up → down as ONE launch.  Returns the mean microseconds per
// launch and (for MoE) the number of 16-row expert blocks the routing holds.
int ferrum_hip_model_time_kernel(FerrumHipModel* m, int which, int n_seqs, int max_kv_len, int reps, float* avg_us,
                                 int* moe_blocks) {
    FH_REQUIRE(m && m->finalized && avg_us && reps > 0, "time_kernel: bad argument");
    const FerrumHipModelConfig& c = m->cfg;
    hipStream_t s = m->stream;
    const int T = n_seqs, H = c.hidden;
    hipEvent_t e0, e1;
    FH_CHECK_HIP(hipEventCreate(&e0));
    FH_CHECK_HIP(hipEventCreate(&e1));
    int blocks = 0;
    const int P = T * std::max(c.top_k, 1), E = c.num_experts > 0 ? m->ep_E : 0;
    const int32_t* eids = moe_ids(m);
    if (E > 0 && (which < 2 || which == 8)) {
        FH_REQUIRE(P <= 1024, "time_kernel: MoE timing uses the decode (inline-align) path, pairs=%d > 1024", P);
        std::vector<int32_t> ids(P);
        FH_CHECK_HIP(hipMemcpyAsync(ids.data(), eids, (size_t)P * 4, hipMemcpyDeviceToHost, s));
        FH_CHECK_HIP(hipStreamSynchronize(s));
        std::vector<int> cnt(E, 0);
        for (int v : ids) if (v >= 0 && v < E) cnt[v]++;
        for (int e = 0; e < E; e++) blocks += (cnt[e] + 15) / 16;
    }
    if (moe_blocks) *moe_blocks = blocks;
    FH_REQUIRE(T <= c.max_tokens, "time_kernel: %d rows > max_tokens %d", T, c.max_tokens);
    const int max_blocks = E > 0 ? std::min((P + E * 16) / 16, P / 16 + std::min(P, E)) : 0;
    int launches = 0, rc = 0;
    const bool em = E > 0 && m->moe_em_min_pairs_x8 > 0 && 8L * P >= (long)m->moe_em_min_pairs_x8 * c.num_experts;   // as moe_decode_gemms
    auto one = [&](int li) -> int {
        LayerWeights& L = m->layers[li];
        switch (which) {
        case 0: {
            if (em) return w4_gemm_moe_expert_major(L.exp_gate_up, m->norm_out, m->moe_act, eids, E, P, c.top_k, 1, s);
            return w4_gemm_moe_inline_align(L.exp_gate_up, m->norm_out, m->moe_act, eids, E, P, max_blocks, c.top_k, 1,
                                            m->sorted_ids, m->block_ids, m->total_post_pad, s);
        }
        case 1:
            if (em) return w4_gemm_moe_expert_major(L.exp_down, m->moe_act, m->moe_down, eids, E, P, 1, 0, s);
            return w4_gemm_moe(L.exp_down, m->moe_act, m->moe_down, m->sorted_ids, m->block_ids, m->total_post_pad, P, max_blocks, 1, 0, s);
        case 8: {   // gate_up → down as the step runs them: one merged launch (falls back to an error when the shapes do not take it)
            unsigned* cur = m->em2_arrive + (size_t)(launches & 1) * m->arrive_half_words;
            unsigned* nxt = m->em2_arrive + (size_t)((launches & 1) ^ 1) * m->arrive_half_words;
            int took = 0;
            // (as the step runs it: with the routing taken from the candidate lists the last forward's chain left behind)
            int Q = m->route_parts;
            const int tiles = (c.num_experts + 15) / 16;
            while (Q > 1 && (tiles % Q != 0 || Q > 8)) Q >>= 1;
            MoeRouteLists lists;
            lists.cand = m->route_cand; lists.stats = m->route_stats; lists.T = T; lists.Q = Q; lists.norm_topk = c.norm_topk_prob;
            lists.pub_expert_ids = m->expert_ids; lists.pub_expert_w = m->expert_w;
            const bool defer = knobs().decode_chain && Q > 1 && moe_deferred_merge_ok(m, L, T, Q);
            if (int r_ = w4_gemm_moe_expert_major_pair(L.exp_gate_up, L.exp_down, m->norm_out, m->moe_act, m->moe_down, eids, E, P, c.top_k,
                                                       cur, nxt, m->inlaunch_timeouts, &took, s, defer ? &lists : nullptr)) return r_;
            if (!took) { fh::set_error("time_kernel: the merged gate_up → down launch does not take these shapes"); return FERRUM_HIP_UNSUPPORTED; }
            return 0;
        }
        case 2: return paged_batched_decode_attention_f16(m->q_out, L.k_pool, L.v_pool, m->attn_out, idx<int32_t>(m, m->il.block_tables),
                                                          idx<uint32_t>(m, m->il.kv_lens), T, max_kv_len, c.num_heads, c.num_kv_heads,
                                                          c.head_dim, KV_BLOCK, m->max_blocks_per_seq, m->workspace, m->workspace_bytes, s);
        case 3: return w4_gemm_dense(L.qkv, m->norm_out, m->qkv_out, T, m->workspace, m->workspace_bytes, s);
        case 4: return w4_gemm_dense(L.o, m->attn_out, m->o_out, T, m->workspace, m->workspace_bytes, s);
        case 6: return w4_gemm_dense(L.gate_up, m->norm_out, m->gate_up_out, T, m->workspace, m->workspace_bytes, s);
        case 7: return w4_gemm_dense(L.down, m->act_out, m->mlp_out, T, m->workspace, m->workspace_bytes, s);
        case 5: return f16t_gemm_f32out(m->sampled_hidden, m->lm_head_t, m->logits, T, m->vp_n, H, m->workspace, m->workspace_bytes, s);
        }
        fh::set_error("time_kernel: which=%d", which);
        return FERRUM_HIP_INVALID;
    };
    FH_REQUIRE((which >= 2 && which != 8) || E > 0, "time_kernel: MoE kernel on a dense model");
    FH_REQUIRE(which < 6 || which == 8 || E == 0, "time_kernel: dense MLP kernel on a MoE model");
    if (which == 8) {
        FH_REQUIRE(em && knobs().moe_em2 && m->em2_arrive && !m->em2_failed, "time_kernel: the merged gate_up → down launch is not what this batch takes");
        FH_CHECK_HIP(hipMemsetAsync(m->em2_arrive, 0, 2 * m->arrive_half_words * sizeof(unsigned), s));
    }
    const bool same_layer = knobs().time_same_layer;   // experiment: weights resident in the Infinity Cache
    // warm-up round (code objects, TLBs), then the timed rounds
    for (int li = 0; li < c.num_layers && !rc; li++) { rc = one(li); if (which == 8) launches++; }
    if (which == 8 && (launches & 1)) { rc = rc ? rc : one(0); launches++; }      // (even count: the counter halves alternate)
    launches = 0;
    if (rc) return rc;
    FH_CHECK_HIP(hipEventRecord(e0, s));
    for (int r = 0; r < reps && !rc; r++)
        for (int li = 0; li < c.num_layers && !rc; li++) { rc = one(which == 5 || same_layer ? 0 : li); launches++; }
    FH_CHECK_HIP(hipEventRecord(e1, s));
    FH_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    FH_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    *avg_us = ms * 1000.0f / (float)launches;
    return 0;
}

int ferrum_hip_tp_loopback_create(FerrumHipTpLoopback** out, int world) {
    FH_REQUIRE(out && world >= 2 && world <= 8, "tp_loopback_create: world=%d", world);
    auto* lb = new FerrumHipTpLoopback();
    lb->world = world;
    if (hipMalloc((void**)&lb->bufs_dev, sizeof(void*) * 8) != hipSuccess) { delete lb; fh::set_error("tp_loopback_create: hipMalloc"); return 1; }
    *out = lb;
    return 0;
}
int ferrum_hip_tp_loopback_destroy(FerrumHipTpLoopback* lb) {
    if (lb) { (void)hipFree((void*)lb->bufs_dev); delete lb; }
    return 0;
}
int ferrum_hip_model_tp_attach_loopback(FerrumHipModel* m, FerrumHipTpLoopback* lb) {
    FH_REQUIRE(m && lb && m->cfg.tp_world == lb->world && m->cfg.tp_rank < lb->world, "tp_attach_loopback: world mismatch");
    m->tp_loopback = lb;
    return 0;
}

// RCCL rank of this model's tensor-parallel group (nccl_comm.rs:21-49): the communicator is owned by the model.
int ferrum_hip_model_tp_init(FerrumHipModel* m, const uint8_t id[128]) {
    FH_REQUIRE(m && id, "tp_init: null");
    if (m->cfg.tp_world <= 1) return 0;
    FerrumHipComm* c = nullptr;
    if (int rc = ferrum_hip_comm_create_rccl(&c, m->cfg.tp_world, m->cfg.tp_rank, id)) return rc;
    if (m->comm && m->comm_owned) ferrum_hip_comm_destroy(m->comm);
    m->comm = c;
    m->comm_owned = true;
    m->graph_refused = false;
    drop_graph(m);
    return 0;
}

// Attach a communicator created by the caller (ferrum_hip_comm_create_*); the model does not own it.  Also the way to
// re-read the all-reduce policy after ferrum_hip_debug_reload_knobs: a captured decode graph is dropped here.
int ferrum_hip_model_set_comm(FerrumHipModel* m, FerrumHipComm* comm) {
    FH_REQUIRE(m, "model_set_comm: null model");
    FH_REQUIRE(!comm || (ferrum_hip_comm_world_size(comm) == m->cfg.tp_world && ferrum_hip_comm_rank(comm) == m->cfg.tp_rank),
               "model_set_comm: communicator is rank %d of %d, model is rank %d of %d", comm ? ferrum_hip_comm_rank(comm) : 0,
               comm ? ferrum_hip_comm_world_size(comm) : 1, m->cfg.tp_rank, m->cfg.tp_world);
    if (m->comm && m->comm_owned) ferrum_hip_comm_destroy(m->comm);
    m->comm = comm;
    m->comm_owned = false;
    m->graph_refused = false;
    drop_graph(m);
    return 0;
}

}  // extern "C"
