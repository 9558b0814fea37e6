// C-ABI layer (include/ferrum_hip.h) over the gfx950 kernels: argument validation, handles,
// memory, and the native-operator descriptor.  Host-only code; compiled with hipcc.
#include <stdarg.h>

#include <mutex>

#include "../../include/ferrum_hip.h"
#include "block_allocator.h"
#include "common.h"
#include "kernels.h"
#include "knobs.h"

#include <atomic>

namespace fh {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

// ── development knobs (read once) and kernel-form accounting ────────────────
static Knobs g_knobs;
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static bool env_flag(const char* name) { const char* e = getenv(name); return e && atoi(e) != 0; }
void reload_knobs() {
    Knobs k;
    k.attn_no_flash = env_flag("FERRUM_HIP_ATTN_NO_FLASH");
    k.attn_flash_min_rows_set = getenv("FERRUM_HIP_ATTN_FLASH_MIN_ROWS") != nullptr;
    k.attn_flash_min_rows = env_int("FERRUM_HIP_ATTN_FLASH_MIN_ROWS", 512);
    k.attn_splits = env_int("FERRUM_HIP_ATTN_SPLITS", 0);
    k.attn_rs_min_wgs = env_int("FERRUM_HIP_ATTN_RS_MIN_WGS", 512);
    k.attn_no_rs = env_flag("FERRUM_HIP_ATTN_NO_RS");
    k.attn_narrow = env_flag("FERRUM_HIP_ATTN_NARROW");
    k.attn_flash32 = env_flag("FERRUM_HIP_ATTN_FLASH32");
    k.attn_no_resident = env_flag("FERRUM_HIP_ATTN_NO_RESIDENT");
    k.attn_resident_min_wgs = env_int("FERRUM_HIP_ATTN_RESIDENT_MIN_WGS", 128);
    k.moe_kw_pairs = env_int("FERRUM_HIP_MOE_KW_PAIRS", 16);
    k.moe_em2 = env_int("FERRUM_HIP_MOE_EM2", 1);
    k.decode_chain = env_int("FERRUM_HIP_DECODE_CHAIN", 1);
    k.moe_deferred_merge = env_int("FERRUM_HIP_MOE_DEFERRED_MERGE", 1);
    k.moe_bm2 = env_int("FERRUM_HIP_MOE_BM2", 0);
    k.dense_chain = env_int("FERRUM_HIP_DENSE_CHAIN", 1);
    k.chain_max_keys = env_int("FERRUM_HIP_CHAIN_MAX_KEYS", 4096);
    k.chain_split_keys = env_int("FERRUM_HIP_CHAIN_SPLIT_KEYS", 256);
    k.chain_attn_splits = env_int("FERRUM_HIP_CHAIN_ATTN_SPLITS", 0);
    k.route_gemm_topk = env_int("FERRUM_HIP_ROUTE_GEMM_TOPK", 1);
    k.chain_qkv_wide = env_int("FERRUM_HIP_CHAIN_QKV_WIDE", -1);
    k.chain_slots = env_int("FERRUM_HIP_CHAIN_SLOTS", 256);
    k.chain_o_half = env_int("FERRUM_HIP_CHAIN_O_HALF", 1);
    k.chain_qkv_half = env_int("FERRUM_HIP_CHAIN_QKV_HALF", 1);
    k.sandwich_wide = env_int("FERRUM_HIP_SANDWICH_WIDE", 1);
    k.chain_max_rows = std::min(128, env_int("FERRUM_HIP_CHAIN_MAX_ROWS", 128));
    k.w4_tile_min_m = env_int("FERRUM_HIP_W4_TILE_MIN_M", 0);
    k.w4_tile_wgs = env_int("FERRUM_HIP_W4_TILE_WGS", 256);
    k.w4_ldsa = env_int("FERRUM_HIP_W4_LDSA", 1);
    k.w4_ldsa_nw = env_int("FERRUM_HIP_W4_LDSA_NW", 0);
    k.w4_ldsa_s = env_int("FERRUM_HIP_W4_LDSA_S", 0);
    k.w4_ldsw = env_int("FERRUM_HIP_W4_LDSW", 0);
    k.w4_ldsk = env_int("FERRUM_HIP_W4_LDSK", 0);
    k.w4_big = env_int("FERRUM_HIP_W4_BIG", 0);
    k.w4_nt = env_int("FERRUM_HIP_W4_NT", 0);
    k.w4_w = env_int("FERRUM_HIP_W4_W", 0);
    k.lds_min_wgs = env_int("FERRUM_HIP_LDS_MIN_WGS", 128);
    k.lds_min_groups = env_int("FERRUM_HIP_LDS_MIN_GROUPS", 8);
    k.no_graph = getenv("FERRUM_HIP_NO_GRAPH") != nullptr;
    k.trace_launches = getenv("FERRUM_HIP_TRACE_LAUNCHES") != nullptr && k.no_graph;
    k.time_same_layer = getenv("FERRUM_HIP_TIME_SAME_LAYER") != nullptr;
    k.tp_oneshot = env_int("FERRUM_HIP_TP_ONESHOT", -1);
    k.tp_fused_norm = env_int("FERRUM_HIP_TP_FUSED_NORM", 1);
    g_knobs = k;
}
namespace { struct KnobsInit { KnobsInit() { reload_knobs(); } } g_knobs_init; }
const Knobs& knobs() { return g_knobs; }

static std::atomic<uint64_t> g_form_hits[FORM_COUNT];
void form_hit(Form f) { g_form_hits[f].fetch_add(1, std::memory_order_relaxed); }
static const char* const g_form_names[FORM_COUNT] = {
    "attn_flash", "attn_row_split", "attn_kv_wide", "attn_kv_narrow", "attn_fused_qkv_wide", "attn_fused_qkv_narrow",
    "attn_split_reduce", "w4_wgsplit", "w4_ldsa", "w4_tilep", "w4_slabs", "w4_slabs_lds", "w4_slabs_tile", "w4_rowsum",
    "moe_expert_major", "moe_inline_align", "moe_block16", "moe_tile64", "moe_tile32", "moe_tile_big", "moe_merge_route", "route_split",
    "route_fused", "route_gemm", "dense_slab_chain", "graph_capture", "graph_replay", "tp_allreduce_rccl",
    "tp_allreduce_loopback", "tp_allreduce_oneshot", "f16_dense_linear", "w4_fused_tail", "attn_resident", "w4_big", "w4_ldsk", "gather_columns",
    "perm_producer", "moe_expert_major_pair", "decode_chain", "moe_deferred_merge", "dense_chain", "tp_allreduce_norm_fused", "moe_block_major_pair", "chain_attn_kv_splits", "chain_qkv_wide", "route_gemm_topk"};
const char* form_name(int f) { return f >= 0 && f < FORM_COUNT ? g_form_names[f] : nullptr; }
}  // namespace fh

using namespace fh;

struct FerrumHipWorkspace {
    float* ptr = nullptr;
    size_t bytes = 0;
};

struct FerrumHipGptq {
    W4Device dev;
    bool symmetric = true;
    __half* gather_scratch = nullptr;   // act-order input gather buffer [m_cap, K]
    int gather_rows = 0;
    std::vector<void*> retired_scratch;  // outgrown gather buffers: a graph captured earlier may still hold them → freed with the handle
    // gemm_phase_batched: rotating (pinned host, device, event) slots for the per-call dispatch arrays
    struct DispatchSlot { int32_t* host = nullptr; int32_t* dev = nullptr; size_t cap = 0; hipEvent_t done = nullptr; };
    DispatchSlot slots[4];
    int next_slot = 0;
    // expert_major_pair: [2][E] arrival counters + the give-up word of the merged gate_up → down launch
    unsigned* pair_arrive = nullptr;
    int pair_experts = 0;
};

struct FerrumHipGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

#define H(p) reinterpret_cast<__half*>(p)
#define CH(p) reinterpret_cast<const __half*>(p)
#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

// ── native operator artifact ────────────────────────────────────────────────
static const FerrumNativeOperatorDescriptor g_desc = {1u, "ferrum_hip_decode", "1"};
int ferrum_native_op_init(void) { return 0; }
const FerrumNativeOperatorDescriptor* ferrum_native_op_descriptor(void) { return &g_desc; }

const char* ferrum_hip_last_error(void) { return fh::last_error(); }

// ── debug: which kernel forms ran, and re-reading the development knobs ─────
int ferrum_hip_debug_form_count(void) { return fh::FORM_COUNT; }
const char* ferrum_hip_debug_form_name(int form) { return fh::form_name(form); }
int ferrum_hip_debug_form_hits(uint64_t* hits, int capacity) {
    FH_REQUIRE(hits && capacity >= fh::FORM_COUNT, "debug_form_hits: need room for %d counters", (int)fh::FORM_COUNT);
    for (int i = 0; i < fh::FORM_COUNT; i++) hits[i] = fh::g_form_hits[i].load(std::memory_order_relaxed);
    return 0;
}
int ferrum_hip_debug_form_reset(void) {
    for (int i = 0; i < fh::FORM_COUNT; i++) fh::g_form_hits[i].store(0, std::memory_order_relaxed);
    return 0;
}
int ferrum_hip_debug_reload_knobs(void) { fh::reload_knobs(); return 0; }

int ferrum_hip_device_count(int* count) {
    FH_REQUIRE(count, "device_count: null output");
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) { *count = 0; fh::set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return 1; }
    return 0;
}
int ferrum_hip_set_device(int ordinal) { FH_CHECK_HIP(hipSetDevice(ordinal)); return 0; }
int ferrum_hip_stream_create(void** stream) {
    FH_REQUIRE(stream, "stream_create: null output");
    hipStream_t s;
    FH_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return 0;
}
int ferrum_hip_stream_destroy(void* stream) { FH_CHECK_HIP(hipStreamDestroy(ST(stream))); return 0; }
int ferrum_hip_stream_synchronize(void* stream) { FH_CHECK_HIP(hipStreamSynchronize(ST(stream))); return 0; }
int ferrum_hip_alloc(void** dev_ptr, size_t bytes) {
    FH_REQUIRE(dev_ptr, "alloc: null output");
    *dev_ptr = nullptr;
    if (bytes == 0) return 0;
    FH_CHECK_HIP(hipMalloc(dev_ptr, bytes));
    FH_CHECK_HIP(hipMemset(*dev_ptr, 0, bytes));
    return 0;
}
int ferrum_hip_free(void* dev_ptr) { if (dev_ptr) FH_CHECK_HIP(hipFree(dev_ptr)); return 0; }
int ferrum_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes == 0) return 0;
    FH_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ST(stream)));
    FH_CHECK_HIP(hipStreamSynchronize(ST(stream)));
    return 0;
}
int ferrum_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes == 0) return 0;
    FH_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ST(stream)));
    FH_CHECK_HIP(hipStreamSynchronize(ST(stream)));
    return 0;
}
int ferrum_hip_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes == 0) return 0;
    FH_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ST(stream)));
    return 0;
}
int ferrum_hip_memset_zero(void* p, size_t bytes, void* stream) {
    if (bytes == 0) return 0;
    FH_CHECK_HIP(hipMemsetAsync(p, 0, bytes, ST(stream)));
    return 0;
}

int ferrum_hip_workspace_create(FerrumHipWorkspace** ws, size_t bytes) {
    FH_REQUIRE(ws, "workspace_create: null output");
    auto* w = new FerrumHipWorkspace();
    if (bytes) {
        hipError_t e = hipMalloc((void**)&w->ptr, bytes);
        if (e != hipSuccess) { delete w; fh::set_error("workspace hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return 1; }
    }
    w->bytes = bytes;
    *ws = w;
    return 0;
}
int ferrum_hip_workspace_destroy(FerrumHipWorkspace* ws) {
    if (!ws) return 0;
    if (ws->ptr) (void)hipFree(ws->ptr);
    delete ws;
    return 0;
}

// ── norms / elementwise ─────────────────────────────────────────────────────
int ferrum_hip_rms_norm_f16(const void* x, const void* w, float eps, void* out, int tokens, int dim, void* stream) {
    FH_REQUIRE(tokens == 0 || (x && w && out), "rms_norm: null buffer");
    return rms_norm_f16(CH(x), CH(w), eps, H(out), tokens, dim, ST(stream));
}
int ferrum_hip_fused_add_rms_norm_f16(void* residual, const void* x, const void* w, float eps, void* out,
                                      int tokens, int dim, void* stream) {
    FH_REQUIRE(tokens == 0 || (residual && x && w && out), "fused_add_rms_norm: null buffer");
    return fused_add_rms_norm_f16(H(residual), CH(x), CH(w), eps, H(out), tokens, dim, ST(stream));
}
int ferrum_hip_embedding_lookup_f16(const void* table, const uint32_t* ids, void* out, int n_ids, int dim, void* stream) {
    FH_REQUIRE(n_ids == 0 || (table && ids && out), "embedding_lookup: null buffer");
    return embedding_lookup_f16(CH(table), ids, H(out), n_ids, dim, ST(stream));
}
int ferrum_hip_fused_silu_mul_split_f16(const void* gate_up, void* out, int tokens, int im, void* stream) {
    return fused_silu_mul_split_f16(CH(gate_up), H(out), tokens, im, ST(stream));
}
int ferrum_hip_fused_gelu_tanh_mul_split_f16(const void* gate_up, void* out, int tokens, int im, void* stream) {
    return fused_gelu_tanh_mul_split_f16(CH(gate_up), H(out), tokens, im, ST(stream));
}
int ferrum_hip_scale_inplace_f16(void* buf, float scale, size_t len, void* stream) {
    return scale_inplace_f16(H(buf), scale, (long)len, ST(stream));
}
int ferrum_hip_add_inplace_f16(void* residual, const void* x, size_t len, void* stream) {
    return add_inplace_f16(H(residual), CH(x), (long)len, ST(stream));
}
int ferrum_hip_add_bias_f16(void* data, const void* bias, int rows, int cols, void* stream) {
    return add_bias_f16(H(data), CH(bias), rows, cols, ST(stream));
}

int ferrum_hip_layer_norm_f16(const void* x, const void* gamma, const void* beta, float eps, void* out, int tokens, int dim, void* stream) {
    FH_REQUIRE(tokens == 0 || (x && gamma && beta && out), "layer_norm: null buffer");
    return layer_norm_f16(CH(x), CH(gamma), CH(beta), eps, H(out), tokens, dim, ST(stream));
}
int ferrum_hip_gelu_f16(const void* x, void* out, size_t len, void* stream) {
    FH_REQUIRE(len == 0 || (x && out), "gelu: null buffer");
    return gelu_f16(CH(x), H(out), (long)len, ST(stream));
}

// ── device timer (Backend::Timer, backend/timer.rs:88-109: record on the context's stream, elapsed after both fired) ──
int ferrum_hip_event_create(void** event) {
    FH_REQUIRE(event, "event_create: null");
    hipEvent_t e;
    FH_CHECK_HIP(hipEventCreate(&e));
    *event = (void*)e;
    return 0;
}
int ferrum_hip_event_destroy(void* event) {
    if (event) (void)hipEventDestroy((hipEvent_t)event);
    return 0;
}
int ferrum_hip_event_record(void* event, void* stream) {
    FH_REQUIRE(event, "event_record: null");
    FH_CHECK_HIP(hipEventRecord((hipEvent_t)event, ST(stream)));
    return 0;
}
int ferrum_hip_event_elapsed_ms(void* start, void* end, float* ms) {
    FH_REQUIRE(start && end && ms, "event_elapsed_ms: null");
    FH_CHECK_HIP(hipEventSynchronize((hipEvent_t)end));
    FH_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)end));
    return 0;
}

// ── dense GEMM ──────────────────────────────────────────────────────────────
int ferrum_hip_gemm_f16(const void* a, const void* b, void* out, int m, int n, int k, FerrumHipWorkspace* ws, void* stream) {
    FH_REQUIRE(m == 0 || (a && b && out), "gemm: null buffer");
    return f16_gemm(CH(a), CH(b), H(out), m, n, k, ws ? ws->ptr : nullptr, ws ? ws->bytes : 0, ST(stream));
}
int ferrum_hip_gemm_f16_f32out(const void* a, const void* b, float* out, int m, int n, int k, FerrumHipWorkspace* ws,
                               void* stream) {
    FH_REQUIRE(m == 0 || (a && b && out), "gemm: null buffer");
    return f16_gemm_f32out(CH(a), CH(b), out, m, n, k, ws ? ws->ptr : nullptr, ws ? ws->bytes : 0, ST(stream));
}

size_t ferrum_hip_dense_f16t_bytes(int n, int k) { return f16t_elems(n, k) * 2; }
int ferrum_hip_dense_repack_f16t(const void* w, void* out, int n, int k, void* stream) {
    FH_REQUIRE(w && out, "dense_repack_f16t: null buffer");
    return f16t_repack(CH(w), H(out), n, k, ST(stream));
}
int ferrum_hip_gemm_f16t(const void* a, const void* b, void* out, int m, int n, int k, FerrumHipWorkspace* ws, void* stream) {
    FH_REQUIRE(m == 0 || (a && b && out), "gemm_f16t: null buffer");
    return f16t_gemm(CH(a), CH(b), H(out), m, n, k, ws ? ws->ptr : nullptr, ws ? ws->bytes : 0, ST(stream));
}
int ferrum_hip_gemm_f16t_f32out(const void* a, const void* b, float* out, int m, int n, int k, FerrumHipWorkspace* ws,
                                void* stream) {
    FH_REQUIRE(m == 0 || (a && b && out), "gemm_f16t: null buffer");
    return f16t_gemm_f32out(CH(a), CH(b), out, m, n, k, ws ? ws->ptr : nullptr, ws ? ws->bytes : 0, ST(stream));
}

// ── GPTQ ────────────────────────────────────────────────────────────────────
static int upload(const void* host, size_t bytes, void** dev) {
    *dev = nullptr;
    if (!bytes) return 0;
    FH_CHECK_HIP(hipMalloc(dev, bytes));
    FH_CHECK_HIP(hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
    return 0;
}

static int check_gptq_args(int bits, int group_size, int k, int n) {
    if (bits != 4) { fh::set_error("gptq: only bits=4 supported (got %d)", bits); return FERRUM_HIP_UNSUPPORTED; }
    if (k <= 0 || n <= 0 || k % 128 != 0 || n % 8 != 0) {
        fh::set_error("gptq: unsupported shape K=%d N=%d (K %% 128, N %% 8 required)", k, n);
        return FERRUM_HIP_UNSUPPORTED;
    }
    if (group_size <= 0 || group_size % 128 != 0 || k % group_size != 0) {
        fh::set_error("gptq: unsupported group_size=%d for K=%d (multiple of 128 required)", group_size, k);
        return FERRUM_HIP_UNSUPPORTED;
    }
    return 0;
}

int ferrum_hip_gptq_load(FerrumHipGptq** handle, const int32_t* qweight, const float* scales, const int32_t* qzeros,
                         const int32_t* g_idx, const float* bias, int bits, int group_size, int k, int n) {
    FH_REQUIRE(handle && qweight && scales && qzeros, "gptq_load: null argument");
    if (int rc = check_gptq_args(bits, group_size, k, n)) return rc;
    W4HostPacked hp;
    if (int rc = w4_repack_host(qweight, scales, qzeros, g_idx, nullptr, group_size, k, n, &hp)) return rc;
    auto* g = new FerrumHipGptq();
    g->symmetric = hp.symmetric;
    g->dev.k = k; g->dev.n = n; g->dev.n64 = hp.n64; g->dev.G = hp.G; g->dev.num_experts = 1;
    int rc = upload(hp.qw.data(), hp.qw.size() * 4, (void**)&g->dev.qw);
    if (!rc) rc = upload(hp.sc.data(), hp.sc.size() * 2, (void**)&g->dev.sc);
    if (!rc && !hp.symmetric) rc = upload(hp.zp.data(), hp.zp.size() * 2, (void**)&g->dev.zp);
    if (!rc && !hp.perm.empty()) rc = upload(hp.perm.data(), hp.perm.size() * 4, (void**)&g->dev.perm);
    if (!rc && bias) {
        std::vector<uint16_t> bh(n);
        for (int i = 0; i < n; i++) { _Float16 h = (_Float16)bias[i]; memcpy(&bh[i], &h, 2); }
        rc = upload(bh.data(), bh.size() * 2, (void**)&g->dev.bias);
    }
    if (rc) { ferrum_hip_gptq_free(g); return rc; }
    *handle = g;
    return 0;
}

int ferrum_hip_gptq_load_stacked(FerrumHipGptq** handle, const int32_t* const* qweights, const float* const* scales,
                                 const int32_t* const* qzeros, const int32_t* g_idx, int bits, int group_size, int k,
                                 int n_per_expert, int num_experts, int fuse_gate_up) {
    FH_REQUIRE(handle && qweights && scales && qzeros && num_experts > 0, "gptq_load_stacked: null argument");
    if (int rc = check_gptq_args(bits, group_size, k, n_per_expert)) return rc;
    const int n = n_per_expert;
    std::vector<int32_t> col_perm;
    if (fuse_gate_up) {
        if (n % 64 != 0) { fh::set_error("gptq_load_stacked: fused gate_up needs N %% 64 == 0 (N=%d)", n); return FERRUM_HIP_UNSUPPORTED; }
        // supertile st = [gate 32st..32st+31 | up I+32st..I+32st+31]
        const int I = n / 2;
        col_perm.resize(n);
        for (int st = 0; st < n / 64; st++)
            for (int c = 0; c < 64; c++) col_perm[st * 64 + c] = c < 32 ? st * 32 + c : I + st * 32 + (c - 32);
    }
    auto* g = new FerrumHipGptq();
    g->dev.k = k; g->dev.n = n; g->dev.num_experts = num_experts; g->dev.fused_gate_up = fuse_gate_up != 0;
    std::vector<uint32_t> all_qw;
    std::vector<uint16_t> all_sc, all_zp;
    bool any_asym = false;
    std::vector<W4HostPacked> packed(num_experts);
    for (int e = 0; e < num_experts; e++) {
        // one g_idx for the whole stack (capabilities.rs:180-189; cuda/quant.rs:862 ff. samples expert 0's): act-order rows
        // are packed in sorted-group order and the phase entry points gather the input columns to match
        if (int rc = w4_repack_host(qweights[e], scales[e], qzeros[e], g_idx, fuse_gate_up ? col_perm.data() : nullptr,
                                    group_size, k, n, &packed[e])) { delete g; return rc; }
        any_asym |= !packed[e].symmetric;
    }
    g->symmetric = !any_asym;
    g->dev.n64 = packed[0].n64; g->dev.G = packed[0].G;
    for (int e = 0; e < num_experts; e++) {
        all_qw.insert(all_qw.end(), packed[e].qw.begin(), packed[e].qw.end());
        all_sc.insert(all_sc.end(), packed[e].sc.begin(), packed[e].sc.end());
        if (any_asym) {
            if (packed[e].symmetric) {   // materialise zero point 8
                _Float16 h8 = (_Float16)8.0f; uint16_t u; memcpy(&u, &h8, 2);
                all_zp.insert(all_zp.end(), packed[e].sc.size(), u);
            } else {
                all_zp.insert(all_zp.end(), packed[e].zp.begin(), packed[e].zp.end());
            }
        }
    }
    int rc = upload(all_qw.data(), all_qw.size() * 4, (void**)&g->dev.qw);
    if (!rc) rc = upload(all_sc.data(), all_sc.size() * 2, (void**)&g->dev.sc);
    if (!rc && any_asym) rc = upload(all_zp.data(), all_zp.size() * 2, (void**)&g->dev.zp);
    if (!rc && !packed[0].perm.empty()) rc = upload(packed[0].perm.data(), packed[0].perm.size() * 4, (void**)&g->dev.perm);
    if (rc) { ferrum_hip_gptq_free(g); return rc; }
    *handle = g;
    return 0;
}

// Act-order stack: the phase entry points read gathered input rows x'[r][j] = x[r][perm[j]] (`rows` input rows of K columns).
// The scratch grows outside of graph capture only; an outgrown buffer stays alive with the handle (a captured graph may hold it).
static int stack_input(const FerrumHipGptq* stack, const void* input, int rows, hipStream_t s, const __half** x) {
    *x = CH(input);
    if (!stack->dev.perm || rows <= 0) return 0;
    auto* g = const_cast<FerrumHipGptq*>(stack);
    if (g->gather_rows < rows) {
        if (g->gather_scratch) g->retired_scratch.push_back(g->gather_scratch);
        g->gather_scratch = nullptr;
        FH_CHECK_HIP(hipMalloc((void**)&g->gather_scratch, (size_t)rows * g->dev.k * 2));
        g->gather_rows = rows;
    }
    if (int rc = gather_columns_f16(CH(input), g->dev.perm, g->gather_scratch, rows, g->dev.k, s)) return rc;
    *x = g->gather_scratch;
    return 0;
}
static inline int input_rows(int prob_m, int top_k) { return (prob_m + top_k - 1) / top_k; }      // pair p reads row p / top_k

int ferrum_hip_gptq_free(FerrumHipGptq* g) {
    if (!g) return 0;
    if (g->dev.qw) (void)hipFree(g->dev.qw);
    if (g->dev.sc) (void)hipFree(g->dev.sc);
    if (g->dev.zp) (void)hipFree(g->dev.zp);
    if (g->dev.perm) (void)hipFree(g->dev.perm);
    if (g->dev.bias) (void)hipFree(g->dev.bias);
    if (g->gather_scratch) (void)hipFree(g->gather_scratch);
    for (void* p : g->retired_scratch) (void)hipFree(p);
    if (g->pair_arrive) (void)hipFree(g->pair_arrive);
    for (auto& sl : g->slots) {
        if (sl.done) { (void)hipEventSynchronize(sl.done); (void)hipEventDestroy(sl.done); }
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.dev) (void)hipFree(sl.dev);
    }
    delete g;
    return 0;
}

int ferrum_hip_gptq_info(const FerrumHipGptq* g, int* k, int* n, int* num_experts, int* symmetric) {
    FH_REQUIRE(g, "gptq_info: null handle");
    if (k) *k = g->dev.k;
    if (n) *n = g->dev.n;
    if (num_experts) *num_experts = g->dev.num_experts;
    if (symmetric) *symmetric = g->symmetric ? 1 : 0;
    return 0;
}

int ferrum_hip_gptq_linear_forward_f16(const FerrumHipGptq* handle, const void* in, void* out, int m,
                                       FerrumHipWorkspace* ws, void* stream) {
    FH_REQUIRE(handle && (m == 0 || (in && out)), "gptq_linear_forward: null argument");
    FH_REQUIRE(handle->dev.num_experts == 1, "gptq_linear_forward: handle is an expert stack");
    auto* g = const_cast<FerrumHipGptq*>(handle);
    const __half* x = CH(in);
    if (g->dev.perm) {
        // act-order: gather input columns first (cuda/quant.rs:434).  The scratch grows outside
        // of graph capture only (first call with a larger m); the outgrown buffer stays allocated until the handle is
        // freed, because a graph captured at the smaller m holds its address.
        if (g->gather_rows < m) {
            if (g->gather_scratch) g->retired_scratch.push_back(g->gather_scratch);
            g->gather_scratch = nullptr;
            FH_CHECK_HIP(hipMalloc((void**)&g->gather_scratch, (size_t)m * g->dev.k * 2));
            g->gather_rows = m;
        }
        if (int rc = gather_columns_f16(x, g->dev.perm, g->gather_scratch, m, g->dev.k, ST(stream))) return rc;
        x = g->gather_scratch;
    }
    // the bias (if any) is added in the GEMM epilogue / split-K reduce (W4Device::bias)
    return w4_gemm_dense(g->dev, x, H(out), m, ws ? ws->ptr : nullptr, ws ? ws->bytes : 0, ST(stream));
}

int ferrum_hip_moe_gemm_phase_f16(const FerrumHipGptq* stack, const void* input, const int32_t* sorted_token_ids,
                                  const int32_t* expert_ids, const int32_t* num_tokens_past_padded, void* output,
                                  int prob_m, int moe_block_size, int top_k, int max_blocks, int fused_silu_mul,
                                  void* stream) {
    FH_REQUIRE(stack && input && sorted_token_ids && expert_ids && num_tokens_past_padded && output,
               "moe_gemm_phase: null argument");
    if (moe_block_size != 16 && moe_block_size != 32 && moe_block_size != 64 && moe_block_size != 96 && moe_block_size != 128) {
        fh::set_error("moe_gemm_phase: moe_block_size=%d unsupported (16, 32, 64, 96 or 128)", moe_block_size);
        return FERRUM_HIP_UNSUPPORTED;
    }
    if (moe_block_size >= 96 && stack->dev.G % 2 != 0) {
        fh::set_error("moe_gemm_phase: 96- / 128-row blocks need an even number of 128-wide quant groups (K=%d)", stack->dev.k);
        return FERRUM_HIP_UNSUPPORTED;
    }
    FH_REQUIRE(top_k >= 1, "moe_gemm_phase: top_k=%d", top_k);
    FH_REQUIRE(!fused_silu_mul || stack->dev.fused_gate_up, "moe_gemm_phase: fused epilogue needs a stack loaded with fuse_gate_up");
    FH_REQUIRE(fused_silu_mul || !stack->dev.fused_gate_up, "moe_gemm_phase: stack was loaded with fuse_gate_up; plain output is column-permuted");
    const __half* x = nullptr;
    if (int rc = stack_input(stack, input, input_rows(prob_m, top_k), ST(stream), &x)) return rc;
    if (moe_block_size >= 32)   // prefill-sized batches: 128-, 64- or 32-row LDS tiles
        return w4_gemm_moe_tile(stack->dev, x, H(output), sorted_token_ids, expert_ids, num_tokens_past_padded, prob_m,
                                max_blocks, moe_block_size, top_k, fused_silu_mul, ST(stream));
    return w4_gemm_moe(stack->dev, x, H(output), sorted_token_ids, expert_ids, num_tokens_past_padded, prob_m,
                       max_blocks, top_k, fused_silu_mul, ST(stream));
}

int ferrum_hip_moe_gemm_phase_inline_align_f16(const FerrumHipGptq* stack, const void* input,
                                               const int32_t* expert_ids_per_pair, void* output, int prob_m,
                                               int num_experts, int top_k, int max_blocks, int fused_silu_mul,
                                               void* stream) {
    FH_REQUIRE(stack && input && expert_ids_per_pair && output, "moe_gemm_phase_inline_align: null argument");
    FH_REQUIRE(top_k >= 1, "moe_gemm_phase_inline_align: top_k=%d", top_k);
    FH_REQUIRE(!fused_silu_mul || stack->dev.fused_gate_up, "moe_gemm_phase_inline_align: fused epilogue needs a stack loaded with fuse_gate_up");
    FH_REQUIRE(fused_silu_mul || !stack->dev.fused_gate_up, "moe_gemm_phase_inline_align: stack was loaded with fuse_gate_up; plain output is column-permuted");
    if (prob_m > 1024) { fh::set_error("moe_gemm_phase_inline_align: prob_m=%d > 1024 (use moe_align_block_size + moe_gemm_phase)", prob_m); return FERRUM_HIP_UNSUPPORTED; }
    const __half* x = nullptr;
    if (int rc = stack_input(stack, input, input_rows(prob_m, top_k), ST(stream), &x)) return rc;
    return w4_gemm_moe_inline_align(stack->dev, x, H(output), expert_ids_per_pair, num_experts, prob_m, max_blocks,
                                    top_k, fused_silu_mul, nullptr, nullptr, nullptr, ST(stream));
}

int ferrum_hip_moe_gemm_phase_expert_major_f16(const FerrumHipGptq* stack, const void* input,
                                               const int32_t* expert_ids_per_pair, void* output, int prob_m,
                                               int num_experts, int top_k, int fused_silu_mul, void* stream) {
    FH_REQUIRE(stack && input && expert_ids_per_pair && output, "moe_gemm_phase_expert_major: null argument");
    FH_REQUIRE(top_k >= 1 && num_experts >= 1, "moe_gemm_phase_expert_major: top_k=%d num_experts=%d", top_k, num_experts);
    FH_REQUIRE(!fused_silu_mul || stack->dev.fused_gate_up, "moe_gemm_phase_expert_major: fused epilogue needs a stack loaded with fuse_gate_up");
    FH_REQUIRE(fused_silu_mul || !stack->dev.fused_gate_up, "moe_gemm_phase_expert_major: stack was loaded with fuse_gate_up; plain output is column-permuted");
    if (prob_m > 1024) { fh::set_error("moe_gemm_phase_expert_major: prob_m=%d > 1024 (use moe_align_block_size + moe_gemm_phase)", prob_m); return FERRUM_HIP_UNSUPPORTED; }
    const __half* x = nullptr;
    if (int rc = stack_input(stack, input, input_rows(prob_m, top_k), ST(stream), &x)) return rc;
    return w4_gemm_moe_expert_major(stack->dev, x, H(output), expert_ids_per_pair, num_experts, prob_m, top_k,
                                    fused_silu_mul, ST(stream));
}

int ferrum_hip_moe_gemm_phase_expert_major_pair_f16(FerrumHipGptq* gate_up_stack, const FerrumHipGptq* down_stack, const void* input,
                                                    const int32_t* expert_ids_per_pair, void* act_out, void* output, int prob_m,
                                                    int num_experts, int top_k, void* stream) {
    FH_REQUIRE(gate_up_stack && down_stack && input && expert_ids_per_pair && act_out && output, "moe_gemm_phase_expert_major_pair: null argument");
    FH_REQUIRE(top_k >= 1 && num_experts >= 1, "moe_gemm_phase_expert_major_pair: top_k=%d num_experts=%d", top_k, num_experts);
    FH_REQUIRE(gate_up_stack->dev.fused_gate_up && !down_stack->dev.fused_gate_up,
               "moe_gemm_phase_expert_major_pair: gate_up stack must be loaded with fuse_gate_up, the down stack without");
    if (prob_m > 1024) { fh::set_error("moe_gemm_phase_expert_major_pair: prob_m=%d > 1024", prob_m); return FERRUM_HIP_UNSUPPORTED; }
    if (down_stack->dev.perm) {       // the gated activations would need a gather between the two GEMMs of the launch
        fh::set_error("moe_gemm_phase_expert_major_pair: act-order down stack (run the two phases separately)");
        return FERRUM_HIP_UNSUPPORTED;
    }
    const __half* x_in = nullptr;
    if (int rc = stack_input(gate_up_stack, input, input_rows(prob_m, top_k), ST(stream), &x_in)) return rc;
    FerrumHipGptq* g = gate_up_stack;
    if (g->pair_experts < num_experts) {
        // (a captured graph may still hold the old counters: retired with the handle, like outgrown gather buffers)
        if (g->pair_arrive) g->retired_scratch.push_back(g->pair_arrive);
        g->pair_arrive = nullptr;
        const size_t words = (size_t)2 * num_experts * MOE_PAIR_COUNTER_STRIDE + 4;
        FH_CHECK_HIP(hipMalloc((void**)&g->pair_arrive, words * sizeof(unsigned)));
        FH_CHECK_HIP(hipMemset(g->pair_arrive, 0, words * sizeof(unsigned)));
        g->pair_experts = num_experts;
    }
    // op level: the counters are zeroed by a memset node in front of every launch (a captured call replays with the same
    // pointers, so the runner's alternating halves do not apply); the give-up word is checked by ferrum_hip_moe_pair_status
    unsigned* arrive = g->pair_arrive;
    const size_t half = (size_t)g->pair_experts * MOE_PAIR_COUNTER_STRIDE;
    FH_CHECK_HIP(hipMemsetAsync(arrive, 0, half * sizeof(unsigned), ST(stream)));
    int took = 0;
    if (int rc = w4_gemm_moe_expert_major_pair(g->dev, down_stack->dev, x_in, H(act_out), H(output), expert_ids_per_pair, num_experts,
                                               prob_m, top_k, arrive, arrive + half, arrive + 2 * half, &took, ST(stream)))
        return rc;
    if (!took) { fh::set_error("moe_gemm_phase_expert_major_pair: shapes not taken by the merged form"); return FERRUM_HIP_UNSUPPORTED; }
    return 0;
}

/* ≤ 64 pairs (decode at c ≤ 8): the same pair of phases as ONE block-major launch — (gate_up + down tiles) × 16-row blocks,
 * four waves per tile splitting K with the whole tile in flight, one arrival counter per block. */
int ferrum_hip_moe_gemm_phase_block_major_pair_f16(FerrumHipGptq* gate_up_stack, const FerrumHipGptq* down_stack, const void* input,
                                                   const int32_t* expert_ids_per_pair, void* act_out, void* output, int prob_m,
                                                   int num_experts, int top_k, int max_blocks, void* stream) {
    FH_REQUIRE(gate_up_stack && down_stack && input && expert_ids_per_pair && act_out && output, "moe_gemm_phase_block_major_pair: null argument");
    FH_REQUIRE(top_k >= 1 && num_experts >= 1 && max_blocks >= 1, "moe_gemm_phase_block_major_pair: top_k=%d num_experts=%d max_blocks=%d", top_k,
               num_experts, max_blocks);
    FH_REQUIRE(gate_up_stack->dev.fused_gate_up && !down_stack->dev.fused_gate_up,
               "moe_gemm_phase_block_major_pair: gate_up stack must be loaded with fuse_gate_up, the down stack without");
    if (prob_m > 64) { fh::set_error("moe_gemm_phase_block_major_pair: prob_m=%d > 64", prob_m); return FERRUM_HIP_UNSUPPORTED; }
    if (down_stack->dev.perm) {
        fh::set_error("moe_gemm_phase_block_major_pair: act-order down stack (run the two phases separately)");
        return FERRUM_HIP_UNSUPPORTED;
    }
    const __half* x_in = nullptr;
    if (int rc = stack_input(gate_up_stack, input, input_rows(prob_m, top_k), ST(stream), &x_in)) return rc;
    FerrumHipGptq* g = gate_up_stack;
    if (g->pair_experts < num_experts) {
        if (g->pair_arrive) g->retired_scratch.push_back(g->pair_arrive);
        g->pair_arrive = nullptr;
        const size_t words = (size_t)2 * num_experts * MOE_PAIR_COUNTER_STRIDE + 4;
        FH_CHECK_HIP(hipMalloc((void**)&g->pair_arrive, words * sizeof(unsigned)));
        FH_CHECK_HIP(hipMemset(g->pair_arrive, 0, words * sizeof(unsigned)));
        g->pair_experts = num_experts;
    }
    unsigned* arrive = g->pair_arrive;
    const size_t half = (size_t)g->pair_experts * MOE_PAIR_COUNTER_STRIDE;
    FH_CHECK_HIP(hipMemsetAsync(arrive, 0, half * sizeof(unsigned), ST(stream)));
    int took = 0;
    if (int rc = w4_gemm_moe_block_major_pair(g->dev, down_stack->dev, x_in, H(act_out), H(output), expert_ids_per_pair, num_experts, prob_m,
                                              std::min(max_blocks, num_experts), top_k, arrive, arrive + half, arrive + 2 * half, &took,
                                              ST(stream)))
        return rc;
    if (!took) { fh::set_error("moe_gemm_phase_block_major_pair: shapes not taken by the merged form"); return FERRUM_HIP_UNSUPPORTED; }
    return 0;
}

/* number of in-launch waits of the stack's merged launches that gave up so far (0 = every hand-off completed) */
int ferrum_hip_moe_pair_status(const FerrumHipGptq* gate_up_stack, unsigned* timeouts) {
    FH_REQUIRE(gate_up_stack && timeouts, "moe_pair_status: null argument");
    *timeouts = 0;
    if (!gate_up_stack->pair_arrive) return 0;
    FH_CHECK_HIP(hipMemcpy(timeouts, gate_up_stack->pair_arrive + 2 * (size_t)gate_up_stack->pair_experts * MOE_PAIR_COUNTER_STRIDE, sizeof(unsigned), hipMemcpyDeviceToHost));
    return 0;
}

int ferrum_hip_sandwich_add_rms_norm_f32(const void* branch_f16, const void* w_branch, float* residual_f32, const void* w_next,
                                         float eps, void* norm_out_f16, int tokens, int dim, void* stream) {
    FH_REQUIRE(tokens == 0 || (branch_f16 && w_branch && residual_f32 && (!w_next || norm_out_f16)), "sandwich_add_rms_norm_f32: null buffer");
    return sandwich_add_rms_norm_f32(CH(branch_f16), CH(w_branch), residual_f32, CH(w_next), eps, H(norm_out_f16), tokens, dim,
                                     ST(stream));
}
int ferrum_hip_rms_norm_f32_to_f16(const float* x_f32, const int32_t* row_idx, const void* w, float eps, void* out_f16, int n_rows,
                                   int dim, void* stream) {
    FH_REQUIRE(n_rows == 0 || (x_f32 && w && out_f16), "rms_norm_f32_to_f16: null buffer");
    return rms_norm_f32_to_f16(x_f32, row_idx, CH(w), eps, H(out_f16), n_rows, dim, ST(stream));
}

// ── contiguous-KV lane of the core trait ─────────────────────────────────────
int ferrum_hip_split_qkv_f16(const void* qkv, void* q, void* k, void* v, int tokens, int q_dim, int kv_dim, void* stream) {
    FH_REQUIRE(tokens == 0 || (qkv && q && k && v), "split_qkv: null buffer");
    return split_qkv_f16(CH(qkv), H(q), H(k), H(v), tokens, q_dim, kv_dim, ST(stream));
}
int ferrum_hip_qk_norm_rope_f16(const void* input, const void* norm_w, const float* cos_tab, const float* sin_tab, void* output,
                                int tokens, int heads, int head_dim, int pos_offset, float eps, int mode, void* stream) {
    FH_REQUIRE(tokens == 0 || (input && output), "qk_norm_rope: null buffer");
    FH_REQUIRE(mode == 0 || (cos_tab && sin_tab), "qk_norm_rope: rope tables missing");
    FH_REQUIRE(mode != 1 || norm_w, "qk_norm_rope: norm weights missing for mode 1");
    return qk_norm_rope_f16(CH(input), CH(norm_w), cos_tab, sin_tab, H(output), tokens, heads, head_dim, pos_offset, eps, mode,
                            ST(stream));
}
int ferrum_hip_kv_cache_append_head_major_f16(void* cache_k, void* cache_v, int cache_len, int cache_capacity, const void* new_k,
                                              const void* new_v, int new_tokens, int nkv, int head_dim, void* stream) {
    FH_REQUIRE(new_tokens == 0 || (cache_k && cache_v && new_k && new_v), "kv_cache_append_head_major: null buffer");
    return kv_cache_append_head_major_f16(H(cache_k), H(cache_v), cache_len, cache_capacity, CH(new_k), CH(new_v), new_tokens, nkv,
                                          head_dim, ST(stream));
}
int ferrum_hip_transpose_head_to_token_f16(const void* src, void* dst, int tokens, int heads, int dim, void* stream) {
    FH_REQUIRE(tokens == 0 || (src && dst), "transpose_head_to_token: null buffer");
    return transpose_head_to_token_f16(CH(src), H(dst), tokens, heads, dim, ST(stream));
}
int ferrum_hip_transpose_token_to_head_f16(const void* src, void* dst, int tokens, int heads, int dim, void* stream) {
    FH_REQUIRE(tokens == 0 || (src && dst), "transpose_token_to_head: null buffer");
    return transpose_token_to_head_f16(CH(src), H(dst), tokens, heads, dim, ST(stream));
}
int ferrum_hip_copy_slice_f16(const void* src, size_t src_offset, void* dst, size_t dst_offset, size_t len, void* stream) {
    FH_REQUIRE(len == 0 || (src && dst), "copy_slice: null buffer");
    return copy_slice_f16(CH(src), (long)src_offset, H(dst), (long)dst_offset, (long)len, ST(stream));
}
int ferrum_hip_scaled_add_inplace_f16(void* dst, const void* src, float scale, size_t len, void* stream) {
    FH_REQUIRE(len == 0 || (src && dst), "scaled_add_inplace: null buffer");
    return scaled_add_inplace_f16(H(dst), CH(src), scale, (long)len, ST(stream));
}
int ferrum_hip_flash_attention_f16(const void* q, const void* k, const void* v, void* out, int batch, int q_len, int kv_len,
                                   int pos_offset, int num_heads, int num_kv_heads, int head_dim, int causal, float scale,
                                   int kv_seq_stride, int sliding_window, void* stream) {
    FH_REQUIRE(q && k && v && out, "flash_attention: null buffer");
    if (batch != 1) { fh::set_error("flash_attention: batch=%d (the reference's CPU and decode paths use batch 1)", batch); return FERRUM_HIP_UNSUPPORTED; }
    return flash_attention_contig_f16(CH(q), CH(k), CH(v), H(out), q_len, kv_len, causal, pos_offset, num_heads, num_kv_heads, head_dim,
                                      scale, kv_seq_stride, sliding_window, ST(stream));
}

// ── paged KV ────────────────────────────────────────────────────────────────
size_t ferrum_hip_paged_pool_bytes(int num_blocks, int kv_heads, int head_dim) {
    return (size_t)num_blocks * kv_heads * 16 * head_dim * 2;
}

int ferrum_hip_split_qkv_norm_rope_into_paged_cache_varlen_f16(
    const void* qkv, const void* q_norm_w, const void* k_norm_w, const float* cos_tab, const float* sin_tab,
    void* q_out, void* cache_k, void* cache_v, const uint32_t* cu_seqlens_q, const uint32_t* pos_offsets,
    const int32_t* block_tables, int num_seqs, int m_total, int q_heads, int kv_heads, int head_dim, float eps,
    int qk_mode, int block_size, int max_blocks_per_seq, void* stream) {
    FH_REQUIRE(m_total == 0 || (qkv && q_out && cache_k && cache_v && cu_seqlens_q && pos_offsets && block_tables),
               "split_qkv_norm_rope_into_paged_cache_varlen: null buffer");
    FH_REQUIRE(qk_mode == 0 || (cos_tab && sin_tab), "split_qkv_norm_rope: rope tables missing");
    FH_REQUIRE(qk_mode != 1 || (q_norm_w && k_norm_w), "split_qkv_norm_rope: norm weights missing for qk_mode 1");
    return split_qkv_norm_rope_into_paged_cache_varlen_f16(CH(qkv), CH(q_norm_w), CH(k_norm_w), cos_tab, sin_tab,
                                                           H(q_out), H(cache_k), H(cache_v), cu_seqlens_q, pos_offsets,
                                                           block_tables, num_seqs, m_total, q_heads, kv_heads, head_dim,
                                                           eps, qk_mode, block_size, max_blocks_per_seq, ST(stream));
}

int ferrum_hip_paged_varlen_attention_f16(const void* q, const void* k_pool, const void* v_pool, void* out,
                                          const uint32_t* cu_seqlens_q, const uint32_t* pos_offsets,
                                          const int32_t* block_tables, int num_seqs, int total_q_tokens, int max_kv_len,
                                          int num_heads, int num_kv_heads, int head_dim, int sliding_window,
                                          int block_size, int max_num_blocks_per_seq, int max_q_len,
                                          FerrumHipWorkspace* ws, void* stream) {
    FH_REQUIRE(total_q_tokens == 0 || (q && k_pool && v_pool && out && cu_seqlens_q && pos_offsets && block_tables),
               "paged_varlen_attention: null buffer");
    return paged_varlen_attention_f16(CH(q), CH(k_pool), CH(v_pool), H(out), cu_seqlens_q, pos_offsets, block_tables,
                                      num_seqs, total_q_tokens, max_q_len, max_kv_len, num_heads, num_kv_heads, head_dim,
                                      sliding_window, block_size, max_num_blocks_per_seq, ws ? ws->ptr : nullptr,
                                      ws ? ws->bytes : 0, ST(stream));
}

int ferrum_hip_paged_batched_decode_attention_f16(const void* q, const void* k_pool, const void* v_pool, void* out,
                                                  const int32_t* block_tables, const uint32_t* valid_kv_lens,
                                                  int num_seqs, int max_kv_len, int num_heads, int num_kv_heads,
                                                  int head_dim, int block_size, int max_num_blocks_per_seq,
                                                  FerrumHipWorkspace* ws, void* stream) {
    FH_REQUIRE(num_seqs == 0 || (q && k_pool && v_pool && out && block_tables && valid_kv_lens),
               "paged_batched_decode_attention: null buffer");
    return paged_batched_decode_attention_f16(CH(q), CH(k_pool), CH(v_pool), H(out), block_tables, valid_kv_lens,
                                              num_seqs, max_kv_len, num_heads, num_kv_heads, head_dim, block_size,
                                              max_num_blocks_per_seq, ws ? ws->ptr : nullptr, ws ? ws->bytes : 0,
                                              ST(stream));
}

// BackendPagedKv::paged_decode_attention (traits.rs:1719-1738).  q_len == 1: q / out [num_seqs, num_heads, head_dim]
// token-major, one query token per sequence.  q_len > 1: ONE sequence in causal prefill, q / out [num_heads, q_len, head_dim]
// head-major; context_lens[0] is the FINAL kv length (token i sees positions [0, context_len − q_len + 1 + i)).
__global__ void single_seq_index_kernel(const uint32_t* __restrict__ context_lens, uint32_t* __restrict__ cu, uint32_t* __restrict__ pos, int q_len) {
    if (threadIdx.x == 0) { cu[0] = 0; cu[1] = (uint32_t)q_len; pos[0] = context_lens[0] - (uint32_t)q_len; }
}
int ferrum_hip_paged_decode_attention_f16(const void* q, const void* k_pool, const void* v_pool, void* out,
                                          const int32_t* block_tables, const uint32_t* context_lens, int num_seqs, int num_heads,
                                          int num_kv_heads, int head_dim, int block_size, int max_num_blocks_per_seq, int q_len,
                                          FerrumHipWorkspace* ws, void* stream) {
    FH_REQUIRE(num_seqs == 0 || (q && k_pool && v_pool && out && block_tables && context_lens), "paged_decode_attention: null buffer");
    FH_REQUIRE(q_len >= 1, "paged_decode_attention: q_len=%d", q_len);
    if (num_seqs <= 0) return 0;
    const int kv_bound = max_num_blocks_per_seq * block_size;            // the kv lengths live on the device: the table width bounds them
    if (q_len == 1)
        return paged_batched_decode_attention_f16(CH(q), CH(k_pool), CH(v_pool), H(out), block_tables, context_lens, num_seqs, kv_bound,
                                                  num_heads, num_kv_heads, head_dim, block_size, max_num_blocks_per_seq,
                                                  ws ? ws->ptr : nullptr, ws ? ws->bytes : 0, ST(stream));
    if (num_seqs != 1) { fh::set_error("paged_decode_attention: q_len=%d > 1 is the single-sequence prefill form (num_seqs=%d)", q_len, num_seqs); return FERRUM_HIP_UNSUPPORTED; }
    const size_t elems = (size_t)q_len * num_heads * head_dim;
    const size_t need = 2 * elems * 2 + 256;
    FH_REQUIRE(ws && ws->ptr && ws->bytes >= need + (1 << 20), "paged_decode_attention: workspace of %zu bytes needed for the head-major transposes", need + (1 << 20));
    __half* q_tm = reinterpret_cast<__half*>(ws->ptr);
    __half* o_tm = q_tm + elems;
    uint32_t* idx = reinterpret_cast<uint32_t*>(o_tm + elems);
    float* attn_ws = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(ws->ptr) + (need + 255) / 256 * 256);
    const size_t attn_ws_bytes = ws->bytes - (need + 255) / 256 * 256;
    hipLaunchKernelGGL(single_seq_index_kernel, dim3(1), dim3(64), 0, ST(stream), context_lens, idx, idx + 2, q_len);
    FH_CHECK_LAUNCH();
    if (int rc = transpose_head_to_token_f16(CH(q), q_tm, q_len, num_heads, head_dim, ST(stream))) return rc;
    if (int rc = paged_varlen_attention_f16(q_tm, CH(k_pool), CH(v_pool), o_tm, idx, idx + 2, block_tables, 1, q_len, q_len, kv_bound,
                                            num_heads, num_kv_heads, head_dim, 0, block_size, max_num_blocks_per_seq, attn_ws,
                                            attn_ws_bytes, ST(stream))) return rc;
    return transpose_token_to_head_f16(o_tm, H(out), q_len, num_heads, head_dim, ST(stream));
}

// BackendGraph (capabilities.rs:35-70): begin / end stream capture, replay, drop.  The key → graph map of the trait
// (`end_graph_capture(key)`, `replay_graph(key)`) is a HashMap in the binding; this side owns the graph objects.
int ferrum_hip_graph_begin_capture(void* stream) {
    FH_CHECK_HIP(hipStreamBeginCapture(ST(stream), hipStreamCaptureModeThreadLocal));
    return 0;
}
int ferrum_hip_graph_end_capture(void* stream, FerrumHipGraph** graph) {
    FH_REQUIRE(graph, "graph_end_capture: null output");
    auto* g = new FerrumHipGraph();
    hipError_t e = hipStreamEndCapture(ST(stream), &g->graph);
    if (e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        if (g->graph) (void)hipGraphDestroy(g->graph);
        delete g;
        fh::set_error("graph_end_capture: %s", hipGetErrorString(e));
        return 1;
    }
    *graph = g;
    return 0;
}
int ferrum_hip_graph_replay(FerrumHipGraph* graph, void* stream) {
    FH_REQUIRE(graph && graph->exec, "graph_replay: null graph");
    FH_CHECK_HIP(hipGraphLaunch(graph->exec, ST(stream)));
    return 0;
}
int ferrum_hip_graph_destroy(FerrumHipGraph* graph) {
    if (!graph) return 0;
    if (graph->exec) (void)hipGraphExecDestroy(graph->exec);
    if (graph->graph) (void)hipGraphDestroy(graph->graph);
    delete graph;
    return 0;
}

int ferrum_hip_paged_decode_attention_fused_qkv_f16(const void* qkv, const void* q_norm_w, const void* k_norm_w,
                                                    const float* cos_tab, const float* sin_tab, float eps, int qk_mode,
                                                    void* k_pool, void* v_pool, void* out, const int32_t* block_tables,
                                                    const uint32_t* valid_kv_lens, int num_seqs, int max_kv_len,
                                                    int num_heads, int num_kv_heads, int head_dim, int sliding_window,
                                                    int block_size, int max_num_blocks_per_seq, FerrumHipWorkspace* ws,
                                                    void* stream) {
    FH_REQUIRE(num_seqs == 0 || (qkv && k_pool && v_pool && out && block_tables && valid_kv_lens),
               "paged_decode_attention_fused_qkv: null buffer");
    FH_REQUIRE(qk_mode == 0 || (cos_tab && sin_tab), "paged_decode_attention_fused_qkv: rope tables missing");
    FH_REQUIRE(qk_mode != 1 || (q_norm_w && k_norm_w), "paged_decode_attention_fused_qkv: norm weights missing for qk_mode 1");
    return paged_decode_attention_fused_qkv_f16(CH(qkv), CH(q_norm_w), CH(k_norm_w), cos_tab, sin_tab, eps, qk_mode,
                                                H(k_pool), H(v_pool), H(out), block_tables, valid_kv_lens, num_seqs,
                                                max_kv_len, num_heads, num_kv_heads, head_dim, sliding_window, block_size,
                                                max_num_blocks_per_seq, ws ? ws->ptr : nullptr, ws ? ws->bytes : 0,
                                                ST(stream));
}

int ferrum_hip_paged_kv_read_f16(const void* cache_k, const void* cache_v, const int32_t* block_table, int kv_len,
                                 int kv_heads, int head_dim, int block_size, void* k_out, void* v_out, void* stream) {
    return paged_kv_read_f16(CH(cache_k), CH(cache_v), block_table, kv_len, kv_heads, head_dim, block_size, H(k_out),
                             H(v_out), ST(stream));
}

// ── MoE routing ─────────────────────────────────────────────────────────────
int ferrum_hip_moe_route_topk_softmax_f16(const void* logits, int32_t* ids, float* w, int tokens, int num_experts,
                                          int top_k, int norm, void* stream) {
    return moe_route_topk_softmax_f16(CH(logits), ids, w, tokens, num_experts, top_k, norm, ST(stream));
}
int ferrum_hip_moe_route_topk_softmax_f32(const float* logits, int32_t* ids, float* w, int tokens, int num_experts,
                                          int top_k, int norm, void* stream) {
    return moe_route_topk_softmax_f32(logits, ids, w, tokens, num_experts, top_k, norm, ST(stream));
}
int ferrum_hip_moe_align_block_size(const int32_t* expert_ids, int32_t* sorted_token_ids, int32_t* block_ids,
                                    int32_t* total_post_pad, int batch_x_topk, int num_experts, int block_size,
                                    int sorted_max, void* stream) {
    FH_REQUIRE((expert_ids || batch_x_topk == 0) && sorted_token_ids && block_ids && total_post_pad, "moe_align_block_size: null buffer");
    return moe_align_block_size(expert_ids, sorted_token_ids, block_ids, total_post_pad, batch_x_topk, num_experts,
                                block_size, sorted_max, ST(stream));
}
int ferrum_hip_moe_combine_f16(const void* down, const float* weights, void* out, int tokens, int top_k, int hidden,
                               int accumulate, void* stream) {
    return moe_combine_f16(CH(down), weights, H(out), tokens, top_k, hidden, accumulate, ST(stream));
}

int ferrum_hip_moe_align_block_size_packed_rows(const int32_t* expert_ids, int32_t* sorted_token_ids, int32_t* block_ids,
                                                int32_t* total_post_pad, int batch_x_topk, int num_experts, int block_size,
                                                int sorted_max, void* stream) {
    FH_REQUIRE(batch_x_topk == 0 || (expert_ids && sorted_token_ids && block_ids && total_post_pad), "moe_align_block_size: null buffer");
    return moe_align_block_size_packed_rows(expert_ids, sorted_token_ids, block_ids, total_post_pad, batch_x_topk, num_experts,
                                            block_size, sorted_max, ST(stream));
}
int ferrum_hip_moe_build_pairs_by_token(const int32_t* expert_ids, int32_t* pairs_by_token, int32_t* packed_token_idx,
                                        int32_t* expert_offsets, int batch_x_topk, int num_experts, int top_k, void* stream) {
    FH_REQUIRE(expert_offsets && (batch_x_topk == 0 || (expert_ids && pairs_by_token && packed_token_idx)), "moe_build_pairs_by_token: null buffer");
    return moe_build_pairs_by_token(expert_ids, pairs_by_token, packed_token_idx, expert_offsets, batch_x_topk, num_experts, top_k,
                                    ST(stream));
}
int ferrum_hip_moe_combine_pairs_f16(const void* packed_down, const int32_t* pairs_by_token, const float* pair_weights, void* out,
                                     int batch, int hidden, int top_k, int total_pairs, void* stream) {
    FH_REQUIRE(batch == 0 || (packed_down && pairs_by_token && pair_weights && out), "moe_combine: null buffer");
    return moe_combine_pairs_f16(CH(packed_down), pairs_by_token, pair_weights, H(out), batch, hidden, top_k, total_pairs, ST(stream));
}
// weighted_sum_batched / weighted_sum_batched_offset (capabilities.rs:560-600): out[b, h] = Σ_k weights[b, k]·slots[b, k, h];
// offsets in ELEMENTS into `weights` (start of [batch, top_k]) and `out` (start of [batch, hidden]).
int ferrum_hip_weighted_sum_batched_f16(const void* slots, const float* weights, size_t weights_offset, void* out, size_t out_offset,
                                        int batch, int top_k, int hidden, void* stream) {
    FH_REQUIRE(batch == 0 || (slots && weights && out), "weighted_sum_batched: null buffer");
    FH_REQUIRE(out_offset % 8 == 0, "weighted_sum_batched: out_offset=%zu must be a multiple of 8 elements", out_offset);
    return moe_combine_f16(CH(slots), weights + weights_offset, H(out) + out_offset, batch, top_k, hidden, 0, ST(stream));
}

// MarlinExpertStack::gemm_phase_batched (marlin_expert_stack.rs:63-74): dispatches[i] = (expert, in_row_offset,
// out_row_offset, m); rows [in_off, in_off + m) of `input` × tile[expert] → rows [out_off, out_off + m) of `output`.
// One grouped launch per distinct (out_off − in_off): the dispatch list becomes the block-major routing arrays of the
// grouped GEMM (16-row blocks, or 64-row LDS tiles when the experts see ≥ 32 rows on average).
int ferrum_hip_moe_gemm_phase_batched_f16(FerrumHipGptq* stack, const void* input, const int32_t* dispatches, int num_dispatches,
                                          void* output, int k, int fused_silu_mul, void* stream) {
    FH_REQUIRE(stack && (num_dispatches == 0 || (input && dispatches && output)), "gemm_phase_batched: null argument");
    FH_REQUIRE(k == stack->dev.k, "gemm_phase_batched: k=%d but the stack was packed with K=%d", k, stack->dev.k);
    FH_REQUIRE(!fused_silu_mul || stack->dev.fused_gate_up, "gemm_phase_batched: fused epilogue needs a stack loaded with fuse_gate_up");
    FH_REQUIRE(fused_silu_mul || !stack->dev.fused_gate_up, "gemm_phase_batched: stack was loaded with fuse_gate_up; plain output is column-permuted");
    if (num_dispatches <= 0) return 0;
    long rows = 0, max_row = 0;
    for (int i = 0; i < num_dispatches; i++) {
        const int32_t* d = dispatches + 4 * i;
        FH_REQUIRE(d[0] >= 0 && d[0] < stack->dev.num_experts && d[1] >= 0 && d[2] >= 0 && d[3] >= 0,
                   "gemm_phase_batched: dispatch %d = (%d, %d, %d, %d) out of range", i, d[0], d[1], d[2], d[3]);
        rows += d[3];
        max_row = std::max<long>(max_row, (long)d[1] + d[3]);
    }
    if (rows == 0) return 0;
    const int br = rows >= 32L * num_dispatches ? 64 : 16;
    // group by output displacement (normally one group: in_off == out_off, the packed-row convention)
    std::vector<long> deltas;
    for (int i = 0; i < num_dispatches; i++) {
        const long dl = (long)dispatches[4 * i + 2] - dispatches[4 * i + 1];
        if (dispatches[4 * i + 3] > 0 && std::find(deltas.begin(), deltas.end(), dl) == deltas.end()) deltas.push_back(dl);
    }
    const int ldo = fused_silu_mul ? stack->dev.n / 2 : stack->dev.n;
    const __half* x_in = nullptr;
    if (int rc = stack_input(stack, input, (int)max_row, ST(stream), &x_in)) return rc;
    for (long dl : deltas) {
        std::vector<int32_t> sorted, blocks;
        for (int i = 0; i < num_dispatches; i++) {
            const int32_t* d = dispatches + 4 * i;
            if (d[3] == 0 || (long)d[2] - d[1] != dl) continue;
            for (int r0 = 0; r0 < d[3]; r0 += br) {
                blocks.push_back(d[0]);
                for (int r = 0; r < br; r++) sorted.push_back(r0 + r < d[3] ? d[1] + r0 + r : (int32_t)max_row);   // sentinel = prob_m
            }
        }
        const size_t nb = blocks.size(), need = sorted.size() + nb + 4;
        FerrumHipGptq::DispatchSlot& sl = stack->slots[stack->next_slot];
        stack->next_slot = (stack->next_slot + 1) % 4;
        if (sl.done) FH_CHECK_HIP(hipEventSynchronize(sl.done));      // the launch that last used this slot has consumed it
        else FH_CHECK_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        if (sl.cap < need) {
            if (sl.host) (void)hipHostFree(sl.host);
            if (sl.dev) (void)hipFree(sl.dev);
            sl.host = nullptr; sl.dev = nullptr; sl.cap = 0;
            FH_CHECK_HIP(hipHostMalloc((void**)&sl.host, need * 2 * 4, hipHostMallocDefault));
            FH_CHECK_HIP(hipMalloc((void**)&sl.dev, need * 2 * 4));
            sl.cap = need * 2;
        }
        memcpy(sl.host, sorted.data(), sorted.size() * 4);
        memcpy(sl.host + sorted.size(), blocks.data(), nb * 4);
        sl.host[sorted.size() + nb] = (int32_t)sorted.size();           // total_tokens_post_pad
        FH_CHECK_HIP(hipMemcpyAsync(sl.dev, sl.host, need * 4, hipMemcpyHostToDevice, ST(stream)));
        const int32_t* d_sorted = sl.dev;
        const int32_t* d_blocks = sl.dev + sorted.size();
        const int32_t* d_total = sl.dev + sorted.size() + nb;
        __half* out_shift = H(output) + dl * ldo;                        // row (in_off + r) of the shifted view = out_off + r
        int rc = br == 64 ? w4_gemm_moe_tile(stack->dev, x_in, out_shift, d_sorted, d_blocks, d_total, (int)max_row, (int)nb, 64, 1,
                                             fused_silu_mul, ST(stream))
                          : w4_gemm_moe(stack->dev, x_in, out_shift, d_sorted, d_blocks, d_total, (int)max_row, (int)nb, 1,
                                        fused_silu_mul, ST(stream));
        if (rc) return rc;
        FH_CHECK_HIP(hipEventRecord(sl.done, ST(stream)));
    }
    return 0;
}

int ferrum_hip_fused_add_rms_norm_route_f16(void* residual, const void* x, const void* w, float eps, void* norm_out,
                                            const void* router_w, int num_experts, int top_k, int norm_topk_prob,
                                            int32_t* expert_ids, float* expert_weights, float* logits_out, int tokens,
                                            int hidden, void* stream) {
    FH_REQUIRE(tokens == 0 || (residual && x && w && norm_out), "fused_add_rms_norm_route: null buffer");
    FH_REQUIRE(num_experts == 0 || (router_w && expert_ids && expert_weights), "fused_add_rms_norm_route: null router buffers");
    return fused_add_rms_norm_route_f16(H(residual), CH(x), CH(w), eps, H(norm_out), CH(router_w), num_experts, top_k,
                                        norm_topk_prob, expert_ids, expert_weights, logits_out, tokens, hidden, ST(stream));
}
int ferrum_hip_moe_combine_add_rms_norm_f16(const void* down, const float* weights, void* residual,
                                            const void* next_norm_w, float eps, void* norm_out, int tokens, int top_k,
                                            int hidden, void* stream) {
    FH_REQUIRE(tokens == 0 || (down && weights && residual && (!next_norm_w || norm_out)), "moe_combine_add_rms_norm: null buffer");
    return moe_combine_add_rms_norm_f16(CH(down), weights, CH(residual), H(residual), CH(next_norm_w), eps, H(norm_out),
                                        tokens, top_k, hidden, ST(stream));
}

int ferrum_hip_fused_add_rms_norm_route_parts_f16(const void* residual_in, void* residual_out, const void* x_f16,
                                                  const float* x_slabs, int num_slabs, long slab_stride, int ld_slab,
                                                  const void* w, float eps, void* norm_out, const void* router_w_tiled,
                                                  int num_experts, int top_k, int num_parts, void* cand, float* stats,
                                                  float* logits_out, int tokens, int hidden, void* stream) {
    FH_REQUIRE(tokens == 0 || (residual_in && residual_out && w && norm_out && (x_f16 || x_slabs)), "route_parts: null buffer");
    FH_REQUIRE(num_experts == 0 || (router_w_tiled && cand && stats), "route_parts: null router buffers");
    return fused_add_rms_norm_route_parts_f16(CH(residual_in), H(residual_out), CH(x_f16), x_slabs, num_slabs, slab_stride,
                                              ld_slab, CH(w), eps, H(norm_out), CH(router_w_tiled), num_experts, top_k,
                                              num_parts, reinterpret_cast<RouteCand*>(cand), stats, logits_out, tokens, hidden,
                                              ST(stream));
}
int ferrum_hip_fused_add_rms_norm_route_split_f16(const void* residual_in, void* residual_out, const void* x_f16,
                                                  const float* x_slabs, int num_slabs, long slab_stride, int ld_slab,
                                                  const void* w, float eps, void* norm_out, const void* router_w_tiled,
                                                  int num_experts, int top_k, int norm_topk_prob, int num_parts,
                                                  void* cand, float* stats, uint32_t* arrive, int32_t* expert_ids,
                                                  float* expert_weights, float* logits_out, int tokens, int hidden,
                                                  void* stream) {
    FH_REQUIRE(tokens == 0 || (residual_in && residual_out && w && norm_out && (x_f16 || x_slabs)), "route_split: null buffer");
    FH_REQUIRE(num_experts == 0 || (router_w_tiled && cand && stats && arrive && expert_ids && expert_weights),
               "route_split: null router buffers");
    return fused_add_rms_norm_route_split_f16(CH(residual_in), H(residual_out), CH(x_f16), x_slabs, num_slabs, slab_stride,
                                              ld_slab, CH(w), eps, H(norm_out), CH(router_w_tiled), num_experts, top_k,
                                              num_parts, reinterpret_cast<RouteCand*>(cand), stats, arrive, norm_topk_prob,
                                              expert_ids, expert_weights, logits_out, tokens, hidden, ST(stream));
}
int ferrum_hip_moe_gemm_phase_merge_route_f16(const FerrumHipGptq* stack, const void* input, const void* cand,
                                              const float* stats, void* output, int tokens, int num_parts, int top_k,
                                              int norm_topk_prob, int num_experts, int max_blocks, int fused_silu_mul,
                                              int32_t* expert_ids_out, float* expert_weights_out,
                                              int32_t* sorted_token_ids_out, int32_t* block_ids_out,
                                              int32_t* total_post_pad_out, void* stream) {
    FH_REQUIRE(stack && input && cand && stats && output && expert_ids_out && expert_weights_out, "merge_route: null argument");
    FH_REQUIRE(!fused_silu_mul || stack->dev.fused_gate_up, "merge_route: fused epilogue needs a stack loaded with fuse_gate_up");
    FH_REQUIRE(fused_silu_mul || !stack->dev.fused_gate_up, "merge_route: stack was loaded with fuse_gate_up; plain output is column-permuted");
    const __half* x_in = nullptr;
    if (int rc = stack_input(stack, input, tokens, ST(stream), &x_in)) return rc;
    return w4_gemm_moe_merge_route(stack->dev, x_in, H(output), reinterpret_cast<const RouteCand*>(cand), stats, tokens,
                                   num_parts, top_k, norm_topk_prob, num_experts, max_blocks, fused_silu_mul, expert_ids_out,
                                   expert_weights_out, sorted_token_ids_out, block_ids_out, total_post_pad_out, ST(stream));
}

// ── sampling ────────────────────────────────────────────────────────────────
int ferrum_hip_argmax_rows_f16(const void* logits, uint32_t* out, const uint8_t* mask, int mask_len, int m, int n, void* stream) {
    return argmax_rows_f16(CH(logits), out, mask, mask_len, m, n, ST(stream));
}
int ferrum_hip_argmax_rows_f32(const float* logits, uint32_t* out, const uint8_t* mask, int mask_len, int m, int n, void* stream) {
    return argmax_rows_f32(logits, out, mask, mask_len, m, n, ST(stream));
}
int ferrum_hip_argmax_rows_f16_ws(const void* logits, uint32_t* out, const uint8_t* mask, int mask_len, int m, int n,
                                  FerrumHipWorkspace* ws, void* stream) {
    return argmax_rows_f16_ws(CH(logits), out, mask, mask_len, m, n, ws ? ws->ptr : nullptr, ws ? ws->bytes : 0, ST(stream));
}
int ferrum_hip_argmax_rows_f32_ws(const float* logits, uint32_t* out, const uint8_t* mask, int mask_len, int m, int n,
                                  FerrumHipWorkspace* ws, void* stream) {
    return argmax_rows_f32_ws(logits, out, mask, mask_len, m, n, ws ? ws->ptr : nullptr, ws ? ws->bytes : 0, ST(stream));
}
int ferrum_hip_apply_repetition_penalties_sparse_f16(void* logits, const uint32_t* row_offsets, const uint32_t* token_ids,
                                                     const float* penalties, int m, int n, void* stream) {
    return apply_repetition_penalties_sparse_f16(H(logits), row_offsets, token_ids, penalties, m, n, ST(stream));
}
int ferrum_hip_apply_repetition_penalties_sparse_f32(float* logits, const uint32_t* row_offsets, const uint32_t* token_ids,
                                                     const float* penalties, int m, int n, void* stream) {
    return apply_repetition_penalties_sparse_f32(logits, row_offsets, token_ids, penalties, m, n, ST(stream));
}

// ── BlockAllocator ──────────────────────────────────────────────────────────
struct FerrumHipBlockAllocator { fh::BlockAllocator impl; explicit FerrumHipBlockAllocator(uint32_t n) : impl(n) {} };

int ferrum_hip_block_allocator_create(FerrumHipBlockAllocator** a, uint32_t num_blocks) {
    FH_REQUIRE(a, "block_allocator_create: null output");
    *a = new FerrumHipBlockAllocator(num_blocks);
    return 0;
}
int ferrum_hip_block_allocator_destroy(FerrumHipBlockAllocator* a) { delete a; return 0; }
int ferrum_hip_block_allocator_allocate(FerrumHipBlockAllocator* a, uint32_t* block) {
    FH_REQUIRE(a && block, "block_allocator_allocate: null argument");
    if (!a->impl.allocate(block)) {
        fh::set_error("paged KV pool exhausted (capacity=%u blocks, all in use)", a->impl.capacity());
        return FERRUM_HIP_INVALID;
    }
    return 0;
}
int ferrum_hip_block_allocator_allocate_n(FerrumHipBlockAllocator* a, uint32_t n, uint32_t* blocks) {
    FH_REQUIRE(a && (n == 0 || blocks), "block_allocator_allocate_n: null argument");
    if (!a->impl.allocate_n(n, blocks)) {
        fh::set_error("paged KV pool exhausted: need %u blocks but only %u free", n, a->impl.free_count());
        return FERRUM_HIP_INVALID;
    }
    return 0;
}
int ferrum_hip_block_allocator_free(FerrumHipBlockAllocator* a, const uint32_t* blocks, uint32_t n) {
    FH_REQUIRE(a && (n == 0 || blocks), "block_allocator_free: null argument");
    // the reference indexes its ref-count vector (a bad id panics, paged_pool.rs:333-345): a bad id from the FFI side must
    // fail here, before anything is freed, instead of writing outside the vectors
    for (uint32_t i = 0; i < n; i++)
        FH_REQUIRE(blocks[i] < a->impl.capacity(), "block_allocator_free: block %u >= capacity %u", blocks[i], a->impl.capacity());
    a->impl.free(blocks, n);
    return 0;
}
int ferrum_hip_block_allocator_acquire(FerrumHipBlockAllocator* a, uint32_t block) {
    FH_REQUIRE(a, "block_allocator_acquire: null");
    FH_REQUIRE(block < a->impl.capacity(), "block_allocator_acquire: block %u >= capacity %u", block, a->impl.capacity());
    // paged_pool.rs acquire: checked_add on the ref count (panics on overflow) — here: an error, the count is left alone
    FH_REQUIRE(a->impl.ref_count(block) < 0xFFFFu, "block_allocator_acquire: ref count of block %u would overflow", block);
    a->impl.acquire(block);
    return 0;
}
int ferrum_hip_block_allocator_register_hash(FerrumHipBlockAllocator* a, uint32_t block, uint64_t hash) {
    FH_REQUIRE(a, "block_allocator_register_hash: null");
    FH_REQUIRE(block < a->impl.capacity(), "block_allocator_register_hash: block %u >= capacity %u", block, a->impl.capacity());
    a->impl.register_block_hash(block, hash);
    return 0;
}
int ferrum_hip_block_allocator_try_acquire_by_hash(FerrumHipBlockAllocator* a, uint64_t hash, int64_t* block) {
    FH_REQUIRE(a && block, "block_allocator_try_acquire_by_hash: null");
    *block = a->impl.try_acquire_by_hash(hash);
    return 0;
}
uint32_t ferrum_hip_block_allocator_free_count(const FerrumHipBlockAllocator* a) { return a->impl.free_count(); }
uint32_t ferrum_hip_block_allocator_ref_count(const FerrumHipBlockAllocator* a, uint32_t b) { return b < a->impl.capacity() ? a->impl.ref_count(b) : 0; }
uint32_t ferrum_hip_block_allocator_peak_in_use(const FerrumHipBlockAllocator* a) { return a->impl.peak_in_use(); }
uint32_t ferrum_hip_block_allocator_hash_table_size(const FerrumHipBlockAllocator* a) { return a->impl.hash_table_size(); }

}  // extern "C"
