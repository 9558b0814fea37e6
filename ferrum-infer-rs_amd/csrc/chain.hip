// The attention half of a MoE decode layer as ONE launch (gfx950).
//
// A decode step of ≤ 32 rows is a chain of small dependent ops (qwen3_moe_forward_unified_layer.rs:46-455 issues them one
// launch at a time): [tail of the previous layer: combine + add + norm] → q|k|v GEMM → QK-norm + RoPE + paged-KV write +
// attention → o_proj GEMM → add + norm + router + top-k.  As five launches every link pays its own ramp, its first memory
// round trips and its drain (≈ 4–5 µs each even when trivial) on top of the kernel boundary, while HBM idles.  Here the five
// links are ROLES of one grid, laid out in dependency order (blocks of role A first, …): every workgroup requests what does
// not depend on its predecessor — its INT4 weights (all of them: a wave's K slice is 2–4 quant groups), its block table and
// first K/V tiles, its norm and router weights — and only then waits for the predecessor's counter.  What is left on the
// critical path of a link is: activations in, arithmetic, result out.  One 512-thread workgroup per CU is resident (the
// attention role holds two K/V fragment sets, 232 registers): at 32 rows the tail (32 workgroups), the q|k|v projection (80)
// and attention (128) are all there from the start, the o projection's workgroups move in as the tail's and q|k|v's leave
// (long before attention is done) and the router's as attention's leave — each early enough to have its weights on the way.
//
// Hand-offs (MI355X_MICROARCH.md § visibility, valid forms; cdna_hip_programming.md Guideline 16): every byte that crosses
// workgroups inside the launch is stored write-through (sc1, 8- or 16-byte stores), every storing wave drains its stores
// (s_waitcnt vmcnt(0)), the workgroup meets at a barrier and ONE lane adds to the consumer's counter; a consumer polls its
// counter with relaxed agent-scope loads from one lane (s_sleep between polls), releases its other waves through a workgroup
// barrier, and reads the handed-off bytes with sc1 loads only.  No role waits before all its predecessors' workgroups have been
// dispatched (dependency order = dispatch order, and nothing that waits comes before something it waits for); every wait
// is bounded (≈ 20 ms) and a give-up is counted in a host-visible word, so a missing producer costs a wrong result that the
// runner reports, never a hang.  Counters live on 256-byte lines of their own, are zero on entry and the launch zeroes the
// OTHER half of a double buffer for the next launch (no memset node per layer).
//
// Arithmetic: each role is the arithmetic of the stand-alone kernel it replaces, operation for operation —
// moe_combine_add_rmsnorm_kernel, w4_gemm_wgsplit_kernel<MT, 2> (K split over the 8 waves, LDS reduction in wave order),
// paged_attn_kernel<128, true, 8> (fused QK-norm / RoPE / KV write, 8 KV-splitting waves merged in wave order),
// add_rmsnorm_route_part_kernel (Q expert parts per token, in-launch ticket merge) — so the results are the same bits.
#include "common.h"
#include "kernels.h"
#include "knobs.h"
#include "kv_layout.h"
#include "rope_rows.h"
#include "w4_device.h"
#include "route_merge.h"

namespace fh {

namespace {

constexpr int CH_W = 8;                 // waves per workgroup
constexpr int CH_STRIDE = 64;           // words between counters (one 256-byte line each)
// Counter groups (slots of CH_STRIDE words).  A group is `shards` × `reps` counters: producer i adds to EVERY replica of shard
// i mod shards (one wave instruction, `reps` lanes), a waiting workgroup polls ONE replica of every shard (one wave instruction,
// `shards` lanes).  Shards cut the serialised arrivals per line (≈ 12 ns each), replicas the pollers per line — with one
// counter per edge the hand-offs took 2–4 µs each (128 pollers + 64 arrivals on one line: 4.2 µs).
constexpr int CH_NORM_SLOT = 0, CH_NORM_SH = 1, CH_NORM_R = 8;          // tail → q|k|v: T arrivals, ≤ 80 pollers
constexpr int CH_O_SLOT = 8, CH_O_SH = 4, CH_O_R = 4;                   // o_proj → route: 64 arrivals, 128 pollers
constexpr int CH_ATTN_SLOT = 24, CH_ATTN_SH = 8, CH_ATTN_R = 4;         // attention → o_proj: T·nkv arrivals, 64 pollers
constexpr int CH_QKV_SLOT = 56, CH_QKV_R = 4;                           // q|k|v → attention, one group per kv head: (G + 2)·row blocks arrivals, T pollers
constexpr int CH_MAX_KVH = 16;
constexpr int CH_SLOTS = CH_QKV_SLOT + CH_QKV_R * CH_MAX_KVH;
constexpr int CH_GRAN = 10;              // granules per (token, router part): 8 candidates + max + Σexp
constexpr int CH_O_KS = 2;                // o_proj: K parts per block, each a workgroup of its own (fp32 partial rows, role B adds them)
constexpr int CH_MAX_SPLITS = 16;          // KV ranges per (sequence, kv head) in the attention role
constexpr int CH_MAX_T = 128;
constexpr int CH_GRAN_WORDS = CH_MAX_T * 4 * CH_GRAN * 2;      // T ≤ 64 tokens × Q ≤ 4 parts, 8-byte granules
constexpr int CH_SMEM = 8 * 16 * (128 + 4) * 4 + 2 * 8 * 16 * 4 + 16 * 128 * 2 + 2 * 128 * 2;     // the attention role's arena: 73,216 B

constexpr int CH_QKV_NST = 1, CH_O_NST = 1;     // 64-column supertiles per workgroup: one head (128 columns) for q|k|v, 64 columns for o_proj; 16 rows each

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct ChainGemm {
    const uint32_t* qw; const __half* sc; const __half* zp;
    int G, N, K;
};

struct ChainArgs {
    int T, H, nq, nkv;
    int qkv_half;                 // q|k|v in 32-column blocks (one 16-row block only: few rows, every CU gets a block)
    int o_half;                   // o_proj likewise
    // role A: residual' = residual + Σ_k w_k·down_k; norm1 = rms_norm(residual')·ln_in   (absent for the first layer)
    int has_a, top_k;              // has_a: 0 no tail, 1 MoE combine (down / comb_w / top_k), 2 dense (the down projection's split-K slabs)
    const __half* down; const float* comb_w; const __half* res_in; const __half* ln_in;
    const float* a_slabs; int a_S; long a_slab_stride; int a_ld;
    const __half* a_x;              // has_a == 3: the MLP output as fp16 rows [T, H]
    float eps;
    __half* res_a;                // residual' — also what role B adds the o projection to
    __half* norm1;                // [T, H]
    ChainGemm qkv;
    __half* qkv_out;              // [T, (nq + 2 nkv)·128]
    // attention
    __half* k_pool; __half* v_pool;
    const int32_t* block_tables; const uint32_t* kv_lens;
    const __half* q_norm_w; const __half* k_norm_w; const float* cos_t; const float* sin_t;
    int qk_mode, max_blocks, sliding_window;
    float scale;
    __half* attn_out;             // [T, nq·128]
    ChainGemm o;
    __half* o_out;                // [T, H]
    // role B
    __half* res_b_out; const __half* post_ln; __half* norm2; const __half* router_w;
    int E, r_top_k, Q, norm_topk, defer_merge;
    RouteCand* cand; float* stats; unsigned* route_arrive; int32_t* ids; float* weights;
    unsigned* cnt; unsigned* cnt_next; unsigned* timeout;
#ifdef FERRUM_HIP_EXPERIMENTS
    unsigned long long* tl;       // development: per-workgroup wall-clock stamps (tools/exp_timeline_chain.py)
#endif
    float* o_part;                // [CH_O_KS][T][H] fp32: o_proj's K parts
    int qkv_wide;                 // q|k|v in 128-column blocks (one head per workgroup)
    // (behind everything else: the one-range kernel's argument layout is the one it was tuned with)
    int attn_splits;              // KV ranges per (sequence, kv head): > 1 → partial states meet by ticket, the last arriver merges them
    float* attn_partial;          // [T·nkv][splits][16 rows][HD + 4] fp32 (m, l in the pad)
    unsigned* attn_tickets;       // [T·nkv] self-resetting
};
#ifdef FERRUM_HIP_EXPERIMENTS
unsigned long long* g_chain_timeline = nullptr;
#define CH_TL(i) do { if (p.tl && threadIdx.x == 0) p.tl[(long)blockIdx.x * 4 + (i)] = wall_clock64(); } while (0)
#else
#define CH_TL(i) do {} while (0)
#endif

// `shards` lanes of wave 0 poll (one replica of every shard), the workgroup barrier releases the other waves (the polling
// wave's own loads follow its match).  `total` producers, producer i ↔ shard i mod shards.
__device__ __forceinline__ void chain_wait(unsigned* base, int shards, int reps, unsigned total, unsigned* timeout) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const unsigned need = lane < shards ? (total + (unsigned)(shards - 1 - lane)) / (unsigned)shards : 0u;
        unsigned* slot = base + ((lane < shards ? lane : 0) * reps + (int)(blockIdx.x % (unsigned)reps)) * CH_STRIDE;
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
            const unsigned c = lane < shards ? __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (__ballot(c < need) == 0ull) break;
            __builtin_amdgcn_s_sleep(4);
            if ((++spins & 255u) == 0u && wall_clock64() - t0 > 2000000ull) {
                if (lane == 0) __hip_atomic_fetch_add(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // no instruction: keeps later loads behind the poll
    __syncthreads();
}

// Every wave drains its (write-through) stores, the workgroup meets, `reps` lanes signal (every replica of the shard).
__device__ __forceinline__ void chain_signal(unsigned* base, int shard, int reps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if ((int)threadIdx.x < reps) __hip_atomic_fetch_add(base + (shard * reps + (int)threadIdx.x) * CH_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t chain_rsrc(const void* base, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ half8 load16_sc1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));      // aux 16 = sc1
}
__device__ __forceinline__ void store16_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, half8 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 16);
}

// ── role A: MoE combine + residual add + the next input norm (fused.hip kernel A), one workgroup per token ─────────────
__device__ __forceinline__ void chain_role_a(const ChainArgs& p, int row, unsigned char* smem) {
    float* red = reinterpret_cast<float*>(smem);
    CH_TL(0); CH_TL(1);
    const int H = p.H, nvec = H >> 3, top_k = p.top_k;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int CH = 2;                        // H ≤ 8192 with 512 threads
    half8 v[CH];
    float ss = 0.f;
    const __amdgpu_buffer_rsrc_t r_res = chain_rsrc(p.res_a, (long)p.T * H * 2), r_n1 = chain_rsrc(p.norm1, (long)p.T * H * 2);
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int i = threadIdx.x + c * 512;
        if (i < nvec && p.has_a == 3) {
            // dense model, the MLP's output as fp16 rows (≤ 16 or > 32 rows: no slabs): residual add, as fused_add_rms_norm
            const half8 xv = *reinterpret_cast<const half8*>(p.a_x + (long)row * H + i * 8);
            half8 rv = *reinterpret_cast<const half8*>(p.res_in + (long)row * H + i * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) rv[j] = (_Float16)((float)rv[j] + (float)xv[j]);
            store16_sc1(r_res, (int)(((long)row * H + i * 8) * 2), rv);
            v[c] = rv;
#pragma unroll
            for (int j = 0; j < 8; j++) ss += (float)rv[j] * (float)rv[j];
        } else if (i < nvec && p.has_a == 2) {
            // dense model: the MLP's down projection arrives as S fp32 split-K slabs — summed in slab order and rounded like the
            // fp16 op output (add_rmsnorm_route_kernel<true>, fused.hip), then the residual add
            float o[8];
            reduce_slabs8(p.a_slabs + (long)row * p.a_ld + i * 8, p.a_slab_stride, p.a_S, o);
            half8 rv = *reinterpret_cast<const half8*>(p.res_in + (long)row * H + i * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) rv[j] = (_Float16)((float)rv[j] + (float)(_Float16)o[j]);
            store16_sc1(r_res, (int)(((long)row * H + i * 8) * 2), rv);
            v[c] = rv;
#pragma unroll
            for (int j = 0; j < 8; j++) ss += (float)rv[j] * (float)rv[j];
        } else if (i < nvec) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float wk8[8];
            half8 d8[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {        // the first 8 expert rows and the residual row are requested together
                const int kc = k < top_k ? k : top_k - 1;
                wk8[k] = p.comb_w[(long)row * top_k + kc];
                d8[k] = *reinterpret_cast<const half8*>(p.down + ((long)row * top_k + kc) * H + i * 8);
            }
            half8 rv = *reinterpret_cast<const half8*>(p.res_in + (long)row * H + i * 8);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (k < top_k) {
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[j] += wk8[k] * (float)d8[k][j];
                }
            }
            for (int k = 8; k < top_k; k++) {
                const float wk = p.comb_w[(long)row * top_k + k];
                const half8 d = *reinterpret_cast<const half8*>(p.down + ((long)row * top_k + k) * H + i * 8);
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] += wk * (float)d[j];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) rv[j] = (_Float16)((float)rv[j] + acc[j]);
            store16_sc1(r_res, (int)(((long)row * H + i * 8) * 2), rv);
            v[c] = rv;
#pragma unroll
            for (int j = 0; j < 8; j++) ss += (float)rv[j] * (float)rv[j];
        }
    }
    // the stand-alone kernel sums four wave partials of 256 threads; rows of ≤ 2048 elements live in the first four waves
    // here too, wider rows use all eight (same order: wave 0 … 7)
    ss = wave_reduce_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    float total = red[0] + red[1] + red[2] + red[3];
    if (nvec > 256) total += red[4] + red[5] + red[6] + red[7];
    const float inv = 1.0f / sqrtf(total / (float)H + p.eps);
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int i = threadIdx.x + c * 512;
        if (i < nvec) {
            const half8 wv = *reinterpret_cast<const half8*>(p.ln_in + i * 8);
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (_Float16)((float)v[c][j] * inv * (float)wv[j]);
            store16_sc1(r_n1, (int)(((long)row * H + i * 8) * 2), o);
        }
    }
    CH_TL(2);
    chain_signal(p.cnt + CH_NORM_SLOT * CH_STRIDE, 0, CH_NORM_R);
    CH_TL(3);
}

// ── GEMM roles (q|k|v and o_proj): one 16-row block × NST 64-column supertiles per workgroup, K split over its 8 waves ───
// All GPW quant groups of a wave's K slice are requested before the wait; the activations follow the wait (sc1), two groups
// in flight; the 8 partial sums meet in LDS (summed in wave order) and leave as 16-byte write-through stores.  A workgroup
// reads rows × K activation bytes after its wait, at the ≈ 70 GB/s a workgroup gets for handed-off bytes — so the rows are
// split over workgroups (16 each) rather than the columns made narrower (32 rows × 4096: 8.0 µs behind the wait).
struct ChainEdge { unsigned* base; int shards, reps; unsigned total; };      // base == nullptr: nothing to wait for

// NTL = 16-column tiles of the 64-column supertile a workgroup takes (NST = 1): 4, or 2 — a 32-column block `cb` = supertile cb / 2,
// half cb % 2: half the weight bytes per workgroup, twice the workgroups (the stage is as long as its slowest workgroup's fetch).
// F32OUT: the workgroup covers only the quant groups from g_base on (8·GPW of them) and leaves its fp32 partial sums in `out32`
// ([K part][T][N]) — o_proj split over two workgroups per block: the rows a workgroup reads behind its wait (T·K·2 B through one CU)
// are what its stage takes, and role B adds the two parts
template <int NST, int GPW, bool HAS_ZP, int NTL = 4, bool F32OUT = false>
__device__ __forceinline__ void chain_role_gemm(const ChainGemm& w, int cb, int rb, const __half* x_in, __half* out, int T,
                                                const ChainEdge& wait, unsigned* sig_base, int sig_shard, int sig_reps,
                                                unsigned* timeout, unsigned char* smem, const ChainArgs& p, int g_base = 0, float* out32 = nullptr) {
    static_assert(NTL == 4 || (NTL == 2 && NST == 1), "half supertiles only for single-supertile blocks");
    constexpr int V = NST * NTL * 4, CPR = NST * NTL * 2;                       // accumulator floats per lane; 16-byte chunks per output row
    float* red = reinterpret_cast<float*>(smem);                     // [8][V][64]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int a = lane >> 4, b = lane & 15;
    CH_TL(0);
    const int g0 = g_base + wave * GPW;                               // G = 8·GPW (· 2 with F32OUT; checked by the launcher)
    const int nt0 = NTL == 2 ? (cb & 1) * 2 : 0;                      // first tile of the block inside its supertile
    u32x4 wq[GPW][NST][NTL];
    uint2 scv[GPW][NST], zpv[GPW][NST];
#pragma unroll
    for (int i = 0; i < GPW; i++)
#pragma unroll
        for (int s = 0; s < NST; s++) {
            const long st = NTL == 2 ? (long)(cb >> 1) : (long)cb * NST + s;
            const u32x4* qw_lane = reinterpret_cast<const u32x4*>(w.qw) + (st * w.G * 4) * 64 + lane;
#pragma unroll
            for (int nt = 0; nt < NTL; nt++) wq[i][s][nt] = __builtin_nontemporal_load(qw_lane + ((long)(g0 + i) * 4 + nt0 + nt) * 64);
            scv[i][s] = (reinterpret_cast<const uint2*>(w.sc) + (st * w.G) * 16 + b)[(long)(g0 + i) * 16];
            if (HAS_ZP) zpv[i][s] = (reinterpret_cast<const uint2*>(w.zp) + (st * w.G) * 16 + b)[(long)(g0 + i) * 16];
        }
    __builtin_amdgcn_sched_barrier(0);
    if (wait.base) chain_wait(wait.base, wait.shards, wait.reps, wait.total, timeout);
    CH_TL(1);
    const __amdgpu_buffer_rsrc_t r_x = chain_rsrc(x_in, (long)T * w.K * 2);
    const int r_in = rb * 16 + b;
    const int xoff = ((r_in < T ? r_in : T - 1) * w.K + 8 * a) * 2;
    half8 af[2][1][4];
    auto issue_a = [&](int buf, int g) {
#pragma unroll
        for (int s = 0; s < 4; s++) af[buf][0][s] = load16_sc1(r_x, xoff + (g * 128 + 32 * s) * 2);
    };
    float4v acc[NST][1][NTL];
#pragma unroll
    for (int s = 0; s < NST; s++)
#pragma unroll
        for (int nt = 0; nt < NTL; nt++) acc[s][0][nt] = (float4v){0.f, 0.f, 0.f, 0.f};
    issue_a(0, g0);
    if (GPW > 1) issue_a(1, g0 + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < GPW; i++) {
#pragma unroll
        for (int s = 0; s < NST; s++) {
            const unsigned long long sb = ((unsigned long long)scv[i][s].y << 32) | scv[i][s].x;
            const unsigned long long zb = HAS_ZP ? (((unsigned long long)zpv[i][s].y << 32) | zpv[i][s].x) : 0ull;
            w4_consume_group<1, NTL, HAS_ZP, false>(wq[i][s], sb, zb, nt0, af[i & 1], acc[s]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (i + 2 < GPW) issue_a(i & 1, g0 + i + 2);
        __builtin_amdgcn_sched_barrier(0);
    }
    // cross-wave reduction through LDS: red[wave][v][lane], v = (supertile·4 + tile)·4 + r
#pragma unroll
    for (int s = 0; s < NST; s++)
#pragma unroll
        for (int nt = 0; nt < NTL; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[((wave * V) + (s * NTL + nt) * 4 + r) * 64 + lane] = acc[s][0][nt][r];
    __syncthreads();
    // thread → (row, 8 consecutive columns): one 16-byte write-through store
    if (threadIdx.x < 16 * CPR) {
        const int row = threadIdx.x / CPR, c8 = (threadIdx.x % CPR) * 8;
        const int aa = row >> 2, r = row & 3;
        // the row's 8 columns c8 … c8+7 lie in one tile (8 | 16): lanes aa·16 + bb … + 7 of accumulator slot v — 32 contiguous bytes
        const int tile = c8 >> 4, bb = c8 & 15;                               // tile = supertile·4 + nt
        const int v = tile * 4 + r, ln = aa * 16 + bb;
        float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ww = 0; ww < CH_W; ww++) {
            const float4v lo = *reinterpret_cast<const float4v*>(&red[(ww * V + v) * 64 + ln]);
            const float4v hi = *reinterpret_cast<const float4v*>(&red[(ww * V + v) * 64 + ln + 4]);
#pragma unroll
            for (int j = 0; j < 4; j++) { sum[j] += lo[j]; sum[4 + j] += hi[j]; }
        }
        const int r_out = rb * 16 + row;
        if constexpr (F32OUT) {
            if (r_out < T) {
                const __amdgpu_buffer_rsrc_t r_o = chain_rsrc(out32, (long)T * w.N * 4);
                const int off = (r_out * w.N + cb * 16 * NTL * NST + c8) * 4;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (float4v){sum[0], sum[1], sum[2], sum[3]}), r_o, off, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (float4v){sum[4], sum[5], sum[6], sum[7]}), r_o, off + 16, 0, 16);
            }
        } else {
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (_Float16)sum[j];
            if (r_out < T) {
                const __amdgpu_buffer_rsrc_t r_o = chain_rsrc(out, (long)T * w.N * 2);
                store16_sc1(r_o, (r_out * w.N + cb * 16 * NTL * NST + c8) * 2, o);
            }
        }
    }
    CH_TL(2);
    chain_signal(sig_base, sig_shard, sig_reps);
    CH_TL(3);
}

// ── attention role: paged_attn_kernel<128, true, 8> (decode, one new token per sequence) ────────────────────────────────
template <bool KVS, bool WIDE>
__device__ __forceinline__ void chain_role_attn(const ChainArgs& p, int wg, unsigned char* smem) {
    constexpr int HD = 128, NW = CH_W, DT = HD / 16, KS = HD / 32, OSTRIDE = HD + 4;
    float* lds_o = reinterpret_cast<float*>(smem);                    // [NW·16][OSTRIDE]
    float* lds_m = lds_o + NW * 16 * OSTRIDE;
    float* lds_l = lds_m + NW * 16;
    __half* lds_q = reinterpret_cast<__half*>(lds_l + NW * 16);       // [16][HD]
    __half* lds_kv = lds_q + 16 * HD;                                 // [2][HD]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int a = lane >> 4, b = lane & 15;
    // (the ranges of a (sequence, kv head) are neighbours; one range: no divisions on the way to the first K/V request — the role's
    // later workgroups enter when the q|k|v blocks leave their CUs, with the projection already done)
    const int NS = KVS ? p.attn_splits : 1;                          // (a kernel of its own: the one-range form stays as lean as it was)
    const int unit = NS == 1 ? wg : (int)((unsigned)wg / (unsigned)NS), split = NS == 1 ? 0 : wg - unit * NS;
    const int seq = unit % p.T, kvh = unit / p.T;
    CH_TL(0);
    const int G = p.nq / p.nkv;
    const int pos0 = (int)p.kv_lens[seq] - 1;
    const bool row_ok = b < G;
    const int row_pos = pos0;
    const int win_lo = p.sliding_window > 0 ? max(0, row_pos + 1 - p.sliding_window) : 0;
    const int kv_end = pos0 + 1;
    const int kv_begin = p.sliding_window > 0 ? max(0, pos0 + 1 - p.sliding_window) : 0;
    const int all_lo = (kv_begin / KV_BLOCK) / 2;
    const int all_hi = (cdiv_dev(kv_end, KV_BLOCK) + 1) / 2;         // exclusive: block pairs of the whole context
    // this split's share of the pairs (the last split owns the pair with the new token: it patches and writes its K / V)
    const unsigned np_all = (unsigned)max(0, all_hi - all_lo);       // (≤ 2^16 pairs × ≤ 16 ranges: 32-bit)
    const int my_lo = NS == 1 ? all_lo : all_lo + (int)(np_all * (unsigned)split / (unsigned)NS);
    const int my_hi = NS == 1 ? all_hi : all_lo + (int)(np_all * (unsigned)(split + 1) / (unsigned)NS);
    const int first = my_lo + wave;
    const int nblocks = cdiv_dev(kv_end, KV_BLOCK);
    const int32_t* bt = p.block_tables + (long)seq * p.max_blocks;
    const long tile_elems = kv_tile_elems(HD);
    int bt_lo = 2 * my_lo;
    int btv = bt[min(bt_lo + lane, p.max_blocks - 1)];
    struct Frags { half8 k0[KS], k1[KS], v0[KS], v1[KS]; };
    auto issue = [&](int pr, Frags& f) {
        const int blk0 = 2 * pr, blk1 = 2 * pr + 1;
        const bool has1 = blk1 < nblocks;
        if (blk1 - bt_lo >= 64) {
            bt_lo = blk0;
            btv = bt[min(bt_lo + lane, p.max_blocks - 1)];
        }
        const long phys0 = __builtin_amdgcn_readlane(btv, blk0 - bt_lo);
        const long phys1 = has1 ? __builtin_amdgcn_readlane(btv, blk1 - bt_lo) : phys0;
        const __half* k0 = p.k_pool + (phys0 * p.nkv + kvh) * tile_elems + lane * 8;
        const __half* k1 = p.k_pool + (phys1 * p.nkv + kvh) * tile_elems + lane * 8;
        const __half* v0 = p.v_pool + (phys0 * p.nkv + kvh) * tile_elems + lane * 8;
        const __half* v1 = p.v_pool + (phys1 * p.nkv + kvh) * tile_elems + lane * 8;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            f.k0[s] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(k0 + s * 512));
            f.k1[s] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(k1 + s * 512));
        }
#pragma unroll
        for (int s = 0; s < KS; s++) {
            f.v0[s] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(v0 + s * 512));
            f.v1[s] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(v1 + s * 512));
        }
    };
    Frags cur, nxt;
    if (first < my_hi) issue(first, cur);
    const int last_blk = pos0 / KV_BLOCK, slot_new = pos0 % KV_BLOCK;
    // norm weights and the RoPE row of this position do not depend on the projection either
    __builtin_amdgcn_sched_barrier(0);
    // the q|k|v columns of this kv head: G + 2 heads, each one workgroup per 16-row block of the projection role
    chain_wait(p.cnt + (CH_QKV_SLOT + kvh * CH_QKV_R) * CH_STRIDE, 1, CH_QKV_R, (unsigned)((G + 2) * (WIDE ? 1 : (p.qkv_half ? 4 : 2)) * ((p.T + 15) >> 4)), p.timeout);
    CH_TL(1);
    half8 qf[KS];
    {
        constexpr int HALF = HD / 2;
        const int q_dim = p.nq * HD, kv_dim = p.nkv * HD;
        const __half* qrow = p.qkv_out + (long)seq * (q_dim + 2 * kv_dim);
        const int r16 = threadIdx.x >> 4, q16 = threadIdx.x & 15;
        const int rc = r16 < G + 2 ? r16 : G + 1;
        const __half* src = rc < G ? qrow + (kvh * G + rc) * HD
                                   : rc == G ? qrow + q_dim + kvh * HD : qrow + q_dim + kv_dim + kvh * HD;
        const RopeRow<HD> rr = rope_row16<HD, true>(src, rc < G ? p.q_norm_w : p.k_norm_w, p.cos_t + (long)pos0 * HALF,
                                                    p.sin_t + (long)pos0 * HALF, rc <= G ? p.qk_mode : 0, p.qk_mode == 1,
                                                    p.qk_mode != 0, p.eps, q16);
        using hv = typename RopeRow<HD>::hv;
        hv z0, z1;
#pragma unroll
        for (int k = 0; k < HD / 32; k++) { z0[k] = (_Float16)0.f; z1[k] = (_Float16)0.f; }
        _Float16* lq = reinterpret_cast<_Float16*>(lds_q) + r16 * HD;
        if (r16 < 16) {
            *reinterpret_cast<hv*>(lq + rr.off0) = r16 < G ? rr.out0 : z0;
            *reinterpret_cast<hv*>(lq + rr.off1) = r16 < G ? rr.out1 : z1;
        }
        if (r16 == G || r16 == G + 1) {
            _Float16* lk = reinterpret_cast<_Float16*>(lds_kv) + (r16 - G) * HD;
            *reinterpret_cast<hv*>(lk + rr.off0) = rr.out0;
            *reinterpret_cast<hv*>(lk + rr.off1) = rr.out1;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < KS; s++) qf[s] = *reinterpret_cast<const half8*>(lds_q + b * HD + 32 * s + 8 * a);
        // the new token's K (wave 0) and V (wave 1) go to the cache (read by later launches only)
        if (wave < 2 && split == NS - 1) {
            const long phys = p.block_tables[(long)seq * p.max_blocks + last_blk];
            const long toff = (phys * p.nkv + kvh) * kv_tile_elems(HD);
            for (int d = lane; d < HD; d += 64) {
                if (wave == 0) p.k_pool[toff + k_tile_off(slot_new, d)] = lds_kv[d];
                else p.v_pool[toff + v_tile_off(slot_new, d)] = lds_kv[HD + d];
            }
        }
    }
    float m_run = -INFINITY, l_run = 0.f;
    float4v o_acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; dt++) o_acc[dt] = (float4v){0.f, 0.f, 0.f, 0.f};
    for (int pr = first; pr < my_hi; pr += NW) {
        const int blk0 = 2 * pr, blk1 = 2 * pr + 1;
        const bool has1 = blk1 < nblocks;
        if (pr + NW < my_hi) issue(pr + NW, nxt);
        __builtin_amdgcn_sched_barrier(0);
        half8 (&kf0)[KS] = cur.k0, (&kf1)[KS] = cur.k1, (&vf0)[KS] = cur.v0, (&vf1)[KS] = cur.v1;
        if (blk0 == last_blk || blk1 == last_blk) {
            // patch the new token's slot (key slot_new of block last_blk) into the loaded fragments
            const bool in1 = blk1 == last_blk;
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const half8 nk = *reinterpret_cast<const half8*>(lds_kv + 32 * s + 8 * a);
                if (b == slot_new) { if (in1) kf1[s] = nk; else kf0[s] = nk; }
            }
            const bool a_hit = a == (slot_new >> 2);
            const int jn = slot_new & 3;
#pragma unroll
            for (int ld = 0; ld < KS; ld++) {
#pragma unroll
                for (int sub = 0; sub < 2; sub++) {
                    const _Float16 nv = (_Float16)lds_kv[HD + 32 * ld + 16 * sub + b];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (a_hit && j == jn) { if (in1) vf1[ld][4 * sub + j] = nv; else vf0[ld][4 * sub + j] = nv; }
                    }
                }
            }
        }
        float4v s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; s++) {
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf0[s], qf[s], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf1[s], qf[s], s1, 0, 0, 0);
        }
        float sc[8];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int kp0 = blk0 * KV_BLOCK + 4 * a + r, kp1 = blk1 * KV_BLOCK + 4 * a + r;
            const bool ok0 = row_ok && kp0 <= row_pos && kp0 >= win_lo;
            const bool ok1 = row_ok && has1 && kp1 <= row_pos && kp1 >= win_lo;
            sc[r] = ok0 ? s0[r] * p.scale : -INFINITY;
            sc[4 + r] = ok1 ? s1[r] * p.scale : -INFINITY;
            mx = fmaxf(mx, fmaxf(sc[r], sc[4 + r]));
        }
        mx = rows_reduce_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float m_safe = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __expf(m_run - m_safe);
        float psum = 0.f;
        half8 pf;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float pv = __expf(sc[j] - m_safe);
            psum += pv;
            pf[j] = (_Float16)pv;
        }
        psum = rows_reduce_sum(psum);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < DT; dt++) {
            const int ld = dt >> 1, sub = (dt & 1) * 2;
            const u32x4 w0 = __builtin_bit_cast(u32x4, vf0[ld]), w1 = __builtin_bit_cast(u32x4, vf1[ld]);
            const half8 vfrag = __builtin_bit_cast(half8, (u32x4){w0[sub], w0[sub + 1], w1[sub], w1[sub + 1]});
#pragma unroll
            for (int r = 0; r < 4; r++) o_acc[dt][r] *= alpha;
            o_acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfrag, pf, o_acc[dt], 0, 0, 0);
        }
        cur = nxt;
    }
    // merge the 8 waves' partial states through LDS (wave order)
    if (a == 0) {
        lds_m[wave * 16 + b] = m_run;
        lds_l[wave * 16 + b] = l_run;
    }
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
        *reinterpret_cast<float4v*>(&lds_o[(wave * 16 + b) * OSTRIDE + dt * 16 + 4 * a]) = o_acc[dt];
    __syncthreads();
    constexpr int TPR = NW * 4;                                       // threads per row: 32 → 4 dims each
    const int row = threadIdx.x / TPR, dl = threadIdx.x % TPR;
    float mw[NW], M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; w++) { mw[w] = lds_m[w * 16 + row]; M = fmaxf(M, mw[w]); }
    const float Ms = M == -INFINITY ? 0.f : M;
    float fw[NW], L = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) { fw[w] = __expf(mw[w] - Ms); L += lds_l[w * 16 + row] * fw[w]; }
    constexpr int DPT = HD / TPR;
    float ov[DPT];
#pragma unroll
    for (int i = 0; i < DPT; i++) {
        const int d = dl * DPT + i;
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) acc += lds_o[(w * 16 + row) * OSTRIDE + d] * fw[w];
        ov[i] = acc;
    }
    float Lm = L, Mm = M;
    if (NS > 1) {
        // several KV ranges per (sequence, kv head): every split leaves its state (m, l, unnormalised o) write-through and takes a
        // ticket; the LAST to arrive reads all of them back (sc1) and merges them in split order — whoever it is, the same bits
        static_assert(DPT == 4, "one 16-byte piece per thread");
        constexpr int PST = HD + 4;
        float* mine = p.attn_partial + ((long)unit * NS + split) * 16 * PST + row * PST;
        const __amdgpu_buffer_rsrc_t pr_ = chain_rsrc(p.attn_partial, (long)p.T * p.nkv * NS * 16 * PST * 4);
        const int my_off = (int)((mine - p.attn_partial) * 4);
        if (row < G) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (float4v){ov[0], ov[1], ov[2], ov[3]}), pr_, my_off + dl * 16, 0, 16);
            if (dl == 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (float4v){M, L, 0.f, 0.f}), pr_, my_off + HD * 4, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* s_tk = reinterpret_cast<int*>(lds_m);                     // (the merge above is done with lds_m)
        if (threadIdx.x == 0) *s_tk = (int)__hip_atomic_fetch_add(p.attn_tickets + unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*s_tk != NS - 1) { CH_TL(2); CH_TL(3); return; }           // not the last: the merge (and the signal) is someone else's
        if (threadIdx.x == 0) __hip_atomic_store(p.attn_tickets + unit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (row < G) {
            const int base = (int)(((long)unit * NS * 16 * PST + row * PST) * 4);
            float ms[CH_MAX_SPLITS], ls[CH_MAX_SPLITS];
            Mm = -INFINITY;
#pragma unroll
            for (int sp = 0; sp < CH_MAX_SPLITS; sp++) {
                if (sp < NS) {
                    const float4v ml = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(pr_, base + sp * 16 * PST * 4 + HD * 4, 0, 16));
                    ms[sp] = ml[0]; ls[sp] = ml[1];
                    Mm = fmaxf(Mm, ms[sp]);
                }
            }
            const float Msafe = Mm == -INFINITY ? 0.f : Mm;
            Lm = 0.f;
#pragma unroll
            for (int i = 0; i < DPT; i++) ov[i] = 0.f;
#pragma unroll
            for (int sp = 0; sp < CH_MAX_SPLITS; sp++) {
                if (sp < NS) {
                    const float4v os = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(pr_, base + sp * 16 * PST * 4 + dl * 16, 0, 16));
                    const float f = __expf(ms[sp] - Msafe);
                    Lm += ls[sp] * f;
#pragma unroll
                    for (int i = 0; i < DPT; i++) ov[i] += os[i] * f;
                }
            }
        }
    }
    if (row < G) {
        const float inv = Lm > 0.f ? 1.0f / Lm : 0.f;
        union { _Float16 h[4]; unsigned long long u; } o;
#pragma unroll
        for (int i = 0; i < DPT; i++) o.h[i] = (_Float16)(ov[i] * inv);
        __half* dst = p.attn_out + ((long)seq * p.nq + kvh * G + row) * HD + dl * DPT;
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), o.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // 8-byte sc1 store
    }
    CH_TL(2);
    chain_signal(p.cnt + CH_ATTN_SLOT * CH_STRIDE, unit % CH_ATTN_SH, CH_ATTN_R);
    CH_TL(3);
}

// ── role B: residual += o; post-attention norm; router logits of this part's experts; candidates; in-launch merge ─────────
// (add_rmsnorm_route_part_kernel<false> with arrive: grid (token, part) → wg = token·Q + part)
__device__ __forceinline__ void chain_role_b(const ChainArgs& p, int wg, unsigned n_o_wgs, unsigned char* smem) {
    const int H = p.H, Q = p.Q, num_experts = p.E, top_k = p.r_top_k;
    const long row = wg / Q;
    const int q = wg % Q;
    CH_TL(0);
    __half* xs = reinterpret_cast<__half*>(smem);
    float* part = reinterpret_cast<float*>(smem + (size_t)H * 2);
    const int nvec = H >> 3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int CH = 2;
    // requested before the wait: the norm weights and this wave's router weights (its whole share: 16 k-steps × 1 KiB)
    half8 wv_pre[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int i = threadIdx.x + c * 512;
        wv_pre[c] = *reinterpret_cast<const half8*>(p.post_ln + (i < nvec ? i : 0) * 8);
    }
    constexpr int U = 16;
    const int tiles = (num_experts + 15) >> 4;
    const int tiles_q = num_experts > 0 ? tiles / Q : 0;              // (dense model: no router, Q = 1)
    const int ksplit = (tiles_q >= 8 || tiles_q == 0) ? 1 : 8 / tiles_q;
    const int ksteps = H >> 5;
    half8 bw_pre[U];
    bool pre = false;
    if (wave < tiles_q * ksplit) {
        const int tl = wave / ksplit, ks = wave % ksplit;
        const int s0 = ksteps * ks / ksplit, s1 = ksteps * (ks + 1) / ksplit;
        if (s0 + U <= s1) {
            const __half* wrow = p.router_w + ((long)(q * tiles_q + tl) * ksteps * 64 + lane) * 8;
#pragma unroll
            for (int kk = 0; kk < U; kk++) bw_pre[kk] = *reinterpret_cast<const half8*>(wrow + (long)(s0 + kk) * 512);
            pre = true;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    chain_wait(p.cnt + CH_O_SLOT * CH_STRIDE, CH_O_SH, CH_O_R, n_o_wgs, p.timeout);
    CH_TL(1);
    float* red = part + tiles_q * 16 * (ksplit + 1);                  // [8] behind the logits
    unsigned long long* cand_s = reinterpret_cast<unsigned long long*>(red + 8);     // [8]
    const __amdgpu_buffer_rsrc_t r_x = chain_rsrc(p.o_part, (long)CH_O_KS * p.T * H * 4), r_res = chain_rsrc(p.res_a, (long)p.T * H * 2);
    half8 v[CH];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int i = threadIdx.x + c * 512;
        if (i < nvec) {
            const int off = (int)((row * H + i * 8) * 2);
            // o_proj's row: its K parts (fp32) added in part order, rounded to the fp16 the projection's output is
            float xs32[8];
#pragma unroll
            for (int ks = 0; ks < CH_O_KS; ks++) {
                const int o32 = (int)((((long)ks * p.T + row) * H + i * 8) * 4);
                const float4v lo = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(r_x, o32, 0, 16));
                const float4v hi = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(r_x, o32 + 16, 0, 16));
#pragma unroll
                for (int j = 0; j < 4; j++) { xs32[j] = ks ? xs32[j] + lo[j] : lo[j]; xs32[4 + j] = ks ? xs32[4 + j] + hi[j] : hi[j]; }
            }
            half8 xv;
#pragma unroll
            for (int j = 0; j < 8; j++) xv[j] = (_Float16)xs32[j];
            half8 rv = load16_sc1(r_res, off);
#pragma unroll
            for (int j = 0; j < 8; j++) rv[j] = (_Float16)((float)rv[j] + (float)xv[j]);
            v[c] = rv;
#pragma unroll
            for (int j = 0; j < 8; j++) ss += (float)rv[j] * (float)rv[j];
        }
    }
    ss = wave_reduce_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float total = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
    const float inv = 1.0f / sqrtf(total / (float)H + p.eps);
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int i = threadIdx.x + c * 512;
        if (i < nvec) {
            const half8 wv = wv_pre[c];
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (_Float16)((float)v[c][j] * inv * (float)wv[j]);
            *reinterpret_cast<half8*>(xs + i * 8) = o;
            if (q == 0) {
                *reinterpret_cast<half8*>(p.norm2 + row * H + i * 8) = o;
                *reinterpret_cast<half8*>(p.res_b_out + row * H + i * 8) = v[c];
            }
        }
    }
    if (num_experts <= 0) { CH_TL(2); CH_TL(3); return; }           // dense model: add + norm only (norm2 / res_b_out written above)
    __syncthreads();
    const int a = lane >> 4, b = lane & 15;
    for (int u = wave; u < tiles_q * ksplit; u += 8) {
        const int tl = u / ksplit, ks = u % ksplit;
        const int tile = q * tiles_q + tl;
        const int s0 = ksteps * ks / ksplit, s1 = ksteps * (ks + 1) / ksplit;
        const __half* wrow = p.router_w + ((long)tile * ksteps * 64 + lane) * 8;
        float4v acc = {0.f, 0.f, 0.f, 0.f};
        int s = s0;
        if (u == wave && pre) {
#pragma unroll
            for (int kk = 0; kk < U; kk++) {
                half8 av = {0, 0, 0, 0, 0, 0, 0, 0};
                if (b == 0) av = *reinterpret_cast<const half8*>(xs + (s + kk) * 32 + 8 * a);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bw_pre[kk], acc, 0, 0, 0);
            }
            s += U;
        }
        for (; s + U <= s1; s += U) {
            half8 bw[U];
#pragma unroll
            for (int k = 0; k < U; k++) bw[k] = *reinterpret_cast<const half8*>(wrow + (long)(s + k) * 512);
#pragma unroll
            for (int k = 0; k < U; k++) {
                half8 av = {0, 0, 0, 0, 0, 0, 0, 0};
                if (b == 0) av = *reinterpret_cast<const half8*>(xs + (s + k) * 32 + 8 * a);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bw[k], acc, 0, 0, 0);
            }
        }
        for (; s < s1; s++) {
            const half8 bwv = *reinterpret_cast<const half8*>(wrow + (long)s * 512);
            half8 av = {0, 0, 0, 0, 0, 0, 0, 0};
            if (b == 0) av = *reinterpret_cast<const half8*>(xs + s * 32 + 8 * a);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bwv, acc, 0, 0, 0);
        }
        if (a == 0) part[ks * tiles_q * 16 + tl * 16 + b] = acc[0];
    }
    __syncthreads();
    const int EQ = tiles_q * 16;
    float* lgs = part + ksplit * EQ;
    const int t = threadIdx.x;
    const int e_glob = q * EQ + t;
    float l = -INFINITY;
    if (t < EQ && e_glob < num_experts) {
        l = 0.f;
        for (int ks = 0; ks < ksplit; ks++) l += part[ks * EQ + t];
    }
    if (t < EQ) lgs[t] = l;
    float mx = wave_reduce_max(l);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(red[0], red[1]);
    __syncthreads();
    const float ex = (t < EQ && e_glob < num_experts) ? expf(l - mx) : 0.f;
    const float sm = wave_reduce_sum(ex);
    if (lane == 0) red[wave] = sm;
    const int keep = top_k < EQ ? top_k : EQ;
    if (t < EQ) {
        int rank = 0;
        for (int j = 0; j < EQ; j++) {
            const float lj = lgs[j];
            rank += (lj > l || (lj == l && j < t)) ? 1 : 0;
        }
        if (rank < keep) cand_s[rank] = ((unsigned long long)(unsigned)e_glob << 32) | __float_as_uint(l);
    }
    if (t >= keep && t < 8) cand_s[t] = (0x7fffffffull << 32) | __float_as_uint(-INFINITY);
    __syncthreads();
    CH_TL(2);
    if (wave != 0) return;
    if (p.defer_merge) {
        // the consumer is the NEXT launch (the grouped gate_up GEMM merges the Q lists of every token in its prologue, under its
        // first weight loads): plain stores, no meeting of the parts here — the ≈ 3.7 µs merge leaves the layer's critical path
        unsigned long long* cand_g = reinterpret_cast<unsigned long long*>(p.cand) + (row * Q + q) * 8;
        if (lane < 8) cand_g[lane] = cand_s[lane];
        if (lane == 8) reinterpret_cast<unsigned long long*>(p.stats)[row * Q + q] =
            ((unsigned long long)__float_as_uint(red[0] + red[1]) << 32) | __float_as_uint(mx);
        CH_TL(3);
        return;
    }
    // The Q parts of a token meet without a counter: every part but the first publishes its 8 candidates and its two softmax
    // statistics as TAGGED 8-byte granules (one write-through store each — the data is the flag, Guideline 16 R2), and part 0
    // sweeps the (Q − 1)·10 granules of its token until every tag is there, merges, and clears them for the launch after the
    // next (the granules live in the double-buffered counter block: zero on entry).  One store → load latency instead of
    // store → drain → returning ticket → loads: 3.4–3.8 µs → ≈ 1.5 µs on the layer's critical path.
    //   candidate granule: [tag 1 : 16][expert id : 16][logit bits : 32]      statistic granule: [tag 1 : 32][value bits : 32]
    unsigned long long* gran = reinterpret_cast<unsigned long long*>(p.cnt + CH_SLOTS * CH_STRIDE) + (row * Q) * CH_GRAN;
    if (q != 0) {
        unsigned long long g = 0;
        if (lane < 8) g = (1ull << 48) | ((cand_s[lane] >> 32 & 0xffffull) << 32) | (cand_s[lane] & 0xffffffffull);
        else if (lane == 8) g = (1ull << 32) | __float_as_uint(mx);
        else if (lane == 9) g = (1ull << 32) | __float_as_uint(red[0] + red[1]);
        if (lane < CH_GRAN) __hip_atomic_store(gran + q * CH_GRAN + lane, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    unsigned long long* gr = cand_s + 8;                               // LDS [Q][CH_GRAN] behind this part's own list
    {
        const int n = (Q - 1) * CH_GRAN;
        unsigned long long* src = gran + CH_GRAN + lane;
        const bool is_stat = (lane % CH_GRAN) >= 8;
        unsigned long long g = 0;
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
            if (lane < n) g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool have = lane >= n || (is_stat ? (g >> 32) == 1ull : (g >> 48) == 1ull);
            if (__ballot(!have) == 0ull) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0u && wall_clock64() - t0 > 2000000ull) {
                if (lane == 0) __hip_atomic_fetch_add(p.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        if (lane < n) {
            gr[CH_GRAN + lane] = g;
            __hip_atomic_store(src, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (also cleared with the half by a later launch)
        }
        if (lane < 8) gr[lane] = (1ull << 48) | ((cand_s[lane] >> 32 & 0xffffull) << 32) | (cand_s[lane] & 0xffffffffull);
        else if (lane == 8) gr[8] = (1ull << 32) | __float_as_uint(mx);
        else if (lane == 9) gr[9] = (1ull << 32) | __float_as_uint(red[0] + red[1]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // one wave: its own LDS writes, in order
        __builtin_amdgcn_wave_barrier();
    }
    const int ncand = Q * 8;
    unsigned long long c = (0x7fffffffull << 32) | __float_as_uint(-INFINITY);
    if (lane < ncand) {
        const unsigned long long g = gr[(lane >> 3) * CH_GRAN + (lane & 7)];
        const unsigned id16 = (unsigned)(g >> 32) & 0xffffu;
        c = ((unsigned long long)(id16 == 0xffffu ? 0x7fffffffu : id16) << 32) | (g & 0xffffffffull);
    }
    float pmx = -INFINITY, psum = 0.f;
    if (lane < Q) {
        pmx = __uint_as_float((unsigned)gr[lane * CH_GRAN + 8]);
        psum = __uint_as_float((unsigned)gr[lane * CH_GRAN + 9]);
    }
    float ww = 0.f;
    const int rank = route_merge_token(c, ncand, pmx, psum, Q, top_k, p.norm_topk, true, &ww);
    if (rank < top_k) {
        p.ids[row * top_k + rank] = (int)(c >> 32);
        p.weights[row * top_k + rank] = ww;
    }
    CH_TL(3);
}

// KVS / WIDE are kernels of their own, not branches: the one-range, 64-column kernel keeps the instruction layout it was tuned
// with (with either compiled in as a runtime branch it lost ≈ 1 % at every batch size — roles entering late have their first
// instructions on the critical path)
template <int GPW_QKV, int GPW_O, bool HAS_ZP, bool KVS = false, bool WIDE = false>
__global__ __launch_bounds__(512, 2) void decode_chain_kernel(ChainArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[CH_SMEM];
    const int RH = (p.T + 15) >> 4;                                   // 16-row blocks
    const int QB = WIDE ? 128 : (p.qkv_half ? 32 : 64);             // q|k|v block width
    const int n_a = p.has_a ? p.T : 0, n_qkv = p.qkv.N / QB * RH, n_attn = p.T * p.nkv * (KVS ? p.attn_splits : 1), n_o = p.o.N / (p.o_half ? 32 : 64 * CH_O_NST) * RH * CH_O_KS;
    int wg = blockIdx.x;
    if (wg == 0) {                                                    // re-arm the other half: counters and route granules
        if (threadIdx.x < CH_QKV_SLOT + CH_QKV_R * p.nkv) p.cnt_next[threadIdx.x * CH_STRIDE] = 0u;
        for (int i = threadIdx.x; i < p.T * p.Q * CH_GRAN * 2; i += 512) p.cnt_next[CH_SLOTS * CH_STRIDE + i] = 0u;
    }
    if (wg < n_a) { chain_role_a(p, wg, smem); return; }
    wg -= n_a;
    if (wg < n_qkv) {
        // a column block is one head: its arrival counts for that head's kv head
        const int cb = wg / RH, head = cb * QB / 128, G = p.nq / p.nkv;           // (a head is 128 columns)
        const int kvh = head < p.nq ? head / G : (head < p.nq + p.nkv ? head - p.nq : head - p.nq - p.nkv);
        const ChainEdge e{p.has_a ? p.cnt + CH_NORM_SLOT * CH_STRIDE : nullptr, CH_NORM_SH, CH_NORM_R, (unsigned)p.T};
        if constexpr (WIDE) {
            // one head (128 columns) per workgroup: half as many workgroups of twice the bytes — taken where the narrow blocks and the
            // attention role together exceed the workgroups the chip holds at once (decode_chain_f16)
            chain_role_gemm<2, GPW_QKV, HAS_ZP>(p.qkv, cb, wg % RH, p.norm1, p.qkv_out, p.T, e,
                                                p.cnt + (CH_QKV_SLOT + kvh * CH_QKV_R) * CH_STRIDE, 0, CH_QKV_R, p.timeout, smem, p);
        } else if (p.qkv_half)
            chain_role_gemm<1, GPW_QKV, HAS_ZP, 2>(p.qkv, cb, wg % RH, p.norm1, p.qkv_out, p.T, e,
                                                   p.cnt + (CH_QKV_SLOT + kvh * CH_QKV_R) * CH_STRIDE, 0, CH_QKV_R, p.timeout, smem, p);
        else
            chain_role_gemm<CH_QKV_NST, GPW_QKV, HAS_ZP>(p.qkv, cb, wg % RH, p.norm1, p.qkv_out, p.T, e,
                                                         p.cnt + (CH_QKV_SLOT + kvh * CH_QKV_R) * CH_STRIDE, 0, CH_QKV_R, p.timeout, smem, p);
        return;
    }
    wg -= n_qkv;
    if (wg < n_attn) { chain_role_attn<KVS, WIDE>(p, wg, smem); return; }
    wg -= n_attn;
    if (wg < n_o) {
        const ChainEdge e{p.cnt + CH_ATTN_SLOT * CH_STRIDE, CH_ATTN_SH, CH_ATTN_R, (unsigned)(p.T * p.nkv)};      // (one arrival per (sequence, kv head): its last split's)
        // (K part fastest: the two workgroups of a block are neighbours)
        static_assert(GPW_O % CH_O_KS == 0, "o_proj's quant groups per wave divide over its K parts");
        const int ks = wg % CH_O_KS, blk = wg / CH_O_KS;
        float* part = p.o_part + (long)ks * p.T * p.o.N;
        if (p.o_half)
            chain_role_gemm<1, GPW_O / CH_O_KS, HAS_ZP, 2, true>(p.o, blk / RH, blk % RH, p.attn_out, nullptr, p.T, e, p.cnt + CH_O_SLOT * CH_STRIDE,
                                                                 wg % CH_O_SH, CH_O_R, p.timeout, smem, p, ks * (p.o.G / CH_O_KS), part);
        else
            chain_role_gemm<CH_O_NST, GPW_O / CH_O_KS, HAS_ZP, 4, true>(p.o, blk / RH, blk % RH, p.attn_out, nullptr, p.T, e, p.cnt + CH_O_SLOT * CH_STRIDE,
                                                                        wg % CH_O_SH, CH_O_R, p.timeout, smem, p, ks * (p.o.G / CH_O_KS), part);
        return;
    }
    wg -= n_o;
    chain_role_b(p, wg, (unsigned)n_o, smem);
}

}  // namespace

#ifdef FERRUM_HIP_EXPERIMENTS
extern "C" __attribute__((visibility("default"))) void ferrum_hip_debug_set_chain_timeline(void* p) { g_chain_timeline = (unsigned long long*)p; }
#endif

int decode_chain_counter_words() { return CH_SLOTS * CH_STRIDE + CH_GRAN_WORDS; }

bool decode_chain_supports(const DecodeChainDesc& d) {
    const auto gemm_ok = [](const W4Device& w) {
        return w.qw && !w.zp && !w.perm && !w.bias && !w.f16t && w.G % 8 == 0 && (w.G / 8 == 2 || w.G / 8 == 4) && w.n % 128 == 0 &&
               (long)CH_MAX_T * std::max(w.n, w.k) * 2 < (1L << 31);
    };
    if (!d.qkv || !d.o || !gemm_ok(*d.qkv) || !gemm_ok(*d.o)) return false;
    if (d.o->G / 8 != 4) return false;              // (the instantiated forms: o_proj's K slice per wave is four groups)
    if (!d.o_part) return false;
    if (d.attn_splits > CH_MAX_SPLITS || (d.attn_splits > 1 && (!d.attn_partial || !d.attn_tickets))) return false;
    if (d.T < 1 || d.T > CH_MAX_T || d.head_dim != 128 || d.nkv < 1 || d.nq % d.nkv != 0 || d.nq / d.nkv > 14 || d.nkv > CH_MAX_KVH) return false;
    if (d.H % 32 != 0 || d.H > 8192 || d.qkv->k != d.H || d.qkv->n != (d.nq + 2 * d.nkv) * 128 || d.o->k != d.nq * 128 || d.o->n != d.H) return false;
    if (d.E <= 0) {       // dense model: role B is add + norm, role A sums the down projection's slabs (or adds its fp16 rows)
        if (d.Q != 1 || (d.has_a && !d.a_x && (!d.a_slabs || d.a_S < 1 || d.a_ld < d.H))) return false;
        return (size_t)d.H * 2 + 64 * 4 <= (size_t)CH_SMEM;
    }
    const int tiles = (d.E + 15) / 16;
    if (d.E > 0xfff0 || d.Q < 1 || d.Q > 4 || tiles % d.Q != 0 || tiles / d.Q > 8 || d.r_top_k < 1 || d.r_top_k > 8) return false;
    if (d.has_a && (d.top_k < 1)) return false;
    // LDS of role B: the row + the part logits
    const int tiles_q = tiles / d.Q, ksplit = tiles_q >= 8 ? 1 : 8 / tiles_q;
    if ((size_t)d.H * 2 + (size_t)tiles_q * 16 * (ksplit + 1) * 4 + 8 * 4 + (8 + 4 * CH_GRAN) * 8 + 16 > (size_t)CH_SMEM) return false;
    return true;
}

int decode_chain_f16(const DecodeChainDesc& d, hipStream_t stream) {
    FH_REQUIRE(decode_chain_supports(d), "decode_chain: shapes not taken by the merged form");
    ChainArgs a{};
    a.T = d.T; a.H = d.H; a.nq = d.nq; a.nkv = d.nkv;
    a.has_a = d.has_a ? (d.E > 0 ? 1 : (d.a_x ? 3 : 2)) : 0; a.top_k = d.top_k;
    a.a_x = d.a_x;
    a.a_slabs = d.a_slabs; a.a_S = d.a_S; a.a_slab_stride = d.a_slab_stride; a.a_ld = d.a_ld; a.down = d.down; a.comb_w = d.comb_w; a.res_in = d.res_in; a.ln_in = d.ln_in;
    a.eps = d.eps; a.res_a = d.res_a; a.norm1 = d.norm1;
    a.qkv = ChainGemm{d.qkv->qw, d.qkv->sc, d.qkv->zp, d.qkv->G, d.qkv->n, d.qkv->k};
    a.qkv_out = d.qkv_out;
    a.k_pool = d.k_pool; a.v_pool = d.v_pool; a.block_tables = d.block_tables; a.kv_lens = d.kv_lens;
    a.q_norm_w = d.q_norm_w; a.k_norm_w = d.k_norm_w; a.cos_t = d.cos_t; a.sin_t = d.sin_t;
    a.qk_mode = d.qk_mode; a.max_blocks = d.max_blocks; a.sliding_window = d.sliding_window;
    a.scale = 1.0f / sqrtf((float)d.head_dim);
    a.attn_out = d.attn_out;
    a.o = ChainGemm{d.o->qw, d.o->sc, d.o->zp, d.o->G, d.o->n, d.o->k};
    a.o_out = d.o_out;
    a.res_b_out = d.res_b_out; a.post_ln = d.post_ln; a.norm2 = d.norm2; a.router_w = d.router_w;
    a.E = d.E; a.r_top_k = d.r_top_k; a.Q = d.Q; a.norm_topk = d.norm_topk; a.defer_merge = d.defer_merge ? 1 : 0;
    a.cand = d.cand; a.stats = d.stats; a.route_arrive = d.route_arrive; a.ids = d.ids; a.weights = d.weights;
    a.cnt = d.cnt; a.cnt_next = d.cnt_next; a.timeout = d.timeout;
#ifdef FERRUM_HIP_EXPERIMENTS
    a.tl = g_chain_timeline;
#endif
    const int rh = (d.T + 15) / 16;
    a.attn_splits = std::max(1, d.attn_splits); a.attn_partial = d.attn_partial; a.attn_tickets = d.attn_tickets;
    a.qkv_half = (knobs().chain_qkv_half && d.T <= 16) ? 1 : 0;
    // (narrow o_proj only: 2048 columns → 64 blocks; Llama-3.1-8B's 4096 columns are 64 blocks of 64 already: c=8 1.908 → 1.930 ms with 32)
    a.o_half = (knobs().chain_o_half && d.T <= 16 && d.o->n <= 2048) ? 1 : 0;
    // 128-column q|k|v blocks (hidden 2048 only: the two-group weight ring): where the 64-column blocks + the attention role do not fit
    // the chip's resident workgroups, the role's last workgroups enter behind the projection (≈ 2.7 µs of K/V round trips on the
    // critical path, profiles/r03_decode_chain_timeline.txt) — c=32 4.025 → 3.995 ms per step, c=64 4.83 → 4.69; c=20–24 (where they fit) 3 % slower
    a.o_part = d.o_part;
    a.qkv_wide = 0;
    if (!a.qkv_half && d.qkv->G / 8 == 2 && a.attn_splits == 1) {
        const int wide = knobs().chain_qkv_wide;
        const int n64 = d.qkv->n / 64 * rh, n_at = d.T * d.nkv * a.attn_splits;
        a.qkv_wide = wide >= 0 ? (wide > 0) : (n64 + n_at > knobs().chain_slots);
    }
    const int blocks = (d.has_a ? d.T : 0) + d.qkv->n / (a.qkv_half ? 32 : (a.qkv_wide ? 128 : 64)) * rh + d.T * d.nkv * a.attn_splits + d.o->n / (a.o_half ? 32 : 64 * CH_O_NST) * rh * CH_O_KS + d.T * d.Q;
    form_hit(FORM_DECODE_CHAIN);
    if (a.attn_splits > 1) form_hit(FORM_CHAIN_ATTN_KV_SPLITS);
    if (a.qkv_wide) form_hit(FORM_CHAIN_QKV_WIDE);
    if (a.attn_splits > 1) {
        if (d.qkv->G / 8 == 2) hipLaunchKernelGGL((decode_chain_kernel<2, 4, false, true>), dim3(blocks), dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((decode_chain_kernel<4, 4, false, true>), dim3(blocks), dim3(512), 0, stream, a);
    } else if (a.qkv_wide) hipLaunchKernelGGL((decode_chain_kernel<2, 4, false, false, true>), dim3(blocks), dim3(512), 0, stream, a);
    else if (d.qkv->G / 8 == 2) hipLaunchKernelGGL((decode_chain_kernel<2, 4, false>), dim3(blocks), dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((decode_chain_kernel<4, 4, false>), dim3(blocks), dim3(512), 0, stream, a);
    FH_CHECK_LAUNCH();
    return 0;
}

}  // namespace fh
