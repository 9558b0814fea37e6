// Tensor-parallel communicator (C-ABI `ferrum_hip_comm_*`): BackendCollective of the reference
// (ferrum-kernels/src/backend/capabilities.rs:84-109; CUDA lane nccl_comm.rs:21-49, tp_decode.rs:350-372 — an fp16 sum
// all-reduce of [T, H] after o_proj and after down_proj, every layer).
//
// Three transports behind one `all_reduce_f16`:
//   * RCCL over xGMI (`ncclAllReduce`, resolved by dlopen): any message size, capturable in a hipGraph.  The default.
//   * one-shot peer reduce (hand-written): every rank copies its partial into a peer-visible buffer, signals every peer with
//     a system-scope flag, waits for all flags and sums the `world` partials in RANK ORDER with fp32 accumulation (every rank
//     computes the same bits).  Decode-sized messages only (≤ the comm buffer; T ≤ 64 rows × H ≤ 8192 = 1 MiB): at 512 KB a
//     ring all-reduce is 2·(world − 1) latency-bound xGMI hops, the one-shot form is one.  Peers are either comm buffers of
//     the SAME process (a local group: the ranks of a test on one GPU, or of one process driving several GPUs) or
//     hipIpc-imported buffers of other processes.  Pure kernels: capturable in a hipGraph, epoch kept on the device.
//   * the host-barrier loopback of runner.hip (tests, eager only).
// Every spin in the one-shot kernel is bounded (2 s of the 100 MHz realtime counter): a missing peer makes the call fail
// through the error counter instead of hanging the GPU.
#include <dlfcn.h>

#include <mutex>
#include <vector>

#include "../../include/ferrum_hip.h"
#include "common.h"
#include "knobs.h"
#include "tp_comm.h"

using namespace fh;

namespace {

// ── RCCL entry points (dlopen) ──────────────────────────────────────────────
struct UidBlob { char b[128]; };   // ncclUniqueId is passed by value (128 bytes)
typedef int (*nccl_get_uid_t)(void*);
typedef int (*nccl_comm_init_rank_t)(void**, int, UidBlob, int);
typedef int (*nccl_all_reduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_all_gather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*nccl_broadcast_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_comm_destroy_t)(void*);
void* g_rccl = nullptr;
nccl_get_uid_t g_get_uid = nullptr;
nccl_comm_init_rank_t g_init_rank = nullptr;
nccl_all_reduce_t g_all_reduce = nullptr;
nccl_all_gather_t g_all_gather = nullptr;
nccl_broadcast_t g_broadcast = nullptr;
nccl_comm_destroy_t g_comm_destroy = nullptr;
std::mutex g_rccl_mu;

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl) return 0;
    // an RCCL the process has ALREADY loaded (e.g. the one torch.distributed brought) is preferred over a second copy
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    FH_REQUIRE(h, "tensor parallel: cannot dlopen librccl.so: %s", dlerror());
    g_get_uid = (nccl_get_uid_t)dlsym(h, "ncclGetUniqueId");
    g_init_rank = (nccl_comm_init_rank_t)dlsym(h, "ncclCommInitRank");
    g_all_reduce = (nccl_all_reduce_t)dlsym(h, "ncclAllReduce");
    g_all_gather = (nccl_all_gather_t)dlsym(h, "ncclAllGather");
    g_broadcast = (nccl_broadcast_t)dlsym(h, "ncclBroadcast");
    g_comm_destroy = (nccl_comm_destroy_t)dlsym(h, "ncclCommDestroy");
    FH_REQUIRE(g_get_uid && g_init_rank && g_all_reduce && g_all_gather && g_broadcast && g_comm_destroy,
               "tensor parallel: RCCL symbols missing in librccl.so");
    g_rccl = h;
    return 0;
}

// ── one-shot peer all-reduce ────────────────────────────────────────────────
// comm buffer of a rank: [flags: 2 parities × 8 ranks × 64-byte slots | data parity 0 | data parity 1]
constexpr size_t ONESHOT_FLAG_BYTES = 4096;
constexpr int ONESHOT_MAX_BLOCKS = 64;
constexpr unsigned long long ONESHOT_SPIN_TICKS = 200000000ull;   // 2 s of s_memrealtime (100 MHz)

struct OneShotArgs {
    const __half* in;
    __half* out;
    long count;                 // fp16 elements, multiple of 4
    uint8_t* const* peers;      // device array [world]: comm buffer base of every rank (own included)
    unsigned* state;            // private device words: [0] epoch, [1] arrive ticket, [2] done ticket, [3] timeouts
    unsigned* timeouts_host;    // pinned host word (device-visible): the same count where the host reads it after its next sync
    long parity_bytes;          // bytes of one data parity
    int world, rank;
};

__device__ __forceinline__ unsigned* oneshot_flag(uint8_t* base, int parity, int from_rank) {
    return reinterpret_cast<unsigned*>(base + ((size_t)parity * 8 + from_rank) * 64);
}

__global__ __launch_bounds__(256) void tp_oneshot_all_reduce_kernel(OneShotArgs a) {
    const int tid = threadIdx.x, nb = gridDim.x;
    // every block reads the epoch before any block can advance it (it is advanced by the LAST block to finish)
    const unsigned epoch = __hip_atomic_load(&a.state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int parity = (int)(epoch & 1u);
    const unsigned token = epoch + 1u;
    const long n8 = a.count / 4;                       // 8-byte granules
    const long per = (n8 + nb - 1) / nb, g0 = (long)blockIdx.x * per, g1 = g0 + per < n8 ? g0 + per : n8;
    // stage 1: my partial → my comm buffer, written through to memory (system scope: peers read it over xGMI)
    {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(a.in);
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.peers[a.rank] + ONESHOT_FLAG_BYTES + (size_t)parity * a.parity_bytes);
        for (long g = g0 + tid; g < g1; g += 256) __hip_atomic_store(&dst[g], src[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores …
    __syncthreads();                                   // … before the block's one lane counts the block in
    __shared__ int timed_out;
    if (tid == 0) {
        timed_out = 0;
        const unsigned t = __hip_atomic_fetch_add(&a.state[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned)nb - 1) {                   // last block of this rank: all of the partial is in memory
            __hip_atomic_store(&a.state[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // system-scope release in front of the flags (the data went out write-through, so nothing is left to write back;
            // the fence orders the flag stores behind the data for a peer GPU that polls them over xGMI)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int p = 0; p < a.world; p++)
                __hip_atomic_store(oneshot_flag(a.peers[p], parity, a.rank), token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
    // stage 2: wait for every rank's partial (bounded)
    if (tid < a.world) {
        const unsigned* f = oneshot_flag(a.peers[a.rank], parity, tid);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != token) {
            __builtin_amdgcn_s_sleep(4);
            if (__builtin_amdgcn_s_memrealtime() - t0 > ONESHOT_SPIN_TICKS) { timed_out = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");          // ONE system-scope acquire behind the relaxed polls
    }
    __syncthreads();
    // stage 3: rank-ordered fp32 sum of the `world` partials, one rounding (the same bits on every rank)
    if (!timed_out) {
        for (long g = g0 + tid; g < g1; g += 256) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int p = 0; p < a.world; p++) {
                const unsigned long long* src = reinterpret_cast<const unsigned long long*>(a.peers[p] + ONESHOT_FLAG_BYTES + (size_t)parity * a.parity_bytes);
                const unsigned long long v = __hip_atomic_load(&src[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                half4 h = __builtin_bit_cast(half4, v);
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] += (float)h[j];
            }
            half4 o;
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = (_Float16)acc[j];
            reinterpret_cast<half4*>(a.out)[g] = o;
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (timed_out) {
            __hip_atomic_fetch_add(&a.state[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.timeouts_host) __hip_atomic_fetch_add(a.timeouts_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const unsigned t = __hip_atomic_fetch_add(&a.state[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned)nb - 1) {
            __hip_atomic_store(&a.state[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.state[0], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// One-shot all-reduce FOLDED INTO ITS CONSUMER (tp_decode.rs:350-372 runs all-reduce, then the residual add + norm of the layer,
// as two launches): one 256-thread block per token row sends the row's partial, waits for the peers' flags with the norm weights
// already requested, sums the `world` partials in rank order (fp32, one rounding — the all-reduce's output bits), adds the
// residual, and normalises: residual' = residual + Σ_r x_r, out = rms_norm(residual')·w.  Same flags, parity, epoch and tickets
// as the all-reduce above; same thread ↔ element mapping and reduction order as rms_norm_kernel<true, ·> (norm.hip), so the
// result is the two launches' bit for bit.
struct OneShotNormArgs {
    OneShotArgs r;              // in = this rank's partial [rows, dim]; out unused
    __half* residual;           // [rows, dim] in / out
    const __half* w;            // [dim]
    __half* norm_out;           // [rows, dim]
    float eps;
    int dim;
};

template <int CHUNKS>
__global__ __launch_bounds__(256) void tp_oneshot_reduce_add_norm_kernel(OneShotNormArgs q) {
    const OneShotArgs& a = q.r;
    __shared__ float smem[4];
    __shared__ int timed_out;
    const int tid = threadIdx.x, nb = gridDim.x;
    const long row = blockIdx.x;
    const int dim = q.dim, nvec = dim >> 3;
    const unsigned epoch = __hip_atomic_load(&a.state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int parity = (int)(epoch & 1u);
    const unsigned token = epoch + 1u;
    // stage 1: this row of my partial → my comm buffer (write-through, system scope)
    {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(a.in + row * dim);
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.peers[a.rank] + ONESHOT_FLAG_BYTES + (size_t)parity * a.parity_bytes) + row * (dim >> 2);
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) {
            const int i = tid + c * 256;
            if (i < nvec) {
                __hip_atomic_store(&dst[2 * i], src[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&dst[2 * i + 1], src[2 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    // requested before the wait: the row's residual and norm weights
    half8 rv[CHUNKS], wv[CHUNKS];
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
        const int i = tid + c * 256;
        if (i < nvec) {
            rv[c] = *reinterpret_cast<const half8*>(q.residual + row * dim + i * 8);
            wv[c] = *reinterpret_cast<const half8*>(q.w + i * 8);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        timed_out = 0;
        const unsigned t = __hip_atomic_fetch_add(&a.state[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned)nb - 1) {
            __hip_atomic_store(&a.state[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int p = 0; p < a.world; p++)
                __hip_atomic_store(oneshot_flag(a.peers[p], parity, a.rank), token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
    if (tid < a.world) {
        const unsigned* f = oneshot_flag(a.peers[a.rank], parity, tid);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != token) {
            __builtin_amdgcn_s_sleep(4);
            if (__builtin_amdgcn_s_memrealtime() - t0 > ONESHOT_SPIN_TICKS) { timed_out = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
    if (!timed_out) {
        float ss = 0.f;
        half8 v[CHUNKS];
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) {
            const int i = tid + c * 256;
            if (i < nvec) {
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int p = 0; p < a.world; p++) {
                    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(a.peers[p] + ONESHOT_FLAG_BYTES + (size_t)parity * a.parity_bytes) + row * (dim >> 2);
                    const unsigned long long lo = __hip_atomic_load(&src[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const unsigned long long hi = __hip_atomic_load(&src[2 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const half4 hl = __builtin_bit_cast(half4, lo), hh = __builtin_bit_cast(half4, hi);
#pragma unroll
                    for (int j = 0; j < 4; j++) { acc[j] += (float)hl[j]; acc[4 + j] += (float)hh[j]; }
                }
                half8 r = rv[c];
#pragma unroll
                for (int j = 0; j < 8; j++) r[j] = (_Float16)((float)r[j] + (float)(_Float16)acc[j]);     // (the all-reduce's fp16 output, then the add)
                *reinterpret_cast<half8*>(q.residual + row * dim + i * 8) = r;
                v[c] = r;
#pragma unroll
                for (int j = 0; j < 8; j++) ss += (float)r[j] * (float)r[j];
            }
        }
        // block_reduce_sum_256 of norm.hip: wave sums, then the four partials in wave order
        ss = wave_reduce_sum(ss);
        if ((tid & 63) == 0) smem[tid >> 6] = ss;
        __syncthreads();
        const float total = smem[0] + smem[1] + smem[2] + smem[3];
        const float inv = 1.0f / sqrtf(total / (float)dim + q.eps);
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) {
            const int i = tid + c * 256;
            if (i < nvec) {
                half8 o;
#pragma unroll
                for (int j = 0; j < 8; j++) o[j] = (_Float16)((float)v[c][j] * inv * (float)wv[c][j]);
                *reinterpret_cast<half8*>(q.norm_out + row * dim + i * 8) = o;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (timed_out) {
            __hip_atomic_fetch_add(&a.state[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.timeouts_host) __hip_atomic_fetch_add(a.timeouts_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const unsigned t = __hip_atomic_fetch_add(&a.state[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned)nb - 1) {
            __hip_atomic_store(&a.state[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.state[0], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// One-shot all-gather of a small record per rank (≤ 64 KB; the per-row argmax pairs of a vocabulary-parallel lm_head): the
// same buffers, flags, parity and epoch as the all-reduce above (one block, so it is its own last arriver).
__global__ __launch_bounds__(256) void tp_oneshot_all_gather_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, long bytes,
                                                                    uint8_t* const* peers, unsigned* state, long parity_bytes, int world,
                                                                    int rank, unsigned* timeouts_host) {
    const int tid = threadIdx.x;
    const unsigned epoch = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int parity = (int)(epoch & 1u);
    const unsigned token = epoch + 1u;
    const long n8 = bytes / 8;
    {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(in);
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(peers[rank] + ONESHOT_FLAG_BYTES + (size_t)parity * parity_bytes);
        for (long g = tid; g < n8; g += 256) __hip_atomic_store(&dst[g], src[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int timed_out;
    if (tid == 0) timed_out = 0;
    if (tid < world) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(oneshot_flag(peers[tid], parity, rank), token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (tid < world) {
        const unsigned* f = oneshot_flag(peers[rank], parity, tid);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != token) {
            __builtin_amdgcn_s_sleep(4);
            if (__builtin_amdgcn_s_memrealtime() - t0 > ONESHOT_SPIN_TICKS) { timed_out = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
    if (!timed_out)
        for (int p = 0; p < world; p++) {
            const unsigned long long* src = reinterpret_cast<const unsigned long long*>(peers[p] + ONESHOT_FLAG_BYTES + (size_t)parity * parity_bytes);
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(out + (size_t)p * bytes);
            for (long g = tid; g < n8; g += 256) dst[g] = __hip_atomic_load(&src[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    __syncthreads();
    if (tid == 0) {
        if (timed_out) {
            __hip_atomic_fetch_add(&state[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (timeouts_host) __hip_atomic_fetch_add(timeouts_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __hip_atomic_store(&state[0], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

struct FerrumHipComm {
    int world = 1, rank = 0;
    void* nccl = nullptr;                 // RCCL communicator (null: none)
    // one-shot
    uint8_t* buf = nullptr;               // own comm buffer (owned)
    size_t parity_bytes = 0;
    uint8_t* peer_ptrs[8] = {};           // every rank's comm buffer (own included); imported ones are closed at destroy
    bool peer_imported[8] = {};
    uint8_t** peers_dev = nullptr;        // device copy of peer_ptrs
    unsigned* state = nullptr;            // device: epoch, tickets, timeouts
    unsigned* timeouts_host = nullptr;    // pinned host word bumped by a one-shot wait that gave up (read by the runner after every host sync)
    unsigned timeouts_seen = 0;
    bool oneshot_failed = false;          // a one-shot call timed out: its output was the rank's un-reduced partial and the epochs may have diverged — the transport stays off
    bool finegrained = false;             // the comm buffer is fine-grained device memory (hipExtMallocWithFlags)
    bool oneshot_ready = false;
};

namespace fh {

int comm_world(const FerrumHipComm* c) { return c ? c->world : 1; }
bool comm_graph_safe(const FerrumHipComm* c) { return c != nullptr; }   // RCCL and one-shot are both stream-ordered device work

bool comm_oneshot_fits(const FerrumHipComm* c, size_t count) {
    if (!c || !c->oneshot_ready || c->oneshot_failed || knobs().tp_oneshot == 0) return false;
    if (count % 4 != 0 || count * 2 > c->parity_bytes) return false;
    return knobs().tp_oneshot == 1 || c->nccl == nullptr;      // auto: RCCL when there is one, one-shot for RCCL-less groups
}

int comm_all_reduce_f16(FerrumHipComm* c, __half* buf, size_t count, hipStream_t s) {
    if (!c || c->world <= 1 || count == 0) return 0;
    if (comm_oneshot_fits(c, count)) {
        OneShotArgs a{buf, buf, (long)count, c->peers_dev, c->state, c->timeouts_host, (long)c->parity_bytes, c->world, c->rank};
        const int blocks = std::max(1, std::min(ONESHOT_MAX_BLOCKS, cdiv((long)count / 4, 256)));
        hipLaunchKernelGGL(tp_oneshot_all_reduce_kernel, dim3(blocks), dim3(256), 0, s, a);
        FH_CHECK_LAUNCH();
        form_hit(FORM_TP_ALLREDUCE_ONESHOT);
        return 0;
    }
    FH_REQUIRE(c->nccl && g_all_reduce, "all_reduce: %zu fp16 elements: the one-shot transport is unavailable (message too large, or switched off after a call that gave up waiting for a peer) and the communicator has no RCCL rank", count);
    // ncclFloat16 = 6, ncclSum = 0 (rccl.h); in place like nccl_comm.rs all_reduce_in_place
    const int rc = g_all_reduce(buf, buf, count, 6, 0, c->nccl, s);
    FH_REQUIRE(rc == 0, "ncclAllReduce failed: %d", rc);
    form_hit(FORM_TP_ALLREDUCE_RCCL);
    return 0;
}

// residual += Σ_ranks x; norm_out = rms_norm(residual)·w — the all-reduce and its consumer as ONE launch where the one-shot
// transport carries the message (rows ≤ 64, dim ≤ 8192); *fused = 0 otherwise (the caller then runs all-reduce + add + norm).
int comm_all_reduce_add_rms_norm_f16(FerrumHipComm* c, const __half* x, __half* residual, const __half* w, float eps, __half* norm_out,
                                     int rows, int dim, int* fused, hipStream_t s) {
    *fused = 0;
    if (!c || c->world <= 1 || rows <= 0) return 0;
    if (!knobs().tp_fused_norm || rows > ONESHOT_MAX_BLOCKS || dim % 8 != 0 || dim > 8 * 256 * 4 || !comm_oneshot_fits(c, (size_t)rows * dim)) return 0;
    OneShotNormArgs q{};
    q.r = OneShotArgs{x, nullptr, (long)rows * dim, c->peers_dev, c->state, c->timeouts_host, (long)c->parity_bytes, c->world, c->rank};
    q.residual = residual; q.w = w; q.norm_out = norm_out; q.eps = eps; q.dim = dim;
    const int chunks = cdiv(dim / 8, 256);
    if (chunks <= 1) hipLaunchKernelGGL((tp_oneshot_reduce_add_norm_kernel<1>), dim3(rows), dim3(256), 0, s, q);
    else if (chunks <= 2) hipLaunchKernelGGL((tp_oneshot_reduce_add_norm_kernel<2>), dim3(rows), dim3(256), 0, s, q);
    else hipLaunchKernelGGL((tp_oneshot_reduce_add_norm_kernel<4>), dim3(rows), dim3(256), 0, s, q);
    FH_CHECK_LAUNCH();
    form_hit(FORM_TP_ALLREDUCE_NORM_FUSED);
    *fused = 1;
    return 0;
}

// All-gather of `bytes` (a multiple of 8, small) per rank into out[world][bytes], rank order.
int comm_all_gather_bytes(FerrumHipComm* c, const void* in, void* out, size_t bytes, hipStream_t s) {
    if (!c || c->world <= 1) {
        if (in != out && bytes) FH_CHECK_HIP(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    FH_REQUIRE(bytes % 8 == 0, "all_gather: %zu bytes per rank must be a multiple of 8", bytes);
    const bool fits = c->oneshot_ready && !c->oneshot_failed && knobs().tp_oneshot != 0 && bytes <= c->parity_bytes && bytes <= (64u << 10);
    if (fits && (knobs().tp_oneshot == 1 || c->nccl == nullptr)) {
        hipLaunchKernelGGL(tp_oneshot_all_gather_kernel, dim3(1), dim3(256), 0, s, (const uint8_t*)in, (uint8_t*)out, (long)bytes, c->peers_dev,
                           c->state, (long)c->parity_bytes, c->world, c->rank, c->timeouts_host);
        FH_CHECK_LAUNCH();
        form_hit(FORM_TP_ALLREDUCE_ONESHOT);
        return 0;
    }
    FH_REQUIRE(c->nccl && g_all_gather, "all_gather: the communicator has neither a fitting one-shot buffer nor an RCCL rank");
    const int rc = g_all_gather(in, out, bytes, 0 /* ncclInt8 */, c->nccl, s);
    FH_REQUIRE(rc == 0, "ncclAllGather failed: %d", rc);
    form_hit(FORM_TP_ALLREDUCE_RCCL);
    return 0;
}

// Called by the runner after every host synchronisation of a forward: a one-shot wait that gave up skipped its reduction
// (the output kept the rank's partial).  Returns the number of new give-ups and turns the transport off for good.
unsigned comm_take_timeouts(FerrumHipComm* c) {
    if (!c || !c->timeouts_host) return 0;
    const unsigned now = *reinterpret_cast<volatile unsigned*>(c->timeouts_host);
    const unsigned n = now - c->timeouts_seen;
    c->timeouts_seen = now;
    if (n) c->oneshot_failed = true;
    return n;
}

}  // namespace fh

namespace {

int oneshot_alloc(FerrumHipComm* c, size_t max_bytes) {
    FH_REQUIRE(max_bytes >= 4096 && max_bytes <= ((size_t)64 << 20), "comm: one-shot message cap %zu out of range", max_bytes);
    c->parity_bytes = (max_bytes + 255) / 256 * 256;
    const size_t total = ONESHOT_FLAG_BYTES + 2 * c->parity_bytes;
    // The flags of this buffer are polled by PEER GPUs inside a running kernel and the data parities are read by them while
    // this rank's kernel is still in flight: plain hipMalloc memory is coarse-grained (cross-device visibility only at kernel
    // boundaries), so the buffer is fine-grained device memory like RCCL's own flag / LL buffers; coarse-grained only as a
    // fallback (single-GPU rehearsals work either way, a multi-GPU group then relies on the write-through system-scope stores).
    if (hipExtMallocWithFlags((void**)&c->buf, total, hipDeviceMallocFinegrained) == hipSuccess) {
        c->finegrained = true;
    } else {
        (void)hipGetLastError();
        c->buf = nullptr;
        FH_CHECK_HIP(hipMalloc((void**)&c->buf, total));
    }
    FH_CHECK_HIP(hipMemset(c->buf, 0, total));
    FH_CHECK_HIP(hipHostMalloc((void**)&c->timeouts_host, 64, hipHostMallocDefault));
    *c->timeouts_host = 0u;
    FH_CHECK_HIP(hipMalloc((void**)&c->state, 64));
    FH_CHECK_HIP(hipMemset(c->state, 0, 64));
    FH_CHECK_HIP(hipMalloc((void**)&c->peers_dev, sizeof(void*) * 8));
    FH_CHECK_HIP(hipDeviceSynchronize());
    return 0;
}

int oneshot_publish_peers(FerrumHipComm* c) {
    FH_CHECK_HIP(hipMemcpy(c->peers_dev, c->peer_ptrs, sizeof(void*) * 8, hipMemcpyHostToDevice));
    c->oneshot_ready = true;
    return 0;
}

}  // namespace

extern "C" {

int ferrum_hip_comm_unique_id(uint8_t id[128]) {
    FH_REQUIRE(id, "comm_unique_id: null");
    if (int rc = load_rccl()) return rc;
    const int rc = g_get_uid(id);
    FH_REQUIRE(rc == 0, "ncclGetUniqueId failed: %d", rc);
    return 0;
}
int ferrum_hip_tp_unique_id(uint8_t id[128]) { return ferrum_hip_comm_unique_id(id); }

int ferrum_hip_comm_create_rccl(FerrumHipComm** out, int world, int rank, const uint8_t id[128]) {
    FH_REQUIRE(out && id && world >= 1 && world <= 8 && rank >= 0 && rank < world, "comm_create_rccl: world=%d rank=%d", world, rank);
    if (int rc = load_rccl()) return rc;
    auto* c = new FerrumHipComm();
    c->world = world; c->rank = rank;
    UidBlob blob;
    memcpy(blob.b, id, 128);
    const int rc = g_init_rank(&c->nccl, world, blob, rank);
    if (rc != 0 || !c->nccl) { delete c; fh::set_error("ncclCommInitRank failed: %d", rc); return 1; }
    *out = c;
    return 0;
}

// A rank without a transport yet: the one-shot buffers are attached afterwards (oneshot_export / oneshot_attach).  What a
// multi-process group uses when it has no RCCL rank — e.g. the ranks of a rehearsal that share ONE GPU (RCCL refuses
// two ranks on a device), or a deployment that wants only the one-shot path.
int ferrum_hip_comm_create_bare(FerrumHipComm** out, int world, int rank) {
    FH_REQUIRE(out && world >= 1 && world <= 8 && rank >= 0 && rank < world, "comm_create_bare: world=%d rank=%d", world, rank);
    auto* c = new FerrumHipComm();
    c->world = world; c->rank = rank;
    *out = c;
    return 0;
}

// The ranks of one process (threads of a test on one GPU, or one process driving several GPUs with peer access enabled):
// `world` communicators whose one-shot buffers see each other directly.  Call with the device of rank r current when the
// ranks live on different devices (devices[] = NULL: everything on the current device).
int ferrum_hip_comm_create_local_group(FerrumHipComm** out, int world, size_t max_message_bytes, const int* devices) {
    FH_REQUIRE(out && world >= 2 && world <= 8, "comm_create_local_group: world=%d", world);
    int cur = 0;
    FH_CHECK_HIP(hipGetDevice(&cur));
    std::vector<FerrumHipComm*> cs(world, nullptr);
    int rc = 0;
    for (int r = 0; r < world && !rc; r++) {
        cs[r] = new FerrumHipComm();
        cs[r]->world = world; cs[r]->rank = r;
        if (devices) rc = hipSetDevice(devices[r]) == hipSuccess ? 0 : 1;
        if (!rc) rc = oneshot_alloc(cs[r], max_message_bytes);
    }
    for (int r = 0; r < world && !rc; r++) {
        for (int p = 0; p < world; p++) cs[r]->peer_ptrs[p] = cs[p]->buf;
        if (devices) rc = hipSetDevice(devices[r]) == hipSuccess ? 0 : 1;
        if (!rc) rc = oneshot_publish_peers(cs[r]);
    }
    (void)hipSetDevice(cur);
    if (rc) { for (auto* c : cs) ferrum_hip_comm_destroy(c); return rc; }
    for (int r = 0; r < world; r++) out[r] = cs[r];
    return 0;
}

// Multi-process one-shot: every rank allocates its buffer and exports a 64-byte hipIpc handle; the host exchanges the
// handles (torch.distributed / the engine's control plane) and every rank attaches the others'.
int ferrum_hip_comm_oneshot_export(FerrumHipComm* c, size_t max_message_bytes, uint8_t handle[64]) {
    FH_REQUIRE(c && handle, "comm_oneshot_export: null");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (!c->buf)
        if (int rc = oneshot_alloc(c, max_message_bytes)) return rc;
    hipIpcMemHandle_t h;
    FH_CHECK_HIP(hipIpcGetMemHandle(&h, c->buf));
    memcpy(handle, &h, 64);
    return 0;
}
int ferrum_hip_comm_oneshot_attach(FerrumHipComm* c, const uint8_t* handles, int world) {
    FH_REQUIRE(c && handles && world == c->world && c->buf, "comm_oneshot_attach: export first; world=%d", world);
    for (int p = 0; p < world; p++) {
        if (p == c->rank) { c->peer_ptrs[p] = c->buf; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)p * 64, 64);
        void* ptr = nullptr;
        FH_CHECK_HIP(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
        c->peer_ptrs[p] = (uint8_t*)ptr;
        c->peer_imported[p] = true;
    }
    return oneshot_publish_peers(c);
}

int ferrum_hip_comm_destroy(FerrumHipComm* c) {
    if (!c) return 0;
    if (c->nccl && g_comm_destroy) (void)g_comm_destroy(c->nccl);
    for (int p = 0; p < 8; p++)
        if (c->peer_imported[p] && c->peer_ptrs[p]) (void)hipIpcCloseMemHandle(c->peer_ptrs[p]);
    if (c->buf) (void)hipFree(c->buf);
    if (c->state) (void)hipFree(c->state);
    if (c->timeouts_host) (void)hipHostFree(c->timeouts_host);
    if (c->peers_dev) (void)hipFree(c->peers_dev);
    delete c;
    return 0;
}

int ferrum_hip_comm_world_size(const FerrumHipComm* c) { return c ? c->world : 1; }
int ferrum_hip_comm_rank(const FerrumHipComm* c) { return c ? c->rank : 0; }

// BackendCollective::all_reduce (capabilities.rs:92), ReduceOp::Sum over fp16, in place, on `stream`.
int ferrum_hip_all_reduce_f16(FerrumHipComm* c, void* buf, size_t count, void* stream) {
    FH_REQUIRE(buf || count == 0, "all_reduce: null buffer");
    return comm_all_reduce_f16(c, (__half*)buf, count, as_stream(stream));
}
int ferrum_hip_all_reduce_add_rms_norm_f16(FerrumHipComm* c, const void* x, void* residual, const void* w, float eps, void* norm_out,
                                           int rows, int dim, int* fused, void* stream) {
    FH_REQUIRE(fused && (rows == 0 || (x && residual && w && norm_out)), "all_reduce_add_rms_norm: null argument");
    return comm_all_reduce_add_rms_norm_f16(c, reinterpret_cast<const __half*>(x), reinterpret_cast<__half*>(residual),
                                            reinterpret_cast<const __half*>(w), eps, reinterpret_cast<__half*>(norm_out), rows, dim, fused,
                                            as_stream(stream));
}

// BackendCollective::all_gather / broadcast (capabilities.rs:95-108): RCCL ranks only.
int ferrum_hip_all_gather_f16(FerrumHipComm* c, const void* local, void* global, size_t local_count, void* stream) {
    if (!c || c->world <= 1) {
        if (local != global && local_count) FH_CHECK_HIP(hipMemcpyAsync(global, local, local_count * 2, hipMemcpyDeviceToDevice, as_stream(stream)));
        return 0;
    }
    if (!c->nccl) { fh::set_error("all_gather: communicator has no RCCL rank"); return FERRUM_HIP_UNSUPPORTED; }
    const int rc = g_all_gather(local, global, local_count, 6, c->nccl, as_stream(stream));
    FH_REQUIRE(rc == 0, "ncclAllGather failed: %d", rc);
    return 0;
}
int ferrum_hip_broadcast_f16(FerrumHipComm* c, void* buf, size_t count, int src_rank, void* stream) {
    if (!c || c->world <= 1) return 0;
    if (!c->nccl) { fh::set_error("broadcast: communicator has no RCCL rank"); return FERRUM_HIP_UNSUPPORTED; }
    const int rc = g_broadcast(buf, buf, count, 6, src_rank, c->nccl, as_stream(stream));
    FH_REQUIRE(rc == 0, "ncclBroadcast failed: %d", rc);
    return 0;
}

// Number of one-shot calls on this rank whose wait for a peer timed out (0 in a healthy run); also the epoch.
int ferrum_hip_comm_oneshot_status(FerrumHipComm* c, unsigned* epoch, unsigned* timeouts) {
    FH_REQUIRE(c, "comm_oneshot_status: null");
    unsigned st[4] = {0, 0, 0, 0};
    if (c->state) FH_CHECK_HIP(hipMemcpy(st, c->state, sizeof(st), hipMemcpyDeviceToHost));
    if (epoch) *epoch = st[0];
    if (timeouts) *timeouts = st[3];
    if (st[3]) c->oneshot_failed = true;      // a call that gave up left un-reduced data behind and the ranks' epochs may differ: the transport stays off
    return 0;
}

// Plumbing self-test against the installed librccl on ONE device: a 1-rank communicator all-reduces (fp16, sum, in place) a
// known vector — once eagerly and once from inside a captured hipGraph that is then replayed twice — and the vector must
// come back unchanged.  Exercises exactly the entry points, enum values, by-value ncclUniqueId passing and the stream
// capture the tensor-parallel decode loop relies on (which needs ≥ 2 GPUs to run for real).
int ferrum_hip_tp_selftest(int count) {
    FH_REQUIRE(count > 0 && count <= (1 << 20), "tp_selftest: count=%d", count);
    uint8_t id[128];
    if (int rc = ferrum_hip_comm_unique_id(id)) return rc;
    FerrumHipComm* c = nullptr;
    if (int rc = ferrum_hip_comm_create_rccl(&c, 1, 0, id)) return rc;
    c->world = 2;                                     // force the RCCL call (a 1-rank comm would otherwise short-cut)
    std::vector<__half> host(count), back(count);
    for (int i = 0; i < count; i++) host[i] = __float2half((float)(i % 257) * 0.25f - 16.0f);
    __half* dev = nullptr;
    hipStream_t s = nullptr;
    FH_CHECK_HIP(hipMalloc((void**)&dev, (size_t)count * 2));
    FH_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    FH_CHECK_HIP(hipMemcpyAsync(dev, host.data(), (size_t)count * 2, hipMemcpyHostToDevice, s));
    int rc = comm_all_reduce_f16(c, dev, (size_t)count, s);
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    int graph_rc = 0;
    if (!rc) {
        FH_CHECK_HIP(hipStreamSynchronize(s));
        FH_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        graph_rc = comm_all_reduce_f16(c, dev, (size_t)count, s);
        hipError_t e = hipStreamEndCapture(s, &g);
        if (!graph_rc && e != hipSuccess) { fh::set_error("tp_selftest: capture of ncclAllReduce failed: %s", hipGetErrorString(e)); graph_rc = 1; }
        if (!graph_rc && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { fh::set_error("tp_selftest: graph instantiate failed"); graph_rc = 1; }
        for (int i = 0; i < 2 && !graph_rc; i++)
            if (hipGraphLaunch(ge, s) != hipSuccess) { fh::set_error("tp_selftest: graph launch failed"); graph_rc = 1; }
    }
    // the small all-gather of the vocabulary-parallel sampler through the same communicator (ncclAllGather, bytes as int8)
    int gather_rc = 0;
    if (!rc && !graph_rc) {
        std::vector<uint8_t> gin(256), gout(256, 0);
        for (int i = 0; i < 256; i++) gin[i] = (uint8_t)(i * 7 + 3);
        uint8_t *gi = nullptr, *go = nullptr;
        FH_CHECK_HIP(hipMalloc((void**)&gi, 256));
        FH_CHECK_HIP(hipMalloc((void**)&go, 512));
        FH_CHECK_HIP(hipMemcpyAsync(gi, gin.data(), 256, hipMemcpyHostToDevice, s));
        gather_rc = comm_all_gather_bytes(c, gi, go, 256, s);      // 1-rank communicator: the rank's chunk lands at offset 0
        (void)hipMemcpyAsync(gout.data(), go, 256, hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        (void)hipFree(gi); (void)hipFree(go);
        if (!gather_rc && memcmp(gin.data(), gout.data(), 256) != 0) { fh::set_error("tp_selftest: 1-rank all-gather changed the data"); gather_rc = 1; }
    }
    (void)hipMemcpyAsync(back.data(), dev, (size_t)count * 2, hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    if (ge) (void)hipGraphExecDestroy(ge);
    if (g) (void)hipGraphDestroy(g);
    c->world = 1;
    ferrum_hip_comm_destroy(c);
    (void)hipStreamDestroy(s);
    (void)hipFree(dev);
    if (rc) return rc;
    if (graph_rc) return graph_rc;
    if (gather_rc) return gather_rc;
    FH_REQUIRE(memcmp(host.data(), back.data(), (size_t)count * 2) == 0, "tp_selftest: 1-rank all-reduce changed the data");
    return 0;
}

}  // extern "C"
