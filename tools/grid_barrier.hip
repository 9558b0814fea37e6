// Development micro-benchmark: what does a grid-wide barrier cost on MI355X (8 XCDs, one L2 each) next to the ≈ 4.7 µs a
// dependent kernel boundary costs inside a hipGraph?  A persistent per-layer decode kernel replaces launches by barriers,
// so this number decides whether one is worth building (rejected decode forms).
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
// Every spin is bounded (a barrier that never completes sets `err` and falls through).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int MAX_POLLS = 1 << 21;

// MODE 0: relaxed atomics only (no fences: the lower bound, carries no data)
// MODE 1: release add / acquire load at agent scope (what a data-carrying barrier needs)
// MODE 2: as 1, hierarchical: WGs of an XCD meet on the XCD's counter, the last arriver of each XCD meets the other 7
// MODE 3 / 4 (round 3, the forms of MI355X_MICROARCH.md's price table): the data goes out with write-through (sc1) stores that every
//   wave drains before the workgroup's one RELAXED arrival; the poll is a relaxed sc1 load with s_sleep, ONE wavefront-scope acquire
//   fence behind it, and the data is read back with sc1 loads.  3 flat, 4 XCD-hierarchical (per-XCD counter, then 8 arrivals).
template <int MODE>
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned* xcd_ctr, unsigned nwg, unsigned round, unsigned* err) {
    if (MODE >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE >= 3) {
            unsigned target;
            if (MODE == 4) {
                const unsigned xcd = blockIdx.x & 7, per = nwg >> 3;
                const unsigned old = __hip_atomic_fetch_add(&xcd_ctr[xcd * 64], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == round * per + per - 1) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                target = (round + 1) * 8;
            } else {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                target = (round + 1) * nwg;
            }
            int polls = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++polls > MAX_POLLS) { *err = 1; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else if (MODE == 2) {
            const unsigned xcd = blockIdx.x & 7, per = nwg >> 3;
            const unsigned old = __hip_atomic_fetch_add(&xcd_ctr[xcd * 32], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (old == round * per + per - 1) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            int polls = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (round + 1) * 8) {
                __builtin_amdgcn_s_sleep(1);
                if (++polls > MAX_POLLS) { *err = 1; break; }
            }
        } else {
            const auto order_add = MODE == 0 ? __ATOMIC_RELAXED : __ATOMIC_RELEASE;
            const auto order_ld = MODE == 0 ? __ATOMIC_RELAXED : __ATOMIC_ACQUIRE;
            __hip_atomic_fetch_add(ctr, 1u, order_add, __HIP_MEMORY_SCOPE_AGENT);
            int polls = 0;
            while (__hip_atomic_load(ctr, order_ld, __HIP_MEMORY_SCOPE_AGENT) < (round + 1) * nwg) {
                __builtin_amdgcn_s_sleep(1);
                if (++polls > MAX_POLLS) { *err = 1; break; }
            }
        }
    }
    __syncthreads();
}

// each round: every workgroup writes 1 KiB (value = round + wg), barrier, reads the 1 KiB of workgroup (wg + 37) % nwg and checks it
template <int MODE, bool DATA>
__global__ __launch_bounds__(256) void barrier_loop(unsigned* ctr, unsigned* xcd_ctr, unsigned* buf, int rounds, unsigned* err, unsigned* bad) {
    const unsigned nwg = gridDim.x, wg = blockIdx.x;
    for (int r = 0; r < rounds; r++) {
        if (DATA) {
            unsigned* dst = &buf[(size_t)(r & 1) * nwg * 256 + wg * 256 + threadIdx.x];
            if (MODE >= 3) __hip_atomic_store(dst, (unsigned)r * 1000003u + wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sc1 write-through
            else *dst = (unsigned)r * 1000003u + wg;
        }
        grid_barrier<MODE>(ctr, xcd_ctr, nwg, (unsigned)r, err);
        if (DATA) {
            const unsigned src = (wg + 37) % nwg;
            const unsigned* sp = &buf[(size_t)(r & 1) * nwg * 256 + src * 256 + threadIdx.x];
            const unsigned v = MODE >= 3 ? __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __builtin_nontemporal_load(sp);
            if (v != (unsigned)r * 1000003u + src) atomicAdd(bad, 1u);
        }
    }
}

template <int MODE, bool DATA>
static void run(const char* name, int nwg, unsigned* ctr, unsigned* xcd, unsigned* buf, unsigned* err, unsigned* bad) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float t[2];
    const int rounds[2] = {8, 408};
    for (int k = 0; k < 2; k++) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipMemset(ctr, 0, 4)); CK(hipMemset(xcd, 0, 8 * 64 * 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(bad, 0, 4));
            CK(hipDeviceSynchronize());
            hipEventRecord(a);
            hipLaunchKernelGGL((barrier_loop<MODE, DATA>), dim3(nwg), dim3(256), 0, 0, ctr, xcd, buf, rounds[k], err, bad);
            hipEventRecord(b); CK(hipEventSynchronize(b));
            float ms; hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        t[k] = best;
    }
    unsigned he = 0, hb = 0;
    CK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("%-44s wgs=%4d  %.2f us per barrier  (timeouts=%u stale_reads=%u)\n", name, nwg, (t[1] - t[0]) * 1000.f / 400.f, he, hb);
}

// producer → consumer hand-off (the form every merged launch of csrc/ uses): workgroup i waits for workgroup i − 1's counter, reads
// its 1 KiB with sc1 loads, writes its own 1 KiB (sc1), drains, bumps its counter.  Time per link = one hand-off on a critical path.
__global__ __launch_bounds__(256) void handoff_chain(unsigned* flags, unsigned* buf, unsigned* err, unsigned* bad, int rounds) {
    const unsigned nwg = gridDim.x, wg = blockIdx.x;
    for (int r = 0; r < rounds; r++) {
        unsigned v = 0;
        if (wg > 0 || r > 0) {
            // predecessor in the ring: wg − 1 of this round, or the last workgroup of the previous round for wg 0
            const unsigned pred = wg > 0 ? wg - 1 : nwg - 1;
            const unsigned need = wg > 0 ? (unsigned)r + 1 : (unsigned)r;
            if (threadIdx.x == 0) {
                int polls = 0;
                while (__hip_atomic_load(&flags[pred * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++polls > MAX_POLLS) { *err = 1; break; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            __syncthreads();
            v = __hip_atomic_load(&buf[pred * 256 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (wg > 0 ? (unsigned)r : (unsigned)r - 1) * 1000003u + pred;
            if (v != want) atomicAdd(bad, 1u);
        }
        __hip_atomic_store(&buf[wg * 256 + threadIdx.x], (unsigned)r * 1000003u + wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&flags[wg * 64], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static void run_handoff(int nwg, unsigned* flags, unsigned* buf, unsigned* err, unsigned* bad) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float t[2];
    const int rounds[2] = {2, 22};
    for (int k = 0; k < 2; k++) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipMemset(flags, 0, (size_t)nwg * 64 * 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(bad, 0, 4));
            CK(hipDeviceSynchronize());
            hipEventRecord(a);
            hipLaunchKernelGGL(handoff_chain, dim3(nwg), dim3(256), 0, 0, flags, buf, err, bad, rounds[k]);
            hipEventRecord(b); CK(hipEventSynchronize(b));
            float ms; hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        t[k] = best;
    }
    unsigned he = 0, hb = 0;
    CK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("%-44s wgs=%4d  %.2f us per hand-off   (timeouts=%u stale_reads=%u)\n", "hand-off flag, 1 KiB (sc1 store/drain/flag/poll/sc1 load)", nwg,
           (t[1] - t[0]) * 1000.f / (20.f * nwg), he, hb);
}

int main() {
    unsigned *ctr, *xcd, *buf, *err, *bad;
    CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&xcd, 8 * 64 * 4)); CK(hipMalloc(&buf, 2 * 2048 * 256 * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&bad, 4));
    for (int nwg : {64, 256, 512}) {
        run<0, false>("relaxed atomics, no data", nwg, ctr, xcd, buf, err, bad);
        run<1, false>("release/acquire, no data", nwg, ctr, xcd, buf, err, bad);
        run<1, true>("release/acquire, 1 KiB per WG exchanged", nwg, ctr, xcd, buf, err, bad);
        run<2, true>("hierarchical (per-XCD, then 8), 1 KiB", nwg, ctr, xcd, buf, err, bad);
        run<3, true>("sc1 data, relaxed polls + 1 fence, flat", nwg, ctr, xcd, buf, err, bad);
        run<4, true>("sc1 data, relaxed polls + 1 fence, per-XCD", nwg, ctr, xcd, buf, err, bad);
    }
    unsigned* flags;
    CK(hipMalloc(&flags, 256 * 64 * 4));
    for (int nwg : {8, 64, 256}) run_handoff(nwg, flags, buf, err, bad);
    return 0;
}
