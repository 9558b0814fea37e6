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
template <int MODE>
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned* xcd_ctr, unsigned nwg, unsigned round, unsigned* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 2) {
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
        if (DATA) buf[(size_t)(r & 1) * nwg * 256 + wg * 256 + threadIdx.x] = (unsigned)r * 1000003u + wg;
        grid_barrier<MODE>(ctr, xcd_ctr, nwg, (unsigned)r, err);
        if (DATA) {
            const unsigned src = (wg + 37) % nwg;
            const unsigned v = __builtin_nontemporal_load(&buf[(size_t)(r & 1) * nwg * 256 + src * 256 + threadIdx.x]);
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
            CK(hipMemset(ctr, 0, 4)); CK(hipMemset(xcd, 0, 8 * 32 * 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(bad, 0, 4));
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

int main() {
    unsigned *ctr, *xcd, *buf, *err, *bad;
    CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&xcd, 8 * 32 * 4)); CK(hipMalloc(&buf, 2 * 2048 * 256 * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&bad, 4));
    for (int nwg : {64, 256, 512}) {
        run<0, false>("relaxed atomics, no data", nwg, ctr, xcd, buf, err, bad);
        run<1, false>("release/acquire, no data", nwg, ctr, xcd, buf, err, bad);
        run<1, true>("release/acquire, 1 KiB per WG exchanged", nwg, ctr, xcd, buf, err, bad);
        run<2, true>("hierarchical (per-XCD, then 8), 1 KiB", nwg, ctr, xcd, buf, err, bad);
    }
    return 0;
}
