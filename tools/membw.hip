// Development micro-benchmark: HBM read ceilings for the access patterns of the INT4 grouped GEMM.
// hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o /tmp/membw && /tmp/membw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// each wave streams `bytes_per_wave` contiguous bytes, U loads (1 KiB each) in flight
template <int U, bool NT>
__global__ void stream_waves(const u32x4* __restrict__ src, unsigned* __restrict__ sink, long bytes_per_wave, long nwaves) {
    long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    if (wave >= nwaves) return;
    const u32x4* p = src + wave * (bytes_per_wave / 16) + lane;
    long iters = bytes_per_wave / 1024;
    u32x4 acc = {0, 0, 0, 0};
    for (long i = 0; i < iters; i += U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(p + (i + u) * 64) : p[(i + u) * 64];
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}

int main() {
    const long total = 4L << 30;            // 4 GiB pool, rotate through it so nothing stays in the 256 MiB cache
    u32x4* buf; unsigned* sink;
    CK(hipMalloc(&buf, total)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, total));
    long region = 185L << 20;                // one "launch" reads 185 MB like the MoE gate_up GEMM
    int nreg = (int)(total / region);
    int rot = 0;
    auto run = [&](auto kern, long bytes_per_wave, int wg_threads, const char* name) {
        long nwaves = region / bytes_per_wave;
        long threads = nwaves * 64;
        dim3 grid((threads + wg_threads - 1) / wg_threads), block(wg_threads);
        float us = timeit([&]() { const u32x4* src = (const u32x4*)((const char*)buf + (long)(rot++ % nreg) * region);
                                  hipLaunchKernelGGL(kern, grid, block, 0, 0, src, sink, bytes_per_wave, nwaves); }, 40);
        printf("%-44s waves=%6ld wg=%4d : %7.2f us  %6.2f TB/s\n", name, nwaves, wg_threads, us, region / us / 1e6);
    };
    if (getenv("MEMBW_MALL")) {
        // Does the 256 MiB Infinity Cache (memory-side) serve a 185 MB working set faster than HBM, and does a PREFETCH pass — plain or
        // non-temporal loads by a few waves — leave the region there for a later kernel?  flush = stream 1 GiB of other data.
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const u32x4* R = buf;                                                  // the region under test
        const u32x4* F = (const u32x4*)((const char*)buf + (2L << 30));        // flush source
        auto G = [&](auto kern, const u32x4* src, long bytes, long bpw, int wg) {
            long nwaves = bytes / bpw; long threads = nwaves * 64;
            hipLaunchKernelGGL(kern, dim3((threads + wg - 1) / wg), dim3(wg), 0, 0, src, sink, bpw, nwaves);
        };
        auto timed = [&](const char* name, auto prep) {
            float best = 1e30f, sum = 0;
            for (int rep = 0; rep < 12; rep++) {
                G(stream_waves<8, true>, F, 1L << 30, 64 << 10, 256);          // flush
                prep();
                hipEventRecord(e0);
                G(stream_waves<4, true>, R, region, 64 << 10, 64);             // the consumer: the MoE GEMM's access shape
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2) { best = ms < best ? ms : best; sum += ms; }
            }
            printf("%-58s consumer: best %6.2f us (%5.2f TB/s)  mean %6.2f us\n", name, best * 1e3f, region / (best * 1e3) / 1e6, sum / 10 * 1e3f);
        };
        timed("cold (flushed)", [&]() {});
        timed("after a full-rate nt read of the region", [&]() { G(stream_waves<4, true>, R, region, 64 << 10, 64); });
        timed("after a full-rate plain read of the region", [&]() { G(stream_waves<4, false>, R, region, 64 << 10, 64); });
        timed("after a prefetch pass: 512 waves, nt, U=8", [&]() { G(stream_waves<8, true>, R, region, region / 512 / 8192 * 8192, 64); });
        timed("after a prefetch pass: 512 waves, plain, U=8", [&]() { G(stream_waves<8, false>, R, region, region / 512 / 8192 * 8192, 64); });
        timed("after a prefetch pass: 1024 waves, plain, U=16", [&]() { G(stream_waves<16, false>, R, region, region / 1024 / 16384 * 16384, 64); });
        // how fast are the prefetch passes themselves?
        for (int waves : {256, 512, 1024, 2048}) {
            long bpw = region / waves / 16384 * 16384;
            float us = timeit([&]() { G(stream_waves<16, false>, (const u32x4*)((const char*)buf + (long)(rot++ % nreg) * region), region, bpw, 64); }, 20);
            printf("prefetch pass alone, %4d waves plain U=16: %7.2f us  %5.2f TB/s\n", waves, us, (bpw * waves) / us / 1e6);
        }
        return 0;
    }
    if (getenv("MEMBW_DENSE")) {
        // one decode-sized dense projection per launch (Llama-3.1-8B gate_up: 58.7 MB): waves × KiB in flight per wave
        region = 58720256L; nreg = (int)(total / region);
        printf("58.7 MB per launch\n");
        for (int wg : {256, 512}) {
            for (long waves : {896L, 1792L, 3584L, 7168L, 14336L}) {
                long bpw = region / waves;
                char nm[64];
                snprintf(nm, 64, "%ldKB/wave U=2", bpw >> 10); run(stream_waves<2, true>, bpw, wg, nm);
                snprintf(nm, 64, "%ldKB/wave U=4", bpw >> 10); run(stream_waves<4, true>, bpw, wg, nm);
                snprintf(nm, 64, "%ldKB/wave U=8", bpw >> 10); if (bpw >= 8192) run(stream_waves<8, true>, bpw, wg, nm);
                snprintf(nm, 64, "%ldKB/wave U=16", bpw >> 10); if (bpw >= 16384) run(stream_waves<16, true>, bpw, wg, nm);
            }
        }
        return 0;
    }
    run(stream_waves<4, true>, 64 << 10, 64, "64KB/wave U=4 nt wg64");
    run(stream_waves<4, false>, 64 << 10, 64, "64KB/wave U=4 plain wg64");
    run(stream_waves<8, true>, 64 << 10, 64, "64KB/wave U=8 nt wg64");
    run(stream_waves<8, true>, 64 << 10, 256, "64KB/wave U=8 nt wg256");
    run(stream_waves<16, true>, 64 << 10, 64, "64KB/wave U=16 nt wg64");
    run(stream_waves<8, true>, 16 << 10, 256, "16KB/wave U=8 nt wg256");
    run(stream_waves<8, true>, 32 << 10, 256, "32KB/wave U=8 nt wg256");
    run(stream_waves<16, true>, 16 << 10, 256, "16KB/wave U=16 nt wg256");
    run(stream_waves<8, false>, 16 << 10, 256, "16KB/wave U=8 plain wg256");
    run(stream_waves<4, true>, 8 << 10, 256, "8KB/wave U=4 nt wg256");
    run(stream_waves<8, true>, 8 << 10, 256, "8KB/wave U=8 nt wg256");
    run(stream_waves<8, true>, 256 << 10, 64, "256KB/wave U=8 nt wg64");
    return 0;
}
