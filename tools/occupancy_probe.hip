// Development micro-benchmark: how many waves does MI355X keep resident per CU for a given VGPR count and workgroup size?
// Every wave stamps its start (100 MHz wall clock) and spins ~30 µs; waves that start within 3 µs of the first were resident at once.
//   hipcc --offload-arch=gfx950 -O3 tools/occupancy_probe.hip -o /tmp/occupancy_probe && /tmp/occupancy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int VGPRS, int WG, int LDS>
__global__ __launch_bounds__(WG) void probe(unsigned long long* start, float* sink) {
    __shared__ int pad[LDS / 4 > 0 ? LDS / 4 : 1];
    const unsigned long long t0 = wall_clock64();
    if ((threadIdx.x & 63) == 0) start[(blockIdx.x * WG + threadIdx.x) >> 6] = t0;
    if (LDS > 0) pad[threadIdx.x % (LDS / 4 > 0 ? LDS / 4 : 1)] = (int)t0;
    // keep VGPRS registers live: an array of floats updated in a loop the compiler cannot collapse
    float r[VGPRS > 24 ? VGPRS - 24 : 1];
#pragma unroll
    for (int i = 0; i < (VGPRS > 24 ? VGPRS - 24 : 1); i++) r[i] = (float)(threadIdx.x + i);
    while (wall_clock64() - t0 < 3000) {           // 30 µs
#pragma unroll
        for (int i = 0; i < (VGPRS > 24 ? VGPRS - 24 : 1); i++) r[i] = r[i] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < (VGPRS > 24 ? VGPRS - 24 : 1); i++) s += r[i];
    if (s == 12345.678f) sink[0] = s + (LDS > 0 ? pad[0] : 0);
}

template <int VGPRS, int WG, int LDS>
static void run(unsigned long long* dstart, float* sink) {
    const int waves = 16384, wgs = waves / (WG / 64);
    CK(hipMemset(dstart, 0, waves * 8));
    hipLaunchKernelGGL((probe<VGPRS, WG, LDS>), dim3(wgs), dim3(WG), 0, 0, dstart, sink);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(waves);
    CK(hipMemcpy(h.data(), dstart, waves * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = *std::min_element(h.begin(), h.end());
    int early = 0;
    for (auto t : h) early += (t - t0) < 300;
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)probe<VGPRS, WG, LDS>));
    int nb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, probe<VGPRS, WG, LDS>, WG, 0));
    printf("asked vgpr~%3d (compiled %3d) wg=%4d lds=%5d B: resident at once %5d waves = %5.1f per CU (%4.1f per SIMD); runtime says %d workgroups per CU\n",
           VGPRS, fa.numRegs, WG, (int)fa.sharedSizeBytes, early, early / 256.0, early / 1024.0, nb);
}

int main() {
    unsigned long long* dstart; float* sink;
    CK(hipMalloc(&dstart, 16384 * 8)); CK(hipMalloc(&sink, 4));
    run<32, 64, 0>(dstart, sink);
    run<64, 64, 0>(dstart, sink);
    run<80, 64, 0>(dstart, sink);
    run<80, 64, 4096>(dstart, sink);
    run<80, 64, 2048>(dstart, sink);
    run<96, 64, 0>(dstart, sink);
    run<128, 64, 0>(dstart, sink);
    run<80, 256, 0>(dstart, sink);
    run<80, 256, 4096>(dstart, sink);
    run<128, 256, 0>(dstart, sink);
    run<32, 256, 0>(dstart, sink);
    return 0;
}
