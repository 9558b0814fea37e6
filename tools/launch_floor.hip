// Development micro-benchmark: the floor of a dependent kernel chain replayed from a hipGraph on MI355X — how much of the
// ≈ 4.7 µs a trivial kernel of the decode step costs is dispatch, and how much is the kernel's own chain of memory round trips.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_empty() {}
// DEPTH dependent global round trips: p = buf[p] …, then one store (what the next kernel reads)
template <int DEPTH>
__global__ void k_chase(const unsigned* __restrict__ in, unsigned* __restrict__ out, int n) {
    unsigned p = (blockIdx.x * blockDim.x + threadIdx.x) % n;
#pragma unroll
    for (int d = 0; d < DEPTH; d++) p = in[p];
    out[(blockIdx.x * blockDim.x + threadIdx.x) % n] = p;
}

template <typename F> static float graph_us_per_kernel(F enqueue, int chain, hipStream_t s) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < chain; i++) enqueue(i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(a, s));
        for (int i = 0; i < 4; i++) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best * 1000.f / (4.f * chain);
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int n = 1 << 16;
    unsigned *a, *b;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    unsigned* h = (unsigned*)malloc(n * 4);
    for (int i = 0; i < n; i++) h[i] = (unsigned)((i * 7919u + 13u) % n);
    CK(hipMemcpy(a, h, n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, h, n * 4, hipMemcpyHostToDevice));
    const int chain = 300;
    for (int wgs : {1, 32, 256, 2048}) {
        printf("workgroups=%4d x 256 threads, chain of %d kernels in one graph:\n", wgs, chain);
        printf("  empty kernel                         %.2f us per kernel\n", graph_us_per_kernel([&](int) { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, s); }, chain, s));
        printf("  1 load -> store (ping-pong buffers)  %.2f us per kernel\n", graph_us_per_kernel([&](int i) { hipLaunchKernelGGL(k_chase<1>, dim3(wgs), dim3(256), 0, s, i & 1 ? b : a, i & 1 ? a : b, n); }, chain, s));
        printf("  2 dependent loads -> store           %.2f us per kernel\n", graph_us_per_kernel([&](int i) { hipLaunchKernelGGL(k_chase<2>, dim3(wgs), dim3(256), 0, s, i & 1 ? b : a, i & 1 ? a : b, n); }, chain, s));
        printf("  4 dependent loads -> store           %.2f us per kernel\n", graph_us_per_kernel([&](int i) { hipLaunchKernelGGL(k_chase<4>, dim3(wgs), dim3(256), 0, s, i & 1 ? b : a, i & 1 ? a : b, n); }, chain, s));
    }
    return 0;
}
