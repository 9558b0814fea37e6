// Microbenchmark: how many VALU instructions fit beside an MFMA stream on gfx950, for the two f16 MFMA shapes.
// Each wave runs ITER × (4 independent MFMAs, each followed by NV v_and_or_b32 on private registers); one or two waves per
// SIMD.  Prints ns per MFMA and the implied cycles at the measured clock (s_memrealtime is 100 MHz, so wall time is used).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_issue.hip -o ferrum-infer-rs_amd/bin/mfma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <int NV>
__device__ __forceinline__ void valu(uint32_t (&x)[4], uint32_t m, uint32_t c) {
#pragma unroll
    for (int i = 0; i < NV; i++) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[i & 3]) : "s"(m), "v"(c));
}

template <int NV>
__global__ __launch_bounds__(512) void k16(float* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
    float4v acc[4] = {};
    uint32_t x[4] = {threadIdx.x, 2, 3, 4};
    const uint32_t m = 0x0f0f0f0fu, c = threadIdx.x | 0x64006400u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
            valu<NV>(x, m, c);
        }
    }
    float s = 0;
    for (int j = 0; j < 4; j++) s += acc[j][0] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(x[0] ^ x[1] ^ x[2] ^ x[3]);
}

template <int NV>
__global__ __launch_bounds__(512) void k32(float* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
    float16v acc[4] = {};
    uint32_t x[4] = {threadIdx.x, 2, 3, 4};
    const uint32_t m = 0x0f0f0f0fu, c = threadIdx.x | 0x64006400u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
            valu<NV>(x, m, c);
        }
    }
    float s = 0;
    for (int j = 0; j < 4; j++) s += acc[j][0] + acc[j][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(x[0] ^ x[1] ^ x[2] ^ x[3]);
}

template <typename F>
static float time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    const int iters = 20000;
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("clock %d kHz (nominal); %d iterations x 4 MFMAs per wave; 256 workgroups\n", clk_khz, iters);
    printf("%-10s %4s %6s %12s %12s\n", "mfma", "NV", "waves", "ns/mfma", "cyc/mfma");
#define RUN(KN, NVV, label)                                                                                 \
    for (int threads : {256, 512}) {                                                                        \
        float ms = time_ms([&] { hipLaunchKernelGGL(KN<NVV>, dim3(256), dim3(threads), 0, 0, out, iters); }); \
        /* per SIMD: waves/SIMD × iters × 4 MFMAs share one MFMA pipe */                                     \
        double per = ms * 1e6 / ((double)iters * 4 * (threads / 256));                                      \
        printf("%-10s %4d %6d %12.2f %12.1f\n", label, NVV, threads / 256, per, per * clk_khz * 1e-6);       \
    }
    RUN(k16, 0, "16x16x32") RUN(k16, 2, "16x16x32") RUN(k16, 4, "16x16x32") RUN(k16, 6, "16x16x32")
    RUN(k16, 8, "16x16x32") RUN(k16, 12, "16x16x32") RUN(k16, 16, "16x16x32")
    RUN(k32, 0, "32x32x16") RUN(k32, 4, "32x32x16") RUN(k32, 8, "32x32x16") RUN(k32, 12, "32x32x16")
    RUN(k32, 16, "32x32x16") RUN(k32, 24, "32x32x16") RUN(k32, 32, "32x32x16")
    hipFree(out);
    return 0;
}
