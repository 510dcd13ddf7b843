// micro-benchmark: fp32 MFMA issue rate for dependent accumulator chains (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float av = a + threadIdx.x * 1e-9f, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    float av = a + threadIdx.x * 1e-9f, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
void run(const char* name, F launch, double flop_per_mfma, int nacc, int waves_per_simd) {
    float* out; hipMalloc(&out, 1 << 24);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n_mfma_per_wave = (double)iters * 8 * nacc;
    double total = n_mfma_per_wave * 256 * 4 * waves_per_simd;       // 256 CUs x 4 SIMDs
    printf("%-34s %8.3f ms  %7.1f TFLOP/s   %.1f ns per MFMA per SIMD\n", name, ms, total * flop_per_mfma / ms / 1e9,
           ms * 1e6 / (n_mfma_per_wave * waves_per_simd));
    hipFree(out);
}
int main() {
    run("32x32x2 1 acc, 1 wave/SIMD", [](float* o, int it) { hipLaunchKernelGGL(k32<1>, dim3(256), dim3(256), 0, 0, o, it, 1.f, 2.f); }, 4096, 1, 1);
    run("32x32x2 2 acc, 1 wave/SIMD", [](float* o, int it) { hipLaunchKernelGGL(k32<2>, dim3(256), dim3(256), 0, 0, o, it, 1.f, 2.f); }, 4096, 2, 1);
    run("32x32x2 4 acc, 1 wave/SIMD", [](float* o, int it) { hipLaunchKernelGGL(k32<4>, dim3(256), dim3(256), 0, 0, o, it, 1.f, 2.f); }, 4096, 4, 1);
    run("32x32x2 1 acc, 2 waves/SIMD", [](float* o, int it) { hipLaunchKernelGGL(k32<1>, dim3(512), dim3(256), 0, 0, o, it, 1.f, 2.f); }, 4096, 1, 2);
    run("32x32x2 1 acc, 4 waves/SIMD", [](float* o, int it) { hipLaunchKernelGGL(k32<1>, dim3(1024), dim3(256), 0, 0, o, it, 1.f, 2.f); }, 4096, 1, 4);
    run("16x16x4 1 acc, 1 wave/SIMD", [](float* o, int it) { hipLaunchKernelGGL(k16<1>, dim3(256), dim3(256), 0, 0, o, it, 1.f, 2.f); }, 2048, 1, 1);
    run("16x16x4 2 acc, 1 wave/SIMD", [](float* o, int it) { hipLaunchKernelGGL(k16<2>, dim3(256), dim3(256), 0, 0, o, it, 1.f, 2.f); }, 2048, 2, 1);
    run("16x16x4 4 acc, 1 wave/SIMD", [](float* o, int it) { hipLaunchKernelGGL(k16<4>, dim3(256), dim3(256), 0, 0, o, it, 1.f, 2.f); }, 2048, 4, 1);
    run("16x16x4 1 acc, 2 waves/SIMD", [](float* o, int it) { hipLaunchKernelGGL(k16<1>, dim3(512), dim3(256), 0, 0, o, it, 1.f, 2.f); }, 2048, 1, 2);
    return 0;
}
