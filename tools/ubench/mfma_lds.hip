// micro-benchmark: fp32 MFMA chain fed from LDS (the conv3 inner loop in isolation), gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define XP 68
// MODE 0: 16-byte reads, no barrier;  1: 16-byte reads + __syncthreads per tap;  2: 4-byte reads (pitch 65), no barrier
// MODE 3: 16-byte reads, random data (same as 0 but LDS filled with pseudo-random values)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int rnd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                 // 3*66*68
    float* Ws = smem + 3 * 66 * XP;   // 64*68
    for (int i = threadIdx.x; i < 3 * 66 * XP + 64 * XP; i += 256) {
        unsigned h = (i * 2654435761u) ^ (blockIdx.x * 40503u);
        smem[i] = rnd ? ((h >> 9) * (1.0f / 8388608.0f) - 1.0f) : 1.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave & 1, wn = wave >> 1;
    f32x16 acc;
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - 3 * kh;
            if (MODE == 2) {
                const float* arow = Xs + (kh * 66 + wm * 32 + (lane & 31) + kw) * 65 + (lane >> 5);
                const float* brow = Ws + (lane >> 5) * 64 + wn * 32 + (lane & 31);
#pragma unroll
                for (int kk = 0; kk < 64; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[kk], brow[kk * 64], acc, 0, 0, 0);
            } else {
                const float* arow = Xs + (kh * 66 + wm * 32 + (lane & 31) + kw) * XP + 4 * (lane >> 5);
                const float* brow = Ws + (wn * 32 + (lane & 31)) * XP + 4 * (lane >> 5);
                f32x4 va[8], vb[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) { va[c] = *(const f32x4*)(arow + 8 * c); vb[c] = *(const f32x4*)(brow + 8 * c); }
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[c][u], vb[c][u], acc, 0, 0, 0);
            }
            if (MODE == 1) __syncthreads();
        }
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += acc[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int rnd) {
    float* out; hipMalloc(&out, 1 << 22);
    const int lds = (3 * 66 * XP + 64 * XP) * 4;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), lds, 0, out, 10, rnd);
    hipDeviceSynchronize();
    const int iters = 300;
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), lds, 0, out, iters, rnd);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * 9 * 32 * 256 * 4;
    printf("%-46s %8.3f ms  %7.1f TFLOP/s  %.1f ns/MFMA/SIMD\n", name, ms, n * 4096 / ms / 1e9, ms * 1e6 / (iters * 9.0 * 32));
    hipFree(out);
}
int main() {
    run<0>("b128 LDS reads, no barrier, const data", 0);
    run<0>("b128 LDS reads, no barrier, random data", 1);
    run<1>("b128 LDS reads, barrier per tap, random data", 1);
    run<2>("b32  LDS reads, no barrier, random data", 1);
    return 0;
}
