// Do a VALU-bound wave and an MFMA-bound wave of the same SIMD overlap?  One 512-thread work-group per CU: waves 0-3 (one per SIMD) run
// dependent chains of v_mfma_f32_32x32x16_bf16, waves 4-7 run VALU work (fma, or v_exp, or v_cvt_pk).  mode bit 0: matrix waves active,
// bit 1: VALU waves active.  build: hipcc --offload-arch=gfx950 -O2 tools/ubench/coissue.hip -o tools/ubench/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int valu_kind, int nchain>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            bf16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
            f32x16 c0 = {}, c1 = {};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                    if (nchain == 2) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
                    else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c0, 0, 0, 0);
                }
            }
            for (int i = 0; i < 16; ++i) r += c0[i] + c1[i];
        }
    } else if (mode & 2) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.01f + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j)            // 128 VALU instructions per iteration = 512 cycles: the matrix waves do 16 x 32 = 512
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (valu_kind == 0) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
                    else if (valu_kind == 1) x[i] = __builtin_amdgcn_exp2f(x[i]) * 0.0f + x[i];
                    else { unsigned u = __builtin_bit_cast(unsigned, x[i]); u ^= u >> 16; u *= 0x85EBCA6Bu; x[i] = __builtin_bit_cast(float, u); }
                }
        }
        for (int i = 0; i < 8; ++i) r += x[i];
    }
    if (r == 12345.f) out[threadIdx.x] = r;
}
int main() {
    float* d; hipMalloc(&d, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
#define RUN(kind, nchain) \
    for (int mode = 1; mode <= 3; ++mode) { \
        k<kind, nchain><<<256, 512>>>(d, iters, mode); \
        hipEventRecord(e0); \
        k<kind, nchain><<<256, 512>>>(d, iters, mode); \
        hipEventRecord(e1); hipEventSynchronize(e1); \
        float ms; hipEventElapsedTime(&ms, e0, e1); \
        printf("valu kind %d (0 fma, 1 exp+fma, 2 int hash)  mfma chains %d  mode %d (1 = matrix waves, 2 = VALU waves, 3 = both): %.1f us\n", kind, nchain, mode, ms * 1e3); \
    }
    RUN(0, 1) RUN(0, 2) RUN(1, 2) RUN(2, 2)
    return 0;
}
