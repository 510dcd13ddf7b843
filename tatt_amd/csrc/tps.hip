// Thin-plate-spline rectification of the LR input (train-time STN path):
//   reference TPSSpatialTransformer.forward, model/tps_spatial_transformer.py:97-112
//   + F.grid_sample(bilinear, zeros padding, align_corners=False), :11
// tps_grid:   src[b,p,:] = repr[p,:] @ (inverse_kernel @ [ctrl[b]; padding])        (B,P,2)
// grid_sample: out[b,h,w,c] = bilinear(x[b,c], 2*clamp(src,0,1)-1)   -> NHWC output, strided input
// backward: d ctrl through both (the LR image itself carries no gradient).
#include "common.h"

#define TPS_MAXNP 32

__global__ __launch_bounds__(256) void tps_grid_fwd_kernel(const float* __restrict__ ctrl, const float* __restrict__ inv,
                                                           const float* __restrict__ pad, const float* __restrict__ repr,
                                                           float* __restrict__ src, int N, int P) {
    __shared__ float Y[TPS_MAXNP][2];
    __shared__ double Mp[TPS_MAXNP][2];
    const int NP = N + 3, b = blockIdx.y, t = threadIdx.x;
    if (t < NP * 2) {
        int i = t >> 1, d = t & 1;
        Y[i][d] = i < N ? ctrl[((long)b * N + i) * 2 + d] : pad[(i - N) * 2 + d];
    }
    __syncthreads();
    if (t < NP * 2) {
        int j = t >> 1, d = t & 1;
        double s = 0.0;          // the kernel inverse has large cancelling entries: accumulate in fp64
        for (int i = 0; i < NP; ++i) s += (double)inv[j * NP + i] * (double)Y[i][d];
        Mp[j][d] = s;
    }
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + t;
    if (p >= P) return;
    double sx = 0.0, sy = 0.0;
    for (int j = 0; j < NP; ++j) {
        double r = (double)repr[(long)p * NP + j];
        sx += r * Mp[j][0]; sy += r * Mp[j][1];
    }
    src[((long)b * P + p) * 2] = (float)sx;
    src[((long)b * P + p) * 2 + 1] = (float)sy;
}
TATT_API int tatt_tps_grid_fwd(const float* ctrl, const float* inv, const float* pad, const float* repr, float* src,
                               int B, int N, int P, hipStream_t st) {
    if (N + 3 > TPS_MAXNP) return 1;
    hipLaunchKernelGGL(tps_grid_fwd_kernel, dim3(cdiv(P, 256), B), dim3(256), 0, st, ctrl, inv, pad, repr, src, N, P);
    return LAUNCH_CHECK();
}

// one block per image: dM[j][d] = sum_p repr[p][j] * dsrc[b][p][d];  dctrl[b][i][d] = sum_j inv[j][i] * dM[j][d]
__global__ __launch_bounds__(256) void tps_grid_bwd_kernel(const float* __restrict__ dsrc, const float* __restrict__ inv,
                                                           const float* __restrict__ repr, float* __restrict__ dctrl,
                                                           int N, int P) {
    __shared__ double red[4][TPS_MAXNP * 2];
    __shared__ double dM[TPS_MAXNP][2];       // the kernel inverse has large cancelling entries: reduce and contract in fp64
    const int NP = N + 3, b = blockIdx.x, t = threadIdx.x;
    float acc[TPS_MAXNP * 2];
#pragma unroll
    for (int k = 0; k < TPS_MAXNP * 2; ++k) acc[k] = 0.f;
    for (int p = t; p < P; p += blockDim.x) {
        const float gx = dsrc[((long)b * P + p) * 2], gy = dsrc[((long)b * P + p) * 2 + 1];
#pragma unroll
        for (int j = 0; j < TPS_MAXNP; ++j)
            if (j < NP) {
                float r = repr[(long)p * NP + j];
                acc[2 * j] = fmaf(r, gx, acc[2 * j]); acc[2 * j + 1] = fmaf(r, gy, acc[2 * j + 1]);
            }
    }
#pragma unroll
    for (int k = 0; k < TPS_MAXNP * 2; ++k) {
        double v = wave_sum_d((double)acc[k]);   // per-thread sums hold P/256 = 4 terms
        if ((t & 63) == 0) red[t >> 6][k] = v;
    }
    __syncthreads();
    if (t < NP * 2) dM[t >> 1][t & 1] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    __syncthreads();
    if (t < N * 2) {
        int i = t >> 1, d = t & 1;
        double s = 0.0;
        for (int j = 0; j < NP; ++j) s += (double)inv[j * NP + i] * dM[j][d];
        dctrl[((long)b * N + i) * 2 + d] = (float)s;
    }
}
TATT_API int tatt_tps_grid_bwd(const float* dsrc, const float* inv, const float* repr, float* dctrl, int B, int N,
                               int P, hipStream_t st) {
    if (N + 3 > TPS_MAXNP) return 1;
    hipLaunchKernelGGL(tps_grid_bwd_kernel, dim3(B), dim3(256), 0, st, dsrc, inv, repr, dctrl, N, P);
    return LAUNCH_CHECK();
}

struct SampleGeom { int B, C, H, W; long xsn, xsc, xsh, xsw; };

__device__ __forceinline__ float tap(const float* x, const SampleGeom& g, int b, int c, int yi, int xi) {
    if (xi < 0 || xi >= g.W || yi < 0 || yi >= g.H) return 0.f;
    return x[b * g.xsn + c * g.xsc + yi * g.xsh + xi * g.xsw];
}

// out (B,H,W,C) NHWC; one thread per output pixel (C <= 4)
__global__ void grid_sample_fwd_kernel(const float* __restrict__ x, const float* __restrict__ src,
                                       float* __restrict__ out, SampleGeom g) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)g.B * g.H * g.W;
    if (idx >= total) return;
    const int b = idx / ((long)g.H * g.W);
    const float cx = fminf(fmaxf(src[idx * 2], 0.f), 1.f), cy = fminf(fmaxf(src[idx * 2 + 1], 0.f), 1.f);
    const float gx = 2.f * cx - 1.f, gy = 2.f * cy - 1.f;
    const float ix = ((gx + 1.f) * g.W - 1.f) * 0.5f, iy = ((gy + 1.f) * g.H - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    for (int c = 0; c < g.C; ++c) {
        float v00 = tap(x, g, b, c, y0, x0), v01 = tap(x, g, b, c, y0, x0 + 1);
        float v10 = tap(x, g, b, c, y0 + 1, x0), v11 = tap(x, g, b, c, y0 + 1, x0 + 1);
        out[idx * g.C + c] = v00 * (1.f - tx) * (1.f - ty) + v01 * tx * (1.f - ty) + v10 * (1.f - tx) * ty + v11 * tx * ty;
    }
}
TATT_API int tatt_grid_sample_fwd(const float* x, long xsn, long xsc, long xsh, long xsw, const float* src, float* out,
                                  int B, int C, int H, int W, hipStream_t st) {
    SampleGeom g = {B, C, H, W, xsn, xsc, xsh, xsw};
    hipLaunchKernelGGL(grid_sample_fwd_kernel, dim3(cdiv((long)B * H * W, 256)), dim3(256), 0, st, x, src, out, g);
    return LAUNCH_CHECK();
}
// dsrc[b,p,:] = sum_c dout[b,p,c] * d out/d (ix,iy) * (W/2, H/2) * 2 * [0 <= src <= 1]
__global__ void grid_sample_bwd_kernel(const float* __restrict__ x, const float* __restrict__ src,
                                       const float* __restrict__ dout, float* __restrict__ dsrc, SampleGeom g) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)g.B * g.H * g.W;
    if (idx >= total) return;
    const int b = idx / ((long)g.H * g.W);
    const float sx = src[idx * 2], sy = src[idx * 2 + 1];
    const float cx = fminf(fmaxf(sx, 0.f), 1.f), cy = fminf(fmaxf(sy, 0.f), 1.f);
    const float gx = 2.f * cx - 1.f, gy = 2.f * cy - 1.f;
    const float ix = ((gx + 1.f) * g.W - 1.f) * 0.5f, iy = ((gy + 1.f) * g.H - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    float dix = 0.f, diy = 0.f;
    for (int c = 0; c < g.C; ++c) {
        float v00 = tap(x, g, b, c, y0, x0), v01 = tap(x, g, b, c, y0, x0 + 1);
        float v10 = tap(x, g, b, c, y0 + 1, x0), v11 = tap(x, g, b, c, y0 + 1, x0 + 1);
        float go = dout[idx * g.C + c];
        dix += go * ((v01 - v00) * (1.f - ty) + (v11 - v10) * ty);
        diy += go * ((v10 - v00) * (1.f - tx) + (v11 - v01) * tx);
    }
    const float mx = (sx >= 0.f && sx <= 1.f) ? 1.f : 0.f, my = (sy >= 0.f && sy <= 1.f) ? 1.f : 0.f;
    dsrc[idx * 2] = dix * (0.5f * g.W) * 2.f * mx;
    dsrc[idx * 2 + 1] = diy * (0.5f * g.H) * 2.f * my;
}
TATT_API int tatt_grid_sample_bwd(const float* x, long xsn, long xsc, long xsh, long xsw, const float* src,
                                  const float* dout, float* dsrc, int B, int C, int H, int W, hipStream_t st) {
    SampleGeom g = {B, C, H, W, xsn, xsc, xsh, xsw};
    hipLaunchKernelGGL(grid_sample_bwd_kernel, dim3(cdiv((long)B * H * W, 256)), dim3(256), 0, st, x, src, dout, dsrc, g);
    return LAUNCH_CHECK();
}
