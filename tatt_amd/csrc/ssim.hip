// SSIM / TRI_SSIM losses and the rotation augmentation of the shipped TATT training recipe (gfx950)  -- SURVEY.md 8f-2.
//
//   reference utils/ssim_psnr.py:76-97 (_ssim), :99-129 (_tri_ssim), :28-37 (11x11 Gaussian window, sigma 1.5, zero padding,
//   one depth-wise filter per channel); model/__init__.py:4-29 / interfaces/super_resolution.py:126-157 (torch_distortion:
//   F.affine_grid + F.grid_sample, bilinear, zeros padding, align_corners=False).
//
// Per pixel q the loss needs nine windowed raw moments of the three images (three means m_i = w*x_i, three E[x_i^2], three
// E[x_i x_j]); with  P = m1m2+m2m3+m3m1+C1,  Q = s12+s23+s31+C2,  R = m1^2+m2^2+m3^2+C1,  T = s1+s2+s3+C2  (s.. = E[..] - m.m.)
// the map is S = PQ/(RT) (pair form: P = 2m1m2+C1, Q = 2 s12+C2, R = m1^2+m2^2+C1, T = s1+s2+C2).
//
// forward : one kernel; a 16x64-pixel tile of one (sample, channel) plane + 5-pixel halo of all three images in LDS (23 KB), every
//           thread filters 4 pixels with the 121 taps directly, fp64 partial sums per tile -> deterministic per-sample mean.
// backward: S depends on x_i[p] only through the moments, so
//              dL/dx1[p] = (w * A1)[p] + 2 x1[p] (w * U)[p] + (x2[p] + x3[p]) (w * V)[p]
//           with the per-pixel maps A_i = dS/dm_i (total), U = dS/dE[x_i^2] = -S/T, V = dS/dE[x_i x_j] = P/(RT) (x2 in pair form),
//           all scaled by the upstream weight of the sample.  Kernel 1 = the forward filter again, writing the 5 maps (28 MB at
//           B = 48, HR 32x128); kernel 2 filters the maps (the window is symmetric) and combines them with the pixel values.
// Images are addressed by explicit strides: the SR output is NCHW-shaped over NHWC memory.  HBM traffic is tiny; the kernels are
// LDS/VALU-bound (121 taps x 12 FMA per pixel = 1.1 GFLOP forward at B = 48).
#include "common.h"

#define SS_R 5
#define SS_K 11
#define SS_TH 16
#define SS_TW 64
#define SS_LH (SS_TH + 2 * SS_R)
#define SS_LW (SS_TW + 2 * SS_R)
#define SS_LP (SS_LW + 1)

struct SsimImg {
    const float* p;
    long sn, sc, sh, sw;
};
struct SsimP {
    SsimImg x[3];
    int B, C, H, W, tri, tiles_h, tiles_w;
    float g[SS_K];              // 1-D Gaussian (float, normalised) -- the 2-D window is fl(g[i]*g[j]) like the reference's outer product
};

__device__ __forceinline__ void ssim_load_tile(const SsimP& p, int n, int c, int h0, int w0, int nimg, float (*t)[SS_LH][SS_LP]) {
    for (int i = threadIdx.x; i < SS_LH * SS_LW; i += 256) {
        const int r = i / SS_LW, q = i - r * SS_LW;
        const int h = h0 + r - SS_R, w = w0 + q - SS_R;
        const bool in = h >= 0 && h < p.H && w >= 0 && w < p.W;
        for (int k = 0; k < nimg; ++k)
            t[k][r][q] = in ? p.x[k].p[n * p.x[k].sn + c * p.x[k].sc + h * p.x[k].sh + w * p.x[k].sw] : 0.f;
    }
}

struct Moments { float m[3], e[3], x[3]; };      // means, E[x_i^2], E[x1x2], E[x2x3], E[x3x1]

__device__ __forceinline__ void ssim_moments(const SsimP& p, const float (*t)[SS_LH][SS_LP], int r, int q, Moments& o) {
#pragma unroll
    for (int k = 0; k < 3; ++k) o.m[k] = o.e[k] = o.x[k] = 0.f;
    for (int i = 0; i < SS_K; ++i) {
#pragma unroll
        for (int j = 0; j < SS_K; ++j) {
            const float w = p.g[i] * p.g[j];
            const float a = t[0][r + i][q + j], b = t[1][r + i][q + j];
            o.m[0] = fmaf(w, a, o.m[0]); o.m[1] = fmaf(w, b, o.m[1]);
            o.e[0] = fmaf(w, a * a, o.e[0]); o.e[1] = fmaf(w, b * b, o.e[1]);
            o.x[0] = fmaf(w, a * b, o.x[0]);
            if (p.tri) {
                const float c = t[2][r + i][q + j];
                o.m[2] = fmaf(w, c, o.m[2]); o.e[2] = fmaf(w, c * c, o.e[2]);
                o.x[1] = fmaf(w, b * c, o.x[1]); o.x[2] = fmaf(w, c * a, o.x[2]);
            }
        }
    }
}

// S and (optionally) its partial derivatives w.r.t. the raw moments
__device__ __forceinline__ float ssim_value(const SsimP& p, const Moments& o, float* A, float* U, float* V) {
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float P, Q, R, T;
    const float m1 = o.m[0], m2 = o.m[1], m3 = o.m[2];
    if (p.tri) {
        P = m1 * m2 + m2 * m3 + m3 * m1 + C1;
        Q = (o.x[0] - m1 * m2) + (o.x[1] - m2 * m3) + (o.x[2] - m3 * m1) + C2;
        R = m1 * m1 + m2 * m2 + m3 * m3 + C1;
        T = (o.e[0] - m1 * m1) + (o.e[1] - m2 * m2) + (o.e[2] - m3 * m3) + C2;
    } else {
        P = 2.f * m1 * m2 + C1;
        Q = 2.f * (o.x[0] - m1 * m2) + C2;
        R = m1 * m1 + m2 * m2 + C1;
        T = (o.e[0] - m1 * m1) + (o.e[1] - m2 * m2) + C2;
    }
    const float S = (P * Q) / (R * T);
    if (A) {
        const float irt = 1.f / (R * T);
        const float SP = Q * irt, SQ = P * irt, SR = -S / R, ST = -S / T;
        if (p.tri) {
            A[0] = (m2 + m3) * (SP - SQ) + 2.f * m1 * (SR - ST);
            A[1] = (m1 + m3) * (SP - SQ) + 2.f * m2 * (SR - ST);
            A[2] = (m1 + m2) * (SP - SQ) + 2.f * m3 * (SR - ST);
            *V = SQ;
        } else {
            A[0] = 2.f * m2 * (SP - SQ) + 2.f * m1 * (SR - ST);
            A[1] = 2.f * m1 * (SP - SQ) + 2.f * m2 * (SR - ST);
            A[2] = 0.f;
            *V = 2.f * SQ;
        }
        *U = ST;
    }
    return S;
}

// grid: (tiles_w * tiles_h, C, B); part[((n*C + c)*tiles + tile)] = sum of S over the tile (fp64);  maps (optional):
// [n][c][5][H][W] = (A1, A2, A3, U, V) * gs[n] / (C*H*W)
__global__ __launch_bounds__(256) void ssim_fwd_kernel(SsimP p, double* __restrict__ part, const float* __restrict__ gs,
                                                       float* __restrict__ maps) {
    __shared__ float t[3][SS_LH][SS_LP];
    __shared__ double red[4];
    const int tile = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
    const int th = tile / p.tiles_w, tw = tile - th * p.tiles_w;
    const int h0 = th * SS_TH, w0 = tw * SS_TW;
    ssim_load_tile(p, n, c, h0, w0, p.tri ? 3 : 2, t);
    __syncthreads();
    const int r = threadIdx.x >> 4, q0 = (threadIdx.x & 15) * 4;        // 16 rows x 16 groups of 4 pixels
    double acc = 0.0;
    const float up = gs ? gs[n] / (float)((long)p.C * p.H * p.W) : 0.f;
#pragma unroll 1
    for (int u = 0; u < 4; ++u) {
        const int q = q0 + u, h = h0 + r, w = w0 + q;
        if (h >= p.H || w >= p.W) continue;
        Moments o;
        ssim_moments(p, t, r, q, o);
        float A[3], U, V;
        const float S = ssim_value(p, o, maps ? A : nullptr, &U, &V);
        acc += (double)S;
        if (maps) {
            const long hw = (long)p.H * p.W;
            float* mp = maps + ((long)(n * p.C + c) * 5) * hw + (long)h * p.W + w;
            mp[0] = A[0] * up; mp[hw] = A[1] * up; mp[2 * hw] = A[2] * up; mp[3 * hw] = U * up; mp[4 * hw] = V * up;
        }
    }
    if (part) {
        acc = wave_sum_d(acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) part[(long)(n * p.C + c) * (p.tiles_h * p.tiles_w) + tile] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}
// per-sample mean of S over (C, H, W)
__global__ void ssim_mean_kernel(const double* __restrict__ part, int per_sample, double inv_count, float* __restrict__ out) {
    const int n = blockIdx.x;
    double a = 0.0;
    for (int i = threadIdx.x; i < per_sample; i += 64) a += part[(long)n * per_sample + i];
    a = wave_sum_d(a);
    if (threadIdx.x == 0) out[n] = (float)(a * inv_count);
}

// dx_i[n,c,h,w] (contiguous NCHW) from the 5 maps; same tiling, the maps' tile + halo in LDS
__global__ __launch_bounds__(256) void ssim_bwd_kernel(SsimP p, const float* __restrict__ maps, float* __restrict__ dx1,
                                                       float* __restrict__ dx2, float* __restrict__ dx3) {
    __shared__ float t[5][SS_LH][SS_LP];
    const int tile = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
    const int th = tile / p.tiles_w, tw = tile - th * p.tiles_w;
    const int h0 = th * SS_TH, w0 = tw * SS_TW;
    const long hw = (long)p.H * p.W;
    const float* mp = maps + ((long)(n * p.C + c) * 5) * hw;
    for (int i = threadIdx.x; i < SS_LH * SS_LW; i += 256) {
        const int r = i / SS_LW, q = i - r * SS_LW;
        const int h = h0 + r - SS_R, w = w0 + q - SS_R;
        const bool in = h >= 0 && h < p.H && w >= 0 && w < p.W;
#pragma unroll
        for (int k = 0; k < 5; ++k) t[k][r][q] = in ? mp[k * hw + (long)h * p.W + w] : 0.f;
    }
    __syncthreads();
    const int r = threadIdx.x >> 4, q0 = (threadIdx.x & 15) * 4;
#pragma unroll 1
    for (int u = 0; u < 4; ++u) {
        const int q = q0 + u, h = h0 + r, w = w0 + q;
        if (h >= p.H || w >= p.W) continue;
        float f[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < SS_K; ++i)
#pragma unroll
            for (int j = 0; j < SS_K; ++j) {
                const float wgt = p.g[i] * p.g[j];
#pragma unroll
                for (int k = 0; k < 5; ++k) f[k] = fmaf(wgt, t[k][r + i][q + j], f[k]);
            }
        float x[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            x[k] = (k < 2 || p.tri) ? p.x[k].p[n * p.x[k].sn + c * p.x[k].sc + h * p.x[k].sh + w * p.x[k].sw] : 0.f;
        const long o = ((long)(n * p.C + c)) * hw + (long)h * p.W + w;
        if (dx1) dx1[o] = f[0] + 2.f * x[0] * f[3] + (x[1] + x[2]) * f[4];
        if (dx2) dx2[o] = f[1] + 2.f * x[1] * f[3] + (x[0] + x[2]) * f[4];
        if (dx3 && p.tri) dx3[o] = f[2] + 2.f * x[2] * f[3] + (x[0] + x[1]) * f[4];
    }
}

static SsimP make_ssimp(const float* x1, const long* s1, const float* x2, const long* s2, const float* x3, const long* s3, int B,
                        int C, int H, int W) {
    SsimP p;
    const float* xs[3] = {x1, x2, x3};
    const long* ss[3] = {s1, s2, s3};
    for (int k = 0; k < 3; ++k) {
        p.x[k].p = xs[k];
        p.x[k].sn = ss[k][0]; p.x[k].sc = ss[k][1]; p.x[k].sh = ss[k][2]; p.x[k].sw = ss[k][3];
    }
    p.B = B; p.C = C; p.H = H; p.W = W;
    p.tri = x3 != nullptr;
    p.tiles_h = cdiv(H, SS_TH); p.tiles_w = cdiv(W, SS_TW);
    // reference gaussian(11, 1.5): exp(-(x - 5)^2 / (2 * 1.5^2)) in double, stored as float, normalised by the float sum
    float g[SS_K], s = 0.f;
    for (int i = 0; i < SS_K; ++i) { g[i] = (float)exp(-(double)((i - SS_R) * (i - SS_R)) / (2.0 * 1.5 * 1.5)); s += g[i]; }
    for (int i = 0; i < SS_K; ++i) p.g[i] = g[i] / s;
    return p;
}

// out[n] = mean_{c,h,w} S;  x3 == NULL: SSIM of two images, else TRI_SSIM.  ws: B*C*tiles doubles.
TATT_API int tatt_ssim_fwd(const float* x1, long a_n, long a_c, long a_h, long a_w, const float* x2, long b_n, long b_c, long b_h,
                           long b_w, const float* x3, long c_n, long c_c, long c_h, long c_w, int B, int C, int H, int W,
                           float* out, double* ws, hipStream_t st) {
    const long s1[4] = {a_n, a_c, a_h, a_w}, s2[4] = {b_n, b_c, b_h, b_w}, s3[4] = {c_n, c_c, c_h, c_w};
    SsimP p = make_ssimp(x1, s1, x2, s2, x3, s3, B, C, H, W);
    const int tiles = p.tiles_h * p.tiles_w;
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3(tiles, C, B), dim3(256), 0, st, p, ws, (const float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(ssim_mean_kernel, dim3(B), dim3(64), 0, st, ws, C * tiles, 1.0 / ((double)C * H * W), out);
    return LAUNCH_CHECK();
}
// gs[n]: upstream gradient of out[n].  maps: workspace of B*C*5*H*W floats.  dx1/dx2/dx3: contiguous (B,C,H,W), any may be NULL.
TATT_API int tatt_ssim_bwd(const float* x1, long a_n, long a_c, long a_h, long a_w, const float* x2, long b_n, long b_c, long b_h,
                           long b_w, const float* x3, long c_n, long c_c, long c_h, long c_w, int B, int C, int H, int W,
                           const float* gs, float* maps, float* dx1, float* dx2, float* dx3, hipStream_t st) {
    const long s1[4] = {a_n, a_c, a_h, a_w}, s2[4] = {b_n, b_c, b_h, b_w}, s3[4] = {c_n, c_c, c_h, c_w};
    SsimP p = make_ssimp(x1, s1, x2, s2, x3, s3, B, C, H, W);
    const int tiles = p.tiles_h * p.tiles_w;
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3(tiles, C, B), dim3(256), 0, st, p, (double*)nullptr, gs, maps);
    hipLaunchKernelGGL(ssim_bwd_kernel, dim3(tiles, C, B), dim3(256), 0, st, p, maps, dx1, dx2, dx3);
    return LAUNCH_CHECK();
}

// ---- rotation / aspect-jitter resampling: out = grid_sample(img, affine_grid(theta)) --------------------------------------------
// theta (N,2,3) row-major.  Sampling position of output pixel (h,w), align_corners=False:
//   xn = (2w+1)/W - 1, yn = (2h+1)/H - 1;  gx = t00 xn + t01 yn + t02, gy = t10 xn + t11 yn + t12;
//   ix = ((gx+1) W - 1)/2, iy = ((gy+1) H - 1)/2;  bilinear, zeros padding.
struct AffP {
    const float* x; long sn, sc, sh, sw;
    const float* theta;
    int B, C, H, W;
};
__device__ __forceinline__ void aff_pos(const AffP& p, int n, int h, int w, float& ix, float& iy) {
    const float* t = p.theta + n * 6;
    const float xn = (2.f * w + 1.f) / p.W - 1.f, yn = (2.f * h + 1.f) / p.H - 1.f;
    const float gx = t[0] * xn + t[1] * yn + t[2], gy = t[3] * xn + t[4] * yn + t[5];
    ix = ((gx + 1.f) * p.W - 1.f) * 0.5f;
    iy = ((gy + 1.f) * p.H - 1.f) * 0.5f;
}
// out contiguous (B,C,H,W); one thread per output pixel, all channels
__global__ void affine_sample_fwd_kernel(AffP p, float* __restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long hw = (long)p.H * p.W;
    if (idx >= p.B * hw) return;
    const int n = idx / hw, r = idx - n * hw, h = r / p.W, w = r - h * p.W;
    float ix, iy;
    aff_pos(p, n, h, w, ix, iy);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    for (int c = 0; c < p.C; ++c) {
        const float* xc = p.x + n * p.sn + c * p.sc;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xi = x0 + (k & 1), yi = y0 + (k >> 1);
            if (xi < 0 || xi >= p.W || yi < 0 || yi >= p.H) continue;
            v += xc[yi * p.sh + xi * p.sw] * ((k & 1) ? tx : 1.f - tx) * ((k >> 1) ? ty : 1.f - ty);
        }
        out[((long)n * p.C + c) * hw + r] = v;
    }
}
// Gradient w.r.t. the IMAGE, as a gather (deterministic, no atomics): input pixel (y,x) collects from every output pixel whose
// bilinear footprint covers it.  Those lie around J^-1 (x,y): the pixel-space map is affine with Jacobian
// J = [[t00, t01 W/H], [t10 H/W, t11]], so a +-1 box around the pixel maps into a window of half-width
// |J^-1_00| + |J^-1_01| (+1 for rounding) output columns and |J^-1_10| + |J^-1_11| (+1) rows around the centre.
// dimg contiguous (B,C,H,W); dout contiguous (B,C,H,W).
__global__ void affine_sample_bwd_kernel(AffP p, const float* __restrict__ dout, float* __restrict__ dimg) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long hw = (long)p.H * p.W;
    if (idx >= p.B * hw) return;
    const int n = idx / hw, r = idx - n * hw, y = r / p.W, x = r - y * p.W;
    const float* t = p.theta + n * 6;
    const float j00 = t[0], j01 = t[1] * p.W / p.H, j10 = t[3] * p.H / p.W, j11 = t[4];
    const float det = j00 * j11 - j01 * j10;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (fabsf(det) > 1e-12f) {
        const float i00 = j11 / det, i01 = -j01 / det, i10 = -j10 / det, i11 = j00 / det;
        // pixel-space offset of the map: ix = j00 w + j01 h + bx  -> take it from output pixel (0,0)
        float bx, by;
        aff_pos(p, n, 0, 0, bx, by);
        const float cw = i00 * (x - bx) + i01 * (y - by), ch = i10 * (x - bx) + i11 * (y - by);
        const int rw = (int)ceilf(fabsf(i00) + fabsf(i01)) + 1, rh = (int)ceilf(fabsf(i10) + fabsf(i11)) + 1;
        const int w_lo = max(0, (int)floorf(cw) - rw), w_hi = min(p.W - 1, (int)ceilf(cw) + rw);
        const int h_lo = max(0, (int)floorf(ch) - rh), h_hi = min(p.H - 1, (int)ceilf(ch) + rh);
        for (int h = h_lo; h <= h_hi; ++h)
            for (int w = w_lo; w <= w_hi; ++w) {
                float ix, iy;
                aff_pos(p, n, h, w, ix, iy);
                const float fx = floorf(ix), fy = floorf(iy);
                const int dxi = x - (int)fx, dyi = y - (int)fy;
                if (dxi < 0 || dxi > 1 || dyi < 0 || dyi > 1) continue;
                const float tx = ix - fx, ty = iy - fy;
                const float wgt = (dxi ? tx : 1.f - tx) * (dyi ? ty : 1.f - ty);
                for (int c = 0; c < p.C && c < 4; ++c) acc[c] = fmaf(wgt, dout[((long)n * p.C + c) * hw + (long)h * p.W + w], acc[c]);
            }
    }
    for (int c = 0; c < p.C && c < 4; ++c) dimg[((long)n * p.C + c) * hw + r] = acc[c];
}
TATT_API int tatt_affine_sample_fwd(const float* x, long xsn, long xsc, long xsh, long xsw, const float* theta, float* out, int B,
                                    int C, int H, int W, hipStream_t st) {
    AffP p = {x, xsn, xsc, xsh, xsw, theta, B, C, H, W};
    hipLaunchKernelGGL(affine_sample_fwd_kernel, dim3(cdiv((long)B * H * W, 256)), dim3(256), 0, st, p, out);
    return LAUNCH_CHECK();
}
TATT_API int tatt_affine_sample_bwd(const float* theta, const float* dout, float* dimg, int B, int C, int H, int W,
                                    hipStream_t st) {
    if (C > 4) return 1;
    AffP p = {nullptr, 0, 0, 0, 0, theta, B, C, H, W};
    hipLaunchKernelGGL(affine_sample_bwd_kernel, dim3(cdiv((long)B * H * W, 256)), dim3(256), 0, st, p, dout, dimg);
    return LAUNCH_CHECK();
}
