// Streaming element-wise kernels of the TATT hot path (all HBM-bound, lanes along the contiguous axis).
#include "common.h"

#define EW_GRID(total) dim3(cdiv((total), 256)), dim3(256)

// ---- PReLU with ONE shared slope (reference nn.PReLU(): model/tsrn.py:598,173) ---------------------
__global__ void prelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ a,
                                 long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    y[i] = v >= 0.f ? v : a[0] * v;
}
TATT_API int tatt_prelu_fwd(const float* x, float* y, const float* alpha, long n, hipStream_t st) {
    hipLaunchKernelGGL(prelu_fwd_kernel, EW_GRID(n), 0, st, x, y, alpha, n);
    return LAUNCH_CHECK();
}
// dx = dy * (x>=0 ? 1 : a);  dalpha partial[block] = sum dy*x*[x<0]
__global__ void prelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                 const float* __restrict__ a, long n, float* __restrict__ part) {
    __shared__ double sh[4];
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0.0;                                   // the slope gradient is one cancelling sum over every element: fp64 partials
    if (i < n) {
        float v = x[i], g = dy[i];
        dx[i] = v >= 0.f ? g : a[0] * g;
        if (v < 0.f) s = (double)g * (double)v;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (float)((sh[0] + sh[1]) + (sh[2] + sh[3]));
}
// part: cdiv(n,256) floats (reduce with tatt_colsum(part, 1, G, 1, dalpha, ...))
TATT_API int tatt_prelu_bwd(const float* x, const float* dy, float* dx, const float* alpha, long n, float* part,
                            hipStream_t st) {
    hipLaunchKernelGGL(prelu_bwd_kernel, EW_GRID(n), 0, st, x, dy, dx, alpha, n, part);
    return LAUNCH_CHECK();
}

// ---- generic activation forward / backward ----------------------------------------------------------
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int act) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = apply_act(x[i], act);
}
TATT_API int tatt_act_fwd(const float* x, float* y, long n, int act, hipStream_t st) {
    hipLaunchKernelGGL(act_fwd_kernel, EW_GRID(n), 0, st, x, y, n, act);
    return LAUNCH_CHECK();
}
// from_output: ref holds y = act(u) (valid for relu / tanh); else ref holds the pre-activation u
__global__ void act_bwd_kernel(const float* __restrict__ ref, const float* __restrict__ dy, float* __restrict__ dx,
                               long n, int act, int from_output) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = ref[i], g;
    if (from_output) {
        if (act == ACT_RELU) g = r > 0.f ? 1.f : 0.f;
        else if (act == ACT_TANH) g = 1.f - r * r;
        else g = 1.f;
    } else g = act_grad(r, act);
    dx[i] = dy[i] * g;
}
TATT_API int tatt_act_bwd(const float* ref, const float* dy, float* dx, long n, int act, int from_output,
                          hipStream_t st) {
    hipLaunchKernelGGL(act_bwd_kernel, EW_GRID(n), 0, st, ref, dy, dx, n, act, from_output);
    return LAUNCH_CHECK();
}

// ---- y = alpha*a + beta*b  (b may be null) ------------------------------------------------------------
__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                             float alpha, float beta, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = alpha * a[i];
    if (b) v += beta * b[i];
    y[i] = v;
}
TATT_API int tatt_axpby(const float* a, const float* b, float* y, float alpha, float beta, long n, hipStream_t st) {
    hipLaunchKernelGGL(axpby_kernel, EW_GRID(n), 0, st, a, b, y, alpha, beta, n);
    return LAUNCH_CHECK();
}
// y = ((s0 + s1) + s2) + ... over n <= 8 equally shaped tensors, one launch (gradient contributions of a tensor with several
// consumers, summed left to right like a chain of binary adds)
struct AddNP { const float* s[8]; int n; };
__global__ void add_n_kernel(AddNP p, float* __restrict__ y, long numel) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= numel) return;
    if (i + 4 <= numel) {
        f32x4 v = *reinterpret_cast<const f32x4*>(p.s[0] + i);
        for (int k = 1; k < p.n; ++k) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(p.s[k] + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += u[e];
        }
        *reinterpret_cast<f32x4*>(y + i) = v;
    } else {
        for (long j = i; j < numel; ++j) {
            float v = p.s[0][j];
            for (int k = 1; k < p.n; ++k) v += p.s[k][j];
            y[j] = v;
        }
    }
}
TATT_API int tatt_add_n(const float* const* srcs, int n, float* y, long numel, hipStream_t st) {
    if (n < 1 || n > 8) return 1;
    AddNP p;
    for (int k = 0; k < 8; ++k) p.s[k] = k < n ? srcs[k] : nullptr;
    p.n = n;
    for (int k = 0; k < n; ++k) if (((uintptr_t)srcs[k]) & 15) return 2;
    if (((uintptr_t)y) & 15) return 2;
    hipLaunchKernelGGL(add_n_kernel, dim3(cdiv(cdiv(numel, 4), 256)), dim3(256), 0, st, p, y, numel);
    return LAUNCH_CHECK();
}
// y[m, :] = a[m, :] + b[(m % period), :]   (row-broadcast add: positional / query embeddings)
__global__ void add_rowbcast_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                    long rows, int C, long period) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    long m = i / C; int c = i % C;
    y[i] = a[i] + b[(m % period) * C + c];
}
TATT_API int tatt_add_rowbcast(const float* a, const float* b, float* y, long rows, int C, long period,
                               hipStream_t st) {
    hipLaunchKernelGGL(add_rowbcast_kernel, EW_GRID(rows * C), 0, st, a, b, y, rows, C, period);
    return LAUNCH_CHECK();
}

// ---- PixelShuffle(2) + activation on NHWC maps --------------------------------------------------------
// out[b, 2h+i, 2w+j, c] = act(in[b, h, w, 4c+2i+j])    (reference nn.PixelShuffle + mish: model/tsrn.py:1045-1052)
__global__ void pixel_shuffle_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                                         int C, int act) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // enumerates OUTPUT (b, oh, ow, c)
    long total = (long)B * 4 * H * W * C;
    if (idx >= total) return;
    int c = idx % C; long r = idx / C;
    int ow = r % (2 * W); r /= (2 * W);
    int oh = r % (2 * H); int b = r / (2 * H);
    int h = oh >> 1, i = oh & 1, w = ow >> 1, j = ow & 1;
    float v = in[(((long)b * H + h) * W + w) * (4 * C) + 4 * c + 2 * i + j];
    out[idx] = apply_act(v, act);
}
TATT_API int tatt_pixel_shuffle_fwd(const float* in, float* out, int B, int H, int W, int C, int act,
                                    hipStream_t st) {
    long total = (long)B * 4 * H * W * C;
    hipLaunchKernelGGL(pixel_shuffle_fwd_kernel, EW_GRID(total), 0, st, in, out, B, H, W, C, act);
    return LAUNCH_CHECK();
}
// din[b,h,w,4c+2i+j] = dout[b,2h+i,2w+j,c] * act'(in[...])
__global__ void pixel_shuffle_bwd_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                         float* __restrict__ din, int B, int H, int W, int C, int act) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // enumerates INPUT (b, h, w, ch)
    long total = (long)B * H * W * 4 * C;
    if (idx >= total) return;
    int ch = idx % (4 * C); long r = idx / (4 * C);
    int w = r % W; r /= W;
    int h = r % H; int b = r / H;
    int c = ch >> 2, i = (ch >> 1) & 1, j = ch & 1;
    float g = dout[(((long)b * 2 * H + 2 * h + i) * (2 * W) + 2 * w + j) * C + c];
    din[idx] = g * act_grad(in[idx], act);
}
TATT_API int tatt_pixel_shuffle_bwd(const float* in, const float* dout, float* din, int B, int H, int W, int C,
                                    int act, hipStream_t st) {
    long total = (long)B * H * W * 4 * C;
    hipLaunchKernelGGL(pixel_shuffle_bwd_kernel, EW_GRID(total), 0, st, in, dout, din, B, H, W, C, act);
    return LAUNCH_CHECK();
}

// ---- nn.MaxPool2d(kernel (kh,kw), stride (sh,sw), padding (ph,pw)) on NHWC maps (reference model/stn_head.py:36-44: 2x2/2;
// model/crnn/crnn.py:57-69: 2x2/2 and the (2,2),(2,1),(0,1) pools whose windows overlap along W) -------------------------------
struct PoolP { int B, H, W, C, kh, kw, sh, sw, ph, pw, Ho, Wo; };
// first maximum of window (oh, ow) in (i, j) scan order (strict '>' like ATen); padding cells never win
__device__ __forceinline__ float pool_window(const float* __restrict__ in, const PoolP& p, int b, int oh, int ow, int c, int& ai, int& aj) {
    float m = -INFINITY;
    ai = -1; aj = -1;
    for (int i = 0; i < p.kh; ++i) {
        const int h = oh * p.sh - p.ph + i;
        if (h < 0 || h >= p.H) continue;
        for (int j = 0; j < p.kw; ++j) {
            const int w = ow * p.sw - p.pw + j;
            if (w < 0 || w >= p.W) continue;
            const float v = in[(((long)b * p.H + h) * p.W + w) * p.C + c];
            if (v > m || ai < 0) { m = v; ai = h; aj = w; }
        }
    }
    return m;
}
__global__ void maxpool_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, PoolP p) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)p.B * p.Ho * p.Wo * p.C;
    if (idx >= total) return;
    int c = idx % p.C; long r = idx / p.C;
    int ow = r % p.Wo; r /= p.Wo;
    int oh = r % p.Ho; int b = r / p.Ho;
    int ai, aj;
    out[idx] = pool_window(in, p, b, oh, ow, c, ai, aj);
}
static PoolP make_poolp(int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw) {
    PoolP p = {B, H, W, C, kh, kw, sh, sw, ph, pw, (H + 2 * ph - kh) / sh + 1, (W + 2 * pw - kw) / sw + 1};
    return p;
}
TATT_API int tatt_maxpool_fwd(const float* in, float* out, int B, int H, int W, int C, int kh, int kw, int sh, int sw,
                              int ph, int pw, hipStream_t st) {
    PoolP p = make_poolp(B, H, W, C, kh, kw, sh, sw, ph, pw);
    long total = (long)B * p.Ho * p.Wo * C;
    hipLaunchKernelGGL(maxpool_fwd_kernel, EW_GRID(total), 0, st, in, out, p);
    return LAUNCH_CHECK();
}
// one thread per INPUT element: it collects the gradient of every window whose (first) maximum it is -- windows may overlap
__global__ void maxpool_bwd_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                   float* __restrict__ din, PoolP p) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)p.B * p.H * p.W * p.C;
    if (idx >= total) return;
    int c = idx % p.C; long r = idx / p.C;
    int w = r % p.W; r /= p.W;
    int h = r % p.H; int b = r / p.H;
    // windows containing (h, w): oh*sh - ph <= h < oh*sh - ph + kh
    int oh0 = h + p.ph - p.kh + 1; oh0 = oh0 <= 0 ? 0 : (oh0 + p.sh - 1) / p.sh;
    int ow0 = w + p.pw - p.kw + 1; ow0 = ow0 <= 0 ? 0 : (ow0 + p.sw - 1) / p.sw;
    int oh1 = (h + p.ph) / p.sh; if (oh1 > p.Ho - 1) oh1 = p.Ho - 1;
    int ow1 = (w + p.pw) / p.sw; if (ow1 > p.Wo - 1) ow1 = p.Wo - 1;
    float g = 0.f;
    for (int oh = oh0; oh <= oh1; ++oh)
        for (int ow = ow0; ow <= ow1; ++ow) {
            int ai, aj;
            pool_window(in, p, b, oh, ow, c, ai, aj);
            if (ai == h && aj == w) g += dout[(((long)b * p.Ho + oh) * p.Wo + ow) * p.C + c];
        }
    din[idx] = g;
}
TATT_API int tatt_maxpool_bwd(const float* in, const float* dout, float* din, int B, int H, int W, int C, int kh,
                              int kw, int sh, int sw, int ph, int pw, hipStream_t st) {
    PoolP p = make_poolp(B, H, W, C, kh, kw, sh, sw, ph, pw);
    long total = (long)B * H * W * C;
    hipLaunchKernelGGL(maxpool_bwd_kernel, EW_GRID(total), 0, st, in, dout, din, p);
    return LAUNCH_CHECK();
}

// ---- Dropout (reference nn.Dropout(0.1): model/transformer_v2.py:27,456,461-462,789,795-797) ----------
// y = keep ? x/(1-p) : 0, keep drawn from a counter-based hash of (seed word in device memory, site, index):
// forward and backward regenerate the same mask; the same kernel serves both.
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float p,
                               const unsigned long long* __restrict__ seed, unsigned site) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t th = dropout_thresh(p);
    y[i] = dropout_keep(seed[0], site, (uint64_t)i, th) ? x[i] * (1.f / (1.f - p)) : 0.f;
}
TATT_API int tatt_dropout(const float* x, float* y, long n, float p, const unsigned long long* seed, unsigned site,
                          hipStream_t st) {
    hipLaunchKernelGGL(dropout_kernel, EW_GRID(n), 0, st, x, y, n, p, seed, site);
    return LAUNCH_CHECK();
}
// seed word += odd constant; `snap` (optional) receives the new value: a training forward bumps the device-resident word once and
// hands every dropout site of THAT forward (and of its backward, whenever it runs) the private snapshot.
__global__ void bump_seed_kernel(unsigned long long* seed, unsigned long long* snap) {
    const unsigned long long v = seed[0] + 0x632BE59BD9B4E019ull;
    seed[0] = v;
    if (snap) snap[0] = v;
}
TATT_API int tatt_bump_seed(unsigned long long* seed, unsigned long long* snap, hipStream_t st) {
    hipLaunchKernelGGL(bump_seed_kernel, dim3(1), dim3(1), 0, st, seed, snap);
    return LAUNCH_CHECK();
}

// out[0] = the 100 MHz wall clock when this point of the stream is reached: a timeline that is valid INSIDE a replayed hipGraph
// (rocprofv3 serialises graph nodes; this does not).  Measurement tooling only (tatt_amd.functional.stamp).
__global__ void stamp_kernel(unsigned long long* out) { out[0] = wall_clock64(); }
TATT_API int tatt_stamp(unsigned long long* out, hipStream_t st) {
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, st, out);
    return LAUNCH_CHECK();
}

// ---- generic 4-D strided copy (layout changes: NCHW<->NHWC, parameter gathers) ------------------------
__global__ void copy4d_kernel(const float* __restrict__ src, float* __restrict__ dst, int n0, int n1, int n2, int n3,
                              long s0, long s1, long s2, long s3, long d0, long d1, long d2, long d3, float beta) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)n0 * n1 * n2 * n3;
    if (idx >= total) return;
    int i3 = idx % n3; long r = idx / n3;
    int i2 = r % n2; r /= n2;
    int i1 = r % n1; int i0 = r / n1;
    float v = src[i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3];
    long o = i0 * d0 + i1 * d1 + i2 * d2 + i3 * d3;
    dst[o] = beta != 0.f ? v + beta * dst[o] : v;
}
// the LAST index (n3) is the fastest-varying thread index: pick it along the contiguous axis of dst
TATT_API int tatt_copy4d(const float* src, float* dst, int n0, int n1, int n2, int n3, long s0, long s1, long s2,
                         long s3, long d0, long d1, long d2, long d3, float beta, hipStream_t st) {
    long total = (long)n0 * n1 * n2 * n3;
    hipLaunchKernelGGL(copy4d_kernel, EW_GRID(total), 0, st, src, dst, n0, n1, n2, n3, s0, s1, s2, s3, d0, d1, d2, d3,
                       beta);
    return LAUNCH_CHECK();
}

// ---- optimiser: global-norm clip + Adam on flat parameter/gradient buffers ----------------------------
// (reference clip_grad_norm_(0.25) + Adam(lr, betas=(0.5,0.999)): interfaces/super_resolution.py:1083-1085,
//  interfaces/base.py:527).  Everything that changes from step to step (norm, step count) is read from
// DEVICE memory so that the captured hipGraph of a training step replays correctly.
__global__ void sumsq_stage1(const float* __restrict__ g, long n, double* __restrict__ part) {
    __shared__ double sh[4];
    double s = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        double v = (double)g[i];
        s += v * v;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ void sumsq_stage2(const double* __restrict__ part, int G, float* __restrict__ out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < G; i += 256) s += part[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)sqrt((sh[0] + sh[1]) + (sh[2] + sh[3]));
}
// out[0] = ||g||_2 ; ws >= 1024 doubles
TATT_API int tatt_l2norm(const float* g, long n, float* out, double* ws, hipStream_t st) {
    int G = (int)((n + 255) / 256);
    if (G > 1024) G = 1024;
    if (G < 1) G = 1;
    hipLaunchKernelGGL(sumsq_stage1, dim3(G), dim3(256), 0, st, g, n, ws);
    hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(256), 0, st, ws, G, out);
    return LAUNCH_CHECK();
}
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float coef, float b1, float b2, float a1, float rs2,
                                         float eps) {
    const float gi = g * coef;
    const float mi = b1 * m + (1.f - b1) * gi;
    const float vi = b2 * v + (1.f - b2) * gi * gi;
    m = mi; v = vi;
    p -= a1 * mi / (sqrtf(vi) * rs2 + eps);
}
// 16 bytes per lane and array: the four flat buffers start 256-byte aligned and every segment handed in starts on a 64-byte
// boundary (FlatParams.ALIGN); a ragged tail (n % 4) is finished element-wise by the last thread.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                            const float* __restrict__ gnorm, float max_norm, float gscale,
                            const long long* __restrict__ step) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n4 = n >> 2;
    if (i > n4) return;
    const double t = (double)step[0];
    const float bc1 = (float)(1.0 - pow((double)b1, t)), bc2 = (float)(1.0 - pow((double)b2, t));
    float coef = gscale;
    if (max_norm > 0.f) { float c = max_norm / (gnorm[0] * gscale + 1e-6f); coef *= c < 1.f ? c : 1.f; }
    const float a1 = lr / bc1, rs2 = 1.f / sqrtf(bc2);
    if (i < n4) {
        float4 pp = ((float4*)p)[i], mm = ((float4*)m)[i], vv = ((float4*)v)[i];
        const float4 gg = ((const float4*)g)[i];
        adam_one(pp.x, gg.x, mm.x, vv.x, coef, b1, b2, a1, rs2, eps);
        adam_one(pp.y, gg.y, mm.y, vv.y, coef, b1, b2, a1, rs2, eps);
        adam_one(pp.z, gg.z, mm.z, vv.z, coef, b1, b2, a1, rs2, eps);
        adam_one(pp.w, gg.w, mm.w, vv.w, coef, b1, b2, a1, rs2, eps);
        ((float4*)p)[i] = pp; ((float4*)m)[i] = mm; ((float4*)v)[i] = vv;
    } else {
        for (long k = n4 << 2; k < n; ++k) adam_one(p[k], g[k], m[k], v[k], coef, b1, b2, a1, rs2, eps);
    }
}
// gnorm: device float = ||g||_2 of the UNSCALED buffer the clip refers to (a parameter group of its own may pass max_norm = 0:
// no clipping); gscale: constant pre-scale (1/world_size after a sum all-reduce); step: device int64 holding the 1-based step count.
TATT_API int tatt_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                            float eps, const float* gnorm, float max_norm, float gscale, const long long* step,
                            hipStream_t st) {
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return 1001;      // segments must be 16-byte aligned
    hipLaunchKernelGGL(adam_kernel, EW_GRID((n >> 2) + 1), 0, st, p, g, m, v, n, lr, b1, b2, eps, gnorm, max_norm, gscale, step);
    return LAUNCH_CHECK();
}

// ---- sticky error word + guard (see common.h) ---------------------------------------------------------------------------------------
#include <mutex>
static std::mutex g_sticky_mu;
static unsigned* g_sticky[64] = {};
unsigned* tatt_sticky_ptr() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_sticky_mu);
    return g_sticky[dev];
}
// word: one zero-initialised 32-bit word in the CURRENT device's memory that outlives every launch (NULL: unregister)
TATT_API int tatt_set_sticky(unsigned* word) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1;
    std::lock_guard<std::mutex> lk(g_sticky_mu);
    g_sticky[dev] = word;
    return 0;
}
__global__ void sync_guard_kernel(const unsigned* word) {
    if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) __builtin_trap();
}
// One-thread launch that TRAPS (the process dies with a GPU exception) if a launch that synchronises its work-groups in flight has given
// up waiting since the word was registered: issued in front of the optimiser, it keeps invalid gradients from ever reaching the weights.
// Returns 0 without launching when no word is registered.
TATT_API int tatt_sync_guard(hipStream_t st) {
    unsigned* w = tatt_sticky_ptr();
    if (!w) return 0;
    hipLaunchKernelGGL(sync_guard_kernel, dim3(1), dim3(1), 0, st, w);
    return LAUNCH_CHECK();
}

// ---- a resident "CU holder" (diagnostic) -----------------------------------------------------------------------------------------------
// `groups` work-groups of 512 threads that do nothing but stay resident for `ticks` of the 100 MHz wall clock: what a collective's channel
// kernels look like to the launches that need their whole grid co-resident (RCCL keeps one work-group per channel on a CU for the length
// of the collective).  lds_bytes of dynamic LDS per group are touched so that the allocation is real.  The tests launch it on a second
// stream beside the persistent query-GRU chains and the STN-head launches.
__global__ __launch_bounds__(512) void cu_holder_kernel(long ticks, unsigned* sink, int lds_words) {
    extern __shared__ unsigned hold_lds[];
    for (int i = threadIdx.x; i < lds_words; i += 512) hold_lds[i] = i;
    __syncthreads();
    const long t0 = wall_clock64();
    unsigned acc = 0;
    while (wall_clock64() - t0 < ticks) {
        acc += lds_words ? hold_lds[(threadIdx.x + acc) % lds_words] : 1u;
        __builtin_amdgcn_s_sleep(8);
    }
    if (acc == 0xffffffffu && sink) sink[0] = acc;            // (keeps the loop alive for the compiler)
}
TATT_API int tatt_cu_holder(int groups, long ticks, int lds_bytes, unsigned* sink, hipStream_t st) {
    if (groups < 1 || groups > 1024 || ticks < 0 || ticks > 100000000L || lds_bytes < 0 || lds_bytes > 65536) return 1;
    hipLaunchKernelGGL(cu_holder_kernel, dim3(groups), dim3(512), lds_bytes, st, ticks, sink, lds_bytes / 4);
    return LAUNCH_CHECK();
}

// ---- gradient gather: one bucket of the flat gradient buffer <- the parameters' fresh gradient tensors ----------------------------------
// (replaces a fill + a multi-tensor ATen copy per bucket: one launch, descriptor table in the kernel arguments -- fixed at hipGraph
// capture like the tables of the split-K reducer).  An entry without a source writes zeros (a parameter that received no gradient).
#define GG_MAX 112
struct GGEntry { const float* src; long off; int n; int block0; };
struct GGTable { GGEntry e[GG_MAX]; float* dst; int n; };
__global__ __launch_bounds__(256) void gather_grads_kernel(GGTable t) {
    int k = 0;
    while (k + 1 < t.n && (int)blockIdx.x >= t.e[k + 1].block0) ++k;
    const GGEntry& e = t.e[k];
    const long i = ((long)((int)blockIdx.x - e.block0) * 256 + threadIdx.x) * 4;      // (an entry of n < 2^31 elements: 4 i can pass 2^31)
    if (i >= e.n) return;
    float* d = t.dst + e.off + i;
    const bool vec = i + 4 <= e.n && (((uintptr_t)d | (e.src ? (uintptr_t)(e.src + i) : 0)) & 15) == 0;
    if (vec) {
        *reinterpret_cast<f32x4*>(d) = e.src ? *reinterpret_cast<const f32x4*>(e.src + i) : (f32x4){0.f, 0.f, 0.f, 0.f};
    } else {
        for (int j = 0; j < 4 && i + j < e.n; ++j) d[j] = e.src ? e.src[i + j] : 0.f;
    }
}
// srcs: HOST array of `count` device pointers (NULL: zeros), offs / ns: HOST arrays (element offset into dst, element count)
TATT_API int tatt_gather_grads(const float* const* srcs, const long* offs, const int* ns, int count, float* dst, hipStream_t st) {
    for (int base = 0; base < count; base += GG_MAX) {
        GGTable t;
        t.dst = dst;
        t.n = count - base < GG_MAX ? count - base : GG_MAX;
        int blocks = 0;
        for (int k = 0; k < t.n; ++k) {
            if (ns[base + k] < 0) return 1;
            t.e[k] = {srcs[base + k], offs[base + k], ns[base + k], blocks};
            blocks += cdiv(ns[base + k], 1024);
        }
        if (blocks) hipLaunchKernelGGL(gather_grads_kernel, dim3(blocks), dim3(256), 0, st, t);
    }
    return LAUNCH_CHECK();
}
// v[i] += 1 for n 64-bit counters (the optimiser's step count, the BatchNorms' num_batches_tracked group)
__global__ void inc_i64_kernel(long long* v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] += 1;
}
TATT_API int tatt_inc_i64(long long* v, int n, hipStream_t st) {
    if (n < 1) return 0;
    hipLaunchKernelGGL(inc_i64_kernel, dim3(cdiv(n, 64)), dim3(64), 0, st, v, n);
    return LAUNCH_CHECK();
}
// y[i] = 0 (n floats): the zero time slots of the query GRU's state buffer and other small clears, without an ATen fill
__global__ void zero_f32_kernel(float* y, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = 0.f;
}
TATT_API int tatt_zero_f32(float* y, long n, hipStream_t st) {
    if (n < 1) return 0;
    hipLaunchKernelGGL(zero_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, y, n);
    return LAUNCH_CHECK();
}
