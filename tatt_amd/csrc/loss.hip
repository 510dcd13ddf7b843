// ImageLoss of the TATT training step (gfx950): reference loss/image_loss.py:19-34 (ImageLoss.forward, gradient=True,
// loss_weight=[1, 1e-4]) with GradientPriorLoss.gradient_map (:50-58):
//
//   loss_b = w0 * mean_{c,h,w} (sr - hr)^2  +  w1 * mean_{c<3,h,w} | G(sr) - G(hr) |
//   G(x)[h,w] = sqrt( ((x[h,w+1] - x[h,w-1]) / 2)^2 + ((x[h-1,w] - x[h+1,w]) / 2)^2 + 1e-6 )        (zero padding)
//
// The reference evaluates this with ~40 element-wise torch kernels forward and as many backward on (B,4,2H,2W) images; here
// it is one forward kernel (per-sample loss, optionally followed by the batch mean * scale the training loop applies,
// interfaces/super_resolution.py:889-894) and one backward kernel that recomputes the four neighbouring gradient-map terms of
// every pixel instead of saving intermediate maps.  Both images are addressed by explicit strides: the SR output of the
// generator is NCHW-shaped over NHWC memory, the HR target is plain NCHW.  HBM-bound and tiny (0.8 MB per image pair).
#include "common.h"

struct ImgView {
    const float* p;
    long sn, sc, sh, sw;
};
struct LossP {
    ImgView sr, hr;
    int B, C, H, W;
    float w0, w1;
};

// central differences of channel plane x (pointer already offset to (n, c)) at (h, w), zero padded
__device__ __forceinline__ float grad_terms(const float* x, long sh, long sw, int h, int w, int H, int W, float& dx, float& dy) {
    const float r = (w + 1 < W) ? x[h * sh + (w + 1) * sw] : 0.f;
    const float l = (w > 0) ? x[h * sh + (w - 1) * sw] : 0.f;
    const float t = (h > 0) ? x[(h - 1) * sh + w * sw] : 0.f;
    const float b = (h + 1 < H) ? x[(h + 1) * sh + w * sw] : 0.f;
    dx = (r - l) * 0.5f;
    dy = (t - b) * 0.5f;
    return sqrtf(dx * dx + dy * dy + 1e-6f);
}

// one work-group per image; fp64 accumulation, deterministic tree
__global__ __launch_bounds__(1024) void image_loss_fwd_kernel(LossP p, float* __restrict__ loss) {
    __shared__ double red[2][16];
    const int n = blockIdx.x, t = threadIdx.x;
    const float* s = p.sr.p + n * p.sr.sn;
    const float* g = p.hr.p + n * p.hr.sn;
    const int gc = p.C < 3 ? p.C : 3;
    double am = 0.0, ag = 0.0;
    for (int i = t; i < p.H * p.W; i += 1024) {
        const int h = i / p.W, w = i - h * p.W;
        for (int c = 0; c < p.C; ++c) {
            const float d = s[c * p.sr.sc + h * p.sr.sh + w * p.sr.sw] - g[c * p.hr.sc + h * p.hr.sh + w * p.hr.sw];
            am += (double)(d * d);
        }
        for (int c = 0; c < gc; ++c) {
            float dx, dy;
            const float gs = grad_terms(s + c * p.sr.sc, p.sr.sh, p.sr.sw, h, w, p.H, p.W, dx, dy);
            const float gh = grad_terms(g + c * p.hr.sc, p.hr.sh, p.hr.sw, h, w, p.H, p.W, dx, dy);
            ag += (double)fabsf(gs - gh);
        }
    }
    am = wave_sum_d(am);
    ag = wave_sum_d(ag);
    if ((t & 63) == 0) { red[0][t >> 6] = am; red[1][t >> 6] = ag; }
    __syncthreads();
    if (t == 0) {
        double m = 0.0, q = 0.0;
        for (int k = 0; k < 16; ++k) { m += red[0][k]; q += red[1][k]; }
        const double hw = (double)p.H * p.W;
        loss[n] = (float)(p.w0 * m / (p.C * hw) + p.w1 * q / (gc * hw));
    }
}

__global__ void loss_mean_kernel(const float* __restrict__ loss, int B, float scale, float* __restrict__ out) {
    double a = 0.0;
    for (int i = threadIdx.x; i < B; i += 64) a += (double)loss[i];
    a = wave_sum_d(a);
    if (threadIdx.x == 0) out[0] = (float)(a / B * scale);
}

// d loss / d sr, one thread per pixel (all channels).  gper: per-sample upstream gradients (B) or NULL; gscalar: upstream
// gradient of the scaled batch mean (1 element) -- then every sample gets gscalar * scale_over_b.
__global__ __launch_bounds__(256) void image_loss_bwd_kernel(LossP p, const float* __restrict__ gper, const float* __restrict__ gscalar,
                                                             float scale_over_b, float* __restrict__ dsr) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long hw = (long)p.H * p.W;
    if (i >= p.B * hw) return;
    const int n = i / hw, r = i - n * hw;
    const int h = r / p.W, w = r - h * p.W;
    const float up = gper ? gper[n] : gscalar[0] * scale_over_b;
    const float* s = p.sr.p + n * p.sr.sn;
    const float* g = p.hr.p + n * p.hr.sn;
    const int gc = p.C < 3 ? p.C : 3;
    const float km = up * p.w0 * 2.f / (float)(p.C * hw);
    const float kg = up * p.w1 * 0.5f / (float)(gc * hw);
    for (int c = 0; c < p.C; ++c) {
        const long o = c * p.sr.sc + h * p.sr.sh + w * p.sr.sw;
        float v = km * (s[o] - g[c * p.hr.sc + h * p.hr.sh + w * p.hr.sw]);
        if (c < gc) {
            const float* sc = s + c * p.sr.sc;
            const float* hc = g + c * p.hr.sc;
            // pixel (h,w) is the RIGHT neighbour of (h,w-1), the LEFT one of (h,w+1), the TOP one of (h+1,w), the BOTTOM one of (h-1,w)
            float acc = 0.f;
            const int qh[4] = {h, h, h + 1, h - 1}, qw[4] = {w - 1, w + 1, w, w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (qh[k] < 0 || qh[k] >= p.H || qw[k] < 0 || qw[k] >= p.W) continue;
                float dx, dy, hx, hy;
                const float gs = grad_terms(sc, p.sr.sh, p.sr.sw, qh[k], qw[k], p.H, p.W, dx, dy);
                const float gh = grad_terms(hc, p.hr.sh, p.hr.sw, qh[k], qw[k], p.H, p.W, hx, hy);
                const float df = gs - gh;
                const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);          // torch: d|x| = sgn(x)
                const float term = (k < 2 ? dx : dy) / gs;
                acc += ((k & 1) ? -sg : sg) * term;
            }
            v += kg * acc;
        }
        dsr[n * p.sr.sn + o] = v;
    }
}

static LossP make_lossp(const float* sr, long s_n, long s_c, long s_h, long s_w, const float* hr, long h_n, long h_c, long h_h,
                        long h_w, int B, int C, int H, int W, float w0, float w1) {
    LossP p = {{sr, s_n, s_c, s_h, s_w}, {hr, h_n, h_c, h_h, h_w}, B, C, H, W, w0, w1};
    return p;
}

// loss[b] per sample; if loss_mean != NULL also loss_mean[0] = scale * mean_b loss[b]
TATT_API int tatt_image_loss_fwd(const float* sr, long s_n, long s_c, long s_h, long s_w, const float* hr, long h_n, long h_c,
                                 long h_h, long h_w, float* loss, float* loss_mean, float scale, int B, int C, int H, int W,
                                 float w0, float w1, hipStream_t st) {
    LossP p = make_lossp(sr, s_n, s_c, s_h, s_w, hr, h_n, h_c, h_h, h_w, B, C, H, W, w0, w1);
    hipLaunchKernelGGL(image_loss_fwd_kernel, dim3(B), dim3(1024), 0, st, p, loss);
    if (loss_mean) hipLaunchKernelGGL(loss_mean_kernel, dim3(1), dim3(64), 0, st, loss, B, scale, loss_mean);
    return LAUNCH_CHECK();
}

// dsr has the strides of sr.  Exactly one of gper (B per-sample gradients) / gscalar (gradient of the scaled mean) is non-NULL.
TATT_API int tatt_image_loss_bwd(const float* sr, long s_n, long s_c, long s_h, long s_w, const float* hr, long h_n, long h_c,
                                 long h_h, long h_w, const float* gper, const float* gscalar, float scale, float* dsr,
                                 int B, int C, int H, int W, float w0, float w1, hipStream_t st) {
    if ((gper == nullptr) == (gscalar == nullptr)) return 1;
    LossP p = make_lossp(sr, s_n, s_c, s_h, s_w, hr, h_n, h_c, h_h, h_w, B, C, H, W, w0, w1);
    const long total = (long)B * H * W;
    hipLaunchKernelGGL(image_loss_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, p, gper, gscalar, scale / B, dsr);
    return LAUNCH_CHECK();
}

// ---- SemanticLoss: the distillation loss between the student's and the teacher's text priors (reference
// loss/semantic_loss.py:21-38, used at interfaces/super_resolution.py:711):
//   L = mean |gt - pred|  +  mean (gt + e) (log(gt + e) - log(pred + e)),   e = 1e-20   (nn.KLDivLoss, reduction 'mean')
// One work-group (the priors are B x 26 x 37 values); backward w.r.t. pred only (the teacher is detached). ----
__global__ __launch_bounds__(1024) void semantic_loss_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt, long n,
                                                                 float* __restrict__ out) {
    __shared__ double red[16];
    double a = 0.0;
    for (long i = threadIdx.x; i < n; i += 1024) {
        const float p = pred[i], g = gt[i];
        const float ge = g + 1e-20f;
        a += (double)fabsf(g - p) + (double)(ge * (logf(ge) - logf(p + 1e-20f)));
    }
    a = wave_sum_d(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < 16; ++k) s += red[k];
        out[0] = (float)(s / (double)n);
    }
}
__global__ void semantic_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ gout,
                                         long n, float* __restrict__ dpred) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float p = pred[i], g = gt[i];
    const float d = g - p;
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    dpred[i] = gout[0] * (-sg - (g + 1e-20f) / (p + 1e-20f)) / (float)n;
}
TATT_API int tatt_semantic_loss_fwd(const float* pred, const float* gt, long n, float* out, hipStream_t st) {
    hipLaunchKernelGGL(semantic_loss_fwd_kernel, dim3(1), dim3(1024), 0, st, pred, gt, n, out);
    return LAUNCH_CHECK();
}
TATT_API int tatt_semantic_loss_bwd(const float* pred, const float* gt, const float* gout, long n, float* dpred, hipStream_t st) {
    hipLaunchKernelGGL(semantic_loss_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, pred, gt, gout, n, dpred);
    return LAUNCH_CHECK();
}

// ---- calculate_psnr (reference utils/ssim_psnr.py:9-15): 20 log10(255 / sqrt(mean((255 a - 255 b)^2))) over the first 3 channels;
// +inf when the images are identical.  Strided (B,C,H,W) views like the loss above. ----
__global__ __launch_bounds__(1024) void psnr_kernel(LossP p, float* __restrict__ out) {
    __shared__ double red[16];
    const int gc = p.C < 3 ? p.C : 3;
    const long hw = (long)p.H * p.W, total = (long)p.B * gc * hw;
    double a = 0.0;
    for (long i = threadIdx.x; i < total; i += 1024) {
        const int w = i % p.W; long r = i / p.W;
        const int h = r % p.H; r /= p.H;
        const int c = r % gc; const int n = r / gc;
        const float d = 255.f * p.sr.p[n * p.sr.sn + c * p.sr.sc + h * p.sr.sh + w * p.sr.sw] -
                        255.f * p.hr.p[n * p.hr.sn + c * p.hr.sc + h * p.hr.sh + w * p.hr.sw];
        a += (double)(d * d);
    }
    a = wave_sum_d(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < 16; ++k) s += red[k];
        const double mse = s / (double)total;
        out[0] = mse == 0.0 ? INFINITY : (float)(20.0 * log10(255.0 / sqrt(mse)));
    }
}
TATT_API int tatt_psnr(const float* a, long a_n, long a_c, long a_h, long a_w, const float* b, long b_n, long b_c, long b_h, long b_w,
                       float* out, int B, int C, int H, int W, hipStream_t st) {
    LossP p = make_lossp(a, a_n, a_c, a_h, a_w, b, b_n, b_c, b_h, b_w, B, C, H, W, 0.f, 0.f);
    hipLaunchKernelGGL(psnr_kernel, dim3(1), dim3(1024), 0, st, p, out);
    return LAUNCH_CHECK();
}
