// BatchNorm (train / eval, with fused activation), LayerNorm (with fused residual add) and column
// reductions over token-major (rows x C) activations.  Memory-bound streaming kernels: rows are read
// with lanes along the contiguous channel axis; statistics are accumulated in fp64 so that the
// E[x^2]-E[x]^2 form is safe.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// column sums:  out[c] (+)= scale * sum_m X[m*ld + c]      (two stages, deterministic)
// ---------------------------------------------------------------------------------------------
#define CS_MAXG 256   // stage-1 blocks (partials per column)

// stage 1: block g sums rows [g*rpb, (g+1)*rpb); thread = (channel lane tx = t&63, row lane ty = t>>6)
template <typename T>
__global__ __launch_bounds__(256) void colsum_stage1(const T* __restrict__ X, long ld, int M, int C, int rpb,
                                                     double* __restrict__ part) {
    __shared__ double sh[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int m_begin = blockIdx.x * rpb;
    const int m_end = min(M, m_begin + rpb);
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + tx;
        double s0 = 0.0;
        if (c < C && m_begin + ty < m_end)
            s0 = sum_strided<T, 8>(X + (long)(m_begin + ty) * ld + c, (m_end - m_begin - ty + 3) / 4, 4 * ld);
        sh[ty][tx] = s0;
        __syncthreads();
        if (ty == 0 && c < C) part[(long)blockIdx.x * C + c] = (sh[0][tx] + sh[1][tx]) + (sh[2][tx] + sh[3][tx]);
        __syncthreads();
    }
}
// stage 2: out[c] = scale * sum_g part[g*ldp + c] (+ beta*out[c]); block = 64 columns x S2_L g-lanes.  The kernel is one
// dependent chain of loads per thread: 16 lanes of 16 partials (2 round trips to L2) instead of 4 lanes of 64 (8 round trips).
#define S2_L 16
__device__ __forceinline__ double s2_combine(double (*sh)[64], int tx) {
    double s = 0.0;
#pragma unroll
    for (int l = 0; l < S2_L; l += 4) s += (sh[l][tx] + sh[l + 1][tx]) + (sh[l + 2][tx] + sh[l + 3][tx]);
    return s;
}
__global__ __launch_bounds__(64 * S2_L) void colsum_stage2(const double* __restrict__ part, int G, int C, long ldp,
                                                           float* __restrict__ out, float scale, float beta) {
    __shared__ double sh[S2_L][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    double s = 0.0;
    if (c < C && ty < G) s = sum_strided<double, 8>(part + (long)ty * ldp + c, (G - ty + S2_L - 1) / S2_L, S2_L * ldp);
    sh[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) {
        float v = (float)(s2_combine(sh, tx) * scale);
        out[c] = beta != 0.f ? v + beta * out[c] : v;
    }
}
// ---- 16-byte variants (C % 4 == 0, 16-byte aligned rows) -------------------------------------------------------------------
// The scalar kernels above keep 4-8 four-byte loads in flight per thread and ran at 0.7-1 TB/s (a reduction over 49,152 x 64
// floats took 13-18 us: ~12 dependent round trips to HBM).  Here a thread owns 4 consecutive channels, 256 / (C/4) rows are read
// in parallel and a thread issues up to V4_U sixteen-byte loads before it consumes the first one.
#define V4_U 12
static inline bool v4_ok(const void* p, long ld, int C) {
    return C % 4 == 0 && C <= 1024 && ld % 4 == 0 && (((uintptr_t)p) & 15) == 0 && (256 % (C / 4) == 0 || (C / 4) % 256 == 0);
}
// rows [m_begin, m_end) of X, channels 4*cl .. 4*cl+3, rows m_begin + rl, + rpar, ...: f(row-value float4) for every row
template <typename F>
__device__ __forceinline__ void v4_rows(const float* __restrict__ X, long ld, int m_begin, int m_end, int rl, int rpar, int c4,
                                        F&& f) {
    int m = m_begin + rl;
    for (; m + (V4_U - 1) * rpar < m_end; m += V4_U * rpar) {
        f32x4 v[V4_U];
#pragma unroll
        for (int u = 0; u < V4_U; ++u) v[u] = *reinterpret_cast<const f32x4*>(X + (long)(m + u * rpar) * ld + c4);
#pragma unroll
        for (int u = 0; u < V4_U; ++u) f(v[u]);
    }
    for (; m < m_end; m += rpar) f(*reinterpret_cast<const f32x4*>(X + (long)m * ld + c4));
}
// same over two row-aligned matrices
template <typename F>
__device__ __forceinline__ void v4_rows2(const float* __restrict__ X, long ldx, const float* __restrict__ Y, long ldy,
                                         int m_begin, int m_end, int rl, int rpar, int c4, F&& f) {
    constexpr int U = V4_U / 2;
    int m = m_begin + rl;
    for (; m + (U - 1) * rpar < m_end; m += U * rpar) {
        f32x4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = *reinterpret_cast<const f32x4*>(X + (long)(m + u * rpar) * ldx + c4);
            b[u] = *reinterpret_cast<const f32x4*>(Y + (long)(m + u * rpar) * ldy + c4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) f(a[u], b[u]);
    }
    for (; m < m_end; m += rpar)
        f(*reinterpret_cast<const f32x4*>(X + (long)m * ldx + c4), *reinterpret_cast<const f32x4*>(Y + (long)m * ldy + c4));
}
// combine the row lanes of a block: acc[NQ][4] per thread -> part[(g*NQ + q)*C + c]; sh: NQ*1024 doubles
template <int NQ>
__device__ __forceinline__ void v4_block_partials(const double (&acc)[NQ][4], int C, int cv, int cl, int rl, int rpar,
                                                  int c_base, double* __restrict__ part, double* sh) {
    const int t = threadIdx.x;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) sh[q * 1024 + rl * (4 * cv) + 4 * cl + e] = acc[q][e];
    __syncthreads();
    const int width = 4 * cv;                   // channels handled in this pass (<= 1024)
    for (int c = t; c < width; c += 256) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            double s = 0.0;
            for (int r = 0; r < rpar; ++r) s += sh[q * 1024 + r * width + c];
            part[((long)blockIdx.x * NQ + q) * C + c_base + c] = s;
        }
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void colsum_stage1_v4(const float* __restrict__ X, long ld, int M, int C, int rpb,
                                                        double* __restrict__ part) {
    __shared__ double sh[1024];
    const int t = threadIdx.x;
    const int cv = min(C >> 2, 256), rpar = 256 / cv, cl = t % cv, rl = t / cv;
    const int m_begin = blockIdx.x * rpb, m_end = min(M, m_begin + rpb);
    for (int c0 = 0; c0 < C; c0 += 4 * cv) {
        double acc[1][4] = {{0.0, 0.0, 0.0, 0.0}};
        v4_rows(X, ld, m_begin, m_end, rl, rpar, c0 + 4 * cl, [&](const f32x4& v) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0][e] += (double)v[e];
        });
        v4_block_partials<1>(acc, C, cv, cl, rl, rpar, c0, part, sh);
    }
}
static inline int cs_groups(int M) { int g = cdiv(M, 64); return g > CS_MAXG ? CS_MAXG : (g < 1 ? 1 : g); }
// ws: min(ceil(M/64),256)*C doubles
TATT_API int tatt_colsum(const float* X, long ld, int M, int C, float* out, float scale, float beta,
                         double* ws, hipStream_t st) {
    int G = cs_groups(M);
    int rpb = cdiv(M, G);
    if (v4_ok(X, ld, C)) hipLaunchKernelGGL(colsum_stage1_v4, dim3(G), dim3(256), 0, st, X, ld, M, C, rpb, ws);
    else hipLaunchKernelGGL((colsum_stage1<float>), dim3(G), dim3(256), 0, st, X, ld, M, C, rpb, ws);
    hipLaunchKernelGGL(colsum_stage2, dim3(cdiv(C, 64)), dim3(64 * S2_L), 0, st, ws, G, C, (long)C, out, scale, beta);
    return LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// BatchNorm  (reference nn.BatchNorm2d / BatchNorm1d: model/tsrn.py:878,886,613; model/stn_head.py:19,51)
// ---------------------------------------------------------------------------------------------
// stage 1: per-block partial sums (sum, sumsq) per channel in fp64.  part[g][2][C]
__global__ void bn_stats_stage1(const float* __restrict__ X, long ld, int M, int C, int rows_per_block,
                                double* __restrict__ part) {
    __shared__ double sh[2][256];
    const int cpb = C < 256 ? C : 256;           // channels per pass (C is a power of two <= 512 here)
    const int rpar = 256 / cpb;                  // rows handled in parallel
    const int t = threadIdx.x;
    const int cl = t % cpb, rl = t / cpb;
    const int m_begin = blockIdx.x * rows_per_block;
    const int m_end = min(M, m_begin + rows_per_block);
    for (int c0 = 0; c0 < C; c0 += cpb) {
        const int c = c0 + cl;
        double s = 0.0, q = 0.0;
        if (rl < rpar && c < C) {
            int m = m_begin + rl;
            for (; m + 3 * rpar < m_end; m += 4 * rpar) {
                float v0 = X[(long)m * ld + c], v1 = X[(long)(m + rpar) * ld + c];
                float v2 = X[(long)(m + 2 * rpar) * ld + c], v3 = X[(long)(m + 3 * rpar) * ld + c];
                s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
                q += ((double)v0 * v0 + (double)v1 * v1) + ((double)v2 * v2 + (double)v3 * v3);
            }
            for (; m < m_end; m += rpar) { double v = (double)X[(long)m * ld + c]; s += v; q += v * v; }
        }
        sh[0][t] = s; sh[1][t] = q;
        __syncthreads();
        if (t < cpb && c < C) {
            double ss = 0.0, qq = 0.0;
            for (int r = 0; r < rpar; ++r) { ss += sh[0][r * cpb + t]; qq += sh[1][r * cpb + t]; }
            part[((long)blockIdx.x * 2 + 0) * C + c] = ss;
            part[((long)blockIdx.x * 2 + 1) * C + c] = qq;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void bn_stats_stage1_v4(const float* __restrict__ X, long ld, int M, int C, int rpb,
                                                          double* __restrict__ part) {
    __shared__ double sh[2 * 1024];
    const int t = threadIdx.x;
    const int cv = min(C >> 2, 256), rpar = 256 / cv, cl = t % cv, rl = t / cv;
    const int m_begin = blockIdx.x * rpb, m_end = min(M, m_begin + rpb);
    for (int c0 = 0; c0 < C; c0 += 4 * cv) {
        double acc[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
        v4_rows(X, ld, m_begin, m_end, rl, rpar, c0 + 4 * cl, [&](const f32x4& v) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double d = (double)v[e]; acc[0][e] += d; acc[1][e] += d * d; }
        });
        v4_block_partials<2>(acc, C, cv, cl, rl, rpar, c0, part, sh);
    }
}
// stage 2: mean / rstd (biased var) + running-stat update (unbiased var, momentum) -- one thread per channel
__global__ __launch_bounds__(64 * S2_L) void bn_stats_stage2(const double* __restrict__ part, int G, int C, int M, float eps,
                                                             float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                             float* __restrict__ running_mean, float* __restrict__ running_var,
                                                             const float* __restrict__ gamma = nullptr,
                                                             const float* __restrict__ beta = nullptr,
                                                             float* __restrict__ scale = nullptr, float* __restrict__ shift = nullptr) {
    __shared__ double sh[2][S2_L][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    double s = 0.0, q = 0.0;
    if (c < C && ty < G) {
        s = sum_strided<double, 8>(part + ((long)ty * 2 + 0) * C + c, (G - ty + S2_L - 1) / S2_L, (long)2 * S2_L * C);
        q = sum_strided<double, 8>(part + ((long)ty * 2 + 1) * C + c, (G - ty + S2_L - 1) / S2_L, (long)2 * S2_L * C);
    }
    sh[0][ty][tx] = s; sh[1][ty][tx] = q;
    __syncthreads();
    if (ty != 0 || c >= C) return;
    s = s2_combine(sh[0], tx);
    q = s2_combine(sh[1], tx);
    double mu = s / M;
    double var = q / M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (scale) {                                             // the affine map a consumer folds into its input staging
        const float sc = gamma[c] * rstd[c];
        scale[c] = sc;
        shift[c] = beta[c] - mean[c] * sc;
    }
    if (running_mean) {
        double unb = M > 1 ? var * ((double)M / (M - 1)) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
}
// ws: cdiv(M, rows_per_block)*2*C doubles
TATT_API int tatt_bn_stats(const float* X, long ld, int M, int C, float eps, float momentum, float* mean,
                           float* rstd, float* running_mean, float* running_var, double* ws, hipStream_t st) {
    int G = cs_groups(M);
    int rpb = cdiv(M, G);
    if (v4_ok(X, ld, C)) hipLaunchKernelGGL(bn_stats_stage1_v4, dim3(G), dim3(256), 0, st, X, ld, M, C, rpb, ws);
    else hipLaunchKernelGGL(bn_stats_stage1, dim3(G), dim3(256), 0, st, X, ld, M, C, rpb, ws);
    hipLaunchKernelGGL(bn_stats_stage2, dim3(cdiv(C, 64)), dim3(64 * S2_L), 0, st, ws, G, C, M, eps, momentum, mean, rstd,
                       running_mean, running_var);
    return LAUNCH_CHECK();
}
TATT_API int tatt_bn_stats_finish(const double* part, int G, int C, int M, float eps, float momentum, const float* gamma,
                                  const float* beta, float* mean, float* rstd, float* running_mean, float* running_var,
                                  float* scale, float* shift, hipStream_t st) {
    if (scale && (!gamma || !beta || !shift)) return 1;
    hipLaunchKernelGGL(bn_stats_stage2, dim3(cdiv(C, 64)), dim3(64 * S2_L), 0, st, part, G, C, M, eps, momentum, mean, rstd,
                       running_mean, running_var, gamma, beta, scale, shift);
    return LAUNCH_CHECK();
}
// eval mode: rstd from running_var
__global__ void bn_rstd_kernel(const float* __restrict__ var, float* __restrict__ rstd, int C, float eps) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) rstd[c] = 1.f / sqrtf(var[c] + eps);
}
TATT_API int tatt_bn_rstd(const float* var, float* rstd, int C, float eps, hipStream_t st) {
    hipLaunchKernelGGL(bn_rstd_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, var, rstd, C, eps);
    return LAUNCH_CHECK();
}

// y = act((x - mean) * rstd * gamma + beta)
__global__ void bn_apply_kernel(const float* __restrict__ X, long ldx, float* __restrict__ Y, long ldy, int M, int C,
                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, int act) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)M * C;
    if (idx >= total) return;
    int c = idx % C;
    long m = idx / C;
    float u = (X[m * ldx + c] - mean[c]) * rstd[c] * gamma[c] + beta[c];
    Y[m * ldy + c] = apply_act(u, act);
}
// 4 channels per thread, BA_R rows per thread (the per-channel constants are loaded once)
#define BA_R 4
__global__ __launch_bounds__(256) void bn_apply_v4_kernel(const float* __restrict__ X, long ldx, float* __restrict__ Y, long ldy,
                                                          int M, int C, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int act) {
    const int cv = C >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // (row group, channel vector)
    const int c4 = (int)(idx % cv) * 4;
    const long m0 = (idx / cv) * BA_R;
    if (m0 >= M) return;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4), rs = *reinterpret_cast<const f32x4*>(rstd + c4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c4), b = *reinterpret_cast<const f32x4*>(beta + c4);
    f32x4 v[BA_R];
#pragma unroll
    for (int r = 0; r < BA_R; ++r) v[r] = *reinterpret_cast<const f32x4*>(X + min(m0 + r, (long)M - 1) * ldx + c4);
#pragma unroll
    for (int r = 0; r < BA_R; ++r) {
        if (m0 + r >= M) break;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = apply_act((v[r][e] - mu[e]) * rs[e] * g[e] + b[e], act);
        *reinterpret_cast<f32x4*>(Y + (m0 + r) * ldy + c4) = o;
    }
}
TATT_API int tatt_bn_apply(const float* X, long ldx, float* Y, long ldy, int M, int C, const float* mean,
                           const float* rstd, const float* gamma, const float* beta, int act, hipStream_t st) {
    long total = (long)M * C;
    if (C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && !((((uintptr_t)X) | ((uintptr_t)Y) | ((uintptr_t)mean) | ((uintptr_t)rstd) |
                                                          ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15)) {
        const long items = (long)cdiv(M, BA_R) * (C / 4);
        hipLaunchKernelGGL(bn_apply_v4_kernel, dim3(cdiv(items, 256)), dim3(256), 0, st, X, ldx, Y, ldy, M, C, mean, rstd, gamma,
                           beta, act);
        return LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(bn_apply_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, X, ldx, Y, ldy, M, C, mean, rstd,
                       gamma, beta, act);
    return LAUNCH_CHECK();
}

// backward stage 1: per-channel partial sums of du and du*xhat, du = dy * act'(gamma*xhat+beta)
__global__ void bn_bwd_stage1(const float* __restrict__ X, long ldx, const float* __restrict__ dY, long lddy, int M,
                              int C, int rows_per_block, const float* __restrict__ mean,
                              const float* __restrict__ rstd, const float* __restrict__ gamma,
                              const float* __restrict__ beta, int act, double* __restrict__ part) {
    __shared__ double sh[2][256];
    const int cpb = C < 256 ? C : 256;
    const int rpar = 256 / cpb;
    const int t = threadIdx.x;
    const int cl = t % cpb, rl = t / cpb;
    const int m_begin = blockIdx.x * rows_per_block;
    const int m_end = min(M, m_begin + rows_per_block);
    for (int c0 = 0; c0 < C; c0 += cpb) {
        const int c = c0 + cl;
        double s = 0.0, q = 0.0;
        if (rl < rpar && c < C) {
            const float mu = mean[c], rs = rstd[c], g = gamma[c], b = beta[c];
            int m = m_begin + rl;
            for (; m + 3 * rpar < m_end; m += 4 * rpar) {
                float xv[4], dv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { xv[u] = X[(long)(m + u * rpar) * ldx + c]; dv[u] = dY[(long)(m + u * rpar) * lddy + c]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float xh = (xv[u] - mu) * rs;
                    float du = dv[u];
                    if (act != ACT_NONE) du *= act_grad(g * xh + b, act);
                    s += (double)du; q += (double)du * xh;
                }
            }
            for (; m < m_end; m += rpar) {
                float xh = (X[(long)m * ldx + c] - mu) * rs;
                float du = dY[(long)m * lddy + c];
                if (act != ACT_NONE) du *= act_grad(g * xh + b, act);
                s += (double)du; q += (double)du * xh;
            }
        }
        sh[0][t] = s; sh[1][t] = q;
        __syncthreads();
        if (t < cpb && c < C) {
            double ss = 0.0, qq = 0.0;
            for (int r = 0; r < rpar; ++r) { ss += sh[0][r * cpb + t]; qq += sh[1][r * cpb + t]; }
            part[((long)blockIdx.x * 2 + 0) * C + c] = ss;
            part[((long)blockIdx.x * 2 + 1) * C + c] = qq;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void bn_bwd_stage1_v4(const float* __restrict__ X, long ldx, const float* __restrict__ dY,
                                                        long lddy, int M, int C, int rpb, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int act, double* __restrict__ part) {
    __shared__ double sh[2 * 1024];
    const int t = threadIdx.x;
    const int cv = min(C >> 2, 256), rpar = 256 / cv, cl = t % cv, rl = t / cv;
    const int m_begin = blockIdx.x * rpb, m_end = min(M, m_begin + rpb);
    for (int c0 = 0; c0 < C; c0 += 4 * cv) {
        const int c4 = c0 + 4 * cl;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4), rs = *reinterpret_cast<const f32x4*>(rstd + c4);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c4), b = *reinterpret_cast<const f32x4*>(beta + c4);
        double acc[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
        v4_rows2(X, ldx, dY, lddy, m_begin, m_end, rl, rpar, c4, [&](const f32x4& xv, const f32x4& dv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xv[e] - mu[e]) * rs[e];
                float du = dv[e];
                if (act != ACT_NONE) du *= act_grad(g[e] * xh + b[e], act);
                acc[0][e] += (double)du; acc[1][e] += (double)du * xh;
            }
        });
        v4_block_partials<2>(acc, C, cv, cl, rl, rpar, c0, part, sh);
    }
}
__global__ __launch_bounds__(64 * S2_L) void bn_bwd_stage2(const double* __restrict__ part, int G, int C,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ sums) {
    __shared__ double sh[2][S2_L][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    double s = 0.0, q = 0.0;
    if (c < C && ty < G) {
        s = sum_strided<double, 8>(part + ((long)ty * 2 + 0) * C + c, (G - ty + S2_L - 1) / S2_L, (long)2 * S2_L * C);
        q = sum_strided<double, 8>(part + ((long)ty * 2 + 1) * C + c, (G - ty + S2_L - 1) / S2_L, (long)2 * S2_L * C);
    }
    sh[0][ty][tx] = s; sh[1][ty][tx] = q;
    __syncthreads();
    if (ty != 0 || c >= C) return;
    s = s2_combine(sh[0], tx);
    q = s2_combine(sh[1], tx);
    dbeta[c] = (float)s; dgamma[c] = (float)q;
    sums[c] = (float)s; sums[C + c] = (float)q;
}
// dx = gamma*rstd*(du - mean(du) - xhat*mean(du*xhat))   [training]   or   gamma*rstd*du   [eval]
__global__ void bn_bwd_apply_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ dY, long lddy,
                                    float* __restrict__ dX, long lddx, int M, int C,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                    const float* __restrict__ sums, int training) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)M * C;
    if (idx >= total) return;
    int c = idx % C;
    long m = idx / C;
    float rs = rstd[c], g = gamma[c];
    float xh = (X[m * ldx + c] - mean[c]) * rs;
    float du = dY[m * lddy + c];
    if (act != ACT_NONE) du *= act_grad(g * xh + beta[c], act);
    float v = du;
    if (training) {
        float inv = 1.f / (float)M;
        v = du - sums[c] * inv - xh * sums[C + c] * inv;
    }
    dX[m * lddx + c] = g * rs * v;
}
__global__ __launch_bounds__(256) void bn_bwd_apply_v4_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ dY,
                                                              long lddy, float* __restrict__ dX, long lddx, int M, int C,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int act, const float* __restrict__ sums, int training) {
    const int cv = C >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int c4 = (int)(idx % cv) * 4;
    const long m0 = (idx / cv) * BA_R;
    if (m0 >= M) return;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4), rs = *reinterpret_cast<const f32x4*>(rstd + c4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c4), b = *reinterpret_cast<const f32x4*>(beta + c4);
    const f32x4 s1 = *reinterpret_cast<const f32x4*>(sums + c4), s2 = *reinterpret_cast<const f32x4*>(sums + C + c4);
    const float inv = 1.f / (float)M;
    f32x4 xv[BA_R], dv[BA_R];
#pragma unroll
    for (int r = 0; r < BA_R; ++r) {
        const long m = min(m0 + r, (long)M - 1);
        xv[r] = *reinterpret_cast<const f32x4*>(X + m * ldx + c4);
        dv[r] = *reinterpret_cast<const f32x4*>(dY + m * lddy + c4);
    }
#pragma unroll
    for (int r = 0; r < BA_R; ++r) {
        if (m0 + r >= M) break;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xv[r][e] - mu[e]) * rs[e];
            float du = dv[r][e];
            if (act != ACT_NONE) du *= act_grad(g[e] * xh + b[e], act);
            float v = du;
            if (training) v = du - s1[e] * inv - xh * s2[e] * inv;
            o[e] = g[e] * rs[e] * v;
        }
        *reinterpret_cast<f32x4*>(dX + (m0 + r) * lddx + c4) = o;
    }
}
// ws: cdiv(M,128)*2*C doubles; sums: 2*C floats scratch
TATT_API int tatt_bn_bwd(const float* X, long ldx, const float* dY, long lddy, float* dX, long lddx, int M, int C,
                         const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                         int training, float* dgamma, float* dbeta, float* sums, double* ws, hipStream_t st) {
    int G = cs_groups(M);
    int rpb = cdiv(M, G);
    const bool v4 = v4_ok(X, ldx, C) && v4_ok(dY, lddy, C) && v4_ok(dX, lddx, C) &&
                    !((((uintptr_t)mean) | ((uintptr_t)rstd) | ((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)sums)) & 15);
    if (v4) hipLaunchKernelGGL(bn_bwd_stage1_v4, dim3(G), dim3(256), 0, st, X, ldx, dY, lddy, M, C, rpb, mean, rstd, gamma, beta,
                               act, ws);
    else hipLaunchKernelGGL(bn_bwd_stage1, dim3(G), dim3(256), 0, st, X, ldx, dY, lddy, M, C, rpb, mean, rstd, gamma, beta,
                            act, ws);
    hipLaunchKernelGGL(bn_bwd_stage2, dim3(cdiv(C, 64)), dim3(64 * S2_L), 0, st, ws, G, C, dgamma, dbeta, sums);
    if (v4) {
        const long items = (long)cdiv(M, BA_R) * (C / 4);
        hipLaunchKernelGGL(bn_bwd_apply_v4_kernel, dim3(cdiv(items, 256)), dim3(256), 0, st, X, ldx, dY, lddy, dX, lddx, M, C,
                           mean, rstd, gamma, beta, act, sums, training);
        return LAUNCH_CHECK();
    }
    long total = (long)M * C;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, X, ldx, dY, lddy, dX, lddx, M, C,
                       mean, rstd, gamma, beta, act, sums, training);
    return LAUNCH_CHECK();
}

// ---- BatchNorm backward in pieces (round 4): the stage-1 partials may come from the PRODUCER of the upstream gradient (the data-gradient
// convolution's epilogue, tatt_conv3_c64_dgrad_bn_sb) and the result may be applied by the CONSUMER while it stages its input:
//   dx = gamma rstd (du - mean(du) - xhat mean(du xhat)),  xhat = (x - mean) rstd      is affine per channel in (du, x):
//   dx = a du + b x + c,   a = gamma rstd,  b = -gamma rstd^2 mean(du xhat),  c = -gamma rstd mean(du) - b mean
__global__ __launch_bounds__(64 * S2_L) void bn_bwd_finish_kernel(const double* __restrict__ part, int G, int C, int M,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  const float* __restrict__ gamma, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, float* __restrict__ coef) {
    __shared__ double sh[2][S2_L][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    double s = 0.0, q = 0.0;
    if (c < C && ty < G) {
        s = sum_strided<double, 8>(part + ((long)ty * 2 + 0) * C + c, (G - ty + S2_L - 1) / S2_L, (long)2 * S2_L * C);
        q = sum_strided<double, 8>(part + ((long)ty * 2 + 1) * C + c, (G - ty + S2_L - 1) / S2_L, (long)2 * S2_L * C);
    }
    sh[0][ty][tx] = s; sh[1][ty][tx] = q;
    __syncthreads();
    if (ty != 0 || c >= C) return;
    s = s2_combine(sh[0], tx);
    q = s2_combine(sh[1], tx);
    dbeta[c] = (float)s; dgamma[c] = (float)q;
    const float inv = 1.f / (float)M, g = gamma[c], rs = rstd[c];
    const float a = g * rs;
    const float b = -a * rs * ((float)q * inv);
    coef[c] = a;
    coef[C + c] = b;
    coef[2 * C + c] = -a * ((float)s * inv) - b * mean[c];
}
// stage 1 alone: part[G][2][C] doubles (G = tatt_bn_bwd_groups(M)) of du = dy act'(gamma xhat + beta) and du xhat
TATT_API int tatt_bn_bwd_groups(int M) { return cs_groups(M); }
TATT_API int tatt_bn_bwd_partials(const float* X, long ldx, const float* dY, long lddy, int M, int C, const float* mean,
                                  const float* rstd, const float* gamma, const float* beta, int act, double* part, hipStream_t st) {
    const int G = cs_groups(M), rpb = cdiv(M, G);
    const bool v4 = v4_ok(X, ldx, C) && v4_ok(dY, lddy, C) &&
                    !((((uintptr_t)mean) | ((uintptr_t)rstd) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15);
    if (v4) hipLaunchKernelGGL(bn_bwd_stage1_v4, dim3(G), dim3(256), 0, st, X, ldx, dY, lddy, M, C, rpb, mean, rstd, gamma, beta, act, part);
    else hipLaunchKernelGGL(bn_bwd_stage1, dim3(G), dim3(256), 0, st, X, ldx, dY, lddy, M, C, rpb, mean, rstd, gamma, beta, act, part);
    return LAUNCH_CHECK();
}
// partials of G groups -> dgamma, dbeta (C each) and coef[3][C] = (a, b, c) of  dx = a du + b x + c
TATT_API int tatt_bn_bwd_finish(const double* part, int G, int C, int M, const float* mean, const float* rstd, const float* gamma,
                                float* dgamma, float* dbeta, float* coef, hipStream_t st) {
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(cdiv(C, 64)), dim3(64 * S2_L), 0, st, part, G, C, M, mean, rstd, gamma, dgamma, dbeta, coef);
    return LAUNCH_CHECK();
}
// dX = a dU + b X + c per channel (coef from tatt_bn_bwd_finish): the materialised BatchNorm backward, for consumers that cannot
// apply it themselves (the weight-gradient pass).  X, dU, dX (M, C) contiguous, C % 4 == 0.
__global__ __launch_bounds__(256) void bn_bwd_affine_kernel(const float* __restrict__ X, const float* __restrict__ dU, float* __restrict__ dX,
                                                            long n4, int C, const float* __restrict__ coef) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c4 = (int)(i % (C >> 2)) * 4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(coef + c4), b = *reinterpret_cast<const f32x4*>(coef + C + c4);
    const f32x4 c = *reinterpret_cast<const f32x4*>(coef + 2 * C + c4);
    const f32x4 x = reinterpret_cast<const f32x4*>(X)[i], u = reinterpret_cast<const f32x4*>(dU)[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaf(a[e], u[e], fmaf(b[e], x[e], c[e]));
    reinterpret_cast<f32x4*>(dX)[i] = o;
}
TATT_API int tatt_bn_bwd_affine(const float* X, const float* dU, float* dX, int M, int C, const float* coef, hipStream_t st) {
    if (C % 4) return 1;
    const long n4 = (long)M * (C / 4);
    hipLaunchKernelGGL(bn_bwd_affine_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, st, X, dU, dX, n4, C, coef);
    return LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last axis with fused residual:  y = LN(a + b) * gamma + beta
// (reference nn.LayerNorm: model/transformer_v2.py:459-460,793-795; residual adds :478-483,826-832)
// one wave per row; lanes stride the C channels.  stats: [M][2] = (mean, rstd)
// ---------------------------------------------------------------------------------------------
#define LN_MAXPER 4   // supports C <= 256

// mode 0: nn.LayerNorm  y = (x-mu)/sqrt(biased var + eps)*g + b;   mode 1: the TBSRN variant's own LayerNorm
// (reference model/tbsrn.py:23-36)  y = g*(x-mu)/(UNBIASED std + eps) + b.  stats = (mu, 1/denominator) in both modes.
// pdrop > 0: Bres passes through nn.Dropout(pdrop) on the way in (the mask of tatt_dropout for the same seed word / site / flat
// index row*C + c, so fused and separate forms agree bit for bit): LayerNorm(a + Dropout(b)), transformer_v2.py:478-483,826-832
__device__ __forceinline__ float ln_residual(const float* __restrict__ Bres, long idx, float pdrop, uint64_t seedv, unsigned site,
                                             uint32_t th) {
    const float b = Bres[idx];
    if (pdrop <= 0.f) return b;
    return dropout_keep(seedv, site, (uint64_t)idx, th) ? b * (1.f / (1.f - pdrop)) : 0.f;
}
__global__ void ln_fwd_kernel(const float* __restrict__ A, const float* __restrict__ Bres, float* __restrict__ Y,
                              float* __restrict__ stats, int M, int C, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float eps, int mode, float pdrop,
                              const unsigned long long* __restrict__ seed, unsigned site) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const uint64_t seedv = pdrop > 0.f ? seed[0] : 0;
    const uint32_t th = dropout_thresh(pdrop);
    float v[LN_MAXPER];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXPER; ++k) {
        int c = lane + 64 * k;
        float x = 0.f;
        if (c < C) { x = A[row * C + c]; if (Bres) x += ln_residual(Bres, row * C + c, pdrop, seedv, site, th); }
        v[k] = x; s += x;
    }
    float mu = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXPER; ++k) { int c = lane + 64 * k; if (c < C) { float d = v[k] - mu; q += d * d; } }
    const float ssq = wave_sum(q);
    float rs = mode == 0 ? 1.f / sqrtf(ssq / C + eps) : 1.f / (sqrtf(ssq / (C - 1)) + eps);
#pragma unroll
    for (int k = 0; k < LN_MAXPER; ++k) {
        int c = lane + 64 * k;
        if (c < C) Y[row * C + c] = (v[k] - mu) * rs * gamma[c] + beta[c];
    }
    if (lane == 0 && stats) { stats[row * 2] = mu; stats[row * 2 + 1] = rs; }
}
TATT_API int tatt_ln_fwd(const float* A, const float* Bres, float* Y, float* stats, int M, int C,
                         const float* gamma, const float* beta, float eps, int mode, float pdrop,
                         const unsigned long long* seed, unsigned site, hipStream_t st) {
    if (C > 64 * LN_MAXPER) return 1;
    if (pdrop > 0.f && (!Bres || !seed)) return 2;
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, A, Bres, Y, stats, M, C, gamma, beta, eps, mode,
                       pdrop, seed, site);
    return LAUNCH_CHECK();
}

// backward: dX (= d(a+b)) per row; per-block partial dgamma/dbeta -> part[G][2][C] (float), reduced by colsum.
#define LN_BWD_ROWS 64   // rows per wave-loop block (4 waves x 16 rows)
__global__ void ln_bwd_kernel(const float* __restrict__ A, const float* __restrict__ Bres,
                              const float* __restrict__ dY, const float* __restrict__ stats, float* __restrict__ dX,
                              float* __restrict__ dB, int M, int C, const float* __restrict__ gamma, float* __restrict__ part,
                              float eps, int mode, float pdrop, const unsigned long long* __restrict__ seed, unsigned site) {
    __shared__ float sh[4][2][64 * LN_MAXPER];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t seedv = pdrop > 0.f ? seed[0] : 0;
    const uint32_t th = dropout_thresh(pdrop);
    float dg[LN_MAXPER], db[LN_MAXPER];
#pragma unroll
    for (int k = 0; k < LN_MAXPER; ++k) { dg[k] = 0.f; db[k] = 0.f; }
    const long r0 = (long)blockIdx.x * LN_BWD_ROWS;
    for (int rr = wave; rr < LN_BWD_ROWS; rr += 4) {
        long row = r0 + rr;
        if (row >= M) break;
        const float mu = stats[row * 2], rs = stats[row * 2 + 1];
        float xh[LN_MAXPER], dxh[LN_MAXPER];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXPER; ++k) {
            int c = lane + 64 * k;
            xh[k] = 0.f; dxh[k] = 0.f;
            if (c < C) {
                float x = A[row * C + c]; if (Bres) x += ln_residual(Bres, row * C + c, pdrop, seedv, site, th);
                float dy = dY[row * C + c];
                xh[k] = (x - mu) * rs;
                dxh[k] = dy * gamma[c];
                dg[k] += dy * xh[k]; db[k] += dy;
                s1 += dxh[k]; s2 += dxh[k] * xh[k];
            }
        }
        s1 = wave_sum(s1) / C; s2 = wave_sum(s2);
        // mode 0: dx = rs*(g - mean(g) - xh*mean(g*xh));  mode 1: dx = rs*(g - mean(g)) - xh*sum(g*xh)/((C-1)*std), std = 1/rs - eps
        const float c2 = mode == 0 ? rs * s2 / C : s2 / ((C - 1) * (1.f / rs - eps));
#pragma unroll
        for (int k = 0; k < LN_MAXPER; ++k) {
            int c = lane + 64 * k;
            if (c < C) {
                const float d = rs * (dxh[k] - s1) - xh[k] * c2;
                dX[row * C + c] = d;
                if (dB) dB[row * C + c] = dropout_keep(seedv, site, (uint64_t)(row * C + c), th) ? d * (1.f / (1.f - pdrop)) : 0.f;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < LN_MAXPER; ++k) { sh[wave][0][lane + 64 * k] = dg[k]; sh[wave][1][lane + 64 * k] = db[k]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float g = 0.f, b = 0.f;
        for (int w = 0; w < 4; ++w) { g += sh[w][0][c]; b += sh[w][1][c]; }
        part[((long)blockIdx.x * 2 + 0) * C + c] = g;
        part[((long)blockIdx.x * 2 + 1) * C + c] = b;
    }
}
// C = 128 without dropout (the two LayerNorms of every TBSRN FeatureEnhancer: 10 launches of 49,152 rows per step, 54 us each in the
// generic kernel -- one row per wave iteration, scalar loads, two 6-step cross-lane sums per row; HBM needs 25 us for the 100 MB).
// Here a 16-lane group owns a row (lane = channels 4 l .. + 3 and 64 + 4 l .. + 3: two 16-byte loads per tensor), four rows per
// group in flight, row sums by DPP; the same part[G][2][C] partial layout as ln_bwd_kernel (G = blocks of 64 rows).
__global__ __launch_bounds__(256) void ln_bwd_c128_kernel(const float* __restrict__ A, const float* __restrict__ Bres, const float* __restrict__ dY,
                                                          const float* __restrict__ stats, float* __restrict__ dX, int M,
                                                          const float* __restrict__ gamma, float* __restrict__ part, float eps, int mode) {
    __shared__ float sh[16][2][128];
    const int grp = threadIdx.x >> 4, l = threadIdx.x & 15;
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + 4 * l), g1 = *reinterpret_cast<const f32x4*>(gamma + 64 + 4 * l);
    f32x4 dg0 = {0.f, 0.f, 0.f, 0.f}, dg1 = dg0, db0 = dg0, db1 = dg0;
    const long r0 = (long)blockIdx.x * LN_BWD_ROWS + 4 * grp;
    f32x4 x0[4], x1[4], y0[4], y1[4];
    float mu[4], rs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long row = r0 + j < M ? r0 + j : M - 1;
        const float* a = A + row * 128 + 4 * l;
        x0[j] = *reinterpret_cast<const f32x4*>(a); x1[j] = *reinterpret_cast<const f32x4*>(a + 64);
        if (Bres) { x0[j] += *reinterpret_cast<const f32x4*>(Bres + row * 128 + 4 * l); x1[j] += *reinterpret_cast<const f32x4*>(Bres + row * 128 + 64 + 4 * l); }
        y0[j] = *reinterpret_cast<const f32x4*>(dY + row * 128 + 4 * l); y1[j] = *reinterpret_cast<const f32x4*>(dY + row * 128 + 64 + 4 * l);
        mu[j] = stats[row * 2]; rs[j] = stats[row * 2 + 1];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (r0 + j >= M) break;
        f32x4 xh0, xh1, dx0, dx1;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xh0[e] = (x0[j][e] - mu[j]) * rs[j]; xh1[e] = (x1[j][e] - mu[j]) * rs[j];
            dx0[e] = y0[j][e] * g0[e]; dx1[e] = y1[j][e] * g1[e];
            dg0[e] += y0[j][e] * xh0[e]; dg1[e] += y1[j][e] * xh1[e];
            db0[e] += y0[j][e]; db1[e] += y1[j][e];
            s1 += dx0[e] + dx1[e];
            s2 += dx0[e] * xh0[e] + dx1[e] * xh1[e];
        }
        s1 = row16_sum(s1) * (1.f / 128.f); s2 = row16_sum(s2);
        const float c2 = mode == 0 ? rs[j] * s2 * (1.f / 128.f) : s2 / (127.f * (1.f / rs[j] - eps));
        f32x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o0[e] = rs[j] * (dx0[e] - s1) - xh0[e] * c2; o1[e] = rs[j] * (dx1[e] - s1) - xh1[e] * c2; }
        float* d = dX + (r0 + j) * 128 + 4 * l;
        *reinterpret_cast<f32x4*>(d) = o0; *reinterpret_cast<f32x4*>(d + 64) = o1;
    }
    *reinterpret_cast<f32x4*>(&sh[grp][0][4 * l]) = dg0; *reinterpret_cast<f32x4*>(&sh[grp][0][64 + 4 * l]) = dg1;
    *reinterpret_cast<f32x4*>(&sh[grp][1][4 * l]) = db0; *reinterpret_cast<f32x4*>(&sh[grp][1][64 + 4 * l]) = db1;
    __syncthreads();
    {
        const int which = threadIdx.x >> 7, c = threadIdx.x & 127;
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) v += sh[g][which][c];
        part[((long)blockIdx.x * 2 + which) * 128 + c] = v;
    }
}
// part: cdiv(M,64)*2*C floats;  ws: doubles for the colsum (cdiv(G,256)*2*C)
// dB (with pdrop > 0): the gradient of the residual input in front of its dropout, Dropout'(dX); without dropout it equals dX
TATT_API int tatt_ln_bwd(const float* A, const float* Bres, const float* dY, const float* stats, float* dX, float* dB, int M,
                         int C, const float* gamma, float* dgamma, float* dbeta, float* part, double* ws, float eps,
                         int mode, float pdrop, const unsigned long long* seed, unsigned site, hipStream_t st) {
    if (C > 64 * LN_MAXPER) return 1;
    if (pdrop > 0.f && (!Bres || !seed || !dB)) return 2;
    if (pdrop <= 0.f) dB = nullptr;
    int G = cdiv(M, LN_BWD_ROWS);
    if (C == 128 && pdrop <= 0.f)
        hipLaunchKernelGGL(ln_bwd_c128_kernel, dim3(G), dim3(256), 0, st, A, Bres, dY, stats, dX, M, gamma, part, eps, mode);
    else
        hipLaunchKernelGGL(ln_bwd_kernel, dim3(G), dim3(256), 0, st, A, Bres, dY, stats, dX, dB, M, C, gamma, part, eps, mode, pdrop,
                           seed, site);
    // part viewed as (G, 2C) -> column sums give [dgamma | dbeta]
    int G2 = cs_groups(G);
    int rpb = cdiv(G, G2);
    hipLaunchKernelGGL((colsum_stage1<float>), dim3(G2), dim3(256), 0, st, part, (long)2 * C, G, 2 * C, rpb, ws);
    if (dbeta == dgamma + C) {                      // adjacent outputs: one launch over the 2C columns
        hipLaunchKernelGGL(colsum_stage2, dim3(cdiv(2 * C, 64)), dim3(64 * S2_L), 0, st, ws, G2, 2 * C, (long)2 * C, dgamma, 1.f, 0.f);
        return LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(colsum_stage2, dim3(cdiv(C, 64)), dim3(64 * S2_L), 0, st, ws, G2, C, (long)2 * C, dgamma, 1.f, 0.f);
    hipLaunchKernelGGL(colsum_stage2, dim3(cdiv(C, 64)), dim3(64 * S2_L), 0, st, ws + C, G2, C, (long)2 * C, dbeta, 1.f, 0.f);
    return LAUNCH_CHECK();
}
