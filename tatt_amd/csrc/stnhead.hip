// STN head (reference model/stn_head.py:25-106: six conv3x3 + BatchNorm + ReLU (+ MaxPool) layers on maps of <= 6 MB, then
// Linear + BatchNorm1d + ReLU and Linear) as a FEW launches.  Operator by operator the head is 53 dependent launches in the forward
// and 62 in the backward, each about 5 us of launch boundary around a microsecond of work, and it sits at BOTH exposed ends of the
// training step (nothing can run before its forward or after its backward).  Here everything between two convolutions is ONE launch:
//
//   stn_bn_pool_fwd_kernel   batch statistics (+ running statistics) + normalise + ReLU + max-pool
//   stn_bn_pool_bwd_kernel   pool routing + ReLU mask + BatchNorm backward (+ dgamma, dbeta, the convolution's bias gradient)
//   stn_fc_fwd_kernel        flatten + Linear + BatchNorm1d + ReLU + 0.1 x + Linear
//   stn_fc_bwd_kernel        the backward of that, every parameter gradient included
//
// A BatchNorm is a reduction over the whole map between two passes over it -- the reason the operator chain needs 3-4 launches per
// layer.  These kernels keep it inside one launch: the G <= 128 work-groups of a launch publish partial sums with write-through (sc1)
// stores, raise a flag word each, and wait for the others' flags (the agent-scope hand-off forms of MI355X_MICROARCH.md, validated on
// this chip by the persistent query-GRU launches in gru.hip); every work-group then reduces the G partials in the same fixed order, so
// all of them -- and every run -- compute bit-identical statistics.  Flags are epoch-stamped: a site's 256-word sync buffer is zeroed
// once when it is allocated and reused by every later launch of that site (word 254 = epoch, word 255 = error); launches of one site
// must not overlap (they are issued on one stream).  Spins are bounded by the wall clock: on expiry word 63 is raised and the launch
// runs on (results invalid, reported by the host when it next synchronises).
#include "common.h"

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define SN_WORDS 256                      // words of a site's sync buffer: flags 0 .. SN_MAXG-1, epoch, error
#define SN_EPOCH 254
#define SN_ERR 255
#define SN_SPIN_TICKS 200000000L          // 2 s of the 100 MHz wall clock
#define SN_MAXG 128
#define SN_MAXW 4                         // pool windows per thread (they stay in registers between the two passes over the map)

// 16-byte write-through store / L1-bypassing load at base + off bytes (base wave-uniform: the descriptor stays in SGPRs)
__device__ __forceinline__ void st16_sc1(void* base, int off, const void* v16) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffff0, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4_t*>(v16), rs, off, 0, 16 /* sc1: write-through */);
}
__device__ __forceinline__ u32x4_t ld16_sc1(const void* base, int off) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7ffffff0, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16 /* sc1: not from this CU's L1 */);
}

// all of this launch's work-groups have published phase `phase` (1, 2, 3): called by every thread, ends with a barrier.
// Publishing = stores (sc1) -> s_waitcnt vmcnt(0) -> barrier -> thread 0 raises the flag.
__device__ __forceinline__ void sn_publish(unsigned* sync, int g, unsigned val) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(sync + g, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sn_wait(unsigned* sync, int G, unsigned val, int* s_dead, unsigned* sticky) {
    if (threadIdx.x < 64 && !*s_dead) {
        const int lane = threadIdx.x;
        const long t0 = wall_clock64();
        int it = 0;
        for (;;) {
            const unsigned f0 = lane < G ? __hip_atomic_load(sync + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : val;
            const unsigned f1 = lane + 64 < G ? __hip_atomic_load(sync + lane + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : val;
            if (__builtin_amdgcn_ballot_w64((int)(f0 - val) < 0 || (int)(f1 - val) < 0) == 0) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++it & 63) == 0 && wall_clock64() - t0 > SN_SPIN_TICKS) {
                if (lane == 0) { *s_dead = 1; __hip_atomic_store(sync + SN_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); tatt_raise_sticky(sticky, TATT_STICKY_STN); }
                break;
            }
        }
    }
    __syncthreads();
}

struct SnGeom { int B, H, W, C, G; };   // (B, H, W, C) map split over G work-groups

// Thread layout of the map kernels: thread t = (row lane rl = t / cv, channel quad cq = t % cv), cv = C / 4 in {8, 16, 32, 64}.
// sn_allsum: sum of `v` over the row lanes of every channel quad, valid in ALL threads afterwards, in a fixed order (lanes of a wave by
// xor shuffles, then the four waves through LDS): deterministic.  red: [4][64] doubles of scratch per value.
template <int NV>
__device__ __forceinline__ void sn_allsum(double (*red)[4][64], double* v, int cv) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = cv; off < 64; off <<= 1)
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] += __shfl_xor(v[k], off, 64);
    __syncthreads();
    // lane l of wave w holds the wave's sum for channel quad (w * 64 + l) % cv
    if (lane < cv || cv == 64)
#pragma unroll
        for (int k = 0; k < NV; ++k) red[k][wave][lane] = v[k];
    __syncthreads();
    const int src = cv == 64 ? lane : (threadIdx.x % cv);
    if (cv == 64) {
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = red[k][0][src] + red[k][1][src] + red[k][2][src] + red[k][3][src];
    } else {
        // with cv < 64 every wave covers all channel quads: lane l holds quad l % cv, the same in each wave
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = red[k][0][src] + red[k][1][src] + red[k][2][src] + red[k][3][src];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// forward: X (B,H,W,C) raw convolution output -> A (B,H/PH,W/PW,C) = maxpool(relu(bn(X))), batch statistics over B*H*W
// (nn.BatchNorm2d train mode + nn.ReLU + nn.MaxPool2d, model/stn_head.py:9-15,33-49).  part: G * 2 * C doubles.  A thread owns up to
// SN_MAXW pool windows of one channel quad; they are loaded ONCE and stay in registers across the in-launch reduction.
// ---------------------------------------------------------------------------------------------------------------------------------
struct SnFwdP {
    const float* X; float* A; const float* gamma; const float* beta; float* mean; float* rstd; float* running_mean; float* running_var;
    double* part; unsigned* sync; SnGeom g; float eps, momentum; unsigned* sticky;
    // X as S partial maps `slab` floats apart (the split contraction of the convolution before it, left unsummed): they are added
    // here as they are loaded, in slab order, then + bias (tatt_splitk_reduce adds the same terms four-way interleaved: last-bit
    // differences); Xout receives the finished map
    int S; long slab; const float* bias; float* Xout;
};
template <int PH, int PW>
__global__ __launch_bounds__(256) void stn_bn_pool_fwd_kernel(SnFwdP p) {
    constexpr int NW = PH * PW;
    __shared__ double red[8][4][64];
    __shared__ int s_dead;
    const int C = p.g.C, cv = C >> 2, RL = 256 / cv, G = p.g.G;
    const int t = threadIdx.x, cq = t % cv, rl = t / cv, g = blockIdx.x;
    const int Ho = p.g.H / PH, Wo = p.g.W / PW;
    const long P = (long)p.g.B * Ho * Wo;
    const long p0 = P * g / G, p1 = P * (g + 1) / G;
    if (t == 0) s_dead = 0;
    const unsigned ep = __hip_atomic_load(p.sync + SN_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f32x4 x[SN_MAXW][NW];
#pragma unroll
    for (int w = 0; w < SN_MAXW; ++w) {
        const long pp = min(p0 + rl + (long)w * RL, P - 1);
        const int ow = pp % Wo; const long r = pp / Wo; const int oh = r % Ho; const long b = r / Ho;
#pragma unroll
        for (int k = 0; k < NW; ++k)
            x[w][k] = *reinterpret_cast<const f32x4*>(p.X + (((b * p.g.H + oh * PH + k / PW) * p.g.W + ow * PW + k % PW) * C + cq * 4));
    }
    if (p.S > 1 || p.Xout) {                                              // (uniform) the convolution's split contraction ends here
        long idx[SN_MAXW];
#pragma unroll
        for (int w = 0; w < SN_MAXW; ++w) {
            const long pp = min(p0 + rl + (long)w * RL, P - 1);
            const int ow = pp % Wo; const long r = pp / Wo; const int oh = r % Ho; const long b = r / Ho;
            idx[w] = ((b * p.g.H + oh * PH) * p.g.W + ow * PW) * C + cq * 4;
        }
        for (int sl = 1; sl < p.S; ++sl) {                                  // slab-major: the loads of one slab are independent
            const float* Xs = p.X + sl * p.slab;
            f32x4 tv[SN_MAXW][NW];
#pragma unroll
            for (int w = 0; w < SN_MAXW; ++w)
#pragma unroll
                for (int k = 0; k < NW; ++k) tv[w][k] = *reinterpret_cast<const f32x4*>(Xs + idx[w] + ((k / PW) * p.g.W + k % PW) * C);
#pragma unroll
            for (int w = 0; w < SN_MAXW; ++w)
#pragma unroll
                for (int k = 0; k < NW; ++k) x[w][k] += tv[w][k];
        }
        f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.bias) b4 = *reinterpret_cast<const f32x4*>(p.bias + cq * 4);
#pragma unroll
        for (int w = 0; w < SN_MAXW; ++w)
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                x[w][k] += b4;
                if (p.Xout && p0 + rl + (long)w * RL < p1)
                    *reinterpret_cast<f32x4*>(p.Xout + idx[w] + ((k / PW) * p.g.W + k % PW) * C) = x[w][k];
            }
    }
    const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + cq * 4), be = *reinterpret_cast<const f32x4*>(p.beta + cq * 4);
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < SN_MAXW; ++w)
        if (p0 + rl + (long)w * RL < p1)
#pragma unroll
            for (int k = 0; k < NW; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[e] += (double)x[w][k][e]; acc[4 + e] += (double)x[w][k][e] * (double)x[w][k][e]; }
    sn_allsum<8>(red, acc, cv);
    if (rl == 0) {
        const int d = (g * 2 * C + cq * 4) * 8;
        double v[2];
        v[0] = acc[0]; v[1] = acc[1]; st16_sc1(p.part, d, v);
        v[0] = acc[2]; v[1] = acc[3]; st16_sc1(p.part, d + 16, v);
        v[0] = acc[4]; v[1] = acc[5]; st16_sc1(p.part, d + C * 8, v);
        v[0] = acc[6]; v[1] = acc[7]; st16_sc1(p.part, d + C * 8 + 16, v);
    }
    sn_publish(p.sync, g, ep * 4 + 1);
    sn_wait(p.sync, G, ep * 4 + 1, &s_dead, p.sticky);
    if (g == 0 && t == 0) __hip_atomic_store(p.sync + SN_EPOCH, ep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // totals: every work-group adds the G partials in the same order (row lane rl takes partials rl, rl + RL, ...; then sn_allsum)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0;
    for (int g0 = rl; g0 < G; g0 += 4 * RL) {
        u32x4_t raw[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int gg = min(g0 + u * RL, G - 1), d = (gg * 2 * C + cq * 4) * 8;
            raw[u][0] = ld16_sc1(p.part, d); raw[u][1] = ld16_sc1(p.part, d + 16);
            raw[u][2] = ld16_sc1(p.part, d + C * 8); raw[u][3] = ld16_sc1(p.part, d + C * 8 + 16);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (g0 + u * RL < G)
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const double* dv = reinterpret_cast<const double*>(&raw[u][h]);
                    acc[2 * h] += dv[0]; acc[2 * h + 1] += dv[1];
                }
    }
    sn_allsum<8>(red, acc, cv);
    const double N = (double)p.g.B * p.g.H * p.g.W;
    f32x4 mu, rs;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double m = acc[e] / N;
        double var = acc[4 + e] / N - m * m;
        if (var < 0.0) var = 0.0;
        mu[e] = (float)m; rs[e] = (float)(1.0 / sqrt(var + (double)p.eps));
        if (g == 0 && rl == 0) {
            const int c = cq * 4 + e;
            p.mean[c] = mu[e]; p.rstd[c] = rs[e];
            if (p.running_mean) {
                const double unb = N > 1 ? var * (N / (N - 1)) : var;
                p.running_mean[c] = (float)((1.0 - p.momentum) * p.running_mean[c] + p.momentum * m);
                p.running_var[c] = (float)((1.0 - p.momentum) * p.running_var[c] + p.momentum * unb);
            }
        }
    }
#pragma unroll
    for (int w = 0; w < SN_MAXW; ++w) {
        const long pp = p0 + rl + (long)w * RL;
        if (pp >= p1) break;
        f32x4 best = (f32x4){0.f, 0.f, 0.f, 0.f};                      // max over relu(.) >= 0
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float u = (x[w][k][e] - mu[e]) * rs[e] * ga[e] + be[e];
                best[e] = u > best[e] ? u : best[e];
            }
        *reinterpret_cast<f32x4*>(p.A + pp * C + cq * 4) = best;
    }
}
// work-groups of a map launch: enough that a thread owns <= SN_MAXW windows, at least 2 for the small maps, at most SN_MAXG
static int sn_groups(long P, int C) {
    const long per = (long)SN_MAXW * (256 / (C / 4));
    long G = (P + per - 1) / per;
    if (G < 2) G = 2;
    return G > SN_MAXG ? -1 : (int)G;
}
static bool sn_geom_ok(int B, int H, int W, int C, int ph, int pw) {
    return B > 0 && C % 4 == 0 && C >= 32 && C <= 256 && 256 % (C / 4) == 0 && ((ph == 1 && pw == 1) || (ph == 1 && pw == 2) || (ph == 2 && pw == 2)) &&
           H % ph == 0 && W % pw == 0 && sn_groups((long)B * (H / ph) * (W / pw), C) > 0;
}
// part >= 128 * 2 * C doubles, sync = the site's 256-word buffer (see the header of this file).  running_* may be NULL.
static int sn_fwd_launch(const float* X, int S, const float* bias, float* Xout, float* A, const float* gamma, const float* beta,
                         float* mean, float* rstd, float* running_mean, float* running_var, double* part, unsigned* sync, int B, int H,
                         int W, int C, int ph, int pw, float eps, float momentum, hipStream_t st) {
    if (!sn_geom_ok(B, H, W, C, ph, pw) || S < 1) return 1;
    SnGeom g = {B, H, W, C, sn_groups((long)B * (H / ph) * (W / pw), C)};
    SnFwdP p = {X, A, gamma, beta, mean, rstd, running_mean, running_var, part, sync, g, eps, momentum, tatt_sticky_ptr(),
                S, (long)B * H * W * C, bias, Xout};
    if (ph == 2) hipLaunchKernelGGL((stn_bn_pool_fwd_kernel<2, 2>), dim3(g.G), dim3(256), 0, st, p);
    else if (pw == 2) hipLaunchKernelGGL((stn_bn_pool_fwd_kernel<1, 2>), dim3(g.G), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((stn_bn_pool_fwd_kernel<1, 1>), dim3(g.G), dim3(256), 0, st, p);
    return LAUNCH_CHECK();
}
TATT_API int tatt_stn_bn_pool_fwd(const float* X, float* A, const float* gamma, const float* beta, float* mean, float* rstd,
                                  float* running_mean, float* running_var, double* part, unsigned* sync, int B, int H, int W, int C,
                                  int ph, int pw, float eps, float momentum, hipStream_t st) {
    return sn_fwd_launch(X, 1, nullptr, nullptr, A, gamma, beta, mean, rstd, running_mean, running_var, part, sync, B, H, W, C, ph, pw,
                         eps, momentum, st);
}
// The same with the convolution's split contraction folded in: Xparts = S partial maps (S, B, H, W, C) as tatt_conv2d_fwd_partials
// leaves them; X = sum of the slabs in order + bias (bias may be NULL) is written to Xout (the backward reads it) and normalised.
TATT_API int tatt_stn_bn_pool_fwd_parts(const float* Xparts, int S, const float* bias, float* Xout, float* A, const float* gamma,
                                        const float* beta, float* mean, float* rstd, float* running_mean, float* running_var,
                                        double* part, unsigned* sync, int B, int H, int W, int C, int ph, int pw, float eps,
                                        float momentum, hipStream_t st) {
    if (!Xout) return 1;
    return sn_fwd_launch(Xparts, S, bias, Xout, A, gamma, beta, mean, rstd, running_mean, running_var, part, sync, B, H, W, C, ph, pw,
                         eps, momentum, st);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward of the same: dA (B,H/PH,W/PW,C) -> dX (B,H,W,C) (gradient w.r.t. the convolution output), dgamma, dbeta, and dbias = the
// column sums of dX (the convolution's bias gradient).  The forward's y is recomputed from X: a window's gradient goes to its FIRST
// maximum in scan order (tatt_maxpool_bwd's rule) if that element's pre-activation is positive.  part: G * 3 * C doubles.
// ---------------------------------------------------------------------------------------------------------------------------------
struct SnBwdP {
    const float* X; const float* dA; const float* gamma; const float* beta; const float* mean; const float* rstd;
    float* dX; float* dgamma; float* dbeta; float* dbias; double* part; unsigned* sync; SnGeom g; unsigned* sticky;
    int S; long slab;        // dA as S partial maps `slab` floats apart (the data-gradient convolution's split contraction): summed on load
};
template <int PH, int PW>
__global__ __launch_bounds__(256) void stn_bn_pool_bwd_kernel(SnBwdP p) {
    constexpr int NW = PH * PW;
    __shared__ double red[8][4][64];
    __shared__ int s_dead;
    const int C = p.g.C, cv = C >> 2, RL = 256 / cv, G = p.g.G;
    const int t = threadIdx.x, cq = t % cv, rl = t / cv, g = blockIdx.x;
    const int Ho = p.g.H / PH, Wo = p.g.W / PW;
    const long P = (long)p.g.B * Ho * Wo;
    const long p0 = P * g / G, p1 = P * (g + 1) / G;
    if (t == 0) s_dead = 0;
    const unsigned ep = __hip_atomic_load(p.sync + SN_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f32x4 xh[SN_MAXW][NW], da[SN_MAXW];                               // x, then x-hat; the pooled gradient routed: dy per element
    long base[SN_MAXW];
#pragma unroll
    for (int w = 0; w < SN_MAXW; ++w) {
        const long pp = min(p0 + rl + (long)w * RL, P - 1);
        const int ow = pp % Wo; const long r = pp / Wo; const int oh = r % Ho; const long b = r / Ho;
        base[w] = ((b * p.g.H + oh * PH) * p.g.W + ow * PW) * C + cq * 4;
#pragma unroll
        for (int k = 0; k < NW; ++k) xh[w][k] = *reinterpret_cast<const f32x4*>(p.X + base[w] + ((k / PW) * p.g.W + k % PW) * C);
        da[w] = *reinterpret_cast<const f32x4*>(p.dA + pp * C + cq * 4);
    }
    for (int sl = 1; sl < p.S; sl += 2) {                                   // slab-major, two slabs in flight; added in slab order
        const bool two = sl + 1 < p.S;
        f32x4 t0[SN_MAXW], t1[SN_MAXW];
#pragma unroll
        for (int w = 0; w < SN_MAXW; ++w) {
            const long pp = min(p0 + rl + (long)w * RL, P - 1);
            t0[w] = *reinterpret_cast<const f32x4*>(p.dA + sl * p.slab + pp * C + cq * 4);
            t1[w] = two ? *reinterpret_cast<const f32x4*>(p.dA + (sl + 1) * p.slab + pp * C + cq * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int w = 0; w < SN_MAXW; ++w) { da[w] += t0[w]; if (two) da[w] += t1[w]; }
    }
    const f32x4 mu = *reinterpret_cast<const f32x4*>(p.mean + cq * 4), rs = *reinterpret_cast<const f32x4*>(p.rstd + cq * 4);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + cq * 4), be = *reinterpret_cast<const f32x4*>(p.beta + cq * 4);
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned argm[SN_MAXW];                                             // 4 x 4 bits: per channel the element taking the gradient (15: none)
#pragma unroll
    for (int w = 0; w < SN_MAXW; ++w) {
        const bool live = p0 + rl + (long)w * RL < p1;
        unsigned am = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float best = 0.f; int arg = 15;
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const float h = (xh[w][k][e] - mu[e]) * rs[e];
                xh[w][k][e] = h;
                const float u = h * ga[e] + be[e];                          // relu(u) > 0 and a new maximum: takes the gradient
                if (u > best) { best = u; arg = k; }
            }
            am |= (unsigned)arg << (4 * e);
            if (live && arg != 15) {
                float hsel = xh[w][0][e];
#pragma unroll
                for (int k = 1; k < NW; ++k) hsel = arg == k ? xh[w][k][e] : hsel;
                acc[e] += (double)da[w][e]; acc[4 + e] += (double)da[w][e] * (double)hsel;
            }
        }
        argm[w] = am;
    }
    sn_allsum<8>(red, acc, cv);
    if (rl == 0) {
        const int d = (g * 3 * C + cq * 4) * 8;
        double v[2];
        v[0] = acc[0]; v[1] = acc[1]; st16_sc1(p.part, d, v);
        v[0] = acc[2]; v[1] = acc[3]; st16_sc1(p.part, d + 16, v);
        v[0] = acc[4]; v[1] = acc[5]; st16_sc1(p.part, d + C * 8, v);
        v[0] = acc[6]; v[1] = acc[7]; st16_sc1(p.part, d + C * 8 + 16, v);
    }
    sn_publish(p.sync, g, ep * 4 + 1);
    sn_wait(p.sync, G, ep * 4 + 1, &s_dead, p.sticky);
    if (g == 0 && t == 0) __hip_atomic_store(p.sync + SN_EPOCH, ep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0;
    for (int g0 = rl; g0 < G; g0 += 4 * RL) {
        u32x4_t raw[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int gg = min(g0 + u * RL, G - 1), d = (gg * 3 * C + cq * 4) * 8;
            raw[u][0] = ld16_sc1(p.part, d); raw[u][1] = ld16_sc1(p.part, d + 16);
            raw[u][2] = ld16_sc1(p.part, d + C * 8); raw[u][3] = ld16_sc1(p.part, d + C * 8 + 16);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (g0 + u * RL < G)
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const double* dv = reinterpret_cast<const double*>(&raw[u][h]);
                    acc[2 * h] += dv[0]; acc[2 * h + 1] += dv[1];
                }
    }
    sn_allsum<8>(red, acc, cv);
    if (g == 0 && rl == 0)
#pragma unroll
        for (int e = 0; e < 4; ++e) { p.dbeta[cq * 4 + e] = (float)acc[e]; p.dgamma[cq * 4 + e] = (float)acc[4 + e]; }
    const float inv = 1.f / ((float)p.g.B * p.g.H * p.g.W);
    f32x4 s1, s2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[e] = (float)acc[e]; s2[e] = (float)acc[4 + e]; }
    double cs[4] = {0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < SN_MAXW; ++w) {
        if (p0 + rl + (long)w * RL >= p1) break;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float du = ((argm[w] >> (4 * e)) & 15u) == (unsigned)k ? da[w][e] : 0.f;
                const float v = du - s1[e] * inv - xh[w][k][e] * s2[e] * inv;
                o[e] = ga[e] * rs[e] * v;
                cs[e] += (double)o[e];
            }
            *reinterpret_cast<f32x4*>(p.dX + base[w] + ((k / PW) * p.g.W + k % PW) * C) = o;
        }
    }
    if (!p.dbias) return;
    sn_allsum<4>(red, cs, cv);
    if (rl == 0) {
        const int d = (g * 3 * C + 2 * C + cq * 4) * 8;
        double v[2];
        v[0] = cs[0]; v[1] = cs[1]; st16_sc1(p.part, d, v);
        v[0] = cs[2]; v[1] = cs[3]; st16_sc1(p.part, d + 16, v);
    }
    sn_publish(p.sync, g, ep * 4 + 2);
    if (g != 0) return;
    sn_wait(p.sync, G, ep * 4 + 2, &s_dead, p.sticky);
    double s[4] = {0, 0, 0, 0};
    for (int gg = rl; gg < G; gg += RL) {
        const int d = (gg * 3 * C + 2 * C + cq * 4) * 8;
        const u32x4_t r0 = ld16_sc1(p.part, d), r1 = ld16_sc1(p.part, d + 16);
        const double* d0 = reinterpret_cast<const double*>(&r0); const double* d1 = reinterpret_cast<const double*>(&r1);
        s[0] += d0[0]; s[1] += d0[1]; s[2] += d1[0]; s[3] += d1[1];
    }
    sn_allsum<4>(red, s, cv);
    if (rl == 0)
#pragma unroll
        for (int e = 0; e < 4; ++e) p.dbias[cq * 4 + e] = (float)s[e];
}
// part >= 128 * 3 * C doubles; dbias may be NULL.
// dA: S partial maps (S, B, H/ph, W/pw, C) summed in slab order as they are loaded (S = 1: the map itself)
TATT_API int tatt_stn_bn_pool_bwd_parts(const float* X, const float* dAparts, int S, const float* gamma, const float* beta,
                                        const float* mean, const float* rstd, float* dX, float* dgamma, float* dbeta, float* dbias,
                                        double* part, unsigned* sync, int B, int H, int W, int C, int ph, int pw, hipStream_t st) {
    if (!sn_geom_ok(B, H, W, C, ph, pw) || S < 1) return 1;
    SnGeom g = {B, H, W, C, sn_groups((long)B * (H / ph) * (W / pw), C)};
    SnBwdP p = {X, dAparts, gamma, beta, mean, rstd, dX, dgamma, dbeta, dbias, part, sync, g, tatt_sticky_ptr(),
                S, (long)B * (H / ph) * (W / pw) * C};
    if (ph == 2) hipLaunchKernelGGL((stn_bn_pool_bwd_kernel<2, 2>), dim3(g.G), dim3(256), 0, st, p);
    else if (pw == 2) hipLaunchKernelGGL((stn_bn_pool_bwd_kernel<1, 2>), dim3(g.G), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((stn_bn_pool_bwd_kernel<1, 1>), dim3(g.G), dim3(256), 0, st, p);
    return LAUNCH_CHECK();
}
TATT_API int tatt_stn_bn_pool_bwd(const float* X, const float* dA, const float* gamma, const float* beta, const float* mean,
                                  const float* rstd, float* dX, float* dgamma, float* dbeta, float* dbias, double* part,
                                  unsigned* sync, int B, int H, int W, int C, int ph, int pw, hipStream_t st) {
    return tatt_stn_bn_pool_bwd_parts(X, dA, 1, gamma, beta, mean, rstd, dX, dgamma, dbeta, dbias, part, sync, B, H, W, C, ph, pw, st);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// the fully connected end of the head (model/stn_head.py:51-58,96-106): x.view(B, -1) of the (B, 256, 1, 2) map -> Linear(512, 512)
// -> BatchNorm1d (batch statistics over the B rows) -> ReLU -> x 0.1 -> Linear(512, NO).  32 work-groups, each owns 16 of the 512
// hidden features for ALL rows: the BatchNorm1d statistics are local to a work-group; the second Linear leaves as per-work-group
// partial products that work-group 0 adds up in a fixed order.  A6 is the NHWC map (B, 2, 256): feature c*2 + w of the reference's
// NCHW flattening is A6[b][w][c] -- the contraction simply runs in A6's memory order with W1's columns permuted to match.
// MFMA 16x16x4: rows = samples (B <= 64: up to 4 tiles), columns = the 16 features, 8 waves split the 512-long contraction.
// ---------------------------------------------------------------------------------------------------------------------------------
#define FC_H 512
struct SnFcFwdP {
    const float* A6; const float* W1; const float* b1; const float* g1; const float* be1; float* rm1; float* rv1;
    const float* W2; const float* b2; float* U; float* mean1; float* rstd1; float* S; float* ctrl; float* part; unsigned* sync;
    int B, NO; float eps, momentum; unsigned* sticky;
};
__global__ __launch_bounds__(512) void stn_fc_fwd_kernel(SnFcFwdP p) {
    __shared__ float red[8][4][16][17];
    __shared__ double sd[2][16][16];
    __shared__ float bc[2][16];
    __shared__ __attribute__((aligned(16))) float sS[64][16];
    __shared__ __attribute__((aligned(16))) float w2s[64][16];
    __shared__ int s_dead;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), g = blockIdx.x, j0 = g * 16;
    const int i = lane & 15, q = lane >> 4, B = p.B, NO = p.NO, MT = (B + 15) >> 4;
    if (t == 0) s_dead = 0;
    const unsigned ep = __hip_atomic_load(p.sync + SN_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int idx = t; idx < NO * 16; idx += 512) w2s[idx >> 4][idx & 15] = p.W2[(long)(idx >> 4) * FC_H + j0 + (idx & 15)];
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = wave * 32 + 16 * u + 4 * q;
        const f32x4 wlo = *reinterpret_cast<const f32x4*>(p.W1 + (long)(j0 + i) * FC_H + 2 * c);
        const f32x4 whi = *reinterpret_cast<const f32x4*>(p.W1 + (long)(j0 + i) * FC_H + 2 * c + 4);
        const float bw0[4] = {wlo[0], wlo[2], whi[0], whi[2]}, bw1[4] = {wlo[1], wlo[3], whi[1], whi[3]};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (mt >= MT) break;
            const int b = min(mt * 16 + i, B - 1);
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(p.A6 + (long)b * FC_H + c);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(p.A6 + (long)b * FC_H + 256 + c);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[v], bw0[v], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[v], bw1[v], acc[mt], 0, 0, 0);
            }
        }
    }
    {
        const int col = lane & 15, rb = (lane >> 4) * 4;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) red[wave][mt][rb + rr][col] = acc[mt][rr];
    }
    __syncthreads();
    const int m = (t >> 4) & 15, j = t & 15;
    float uv[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < 256) {
        double s = 0.0, s2 = 0.0;
        const float bias = p.b1[j0 + j];
        for (int mt = 0; mt < MT; ++mt) {
            const int b = mt * 16 + m;
            if (b >= B) break;
            float x = bias;
#pragma unroll
            for (int w = 0; w < 8; ++w) x += red[w][mt][m][j];
            uv[mt] = x;
            p.U[(long)b * FC_H + j0 + j] = x;
            s += (double)x; s2 += (double)x * (double)x;
        }
        sd[0][m][j] = s; sd[1][m][j] = s2;
    }
    __syncthreads();
    if (t < 16) {
        double s = 0.0, s2 = 0.0;
        for (int mm = 0; mm < 16; ++mm) { s += sd[0][mm][t]; s2 += sd[1][mm][t]; }
        const double mu = s / B;
        double var = s2 / B - mu * mu;
        if (var < 0.0) var = 0.0;
        const float m_f = (float)mu, r_f = (float)(1.0 / sqrt(var + (double)p.eps));
        bc[0][t] = m_f; bc[1][t] = r_f;
        p.mean1[j0 + t] = m_f; p.rstd1[j0 + t] = r_f;
        if (p.rm1) {
            const double unb = B > 1 ? var * ((double)B / (B - 1)) : var;
            p.rm1[j0 + t] = (float)((1.0 - p.momentum) * p.rm1[j0 + t] + p.momentum * mu);
            p.rv1[j0 + t] = (float)((1.0 - p.momentum) * p.rv1[j0 + t] + p.momentum * unb);
        }
    }
    __syncthreads();
    if (t < 256) {
        const float mu = bc[0][j], rs = bc[1][j], ga = p.g1[j0 + j], be = p.be1[j0 + j];
        for (int mt = 0; mt < MT; ++mt) {
            const int b = mt * 16 + m;
            if (b >= B) break;
            const float v = (uv[mt] - mu) * rs * ga + be;
            const float sv = 0.1f * (v > 0.f ? v : 0.f);
            sS[b][j] = sv;
            p.S[(long)b * FC_H + j0 + j] = sv;
        }
    }
    __syncthreads();
    // this work-group's share of the second Linear: part[g][b][o] = sum_j S[b][j0 + j] W2[o][j0 + j]
    const int nq = B * NO / 4;
    for (int idx = t; idx < nq; idx += 512) {
        const int b = (idx * 4) / NO, o = (idx * 4) % NO;
        f32x4 r = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const float sv = sS[b][jj];
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] += sv * w2s[o + e][jj];
        }
        st16_sc1(p.part, ((g * B * NO) + idx * 4) * 4, &r);
    }
    sn_publish(p.sync, g, ep * 4 + 1);
    if (g != 0) return;
    sn_wait(p.sync, 32, ep * 4 + 1, &s_dead, p.sticky);
    if (t == 0) __hip_atomic_store(p.sync + SN_EPOCH, ep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int idx = t; idx < nq; idx += 512) {
        const int o = (idx * 4) % NO;
        f32x4 r = *reinterpret_cast<const f32x4*>(p.b2 + o);
        for (int gg = 0; gg < 32; ++gg) {
            const u32x4_t raw = ld16_sc1(p.part, ((gg * B * NO) + idx * 4) * 4);
            r += *reinterpret_cast<const f32x4*>(&raw);
        }
        *reinterpret_cast<f32x4*>(p.ctrl + idx * 4) = r;
    }
}
// A6 (B, 2, 256); W1 (512, 512), W2 (NO, 512) row-major [out][in]; U, S (B, 512); part >= 32 * B * NO floats.  B <= 64, NO % 4 == 0,
// NO <= 64, else 1.
TATT_API int tatt_stn_fc_fwd(const float* A6, const float* W1, const float* b1, const float* g1, const float* be1, float* rm1,
                             float* rv1, const float* W2, const float* b2, float* U, float* mean1, float* rstd1, float* S,
                             float* ctrl, float* part, unsigned* sync, int B, int NO, float eps, float momentum, hipStream_t st) {
    if (B < 1 || B > 64 || NO % 4 || NO < 4 || NO > 64) return 1;
    SnFcFwdP p = {A6, W1, b1, g1, be1, rm1, rv1, W2, b2, U, mean1, rstd1, S, ctrl, part, sync, B, NO, eps, momentum, tatt_sticky_ptr()};
    hipLaunchKernelGGL(stn_fc_fwd_kernel, dim3(32), dim3(512), 0, st, p);
    return LAUNCH_CHECK();
}

// backward of the same, every parameter gradient included.  Phase A (per work-group, its 16 features): dS = dctrl W2, ReLU mask,
// BatchNorm1d backward (local), dU -> global (write-through) and LDS; dW2 columns, dW1 rows, db1, dgamma1, dbeta1 (db2: work-group 0).
// Phase B (after all dU are published): dF = dU W1 for this work-group's 16 input features, scattered back to the NHWC map.
struct SnFcBwdP {
    const float* dctrl; const float* W2; const float* S; const float* U; const float* mean1; const float* rstd1; const float* g1;
    const float* W1; const float* A6;
    float* dW2; float* db2; float* dg1; float* dbe1; float* dW1; float* db1; float* dU; float* dA6; unsigned* sync;
    int B, NO; unsigned* sticky;
};
// (96 registers: two of its waves per SIMD fit beside the two of a resident query-GRU recurrence work-group with room to spare)
__global__ __launch_bounds__(512, 5) void stn_fc_bwd_kernel(SnFcBwdP p) {
    __shared__ float red[8][4][16][17];
    __shared__ float dcs[64][65];
    __shared__ __attribute__((aligned(16))) float w2s[64][16];
    __shared__ __attribute__((aligned(16))) float sS[64][16];
    __shared__ __attribute__((aligned(16))) float dus[64][16];
    __shared__ double sd[3][16][16];
    __shared__ float bc[2][16];
    __shared__ int s_dead;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), g = blockIdx.x, j0 = g * 16;
    const int i = lane & 15, q = lane >> 4, B = p.B, NO = p.NO, MT = (B + 15) >> 4;
    if (t == 0) s_dead = 0;
    const unsigned ep = __hip_atomic_load(p.sync + SN_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int idx = t; idx < B * NO; idx += 512) dcs[idx / NO][idx % NO] = p.dctrl[idx];
    for (int idx = t; idx < NO * 16; idx += 512) w2s[idx >> 4][idx & 15] = p.W2[(long)(idx >> 4) * FC_H + j0 + (idx & 15)];
    for (int idx = t; idx < 64 * 16; idx += 512) {
        const int b = idx >> 4;
        sS[b][idx & 15] = b < B ? p.S[(long)b * FC_H + j0 + (idx & 15)] : 0.f;
        dus[b][idx & 15] = 0.f;
    }
    __syncthreads();
    const int m = (t >> 4) & 15, j = t & 15;
    float dbn[4] = {0.f, 0.f, 0.f, 0.f}, uh[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < 256) {
        const float mu = p.mean1[j0 + j], rs = p.rstd1[j0 + j];
        double s1 = 0.0, s2 = 0.0;
        for (int mt = 0; mt < MT; ++mt) {
            const int b = mt * 16 + m;
            if (b >= B) break;
            float ds = 0.f;
            for (int o = 0; o < NO; ++o) ds += dcs[b][o] * w2s[o][j];
            const float d = sS[b][j] > 0.f ? 0.1f * ds : 0.f;
            const float h = (p.U[(long)b * FC_H + j0 + j] - mu) * rs;
            dbn[mt] = d; uh[mt] = h;
            s1 += (double)d; s2 += (double)d * (double)h;
        }
        sd[0][m][j] = s1; sd[1][m][j] = s2;
    }
    __syncthreads();
    if (t < 16) {
        double s1 = 0.0, s2 = 0.0;
        for (int mm = 0; mm < 16; ++mm) { s1 += sd[0][mm][t]; s2 += sd[1][mm][t]; }
        p.dbe1[j0 + t] = (float)s1; p.dg1[j0 + t] = (float)s2;
        bc[0][t] = (float)s1; bc[1][t] = (float)s2;
    }
    __syncthreads();
    if (t < 256) {
        const float inv = 1.f / (float)B, s1 = bc[0][j], s2 = bc[1][j];
        const float gr = p.g1[j0 + j] * p.rstd1[j0 + j];
        double sb = 0.0;
        for (int mt = 0; mt < MT; ++mt) {
            const int b = mt * 16 + m;
            if (b >= B) break;
            const float du = gr * (dbn[mt] - s1 * inv - uh[mt] * s2 * inv);
            dus[b][j] = du;
            sb += (double)du;
        }
        sd[2][m][j] = sb;
    }
    __syncthreads();
    if (t < 16) {
        double s = 0.0;
        for (int mm = 0; mm < 16; ++mm) s += sd[2][mm][t];
        p.db1[j0 + t] = (float)s;
    }
    if (t < B * 4) {                                                // dU rows of this work-group: 16 floats = 4 x 16 bytes per sample
        const int b = t >> 2, c4 = (t & 3) * 4;
        st16_sc1(p.dU, (b * FC_H + j0 + c4) * 4, &dus[b][c4]);
    }
    sn_publish(p.sync, g, ep * 4 + 1);
    // parameter gradients of this work-group's features while the others publish
    for (int idx = t; idx < NO * 16; idx += 512) {                  // dW2[o][j0 + j] = sum_b dctrl[b][o] S[b][j0 + j]
        const int o = idx >> 4, jj = idx & 15;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dcs[b][o] * sS[b][jj];
        p.dW2[(long)o * FC_H + j0 + jj] = s;
    }
    if (g == 0 && t < NO) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dcs[b][t];
        p.db2[t] = s;
    }
    {                                                               // dW1[j0 + j][k] = sum_b dU[b][j] F[b][k], thread = k
        float a[16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) a[jj] = 0.f;
        const int src = (t & 1) * 256 + (t >> 1);                   // F[b][k] = A6[b][k & 1][k >> 1]
        for (int b = 0; b < B; ++b) {
            const float f = p.A6[(long)b * FC_H + src];
#pragma unroll
            for (int jq = 0; jq < 4; ++jq) {
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(&dus[b][jq * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[jq * 4 + e] += d4[e] * f;
            }
        }
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) p.dW1[(long)(j0 + jj) * FC_H + t] = a[jj];
    }
    sn_wait(p.sync, 32, ep * 4 + 1, &s_dead, p.sticky);
    if (g == 0 && t == 0) __hip_atomic_store(p.sync + SN_EPOCH, ep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // phase B: dF[b][k] = sum_j dU[b][j] W1[j][k] for k = 16 g + n
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int jb = wave * 64 + 16 * u + 4 * q;
        float bw[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) bw[v] = p.W1[(long)(jb + v) * FC_H + 16 * g + i];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (mt >= MT) break;
            const int b = min(mt * 16 + i, B - 1);
            const u32x4_t raw = ld16_sc1(p.dU, (b * FC_H + jb) * 4);
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(&raw);
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[v], bw[v], acc[mt], 0, 0, 0);
        }
    }
    {
        const int col = lane & 15, rb = (lane >> 4) * 4;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) red[wave][mt][rb + rr][col] = acc[mt][rr];
    }
    __syncthreads();
    if (t < 256) {
        const int k = 16 * g + j;
        for (int mt = 0; mt < MT; ++mt) {
            const int b = mt * 16 + m;
            if (b >= B) break;
            float x = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) x += red[w][mt][m][j];
            p.dA6[(long)b * FC_H + (k & 1) * 256 + (k >> 1)] = x;
        }
    }
}
TATT_API int tatt_stn_fc_bwd(const float* dctrl, const float* W2, const float* S, const float* U, const float* mean1,
                             const float* rstd1, const float* g1, const float* W1, const float* A6, float* dW2, float* db2,
                             float* dg1, float* dbe1, float* dW1, float* db1, float* dU, float* dA6, unsigned* sync, int B, int NO,
                             hipStream_t st) {
    if (B < 1 || B > 64 || NO % 4 || NO < 4 || NO > 64) return 1;
    SnFcBwdP p = {dctrl, W2, S, U, mean1, rstd1, g1, W1, A6, dW2, db2, dg1, dbe1, dW1, db1, dU, dA6, sync, B, NO, tatt_sticky_ptr()};
    hipLaunchKernelGGL(stn_fc_bwd_kernel, dim3(32), dim3(512), 0, st, p);
    return LAUNCH_CHECK();
}

// Work-groups of each launch above that can be resident on the CURRENT device at once (occupancy per CU x CUs the process sees), the
// smallest over the kernels of a kind: out[0] the map launches (need g.G <= SN_MAXG = 128 co-resident), out[1] the fully connected
// launches (need 32).  On a partitioned / CU-masked device that cannot hold them the caller walks the operator chain instead.
TATT_API int tatt_stn_capacity(int* out) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
    const void* km[6] = {reinterpret_cast<const void*>(stn_bn_pool_fwd_kernel<2, 2>), reinterpret_cast<const void*>(stn_bn_pool_fwd_kernel<1, 2>),
                         reinterpret_cast<const void*>(stn_bn_pool_fwd_kernel<1, 1>), reinterpret_cast<const void*>(stn_bn_pool_bwd_kernel<2, 2>),
                         reinterpret_cast<const void*>(stn_bn_pool_bwd_kernel<1, 2>), reinterpret_cast<const void*>(stn_bn_pool_bwd_kernel<1, 1>)};
    const void* kf[2] = {reinterpret_cast<const void*>(stn_fc_fwd_kernel), reinterpret_cast<const void*>(stn_fc_bwd_kernel)};
    int lo = 1 << 30;
    for (int i = 0; i < 6; ++i) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, km[i], 256, 0) != hipSuccess) return 2;
        lo = n * prop.multiProcessorCount < lo ? n * prop.multiProcessorCount : lo;
    }
    out[0] = lo;
    lo = 1 << 30;
    for (int i = 0; i < 2; ++i) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kf[i], 512, 0) != hipSuccess) return 2;
        lo = n * prop.multiProcessorCount < lo ? n * prop.multiProcessorCount : lo;
    }
    out[1] = lo;
    return 0;
}
