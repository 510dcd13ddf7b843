// Shared device helpers for the TATT hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TATT_API extern "C" __attribute__((visibility("default")))

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LAUNCH_CHECK() (int)hipGetLastError()

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- activations ------------------------------------------------------------------------
// mish(x) = x * tanh(softplus(x)), softplus threshold 20  (reference model/tsrn.py:1056-1064)
// tanh(softplus(x)) = ((1+e^x)^2 - 1) / ((1+e^x)^2 + 1) = n / (n + 2) with n = e^x (e^x + 2): one v_exp + one v_rcp, and no
// cancellation for x << 0 (n -> e^x * 2).  Matches x*tanh(log1p(exp(x))) to fp32 round-off (|err| < 2e-7 on the value).
__device__ __forceinline__ float tanh_softplus_f(float x) {
    if (x > 20.f) return tanhf(x);           // reference: softplus(x) = x beyond the threshold; tanh(x>20) == 1.f in fp32
    const float e = __expf(x);
    const float n = e * (e + 2.f);
    return n * __builtin_amdgcn_rcpf(n + 2.f);
}
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(__expf(x)); }
__device__ __forceinline__ float mish_f(float x) { return x * tanh_softplus_f(x); }
// d mish / dx = tanh(sp) + x * (1 - tanh(sp)^2) * sigmoid(x)   (for x > 20: softplus' = 1)
__device__ __forceinline__ float mish_grad_f(float x) {
    const float t = tanh_softplus_f(x);
    const float sg = x > 20.f ? 1.f : __builtin_amdgcn_rcpf(1.f + __expf(-x));
    return t + x * (1.f - t * t) * sg;
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }
// fast gate functions for the latency-bound GRU recurrences: v_exp_f32 + v_rcp_f32 (abs. error ~1e-7, i.e. fp32 round-off class)
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_MISH = 2, ACT_TANH = 3 };

__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case ACT_RELU: return x > 0.f ? x : 0.f;
        case ACT_MISH: return mish_f(x);
        case ACT_TANH: return tanhf(x);
        default: return x;
    }
}
// derivative of act at pre-activation u
__device__ __forceinline__ float act_grad(float u, int act) {
    switch (act) {
        case ACT_RELU: return u > 0.f ? 1.f : 0.f;
        case ACT_MISH: return mish_grad_f(u);
        case ACT_TANH: { float t = tanhf(u); return 1.f - t * t; }
        default: return 1.f;
    }
}

// ---- counter-based RNG for dropout ---------------------------------------------------------
// keep-mask for element `idx` of dropout site `site`; `seed` is read from DEVICE memory so that a
// captured hipGraph draws fresh masks on every replay (the host bumps the word between replays).
__device__ __forceinline__ uint32_t mix32(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint32_t site, uint64_t idx, uint32_t thresh) {
    return mix32(seed ^ ((uint64_t)site << 40) ^ idx * 0xD1342543DE82EF95ull) >= thresh;
}
__device__ __forceinline__ uint32_t dropout_thresh(float p) { return (uint32_t)((double)p * 4294967296.0); }

// ---- wave / block reductions -----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// LDS hand-off between lanes of ONE wave (a 32-lane group, a wave-private tile): LDS operations of a wave execute in order, so
// only compiler/memory-model ordering is needed -- no work-group barrier.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- latency-tolerant strided sums ------------------------------------------------------------------------------
// The reduction kernels run at low occupancy, so a naive `for (...) s += p[i*stride]` serialises on load latency
// (~1 us per dependent L2 round trip).  U independent loads are issued per trip instead.
template <typename T, int U>
__device__ __forceinline__ double sum_strided(const T* __restrict__ p, int n, long stride) {
    double acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = 0.0;
    int i = 0;
    for (; i + U <= n; i += U) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[(long)(i + u) * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] += (double)v[u];
    }
    for (; i < n; ++i) acc[0] += (double)p[(long)i * stride];
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) s += acc[u];
    return s;
}
