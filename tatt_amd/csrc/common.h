// Shared device helpers for the TATT hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

#define TATT_API extern "C" __attribute__((visibility("default")))

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LAUNCH_CHECK() (int)hipGetLastError()

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Run `f` once per DEVICE at a call site (hipFuncSetAttribute and friends are per device: a process-wide std::once_flag would leave a
// second GPU of the same process without the attribute).  `f` runs under the site's lock, so a concurrent caller on the same device
// cannot launch before the attribute is set.
struct TattPerDevice { std::mutex mu; bool done[64] = {}; };
template <class F>
static inline void tatt_per_device(TattPerDevice& s, F f) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { f(); return; }
    std::lock_guard<std::mutex> lk(s.mu);
    if (!s.done[dev]) { f(); s.done[dev] = true; }
}

// Sticky error word of the launches that synchronise their work-groups in flight (persistent query-GRU recurrences, STN-head launches):
// one word per device, registered once by the host (tatt_set_sticky), never reset by the library.  A bounded spin that expires ORs a
// code into it (besides the launch's own error word); tatt_sync_guard traps on it.  Defined in elementwise.hip; null until registered.
unsigned* tatt_sticky_ptr();
#define TATT_STICKY_QGRU 1u
#define TATT_STICKY_STN 2u
__device__ __forceinline__ void tatt_raise_sticky(unsigned* sticky, unsigned code) {
    if (sticky) __hip_atomic_fetch_or(sticky, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- activations ------------------------------------------------------------------------
// mish(x) = x * tanh(softplus(x)), softplus threshold 20  (reference model/tsrn.py:1056-1064)
// tanh(softplus(x)) = ((1+e^x)^2 - 1) / ((1+e^x)^2 + 1) = n / (n + 2) with n = e^x (e^x + 2): one v_exp + one v_rcp, and no
// cancellation for x << 0 (n -> e^x * 2).  Matches x*tanh(log1p(exp(x))) to fp32 round-off (|err| < 2e-7 on the value).
__device__ __forceinline__ float tanh_softplus_f(float x) {
    // reference: softplus(x) = x beyond the threshold, and tanh(x > 20) == 1.f in fp32 -- a select, not a branch (the exponent is
    // clamped so the unselected lane stays finite)
    const float e = __expf(fminf(x, 20.f));
    const float n = e * (e + 2.f);
    return x > 20.f ? 1.f : n * __builtin_amdgcn_rcpf(n + 2.f);
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
// 32-bit counter hash: the murmur3 finaliser (a bijection of 2^32 with full avalanche) over the element index, keyed by the
// (seed, site) pair before the first and between the two multiplies -- 3 integer multiplies + 9 simple ops per element.  (Round 2
// used two 64-bit splitmix rounds: ~12 quarter-rate multiplies per element, a third of the instructions of the fused
// transformer-layer kernel.)  The keys depend on (seed, site) only: wave-uniform, computed on the scalar unit.
__device__ __forceinline__ uint32_t mix32(uint64_t z) {            // (key derivation / host-visible seed mixing)
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint32_t site, uint64_t idx, uint32_t thresh) {
    const uint32_t k0 = (uint32_t)seed ^ (site * 0x9E3779B9u);
    const uint32_t k1 = (uint32_t)(seed >> 32) + site * 0x85EBCA77u;
    uint32_t h = ((uint32_t)idx ^ k0) + (uint32_t)(idx >> 32) * 0x27D4EB2Fu;
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h += k1;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h >= thresh;
}
__device__ __forceinline__ uint32_t dropout_thresh(float p) { return (uint32_t)((double)p * 4294967296.0); }

// ---- DPP lane exchanges (VALU rate; __shfl_xor compiles to ds_bpermute_b32, an LDS-pipe instruction) ------------------------
// row = 16 consecutive lanes.  DPP_ROR(n): lane i of a row reads lane (i - n) mod 16 of its row; DPP_QXOR1 / 2: lane ^ 1 / ^ 2.
#define DPP_ROR(n) (0x120 + (n))
#define DPP_QXOR1 0xB1
#define DPP_QXOR2 0x4E
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a row, result in every lane
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<DPP_ROR(1)>(v); v += dpp_f<DPP_ROR(2)>(v); v += dpp_f<DPP_ROR(4)>(v); v += dpp_f<DPP_ROR(8)>(v);
    return v;
}
__device__ __forceinline__ float quad_sum(float v) { v += dpp_f<DPP_QXOR1>(v); v += dpp_f<DPP_QXOR2>(v); return v; }
__device__ __forceinline__ float quad_max(float v) { v = fmaxf(v, dpp_f<DPP_QXOR1>(v)); v = fmaxf(v, dpp_f<DPP_QXOR2>(v)); return v; }

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
