// GRU recurrences of the TATT hot path (gfx950).
//
// (1) gru32_*: the 10 small bidirectional GRUs of the sequential-residual blocks (hidden 32, input 64;
//     reference GruBlock, model/tsrn.py:1067-1084).  The input projection gi = x W_ih^T + b_ih is a
//     GEMM done beforehand (tatt_gemm); these kernels run only the latency-bound recurrence: one
//     32-lane group per (sequence, direction), the 96x32 recurrent matrix resident in VGPRs, h broadcast
//     through LDS, gates in registers.  Sequences are addressed on the NHWC token grid by strides so
//     the same kernel scans image columns (vertical, gru1) and image rows (horizontal, gru2).
// (2) qgru_*: the query-embedding GRU (hidden 512, input 1024) whose time axis is the SAMPLE axis
//     (reference InfoTransformer.forward, model/transformer_v2.py:201-221).  One launch
//     per time step: h W_hh^T on v_mfma_f32_16x16x4_f32 with K split over the 4 waves of a work-group
//     and the gate math fused in the epilogue.
#include "common.h"

struct SeqGeom {
    int nseq, T, s_in;
    long stride_hi, stride_lo, stride_t;   // token = (s / s_in)*stride_hi + (s % s_in)*stride_lo + t*stride_t
};
__device__ __forceinline__ long seq_base(const SeqGeom& g, int s) {
    return (long)(s / g.s_in) * g.stride_hi + (long)(s % g.s_in) * g.stride_lo;
}

// ------------------------------------------------------------------------------------------------
// small BiGRU forward.  gi: [tok][192] = [fwd r,z,n | rev r,z,n];  out: [tok][64] = [fwd h | rev h]
// ------------------------------------------------------------------------------------------------
// The recurrence is latency-bound: a step is ~0.3 us of arithmetic but its operands come from HBM / Infinity Cache
// (~1-2 us away), so each group keeps GRU_PF steps of input in flight in a register ring.  A 32-lane group lives inside
// one wave, so the LDS hand-off of h needs only wave-level ordering -- no work-group barrier couples the 8 groups.
#define GRU_PF 4
// Every load is UNCONDITIONAL (step indices clamped into the sequence; groups past the last sequence redo the last one, writing
// identical values to identical addresses) and nothing loaded lives in a register across the loop's back edge: the inputs of a
// group of GRU_PF steps are loaded while the group before it computes and are parked in LDS (each lane its own column) at the
// end of that group.  With a branch around a load, or a register ring carried around the loop, the compiler copies the ring at
// the back edge behind `s_waitcnt vmcnt(0)`, which drains the prefetch just issued (and, on gfx9, every store since): measured
// 1.5-2 us per step instead of 0.3-0.5.
//
// SAVE: also write gates[tok][dir*128 + {r, z, n, W_hn h + b_hn}*32 + j] for the backward pass (it then needs neither gi nor the
// 96x32 gate recomputation, which halves its arithmetic and its registers).
template <bool SAVE>
__global__ __launch_bounds__(256) void gru32_fwd_kernel(const float* __restrict__ gi,
                                                        const float* __restrict__ whh_f, const float* __restrict__ bhh_f,
                                                        const float* __restrict__ whh_r, const float* __restrict__ bhh_r,
                                                        float* __restrict__ out, float* __restrict__ gates, SeqGeom g) {
    __shared__ __attribute__((aligned(16))) float hs[8][32];
    __shared__ float pf[3 * GRU_PF][256];
    const int t = threadIdx.x, grp = t >> 5, j = t & 31;
    const int seq = min((int)blockIdx.x * 4 + (grp >> 1), g.nseq - 1), dir = grp & 1;
    const float* whh = dir ? whh_r : whh_f;
    const float* bhh = dir ? bhh_r : bhh_f;
    float wr[32], wz[32], wn[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        wr[k] = whh[(0 * 32 + j) * 32 + k];
        wz[k] = whh[(1 * 32 + j) * 32 + k];
        wn[k] = whh[(2 * 32 + j) * 32 + k];
    }
    const float br = bhh[j], bz = bhh[32 + j], bn = bhh[64 + j];
    const int T = g.T;
    const long st_t = dir ? -g.stride_t : g.stride_t;
    const long tok0 = seq_base(g, seq) + (dir ? (long)(T - 1) * g.stride_t : 0);          // token of step 0
    const float* gq = gi + dir * 96 + j;
    float* oq = out + dir * 32 + j;
    float* sq = gates + dir * 128 + j;
    auto fetch = [&](int step, float& a, float& b, float& c) {
        const float* q = gq + (tok0 + (long)min(step, T - 1) * st_t) * 192;
        a = q[0]; b = q[32]; c = q[64];
    };
    float h = 0.f;
    auto one_step = [&](int step, float gr, float gz, float gn) {
        hs[grp][j] = h;
        wave_lds_sync();
        float ar = br, az = bz, an = bn;
        const f32x4* hv = reinterpret_cast<const f32x4*>(hs[grp]);
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            f32x4 hh = hv[k4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ar = fmaf(wr[k4 * 4 + e], hh[e], ar);
                az = fmaf(wz[k4 * 4 + e], hh[e], az);
                an = fmaf(wn[k4 * 4 + e], hh[e], an);
            }
        }
        wave_lds_sync();
        const float r = sigmoid_fast(gr + ar);
        const float z = sigmoid_fast(gz + az);
        const float n = tanh_fast(gn + r * an);
        h = (1.f - z) * n + z * h;
        const long tok = tok0 + (long)step * st_t;
        oq[tok * 64] = h;
        if (SAVE) {
            float* q = sq + tok * 256;
            q[0] = r; q[32] = z; q[64] = n; q[96] = an;
        }
    };
    {
        float a, b, c;
#pragma unroll
        for (int u = 0; u < GRU_PF; ++u) {
            fetch(u, a, b, c);
            pf[u * 3 + 0][t] = a; pf[u * 3 + 1][t] = b; pf[u * 3 + 2][t] = c;
        }
    }
    int s0 = 0;
    for (; s0 + GRU_PF <= T; s0 += GRU_PF) {
        float nr[GRU_PF], nz[GRU_PF], nn[GRU_PF];
#pragma unroll
        for (int u = 0; u < GRU_PF; ++u) fetch(s0 + GRU_PF + u, nr[u], nz[u], nn[u]);
#pragma unroll
        for (int u = 0; u < GRU_PF; ++u) one_step(s0 + u, pf[u * 3 + 0][t], pf[u * 3 + 1][t], pf[u * 3 + 2][t]);
#pragma unroll
        for (int u = 0; u < GRU_PF; ++u) { pf[u * 3 + 0][t] = nr[u]; pf[u * 3 + 1][t] = nz[u]; pf[u * 3 + 2][t] = nn[u]; }
    }
    for (; s0 < T; ++s0) {                                   // T % GRU_PF leftover steps: plain loads
        float gr, gz, gn;
        fetch(s0, gr, gz, gn);
        one_step(s0, gr, gz, gn);
    }
}
TATT_API int tatt_gru32_fwd(const float* gi, const float* whh_f, const float* bhh_f, const float* whh_r,
                            const float* bhh_r, float* out, float* gates, int nseq, int T, int s_in, long stride_hi,
                            long stride_lo, long stride_t, hipStream_t st) {
    if (nseq <= 0 || T <= 0) return 0;
    SeqGeom g = {nseq, T, s_in, stride_hi, stride_lo, stride_t};
    if (gates)
        hipLaunchKernelGGL(gru32_fwd_kernel<true>, dim3(cdiv(nseq, 4)), dim3(256), 0, st, gi, whh_f, bhh_f, whh_r, bhh_r, out,
                           gates, g);
    else
        hipLaunchKernelGGL(gru32_fwd_kernel<false>, dim3(cdiv(nseq, 4)), dim3(256), 0, st, gi, whh_f, bhh_f, whh_r, bhh_r, out,
                           gates, g);
    return LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// small BiGRU backward (BPTT) from the gates the forward pass saved.
// writes dgi [tok][192], dgh [tok][192] (recurrent-side gate grads: n-gate scaled by r) and
// hprev [tok][64] (the h_{t-1} each step consumed) -- dW_hh = dgh^T hprev, dW_ih = dgi^T x, dx = dgi W_ih
// are GEMMs issued by the host afterwards.
// Per step only the chain  dh -> gate gradients -> (LDS) -> dh_{t-1} = dh z + W_hh^T [dr, dz, dn r]  is serial.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gru32_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ out,
                                                        const float* __restrict__ dout,
                                                        const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                                        float* __restrict__ dgi, float* __restrict__ dgh,
                                                        float* __restrict__ hprev, SeqGeom g) {
    __shared__ __attribute__((aligned(16))) float ds[8][96];
    __shared__ float pf[6 * GRU_PF][256];
    const int t = threadIdx.x, grp = t >> 5, j = t & 31;
    const int seq = min((int)blockIdx.x * 4 + (grp >> 1), g.nseq - 1), dir = grp & 1;
    const float* whh = dir ? whh_r : whh_f;
    float wt[96];
#pragma unroll
    for (int row = 0; row < 96; ++row) wt[row] = whh[row * 32 + j];
    const int T = g.T;
    const long st_t = dir ? -g.stride_t : g.stride_t;
    const long tok0 = seq_base(g, seq) + (dir ? (long)(T - 1) * g.stride_t : 0);          // token of (forward) step 0
    const float* sq = gates + dir * 128 + j;
    const float* oq = out + dir * 32 + j;
    const float* dq = dout + dir * 32 + j;
    float dhc = 0.f;   // gradient carried to h_{t-1}
    auto fetch = [&](int step, float& hp, float& r, float& z, float& n, float& an, float& go) {
        const int sc = max(step, 0);
        const long tok = tok0 + (long)sc * st_t;
        hp = oq[(tok0 + (long)max(sc - 1, 0) * st_t) * 64];    // raw: h_{-1} = 0 is selected where the value is used
        const float* q = sq + tok * 256;
        r = q[0]; z = q[32]; n = q[64]; an = q[96];
        go = dq[tok * 64];
    };
    auto one_step = [&](int step, float hp_raw, float r, float z, float n, float an, float go) {
        const long tok = tok0 + (long)step * st_t;
        const float hp = step > 0 ? hp_raw : 0.f;
        const float dh = dhc + go;
        const float dn = dh * (1.f - z);
        const float dz = dh * (hp - n);
        const float dnp = dn * (1.f - n * n);
        const float drp = dnp * an * r * (1.f - r);
        const float dzp = dz * z * (1.f - z);
        const float dghn = dnp * r;
        ds[grp][j] = drp; ds[grp][32 + j] = dzp; ds[grp][64 + j] = dghn;
        wave_lds_sync();
        {
            float* a = dgi + tok * 192 + dir * 96 + j;
            float* b = dgh + tok * 192 + dir * 96 + j;
            a[0] = drp; a[32] = dzp; a[64] = dnp;
            b[0] = drp; b[32] = dzp; b[64] = dghn;
            hprev[tok * 64 + dir * 32 + j] = hp;
        }
        // 96-term dot product split over 4 accumulators (a single dependent FMA chain would cost ~96 x 8 cycles per step)
        float c0 = dh * z, c1 = 0.f, c2 = 0.f, c3 = 0.f;
        const f32x4* dv = reinterpret_cast<const f32x4*>(ds[grp]);
#pragma unroll
        for (int k4 = 0; k4 < 24; ++k4) {
            f32x4 dd = dv[k4];
            c0 = fmaf(wt[k4 * 4 + 0], dd[0], c0); c1 = fmaf(wt[k4 * 4 + 1], dd[1], c1);
            c2 = fmaf(wt[k4 * 4 + 2], dd[2], c2); c3 = fmaf(wt[k4 * 4 + 3], dd[3], c3);
        }
        dhc = (c0 + c1) + (c2 + c3);
        wave_lds_sync();
    };
    {
        float a, b, c, d, e, f;
#pragma unroll
        for (int u = 0; u < GRU_PF; ++u) {
            fetch(T - 1 - u, a, b, c, d, e, f);
            pf[u * 6 + 0][t] = a; pf[u * 6 + 1][t] = b; pf[u * 6 + 2][t] = c; pf[u * 6 + 3][t] = d; pf[u * 6 + 4][t] = e;
            pf[u * 6 + 5][t] = f;
        }
    }
    int s0 = T - 1;
    for (; s0 >= GRU_PF - 1; s0 -= GRU_PF) {
        float nx[GRU_PF][6];
#pragma unroll
        for (int u = 0; u < GRU_PF; ++u) fetch(s0 - GRU_PF - u, nx[u][0], nx[u][1], nx[u][2], nx[u][3], nx[u][4], nx[u][5]);
#pragma unroll
        for (int u = 0; u < GRU_PF; ++u)
            one_step(s0 - u, pf[u * 6 + 0][t], pf[u * 6 + 1][t], pf[u * 6 + 2][t], pf[u * 6 + 3][t], pf[u * 6 + 4][t],
                     pf[u * 6 + 5][t]);
#pragma unroll
        for (int u = 0; u < GRU_PF; ++u)
#pragma unroll
            for (int k = 0; k < 6; ++k) pf[u * 6 + k][t] = nx[u][k];
    }
    for (; s0 >= 0; --s0) {                                  // T % GRU_PF leftover steps: plain loads
        float a, b, c, d, e, f;
        fetch(s0, a, b, c, d, e, f);
        one_step(s0, a, b, c, d, e, f);
    }
}
TATT_API int tatt_gru32_bwd(const float* gates, const float* out, const float* dout, const float* whh_f,
                            const float* whh_r, float* dgi, float* dgh, float* hprev, int nseq, int T, int s_in,
                            long stride_hi, long stride_lo, long stride_t, hipStream_t st) {
    if (nseq <= 0 || T <= 0) return 0;
    SeqGeom g = {nseq, T, s_in, stride_hi, stride_lo, stride_t};
    hipLaunchKernelGGL(gru32_bwd_kernel, dim3(cdiv(nseq, 4)), dim3(256), 0, st, gates, out, dout, whh_f, whh_r, dgi, dgh, hprev,
                       g);
    return LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// Second generation of the small BiGRU recurrences (round 4): ONE WAVE per (sequence, direction), two lanes per hidden unit.
// ------------------------------------------------------------------------------------------------
// The first generation (above: a 32-lane group per (sequence, direction), 96 recurrent weights per lane) spends a step waiting
// for LDS: the 96 (backward) / 32 (forward) broadcast values are read two ds_read_b128 at a time because the 235 VGPRs leave no
// room to have more in flight -- 12 dependent LDS round trips per backward step (disassembly: `s_waitcnt lgkmcnt(1)`,
// `lgkmcnt(0)` alternating through the dot product), ~0.9 us per step at T = 64 against ~0.2 us of arithmetic.  Here lane
// (j, half) of a wave owns hidden unit j and HALF of every dot product (48 weights per lane): all 12 (4) LDS reads of a step are
// in flight at once, the two halves meet through v_permlane32_swap (VALU, no LDS), and the loads of a step are split between the
// halves the same way (one 256-byte row per instruction, exchanged by the same swap).  Twice the waves for the same work, which
// is what a latency-bound recurrence wants.
//
// FRAGS (backward): instead of dgh [tok][192] and hprev [tok][64] (1 KB per token written here and read back by the weight-
// gradient pass, which then transposes 36 column tiles through LDS), the kernel leaves the operands of that pass in the order
// the bf16 matrix cores take them.  The contraction there runs over TOKENS, and a lane of the recurrence owns ONE channel over
// CONSECUTIVE time steps -- which is exactly an MFMA operand's "8 consecutive k per lane": 8 steps of one (sequence, direction)
// are one k-octet, 4 octets one K-step of v_mfma_f32_16x16x32_bf16.  Layout (dwords): frag[K-step c][slot 0..19][hi, lo][lane'
// = 16 kq + li][4], octet o = seq * T/8 + (t >> 3) = 4 c + kq, element e = t & 7 (token order, both directions); slots d*8 + {r0 r1
// z0 z1 n0 n1 gn0 gn1} (16 channels each; n = dgi's n gate, gn = dgh's: scaled by r), 16 + d*2 + {0, 1} = h_{t-1}.  1.25 KB
// per token, written as whole 16-byte lane operands (staged per wave in LDS for the 8 steps of a window).
#define GRU2_PF 8
// v_permlane32_swap_b32 a, b: lanes 32-63 of a <-> lanes 0-31 of b, i.e. a <- [a.lo | b.lo], b <- [a.hi | b.hi].  Inline asm on purpose:
// ROCm 7.2's __builtin_amdgcn_permlane32_swap returns its FIRST result for both elements of the pair (checked in the disassembly),
// so the exchange would silently be lost.  s_nop 1 covers the two wait states between a VALU write of an operand and the swap.
__device__ __forceinline__ void lane32_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float half_sum(float v) {          // v[l] + v[l ^ 32] in every lane
    float a = v, b = v;
    lane32_swap(a, b);
    return a + b;
}
__device__ __forceinline__ void half_both(float v, float& lo, float& hi) {   // lo = v of lane l & 31, hi = v of lane (l & 31) + 32
    lo = v; hi = v;
    lane32_swap(lo, hi);
}

// The kernels are bound by INSTRUCTION ISSUE, not by latency (PMC, profiles/r04_c_pmc_gru32_*: a wave spends 57 % of its cycles
// issuing, ~4 cycles per instruction, ~160 instructions per backward step), so the step is written for instruction count: token
// addresses are scalar (wave-uniform sequence / direction) and enter the loads / stores as an SGPR base + one per-lane byte offset
// that never changes; the dot products run as v_pk_fma_f32 on register pairs; nothing is predicated on the half (both halves hold
// identical values wherever only one would need to store: same value, same address).  32-bit token arithmetic: the host checks
// that every byte offset fits 32 bits.
typedef float gr_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float ldg32(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void stg32(float* base, unsigned byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
__device__ __forceinline__ int seq_base_i(const SeqGeom& g, int s) {
    return (s / g.s_in) * (int)g.stride_hi + (s % g.s_in) * (int)g.stride_lo;
}

template <bool SAVE>
__global__ __launch_bounds__(256) void gru32_fwd2_kernel(const float* __restrict__ gi,
                                                         const float* __restrict__ whh_f, const float* __restrict__ bhh_f,
                                                         const float* __restrict__ whh_r, const float* __restrict__ bhh_r,
                                                         float* __restrict__ out, float* __restrict__ gates, SeqGeom g) {
    __shared__ __attribute__((aligned(16))) float hs[4][32];
    __shared__ float pf[2 * GRU2_PF][256];
    const int t = threadIdx.x, lane = t & 63, j = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);       // wave-uniform: sequence, direction and every token address live in SGPRs
    const int sd = (int)blockIdx.x * 4 + wave;
    const int seq = min(sd >> 1, g.nseq - 1), dir = sd & 1;
    const float* whh = dir ? whh_r : whh_f;
    const float* bhh = dir ? bhh_r : bhh_f;
    gr_f32x2 wr[8], wz[8], wn[8];                                   // this half's 16 columns of the three gate rows of unit j, as pairs
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        wr[k] = (gr_f32x2){whh[(0 * 32 + j) * 32 + 16 * half + 2 * k], whh[(0 * 32 + j) * 32 + 16 * half + 2 * k + 1]};
        wz[k] = (gr_f32x2){whh[(1 * 32 + j) * 32 + 16 * half + 2 * k], whh[(1 * 32 + j) * 32 + 16 * half + 2 * k + 1]};
        wn[k] = (gr_f32x2){whh[(2 * 32 + j) * 32 + 16 * half + 2 * k], whh[(2 * 32 + j) * 32 + 16 * half + 2 * k + 1]};
    }
    const float br = half ? 0.f : bhh[j], bz = half ? 0.f : bhh[32 + j], bn = half ? 0.f : bhh[64 + j];   // biases enter once
    const int T = g.T;
    const int st_t = dir ? -(int)g.stride_t : (int)g.stride_t;
    const int tok0 = seq_base_i(g, seq) + (dir ? (T - 1) * (int)g.stride_t : 0);
    const unsigned lane4 = lane * 4, j4 = j * 4;
    const float* gq = gi + dir * 96;
    float* oq = out + dir * 32;
    float* sq = gates + dir * 128;
    auto fetch = [&](int step, float& a, float& b) {
        const float* q = gq + (unsigned)(tok0 + min(step, T - 1) * st_t) * 192u;     // (scalar)
        a = ldg32(q, lane4);                                        // half 0: r gate of unit j, half 1: z gate
        b = ldg32(q + 64, j4);                                      // n gate (both halves)
    };
    float h = 0.f;
    auto one_step = [&](int step, float grz, float gn) {
        hs[wave][j] = h;                                            // both halves hold the same h: same value, same address
        wave_lds_sync();
        gr_f32x2 ar = (gr_f32x2){br, 0.f}, az = (gr_f32x2){bz, 0.f}, an2 = (gr_f32x2){bn, 0.f};
        const f32x4* hv = reinterpret_cast<const f32x4*>(hs[wave] + 16 * half);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const f32x4 hh = hv[k4];
            const gr_f32x2 h01 = (gr_f32x2){hh[0], hh[1]}, h23 = (gr_f32x2){hh[2], hh[3]};
            ar = wr[2 * k4] * h01 + ar;  az = wz[2 * k4] * h01 + az;  an2 = wn[2 * k4] * h01 + an2;
            ar = wr[2 * k4 + 1] * h23 + ar;  az = wz[2 * k4 + 1] * h23 + az;  an2 = wn[2 * k4 + 1] * h23 + an2;
        }
        wave_lds_sync();
        const float sr = half_sum(ar[0] + ar[1]), sz = half_sum(az[0] + az[1]), an = half_sum(an2[0] + an2[1]);
        float gr, gz;
        half_both(grz, gr, gz);
        const float r = sigmoid_fast(gr + sr);
        const float z = sigmoid_fast(gz + sz);
        const float n = tanh_fast(gn + r * an);
        h = (1.f - z) * n + z * h;
        const unsigned tok = (unsigned)(tok0 + step * st_t);        // (scalar)
        stg32(oq + tok * 64u, j4, h);
        if (SAVE) {
            float* q = sq + tok * 256u;                             // [r | z | n | W_hn h + b_hn], 32 each: two 256-byte rows
            stg32(q, lane4, half ? z : r);
            stg32(q + 64, lane4, half ? an : n);
        }
    };
    {
        float a[GRU2_PF], b[GRU2_PF];                               // all loads of the first group in flight before any is parked
#pragma unroll
        for (int u = 0; u < GRU2_PF; ++u) fetch(u, a[u], b[u]);
#pragma unroll
        for (int u = 0; u < GRU2_PF; ++u) { pf[u * 2 + 0][t] = a[u]; pf[u * 2 + 1][t] = b[u]; }
    }
    int s0 = 0;
    for (; s0 + GRU2_PF <= T; s0 += GRU2_PF) {
        float na[GRU2_PF], nb[GRU2_PF];
#pragma unroll
        for (int u = 0; u < GRU2_PF; ++u) fetch(s0 + GRU2_PF + u, na[u], nb[u]);
#pragma unroll
        for (int u = 0; u < GRU2_PF; ++u) one_step(s0 + u, pf[u * 2 + 0][t], pf[u * 2 + 1][t]);
#pragma unroll
        for (int u = 0; u < GRU2_PF; ++u) { pf[u * 2 + 0][t] = na[u]; pf[u * 2 + 1][t] = nb[u]; }
    }
    for (; s0 < T; ++s0) {                                          // T % GRU2_PF leftover steps: plain loads
        float a, b;
        fetch(s0, a, b);
        one_step(s0, a, b);
    }
}
// every byte offset the second-generation kernels form must fit 32 bits: (largest token index + 1) * 1024 bytes (the gates rows)
static bool gru2_fits(int nseq, int T, int s_in, long stride_hi, long stride_lo, long stride_t) {
    if (s_in <= 0 || stride_hi < 0 || stride_lo < 0 || stride_t <= 0) return false;
    const long last = (long)((nseq - 1) / s_in) * stride_hi + (long)(s_in - 1) * stride_lo + (long)(T - 1) * stride_t;
    return last + 1 < (1L << 22);
}
TATT_API int tatt_gru32_fwd2(const float* gi, const float* whh_f, const float* bhh_f, const float* whh_r,
                             const float* bhh_r, float* out, float* gates, int nseq, int T, int s_in, long stride_hi,
                             long stride_lo, long stride_t, hipStream_t st) {
    if (nseq <= 0 || T <= 0) return 0;
    if (!gru2_fits(nseq, T, s_in, stride_hi, stride_lo, stride_t))          // huge token grids: the first generation (64-bit addressing)
        return tatt_gru32_fwd(gi, whh_f, bhh_f, whh_r, bhh_r, out, gates, nseq, T, s_in, stride_hi, stride_lo, stride_t, st);
    SeqGeom g = {nseq, T, s_in, stride_hi, stride_lo, stride_t};
    const dim3 grid(cdiv(2L * nseq, 4)), block(256);
    if (gates) hipLaunchKernelGGL(gru32_fwd2_kernel<true>, grid, block, 0, st, gi, whh_f, bhh_f, whh_r, bhh_r, out, gates, g);
    else hipLaunchKernelGGL(gru32_fwd2_kernel<false>, grid, block, 0, st, gi, whh_f, bhh_f, whh_r, bhh_r, out, gates, g);
    return LAUNCH_CHECK();
}

typedef __bf16 gr_bf16x2 __attribute__((ext_vector_type(2)));
// 8 fp32 values -> the hi (bf16(a)) or lo (bf16(a - hi)) halves, packed two per dword in element order
__device__ __forceinline__ f32x4 gr_split8(const float* v, bool want_lo) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const gr_f32x2 a = (gr_f32x2){v[e], v[e + 1]};
        const gr_bf16x2 hi = __builtin_convertvector(a, gr_bf16x2);
        const gr_bf16x2 lo = __builtin_convertvector(a - __builtin_convertvector(hi, gr_f32x2), gr_bf16x2);
        o[e >> 1] = want_lo ? __builtin_bit_cast(float, lo) : __builtin_bit_cast(float, hi);
    }
    return o;
}

template <bool FRAGS>
__global__ __launch_bounds__(256) void gru32_bwd2_kernel(const float* __restrict__ gates, const float* __restrict__ out,
                                                         const float* __restrict__ dout,
                                                         const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                                         float* __restrict__ dgi, float* __restrict__ dgh,
                                                         float* __restrict__ hprev, float* __restrict__ frag, SeqGeom g) {
    __shared__ __attribute__((aligned(16))) float ds[4][96];
    __shared__ float pf[4 * GRU2_PF][256];                         // parked loads of a group of steps: (r | z), (n | an), h_{t-1}, dout
    __shared__ float stg[FRAGS ? 4 : 1][4][8][32];                 // per wave: [drp, dzp, dnp, dghn][e = t & 7][j]
    const int t = threadIdx.x, lane = t & 63, j = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int sd = (int)blockIdx.x * 4 + wave;
    const int seq = min(sd >> 1, g.nseq - 1), dir = sd & 1;
    const float* whh = dir ? whh_r : whh_f;
    gr_f32x2 wt[24];                                                // this half's 48 rows of column j of W_hh, as pairs
#pragma unroll
    for (int k = 0; k < 24; ++k) wt[k] = (gr_f32x2){whh[(48 * half + 2 * k) * 32 + j], whh[(48 * half + 2 * k + 1) * 32 + j]};
    const int T = g.T;
    const int st_t = dir ? -(int)g.stride_t : (int)g.stride_t;
    const int tok0 = seq_base_i(g, seq) + (dir ? (T - 1) * (int)g.stride_t : 0);          // token of (forward) step 0
    const unsigned lane4 = lane * 4, j4 = j * 4;
    const float* sq = gates + dir * 128;
    const float* oq = out + dir * 32;
    const float* dq = dout + dir * 32;
    float* gq = dgi + dir * 96;
    float dhc = 0.f;                                                // gradient carried to h_{t-1} (both halves)
    auto fetch = [&](int step, float& a, float& b, float& c, float& d) {
        const int sc = max(step, 0);
        const unsigned tok = (unsigned)(tok0 + sc * st_t), tokp = (unsigned)(tok0 + max(sc - 1, 0) * st_t);    // (scalar)
        const float* q = sq + tok * 256u;
        a = ldg32(q, lane4);                                        // half 0: r, half 1: z
        b = ldg32(q + 64, lane4);                                   // half 0: n, half 1: W_hn h + b_hn
        c = ldg32(oq + tokp * 64u, j4);                             // h_{t-1} (raw: 0 is selected at step 0)
        d = ldg32(dq + tok * 64u, j4);                              // dout
    };
    auto one_step = [&](int step, float a, float b, float hp_raw, float go) {
        float r, z, n, an;
        half_both(a, r, z);
        half_both(b, n, an);
        const unsigned tok = (unsigned)(tok0 + step * st_t);        // (scalar)
        const float hp = step > 0 ? hp_raw : 0.f;
        const float dh = dhc + go;
        const float omz = 1.f - z;
        const float dn = dh * omz;
        const float dz = dh * (hp - n);
        const float dnp = dn * (1.f - n * n);
        const float drp = dnp * an * r * (1.f - r);
        const float dzp = dz * z * omz;
        const float dghn = dnp * r;
        const float rz = half ? dzp : drp;
        ds[wave][lane] = rz;                                        // [drp | dzp | dghn]
        ds[wave][64 + j] = dghn;
        wave_lds_sync();
        stg32(gq + tok * 192u, lane4, rz);
        stg32(gq + tok * 192u + 64, j4, dnp);
        if (FRAGS) {
            const int e = (dir ? T - 1 - step : step) & 7;          // (scalar)
            stg[wave][half][e][j] = rz;
            stg[wave][2 + half][e][j] = half ? dghn : dnp;
        } else {
            float* q = dgh + tok * 192u + dir * 96;
            stg32(q, lane4, rz);
            stg32(q + 64, j4, dghn);
            stg32(hprev + tok * 64u + dir * 32, j4, hp);
        }
        gr_f32x2 c01 = (gr_f32x2){half ? 0.f : dh * z, 0.f}, c23 = (gr_f32x2){0.f, 0.f};
        const f32x4* dv = reinterpret_cast<const f32x4*>(ds[wave] + 48 * half);
        // broadcast reads in two batches (8, then 4 behind the first FMAs) -- left to itself the scheduler issues them one by one (9
        // exposed LDS round trips per step); all 12 at once cost the third wave per SIMD (172 VGPRs)
        f32x4 da[8], db[4];
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) da[k4] = dv[k4];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            c01 = wt[2 * k4] * (gr_f32x2){da[k4][0], da[k4][1]} + c01;
            c23 = wt[2 * k4 + 1] * (gr_f32x2){da[k4][2], da[k4][3]} + c23;
        }
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) db[k4] = dv[8 + k4];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k4 = 4; k4 < 8; ++k4) {
            c01 = wt[2 * k4] * (gr_f32x2){da[k4][0], da[k4][1]} + c01;
            c23 = wt[2 * k4 + 1] * (gr_f32x2){da[k4][2], da[k4][3]} + c23;
        }
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            c01 = wt[16 + 2 * k4] * (gr_f32x2){db[k4][0], db[k4][1]} + c01;
            c23 = wt[16 + 2 * k4 + 1] * (gr_f32x2){db[k4][2], db[k4][3]} + c23;
        }
        dhc = half_sum((c01[0] + c01[1]) + (c23[0] + c23[1]));
        wave_lds_sync();
    };
    // the 8 steps s_hi .. s_hi - 7 just processed are one window of the sequence: leave its operand fragments.  Half 0 stores the hi
    // halves, half 1 the lo halves.  h_{t-1} of the window is still parked in the prefetch ring (own column, slot u = step s_hi - u).
    auto flush = [&](int s_hi) {
        const int tt0 = dir ? T - 1 - s_hi : s_hi - 7;              // first token-time of the window (a multiple of 8)
        const int o = seq * (T >> 3) + (tt0 >> 3), c = o >> 2, kq = o & 3;
        float* fb = frag + (long)c * (20 * 2 * 256) + half * 256 + (kq * 16 + (j & 15)) * 4;
        const int jt = j >> 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = stg[wave][q][e][j];
            *reinterpret_cast<f32x4*>(fb + (dir * 8 + q * 2 + jt) * 512) = gr_split8(v, half != 0);
        }
        {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {                           // token-time e of the window <-> ring slot u: step = s_hi - u
                const int u = dir ? e : 7 - e;
                const float raw = pf[u * 4 + 2][t];
                v[e] = (s_hi - u) > 0 ? raw : 0.f;
            }
            *reinterpret_cast<f32x4*>(fb + (16 + dir * 2 + jt) * 512) = gr_split8(v, half != 0);
        }
        wave_lds_sync();
    };
    {
        float a[GRU2_PF][4];
#pragma unroll
        for (int u = 0; u < GRU2_PF; ++u) fetch(T - 1 - u, a[u][0], a[u][1], a[u][2], a[u][3]);
#pragma unroll
        for (int u = 0; u < GRU2_PF; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) pf[u * 4 + k][t] = a[u][k];
    }
    int s0 = T - 1;
    for (; s0 >= GRU2_PF - 1; s0 -= GRU2_PF) {
        float nx[GRU2_PF][4];
#pragma unroll
        for (int u = 0; u < GRU2_PF; ++u) fetch(s0 - GRU2_PF - u, nx[u][0], nx[u][1], nx[u][2], nx[u][3]);
#pragma unroll
        for (int u = 0; u < GRU2_PF; ++u) one_step(s0 - u, pf[u * 4 + 0][t], pf[u * 4 + 1][t], pf[u * 4 + 2][t], pf[u * 4 + 3][t]);
        if (FRAGS) flush(s0);
#pragma unroll
        for (int u = 0; u < GRU2_PF; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) pf[u * 4 + k][t] = nx[u][k];
    }
    for (; s0 >= 0; --s0) {                                         // T % GRU2_PF leftover steps (never with FRAGS): plain loads
        float a, b, c, d;
        fetch(s0, a, b, c, d);
        one_step(s0, a, b, c, d);
    }
}
// frag == NULL: dgh (M, 192) and hprev (M, 64) are written as by tatt_gru32_bwd.  frag != NULL (needs T % 8 == 0 and
// nseq * T / 8 % 4 == 0; dgh / hprev unused): nseq * T / 32 * 10240 floats of operand fragments for tatt_gru_wgrad_frag.
TATT_API int tatt_gru32_bwd2(const float* gates, const float* out, const float* dout, const float* whh_f,
                             const float* whh_r, float* dgi, float* dgh, float* hprev, float* frag, int nseq, int T, int s_in,
                             long stride_hi, long stride_lo, long stride_t, hipStream_t st) {
    if (nseq <= 0 || T <= 0) return 0;
    if (frag && (T % 8 || ((long)nseq * (T / 8)) % 4)) return 1;
    if (!gru2_fits(nseq, T, s_in, stride_hi, stride_lo, stride_t)) {
        if (frag) return 3;
        return tatt_gru32_bwd(gates, out, dout, whh_f, whh_r, dgi, dgh, hprev, nseq, T, s_in, stride_hi, stride_lo, stride_t, st);
    }
    SeqGeom g = {nseq, T, s_in, stride_hi, stride_lo, stride_t};
    const dim3 grid(cdiv(2L * nseq, 4)), block(256);
    if (frag) hipLaunchKernelGGL(gru32_bwd2_kernel<true>, grid, block, 0, st, gates, out, dout, whh_f, whh_r, dgi, dgh, hprev, frag, g);
    else hipLaunchKernelGGL(gru32_bwd2_kernel<false>, grid, block, 0, st, gates, out, dout, whh_f, whh_r, dgi, dgh, hprev, frag, g);
    return LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// query GRU, one time step, both directions (blockIdx.z).
//   gi   [dir][Wb][3*HID]   time-invariant input projection (incl. b_ih)
//   whh  [dir] -> (3*HID, HID) row-major,  bhh [dir] -> (3*HID)
//   hprev[dir] -> (Wb, HID) or null (first step: h = 0)
//   hnew [dir] -> (Wb, HID);  gsave[dir] -> (4, Wb, HID) = r, z, n, (W_hn h + b_hn)
// work-group = 16 rows x 16 hidden units x 3 gates; wave w reduces k in [w*HID/4, (w+1)*HID/4).
// MFMA 16x16x4 operand map: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15]; each lane loads a
// float4 along k for A and B -- the (lane>>4, element) -> k assignment is the same on both sides, which
// is all the contraction needs.  C/D: col = lane&15, row = (lane>>4)*4 + reg.
// ------------------------------------------------------------------------------------------------
struct QStepP {
    const float* gi[2]; const float* whh[2]; const float* bhh[2]; const float* hprev[2];
    float* hnew[2]; float* gsave[2];
    int Wb, HID;
};
// RB: 16-row blocks one work-group walks with the SAME weight rows (B operands loaded once per trip, used RB times).  With one row
// block per work-group (the first version) the Wb / 16 work-groups that share a weight slice are consecutive in dispatch order = on
// different XCDs: at Wb = 128, HID = 1024 (LR 32x128) every XCD streamed all 25 MB of W_hh through its own L2 each step, 200 MB per
// step, 39.5 us (5 TB/s) -- profiles/r06_large_kernel_stats.txt.
template <int RB>
__global__ __launch_bounds__(256) void qgru_fwd_step_kernel(QStepP p) {
    __shared__ float red[RB][4][3][16][17];
    const int d = blockIdx.z;
    const int m0 = blockIdx.x * 16 * RB, j0 = blockIdx.y * 16;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int HID = p.HID;
    const float* hprev = p.hprev[d];
    const float* whh = p.whh[d];
    f32x4 acc[RB][3];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[rb][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (hprev) {
        const int i = lane & 15, q = lane >> 4;
        const int kspan = HID / 4;
        const int kbeg = wave * kspan;
        int arow[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) arow[rb] = min(m0 + 16 * rb + i, p.Wb - 1);
        // 4 sub-steps (64 k) per trip: 12 + 4 RB independent 16-byte loads are in flight before the first MFMA needs data -- the
        // step is latency-bound (one short wave per SIMD), so load-level parallelism is what matters.
        for (int kb = kbeg; kb < kbeg + kspan; kb += 64) {
            f32x4 a[RB][4], b[3][4];
#pragma unroll
            for (int sstep = 0; sstep < 4; ++sstep) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    b[g][sstep] = *reinterpret_cast<const f32x4*>(whh + ((long)g * HID + j0 + i) * HID + kb + 16 * sstep + 4 * q);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    a[rb][sstep] = *reinterpret_cast<const f32x4*>(hprev + (long)arow[rb] * HID + kb + 16 * sstep + 4 * q);
            }
#pragma unroll
            for (int sstep = 0; sstep < 4; ++sstep)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int g = 0; g < 3; ++g)
                            acc[rb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][sstep][u], b[g][sstep][u], acc[rb][g], 0, 0, 0);
        }
    }
    {
        const int col = lane & 15, r4 = (lane >> 4) * 4;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[rb][wave][g][r4 + r][col] = acc[rb][g][r];
    }
    __syncthreads();
    const int m = t >> 4, j = t & 15;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const long row = m0 + 16 * rb + m;
        if (row >= p.Wb) break;
        float gh[3];
#pragma unroll
        for (int g = 0; g < 3; ++g)
            gh[g] = red[rb][0][g][m][j] + red[rb][1][g][m][j] + red[rb][2][g][m][j] + red[rb][3][g][m][j] + p.bhh[d][g * HID + j0 + j];
        const float* gi = p.gi[d] + row * 3 * HID + j0 + j;
        const float r = sigmoid_fast(gi[0] + gh[0]);
        const float z = sigmoid_fast(gi[HID] + gh[1]);
        const float n = tanh_fast(gi[2 * HID] + r * gh[2]);
        const float hp = hprev ? hprev[row * HID + j0 + j] : 0.f;
        const float h = (1.f - z) * n + z * hp;
        p.hnew[d][row * HID + j0 + j] = h;
        float* gs = p.gsave[d];
        if (gs) {
            const long plane = (long)p.Wb * HID, o = row * HID + j0 + j;
            gs[o] = r; gs[plane + o] = z; gs[2 * plane + o] = n; gs[3 * plane + o] = gh[2];
        }
    }
}
// row blocks per work-group of the per-step kernels: as many as keep >= 256 work-groups (at most 4)
static int qgru_step_rb(int Wb, int HID) {
    const int nrb = cdiv(Wb, 16), per_rb = (HID / 16) * 2;
    int rb = 1;
    while (rb < 4 && nrb % (2 * rb) == 0 && (nrb / (2 * rb)) * per_rb >= 256) rb *= 2;
    return rb;
}
TATT_API int tatt_qgru_fwd_step(const float* gi0, const float* gi1, const float* whh0, const float* whh1,
                                const float* bhh0, const float* bhh1, const float* hprev0, const float* hprev1,
                                float* hnew0, float* hnew1, float* gsave0, float* gsave1, int Wb, int HID,
                                hipStream_t st) {
    if (HID % 256) return 1;          // each of the 4 waves reduces HID/4 columns in trips of 64
    QStepP p = {{gi0, gi1}, {whh0, whh1}, {bhh0, bhh1}, {hprev0, hprev1}, {hnew0, hnew1}, {gsave0, gsave1}, Wb, HID};
    const int rb = qgru_step_rb(Wb, HID);
    const dim3 grid(cdiv(cdiv(Wb, 16), rb), HID / 16, 2);
    if (rb == 4) hipLaunchKernelGGL(qgru_fwd_step_kernel<4>, grid, dim3(256), 0, st, p);
    else if (rb == 2) hipLaunchKernelGGL(qgru_fwd_step_kernel<2>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(qgru_fwd_step_kernel<1>, grid, dim3(256), 0, st, p);
    return LAUNCH_CHECK();
}

// backward step part 1 (element-wise): gate gradients of one time step, both directions.
//   dh_in  = dhseq_t + dhcarry   ->  dgi_acc += [dr', dz', dn'];  dgh_t = [dr', dz', dn'*r];  dhcarry = dh_in * z
struct QBwdGateP {
    const float* dhseq[2]; const float* gsave[2]; const float* hprev[2];
    float* dhcarry[2]; float* dgi_acc[2]; float* dgh[2];
    int Wb, HID, first;   // first: dhcarry holds garbage -> treat as zero
};
__global__ void qgru_bwd_gates_kernel(QBwdGateP p) {
    const int d = blockIdx.y;
    const long n_el = (long)p.Wb * p.HID;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_el) return;
    const long row = i / p.HID; const int j = i % p.HID;
    const float* gs = p.gsave[d];
    const float r = gs[i], z = gs[n_el + i], n = gs[2 * n_el + i], hn = gs[3 * n_el + i];
    const float hp = p.hprev[d] ? p.hprev[d][i] : 0.f;
    float dh = p.dhseq[d][i];
    if (!p.first) dh += p.dhcarry[d][i];
    const float dn = dh * (1.f - z), dz = dh * (hp - n);
    const float dnp = dn * (1.f - n * n);
    const float drp = dnp * hn * r * (1.f - r);
    const float dzp = dz * z * (1.f - z);
    const long g3 = row * 3 * p.HID + j;
    float* acc = p.dgi_acc[d];
    if (p.first) { acc[g3] = drp; acc[g3 + p.HID] = dzp; acc[g3 + 2 * p.HID] = dnp; }
    else { acc[g3] += drp; acc[g3 + p.HID] += dzp; acc[g3 + 2 * p.HID] += dnp; }
    float* dg = p.dgh[d];
    dg[g3] = drp; dg[g3 + p.HID] = dzp; dg[g3 + 2 * p.HID] = dnp * r;
    p.dhcarry[d][i] = dh * z;
}
TATT_API int tatt_qgru_bwd_gates(const float* dhseq0, const float* dhseq1, const float* gsave0, const float* gsave1,
                                 const float* hprev0, const float* hprev1, float* dhcarry0, float* dhcarry1,
                                 float* dgi_acc0, float* dgi_acc1, float* dgh0, float* dgh1, int Wb, int HID, int first,
                                 hipStream_t st) {
    QBwdGateP p = {{dhseq0, dhseq1}, {gsave0, gsave1}, {hprev0, hprev1}, {dhcarry0, dhcarry1}, {dgi_acc0, dgi_acc1},
                   {dgh0, dgh1}, Wb, HID, first};
    hipLaunchKernelGGL(qgru_bwd_gates_kernel, dim3(cdiv((long)Wb * HID, 256), 2), dim3(256), 0, st, p);
    return LAUNCH_CHECK();
}

// backward step part 2:  dhcarry[d] (Wb x HID) += dgh[d] (Wb x 3HID) @ whh[d] (3HID x HID)
// whh is passed TRANSPOSED (HID x 3HID) so that both MFMA operands are read as float4 along the contraction axis.
struct QBwdMmP { const float* dgh[2]; const float* whh[2]; float* dhcarry[2]; int Wb, HID; };
__global__ __launch_bounds__(256) void qgru_bwd_mm_kernel(QBwdMmP p) {
    __shared__ float red[4][16][17];
    const int d = blockIdx.z;
    const int m0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int HID = p.HID, K = 3 * p.HID;
    const float* dgh = p.dgh[d];
    const float* whh = p.whh[d];
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int i = lane & 15, q = lane >> 4;
    const int kspan = K / 4, kbeg = wave * kspan;
    const int arow = min(m0 + i, p.Wb - 1);
    for (int kb = kbeg; kb < kbeg + kspan; kb += 64) {         // 4 sub-steps per trip: 8 loads in flight (see forward step)
        f32x4 a[4], b[4];
#pragma unroll
        for (int sstep = 0; sstep < 4; ++sstep) {
            a[sstep] = *reinterpret_cast<const f32x4*>(dgh + (long)arow * K + kb + 16 * sstep + 4 * q);
            b[sstep] = *reinterpret_cast<const f32x4*>(whh + (long)(j0 + i) * K + kb + 16 * sstep + 4 * q);
        }
#pragma unroll
        for (int sstep = 0; sstep < 4; ++sstep)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[sstep][u], b[sstep][u], acc, 0, 0, 0);
    }
    {
        const int col = lane & 15, rb = (lane >> 4) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][rb + r][col] = acc[r];
    }
    __syncthreads();
    const int m = t >> 4, j = t & 15;
    if (m0 + m >= p.Wb) return;
    const float s = red[0][m][j] + red[1][m][j] + red[2][m][j] + red[3][m][j];
    p.dhcarry[d][(long)(m0 + m) * HID + j0 + j] += s;
}
TATT_API int tatt_qgru_bwd_mm(const float* dgh0, const float* dgh1, const float* whhT0, const float* whhT1,
                              float* dhcarry0, float* dhcarry1, int Wb, int HID, hipStream_t st) {
    if (HID % 256) return 1;           // each wave reduces 3*HID/4 rows in trips of 64
    QBwdMmP p = {{dgh0, dgh1}, {whhT0, whhT1}, {dhcarry0, dhcarry1}, Wb, HID};
    hipLaunchKernelGGL(qgru_bwd_mm_kernel, dim3(cdiv(Wb, 16), HID / 16, 2), dim3(256), 0, st, p);
    return LAUNCH_CHECK();
}

// backward step, fused: the matmul part of step s followed, on the same (row, hidden unit) tile, by the gate part of step s+1
//   dh = dhseq_next + dhcarry + dgh_cur @ whh   ->   dgi_acc += ..., dgh_next = ..., dhcarry = dh * z
// (dhcarry holds dh_s * z_s from the previous gate evaluation).  Halves the launches of the backward recurrence.
struct QBwdFusedP {
    const float* dgh_cur[2]; const float* whhT[2];
    const float* dhseq_next[2]; const float* gsave_next[2]; const float* hprev_next[2];
    float* dhcarry[2]; float* dgi_acc[2]; float* dgh_next[2];
    int Wb, HID;
};
// 8 waves split the 3*HID contraction and walk it in trips of 96 with 6 + 6 sixteen-byte loads per lane in flight (HID = 512: two
// trips to L2 instead of the six of a 4-wave / 64-wide walk -- the step is latency, not work: 47 of these follow each other)
#define QB_WAVES 8
template <int RB>                                  // row blocks per work-group, see qgru_fwd_step_kernel
__global__ __launch_bounds__(64 * QB_WAVES) void qgru_bwd_fused_kernel(QBwdFusedP p) {
    __shared__ float red[RB][QB_WAVES][16][17];
    const int d = blockIdx.z;
    const int m0 = blockIdx.x * 16 * RB, j0 = blockIdx.y * 16;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int HID = p.HID, K = 3 * p.HID;
    const float* dgh = p.dgh_cur[d];
    const float* whh = p.whhT[d];
    f32x4 acc0[RB], acc1[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) { acc0[rb] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[rb] = acc0[rb]; }
    const int i = lane & 15, q = lane >> 4;
    const int kspan = K / QB_WAVES, kbeg = wave * kspan;               // HID % 256 == 0: kspan is a multiple of 96
    const float* ap[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) ap[rb] = dgh + (long)min(m0 + 16 * rb + i, p.Wb - 1) * K + kbeg + 4 * q;
    const float* bp = whh + (long)(j0 + i) * K + kbeg + 4 * q;
    for (int kb = 0; kb < kspan; kb += 96) {
        f32x4 a[RB][6], b[6];
#pragma unroll
        for (int sstep = 0; sstep < 6; ++sstep) {
            b[sstep] = *reinterpret_cast<const f32x4*>(bp + kb + 16 * sstep);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a[rb][sstep] = *reinterpret_cast<const f32x4*>(ap[rb] + kb + 16 * sstep);
        }
#pragma unroll
        for (int sstep = 0; sstep < 6; ++sstep)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    if (u & 1) acc1[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][sstep][u], b[sstep][u], acc1[rb], 0, 0, 0);
                    else acc0[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][sstep][u], b[sstep][u], acc0[rb], 0, 0, 0);
                }
    }
    {
        const int col = lane & 15, r4 = (lane >> 4) * 4;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[rb][wave][r4 + r][col] = acc0[rb][r] + acc1[rb][r];
    }
    __syncthreads();
    if (t >= 256) return;
    const int m = t >> 4, j = t & 15;
    const long n_el = (long)p.Wb * HID;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const long row = m0 + 16 * rb + m;
        if (row >= p.Wb) break;
        const long e = row * HID + j0 + j;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < QB_WAVES; ++w) sum += red[rb][w][m][j];
        const float dh = p.dhseq_next[d][e] + p.dhcarry[d][e] + sum;
        const float* gs = p.gsave_next[d];
        const float r = gs[e], z = gs[n_el + e], n = gs[2 * n_el + e], hn = gs[3 * n_el + e];
        const float hp = p.hprev_next[d] ? p.hprev_next[d][e] : 0.f;
        const float dn = dh * (1.f - z), dz = dh * (hp - n);
        const float dnp = dn * (1.f - n * n);
        const float drp = dnp * hn * r * (1.f - r);
        const float dzp = dz * z * (1.f - z);
        const long g3 = row * 3 * HID + j0 + j;
        float* ga = p.dgi_acc[d];
        ga[g3] += drp; ga[g3 + HID] += dzp; ga[g3 + 2 * HID] += dnp;
        float* dg = p.dgh_next[d];
        dg[g3] = drp; dg[g3 + HID] = dzp; dg[g3 + 2 * HID] = dnp * r;
        p.dhcarry[d][e] = dh * z;
    }
}
TATT_API int tatt_qgru_bwd_fused(const float* dgh_cur0, const float* dgh_cur1, const float* whhT0, const float* whhT1,
                                 const float* dhseq_next0, const float* dhseq_next1, const float* gsave_next0,
                                 const float* gsave_next1, const float* hprev_next0, const float* hprev_next1,
                                 float* dhcarry0, float* dhcarry1, float* dgi_acc0, float* dgi_acc1, float* dgh_next0,
                                 float* dgh_next1, int Wb, int HID, hipStream_t st) {
    if (HID % 256) return 1;
    QBwdFusedP p = {{dgh_cur0, dgh_cur1}, {whhT0, whhT1}, {dhseq_next0, dhseq_next1}, {gsave_next0, gsave_next1},
                    {hprev_next0, hprev_next1}, {dhcarry0, dhcarry1}, {dgi_acc0, dgi_acc1}, {dgh_next0, dgh_next1}, Wb, HID};
    const int rb = qgru_step_rb(Wb, HID);
    const dim3 grid(cdiv(cdiv(Wb, 16), rb), HID / 16, 2);
    if (rb == 4) hipLaunchKernelGGL(qgru_bwd_fused_kernel<4>, grid, dim3(64 * QB_WAVES), 0, st, p);
    else if (rb == 2) hipLaunchKernelGGL(qgru_bwd_fused_kernel<2>, grid, dim3(64 * QB_WAVES), 0, st, p);
    else hipLaunchKernelGGL(qgru_bwd_fused_kernel<1>, grid, dim3(64 * QB_WAVES), 0, st, p);
    return LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// backward recurrence of the query GRU as ONE persistent launch (steps s0 .. s1-1 of the T-1 fused steps above).
//
// The per-step launches above are bound by the launch boundary (9.4 us start to start for 0.2 GFLOP), 47 of them at the exposed end
// of the training step.  Here a work-group keeps its (16 rows, 16 hidden units) tile for the whole chain: its slice of W_hh^T lives
// in registers (48 per lane), dhcarry and the three dgi accumulators of its tile too; what crosses work-groups per step is only
// dgh (this step's recurrent-side gate gradients): a tile's contraction needs the 16 x 3*HID rows its ROW BLOCK's HID/16 work-groups
// produced one step earlier -- so synchronisation is per (row block, direction) GROUP of HID/16 members, not grid-wide.  Block b is
// member b / NG of group b % NG: with NG = 8 groups (W = 64) a group's members are the blocks HIP places on ONE XCD (observed
// placement b % 8, MI355X_MICROARCH.md) -- a speed bonus only; correctness uses the agent-scope forms of that guide:
//   producer: 16-byte `sc1` (write-through) stores of its dgh tile -> s_waitcnt vmcnt(0) -> barrier -> ONE `sc1` flag store
//             (flags[group][member] = number of steps published);
//   consumer: wave 0 polls the group's flags (one 4-byte `sc1` load per lane = per member) -> barrier -> `sc1` loads of the rows.
// Every spin is bounded by the wall clock: on expiry the kernel raises flags[QCH_ERR] and runs on without waiting (results are then
// garbage but the launch ends); tatt_qgru_bwd_chain's caller reads the word when it next synchronises.
// Residency: the NG * HID/16 <= 256 work-groups must be co-resident (one per CU at 512 threads / <= 128 VGPRs leaves room beside it).
// ------------------------------------------------------------------------------------------------
#define QCH_ERR 1023                       // index of the error word in the sync buffer (1024 words)
#define QCH_SPIN_TICKS 200000000L          // 2 s of the 100 MHz wall clock: far beyond any stall another launch (a collective waiting for a late rank) can cause
struct QChainP {
    float* dgh[2]; const float* whhT[2]; const float* dhseq[2]; const float* gsave[2]; const float* hbuf[2];
    float* dhcarry[2]; float* dgi_acc[2];
    unsigned* flags;
    int T, Wb, s0, s1, transposed;
    float* xch[2];      // SB: the exchanged tensor in matrix-core operand form, (T, Wb, 3*HID / 8) x {8 bf16 hi, 8 bf16 lo}
    unsigned* sticky;   // the device's sticky error word (common.h), may be null
};
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 qc_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void qc_split8(const float* v, qc_bf16x8& hi, qc_bf16x8& lo) {
    hi = __builtin_bit_cast(qc_bf16x8, gr_split8(v, false));
    lo = __builtin_bit_cast(qc_bf16x8, gr_split8(v, true));
}
__device__ __forceinline__ f32x4 qc_mma3(const qc_bf16x8& ah, const qc_bf16x8& al, const qc_bf16x8& bh, const qc_bf16x8& bl, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
}
__device__ __forceinline__ void qc_split1(float x, unsigned short& hi, unsigned short& lo) {
    const __bf16 h = (__bf16)x;
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, (__bf16)(x - (float)h));
}
// SB = true: the recurrent product on the bf16 matrix cores with split operands (a = hi + lo, a b ~ hi hi + hi lo + lo hi, fp32
// accumulation: 2^-16 relative per product).  The fp32 MFMA form of this product costs a third of the CHIP's fp32 matrix throughput
// for the duration of the chain (0.2 GFLOP per step on 256 CUs) and slows whatever runs beside it; the split form needs a fifth of
// the matrix-core time.  What the work-groups exchange is then already in operand form: the producer of an element splits it ONCE and
// publishes 8-element groups as {8 bf16 hi, 8 bf16 lo} (the same 4 bytes per element), consumers load fragments with no conversion.
template <bool SB>
__global__ __launch_bounds__(512) void qgru_bwd_chain_kernel(QChainP p) {
    constexpr int HID = 512, K = 3 * HID, NM = HID / 16, KS = K / 8 / 16;      // KS = 12 sixteen-byte fragments per lane
    __shared__ float red[8][16][17];
    __shared__ __attribute__((aligned(16))) float stage[16][3][16];
    __shared__ __attribute__((aligned(16))) unsigned short stage_h[SB ? 16 : 1][3][16], stage_l[SB ? 16 : 1][3][16];
    __shared__ int s_dead;
    const int NG = (p.Wb / 16) * 2;
    const int grp = blockIdx.x % NG, mem = blockIdx.x / NG;
    const int d = grp / (p.Wb / 16), m0 = (grp % (p.Wb / 16)) * 16, j0 = mem * 16;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i = lane & 15, q = lane >> 4;
    const long plane = (long)p.Wb * HID;
    // W_hh^T rows j0 .. j0+15, this wave's eighth of the contraction: resident for the whole chain
    f32x4 b[SB ? 1 : KS];
    qc_bf16x8 bh[SB ? 6 : 1], bl[SB ? 6 : 1];
    if constexpr (SB) {
        // operand blocks of 32 k: lane (i, q) holds k = wave * 192 + 32 t + 8 q .. + 7 of column j0 + i
#pragma unroll
        for (int tt = 0; tt < 6; ++tt) {
            float v[8];
            const int k0 = wave * (K / 8) + 32 * tt + 8 * q;
            if (p.transposed) {
                const f32x4 lo4 = *reinterpret_cast<const f32x4*>(p.whhT[d] + (long)(j0 + i) * K + k0);
                const f32x4 hi4 = *reinterpret_cast<const f32x4*>(p.whhT[d] + (long)(j0 + i) * K + k0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = lo4[e]; v[4 + e] = hi4[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = p.whhT[d][(long)(k0 + e) * HID + j0 + i];
            }
            qc_split8(v, bh[tt], bl[tt]);
        }
    } else if (p.transposed) {
        const float* bp = p.whhT[d] + (long)(j0 + i) * K + wave * (K / 8) + 4 * q;
#pragma unroll
        for (int u = 0; u < KS; ++u) b[u] = *reinterpret_cast<const f32x4*>(bp + 16 * u);
    } else {                                                        // W_hh as the parameter stores it (3*HID, HID): column j0 + i
        const float* bp = p.whhT[d] + (long)(wave * (K / 8) + 4 * q) * HID + j0 + i;
#pragma unroll
        for (int u = 0; u < KS; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) b[u][v] = bp[(long)(16 * u + v) * HID];
    }
    // epilogue ownership (threads 0..255): element (m0 + m, j0 + j)
    const int m = (t >> 4) & 15, j = t & 15;
    const long e = (long)(m0 + m) * HID + j0 + j, g3 = (long)(m0 + m) * K + j0 + j;
    float dhc = 0.f, ga0 = 0.f, ga1 = 0.f, ga2 = 0.f;
    if (t < 256) { dhc = p.dhcarry[d][e]; ga0 = p.dgi_acc[d][g3]; ga1 = p.dgi_acc[d][g3 + HID]; ga2 = p.dgi_acc[d][g3 + 2 * HID]; }
    if (t == 0) s_dead = 0;
    unsigned* flags = p.flags + grp * 64;
    const int a_off = ((m0 + i) * K + wave * (K / 8) + 4 * q) * 4;             // byte offset of this lane's A fragments in a dgh step
    for (int s = p.s0; s < p.s1; ++s) {
        const int cur = d ? s : p.T - 1 - s, nxt = d ? s + 1 : p.T - 2 - s;
        // inputs of the gate part do not depend on the chain: in flight before the wait
        float in_dh = 0.f, r = 0.f, z = 0.f, n = 0.f, hn = 0.f, hp = 0.f;
        if (t < 256) {
            const float* gs = p.gsave[d] + (long)nxt * 4 * plane;
            in_dh = p.dhseq[d][(long)nxt * plane + e];
            r = gs[e]; z = gs[plane + e]; n = gs[2 * plane + e]; hn = gs[3 * plane + e];
            hp = p.hbuf[d][(long)(d ? nxt + 1 : nxt) * plane + e];
        }
        if (s > p.s0) {                                                         // the group's members have published step s-1
            if (wave == 0 && !s_dead) {
                const long t0 = wall_clock64();
                int it = 0;
                for (;;) {
                    const unsigned f = lane < NM ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
                    if (__builtin_amdgcn_ballot_w64(f < (unsigned)s) == 0) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++it & 63) == 0 && wall_clock64() - t0 > QCH_SPIN_TICKS) {
                        if (lane == 0) { s_dead = 1; __hip_atomic_store(p.flags + QCH_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); tatt_raise_sticky(p.sticky, TATT_STICKY_QGRU); }
                        break;
                    }
                }
            }
            __syncthreads();
        }
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        if constexpr (SB) {
            qc_bf16x8 ah[6], al[6];
            if (s == 0) {
                // the first step's dgh comes from tatt_qgru_bwd_gates in fp32 only: split here, once
                const float* src = p.dgh[d] + (long)cur * p.Wb * K + (long)(m0 + i) * K + wave * (K / 8) + 8 * q;
                f32x4 raw[12];
#pragma unroll
                for (int tt = 0; tt < 6; ++tt) {
                    raw[2 * tt] = *reinterpret_cast<const f32x4*>(src + 32 * tt);
                    raw[2 * tt + 1] = *reinterpret_cast<const f32x4*>(src + 32 * tt + 4);
                }
#pragma unroll
                for (int tt = 0; tt < 6; ++tt) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = raw[2 * tt][e]; v[4 + e] = raw[2 * tt + 1][e]; }
                    qc_split8(v, ah[tt], al[tt]);
                }
            } else {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.xch[d] + (long)cur * p.Wb * K, 0, p.Wb * K * 4, 0x00020000);
                const int off = (m0 + i) * K * 4 + (wave * 24 + q) * 32;
#pragma unroll
                for (int tt = 0; tt < 6; ++tt) {
                    ah[tt] = __builtin_bit_cast(qc_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 128 * tt, 0, 16 /* sc1 */));
                    al[tt] = __builtin_bit_cast(qc_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 128 * tt + 16, 0, 16 /* sc1 */));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < 6; ++tt) {
                if (tt & 1) acc1 = qc_mma3(ah[tt], al[tt], bh[tt], bl[tt], acc1);
                else acc0 = qc_mma3(ah[tt], al[tt], bh[tt], bl[tt], acc0);
            }
        } else {
            f32x4 a[KS];
            {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.dgh[d] + (long)cur * p.Wb * K, 0, p.Wb * K * 4, 0x00020000);
#pragma unroll
                for (int u = 0; u < KS; ++u)
                    a[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, a_off + 64 * u, 0, 16 /* sc1 */));
            }
            __builtin_amdgcn_sched_barrier(0);                                  // all twelve loads in flight before the first MFMA
#pragma unroll
            for (int u = 0; u < KS; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    if (v & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][v], b[u][v], acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][v], b[u][v], acc0, 0, 0, 0);
                }
        }
        {
            const int col = lane & 15, rb = (lane >> 4) * 4;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) red[wave][rb + rr][col] = acc0[rr] + acc1[rr];
        }
        __syncthreads();
        if (t < 256) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) sum += red[w][m][j];
            const float dh = in_dh + dhc + sum;
            const float dn = dh * (1.f - z), dz = dh * (hp - n);
            const float dnp = dn * (1.f - n * n);
            const float drp = dnp * hn * r * (1.f - r);
            const float dzp = dz * z * (1.f - z);
            ga0 += drp; ga1 += dzp; ga2 += dnp;
            const float g0 = drp, g1 = dzp, g2 = dnp * r;
            if constexpr (SB) {
                qc_split1(g0, stage_h[m][0][j], stage_l[m][0][j]);
                qc_split1(g1, stage_h[m][1][j], stage_l[m][1][j]);
                qc_split1(g2, stage_h[m][2][j], stage_l[m][2][j]);
                float* dg = p.dgh[d] + (long)nxt * p.Wb * K + g3;              // fp32 copy: the W_hh gradient GEMM reads it later
                dg[0] = g0; dg[HID] = g1; dg[2 * HID] = g2;
            } else {
                stage[m][0][j] = g0; stage[m][1][j] = g1; stage[m][2][j] = g2;
            }
            dhc = dh * z;
        }
        __syncthreads();
        if (t < 192) {                                                          // 16 rows x 3 gates x 4 sixteen-byte pieces
            if constexpr (SB) {
                const int row = t / 12, gate = (t % 12) >> 2, grp8 = (t >> 1) & 1, hl = t & 1;
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(hl ? &stage_l[row][gate][grp8 * 8] : &stage_h[row][gate][grp8 * 8]);
                const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(p.xch[d] + (long)nxt * p.Wb * K, 0, p.Wb * K * 4, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(v, ws, (m0 + row) * K * 4 + ((gate * HID + j0) / 8 + grp8) * 32 + hl * 16, 0, 16 /* sc1 */);
            } else {
                const int row = t / 12, gate = (t % 12) >> 2, j4 = (t & 3) * 4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(&stage[row][gate][j4]);
                const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(p.dgh[d] + (long)nxt * p.Wb * K, 0, p.Wb * K * 4, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), ws, ((m0 + row) * K + gate * HID + j0 + j4) * 4, 0, 16 /* sc1 */);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (t == 0) __hip_atomic_store(flags + mem, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t < 256) { p.dhcarry[d][e] = dhc; p.dgi_acc[d][g3] = ga0; p.dgi_acc[d][g3 + HID] = ga1; p.dgi_acc[d][g3 + 2 * HID] = ga2; }
}
// dgh (2 pointers: (T, Wb, 3*HID) per direction, the first step's slot filled by tatt_qgru_bwd_gates), hbuf* = h_prev of time t for
// direction 0 at hbuf0 + t*Wb*HID and of time t for direction 1 at hbuf1 + (t+1)*Wb*HID (the zero-slot layout of the forward);
// sync: 1024 words, zeroed here when s0 == 0; sync[1023] != 0 afterwards = a bounded spin expired (results invalid).
// whhT*: transposed != 0: W_hh^T (HID, 3*HID); 0: W_hh itself (3*HID, HID) -- the slices are read once per launch either way.
// xch* non-NULL ((T, Wb, 3*HID) floats of workspace per direction): the split-bf16 form (see the kernel); NULL: exact fp32 products.
// Returns 1 for geometries it does not take (HID != 512, Wb % 16, more than 256 work-groups): use the per-step entry points.
TATT_API int tatt_qgru_bwd_chain(float* dgh0, float* dgh1, const float* whhT0, const float* whhT1, const float* dhseq0,
                                 const float* dhseq1, const float* gsave0, const float* gsave1, const float* hbuf0,
                                 const float* hbuf1, float* dhcarry0, float* dhcarry1, float* dgi_acc0, float* dgi_acc1,
                                 unsigned* sync, int T, int Wb, int HID, int s0, int s1, int transposed, float* xch0, float* xch1,
                                 hipStream_t st) {
    if (HID != 512 || Wb % 16 || Wb <= 0 || (Wb / 16) * 2 * (HID / 16) > 256 || (Wb / 16) * 2 > 15) return 1;
    if (s0 < 0 || s1 > T - 1 || s0 >= s1) return s0 == s1 ? 0 : 2;
    if ((xch0 == nullptr) != (xch1 == nullptr)) return 1;
    if (s0 == 0 && hipMemsetAsync(sync, 0, 1024 * sizeof(unsigned), st) != hipSuccess) return 3;
    QChainP p = {{dgh0, dgh1}, {whhT0, whhT1}, {dhseq0, dhseq1}, {gsave0, gsave1}, {hbuf0, hbuf1}, {dhcarry0, dhcarry1},
                 {dgi_acc0, dgi_acc1}, sync, T, Wb, s0, s1, transposed, {xch0, xch1}, tatt_sticky_ptr()};
    if (xch0) hipLaunchKernelGGL(qgru_bwd_chain_kernel<true>, dim3((Wb / 16) * 2 * (HID / 16)), dim3(512), 0, st, p);
    else hipLaunchKernelGGL(qgru_bwd_chain_kernel<false>, dim3((Wb / 16) * 2 * (HID / 16)), dim3(512), 0, st, p);
    return LAUNCH_CHECK();
}

// forward recurrence of the query GRU as one persistent launch (time steps s0 .. s1-1; direction 0 visits time s, direction 1 time
// T-1-s): same groups, flags and hand-off forms as qgru_bwd_chain_kernel; what crosses work-groups per step is h (16 x HID rows per
// group).  hbuf* is the zero-slot layout of the host: h of time t at hbuf0 + (t+1)*Wb*HID (slot 0 = zeros) for direction 0 and at
// hbuf1 + t*Wb*HID (slot T = zeros) for direction 1, so h_prev of every step is a valid pointer.  The W_hh rows of the tile (3 gates x
// 16 units, this wave's eighth of the contraction), the tile's gi, b_hh and its own h stay in registers.
struct QFChainP {
    const float* gi[2]; const float* whh[2]; const float* bhh[2]; float* hbuf[2]; float* gsave[2];
    unsigned* flags;
    int T, Wb, s0, s1;
    // gi == NULL: the input projection is computed here, once per launch: gi = x W_ih^T + b_ih for the tile (x (Wb, IN), wih (3*HID, IN))
    const float* x; const float* wih[2]; const float* bih[2]; int IN;
    // q != NULL: h also leaves in the layout the TP interpreter reads, q[n][d*Hh + j / C][w][j % C] (n = time, w = row, j = unit)
    float* q; int C;
    float* xch[2];      // SB: h in matrix-core operand form, (T+1 slots as hbuf, Wb, HID / 8) x {8 bf16 hi, 8 bf16 lo}
    unsigned* sticky;   // the device's sticky error word (common.h), may be null
};
template <bool SB>
__global__ __launch_bounds__(512) void qgru_fwd_chain_kernel(QFChainP p) {
    constexpr int HID = 512, NM = HID / 16;
    __shared__ float red[8][3][16][17];
    __shared__ __attribute__((aligned(16))) float stage[16][16];
    __shared__ __attribute__((aligned(16))) unsigned short stage_h[SB ? 16 : 1][16], stage_l[SB ? 16 : 1][16];
    __shared__ int s_dead;
    const int NG = (p.Wb / 16) * 2;
    const int grp = blockIdx.x % NG, mem = blockIdx.x / NG;
    const int d = grp / (p.Wb / 16), m0 = (grp % (p.Wb / 16)) * 16, j0 = mem * 16;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i = lane & 15, q = lane >> 4;
    const long plane = (long)p.Wb * HID;
    const int m = (t >> 4) & 15, j = t & 15;
    const long e = (long)(m0 + m) * HID + j0 + j;
    float gi0 = 0.f, gi1 = 0.f, gi2 = 0.f, bh0 = 0.f, bh1 = 0.f, bh2 = 0.f, hown = 0.f;
    if (!p.gi[0]) {
        // input projection of this tile: 16 rows x (3 gates x 16 units), contraction over IN split over the 8 waves in trips of 128
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int IN = p.IN, span = IN / 8;
        for (int kb = wave * span; kb < (wave + 1) * span; kb += 128) {
            f32x4 a[8], w[3][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = *reinterpret_cast<const f32x4*>(p.x + (long)(m0 + i) * IN + kb + 16 * u + 4 * q);
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    w[g][u] = *reinterpret_cast<const f32x4*>(p.wih[d] + ((long)g * HID + j0 + i) * IN + kb + 16 * u + 4 * q);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][v], w[g][u][v], acc[g], 0, 0, 0);
        }
        const int col = lane & 15, rb = (lane >> 4) * 4;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) red[wave][g][rb + rr][col] = acc[g][rr];
        __syncthreads();
        if (t < 256) {
            gi0 = p.bih[d][j0 + j]; gi1 = p.bih[d][HID + j0 + j]; gi2 = p.bih[d][2 * HID + j0 + j];
#pragma unroll
            for (int w = 0; w < 8; ++w) { gi0 += red[w][0][m][j]; gi1 += red[w][1][m][j]; gi2 += red[w][2][m][j]; }
        }
        __syncthreads();
    }
    f32x4 b[SB ? 1 : 3][4];
    qc_bf16x8 bh[SB ? 3 : 1][2], bl[SB ? 3 : 1][2];
    if constexpr (SB) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {                        // operand blocks of 32 k: lane (i, q) holds k = wave*64 + 32 tt + 8 q .. + 7
                const float* src = p.whh[d] + ((long)g * HID + j0 + i) * HID + wave * 64 + 32 * tt + 8 * q;
                const f32x4 lo4 = *reinterpret_cast<const f32x4*>(src), hi4 = *reinterpret_cast<const f32x4*>(src + 4);
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = lo4[e]; v[4 + e] = hi4[e]; }
                qc_split8(v, bh[g][tt], bl[g][tt]);
            }
    } else {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                b[g][u] = *reinterpret_cast<const f32x4*>(p.whh[d] + ((long)g * HID + j0 + i) * HID + wave * 64 + 16 * u + 4 * q);
    }
    if (t < 256) {
        if (p.gi[0]) {
            const float* gi = p.gi[d] + (long)(m0 + m) * 3 * HID + j0 + j;
            gi0 = gi[0]; gi1 = gi[HID]; gi2 = gi[2 * HID];
        }
        bh0 = p.bhh[d][j0 + j]; bh1 = p.bhh[d][HID + j0 + j]; bh2 = p.bhh[d][2 * HID + j0 + j];
        const int tp = d ? p.T - 1 - p.s0 + 1 : p.s0;                           // slot of h_prev of the first step of this launch
        hown = p.hbuf[d][(long)tp * plane + e];
    }
    if (t == 0) s_dead = 0;
    unsigned* flags = p.flags + grp * 64;
    const int a_off = ((m0 + i) * HID + wave * 64 + 4 * q) * 4;
    for (int s = p.s0; s < p.s1; ++s) {
        const int slot_prev = d ? p.T - s : s, slot_new = d ? p.T - 1 - s : s + 1, tcur = d ? p.T - 1 - s : s;
        if (s > p.s0) {
            if (wave == 0 && !s_dead) {
                const long t0 = wall_clock64();
                int it = 0;
                for (;;) {
                    const unsigned f = lane < NM ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
                    if (__builtin_amdgcn_ballot_w64(f < (unsigned)s) == 0) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++it & 63) == 0 && wall_clock64() - t0 > QCH_SPIN_TICKS) {
                        if (lane == 0) { s_dead = 1; __hip_atomic_store(p.flags + QCH_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); tatt_raise_sticky(p.sticky, TATT_STICKY_QGRU); }
                        break;
                    }
                }
            }
            __syncthreads();
        }
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (SB) {
            if (s > 0) {                                                        // (h_prev of the first time step is zero: nothing to add)
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.xch[d] + (long)slot_prev * plane, 0, p.Wb * HID * 4, 0x00020000);
                const int off = (m0 + i) * HID * 4 + (wave * 8 + q) * 32;
                qc_bf16x8 ah[2], al[2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    ah[tt] = __builtin_bit_cast(qc_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 128 * tt, 0, 16 /* sc1 */));
                    al[tt] = __builtin_bit_cast(qc_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 128 * tt + 16, 0, 16 /* sc1 */));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g] = qc_mma3(ah[tt], al[tt], bh[g][tt], bl[g][tt], acc[g]);
            }
        } else {
            f32x4 a[4];
            {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.hbuf[d] + (long)slot_prev * plane, 0, p.Wb * HID * 4, 0x00020000);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    a[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, a_off + 64 * u, 0, 16 /* sc1 */));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][v], b[g][u][v], acc[g], 0, 0, 0);
        }
        {
            const int col = lane & 15, rb = (lane >> 4) * 4;
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) red[wave][g][rb + rr][col] = acc[g][rr];
        }
        __syncthreads();
        if (t < 256) {
            float gh0 = bh0, gh1 = bh1, gh2 = bh2;
#pragma unroll
            for (int w = 0; w < 8; ++w) { gh0 += red[w][0][m][j]; gh1 += red[w][1][m][j]; gh2 += red[w][2][m][j]; }
            const float r = sigmoid_fast(gi0 + gh0);
            const float z = sigmoid_fast(gi1 + gh1);
            const float n = tanh_fast(gi2 + r * gh2);
            const float h = (1.f - z) * n + z * hown;
            hown = h;
            if constexpr (SB) {
                qc_split1(h, stage_h[m][j], stage_l[m][j]);
                p.hbuf[d][(long)slot_new * plane + e] = h;                      // fp32 copy: the backward and the W_hh gradient read it
            } else {
                stage[m][j] = h;
            }
            if (p.q) {
                const int Hh = HID / p.C, jj = j0 + j;
                p.q[(((long)tcur * 2 * Hh + d * Hh + jj / p.C) * p.Wb + m0 + m) * p.C + jj % p.C] = h;
            }
            float* gs = p.gsave[d];
            if (gs) {
                gs += (long)tcur * 4 * plane;
                gs[e] = r; gs[plane + e] = z; gs[2 * plane + e] = n; gs[3 * plane + e] = gh2;
            }
        }
        __syncthreads();
        if (t < 64) {                                                           // 16 rows x 4 sixteen-byte pieces
            if constexpr (SB) {
                const int row = t >> 2, grp8 = (t >> 1) & 1, hl = t & 1;
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(hl ? &stage_l[row][grp8 * 8] : &stage_h[row][grp8 * 8]);
                const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(p.xch[d] + (long)slot_new * plane, 0, p.Wb * HID * 4, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(v, ws, (m0 + row) * HID * 4 + (j0 / 8 + grp8) * 32 + hl * 16, 0, 16 /* sc1 */);
            } else {
                const int row = t >> 2, j4 = (t & 3) * 4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(&stage[row][j4]);
                const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(p.hbuf[d] + (long)slot_new * plane, 0, p.Wb * HID * 4, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), ws, ((m0 + row) * HID + j0 + j4) * 4, 0, 16 /* sc1 */);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (t == 0) __hip_atomic_store(flags + mem, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// gi* (Wb, 3*HID) incl. b_ih -- or NULL with x (Wb, IN), wih* (3*HID, IN), bih*: the projection is then computed by the launch itself
// (IN a multiple of 1024); whh* (3*HID, HID); hbuf* (T+1, Wb, HID) with the zero slots already zeroed; gsave* (T, 4, Wb, HID) or NULL;
// q (T, 2*HID/C, Wb, C) or NULL: h in the (sample, H, W, C) layout of the TP interpreter's query embedding; sync as for
// tatt_qgru_bwd_chain.  xch* non-NULL ((T+1, Wb, HID) floats of workspace per direction): the split-bf16 form (s0 must be 0).
// Returns 1 for geometries it does not take.
TATT_API int tatt_qgru_fwd_chain(const float* gi0, const float* gi1, const float* whh0, const float* whh1, const float* bhh0,
                                 const float* bhh1, float* hbuf0, float* hbuf1, float* gsave0, float* gsave1, unsigned* sync,
                                 int T, int Wb, int HID, int s0, int s1, const float* x, const float* wih0, const float* wih1,
                                 const float* bih0, const float* bih1, int IN, float* q, int C, float* xch0, float* xch1,
                                 hipStream_t st) {
    if (HID != 512 || Wb % 16 || Wb <= 0 || (Wb / 16) * 2 * (HID / 16) > 256 || (Wb / 16) * 2 > 15) return 1;
    if (s0 < 0 || s1 > T || s0 >= s1) return s0 == s1 ? 0 : 2;
    if (!gi0 && (!x || !wih0 || !wih1 || !bih0 || !bih1 || IN % 1024)) return 1;
    if (q && (C <= 0 || HID % C)) return 1;
    if (s0 == 0 && hipMemsetAsync(sync, 0, 1024 * sizeof(unsigned), st) != hipSuccess) return 3;
    if ((xch0 == nullptr) != (xch1 == nullptr) || (xch0 && s0 != 0)) return 1;      // (the split form runs the whole chain in one launch)
    QFChainP p = {{gi0, gi1}, {whh0, whh1}, {bhh0, bhh1}, {hbuf0, hbuf1}, {gsave0, gsave1}, sync, T, Wb, s0, s1,
                  x, {wih0, wih1}, {bih0, bih1}, IN, q, C, {xch0, xch1}, tatt_sticky_ptr()};
    if (xch0) hipLaunchKernelGGL(qgru_fwd_chain_kernel<true>, dim3((Wb / 16) * 2 * (HID / 16)), dim3(512), 0, st, p);
    else hipLaunchKernelGGL(qgru_fwd_chain_kernel<false>, dim3((Wb / 16) * 2 * (HID / 16)), dim3(512), 0, st, p);
    return LAUNCH_CHECK();
}

// Work-groups of each persistent launch that can be resident on the CURRENT device at once (occupancy per CU x CUs the process sees):
// out[0] forward split-bf16, out[1] forward fp32, out[2] backward split-bf16, out[3] backward fp32.  The chains need their whole grid
// ((Wb / 16) * 2 * (HID / 16) work-groups) co-resident: on a partitioned / CU-masked device the caller takes the per-step launches.
TATT_API int tatt_qgru_chain_capacity(int* out) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
    const void* k[4] = {reinterpret_cast<const void*>(qgru_fwd_chain_kernel<true>), reinterpret_cast<const void*>(qgru_fwd_chain_kernel<false>),
                        reinterpret_cast<const void*>(qgru_bwd_chain_kernel<true>), reinterpret_cast<const void*>(qgru_bwd_chain_kernel<false>)};
    for (int i = 0; i < 4; ++i) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k[i], 512, 0) != hipSuccess) return 2;
        out[i] = n * prop.multiProcessorCount;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// GruBlock glue (reference GruBlock = 1x1 conv + BiGRU, model/tsrn.py:1067-1084): the 1x1 conv W_c (64 x K), b_c and the
// GRU input projections W_ih (2 x 96 x 64), b_ih are composed into ONE projection  W' = W_ih W_c (192 x K),
// b' = W_ih b_c + b_ih  applied to the token matrix by a single GEMM; `tail` maps the gradients of the composed
// projection back to the reference's parameters.  A few hundred kFLOP each: one small launch instead of 4 / 6 GEMM launches.
// ------------------------------------------------------------------------------------------------
__global__ void gru_compose_kernel(const float* __restrict__ wih_f, const float* __restrict__ wih_r,
                                   const float* __restrict__ bih_f, const float* __restrict__ bih_r,
                                   const float* __restrict__ Wc, const float* __restrict__ bc, float* __restrict__ Wp,
                                   float* __restrict__ bp, int K) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 192 * (K + 1)) return;
    const int row = idx / (K + 1), col = idx - row * (K + 1);
    const float* w = (row < 96 ? wih_f : wih_r) + (row % 96) * 64;
    float s0 = 0.f, s1 = 0.f;
    if (col < K) {
        for (int c = 0; c < 64; c += 2) { s0 = fmaf(w[c], Wc[c * K + col], s0); s1 = fmaf(w[c + 1], Wc[(c + 1) * K + col], s1); }
        Wp[row * K + col] = s0 + s1;
    } else {
        for (int c = 0; c < 64; c += 2) { s0 = fmaf(w[c], bc[c], s0); s1 = fmaf(w[c + 1], bc[c + 1], s1); }
        bp[row] = s0 + s1 + (row < 96 ? bih_f : bih_r)[row % 96];
    }
}
TATT_API int tatt_gru_compose(const float* wih_f, const float* wih_r, const float* bih_f, const float* bih_r,
                              const float* Wc, const float* bc, float* Wp, float* bp, int K, hipStream_t st) {
    hipLaunchKernelGGL(gru_compose_kernel, dim3(cdiv(192 * (K + 1), 256)), dim3(256), 0, st, wih_f, wih_r, bih_f, bih_r, Wc,
                       bc, Wp, bp, K);
    return LAUNCH_CHECK();
}
// The same for up to GC_MAX GruBlocks in ONE launch (all ten of a generator, once per forward: the compositions depend on parameters only)
#define GC_MAX 16
struct GCEntry { const float* wih_f; const float* wih_r; const float* bih_f; const float* bih_r; const float* Wc; const float* bc;
                 float* Wp; float* bp; int K; int block0; };
struct GCTable { GCEntry e[GC_MAX]; int n; };
__global__ void gru_compose_batch_kernel(GCTable t) {
    int k = 0;
    while (k + 1 < t.n && (int)blockIdx.x >= t.e[k + 1].block0) ++k;
    const GCEntry& e = t.e[k];
    const int K = e.K;
    const int idx = ((int)blockIdx.x - e.block0) * blockDim.x + threadIdx.x;
    if (idx >= 192 * (K + 1)) return;
    const int row = idx / (K + 1), col = idx - row * (K + 1);
    const float* w = (row < 96 ? e.wih_f : e.wih_r) + (row % 96) * 64;
    float s0 = 0.f, s1 = 0.f;
    if (col < K) {
        for (int c = 0; c < 64; c += 2) { s0 = fmaf(w[c], e.Wc[c * K + col], s0); s1 = fmaf(w[c + 1], e.Wc[(c + 1) * K + col], s1); }
        e.Wp[row * K + col] = s0 + s1;
    } else {
        for (int c = 0; c < 64; c += 2) { s0 = fmaf(w[c], e.bc[c], s0); s1 = fmaf(w[c + 1], e.bc[c + 1], s1); }
        e.bp[row] = s0 + s1 + (row < 96 ? e.bih_f : e.bih_r)[row % 96];
    }
}
// ptrs: HOST array of n x 8 device pointers (wih_f, wih_r, bih_f, bih_r, Wc, bc, Wp, bp per block); Ks: HOST array of n ints
TATT_API int tatt_gru_compose_batch(const float* const* ptrs, const int* Ks, int n, hipStream_t st) {
    for (int base = 0; base < n; base += GC_MAX) {
        GCTable t;
        t.n = n - base < GC_MAX ? n - base : GC_MAX;
        int blocks = 0;
        for (int k = 0; k < t.n; ++k) {
            const float* const* q = ptrs + (long)(base + k) * 8;
            t.e[k] = {q[0], q[1], q[2], q[3], q[4], q[5], const_cast<float*>(q[6]), const_cast<float*>(q[7]), Ks[base + k], blocks};
            blocks += cdiv(192 * (Ks[base + k] + 1), 256);
        }
        hipLaunchKernelGGL(gru_compose_batch_kernel, dim3(blocks), dim3(256), 0, st, t);
    }
    return LAUNCH_CHECK();
}
// dW_ih_d (96x64) = dW'_d W_c^T + db'_d b_c^T ;  dW_c (64xK) = sum_d W_ih_d^T dW'_d ;  db_c (64) = sum_d W_ih_d^T db'_d ;
// dW_hh_d (96x32) = the d-th diagonal block of dWhh (192x64) = dgh^T hprev
__global__ void gru_tail_kernel(const float* __restrict__ dWp, const float* __restrict__ dbp,
                                const float* __restrict__ Wc, const float* __restrict__ bc,
                                const float* __restrict__ wih_f, const float* __restrict__ wih_r,
                                float* __restrict__ dwih_f, float* __restrict__ dwih_r, float* __restrict__ dWc,
                                float* __restrict__ dbc, int K, const float* __restrict__ dWhh,
                                float* __restrict__ dwhh_f, float* __restrict__ dwhh_r, int whh_ld) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 2 * 96 * 64) {   // 48 whole blocks; a wave = one row (d, r), lane = c.  W_c goes through LDS transposed, 64 k at a time:
        // read straight from memory each lane would walk its own row of W_c (64 cache lines per load instruction)
        __shared__ float WcT[64 * 65];
        const int d = idx / 6144, r = (idx % 6144) / 64, c = idx & 63;
        const float* a = dWp + (long)(96 * d + r) * K;
        float s0 = dbp[96 * d + r] * bc[c], s1 = 0.f;
        for (int kc = 0; kc < K; kc += 64) {
            const int kn = K - kc < 64 ? K - kc : 64;
            __syncthreads();
            for (int e = threadIdx.x; e < 4096; e += 256) {
                const int cc = e >> 6, k = e & 63;
                if (k < kn) WcT[k * 65 + cc] = Wc[(long)cc * K + kc + k];
            }
            __syncthreads();
            for (int k = 0; k < kn; k += 2) {
                s0 = fmaf(a[kc + k], WcT[k * 65 + c], s0);
                s1 = fmaf(a[kc + k + 1], WcT[(k + 1) * 65 + c], s1);
            }
        }
        (d ? dwih_r : dwih_f)[r * 64 + c] = s0 + s1;
        return;
    }
    idx -= 2 * 96 * 64;
    if (idx < 64 * K) {
        const int c = idx / K, k = idx - c * K;
        float s0 = 0.f, s1 = 0.f;
        for (int row = 0; row < 96; ++row) {
            s0 = fmaf(wih_f[row * 64 + c], dWp[(long)row * K + k], s0);
            s1 = fmaf(wih_r[row * 64 + c], dWp[(long)(96 + row) * K + k], s1);
        }
        dWc[idx] = s0 + s1;
        return;
    }
    idx -= 64 * K;
    if (idx < 64) {
        float s0 = 0.f, s1 = 0.f;
        for (int row = 0; row < 96; ++row) {
            s0 = fmaf(wih_f[row * 64 + idx], dbp[row], s0);
            s1 = fmaf(wih_r[row * 64 + idx], dbp[96 + row], s1);
        }
        dbc[idx] = s0 + s1;
        return;
    }
    idx -= 64;
    if (idx < 2 * 96 * 32) {
        const int d = idx / 3072, r = (idx % 3072) / 32, c = idx & 31;
        // whh_ld 64: dWhh (192 x 64) = dgh^T hprev over both directions, the diagonal blocks are wanted; 32: compact [fwd; rev]
        (d ? dwhh_r : dwhh_f)[r * 32 + c] = dWhh[(long)(96 * d + r) * whh_ld + (whh_ld == 64 ? 32 * d : 0) + c];
    }
}
TATT_API int tatt_gru_tail(const float* dWp, const float* dbp, const float* Wc, const float* bc, const float* wih_f,
                           const float* wih_r, float* dwih_f, float* dwih_r, float* dWc, float* dbc, int K,
                           const float* dWhh, float* dwhh_f, float* dwhh_r, hipStream_t st) {
    if (K % 2) return 1;
    const int total = 2 * 96 * 64 + 64 * K + 64 + 2 * 96 * 32;
    hipLaunchKernelGGL(gru_tail_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, dWp, dbp, Wc, bc, wih_f, wih_r, dwih_f,
                       dwih_r, dWc, dbc, K, dWhh, dwhh_f, dwhh_r, 64);
    return LAUNCH_CHECK();
}
// the same with dWhh compact: (192, 32) = [dW_hh forward; dW_hh reverse], as tatt_gru_wgrad_frag's reduction leaves it
TATT_API int tatt_gru_tail_c(const float* dWp, const float* dbp, const float* Wc, const float* bc, const float* wih_f,
                             const float* wih_r, float* dwih_f, float* dwih_r, float* dWc, float* dbc, int K,
                             const float* dWhh_c, float* dwhh_f, float* dwhh_r, hipStream_t st) {
    if (K % 2) return 1;
    const int total = 2 * 96 * 64 + 64 * K + 64 + 2 * 96 * 32;
    hipLaunchKernelGGL(gru_tail_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, dWp, dbp, Wc, bc, wih_f, wih_r, dwih_f,
                       dwih_r, dWc, dbc, K, dWhh_c, dwhh_f, dwhh_r, 32);
    return LAUNCH_CHECK();
}
