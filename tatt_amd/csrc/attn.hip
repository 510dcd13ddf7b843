// Scaled-dot-product attention core over a SHORT key sequence (S <= 32: the 26-token CRNN text prior) for
// the TP interpreter (reference nn.MultiheadAttention inside TransformerDecoderLayer_TP /
// TransformerEncoderLayer, model/transformer_v2.py:453,786,821-824).  The Q/K/V and output projections are
// GEMMs (tatt_gemm; q pre-scaled by 1/sqrt(d_h) in its epilogue); this kernel does
//     P = softmax(Q K^T) -> dropout -> ctx = P V,   w_avg = mean_heads(P_dropped)
// with K/V of one image resident in LDS, one thread per (query, head), d_h = 16, 4 heads (E = 64).
// Threads of a wave are (16 queries x 4 heads) so Q/ctx rows are read/written as contiguous 256 B.
#include "common.h"

#define AT_E 64
#define AT_H 4
#define AT_D 16
#define AT_SMAX 32
#define AT_QPB 64    // queries per block

struct AttnP {
    const float* Q; const float* K; const float* V;   // (B,Lq,64), (B,S,64), (B,S,64)
    float* ctx; float* wavg;                          // (B,Lq,64), (B,Lq,S) (wavg may be null)
    int B, Lq, S;
    float pdrop; const unsigned long long* seed; unsigned site;
};

__device__ __forceinline__ void attn_load_kv(const float* K, const float* V, int b, int S, float (*Ks)[AT_E],
                                             float (*Vs)[AT_E]) {
    for (int i = threadIdx.x; i < S * AT_E; i += blockDim.x) {
        Ks[i / AT_E][i % AT_E] = K[(long)b * S * AT_E + i];
        Vs[i / AT_E][i % AT_E] = V[(long)b * S * AT_E + i];
    }
}

// probabilities of one (query, head) row; returns them in p[] (un-dropped) ; s >= S entries are 0
__device__ __forceinline__ void attn_row_probs(const float* q, const float (*Ks)[AT_E], int head, int S, float* p) {
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < AT_SMAX; ++s) {
        float a = -INFINITY;
        if (s < S) {
            a = 0.f;
#pragma unroll
            for (int i = 0; i < AT_D; ++i) a = fmaf(q[i], Ks[s][head * AT_D + i], a);
        }
        p[s] = a;
        mx = fmaxf(mx, a);
    }
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < AT_SMAX; ++s) {
        float e = s < S ? __expf(p[s] - mx) : 0.f;
        p[s] = e; sum += e;
    }
    const float inv = 1.f / sum;
#pragma unroll
    for (int s = 0; s < AT_SMAX; ++s) p[s] *= inv;
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnP a) {
    __shared__ __attribute__((aligned(16))) float Ks[AT_SMAX][AT_E];
    __shared__ __attribute__((aligned(16))) float Vs[AT_SMAX][AT_E];
    const int b = blockIdx.y;
    attn_load_kv(a.K, a.V, b, a.S, Ks, Vs);
    __syncthreads();
    const int head = threadIdx.x & 3;
    const int qi = blockIdx.x * AT_QPB + (threadIdx.x >> 2);
    const bool valid = qi < a.Lq;
    const int qc = valid ? qi : a.Lq - 1;
    const long rowoff = ((long)b * a.Lq + qc) * AT_E + head * AT_D;
    float q[AT_D];
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(a.Q + rowoff + 4 * i4);
        q[4 * i4] = v[0]; q[4 * i4 + 1] = v[1]; q[4 * i4 + 2] = v[2]; q[4 * i4 + 3] = v[3];
    }
    float p[AT_SMAX];
    attn_row_probs(q, Ks, head, a.S, p);
    if (a.pdrop > 0.f) {
        const uint32_t th = dropout_thresh(a.pdrop);
        const float sc = 1.f / (1.f - a.pdrop);
        const uint64_t base = (((uint64_t)b * AT_H + head) * a.Lq + qc) * a.S;
        const uint64_t sd = a.seed[0];
#pragma unroll
        for (int s = 0; s < AT_SMAX; ++s)
            if (s < a.S) p[s] = dropout_keep(sd, a.site, base + s, th) ? p[s] * sc : 0.f;
    }
    float o[AT_D];
#pragma unroll
    for (int i = 0; i < AT_D; ++i) o[i] = 0.f;
#pragma unroll
    for (int s = 0; s < AT_SMAX; ++s)
        if (s < a.S) {
#pragma unroll
            for (int i = 0; i < AT_D; ++i) o[i] = fmaf(p[s], Vs[s][head * AT_D + i], o[i]);
        }
    if (valid) {
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4)
            *reinterpret_cast<f32x4*>(a.ctx + rowoff + 4 * i4) = (f32x4){o[4 * i4], o[4 * i4 + 1], o[4 * i4 + 2], o[4 * i4 + 3]};
    }
    if (a.wavg) {
        // head average: the 4 heads of a query are adjacent lanes
#pragma unroll
        for (int s = 0; s < AT_SMAX; ++s) {
            float v = p[s];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            if (valid && s < a.S && (s & 3) == head) a.wavg[((long)b * a.Lq + qi) * a.S + s] = 0.25f * v;
        }
    }
}
TATT_API int tatt_attn_fwd(const float* Q, const float* K, const float* V, float* ctx, float* wavg, int B, int Lq,
                           int S, float pdrop, const unsigned long long* seed, unsigned site, hipStream_t st) {
    if (S > AT_SMAX || S < 1) return 1;
    AttnP a = {Q, K, V, ctx, wavg, B, Lq, S, pdrop, seed, site};
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(cdiv(Lq, AT_QPB), B), dim3(256), 0, st, a);
    return LAUNCH_CHECK();
}

// backward: recomputes P; dQ per (query, head); dK/dV reduced over the block's 64 queries on the matrix cores
// (v_mfma_f32_16x16x4_f32) and written as per-block partials part[b][blk][2][S][64], summed deterministically by attn_bwd_reduce.
struct AttnBwdP {
    const float* Q; const float* K; const float* V; const float* dctx; const float* dwavg;   // dwavg may be null
    float* dQ; float* part;
    int B, Lq, S, nblk;
    float pdrop; const unsigned long long* seed; unsigned site;
};
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnBwdP a) {
    __shared__ __attribute__((aligned(16))) float Ks[AT_SMAX][AT_E];
    __shared__ __attribute__((aligned(16))) float Vs[AT_SMAX][AT_E];
    // per head: [query][key] matrix (dropped probabilities, then score gradients) = the A operand of the dV / dK MFMAs
    __shared__ float PS[AT_H][AT_QPB][AT_SMAX + 1];
    const int b = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    attn_load_kv(a.K, a.V, b, a.S, Ks, Vs);
    __syncthreads();
    const int head = threadIdx.x & 3, ql = threadIdx.x >> 2;
    const int q0 = blockIdx.x * AT_QPB;
    const int qi = q0 + ql;
    const bool valid = qi < a.Lq;
    const int qc = valid ? qi : a.Lq - 1;
    const long rowoff = ((long)b * a.Lq + qc) * AT_E + head * AT_D;
    float q[AT_D], g[AT_D];
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(a.Q + rowoff + 4 * i4);
        f32x4 w = *reinterpret_cast<const f32x4*>(a.dctx + rowoff + 4 * i4);
#pragma unroll
        for (int u = 0; u < 4; ++u) { q[4 * i4 + u] = v[u]; g[4 * i4 + u] = valid ? w[u] : 0.f; }
    }
    float p[AT_SMAX];
    attn_row_probs(q, Ks, head, a.S, p);
    const bool drop = a.pdrop > 0.f;
    const uint32_t th = dropout_thresh(a.pdrop);
    const float sc = drop ? 1.f / (1.f - a.pdrop) : 1.f;
    const uint64_t base = (((uint64_t)b * AT_H + head) * a.Lq + qc) * a.S;
    const uint64_t sd = drop ? a.seed[0] : 0ull;
    float dp[AT_SMAX];
    float dot = 0.f;
#pragma unroll
    for (int s = 0; s < AT_SMAX; ++s) {
        float d = 0.f, pd = 0.f;
        if (s < a.S) {
#pragma unroll
            for (int i = 0; i < AT_D; ++i) d = fmaf(g[i], Vs[s][head * AT_D + i], d);
            if (a.dwavg && valid) d += 0.25f * a.dwavg[((long)b * a.Lq + qi) * a.S + s];
            const bool keep = !drop || dropout_keep(sd, a.site, base + s, th);
            pd = keep ? p[s] * sc : 0.f;                 // dropped probability (what multiplied V)
            d = keep ? d * sc : 0.f;                     // gradient w.r.t. the un-dropped probability
        }
        PS[head][ql][s] = pd;
        dp[s] = d;
        dot = fmaf(p[s], d, dot);
    }
    float dq[AT_D];
#pragma unroll
    for (int i = 0; i < AT_D; ++i) dq[i] = 0.f;
#pragma unroll
    for (int s = 0; s < AT_SMAX; ++s) {
        const float dsc = (valid && s < a.S) ? p[s] * (dp[s] - dot) : 0.f;
        dp[s] = dsc;                                     // now holds the score gradient
        if (s < a.S) {
#pragma unroll
            for (int i = 0; i < AT_D; ++i) dq[i] = fmaf(dsc, Ks[s][head * AT_D + i], dq[i]);
        }
    }
    if (valid) {
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4)
            *reinterpret_cast<f32x4*>(a.dQ + rowoff + 4 * i4) = (f32x4){dq[4 * i4], dq[4 * i4 + 1], dq[4 * i4 + 2], dq[4 * i4 + 3]};
    }
    __syncthreads();
    // ---- dV[s][i] = sum_q Pd[q][s] g[q][i],  dK[s][i] = sum_q dS[q][s] Q[q][i]: per head (= wave) a (32 x 64) x (64 x 16)
    //      product on v_mfma_f32_16x16x4_f32.  A[i = key (lane&15)][k = query (lane>>4)] from PS, B[k = query][j = lane&15]
    //      straight from global (dctx / Q rows were just read: L2 hits).  C: col = lane&15, row = (lane>>4)*4 + reg.
    float* P = a.part + ((long)b * a.nblk + blockIdx.x) * 2 * a.S * AT_E;
    const int w = wave;                                   // head handled by this wave
    const int col = lane & 15, kq = lane >> 4;
    for (int pass = 0; pass < 2; ++pass) {
        const float* Bsrc = pass == 0 ? a.dctx : a.Q;
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll 4
        for (int k0 = 0; k0 < AT_QPB; k0 += 4) {
            const int qq = q0 + k0 + kq;
            const float bv = qq < a.Lq ? Bsrc[((long)b * a.Lq + qq) * AT_E + w * AT_D + col] : 0.f;
            const float a0 = PS[w][k0 + kq][col], a1 = PS[w][k0 + kq][16 + col];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc1, 0, 0, 0);
        }
        // pass 0 -> dV (second half of the partial record), pass 1 -> dK (first half)
        float* dst = P + (pass == 0 ? a.S * AT_E : 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s0 = kq * 4 + r, s1 = 16 + kq * 4 + r;
            if (s0 < a.S) dst[s0 * AT_E + w * AT_D + col] = acc0[r];
            if (s1 < a.S) dst[s1 * AT_E + w * AT_D + col] = acc1[r];
        }
        if (pass == 0) {
            __syncthreads();                               // every wave has consumed the probabilities
#pragma unroll
            for (int s = 0; s < AT_SMAX; ++s) PS[head][ql][s] = dp[s];
            __syncthreads();
        }
    }
}
__global__ void attn_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dK, float* __restrict__ dV,
                                       int B, int S, int nblk) {
    const int n = S * AT_E;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * n) return;
    int b = idx / n, i = idx % n;
    float sk = 0.f, sv = 0.f;
    for (int k = 0; k < nblk; ++k) {
        const float* P = part + ((long)b * nblk + k) * 2 * n;
        sk += P[i]; sv += P[n + i];
    }
    dK[idx] = sk; dV[idx] = sv;
}
// part: B * cdiv(Lq,64) * 2 * S * 64 floats
TATT_API int tatt_attn_bwd(const float* Q, const float* K, const float* V, const float* dctx, const float* dwavg,
                           float* dQ, float* dK, float* dV, float* part, int B, int Lq, int S, float pdrop,
                           const unsigned long long* seed, unsigned site, hipStream_t st) {
    if (S > AT_SMAX || S < 1) return 1;
    int nblk = cdiv(Lq, AT_QPB);
    AttnBwdP a = {Q, K, V, dctx, dwavg, dQ, part, B, Lq, S, nblk, pdrop, seed, site};
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(nblk, B), dim3(256), 0, st, a);
    hipLaunchKernelGGL(attn_bwd_reduce_kernel, dim3(cdiv((long)B * S * AT_E, 256)), dim3(256), 0, st, part, dK, dV, B, S,
                       nblk);
    return LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// Row softmax (+ dropout) over materialised score matrices -- the self-attention of the TBSRN variant's FeatureEnhancer
// (reference model/tbsrn.py:130-151: softmax(QK^T/sqrt(d_k)) over ALL H*W positions, dropout(0.1), @V).  With 288 GB of HBM
// the (B,h,P,P) probability tensor is simply kept for the backward pass; the matrix products around it are batched GEMMs
// (tatt_gemm).  One work-group per row, row length L <= 4096 held in registers.
// ------------------------------------------------------------------------------------------------
#define SM_MAXPER 16
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(float* __restrict__ S, float* __restrict__ Pd, int L,
                                                               float pdrop, const unsigned long long* __restrict__ seed,
                                                               unsigned site) {
    __shared__ float sh[4];
    const long row = blockIdx.x;
    float* s = S + row * L;
    const int t = threadIdx.x;
    float v[SM_MAXPER];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < SM_MAXPER; ++k) {
        const int c = t + 256 * k;
        v[k] = c < L ? s[c] : -INFINITY;
        mx = fmaxf(mx, v[k]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((t & 63) == 0) sh[t >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < SM_MAXPER; ++k) {
        const int c = t + 256 * k;
        v[k] = c < L ? __expf(v[k] - mx) : 0.f;
        sum += v[k];
    }
    sum = wave_sum(sum);
    if ((t & 63) == 0) sh[t >> 6] = sum;
    __syncthreads();
    const float inv = 1.f / ((sh[0] + sh[1]) + (sh[2] + sh[3]));
    const bool drop = Pd != nullptr && pdrop > 0.f;
    const uint32_t th = dropout_thresh(pdrop);
    const float sc = 1.f / (1.f - pdrop);
    const uint64_t sd = drop ? seed[0] : 0ull;
#pragma unroll
    for (int k = 0; k < SM_MAXPER; ++k) {
        const int c = t + 256 * k;
        if (c < L) {
            const float p = v[k] * inv;
            s[c] = p;
            if (Pd) Pd[row * L + c] = (!drop || dropout_keep(sd, site, (uint64_t)row * L + c, th)) ? (drop ? p * sc : p) : 0.f;
        }
    }
}
// S (rows x L) is overwritten by P = softmax(S); Pd (nullable) receives dropout(P)
TATT_API int tatt_softmax_rows_fwd(float* S, float* Pd, long rows, int L, float pdrop, const unsigned long long* seed,
                                   unsigned site, hipStream_t st) {
    if (L > 256 * SM_MAXPER) return 1;
    hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, st, S, Pd, L, pdrop, seed, site);
    return LAUNCH_CHECK();
}
// dS = P * (dPm - sum_j P_j dPm_j), dPm = dropout'(dPd); written over dPd
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, int L,
                                                               float pdrop, const unsigned long long* __restrict__ seed,
                                                               unsigned site) {
    __shared__ float sh[4];
    const long row = blockIdx.x;
    const int t = threadIdx.x;
    const bool drop = pdrop > 0.f;
    const uint32_t th = dropout_thresh(pdrop);
    const float sc = drop ? 1.f / (1.f - pdrop) : 1.f;
    const uint64_t sd = drop ? seed[0] : 0ull;
    float p[SM_MAXPER], g[SM_MAXPER];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < SM_MAXPER; ++k) {
        const int c = t + 256 * k;
        p[k] = 0.f; g[k] = 0.f;
        if (c < L) {
            p[k] = P[row * L + c];
            float d = dP[row * L + c];
            if (drop) d = dropout_keep(sd, site, (uint64_t)row * L + c, th) ? d * sc : 0.f;
            g[k] = d;
            dot = fmaf(p[k], d, dot);
        }
    }
    dot = wave_sum(dot);
    if ((t & 63) == 0) sh[t >> 6] = dot;
    __syncthreads();
    dot = (sh[0] + sh[1]) + (sh[2] + sh[3]);
#pragma unroll
    for (int k = 0; k < SM_MAXPER; ++k) {
        const int c = t + 256 * k;
        if (c < L) dP[row * L + c] = p[k] * (g[k] - dot);
    }
}
TATT_API int tatt_softmax_rows_bwd(const float* P, float* dP, long rows, int L, float pdrop,
                                   const unsigned long long* seed, unsigned site, hipStream_t st) {
    if (L > 256 * SM_MAXPER) return 1;
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, st, P, dP, L, pdrop, seed, site);
    return LAUNCH_CHECK();
}
