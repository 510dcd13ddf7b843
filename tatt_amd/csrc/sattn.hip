// Score-free multi-head self-attention over ALL positions of a feature map -- the FeatureEnhancer of the TBSRN variant
// (reference model/tbsrn.py:96-151: `attention`: softmax(Q K^T / sqrt(d_k)) -> Dropout(0.1) -> @ V, h = 4 heads of d_k = 32,
// P = H*W = 1024 .. 4096 positions).  Round 2 materialised the (B, h, P, P) probabilities in HBM (0.8 GB per block at B = 48,
// P = 1024, written and read four times per step); here the scores exist only as 16 x 16 MFMA accumulator blocks:
//
//   forward   one work-group = 64 queries of one (sample, head), 4 waves x 16 queries; K / V stream through LDS in tiles of 64 keys;
//             online softmax (running max / sum per query), dropout by the counter hash, O accumulated in registers; per query the
//             log-sum-exp is saved (B h P floats) -- nothing else.
//   backward  two kernels, both recomputing P = exp(S - lse) tile by tile (deterministic, no atomics):
//             dK / dV: one work-group = 64 keys, loops over the query tiles;   dQ: one work-group = 64 queries, loops over key tiles.
//
// All products run on v_mfma_f32_16x16x4_f32 (exact fp32).  The transposed forms are chosen so that a probability / score-gradient
// block leaves one MFMA in exactly the register layout the next MFMA wants as an operand (no LDS transpose):
//   forward, dQ:  S^T = K Q^T  -> lane (q = lane & 15, kq = lane >> 4) holds S[q][16 jb + 4 kq + r], r = 0..3 = the B operand
//                 (k-slot kq, step r) of  O^T += V^T P^T  /  dQ^T += K^T dS^T;  softmax statistics are per-lane scalars.
//   dK / dV:      S = Q K^T    -> lane (key = lane & 15, kq) holds S[16 qb + 4 kq + r][key] = the A operand of
//                 dV += P^T dO,  dK += dS^T Q.
// Dropout masks are those of tatt_softmax_rows_fwd (same seed word, site, flat index (row * P + key)): the fused path reproduces
// the materialised path mask for mask.
#include "common.h"

#define SA_D 32
#define SA_T 64                  // queries per work-group = keys per tile
#define SA_RP 36                 // LDS pitch of row-major tiles [64 rows][32]
#define SA_TP 68                 // LDS pitch of transposed tiles [32][64 rows]
#define SA_ROWSZ (SA_T * SA_RP)  // 2304 floats
#define SA_TRSZ (SA_D * SA_TP)   // 2176 floats

struct SAttnP {
    const float* Q; const float* K; const float* V;            // (B, P, E), E = h * 32
    float* O; float* lse;                                      // (B, P, E), (B, h, P)
    const float* dO; const float* Dv;                          // backward: upstream gradient, D = rowsum(dO * O) (B, h, P)
    float* dQ; float* dK; float* dV;
    int B, P, h, E;
    float scale, pdrop; const unsigned long long* seed; unsigned site;
};

__device__ __forceinline__ f32x4 sa_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// 64 rows x 32 columns of one head from a (B, P, E) tensor -> registers (two 16-byte loads per thread)
__device__ __forceinline__ void sa_fetch(const float* __restrict__ X, long base_row, int E, int hoff, int tid, f32x4& a, f32x4& b) {
    const int r0 = tid >> 3, c4 = (tid & 7) * 4;
    a = sa_ld4(X + (base_row + r0) * E + hoff + c4);
    b = sa_ld4(X + (base_row + r0 + 32) * E + hoff + c4);
}
__device__ __forceinline__ void sa_store_rows(float* T, int tid, f32x4 a, f32x4 b) {
    const int r0 = tid >> 3, c4 = (tid & 7) * 4;
    *reinterpret_cast<f32x4*>(T + r0 * SA_RP + c4) = a;
    *reinterpret_cast<f32x4*>(T + (r0 + 32) * SA_RP + c4) = b;
}
__device__ __forceinline__ void sa_store_tr(float* T, int tid, f32x4 a, f32x4 b) {
    const int r0 = tid >> 3, c4 = (tid & 7) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) { T[(c4 + e) * SA_TP + r0] = a[e]; T[(c4 + e) * SA_TP + r0 + 32] = b[e]; }
}
// 16 x 16 block: acc += A(rows of a row-major LDS tile: 8 consecutive columns per lane) * B(8 registers per lane)
__device__ __forceinline__ f32x4 sa_mm8(const float* __restrict__ rowp, const float (&b)[8]) {
    const f32x4 a0 = sa_ld4(rowp), a1 = sa_ld4(rowp + 4);
    f32x4 c0 = (f32x4){0.f, 0.f, 0.f, 0.f}, c1 = c0;
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0], b[0], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1], b[1], c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[2], b[2], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[3], b[3], c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[0], b[4], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[1], b[5], c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[2], b[6], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[3], b[7], c1, 0, 0, 0);
    return c0 + c1;
}
__device__ __forceinline__ float sa_rowred_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float sa_rowred_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void sattn_fwd_kernel(SAttnP p) {
    __shared__ __attribute__((aligned(16))) float Ks[2][SA_ROWSZ];       // K tile, row-major  [key][d]
    __shared__ __attribute__((aligned(16))) float Vt[2][SA_TRSZ];        // V tile, transposed [dv][key]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, am = lane & 15, kq = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.h, head = bh - b * p.h, hoff = head * SA_D;
    const int q = blockIdx.x * SA_T + wave * 16 + am;                    // this lane's query (column of S^T)
    const long brow = (long)b * p.P;
    float qb[8];
    {
        const float* qp = p.Q + (brow + q) * p.E + hoff + 8 * kq;
        const f32x4 v0 = sa_ld4(qp), v1 = sa_ld4(qp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { qb[e] = v0[e] * p.scale; qb[4 + e] = v1[e] * p.scale; }
    }
    const bool drop = p.pdrop > 0.f;
    const uint64_t sd = drop ? p.seed[0] : 0ull;
    const uint32_t th = dropout_thresh(p.pdrop);
    const float sc = drop ? 1.f / (1.f - p.pdrop) : 1.f;
    const uint64_t rowidx = ((uint64_t)bh * p.P + q) * (uint64_t)p.P;    // flat index of (b, head, q, key 0)
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    float m = -INFINITY, l = 0.f;
    const int nt = p.P / SA_T;
    f32x4 ka, kb, va, vb;
    sa_fetch(p.K, brow, p.E, hoff, tid, ka, kb);
    sa_fetch(p.V, brow, p.E, hoff, tid, va, vb);
    sa_store_rows(Ks[0], tid, ka, kb);
    sa_store_tr(Vt[0], tid, va, vb);
    __syncthreads();
    for (int kt = 0; kt < nt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nt) {
            sa_fetch(p.K, brow + (long)(kt + 1) * SA_T, p.E, hoff, tid, ka, kb);
            sa_fetch(p.V, brow + (long)(kt + 1) * SA_T, p.E, hoff, tid, va, vb);
        }
        // S^T blocks: lane holds S[q][16 jb + 4 kq + r]
        f32x4 st[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) st[jb] = sa_mm8(Ks[cur] + (16 * jb + am) * SA_RP + 8 * kq, qb);
        float mt = -INFINITY;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) mt = fmaxf(mt, st[jb][r]);
        mt = sa_rowred_max(mt);
        const float mn = fmaxf(m, mt);
        const float alpha = __expf(m - mn);
        float ls = 0.f;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(st[jb][r] - mn);
                ls += e;
                st[jb][r] = e;
            }
        l = l * alpha + sa_rowred_sum(ls);
        m = mn;
        if (drop) {
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint64_t idx = rowidx + (uint64_t)(kt * SA_T + 16 * jb + 4 * kq + r);
                    st[jb][r] = dropout_keep(sd, p.site, idx, th) ? st[jb][r] * sc : 0.f;
                }
        }
        // O^T[dv][q] += V^T[dv][key] P^T[key][q]
#pragma unroll
        for (int dvb = 0; dvb < 2; ++dvb) {
            acc[dvb] *= alpha;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const f32x4 v4 = sa_ld4(Vt[cur] + (16 * dvb + am) * SA_TP + 16 * jb + 4 * kq);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[dvb] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4[r], st[jb][r], acc[dvb], 0, 0, 0);
            }
        }
        if (kt + 1 < nt) {
            sa_store_rows(Ks[cur ^ 1], tid, ka, kb);
            sa_store_tr(Vt[cur ^ 1], tid, va, vb);
        }
        __syncthreads();
    }
    const float inv = 1.f / l;
    float* op = p.O + (brow + q) * p.E + hoff + 4 * kq;
    *reinterpret_cast<f32x4*>(op) = acc[0] * inv;
    *reinterpret_cast<f32x4*>(op + 16) = acc[1] * inv;
    if (kq == 0) p.lse[(long)bh * p.P + q] = m + logf(l);
}

// D[b, head, q] = sum_dv dO[q][dv] O[q][dv]
__global__ __launch_bounds__(256) void sattn_prep_kernel(SAttnP p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;               // (b, q, head)
    if (idx >= (long)p.B * p.P * p.h) return;
    const int head = (int)(idx % p.h);
    const long bq = idx / p.h;
    const int b = (int)(bq / p.P), q = (int)(bq - (long)b * p.P);
    const float* o = p.O + bq * p.E + head * SA_D;
    const float* g = p.dO + bq * p.E + head * SA_D;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < SA_D; i += 4) {
        const f32x4 a = sa_ld4(o + i), c = sa_ld4(g + i);
        s += (a[0] * c[0] + a[1] * c[1]) + (a[2] * c[2] + a[3] * c[3]);
    }
    const_cast<float*>(p.Dv)[((long)b * p.h + head) * p.P + q] = s;
}

// ---------------------------------------------------------------------------------------------------------------------------
// dK, dV: one work-group = 64 keys of one (sample, head); wave = 16 keys; loops over the query tiles.
__global__ __launch_bounds__(256, 2) void sattn_bwd_kv_kernel(SAttnP p) {
    __shared__ __attribute__((aligned(16))) float Qs[2][SA_ROWSZ];       // Q tile  [q][d]
    __shared__ __attribute__((aligned(16))) float Gs[2][SA_ROWSZ];       // dO tile [q][dv]
    __shared__ float Ls[2][SA_T], Ds[2][SA_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, am = lane & 15, kq = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.h, head = bh - b * p.h, hoff = head * SA_D;
    const int key = blockIdx.x * SA_T + wave * 16 + am;
    const long brow = (long)b * p.P;
    float kb_[8], vb_[8];                                               // B operands: K (pre-scaled) and V rows of this lane's key
    {
        const float* kp = p.K + (brow + key) * p.E + hoff + 8 * kq;
        const float* vp = p.V + (brow + key) * p.E + hoff + 8 * kq;
        const f32x4 k0 = sa_ld4(kp), k1 = sa_ld4(kp + 4), v0 = sa_ld4(vp), v1 = sa_ld4(vp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { kb_[e] = k0[e] * p.scale; kb_[4 + e] = k1[e] * p.scale; vb_[e] = v0[e]; vb_[4 + e] = v1[e]; }
    }
    const bool drop = p.pdrop > 0.f;
    const uint64_t sd = drop ? p.seed[0] : 0ull;
    const uint32_t th = dropout_thresh(p.pdrop);
    const float sc = drop ? 1.f / (1.f - p.pdrop) : 1.f;
    f32x4 accK[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}}, accV[2] = {accK[0], accK[0]};
    const int nt = p.P / SA_T;
    f32x4 qa, qb2, ga, gb;
    float lq = 0.f, dq = 0.f;
    auto fetch = [&](int qt) {
        sa_fetch(p.Q, brow + (long)qt * SA_T, p.E, hoff, tid, qa, qb2);
        sa_fetch(p.dO, brow + (long)qt * SA_T, p.E, hoff, tid, ga, gb);
        if (tid < SA_T) { lq = p.lse[(long)bh * p.P + qt * SA_T + tid]; dq = p.Dv[(long)bh * p.P + qt * SA_T + tid]; }
    };
    auto stash = [&](int buf) {
        sa_store_rows(Qs[buf], tid, qa, qb2);
        sa_store_rows(Gs[buf], tid, ga, gb);
        if (tid < SA_T) { Ls[buf][tid] = lq; Ds[buf][tid] = dq; }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    for (int qt = 0; qt < nt; ++qt) {
        const int cur = qt & 1;
        if (qt + 1 < nt) fetch(qt + 1);
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            // S = Q K^T and dP = dO V^T blocks: lane holds [q = 16 qb + 4 kq + r][key]
            const f32x4 s = sa_mm8(Qs[cur] + (16 * qb + am) * SA_RP + 8 * kq, kb_);
            const f32x4 dp = sa_mm8(Gs[cur] + (16 * qb + am) * SA_RP + 8 * kq, vb_);
            float pd[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = 16 * qb + 4 * kq + r;
                const float pr = __expf(s[r] - Ls[cur][ql]);
                bool keep = true;
                if (drop) keep = dropout_keep(sd, p.site, ((uint64_t)bh * p.P + (uint64_t)(qt * SA_T + ql)) * (uint64_t)p.P + key, th);
                pd[r] = keep ? pr * sc : 0.f;
                ds[r] = pr * ((keep ? dp[r] * sc : 0.f) - Ds[cur][ql]);
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int off = (16 * qb + 4 * kq + r) * SA_RP + 16 * blk + am;
                    accV[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(pd[r], Gs[cur][off], accV[blk], 0, 0, 0);
                    accK[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[r], Qs[cur][off], accK[blk], 0, 0, 0);
                }
        }
        if (qt + 1 < nt) stash(cur ^ 1);
        __syncthreads();
    }
    // C layout: column = channel 16 blk + am, rows = keys 4 kq + r of this wave's 16
    const long krow = brow + blockIdx.x * SA_T + wave * 16 + 4 * kq;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long o = (krow + r) * p.E + hoff + 16 * blk + am;
            p.dK[o] = accK[blk][r] * p.scale;
            p.dV[o] = accV[blk][r];
        }
}

// dQ: one work-group = 64 queries of one (sample, head); wave = 16 queries; loops over key tiles.
__global__ __launch_bounds__(256, 2) void sattn_bwd_q_kernel(SAttnP p) {
    __shared__ __attribute__((aligned(16))) float Ks[2][SA_ROWSZ];       // K tile [key][d]
    __shared__ __attribute__((aligned(16))) float Vs[2][SA_ROWSZ];       // V tile [key][dv]
    __shared__ __attribute__((aligned(16))) float Kt[2][SA_TRSZ];        // K tile transposed [d][key]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, am = lane & 15, kq = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.h, head = bh - b * p.h, hoff = head * SA_D;
    const int q = blockIdx.x * SA_T + wave * 16 + am;
    const long brow = (long)b * p.P;
    float qb[8], gb_[8];
    {
        const float* qp = p.Q + (brow + q) * p.E + hoff + 8 * kq;
        const float* gp = p.dO + (brow + q) * p.E + hoff + 8 * kq;
        const f32x4 q0 = sa_ld4(qp), q1 = sa_ld4(qp + 4), g0 = sa_ld4(gp), g1 = sa_ld4(gp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { qb[e] = q0[e] * p.scale; qb[4 + e] = q1[e] * p.scale; gb_[e] = g0[e]; gb_[4 + e] = g1[e]; }
    }
    const float lse = p.lse[(long)bh * p.P + q], Dq = p.Dv[(long)bh * p.P + q];
    const bool drop = p.pdrop > 0.f;
    const uint64_t sd = drop ? p.seed[0] : 0ull;
    const uint32_t th = dropout_thresh(p.pdrop);
    const float sc = drop ? 1.f / (1.f - p.pdrop) : 1.f;
    const uint64_t rowidx = ((uint64_t)bh * p.P + q) * (uint64_t)p.P;
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    const int nt = p.P / SA_T;
    f32x4 ka, kb, va, vb;
    sa_fetch(p.K, brow, p.E, hoff, tid, ka, kb);
    sa_fetch(p.V, brow, p.E, hoff, tid, va, vb);
    sa_store_rows(Ks[0], tid, ka, kb);
    sa_store_tr(Kt[0], tid, ka, kb);
    sa_store_rows(Vs[0], tid, va, vb);
    __syncthreads();
    for (int kt = 0; kt < nt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nt) {
            sa_fetch(p.K, brow + (long)(kt + 1) * SA_T, p.E, hoff, tid, ka, kb);
            sa_fetch(p.V, brow + (long)(kt + 1) * SA_T, p.E, hoff, tid, va, vb);
        }
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            // S^T and dP^T blocks: lane holds [q][key = 16 jb + 4 kq + r]
            const f32x4 s = sa_mm8(Ks[cur] + (16 * jb + am) * SA_RP + 8 * kq, qb);
            const f32x4 dp = sa_mm8(Vs[cur] + (16 * jb + am) * SA_RP + 8 * kq, gb_);
            float ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = __expf(s[r] - lse);
                bool keep = true;
                if (drop) keep = dropout_keep(sd, p.site, rowidx + (uint64_t)(kt * SA_T + 16 * jb + 4 * kq + r), th);
                ds[r] = pr * ((keep ? dp[r] * sc : 0.f) - Dq);
            }
            // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const f32x4 k4 = sa_ld4(Kt[cur] + (16 * db + am) * SA_TP + 16 * jb + 4 * kq);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[r], ds[r], acc[db], 0, 0, 0);
            }
        }
        if (kt + 1 < nt) {
            sa_store_rows(Ks[cur ^ 1], tid, ka, kb);
            sa_store_tr(Kt[cur ^ 1], tid, ka, kb);
            sa_store_rows(Vs[cur ^ 1], tid, va, vb);
        }
        __syncthreads();
    }
    float* op = p.dQ + (brow + q) * p.E + hoff + 4 * kq;
    *reinterpret_cast<f32x4*>(op) = acc[0] * p.scale;
    *reinterpret_cast<f32x4*>(op + 16) = acc[1] * p.scale;
}

static int sa_check(int B, int P, int h, int E) { return (B < 1 || P < SA_T || P % SA_T || h < 1 || E != h * SA_D) ? 1 : 0; }

// O (B,P,E) = dropout(softmax(Q K^T * scale)) V per head of 32 channels (E = 32 h, P a multiple of 64); lse (B,h,P) = log-sum-exp of
// the scaled scores per query (kept for the backward).
// csrc/sattn2.hip: the split-bf16 kernels; -1 = not taken (generation 1 selected, or a geometry they do not cover)
int sattn2_fwd(const float* Q, const float* K, const float* V, float* O, float* lse, int B, int P, int h, float scale, float pdrop,
               const unsigned long long* seed, unsigned site, unsigned* bits, hipStream_t st);
int sattn2_bwd(const float* Q, const float* K, const float* V, const float* lse, const float* dO, const float* Dws, float* dQ, float* dK,
               float* dV, int B, int P, int h, float scale, float pdrop, const unsigned long long* seed, unsigned site, const unsigned* bits,
               hipStream_t st);
// `bits` (null, or B h P P / 32 words): where the forward leaves the keep decisions of the dropout masks for the backward (sattn2.hip);
// with null -- and in the fp32 kernels always -- every kernel recomputes them from (seed, site).  A backward must be given what its
// forward was given.
TATT_API int tatt_sattn_fwd_bits(const float* Q, const float* K, const float* V, float* O, float* lse, unsigned* bits, int B, int P, int h,
                                 float scale, float pdrop, const unsigned long long* seed, unsigned site, hipStream_t st) {
    const int E = h * SA_D;
    if (sa_check(B, P, h, E)) return 1;
    if (pdrop > 0.f && !seed) return 2;
    const int r2 = sattn2_fwd(Q, K, V, O, lse, B, P, h, scale, pdrop, seed, site, bits, st);
    if (r2 >= 0) return r2;
    SAttnP p = {Q, K, V, O, lse, nullptr, nullptr, nullptr, nullptr, nullptr, B, P, h, E, scale, pdrop, seed, site};
    hipLaunchKernelGGL(sattn_fwd_kernel, dim3(P / SA_T, B * h), dim3(256), 0, st, p);
    return LAUNCH_CHECK();
}
TATT_API int tatt_sattn_fwd(const float* Q, const float* K, const float* V, float* O, float* lse, int B, int P, int h, float scale,
                            float pdrop, const unsigned long long* seed, unsigned site, hipStream_t st) {
    return tatt_sattn_fwd_bits(Q, K, V, O, lse, nullptr, B, P, h, scale, pdrop, seed, site, st);
}
// Gradients of the same: Dws = workspace of B*h*P floats.  Three launches (D = rowsum(dO * O); dK, dV; dQ), no atomics.
TATT_API int tatt_sattn_bwd_bits(const float* Q, const float* K, const float* V, const float* O, const float* lse, const float* dO,
                                 const unsigned* bits, float* dQ, float* dK, float* dV, float* Dws, int B, int P, int h, float scale,
                                 float pdrop, const unsigned long long* seed, unsigned site, hipStream_t st) {
    const int E = h * SA_D;
    if (sa_check(B, P, h, E)) return 1;
    if (pdrop > 0.f && !seed) return 2;
    SAttnP p = {Q, K, V, const_cast<float*>(O), const_cast<float*>(lse), dO, Dws, dQ, dK, dV, B, P, h, E, scale, pdrop, seed, site};
    hipLaunchKernelGGL(sattn_prep_kernel, dim3(cdiv((long)B * P * h, 256)), dim3(256), 0, st, p);
    const int r2 = sattn2_bwd(Q, K, V, lse, dO, Dws, dQ, dK, dV, B, P, h, scale, pdrop, seed, site, bits, st);
    if (r2 >= 0) return r2;
    hipLaunchKernelGGL(sattn_bwd_kv_kernel, dim3(P / SA_T, B * h), dim3(256), 0, st, p);
    hipLaunchKernelGGL(sattn_bwd_q_kernel, dim3(P / SA_T, B * h), dim3(256), 0, st, p);
    return LAUNCH_CHECK();
}
TATT_API int tatt_sattn_bwd(const float* Q, const float* K, const float* V, const float* O, const float* lse, const float* dO,
                            float* dQ, float* dK, float* dV, float* Dws, int B, int P, int h, float scale, float pdrop,
                            const unsigned long long* seed, unsigned site, hipStream_t st) {
    return tatt_sattn_bwd_bits(Q, K, V, O, lse, dO, nullptr, dQ, dK, dV, Dws, B, P, h, scale, pdrop, seed, site, st);
}
