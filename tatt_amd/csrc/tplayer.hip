// One transformer layer of the TP interpreter as ONE kernel, forward and backward:
//     q = (x + qpos) Wq^T + bq, scaled by 1/sqrt(16)            (nn.MultiheadAttention in-projection, query rows)
//     P = softmax(q K^T) per head (4 x 16), dropout, ctx = P V,  w_avg = mean over heads of the dropped P
//     x1 = LN_A(x + dropout(ctx Wo^T + bo))
//     x2 = LN_B(x1 + dropout(W2 dropout(relu(W1 x1 + b1)) + b2))
//     [decoder, last layer] fin = c * (LN_F(x) [if fin_both] + LN_F(x2))
// i.e. reference TransformerDecoderLayer_TP.forward_post (model/transformer_v2.py:806-833; its self-attention is commented
// out upstream, :817-819) together with the stacked final norms of TransformerDecoder.forward (:380-390), and
// TransformerEncoderLayer.forward_post (:470-484) when x is the text sequence itself.  K and V (S <= 32 keys per sample:
// the 26-step CRNN text prior) are projected beforehand (tiny GEMMs); every query token is independent given them.
//
// Mapping: persistent work-groups of 8 waves; a tile = 32 tokens of ONE sample x 64 channels in LDS (pitch 68); the four
// 64x64 weight matrices (70 KB), K/V (17 KB) and seven tile buffers stay in LDS (150.8 KB, one work-group per CU).  Every
// 32x64x64 product runs on v_mfma_f32_16x16x4_f32 (exact fp32): wave (rb, cb) owns a 16x16 block, lane (i = lane & 15,
// kq = lane >> 4) reads 16 consecutive contraction elements with four ds_read_b128 and k-slot kq of step 4 c + u stands for
// element 16 kq + 4 c + u on BOTH operands.  The 26-key softmax runs on the vector ALU, four lanes per (token, head),
// probabilities exchanged through LDS inside the wave; LayerNorm rows are reduced with 16-lane shuffles.
// Backward: the tile is recomputed from x (nothing but x, qpos, K, V is saved by the forward), then walked in reverse;
// weight gradients (4 x 64x64) and dK / dV accumulate in MFMA accumulators across the work-group's tiles and leave as one
// partial record per work-group (per sample for dK / dV), summed deterministically by the two small reducers below.
// Dropout masks are those of the stand-alone kernels (same seed word, site and flat element index: tatt_attn_fwd,
// tatt_ln_fwd, tatt_dropout), so the fused layer reproduces the unfused operator chain mask for mask.
#include "common.h"
#include <mutex>

#define TL_M 32                 // tokens per tile
#define TL_P 68                 // LDS pitch (floats) of token tiles, weight matrices and K / V rows
#define TL_TSZ (TL_M * TL_P)    // 2176 floats per tile
#define TL_WSZ (64 * TL_P)      // 4352 floats per weight matrix
#define TL_KSZ (32 * TL_P)
#define TL_PSP 34               // pitch of the probability array [token][head][key]
#define TL_NT 512
#define TL_NVEC 10
#define TL_LDS_FLOATS (4 * TL_WSZ + 2 * TL_KSZ + 7 * TL_TSZ + TL_NVEC * 64 + 64)
#define TL_LDS_BYTES (TL_LDS_FLOATS * 4)
#define TL_PREC (4 * 4096 + TL_NVEC * 64)      // floats of one parameter-gradient partial record
#define TL_KVREC (2 * 32 * 64)                 // floats of one dK / dV partial record

struct TLP {
    const float* x; const float* qpos; long qbs;                 // x (B,L,64); qpos (B,L,64) [qbs = L*64] or (L,64) [qbs = 0]
    const float* K; const float* V;                              // (B,S,64) projected keys / values
    const float* Wm[4]; const float* bv[4];                      // Wq (query rows of in_proj), Wo, W1, W2; bq, bo, b1, b2
    const float* lnw[3]; const float* lnb[3];                    // A (after attention), B (after FFN), F (final norm; null: none)
    float fin_scale; int fin_both;
    float* xout; float* fin; float* wavg;                        // forward outputs (each may be null)
    int B, L, S, tps, ntiles, nper;                              // tiles per sample, tiles in all, tiles per work-group
    float p_attn, p_res, p_ffn; const unsigned long long* seed; unsigned site0; float eps;
    const float* dxout; const float* dfin; const float* dwavg; const float* dqacc;   // backward inputs (nullable)
    float* dx; float* dqpos; float* kvpart; float* ppart; int span;
};

__device__ __forceinline__ float tl_sum16(float v) { return row16_sum(v); }      // the 16 lanes of a token = one DPP row
__device__ __forceinline__ f32x4 tl_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void tl_st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// 16x16 block (rows 16 rb .., columns 16 cb ..) of  A (32 x 64 tile) * op(W):
//   BT = false: op(W)[k][n] = W[n][k]   (y = x W^T, nn.Linear forward);   BT = true: op(W)[k][n] = W[k][n]   (dx = dy W)
template <bool BT>
__device__ __forceinline__ f32x4 tl_gemm(const float* __restrict__ A, const float* __restrict__ W, int rb, int cb, int am, int kq) {
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const float* ap = A + (16 * rb + am) * TL_P + 16 * kq;
    f32x4 a[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] = tl_ld4(ap + 4 * c);
    if (!BT) {
        const float* bp = W + (16 * cb + am) * TL_P + 16 * kq;
        f32x4 b[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) b[c] = tl_ld4(bp + 4 * c);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][0], b[c][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][1], b[c][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][2], b[c][2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][3], b[c][3], acc1, 0, 0, 0);
        }
    } else {
        const float* bp = W + (16 * kq) * TL_P + 16 * cb + am;
        float b[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) b[j] = bp[j * TL_P];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][0], b[4 * c + 0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][1], b[4 * c + 1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][2], b[4 * c + 2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][3], b[4 * c + 3], acc1, 0, 0, 0);
        }
    }
    return acc0 + acc1;
}
// dW[n][k] += sum_tokens G[t][n] X[t][k]: wave (nb, kb2) owns rows 16 nb .. and columns 32 kb2 .. (two 16x16 blocks).
// acc0[r] <-> dW[16 nb + 4 kq + r][32 kb2 + am], acc1[r] <-> column + 16.
// bsum += the A operands this lane feeds: sum over tokens = kq (mod 4) of G[t][16 nb + am] -- the bias gradient rides along.
__device__ __forceinline__ void tl_wgrad(const float* __restrict__ G, const float* __restrict__ X, int nb, int kb2, int am, int kq,
                                         f32x4& acc0, f32x4& acc1, float& bsum) {
    const float* gp = G + kq * TL_P + 16 * nb + am;
    const float* xp = X + kq * TL_P + 32 * kb2 + am;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const float a = gp[4 * s * TL_P];
        const float b0 = xp[4 * s * TL_P], b1 = xp[4 * s * TL_P + 16];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc1, 0, 0, 0);
        bsum += a;
    }
}
// acc[r] <-> [key 16 sb + 4 kq + r][channel 16 h + am] += sum_tokens PS[t][h][key] * T[t][16 h + am]
__device__ __forceinline__ void tl_kvgrad(const float* __restrict__ PS, const float* __restrict__ T, int h, int sb, int am, int kq,
                                          f32x4& acc) {
    const float* pp = PS + (kq * 4 + h) * TL_PSP + 16 * sb + am;
    const float* tp = T + kq * TL_P + 16 * h + am;
#pragma unroll
    for (int s = 0; s < 8; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pp[16 * s * TL_PSP], tp[4 * s * TL_P], acc, 0, 0, 0);
}

template <bool BWD>
__global__ __launch_bounds__(TL_NT, 1) void tplayer_kernel(TLP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Wl = smem;                                  // [4][64][68]
    float* const Ks = Wl + 4 * TL_WSZ;                       // [32][68]
    float* const Vs = Ks + TL_KSZ;
    float* const T0 = Vs + TL_KSZ;                           // x -> y1 = x + drop(attention output)
    float* const T1 = T0 + TL_TSZ;                           // x + qpos -> ctx
    float* const T2 = T1 + TL_TSZ;                           // scaled query projection
    float* const T3 = T2 + TL_TSZ;                           // x1 (-> y2 in the forward kernel)
    float* const T4 = T3 + TL_TSZ;                           // dropout(relu(W1 x1 + b1))
    float* const T5 = T4 + TL_TSZ;                           // gradient ping-pong (y2 in the backward kernel)
    float* const T6 = T5 + TL_TSZ;
    float* const Vec = T6 + TL_TSZ;                          // bq, bo, b1, b2, gA, bA, gB, bB, gF, bF
    float* const Stat = Vec + TL_NVEC * 64;                  // mean / rstd of LN_A per token (backward)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int am = lane & 15, kq = lane >> 4;
    const int rb = wave & 1, cb = wave >> 1;                 // GEMM block of this wave
    const int rt = tid >> 4, rc = 4 * (tid & 15);            // row mapping: token rt, channels rc .. rc + 3
    const int ah = (tid >> 2) & 3, au = tid & 3;             // attention mapping: token rt, head ah, key lane au (16 ah + 4 au == rc)
    const bool fin_on = p.lnw[2] != nullptr;

    // ---- resident operands ------------------------------------------------------------------------------------------------
    for (int i = tid; i < 4 * 64 * 16; i += TL_NT) {
        const int m = i >> 10, r = (i >> 4) & 63, q = i & 15;
        tl_st4(Wl + m * TL_WSZ + r * TL_P + 4 * q, tl_ld4(p.Wm[m] + r * 64 + 4 * q));
    }
    for (int i = tid; i < TL_NVEC * 64; i += TL_NT) {
        const int v = i >> 6, c = i & 63;
        const float* src = v < 4 ? p.bv[v] : (((v - 4) & 1) ? p.lnb[(v - 4) >> 1] : p.lnw[(v - 4) >> 1]);
        Vec[i] = src ? src[c] : 0.f;
    }
    const float* const bq = Vec, * const bo = Vec + 64, * const b1 = Vec + 128, * const b2 = Vec + 192;
    const float* const gA = Vec + 256, * const bA = Vec + 320, * const gB = Vec + 384, * const bB = Vec + 448;
    const float* const gF = Vec + 512, * const bF = Vec + 576;

    const uint64_t sd = (p.p_attn > 0.f || p.p_res > 0.f || p.p_ffn > 0.f) ? p.seed[0] : 0ull;
    const uint32_t th_attn = dropout_thresh(p.p_attn), th_res = dropout_thresh(p.p_res), th_ffn = dropout_thresh(p.p_ffn);
    const float sc_attn = p.p_attn > 0.f ? 1.f / (1.f - p.p_attn) : 1.f;
    const float sc_res = p.p_res > 0.f ? 1.f / (1.f - p.p_res) : 1.f;
    const float sc_ffn = p.p_ffn > 0.f ? 1.f / (1.f - p.p_ffn) : 1.f;

    // ---- accumulators that live across the tiles of this work-group (backward) ------------------------------------------------
    f32x4 dWa[4][2];                                         // [matrix][column half]
    f32x4 accK, accV;
    float dbs[4] = {0.f, 0.f, 0.f, 0.f};                     // bias gradients (bq, bo, b1, b2): per-lane partial sums, see tl_wgrad
    f32x4 dgAa, dbAa, dgBa, dbBa, dgFa, dbFa;
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (BWD) {
#pragma unroll
        for (int m = 0; m < 4; ++m) { dWa[m][0] = z4; dWa[m][1] = z4; }
        accK = z4; accV = z4;
        dgAa = z4; dbAa = z4; dgBa = z4; dbBa = z4; dgFa = z4; dbFa = z4;
    }

    const int tbeg = blockIdx.x * p.nper;
    const int tend = min(p.ntiles, tbeg + p.nper);
    const int b_first = tbeg / p.tps;
    int cur_b = -1;
    // dK / dV of the sample this work-group has been walking: one partial record per (work-group, sample it touches)
    auto flush_kv = [&](int bsample) {
        float* rec = p.kvpart + ((long)blockIdx.x * p.span + (bsample - b_first)) * TL_KVREC;
        const int h = wave & 3, sb = wave >> 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = 16 * sb + 4 * kq + r;
            rec[s * 64 + 16 * h + am] = accK[r];
            rec[32 * 64 + s * 64 + 16 * h + am] = accV[r];
        }
        accK = z4; accV = z4;
    };
    // LayerNorm of a row quad held by the 16 lanes of a token: returns xhat, mean-free; rstd out
    auto ln_row = [&](f32x4 v, f32x4& xh, float& rstd) {
        const float mean = tl_sum16((v[0] + v[1]) + (v[2] + v[3])) * (1.f / 64.f);
        f32x4 d = v - mean;
        const float var = tl_sum16((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.f / 64.f);
        rstd = __builtin_amdgcn_rsqf(var + p.eps);
        xh = d * rstd;
    };
    // LayerNorm backward of a row quad: g = upstream gradient, xh = normalised row, gam = gamma quad -> gradient of the LN input
    auto ln_row_bwd = [&](f32x4 g, f32x4 xh, f32x4 gam, float rstd) -> f32x4 {
        const f32x4 gg = g * gam;
        const float s1 = tl_sum16((gg[0] + gg[1]) + (gg[2] + gg[3])) * (1.f / 64.f);
        const float s2 = tl_sum16((gg[0] * xh[0] + gg[1] * xh[1]) + (gg[2] * xh[2] + gg[3] * xh[3])) * (1.f / 64.f);
        return (gg - s1 - xh * s2) * rstd;
    };

    for (int tile = tbeg; tile < tend; ++tile) {
        const int b = tile / p.tps, tok0 = (tile - b * p.tps) * TL_M;
        const int valid = min(TL_M, p.L - tok0);
        const long row0 = (long)b * p.L + tok0;              // global token row of tile row 0
        __syncthreads();                                     // (S0) the previous tile has left every buffer
        if (b != cur_b) {
            if (BWD && cur_b >= 0) flush_kv(cur_b);
            for (int i = tid; i < 32 * 16; i += TL_NT) {
                const int s = i >> 4, q = i & 15;
                f32x4 kv = z4, vv = z4;
                if (s < p.S) {
                    kv = tl_ld4(p.K + ((long)b * p.S + s) * 64 + 4 * q);
                    vv = tl_ld4(p.V + ((long)b * p.S + s) * 64 + 4 * q);
                }
                tl_st4(Ks + s * TL_P + 4 * q, kv);
                tl_st4(Vs + s * TL_P + 4 * q, vv);
            }
            cur_b = b;
        }
        // ---- P0: x, x + qpos -> LDS ---------------------------------------------------------------------------------------
        const bool rv = rt < valid;
        f32x4 xr = z4, qr = z4;
        if (rv) {
            xr = tl_ld4(p.x + (row0 + rt) * 64 + rc);
            qr = tl_ld4(p.qpos + (long)b * p.qbs + (long)(tok0 + rt) * 64 + rc);
        }
        tl_st4(T0 + rt * TL_P + rc, xr);
        tl_st4(T1 + rt * TL_P + rc, xr + qr);
        f32x4 fin0 = z4;                                     // forward: LN_F(x)
        if (!BWD && fin_on && p.fin_both) {
            f32x4 xh; float rs;
            ln_row(xr, xh, rs);
            fin0 = xh * tl_ld4(gF + rc) + tl_ld4(bF + rc);
        }
        __syncthreads();                                     // (S1)
        // ---- P1: Q = (qin Wq^T + bq) / 4 ------------------------------------------------------------------------------------
        {
            const f32x4 acc = tl_gemm<false>(T1, Wl, rb, cb, am, kq);
            const int col = 16 * cb + am;
            const float bj = bq[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) T2[(16 * rb + 4 * kq + r) * TL_P + col] = 0.25f * (acc[r] + bj);
        }
        __syncthreads();                                     // (S2)
        // ---- P2: attention of (token rt, head ah): keys 4 i + au; probabilities through LDS (wave-local), ctx -> T1 -------------
        float pr[8];                                         // un-dropped probabilities of the own keys (kept for the backward)
        const long abase = (((long)b * 4 + ah) * p.L + tok0 + rt) * p.S;      // flat index of (b, head, query, key 0)
        {
            float q[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = tl_ld4(T2 + rt * TL_P + 16 * ah + 4 * j);
                q[4 * j] = v[0]; q[4 * j + 1] = v[1]; q[4 * j + 2] = v[2]; q[4 * j + 3] = v[3];
            }
            float mx = -INFINITY;
#pragma unroll(BWD ? 2 : 8)
            for (int i = 0; i < 8; ++i) {
                const int s = 4 * i + au;
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 kv = tl_ld4(Ks + s * TL_P + 16 * ah + 4 * j);
                    a = fmaf(q[4 * j], kv[0], a); a = fmaf(q[4 * j + 1], kv[1], a);
                    a = fmaf(q[4 * j + 2], kv[2], a); a = fmaf(q[4 * j + 3], kv[3], a);
                }
                pr[i] = s < p.S ? a : -INFINITY;
                mx = fmaxf(mx, pr[i]);
            }
            mx = quad_max(mx);                              // the four key lanes of a (token, head) are one quad
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { pr[i] = (4 * i + au) < p.S ? __expf(pr[i] - mx) : 0.f; sum += pr[i]; }
            sum = quad_sum(sum);
            const float inv = __builtin_amdgcn_rcpf(sum);
            float* PS = T5 + (rt * 4 + ah) * TL_PSP;         // forward exchange buffer: T5..T6
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int s = 4 * i + au;
                pr[i] *= inv;
                float pd = pr[i];
                if (p.p_attn > 0.f && s < p.S) pd = dropout_keep(sd, p.site0, (uint64_t)(abase + s), th_attn) ? pd * sc_attn : 0.f;
                PS[s] = pd;
            }
            wave_lds_sync();
            f32x4 c4 = z4;
            for (int s = 0; s < p.S; s += 2) {               // (rows / probabilities beyond S are zero)
                const float2 pp = *reinterpret_cast<const float2*>(PS + s);
                c4 += tl_ld4(Vs + s * TL_P + rc) * pp.x + tl_ld4(Vs + (s + 1) * TL_P + rc) * pp.y;
            }
            tl_st4(T1 + rt * TL_P + rc, c4);                 // (every wave finished reading x + qpos at S2)
            if (!BWD && p.wavg) {
                const float* P0 = T5 + (rt * 4) * TL_PSP;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int s = (rc >> 1) + e;             // keys 2 cq, 2 cq + 1 of this token
                    if (rv && s < p.S)
                        p.wavg[(row0 + rt) * p.S + s] = 0.25f * ((P0[s] + P0[TL_PSP + s]) + (P0[2 * TL_PSP + s] + P0[3 * TL_PSP + s]));
                }
            }
        }
        __syncthreads();                                     // (S3)
        // ---- P3: y1 = x + dropout(ctx Wo^T + bo), in place over x ---------------------------------------------------------------
        {
            const f32x4 acc = tl_gemm<false>(T1, Wl + TL_WSZ, rb, cb, am, kq);
            const int col = 16 * cb + am;
            const float bj = bo[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rb + 4 * kq + r;
                float a = acc[r] + bj;
                if (p.p_res > 0.f) a = dropout_keep(sd, p.site0 + 1, (uint64_t)((row0 + row) * 64 + col), th_res) ? a * sc_res : 0.f;
                T0[row * TL_P + col] += a;
            }
        }
        __syncthreads();                                     // (S4)
        // ---- P4: x1 = LN_A(y1) -> T3 ------------------------------------------------------------------------------------------
        {
            f32x4 xh; float rs;
            const f32x4 y = tl_ld4(T0 + rt * TL_P + rc);
            ln_row(y, xh, rs);
            tl_st4(T3 + rt * TL_P + rc, xh * tl_ld4(gA + rc) + tl_ld4(bA + rc));
            if (BWD) {
                tl_st4(T0 + rt * TL_P + rc, xh);             // the backward needs the normalised row, not y1
                if ((tid & 15) == 0) Stat[rt] = rs;
            }
        }
        __syncthreads();                                     // (S5)
        // ---- P5: hd = dropout(relu(x1 W1^T + b1)) -> T4 -------------------------------------------------------------------------
        {
            const f32x4 acc = tl_gemm<false>(T3, Wl + 2 * TL_WSZ, rb, cb, am, kq);
            const int col = 16 * cb + am;
            const float bj = b1[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rb + 4 * kq + r;
                float h = fmaxf(acc[r] + bj, 0.f);
                if (p.p_ffn > 0.f) h = dropout_keep(sd, p.site0 + 2, (uint64_t)((row0 + row) * 64 + col), th_ffn) ? h * sc_ffn : 0.f;
                T4[row * TL_P + col] = h;
            }
        }
        __syncthreads();                                     // (S6)
        // ---- P6: y2 = x1 + dropout(hd W2^T + b2): forward in place over x1, backward into T5 --------------------------------------
        {
            const f32x4 acc = tl_gemm<false>(T4, Wl + 3 * TL_WSZ, rb, cb, am, kq);
            const int col = 16 * cb + am;
            const float bj = b2[col];
            float* dst = BWD ? T5 : T3;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rb + 4 * kq + r;
                float f = acc[r] + bj;
                if (p.p_res > 0.f) f = dropout_keep(sd, p.site0 + 3, (uint64_t)((row0 + row) * 64 + col), th_res) ? f * sc_res : 0.f;
                dst[row * TL_P + col] = T3[row * TL_P + col] + f;
            }
        }
        __syncthreads();                                     // (S7)
        if (!BWD) {
            // ---- P7: x2 = LN_B(y2) -> global; fin = c (LN_F(x) + LN_F(x2)) ---------------------------------------------------------
            f32x4 xh; float rs;
            ln_row(tl_ld4(T3 + rt * TL_P + rc), xh, rs);
            const f32x4 x2 = xh * tl_ld4(gB + rc) + tl_ld4(bB + rc);
            if (rv && p.xout) tl_st4(p.xout + (row0 + rt) * 64 + rc, x2);
            if (fin_on) {
                f32x4 xf; float rf;
                ln_row(x2, xf, rf);
                const f32x4 o = (fin0 + xf * tl_ld4(gF + rc) + tl_ld4(bF + rc)) * p.fin_scale;
                if (rv && p.fin) tl_st4(p.fin + (row0 + rt) * 64 + rc, o);
            }
            continue;
        }
        // =========================================== backward of the tile ======================================================
        // ---- B7: LN_B forward + backward on the row; df = dropout'(dy2) -> T6 ------------------------------------------------------
        f32x4 dx1r;                                          // gradient reaching x1 through the residual of the FFN block
        {
            f32x4 xh2; float rs2;
            ln_row(tl_ld4(T5 + rt * TL_P + rc), xh2, rs2);
            f32x4 g2 = z4;
            if (rv && p.dxout) g2 = tl_ld4(p.dxout + (row0 + rt) * 64 + rc);
            const f32x4 gamB = tl_ld4(gB + rc);
            if (fin_on) {
                f32x4 dfr = z4;
                if (rv) dfr = tl_ld4(p.dfin + (row0 + rt) * 64 + rc) * p.fin_scale;
                const f32x4 x2 = xh2 * gamB + tl_ld4(bB + rc);
                f32x4 xf; float rf;
                ln_row(x2, xf, rf);
                dgFa += dfr * xf; dbFa += dfr;
                g2 += ln_row_bwd(dfr, xf, tl_ld4(gF + rc), rf);
            }
            dgBa += g2 * xh2; dbBa += g2;
            dx1r = ln_row_bwd(g2, xh2, gamB, rs2);
            f32x4 df = dx1r;
            if (p.p_res > 0.f) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    df[e] = dropout_keep(sd, p.site0 + 3, (uint64_t)((row0 + rt) * 64 + rc + e), th_res) ? df[e] * sc_res : 0.f;
            }
            tl_st4(T6 + rt * TL_P + rc, df);
        }
        __syncthreads();                                     // (S8)
        // ---- B8: dhp = relu'/dropout' (df W2) -> T5;  dW2 += df^T hd -----------------------------------------------------------------
        {
            const f32x4 acc = tl_gemm<true>(T6, Wl + 3 * TL_WSZ, rb, cb, am, kq);
            const int col = 16 * cb + am;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rb + 4 * kq + r;
                T5[row * TL_P + col] = T4[row * TL_P + col] > 0.f ? acc[r] * sc_ffn : 0.f;   // hd > 0 <=> relu active and kept
            }
            tl_wgrad(T6, T4, wave & 3, wave >> 2, am, kq, dWa[3][0], dWa[3][1], dbs[3]);
        }
        __syncthreads();                                     // (S9)
        // ---- B9: (dhp W1) -> T6;  dW1 += dhp^T x1 -------------------------------------------------------------------------------------
        {
            const f32x4 acc = tl_gemm<true>(T5, Wl + 2 * TL_WSZ, rb, cb, am, kq);
            const int col = 16 * cb + am;
#pragma unroll
            for (int r = 0; r < 4; ++r) T6[(16 * rb + 4 * kq + r) * TL_P + col] = acc[r];
            tl_wgrad(T5, T3, wave & 3, wave >> 2, am, kq, dWa[2][0], dWa[2][1], dbs[2]);
        }
        __syncthreads();                                     // (S10)
        // ---- B10: LN_A backward; da = dropout'(dy1) -> T5 ------------------------------------------------------------------------------
        f32x4 dxr;                                           // gradient reaching x through the attention block's residual
        {
            const f32x4 g1 = dx1r + tl_ld4(T6 + rt * TL_P + rc);
            const f32x4 xh1 = tl_ld4(T0 + rt * TL_P + rc);
            dgAa += g1 * xh1; dbAa += g1;
            dxr = ln_row_bwd(g1, xh1, tl_ld4(gA + rc), Stat[rt]);
            f32x4 da = dxr;
            if (p.p_res > 0.f) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    da[e] = dropout_keep(sd, p.site0 + 1, (uint64_t)((row0 + rt) * 64 + rc + e), th_res) ? da[e] * sc_res : 0.f;
            }
            tl_st4(T5 + rt * TL_P + rc, da);
        }
        __syncthreads();                                     // (S11)
        // ---- B11: dctx = da Wo -> T6;  dWo += da^T ctx ------------------------------------------------------------------------------------
        {
            const f32x4 acc = tl_gemm<true>(T5, Wl + TL_WSZ, rb, cb, am, kq);
            const int col = 16 * cb + am;
#pragma unroll
            for (int r = 0; r < 4; ++r) T6[(16 * rb + 4 * kq + r) * TL_P + col] = acc[r];
            tl_wgrad(T5, T1, wave & 3, wave >> 2, am, kq, dWa[1][0], dWa[1][1], dbs[1]);
        }
        __syncthreads();                                     // (S12)
        // ---- B12: attention backward, own keys: dropped probabilities -> PS (T3..T4), score gradients in registers ----------------------
        float* const PSb = T3;                               // x1 and hd are consumed
        float ds[8];
        {
            float g[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = tl_ld4(T6 + rt * TL_P + 16 * ah + 4 * j);
                g[4 * j] = v[0]; g[4 * j + 1] = v[1]; g[4 * j + 2] = v[2]; g[4 * j + 3] = v[3];
            }
            float dot = 0.f;
            float* PS = PSb + (rt * 4 + ah) * TL_PSP;
#pragma unroll 2
            for (int i = 0; i < 8; ++i) {
                const int s = 4 * i + au;
                float d = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 vv = tl_ld4(Vs + s * TL_P + 16 * ah + 4 * j);
                    d = fmaf(g[4 * j], vv[0], d); d = fmaf(g[4 * j + 1], vv[1], d);
                    d = fmaf(g[4 * j + 2], vv[2], d); d = fmaf(g[4 * j + 3], vv[3], d);
                }
                const bool live = s < p.S;
                if (p.dwavg && rv && live) d += 0.25f * p.dwavg[(row0 + rt) * p.S + s];
                const bool keep = !(p.p_attn > 0.f) || dropout_keep(sd, p.site0, (uint64_t)(abase + s), th_attn);
                const float pd = (live && keep) ? pr[i] * sc_attn : 0.f;       // what multiplied V
                d = (live && keep) ? d * sc_attn : 0.f;                          // gradient of the un-dropped probability
                PS[s] = pd;
                ds[i] = d;
                dot = fmaf(pr[i], d, dot);
            }
            dot = quad_sum(dot);
#pragma unroll
            for (int i = 0; i < 8; ++i) ds[i] = pr[i] * (ds[i] - dot);
        }
        __syncthreads();                                     // (S13)
        tl_kvgrad(PSb, T6, wave & 3, wave >> 2, am, kq, accV);                   // dV += Pd^T dctx
        __syncthreads();                                     // (S14)
        {
            float* PS = PSb + (rt * 4 + ah) * TL_PSP;
#pragma unroll
            for (int i = 0; i < 8; ++i) PS[4 * i + au] = ds[i];
        }
        __syncthreads();                                     // (S15)
        // ---- B15: dK += dS^T Q;  dq = dS K / 4 -> T5;  x + qpos -> T1 --------------------------------------------------------------------
        {
            tl_kvgrad(PSb, T2, wave & 3, wave >> 2, am, kq, accK);
            const float* PS = PSb + (rt * 4 + ah) * TL_PSP;
            f32x4 dq = z4;
            for (int s = 0; s < p.S; s += 2) {
                const float2 pp = *reinterpret_cast<const float2*>(PS + s);
                dq += tl_ld4(Ks + s * TL_P + rc) * pp.x + tl_ld4(Ks + (s + 1) * TL_P + rc) * pp.y;
            }
            dq *= 0.25f;
            tl_st4(T5 + rt * TL_P + rc, dq);
            f32x4 xq = z4;                                   // x + qpos again (L2-resident; cheaper than 8 registers held all tile)
            if (rv) xq = tl_ld4(p.x + (row0 + rt) * 64 + rc) + tl_ld4(p.qpos + (long)b * p.qbs + (long)(tok0 + rt) * 64 + rc);
            tl_st4(T1 + rt * TL_P + rc, xq);
        }
        __syncthreads();                                     // (S16)
        // ---- B16: dqin = dq Wq -> T6;  dWq += dq^T (x + qpos) ------------------------------------------------------------------------------
        {
            const f32x4 acc = tl_gemm<true>(T5, Wl, rb, cb, am, kq);
            const int col = 16 * cb + am;
#pragma unroll
            for (int r = 0; r < 4; ++r) T6[(16 * rb + 4 * kq + r) * TL_P + col] = acc[r];
            tl_wgrad(T5, T1, wave & 3, wave >> 2, am, kq, dWa[0][0], dWa[0][1], dbs[0]);
        }
        __syncthreads();                                     // (S17)
        // ---- B17: dx = residual path + query path (+ LN_F(x) path); dqpos ---------------------------------------------------------------------
        {
            f32x4 gx = z4;                                   // gradient reaching x through LN_F(x) (last decoder layer)
            if (fin_on && p.fin_both) {
                f32x4 xin = z4, dfr = z4, xh; float rs;
                if (rv) { xin = tl_ld4(p.x + (row0 + rt) * 64 + rc); dfr = tl_ld4(p.dfin + (row0 + rt) * 64 + rc) * p.fin_scale; }
                ln_row(xin, xh, rs);
                dgFa += dfr * xh; dbFa += dfr;
                gx = ln_row_bwd(dfr, xh, tl_ld4(gF + rc), rs);
            }
            dxr += gx;
        }
        if (rv) {
            const f32x4 dqin = tl_ld4(T6 + rt * TL_P + rc);
            tl_st4(p.dx + (row0 + rt) * 64 + rc, dxr + dqin);
            if (p.dqpos) {
                f32x4 o = dqin;
                if (p.dqacc) o += tl_ld4(p.dqacc + (row0 + rt) * 64 + rc);
                tl_st4(p.dqpos + (row0 + rt) * 64 + rc, o);
            }
        }
    }
    if (!BWD) return;
    // ---- one partial record per work-group --------------------------------------------------------------------------------------
    if (cur_b >= 0) flush_kv(cur_b);
    float* rec = p.ppart + (long)blockIdx.x * TL_PREC;
    {
        const int nb = wave & 3, kb2 = wave >> 2;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* d = rec + m * 4096 + (16 * nb + 4 * kq + r) * 64 + 32 * kb2 + am;
                d[0] = dWa[m][0][r];
                d[16] = dWa[m][1][r];
            }
    }
    // row-mapped vectors: token slot rt holds partial column sums of channels rc..rc+3 -> sum over the 32 slots through LDS
    auto col_reduce = [&](f32x4 v, int slot) {
        __syncthreads();
        tl_st4(T0 + rt * TL_P + rc, v);
        __syncthreads();
        if (tid < 64) {
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < TL_M; ++r) s += T0[r * TL_P + tid];
            rec[4 * 4096 + slot * 64 + tid] = s;
        }
    };
    col_reduce(dgAa, 4); col_reduce(dbAa, 5); col_reduce(dgBa, 6); col_reduce(dbBa, 7);
    col_reduce(dgFa, 8); col_reduce(dbFa, 9);
    // bias gradients: column 16 nb + am = wave nb (waves 0-3: kb2 = 0), lanes am + 16 kq, kq = 0..3
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 4; ++v) T1[v * TL_NT + tid] = dbs[v];
    __syncthreads();
    if (tid < 256) {
        const int v = tid >> 6, c = tid & 63;
        const float* src = T1 + v * TL_NT + (c >> 4) * 64 + (c & 15);
        rec[4 * 4096 + v * 64 + c] = (src[0] + src[16]) + (src[32] + src[48]);
    }
}

// ---- geometry shared by the launchers and the reducers ---------------------------------------------------------------------------
struct TLGeom { int tps, ntiles, G, nper, span; };
static inline TLGeom tl_geom(int B, int L) {
    TLGeom g;
    g.tps = cdiv(L, TL_M);
    g.ntiles = B * g.tps;
    const int G0 = g.ntiles < 256 ? g.ntiles : 256;
    g.nper = cdiv(g.ntiles, G0);
    g.G = cdiv(g.ntiles, g.nper);
    g.span = (g.nper + g.tps - 2) / g.tps + 1;               // samples a run of nper consecutive tiles can touch
    return g;
}
static void tl_set_attr() {
    static TattPerDevice once;
    tatt_per_device(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tplayer_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, TL_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tplayer_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, TL_LDS_BYTES);
    });
}
// out[0] = work-groups, out[1] = dK/dV partial records per work-group, out[2] = floats of kvpart, out[3] = floats of ppart
TATT_API int tatt_tplayer_geom(int B, int L, int* out) {
    const TLGeom g = tl_geom(B, L);
    out[0] = g.G; out[1] = g.span; out[2] = g.G * g.span * TL_KVREC; out[3] = g.G * TL_PREC;
    return 0;
}

static TLP tl_params(const float* x, const float* qpos, long qbs, const float* K, const float* V, const float* in_w, const float* in_b,
                     const float* out_w, const float* out_b, const float* w1, const float* b1, const float* w2, const float* b2,
                     const float* lnA_w, const float* lnA_b, const float* lnB_w, const float* lnB_b, const float* lnF_w,
                     const float* lnF_b, float fin_scale, int fin_both, int B, int L, int S, float p_attn, float p_res, float p_ffn,
                     const unsigned long long* seed, unsigned site0, float eps) {
    TLP p = {};
    p.x = x; p.qpos = qpos; p.qbs = qbs; p.K = K; p.V = V;
    p.Wm[0] = in_w; p.Wm[1] = out_w; p.Wm[2] = w1; p.Wm[3] = w2;
    p.bv[0] = in_b; p.bv[1] = out_b; p.bv[2] = b1; p.bv[3] = b2;
    p.lnw[0] = lnA_w; p.lnw[1] = lnB_w; p.lnw[2] = lnF_w;
    p.lnb[0] = lnA_b; p.lnb[1] = lnB_b; p.lnb[2] = lnF_b;
    p.fin_scale = fin_scale; p.fin_both = fin_both;
    p.B = B; p.L = L; p.S = S;
    const TLGeom g = tl_geom(B, L);
    p.tps = g.tps; p.ntiles = g.ntiles; p.nper = g.nper; p.span = g.span;
    p.p_attn = p_attn; p.p_res = p_res; p.p_ffn = p_ffn; p.seed = seed; p.site0 = site0; p.eps = eps;
    return p;
}

TATT_API int tatt_tplayer_fwd(const float* x, const float* qpos, long qbs, const float* K, const float* V, const float* in_w,
                              const float* in_b, const float* out_w, const float* out_b, const float* w1, const float* b1,
                              const float* w2, const float* b2, const float* lnA_w, const float* lnA_b, const float* lnB_w,
                              const float* lnB_b, const float* lnF_w, const float* lnF_b, float fin_scale, int fin_both, float* xout,
                              float* fin, float* wavg, int B, int L, int S, float p_attn, float p_res, float p_ffn,
                              const unsigned long long* seed, unsigned site0, float eps, hipStream_t st) {
    if (S < 1 || S > 32 || B < 1 || L < 1) return 1;
    if ((p_attn > 0.f || p_res > 0.f || p_ffn > 0.f) && !seed) return 2;
    TLP p = tl_params(x, qpos, qbs, K, V, in_w, in_b, out_w, out_b, w1, b1, w2, b2, lnA_w, lnA_b, lnB_w, lnB_b, lnF_w, lnF_b,
                      fin_scale, fin_both, B, L, S, p_attn, p_res, p_ffn, seed, site0, eps);
    p.xout = xout; p.fin = fin; p.wavg = wavg;
    tl_set_attr();
    hipLaunchKernelGGL(tplayer_kernel<false>, dim3(tl_geom(B, L).G), dim3(TL_NT), TL_LDS_BYTES, st, p);
    return LAUNCH_CHECK();
}

TATT_API int tatt_tplayer_bwd(const float* x, const float* qpos, long qbs, const float* K, const float* V, const float* in_w,
                              const float* in_b, const float* out_w, const float* out_b, const float* w1, const float* b1,
                              const float* w2, const float* b2, const float* lnA_w, const float* lnA_b, const float* lnB_w,
                              const float* lnB_b, const float* lnF_w, const float* lnF_b, float fin_scale, int fin_both,
                              const float* dxout, const float* dfin, const float* dwavg, const float* dqacc, float* dx,
                              float* dqpos, float* kvpart, float* ppart, int B, int L, int S, float p_attn, float p_res,
                              float p_ffn, const unsigned long long* seed, unsigned site0, float eps, hipStream_t st) {
    if (S < 1 || S > 32 || B < 1 || L < 1) return 1;
    if ((p_attn > 0.f || p_res > 0.f || p_ffn > 0.f) && !seed) return 2;
    if (lnF_w && !dfin) return 3;
    TLP p = tl_params(x, qpos, qbs, K, V, in_w, in_b, out_w, out_b, w1, b1, w2, b2, lnA_w, lnA_b, lnB_w, lnB_b, lnF_w, lnF_b,
                      fin_scale, fin_both, B, L, S, p_attn, p_res, p_ffn, seed, site0, eps);
    p.dxout = dxout; p.dfin = dfin; p.dwavg = dwavg; p.dqacc = dqacc;
    p.dx = dx; p.dqpos = dqpos; p.kvpart = kvpart; p.ppart = ppart;
    tl_set_attr();
    hipLaunchKernelGGL(tplayer_kernel<true>, dim3(tl_geom(B, L).G), dim3(TL_NT), TL_LDS_BYTES, st, p);
    return LAUNCH_CHECK();
}

// ---- dK / dV: sum the records of the work-groups that walked sample b ------------------------------------------------------------
__global__ __launch_bounds__(256) void tplayer_reduce_kv_kernel(const float* __restrict__ part, float* __restrict__ dK,
                                                                float* __restrict__ dV, int B, int S, TLGeom g) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int n = S * 64;
    if (idx >= (long)B * n) return;
    const int b = (int)(idx / n), i = (int)(idx % n);
    const int w_lo = (b * g.tps) / g.nper, w_hi = ((b + 1) * g.tps - 1) / g.nper;
    float sk = 0.f, sv = 0.f;
    for (int w = w_lo; w <= w_hi; ++w) {
        const float* rec = part + ((long)w * g.span + (b - (w * g.nper) / g.tps)) * TL_KVREC;
        sk += rec[i]; sv += rec[32 * 64 + i];
    }
    dK[idx] = sk; dV[idx] = sv;
}
TATT_API int tatt_tplayer_reduce_kv(const float* kvpart, float* dK, float* dV, int B, int L, int S, hipStream_t st) {
    const TLGeom g = tl_geom(B, L);
    hipLaunchKernelGGL(tplayer_reduce_kv_kernel, dim3(cdiv((long)B * S * 64, 256)), dim3(256), 0, st, kvpart, dK, dV, B, S, g);
    return LAUNCH_CHECK();
}

// ---- parameter gradients: out[i] = sum_g part[g][i], scattered to the parameters' gradient tensors ---------------------------------
struct TLOut { float* dst[4 + TL_NVEC]; };                   // dWq (first 64 rows of in_proj), dWo, dW1, dW2, then the 10 vectors
__global__ __launch_bounds__(256) void tplayer_reduce_params_kernel(const float* __restrict__ part, int G, TLOut o, float betaF) {
    __shared__ float sh[8][32];
    const int t = threadIdx.x, il = t & 31, gl = t >> 5;
    const int i = blockIdx.x * 32 + il;                      // TL_PREC is a multiple of 32
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int g = gl;
    for (; g + 24 < G; g += 32) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = part[(long)(g + 8 * u) * TL_PREC + i];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] += v[u];
    }
    for (; g < G; g += 8) acc[0] += part[(long)g * TL_PREC + i];
    sh[gl][il] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (gl == 0) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += sh[k][il];
        int seg, off;
        if (i < 4 * 4096) { seg = i >> 12; off = i & 4095; }
        else { seg = 4 + ((i - 4 * 4096) >> 6); off = i & 63; }
        float* d = o.dst[seg];
        if (d) d[off] = (seg >= 12 && betaF != 0.f) ? s + betaF * d[off] : s;
    }
}
// Any destination may be null (skipped).  betaF = 1: the final norm's gradients are accumulated (several layers share it).
TATT_API int tatt_tplayer_reduce_params(const float* ppart, int B, int L, float* d_in_w, float* d_in_b, float* d_out_w, float* d_out_b,
                                        float* d_w1, float* d_b1, float* d_w2, float* d_b2, float* d_lnA_w, float* d_lnA_b,
                                        float* d_lnB_w, float* d_lnB_b, float* d_lnF_w, float* d_lnF_b, float betaF, hipStream_t st) {
    const TLGeom g = tl_geom(B, L);
    TLOut o = {{d_in_w, d_out_w, d_w1, d_w2, d_in_b, d_out_b, d_b1, d_b2, d_lnA_w, d_lnA_b, d_lnB_w, d_lnB_b, d_lnF_w, d_lnF_b}};
    hipLaunchKernelGGL(tplayer_reduce_params_kernel, dim3(TL_PREC / 32), dim3(256), 0, st, ppart, g.G, o, betaF);
    return LAUNCH_CHECK();
}
// the same reduction over G records (the second-generation backward, tplayer2.hip, has its own work-group count)
TATT_API int tatt_tplayer_reduce_params_g(const float* ppart, int G, float* d_in_w, float* d_in_b, float* d_out_w, float* d_out_b,
                                          float* d_w1, float* d_b1, float* d_w2, float* d_b2, float* d_lnA_w, float* d_lnA_b,
                                          float* d_lnB_w, float* d_lnB_b, float* d_lnF_w, float* d_lnF_b, float betaF, hipStream_t st) {
    if (G < 1) return 1;
    TLOut o = {{d_in_w, d_out_w, d_w1, d_w2, d_in_b, d_out_b, d_b1, d_b2, d_lnA_w, d_lnA_b, d_lnB_w, d_lnB_b, d_lnF_w, d_lnF_b}};
    hipLaunchKernelGGL(tplayer_reduce_params_kernel, dim3(TL_PREC / 32), dim3(256), 0, st, ppart, G, o, betaF);
    return LAUNCH_CHECK();
}
