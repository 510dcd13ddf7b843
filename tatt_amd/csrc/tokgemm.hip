// Token-matrix projections on the bf16 matrix cores by operand splitting:  Y (M x N) = [X1 | X2] (M x K) W^T (+ bias),  M = all the
// tokens of a feature map (49,152 at B = 48), N, K in {64, 128, 192}.  These are the GRU input projections of every GruBlock --
// the composed 1x1-conv x W_ih product (reference model/tsrn.py:1075-1084) -- forward (N = 192) and their data gradients
// (dx = dgi W: N = 64 / 128, K = 192): 20 launches per TATT step that ran at 2.2 TB/s on the fp32-MFMA tile kernel (gemm_fast: at
// N = 192 it is matrix-core-bound: 2.4 GFLOP per launch = 15 us at the fp32 peak; HBM needs 12 us).
// Same arithmetic as tatt_conv3_c64_fwd_sb: every fp32 operand a = hi + lo (hi = bf16(a), lo = bf16(a - hi)); a*b is evaluated as
// hi hi + (hi lo + lo hi) with fp32 accumulation (2^-16 relative; profiles/r03_split_bf16_probe.txt).  Organisation: weight-stationary
// -- wave (mh, nq) of a persistent work-group owns 32 tokens x N/4 outputs and keeps its slice of W (hi and lo) in registers; the
// token tile (64 x K) is split into a hi and a lo bf16 image while it is staged into LDS (double-buffered), read back with one
// ds_read_b128 per operand (8 consecutive k).  LDS pitch = 2 K + 32 bytes: row i -> 16-byte slot (2 or 10) i (mod 16), conflict-free
// for the lane groups ds_read_b128 is served in.
#include "common.h"
#include <mutex>

typedef __bf16 tg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 tg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float tg_f32x2 __attribute__((ext_vector_type(2)));

struct TokGemmP {
    const float* X1; const float* X2; int K1;        // sources: columns [0, K1) from X1 (row pitch K1), [K1, K) from X2 (row pitch K - K1)
    const float* W; const float* bias;               // packed split-bf16 operand (tatt_tokgemm_pack), bias (N) or null
    float* Y1; float* Y2; int N1;                    // destinations: columns [0, N1) to Y1 (row pitch N1), [N1, N) to Y2 (row pitch N - N1)
    int M;
    int act, accum;                                  // epilogue: ACT_RELU after the bias; accum: Y += (instead of Y =)
    // epilogue of the fused feed-forward (tatt_tokgemm_sb_ffn): dropout of the result (mask of tatt_dropout for the same seed word, site
    // and flat index row * N + col), or a gate -- the result times gate_scale where gate[row * N + col] > 0, else 0
    float pdrop; const unsigned long long* seed; unsigned site;
    const float* gate; float gate_scale;
    const float* addend;                             // Y = ... + addend (same shape as Y, one destination): a sum that leaves its operand intact
};

__device__ __forceinline__ void tg_split(f32x4 v, uint2& hi, uint2& lo) {
    const tg_bf16x2 h0 = __builtin_convertvector((tg_f32x2){v[0], v[1]}, tg_bf16x2), h1 = __builtin_convertvector((tg_f32x2){v[2], v[3]}, tg_bf16x2);
    const tg_f32x2 r0 = (tg_f32x2){v[0], v[1]} - __builtin_convertvector(h0, tg_f32x2), r1 = (tg_f32x2){v[2], v[3]} - __builtin_convertvector(h1, tg_f32x2);
    const tg_bf16x2 l0 = __builtin_convertvector(r0, tg_bf16x2), l1 = __builtin_convertvector(r1, tg_bf16x2);
    hi = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
    lo = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
}

// NCB: 16-column blocks per wave (N = 64 NCB); KS: k-steps of 32 (K = 32 KS)
// EPI: the dropout / gate epilogues are compiled in (kept out of the GruBlock instantiations, which sit at the register limit)
template <int NCB, int KS, bool EPI>
__global__ __launch_bounds__(512, 1) void tokgemm_sb_kernel(TokGemmP p) {
    constexpr int K = 32 * KS, N = 64 * NCB;
    constexpr int PW = K / 2 + 8;                            // tile pitch in 32-bit words (two bf16 each)
    constexpr int IMG = 64 * PW;                             // words of one image (hi or lo) of one tile
    constexpr int F4 = 64 * K / 4 / 512;                     // 16-byte loads per thread and tile (1, 2 or 3 ... K = 64 -> 2, 128 -> 4, 192 -> 6)
    extern __shared__ __attribute__((aligned(16))) unsigned tg_smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int mh = wave & 1, nq = wave >> 1, am = lane & 15, kq = lane >> 4;
    const int ntiles = p.M / 64;
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    // weight slice of this wave: [cb][ks][hl] 16-byte vectors per lane
    f32x4 wq[NCB * KS * 2];
    {
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(p.W) + (long)(nq * NCB) * KS * 2 * 64 + lane;
#pragma unroll
        for (int i = 0; i < NCB * KS * 2; ++i) wq[i] = wsrc[i * 64];
    }
    float bj[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) bj[cb] = p.bias ? p.bias[16 * (nq * NCB + cb) + am] : 0.f;
    const int K2 = K - p.K1;
    // this thread's share of a tile: F4 vectors; vector v covers row (idx / (K/4)), columns 4 (idx % (K/4)) ..
    auto fetch = [&](int tl, f32x4 (&r)[F4]) {
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            const int idx = t + 512 * i, row = idx / (K / 4), c4 = 4 * (idx - row * (K / 4));
            const long m = (long)tl * 64 + row;
            r[i] = c4 < p.K1 ? *reinterpret_cast<const f32x4*>(p.X1 + m * p.K1 + c4)
                             : *reinterpret_cast<const f32x4*>(p.X2 + m * K2 + (c4 - p.K1));
        }
    };
    auto stash = [&](unsigned* buf, const f32x4 (&r)[F4]) {
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            const int idx = t + 512 * i, row = idx / (K / 4), c4 = 4 * (idx - row * (K / 4));
            uint2 hi, lo;
            tg_split(r[i], hi, lo);
            unsigned* d = buf + row * PW + c4 / 2;
            *reinterpret_cast<uint2*>(d) = hi;
            *reinterpret_cast<uint2*>(d + IMG) = lo;
        }
    };
    f32x4 pre[F4];
    fetch(tile, pre);
    stash(tg_smem, pre);
    __syncthreads();
    int buf = 0;
    while (true) {
        const int next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        if (has_next) fetch(next, pre);
        const unsigned* Xs = tg_smem + buf * 2 * IMG + (32 * mh + am) * PW + 4 * kq;
        // what the epilogue reads back from memory (the gate, the destination when accumulating) is requested BEFORE the products
        // (behind them the loads would queue between the stores: 51 us instead of 17 for the gated 128 x 128 launch); the small
        // instantiations only -- the GruBlock ones have no registers to spare and do not accumulate
        constexpr bool PRE = NCB <= 2 && KS <= 4;
        const long row0 = (long)tile * 64 + 32 * mh + 4 * kq;
        float gv[PRE ? NCB : 1][2][4], yv[PRE ? NCB : 1][2][4];
        if (PRE) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int col = 16 * (nq * NCB + cb) + am;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const long flat = (row0 + 16 * m + r) * p.N1 + col;
                        gv[PRE ? cb : 0][m][r] = (EPI && p.gate) ? p.gate[flat] : 1.f;
                        yv[PRE ? cb : 0][m][r] = p.addend ? p.addend[flat] : ((p.accum && col < p.N1) ? p.Y1[flat] : 0.f);
                    }
            }
        }
        f32x4 accM[2][NCB], accC[2][NCB];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) { accM[m][cb] = (f32x4){0.f, 0.f, 0.f, 0.f}; accC[m][cb] = accM[m][cb]; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            tg_bf16x8 ah[2], al[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ah[m] = __builtin_bit_cast(tg_bf16x8, *reinterpret_cast<const f32x4*>(Xs + 16 * m * PW + 16 * ks));
                al[m] = __builtin_bit_cast(tg_bf16x8, *reinterpret_cast<const f32x4*>(Xs + 16 * m * PW + 16 * ks + IMG));
            }
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const tg_bf16x8 wh = __builtin_bit_cast(tg_bf16x8, wq[(cb * KS + ks) * 2]), wl = __builtin_bit_cast(tg_bf16x8, wq[(cb * KS + ks) * 2 + 1]);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    accM[m][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], wh, accM[m][cb], 0, 0, 0);
                    accC[m][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], wl, accC[m][cb], 0, 0, 0);
                    accC[m][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], wh, accC[m][cb], 0, 0, 0);
                }
            }
        }
        // C layout: row (token) = 4 (lane >> 4) + r, column = lane & 15
        {
            const int N2 = N - p.N1;
            const bool drop = EPI && p.pdrop > 0.f;
            const uint64_t sd = drop ? p.seed[0] : 0ull;
            const uint32_t th = dropout_thresh(p.pdrop);
            const float dsc = drop ? 1.f / (1.f - p.pdrop) : 1.f;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int col = 16 * (nq * NCB + cb) + am;
                float* dst; int ld, c;
                if (col < p.N1) { dst = p.Y1; ld = p.N1; c = col; } else { dst = p.Y2; ld = N2; c = col - p.N1; }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* y = dst + (row0 + 16 * m + r) * ld + c;
                        float v = (accM[m][cb][r] + accC[m][cb][r]) + bj[cb];
                        if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
                        const long flat = (row0 + 16 * m + r) * ld + c;            // (drop / gate / accum: one destination, ld = N)
                        if (drop) v = dropout_keep(sd, p.site, (uint64_t)flat, th) ? v * dsc : 0.f;
                        if (EPI && p.gate) v = (PRE ? gv[PRE ? cb : 0][m][r] : p.gate[flat]) > 0.f ? v * p.gate_scale : 0.f;
                        if (p.accum || (PRE && p.addend)) v += PRE ? yv[PRE ? cb : 0][m][r] : *y;
                        *y = v;
                    }
            }
        }
        if (!has_next) break;
        stash(tg_smem + (buf ^ 1) * 2 * IMG, pre);
        __syncthreads();
        tile = next;
        buf ^= 1;
    }
}

// Packed B operand: 32-bit words of two bf16 (k-consecutive),
//   word[(((blk * KS + ks) * 2 + hl) * 64 + lane) * 4 + e2] = split(w(n = 16 blk + (lane & 15), k = 32 ks + 8 (lane >> 4) + 2 e2 + {0, 1}))
//   trans = 0: w(n, k) = W[n * ldw + k]  (y = x W^T: W (N, K));   trans = 1: w(n, k) = W[k * ldw + n]  (dx = dy W: W (K, N))
__global__ void tokgemm_pack_kernel(const float* __restrict__ W, float* __restrict__ out, int N, int K, int ldw, int trans) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * K) return;
    const int KS = K / 32;
    const int e2 = idx & 3, lane = (idx >> 2) & 63, hl = (idx >> 8) & 1;
    int r = idx >> 9;
    const int ks = r % KS, blk = r / KS;
    const int n = 16 * blk + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + 2 * e2;
    tg_f32x2 v;
#pragma unroll
    for (int u = 0; u < 2; ++u) v[u] = trans ? W[(long)(k + u) * ldw + n] : W[(long)n * ldw + k + u];
    const tg_bf16x2 hi = __builtin_convertvector(v, tg_bf16x2);
    const tg_bf16x2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, tg_f32x2), tg_bf16x2);
    out[idx] = __builtin_bit_cast(float, hl ? lo : hi);
}
// out: N*K 32-bit words (hi and lo images, two bf16 per word).  N a multiple of 64, K of 32.
TATT_API int tatt_tokgemm_pack(const float* W, float* out, int N, int K, int ldw, int trans, hipStream_t st) {
    if (N % 64 || K % 32) return 1;
    hipLaunchKernelGGL(tokgemm_pack_kernel, dim3(cdiv((long)N * K, 256)), dim3(256), 0, st, W, out, N, K, ldw, trans);
    return LAUNCH_CHECK();
}

// several packs in one launch (all GruBlocks of a generator, forward and data-gradient operands, once per training forward)
#define TGP_MAX 24
struct TGPEntry { const float* W; float* out; int N, K, ldw, trans, block0; };
struct TGPTable { TGPEntry e[TGP_MAX]; int n; };
__global__ void tokgemm_pack_batch_kernel(TGPTable t) {
    int k = 0;
    while (k + 1 < t.n && (int)blockIdx.x >= t.e[k + 1].block0) ++k;
    const TGPEntry& e = t.e[k];
    const int idx = ((int)blockIdx.x - e.block0) * blockDim.x + threadIdx.x;
    if (idx >= e.N * e.K) return;
    const int KS = e.K / 32;
    const int e2 = idx & 3, lane = (idx >> 2) & 63, hl = (idx >> 8) & 1;
    const int r = idx >> 9;
    const int ks = r % KS, blk = r / KS;
    const int n = 16 * blk + (lane & 15), kk = 32 * ks + 8 * (lane >> 4) + 2 * e2;
    tg_f32x2 v;
#pragma unroll
    for (int u = 0; u < 2; ++u) v[u] = e.trans ? e.W[(long)(kk + u) * e.ldw + n] : e.W[(long)n * e.ldw + kk + u];
    const tg_bf16x2 hi = __builtin_convertvector(v, tg_bf16x2);
    const tg_bf16x2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, tg_f32x2), tg_bf16x2);
    e.out[idx] = __builtin_bit_cast(float, hl ? lo : hi);
}
// ptrs: HOST array of n x 2 device pointers (W, out); dims: HOST array of n x 4 ints (N, K, ldw, trans)
TATT_API int tatt_tokgemm_pack_batch(const float* const* ptrs, const int* dims, int n, hipStream_t st) {
    for (int base = 0; base < n; base += TGP_MAX) {
        TGPTable t;
        t.n = n - base < TGP_MAX ? n - base : TGP_MAX;
        int blocks = 0;
        for (int k = 0; k < t.n; ++k) {
            const int* d = dims + (long)(base + k) * 4;
            if (d[0] % 64 || d[1] % 32) return 1;
            t.e[k] = {ptrs[2 * (base + k)], const_cast<float*>(ptrs[2 * (base + k) + 1]), d[0], d[1], d[2], d[3], blocks};
            blocks += cdiv((long)d[0] * d[1], 256);
        }
        hipLaunchKernelGGL(tokgemm_pack_batch_kernel, dim3(blocks), dim3(256), 0, st, t);
    }
    return LAUNCH_CHECK();
}

template <int NCB, int KS, bool EPI>
static int tg_launch_epi(const TokGemmP& p, hipStream_t st) {
    constexpr int lds = 2 * 2 * 64 * (32 * KS / 2 + 8) * 4;
    static TattPerDevice once;
    tatt_per_device(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tokgemm_sb_kernel<NCB, KS, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    });
    const int ntiles = p.M / 64;
    hipLaunchKernelGGL((tokgemm_sb_kernel<NCB, KS, EPI>), dim3(ntiles < 256 ? ntiles : 256), dim3(512), lds, st, p);
    return LAUNCH_CHECK();
}
template <int NCB, int KS>
static int tg_launch(const TokGemmP& p, hipStream_t st) {
    if (p.pdrop > 0.f || p.gate) {
        if constexpr (NCB <= 2 && KS <= 4) return tg_launch_epi<NCB, KS, true>(p, st);      // (the feed-forward shapes)
        else return 1;
    }
    return tg_launch_epi<NCB, KS, false>(p, st);
}
static int tg_dispatch(const TokGemmP& p, int N, int K, hipStream_t st);
// Y = [X1 | X2] Wp^T + bias with Wp from tatt_tokgemm_pack; X1 (M, K1), X2 (M, K - K1) (null when K1 == K); the first N1 output columns
// go to Y1 (M, N1), the rest to Y2 (M, N - N1) (null when N1 == N).  M a multiple of 64; (N, K) in {64,128,192} x {64,128,192}.
// _ex: act = ACT_RELU applies max(., 0) after the bias; accum != 0 adds the result to what Y holds (the sum of several data gradients
// into one map: dx = dq Wq + dk Wk + dv Wv of the TBSRN attention projections).
TATT_API int tatt_tokgemm_sb_ex(const float* X1, const float* X2, int K1, const float* Wp, const float* bias, float* Y1, float* Y2, int N1,
                                int M, int N, int K, int act, int accum, hipStream_t st) {
    if (M < 64 || M % 64 || K1 % 4 || (K - K1) % 4 || K1 < 0 || K1 > K || N1 < 0 || N1 > N || (K1 < K && !X2) || (N1 < N && !Y2)) return 1;
    if ((act != ACT_NONE && act != ACT_RELU) || (accum && N1 != N)) return 1;
    TokGemmP p = {X1, X2, K1, Wp, bias, Y1, Y2, N1, M, act, accum, 0.f, nullptr, 0u, nullptr, 1.f, nullptr};
    return tg_dispatch(p, N, K, st);
}
// The two epilogues of a position-wise feed-forward w_2(Dropout(relu(w_1 x))) (reference PositionwiseFeedForward, model/tbsrn.py:154-164)
// that make the dropout and both element-wise backward passes disappear into GEMMs:
//   forward  F = Dropout_p(relu(X W1^T + b1)):     act = ACT_RELU, pdrop > 0 (seed, site: the mask tatt_dropout draws for F's flat index)
//   backward dPre = (dY W2) * [F > 0] / (1 - p):   gate = F, gate_scale = 1 / (1 - p) -- F > 0 exactly where the unit was active AND kept
// One source, one destination (N1 = N).
TATT_API int tatt_tokgemm_sb_ffn(const float* X, const float* Wp, const float* bias, float* Y, int M, int N, int K, int act, float pdrop,
                                 const unsigned long long* seed, unsigned site, const float* gate, float gate_scale, hipStream_t st) {
    if (M < 64 || M % 64 || (act != ACT_NONE && act != ACT_RELU) || pdrop < 0.f || pdrop >= 1.f || (pdrop > 0.f && !seed)) return 1;
    if ((double)M * N >= 4.0e18) return 1;
    TokGemmP p = {X, nullptr, K, Wp, bias, Y, nullptr, N, M, act, 0, pdrop, seed, site, gate, gate_scale, nullptr};
    return tg_dispatch(p, N, K, st);
}
// Y = X Wp^T + bias + addend (addend (M, N) contiguous, left intact; Y may not alias it): the sum of a data gradient and the gradient a
// residual connection carries, without an element-wise launch and without overwriting either.  (N, K) with N <= 128, K <= 128.
TATT_API int tatt_tokgemm_sb_add(const float* X, const float* Wp, const float* bias, const float* addend, float* Y, int M, int N, int K,
                                 hipStream_t st) {
    if (M < 64 || M % 64 || !addend || addend == Y || N > 128 || K > 128) return 1;
    TokGemmP p = {X, nullptr, K, Wp, bias, Y, nullptr, N, M, ACT_NONE, 0, 0.f, nullptr, 0u, nullptr, 1.f, addend};
    return tg_dispatch(p, N, K, st);
}
static int tg_dispatch(const TokGemmP& p, int N, int K, hipStream_t st) {
    if (N == 192 && K == 128) return tg_launch<3, 4>(p, st);
    if (N == 192 && K == 64) return tg_launch<3, 2>(p, st);
    if (N == 128 && K == 192) return tg_launch<2, 6>(p, st);
    if (N == 128 && K == 128) return tg_launch<2, 4>(p, st);
    if (N == 128 && K == 64) return tg_launch<2, 2>(p, st);
    if (N == 64 && K == 192) return tg_launch<1, 6>(p, st);
    if (N == 64 && K == 64) return tg_launch<1, 2>(p, st);
    if (N == 64 && K == 128) return tg_launch<1, 4>(p, st);
    return 1;
}
TATT_API int tatt_tokgemm_sb(const float* X1, const float* X2, int K1, const float* Wp, const float* bias, float* Y1, float* Y2, int N1,
                             int M, int N, int K, hipStream_t st) {
    return tatt_tokgemm_sb_ex(X1, X2, K1, Wp, bias, Y1, Y2, N1, M, N, K, ACT_NONE, 0, st);
}
