// Weight gradient of a token projection y = x W^T + b on the bf16 matrix cores with split operands:
//   dW (N x K) = dY^T X   (contraction over the M tokens),   db (N) = column sums of dY,        N, K in {64, 128}
// The 35 nn.Linear of the TBSRN FeatureEnhancers (reference model/tbsrn.py:77-164; M = 49,152 tokens at B = 48) ran these as fp32-pipe
// GEMMs (gemm_fast, 31-38 us each, 1.3 ms of the backward's side lane).  Same arithmetic as csrc/tokgemm.hip / csrc/gruwgrad.hip
// (a = hi + lo bf16, a b ~ hi hi + hi lo + lo hi, fp32 accumulation); what is new is the operand path: the contraction runs over
// TOKENS, so an MFMA operand needs 8 consecutive tokens of one channel per lane -- csrc/gruwgrad.hip gathers those with 8 ds_read_b32
// down a column of an fp32 image, converts, and goes through LDS a second time.  Here a chunk (32 tokens) is split ONCE while it is
// staged (token-major bf16 hi / lo images, as it lies in memory) and the operands come out of ds_read_b64_tr_b16, the transposing LDS
// read of gfx950 (a lane receives 4 consecutive rows of one column): A = dY^T and B = X are both read that way.  Row pitch 2 C + 64
// bytes (= 64 mod 256): the four rows of a read fall into the four bank quarters.
// Work-group = 4 waves, wave (nh, kh) owns the (N / 2) x (K / 2) quadrant as 32 x 32 blocks of v_mfma_f32_32x32x16_bf16; the tokens are
// split S ways over the grid (per-split partials [S][N][K] then [S][N]: the layout tatt_splitk_reduce sums deterministically).
#include "common.h"

typedef __bf16 tw_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 tw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float tw_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned tw_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned tw_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 tw_bf16x4_t;
typedef __attribute__((address_space(3))) tw_bf16x4_t* tw_lds_b64;

#define TW_TOK 32

struct TokWgP { const float* A; const float* B; float* part; int nchunks, per, S; };

__device__ __forceinline__ void tw_split2(float a, float b, unsigned& hi, unsigned& lo) {
    const tw_bf16x2 h = __builtin_convertvector((tw_f32x2){a, b}, tw_bf16x2);
    const unsigned u = __builtin_bit_cast(unsigned, h);
    const float ra = a - __builtin_bit_cast(float, u << 16), rb = b - __builtin_bit_cast(float, u & 0xffff0000u);
    hi = u;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector((tw_f32x2){ra, rb}, tw_bf16x2));
}
// 8 consecutive tokens of this lane's channel: two transposing reads 4 rows apart
template <int PITCH>
__device__ __forceinline__ tw_u32x4 tw_tr8(const char* p) {
    const tw_bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tw_lds_b64)(p));
    const tw_bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tw_lds_b64)(p + 4 * PITCH));
    const tw_u32x2 ua = __builtin_bit_cast(tw_u32x2, a), ub = __builtin_bit_cast(tw_u32x2, b);
    return (tw_u32x4){ua[0], ua[1], ub[0], ub[1]};
}
__device__ __forceinline__ f32x16 tw_mfma(tw_u32x4 a, tw_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tw_bf16x8, a), __builtin_bit_cast(tw_bf16x8, b), c, 0, 0, 0);
}

template <int N, int K>
__global__ __launch_bounds__(256, 2) void tok_wgrad_sb_kernel(TokWgP p) {
    constexpr int PA = 2 * N + 64, PB = 2 * K + 64;                 // row pitches in bytes
    constexpr int IA = TW_TOK * PA, IB = TW_TOK * PB;               // bytes of one image (hi or lo)
    constexpr int FA = TW_TOK * N / 4 / 256, FB = TW_TOK * K / 4 / 256;   // 16-byte loads per thread and chunk
    constexpr int BN = N / 64, BK = K / 64;                          // 32 x 32 blocks per wave along n and k
    __shared__ __attribute__((aligned(16))) char img[2 * IA + 2 * IB];
    __shared__ float bsum[256 / (N / 4)][N + 4];
    char* const Ah = img; char* const Al = img + IA; char* const Bh = img + 2 * IA; char* const Bl = img + 2 * IA + IB;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), nh = wave >> 1, kh = wave & 1;
    const int lj = lane & 31, kb = lane >> 5, a16 = lane & 15, g16 = (lane >> 4) & 1;
    const int s = blockIdx.x;
    const int c0 = s * p.per, c1 = min(c0 + p.per, p.nchunks);
    f32x4 pa[FA], pb[FB];
    auto fetch = [&](int c) {
#pragma unroll
        for (int i = 0; i < FA; ++i) {
            const int f = t + 256 * i, row = f / (N / 4), c4 = f % (N / 4);
            pa[i] = *reinterpret_cast<const f32x4*>(p.A + ((long)c * TW_TOK + row) * N + 4 * c4);
        }
#pragma unroll
        for (int i = 0; i < FB; ++i) {
            const int f = t + 256 * i, row = f / (K / 4), c4 = f % (K / 4);
            pb[i] = *reinterpret_cast<const f32x4*>(p.B + ((long)c * TW_TOK + row) * K + 4 * c4);
        }
    };
    f32x4 colsum = {0.f, 0.f, 0.f, 0.f};                            // of this thread's 4 channels of dY (c4 = t % (N / 4) for every i)
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < FA; ++i) {
            const int f = t + 256 * i, row = f / (N / 4), c4 = f % (N / 4);
            unsigned h0, l0, h1, l1;
            tw_split2(pa[i][0], pa[i][1], h0, l0); tw_split2(pa[i][2], pa[i][3], h1, l1);
            *reinterpret_cast<uint2*>(Ah + row * PA + 8 * c4) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(Al + row * PA + 8 * c4) = make_uint2(l0, l1);
            colsum += pa[i];
        }
#pragma unroll
        for (int i = 0; i < FB; ++i) {
            const int f = t + 256 * i, row = f / (K / 4), c4 = f % (K / 4);
            unsigned h0, l0, h1, l1;
            tw_split2(pb[i][0], pb[i][1], h0, l0); tw_split2(pb[i][2], pb[i][3], h1, l1);
            *reinterpret_cast<uint2*>(Bh + row * PB + 8 * c4) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(Bl + row * PB + 8 * c4) = make_uint2(l0, l1);
        }
    };
    f32x16 acc[BN][BK];
#pragma unroll
    for (int i = 0; i < BN; ++i)
#pragma unroll
        for (int j = 0; j < BK; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    // this lane's piece of a [4 tokens][16 channels] block: token 8 kb + a16 / 4, channels 16 g16 + 4 (a16 % 4) ..
    const int pieceA = (8 * kb + (a16 >> 2)) * PA + 2 * (N / 2 * nh + 16 * g16 + 4 * (a16 & 3));
    const int pieceB = (8 * kb + (a16 >> 2)) * PB + 2 * (K / 2 * kh + 16 * g16 + 4 * (a16 & 3));
    if (c0 < c1) fetch(c0);
    for (int c = c0; c < c1; ++c) {
        stash();
        __syncthreads();                                            // images complete
        if (c + 1 < c1) fetch(c + 1);
#pragma unroll
        for (int st = 0; st < TW_TOK / 16; ++st) {
            tw_u32x4 ah[BN], al[BN], bh[BK], bl[BK];
#pragma unroll
            for (int i = 0; i < BN; ++i) {
                ah[i] = tw_tr8<PA>(Ah + pieceA + 16 * st * PA + 64 * i);
                al[i] = tw_tr8<PA>(Al + pieceA + 16 * st * PA + 64 * i);
            }
#pragma unroll
            for (int j = 0; j < BK; ++j) {
                bh[j] = tw_tr8<PB>(Bh + pieceB + 16 * st * PB + 64 * j);
                bl[j] = tw_tr8<PB>(Bl + pieceB + 16 * st * PB + 64 * j);
            }
#pragma unroll
            for (int i = 0; i < BN; ++i)
#pragma unroll
                for (int j = 0; j < BK; ++j) {
                    acc[i][j] = tw_mfma(ah[i], bh[j], acc[i][j]);
                    acc[i][j] = tw_mfma(ah[i], bl[j], acc[i][j]);
                    acc[i][j] = tw_mfma(al[i], bh[j], acc[i][j]);
                }
        }
        __syncthreads();                                            // every wave has left the images
    }
    // block (i, j): register 4 g + r = row n0 + 32 i + 8 g + 4 kb + r, column k0 + 32 j + lj
    float* __restrict__ slab = p.part + (long)s * N * K;
#pragma unroll
    for (int i = 0; i < BN; ++i)
#pragma unroll
        for (int j = 0; j < BK; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = N / 2 * nh + 32 * i + 8 * (v >> 2) + 4 * kb + (v & 3);
                slab[(long)row * K + K / 2 * kh + 32 * j + lj] = acc[i][j][v];
            }
    // column sums of dY: thread t summed channels 4 (t % (N / 4)) .. + 3 over the rows it staged
    {
        constexpr int TPC = 256 / (N / 4);                          // threads per channel quad
        const int c4 = t % (N / 4), which = t / (N / 4);
        *reinterpret_cast<f32x4*>(&bsum[which][4 * c4]) = colsum;
        __syncthreads();
        if (t < N) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < TPC; ++w) v += bsum[w][t];
            p.part[(long)p.S * N * K + (long)s * N + t] = v;
        }
    }
}

template <int N, int K>
static int tw_launch(const TokWgP& p, hipStream_t st) {
    hipLaunchKernelGGL((tok_wgrad_sb_kernel<N, K>), dim3(p.S), dim3(256), 0, st, p);
    return LAUNCH_CHECK();
}
// dW (N x K) = A^T B and db (N) = column sums of A: A (M, N), B (M, K) contiguous, M % 32 == 0, N, K in {64, 128}, 1 <= S <= M / 32 with
// every split owning at least one 32-token chunk; ws >= S*N*K + S*N floats.  Leaves the per-split partials: finish with
// tatt_splitk_reduce(ws, dW, N, K, S, 0, 0, 0, db, N) (batched with the stage's other reductions).
TATT_API int tatt_tok_wgrad_sb(const float* A, const float* B, float* ws, int M, int N, int K, int S, hipStream_t st) {
    if (M <= 0 || M % TW_TOK) return 1;
    const int nchunks = M / TW_TOK;
    if (S < 1 || S > nchunks) return 2;
    const int per = (nchunks + S - 1) / S;
    if ((long)(S - 1) * per >= nchunks) return 2;
    TokWgP p = {A, B, ws, nchunks, per, S};
    if (N == 128 && K == 128) return tw_launch<128, 128>(p, st);
    if (N == 64 && K == 128) return tw_launch<64, 128>(p, st);
    if (N == 128 && K == 64) return tw_launch<128, 64>(p, st);
    if (N == 64 && K == 64) return tw_launch<64, 64>(p, st);
    return 1;
}
