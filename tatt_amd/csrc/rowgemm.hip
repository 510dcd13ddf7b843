// Row-streaming fp32 MFMA GEMM for the token matrices of the TATT hot path (gfx950).
//
//   C (M x N) = act(alpha * ([A | A2] (M x K) @ B (K x N) + bias)) + beta * C
//
// with M = B*H*W tokens (tens of thousands) and a SMALL weight matrix (K, N <= 256): every nn.Linear / 1x1 convolution /
// GRU input projection of the path and their input gradients (reference model/tsrn.py:1071, model/transformer_v2.py:453-459,
// 786-795, nn.Linear / nn.GRU input projections).  These GEMMs sit at or below the HBM balance point (K = N = 64:
// 16 FLOP/B), so the kernel is organised around the row stream, not around an output tile grid:
//   * persistent work-groups; each wave keeps its slice of B -- (N/2 columns) x K -- in registers for its whole lifetime
//     (loaded once: global -> LDS transpose -> registers), so B costs no LDS/L2 traffic in the loop;
//   * 64-row tiles of A are prefetched global -> registers with whole-row 16-byte loads one tile ahead, published to a
//     double-buffered LDS tile (pitch K+4: the 16-byte MFMA operand reads are bank-conflict free), one barrier per tile;
//   * per tile a wave issues K/8 ds_read_b128 and (K/2)*(N/64) v_mfma_f32_32x32x2_f32; C is written straight from the
//     accumulators (128-byte row segments).
// HBM traffic = A once + C once (+ C once more when beta != 0): the algorithmic minimum.
#include "common.h"

struct RowGemmP {
    const float* A; const float* A2; const float* B; const float* bias; float* C;
    long lda, lda2, sbk, sbn, ldc;
    int M, K1, act;
    float alpha, beta;
};

template <int KG, int NCB, int MINB>
__global__ __launch_bounds__(256, MINB) void rowgemm_kernel(RowGemmP p) {
    constexpr int K = KG * 8, N = NCB * 64, XP = K + 4, BP = N + 8;
    constexpr int Q4 = K / 4;              // float4 per row
    constexpr int QT = K / 16;             // float4 per thread per 64-row tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int ntile = (p.M + 63) >> 6;
    int tile = blockIdx.x;
    if (tile >= ntile) return;

    f32x4 pa[QT];
    auto fetch = [&](int tl) {
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const int idx = t + 256 * q;
            const int row = idx / Q4, c = (idx - row * Q4) * 4;
            const long gi = (long)tl * 64 + row;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (gi < p.M)
                v = (c < p.K1) ? *reinterpret_cast<const f32x4*>(p.A + gi * p.lda + c)
                               : *reinterpret_cast<const f32x4*>(p.A2 + gi * p.lda2 + (c - p.K1));
            pa[q] = v;
        }
    };
    auto publish = [&](float* Xs) {
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const int idx = t + 256 * q;
            const int row = idx / Q4, c = (idx - row * Q4) * 4;
            *reinterpret_cast<f32x4*>(Xs + row * XP + c) = pa[q];
        }
    };
    fetch(tile);                                           // first A tile in flight while the weights are staged

    // ---- B: global (any strides, coalesced along its contiguous axis) -> LDS [k][j] -> registers in MFMA B-operand order ----
    if (p.sbn == 1) {
        for (int idx = t; idx < K * N; idx += 256) {
            const int r = idx / N, j = idx - r * N;
            smem[r * BP + j] = p.B[(long)r * p.sbk + j];
        }
    } else {
        for (int idx = t; idx < K * N; idx += 256) {
            const int j = idx / K, r = idx - j * K;
            smem[r * BP + j] = p.B[(long)r * p.sbk + (long)j * p.sbn];
        }
    }
    __syncthreads();
    float w[NCB][KG][4];                                   // w[cb][g][u] = B(k = 8g + 4 (lane >> 5) + u, col block cb, j = lane & 31)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int g = 0; g < KG; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                w[cb][g][u] = smem[(8 * g + 4 * (lane >> 5) + u) * BP + (wn * NCB + cb) * 32 + (lane & 31)];
    float bj[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) bj[cb] = p.bias ? p.bias[(wn * NCB + cb) * 32 + (lane & 31)] : 0.f;
    __syncthreads();                                       // every wave has its weights: the staging area becomes the A tiles
    publish(smem);
    __syncthreads();

    int buf = 0;
    while (true) {
        const int next = tile + gridDim.x;
        const bool has_next = next < ntile;
        if (has_next) fetch(next);
        const float* Xs = smem + buf * (64 * XP) + (wm * 32 + (lane & 31)) * XP + 4 * (lane >> 5);
        f32x16 acc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const f32x4 va = *reinterpret_cast<const f32x4*>(Xs + 8 * g);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[u], w[cb][g][u], acc[cb], 0, 0, 0);
        }
        // ---- epilogue: C/D layout  col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) ----
        const long row0 = (long)tile * 64 + wm * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int col = (wn * NCB + cb) * 32 + (lane & 31);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const long row = row0 + (reg & 3) + 8 * (reg >> 2);
                if (row < p.M) {
                    float v = apply_act(p.alpha * (acc[cb][reg] + bj[cb]), p.act);
                    float* dst = p.C + row * p.ldc + col;
                    if (p.beta != 0.f) v += p.beta * *dst;
                    *dst = v;
                }
            }
        }
        if (!has_next) break;
        publish(smem + (buf ^ 1) * (64 * XP));
        __syncthreads();
        tile = next;
        buf ^= 1;
    }
}

template <int KG, int NCB>
static int launch_rowgemm(const RowGemmP& p, hipStream_t st) {
    constexpr int K = KG * 8, N = NCB * 64;
    constexpr int MINB = (NCB * KG * 4 <= 96) ? 2 : 1;
    constexpr int LDS = 4 * ((K * (N + 8) > 2 * 64 * (K + 4)) ? K * (N + 8) : 2 * 64 * (K + 4));
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm_kernel<KG, NCB, MINB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const int ntile = (p.M + 63) / 64;
    int per_cu = 160 * 1024 / LDS;
    if (per_cu > MINB) per_cu = MINB;
    if (per_cu < 1) per_cu = 1;
    const int G = ntile < 256 * per_cu ? ntile : 256 * per_cu;
    hipLaunchKernelGGL((rowgemm_kernel<KG, NCB, MINB>), dim3(G), dim3(256), LDS, st, p);
    return LAUNCH_CHECK();
}

// C (M x N, row pitch ldc) = act(alpha * ([A | A2] @ B + bias)) + beta * C.
//   A (M x K1, row pitch lda) and the optional A2 (M x (K - K1), row pitch lda2) are row-major with unit column stride,
//   16-byte aligned, pitches and K1 multiples of 4;  B(r, j) = B[r*sbk + j*sbn] (either stride may be the unit one);
//   K in {64,128,192,256}, N in {64,128,192,256}, K*N <= 36864.  Returns 2 (nothing launched) for unsupported shapes --
//   the caller falls back to tatt_gemm.
TATT_API int tatt_rowgemm(const float* A, long lda, const float* A2, long lda2, int K1, const float* B, long sbk, long sbn,
                          const float* bias, float* C, long ldc, int M, int N, int K, float alpha, float beta, int act,
                          hipStream_t st) {
    if (M <= 0 || K % 64 || N % 64 || K > 256 || N > 256 || K * N > 36864) return 2;
    if ((lda & 3) || (A2 && ((lda2 & 3) || (K1 & 3))) || ((uintptr_t)A & 15) || ((uintptr_t)A2 & 15)) return 2;
    if (!A2) K1 = K;
    RowGemmP p = {A, A2, B, bias, C, lda, lda2, sbk, sbn, ldc, M, K1, act, alpha, beta};
    const int kg = K / 8, ncb = N / 64;
#define RG_CASE(KGV, NCBV) if (kg == KGV && ncb == NCBV) return launch_rowgemm<KGV, NCBV>(p, st);
    RG_CASE(8, 1) RG_CASE(8, 2) RG_CASE(8, 3) RG_CASE(8, 4)
    RG_CASE(16, 1) RG_CASE(16, 2) RG_CASE(16, 3) RG_CASE(16, 4)
    RG_CASE(24, 1) RG_CASE(24, 2) RG_CASE(24, 3)
    RG_CASE(32, 1) RG_CASE(32, 2)
#undef RG_CASE
    return 2;
}
