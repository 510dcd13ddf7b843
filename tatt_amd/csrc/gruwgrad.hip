// Weight gradients of one GruBlock in ONE pass over the tokens, on the bf16 matrix cores by operand splitting.
//   dW' (192 x K)  = dgi^T [x | xb]      (composed input projection: 1x1 conv x W_ih, reference model/tsrn.py:1075-1084)
//   dW_hh (192 x 64) = dgh^T hprev         (both directions: the two diagonal 96 x 32 blocks are the nn.GRU gradients)
//   db' (192) = column sums of dgi,  db_hh (192) = column sums of dgh
// M = 49,152 tokens at B = 48.  The three fp32-MFMA GEMMs this replaces (gemm_fast, 3 launches per block, 30 per TATT step) were
// matrix-core-bound (3.6 GFLOP per block on the fp32 pipe) and read dgi twice.  Here every token row (dgi 192 | dgh 192 | x 64 |
// xb 64 | hprev 64 = 576 floats) is read once, and the products run as hi hi + hi lo + lo hi on v_mfma_f32_16x16x32_bf16 (a = hi +
// lo, hi = bf16(a), lo = bf16(a - hi): 2^-16 relative, fp32 accumulation -- the arithmetic of tatt_conv3_c64_fwd_sb / tatt_tokgemm_sb;
// profiles/r03_split_bf16_probe.txt).  The contraction runs over TOKENS, so an MFMA operand needs 8 consecutive tokens of one
// channel per lane: a chunk (32 tokens x 576 floats) is staged token-major in LDS exactly as it lies in memory (coalesced 16-byte
// loads, one chunk ahead in registers); then the 8 waves share the transposition: each gathers 4-5 of the 36 column tiles with 8
// ds_read_b32 down a column, splits them in registers and leaves the hi / lo fragments in LDS in MFMA order (every element is
// converted ONCE; a fragment is then one conflict-free ds_read_b128 for whichever wave needs it).  Row pitch of the fp32 image 578
// dwords: the two token octets of a 32-lane read group are 8 x 578 = 16 (mod 32) banks apart -- conflict-free.  One MFMA contracts
// the 32 tokens of a chunk; two barriers per chunk.
// Work-group = 8 waves, persistent over chunks g, g + G, ..: waves 0-5 own dW' rows 32 w .. 32 w + 31 (2 row tiles x 8 column tiles
// + the ones column = 18 accumulator tiles), waves 6-7 own dW_hh rows 96 (w - 6) .. + 95 (6 x (4 + 1) = 30 tiles); either kind
// gathers and splits 20 fragments per chunk, which is what the waves spend their time on.  The bias gradients ride along as a
// constant all-ones B fragment.  Per-group partial results go to the split-K workspace layout of
// tatt_splitk_reduce ([G][192][N] slabs, then [G][192] row sums), which sums them deterministically.
#include "common.h"
#include <mutex>

typedef __bf16 gw_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gw_f32x2 __attribute__((ext_vector_type(2)));

#define GW_TOK 32
#define GW_PITCH 578
#define GW_IMG (GW_TOK * GW_PITCH)               // floats of the fp32 chunk image (73,984 B)
#define GW_FRAGS (36 * 2 * 64 * 4)               // floats of the fragment buffer: 36 column tiles x {hi, lo} x 64 lanes x 16 B (73,728 B)
#define GW_LDS ((GW_IMG + GW_FRAGS) * 4)         // 147,712 bytes
#define GW_THREADS 512

struct GruWgP {
    const float* dgi; const float* dgh;           // (M, 192) each
    const float* x; const float* xb;              // (M, 64); xb may be null (K = 64)
    const float* hprev;                           // (M, 64)
    float* p1; float* p2;                         // workspaces: G*192*K + G*192 and G*192*64 + G*192 floats
    int nchunks, K;
};

// 8 consecutive tokens (rows 8 kq .. 8 kq + 7 of the chunk) of one column -> the hi and lo bf16 fragments of an MFMA operand
__device__ __forceinline__ void gw_frag(const float* __restrict__ col, gw_bf16x8& hi, gw_bf16x8& lo) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = col[e * GW_PITCH];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const gw_f32x2 a = (gw_f32x2){v[e], v[e + 1]};
        const gw_bf16x2 h = __builtin_convertvector(a, gw_bf16x2);
        const gw_bf16x2 l = __builtin_convertvector(a - __builtin_convertvector(h, gw_f32x2), gw_bf16x2);
        hi[e] = h[0]; hi[e + 1] = h[1];
        lo[e] = l[0]; lo[e + 1] = l[1];
    }
}
__device__ __forceinline__ f32x4 gw_mma3(const gw_bf16x8& ah, const gw_bf16x8& al, const gw_bf16x8& bh, const gw_bf16x8& bl, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
}
__device__ __forceinline__ gw_bf16x8 gw_ones() {
    gw_bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)1.0f;
    return o;
}
// fragment of column tile `tile` (hl = 0: hi, 1: lo) for this lane
__device__ __forceinline__ gw_bf16x8 gw_ld(const float* __restrict__ F, int tile, int hl, int lane) {
    return __builtin_bit_cast(gw_bf16x8, *reinterpret_cast<const f32x4*>(F + ((tile * 2 + hl) * 64 + lane) * 4));
}
// dW' waves: row tiles a0, a0 + 1 (kept) x NT column tiles from tile 24 on (streamed) + the ones column.  acc[m * (NT + 1) + n]
template <int NT>
__device__ __forceinline__ void gw_compute_gi(const float* __restrict__ F, int a0, int lane, f32x4* __restrict__ acc) {
    const gw_bf16x8 ones = gw_ones();
    gw_bf16x8 ah[2], al[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        ah[m] = gw_ld(F, a0 + m, 0, lane);
        al[m] = gw_ld(F, a0 + m, 1, lane);
        acc[m * (NT + 1) + NT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], ones, acc[m * (NT + 1) + NT], 0, 0, 0);
        acc[m * (NT + 1) + NT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], ones, acc[m * (NT + 1) + NT], 0, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const gw_bf16x8 bh = gw_ld(F, 24 + n, 0, lane), bl = gw_ld(F, 24 + n, 1, lane);
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m * (NT + 1) + n] = gw_mma3(ah[m], al[m], bh, bl, acc[m * (NT + 1) + n]);
    }
}
// dW_hh waves: the 4 column tiles of hprev (tiles 32..35, kept) x row tiles a0 .. a0 + 5 (streamed) + the ones column.  acc[m * 5 + n]
__device__ __forceinline__ void gw_compute_gh(const float* __restrict__ F, int a0, int lane, f32x4* __restrict__ acc) {
    const gw_bf16x8 ones = gw_ones();
    gw_bf16x8 bh[4], bl[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) { bh[n] = gw_ld(F, 32 + n, 0, lane); bl[n] = gw_ld(F, 32 + n, 1, lane); }
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        const gw_bf16x8 ah = gw_ld(F, a0 + m, 0, lane), al = gw_ld(F, a0 + m, 1, lane);
        acc[m * 5 + 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, ones, acc[m * 5 + 4], 0, 0, 0);
        acc[m * 5 + 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, ones, acc[m * 5 + 4], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m * 5 + n] = gw_mma3(ah, al, bh[n], bl[n], acc[m * 5 + n]);
    }
}
// C layout: row = 4 (lane >> 4) + r, column = lane & 15.  slab: (192 x N) partial of this group, sums: its 192 row sums
template <int MT, int NT>
__device__ __forceinline__ void gw_store(const f32x4* __restrict__ acc, float* __restrict__ slab, float* __restrict__ sums, int N,
                                         int row0, int li, int kq) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 16 * m + 4 * kq + r;
#pragma unroll
            for (int n = 0; n < NT; ++n) slab[(long)row * N + 16 * n + li] = acc[m * (NT + 1) + n][r];
            if (li == 0) sums[row] = acc[m * (NT + 1) + NT][r];
        }
    }
}

template <bool HAS_XB>
__global__ __launch_bounds__(GW_THREADS) void gru_wgrad_sb_kernel(GruWgP p) {
    extern __shared__ __attribute__((aligned(16))) float gw_T[];
    constexpr int NX = HAS_XB ? 8 : 4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
    const int G = gridDim.x, g = blockIdx.x;
    f32x4 pre[9];
    // chunk staging: three sets of 32 rows x 48 16-byte vectors (dgi | dgh | x, xb, hprev), 3 vectors of each per thread
    auto fetch = [&](int chunk) {
        const long tok0 = (long)chunk * GW_TOK;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int f = t + GW_THREADS * i, row = f / 48, c4 = f - row * 48;
            const long tok = tok0 + row;
            pre[i] = *reinterpret_cast<const f32x4*>(p.dgi + tok * 192 + 4 * c4);
            pre[3 + i] = *reinterpret_cast<const f32x4*>(p.dgh + tok * 192 + 4 * c4);
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (c4 < 16) v = *reinterpret_cast<const f32x4*>(p.x + tok * 64 + 4 * c4);
            else if (c4 < 32) { if (HAS_XB) v = *reinterpret_cast<const f32x4*>(p.xb + tok * 64 + 4 * (c4 - 16)); }
            else v = *reinterpret_cast<const f32x4*>(p.hprev + tok * 64 + 4 * (c4 - 32));
            pre[6 + i] = v;
        }
    };
    auto stash = [&](float* T) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int f = t + GW_THREADS * i, row = f / 48, c4 = f - row * 48;
            float* d = T + row * GW_PITCH + 4 * c4;               // rows are 8-byte aligned (578 dwords): two 8-byte stores
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const f32x4 v = pre[3 * s + i];
                *reinterpret_cast<float2*>(d + 192 * s) = make_float2(v[0], v[1]);
                *reinterpret_cast<float2*>(d + 192 * s + 2) = make_float2(v[2], v[3]);
            }
        }
    };
    f32x4 acc[30];                                                // waves 0-5: [2][NX + 1], waves 6-7: [6][5]
#pragma unroll
    for (int q = 0; q < 30; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float* const T = gw_T;
    float* const F = gw_T + GW_IMG;
    if (g < p.nchunks) fetch(g);
    for (int chunk = g; chunk < p.nchunks; chunk += G) {
        stash(T);                                                 // T: last read by the conversions of the previous chunk (before barrier 2)
        __syncthreads();                                          // 1: image complete; every wave has left the previous chunk's MFMAs (F is free)
        if (chunk + G < p.nchunks) fetch(chunk + G);
        for (int tile = wave; tile < 36; tile += 8) {             // column tiles 0-11 dgi, 12-23 dgh, 24-27 x, 28-31 xb, 32-35 hprev
            if (!HAS_XB && tile >= 28 && tile < 32) continue;
            gw_bf16x8 hi, lo;
            gw_frag(T + 8 * kq * GW_PITCH + 16 * tile + li, hi, lo);
            *reinterpret_cast<f32x4*>(F + ((tile * 2 + 0) * 64 + lane) * 4) = __builtin_bit_cast(f32x4, hi);
            *reinterpret_cast<f32x4*>(F + ((tile * 2 + 1) * 64 + lane) * 4) = __builtin_bit_cast(f32x4, lo);
        }
        __syncthreads();                                          // 2: fragments complete; T may be overwritten
        if (wave < 6) gw_compute_gi<NX>(F, 2 * wave, lane, acc);
        else gw_compute_gh(F, 12 + 6 * (wave - 6), lane, acc);
    }
    if (wave < 6)
        gw_store<2, NX>(acc, p.p1 + (long)g * 192 * p.K, p.p1 + (long)G * 192 * p.K + g * 192, p.K, 32 * wave, li, kq);
    else
        gw_store<6, 4>(acc, p.p2 + (long)g * 192 * 64, p.p2 + (long)G * 192 * 64 + g * 192, 64, 96 * (wave - 6), li, kq);
}

// dgi, dgh (M, 192); x (M, 64); xb (M, 64) or null; hprev (M, 64) -- all contiguous; M % 32 == 0.
// G = number of persistent work-groups = partial slabs (1 .. min(M / 32, 256); fewer groups = less reduction traffic and fewer
// CUs taken from the kernels running beside it); ws1 >= G*192*K + G*192 floats (K = 128 with xb, 64 without), ws2 >= G*192*64 + G*192.
// Leaves the per-group partials; the caller sums them with tatt_splitk_reduce(ws1, dWp, 192, K, G, 0, 0, 0, dbp, 192) and
// tatt_splitk_reduce(ws2, dWhh, 192, 64, G, 0, 0, 0, dbhh, 192) (batched with the stage's other reductions).
TATT_API int tatt_gru_wgrad_sb(const float* dgi, const float* dgh, const float* x, const float* xb, const float* hprev,
                               float* ws1, float* ws2, int M, int G, hipStream_t st) {
    if (M <= 0 || M % GW_TOK) return 1;
    const int nchunks = M / GW_TOK;
    if (G < 1 || G > nchunks || G > 256) return 2;
    static TattPerDevice attr_once;
    tatt_per_device(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gru_wgrad_sb_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, GW_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gru_wgrad_sb_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, GW_LDS);
    });
    GruWgP p = {dgi, dgh, x, xb, hprev, ws1, ws2, nchunks, xb ? 128 : 64};
    if (xb) hipLaunchKernelGGL(gru_wgrad_sb_kernel<true>, dim3(G), dim3(GW_THREADS), GW_LDS, st, p);
    else hipLaunchKernelGGL(gru_wgrad_sb_kernel<false>, dim3(G), dim3(GW_THREADS), GW_LDS, st, p);
    return LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// Round 4: the same weight gradients from the OPERAND FRAGMENTS tatt_gru32_bwd2 leaves behind (gru.hip).
// ------------------------------------------------------------------------------------------------
// The recurrence's lanes own one channel over consecutive time steps, which is the "8 consecutive k per lane" of an MFMA operand
// when the contraction runs over tokens -- so the gate-gradient side (dgi, dgh's n rows, h_{t-1}: 1.25 KB per token, already split
// into bf16 hi / lo) arrives from HBM as ready operands: one 16-byte load per lane per (tile, half), no LDS, no conversion.  Only
// the token-major activations x | xb (4 or 8 column tiles instead of 36) still pass through LDS to be transposed and split.
// K-step c = 4 octets = 32 tokens: octet o = 4 c + kq = seq * T/8 + window, tokens seq_base(seq) + (8 window + e) * stride_t.
// Work-group = 6 waves: wave w = (direction d = w / 3, gate = w % 3) owns rows 32 w .. 32 w + 31 of dW' (2 row tiles x K/16 column
// tiles + the bias column) and of dW_hh (2 x 2 tiles against the h_{t-1} fragments of direction d; its A operand is dgi's r / z
// fragment again, or the gn fragment for the n gate).  LDS: 32-token image of x | xb + its fragments = 17 / 33 KB (was 148 KB), so
// the kernel shares a CU with the main lane's kernels.  Partials: ws1 [G][192][K] + [G][192] (as tatt_gru_wgrad_sb), ws2
// [G][192][32] + [G][192]: dW_hh COMPACT -- rows 96 d .. 96 d + 95 are direction d's 96 x 32 matrix.
#define GF_THREADS 384
struct GruWfP {
    const float* frag; const float* x; const float* xb;
    float* p1; float* p2;
    int nK, T8, s_in;
    long stride_hi, stride_lo, stride_t;
};
template <int PITCH>
__device__ __forceinline__ void gf_frag(const float* __restrict__ col, gw_bf16x8& hi, gw_bf16x8& lo) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = col[e * PITCH];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const gw_f32x2 a = (gw_f32x2){v[e], v[e + 1]};
        const gw_bf16x2 h = __builtin_convertvector(a, gw_bf16x2);
        const gw_bf16x2 l = __builtin_convertvector(a - __builtin_convertvector(h, gw_f32x2), gw_bf16x2);
        hi[e] = h[0]; hi[e + 1] = h[1];
        lo[e] = l[0]; lo[e + 1] = l[1];
    }
}
template <bool HAS_XB>
__global__ __launch_bounds__(GF_THREADS) void gru_wgrad_frag_kernel(GruWfP p) {
    constexpr int K = HAS_XB ? 128 : 64, NT = K / 16, PITCH = K + 2, NV = 32 * K / 4, VPT = (NV + GF_THREADS - 1) / GF_THREADS;
    __shared__ __attribute__((aligned(16))) float gf_T[32 * PITCH];            // token-major image of the chunk's x | xb rows
    __shared__ __attribute__((aligned(16))) float gf_F[NT * 2 * 64 * 4];       // their fragments: [tile][hi, lo][lane][4]
    const int t = threadIdx.x, lane = t & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int G = gridDim.x, g = blockIdx.x;
    const int d = wave / 3, gate = wave - 3 * d;
    const int a_slot = d * 8 + gate * 2;                           // dgi rows of (d, gate): the A operand of dW'
    const int ah_slot = gate == 2 ? d * 8 + 6 : a_slot;            // dgh rows of (d, gate): dgi's for r / z, the gn fragments for n
    const int bh_slot = 16 + d * 2;                                // h_{t-1} of direction d
    f32x4 xpre[VPT], apre[12];
    auto fetch = [&](int c) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int f = t + GF_THREADS * i;
            if (NV % GF_THREADS == 0 || f < NV) {
                const int row = f / (K / 4), c4 = f - row * (K / 4);
                const int o = 4 * c + (row >> 3), s = o / p.T8, w = o - s * p.T8;
                const long tok = (long)(s / p.s_in) * p.stride_hi + (long)(s % p.s_in) * p.stride_lo + (long)(8 * w + (row & 7)) * p.stride_t;
                xpre[i] = (!HAS_XB || c4 < 16) ? *reinterpret_cast<const f32x4*>(p.x + tok * 64 + 4 * c4)
                                               : *reinterpret_cast<const f32x4*>(p.xb + tok * 64 + 4 * (c4 - 16));
            }
        }
        const float* base = p.frag + (long)c * (20 * 2 * 256) + lane * 4;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) {
                apre[m * 2 + hl] = *reinterpret_cast<const f32x4*>(base + ((a_slot + m) * 2 + hl) * 256);
                apre[4 + m * 2 + hl] = *reinterpret_cast<const f32x4*>(base + ((ah_slot + m) * 2 + hl) * 256);
                apre[8 + m * 2 + hl] = *reinterpret_cast<const f32x4*>(base + ((bh_slot + m) * 2 + hl) * 256);
            }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int f = t + GF_THREADS * i;
            if (NV % GF_THREADS == 0 || f < NV) {
                const int row = f / (K / 4), c4 = f - row * (K / 4);
                float* dst = gf_T + row * PITCH + 4 * c4;          // rows are 8-byte aligned (PITCH = 2 mod 4): two 8-byte stores
                *reinterpret_cast<float2*>(dst) = make_float2(xpre[i][0], xpre[i][1]);
                *reinterpret_cast<float2*>(dst + 2) = make_float2(xpre[i][2], xpre[i][3]);
            }
        }
    };
    f32x4 acc1[2 * (NT + 1)], acc2[2 * 3];                         // dW' [m][n | bias], dW_hh [m][n | bias]
#pragma unroll
    for (int q = 0; q < 2 * (NT + 1); ++q) acc1[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 6; ++q) acc2[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const gw_bf16x8 ones = gw_ones();
    if (g < p.nK) fetch(g);
    for (int c = g; c < p.nK; c += G) {
        stash();                                                   // gf_T: last read by the conversions of the previous chunk (before barrier 2)
        __syncthreads();                                           // 1: image complete; every wave has left the previous chunk's MFMAs (gf_F is free)
        gw_bf16x8 a[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) a[q] = __builtin_bit_cast(gw_bf16x8, apre[q]);
        if (c + G < p.nK) fetch(c + G);
        for (int tile = wave; tile < NT; tile += GF_THREADS / 64) {
            gw_bf16x8 hi, lo;
            gf_frag<PITCH>(gf_T + 8 * kq * PITCH + 16 * tile + li, hi, lo);
            *reinterpret_cast<f32x4*>(gf_F + ((tile * 2 + 0) * 64 + lane) * 4) = __builtin_bit_cast(f32x4, hi);
            *reinterpret_cast<f32x4*>(gf_F + ((tile * 2 + 1) * 64 + lane) * 4) = __builtin_bit_cast(f32x4, lo);
        }
        __syncthreads();                                           // 2: fragments complete; gf_T may be overwritten
        // MFMA order: a dependent v_mfma on the same accumulator waits for the previous one's passes, so the three products of a tile
        // (hi hi, hi lo, lo hi) are issued product-major over FOUR accumulators (2 row tiles x 2 column tiles) -- three independent
        // MFMAs between dependent ones (PMC of the first version: 39 % of the wave cycles were MFMA issue stalls)
        auto quad = [&](f32x4& c00, f32x4& c01, f32x4& c10, f32x4& c11, const gw_bf16x8& a0h, const gw_bf16x8& a0l, const gw_bf16x8& a1h,
                        const gw_bf16x8& a1l, const gw_bf16x8& b0h, const gw_bf16x8& b0l, const gw_bf16x8& b1h, const gw_bf16x8& b1l) {
            c00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, b0h, c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, b1h, c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, b0h, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, b1h, c11, 0, 0, 0);
            c00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, b0l, c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, b1l, c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, b0l, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, b1l, c11, 0, 0, 0);
            c00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0l, b0h, c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0l, b1h, c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1l, b0h, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1l, b1h, c11, 0, 0, 0);
        };
        // bias columns (row sums through an all-ones B operand): four independent accumulators, hi then lo
        acc1[NT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], ones, acc1[NT], 0, 0, 0);
        acc1[(NT + 1) + NT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], ones, acc1[(NT + 1) + NT], 0, 0, 0);
        acc2[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[4], ones, acc2[2], 0, 0, 0);
        acc2[5] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[6], ones, acc2[5], 0, 0, 0);
        acc1[NT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], ones, acc1[NT], 0, 0, 0);
        acc1[(NT + 1) + NT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[3], ones, acc1[(NT + 1) + NT], 0, 0, 0);
        acc2[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[5], ones, acc2[2], 0, 0, 0);
        acc2[5] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[7], ones, acc2[5], 0, 0, 0);
        quad(acc2[0], acc2[1], acc2[3], acc2[4], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11]);        // dW_hh: 2 x 2 tiles
#pragma unroll
        for (int n = 0; n < NT; n += 2) {
            const gw_bf16x8 b0h = gw_ld(gf_F, n, 0, lane), b0l = gw_ld(gf_F, n, 1, lane);
            const gw_bf16x8 b1h = gw_ld(gf_F, n + 1, 0, lane), b1l = gw_ld(gf_F, n + 1, 1, lane);
            quad(acc1[n], acc1[n + 1], acc1[(NT + 1) + n], acc1[(NT + 1) + n + 1], a[0], a[1], a[2], a[3], b0h, b0l, b1h, b1l);
        }
    }
    gw_store<2, NT>(acc1, p.p1 + (long)g * 192 * K, p.p1 + (long)G * 192 * K + g * 192, K, 32 * wave, li, kq);
    gw_store<2, 2>(acc2, p.p2 + (long)g * 192 * 32, p.p2 + (long)G * 192 * 32 + g * 192, 32, 32 * wave, li, kq);
}

// frag: the fragment stream of tatt_gru32_bwd2 (nK = nseq * T / 32 K-steps of 10240 floats) for the sequence geometry given;
// x (M, 64), xb (M, 64) or null, contiguous.  1 <= G <= min(nK, 256).  ws1 >= G*192*K + G*192 floats (K = 128 with xb, 64 without),
// ws2 >= G*192*32 + G*192.  Finish with tatt_splitk_reduce(ws1, dWp, 192, K, G, 0, 0, 0, dbp, 192) and
// tatt_splitk_reduce(ws2, dWhh_c, 192, 32, G, 0, 0, 0, dbhh, 192): dWhh_c (192, 32) = [dW_hh forward; dW_hh reverse].
TATT_API int tatt_gru_wgrad_frag(const float* frag, const float* x, const float* xb, float* ws1, float* ws2, int nseq, int T,
                                 int s_in, long stride_hi, long stride_lo, long stride_t, int G, hipStream_t st) {
    if (nseq <= 0 || T <= 0 || T % 8 || ((long)nseq * (T / 8)) % 4 || s_in <= 0) return 1;
    const int nK = (int)((long)nseq * (T / 8) / 4);
    if (G < 1 || G > nK || G > 256) return 2;
    GruWfP p = {frag, x, xb, ws1, ws2, nK, T / 8, s_in, stride_hi, stride_lo, stride_t};
    if (xb) hipLaunchKernelGGL(gru_wgrad_frag_kernel<true>, dim3(G), dim3(GF_THREADS), 0, st, p);
    else hipLaunchKernelGGL(gru_wgrad_frag_kernel<false>, dim3(G), dim3(GF_THREADS), 0, st, p);
    return LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// Round 5: the recurrent weight gradient of the QUERY GRU (nn.GRU(64 -> 512, bidirectional) over the batch axis, reference
// model/tsrn.py:248-262 through InfoGen's query embedding): dW_hh (1536 x 512) = dgh^T h_prev, contraction over the B * 64 = 3072
// (time step, sequence) tokens, and db_hh = column sums of dgh -- per direction 4.8 GFLOP that ran as an fp32-pipe GEMM (gemm_fast,
// 70 us each, the longest kernels of the last backward pass's side lane).  Same arithmetic as the kernels above (split bf16, three
// products, fp32 accumulation); a different shape: a LARGE output and a SHORT contraction, so the output is tiled (128 x 128 per
// work-group, 64 x 64 per wave = 16 accumulator tiles) and the contraction split S ways over grid.y; both directions in one launch
// (grid.z).  Per chunk of 32 tokens the 128 dgh columns and the 128 h_prev columns of the tile are staged token-major (pitch 130:
// conflict-free column gathers as above), each wave transposes + splits 4 of the 16 column tiles, then runs 48 MFMAs on them.
// Partials: [S][N][K] slabs then [S][N] row sums per direction -- the layout tatt_splitk_reduce sums.
// ------------------------------------------------------------------------------------------------
#define QW_PITCH 130
#define QW_IMG (GW_TOK * QW_PITCH)                // floats of one operand's chunk image
#define QW_LDS ((2 * QW_IMG + 16 * 2 * 64 * 4) * 4)   // 33,280 + 32,768 = 66,048 bytes: two work-groups per CU
struct QgruWgP {
    const float* A[2]; const float* Bm[2]; float* part[2];
    int N, K, nchunks, per, S;
};
__global__ __launch_bounds__(256) void qgru_wgrad_sb_kernel(QgruWgP p) {
    extern __shared__ __attribute__((aligned(16))) float qw_lds[];
    float* const T = qw_lds;
    float* const F = qw_lds + 2 * QW_IMG;
    const int t = threadIdx.x, lane = t & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), wr = wave >> 1, wc = wave & 1;
    const int d = blockIdx.z, s = blockIdx.y, tiles_k = p.K >> 7;
    const int tn = blockIdx.x / tiles_k, tk = blockIdx.x - tn * tiles_k;
    const float* __restrict__ A = p.A[d] + 128 * tn;
    const float* __restrict__ B = p.Bm[d] + 128 * tk;
    const int c0 = s * p.per, c1 = min(c0 + p.per, p.nchunks);
    const bool sums = tk == 0 && wc == 0;                          // wave-uniform: these waves also produce the row sums
    f32x4 pa[4], pb[4];
    auto fetch = [&](int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = t + 256 * i, row = f >> 5, c4 = f & 31;
            const long tok = (long)c * GW_TOK + row;
            pa[i] = *reinterpret_cast<const f32x4*>(A + tok * p.N + 4 * c4);
            pb[i] = *reinterpret_cast<const f32x4*>(B + tok * p.K + 4 * c4);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = t + 256 * i, row = f >> 5, c4 = f & 31;
            float* da = T + row * QW_PITCH + 4 * c4;               // rows are 8-byte aligned (130 dwords): two 8-byte stores
            *reinterpret_cast<float2*>(da) = make_float2(pa[i][0], pa[i][1]);
            *reinterpret_cast<float2*>(da + 2) = make_float2(pa[i][2], pa[i][3]);
            *reinterpret_cast<float2*>(da + QW_IMG) = make_float2(pb[i][0], pb[i][1]);
            *reinterpret_cast<float2*>(da + QW_IMG + 2) = make_float2(pb[i][2], pb[i][3]);
        }
    };
    f32x4 acc[16], accs[4];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) accs[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const gw_bf16x8 ones = gw_ones();
    if (c0 < c1) fetch(c0);
    for (int c = c0; c < c1; ++c) {
        stash();                                                   // T: last read by the conversions of the previous chunk (before barrier 2)
        __syncthreads();                                           // 1: images complete; every wave has left the previous chunk's MFMAs
        if (c + 1 < c1) fetch(c + 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {                              // column tiles 0-7: dgh, 8-15: h_prev
            const int tile = wave + 4 * q;
            gw_bf16x8 hi, lo;
            gf_frag<QW_PITCH>(T + (tile >> 3) * QW_IMG + 8 * kq * QW_PITCH + 16 * (tile & 7) + li, hi, lo);
            *reinterpret_cast<f32x4*>(F + ((tile * 2 + 0) * 64 + lane) * 4) = __builtin_bit_cast(f32x4, hi);
            *reinterpret_cast<f32x4*>(F + ((tile * 2 + 1) * 64 + lane) * 4) = __builtin_bit_cast(f32x4, lo);
        }
        __syncthreads();                                           // 2: fragments complete; T may be overwritten
        gw_bf16x8 bh[4], bl[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) { bh[n] = gw_ld(F, 8 + 4 * wc + n, 0, lane); bl[n] = gw_ld(F, 8 + 4 * wc + n, 1, lane); }
#pragma unroll
        for (int m = 0; m < 4; m += 2) {
            const gw_bf16x8 a0h = gw_ld(F, 4 * wr + m, 0, lane), a0l = gw_ld(F, 4 * wr + m, 1, lane);
            const gw_bf16x8 a1h = gw_ld(F, 4 * wr + m + 1, 0, lane), a1l = gw_ld(F, 4 * wr + m + 1, 1, lane);
            if (sums) {
                accs[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, ones, accs[m], 0, 0, 0);
                accs[m + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, ones, accs[m + 1], 0, 0, 0);
                accs[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0l, ones, accs[m], 0, 0, 0);
                accs[m + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1l, ones, accs[m + 1], 0, 0, 0);
            }
#pragma unroll
            for (int n = 0; n < 4; n += 2) {                       // product-major over four accumulators (see the kernel above)
                f32x4 &c00 = acc[m * 4 + n], &c01 = acc[m * 4 + n + 1], &c10 = acc[(m + 1) * 4 + n], &c11 = acc[(m + 1) * 4 + n + 1];
                c00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, bh[n], c00, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, bh[n + 1], c01, 0, 0, 0);
                c10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, bh[n], c10, 0, 0, 0);
                c11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, bh[n + 1], c11, 0, 0, 0);
                c00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, bl[n], c00, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, bl[n + 1], c01, 0, 0, 0);
                c10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, bl[n], c10, 0, 0, 0);
                c11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, bl[n + 1], c11, 0, 0, 0);
                c00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0l, bh[n], c00, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0l, bh[n + 1], c01, 0, 0, 0);
                c10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1l, bh[n], c10, 0, 0, 0);
                c11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1l, bh[n + 1], c11, 0, 0, 0);
            }
        }
    }
    // C layout: row = 4 kq + r, column = li
    float* __restrict__ slab = p.part[d] + (long)s * p.N * p.K;
    float* __restrict__ rs = p.part[d] + (long)p.S * p.N * p.K + (long)s * p.N;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 128 * tn + 64 * wr + 16 * m + 4 * kq + r;
#pragma unroll
            for (int n = 0; n < 4; ++n) slab[(long)row * p.K + 128 * tk + 64 * wc + 16 * n + li] = acc[m * 4 + n][r];
            if (sums && li == 0) rs[row] = accs[m][r];
        }
}

// dW_d (N x K) = A_d^T B_d and db_d (N) = column sums of A_d for d = 0, 1 in one launch: A_d (M, N), B_d (M, K) contiguous,
// M % 32 == 0, N % 128 == 0, K % 128 == 0, 1 <= S <= M / 32; ws_d >= S*N*K + S*N floats.  Leaves the per-split partials: finish each
// direction with tatt_splitk_reduce(ws_d, dW_d, N, K, S, 0, 0, 0, db_d, N) (batched with the stage's other reductions).
TATT_API int tatt_qgru_wgrad_sb(const float* A0, const float* A1, const float* B0, const float* B1, float* ws0, float* ws1, int M, int N,
                                int K, int S, hipStream_t st) {
    if (M <= 0 || M % GW_TOK || N <= 0 || (N & 127) || K <= 0 || (K & 127)) return 1;
    const int nchunks = M / GW_TOK;
    if (S < 1 || S > nchunks) return 2;
    const int per = (nchunks + S - 1) / S;
    if ((long)(S - 1) * per >= nchunks) return 2;                  // (every split gets at least one chunk)
    static TattPerDevice attr_once;
    tatt_per_device(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(qgru_wgrad_sb_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, QW_LDS);
    });
    QgruWgP p = {{A0, A1}, {B0, B1}, {ws0, ws1}, N, K, nchunks, per, S};
    hipLaunchKernelGGL(qgru_wgrad_sb_kernel, dim3((N >> 7) * (K >> 7), S, 2), dim3(256), QW_LDS, st, p);
    return LAUNCH_CHECK();
}
