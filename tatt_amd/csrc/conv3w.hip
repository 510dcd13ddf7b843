// Weight gradient of the 3x3 64-channel convolutions on the bf16 matrix cores by operand splitting (tatt_conv3_c64_wgrad_partial_sb).
// Written in round 3, validated on the GPU and made the default in round 4 (ops.CONV3_WGRAD_SB = True): against fp64 at all four
// shapes of the step + the bias gradient (tests/test_kernels_gpu.py::test_conv3_wgrad_split_bf16_vs_fp64, ::test_conv3_wgrad_bias_gradient),
// inside the model-level fp64 yardsticks (tests/test_model_gpu.py), and in tools/emulate_conv3w.py (CPU emulation of the fragment algebra).
//   dW[tap][ci][co] = sum over pixels of x[px + tap - 1][ci] * dy[px][co]        (+ db[co] = sum of dy[px][co])
// Same arithmetic as tatt_conv3_c64_fwd_sb / tatt_gru_wgrad_sb: a = hi + lo (bf16 each), a*b = hi hi + hi lo + lo hi, fp32
// accumulation.  The contraction runs over PIXELS, so an MFMA operand (v_mfma_f32_16x16x32_bf16) needs 8 consecutive pixels of one
// channel per lane: a segment (one image row of 64 pixels: x halo 3 x 66 pixels x 64 ci, dy 64 pixels x 64 co) is staged
// pixel-major in LDS as it lies in memory (fp32, channel pitch 66 dwords: the two pixel octets of a 32-lane read group are 8 x 66 =
// 16 (mod 32) banks apart); the 8 waves share the transposition: per filter row ky each wave gathers ONE 10-pixel window of one
// (channel tile, pixel half) down the columns and emits its three horizontally shifted fragments (kx = 0, 1, 2), split hi / lo, into
// a fragment buffer in MFMA order; the dy fragments are produced once per segment.  Then wave (ct = w & 3, cg = w >> 2) accumulates
// the 9 taps x (channel tile ct) x (output channel tiles 2 cg, 2 cg + 1) = 18 accumulator tiles it owns for the whole launch.
// Output = the per-work-group partials of tatt_conv3_c64_wgrad_partial: part[g][tap*Cin + ci][Cout], pdb[g][Cout].
#include "common.h"
#include <mutex>

typedef __bf16 cw_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float cw_f32x2 __attribute__((ext_vector_type(2)));

#define CW_PX 64
#define CW_HW 66                                   // halo width (pixels)
#define CW_CP 66                                   // channel pitch of a pixel in the fp32 image (dwords)
#define CW_HALO (3 * CW_HW * CW_CP)                // 13,068 floats
#define CW_IMG (CW_HALO + CW_PX * CW_CP)           // + dy tile: 17,292 floats = 69,168 B (a multiple of 16)
#define CW_NFRAG 32                                // 24 x fragments of one filter row (kx, channel tile, pixel half) + 8 dy fragments
#define CW_FRAGS (CW_NFRAG * 2 * 64 * 4)           // floats: {hi, lo} x 64 lanes x 16 B each = 65,536 B
#define CW_LDS ((CW_IMG + CW_FRAGS) * 4)           // 134,704 B
#define CW_F4 ((3 * CW_HW * 64 + CW_PX * 64) / 4)  // 16-byte vectors of one segment in memory: 3168 + 1024 = 4192

struct Conv3WSP {
    const float* x; const float* dy; float* part;
    int B, H, W, Cin, Cout, nseg;
    float* pdb;                                    // per-work-group partial bias gradient [gridDim.x][Cout], or null
};

__device__ __forceinline__ void cw_split(const float* v, cw_bf16x8& hi, cw_bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const cw_f32x2 a = (cw_f32x2){v[e], v[e + 1]};
        const cw_bf16x2 h = __builtin_convertvector(a, cw_bf16x2);
        const cw_bf16x2 l = __builtin_convertvector(a - __builtin_convertvector(h, cw_f32x2), cw_bf16x2);
        hi[e] = h[0]; hi[e + 1] = h[1];
        lo[e] = l[0]; lo[e + 1] = l[1];
    }
}
__device__ __forceinline__ void cw_put(float* F, int f, int lane, const cw_bf16x8& hi, const cw_bf16x8& lo) {
    *reinterpret_cast<f32x4*>(F + ((f * 2 + 0) * 64 + lane) * 4) = __builtin_bit_cast(f32x4, hi);
    *reinterpret_cast<f32x4*>(F + ((f * 2 + 1) * 64 + lane) * 4) = __builtin_bit_cast(f32x4, lo);
}
__device__ __forceinline__ cw_bf16x8 cw_get(const float* F, int f, int hl, int lane) {
    return __builtin_bit_cast(cw_bf16x8, *reinterpret_cast<const f32x4*>(F + ((f * 2 + hl) * 64 + lane) * 4));
}
__device__ __forceinline__ f32x4 cw_mma3(const cw_bf16x8& ah, const cw_bf16x8& al, const cw_bf16x8& bh, const cw_bf16x8& bl, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
}

__global__ __launch_bounds__(512, 1) void conv3_c64_wgrad_sb_kernel(Conv3WSP p) {
    extern __shared__ __attribute__((aligned(16))) float cw_smem[];
    float* const IMG = cw_smem;
    float* const F = cw_smem + CW_IMG;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
    const int ct = wave & 3, cg = wave >> 2;                 // input-channel tile; pair of output-channel tiles 2 cg, 2 cg + 1
    const int cib = blockIdx.y % (p.Cin / 64), cob = blockIdx.y / (p.Cin / 64);
    const int ci0 = cib * 64, co0 = cob * 64;
    const int segs = p.W / CW_PX;
    const bool want_db = p.pdb != nullptr && cib == 0 && ct == 0;
    f32x4 acc[9][2], accdb[2];
#pragma unroll
    for (int a = 0; a < 9; ++a) { acc[a][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[a][1] = acc[a][0]; }
    accdb[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; accdb[1] = accdb[0];
    cw_bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    // A group walks a CONTIGUOUS run of segments, ordered (image, column strip, row) with the row fastest: consecutive segments are
    // consecutive image rows, and two of a segment's three halo rows are already in LDS -- the halo is a ring of three row slots (row hh
    // of the strip lives in slot (hh + 3) % 3) and only the new bottom row is fetched (round 4 strided the segments over the groups and
    // fetched every x row three times: 67.8 MB per launch against 44 MB algorithmic, profiles/r04_g_pmc_traffic_conv3_wgrad_64_64.txt).
    const int per = (p.nseg + (int)gridDim.x - 1) / (int)gridDim.x;
    const int s_beg = blockIdx.x * per, s_end = min(p.nseg, s_beg + per);
    f32x4 pre[9];
    auto seg_of = [&](int s, int& n, int& h, int& w0) { h = s % p.H; s /= p.H; w0 = (s % segs) * CW_PX; n = s / segs; };
    // rows to fetch: all three for the first segment of the run and of a strip (h == 0), else the bottom one
    auto fetch = [&](int s) {
        int n, h, w0;
        seg_of(s, n, h, w0);
        const bool fresh = s == s_beg || h == 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int idx = t + 512 * r;
            f32x4 u = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (idx < 3 * CW_HW * 16) {
                const int c4 = idx & 15, pp = idx >> 4;
                const int rr = pp / CW_HW, px = pp - rr * CW_HW;
                const int hh = h + rr - 1, ww = w0 + px - 1;
                if ((fresh || rr == 2) && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W)
                    u = *reinterpret_cast<const f32x4*>(p.x + (((long)n * p.H + hh) * p.W + ww) * p.Cin + ci0 + 4 * c4);
            } else if (idx < CW_F4) {
                const int j = idx - 3 * CW_HW * 16, c4 = j & 15, px = j >> 4;
                u = *reinterpret_cast<const f32x4*>(p.dy + (((long)n * p.H + h) * p.W + w0 + px) * p.Cout + co0 + 4 * c4);
            }
            pre[r] = u;
        }
    };
    auto stash = [&](int s) {
        int n, h, w0;
        seg_of(s, n, h, w0);
        const bool fresh = s == s_beg || h == 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int idx = t + 512 * r;
            if (idx < CW_F4) {
                float* d;
                if (idx < 3 * CW_HW * 16) {
                    const int pp = idx >> 4, rr = pp / CW_HW, px = pp - rr * CW_HW;
                    if (!(fresh || rr == 2)) continue;                                              // rows h - 1, h: already in their slots
                    d = IMG + (((h + rr + 2) % 3) * CW_HW + px) * CW_CP + 4 * (idx & 15);           // slot of row h + rr - 1
                } else { const int j = idx - 3 * CW_HW * 16; d = IMG + CW_HALO + (j >> 4) * CW_CP + 4 * (j & 15); }
                *reinterpret_cast<float2*>(d) = make_float2(pre[r][0], pre[r][1]);                 // pixel rows are 8-byte aligned (66 dwords)
                *reinterpret_cast<float2*>(d + 2) = make_float2(pre[r][2], pre[r][3]);
            }
        }
    };
    int s = s_beg;
    if (s < s_end) fetch(s);
    for (; s < s_end; ++s) {
        stash(s);                                            // IMG: last read by the conversions of the previous segment (before its barriers)
        __syncthreads();                                     // image complete; every wave has left the previous segment's MFMAs (F is free)
        if (s + 1 < s_end) fetch(s + 1);
        const int hrow = s % p.H;                            // this segment's image row: halo row ky sits in slot (hrow + ky + 2) % 3
        cw_bf16x8 dh[2][2], dl[2][2];                        // this wave's dy fragments [output-channel tile][pixel half]
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            {   // this wave's share of the transposition: the three shifts of (channel tile (wave >> 1) & 3, pixel half wave & 1) of halo row ky
                const int xt = (wave >> 1) & 3, ks = wave & 1;
                const float* col = IMG + ((((hrow + ky + 2) % 3) * CW_HW + 32 * ks + 8 * kq) * CW_CP) + 16 * xt + li;
                float v[10];
#pragma unroll
                for (int e = 0; e < 10; ++e) v[e] = col[e * CW_CP];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    cw_bf16x8 hi, lo;
                    cw_split(v + kx, hi, lo);
                    cw_put(F, (kx * 4 + xt) * 2 + ks, lane, hi, lo);
                }
                if (ky == 0) {                               // dy fragment (output-channel tile wave >> 1, pixel half wave & 1)
                    const float* dcol = IMG + CW_HALO + (32 * ks + 8 * kq) * CW_CP + 16 * (wave >> 1) + li;
                    float w8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) w8[e] = dcol[e * CW_CP];
                    cw_bf16x8 hi, lo;
                    cw_split(w8, hi, lo);
                    cw_put(F, 24 + (wave >> 1) * 2 + ks, lane, hi, lo);
                }
            }
            __syncthreads();                                 // fragments of row ky complete
            if (ky == 0) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        dh[c][ks] = cw_get(F, 24 + (2 * cg + c) * 2 + ks, 0, lane);
                        dl[c][ks] = cw_get(F, 24 + (2 * cg + c) * 2 + ks, 1, lane);
                        if (want_db) {
                            accdb[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, dh[c][ks], accdb[c], 0, 0, 0);
                            accdb[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, dl[c][ks], accdb[c], 0, 0, 0);
                        }
                    }
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const cw_bf16x8 ah = cw_get(F, (kx * 4 + ct) * 2 + ks, 0, lane), al = cw_get(F, (kx * 4 + ct) * 2 + ks, 1, lane);
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[ky * 3 + kx][c] = cw_mma3(ah, al, dh[c][ks], dl[c][ks], acc[ky * 3 + kx][c]);
                }
            __syncthreads();                                 // everyone has read row ky's fragments: the next row may overwrite them
        }
    }
    // C layout: row = 4 (lane >> 4) + r = input channel within the tile, column = lane & 15 = output channel within the tile
    float* P = p.part + (long)blockIdx.x * 9 * p.Cin * p.Cout;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                P[((long)tap * p.Cin + ci0 + 16 * ct + 4 * kq + r) * p.Cout + co0 + 16 * (2 * cg + c) + li] = acc[tap][c][r];
    if (want_db && kq == 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) p.pdb[(long)blockIdx.x * p.Cout + co0 + 16 * (2 * cg + c) + li] = accdb[c][0];
    }
}

// same contract as tatt_conv3_c64_wgrad_partial (conv3.hip): partials part[G][9*Cin][Cout] (+ pdb[G][Cout]) for the split-K reducer
TATT_API int tatt_conv3_c64_wgrad_partial_sb(const float* x, const float* dy, float* part, float* pdb, int B, int H, int W,
                                             int Cin, int Cout, int G, hipStream_t st) {
    if (Cin % 64 || Cout % 64 || W % CW_PX) return 1;
    const int nseg = B * H * (W / CW_PX);
    Conv3WSP p = {x, dy, part, B, H, W, Cin, Cout, nseg, pdb};
    static TattPerDevice attr_once;
    tatt_per_device(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_wgrad_sb_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CW_LDS);
    });
    hipLaunchKernelGGL(conv3_c64_wgrad_sb_kernel, dim3(G, (Cin / 64) * (Cout / 64)), dim3(512), CW_LDS, st, p);
    return LAUNCH_CHECK();
}
