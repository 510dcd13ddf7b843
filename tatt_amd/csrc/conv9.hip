// 9x9 convolutions with 4 channels on one side (the RGB+mask image end of the network), fp32 VALU.
//
//   tatt_conv9_c64_to_c4:      y[px][4]  = sum_{tap,ci<64} x[px+tap][ci] * w[tap][ci][4]   (+bias)
//       = the final reconstruction conv (reference model/tsrn.py:623, 64->4 at HR resolution) and, with the flipped
//         filter, the data-gradient of block1 (4->64, model/tsrn.py:597).
//   tatt_conv9_c64_c4_wgrad:   dw[co<4][ci<64][tap] = sum_px x[px+tap][ci] * dy[px][co]
//
// With only 4 output channels the MFMA tile (32x32) would be 8x padded, so these stay on the vector ALU at the
// same 157 TFLOP/s fp32 peak: one thread per output pixel holds its 4 accumulators, a (8+8)x(32+8) halo tile of 16
// input channels is staged in LDS channel-quad-major ([c4][row][col] float4: lanes read consecutive 16 B), and the
// filter -- uniform across the wave -- is read through the scalar cache (s_load) straight into SGPR operands.
#include "common.h"
#include <mutex>

#define T9_H 8
#define T9_W 32
#define T9_HH (T9_H + 8)
#define T9_WW (T9_W + 8)
#define T9_CK 16

// stage channels [c0, c0+16) of the halo tile; Xs[c4][row][col] as float4
__device__ __forceinline__ void stage_halo_q(const float* __restrict__ x, f32x4 (*Xs)[T9_HH][T9_WW], int n, int h0, int w0,
                                             int H, int W, int C, int c0) {
    for (int i = threadIdx.x; i < T9_HH * T9_WW * 4; i += 256) {
        const int c4 = i & 3, p = i >> 2;
        const int r = p / T9_WW, cc = p - r * T9_WW;
        const int hh = h0 + r - 4, ww = w0 + cc - 4;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (hh >= 0 && hh < H && ww >= 0 && ww < W)
            v = *reinterpret_cast<const f32x4*>(x + (((long)n * H + hh) * W + ww) * C + c0 + 4 * c4);
        Xs[c4][r][cc] = v;
    }
}

__global__ __launch_bounds__(256) void conv9_c64_to_c4_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              int B, int H, int W, int C) {
    __shared__ f32x4 Xs[4][T9_HH][T9_WW];
    const int tiles_w = W / T9_W, tiles_h = H / T9_H;
    int bid = blockIdx.x;
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h; const int n = bid / tiles_h;
    const int h0 = th * T9_H, w0 = tw * T9_W;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int c0 = 0; c0 < C; c0 += T9_CK) {
        __syncthreads();
        stage_halo_q(x, Xs, n, h0, w0, H, W, C, c0);
        __syncthreads();
        for (int kh = 0; kh < 9; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 9; ++kw) {
                const float* wt = wp + ((long)(kh * 9 + kw) * C + c0) * 4;      // uniform -> scalar loads
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const f32x4 v = Xs[c4][ty + kh][tx + kw];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float* wv = wt + (c4 * 4 + u) * 4;
                        a0 = fmaf(v[u], wv[0], a0); a1 = fmaf(v[u], wv[1], a1);
                        a2 = fmaf(v[u], wv[2], a2); a3 = fmaf(v[u], wv[3], a3);
                    }
                }
            }
        }
    }
    if (bias) { a0 += bias[0]; a1 += bias[1]; a2 += bias[2]; a3 += bias[3]; }
    const long o = (((long)n * H + h0 + ty) * W + w0 + tx) * 4;
    *reinterpret_cast<f32x4*>(y + o) = (f32x4){a0, a1, a2, a3};
}
// x (B,H,W,C) NHWC contiguous, C % 16 == 0, H % 8 == 0, W % 32 == 0; wp = [81][C][4]; y (B,H,W,4)
TATT_API int tatt_conv9_c64_to_c4(const float* x, const float* wpacked, const float* bias, float* y, int B, int H, int W,
                                  int C, hipStream_t st) {
    if (C % T9_CK || H % T9_H || W % T9_W) return 1;
    hipLaunchKernelGGL(conv9_c64_to_c4_kernel, dim3(B * (H / T9_H) * (W / T9_W)), dim3(256), 0, st, x, wpacked, bias, y, B,
                       H, W, C);
    return LAUNCH_CHECK();
}

// ---- the same convolution on the matrix cores ---------------------------------------------------------------------------------
// Four output channels do not fill an MFMA tile -- four output channels of FOUR NEIGHBOURING PIXELS do: the 16 columns of a
// v_mfma_f32_16x16x4_f32 tile are n = (j, o) = (pixel offset 0..3, output channel 0..3), its 16 rows are 16 groups of 4 pixels
// (64 pixels of one image row), and the contraction runs over (ky, dx, ci) with dx = kx + j in [0, 12): the filter becomes a
// Toeplitz-expanded matrix Wt[ky][ci][n][dx] = w[o][ci][ky][dx - j] (zero outside the 9 taps) -- 12/9 of the useful FLOPs instead
// of 4x, i.e. 75 % of the fp32 matrix peak is the ceiling (the vector-ALU kernel above reaches 24 %).
//   work-group = 8 rows x 64 pixels, 4 waves x 2 rows; input channels in chunks of 16:
//   Xs[16 halo rows][16 ci][96]  (pixel-contiguous): lane (i, kq) reads pixels 4 g(i) + 4 d .. + 3 of channel 4 cq + kq with one
//       ds_read_b128 and feeds MFMA u (dx = 4 d + u) with element u.  Channel rows are 96 floats = 24 slots of 16 bytes apart
//       (= 8 mod 16) and matrix row i stands for pixel group g(i) = i ^ 4 for i < 8, i otherwise: with that the four lane groups a
//       ds_read_b128 is served in hit 16 distinct slots each.
//   Ws[16 ci][16 n][20]  (dx-contiguous, one (ky, chunk) slab of the expanded filter, 20 KB, re-staged per ky): pitch 20 floats =
//       5 slots per n, channel rows 80 slots (= 0 mod 16) apart -- conflict-free as well.
#define M9_TH 8
#define M9_TW 64
#define M9_PW 96
#define M9_ROWS (M9_TH + 8)
#define M9_XS (M9_ROWS * 16 * M9_PW)          // floats: 24,576
#define M9_WS (16 * 16 * 20)                  // floats: 5,120
#define M9_LDS ((M9_XS + 2 * M9_WS) * 4)      // 139,264 B
__global__ __launch_bounds__(256) void conv9_c64_to_c4_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                                   const float* __restrict__ bias, float* __restrict__ y,
                                                                   int B, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float smem9[];
    float* Xs = smem9;
    float* Ws = smem9 + M9_XS;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tiles_w = W / M9_TW, tiles_h = H / M9_TH;
    int bid = blockIdx.x;
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h; const int n = bid / tiles_h;
    const int h0 = th * M9_TH, w0 = tw * M9_TW;
    const int i = lane & 15, kq = lane >> 4;
    const int g = (i & 8) ? i : (i ^ 4);                      // pixel group of matrix row i
    f32x4 acc[2];
    acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* xa = Xs + (2 * wave * 16 + kq) * M9_PW + 4 * g;      // + ((ky + r2) * 16 + 4 cq) * PW + 4 d
    const float* wb = Ws + (kq * 16 + i) * 20;                         // + (4 cq * 16) * 20 + 4 d
    // 36 phases (4 channel chunks x 9 filter rows): the NEXT phase's filter slab travels global -> registers under the MFMAs of the
    // current one and is stored to the other Ws buffer afterwards (one barrier per phase); the halo is re-staged per chunk.
    f32x4 wpre[5];
    auto slab_load = [&](int p) {
        const f32x4* src = reinterpret_cast<const f32x4*>(wt + ((long)(p % 9) * 64 + 16 * (p / 9)) * 320);
#pragma unroll
        for (int q = 0; q < 5; ++q) wpre[q] = src[t + 256 * q];
    };
    auto slab_store = [&](int buf) {
        f32x4* dst = reinterpret_cast<f32x4*>(Ws + buf * M9_WS);
#pragma unroll
        for (int q = 0; q < 5; ++q) dst[t + 256 * q] = wpre[q];
    };
    auto halo_stage = [&](int c0) {
        // halo rows h0-4 .. h0+11, pixels w0-4 .. w0+67, channels c0 .. c0+15 (quads of lanes read 64 contiguous bytes)
        for (int idx = t; idx < M9_ROWS * 72 * 4; idx += 256) {
            const int c4 = idx & 3, pp = idx >> 2;
            const int r = pp / 72, px = pp - r * 72;
            const int hh = h0 + r - 4, ww = w0 + px - 4;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (hh >= 0 && hh < H && ww >= 0 && ww < W)
                v = *reinterpret_cast<const f32x4*>(x + (((long)n * H + hh) * W + ww) * 64 + c0 + 4 * c4);
            float* d = Xs + (r * 16 + 4 * c4) * M9_PW + px;
            d[0] = v[0]; d[M9_PW] = v[1]; d[2 * M9_PW] = v[2]; d[3 * M9_PW] = v[3];
        }
    };
    slab_load(0);
    halo_stage(0);
    slab_store(0);
    __syncthreads();
#pragma unroll 1
    for (int p = 0; p < 36; ++p) {
        const int ky = p % 9;
        if (p + 1 < 36) slab_load(p + 1);
        const float* wbp = wb + (p & 1) * M9_WS;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(wbp + cq * 4 * 320 + 4 * d);
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(xa + ((ky + 0) * 16 + 4 * cq) * M9_PW + 4 * d);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(xa + ((ky + 1) * 16 + 4 * cq) * M9_PW + 4 * d);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], bv[u], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], bv[u], acc[1], 0, 0, 0);
                }
            }
        if (p + 1 < 36) {
            slab_store((p + 1) & 1);                          // the other buffer: last read in phase p - 1, before the previous barrier
            if (ky == 8) {                                    // chunk boundary: every wave must have left the halo first
                __syncthreads();
                halo_stage(16 * ((p + 1) / 9));
            }
            __syncthreads();
        }
    }
    // C layout: column n = lane & 15 = (j, o); row 4 (lane >> 4) + reg -> pixel group g(row).  The 16 columns of a row are 16
    // consecutive floats of y: pixel 4 g + j, channel o.
    const float bo = bias ? bias[i & 3] : 0.f;
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 4 * kq + reg;
            const int gg = (row & 8) ? row : (row ^ 4);
            y[(((long)n * H + h0 + 2 * wave + r2) * W + w0 + 4 * gg) * 4 + i] = acc[r2][reg] + bo;
        }
}
// x (B,H,W,64) NHWC contiguous, H % 8 == 0, W % 64 == 0; wt = Toeplitz-expanded filter [9][64][16][20] from
// tatt_repack_conv_weight mode 8 (forward filter of a 64->4 convolution) / mode 9 (data gradient of a 4->64 one); y (B,H,W,4)
TATT_API int tatt_conv9_c64_to_c4_mfma(const float* x, const float* wt, const float* bias, float* y, int B, int H, int W,
                                       hipStream_t st) {
    if (H % M9_TH || W % M9_TW) return 1;
    static std::once_flag attr_once;                 // C++11 call_once: safe if several host threads launch
    std::call_once(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv9_c64_to_c4_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, M9_LDS);
    });
    hipLaunchKernelGGL(conv9_c64_to_c4_mfma_kernel, dim3(B * (H / M9_TH) * (W / M9_TW)), dim3(256), M9_LDS, st, x, wt, bias, y, B, H, W);
    return LAUNCH_CHECK();
}

// ---- weight gradient on the matrix cores -------------------------------------------------------------------------
// dW[ky][kx][ci][co] = sum_{b,r,p} X[b][r][p][ci] * dY[b][r-ky+4][p-kx+4][co]  (r, p = position of the INPUT pixel).
// As a GEMM per input row: A[m = ci][k = p] = X[r][p][ci] (64 rows), B[k = p][n = tap*4 + co] = dY[r-ky+4][p-kx+4][co]: the
// Toeplitz expansion sits on the small operand (4 channels) and is never materialised -- an MFMA B fragment is one LDS dword per
// lane and every lane may take it from its own (row, pixel, channel) address.  n = 81 taps x 4 channels = 324 -> 21 column
// tiles of 16 (96 % useful), 4 row tiles of 16 input channels: 84 accumulator tiles (v_mfma_f32_16x16x4_f32).
// Work-group = 12 waves: wave (mt = w & 3, ng = w >> 2) owns row tile mt and column tiles 7*ng .. 7*ng+6 -> 28 accumulator
// registers that stay live over every tile the (persistent) group walks; per k-step of 4 pixels a wave reads 1 A and 7 B dwords
// and issues 7 MFMAs.  Tile = 4 input rows x 64 pixels of one image: the dY window (12 rows x 72 pixels x 4 channels, zeros
// outside the image) is staged once per tile, the X rows (64 pixels x 64 channels, LDS pitch 80 dwords so that the two pixels
// of an A read's lane group fall in different bank halves) are double-buffered through registers one row ahead.
#define W9_R 4
#define W9_P 64
#define W9_XP 80
#define W9_DW (W9_P + 8)
#define W9_DROWS (W9_R + 8)
#define W9_NT 21
#define W9_N (W9_NT * 16)        // 336 columns per partial row (324 used)
struct W9P { const float* x; const float* dy; float* part; int B, H, W, ntiles; };
__global__ __launch_bounds__(768) void conv9_c64_c4_wgrad_mfma_kernel(W9P p) {
    __shared__ __attribute__((aligned(16))) float Xs[2][W9_P * W9_XP];
    __shared__ __attribute__((aligned(16))) float Ds[(W9_DROWS + 1) * W9_DW * 4];     // + one row of zeros for the 12 unused columns
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int mt = wave & 3, ng = wave >> 2, li = lane & 15, lk = lane >> 4;
    const int tiles_w = p.W / W9_P, tiles_h = p.H / W9_R;
    int bbase[7], bstep[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int n = 16 * (7 * ng + j) + li, tap = n >> 2, co = n & 3;
        if (tap < 81) {
            const int ky = tap / 9, kx = tap - 9 * ky;
            bbase[j] = ((8 - ky) * W9_DW + lk + 8 - kx) * 4 + co;
            bstep[j] = W9_DW * 4;
        } else { bbase[j] = W9_DROWS * W9_DW * 4; bstep[j] = 0; }
    }
    for (int e = t; e < W9_DW * 4; e += 768) Ds[W9_DROWS * W9_DW * 4 + e] = 0.f;
    f32x4 acc[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int my_tiles = p.ntiles > (int)blockIdx.x ? (p.ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int nsteps = my_tiles * W9_R;
    // staging registers: X row of step s+1 (1024 float4 over 768 threads), dY window of the next tile (864 float4)
    f32x4 xr0, xr1 = (f32x4){0.f, 0.f, 0.f, 0.f}, dr0, dr1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto tile_of = [&](int s, int& b, int& r0, int& p0) {
        int tile = blockIdx.x + (s / W9_R) * gridDim.x;
        const int tw = tile % tiles_w; tile /= tiles_w;
        r0 = (tile % tiles_h) * W9_R; b = tile / tiles_h; p0 = tw * W9_P;
    };
    auto load_x = [&](int s) {
        int b, r0, p0; tile_of(s, b, r0, p0);
        const float* src = p.x + (((long)b * p.H + r0 + s % W9_R) * p.W + p0) * 64;
        xr0 = *reinterpret_cast<const f32x4*>(src + 4 * t);
        if (t < 256) xr1 = *reinterpret_cast<const f32x4*>(src + 4 * (t + 768));
    };
    auto store_x = [&](int buf) {
        *reinterpret_cast<f32x4*>(&Xs[buf][(t >> 4) * W9_XP + 4 * (t & 15)]) = xr0;
        if (t < 256) *reinterpret_cast<f32x4*>(&Xs[buf][((t + 768) >> 4) * W9_XP + 4 * (t & 15)]) = xr1;
    };
    auto load_d1 = [&](int e, int b, int r0, int p0) -> f32x4 {
        const int d = e / W9_DW, q = e - d * W9_DW, row = r0 - 4 + d, px = p0 - 4 + q;
        if (row < 0 || row >= p.H || px < 0 || px >= p.W) return (f32x4){0.f, 0.f, 0.f, 0.f};
        return *reinterpret_cast<const f32x4*>(p.dy + (((long)b * p.H + row) * p.W + px) * 4);
    };
    auto load_d = [&](int s) {
        int b, r0, p0; tile_of(s, b, r0, p0);
        dr0 = load_d1(t, b, r0, p0);
        if (t < W9_DROWS * W9_DW - 768) dr1 = load_d1(t + 768, b, r0, p0);
    };
    auto store_d = [&]() {
        *reinterpret_cast<f32x4*>(&Ds[4 * t]) = dr0;
        if (t < W9_DROWS * W9_DW - 768) *reinterpret_cast<f32x4*>(&Ds[4 * (t + 768)]) = dr1;
    };
    if (nsteps > 0) { load_x(0); load_d(0); }
    for (int s = 0; s < nsteps; ++s) {
        const int rr = s % W9_R, buf = s & 1;
        if (rr == 0) __syncthreads();            // the previous tile's last row has been consumed: Ds may change
        store_x(buf);
        if (rr == 0) store_d();
        if (s + 1 < nsteps) { load_x(s + 1); if (rr == W9_R - 1) load_d(s + 1); }
        __syncthreads();
        const float* xa = &Xs[buf][lk * W9_XP + 16 * mt + li];
        const float* db[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) db[j] = &Ds[bbase[j] + rr * bstep[j]];
#pragma unroll 4
        for (int ks = 0; ks < W9_P / 4; ++ks) {
            const float a = xa[ks * 4 * W9_XP];
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, db[j][ks * 16], acc[j], 0, 0, 0);
        }
    }
    // partial[block][ci][n]: C row = 4*(lane>>4)+e -> ci = 16*mt + 4*lk + e, column = lane&15 -> n = 16*(7*ng+j) + li
    float* P = p.part + (long)blockIdx.x * 64 * W9_N;
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) P[(long)(16 * mt + 4 * lk + e) * W9_N + 16 * (7 * ng + j) + li] = acc[j][e];
}
// dw[co][ci][tap] (OIHW, Cout = 4, Cin = 64) = sum_g part[g][ci][tap*4 + co]; block = 64 outputs x 16 lanes over g
__global__ __launch_bounds__(1024) void conv9_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int G) {
    __shared__ float sh[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + tx;                      // enumerates ci*324 + n
    const int ci = idx / 324, n = idx - ci * 324;
    const float* src = part + (long)ci * W9_N + n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = ty;
    for (; g + 48 < G; g += 64) {
        s0 += src[(long)g * 64 * W9_N]; s1 += src[(long)(g + 16) * 64 * W9_N];
        s2 += src[(long)(g + 32) * 64 * W9_N]; s3 += src[(long)(g + 48) * 64 * W9_N];
    }
    for (; g < G; g += 16) s0 += src[(long)g * 64 * W9_N];
    sh[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty == 0) {
        float s = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) s += sh[l][tx];
        dw[((long)(n & 3) * 64 + ci) * 81 + (n >> 2)] = s;
    }
}
// x (B,H,W,64), dy (B,H,W,4) -> dw (4,64,9,9); H % 4 == 0, W % 64 == 0; part >= min(#tiles, 256) * 64 * 336 floats,
// #tiles = B * (H/4) * (W/64)
TATT_API int tatt_conv9_c64_c4_wgrad(const float* x, const float* dy, float* dw, float* part, int B, int H, int W,
                                     hipStream_t st) {
    if (H % W9_R || W % W9_P) return 1;
    W9P p = {x, dy, part, B, H, W, B * (H / W9_R) * (W / W9_P)};
    const int G = p.ntiles < 256 ? p.ntiles : 256;
    hipLaunchKernelGGL(conv9_c64_c4_wgrad_mfma_kernel, dim3(G), dim3(768), 0, st, p);
    hipLaunchKernelGGL(conv9_wgrad_reduce_kernel, dim3(64 * 324 / 64), dim3(1024), 0, st, part, dw, G);
    return LAUNCH_CHECK();
}
