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

// ---- weight gradient ---------------------------------------------------------------------------------------------
// thread (ci = t & 15, tap lane tl = t >> 4): taps tl, tl+16, ..., 4 output channels each -> 6 x 4 accumulators per 16-channel
// chunk, kept in registers over all the tiles a (persistent) block walks.  Halo tile channel-contiguous: Xc[row][col][16].
#define W9_TAPS 6
__global__ __launch_bounds__(256) void conv9_c64_c4_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 float* __restrict__ part, int B, int H, int W) {
    __shared__ float Xc[T9_HH][T9_WW][T9_CK];
    __shared__ f32x4 Dy[T9_H][T9_W];
    const int tiles_w = W / T9_W, tiles_h = H / T9_H;
    const int ntiles = B * tiles_h * tiles_w;
    const int ci = threadIdx.x & 15, tl = threadIdx.x >> 4;
    float acc[4][W9_TAPS][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < W9_TAPS; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][b][c] = 0.f;
    int kh[W9_TAPS], kw[W9_TAPS];
#pragma unroll
    for (int j = 0; j < W9_TAPS; ++j) {
        int tap = tl + 16 * j;
        if (tap > 80) tap = 80;          // clamped lanes recompute tap 80; their result is discarded at the end
        kh[j] = tap / 9; kw[j] = tap - 9 * (tap / 9);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int bid = tile;
        const int tw = bid % tiles_w; bid /= tiles_w;
        const int th = bid % tiles_h; const int n = bid / tiles_h;
        const int h0 = th * T9_H, w0 = tw * T9_W;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            __syncthreads();
            for (int i = threadIdx.x; i < T9_HH * T9_WW * 4; i += 256) {
                const int c4 = i & 3, p = i >> 2;
                const int r = p / T9_WW, cc = p - r * T9_WW;
                const int hh = h0 + r - 4, ww = w0 + cc - 4;
                f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (hh >= 0 && hh < H && ww >= 0 && ww < W)
                    v = *reinterpret_cast<const f32x4*>(x + (((long)n * H + hh) * W + ww) * 64 + ch * T9_CK + 4 * c4);
                *reinterpret_cast<f32x4*>(&Xc[r][cc][4 * c4]) = v;
            }
            if (ch == 0) {
                const int r = threadIdx.x >> 5, cc = threadIdx.x & 31;
                Dy[r][cc] = *reinterpret_cast<const f32x4*>(dy + (((long)n * H + h0 + r) * W + w0 + cc) * 4);
            }
            __syncthreads();
            for (int r = 0; r < T9_H; ++r)
                for (int cc = 0; cc < T9_W; ++cc) {
                    const f32x4 g = Dy[r][cc];
#pragma unroll
                    for (int j = 0; j < W9_TAPS; ++j) {
                        const float xv = Xc[r + kh[j]][cc + kw[j]][ci];
                        acc[ch][j][0] = fmaf(xv, g[0], acc[ch][j][0]); acc[ch][j][1] = fmaf(xv, g[1], acc[ch][j][1]);
                        acc[ch][j][2] = fmaf(xv, g[2], acc[ch][j][2]); acc[ch][j][3] = fmaf(xv, g[3], acc[ch][j][3]);
                    }
                }
        }
    }
    // partial[block][(tap*64 + c)][4]
    float* P = part + (long)blockIdx.x * 81 * 64 * 4;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
#pragma unroll
        for (int j = 0; j < W9_TAPS; ++j) {
            const int tap = tl + 16 * j;
            if (tap <= 80)
                *reinterpret_cast<f32x4*>(P + ((long)tap * 64 + ch * T9_CK + ci) * 4) =
                    (f32x4){acc[ch][j][0], acc[ch][j][1], acc[ch][j][2], acc[ch][j][3]};
        }
}
__global__ void conv9_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int G) {
    // dw[co][ci][tap] (OIHW, Cout=4, Cin=64) = sum_g part[g][tap*64+ci][co]
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // enumerates (tap*64+ci)*4 + co
    if (idx >= 81 * 64 * 4) return;
    float s0 = 0.f, s1 = 0.f;
    int g = 0;
    for (; g + 2 <= G; g += 2) { s0 += part[(long)g * 20736 + idx]; s1 += part[(long)(g + 1) * 20736 + idx]; }
    if (g < G) s0 += part[(long)g * 20736 + idx];
    const int co = idx & 3, r = idx >> 2, ci = r & 63, tap = r >> 6;
    dw[((long)co * 64 + ci) * 81 + tap] = s0 + s1;
}
// x (B,H,W,64), dy (B,H,W,4) -> dw (4,64,9,9); part >= nblocks*81*64*4 floats, nblocks = min(#tiles, 256)
TATT_API int tatt_conv9_c64_c4_wgrad(const float* x, const float* dy, float* dw, float* part, int B, int H, int W,
                                     hipStream_t st) {
    if (H % T9_H || W % T9_W) return 1;
    int ntiles = B * (H / T9_H) * (W / T9_W);
    int G = ntiles < 256 ? ntiles : 256;
    hipLaunchKernelGGL(conv9_c64_c4_wgrad_kernel, dim3(G), dim3(256), 0, st, x, dy, part, B, H, W);
    hipLaunchKernelGGL(conv9_wgrad_reduce_kernel, dim3(cdiv(81 * 64 * 4, 256)), dim3(256), 0, st, part, dw, G);
    return LAUNCH_CHECK();
}
