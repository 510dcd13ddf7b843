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
//   (persistent) work-group = 4 rows x 64 pixels per tile, 8 waves = 2 per SIMD: wave (row = w & 3, kh = w >> 2) computes one output row over half of
//   the input channels (channel quads kh and kh + 2 of every chunk of 16); the two halves meet through LDS at the end.
//   Xs[12 halo rows][16 ci][96]  (pixel-contiguous): lane (i, kq) reads pixels 4 g(i) + 4 d .. + 3 of channel 4 cq + kq with one
//       ds_read_b128 and feeds MFMA u (dx = 4 d + u) with element u.  Channel rows are 96 floats = 24 slots of 16 bytes apart
//       (= 8 mod 16) and matrix row i stands for pixel group g(i) = i ^ 4 for i < 8, i otherwise: with that the four lane groups a
//       ds_read_b128 is served in hit 16 distinct slots each.
//   The expanded filter never touches LDS: a lane's B fragments come straight from global memory (packed in fragment order: a
//   wave's load reads 1 KB of consecutive memory; the whole filter is 442 KB, L2-resident) into registers THREE PHASES AHEAD of their
//   use (four register sets; a phase = one filter row of one chunk = 24 MFMAs per wave, shorter than a trip to L2 -- and loads
//   retire in order, so a filter fragment also waits for the halo prefetches issued before it, which come from HBM).  Without filter
//   slabs in LDS there is no barrier per phase: the waves only meet at the 3 chunk boundaries, where the next chunk's halo
//   (prefetched to registers under the MFMAs: a thread holds the 16 channels of up to 2 halo pixels) replaces the current one.  The
//   halo stores walk pixels along the lanes (bank = pixel mod 32; channel rows are 0 mod 32 apart, so channel-along-lanes would be
//   a 4-way conflict).
#define M9_TH 4
#define M9_TW 64
#define M9_PW 96
#define M9_ROWS (M9_TH + 8)
#define M9_HPX (M9_ROWS * 72)                 // halo pixels per chunk: 864
#define M9_HQ 2                               // halo pixels per thread (512 threads)
#define M9_D 3                                // filter fragments are fetched M9_D phases ahead (M9_D + 1 divides 36)
#define M9_XS (M9_ROWS * 16 * M9_PW)          // floats: 18,432 = 73,728 B
__global__ __launch_bounds__(512) void conv9_c64_to_c4_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                                   const float* __restrict__ bias, float* __restrict__ y,
                                                                   int B, int H, int W, int ntiles) {
    __shared__ __attribute__((aligned(16))) float Xs[M9_XS];
    __shared__ __attribute__((aligned(16))) float Red[4 * 64 * 4];    // the kh = 1 accumulators of the four rows
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int row = wave & 3, kh = wave >> 2;
    const int tiles_w = W / M9_TW, tiles_h = H / M9_TH;
    const int i = lane & 15, kq = lane >> 4;
    const int g = (i & 8) ? i : (i ^ 4);                      // pixel group of matrix row i
    const float* xa = Xs + (row * 16 + 4 * kh + kq) * M9_PW + 4 * g;  // + (ky * 16 + 8 cqi) * PW + 4 d
    // filter fragments: buffer loads with the lane part of the address in voffset and the (phase, fragment) part in soffset -- with
    // plain pointers the compiler hoists a 64-bit address per fragment out of the tile loop (200+ registers, spills)
    const int wl = (kh * 6 * 64 + lane) * 16;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wt), 0, 9 * 4 * 2 * 6 * 64 * 4 * 4, 0x00020000);
    f32x4 wpre[M9_D + 1][6];                                           // filter fragments of phase p sit in set p % (M9_D + 1)
    f32x4 hpre[M9_HQ][4];
    auto filt_load = [&](int p) {
#pragma unroll
        for (int f = 0; f < 6; ++f)
            wpre[p % (M9_D + 1)][f] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                wrs, wl, (((p % 9) * 4 + p / 9) * 12 + f) * 1024, 0));
    };
    // halo rows h0-4 .. h0+7, pixels w0-4 .. w0+67, channels c0 .. c0+15 of tile `tile`: thread t owns halo pixels t and t+512 (< 864)
    auto halo_load = [&](int tile, int c0, int q) {
        const int tw = tile % tiles_w, th = (tile / tiles_w) % tiles_h, n = tile / (tiles_w * tiles_h);
        const int pp = t + 512 * q;
        const int r = pp / 72, px = pp - r * 72;
        const int hh = th * M9_TH + r - 4, ww = tw * M9_TW + px - 4;
        const bool ok = tile < ntiles && pp < M9_HPX && hh >= 0 && hh < H && ww >= 0 && ww < W;
        const f32x4* src = reinterpret_cast<const f32x4*>(x + (((long)n * H + hh) * W + ww) * 64 + c0);
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) hpre[q][c4] = ok ? src[c4] : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto halo_store = [&](int q) {
        const int pp = t + 512 * q;
        if (pp < M9_HPX) {
            const int r = pp / 72, px = pp - r * 72;
            float* d = Xs + r * 16 * M9_PW + px;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                for (int e = 0; e < 4; ++e) d[(4 * c4 + e) * M9_PW] = hpre[q][c4][e];
        }
    };
#pragma unroll
    for (int q = 0; q < M9_D; ++q) filt_load(q);
#pragma unroll
    for (int q = 0; q < M9_HQ; ++q) halo_load(blockIdx.x, 0, q);
#pragma unroll
    for (int q = 0; q < M9_HQ; ++q) halo_store(q);
    __syncthreads();
    const float bo = bias ? bias[i & 3] : 0.f;
    // persistent over tiles; per tile 36 phases = 4 channel chunks x 9 filter rows, unrolled so that the register-set indices are
    // static (36 is a multiple of the M9_D + 1 sets, so the rotation continues seamlessly into the next tile)
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        f32x4 acc4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc4[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int ky = 0; ky < 9; ++ky) {
                const int p = 9 * c + ky;
                filt_load((p + M9_D) % 36);                   // beyond phase 35: the next tile's first phases (same filter)
                if (ky < M9_HQ) {                             // the next chunk's halo: of this tile, or chunk 0 of the next one
                    if (c < 3) halo_load(tile, 16 * (c + 1), ky);
                    else halo_load(tile + gridDim.x, 0, ky);
                }
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(xa + (ky * 16 + 0) * M9_PW + 4 * d);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(xa + (ky * 16 + 8) * M9_PW + 4 * d);
                    const f32x4 b0 = wpre[p % (M9_D + 1)][d], b1 = wpre[p % (M9_D + 1)][3 + d];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {             // four accumulator chains, consecutive MFMAs never depend on each other
                        acc4[u & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], b0[u], acc4[u & 1], 0, 0, 0);
                        acc4[2 + (u & 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], b1[u], acc4[2 + (u & 1)], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);            // keep the scheduler from hoisting a whole chunk's LDS reads (it spills)
                if (ky == 8) {                                // chunk boundary: every wave must have left the halo first
                    f32x4 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < M9_HQ; ++q) halo_store(q);
                    if (c == 3 && kh == 1) *reinterpret_cast<f32x4*>(&Red[(row * 64 + lane) * 4]) = acc;
                    __syncthreads();
                    if (c == 3 && kh == 0) {                  // the two channel halves of a row meet
                        acc += *reinterpret_cast<const f32x4*>(&Red[(row * 64 + lane) * 4]);
                        // C layout: column n = lane & 15 = (j, o); row 4 (lane >> 4) + reg -> pixel group g(row).  The 16 columns
                        // of a row are 16 consecutive floats of y: pixel 4 g + j, channel o.
                        const int tw = tile % tiles_w, th = (tile / tiles_w) % tiles_h, n = tile / (tiles_w * tiles_h);
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int mrow = 4 * kq + reg;
                            const int gg = (mrow & 8) ? mrow : (mrow ^ 4);
                            y[(((long)n * H + th * M9_TH + row) * W + tw * M9_TW + 4 * gg) * 4 + i] = acc[reg] + bo;
                        }
                    }
                }
            }
        }
    }
}
// x (B,H,W,64) NHWC contiguous, H % 4 == 0, W % 64 == 0; wt = Toeplitz-expanded filter (110,592 floats in MFMA fragment order) from
// tatt_repack_conv_weight mode 8 (forward filter of a 64->4 convolution) / mode 9 (data gradient of a 4->64 one); y (B,H,W,4)
TATT_API int tatt_conv9_c64_to_c4_mfma(const float* x, const float* wt, const float* bias, float* y, int B, int H, int W,
                                       hipStream_t st) {
    if (H % M9_TH || W % M9_TW) return 1;
    const int ntiles = B * (H / M9_TH) * (W / M9_TW);
    hipLaunchKernelGGL(conv9_c64_to_c4_mfma_kernel, dim3(ntiles < 256 ? ntiles : 256), dim3(512), 0, st, x, wt, bias, y, B, H, W,
                       ntiles);
    return LAUNCH_CHECK();
}

// ---- the same Toeplitz convolution on the bf16 matrix cores with split operands (round 5) ------------------------------------------
// The fp32 form above runs at 65 % of the fp32 MFMA peak: 107 us for the 64 -> 4 output convolution at HR resolution, the longest
// kernel of the training step.  Here every fp32 operand is a = hi + lo (hi = bf16(a), lo = bf16(a - hi)) and a b = hi hi + hi lo +
// lo hi with fp32 accumulation (2^-16 relative per product -- the arithmetic of tatt_conv3_c64_fwd_sb), on v_mfma_f32_16x16x32_bf16:
// 16x the fp32 rate for three products.  Same tile (4 rows x 64 pixels), 16-channel chunks, phases = (chunk, filter row), filter
// fragments from global memory some phases ahead, halo of the next chunk prefetched to registers.
//   rows i = pixel group g = i (4 pixels), columns n = (j, o), one MFMA contracts k = 32 = 2 pixel offsets x 16 channels:
//   lane (i, kq): dx = 2 pair + (kq >> 1), channels 8 (kq & 1) .. + 7 of the chunk -- 8 consecutive bf16 of ONE halo pixel, one ds_read_b128.
//   12 waves: wave = (dx pair 0..5, row half): ONE pair of filter fragments (hi, lo) per phase serves its two rows -- the first version
//   (wave = (row, three pairs), 6 fragments per wave and phase, each fragment fetched by four waves) was bound by the filter's way
//   through the vector memory path and by fetching only three now much shorter phases ahead: 75 us; with two fragments per phase
//   the same registers hold five phases: 67 us.  Six accumulator chains per wave ((row, product)): no MFMA waits for its predecessor.
//   With the images dealt to the XCDs (below) HBM traffic fell from 221 to 53 MB: 58 us (fp32 form: 115).  What is left (PMC,
//   profiles/r05_pmc_conv9_sb.txt): the MFMA pipe is busy 27 % of the time; a wave spends 36 % of its cycles at s_waitcnt -- vector
//   loads return in order, so the filter fragments (L2 hits, needed 5 phases later) queue behind the next chunk's halo loads (first
//   touch of the lines), and a chunk's nine phases (1.1 us) are shorter than that latency.
//   Halo: a hi and a lo bf16 image, pixel-major with the four pixel PHASES (px & 3) of a row apart: [row 12][px & 3][px >> 2 (18)][16 ch]
//   = 32 B per pixel.  A lane group of a ds_read_b128 then holds 8 lanes of one channel half and 8 of the other over consecutive pixel
//   groups: 16 distinct 16-byte slots of the 256-byte bank row (MI355X_MICROARCH.md, LDS) -- conflict-free without padding.  The
//   operands are split ONCE, when the halo is stored.  The six pairs' partial sums of a row meet through LDS at the end of a tile.
#define S9_ROWB (4 * 18 * 32)                    // bytes of one halo row of one image: 2304
#define S9_IMG (M9_ROWS * S9_ROWB)               // bytes of one image (hi or lo): 27,648
#define S9_THREADS 768
#define S9_D 5                                   // filter fragments are fetched S9_D phases ahead (S9_D + 1 divides 36)
#define S9_RED (12 * 2 * 64 * 16)                // bytes: every wave's two row accumulators
#define S9_LDS (2 * S9_IMG + S9_RED)             // 79,872 B
typedef __bf16 s9_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 s9_bf16x2 __attribute__((ext_vector_type(2)));
typedef float s9_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned s9_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void s9_split(f32x4 v, unsigned& h0, unsigned& h1, unsigned& l0, unsigned& l1) {
    const s9_f32x2 a = (s9_f32x2){v[0], v[1]}, b = (s9_f32x2){v[2], v[3]};
    const s9_bf16x2 ha = __builtin_convertvector(a, s9_bf16x2), hb = __builtin_convertvector(b, s9_bf16x2);
    const s9_bf16x2 la = __builtin_convertvector(a - __builtin_convertvector(ha, s9_f32x2), s9_bf16x2);
    const s9_bf16x2 lb = __builtin_convertvector(b - __builtin_convertvector(hb, s9_f32x2), s9_bf16x2);
    h0 = __builtin_bit_cast(unsigned, ha); h1 = __builtin_bit_cast(unsigned, hb);
    l0 = __builtin_bit_cast(unsigned, la); l1 = __builtin_bit_cast(unsigned, lb);
}
__global__ __launch_bounds__(S9_THREADS) void conv9_c64_to_c4_sb_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                                        int B, int H, int W, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s9_lds[];
    unsigned char* const Xs = s9_lds;                                       // hi image, lo image
    float* const Red = reinterpret_cast<float*>(s9_lds + 2 * S9_IMG);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int pair = wave % 6, rh = wave / 6;
    const int tiles_w = W / M9_TW, tiles_h = H / M9_TH;
    const int i = lane & 15, kq = lane >> 4;
    // A fragment of (row, ky): lane base + the pair's (pixel phase, group) offset + ky rows
    const int pair_off = (((2 * pair) & 3) * 18 + (pair >> 1)) * 32;
    int xa[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) xa[r] = (((2 * rh + r) * 4 + (kq >> 1)) * 18 + i) * 32 + (kq & 1) * 16 + pair_off;
    const int wl = (pair * 2 * 64 + lane) * 16;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wt), 0, 9 * 4 * 2 * 6 * 64 * 4 * 4, 0x00020000);
    f32x4 wpre[S9_D + 1][2];                                           // {hi, lo} of phase p in set p % (S9_D + 1)
    f32x4 hpre[M9_HQ][4];
    auto filt_load = [&](int p) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
            wpre[p % (S9_D + 1)][f] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                wrs, wl, (((p % 9) * 4 + p / 9) * 12 + f) * 1024, 0));
    };
    // halo rows h0-4 .. h0+7, pixels w0-4 .. w0+67, channels c0 .. c0+15 of tile `tile`: thread t owns halo pixels t and t+768 (< 864)
    auto halo_load = [&](int tile, int c0, int q) {
        const int tw = tile % tiles_w, th = (tile / tiles_w) % tiles_h, n = tile / (tiles_w * tiles_h);
        const int pp = t + S9_THREADS * q;
        const int r = pp / 72, px = pp - r * 72;
        const int hh = th * M9_TH + r - 4, ww = tw * M9_TW + px - 4;
        const bool ok = tile < ntiles && pp < M9_HPX && hh >= 0 && hh < H && ww >= 0 && ww < W;
        const f32x4* src = reinterpret_cast<const f32x4*>(x + (((long)n * H + hh) * W + ww) * 64 + c0);
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) hpre[q][c4] = ok ? src[c4] : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto halo_store = [&](int q) {
        const int pp = t + S9_THREADS * q;
        if (pp < M9_HPX) {
            const int r = pp / 72, px = pp - r * 72;
            unsigned char* d = Xs + ((r * 4 + (px & 3)) * 18 + (px >> 2)) * 32;
            unsigned h[8], l[8];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) s9_split(hpre[q][c4], h[2 * c4], h[2 * c4 + 1], l[2 * c4], l[2 * c4 + 1]);
            *reinterpret_cast<s9_u32x4*>(d) = (s9_u32x4){h[0], h[1], h[2], h[3]};
            *reinterpret_cast<s9_u32x4*>(d + 16) = (s9_u32x4){h[4], h[5], h[6], h[7]};
            *reinterpret_cast<s9_u32x4*>(d + S9_IMG) = (s9_u32x4){l[0], l[1], l[2], l[3]};
            *reinterpret_cast<s9_u32x4*>(d + S9_IMG + 16) = (s9_u32x4){l[4], l[5], l[6], l[7]};
        }
    };
    // Tile order.  A tile's halo is 12 rows for 4 rows of output: neighbouring tiles read each other's rows, and in launch order they
    // run on different XCDs -- the fp32 kernel's order fetches 221 MB from HBM for a 50 MB input (PMC, B = 48), which binds this
    // 5x shorter kernel.  Here an XCD (= blockIdx.x & 7) owns whole images (x, x + 8, ...) and its work-groups walk their tiles in
    // image order, so that vertical neighbours are in flight on the same L2 together.
    const int per_img = tiles_w * tiles_h;
    const bool by_xcd = (gridDim.x & 7) == 0 && (B & 7) == 0;       // (whole images per XCD: other batch sizes keep the launch order)
    const int xcd = blockIdx.x & 7, wg_m = blockIdx.x >> 3, wg_M = gridDim.x >> 3;
    auto tile_of = [&](int s) -> int {                        // s-th tile of this work-group; ntiles: none
        if (!by_xcd) { const long tl = (long)blockIdx.x + (long)s * gridDim.x; return tl < ntiles ? (int)tl : ntiles; }
        const int sq = wg_m + s * wg_M, img = xcd + 8 * (sq / per_img);
        return img < B ? img * per_img + sq % per_img : ntiles;
    };
#pragma unroll
    for (int q = 0; q < S9_D; ++q) filt_load(q);
#pragma unroll
    for (int q = 0; q < M9_HQ; ++q) halo_load(tile_of(0), 0, q);
#pragma unroll
    for (int q = 0; q < M9_HQ; ++q) halo_store(q);
    __syncthreads();
    const float bo = bias ? bias[i & 3] : 0.f;
#pragma unroll 1
    for (int ts = 0;; ++ts) {
        const int tile = tile_of(ts);
        if (tile >= ntiles) break;
        const int tile_next = tile_of(ts + 1);
        f32x4 acc6[2][3];                                     // [row][product]: six independent chains
#pragma unroll
        for (int q = 0; q < 6; ++q) acc6[q / 3][q % 3] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int ky = 0; ky < 9; ++ky) {
                const int p = 9 * c + ky;
                filt_load((p + S9_D) % 36);                   // beyond phase 35: the next tile's first phases (same filter)
                if (ky < M9_HQ) {                             // the next chunk's halo: of this tile, or chunk 0 of the next one
                    if (c < 3) halo_load(tile, 16 * (c + 1), ky);
                    else halo_load(tile_next, 0, ky);
                }
                const s9_bf16x8 bh = __builtin_bit_cast(s9_bf16x8, wpre[p % (S9_D + 1)][0]);
                const s9_bf16x8 bl = __builtin_bit_cast(s9_bf16x8, wpre[p % (S9_D + 1)][1]);
                s9_bf16x8 ah[2], al[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    ah[r] = *reinterpret_cast<const s9_bf16x8*>(Xs + xa[r] + ky * S9_ROWB);
                    al[r] = *reinterpret_cast<const s9_bf16x8*>(Xs + xa[r] + ky * S9_ROWB + S9_IMG);
                }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    acc6[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[r], bh, acc6[r][0], 0, 0, 0);
                    acc6[r][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[r], bl, acc6[r][1], 0, 0, 0);
                    acc6[r][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[r], bh, acc6[r][2], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ky == 8) {                                // chunk boundary: every wave must have left the halo first
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < M9_HQ; ++q) halo_store(q);
                    if (c == 3) {
#pragma unroll
                        for (int r = 0; r < 2; ++r)
                            *reinterpret_cast<f32x4*>(&Red[((wave * 2 + r) * 64 + lane) * 4]) = (acc6[r][0] + acc6[r][1]) + acc6[r][2];
                    }
                    __syncthreads();
                    if (c == 3 && pair < 2) {                 // the six dx pairs of a row meet: wave (pair 0 / 1, rh) finishes row 2 rh + pair
                        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int pw = 0; pw < 6; ++pw)
                            acc += *reinterpret_cast<const f32x4*>(&Red[(((rh * 6 + pw) * 2 + pair) * 64 + lane) * 4]);
                        // C layout: column n = lane & 15 = (j, o); row 4 (lane >> 4) + reg = pixel group: 16 consecutive floats of y
                        const int tw = tile % tiles_w, th = (tile / tiles_w) % tiles_h, n = tile / (tiles_w * tiles_h);
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg)
                            y[(((long)n * H + th * M9_TH + 2 * rh + pair) * W + tw * M9_TW + 4 * (4 * kq + reg)) * 4 + i] = acc[reg] + bo;
                    }
                }
            }
        }
    }
}
// as tatt_conv9_c64_to_c4_mfma; wt = the split-bf16 Toeplitz filter from tatt_repack_conv_weight mode 12 (forward filter of a
// 64 -> 4 convolution) / mode 13 (data gradient of a 4 -> 64 one)
TATT_API int tatt_conv9_c64_to_c4_sb(const float* x, const float* wt, const float* bias, float* y, int B, int H, int W,
                                     hipStream_t st) {
    if (H % M9_TH || W % M9_TW) return 1;
    const int ntiles = B * (H / M9_TH) * (W / M9_TW);
    static TattPerDevice attr_once;
    tatt_per_device(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv9_c64_to_c4_sb_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, S9_LDS);
    });
    hipLaunchKernelGGL(conv9_c64_to_c4_sb_kernel, dim3(ntiles < 256 ? ntiles : 256), dim3(S9_THREADS), S9_LDS, st, x, wt, bias, y, B,
                       H, W, ntiles);
    return LAUNCH_CHECK();
}

// ---- 9x9 convolution FROM 4 channels TO 64 on the matrix cores -----------------------------------------------------------------
// y[px][n] = act(sum_{tap, k} in[px + tap - 4][k] * wp[tap][k][n] + bias[n]),  wp = [81][4][64] (repack mode 0: block1's forward
// convolution, reference model/tsrn.py:597; mode 1: the data gradient of the 64->4 reconstruction convolution, :623).
// The 4 input channels are exactly the k = 4 of v_mfma_f32_16x16x4_f32 -- one MFMA per (tap, 16 outputs x 16 pixels), no padding
// anywhere.  A[i = output channel][k] = wp[tap][k][16 nt + i] is WEIGHT-STATIONARY: 81 registers per lane hold a wave's whole
// filter slice for every tile the (persistent) group walks.  B[k][j = pixel] = one LDS dword per lane from the 4-channel halo tile
// (12 rows x 72 pixels x 4 channels = 13.8 KB; the 64 lanes of a read cover 64 consecutive dwords).  C rows = output channels:
// a lane ends up with 4 consecutive channels of one pixel = one 16-byte store.
// Work-group = 8 waves on a 4-row x 64-pixel tile: wave (nt = w & 3, half = w >> 2) computes output channels 16 nt .. +15 of rows
// 2 half, 2 half + 1; the four 16-pixel groups of a row are four independent accumulator chains.  The next tile's halo travels
// global -> registers under the MFMAs.
#define F9_TH 4
#define F9_TW 64
#define F9_DW (F9_TW + 8)
#define F9_ROWS (F9_TH + 8)
struct F9P { const float* in; const float* wp; const float* bias; float* out; int B, H, W, ntiles, act; };
__global__ __launch_bounds__(512) void conv9_c4_to_c64_kernel(F9P p) {
    __shared__ __attribute__((aligned(16))) float Ds[F9_ROWS * F9_DW * 4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nt = wave & 3, half = wave >> 2, li = lane & 15, lk = lane >> 4;
    const int tiles_w = p.W / F9_TW, tiles_h = p.H / F9_TH;
    float wreg[81];
#pragma unroll
    for (int tap = 0; tap < 81; ++tap) wreg[tap] = p.wp[(tap * 4 + lk) * 64 + 16 * nt + li];
    f32x4 bo = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.bias) bo = *reinterpret_cast<const f32x4*>(p.bias + 16 * nt + 4 * lk);
    f32x4 dr0, dr1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto decode = [&](int tile, int& b, int& h0, int& w0) {
        const int tw = tile % tiles_w; tile /= tiles_w;
        h0 = (tile % tiles_h) * F9_TH; b = tile / tiles_h; w0 = tw * F9_TW;
    };
    auto load1 = [&](int e, int b, int h0, int w0) -> f32x4 {
        const int d = e / F9_DW, q = e - d * F9_DW, row = h0 - 4 + d, px = w0 - 4 + q;
        if (row < 0 || row >= p.H || px < 0 || px >= p.W) return (f32x4){0.f, 0.f, 0.f, 0.f};
        return *reinterpret_cast<const f32x4*>(p.in + (((long)b * p.H + row) * p.W + px) * 4);
    };
    auto load_halo = [&](int tile) {
        int b, h0, w0; decode(tile, b, h0, w0);
        dr0 = load1(t, b, h0, w0);
        if (t < F9_ROWS * F9_DW - 512) dr1 = load1(t + 512, b, h0, w0);
    };
    if ((int)blockIdx.x < p.ntiles) load_halo(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        __syncthreads();                                   // the previous tile has been consumed
        *reinterpret_cast<f32x4*>(&Ds[4 * t]) = dr0;
        if (t < F9_ROWS * F9_DW - 512) *reinterpret_cast<f32x4*>(&Ds[4 * (t + 512)]) = dr1;
        if (tile + (int)gridDim.x < p.ntiles) load_halo(tile + gridDim.x);
        __syncthreads();
        int b, h0, w0; decode(tile, b, h0, w0);
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {
            const int row = 2 * half + r;
            const float* db = &Ds[(row * F9_DW + li) * 4 + lk];
            f32x4 acc[4];
#pragma unroll
            for (int mg = 0; mg < 4; ++mg) acc[mg] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 9; ++ky)
#pragma unroll
                for (int kx = 0; kx < 9; ++kx)
#pragma unroll
                    for (int mg = 0; mg < 4; ++mg)
                        acc[mg] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[ky * 9 + kx], db[(ky * F9_DW + kx + 16 * mg) * 4], acc[mg], 0, 0, 0);
            float* o = p.out + (((long)b * p.H + h0 + row) * p.W + w0 + li) * 64 + 16 * nt + 4 * lk;
#pragma unroll
            for (int mg = 0; mg < 4; ++mg) {
                f32x4 v = acc[mg] + bo;
                if (p.act != ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
                }
                *reinterpret_cast<f32x4*>(o + (long)16 * mg * 64) = v;
            }
        }
    }
}
// in (B,H,W,4) NHWC contiguous, H % 4 == 0, W % 64 == 0; wp [81][4][64]; out (B,H,W,64)
TATT_API int tatt_conv9_c4_to_c64(const float* in, const float* wp, const float* bias, float* out, int B, int H, int W, int act,
                                  hipStream_t st) {
    if (H % F9_TH || W % F9_TW) return 1;
    F9P p = {in, wp, bias, out, B, H, W, B * (H / F9_TH) * (W / F9_TW), act};
    const int G = p.ntiles < 256 ? p.ntiles : 256;
    hipLaunchKernelGGL(conv9_c4_to_c64_kernel, dim3(G), dim3(512), 0, st, p);
    return LAUNCH_CHECK();
}

// ---- the same 4 -> 64 convolution on the bf16 matrix cores with split operands (round 5) ------------------------------------------
// k = 32 per v_mfma_f32_16x16x32_bf16 = 8 taps x 4 input channels: the 81 x 4 = 324-long contraction is 11 MFMAs (92 % useful) x 3
// products (hi hi, hi lo, lo hi; a = hi + lo, 2^-16 relative per product) instead of 81 fp32 ones at a sixteenth of the rate.
//   k-slot (m, kq, e): tap = 8 m + 2 kq + (e >> 2), channel e & 3.  A (weights, rows = 16 output channels) is stationary: 11 x (hi, lo)
//   fragments = 88 registers, split once per work-group from the fp32 filter [81][4][64].  B (columns = 16 pixels): a lane's 8 bf16
//   are the 4 channels of TWO taps' pixels = two 8-byte LDS reads from a pixel-major bf16 halo image (12 rows x 72 pixels x 4 channels,
//   a hi and a lo one: 13.8 KB); their addresses differ per lane quarter (a tap pair may wrap to the next filter row), so each lane
//   keeps its 22 tap addresses in registers and (row, pixel group) go into the instructions' immediate offsets.
// Same tile, wave roles and epilogue as the fp32 kernel: 4 rows x 64 pixels, wave (nt, half) = 16 output channels of 2 rows, the next
// tile's halo prefetched to registers.
#define G9_IMG (F9_ROWS * F9_DW * 8)              // bytes of one halo image (hi or lo): 6912
typedef __bf16 g9_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned g9_u32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void conv9_c4_to_c64_sb_kernel(F9P p) {
    __shared__ __attribute__((aligned(16))) unsigned char Ds[2 * G9_IMG];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nt = wave & 3, half = wave >> 2, li = lane & 15, lk = lane >> 4;
    const int tiles_w = p.W / F9_TW, tiles_h = p.H / F9_TH;
    // stationary A fragments: output channel 16 nt + li, k-slots (tap 8 m + 2 lk + (e >> 2), channel e & 3); taps >= 81 are zero
    g9_bf16x8 wh[11], wl[11];
    int ta[11][2];                                         // byte address of the lane's two taps' pixels for (row 2 half, group 0)
#pragma unroll
    for (int m = 0; m < 11; ++m) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int tap = 8 * m + 2 * lk + (e >> 2);
            v[e] = tap < 81 ? p.wp[(tap * 4 + (e & 3)) * 64 + 16 * nt + li] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const s9_f32x2 a = (s9_f32x2){v[e], v[e + 1]};
            const s9_bf16x2 h = __builtin_convertvector(a, s9_bf16x2);
            const s9_bf16x2 l = __builtin_convertvector(a - __builtin_convertvector(h, s9_f32x2), s9_bf16x2);
            wh[m][e] = h[0]; wh[m][e + 1] = h[1];
            wl[m][e] = l[0]; wl[m][e + 1] = l[1];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tap = min(8 * m + 2 * lk + u, 80), ky = tap / 9, kx = tap - 9 * ky;
            ta[m][u] = (((2 * half + ky) * F9_DW) + li + kx) * 8;
        }
    }
    f32x4 bo = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.bias) bo = *reinterpret_cast<const f32x4*>(p.bias + 16 * nt + 4 * lk);
    f32x4 dr0, dr1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto decode = [&](int tile, int& b, int& h0, int& w0) {
        const int tw = tile % tiles_w; tile /= tiles_w;
        h0 = (tile % tiles_h) * F9_TH; b = tile / tiles_h; w0 = tw * F9_TW;
    };
    auto load1 = [&](int e, int b, int h0, int w0) -> f32x4 {
        const int d = e / F9_DW, q = e - d * F9_DW, row = h0 - 4 + d, px = w0 - 4 + q;
        if (row < 0 || row >= p.H || px < 0 || px >= p.W) return (f32x4){0.f, 0.f, 0.f, 0.f};
        return *reinterpret_cast<const f32x4*>(p.in + (((long)b * p.H + row) * p.W + px) * 4);
    };
    auto load_halo = [&](int tile) {
        int b, h0, w0; decode(tile, b, h0, w0);
        dr0 = load1(t, b, h0, w0);
        if (t < F9_ROWS * F9_DW - 512) dr1 = load1(t + 512, b, h0, w0);
    };
    auto store1 = [&](int e, f32x4 v) {                    // split once: 4 channels -> 8 bytes of hi, 8 bytes of lo
        unsigned h0, h1, l0, l1;
        s9_split(v, h0, h1, l0, l1);
        *reinterpret_cast<g9_u32x2*>(Ds + e * 8) = (g9_u32x2){h0, h1};
        *reinterpret_cast<g9_u32x2*>(Ds + G9_IMG + e * 8) = (g9_u32x2){l0, l1};
    };
    if ((int)blockIdx.x < p.ntiles) load_halo(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        __syncthreads();                                   // the previous tile has been consumed
        store1(t, dr0);
        if (t < F9_ROWS * F9_DW - 512) store1(t + 512, dr1);
        if (tile + (int)gridDim.x < p.ntiles) load_halo(tile + gridDim.x);
        __syncthreads();
        int b, h0, w0; decode(tile, b, h0, w0);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            f32x4 acc[4];
#pragma unroll
            for (int mg = 0; mg < 4; ++mg) acc[mg] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 11; ++m) {
                g9_bf16x8 bh[4], bl[4];
#pragma unroll
                for (int mg = 0; mg < 4; ++mg) {
                    const int off = (r * F9_DW + 16 * mg) * 8;
                    g9_u32x2 q0 = *reinterpret_cast<const g9_u32x2*>(Ds + ta[m][0] + off), q1 = *reinterpret_cast<const g9_u32x2*>(Ds + ta[m][1] + off);
                    g9_u32x2 q2 = *reinterpret_cast<const g9_u32x2*>(Ds + ta[m][0] + off + G9_IMG), q3 = *reinterpret_cast<const g9_u32x2*>(Ds + ta[m][1] + off + G9_IMG);
                    bh[mg] = __builtin_bit_cast(g9_bf16x8, (s9_u32x4){q0[0], q0[1], q1[0], q1[1]});
                    bl[mg] = __builtin_bit_cast(g9_bf16x8, (s9_u32x4){q2[0], q2[1], q3[0], q3[1]});
                }
                // product-major over the four pixel groups: consecutive MFMAs never depend on each other
#pragma unroll
                for (int mg = 0; mg < 4; ++mg) acc[mg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[m], bh[mg], acc[mg], 0, 0, 0);
#pragma unroll
                for (int mg = 0; mg < 4; ++mg) acc[mg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[m], bl[mg], acc[mg], 0, 0, 0);
#pragma unroll
                for (int mg = 0; mg < 4; ++mg) acc[mg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[m], bh[mg], acc[mg], 0, 0, 0);
            }
            const int row = 2 * half + r;
            float* o = p.out + (((long)b * p.H + h0 + row) * p.W + w0 + li) * 64 + 16 * nt + 4 * lk;
#pragma unroll
            for (int mg = 0; mg < 4; ++mg) {
                f32x4 v = acc[mg] + bo;
                if (p.act != ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
                }
                *reinterpret_cast<f32x4*>(o + (long)16 * mg * 64) = v;
            }
        }
    }
}
// as tatt_conv9_c4_to_c64 (same packed fp32 filter wp [81][4][64]: the kernel splits it itself), products in split bf16
TATT_API int tatt_conv9_c4_to_c64_sb(const float* in, const float* wp, const float* bias, float* out, int B, int H, int W, int act,
                                     hipStream_t st) {
    if (H % F9_TH || W % F9_TW) return 1;
    F9P p = {in, wp, bias, out, B, H, W, B * (H / F9_TH) * (W / F9_TW), act};
    const int G = p.ntiles < 256 ? p.ntiles : 256;
    hipLaunchKernelGGL(conv9_c4_to_c64_sb_kernel, dim3(G), dim3(512), 0, st, p);
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
struct W9P { const float* x; const float* dy; float* part; int B, H, W, ntiles, flip; };
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
            // flip: the 4-channel tensor is the convolution's INPUT (weight gradient of a 4->64 convolution): rows r+ky-4, pixels p+kx-4
            bbase[j] = p.flip ? (ky * W9_DW + lk + kx) * 4 + co : ((8 - ky) * W9_DW + lk + 8 - kx) * 4 + co;
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
// ---- the same weight gradient on the bf16 matrix cores with split operands (round 5) --------------------------------------------
// The contraction runs over PIXELS, so an operand of v_mfma_f32_16x16x32_bf16 needs 8 consecutive pixels of one channel per lane: the
// operands are gathered from the token-major fp32 images with 8 ds_read_b32 down a column and split in registers (a = hi + lo,
// hi hi + hi lo + lo hi, fp32 accumulation: 2^-16 relative per product).  Per input row there are 8 A fragments (4 channel tiles x 2
// halves of the 64 pixels) and 42 B fragments (21 column tiles x 2); every A fragment is needed by three waves and every B fragment
// by four.  The first version gathered each in every wave that needs it (192 gathers per row, LDS-bound: 52 us at HR against 83 for
// the fp32 form); here the 12 waves share the 50 gathers of a row, leave the hi / lo fragments in LDS in MFMA order (as
// gru_wgrad_sb_kernel does) and every wave then reads its 1 + 7 fragments per half with ds_read_b128: two barriers per row.
// Same tile walk, staging and partial layout as the fp32 kernel; the X image's pitch is 82 dwords (the two pixel octets of a 32-lane
// read group fall in different bank halves; rows are 8-byte aligned: two 8-byte stores per vector).
#define W9S_XP 82
#define W9S_NFRAG 50
#define W9S_XS (2 * W9_P * W9S_XP)                          // floats
#define W9S_DS ((W9_DROWS + 1) * W9_DW * 4)                  // floats (+ one row of zeros for the 12 unused columns)
#define W9S_LDS ((W9S_XS + W9S_DS) * 4 + W9S_NFRAG * 2 * 64 * 16)      // 41,984 + 14,976 + 102,400 = 159,360 bytes
__device__ __forceinline__ void w9s_frag(const float* __restrict__ col, int stride, g9_bf16x8& hi, g9_bf16x8& lo) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = col[e * stride];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const s9_f32x2 a = (s9_f32x2){v[e], v[e + 1]};
        const s9_bf16x2 h = __builtin_convertvector(a, s9_bf16x2);
        const s9_bf16x2 l = __builtin_convertvector(a - __builtin_convertvector(h, s9_f32x2), s9_bf16x2);
        hi[e] = h[0]; hi[e + 1] = h[1];
        lo[e] = l[0]; lo[e + 1] = l[1];
    }
}
__global__ __launch_bounds__(768) void conv9_c64_c4_wgrad_sb_kernel(W9P p) {
    extern __shared__ __attribute__((aligned(16))) float w9s_lds[];
    float* const Xs = w9s_lds;                                             // [2][64 pixels][82]
    float* const Ds = w9s_lds + W9S_XS;
    f32x4* const Fr = reinterpret_cast<f32x4*>(w9s_lds + W9S_XS + W9S_DS); // [fragment][hi, lo][lane]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int mt = wave & 3, ng = wave >> 2, li = lane & 15, lk = lane >> 4;
    const int tiles_w = p.W / W9_P, tiles_h = p.H / W9_R;
    // this wave's share of a row's 50 gathers: fragments wave, wave + 12, ... ; < 8: A (channel tile f >> 1, half f & 1), else B
    // (column tile (f - 8) >> 1, half (f - 8) & 1).  gb: the lane's gather base (dwords), gs: stride between the 8 pixels, gr: per-row step
    int gb[5], gs[5], gr[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int f = wave + 12 * q;
        if (f < 8) {
            gb[q] = ((f & 1) * 32 + 8 * lk) * W9S_XP + 16 * (f >> 1) + li; gs[q] = W9S_XP; gr[q] = 0;
        } else {
            const int ntile = (f - 8) >> 1, ks = (f - 8) & 1;
            const int n = 16 * ntile + li, tap = n >> 2, co = n & 3;
            if (tap < 81 && f < W9S_NFRAG) {
                const int ky = tap / 9, kx = tap - 9 * ky;
                // flip: the 4-channel tensor is the convolution's INPUT (weight gradient of a 4->64 convolution): rows r+ky-4, pixels p+kx-4
                gb[q] = W9S_XS + (p.flip ? (ky * W9_DW + 8 * lk + kx) * 4 + co : ((8 - ky) * W9_DW + 8 * lk + 8 - kx) * 4 + co) + ks * 128;
                gr[q] = W9_DW * 4;
            } else { gb[q] = W9S_XS + W9_DROWS * W9_DW * 4; gr[q] = 0; }
            gs[q] = 4;
        }
    }
    for (int e = t; e < W9_DW * 4; e += 768) Ds[W9_DROWS * W9_DW * 4 + e] = 0.f;
    f32x4 acc[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int my_tiles = p.ntiles > (int)blockIdx.x ? (p.ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int nsteps = my_tiles * W9_R;
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
        float* d0 = Xs + buf * W9_P * W9S_XP + (t >> 4) * W9S_XP + 4 * (t & 15);
        *reinterpret_cast<float2*>(d0) = make_float2(xr0[0], xr0[1]);
        *reinterpret_cast<float2*>(d0 + 2) = make_float2(xr0[2], xr0[3]);
        if (t < 256) {
            float* d1 = Xs + buf * W9_P * W9S_XP + ((t + 768) >> 4) * W9S_XP + 4 * (t & 15);
            *reinterpret_cast<float2*>(d1) = make_float2(xr1[0], xr1[1]);
            *reinterpret_cast<float2*>(d1 + 2) = make_float2(xr1[2], xr1[3]);
        }
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
        store_x(buf);                            // (the other X buffer than the one the previous row's gathers read)
        if (rr == 0) store_d();                  // Ds: last read by the previous row's gathers, which every wave left at ITS barrier 2
        if (s + 1 < nsteps) { load_x(s + 1); if (rr == W9_R - 1) load_d(s + 1); }
        __syncthreads();                         // 1: images complete; every wave has left the previous row's MFMAs (the fragments are free)
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int f = wave + 12 * q;
            if (f < W9S_NFRAG) {
                g9_bf16x8 hi, lo;
                w9s_frag(w9s_lds + gb[q] + (f < 8 ? buf * W9_P * W9S_XP : rr * gr[q]), gs[q], hi, lo);
                Fr[(f * 2 + 0) * 64 + lane] = __builtin_bit_cast(f32x4, hi);
                Fr[(f * 2 + 1) * 64 + lane] = __builtin_bit_cast(f32x4, lo);
            }
        }
        __syncthreads();                         // 2: fragments complete
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const g9_bf16x8 ah = __builtin_bit_cast(g9_bf16x8, Fr[((2 * mt + ks) * 2 + 0) * 64 + lane]);
            const g9_bf16x8 al = __builtin_bit_cast(g9_bf16x8, Fr[((2 * mt + ks) * 2 + 1) * 64 + lane]);
            g9_bf16x8 bh[7], bl[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int f = 8 + 2 * (7 * ng + j) + ks;
                bh[j] = __builtin_bit_cast(g9_bf16x8, Fr[(f * 2 + 0) * 64 + lane]);
                bl[j] = __builtin_bit_cast(g9_bf16x8, Fr[(f * 2 + 1) * 64 + lane]);
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[j], 0, 0, 0);
        }
    }
    float* P = p.part + (long)blockIdx.x * 64 * W9_N;
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) P[(long)(16 * mt + 4 * lk + e) * W9_N + 16 * (7 * ng + j) + li] = acc[j][e];
}
// dw[co][ci][tap] (OIHW, Cout = 4, Cin = 64) = sum_g part[g][ci][tap*4 + co]; block = 64 outputs x 16 lanes over g
// (swap: dw[ci][co][tap], the OIHW gradient of the 4->64 convolution whose 64 OUTPUT channels are the matrix rows)
__global__ __launch_bounds__(1024) void conv9_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int G,
                                                                  int swap) {
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
        dw[(swap ? (long)ci * 4 + (n & 3) : (long)(n & 3) * 64 + ci) * 81 + (n >> 2)] = s;
    }
}
// x (B,H,W,64), dy (B,H,W,4) -> dw (4,64,9,9); H % 4 == 0, W % 64 == 0; part >= min(#tiles, 256) * 64 * 336 floats,
// #tiles = B * (H/4) * (W/64)
static int conv9_wgrad_launch(const float* x64, const float* t4, float* dw, float* part, int B, int H, int W, int flip,
                              hipStream_t st, bool sb = false) {
    if (H % W9_R || W % W9_P) return 1;
    W9P p = {x64, t4, part, B, H, W, B * (H / W9_R) * (W / W9_P), flip};
    const int G = p.ntiles < 256 ? p.ntiles : 256;
    if (sb) {
        static TattPerDevice attr_once;
        tatt_per_device(attr_once, [&] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv9_c64_c4_wgrad_sb_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W9S_LDS);
        });
        hipLaunchKernelGGL(conv9_c64_c4_wgrad_sb_kernel, dim3(G), dim3(768), W9S_LDS, st, p);
    }
    else hipLaunchKernelGGL(conv9_c64_c4_wgrad_mfma_kernel, dim3(G), dim3(768), 0, st, p);
    hipLaunchKernelGGL(conv9_wgrad_reduce_kernel, dim3(64 * 324 / 64), dim3(1024), 0, st, part, dw, G, flip);
    return LAUNCH_CHECK();
}
TATT_API int tatt_conv9_c64_c4_wgrad(const float* x, const float* dy, float* dw, float* part, int B, int H, int W,
                                     hipStream_t st) {
    return conv9_wgrad_launch(x, dy, dw, part, B, H, W, 0, st);
}
// x (B,H,W,4), dy (B,H,W,64) -> dw (64,4,9,9): the same GEMM with the roles of the two tensors exchanged (rows = the 64 OUTPUT
// channels from dy, Toeplitz columns = (tap, input channel) from x); weight gradient of block1 (reference model/tsrn.py:597)
TATT_API int tatt_conv9_c4_c64_wgrad(const float* x, const float* dy, float* dw, float* part, int B, int H, int W,
                                     hipStream_t st) {
    return conv9_wgrad_launch(dy, x, dw, part, B, H, W, 1, st);
}
// the two weight gradients with split-bf16 products on the bf16 matrix cores (same arguments, same workspace)
TATT_API int tatt_conv9_c64_c4_wgrad_sb(const float* x, const float* dy, float* dw, float* part, int B, int H, int W,
                                        hipStream_t st) {
    return conv9_wgrad_launch(x, dy, dw, part, B, H, W, 0, st, true);
}
TATT_API int tatt_conv9_c4_c64_wgrad_sb(const float* x, const float* dy, float* dw, float* part, int B, int H, int W,
                                        hipStream_t st) {
    return conv9_wgrad_launch(dy, x, dw, part, B, H, W, 1, st, true);
}
