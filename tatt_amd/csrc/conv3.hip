// 3x3 convolutions between multiples of 64 channels on NHWC maps -- the bulk of the backbone's FLOPs (11 of them per
// forward, reference model/tsrn.py:877,885,612,1043) -- on v_mfma_f32_32x32x2_f32 (exact fp32).
//
// Forward / data-gradient, one work-group tile = one 64-pixel row segment x 64 output channels, persistent work-groups:
//   * 64-channel contractions on the bf16 matrix cores by operand splitting (tatt_conv3_c64_fwd_sb, the default: three bf16 products
//     per fp32 product, fp32 accumulation; wider inputs are chunked) or in exact fp32 (tatt_conv3_c64_fwd_ws16): WEIGHT-STATIONARY,
//     the filter lives in registers for the lifetime of the work-group, the loop streams only activations through a double-buffered
//     LDS halo; both fold the producer's BatchNorm + activation into the halo staging and emit BatchNorm statistics of their output;
//   * more input channels in exact fp32 (tatt_conv3_c64_fwd_t): the filter slice of each tap is re-staged through LDS
//     (conv3_c64_fwd_v5_kernel), both MFMA operands read with 16-byte LDS loads.
// Weight-gradient (tatt_conv3_c64_wgrad_partial): persistent work-groups walk row segments; each wave keeps the 9 taps x
//   (32 ci x 32 co) quadrant in 9 accumulators (144 VGPRs); A = x halo read channel-contiguous, B = dy tile; per-block
//   partials are summed deterministically and scattered to the OIHW parameter layout by the split-K reducer of gemm.hip.
// History (measured on MI355X, B=48, 64->64 channels, 3.62 GFLOP/launch): one tile per work-group 59.6 us; persistent +
// prefetch 54 us; 16-byte LDS operand reads (v5) 52 us; weight-stationary 39 us (round 1), 38 us (ws16, round 2); split-bf16 18.6 us
// (round 3).
#include "common.h"
#include <mutex>
#include <stdlib.h>
#include <type_traits>

#define C3_PX 64
#define C3_HW 66     // halo width (pixels)

struct Conv3P {
    const float* x; const float* w; const float* bias; float* y;
    int B, H, W, Cin, Cout, act;
    float beta;
    // ws16 kernel only: BatchNorm folded into the convolution on either side (reference model/tsrn.py:877-886: conv -> bn -> mish -> conv -> bn)
    const float* in_scale; const float* in_shift; int in_act;    // input pixels pass through act(x * scale[c] + shift[c]) while the halo is staged
    double* stats;                                               // [work-group][2][64]: sum / sum of squares of the output per channel
};
// ---- filter re-staged through LDS per tap (any multiple of 64 input channels) ------------------------------------------------
// Persistent work-groups; the [co][CK ci] filter slice of each tap and 1/9 of the next halo are prefetched global -> registers
// -> LDS under the current tap's MFMAs (double-buffered).  Both operands are stored with the CONTRACTION axis contiguous
// (halo [row][px][CK+4], filter slice [co][CK+4]; the pitch keeps ds_read_b128 conflict-free): each lane reads 4 consecutive
// input channels per 16-byte load and feeds 4 MFMAs with it -- lane (i, kq = lane>>5) uses channels 8c + 4kq + u for u = 0..3
// on BOTH operands, which is all the contraction needs (8x fewer LDS instructions than 4-byte operand reads).
// Template CK = input channels per work item: 64 (one work-group per CU, 142.5 KB LDS), 32 (75.5 KB => two per CU, default) or 16.
template <int CK>
struct C5 {
    static constexpr int XP = CK + 4;                     // pitch: 16-byte aligned rows, conflict-free ds_read_b128
    static constexpr int HALO = 3 * C3_HW * XP;
    static constexpr int WT = 64 * XP;
    static constexpr int LDS = (2 * HALO + 2 * WT) * 4;   // CK=64: 142,528 B; CK=32: 75,456 B
    static constexpr int Q4 = CK / 4;                     // float4 per pixel
    static constexpr int SLICE = 22 * Q4;                 // float4 of the next halo fetched per tap
    static constexpr int WF4 = 64 * Q4;                   // float4 of one filter slice
};
template <int CK>
__global__ __launch_bounds__(256, (CK == 64 ? 1 : (CK == 32 ? 2 : 3))) void conv3_c64_fwd_v5_kernel(Conv3P p) {
    constexpr int C5_XP = C5<CK>::XP, C5_HALO = C5<CK>::HALO, C5_WT = C5<CK>::WT, Q4 = C5<CK>::Q4;
    constexpr int SLICE = C5<CK>::SLICE, WF4 = C5<CK>::WF4, QS = (CK == 64 ? 4 : 5);   // log2(Q4)+... see idx split below
    constexpr int Q4S = (CK == 64 ? 4 : (CK == 32 ? 3 : 2));   // log2(Q4)
    (void)QS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* XsB = smem;                                   // [2][3][66][68]
    float* WsB = smem + 2 * C5_HALO;                     // [2][64 co][68]   (filter slice, ci contiguous)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int segs = p.W / C3_PX, cob = p.Cout / 64, nch = p.Cin / CK;
    const int ntiles = p.B * p.H * segs * cob;
    const int G = gridDim.x;
    auto decode = [&](int tile, int& n, int& h, int& w0, int& co0) {
        int bid = tile;
        const int cb = bid % cob; bid /= cob;
        const int seg = bid % segs; bid /= segs;
        h = bid % p.H; n = bid / p.H;
        w0 = seg * C3_PX; co0 = cb * 64;
    };
    // float4 #idx (c4 = idx % Q4, pixel = idx / Q4) of halo row r, pixels [pxb, pxb+22)
    auto halo_load = [&](int n, int hh, int w0, int ci0, int pxb, int idx) -> f32x4 {
        const int ww = w0 + pxb + (idx >> Q4S) - 1;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (hh >= 0 && hh < p.H && ww >= 0 && ww < p.W)
            v = *reinterpret_cast<const f32x4*>(p.x + (((long)n * p.H + hh) * p.W + ww) * p.Cin + ci0 + 4 * (idx & (Q4 - 1)));
        return v;
    };
    auto halo_store = [&](float* Xs, int r, int pxb, int idx, f32x4 v) {
        *reinterpret_cast<f32x4*>(Xs + (r * C3_HW + pxb + (idx >> Q4S)) * C5_XP + 4 * (idx & (Q4 - 1))) = v;
    };
    // packed filter (mode 2/3 of tatt_repack_conv_weight): wt[tap][co][ci], ci contiguous
    constexpr int WQ = WF4 / 256;                        // filter float4 per thread per tap: 4 (CK=64) or 2 (CK=32)
    f32x4 wreg[WQ];
    auto load_w = [&](int tap, int ci0, int co0) {
        const float* src = p.w + ((long)tap * p.Cout + co0) * p.Cin + ci0;
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int idx = t + 256 * q;                 // co = idx / Q4, ci quad = idx % Q4
            wreg[q] = *reinterpret_cast<const f32x4*>(src + (long)(idx >> Q4S) * p.Cin + 4 * (idx & (Q4 - 1)));
        }
    };
    auto store_w = [&](float* Ws) {
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int idx = t + 256 * q;
            *reinterpret_cast<f32x4*>(Ws + (idx >> Q4S) * C5_XP + 4 * (idx & (Q4 - 1))) = wreg[q];
        }
    };
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    int n, h, w0, co0;
    decode(tile, n, h, w0, co0);
    // ---- prologue ----
    for (int s9 = 0; s9 < 9; ++s9) {
        const int r = s9 / 3, pxb = (s9 - 3 * r) * 22;
        for (int i = t; i < SLICE; i += 256) halo_store(XsB, r, pxb, i, halo_load(n, h + r - 1, w0, 0, pxb, i));
    }
    load_w(0, 0, co0);
    store_w(WsB);
    __syncthreads();

    int xbuf = 0, wbuf = 0, ch = 0;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    while (true) {
        int ntile = tile, nchk = ch + 1;
        if (nchk == nch) { nchk = 0; ntile = tile + G; }
        const bool has_next = ntile < ntiles;
        int nn = n, nh = h, nw0 = w0, nco0 = co0;
        if (has_next && nchk == 0) decode(ntile, nn, nh, nw0, nco0);
        const float* Xs = XsB + xbuf * C5_HALO;
        float* XsN = XsB + (xbuf ^ 1) * C5_HALO;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            // ---- issue the loads that the MFMAs below will hide: next filter slice + 1/9 of the next halo ----
            const bool more_w = tap < 8 || has_next;
            if (tap < 8) load_w(tap + 1, ch * CK, co0);
            else if (has_next) load_w(0, nchk * CK, nco0);
            const int r = tap / 3, pxb = (tap - 3 * r) * 22;      // this tap's slice of the NEXT halo: row r, 22 pixels
            f32x4 h0 = (f32x4){0.f, 0.f, 0.f, 0.f}, h1 = h0;
            if (has_next) {
                if (t < SLICE) h0 = halo_load(nn, nh + r - 1, nw0, nchk * CK, pxb, t);
                if (SLICE > 256 && t < SLICE - 256) h1 = halo_load(nn, nh + r - 1, nw0, nchk * CK, pxb, 256 + t);
            }
            // ---- 32 MFMAs fed by 8 + 8 sixteen-byte LDS reads, kept two steps ahead ----
            const int kh = r, kw = tap - 3 * r;
            const float* arow = Xs + (kh * C3_HW + wm * 32 + (lane & 31) + kw) * C5_XP + 4 * (lane >> 5);
            const float* brow = WsB + wbuf * C5_WT + (wn * 32 + (lane & 31)) * C5_XP + 4 * (lane >> 5);
            f32x4 va[CK / 8], vb[CK / 8];
#define C5_LD(c) va[c] = *reinterpret_cast<const f32x4*>(arow + 8 * (c)); vb[c] = *reinterpret_cast<const f32x4*>(brow + 8 * (c));
#define C5_MM(c) _Pragma("unroll") for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[c][u], vb[c][u], acc, 0, 0, 0);
            if constexpr (CK == 16) {
                C5_LD(0) C5_LD(1)
                __builtin_amdgcn_sched_barrier(0);
                C5_MM(0) C5_MM(1)
                __builtin_amdgcn_sched_barrier(0);
            } else {
                C5_LD(0) C5_LD(1)
                __builtin_amdgcn_sched_barrier(0);
                C5_LD(2) C5_LD(3)
                __builtin_amdgcn_sched_barrier(0);
                C5_MM(0) C5_MM(1)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (CK == 64) {
                    C5_LD(4) C5_LD(5)
                    __builtin_amdgcn_sched_barrier(0);
                }
                C5_MM(2) C5_MM(3)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (CK == 64) {
                    C5_LD(6) C5_LD(7)
                    __builtin_amdgcn_sched_barrier(0);
                    C5_MM(4) C5_MM(5)
                    __builtin_amdgcn_sched_barrier(0);
                    C5_MM(6) C5_MM(7)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- publish the prefetched data ----
            if (more_w) store_w(WsB + (wbuf ^ 1) * C5_WT);
            if (has_next) {
                if (t < SLICE) halo_store(XsN, r, pxb, t, h0);
                if (SLICE > 256 && t < SLICE - 256) halo_store(XsN, r, pxb, 256 + t, h1);
            }
            __syncthreads();
            wbuf ^= 1;
        }
        if (ch == nch - 1) {
            const int co = co0 + wn * 32 + (lane & 31);
            const float bj = p.bias ? p.bias[co] : 0.f;
            const long rowbase = ((long)n * p.H + h) * p.W + w0 + wm * 32;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int px = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                float v = apply_act(acc[reg] + bj, p.act);
                const long o = (rowbase + px) * p.Cout + co;
                if (p.beta != 0.f) v += p.beta * p.y[o];
                p.y[o] = v;
                acc[reg] = 0.f;
            }
        }
        if (!has_next) break;
        tile = ntile; ch = nchk; n = nn; h = nh; w0 = nw0; co0 = nco0;
        xbuf ^= 1;
    }
}

// x (B,H,W,Cin) NHWC contiguous; wt = filter packed [9][Cout][Cin] (tatt_repack_conv_weight mode 2; mode 3 for the data
// gradient); y (B,H,W,Cout)
TATT_API int tatt_conv3_c64_fwd_t(const float* x, const float* wt, const float* bias, float* y, int B, int H, int W, int Cin,
                                  int Cout, int act, float beta, hipStream_t st) {
    if (Cin % 64 || Cout % 64 || W % C3_PX) return 1;
    Conv3P p = {x, wt, bias, y, B, H, W, Cin, Cout, act, beta};
    static TattPerDevice attr_once;                 // once per device, under the site lock (common.h)
    tatt_per_device(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_fwd_v5_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  C5<32>::LDS);
    });
    // 32 input channels per work item: 75.5 KB of LDS, two work-groups per CU (64: one per CU and 16: three per CU measured slower)
    const int ntiles = B * H * (W / C3_PX) * (Cout / 64);
    const int G = ntiles < 512 ? ntiles : 512;
    hipLaunchKernelGGL((conv3_c64_fwd_v5_kernel<32>), dim3(G), dim3(256), C5<32>::LDS, st, p);
    return LAUNCH_CHECK();
}

// ---- weight-stationary forward, 16 output channels per wave (Cin == 64) ----------------------------------------------------------
// WEIGHT-STATIONARY: the filter lives in registers for the lifetime of a persistent work-group (one per CU, 8 waves), the loop
// streams activations through a double-buffered LDS halo.  The 64 px x 64 co tile is cut along the OUTPUT CHANNELS: wave (pxh, cq)
// owns 32 pixels x 16 channels and contracts all 9 x 64 input channels itself on v_mfma_f32_16x16x4_f32 (144 filter registers per
// lane, two waves per SIMD).  There is no partial sum to exchange, so the epilogue is nothing but stores (they retire under the next
// tile's MFMAs) and ONE barrier per tile is left (the halo hand-off).  (Rounds 1-2 also carried a variant with 32 x 32 blocks and
// the contraction split over a wave pair: its two barriers + LDS exchange per tile stalled all eight waves at once, 24 k of a
// launch's 79 k resident cycles, profiles/r01_step10_pmc_sq_counters.txt; removed in round 3.)
//   A operand (activations): lane (i = lane & 15, kq = lane >> 4) reads 4 consecutive input channels 16 g + 4 kq + u of pixel i
//     with one ds_read_b128 and feeds MFMA u of group (tap, g) with element u -- the k-slot kq of that MFMA then stands for input
//     channel 16 g + 4 kq + u on BOTH operands (any bijection slot -> channel is a valid contraction order).
//   halo pitch 72 floats: pixel i -> 16-byte slot 2 i (mod 16), k-slot kq -> +kq: the lane groups ds_read_b128 is served in
//     ({0-3,12-15,20-27}, ...) touch 16 distinct slots -- conflict-free (a pitch of 68 is 2-way on two of four groups).
#define W16_XP 72
#define W16_HALO (3 * C3_HW * W16_XP)                        // floats per halo buffer (57.0 KB)
#define W16_LDS (2 * W16_HALO * 4)
__global__ __launch_bounds__(512, 1) void conv3_c64_ws16_kernel(Conv3P p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cq = wave & 3, pxh = wave >> 2;
    const int segs = p.W / C3_PX, cob = p.Cout / 64;
    const int npt = p.B * p.H * segs;
    // XCD-aware tile order (see conv3_c64_ws_kernel)
    const int stride = gridDim.x / cob;
    int cb = blockIdx.x % cob, pt = blockIdx.x / cob;
    if (gridDim.x % 8 == 0 && 8 % cob == 0 && stride % (8 / cob) == 0) {
        // XCD-aware tile order: work-groups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8); give each XCD a contiguous
        // run of rows so that a row's halo is fetched by one L2 only (round 1: 37.8 MB fetched for a 12.6 MB input otherwise)
        const int x = blockIdx.x & 7, m = blockIdx.x >> 3, xg = 8 / cob;
        cb = x % cob;
        pt = (x / cob) * (stride / xg) + m;
    }
    __shared__ float st_red[2][2][64];                       // [pixel half][sum, sum of squares][channel]
    if (pt >= npt) {
        if (p.stats && t < 128) p.stats[(long)blockIdx.x * 128 + t] = 0.0;
        return;
    }
    const int co0 = cb * 64 + cq * 16;
    // BatchNorm + activation of the PRODUCER applied while the halo is staged: this thread always stages channels 4 (t & 15) ..
    f32x4 isc = (f32x4){1.f, 1.f, 1.f, 1.f}, ish = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.in_scale) {
        isc = *reinterpret_cast<const f32x4*>(p.in_scale + 4 * (t & 15));
        ish = *reinterpret_cast<const f32x4*>(p.in_shift + 4 * (t & 15));
    }
    float st_s = 0.f, st_q = 0.f;                            // statistics of output channel co0 + (lane & 15) over this lane's pixels
    f32x4 wq[36];                                             // [tap * 4 + g][u]: input channel 16 g + 4 (lane >> 4) + u, output channel co0 + (lane & 15)
    {
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(p.w) + (long)(cb * 4 + cq) * 36 * 64 + lane;
#pragma unroll
        for (int q = 0; q < 36; ++q) wq[q] = wsrc[q * 64];
    }
    auto decode = [&](int tile, int& n, int& h, int& w0) {
        const int seg = tile % segs; tile /= segs;
        h = tile % p.H; n = tile / p.H; w0 = seg * C3_PX;
    };
    // halo float4 #idx of 3168: c4 = idx & 15, pixel = (idx >> 4) % 66, row = (idx >> 4) / 66
    auto halo_load = [&](int n, int h, int w0, int idx) -> f32x4 {
        const int pix = idx >> 4, r = pix / C3_HW, px = pix - r * C3_HW;
        const int hh = h + r - 1, ww = w0 + px - 1;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (idx < 3 * C3_HW * 16 && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W) {
            v = *reinterpret_cast<const f32x4*>(p.x + (((long)n * p.H + hh) * p.W + ww) * 64 + 4 * (idx & 15));
            if (p.in_scale) {                                // (the zero padding stays zero: it pads the TRANSFORMED map)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = fmaf(v[e], isc[e], ish[e]);
                    v[e] = p.in_act == ACT_MISH ? mish_f(u) : (p.in_act == ACT_RELU ? fmaxf(u, 0.f) : u);
                }
            }
        }
        return v;
    };
    auto halo_store = [&](float* Xs, int idx, f32x4 v) {
        if (idx < 3 * C3_HW * 16) *reinterpret_cast<f32x4*>(Xs + (idx >> 4) * W16_XP + 4 * (idx & 15)) = v;
    };
    int n, h, w0;
    decode(pt, n, h, w0);
    {
        f32x4 hp[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) hp[q] = halo_load(n, h, w0, t + 512 * q);
#pragma unroll
        for (int q = 0; q < 7; ++q) halo_store(smem, t + 512 * q, hp[q]);
    }
    __syncthreads();
    const int abase = (pxh * 32 + (lane & 15)) * W16_XP + 4 * (lane >> 4);
    const float bj = p.bias ? p.bias[co0 + (lane & 15)] : 0.f;
    int xbuf = 0;
    while (true) {
        const int npt_next = pt + stride;
        const bool has_next = npt_next < npt;
        int nn = n, nh = h, nw0 = w0;
        if (has_next) decode(npt_next, nn, nh, nw0);
        const float* Xs = smem + xbuf * W16_HALO + abase;
        float* XsN = smem + (xbuf ^ 1) * W16_HALO;
        f32x4 acc[2];
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 va[3][2];                                      // ring: the reads of group l + 2 are issued before the MFMAs of group l
        f32x4 hq = (f32x4){0.f, 0.f, 0.f, 0.f};
        // group l = tap * 4 + g; M-tile m: pixels pxh * 32 + 16 m + i
#define W16_AOFF(l, m) ((((l) >> 2) / 3 * C3_HW + ((l) >> 2) % 3 + 16 * (m)) * W16_XP + 16 * ((l) & 3))
#define W16_LD(l) va[(l) % 3][0] = *reinterpret_cast<const f32x4*>(Xs + W16_AOFF(l, 0)); \
                  va[(l) % 3][1] = *reinterpret_cast<const f32x4*>(Xs + W16_AOFF(l, 1));
        W16_LD(0) W16_LD(1)
#pragma unroll
        for (int l = 0; l < 36; ++l) {
            if (l % 5 == 0 && has_next) hq = halo_load(nn, nh, nw0, t + 512 * (l / 5));       // 7 rounds: l = 0,5,...,30
            if (l + 2 < 36) { W16_LD(l + 2) }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[l % 3][0][u], wq[l][u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[l % 3][1][u], wq[l][u], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (l % 5 == 4 && has_next) halo_store(XsN, t + 512 * (l / 5), hq);
        }
        // ---- epilogue: C layout of the 16x16 tile: row (pixel) = 4 (lane >> 4) + reg, column (channel) = lane & 15.  Each
        // 16-lane group stores 64 contiguous bytes of one pixel; the four waves of a pixel half complete its 256-byte row. ----
        {
            const long rowbase = ((long)n * p.H + h) * p.W + w0 + pxh * 32 + 4 * (lane >> 4);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* dst = p.y + (rowbase + 16 * m + r) * p.Cout + co0 + (lane & 15);
                    float v = apply_act(acc[m][r] + bj, p.act);
                    if (p.beta != 0.f) v += p.beta * *dst;
                    *dst = v;
                    st_s += v; st_q += v * v;
                }
        }
        if (!has_next) break;
        __syncthreads();                                     // the next halo is complete; every wave has left the current one
        pt = npt_next; n = nn; h = nh; w0 = nw0;
        xbuf ^= 1;
    }
    if (p.stats) {
        // per-channel sum / sum of squares of everything this work-group wrote: lanes of a channel, then the two pixel halves
        st_s += __shfl_xor(st_s, 16, 64); st_s += __shfl_xor(st_s, 32, 64);
        st_q += __shfl_xor(st_q, 16, 64); st_q += __shfl_xor(st_q, 32, 64);
        if (lane < 16) { st_red[pxh][0][cq * 16 + lane] = st_s; st_red[pxh][1][cq * 16 + lane] = st_q; }
        __syncthreads();
        if (t < 128) p.stats[(long)blockIdx.x * 128 + t] = (double)st_red[0][t >> 6][t & 63] + (double)st_red[1][t >> 6][t & 63];
    }
}

TATT_API int tatt_conv3_c64_fwd_ws16_bn(const float* x, const float* wl, const float* bias, float* y, int B, int H, int W,
                                        int Cout, int act, float beta, const float* in_scale, const float* in_shift, int in_act,
                                        double* stats, hipStream_t st);
// wl = filter from tatt_repack_conv_weight mode 6 (forward) / mode 7 (data gradient of a 64-output-channel convolution)
TATT_API int tatt_conv3_c64_fwd_ws16(const float* x, const float* wl, const float* bias, float* y, int B, int H, int W,
                                     int Cout, int act, float beta, hipStream_t st) {
    return tatt_conv3_c64_fwd_ws16_bn(x, wl, bias, y, B, H, W, Cout, act, beta, nullptr, nullptr, 0, nullptr, st);
}
// The same convolution with BatchNorm folded in on either side:
//   in_scale / in_shift (64 floats each, nullable): every input pixel passes through in_act(x * in_scale[c] + in_shift[c]) first
//     (BatchNorm + activation of the producing layer, applied while the halo is staged; zero padding pads the transformed map);
//   stats (nullable; Cout == 64, act == none, beta == 0): [grid][2][64] doubles -- per work-group sum and sum of squares of the
//     output per channel, the stage-1 partials of tatt_bn_stats_finish; grid = min(256, B*H*W/64) work-groups.
TATT_API int tatt_conv3_c64_fwd_ws16_bn(const float* x, const float* wl, const float* bias, float* y, int B, int H, int W,
                                        int Cout, int act, float beta, const float* in_scale, const float* in_shift, int in_act,
                                        double* stats, hipStream_t st) {
    if (Cout % 64 || W % C3_PX) return 1;
    if (stats && (Cout != 64 || act != ACT_NONE || beta != 0.f)) return 2;
    if (in_scale && (!in_shift || in_act == ACT_TANH)) return 3;
    Conv3P p = {x, wl, bias, y, B, H, W, 64, Cout, act, beta, in_scale, in_shift, in_act, stats};
    static TattPerDevice attr_once;                 // once per device, under the site lock (common.h)
    tatt_per_device(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_ws16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W16_LDS);
    });
    const int cob = Cout / 64, npt = B * H * (W / C3_PX);
    int per = 256 / cob;
    if (per > npt) per = npt;
    hipLaunchKernelGGL(conv3_c64_ws16_kernel, dim3(per * cob), dim3(512), W16_LDS, st, p);
    return LAUNCH_CHECK();
}

// ---- weight-stationary forward on the bf16 matrix cores: split-bf16 (hi + lo), three products per fp32 product -------------------
// bf16 MFMA runs at 16x the fp32-MFMA rate on gfx950.  Every fp32 operand is split a = hi + lo, hi = bf16(a), lo = bf16(a - hi)
// (16 mantissa bits kept), and  a * b  is evaluated as  hi_a hi_b + (hi_a lo_b + lo_a hi_b)  with fp32 accumulation: the dropped
// lo_a lo_b term is 2^-16 relative.  Measured on the CPU oracle (tools/split_bf16_probe.py, profiles/r03_split_bf16_probe.txt): the
// eval SR moves by 1.1e-6 (fp32 itself is 2.6e-7 from fp64; parity bar 1e-3, test bound 2e-5) and every training gradient stays inside
// the fp64 yardstick (worst ratio 0.18 of the test limit).  Same organisation as conv3_c64_ws16_kernel: wave (pxh, cq) owns 32 px x 16 co,
// the filter (hi and lo, 144 registers per lane again) lives in registers, activations stream through a double-buffered LDS halo that
// is split into a hi and a lo bf16 image while it is staged (two v_cvt_pk_bf16_f32 + one subtraction per pair of values).
//   v_mfma_f32_16x16x32_bf16: lane (i = lane & 15, kq = lane >> 4) supplies 8 consecutive k of row i; k-step ks = (tap, half):
//   input channels 32 half + 8 kq .. + 7 of pixel i  =  ONE ds_read_b128 per operand image.
//   halo pitch 160 B per pixel (128 B of channels + 32): pixel i -> 16-byte slot 10 i, k-slot kq -> + kq: the lane groups a
//   ds_read_b128 is served in touch 16 distinct slots (conflict-free).
// Handles a 64-channel slice [ci0, ci0 + 64) of a wider input (cin_total): wider contractions are chunked by the host (beta = 1).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define SB_PW 40                                             // halo pitch in 32-bit words (two bf16 each)
#define SB_IMG (3 * C3_HW * SB_PW)                           // words of one image (hi or lo) of one halo buffer: 7920
#define SB_LDS (4 * SB_IMG * 4)                              // two buffers x (hi, lo): 126,720 B
// Round 4, backward of conv -> bn -> mish -> conv -> bn (model/tsrn.py:877-886) without the BatchNorm-backward passes over the maps:
//   IN2:  the kernel's input is the BatchNorm backward of the NEXT layer applied on the fly: two maps are staged per pixel and
//         combined per channel, v = x * in_scale + x2 * in_scale2 + in_shift  (x = upstream gradient du, x2 = that BatchNorm's input y;
//         dy = gamma rstd (du - mean(du) - xhat mean(du xhat)) is affine in (du, y) once the two means are known: tatt_bn_bwd_finish);
//   EPBN: the kernel's output is the gradient w.r.t. act(bn(ep_x)): the epilogue multiplies by act'(gamma xhat + beta) (xhat from ep_x
//         and that BatchNorm's batch statistics), stores du, and leaves the per-work-group partial sums of du and du * xhat in `stats`
//         (the stage-1 partials of THAT BatchNorm's backward) -- the mirror of the forward's epilogue statistics.
struct Conv3SB {
    Conv3P c; int cin_total, ci0;
    const float* x2; const float* in_scale2;                                            // IN2
    const float* ep_x; const float* ep_mean; const float* ep_rstd; const float* ep_gamma; const float* ep_beta; int ep_act;   // EPBN
};
__device__ __forceinline__ void sb_split(f32x4 v, uint2& hi, uint2& lo) {
    // scalar arithmetic on purpose (and the file is built with -fno-slp-vectorize): packed fp32 VALU forms (v_pk_add_f32 ...) cost the
    // MFMA waves beside them issue time (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
    const bf16x2 h0 = __builtin_convertvector((f32x2){v[0], v[1]}, bf16x2), h1 = __builtin_convertvector((f32x2){v[2], v[3]}, bf16x2);
    const unsigned u0 = __builtin_bit_cast(unsigned, h0), u1 = __builtin_bit_cast(unsigned, h1);
    const float r0 = v[0] - __builtin_bit_cast(float, u0 << 16), r1 = v[1] - __builtin_bit_cast(float, u0 & 0xffff0000u);
    const float r2 = v[2] - __builtin_bit_cast(float, u1 << 16), r3 = v[3] - __builtin_bit_cast(float, u1 & 0xffff0000u);
    const bf16x2 l0 = __builtin_convertvector((f32x2){r0, r1}, bf16x2), l1 = __builtin_convertvector((f32x2){r2, r3}, bf16x2);
    hi = make_uint2(u0, u1);
    lo = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
}
template <bool IN2, bool EPBN>
__global__ __launch_bounds__(512, 1) void conv3_c64_sb_kernel(Conv3SB q) {
    const Conv3P& p = q.c;
    extern __shared__ __attribute__((aligned(16))) unsigned smem_u[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cq = wave & 3, pxh = wave >> 2;
    const int segs = p.W / C3_PX, cob = p.Cout / 64;
    const int npt = p.B * p.H * segs;
    const int stride = gridDim.x / cob;
    int cb = blockIdx.x % cob, pt = blockIdx.x / cob;
    if (gridDim.x % 8 == 0 && 8 % cob == 0 && stride % (8 / cob) == 0) {      // XCD-aware tile order (see conv3_c64_ws_kernel)
        const int x = blockIdx.x & 7, m = blockIdx.x >> 3, xg = 8 / cob;
        cb = x % cob;
        pt = (x / cob) * (stride / xg) + m;
    }
    __shared__ float st_red[2][2][64];
    if (pt >= npt) {
        if (p.stats && t < 128) p.stats[(long)blockIdx.x * 128 + t] = 0.0;
        return;
    }
    const int co0 = cb * 64 + cq * 16;
    f32x4 isc = (f32x4){1.f, 1.f, 1.f, 1.f}, ish = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.in_scale) {
        isc = *reinterpret_cast<const f32x4*>(p.in_scale + 4 * (t & 15));
        ish = *reinterpret_cast<const f32x4*>(p.in_shift + 4 * (t & 15));
    }
    f32x4 isc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (IN2) isc2 = *reinterpret_cast<const f32x4*>(q.in_scale2 + 4 * (t & 15));
    float ep_mu = 0.f, ep_rs = 0.f, ep_g = 0.f, ep_b = 0.f;
    if (EPBN) {
        const int c = co0 + (lane & 15);
        ep_mu = q.ep_mean[c]; ep_rs = q.ep_rstd[c]; ep_g = q.ep_gamma[c]; ep_b = q.ep_beta[c];
    }
    float st_s = 0.f, st_q = 0.f;
    f32x4 wq[36];                                            // [kstep * 2 + {hi, lo}]: 8 bf16 = input channels 32 half + 8 (lane >> 4) .. of output channel co0 + (lane & 15)
    {
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(p.w) + (long)(cb * 4 + cq) * 36 * 64 + lane;
#pragma unroll
        for (int k = 0; k < 36; ++k) wq[k] = wsrc[k * 64];
    }
    auto decode = [&](int tile, int& n, int& h, int& w0) {
        const int seg = tile % segs; tile /= segs;
        h = tile % p.H; n = tile / p.H; w0 = seg * C3_PX;
    };
    auto halo_load = [&](int n, int h, int w0, int idx) -> f32x4 {
        const int pix = idx >> 4, r = pix / C3_HW, px = pix - r * C3_HW;
        const int hh = h + r - 1, ww = w0 + px - 1;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (idx < 3 * C3_HW * 16 && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W) {
            v = *reinterpret_cast<const f32x4*>(p.x + (((long)n * p.H + hh) * p.W + ww) * q.cin_total + q.ci0 + 4 * (idx & 15));
            if (IN2) {                                       // (both maps hold exactly 64 channels per pixel)
                const f32x4 v2 = *reinterpret_cast<const f32x4*>(q.x2 + (((long)n * p.H + hh) * p.W + ww) * 64 + 4 * (idx & 15));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], isc[e], fmaf(v2[e], isc2[e], ish[e]));
            } else if (p.in_scale) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = fmaf(v[e], isc[e], ish[e]);
                    v[e] = p.in_act == ACT_MISH ? mish_f(u) : (p.in_act == ACT_RELU ? fmaxf(u, 0.f) : u);
                }
            }
        }
        return v;
    };
    auto halo_store = [&](unsigned* Xs, int idx, f32x4 v) {    // Xs: hi image; the lo image follows SB_IMG words later
        if (idx < 3 * C3_HW * 16) {
            uint2 hi, lo;
            sb_split(v, hi, lo);
            unsigned* d = Xs + (idx >> 4) * SB_PW + 2 * (idx & 15);
            *reinterpret_cast<uint2*>(d) = hi;
            *reinterpret_cast<uint2*>(d + SB_IMG) = lo;
        }
    };
    int n, h, w0;
    decode(pt, n, h, w0);
    {
        f32x4 hp[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) hp[k] = halo_load(n, h, w0, t + 512 * k);
#pragma unroll
        for (int k = 0; k < 7; ++k) halo_store(smem_u, t + 512 * k, hp[k]);
    }
    __syncthreads();
    const int abase = (pxh * 32 + (lane & 15)) * SB_PW + 4 * (lane >> 4);
    const float bj = p.bias ? p.bias[co0 + (lane & 15)] : 0.f;
    int xbuf = 0;
    while (true) {
        const int npt_next = pt + stride;
        const bool has_next = npt_next < npt;
        int nn = n, nh = h, nw0 = w0;
        if (has_next) decode(npt_next, nn, nh, nw0);
        const unsigned* Xs = smem_u + xbuf * 2 * SB_IMG + abase;
        unsigned* XsN = smem_u + (xbuf ^ 1) * 2 * SB_IMG;
        f32x4 accM[2], accC[2];                              // hi*hi products; cross products (hi*lo + lo*hi), summed separately
#pragma unroll
        for (int m = 0; m < 2; ++m) { accM[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; accC[m] = accM[m]; }
        f32x4 va[2][4];                                      // ring: [.][2 m + {hi, lo}]; reads of k-step ks + 1 are issued before the MFMAs of ks
        f32x4 hq = (f32x4){0.f, 0.f, 0.f, 0.f};
        // k-step ks = tap * 2 + half; M-tile m: pixels pxh * 32 + 16 m + i
#define SB_AOFF(ks, m) ((((ks) >> 1) / 3 * C3_HW + ((ks) >> 1) % 3 + 16 * (m)) * SB_PW + 16 * ((ks) & 1))
#define SB_LD(ks) va[(ks) % 2][0] = *reinterpret_cast<const f32x4*>(Xs + SB_AOFF(ks, 0)); \
                  va[(ks) % 2][1] = *reinterpret_cast<const f32x4*>(Xs + SB_AOFF(ks, 0) + SB_IMG); \
                  va[(ks) % 2][2] = *reinterpret_cast<const f32x4*>(Xs + SB_AOFF(ks, 1)); \
                  va[(ks) % 2][3] = *reinterpret_cast<const f32x4*>(Xs + SB_AOFF(ks, 1) + SB_IMG);
        SB_LD(0)
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
            if (ks % 2 == 0 && ks / 2 < 7 && has_next) hq = halo_load(nn, nh, nw0, t + 512 * (ks / 2));
            if (ks + 1 < 18) { SB_LD(ks + 1) }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 wh = __builtin_bit_cast(bf16x8, wq[2 * ks]), wl = __builtin_bit_cast(bf16x8, wq[2 * ks + 1]);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const bf16x8 ah = __builtin_bit_cast(bf16x8, va[ks % 2][2 * m]), al = __builtin_bit_cast(bf16x8, va[ks % 2][2 * m + 1]);
                accM[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh, accM[m], 0, 0, 0);
                accC[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl, accC[m], 0, 0, 0);
                accC[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh, accC[m], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ks % 2 == 1 && ks / 2 < 7 && has_next) halo_store(XsN, t + 512 * (ks / 2), hq);
        }
        {
            const long rowbase = ((long)n * p.H + h) * p.W + w0 + pxh * 32 + 4 * (lane >> 4);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* dst = p.y + (rowbase + 16 * m + r) * p.Cout + co0 + (lane & 15);
                    float v = apply_act((accM[m][r] + accC[m][r]) + bj, p.act);
                    if (p.beta != 0.f) v += p.beta * *dst;
                    if (EPBN) {                              // v = gradient w.r.t. act(bn(ep_x)); Cout == 64
                        const float xh = (q.ep_x[(rowbase + 16 * m + r) * 64 + co0 + (lane & 15)] - ep_mu) * ep_rs;
                        if (q.ep_act != ACT_NONE) v *= act_grad(fmaf(ep_g, xh, ep_b), q.ep_act);
                        *dst = v;
                        st_s += v; st_q += v * xh;
                    } else {
                        *dst = v;
                        st_s += v; st_q += v * v;
                    }
                }
        }
        if (!has_next) break;
        __syncthreads();
        pt = npt_next; n = nn; h = nh; w0 = nw0;
        xbuf ^= 1;
    }
    if (p.stats) {
        st_s += __shfl_xor(st_s, 16, 64); st_s += __shfl_xor(st_s, 32, 64);
        st_q += __shfl_xor(st_q, 16, 64); st_q += __shfl_xor(st_q, 32, 64);
        if (lane < 16) { st_red[pxh][0][cq * 16 + lane] = st_s; st_red[pxh][1][cq * 16 + lane] = st_q; }
        __syncthreads();
        if (t < 128) p.stats[(long)blockIdx.x * 128 + t] = (double)st_red[0][t >> 6][t & 63] + (double)st_red[1][t >> 6][t & 63];
    }
}
// ---- round 6: the same arithmetic on square tiles -------------------------------------------------------------------------------------------
// What bound the kernel above (profiles/r06_pmc_conv3_sb_old_and_g2.txt): every tile's next halo was fetched as SEVEN dependent global-load
// -> LDS-store round trips spread over its k-steps (one load, waited for one k-step later), the 144-register filter copy was loaded twice
// per work-group (pixel halves), a one-row tile fetched and transformed (BatchNorm + mish + split) every input pixel three times, and
// the epilogue left as 4-byte stores.  Common to the kernels below:
//   * tile = 4 rows x 16 pixels (halo 6 x 18 = 108 pixels: 1.69 pixels staged per output pixel instead of 3.09);
//   * the contraction is split over wave pairs (kh = input-channel half): half the filter registers per wave, no duplicate filter loads;
//     the halves meet through an LDS exchange image [kh][pixel][channel] in front of the tile's only barrier;
//   * the next tile's halo and the epilogue's own inputs are requested a tile ahead with buffer loads (32-bit offsets; the zero padding is
//     the descriptor's out-of-range answer: no divergent branches, no 64-bit address math) and staged into the other LDS buffer;
//   * the MFMA operands are swapped (A = filter, B = activations): a lane ends up with consecutive output channels of ONE pixel, and the
//     epilogue is re-threaded over the exchange image: 16 lanes finish the 256 bytes of one pixel, a wave stores 1 KB of one map row per
//     instruction (whole cache lines; the same for the BatchNorm-backward map it reads).
// Same statistics layout, same entry points; geometries with H % 4 != 0 keep the row-tile kernel.
#define S2_HW 18                                             // halo width (pixels)
#define S2_NPX (6 * S2_HW)                                   // 108 halo pixels
#define S2_XP 68                                             // words per pixel of an exchange image (64 channels + 4: conflict-free 16-byte writes)
#define S2_XCH (2 * 64 * S2_XP)                              // words of one exchange buffer: [kh 2][pixel 64][S2_XP]: 34,816 B
// instrumentation, off in the product build: S2_STAMP 1 = s_memtime stamps of the generation-3 kernel (tools/conv3_stamps.py),
// S2_DEBUG 1 = the role switches of the generation-4 kernel (tools/conv3_sweep.py <gens> <mode>; profiles/r06_conv3_sb4_roles.txt)
#define S2_STAMP 0
#define S2_DEBUG 0
#if S2_DEBUG
__device__ int s2_wrep = 0;
extern "C" __attribute__((visibility("default"))) int tatt_conv3_debug_wrep(int r) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(s2_wrep), &r, sizeof(r)); }
#endif
#if S2_STAMP
__device__ unsigned long long* s2_stamp_buf = nullptr;
extern "C" __attribute__((visibility("default"))) int tatt_conv3_debug_stamps(unsigned long long* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(s2_stamp_buf), &buf, sizeof(buf)); }
#define S2_STAMP_AT(i) ts[i] = __builtin_amdgcn_s_memtime();
#else
#define S2_STAMP_AT(i)
#endif
#define S2_OOB 0x80000000u                                   // a byte offset beyond every descriptor's range: the load answers 0
typedef unsigned u32x4_s2 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 s2_ld(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
// INMODE 0: the input as it is; 1: mish(x * in_scale + in_shift); 2: x * in_scale + x2 * in_scale2 + in_shift; 3: x * in_scale + in_shift,
// through ReLU if in_act says so.  The staging code is written with selects, never branches (uniform conditions included); what that
// cannot express (an output activation, a tanh) runs the row-tile kernel.
// (The first cut of this rebuild -- eight waves of 16 output channels on v_mfma_f32_16x16x32_bf16, everything in every wave -- is in the
// history of this file (commit "3x3 split-bf16 convolution rebuilt"); its counters, stamps and ablation are profiles/r06_pmc_conv3_sb_old_and_g2.txt,
// r06_conv3_sb2_stamps.txt, r06_conv3_sb2_ablation.txt.  The kernels below keep its tiles, exchange image and request schedule.)
// ---- round 6, second cut: one wave per SIMD, 32 output channels per wave, v_mfma_f32_32x32x16_bf16 ----------------------------------------
// What the ablation of the kernel above showed (profiles/r06_conv3_sb2_ablation.txt, B = 48, graph-captured launches): with every
// MFMA, global load and store removed the launch still took 10.2 of 15.2 us -- 2.4 us per tile of LDS operand reads (576 16-byte
// wave-reads = 2,700 LDS cycles: every activation fragment is read by the 4 waves that own 16 output channels each), staging and
// barriers; and v_mfma_f32_16x16x32_bf16 issues at ~20 cycles per SIMD (4,320 cycles per tile) against 32 for twice the FLOPs of the
// 32x32x16 form (3,456).  Here a work-group is 4 waves (one per SIMD, up to 512 registers each): wave (coh, kh) owns all 64 pixels x 32
// output channels over half the contraction: 144 filter registers, 64 accumulator registers, every activation fragment is read by TWO
// waves (288 wave-reads per tile), 108 MFMAs of 32 cycles per tile per SIMD.  Tiles, halo staging, LDS exchange, re-threaded epilogue
// and request schedule as above.  Filter packing: modes 14 / 15 (gemm.hip).
typedef float f32x16_s3 __attribute__((ext_vector_type(16)));
#define S3_RP (S2_HW * SB_PW + 4)                            // halo row pitch (words): + 16 bytes, so that the two pixel rows a 32-lane operand read
                                                             // spans fall into disjoint LDS bank slots (odd / even 16-byte slots)
#define S3_IMG (6 * S3_RP)                                   // words of one image (hi or lo) of one halo buffer: 4344
#define S3_LDS ((4 * S3_IMG + 2 * S2_XCH) * 4)               // 139,136 B
#define S3_NIT 7                                             // 16-byte halo items per thread (1728 <= 7 x 256)
#define S3_NEP 4                                             // epilogue items per thread (64 pixels x 16 channel groups / 256)
template <int INMODE, bool EPBN>
__global__ __launch_bounds__(256, 1) void conv3_c64_sb3_kernel(Conv3SB q) {
    const Conv3P& p = q.c;
    extern __shared__ __attribute__((aligned(16))) unsigned smem_u[];
    unsigned* const XCH = smem_u + 4 * S3_IMG;
    const unsigned t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned coh = wave & 1, kh = wave >> 1, lj = lane & 31, kb = lane >> 5;
    const unsigned W = p.W, H = p.H, tws = W / 16, ths = H / 4, cob = p.Cout / 64;
    const unsigned npt = p.B * ths * tws;
    const unsigned stride = gridDim.x / cob;
    unsigned cb = blockIdx.x % cob, pt = blockIdx.x / cob;
    if (gridDim.x % 8 == 0 && 8 % cob == 0 && stride % (8 / cob) == 0) {      // XCD-aware tile order (see conv3_c64_ws_kernel)
        const unsigned x = blockIdx.x & 7, m = blockIdx.x >> 3, xg = 8 / cob;
        cb = x % cob;
        pt = (x / cob) * (stride / xg) + m;
    }
    __shared__ float st_red[4][2][64];
    if (pt >= npt) {
        if (p.stats && t < 128) p.stats[(long)blockIdx.x * 128 + t] = 0.0;
        return;
    }
#if S2_STAMP
    unsigned long long* const sbuf = s2_stamp_buf;
    unsigned long long ts[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) ts[i] = 0;
    int tix = 0;
    ts[0] = __builtin_amdgcn_s_memtime();
    ts[30] = wall_clock64();
#endif
    const unsigned npix = p.B * H * W;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, npix * q.cin_total * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(INMODE == 2 ? q.x2 : p.x), 0, npix * 256, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, npix * p.Cout * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(EPBN ? q.ep_x : p.y), 0, npix * (EPBN ? 64 : p.Cout) * 4, 0x00020000);
    __shared__ __attribute__((aligned(16))) float cst[7][64];   // in_scale, in_shift, in_scale2, ep_mean, ep_rstd, ep_gamma, ep_beta
    __shared__ __attribute__((aligned(16))) float cbias[64];    // this work-group's 64 output channels
    const bool in_relu = p.in_act == ACT_RELU;
    f32x4 st_s = (f32x4){0.f, 0.f, 0.f, 0.f}, st_q = st_s;
    const unsigned xps = q.cin_total * 4, yps = p.Cout * 4, eps_ = EPBN ? 256u : yps;     // bytes per pixel of the maps
    // halo item k of this thread: 16-byte channel group t & 15 of halo pixel (t >> 4) + 16 k (row-major 6 x 18): byte offset from the
    // tile's origin pixel, word offset in the LDS image, the tile edges at which it falls outside the map (as in the kernel above)
    unsigned boff[S3_NIT], poff[S3_NIT], edge[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < S3_NIT; ++k) {
        const unsigned pix = (t >> 4) + 16 * k, r = pix / S2_HW, c = pix - r * S2_HW;
        boff[k] = (unsigned)(((int)r - 1) * (int)W + (int)c - 1) * xps + (q.ci0 + 4 * (t & 15)) * 4;
        poff[k] = r * S3_RP + c * SB_PW + 2 * (t & 15);
        edge[k >> 2] |= ((r == 0 ? 1u : 0u) | (r == 5 ? 2u : 0u) | (c == 0 ? 4u : 0u) | (c == S2_HW - 1 ? 8u : 0u) | (pix >= S2_NPX ? 16u : 0u)) << (8 * (k & 3));
    }
    auto decode = [&](unsigned tile, unsigned& org, unsigned& te) {                     // -> origin pixel index, edge mask of the tile
        const unsigned tw = tile % tws; tile /= tws;
        const unsigned th = tile % ths, n = tile / ths;
        org = (n * H + th * 4) * W + tw * 16;
        te = (th == 0 ? 1u : 0u) | (th == ths - 1 ? 2u : 0u) | (tw == 0 ? 4u : 0u) | (tw == tws - 1 ? 8u : 0u) | 16u;
    };
    f32x4 hp[S3_NIT], hp2[S3_NIT];
    auto halo_issue1 = [&](unsigned org, unsigned te, bool valid, int k) {   // !valid: the request is out of range (no memory traffic)
        const unsigned a = (org * xps + boff[k]) | ((((edge[k >> 2] >> (8 * (k & 3))) & te) != 0) ? S2_OOB : (valid ? 0u : S2_OOB));
        hp[k] = s2_ld(rs_x, a);
        if (INMODE == 2) hp2[k] = s2_ld(rs_x2, a);
    };
    auto halo_put = [&](unsigned* img, unsigned te, int k) {       // transform, split, store item k into the image at `img`
        const bool ok = ((edge[k >> 2] >> (8 * (k & 3))) & te) == 0;
        f32x4 v = hp[k];
        f32x4 isc, ish;
        if (INMODE != 0) { isc = *reinterpret_cast<const f32x4*>(&cst[0][4 * (t & 15)]); ish = *reinterpret_cast<const f32x4*>(&cst[1][4 * (t & 15)]); }
        if (INMODE == 2) {
            const f32x4 isc2 = *reinterpret_cast<const f32x4*>(&cst[2][4 * (t & 15)]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], isc[e], fmaf(hp2[k][e], isc2[e], ish[e]));
        } else if (INMODE == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = mish_f(fmaf(v[e], isc[e], ish[e]));
        } else if (INMODE == 3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float u = fmaf(v[e], isc[e], ish[e]);
                v[e] = in_relu ? fmaxf(u, 0.f) : u;
            }
        }
        if (INMODE != 0) {                                   // (the zero padding is a padding of the TRANSFORMED map)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
        }
        if (k < S3_NIT - 1 || t < S2_NPX * 16 - 256 * (S3_NIT - 1)) {
            uint2 hi, lo;
            sb_split(v, hi, lo);
            unsigned* d = img + poff[k];
            *reinterpret_cast<uint2*>(d) = hi;
            *reinterpret_cast<uint2*>(d + S3_IMG) = lo;
        }
    };
    // first tile: its halo is requested before anything else (the first HBM round trip is the longest wait of the kernel)
    unsigned org, te;
    decode(pt, org, te);
#pragma unroll
    for (int k = 0; k < S3_NIT; ++k) halo_issue1(org, te, true, k);
    if (t < 64) {
        cbias[t] = p.bias ? p.bias[cb * 64 + t] : 0.f;
        if (INMODE != 0) { cst[0][t] = p.in_scale[t]; cst[1][t] = p.in_shift[t]; }
        if (INMODE == 2) cst[2][t] = q.in_scale2[t];
        if (EPBN) { cst[3][t] = q.ep_mean[t]; cst[4][t] = q.ep_rstd[t]; cst[5][t] = q.ep_gamma[t]; cst[6][t] = q.ep_beta[t]; }
    }
    __syncthreads();                                         // constants in place (the staging below reads them)
#pragma unroll
    for (int k = 0; k < S3_NIT; ++k) halo_put(smem_u, te, k);
    unsigned norg, nte;
    bool has_next = pt + stride < npt;
    decode(pt + stride, norg, nte);
#pragma unroll
    for (int k = 0; k < S3_NIT; ++k) halo_issue1(norg, nte, has_next, k);
    __syncthreads();
    S2_STAMP_AT(1)
    // The filter: [(tap * 2 + s) * 2 + {hi, lo}] = 8 bf16 = input channels 32 kh + 16 s + 8 kb .. of output channel 32 coh + lj (mode 14 / 15
    // packing).  The first tile's k-loop is peeled and requests each tap's registers three taps ahead of its MFMAs.
    f32x4 wq[36];
    const f32x4* const wsrc = reinterpret_cast<const f32x4*>(p.w) + (cb * 2 + coh) * 72 * 64 + lane;
    auto w_issue = [&](int tap) {
#pragma unroll
        for (int f = 0; f < 4; ++f) wq[4 * tap + f] = wsrc[((tap * 2 + kh) * 4 + f) * 64];
    };
    const unsigned bbase = (lj >> 4) * S3_RP + (lj & 15) * SB_PW + 16 * kh + 4 * kb;
    const bool use_beta = p.beta != 0.f;
    const bool ep_mish = q.ep_act == ACT_MISH, ep_relu = q.ep_act == ACT_RELU;
    f32x4 epx[S3_NEP];                                       // (zeroed: the first tile's idle epilogue pass multiplies by what is in here)
#pragma unroll
    for (int e = 0; e < S3_NEP; ++e) epx[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned eorg = 0, ebuf = 0;
    bool epi_pending = false;
    // epilogue item e of this thread: channels 4 (t & 15) .. + 3 of tile pixel (t >> 4) + 16 e (row e, column t >> 4)
    const unsigned ecol = t >> 4, ec4 = 4 * (t & 15);
    auto epi_request = [&](unsigned org_, int e) {           // (neither an EPBN map nor a beta: out of range, the answer is 0)
        epx[e] = s2_ld(rs_e, ((org_ + e * W + ecol) * eps_ + (cb * 64 + ec4) * 4) | ((EPBN || use_beta) ? 0u : S2_OOB));
    };
    auto epilogue = [&](bool live, int e) {                  // !live (no tile is waiting): nothing is stored, nothing is counted
        const unsigned* X = XCH + ebuf * S2_XCH + ((t >> 4) + 16 * e) * S2_XP + ec4;
        const f32x4 bj = *reinterpret_cast<const f32x4*>(&cbias[ec4]);
        f32x4 ep_mu, ep_rs, ep_g, ep_b;
        if (EPBN) {                                          // (Cout == 64)
            ep_mu = *reinterpret_cast<const f32x4*>(&cst[3][ec4]); ep_rs = *reinterpret_cast<const f32x4*>(&cst[4][ec4]);
            ep_g = *reinterpret_cast<const f32x4*>(&cst[5][ec4]); ep_b = *reinterpret_cast<const f32x4*>(&cst[6][ec4]);
        }
        f32x4 v = *reinterpret_cast<const f32x4*>(X) + *reinterpret_cast<const f32x4*>(X + 64 * S2_XP);      // the two halves of the contraction
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] += bj[r];                                   // (no output activation in this kernel)
            if (EPBN) {                                      // v = gradient w.r.t. act(bn(ep_x)); Cout == 64
                const float xh = (epx[e][r] - ep_mu[r]) * ep_rs[r], u = fmaf(ep_g[r], xh, ep_b[r]);
                const float g = ep_mish ? mish_grad_f(u) : 1.f;
                v[r] *= ep_relu ? (u > 0.f ? 1.f : 0.f) : g;
                v[r] = live ? v[r] : 0.f;
                st_s[r] += v[r]; st_q[r] += v[r] * xh;
            } else {
                v[r] = fmaf(p.beta, epx[e][r], v[r]);
                v[r] = live ? v[r] : 0.f;
                st_s[r] += v[r]; st_q[r] += v[r] * v[r];
            }
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_s2, v), rs_y,
                                               ((eorg + e * W + ecol) * yps + (cb * 64 + ec4) * 4) | (live ? 0u : S2_OOB), 0, 0);
    };
    unsigned xbuf = 0;
    bool had_next = false;
    auto tile_body = [&](auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const unsigned* Xs = smem_u + xbuf * 2 * S3_IMG + bbase;
        unsigned* imgN = smem_u + (xbuf ^ 1) * 2 * S3_IMG;
        f32x16_s3 accM[2], accC[2];                          // hi*hi products; cross products (hi*lo + lo*hi), summed separately; [pixel rows 2 p, 2 p + 1]
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int i = 0; i < 16; ++i) { accM[pp][i] = 0.f; accC[pp][i] = 0.f; }
        f32x4 vb[2][4];                                      // ring: [.][2 p + {hi, lo}]; reads of step st + 1 are issued before the MFMAs of st
        // step st = tap * 2 + s: input channels 32 kh + 16 s .. + 15 against tap (dy, dx) = (tap / 3, tap % 3); operand p: tile pixel rows 2 p, 2 p + 1
#define S3_BOFF(st, pp) ((2 * (pp) + ((st) >> 1) / 3) * S3_RP + (((st) >> 1) % 3) * SB_PW + 8 * ((st) & 1))
#define S3_LD(st) vb[(st) % 2][0] = *reinterpret_cast<const f32x4*>(Xs + S3_BOFF(st, 0)); \
                  vb[(st) % 2][1] = *reinterpret_cast<const f32x4*>(Xs + S3_BOFF(st, 0) + S3_IMG); \
                  vb[(st) % 2][2] = *reinterpret_cast<const f32x4*>(Xs + S3_BOFF(st, 1)); \
                  vb[(st) % 2][3] = *reinterpret_cast<const f32x4*>(Xs + S3_BOFF(st, 1) + S3_IMG);
#define S3_INTERLEAVE(nvalu) _Pragma("unroll") for (int g_ = 0; g_ < 6; ++g_) { \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, nvalu, 0); }
#if S2_STAMP
        { const unsigned long long now = __builtin_amdgcn_s_memtime();
          if (tix == 0) ts[2] = now; else if (tix == 1) ts[5] = now; else if (tix == 2) ts[8] = now; }
#endif
        if (FIRST) { w_issue(0); w_issue(1); w_issue(2); }
        S3_LD(0)
        had_next = has_next;
        unsigned nnorg, nnte;
        decode(pt + 2 * stride, nnorg, nnte);                // (past the last tile: unused)
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            if (st + 1 < 18) { S3_LD(st + 1) }
            if (FIRST && st % 2 == 0 && st / 2 + 3 < 9) w_issue(st / 2 + 3);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 wh = __builtin_bit_cast(bf16x8, wq[2 * st]), wl = __builtin_bit_cast(bf16x8, wq[2 * st + 1]);
            const bf16x8 b0h = __builtin_bit_cast(bf16x8, vb[st % 2][0]), b0l = __builtin_bit_cast(bf16x8, vb[st % 2][1]);
            const bf16x8 b1h = __builtin_bit_cast(bf16x8, vb[st % 2][2]), b1l = __builtin_bit_cast(bf16x8, vb[st % 2][3]);
            accM[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, b0h, accM[0], 0, 0, 0);
            accM[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, b1h, accM[1], 0, 0, 0);
            accC[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, b0h, accC[0], 0, 0, 0);
            accC[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, b1h, accC[1], 0, 0, 0);
            accC[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, b0l, accC[0], 0, 0, 0);
            accC[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, b1l, accC[1], 0, 0, 0);
            // Side work, at most one memory instruction per step (see the kernel above): odd steps 1..13: stage item k of the next tile's
            // halo (requested a tile ago) and request that register again for the tile after; steps 2, 6, 10, 14: the previous tile's
            // epilogue, one pixel row each (one store); steps 4, 8, 12, 16: this tile's epilogue inputs.
            if (st % 2 == 1 && st / 2 < S3_NIT) {
                halo_put(imgN, nte, st / 2);                 // (without a next tile the item is zeros and nobody reads the other buffer)
                halo_issue1(nnorg, nnte, had_next && pt + 2 * stride < npt, st / 2);
                S3_INTERLEAVE(INMODE == 1 ? 14 : 8)
            }
            if (st % 4 == 2 && st / 4 < S3_NEP) { epilogue(epi_pending, st / 4); S3_INTERLEAVE(8) }
            if (st % 4 == 0 && st > 0 && st / 4 - 1 < S3_NEP) epi_request(org, st / 4 - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
#if S2_STAMP
        { const unsigned long long now = __builtin_amdgcn_s_memtime();
          if (tix == 0) ts[3] = now; else if (tix == 1) ts[6] = now; else if (tix == 2) ts[9] = now; }
#endif
        has_next = had_next && pt + 2 * stride < npt;
        // the exchange: every wave leaves its half sums as an image [kh][pixel][channel] (epilogue: re-threaded over pixels).
        // 32x32 accumulator: register 4 g + r of lane (lj, kb) = output channel 32 coh + 8 g + 4 kb + r of pixel 32 p + lj
        {
            unsigned* xw = XCH + xbuf * S2_XCH + (kh * 64 + lj) * S2_XP + 32 * coh + 4 * kb;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = accM[pp][4 * g + r] + accC[pp][4 * g + r];
                    *reinterpret_cast<f32x4*>(xw + 32 * pp * S2_XP + 8 * g) = v;
                }
        }
        eorg = org; ebuf = xbuf; epi_pending = true;
#if S2_STAMP
        { const unsigned long long now = __builtin_amdgcn_s_memtime();
          if (tix == 0) ts[4] = now; else if (tix == 1) ts[7] = now; else if (tix == 2) ts[10] = now; }
        ++tix;
#endif
        __syncthreads();                                     // exchange + next halo published; everyone has left this tile's k-loop
        if (had_next) { pt += stride; org = norg; te = nte; norg = nnorg; nte = nnte; xbuf ^= 1; }
    };
    tile_body(std::true_type{});
    while (had_next) tile_body(std::false_type{});
    S2_STAMP_AT(11)
#pragma unroll
    for (int e = 0; e < S3_NEP; ++e) epilogue(true, e);
    S2_STAMP_AT(12)
#if S2_STAMP
    ts[31] = wall_clock64();
    if (lane == 0 && sbuf) {
#pragma unroll
        for (int i = 0; i < 32; ++i) sbuf[(blockIdx.x * 8 + wave) * 32 + i] = ts[i];
    }
#endif
    if (p.stats) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {                        // lanes l, l + 16, l + 32, l + 48 hold four pixels of the same channels
            st_s[r] += __shfl_xor(st_s[r], 16, 64); st_s[r] += __shfl_xor(st_s[r], 32, 64);
            st_q[r] += __shfl_xor(st_q[r], 16, 64); st_q[r] += __shfl_xor(st_q[r], 32, 64);
        }
        if (lane < 16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { st_red[wave][0][ec4 + r] = st_s[r]; st_red[wave][1][ec4 + r] = st_q[r]; }
        }
        __syncthreads();
        if (t < 128) {
            double a = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) a += (double)st_red[w][t >> 6][t & 63];
            p.stats[(long)blockIdx.x * 128 + t] = a;
        }
    }
}
// ---- round 6, third cut: the 32-channel MFMA waves of the kernel above with STAGING WAVES beside them -------------------------------------
// profiles/r06_conv3_sb2_stamps.txt: the one-wave-per-SIMD kernel above issues 740 instructions per tile per wave (108 MFMA, 304 VALU,
// 152 SALU, 106 LDS, 15 memory, 56 waits) and needs 5,200 cycles for 3,456 cycles of MFMAs: ONE in-order wave cannot issue the staging
// arithmetic and feed the matrix pipe.  The weight-gradient kernel (conv3w.hip) showed the way out: a second wave per SIMD that does
// everything except MFMAs.  Here waves 0-3 are the (coh, kh) MFMA waves -- operand reads and MFMAs only, one accumulator pair (the
// cross products join the hi*hi sum in the same register, freeing 32 registers: the 256-register budget of two waves per SIMD) -- and
// waves 4-7 stage the next tile's halo, request the tile after, run the previous tile's epilogue out of the exchange image and count the
// BatchNorm statistics.  One barrier per tile.  Same images, exchange, filter packing (modes 14 / 15) and results as generation 3 up to
// the order of fp32 summation.
template <int INMODE, bool EPBN>
__global__ __launch_bounds__(512, 2) void conv3_c64_sb4_kernel(Conv3SB q) {
    const Conv3P& p = q.c;
    extern __shared__ __attribute__((aligned(16))) unsigned smem_u[];
    unsigned* const XCH = smem_u + 4 * S3_IMG;
    const unsigned t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned W = p.W, H = p.H, tws = (W + 15) / 16, ths = H / 4, cob = p.Cout / 64;     // (W may be ragged: the last tile column is cut by the map)
    const unsigned npt = p.B * ths * tws;
    const unsigned stride = gridDim.x / cob;
    unsigned cb = blockIdx.x % cob, pt = blockIdx.x / cob;
    if (gridDim.x % 8 == 0 && 8 % cob == 0 && stride % (8 / cob) == 0) {      // XCD-aware tile order (see conv3_c64_ws_kernel)
        const unsigned x = blockIdx.x & 7, m = blockIdx.x >> 3, xg = 8 / cob;
        cb = x % cob;
        pt = (x / cob) * (stride / xg) + m;
    }
    __shared__ float st_red[4][2][64];
    __shared__ __attribute__((aligned(16))) float cst[7][64];   // in_scale, in_shift, in_scale2, ep_mean, ep_rstd, ep_gamma, ep_beta
    __shared__ __attribute__((aligned(16))) float cbias[64];    // this work-group's 64 output channels
    if (pt >= npt) {
        if (p.stats && t < 128) p.stats[(long)blockIdx.x * 128 + t] = 0.0;
        return;
    }
    const unsigned ntile = (npt - pt + stride - 1) / stride;    // tiles of this work-group: pt, pt + stride, ...
    // -> origin pixel index; te: bit 0 / 1 / 2 = the tile touches the top / bottom / left edge of the map, bits 8-15 = the number of its
    // 18 halo columns that lie inside the map on the right (17 + 1 for an inner tile; fewer in the last column of a ragged map)
    auto decode = [&](unsigned tile, unsigned& org, unsigned& te) {
        const unsigned tw = tile % tws; tile /= tws;
        const unsigned th = tile % ths, n = tile / ths;
        org = (n * H + th * 4) * W + tw * 16;
        const unsigned cin = min(18u, W - tw * 16 + 1);      // halo column c holds map column 16 tw + c - 1: inside while c <= W - 16 tw
        te = (th == 0 ? 1u : 0u) | (th == ths - 1 ? 2u : 0u) | (tw == 0 ? 4u : 0u) | (cin << 8);
    };
    if (wave >= 4) {
        // ------------------------------------------------ staging waves (256 threads) ------------------------------------------------
        const unsigned h = t - 256;
        const unsigned npix = p.B * H * W;
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, npix * q.cin_total * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(INMODE == 2 ? q.x2 : p.x), 0, npix * 256, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, npix * p.Cout * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(EPBN ? q.ep_x : p.y), 0, npix * (EPBN ? 64 : p.Cout) * 4, 0x00020000);
        const bool in_relu = p.in_act == ACT_RELU;
        const unsigned xps = q.cin_total * 4, yps = p.Cout * 4, eps_ = EPBN ? 256u : yps;     // bytes per pixel of the maps
        unsigned boff[S3_NIT], poff[S3_NIT], edge[2] = {0, 0}, colk[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < S3_NIT; ++k) {                   // halo item k: 16-byte channel group h & 15 of halo pixel (h >> 4) + 16 k (row-major 6 x 18)
            const unsigned pix = (h >> 4) + 16 * k, r = pix / S2_HW, c = pix - r * S2_HW;
            boff[k] = (unsigned)(((int)r - 1) * (int)W + (int)c - 1) * xps + (q.ci0 + 4 * (h & 15)) * 4;
            poff[k] = r * S3_RP + c * SB_PW + 2 * (h & 15);
            edge[k >> 2] |= ((r == 0 ? 1u : 0u) | (r == 5 ? 2u : 0u) | (c == 0 ? 4u : 0u) | (pix >= S2_NPX ? 8u : 0u)) << (8 * (k & 3));
            colk[k >> 2] |= c << (8 * (k & 3));
        }
        // item k lies outside the map: a top / bottom / left edge it sits on, no item at all, or its column beyond the map's last
        auto outside = [&](int k, unsigned te) -> bool {
            return (((edge[k >> 2] >> (8 * (k & 3))) & (te | 8u) & 15u) != 0) || (((colk[k >> 2] >> (8 * (k & 3))) & 255u) >= (te >> 8));
        };
        f32x4 hp[S3_NIT], hp2[S3_NIT];
        auto halo_issue = [&](unsigned tile, bool valid) {   // !valid: every request is out of range (no memory traffic)
            unsigned org, te;
            decode(tile, org, te);
#if S2_DEBUG
            if (s2_wrep & 4) valid = false;
#endif
#pragma unroll
            for (int k = 0; k < S3_NIT; ++k) {
                const unsigned a = (org * xps + boff[k]) | (outside(k, te) ? S2_OOB : (valid ? 0u : S2_OOB));
                hp[k] = s2_ld(rs_x, a);
                if (INMODE == 2) hp2[k] = s2_ld(rs_x2, a);
            }
        };
        auto halo_put = [&](unsigned* img, unsigned tile) {  // transform, split, store the seven items of `tile` into the image at `img`
            unsigned org, te;
            decode(tile, org, te);
            f32x4 isc, ish, isc2;
            if (INMODE != 0) { isc = *reinterpret_cast<const f32x4*>(&cst[0][4 * (h & 15)]); ish = *reinterpret_cast<const f32x4*>(&cst[1][4 * (h & 15)]); }
            if (INMODE == 2) isc2 = *reinterpret_cast<const f32x4*>(&cst[2][4 * (h & 15)]);
#pragma unroll
            for (int k = 0; k < S3_NIT; ++k) {
                const bool ok = !outside(k, te);
                f32x4 v = hp[k];
                if (INMODE == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], isc[e], fmaf(hp2[k][e], isc2[e], ish[e]));
                } else if (INMODE == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = mish_f(fmaf(v[e], isc[e], ish[e]));
                } else if (INMODE == 3) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = fmaf(v[e], isc[e], ish[e]);
                        v[e] = in_relu ? fmaxf(u, 0.f) : u;
                    }
                }
                if (INMODE != 0) {                           // (the zero padding is a padding of the TRANSFORMED map)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
                }
                if (k < S3_NIT - 1 || h < S2_NPX * 16 - 256 * (S3_NIT - 1)) {
                    uint2 hi, lo;
                    sb_split(v, hi, lo);
                    unsigned* d = img + poff[k];
                    *reinterpret_cast<uint2*>(d) = hi;
                    *reinterpret_cast<uint2*>(d + S3_IMG) = lo;
                }
            }
        };
        halo_issue(pt, true);                                // (the first HBM round trip is the longest wait of the kernel: requested before anything else)
        if (h < 64) {
            cbias[h] = p.bias ? p.bias[cb * 64 + h] : 0.f;
            if (INMODE != 0) { cst[0][h] = p.in_scale[h]; cst[1][h] = p.in_shift[h]; }
            if (INMODE == 2) cst[2][h] = q.in_scale2[h];
            if (EPBN) { cst[3][h] = q.ep_mean[h]; cst[4][h] = q.ep_rstd[h]; cst[5][h] = q.ep_gamma[h]; cst[6][h] = q.ep_beta[h]; }
        }
        __syncthreads();                                     // #1: constants in place
        halo_put(smem_u, pt);
        halo_issue(pt + stride, ntile > 1);
        __syncthreads();                                     // #2: tile 0 staged
        // epilogue item e of this thread: channels 4 (h & 15) .. + 3 of tile pixel (h >> 4) + 16 e (row e, column h >> 4)
        const bool use_beta = p.beta != 0.f, out_relu = p.act == ACT_RELU;
        const bool ep_mish = q.ep_act == ACT_MISH, ep_relu = q.ep_act == ACT_RELU;
        const unsigned ecol = h >> 4, ec4 = 4 * (h & 15);
        f32x4 epx[S3_NEP], st_s = (f32x4){0.f, 0.f, 0.f, 0.f}, st_q = st_s;
        auto epi_request = [&](unsigned tile) {              // (neither an EPBN map nor a beta: out of range, the answer is 0)
            unsigned org, te;
            decode(tile, org, te);
            const bool in_map = ecol + 1 < (te >> 8);        // (tile column ecol = halo column ecol + 1)
#pragma unroll
            for (int e = 0; e < S3_NEP; ++e)
                epx[e] = s2_ld(rs_e, ((org + e * W + ecol) * eps_ + (cb * 64 + ec4) * 4) | (((EPBN || use_beta) && in_map) ? 0u : S2_OOB));
        };
        auto epilogue = [&](unsigned tile, unsigned xb) {
            unsigned org, te;
            decode(tile, org, te);
            const f32x4 bj = *reinterpret_cast<const f32x4*>(&cbias[ec4]);
            const bool in_map = ecol + 1 < (te >> 8);        // pixels of a ragged map's last tile column beyond the map: not stored, not counted
            f32x4 ep_mu, ep_rs, ep_g, ep_b;
            if (EPBN) {                                      // (Cout == 64)
                ep_mu = *reinterpret_cast<const f32x4*>(&cst[3][ec4]); ep_rs = *reinterpret_cast<const f32x4*>(&cst[4][ec4]);
                ep_g = *reinterpret_cast<const f32x4*>(&cst[5][ec4]); ep_b = *reinterpret_cast<const f32x4*>(&cst[6][ec4]);
            }
#pragma unroll
            for (int e = 0; e < S3_NEP; ++e) {
                const unsigned* X = XCH + xb * S2_XCH + ((h >> 4) + 16 * e) * S2_XP + ec4;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(X), v1 = *reinterpret_cast<const f32x4*>(X + 64 * S2_XP);   // the two halves of the contraction
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = (v0[r] + v1[r]) + bj[r];
                    if (EPBN) {                              // v = gradient w.r.t. act(bn(ep_x)); Cout == 64
                        const float xh = (epx[e][r] - ep_mu[r]) * ep_rs[r], u = fmaf(ep_g[r], xh, ep_b[r]);
                        const float g = ep_mish ? mish_grad_f(u) : 1.f;
                        v[r] *= ep_relu ? (u > 0.f ? 1.f : 0.f) : g;
                        v[r] = in_map ? v[r] : 0.f;
                        st_s[r] += v[r]; st_q[r] += v[r] * xh;
                    } else {
                        v[r] = fmaf(p.beta, epx[e][r], v[r]);
                        v[r] = out_relu ? fmaxf(v[r], 0.f) : v[r];   // (the only output activation of this kernel; mish / tanh: row-tile kernel)
                        v[r] = in_map ? v[r] : 0.f;
                        st_s[r] += v[r]; st_q[r] += v[r] * v[r];
                    }
                }
#if S2_DEBUG
                if (s2_wrep & 8) continue;
#endif
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_s2, v), rs_y, ((org + e * W + ecol) * yps + (cb * 64 + ec4) * 4) | (in_map ? 0u : S2_OOB), 0, 0);
            }
        };
        for (unsigned i = 0; i < ntile; ++i) {
            // beside the MFMAs of tile i: finish tile i - 1, stage tile i + 1 (requested a tile ago), request tile i + 2 and tile i's epilogue inputs
            const unsigned tile = pt + i * stride;
#if S2_DEBUG
            if (s2_wrep & 1) { __syncthreads(); continue; }
#endif
#if S2_DEBUG
            if (!(s2_wrep & 32))
#endif
            if (i > 0) epilogue(tile - stride, (i - 1) & 1);
#if S2_DEBUG
            if (!(s2_wrep & 16))
#endif
            if (i + 1 < ntile) halo_put(smem_u + ((i + 1) & 1) * 2 * S3_IMG, tile + stride);
            halo_issue(tile + 2 * stride, i + 2 < ntile);
            epi_request(tile);
            __syncthreads();                                 // tile i's exchange image written; tile i + 1 staged
        }
        epilogue(pt + (ntile - 1) * stride, (ntile - 1) & 1);
        if (p.stats) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {                    // lanes l, l + 16, l + 32, l + 48 hold four pixels of the same channels
                st_s[r] += __shfl_xor(st_s[r], 16, 64); st_s[r] += __shfl_xor(st_s[r], 32, 64);
                st_q[r] += __shfl_xor(st_q[r], 16, 64); st_q[r] += __shfl_xor(st_q[r], 32, 64);
            }
            if (lane < 16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { st_red[wave - 4][0][ec4 + r] = st_s[r]; st_red[wave - 4][1][ec4 + r] = st_q[r]; }
            }
        }
        __syncthreads();                                     // (last: the statistics of the four staging waves)
        if (p.stats && h < 128) {
            double a = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) a += (double)st_red[w][h >> 6][h & 63];
            p.stats[(long)blockIdx.x * 128 + h] = a;
        }
        return;
    }
    // ------------------------------------------------------ MFMA waves (coh, kh) ------------------------------------------------------
    const unsigned coh = wave & 1, kh = wave >> 1, lj = lane & 31, kb = lane >> 5;
    // The filter: [(tap * 2 + s) * 2 + {hi, lo}] = 8 bf16 = input channels 32 kh + 16 s + 8 kb .. of output channel 32 coh + lj (mode 14 / 15
    // packing); requested at once, consumed tap by tap as it arrives (the first tile is paced by it: 147 KB per work-group at ~27 B/clk)
    f32x4 wq[36];
    {
        const f32x4* const wsrc = reinterpret_cast<const f32x4*>(p.w) + (cb * 2 + coh) * 72 * 64 + lane;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int f = 0; f < 4; ++f) wq[4 * tap + f] = wsrc[((tap * 2 + kh) * 4 + f) * 64];
    }
    const unsigned bbase = (lj >> 4) * S3_RP + (lj & 15) * SB_PW + 16 * kh + 4 * kb;
    __syncthreads();                                         // #1
    __syncthreads();                                         // #2: tile 0 staged
    for (unsigned i = 0; i < ntile; ++i) {
#if S2_DEBUG
        if (s2_wrep & 2) { __syncthreads(); continue; }
#endif
        const unsigned* Xs = smem_u + (i & 1) * 2 * S3_IMG + bbase;
        f32x16_s3 acc[2];                                    // [pixel rows 2 p, 2 p + 1]: hi*hi and the cross products in one sum
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[pp][v] = 0.f;
        f32x4 vb[2][4];                                      // ring: [.][2 p + {hi, lo}]; reads of step st + 1 are issued before the MFMAs of st
        S3_LD(0)                                             // (two steps ahead measured the same: profiles/r06_conv3_sb4_roles.txt)
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            if (st + 1 < 18) { S3_LD(st + 1) }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 wh = __builtin_bit_cast(bf16x8, wq[2 * st]), wl = __builtin_bit_cast(bf16x8, wq[2 * st + 1]);
            const bf16x8 b0h = __builtin_bit_cast(bf16x8, vb[st % 2][0]), b0l = __builtin_bit_cast(bf16x8, vb[st % 2][1]);
            const bf16x8 b1h = __builtin_bit_cast(bf16x8, vb[st % 2][2]), b1l = __builtin_bit_cast(bf16x8, vb[st % 2][3]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, b0h, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, b1h, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, b0h, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, b1h, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, b0l, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, b1l, acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the exchange: every MFMA wave leaves its half sums as an image [kh][pixel][channel] for the staging waves' epilogue.
        // 32x32 accumulator: register 4 g + r of lane (lj, kb) = output channel 32 coh + 8 g + 4 kb + r of pixel 32 p + lj
        unsigned* xw = XCH + (i & 1) * S2_XCH + (kh * 64 + lj) * S2_XP + 32 * coh + 4 * kb;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[pp][4 * g + r];
                *reinterpret_cast<f32x4*>(xw + 32 * pp * S2_XP + 8 * g) = v;
            }
        __syncthreads();
    }
    __syncthreads();                                         // (last: the staging waves' statistics hand-over)
}
static int conv3_sb_generation = 4;                          // A/B hook (tatt_conv3_sb_generation): 1 = the row-tile kernel of rounds 3-5
TATT_API int tatt_conv3_sb_generation(int gen) {
    const int old = conv3_sb_generation;
    if (gen == 1 || gen == 3 || gen == 4) conv3_sb_generation = gen;
    return old;
}
template <int INMODE, bool EPBN>
static void conv3_sb4_go(const Conv3SB& q, dim3 grid, hipStream_t st) {
    static TattPerDevice attr_once;
    tatt_per_device(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_sb4_kernel<INMODE, EPBN>), hipFuncAttributeMaxDynamicSharedMemorySize, S3_LDS);
    });
    hipLaunchKernelGGL((conv3_c64_sb4_kernel<INMODE, EPBN>), grid, dim3(512), S3_LDS, st, q);
}
template <int INMODE, bool EPBN>
static void conv3_sb3_go(const Conv3SB& q, dim3 grid, hipStream_t st) {
    static TattPerDevice attr_once;
    tatt_per_device(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_sb3_kernel<INMODE, EPBN>), hipFuncAttributeMaxDynamicSharedMemorySize, S3_LDS);
    });
    hipLaunchKernelGGL((conv3_c64_sb3_kernel<INMODE, EPBN>), grid, dim3(256), S3_LDS, st, q);
}
// which kernel a geometry runs: 1 = row tiles (filter packing modes 10 / 11), 3 = square tiles, 32-channel waves (14 / 15), 4 = 3 with
// staging waves beside the MFMA waves (14 / 15)
static int conv3_sb_pick(int B, int H, int W, int cin_total, int Cout, int act, int ep_act) {
    const long maxb = (long)B * H * W * (cin_total > Cout ? cin_total : Cout) * 4;       // 32-bit buffer offsets
    if (conv3_sb_generation < 3 || H % 4 || maxb >= 0x7fffffffL || ep_act == ACT_TANH) return 1;
    // what the square-tile kernels evaluate without branches: generation 4 also a ReLU on its output and map widths that are not a
    // multiple of 16 (the CRNN's 50-, 25-, 26-pixel maps: the last tile column is cut by the map)
    if (conv3_sb_generation == 4 && (act == ACT_NONE || (act == ACT_RELU && ep_act == ACT_NONE))) return 4;
    return (conv3_sb_generation == 3 && W % 16 == 0 && act == ACT_NONE) ? 3 : 1;
}
TATT_API int tatt_conv3_sb_packing(int B, int H, int W, int cin_total, int Cout, int act, int ep_act) {
    return conv3_sb_pick(B, H, W, cin_total, Cout, act, ep_act) >= 3 ? 14 : 10;
}
static int conv3_sb_launch(const Conv3SB& q, hipStream_t st) {
    const Conv3P& p = q.c;
    const int pick = conv3_sb_pick(p.B, p.H, p.W, q.cin_total, p.Cout, p.act, q.ep_act);
    if (pick >= 3) {
        const int cob = p.Cout / 64, npt = p.B * (p.H / 4) * ((p.W + 15) / 16);
        int per = 256 / cob;
        if (per > npt) per = npt;
        const dim3 grid(per * cob);
        const bool ep = q.ep_x != nullptr;
        const int inmode = q.x2 ? 2 : (p.in_scale ? (p.in_act == ACT_MISH ? 1 : 3) : 0);
        if (pick == 4) {                                     // (filter packing: modes 14 / 15)
            if (inmode == 2) { if (ep) conv3_sb4_go<2, true>(q, grid, st); else conv3_sb4_go<2, false>(q, grid, st); }
            else if (inmode == 1) { if (ep) conv3_sb4_go<1, true>(q, grid, st); else conv3_sb4_go<1, false>(q, grid, st); }
            else if (inmode == 3) { if (ep) conv3_sb4_go<3, true>(q, grid, st); else conv3_sb4_go<3, false>(q, grid, st); }
            else { if (ep) conv3_sb4_go<0, true>(q, grid, st); else conv3_sb4_go<0, false>(q, grid, st); }
            return LAUNCH_CHECK();
        }
        if (pick == 3) {                                     // (filter packing: modes 14 / 15)
            if (inmode == 2) { if (ep) conv3_sb3_go<2, true>(q, grid, st); else conv3_sb3_go<2, false>(q, grid, st); }
            else if (inmode == 1) { if (ep) conv3_sb3_go<1, true>(q, grid, st); else conv3_sb3_go<1, false>(q, grid, st); }
            else if (inmode == 3) { if (ep) conv3_sb3_go<3, true>(q, grid, st); else conv3_sb3_go<3, false>(q, grid, st); }
            else { if (ep) conv3_sb3_go<0, true>(q, grid, st); else conv3_sb3_go<0, false>(q, grid, st); }
            return LAUNCH_CHECK();
        }
        return LAUNCH_CHECK();
    }
    if (p.W % C3_PX) return 1;                               // the row-tile kernel walks 64-pixel row segments
    static TattPerDevice attr_once;
    tatt_per_device(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_sb_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, SB_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_sb_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, SB_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_sb_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SB_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_sb_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SB_LDS);
    });
    const int cob = p.Cout / 64, npt = p.B * p.H * (p.W / C3_PX);
    int per = 256 / cob;
    if (per > npt) per = npt;
    const dim3 grid(per * cob), block(512);
    const bool in2 = q.x2 != nullptr, ep = q.ep_x != nullptr;
    if (in2 && ep) hipLaunchKernelGGL((conv3_c64_sb_kernel<true, true>), grid, block, SB_LDS, st, q);
    else if (in2) hipLaunchKernelGGL((conv3_c64_sb_kernel<true, false>), grid, block, SB_LDS, st, q);
    else if (ep) hipLaunchKernelGGL((conv3_c64_sb_kernel<false, true>), grid, block, SB_LDS, st, q);
    else hipLaunchKernelGGL((conv3_c64_sb_kernel<false, false>), grid, block, SB_LDS, st, q);
    return LAUNCH_CHECK();
}
TATT_API int tatt_conv3_c64_fwd_sb(const float* x, int cin_total, int ci0, const float* wl, const float* bias, float* y, int B, int H,
                                   int W, int Cout, int act, float beta, const float* in_scale, const float* in_shift, int in_act,
                                   double* stats, hipStream_t st) {
    if (Cout % 64 || cin_total % 4 || ci0 % 4 || ci0 + 64 > cin_total) return 1;
    if (stats && (Cout != 64 || act != ACT_NONE || beta != 0.f)) return 2;
    if (in_scale && (!in_shift || in_act == ACT_TANH)) return 3;
    Conv3SB q = {{x, wl, bias, y, B, H, W, 64, Cout, act, beta, in_scale, in_shift, in_act, stats}, cin_total, ci0,
                 nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    return conv3_sb_launch(q, st);
}
// The data-gradient convolution of a conv -> bn -> act -> conv -> bn chain with the BatchNorm backward folded in on both sides (64 -> 64
// channels, wl = mode-11 packed filter of the convolution whose data gradient this is):
//   input side (x2 != NULL): the gradient entering is  x * in_scale + x2 * in_scale2 + in_shift  per channel (coefficients from
//       tatt_bn_bwd_finish: x = upstream gradient du, x2 = the BatchNorm's input);  x2 == NULL: x is taken as it is;
//   output side (ep_x != NULL): y = (conv result) * ep_act'(ep_gamma xhat + ep_beta), xhat = (ep_x - ep_mean) ep_rstd, and stats
//       [min(256, B*H*W/64)][2][64] doubles = per-work-group sums of y and of y * xhat (stage-1 partials of that BatchNorm's backward).
TATT_API int tatt_conv3_c64_dgrad_bn_sb(const float* x, const float* x2, const float* in_scale, const float* in_scale2,
                                        const float* in_shift, const float* wl, float* y, int B, int H, int W, const float* ep_x,
                                        const float* ep_mean, const float* ep_rstd, const float* ep_gamma, const float* ep_beta,
                                        int ep_act, double* stats, hipStream_t st) {
    if (x2 && (!in_scale || !in_scale2 || !in_shift)) return 2;
    if (ep_x && (!ep_mean || !ep_rstd || !ep_gamma || !ep_beta || !stats)) return 3;
    if (!ep_x && stats) return 4;
    Conv3SB q = {{x, wl, nullptr, y, B, H, W, 64, 64, ACT_NONE, 0.f, x2 ? in_scale : nullptr, x2 ? in_shift : nullptr, ACT_NONE, stats}, 64, 0,
                 x2, in_scale2, ep_x, ep_mean, ep_rstd, ep_gamma, ep_beta, ep_act};
    return conv3_sb_launch(q, st);
}

// ---- weight gradient -------------------------------------------------------------------------------------------------
// dW[tap][ci][co] = sum over pixels x[pixel + tap][ci] * dy[pixel][co]: the pixels are the contraction axis.  Persistent
// work-groups of 8 waves walk 64-pixel row segments; wave w owns the (ci half, co half) quadrant w & 3 of one 64 ci x 64 co
// block for the taps of its group (waves 0-3: taps 0-4, waves 4-7: taps 5-8 -- five / four 32x32 accumulators that live for
// the whole kernel), so two waves share each SIMD and hide each other's LDS latency and prefetch traffic.  The halo of x and
// the dy tile of the NEXT segment are prefetched global -> registers -> LDS under the MFMAs (double-buffered, one barrier per
// segment).  Per-work-group partials leave as 16-byte stores (transposed through LDS) and are summed deterministically and
// scattered to the OIHW parameter layout by the split-K reducer of gemm.hip.
struct Conv3WP {
    const float* x; const float* dy; float* part;
    int B, H, W, Cin, Cout, nseg;
    float* pdb;      // per-work-group partial bias gradient [gridDim.x][Cout] (column sums of dy), or null
};
#define WG_HALO (3 * C3_HW * 64)                     // floats: x halo [3][66][64 ci]
#define WG_BUF (WG_HALO + 64 * 64)                   // + dy tile [64 px][64 co]
#define C3_WG_LDS (2 * WG_BUF * 4)                   // 133.9 KB
#define WG_F4 (WG_BUF / 4)                           // float4 per buffer: 3168 + 1024 = 4192

// dbacc: running sum of the B operands (dy values) this lane feeds to the MFMAs -- over a launch that is the column sum of dy
// for output channel (lane & 31) of the wave's half over the pixels of parity lane >> 5: the bias gradient, at one add per k-step.
template <int T0, int NT>
__device__ __forceinline__ void wgrad_segment(f32x16 (&acc)[5], const float* __restrict__ acol, const float* __restrict__ bcol,
                                              const Conv3WP& p, bool has_next, int nn, int nh, int nw0, int ci0, int co0,
                                              float* __restrict__ nbuf, int t, float& dbacc) {
    // float4 #idx of the next segment's buffer: [0, 3168) halo (c4 = idx & 15, pixel = idx >> 4), then the dy tile
    auto fetch = [&](int idx) -> f32x4 {
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (idx < WG_HALO / 4) {
            const int c4 = idx & 15, pp = idx >> 4;
            const int r = pp / C3_HW, px = pp - r * C3_HW;
            const int hh = nh + r - 1, ww = nw0 + px - 1;
            if (hh >= 0 && hh < p.H && ww >= 0 && ww < p.W)
                v = *reinterpret_cast<const f32x4*>(p.x + (((long)nn * p.H + hh) * p.W + ww) * p.Cin + ci0 + 4 * c4);
        } else if (idx < WG_F4) {
            const int j = idx - WG_HALO / 4, c4 = j & 15, px = j >> 4;
            v = *reinterpret_cast<const f32x4*>(p.dy + (((long)nn * p.H + nh) * p.W + nw0 + px) * p.Cout + co0 + 4 * c4);
        }
        return v;
    };
    // 32 k-steps (pixel pairs); operands of step k+1 are read (pinned by sched_barrier) before the NT MFMAs of step k issue.
    // Every 3rd k-step carries one float4 of the next segment: load at k % 6 == 0, publish at k % 6 == 4 (9 rounds of 512).
    float wa[2][NT], wb[2];
    f32x4 pf = (f32x4){0.f, 0.f, 0.f, 0.f};
#define C3W_LOAD(buf, k)                                                                   \
    wb[buf] = bcol[(k) * 64];                                                              \
    _Pragma("unroll") for (int tp = 0; tp < NT; ++tp)                                      \
        wa[buf][tp] = acol[(((T0 + tp) / 3) * C3_HW + (k) + ((T0 + tp) % 3)) * 64];
    C3W_LOAD(0, 0)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 64; k += 2) {
        const int cur = (k >> 1) & 1;
        if (k + 2 < 64) { C3W_LOAD(cur ^ 1, k + 2) }
        if (has_next && k % 6 == 0 && k / 6 < 9) pf = fetch(t + 512 * (k / 6));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tp = 0; tp < NT; ++tp)
            acc[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cur][tp], wb[cur], acc[tp], 0, 0, 0);
        if (T0 == 0) dbacc += wb[cur];
        __builtin_amdgcn_sched_barrier(0);
        if (has_next && k % 6 == 4 && k / 6 < 9) {
            const int idx = t + 512 * (k / 6);
            if (idx < WG_F4) *reinterpret_cast<f32x4*>(nbuf + 4 * idx) = pf;
        }
    }
#undef C3W_LOAD
}

#define WS_TP 36                              // transpose-tile pitch (floats) of the weight-gradient epilogue
#define WS_TT (32 * WS_TP)                    // floats per per-wave epilogue tile
__global__ __launch_bounds__(512, 1) void conv3_c64_wgrad_kernel(Conv3WP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int qi = wave & 1, qo = (wave >> 1) & 1, tg = wave >> 2;   // (ci half, co half) quadrant; tap group
    const int cib = blockIdx.y % (p.Cin / 64), cob = blockIdx.y / (p.Cin / 64);
    const int ci0 = cib * 64, co0 = cob * 64;
    const int segs = p.W / C3_PX;
    f32x16 acc[5];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    auto decode = [&](int s, int& n, int& h, int& w0) {
        const int seg = s % segs; s /= segs;
        h = s % p.H; n = s / p.H; w0 = seg * C3_PX;
    };
    float dbacc = 0.f;
    int s = blockIdx.x;
    if (s < p.nseg) {
        int n, h, w0;
        decode(s, n, h, w0);
        {   // first segment: all loads in flight at once
            f32x4 v[9];
            const Conv3WP& q = p;
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const int idx = t + 512 * r;
                f32x4 u = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (idx < WG_HALO / 4) {
                    const int c4 = idx & 15, pp = idx >> 4;
                    const int rr = pp / C3_HW, px = pp - rr * C3_HW;
                    const int hh = h + rr - 1, ww = w0 + px - 1;
                    if (hh >= 0 && hh < q.H && ww >= 0 && ww < q.W)
                        u = *reinterpret_cast<const f32x4*>(q.x + (((long)n * q.H + hh) * q.W + ww) * q.Cin + ci0 + 4 * c4);
                } else if (idx < WG_F4) {
                    const int j = idx - WG_HALO / 4, c4 = j & 15, px = j >> 4;
                    u = *reinterpret_cast<const f32x4*>(q.dy + (((long)n * q.H + h) * q.W + w0 + px) * q.Cout + co0 + 4 * c4);
                }
                v[r] = u;
            }
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const int idx = t + 512 * r;
                if (idx < WG_F4) *reinterpret_cast<f32x4*>(smem + 4 * idx) = v[r];
            }
        }
        __syncthreads();
        int buf = 0;
        while (true) {
            const int sn = s + gridDim.x;
            const bool has_next = sn < p.nseg;
            int nn = 0, nh = 0, nw0 = 0;
            if (has_next) decode(sn, nn, nh, nw0);
            const float* cur = smem + buf * WG_BUF;
            float* nxt = smem + (buf ^ 1) * WG_BUF;
            // A(i = ci, k = pixel) = Xs[kh][pixel + kw][ci];  B(k = pixel, j = co) = Ds[pixel][co]
            const int kq = lane >> 5;
            const float* acol = cur + kq * 64 + qi * 32 + (lane & 31);
            const float* bcol = cur + WG_HALO + kq * 64 + qo * 32 + (lane & 31);
            if (tg == 0) wgrad_segment<0, 5>(acc, acol, bcol, p, has_next, nn, nh, nw0, ci0, co0, nxt, t, dbacc);
            else wgrad_segment<5, 4>(acc, acol, bcol, p, has_next, nn, nh, nw0, ci0, co0, nxt, t, dbacc);
            __syncthreads();                               // next buffer published; everyone is done with the current one
            if (!has_next) break;
            s = sn;
            buf ^= 1;
        }
    }
    if (p.pdb && cib == 0 && tg == 0 && qi == 0) {           // bias gradient partial of this work-group: both pixel parities
        const float v = dbacc + __shfl_xor(dbacc, 32, 64);
        if (lane < 32) p.pdb[(long)blockIdx.x * p.Cout + co0 + qo * 32 + lane] = v;
    }
    // ---- partial[blockIdx.x][(tap*Cin + ci)][Cout]: each 32 ci x 32 co accumulator is transposed through a per-wave LDS tile
    // (the segment buffers are free now) and leaves as 4 sixteen-byte stores per lane ----
    float* P = p.part + (long)blockIdx.x * 9 * p.Cin * p.Cout;
    float* T = smem + wave * WS_TT;
    const int ntap = tg == 0 ? 5 : 4, tap0 = tg == 0 ? 0 : 5;
#pragma unroll
    for (int tp = 0; tp < 5; ++tp) {
        if (tp < ntap) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                T[((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * WS_TP + (lane & 31)] = acc[tp][reg];
            wave_lds_sync();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = (lane >> 3) + 8 * q, c4 = lane & 7;
                const f32x4 v = *reinterpret_cast<const f32x4*>(T + r * WS_TP + 4 * c4);
                *reinterpret_cast<f32x4*>(P + ((long)(tap0 + tp) * p.Cin + ci0 + qi * 32 + r) * p.Cout + co0 + qo * 32 + 4 * c4) = v;
            }
            wave_lds_sync();
        }
    }
}
// partials: part[G][9*Cin][Cout] with G work-groups along x (<= 256 / blocks); reduce with the split-K reducer
TATT_API int tatt_conv3_c64_wgrad_partial(const float* x, const float* dy, float* part, float* pdb, int B, int H, int W,
                                          int Cin, int Cout, int G, hipStream_t st) {
    if (Cin % 64 || Cout % 64 || W % C3_PX) return 1;
    const int nseg = B * H * (W / C3_PX);
    Conv3WP p = {x, dy, part, B, H, W, Cin, Cout, nseg, pdb};
    static TattPerDevice attr_once;                 // once per device, under the site lock (common.h)
    tatt_per_device(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  C3_WG_LDS);
    });
    hipLaunchKernelGGL(conv3_c64_wgrad_kernel, dim3(G, (Cin / 64) * (Cout / 64)), dim3(512), C3_WG_LDS, st, p);
    return LAUNCH_CHECK();
}
