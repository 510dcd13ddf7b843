// 3x3 convolutions between multiples of 64 channels on NHWC maps -- the bulk of the backbone's FLOPs (11 of them per
// forward, reference model/tsrn.py:877,885,612,1043) -- on v_mfma_f32_32x32x2_f32 (exact fp32).
//
// Forward / data-gradient (tatt_conv3_c64_fwd): one work-group = one 64-pixel row segment x 64 output channels.
//   * the (3 x 66)-pixel halo of 64 input channels is staged ONCE in LDS ([row][px][65]: the odd pitch makes the MFMA
//     A-operand read  lane -> (pixel = lane&31, k = lane>>5)  bank-conflict free) and reused by all 9 taps;
//   * the 64x64 filter slice of each tap streams through a single LDS buffer, prefetched into registers while the
//     previous tap's 32 MFMAs per wave run; 67.5 KB LDS => 2 work-groups per CU overlap each other's barriers;
//   * 4 waves = 2 (pixel halves) x 2 (channel halves), one 32x32 accumulator each; 288 MFMAs per wave per 64 input channels.
// Weight-gradient (tatt_conv3_c64_wgrad): persistent work-groups walk row segments; each wave keeps the 9 taps x (32 ci x 32 co)
//   quadrant in 9 accumulators (144 VGPRs); A = x halo read channel-contiguous, B = dy tile; per-block partials are
//   summed deterministically and scattered to the OIHW parameter layout by the split-K reducer of gemm.hip.
#include "common.h"
#include <stdlib.h>

#define C3_PX 64
#define C3_XP 65     // halo channel pitch (floats)
#define C3_HW 66     // halo width (pixels)

struct Conv3P {
    const float* x; const float* w; const float* bias; float* y;
    int B, H, W, Cin, Cout, act;
    float beta;
    long long* prof;      // diagnostics only (tools/bench_kernels.py --prof): per-work-group cycle breakdown, else nullptr
};

__global__ __launch_bounds__(256, 2) void conv3_c64_fwd_kernel(Conv3P p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*Xs)[C3_HW][C3_XP] = reinterpret_cast<float (*)[C3_HW][C3_XP]>(smem);           // [3][66][65]
    float (*Ws)[64] = reinterpret_cast<float (*)[64]>(smem + 3 * C3_HW * C3_XP + 2);       // [64][64], 16B aligned (12872 floats)

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int segs = p.W / C3_PX, cob = p.Cout / 64;
    int bid = blockIdx.x;
    const int cb = bid % cob; bid /= cob;
    const int seg = bid % segs; bid /= segs;
    const int h = bid % p.H; const int n = bid / p.H;
    const int w0 = seg * C3_PX, co0 = cb * 64;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    f32x4 wreg[4];
    auto load_w = [&](int tap, int ci0) {
        // filter slice [tap][ci0..ci0+64][co0..co0+64] of the packed [9][Cin][Cout] operand: 64 rows of 256 B
        const float* src = p.w + ((long)tap * p.Cin + ci0) * p.Cout + co0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = t + 256 * q;             // 0..1023 float4s
            const int r = idx >> 4, c4 = idx & 15;
            wreg[q] = *reinterpret_cast<const f32x4*>(src + (long)r * p.Cout + 4 * c4);
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = t + 256 * q;
            *reinterpret_cast<f32x4*>(&Ws[idx >> 4][4 * (idx & 15)]) = wreg[q];
        }
    };

    for (int ci0 = 0; ci0 < p.Cin; ci0 += 64) {
        __syncthreads();                      // previous chunk's readers are done with Xs / Ws
        // ---- stage the halo: 3 rows x 66 pixels x 64 channels (16 float4 per pixel) ----
        for (int i = t; i < 3 * C3_HW * 16; i += 256) {
            const int c4 = i & 15, pp = i >> 4;
            const int r = pp / C3_HW, px = pp - r * C3_HW;
            const int hh = h + r - 1, ww = w0 + px - 1;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (hh >= 0 && hh < p.H && ww >= 0 && ww < p.W)
                v = *reinterpret_cast<const f32x4*>(p.x + (((long)n * p.H + hh) * p.W + ww) * p.Cin + ci0 + 4 * c4);
            float* d = &Xs[r][px][4 * c4];
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
        load_w(0, ci0);
        store_w();
        __syncthreads();
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) load_w(tap + 1, ci0);
            const int kh = tap / 3, kw = tap - 3 * kh;
            const float* arow = &Xs[kh][wm * 32 + (lane & 31) + kw][lane >> 5];
            const float* brow = &Ws[lane >> 5][wn * 32 + (lane & 31)];
#pragma unroll
            for (int k = 0; k < 64; k += 2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[k], brow[k * 64], acc, 0, 0, 0);
            if (tap + 1 < 9) {
                __syncthreads();              // everyone finished reading Ws
                store_w();
                __syncthreads();
            }
        }
    }
    // ---- epilogue: col = lane&31 (output channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel) ----
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
    }
}
// ---- persistent, software-pipelined variant (default) ------------------------------------------------------------------------
// One work-group per CU walks its (tile, 64-input-channel chunk) work items.  While the 32 MFMAs per wave of a tap run, the
// filter slice of the NEXT tap and one ninth of the NEXT work item's halo are in flight from L2/HBM into registers; they are
// written to the other LDS buffer after the MFMAs and published by the single barrier that closes the tap.  LDS: 2 halos
// (2 x 51.5 KB) + 2 filter slices (2 x 16 KB) = 135.7 KB.
#define C3_HALO_F (3 * C3_HW * C3_XP)            // 12870 floats
#define C3_HALO_PITCH 12872                      // 16-byte aligned pitch between the two halo buffers
#define C3_V2_LDS ((2 * C3_HALO_PITCH + 2 * 64 * 64) * 4)
#define C3_SLICE 352                             // float4s of the next halo fetched per tap (9 * 352 = 3168)

__global__ __launch_bounds__(256) void conv3_c64_fwd_v2_kernel(Conv3P p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* XsB = smem;                                   // [2][3][66][65]
    float* WsB = smem + 2 * C3_HALO_PITCH;               // [2][64][64]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int segs = p.W / C3_PX, cob = p.Cout / 64, nch = p.Cin / 64;
    const int ntiles = p.B * p.H * segs * cob;
    const int G = gridDim.x;

    auto decode = [&](int tile, int& n, int& h, int& w0, int& co0) {
        int bid = tile;
        const int cb = bid % cob; bid /= cob;
        const int seg = bid % segs; bid /= segs;
        h = bid % p.H; n = bid / p.H;
        w0 = seg * C3_PX; co0 = cb * 64;
    };
    // one float4 of the halo of (tile, chunk): idx in [0, 3168)
    auto halo_load = [&](int n, int h, int w0, int ci0, int idx) -> f32x4 {
        const int c4 = idx & 15, pp = idx >> 4;
        const int r = pp / C3_HW, px = pp - r * C3_HW;
        const int hh = h + r - 1, ww = w0 + px - 1;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (hh >= 0 && hh < p.H && ww >= 0 && ww < p.W)
            v = *reinterpret_cast<const f32x4*>(p.x + (((long)n * p.H + hh) * p.W + ww) * p.Cin + ci0 + 4 * c4);
        return v;
    };
    auto halo_store = [&](float* Xs, int idx, f32x4 v) {
        const int c4 = idx & 15, pp = idx >> 4;
        float* d = Xs + pp * C3_XP + 4 * c4;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    };
    f32x4 wreg[4];
    auto load_w = [&](int tap, int ci0, int co0) {
        const float* src = p.w + ((long)tap * p.Cin + ci0) * p.Cout + co0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = t + 256 * q;
            wreg[q] = *reinterpret_cast<const f32x4*>(src + (long)(idx >> 4) * p.Cout + 4 * (idx & 15));
        }
    };
    auto store_w = [&](float* Ws) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = t + 256 * q;
            *reinterpret_cast<f32x4*>(Ws + (idx >> 4) * 64 + 4 * (idx & 15)) = wreg[q];
        }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    int n, h, w0, co0;
    decode(tile, n, h, w0, co0);
    // ---- prologue: first halo + first filter slice ----
    for (int i = t; i < 9 * C3_SLICE; i += 256) halo_store(XsB, i, halo_load(n, h, w0, 0, i));
    load_w(0, 0, co0);
    store_w(WsB);
    __syncthreads();

    int xbuf = 0, wbuf = 0, ch = 0;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    while (true) {
        // next work item
        int ntile = tile, nchk = ch + 1;
        if (nchk == nch) { nchk = 0; ntile = tile + G; }
        const bool has_next = ntile < ntiles;
        int nn = n, nh = h, nw0 = w0, nco0 = co0;
        if (has_next && nchk == 0) decode(ntile, nn, nh, nw0, nco0);
        const float* Xs = XsB + xbuf * C3_HALO_PITCH;
        float* XsN = XsB + (xbuf ^ 1) * C3_HALO_PITCH;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            // ---- issue the loads that the MFMAs below will hide ----
            const bool more_w = tap < 8 || has_next;
            if (tap < 8) load_w(tap + 1, ch * 64, co0);
            else if (has_next) load_w(0, nchk * 64, nco0);
            f32x4 h0 = (f32x4){0.f, 0.f, 0.f, 0.f}, h1 = h0;
            const int i0 = tap * C3_SLICE + t, i1 = tap * C3_SLICE + 256 + t;
            if (has_next) {
                h0 = halo_load(nn, nh, nw0, nchk * 64, i0);
                if (t < C3_SLICE - 256) h1 = halo_load(nn, nh, nw0, nchk * 64, i1);
            }
            // ---- 32 MFMAs: A = halo (pixel, channel pair), B = filter slice (channel pair, output channel) ----
            const int kh = tap / 3, kw = tap - 3 * kh;
            const float* arow = Xs + (kh * C3_HW + wm * 32 + (lane & 31) + kw) * C3_XP + (lane >> 5);
            const float* brow = WsB + wbuf * 4096 + (lane >> 5) * 64 + wn * 32 + (lane & 31);
            // 32 dependent MFMAs in 4 groups of 8.  The LDS operand reads of group g+1 are pinned (sched_barrier) in front of the
            // MFMAs of group g, so they complete under 512 cycles of matrix work; left alone, the scheduler sinks every read pair
            // right in front of its MFMA and exposes the LDS latency 16 times per tap (measured: MFMA pipe 45 % busy).
            float ra[4][8], rb[4][8];
#define C3_LOADG(g)                                                                                   \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                            \
                ra[g][j] = arow[(g) * 16 + 2 * j]; rb[g][j] = brow[((g) * 16 + 2 * j) * 64]; }
#define C3_MFMAG(g)                                                                                   \
            _Pragma("unroll") for (int j = 0; j < 8; ++j)                                              \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[g][j], rb[g][j], acc, 0, 0, 0);
            C3_LOADG(0)
            __builtin_amdgcn_sched_barrier(0);
            C3_LOADG(1)
            __builtin_amdgcn_sched_barrier(0);
            C3_MFMAG(0)
            __builtin_amdgcn_sched_barrier(0);
            C3_LOADG(2)
            __builtin_amdgcn_sched_barrier(0);
            C3_MFMAG(1)
            __builtin_amdgcn_sched_barrier(0);
            C3_LOADG(3)
            __builtin_amdgcn_sched_barrier(0);
            C3_MFMAG(2)
            __builtin_amdgcn_sched_barrier(0);
            C3_MFMAG(3)
            __builtin_amdgcn_sched_barrier(0);
            // ---- publish the prefetched data ----
            if (more_w) store_w(WsB + (wbuf ^ 1) * 4096);
            if (has_next) {
                halo_store(XsN, i0, h0);
                if (t < C3_SLICE - 256) halo_store(XsN, i1, h1);
            }
            __syncthreads();
            wbuf ^= 1;
        }
        if (ch == nch - 1) {
            // ---- epilogue: col = lane&31 (output channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel) ----
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

// ---- v3: v2 with 8 waves per work-group: the input channels of a tap are split between two groups of 4 waves (two waves
// per SIMD), so the per-tap bookkeeping of one wave (address math, LDS publishes, barrier skew: ~1.3k cycles, which a lone
// wave per SIMD cannot overlap with its own dependent MFMA chain) runs under the other wave's MFMAs.  The two partial sums
// are combined through LDS in the epilogue.
#define C3_V3_LDS (C3_V2_LDS + 16 * 256 * 4)
__global__ __launch_bounds__(512) void conv3_c64_fwd_v3_kernel(Conv3P p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* XsB = smem;                                   // [2][3][66][65]
    float* WsB = smem + 2 * C3_HALO_PITCH;               // [2][64][64]
    float* Red = WsB + 2 * 4096;                         // [16][256] partial accumulators of the second channel half
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int half = wave >> 2, wm = wave & 1, wn = (wave >> 1) & 1;
    const int segs = p.W / C3_PX, cob = p.Cout / 64, nch = p.Cin / 64;
    const int ntiles = p.B * p.H * segs * cob;
    const int G = gridDim.x;

    auto decode = [&](int tile, int& n, int& h, int& w0, int& co0) {
        int bid = tile;
        const int cb = bid % cob; bid /= cob;
        const int seg = bid % segs; bid /= segs;
        h = bid % p.H; n = bid / p.H;
        w0 = seg * C3_PX; co0 = cb * 64;
    };
    // one float4 of the halo of (tile, chunk): idx in [0, 3168)
    auto halo_load = [&](int n, int h, int w0, int ci0, int idx) -> f32x4 {
        const int c4 = idx & 15, pp = idx >> 4;
        const int r = pp / C3_HW, px = pp - r * C3_HW;
        const int hh = h + r - 1, ww = w0 + px - 1;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (hh >= 0 && hh < p.H && ww >= 0 && ww < p.W)
            v = *reinterpret_cast<const f32x4*>(p.x + (((long)n * p.H + hh) * p.W + ww) * p.Cin + ci0 + 4 * c4);
        return v;
    };
    auto halo_store = [&](float* Xs, int idx, f32x4 v) {
        const int c4 = idx & 15, pp = idx >> 4;
        float* d = Xs + pp * C3_XP + 4 * c4;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    };
    f32x4 wreg[2];
    auto load_w = [&](int tap, int ci0, int co0) {
        const float* src = p.w + ((long)tap * p.Cin + ci0) * p.Cout + co0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = t + 512 * q;
            wreg[q] = *reinterpret_cast<const f32x4*>(src + (long)(idx >> 4) * p.Cout + 4 * (idx & 15));
        }
    };
    auto store_w = [&](float* Ws) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = t + 512 * q;
            *reinterpret_cast<f32x4*>(Ws + (idx >> 4) * 64 + 4 * (idx & 15)) = wreg[q];
        }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    int n, h, w0, co0;
    decode(tile, n, h, w0, co0);
    // ---- prologue: first halo + first filter slice ----
    for (int i = t; i < 9 * C3_SLICE; i += 512) halo_store(XsB, i, halo_load(n, h, w0, 0, i));
    load_w(0, 0, co0);
    store_w(WsB);
    __syncthreads();

    int xbuf = 0, wbuf = 0, ch = 0;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    while (true) {
        // next work item
        int ntile = tile, nchk = ch + 1;
        if (nchk == nch) { nchk = 0; ntile = tile + G; }
        const bool has_next = ntile < ntiles;
        int nn = n, nh = h, nw0 = w0, nco0 = co0;
        if (has_next && nchk == 0) decode(ntile, nn, nh, nw0, nco0);
        const float* Xs = XsB + xbuf * C3_HALO_PITCH;
        float* XsN = XsB + (xbuf ^ 1) * C3_HALO_PITCH;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            // ---- issue the loads that the MFMAs below will hide ----
            const bool more_w = tap < 8 || has_next;
            if (tap < 8) load_w(tap + 1, ch * 64, co0);
            else if (has_next) load_w(0, nchk * 64, nco0);
            f32x4 h0 = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int i0 = tap * C3_SLICE + t;
            const bool hload = has_next && t < C3_SLICE;
            if (hload) h0 = halo_load(nn, nh, nw0, nchk * 64, i0);
            // ---- 32 MFMAs: A = halo (pixel, channel pair), B = filter slice (channel pair, output channel) ----
            const int kh = tap / 3, kw = tap - 3 * kh;
            const float* arow = Xs + (kh * C3_HW + wm * 32 + (lane & 31) + kw) * C3_XP + half * 32 + (lane >> 5);
            const float* brow = WsB + wbuf * 4096 + (half * 32 + (lane >> 5)) * 64 + wn * 32 + (lane & 31);
            // 32 dependent MFMAs in 4 groups of 8.  The LDS operand reads of group g+1 are pinned (sched_barrier) in front of the
            // MFMAs of group g, so they complete under 512 cycles of matrix work; left alone, the scheduler sinks every read pair
            // right in front of its MFMA and exposes the LDS latency 16 times per tap (measured: MFMA pipe 45 % busy).
            float ra[2][8], rb[2][8];
            C3_LOADG(0)
            __builtin_amdgcn_sched_barrier(0);
            C3_LOADG(1)
            __builtin_amdgcn_sched_barrier(0);
            C3_MFMAG(0)
            __builtin_amdgcn_sched_barrier(0);
            C3_MFMAG(1)
            __builtin_amdgcn_sched_barrier(0);
            // ---- publish the prefetched data ----
            if (more_w) store_w(WsB + (wbuf ^ 1) * 4096);
            if (hload) halo_store(XsN, i0, h0);
            __syncthreads();
            wbuf ^= 1;
        }
        if (ch == nch - 1) {
            // ---- combine the two channel halves through LDS ----
            if (half == 1) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) { Red[reg * 256 + (t & 255)] = acc[reg]; acc[reg] = 0.f; }
            }
            __syncthreads();
            if (half == 0) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) acc[reg] += Red[reg * 256 + t];
            }
        }
        if (ch == nch - 1 && half == 0) {
            // ---- epilogue: col = lane&31 (output channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel) ----
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

// ---- v4: wave-specialised: 4 MFMA waves + 1 loader wave ----------------------------------------------------------------------------
// Measured on v2: per tap a wave spends ~2.0k cycles on 32 MFMAs and another ~2.1k on everything else (prefetch address math,
// global loads, LDS publishes, barrier, operand-read latency), and with one wave per SIMD nothing overlaps the two.  Here the
// four MFMA waves only read LDS and issue MFMAs; a fifth wave owns ALL global->LDS traffic (the next tap's 16 KB filter slice
// and 1/9 of the next work item's halo per tap).  One barrier per tap hands the buffers over.
__global__ __launch_bounds__(320) void conv3_c64_fwd_v4_kernel(Conv3P p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* XsB = smem;                                   // [2][3][66][65]
    float* WsB = smem + 2 * C3_HALO_PITCH;               // [2][64][64]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int segs = p.W / C3_PX, cob = p.Cout / 64, nch = p.Cin / 64;
    const int ntiles = p.B * p.H * segs * cob;
    const int G = gridDim.x;
    auto decode = [&](int tile, int& n, int& h, int& w0, int& co0) {
        int bid = tile;
        const int cb = bid % cob; bid /= cob;
        const int seg = bid % segs; bid /= segs;
        h = bid % p.H; n = bid / p.H;
        w0 = seg * C3_PX; co0 = cb * 64;
    };
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    int n, h, w0, co0;
    decode(tile, n, h, w0, co0);
    // ---- prologue (all 5 waves): first halo + first filter slice ----
    for (int i = t; i < 9 * C3_SLICE; i += 320) {
        const int c4 = i & 15, pp = i >> 4;
        const int r = pp / C3_HW, px = pp - r * C3_HW;
        const int hh = h + r - 1, ww = w0 + px - 1;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (hh >= 0 && hh < p.H && ww >= 0 && ww < p.W)
            v = *reinterpret_cast<const f32x4*>(p.x + (((long)n * p.H + hh) * p.W + ww) * p.Cin + 4 * c4);
        float* d = XsB + pp * C3_XP + 4 * c4;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    for (int i = t; i < 1024; i += 320)
        *reinterpret_cast<f32x4*>(WsB + (i >> 4) * 64 + 4 * (i & 15)) =
            *reinterpret_cast<const f32x4*>(p.w + (long)(i >> 4) * p.Cout + co0 + 4 * (i & 15));
    __syncthreads();

    int xbuf = 0, wbuf = 0, ch = 0;
    if (wave == 4) {
        // ================================ loader wave ================================
        while (true) {
            int ntile = tile, nchk = ch + 1;
            if (nchk == nch) { nchk = 0; ntile = tile + G; }
            const bool has_next = ntile < ntiles;
            int nn = n, nh = h, nw0 = w0, nco0 = co0;
            if (has_next && nchk == 0) decode(ntile, nn, nh, nw0, nco0);
            float* XsN = XsB + (xbuf ^ 1) * C3_HALO_PITCH;
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const bool more_w = tap < 8 || has_next;
                f32x4 wv[16], hv[6];
                if (more_w) {
                    const int ntap = tap < 8 ? tap + 1 : 0;
                    const int wci = tap < 8 ? ch * 64 : nchk * 64;
                    const int wco = tap < 8 ? co0 : nco0;
                    const float* src = p.w + ((long)ntap * p.Cin + wci) * p.Cout + wco;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int idx = lane + 64 * q;
                        wv[q] = *reinterpret_cast<const f32x4*>(src + (long)(idx >> 4) * p.Cout + 4 * (idx & 15));
                    }
                }
                // halo slice of the next item: row r = tap/3, pixels (tap%3)*22 .. +21  (66 = 3 x 22: no division per element)
                const int r = tap / 3, pxb = (tap - 3 * r) * 22;
                const int hh = nh + r - 1;
                const bool row_ok = has_next && hh >= 0 && hh < p.H;
                if (has_next) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        const int idx = lane + 64 * q;
                        const int px = pxb + (idx >> 4), ww = nw0 + px - 1;
                        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (idx < C3_SLICE && row_ok && ww >= 0 && ww < p.W)
                            v = *reinterpret_cast<const f32x4*>(p.x + (((long)nn * p.H + hh) * p.W + ww) * p.Cin + nchk * 64 +
                                                                4 * (idx & 15));
                        hv[q] = v;
                    }
                }
                if (more_w) {
                    float* Wd = WsB + (wbuf ^ 1) * 4096;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int idx = lane + 64 * q;
                        *reinterpret_cast<f32x4*>(Wd + (idx >> 4) * 64 + 4 * (idx & 15)) = wv[q];
                    }
                }
                if (has_next) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        const int idx = lane + 64 * q;
                        if (idx < C3_SLICE) {
                            float* d = XsN + (r * C3_HW + pxb + (idx >> 4)) * C3_XP + 4 * (idx & 15);
                            d[0] = hv[q][0]; d[1] = hv[q][1]; d[2] = hv[q][2]; d[3] = hv[q][3];
                        }
                    }
                }
                __syncthreads();
                wbuf ^= 1;
            }
            if (!has_next) break;
            tile = ntile; ch = nchk; n = nn; h = nh; w0 = nw0; co0 = nco0;
            xbuf ^= 1;
        }
        return;
    }
    // ================================ MFMA waves ================================
    const int wm = wave & 1, wn = wave >> 1;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    while (true) {
        int ntile = tile, nchk = ch + 1;
        if (nchk == nch) { nchk = 0; ntile = tile + G; }
        const bool has_next = ntile < ntiles;
        const float* Xs = XsB + xbuf * C3_HALO_PITCH;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - 3 * kh;
            const float* arow = Xs + (kh * C3_HW + wm * 32 + (lane & 31) + kw) * C3_XP + (lane >> 5);
            const float* brow = WsB + wbuf * 4096 + (lane >> 5) * 64 + wn * 32 + (lane & 31);
            float ra[4][8], rb[4][8];
            C3_LOADG(0)
            __builtin_amdgcn_sched_barrier(0);
            C3_LOADG(1)
            __builtin_amdgcn_sched_barrier(0);
            C3_MFMAG(0)
            __builtin_amdgcn_sched_barrier(0);
            C3_LOADG(2)
            __builtin_amdgcn_sched_barrier(0);
            C3_MFMAG(1)
            __builtin_amdgcn_sched_barrier(0);
            C3_LOADG(3)
            __builtin_amdgcn_sched_barrier(0);
            C3_MFMAG(2)
            __builtin_amdgcn_sched_barrier(0);
            C3_MFMAG(3)
            __builtin_amdgcn_sched_barrier(0);
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
        if (nchk == 0) decode(ntile, n, h, w0, co0);
        tile = ntile; ch = nchk;
        xbuf ^= 1;
    }
}

// ---- v5: v2 (persistent, in-wave prefetch) with 16-BYTE LDS operand reads ---------------------------------------------------------
// Measured: v1..v4 all stall at ~45 % MFMA utilisation however the global traffic is organised -- the limiter is the LDS
// *instruction* rate: with one wave per SIMD a 4-byte-per-lane ds_read reaches only ~1/5 of its peak (MI355X_MICROARCH.md,
// LDS), and the 32x32x2 MFMA wants two of them every 64 cycles.  Here both operands are stored with the CONTRACTION axis
// contiguous (halo [row][px][68], filter slice transposed [co][68]; pitch 68 = conflict-free for ds_read_b128), each lane
// reads 4 consecutive input channels per 16-byte load and feeds 4 MFMAs with it: lane (i, kq = lane>>5) uses channels
// 8c + 4kq + u for u = 0..3 on BOTH operands, which is all the contraction needs.  8x fewer LDS instructions.
// Template CK = input channels per work item: 64 (one work-group per CU, 142.5 KB LDS) or 32 (75.5 KB LDS => TWO work-groups per
// CU: the in-kernel cycle breakdown (tools/conv3_prof.py) shows a wave spends 2.17k cycles per tap in its 32 MFMAs and another
// ~1.6k in load issue, LDS publishes, barrier and epilogue that a lone wave per SIMD cannot overlap; a second resident
// work-group fills those gaps).
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
template <int CK, bool PROF>
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
    long long pc[6] = {0, 0, 0, 0, 0, 0};       // issue-loads | mfma block | publish (vmcnt wait + LDS writes) | barrier | epilogue | total
    constexpr bool prof = PROF;
    const long long tstart = prof ? clock64() : 0;
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
            const long long c0 = prof ? clock64() : 0;
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
            const long long c1 = prof ? clock64() : 0;
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
            const long long c2 = prof ? clock64() : 0;
            if (more_w) store_w(WsB + (wbuf ^ 1) * C5_WT);
            if (has_next) {
                if (t < SLICE) halo_store(XsN, r, pxb, t, h0);
                if (SLICE > 256 && t < SLICE - 256) halo_store(XsN, r, pxb, 256 + t, h1);
            }
            const long long c3 = prof ? clock64() : 0;
            __syncthreads();
            if (prof) { const long long c4 = clock64(); pc[0] += c1 - c0; pc[1] += c2 - c1; pc[2] += c3 - c2; pc[3] += c4 - c3; }
            wbuf ^= 1;
        }
        const long long e0 = prof ? clock64() : 0;
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
        if (prof) pc[4] += clock64() - e0;
        if (!has_next) break;
        tile = ntile; ch = nchk; n = nn; h = nh; w0 = nw0; co0 = nco0;
        xbuf ^= 1;
    }
    if (prof && lane == 0) {
        pc[5] = clock64() - tstart;   // (only the first 256 work-groups report)
        if (blockIdx.x < 256) for (int i = 0; i < 6; ++i) p.prof[((long)blockIdx.x * 4 + wave) * 6 + i] = pc[i];
    }
}

#define C3_FWD_LDS ((3 * C3_HW * C3_XP + 2 + 64 * 64) * 4)
// x (B,H,W,Cin) NHWC contiguous; w = packed [9][Cin][Cout]; y (B,H,W,Cout); Cin, Cout, W multiples of 64
TATT_API int tatt_conv3_c64_fwd(const float* x, const float* wpacked, const float* bias, float* y, int B, int H, int W,
                                int Cin, int Cout, int act, float beta, hipStream_t st) {
    if (Cin % 64 || Cout % 64 || W % C3_PX) return 1;
    Conv3P p = {x, wpacked, bias, y, B, H, W, Cin, Cout, act, beta, nullptr};
    static bool attr_set = false;
    static int variant = 4;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C3_FWD_LDS);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_fwd_v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C3_V2_LDS);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_fwd_v3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C3_V3_LDS);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_fwd_v4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C3_V2_LDS);
        const char* e = getenv("TATT_CONV3_VARIANT");       // 1 = one tile per work-group, 2 = persistent pipelined (4 waves), 3 = persistent, 8 waves (measured slower: 60.6 vs 54.4 us), 4 = 4 MFMA waves + 1 loader wave (default)
        if (e) variant = atoi(e);
        attr_set = true;
    }
    const int ntiles = B * H * (W / C3_PX) * (Cout / 64);
    if (variant == 1) {
        hipLaunchKernelGGL(conv3_c64_fwd_kernel, dim3(ntiles), dim3(256), C3_FWD_LDS, st, p);
    } else if (variant == 2) {
        const int G = ntiles < 256 ? ntiles : 256;
        hipLaunchKernelGGL(conv3_c64_fwd_v2_kernel, dim3(G), dim3(256), C3_V2_LDS, st, p);
    } else if (variant == 3) {
        const int G = ntiles < 256 ? ntiles : 256;
        hipLaunchKernelGGL(conv3_c64_fwd_v3_kernel, dim3(G), dim3(512), C3_V3_LDS, st, p);
    } else {
        const int G = ntiles < 256 ? ntiles : 256;
        hipLaunchKernelGGL(conv3_c64_fwd_v4_kernel, dim3(G), dim3(320), C3_V2_LDS, st, p);
    }
    return LAUNCH_CHECK();
}

// x (B,H,W,Cin) NHWC contiguous; wt = filter packed [9][Cout][Cin] (tatt_repack_conv_weight mode 2; mode 3 for the data
// gradient); y (B,H,W,Cout)
static long long* g_conv3_prof = nullptr;
// diagnostics: cycle breakdown buffer (256 work-groups x 4 waves x 6 counters) filled by the next conv3 launches; NULL disables
TATT_API int tatt_conv3_set_prof(long long* buf) { g_conv3_prof = buf; return 0; }

TATT_API int tatt_conv3_c64_fwd_t(const float* x, const float* wt, const float* bias, float* y, int B, int H, int W, int Cin,
                                  int Cout, int act, float beta, hipStream_t st) {
    if (Cin % 64 || Cout % 64 || W % C3_PX) return 1;
    Conv3P p = {x, wt, bias, y, B, H, W, Cin, Cout, act, beta, g_conv3_prof};
    static bool attr_set = false;
    static int ck = 32;
    if (!attr_set) {
#define C5_ATTR(CKV, PV) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_fwd_v5_kernel<CKV, PV>), \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, C5<CKV>::LDS);
        C5_ATTR(64, false) C5_ATTR(32, false) C5_ATTR(16, false) C5_ATTR(64, true) C5_ATTR(32, true) C5_ATTR(16, true)
        const char* e = getenv("TATT_CONV3_CK");   // 64: one work-group per CU; 32 (default): two per CU; 16: three per CU
        if (e) ck = atoi(e);
        attr_set = true;
    }
    const int ntiles = B * H * (W / C3_PX) * (Cout / 64);
    const int per_cu = ck == 64 ? 1 : (ck == 16 ? 3 : 2);
    const int G = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
#define C5_LAUNCH(CKV)                                                                                              \
    if (p.prof) hipLaunchKernelGGL((conv3_c64_fwd_v5_kernel<CKV, true>), dim3(G), dim3(256), C5<CKV>::LDS, st, p); \
    else hipLaunchKernelGGL((conv3_c64_fwd_v5_kernel<CKV, false>), dim3(G), dim3(256), C5<CKV>::LDS, st, p);
    if (ck == 64) { C5_LAUNCH(64) } else if (ck == 16) { C5_LAUNCH(16) } else { C5_LAUNCH(32) }
    return LAUNCH_CHECK();
}

// ---- weight-stationary forward (Cin == 64) ---------------------------------------------------------------------------
// Each wave keeps the WHOLE filter of its 32 output channels (9 taps x 64 ci = K 576 -> 288 B-operand registers per lane) in
// the unified VGPR/AGPR file for the lifetime of the (persistent) work-group, so the main loop streams only activations:
// per 64-pixel row tile a wave issues 72 sixteen-byte LDS reads (A operand) and 288 MFMAs, the next tile's halo is
// prefetched global -> registers -> LDS underneath them, and there is ONE barrier per tile instead of one per tap.  The
// filter is read from L2 once per work-group (147 KB) instead of once per tile (113 MB per launch in the v5 kernel).
// One work-group per CU (1 wave / SIMD, 512 registers each); latency is hidden by software pipelining, not occupancy.
#define WS_XP 68                              // halo pitch (floats): 64 ci + 4 pad -> conflict-free ds_read_b128
#define WS_HALO (3 * C3_HW * WS_XP)           // floats per halo buffer (53.9 KB)
#define WS_LDS (2 * WS_HALO * 4)
__global__ __launch_bounds__(256, 1) void conv3_c64_ws_kernel(Conv3P p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int segs = p.W / C3_PX, cob = p.Cout / 64;
    const int npt = p.B * p.H * segs;                     // pixel tiles (one 64-pixel row segment each)
    // XCD-aware tile order: work-groups are dealt round-robin to the 8 XCDs (private L2 each), so XCD x = blockIdx % 8 takes
    // a CONTIGUOUS run of row tiles (the 3-row halos of neighbouring rows then hit the same L2 instead of being fetched
    // through the fabric once per XCD) and a single cout block (one filter per L2).
    const int stride = gridDim.x / cob;
    int cb = blockIdx.x % cob, pt = blockIdx.x / cob;
    if (gridDim.x % 8 == 0 && 8 % cob == 0 && stride % (8 / cob) == 0) {
        const int x = blockIdx.x & 7, m = blockIdx.x >> 3, xg = 8 / cob;
        cb = x % cob;
        pt = (x / cob) * (stride / xg) + m;
    }
    if (pt >= npt) return;
    const int co0 = cb * 64;
    // ---- the filter: 72 float4 per lane, packed by tatt_repack_conv_weight mode 4 / 5 ----
    f32x4 wq[72];
    {
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(p.w) + ((long)(cb * 2 + wn) * 72) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 72; ++q) wq[q] = wsrc[q * 64];
    }
    auto decode = [&](int tile, int& n, int& h, int& w0) {
        const int seg = tile % segs; tile /= segs;
        h = tile % p.H; n = tile / p.H; w0 = seg * C3_PX;
    };
    // float4 #idx (c4 = idx & 15, pixel = idx >> 4) of halo row r, pixels [pxb, pxb + 22)
    auto halo_load = [&](int n, int hh, int w0, int pxb, int idx) -> f32x4 {
        const int ww = w0 + pxb + (idx >> 4) - 1;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (hh >= 0 && hh < p.H && ww >= 0 && ww < p.W)
            v = *reinterpret_cast<const f32x4*>(p.x + (((long)n * p.H + hh) * p.W + ww) * 64 + 4 * (idx & 15));
        return v;
    };
    auto halo_store = [&](float* Xs, int r, int pxb, int idx, f32x4 v) {
        *reinterpret_cast<f32x4*>(Xs + (r * C3_HW + pxb + (idx >> 4)) * WS_XP + 4 * (idx & 15)) = v;
    };
    int n, h, w0;
    decode(pt, n, h, w0);
    {   // first halo: all 18 loads in flight together with the 72 filter loads (one memory round trip, not 9)
        f32x4 hp0[9], hp1[9];
#pragma unroll
        for (int s9 = 0; s9 < 9; ++s9) {
            const int r = s9 / 3, pxb = (s9 - 3 * r) * 22;
            hp0[s9] = halo_load(n, h + r - 1, w0, pxb, t);
            hp1[s9] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (t < 96) hp1[s9] = halo_load(n, h + r - 1, w0, pxb, 256 + t);
        }
#pragma unroll
        for (int s9 = 0; s9 < 9; ++s9) {
            const int r = s9 / 3, pxb = (s9 - 3 * r) * 22;
            halo_store(smem, r, pxb, t, hp0[s9]);
            if (t < 96) halo_store(smem, r, pxb, 256 + t, hp1[s9]);
        }
    }
    __syncthreads();
    const int abase = (wm * 32 + (lane & 31)) * WS_XP + 4 * (lane >> 5);
    const int co = co0 + wn * 32 + (lane & 31);
    const float bj = p.bias ? p.bias[co] : 0.f;
    int xbuf = 0;
    while (true) {
        const int npt_next = pt + stride;
        const bool has_next = npt_next < npt;
        int nn = n, nh = h, nw0 = w0;
        if (has_next) decode(npt_next, nn, nh, nw0);
        const float* Xs = smem + xbuf * WS_HALO + abase;
        float* XsN = smem + (xbuf ^ 1) * WS_HALO;
        f32x16 acc, acc1;            // two independent accumulation chains (even / odd groups): no back-to-back dependent MFMAs
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc1[i] = 0.f; }
        f32x4 va[8];
        f32x4 h0, h1;
        // group g = tap * 8 + c covers input channels [8c, 8c+8) of tap (kh, kw): one 16-byte A read, four MFMAs
#define WS_AOFF(g) ((((g) >> 3) / 3 * C3_HW + ((g) >> 3) % 3) * WS_XP + 8 * ((g) & 7))
#define WS_LD(g) va[(g) & 7] = *reinterpret_cast<const f32x4*>(Xs + WS_AOFF(g));
        WS_LD(0) WS_LD(1) WS_LD(2) WS_LD(3)
#pragma unroll
        for (int g = 0; g < 72; ++g) {
            const int tap = g >> 3, c = g & 7;
            const int r = tap / 3, pxb = (tap - 3 * r) * 22;
            if (c == 0) {                                  // this tap's 1/9 of the NEXT tile's halo: global -> registers
                h0 = (f32x4){0.f, 0.f, 0.f, 0.f}; h1 = h0;
                if (has_next) {
                    h0 = halo_load(nn, nh + r - 1, nw0, pxb, t);
                    if (t < 96) h1 = halo_load(nn, nh + r - 1, nw0, pxb, 256 + t);
                }
            }
            if (g + 4 < 72) { WS_LD(g + 4) }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[g & 7][u], wq[g][u], acc1, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[g & 7][u], wq[g][u], acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (c == 7 && has_next) {                      // ... registers -> LDS, 28 MFMAs after the loads were issued
                halo_store(XsN, r, pxb, t, h0);
                if (t < 96) halo_store(XsN, r, pxb, 256 + t, h1);
            }
        }
        {
            const long rowbase = ((long)n * p.H + h) * p.W + w0 + wm * 32;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int px = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                float v = apply_act(acc[reg] + acc1[reg] + bj, p.act);
                const long o = (rowbase + px) * p.Cout + co;
                if (p.beta != 0.f) v += p.beta * p.y[o];
                p.y[o] = v;
            }
        }
        if (!has_next) break;
        __syncthreads();
        pt = npt_next; n = nn; h = nh; w0 = nw0;
        xbuf ^= 1;
    }
}

// x (B,H,W,64) NHWC contiguous; wl = filter in the per-lane register order (tatt_repack_conv_weight mode 4; mode 5 for the
// data gradient of a 64-output-channel convolution); y (B,H,W,Cout)
TATT_API int tatt_conv3_c64_fwd_ws(const float* x, const float* wl, const float* bias, float* y, int B, int H, int W,
                                   int Cout, int act, float beta, hipStream_t st) {
    if (Cout % 64 || W % C3_PX) return 1;
    Conv3P p = {x, wl, bias, y, B, H, W, 64, Cout, act, beta, nullptr};
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
        attr_set = true;
    }
    const int cob = Cout / 64, npt = B * H * (W / C3_PX);
    int per = 256 / cob;
    if (per > npt) per = npt;
    hipLaunchKernelGGL(conv3_c64_ws_kernel, dim3(per * cob), dim3(256), WS_LDS, st, p);
    return LAUNCH_CHECK();
}

// ---- weight gradient -------------------------------------------------------------------------------------------------
struct Conv3WP {
    const float* x; const float* dy; float* part;
    int B, H, W, Cin, Cout, nseg;
};
__global__ __launch_bounds__(256) void conv3_c64_wgrad_kernel(Conv3WP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*Xs)[C3_HW][64] = reinterpret_cast<float (*)[C3_HW][64]>(smem);                  // [3][66][64]
    float (*Ds)[64] = reinterpret_cast<float (*)[64]>(smem + 3 * C3_HW * 64);               // [64 px][64 co]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int qi = wave & 1, qo = wave >> 1;             // (ci half, co half) quadrant of this wave
    const int cib = blockIdx.y % (p.Cin / 64), cob = blockIdx.y / (p.Cin / 64);
    const int ci0 = cib * 64, co0 = cob * 64;
    const int segs = p.W / C3_PX;
    f32x16 acc[9];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;

    for (int s = blockIdx.x; s < p.nseg; s += gridDim.x) {
        int bid = s;
        const int seg = bid % segs; bid /= segs;
        const int h = bid % p.H; const int n = bid / p.H;
        const int w0 = seg * C3_PX;
        __syncthreads();
        for (int i = t; i < 3 * C3_HW * 16; i += 256) {
            const int c4 = i & 15, pp = i >> 4;
            const int r = pp / C3_HW, px = pp - r * C3_HW;
            const int hh = h + r - 1, ww = w0 + px - 1;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (hh >= 0 && hh < p.H && ww >= 0 && ww < p.W)
                v = *reinterpret_cast<const f32x4*>(p.x + (((long)n * p.H + hh) * p.W + ww) * p.Cin + ci0 + 4 * c4);
            *reinterpret_cast<f32x4*>(&Xs[r][px][4 * c4]) = v;
        }
        for (int i = t; i < 64 * 16; i += 256) {
            const int c4 = i & 15, px = i >> 4;
            *reinterpret_cast<f32x4*>(&Ds[px][4 * c4]) =
                *reinterpret_cast<const f32x4*>(p.dy + (((long)n * p.H + h) * p.W + w0 + px) * p.Cout + co0 + 4 * c4);
        }
        __syncthreads();
        // A(i = ci, k = pixel) = Xs[kh][pixel + kw][ci];  B(k = pixel, j = co) = Ds[pixel][co]
        const int kq = lane >> 5;
        const float* bcol = &Ds[kq][qo * 32 + (lane & 31)];
        const float* acol = &Xs[0][kq][qi * 32 + (lane & 31)];
        // 32 k-steps (pixel pairs) x 9 independent accumulators; operands of step k+1 are read (pinned by sched_barrier)
        // before the 9 MFMAs of step k issue, so the LDS latency hides under 576 cycles of matrix work.
        float wa[2][9], wb[2];
#define C3W_LOAD(buf, k)                                                                   \
        wb[buf] = bcol[(k) * 64];                                                          \
        _Pragma("unroll") for (int tap = 0; tap < 9; ++tap)                                 \
            wa[buf][tap] = acol[((tap / 3) * C3_HW + (k) + (tap - 3 * (tap / 3))) * 64];
        C3W_LOAD(0, 0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 64; k += 2) {
            const int cur = (k >> 1) & 1;
            if (k + 2 < 64) { C3W_LOAD(cur ^ 1, k + 2) }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cur][tap], wb[cur], acc[tap], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // partial[blockIdx.x][(tap*Cin + ci)][Cout]
    float* P = p.part + (long)blockIdx.x * 9 * p.Cin * p.Cout;
    const int co = co0 + qo * 32 + (lane & 31);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int ci = ci0 + qi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            P[((long)tap * p.Cin + ci) * p.Cout + co] = acc[tap][reg];
        }
}
#define C3_WG_LDS ((3 * C3_HW * 64 + 64 * 64) * 4)
// partials: part[G][9*Cin][Cout] with G = *nblocks_out work-groups along x (<= 256); reduce with the split-K reducer
TATT_API int tatt_conv3_c64_wgrad_partial(const float* x, const float* dy, float* part, int B, int H, int W, int Cin,
                                          int Cout, int G, hipStream_t st) {
    if (Cin % 64 || Cout % 64 || W % C3_PX) return 1;
    const int nseg = B * H * (W / C3_PX);
    Conv3WP p = {x, dy, part, B, H, W, Cin, Cout, nseg};
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C3_WG_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL(conv3_c64_wgrad_kernel, dim3(G, (Cin / 64) * (Cout / 64)), dim3(256), C3_WG_LDS, st, p);
    return LAUNCH_CHECK();
}
