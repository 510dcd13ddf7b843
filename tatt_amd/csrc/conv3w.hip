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

// ---- round 6: square tiles, hardware-transposed operand reads, staging waves beside MFMA waves ------------------------------------------
// The kernel above (46.9 us per launch in the step, 0.035 of the bf16 pipe) transposes every operand with scalar LDS reads behind seven
// work-group barriers per 64-pixel segment, and all of its 8 waves do everything in lock step.  What the forward kernel's rebuild
// measured (profiles/r06_conv3_sb2_stamps.txt) applies here as well: one wave per SIMD cannot issue staging arithmetic AND feed the MFMA
// pipe; the CU's memory pipe takes ~37 cycles per 1 KB request; v_mfma_f32_32x32x16_bf16 is the faster form.  Here:
//   * tile = 4 rows x 16 pixels (as the forward kernel: x halo 6 x 18); one tile row = the 16 pixels of ONE MFMA's contraction;
//   * both maps are staged as they lie in memory -- pixel-major, channel-contiguous -- as a hi and a lo bf16 image (pitch 192 B per
//     pixel), and the MFMA operands (8 consecutive PIXELS of one channel per lane) come out of ds_read_b64_tr_b16, gfx950's transposing
//     LDS read: lane a of a 16-lane group points at the piece (pixel a / 4, channel quad a % 4) of a [4 pixels][16 channels] block and
//     receives channel a of the four pixels (tools/ubench/tr_probe.hip); with the 192-byte pitch the four pixel rows of a read fall into
//     disjoint bank quarters;
//   * waves 0-3 only read operands and issue MFMAs: wave (cih, coh) owns the (32 ci x 32 co) quadrant of all nine taps (9 accumulators,
//     144 registers) for the whole launch; waves 4-7 only stage: global -> registers (requested a tile ahead) -> split -> LDS, and sum
//     the bias gradient in fp32 from the dy values passing through; the two kinds meet at one barrier per tile (double-buffered images);
//   * the partial slab leaves as whole 128-byte lines (each accumulator transposed through a wave-private LDS tile).
// Same contract and outputs as the kernel above.  Geometries with H % 4 != 0 keep it.
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 cw_bf16x4_t;
typedef __attribute__((address_space(3))) cw_bf16x4_t* cw_lds_b64;
typedef float cw_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned cw_u32x4 __attribute__((ext_vector_type(4)));
#define W2_PB 192                                   // bytes per pixel of an image (64 bf16 + 64 B: the bank spread of the transposing reads)
#define W2_XIMG (108 * W2_PB)                       // x halo image (hi or lo): 20,736 B
#define W2_DIMG (64 * W2_PB)                        // dy tile image: 12,288 B
#define W2_BUF (2 * W2_XIMG + 2 * W2_DIMG)          // one buffer: x hi, x lo, dy hi, dy lo = 66,048 B
#define W2_LDS (2 * W2_BUF)                         // 132,096 B
#define W2_OOB 0x80000000u
__device__ __forceinline__ cw_bf16x8 w2_tr(const char* p0, int off) {        // 8 consecutive pixels of this lane's channel: two transposing reads
    const cw_bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((cw_lds_b64)(p0 + off));
    const cw_bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((cw_lds_b64)(p0 + off + 4 * W2_PB));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ void w2_split_put(char* dst_hi, int lo_off, f32x4 v) {
    const cw_bf16x2 h0 = __builtin_convertvector((cw_f32x2){v[0], v[1]}, cw_bf16x2), h1 = __builtin_convertvector((cw_f32x2){v[2], v[3]}, cw_bf16x2);
    const cw_f32x2 r0 = (cw_f32x2){v[0], v[1]} - __builtin_convertvector(h0, cw_f32x2), r1 = (cw_f32x2){v[2], v[3]} - __builtin_convertvector(h1, cw_f32x2);
    const cw_bf16x2 l0 = __builtin_convertvector(r0, cw_bf16x2), l1 = __builtin_convertvector(r1, cw_bf16x2);
    *reinterpret_cast<uint2*>(dst_hi) = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
    *reinterpret_cast<uint2*>(dst_hi + lo_off) = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
}
__global__ __launch_bounds__(512, 2) void conv3_c64_wgrad_sb2_kernel(Conv3WSP p) {
    extern __shared__ __attribute__((aligned(16))) char w2_smem[];
    __shared__ float dbred[16][64];
    const unsigned t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned cib = blockIdx.y % (p.Cin / 64), cob = blockIdx.y / (p.Cin / 64);
    const unsigned ci0 = cib * 64, co0 = cob * 64;
    const unsigned W = p.W, H = p.H, tws = (W + 15) / 16, ths = H / 4;        // (W may be ragged: the last tile column is cut by the map)
    const unsigned per = (p.nseg + gridDim.x - 1) / gridDim.x;
    const unsigned s_beg = blockIdx.x * per, s_end = min((unsigned)p.nseg, s_beg + per);
    const bool want_db = p.pdb != nullptr && cib == 0;
    // -> origin pixel index; te: bit 0 / 1 / 2 = the tile touches the top / bottom / left edge, bits 8-15 = its halo columns inside the map
    auto decode = [&](unsigned tile, unsigned& org, unsigned& te) {
        const unsigned tw = tile % tws; tile /= tws;
        const unsigned th = tile % ths, n = tile / ths;
        org = (n * H + th * 4) * W + tw * 16;
        te = (th == 0 ? 1u : 0u) | (th == ths - 1 ? 2u : 0u) | (tw == 0 ? 4u : 0u) | (min(18u, W - tw * 16 + 1) << 8);
    };
    if (wave >= 4) {
        // ---------------- staging waves: 256 threads; x halo = 1728 16-byte items (7 per thread), dy tile = 1024 (4 per thread) ----------------
        const unsigned h = t - 256, c4 = h & 15;
        const unsigned npix = p.B * H * W;
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, npix * p.Cin * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, npix * p.Cout * 4, 0x00020000);
        const unsigned xps = p.Cin * 4, dps = p.Cout * 4;
        unsigned xoff[7], xput[7], edge[2] = {0, 0}, colk[2] = {0, 0}, doff[4], dput[4];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const unsigned pix = (h >> 4) + 16 * k, r = pix / 18, c = pix - r * 18;
            xoff[k] = (unsigned)(((int)r - 1) * (int)W + (int)c - 1) * xps + (ci0 + 4 * c4) * 4;
            xput[k] = pix * W2_PB + 8 * c4;
            edge[k >> 2] |= ((r == 0 ? 1u : 0u) | (r == 5 ? 2u : 0u) | (c == 0 ? 4u : 0u) | (pix >= 108 ? 8u : 0u)) << (8 * (k & 3));
            colk[k >> 2] |= c << (8 * (k & 3));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned pix = (h >> 4) + 16 * k;                           // tile pixel: row k, column h >> 4
            doff[k] = (k * W + (h >> 4)) * dps + (co0 + 4 * c4) * 4;
            dput[k] = 2 * W2_XIMG + pix * W2_PB + 8 * c4;
        }
        f32x4 hx[7], hd[4], dbacc = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto issue = [&](unsigned tile, bool valid) {
            unsigned org, te;
            decode(tile, org, te);
            const unsigned inv = valid ? 0u : W2_OOB;
#pragma unroll
            for (int k = 0; k < 7; ++k) {                    // outside the map: an edge the item sits on, no item, or a column beyond the map's last
                const bool out = (((edge[k >> 2] >> (8 * (k & 3))) & (te | 8u) & 15u) != 0) || (((colk[k >> 2] >> (8 * (k & 3))) & 255u) >= (te >> 8));
                hx[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (org * xps + xoff[k]) | (out ? W2_OOB : inv), 0, 0));
            }
            const unsigned dinv = ((h >> 4) + 1 < (te >> 8)) ? inv : W2_OOB;     // (tile column h >> 4 = halo column + 1)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                hd[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, (org * dps + doff[k]) | dinv, 0, 0));
        };
        auto put = [&](char* buf) {
#pragma unroll
            for (int k = 0; k < 7; ++k)
                if (k < 6 || h < 1728 - 6 * 256) w2_split_put(buf + xput[k], W2_XIMG, hx[k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) { w2_split_put(buf + dput[k], W2_DIMG, hd[k]); dbacc += hd[k]; }
        };
        if (s_beg < s_end) {
            issue(s_beg, true);
            put(w2_smem);
            issue(s_beg + 1, s_beg + 1 < s_end);
        }
        __syncthreads();                                                      // tile s_beg staged
        for (unsigned s = s_beg; s < s_end; ++s) {
            // while the MFMA waves work on tile s: stage tile s + 1 (requested a tile ago) into the other buffer, request tile s + 2
            if (s + 1 < s_end) {
                put(w2_smem + ((s + 1 - s_beg) & 1) * W2_BUF);
                issue(s + 2, s + 2 < s_end);
            }
            __syncthreads();
        }
        if (want_db) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dbred[h >> 4][4 * c4 + e] = dbacc[e];
        }
        __syncthreads();                                                      // (the MFMA waves' last barrier too)
        if (want_db && h < 64) {
            float a = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) a += dbred[g][h];
            p.pdb[(long)blockIdx.x * p.Cout + co0 + h] = a;
        }
        return;
    }
    // ---------------- MFMA waves: (cih, coh) quadrant of all nine taps ----------------
    const unsigned cih = wave & 1, coh = wave >> 1, a16 = lane & 15, g16 = (lane >> 4) & 1, kb = lane >> 5;
    cw_f32x16 acc[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tp][i] = 0.f;
    // this lane's piece of the [4 pixels][16 channels] blocks: pixel 8 kb + a16 / 4 (+ 4 for the second read), channel quad a16 % 4 of
    // channels 32 half + 16 g16 ..
    const unsigned la = (8 * kb + (a16 >> 2)) * W2_PB + (32 * cih + 16 * g16) * 2 + (a16 & 3) * 8;
    const unsigned lb = 2 * W2_XIMG + (8 * kb + (a16 >> 2)) * W2_PB + (32 * coh + 16 * g16) * 2 + (a16 & 3) * 8;
    __syncthreads();                                                          // tile s_beg staged
    for (unsigned s = s_beg; s < s_end; ++s) {
        const char* buf = w2_smem + ((s - s_beg) & 1) * W2_BUF;
        const char* pa = buf + la;
        const char* pb = buf + lb;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                         // tile row r = 16 pixels of contraction
            const cw_bf16x8 bh = w2_tr(pb, r * 16 * W2_PB), bl = w2_tr(pb, r * 16 * W2_PB + W2_DIMG);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                cw_bf16x8 ah[3], al[3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    ah[kx] = w2_tr(pa, ((r + ky) * 18 + kx) * W2_PB);
                    al[kx] = w2_tr(pa, ((r + ky) * 18 + kx) * W2_PB + W2_XIMG);
                }
                // D[ci][co] += x^T[ci][pixel] dy[pixel][co]; the three MFMAs of an accumulator are three issues apart
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kx], bh, acc[ky * 3 + kx], 0, 0, 0);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kx], bl, acc[ky * 3 + kx], 0, 0, 0);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kx], bh, acc[ky * 3 + kx], 0, 0, 0);
            }
        }
        __syncthreads();                                                      // next tile staged; everyone has left this tile's images
    }
    __syncthreads();                                                          // (the staging waves' bias-gradient hand-over)
    // ---- partial[blockIdx.x][tap * Cin + ci][Cout]: accumulator register 4 q + e of lane (j = lane & 31, kb) = row ci = 8 q + 4 kb + e, column
    // co = j of the quadrant; transposed through a wave-private LDS tile [32 ci][36] and stored as 128-byte rows, 8 rows per instruction ----
    float* P = p.part + (long)blockIdx.x * 9 * p.Cin * p.Cout;
    float* T = reinterpret_cast<float*>(w2_smem) + wave * (32 * 36);
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
        for (int v = 0; v < 16; ++v) T[((v & 3) + 8 * (v >> 2) + 4 * kb) * 36 + (lane & 31)] = acc[tp][v];
        wave_lds_sync();
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int row = (lane >> 3) + 8 * qd, cq4 = lane & 7;
            const f32x4 v = *reinterpret_cast<const f32x4*>(T + row * 36 + 4 * cq4);
            *reinterpret_cast<f32x4*>(P + ((long)tp * p.Cin + ci0 + 32 * cih + row) * p.Cout + co0 + 32 * coh + 4 * cq4) = v;
        }
        wave_lds_sync();
    }
}
static int conv3_wgrad_sb_generation = 2;                 // A/B hook (tatt_conv3_wgrad_sb_generation): 1 = the row-segment kernel of rounds 3-5
TATT_API int tatt_conv3_wgrad_sb_generation(int gen) {
    const int old = conv3_wgrad_sb_generation;
    if (gen == 1 || gen == 2) conv3_wgrad_sb_generation = gen;
    return old;
}

// same contract as tatt_conv3_c64_wgrad_partial (conv3.hip): partials part[G][9*Cin][Cout] (+ pdb[G][Cout]) for the split-K reducer
TATT_API int tatt_conv3_c64_wgrad_partial_sb(const float* x, const float* dy, float* part, float* pdb, int B, int H, int W,
                                             int Cin, int Cout, int G, hipStream_t st) {
    if (Cin % 64 || Cout % 64) return 1;
    const long maxb = (long)B * H * W * (Cin > Cout ? Cin : Cout) * 4;       // (32-bit buffer offsets)
    if (conv3_wgrad_sb_generation == 2 && H % 4 == 0 && maxb < 0x7fffffffL) {
        const int ntile = B * (H / 4) * ((W + 15) / 16);
        Conv3WSP p2 = {x, dy, part, B, H, W, Cin, Cout, ntile, pdb};
        static TattPerDevice attr2_once;
        tatt_per_device(attr2_once, [&] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_wgrad_sb2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS);
        });
        hipLaunchKernelGGL(conv3_c64_wgrad_sb2_kernel, dim3(G, (Cin / 64) * (Cout / 64)), dim3(512), W2_LDS, st, p2);
        return LAUNCH_CHECK();
    }
    if (W % CW_PX) return 1;
    const int nseg = B * H * (W / CW_PX);
    Conv3WSP p = {x, dy, part, B, H, W, Cin, Cout, nseg, pdb};
    static TattPerDevice attr_once;
    tatt_per_device(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_wgrad_sb_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CW_LDS);
    });
    hipLaunchKernelGGL(conv3_c64_wgrad_sb_kernel, dim3(G, (Cin / 64) * (Cout / 64)), dim3(512), CW_LDS, st, p);
    return LAUNCH_CHECK();
}
