// fp32 MFMA GEMM / implicit-GEMM convolution for the TATT hot path (gfx950).
//
// One kernel template computes  C(i,j) = epilogue( sum_r A(i,r) * B(r,j) )  on v_mfma_f32_32x32x2_f32
// (exact fp32, 64 FLOP/clk/SIMD).  A is a *virtual* matrix:
//   AMODE 0  strided            A[i*sam + r*sak]                       (Linear layers, 1x1 convs)
//   AMODE 2  K-concat           r < K1 ? A[...] : A2[i*sa2m+(r-K1)*sa2k] (1x1 conv over cat[res, tp_map])
//   AMODE 3  im2col             i = output pixel, r = (kh,kw,ci)        (conv forward / data-gradient)
//   AMODE 4  im2col transposed  i = (kh,kw,ci),   r = pixel             (conv weight-gradient)
// B and C are strided.  Work-group tile 64x64, 4 waves (2x2) each owning one 32x32 accumulator;
// K is consumed in chunks of 16 staged through double-buffered LDS (A stored k-major so that the
// MFMA operand read  lane -> (row = lane&31, k = lane>>5)  is bank-conflict free).
// Split-K (grid.y) writes raw partial tiles that tatt_splitk_reduce sums deterministically.
#include "common.h"
#include <mutex>
#include <unordered_map>
#include <stdlib.h>

#define BM 64
#define BN 64
#define KC 16
#define LDT 65   // padded LDS leading dimension

struct GemmP {
    const float* A; const float* A2; const float* B; const float* bias; float* C;
    int M, N, K, K1;
    long sam, sak, sa2m, sa2k, sbk, sbn, scm, scn;
    long bsA, bsA2, bsB, bsC, bsBias;
    float alpha, beta;
    int act;
    int splitk, chunks_per_split;
    float* partial;
    float* rowsum;       // non-null: the row sums of A (sum_r A(i,r)) are also produced (from the A tiles already staged in LDS, by
                         // the tile_n == 0 work-groups) -- bias gradients ride along with the weight-gradient GEMM
    // conv geometry (AMODE 3/4): input addressed as X[n*xsn + h*xsh + w*xsw + c*xsc]
    int H, W, Cin, KH, KW, pad, HW, Wshift, HWshift, Cshift;
    long xsn, xsh, xsw, xsc;
};

__device__ __forceinline__ void decode_pixel(const GemmP& p, int pix, int& n, int& h, int& w) {
    int hw;
    if (p.HWshift >= 0) { n = pix >> p.HWshift; hw = pix & (p.HW - 1); }
    else { n = pix / p.HW; hw = pix - n * p.HW; }
    if (p.Wshift >= 0) { h = hw >> p.Wshift; w = hw & (p.W - 1); }
    else { h = hw / p.W; w = hw - h * p.W; }
}
__device__ __forceinline__ void decode_kidx(const GemmP& p, int k, int& kh, int& kw, int& ci) {
    int tap;
    if (p.Cshift >= 0) { tap = k >> p.Cshift; ci = k & (p.Cin - 1); }
    else { tap = k / p.Cin; ci = k - tap * p.Cin; }
    kh = tap / p.KW; kw = tap - kh * p.KW;
}

template <int AMODE, int AK, int BK, int KCT>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(GemmP p) {
    // KCT = K-chunk (16)
    constexpr int NP = KCT / 4;          // elements per thread per operand per chunk
    constexpr int KL = KCT;              // lanes along k when an operand is k-contiguous
    constexpr int RP = 256 / KCT;        // rows covered per pass in that case
    __shared__ float As[2][KCT][LDT];
    __shared__ float Bs[2][KCT][LDT];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int z = blockIdx.z;
    const float* A = p.A + (long)z * p.bsA;
    const float* A2 = (AMODE == 2) ? p.A2 + (long)z * p.bsA2 : nullptr;
    const float* B = p.B + (long)z * p.bsB;

    const int nchunks = (p.K + KCT - 1) / KCT;
    int c_begin = 0, c_end = nchunks;
    if (p.splitk > 1) {
        c_begin = blockIdx.y * p.chunks_per_split;
        c_end = min(nchunks, c_begin + p.chunks_per_split);
    }

    // per-thread fixed decodes for the im2col modes
    int pn[4], ph[4], pw[4];          // AMODE 3: the thread's 4 pixel rows
    int fkh = 0, fkw = 0, fci = 0;    // AMODE 4: the thread's fixed (kh,kw,ci) row
    bool frow_ok = false;
    if (AMODE == 3) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int i = m0 + (t >> 4) + 16 * q;
            if (i < p.M) decode_pixel(p, i, pn[q], ph[q], pw[q]);
            else { pn[q] = -1; ph[q] = 0; pw[q] = 0; }
        }
    }
    if (AMODE == 4) {
        int i = m0 + (t & 63);
        frow_ok = i < p.M;
        if (frow_ok) decode_kidx(p, i, fkh, fkw, fci);
    }

    float ra[NP], rb[NP];
    auto load_chunk = [&](int c) {
        const int k0 = c * KCT;
        // ---- A ----
        if (AMODE == 3) {
            int r = k0 + (t & 15);
            int kh = 0, kw = 0, ci = 0;
            bool rok = r < p.K;
            if (rok) decode_kidx(p, r, kh, kw, ci);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = 0.f;
                if (rok && pn[q] >= 0) {
                    int hh = ph[q] + kh - p.pad, ww = pw[q] + kw - p.pad;
                    if (hh >= 0 && hh < p.H && ww >= 0 && ww < p.W)
                        v = A[pn[q] * p.xsn + hh * p.xsh + ww * p.xsw + ci * p.xsc];
                }
                ra[q] = v;
            }
        } else if (AMODE == 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int r = k0 + (t >> 6) + 4 * q;
                float v = 0.f;
                if (frow_ok && r < p.K) {
                    int n, h, w;
                    decode_pixel(p, r, n, h, w);
                    int hh = h + fkh - p.pad, ww = w + fkw - p.pad;
                    if (hh >= 0 && hh < p.H && ww >= 0 && ww < p.W)
                        v = A[n * p.xsn + hh * p.xsh + ww * p.xsw + fci * p.xsc];
                }
                ra[q] = v;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                int i, r;
                if (AK) { r = k0 + (t & (KL - 1)); i = m0 + t / KL + RP * q; }
                else    { i = m0 + (t & 63); r = k0 + (t >> 6) + 4 * q; }
                float v = 0.f;
                if (i < p.M && r < p.K) {
                    if (AMODE == 2 && r >= p.K1) v = A2[i * p.sa2m + (long)(r - p.K1) * p.sa2k];
                    else v = A[i * p.sam + r * p.sak];
                }
                ra[q] = v;
            }
        }
        // ---- B ----
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            int j, r;
            if (BK) { r = k0 + (t & (KL - 1)); j = n0 + t / KL + RP * q; }
            else    { j = n0 + (t & 63); r = k0 + (t >> 6) + 4 * q; }
            rb[q] = (j < p.N && r < p.K) ? B[r * p.sbk + j * p.sbn] : 0.f;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (AMODE == 3 || (AMODE != 4 && AK)) As[buf][t & (KL - 1)][t / KL + RP * q] = ra[q];
            else As[buf][(t >> 6) + 4 * q][t & 63] = ra[q];
            if (BK) Bs[buf][t & (KL - 1)][t / KL + RP * q] = rb[q];
            else Bs[buf][(t >> 6) + 4 * q][t & 63] = rb[q];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const bool do_rs = p.rowsum != nullptr && tile_n == 0;
    float rs_acc = 0.f;

    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk(0);
        __syncthreads();
        for (int c = c_begin; c < c_end; ++c) {
            const int buf = (c - c_begin) & 1;
            if (c + 1 < c_end) load_chunk(c + 1);
            const int ar = wm * 32 + (lane & 31), bc = wn * 32 + (lane & 31), kq = lane >> 5;
#pragma unroll
            for (int kk = 0; kk < KCT; kk += 2) {
                float a = As[buf][kk + kq][ar];
                float b = Bs[buf][kk + kq][bc];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            if (do_rs) {
#pragma unroll
                for (int kk = 0; kk < KCT / 4; ++kk) rs_acc += As[buf][(t >> 6) * (KCT / 4) + kk][t & 63];
            }
            if (c + 1 < c_end) store_chunk(buf ^ 1);
            __syncthreads();
        }
    }
    if (do_rs) {
        // combine the 4 k-groups through LDS (the chunk buffers are free now), thread t < 64 owns row m0 + t
        As[0][t >> 6][t & 63] = rs_acc;
        __syncthreads();
        if (t < 64 && m0 + t < p.M) {
            const float v = (As[0][0][t] + As[0][1][t]) + (As[0][2][t] + As[0][3][t]);
            if (p.splitk > 1) p.partial[(long)p.splitk * p.M * p.N + (long)blockIdx.y * p.M + m0 + t] = v;
            else p.rowsum[m0 + t] = p.alpha * v;
        }
    }

    // ---- epilogue: C/D layout  col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
    const int j = n0 + wn * 32 + (lane & 31);
    if (j >= p.N) return;
    if (p.splitk > 1) {
        float* P = p.partial + ((long)(z * p.splitk + blockIdx.y) * p.M) * p.N;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            int i = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (i < p.M) P[(long)i * p.N + j] = acc[reg];
        }
        return;
    }
    float* C = p.C + (long)z * p.bsC;
    const float bj = p.bias ? p.bias[(long)z * p.bsBias + j] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        int i = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (i < p.M) {
            float v = apply_act(p.alpha * (acc[reg] + bj), p.act);
            long off = i * p.scm + j * p.scn;
            if (p.beta != 0.f) v += p.beta * C[off];
            C[off] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fast path of the strided GEMM for aligned operands (M, N multiples of 64, K of 16, 16-byte aligned rows): the same
// 64x64 tile / 4 waves / K-chunks of 16 / double-buffered LDS, but the per-element address arithmetic and bounds checks of the
// general kernel -- measured at ~580 VALU + ~320 SALU instructions per wave around 32 MFMAs on the token GEMMs -- are gone:
// every thread moves ONE 16-byte vector per operand per chunk through a pointer that advances by a constant.
//   operand contiguous along K ("KC": activations x[m][k], nn.Linear weights W[n][k]):  tile [64 rows][16 k], pitch 20 floats;
//       MFMA operands come from 16-byte LDS reads (4 consecutive k per lane, the same k <-> (lane>>5, u) map on both sides);
//   operand contiguous along its row/column index (dy^T for weight gradients, W for input gradients):  tile [16 k][64], pitch
//       96 floats (the two lane halves hit different bank halves), 4-byte operand reads.
// The C tile leaves through a per-wave LDS transpose as 16-byte stores when C is row-major.
// ------------------------------------------------------------------------------------------------
// IM2COL (requires AKC, !BKC): A is the virtual im2col matrix of an NHWC map with Cin % 16 == 0 -- a K-chunk of 16 lies inside
// one filter tap, so it is 16 contiguous channels of ONE (shifted) pixel: the thread's pixel is decoded once, the tap walks
// with wave-uniform counters, and the row count M may be ragged (rows >= M load zeros and are not stored).
// IM2COLT (requires !AKC, !BKC): the transposed im2col matrix of the weight gradient, A(i = (tap, ci), r = pixel), Cin % 64 == 0:
// the 64 rows of a tile are 64 consecutive channels of ONE tap, so a K-chunk is 16 pixels x 64 contiguous channels.
template <bool AKC, bool BKC, bool CAT, bool IM2COL, bool IM2COLT = false>
__global__ __launch_bounds__(256) void gemm_fast_kernel(GemmP p, int cvec) {
    constexpr int TKC = 64 * 20, TMC = 16 * 96;
    constexpr int TA = AKC ? TKC : TMC, TB = BKC ? TKC : TMC;
    __shared__ __attribute__((aligned(16))) float smem[2 * TA + 2 * TB];
    float* AsB = smem;
    float* BsB = smem + 2 * TA;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int tiles_n = p.N >> 6;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * 64, n0 = tile_n * 64;
    const int z = blockIdx.z;
    const int nchunks = p.K >> 4;
    int c_begin = 0, c_end = nchunks;
    if (p.splitk > 1) {
        c_begin = blockIdx.y * p.chunks_per_split;
        c_end = min(nchunks, c_begin + p.chunks_per_split);
    }
    // ---- per-thread source pointers (advance by a constant per chunk) and LDS slots ----
    const float* ap; const float* ap2 = nullptr; const float* bp;
    long astep, bstep;
    int aslot, bslot;
    // im2col state: this thread's pixel (fixed) and the wave-uniform tap walk
    int ih = 0, iw = 0, tkh = 0, tkw = 0, tcc = 0;
    bool irow = false;
    const int cpt = IM2COL ? (p.Cin >> 4) : 1;                  // chunks per tap
    if (IM2COL) {
        const int row = t >> 2, k4 = (t & 3) * 4;
        const int pix = m0 + row;
        irow = pix < p.M;
        int n = 0;
        if (irow) decode_pixel(p, pix, n, ih, iw);
        const int tap = c_begin / cpt;
        tcc = c_begin - tap * cpt; tkh = tap / p.KW; tkw = tap - tkh * p.KW;
        ap = p.A + (long)n * p.xsn + k4;
        astep = 0; aslot = row * 20 + k4;
    } else if (IM2COLT) {
        const int kk = t >> 4, c4 = (t & 15) * 4;
        const int tap = m0 / p.Cin;
        tkh = tap / p.KW; tkw = tap - tkh * p.KW;
        ap = p.A + (m0 - tap * p.Cin) + c4;
        tcc = 16 * c_begin + kk;                                   // this thread's pixel index, advances by 16 per chunk
        astep = 0; aslot = kk * 96 + c4;
    } else if (AKC) {
        const int row = t >> 2, k4 = (t & 3) * 4;
        const int rowc = min(m0 + row, p.M - 1);                  // ragged M: rows past the end re-read the last row, never stored
        ap = p.A + (long)z * p.bsA + (long)rowc * p.sam + k4 + 16L * c_begin;
        if (CAT) ap2 = p.A2 + (long)z * p.bsA2 + (long)rowc * p.sa2m + k4 + 16L * c_begin - p.K1;
        astep = 16; aslot = row * 20 + k4;
    } else {
        const int k = t >> 4, c4 = (t & 15) * 4;
        ap = p.A + (long)z * p.bsA + (long)(16 * c_begin + k) * p.sak + m0 + c4;
        astep = 16 * p.sak; aslot = k * 96 + c4;
    }
    if (BKC) {
        const int row = t >> 2, k4 = (t & 3) * 4;
        bp = p.B + (long)z * p.bsB + (long)(n0 + row) * p.sbn + k4 + 16L * c_begin;
        bstep = 16; bslot = row * 20 + k4;
    } else {
        const int k = t >> 4, c4 = (t & 15) * 4;
        bp = p.B + (long)z * p.bsB + (long)(16 * c_begin + k) * p.sbk + n0 + c4;
        bstep = 16 * p.sbk; bslot = k * 96 + c4;
    }
    f32x4 ra, rb;
    auto load_chunk = [&](int c) {
        if (IM2COL) {
            const int hh = ih + tkh - p.pad, ww = iw + tkw - p.pad;
            ra = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (irow && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W)
                ra = *reinterpret_cast<const f32x4*>(ap + (long)hh * p.xsh + (long)ww * p.xsw + 16 * tcc);
            if (++tcc == cpt) { tcc = 0; if (++tkw == p.KW) { tkw = 0; ++tkh; } }
        } else if (IM2COLT) {
            int n, h, w;
            decode_pixel(p, tcc, n, h, w);
            const int hh = h + tkh - p.pad, ww = w + tkw - p.pad;
            ra = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (hh >= 0 && hh < p.H && ww >= 0 && ww < p.W)
                ra = *reinterpret_cast<const f32x4*>(ap + (long)n * p.xsn + (long)hh * p.xsh + (long)ww * p.xsw);
            tcc += 16;
        } else if (CAT && 16 * c >= p.K1) ra = *reinterpret_cast<const f32x4*>(ap2);
        else ra = *reinterpret_cast<const f32x4*>(ap);
        rb = *reinterpret_cast<const f32x4*>(bp);
        ap += astep; bp += bstep;
        if (CAT) ap2 += 16;
    };
    auto store_chunk = [&](int buf) {
        *reinterpret_cast<f32x4*>(AsB + buf * TA + aslot) = ra;
        *reinterpret_cast<f32x4*>(BsB + buf * TB + bslot) = rb;
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const bool do_rs = !AKC && p.rowsum != nullptr && tile_n == 0;
    float rs_acc = 0.f;
    const int arow = wm * 32 + (lane & 31), bcol = wn * 32 + (lane & 31), kq = lane >> 5;
    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk(0);
        __syncthreads();
        for (int c = c_begin; c < c_end; ++c) {
            const int buf = (c - c_begin) & 1;
            if (c + 1 < c_end) load_chunk(c + 1);
            const float* As = AsB + buf * TA;
            const float* Bs = BsB + buf * TB;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 va, vb;
                if (AKC) va = *reinterpret_cast<const f32x4*>(As + arow * 20 + 8 * h + 4 * kq);
                else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) va[u] = As[(8 * h + 4 * kq + u) * 96 + arow];
                }
                if (BKC) vb = *reinterpret_cast<const f32x4*>(Bs + bcol * 20 + 8 * h + 4 * kq);
                else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) vb[u] = Bs[(8 * h + 4 * kq + u) * 96 + bcol];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[u], vb[u], acc, 0, 0, 0);
            }
            if (do_rs) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) rs_acc += As[((t >> 6) * 4 + kk) * 96 + (t & 63)];
            }
            if (c + 1 < c_end) store_chunk(buf ^ 1);
            __syncthreads();
        }
    }
    if (do_rs) {
        smem[(t >> 6) * 64 + (t & 63)] = rs_acc;
        __syncthreads();
        if (t < 64) {
            const float v = (smem[t] + smem[64 + t]) + (smem[128 + t] + smem[192 + t]);
            if (p.splitk > 1) p.partial[(long)p.splitk * p.M * p.N + (long)blockIdx.y * p.M + m0 + t] = v;
            else p.rowsum[m0 + t] = p.alpha * v;
        }
        __syncthreads();
    }
    // ---- epilogue ----
    const bool part = p.splitk > 1;
    if (part || cvec) {
        // transpose the wave's 32x32 block through LDS: lane -> (row = lane >> 3 + 8q, 4 consecutive columns)
        float* T = smem + wave * (32 * 36);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            T[((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = acc[reg];
        wave_lds_sync();
        const int c4 = lane & 7;
        const int j = n0 + wn * 32 + 4 * c4;
        f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!part && p.bias) b4 = *reinterpret_cast<const f32x4*>(p.bias + (long)z * p.bsBias + j);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (lane >> 3) + 8 * q;
            const long i = m0 + wm * 32 + r;
            if (AKC && i >= p.M) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(T + r * 36 + 4 * c4);
            if (part) {
                *reinterpret_cast<f32x4*>(p.partial + ((long)(z * p.splitk + blockIdx.y) * p.M + i) * p.N + j) = v;
            } else {
                float* dst = p.C + (long)z * p.bsC + i * p.scm + j;
                f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (p.beta != 0.f) o = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = apply_act(p.alpha * (v[e] + b4[e]), p.act) + p.beta * o[e];
                *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
        return;
    }
    const int j = n0 + wn * 32 + (lane & 31);
    float* C = p.C + (long)z * p.bsC;
    const float bj = p.bias ? p.bias[(long)z * p.bsBias + j] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const long i = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (AKC && i >= p.M) continue;
        float v = apply_act(p.alpha * (acc[reg] + bj), p.act);
        const long off = i * p.scm + j * p.scn;
        if (p.beta != 0.f) v += p.beta * C[off];
        C[off] = v;
    }
}

// Narrow-N variant of the aligned fast path: 128 x 32 tile (4 waves stacked along M) for N % 64 == 32 -- the per-head products
// of TBSRN's self-attention (d_k = 32: P.V, P^T.dO, dS.K, dS^T.Q; reference model/tbsrn.py:130-151).  Same loaders and LDS
// layouts as gemm_fast_kernel with twice the A rows per chunk; no K-concat / im2col / row sums.
template <bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_fast32_kernel(GemmP p, int cvec) {
    constexpr int TA = AKC ? 128 * 20 : 16 * 160;            // [128 rows][20] or [16 k][160]
    constexpr int TB = BKC ? 32 * 20 : 16 * 96;              // [32 cols][20]  or [16 k][96]
    __shared__ __attribute__((aligned(16))) float smem[(2 * TA + 2 * TB) > 4 * 32 * 36 ? (2 * TA + 2 * TB) : 4 * 32 * 36];
    float* AsB = smem;
    float* BsB = smem + 2 * TA;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tiles_n = p.N >> 5;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * 128, n0 = tile_n * 32;
    const int z = blockIdx.z;
    const int nchunks = p.K >> 4;
    int c_begin = 0, c_end = nchunks;
    if (p.splitk > 1) {
        c_begin = blockIdx.y * p.chunks_per_split;
        c_end = min(nchunks, c_begin + p.chunks_per_split);
    }
    const float* ap[2]; const float* bp;
    long astep, bstep;
    int aslot[2], bslot;
    const bool bload = t < 128;
    if (AKC) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = (t >> 2) + 64 * q, k4 = (t & 3) * 4;
            ap[q] = p.A + (long)z * p.bsA + (long)min(m0 + row, p.M - 1) * p.sam + k4 + 16L * c_begin;
            aslot[q] = row * 20 + k4;
        }
        astep = 16;
    } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = t + 256 * q, k = idx >> 5, c4 = (idx & 31) * 4;
            ap[q] = p.A + (long)z * p.bsA + (long)(16 * c_begin + k) * p.sak + m0 + c4;
            aslot[q] = k * 160 + c4;
        }
        astep = 16 * p.sak;
    }
    if (BKC) {
        const int row = (t & 127) >> 2, k4 = (t & 3) * 4;
        bp = p.B + (long)z * p.bsB + (long)(n0 + row) * p.sbn + k4 + 16L * c_begin;
        bstep = 16; bslot = row * 20 + k4;
    } else {
        const int k = (t & 127) >> 3, c4 = (t & 7) * 4;
        bp = p.B + (long)z * p.bsB + (long)(16 * c_begin + k) * p.sbk + n0 + c4;
        bstep = 16 * p.sbk; bslot = k * 96 + c4;
    }
    f32x4 ra[2], rb = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_chunk = [&]() {
        ra[0] = *reinterpret_cast<const f32x4*>(ap[0]);
        ra[1] = *reinterpret_cast<const f32x4*>(ap[1]);
        if (bload) rb = *reinterpret_cast<const f32x4*>(bp);
        ap[0] += astep; ap[1] += astep; bp += bstep;
    };
    auto store_chunk = [&](int buf) {
        *reinterpret_cast<f32x4*>(AsB + buf * TA + aslot[0]) = ra[0];
        *reinterpret_cast<f32x4*>(AsB + buf * TA + aslot[1]) = ra[1];
        if (bload) *reinterpret_cast<f32x4*>(BsB + buf * TB + bslot) = rb;
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int arow = wave * 32 + (lane & 31), bcol = lane & 31, kq = lane >> 5;
    if (c_begin < c_end) {
        load_chunk();
        store_chunk(0);
        __syncthreads();
        for (int c = c_begin; c < c_end; ++c) {
            const int buf = (c - c_begin) & 1;
            if (c + 1 < c_end) load_chunk();
            const float* As = AsB + buf * TA;
            const float* Bs = BsB + buf * TB;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 va, vb;
                if (AKC) va = *reinterpret_cast<const f32x4*>(As + arow * 20 + 8 * h + 4 * kq);
                else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) va[u] = As[(8 * h + 4 * kq + u) * 160 + arow];
                }
                if (BKC) vb = *reinterpret_cast<const f32x4*>(Bs + bcol * 20 + 8 * h + 4 * kq);
                else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) vb[u] = Bs[(8 * h + 4 * kq + u) * 96 + bcol];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[u], vb[u], acc, 0, 0, 0);
            }
            if (c + 1 < c_end) store_chunk(buf ^ 1);
            __syncthreads();
        }
    }
    const bool part = p.splitk > 1;
    if (part || cvec) {
        float* T = smem + wave * (32 * 36);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            T[((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = acc[reg];
        wave_lds_sync();
        const int c4 = lane & 7;
        const int j = n0 + 4 * c4;
        f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!part && p.bias) b4 = *reinterpret_cast<const f32x4*>(p.bias + (long)z * p.bsBias + j);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (lane >> 3) + 8 * q;
            const long i = m0 + wave * 32 + r;
            if (i >= p.M) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(T + r * 36 + 4 * c4);
            if (part) {
                *reinterpret_cast<f32x4*>(p.partial + ((long)(z * p.splitk + blockIdx.y) * p.M + i) * p.N + j) = v;
            } else {
                float* dst = p.C + (long)z * p.bsC + i * p.scm + j;
                f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (p.beta != 0.f) o = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = apply_act(p.alpha * (v[e] + b4[e]), p.act) + p.beta * o[e];
                *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
        return;
    }
    const int j = n0 + (lane & 31);
    float* C = p.C + (long)z * p.bsC;
    const float bj = p.bias ? p.bias[(long)z * p.bsBias + j] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const long i = m0 + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (i >= p.M) continue;
        float v = apply_act(p.alpha * (acc[reg] + bj), p.act);
        const long off = i * p.scm + j * p.scn;
        if (p.beta != 0.f) v += p.beta * C[off];
        C[off] = v;
    }
}

static inline bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }
// 0 = launched, -1 = shape/alignment not eligible (caller uses the general kernel)
static int try_gemm_fast(const GemmP& p, int Z, hipStream_t st) {
    const bool narrow = (p.N & 63) == 32;                     // 128 x 32 tiles
    if ((!narrow && (p.N & 63)) || (p.K & 15)) return -1;
    const bool akc = p.sak == 1, amc = p.sam == 1 && !akc;
    if ((p.M & (narrow ? 127 : 63)) && !akc) return -1;       // ragged M only when A is read row by row
    if (narrow && (p.A2 || p.rowsum)) return -1;
    const bool bkc = p.sbk == 1 && p.sbn != 1, bnc = p.sbn == 1;
    if (!(akc || amc) || !(bkc || bnc)) return -1;
    if (!al16(p.A) || !al16(p.B) || (p.bsA & 3) || (p.bsB & 3)) return -1;
    if (akc ? (p.sam & 3) : (p.sak & 3)) return -1;
    if (bkc ? (p.sbn & 3) : (p.sbk & 3)) return -1;
    const bool cat = p.A2 != nullptr;
    if (cat && (!akc || p.sa2k != 1 || (p.sa2m & 3) || (p.K1 & 15) || !al16(p.A2) || (p.bsA2 & 3))) return -1;
    if (p.rowsum && akc) return -1;
    if (p.splitk > 1 && !al16(p.partial)) return -1;
    const int cvec = (p.scn == 1 && !(p.scm & 3) && al16(p.C) && !(p.bsC & 3) && (!p.bias || (al16(p.bias) && !(p.bsBias & 3)))) ? 1 : 0;
    if (narrow) {
        dim3 grid32(cdiv(p.M, 128) * (p.N >> 5), p.splitk, Z);
        if (akc && bkc) hipLaunchKernelGGL((gemm_fast32_kernel<true, true>), grid32, dim3(256), 0, st, p, cvec);
        else if (akc) hipLaunchKernelGGL((gemm_fast32_kernel<true, false>), grid32, dim3(256), 0, st, p, cvec);
        else if (bkc) hipLaunchKernelGGL((gemm_fast32_kernel<false, true>), grid32, dim3(256), 0, st, p, cvec);
        else hipLaunchKernelGGL((gemm_fast32_kernel<false, false>), grid32, dim3(256), 0, st, p, cvec);
        return LAUNCH_CHECK();
    }
    dim3 grid(cdiv(p.M, 64) * (p.N >> 6), p.splitk, Z), block(256);
#define GF_LAUNCH(A_, B_, C_) hipLaunchKernelGGL((gemm_fast_kernel<A_, B_, C_, false>), grid, block, 0, st, p, cvec)
    if (cat) { if (bkc) GF_LAUNCH(true, true, true); else GF_LAUNCH(true, false, true); }
    else if (akc) { if (bkc) GF_LAUNCH(true, true, false); else GF_LAUNCH(true, false, false); }
    else { if (bkc) GF_LAUNCH(false, true, false); else GF_LAUNCH(false, false, false); }
#undef GF_LAUNCH
    return LAUNCH_CHECK();
}
// transposed-im2col fast path of tatt_conv2d_wgrad: NHWC input, Cin % 64 == 0, Cout % 64 == 0, pixel count % 16 == 0
static int try_wgrad_fast(const GemmP& p, hipStream_t st) {
    if ((p.N & 63) || (p.Cin & 63) || (p.K & 15) || p.xsc != 1 || p.sbn != 1 || (p.sbk & 3) || p.splitk < 2) return -1;
    if (!al16(p.A) || !al16(p.B) || !al16(p.partial) || (p.xsn & 3) || (p.xsh & 3) || (p.xsw & 3)) return -1;
    dim3 grid((p.M >> 6) * (p.N >> 6), p.splitk, 1), block(256);
    hipLaunchKernelGGL((gemm_fast_kernel<false, false, false, false, true>), grid, block, 0, st, p, 0);
    return LAUNCH_CHECK();
}
// im2col fast path of tatt_conv2d_fwd: NHWC input with unit channel stride, Cin % 16 == 0, Cout % 64 == 0
static int try_conv_fast(const GemmP& p, hipStream_t st) {
    if ((p.N & 63) || (p.Cin & 15) || p.xsc != 1 || p.sbn != 1 || (p.sbk & 3)) return -1;
    if (!al16(p.A) || !al16(p.B) || (p.xsn & 3) || (p.xsh & 3) || (p.xsw & 3)) return -1;
    if (p.splitk > 1 && !al16(p.partial)) return -1;
    const int cvec = (p.scn == 1 && !(p.scm & 3) && al16(p.C) && (!p.bias || al16(p.bias))) ? 1 : 0;
    dim3 grid(cdiv(p.M, 64) * (p.N >> 6), p.splitk, 1), block(256);
    hipLaunchKernelGGL((gemm_fast_kernel<true, false, false, true>), grid, block, 0, st, p, cvec);
    return LAUNCH_CHECK();
}

// Deterministic split-K reduction + epilogue.  remap_cin > 0: the (i,j) result of a conv
// weight-gradient GEMM (i = (tap,ci), j = co) is scattered to the reference's OIHW layout
// dW[co][ci][tap]  (reference nn.Conv2d weight layout, model/tsrn.py:597).
struct ReduceArgs {
    const float* partial; float* C; const float* bias; float* rowsum;
    long scm, scn, bsC, bsBias;
    int M, N, S, Z;
    float alpha, beta;
    int act, remap_cin, remap_taps, block0;
    int RM;     // length of the row-sum vector (0: M)
};
__device__ __forceinline__ void splitk_reduce_body(const ReduceArgs& a, long block) {
    __shared__ float sh[4][64];
    const float* __restrict__ partial = a.partial;
    float* __restrict__ C = a.C;
    const int M = a.M, N = a.N, S = a.S, Z = a.Z;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // 64 outputs x 4 slab lanes per block
    const long idx = block * 64 + tx;
    const long total = (long)Z * M * N;
    const long total64 = ((total + 63) / 64) * 64;
    if (a.rowsum && idx >= total64) {
        // trailing blocks: row sums of A, partial slabs [S][M] stored behind the S (M x N) slabs  (Z == 1)
        const long i2 = idx - total64;
        const int rm = a.RM > 0 ? a.RM : M;
        float r = 0.f;
        if (i2 < rm) for (int k = ty; k < S; k += 4) r += partial[(long)S * M * N + (long)k * rm + i2];
        sh[ty][tx] = r;
        __syncthreads();
        if (ty == 0 && i2 < rm) a.rowsum[i2] = a.alpha * ((sh[0][tx] + sh[1][tx]) + (sh[2][tx] + sh[3][tx]));
        return;
    }
    const bool ok = idx < total;
    int j = 0, i = 0, z = 0;
    float s = 0.f;
    if (ok) {
        j = idx % N;
        long rest = idx / N;
        i = rest % M;
        z = rest / M;
        const long step = (long)M * N;
        const float* pp = partial + ((long)z * S * M + i) * N + j;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int k = ty;
        for (; k + 12 < S; k += 16) {
            float v0 = pp[(long)k * step], v1 = pp[(long)(k + 4) * step], v2 = pp[(long)(k + 8) * step], v3 = pp[(long)(k + 12) * step];
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; k < S; k += 4) a0 += pp[(long)k * step];
        s = (a0 + a1) + (a2 + a3);
    }
    sh[ty][tx] = s;
    __syncthreads();
    if (ty != 0 || !ok) return;
    s = (sh[0][tx] + sh[1][tx]) + (sh[2][tx] + sh[3][tx]);
    float bj = a.bias ? a.bias[(long)z * a.bsBias + j] : 0.f;
    float v = apply_act(a.alpha * (s + bj), a.act);
    long off;
    if (a.remap_cin > 0) {
        int tap = i / a.remap_cin, ci = i - tap * a.remap_cin;
        off = ((long)j * a.remap_cin + ci) * a.remap_taps + tap;
    } else {
        off = (long)z * a.bsC + i * a.scm + j * a.scn;
    }
    if (a.beta != 0.f) v += a.beta * C[off];
    C[off] = v;
}
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ReduceArgs a) { splitk_reduce_body(a, blockIdx.x); }

// Many reductions in ONE launch: weight-gradient GEMMs of a whole backward stage leave their partial slabs behind and register
// here (tatt_reduce_defer); tatt_reduce_flush sums them all.  The table travels in the kernel arguments (pointers fixed at
// hipGraph capture; the slabs stay allocated until the flush).
#define REDUCE_MAX 36
struct ReduceTable { ReduceArgs e[REDUCE_MAX]; int n; };
__global__ __launch_bounds__(256) void splitk_reduce_batch_kernel(ReduceTable t) {
    int k = 0;
    while (k + 1 < t.n && (int)blockIdx.x >= t.e[k + 1].block0) ++k;
    splitk_reduce_body(t.e[k], (long)blockIdx.x - t.e[k].block0);
}
// Host-side state of the deferred reductions, ONE RECORD PER STREAM (guarded by a mutex): hosts driving different streams -- two
// trainers, two threads, the two lanes of one training step -- never see each other's registrations.
struct ReduceState { bool defer = false; int n = 0; ReduceArgs pending[REDUCE_MAX]; };
static std::mutex g_reduce_mu;
static std::unordered_map<hipStream_t, ReduceState> g_reduce_states;
static long reduce_blocks(const ReduceArgs& a) {
    long total = (long)a.Z * a.M * a.N;
    if (a.rowsum) total = (long)cdiv(total, 64) * 64 + (a.RM > 0 ? a.RM : a.M);   // extra thread range (64-aligned start) for the row sums
    return cdiv(total, 64);
}
static int reduce_flush_locked(ReduceState& rs, hipStream_t st) {
    if (rs.n == 0) return 0;
    ReduceTable t;
    t.n = rs.n;
    int blocks = 0;
    for (int k = 0; k < t.n; ++k) {
        t.e[k] = rs.pending[k];
        t.e[k].block0 = blocks;
        blocks += (int)reduce_blocks(t.e[k]);
    }
    rs.n = 0;
    hipLaunchKernelGGL(splitk_reduce_batch_kernel, dim3(blocks), dim3(256), 0, st, t);
    return LAUNCH_CHECK();
}
static int reduce_flush(hipStream_t st) {
    std::lock_guard<std::mutex> lock(g_reduce_mu);
    auto it = g_reduce_states.find(st);
    return it == g_reduce_states.end() ? 0 : reduce_flush_locked(it->second, st);
}
static int reduce_submit(const ReduceArgs& a, hipStream_t st) {
    std::lock_guard<std::mutex> lock(g_reduce_mu);
    auto it = g_reduce_states.find(st);
    if (it == g_reduce_states.end() || !it->second.defer || a.beta != 0.f) {   // an accumulating reduce must see earlier ones: flush, then run it
        if (it != g_reduce_states.end()) {
            int rc = reduce_flush_locked(it->second, st);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)reduce_blocks(a)), dim3(256), 0, st, a);
        return LAUNCH_CHECK();
    }
    ReduceState& rs = it->second;
    rs.pending[rs.n++] = a;
    return rs.n == REDUCE_MAX ? reduce_flush_locked(rs, st) : 0;
}
// on != 0: split-K reductions issued from now on are only registered (their workspaces must stay allocated);
// on == 0 (or tatt_reduce_flush): everything registered ON THAT STREAM is reduced by one launch per 36 entries.
TATT_API int tatt_reduce_defer(int on, hipStream_t st) {
    std::lock_guard<std::mutex> lock(g_reduce_mu);
    if (on) { g_reduce_states[st].defer = true; return 0; }
    auto it = g_reduce_states.find(st);
    if (it == g_reduce_states.end()) return 0;
    const int rc = reduce_flush_locked(it->second, st);
    g_reduce_states.erase(it);
    return rc;
}
TATT_API int tatt_reduce_flush(hipStream_t st) { return reduce_flush(st); }

static int ilog2_or_neg(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

template <int AMODE, int KCT>
static int launch_gemm(const GemmP& p, int Z, bool ak, bool bk, hipStream_t st) {
    dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), p.splitk, Z), block(256);
    if (AMODE == 3) {
        if (bk) hipLaunchKernelGGL((gemm_mfma_kernel<3, 1, 1, 16>), grid, block, 0, st, p);
        else    hipLaunchKernelGGL((gemm_mfma_kernel<3, 1, 0, 16>), grid, block, 0, st, p);
    } else if (AMODE == 4) {
        if (bk) hipLaunchKernelGGL((gemm_mfma_kernel<4, 0, 1, 16>), grid, block, 0, st, p);
        else    hipLaunchKernelGGL((gemm_mfma_kernel<4, 0, 0, 16>), grid, block, 0, st, p);
    } else {
        if (ak && bk)       hipLaunchKernelGGL((gemm_mfma_kernel<AMODE, 1, 1, KCT>), grid, block, 0, st, p);
        else if (ak && !bk) hipLaunchKernelGGL((gemm_mfma_kernel<AMODE, 1, 0, KCT>), grid, block, 0, st, p);
        else if (!ak && bk) hipLaunchKernelGGL((gemm_mfma_kernel<AMODE, 0, 1, KCT>), grid, block, 0, st, p);
        else                hipLaunchKernelGGL((gemm_mfma_kernel<AMODE, 0, 0, KCT>), grid, block, 0, st, p);
    }
    return LAUNCH_CHECK();
}

static int finish_splitk(const GemmP& p, int Z, int remap_cin, int remap_taps, hipStream_t st) {
    ReduceArgs a = {p.partial, p.C, p.bias, p.rowsum, p.scm, p.scn, p.bsC, p.bsBias, p.M, p.N, p.splitk, Z, p.alpha, p.beta,
                    p.act, remap_cin, remap_taps, 0, 0};
    return reduce_submit(a, st);
}

static void set_split(GemmP& p, int splitk, float* ws, int kc = KC) {
    int nchunks = cdiv(p.K, kc);
    if (splitk < 1) splitk = 1;
    if (splitk > nchunks) splitk = nchunks;
    p.chunks_per_split = cdiv(nchunks, splitk);
    p.splitk = cdiv(nchunks, p.chunks_per_split);
    p.partial = ws;
    if (p.splitk > 1 && ws == nullptr) p.splitk = 1, p.chunks_per_split = nchunks;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------

// General strided (batched) GEMM:  C = act(alpha * (A @ B + bias)) + beta * C.
// A2/K1: optional second A source for r >= K1 (K-concatenation).  splitk > 1 needs `ws` of
// Z*splitk*M*N floats.  Replaces nn.Linear / 1x1 nn.Conv2d / nn.GRU input projections
// (reference model/tsrn.py:170,1071-1072; model/transformer_v2.py:455-457,788-790).
TATT_API int tatt_gemm(const float* A, long sam, long sak, const float* A2, long sa2m, long sa2k, int K1,
                       const float* B, long sbk, long sbn, const float* bias, float* C, long scm, long scn,
                       int M, int N, int K, int Z, long bsA, long bsA2, long bsB, long bsC, long bsBias,
                       float alpha, float beta, int act, int splitk, float* ws, float* rowsum, hipStream_t st) {
    if (M <= 0 || N <= 0 || Z <= 0) return 0;
    GemmP p = {};
    p.A = A; p.A2 = A2; p.B = B; p.bias = bias; p.C = C;
    p.M = M; p.N = N; p.K = K; p.K1 = K1;
    p.sam = sam; p.sak = sak; p.sa2m = sa2m; p.sa2k = sa2k; p.sbk = sbk; p.sbn = sbn; p.scm = scm; p.scn = scn;
    p.bsA = bsA; p.bsA2 = bsA2; p.bsB = bsB; p.bsC = bsC; p.bsBias = bsBias;
    p.alpha = alpha; p.beta = beta; p.act = act;
    p.rowsum = rowsum;                    // row sums of A (bias gradient); with split-K, ws needs splitk*M extra floats
    // (64-deep K chunks were measured slower than 16-deep on the token GEMMs: 2 work-groups/CU instead of 8.)
    set_split(p, splitk, ws, KC);
    bool ak = (sak == 1), bk = (sbk == 1 && sbn != 1);
    int rc = try_gemm_fast(p, Z, st);
    if (rc == 0) return p.splitk > 1 ? finish_splitk(p, Z, 0, 0, st) : 0;
    if (rc > 0) return rc;
    rc = A2 ? launch_gemm<2, 16>(p, Z, ak, bk, st) : launch_gemm<0, 16>(p, Z, ak, bk, st);
    if (rc) return rc;
    if (p.splitk > 1) return finish_splitk(p, Z, 0, 0, st);
    return 0;
}

static void fill_conv(GemmP& p, int H, int W, int Cin, int KH, int KW, long xsn, long xsh, long xsw, long xsc) {
    p.H = H; p.W = W; p.Cin = Cin; p.KH = KH; p.KW = KW; p.pad = (KH - 1) / 2; p.HW = H * W;
    p.Wshift = ilog2_or_neg(W); p.HWshift = ilog2_or_neg(H * W); p.Cshift = ilog2_or_neg(Cin);
    p.xsn = xsn; p.xsh = xsh; p.xsw = xsw; p.xsc = xsc;
}

// Stride-1 'same' convolution as implicit GEMM (forward, and data-gradient when `wpacked` holds the
// flipped/transposed filter).  x addressed with explicit strides (NCHW or NHWC), wpacked is
// [KH][KW][Cin][Cout], output y[pixel*ldy + co] (NHWC rows).  y = act(conv + bias) + beta*y.
// Replaces nn.Conv2d (reference model/tsrn.py:597,612,877,885,1043; model/stn_head.py:15).
TATT_API int tatt_conv2d_fwd(const float* x, long xsn, long xsh, long xsw, long xsc, const float* wpacked,
                             const float* bias, float* y, long ldy, int Bn, int H, int W, int Cin, int Cout,
                             int KH, int KW, int act, float beta, int splitk, float* ws, hipStream_t st) {
    GemmP p = {};
    p.A = x; p.B = wpacked; p.bias = bias; p.C = y;
    p.M = Bn * H * W; p.N = Cout; p.K = KH * KW * Cin;
    p.sbk = Cout; p.sbn = 1; p.scm = ldy; p.scn = 1;
    p.alpha = 1.f; p.beta = beta; p.act = act;
    fill_conv(p, H, W, Cin, KH, KW, xsn, xsh, xsw, xsc);
    set_split(p, splitk, ws);        // split-K (ws >= splitk*Bn*H*W*Cout floats) spreads small-M / deep-K convs (STN tail) over the CUs
    int rc = try_conv_fast(p, st);
    if (rc == 0) return p.splitk > 1 ? finish_splitk(p, 1, 0, 0, st) : 0;
    if (rc > 0) return rc;
    rc = launch_gemm<3, 16>(p, 1, true, false, st);
    if (rc) return rc;
    if (p.splitk > 1) return finish_splitk(p, 1, 0, 0, st);
    return 0;
}

// tatt_conv2d_fwd without bias / activation / beta that leaves the split contraction UNSUMMED: ws receives *splits partial maps
// ([s][pixel][Cout], *splits <= splitk; 1: the finished map) for a consumer that adds them as it loads them
// (tatt_stn_bn_pool_fwd_parts / _bwd_parts) -- one launch less between two links of the STN head's dependent chain.
// ws >= max(splitk, 1) * Bn*H*W*Cout floats; `splits` is a HOST pointer, written before the function returns.
TATT_API int tatt_conv2d_fwd_partials(const float* x, long xsn, long xsh, long xsw, long xsc, const float* wpacked, int Bn, int H,
                                      int W, int Cin, int Cout, int KH, int KW, int splitk, float* ws, int* splits, hipStream_t st) {
    if (!ws || !splits) return 1;
    GemmP p = {};
    p.A = x; p.B = wpacked; p.bias = nullptr; p.C = ws;
    p.M = Bn * H * W; p.N = Cout; p.K = KH * KW * Cin;
    p.sbk = Cout; p.sbn = 1; p.scm = Cout; p.scn = 1;
    p.alpha = 1.f; p.beta = 0.f; p.act = ACT_NONE;
    fill_conv(p, H, W, Cin, KH, KW, xsn, xsh, xsw, xsc);
    set_split(p, splitk, ws);
    *splits = p.splitk;
    int rc = try_conv_fast(p, st);
    if (rc >= 0) return rc;
    return launch_gemm<3, 16>(p, 1, true, false, st);
}

// Convolution weight gradient: dW[co][ci][kh][kw] (OIHW, the reference parameter layout)
//   = sum_pixels x[pixel + (kh,kw), ci] * dy[pixel, co];  dW = result + beta*dW.
// Split over pixels (splitk) with a deterministic second-stage reduction; ws >= splitk*KH*KW*Cin*Cout floats.
TATT_API int tatt_conv2d_wgrad(const float* x, long xsn, long xsh, long xsw, long xsc, const float* dy,
                               long lddy, float* dw_oihw, int Bn, int H, int W, int Cin, int Cout, int KH,
                               int KW, float beta, int splitk, float* ws, hipStream_t st) {
    GemmP p = {};
    p.A = x; p.B = dy; p.bias = nullptr; p.C = dw_oihw;
    p.M = KH * KW * Cin; p.N = Cout; p.K = Bn * H * W;
    p.sbk = lddy; p.sbn = 1; p.scm = Cout; p.scn = 1;
    p.alpha = 1.f; p.beta = beta; p.act = ACT_NONE;
    fill_conv(p, H, W, Cin, KH, KW, xsn, xsh, xsw, xsc);
    set_split(p, splitk < 2 ? 2 : splitk, ws);
    if (p.splitk < 2) { p.splitk = 2; p.chunks_per_split = cdiv(cdiv(p.K, KC), 2); }
    int rc = try_wgrad_fast(p, st);
    if (rc < 0) rc = launch_gemm<4, 16>(p, 1, false, false, st);
    if (rc) return rc;
    return finish_splitk(p, 1, Cin, KH * KW, st);
}

// OIHW filter -> implicit-GEMM operand.  mode 0: [KH][KW][Cin][Cout] (forward);
// mode 1: [KH][KW][Cout][Cin] spatially flipped (data-gradient: dX = conv(dY, flip(W)^T)).
// modes 2 / 3: the same two filters with the contraction axis contiguous ([tap][out][in]) for tatt_conv3_c64_fwd_t.
// (modes 4 / 5 belonged to the retired 32x32 weight-stationary kernel;)
// modes 6 / 7: in the register order of tatt_conv3_c64_fwd_ws16.
// modes 8 / 9: Toeplitz-expanded 9x9 filter of tatt_conv9_c64_to_c4_mfma in the per-lane MFMA B-fragment order of that kernel
// (a wave's 16-byte fragment load reads 1 KB of consecutive memory):
//   out[((((ky * 4 + c) * 2 + kh) * 6 + cqi * 3 + d) * 64 + lane) * 4 + u] = Wt[ky][ci][n][dx],
//   ci = 16 c + 4 (kh + 2 cqi) + (lane >> 4), n = lane & 15 = 4 j + o, dx = 4 d + u (< 12),
//   Wt = f[o][ci][ky][dx - j] if 0 <= dx - j < 9 else 0;  mode 8: f = w (OIHW, Cout = 4, Cin = 64);
//   mode 9 (data gradient of a 4 -> 64 convolution, w OIHW with Cout = 64, Cin = 4): f[o][ci][ky][kx] = w[ci][o][8 - ky][8 - kx].
#define TOEPLITZ9_WORDS (9 * 4 * 2 * 6 * 64 * 4)
__device__ __forceinline__ float toeplitz9(const float* __restrict__ w, int idx, int mode) {
    const int u = idx & 3, lane = (idx >> 2) & 63;
    int r = idx >> 8;
    const int f = r % 6; r /= 6;
    const int kh = r & 1; r >>= 1;
    const int c = r & 3, ky = r >> 2;
    const int cqi = f / 3, d = f - 3 * cqi;
    const int ci = 16 * c + 4 * (kh + 2 * cqi) + (lane >> 4), nn = lane & 15;
    const int j = nn >> 2, o = nn & 3, kx = 4 * d + u - j;
    if (kx < 0 || kx >= 9) return 0.f;
    return mode == 8 ? w[(((long)o * 64 + ci) * 9 + ky) * 9 + kx] : w[(((long)ci * 4 + o) * 9 + (8 - ky)) * 9 + (8 - kx)];
}
static inline long repack_total(int Cout, int Cin, int KH, int KW, int mode) {
    return (mode == 8 || mode == 9 || mode == 12 || mode == 13) ? (long)TOEPLITZ9_WORDS : (long)Cout * Cin * KH * KW;
}
// modes 10 / 11: the split-bf16 B operand of tatt_conv3_c64_fwd_sb (3x3; conv input channels a multiple of 64, output channels of 16):
// 32-bit words of two bf16 with consecutive input channels,
//   word[((((chunk * nblk + blk) * 18 + ks) * 2 + hl) * 64 + lane) * 4 + e2],  ks = tap * 2 + half, hl = 0: hi = bf16(v), 1: lo = bf16(v - hi),
//   conv output channel o = 16 blk + (lane & 15), conv input channel i = 64 chunk + 32 half + 8 (lane >> 4) + 2 e2 + {0, 1};
//   mode 10: v = filter[o][i][tap] (forward);  mode 11: v = filter[i][o][8 - tap] (data gradient: o = ci, i = co of the OIHW filter)
typedef __bf16 rp_bf16x2 __attribute__((ext_vector_type(2)));
typedef float rp_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float repack_sb(const float* __restrict__ w, long idx, int Cout, int Cin, int mode) {
    const int e2 = (int)(idx & 3), lane = (int)((idx >> 2) & 63), hl = (int)((idx >> 8) & 1);
    long r = idx >> 9;
    const int ks = (int)(r % 18); r /= 18;
    const int nblk = (mode == 10 ? Cout : Cin) / 16;
    const int blk = (int)(r % nblk), chunk = (int)(r / nblk);
    const int tap = ks >> 1, half = ks & 1;
    const int o = 16 * blk + (lane & 15), i = 64 * chunk + 32 * half + 8 * (lane >> 4) + 2 * e2;
    rp_f32x2 v;
#pragma unroll
    for (int u = 0; u < 2; ++u) v[u] = mode == 10 ? w[((long)o * Cin + (i + u)) * 9 + tap] : w[((long)(i + u) * Cin + o) * 9 + (8 - tap)];
    const rp_bf16x2 hi = __builtin_convertvector(v, rp_bf16x2);
    const rp_bf16x2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, rp_f32x2), rp_bf16x2);
    return __builtin_bit_cast(float, hl ? lo : hi);
}
// modes 14 / 15: the same values for tatt_conv3_c64_fwd_sb's 32x32x16 kernel (round 6: 32 output channels per wave, A operand of
// v_mfma_f32_32x32x16_bf16: lane (i = lane & 31, kb = lane >> 5) supplies 8 consecutive k of row i):
//   word[((((chunk * nblk + blk) * 9 + tap) * 2 + kh) * 4 + s * 2 + hl) * 64 + lane) * 4 + e2],  nblk = conv output channels / 32,
//   conv output channel o = 32 blk + (lane & 31), conv input channel i = 64 chunk + 32 kh + 16 s + 8 (lane >> 5) + 2 e2 + {0, 1};
//   mode 14: v = filter[o][i][tap] (forward);  mode 15: v = filter[i][o][8 - tap] (data gradient)
__device__ __forceinline__ float repack_sb32(const float* __restrict__ w, long idx, int Cout, int Cin, int mode) {
    const int e2 = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    long r = idx >> 8;
    const int F = (int)(r % 72); r /= 72;
    const int nblk = (mode == 14 ? Cout : Cin) / 32;
    const int blk = (int)(r % nblk), chunk = (int)(r / nblk);
    const int hl = F & 1, sstep = (F >> 1) & 1, kh = (F >> 2) & 1, tap = F >> 3;
    const int o = 32 * blk + (lane & 31), i = 64 * chunk + 32 * kh + 16 * sstep + 8 * (lane >> 5) + 2 * e2;
    rp_f32x2 v;
#pragma unroll
    for (int u = 0; u < 2; ++u) v[u] = mode == 14 ? w[((long)o * Cin + (i + u)) * 9 + tap] : w[((long)(i + u) * Cin + o) * 9 + (8 - tap)];
    const rp_bf16x2 hi = __builtin_convertvector(v, rp_bf16x2);
    const rp_bf16x2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, rp_f32x2), rp_bf16x2);
    return __builtin_bit_cast(float, hl ? lo : hi);
}
// modes 12 / 13: the Toeplitz-expanded 9x9 filter of modes 8 / 9 as the split-bf16 B operand of tatt_conv9_c64_to_c4_sb
// (v_mfma_f32_16x16x32_bf16: a lane supplies 8 consecutive k = two pixel offsets x 16 channels spread over the four k-quarters):
//   word[((((ky * 4 + c) * 6 + pair) * 2 + hl) * 64 + lane) * 4 + e2]  = two bf16 of input channels ci, ci + 1,
//   n = lane & 15 = 4 j + o, kq = lane >> 4, dx = 2 pair + (kq >> 1), ci = 16 c + 8 (kq & 1) + 2 e2,
//   value = f[o][ci][ky][dx - j] if 0 <= dx - j < 9 else 0 (f as for modes 8 / 9), hl = 0: hi = bf16(v), 1: lo = bf16(v - hi).
__device__ __forceinline__ float toeplitz9_sb(const float* __restrict__ w, int idx, int mode) {
    const int e2 = idx & 3, lane = (idx >> 2) & 63, hl = (idx >> 8) & 1;
    int r = idx >> 9;
    const int pair = r % 6; r /= 6;
    const int c = r & 3, ky = r >> 2;
    const int nn = lane & 15, kq = lane >> 4;
    const int j = nn >> 2, o = nn & 3, kx = 2 * pair + (kq >> 1) - j, ci = 16 * c + 8 * (kq & 1) + 2 * e2;
    rp_f32x2 v = (rp_f32x2){0.f, 0.f};
    if (kx >= 0 && kx < 9) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
            v[u] = mode == 12 ? w[(((long)o * 64 + ci + u) * 9 + ky) * 9 + kx] : w[(((long)(ci + u) * 4 + o) * 9 + (8 - ky)) * 9 + (8 - kx)];
    }
    const rp_bf16x2 hi = __builtin_convertvector(v, rp_bf16x2);
    const rp_bf16x2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, rp_f32x2), rp_bf16x2);
    return __builtin_bit_cast(float, hl ? lo : hi);
}
__global__ void repack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin,
                                     int KH, int KW, int mode) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (mode >= 14) {
        if (idx < Cout * Cin * 9) out[idx] = repack_sb32(w, idx, Cout, Cin, mode);
        return;
    }
    if (mode >= 12) {
        if (idx < TOEPLITZ9_WORDS) out[idx] = toeplitz9_sb(w, idx, mode);
        return;
    }
    if (mode >= 10) {
        if (idx < Cout * Cin * 9) out[idx] = repack_sb(w, idx, Cout, Cin, mode);
        return;
    }
    if (mode >= 8) {
        if (idx < TOEPLITZ9_WORDS) out[idx] = toeplitz9(w, idx, mode);
        return;
    }
    int total = Cout * Cin * KH * KW;
    if (idx >= total) return;
    // idx enumerates the OUTPUT
    int T = KH * KW;
    if (mode == 0) {
        int co = idx % Cout; int r = idx / Cout; int ci = r % Cin; int tap = r / Cin;
        out[idx] = w[((long)co * Cin + ci) * T + tap];
    } else if (mode == 1) {
        int ci = idx % Cin; int r = idx / Cin; int co = r % Cout; int tap = r / Cout;
        out[idx] = w[((long)co * Cin + ci) * T + (T - 1 - tap)];
    } else if (mode == 2) {      // [tap][Cout][Cin]: forward filter with the contraction (input-channel) axis contiguous
        int ci = idx % Cin; int r = idx / Cin; int co = r % Cout; int tap = r / Cout;
        out[idx] = w[((long)co * Cin + ci) * T + tap];
    } else if (mode == 3) {      // [tap][Cin][Cout] flipped: data-gradient filter, contraction (Cout) axis contiguous
        int co = idx % Cout; int r = idx / Cout; int ci = r % Cin; int tap = r / Cin;
        out[idx] = w[((long)co * Cin + ci) * T + (T - 1 - tap)];
    } else {
        // modes 6 / 7 (3x3, 64 contraction channels): the per-lane MFMA B-operand register order of tatt_conv3_c64_fwd_ws16
        // (16 output channels per wave, v_mfma_f32_16x16x4_f32):
        // out[((blk * 36 + tap * 4 + g) * 64 + lane) * 4 + u] = filter[out ch 16 blk + (lane & 15)][in ch 16 g + 4 (lane >> 4) + u][tap]
        const int u = idx & 3, lane = (idx >> 2) & 63;
        const int q6 = (idx >> 8) % 36, blk = idx / (36 * 256);
        const int tap6 = q6 >> 2, g6 = q6 & 3;
        const int o6 = blk * 16 + (lane & 15), i6 = 16 * g6 + 4 * (lane >> 4) + u;
        if (mode == 6) out[idx] = w[((long)o6 * Cin + i6) * T + tap6];
        else out[idx] = w[((long)i6 * Cin + o6) * T + (T - 1 - tap6)];
    }
}
// 32-bit words a packed layout occupies (what `out` of tatt_repack_conv_weight must hold); -1 for an unknown mode.  Host-only.
TATT_API int tatt_repack_words(int Cout, int Cin, int KH, int KW, int mode) {
    if (mode == 4 || mode == 5 || mode < 0 || mode > 15) return -1;
    return (int)repack_total(Cout, Cin, KH, KW, mode);
}
TATT_API int tatt_repack_conv_weight(const float* w_oihw, float* out, int Cout, int Cin, int KH, int KW,
                                     int mode, hipStream_t st) {
    const bool fwd9 = mode == 8 || mode == 12, dgrad9 = mode == 9 || mode == 13;
    if ((fwd9 || dgrad9) && !(KH == 9 && KW == 9 && ((fwd9 && Cout == 4 && Cin == 64) || (dgrad9 && Cout == 64 && Cin == 4)))) return 1;
    if ((mode == 10 || mode == 11 || mode == 14 || mode == 15) && !(KH == 3 && KW == 3 && Cout % 64 == 0 && Cin % 64 == 0)) return 1;
    if (mode == 4 || mode == 5 || mode < 0 || mode > 15) return 1;     // (4 / 5: the retired 32x32 weight-stationary kernel)
    long total = repack_total(Cout, Cin, KH, KW, mode);
    hipLaunchKernelGGL(repack_weight_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w_oihw, out, Cout, Cin,
                       KH, KW, mode);
    return LAUNCH_CHECK();
}

// All packed filter layouts of a model in ONE launch (after the optimiser has moved the weights): up to REPACK_MAX entries, the
// descriptor table travels in the kernel arguments (fixed at hipGraph capture: the buffers are persistent).
#define REPACK_MAX 96
struct RepackEntry { const float* w; float* out; int Cout, Cin, KH, KW, mode, block0; };
struct RepackTable { RepackEntry e[REPACK_MAX]; int n; };
__global__ void repack_batch_kernel(RepackTable t) {
    int k = 0;
    while (k + 1 < t.n && (int)blockIdx.x >= t.e[k + 1].block0) ++k;          // wave-uniform walk over <= 96 entries
    const RepackEntry& e = t.e[k];
    const int idx = ((int)blockIdx.x - e.block0) * blockDim.x + threadIdx.x;
    if (e.mode >= 14) {
        if (idx < e.Cout * e.Cin * 9) e.out[idx] = repack_sb32(e.w, idx, e.Cout, e.Cin, e.mode);
        return;
    }
    if (e.mode >= 12) {
        if (idx < TOEPLITZ9_WORDS) e.out[idx] = toeplitz9_sb(e.w, idx, e.mode);
        return;
    }
    if (e.mode >= 10) {
        if (idx < e.Cout * e.Cin * 9) e.out[idx] = repack_sb(e.w, idx, e.Cout, e.Cin, e.mode);
        return;
    }
    if (e.mode >= 8) {
        if (idx < TOEPLITZ9_WORDS) e.out[idx] = toeplitz9(e.w, idx, e.mode);
        return;
    }
    const int total = e.Cout * e.Cin * e.KH * e.KW;
    if (idx >= total) return;
    const int T = e.KH * e.KW, Cout = e.Cout, Cin = e.Cin;
    const float* w = e.w;
    float v;
    if (e.mode == 0) { int co = idx % Cout; int r = idx / Cout; int ci = r % Cin; int tap = r / Cin; v = w[((long)co * Cin + ci) * T + tap]; }
    else if (e.mode == 1) { int ci = idx % Cin; int r = idx / Cin; int co = r % Cout; int tap = r / Cout; v = w[((long)co * Cin + ci) * T + (T - 1 - tap)]; }
    else if (e.mode == 2) { int ci = idx % Cin; int r = idx / Cin; int co = r % Cout; int tap = r / Cout; v = w[((long)co * Cin + ci) * T + tap]; }
    else if (e.mode == 3) { int co = idx % Cout; int r = idx / Cout; int ci = r % Cin; int tap = r / Cin; v = w[((long)co * Cin + ci) * T + (T - 1 - tap)]; }
    else {
        const int u = idx & 3, lane = (idx >> 2) & 63;
        const int q6 = (idx >> 8) % 36, blk = idx / (36 * 256);
        const int tap6 = q6 >> 2, g6 = q6 & 3;
        const int o6 = blk * 16 + (lane & 15), i6 = 16 * g6 + 4 * (lane >> 4) + u;
        v = e.mode == 6 ? w[((long)o6 * Cin + i6) * T + tap6] : w[((long)i6 * Cin + o6) * T + (T - 1 - tap6)];
    }
    e.out[idx] = v;
}
// ws/outs: n source / destination pointers; dims: n x 5 ints (Cout, Cin, KH, KW, mode).  Host arrays, read during the call.
TATT_API int tatt_repack_conv_weight_batch(const float* const* ws, float* const* outs, const int* dims, int n, hipStream_t st) {
    if (n <= 0) return 0;
    for (int base = 0; base < n; base += REPACK_MAX) {
        RepackTable t;
        t.n = n - base < REPACK_MAX ? n - base : REPACK_MAX;
        int blocks = 0;
        for (int k = 0; k < t.n; ++k) {
            const int* d = dims + (long)(base + k) * 5;
            t.e[k] = {ws[base + k], outs[base + k], d[0], d[1], d[2], d[3], d[4], blocks};
            blocks += cdiv(repack_total(d[0], d[1], d[2], d[3], d[4]), 256);
        }
        hipLaunchKernelGGL(repack_batch_kernel, dim3(blocks), dim3(256), 0, st, t);
    }
    return LAUNCH_CHECK();
}

// Deterministic reduction of S partial (M x N) slabs: C = sum_s partial[s] (+ beta*C).  remap_cin > 0 scatters row
// i = tap*remap_cin + ci, column j = co to the OIHW filter layout dW[co][ci][tap] (used by the specialised conv
// weight-gradient kernels of conv3.hip).
TATT_API int tatt_splitk_reduce(const float* partial, float* C, int M, int N, int S, int remap_cin, int remap_taps,
                                float beta, float* vec, int vec_len, hipStream_t st) {
    ReduceArgs a = {partial, C, nullptr, vec, (long)N, 1L, 0L, 0L, M, N, S, 1, 1.f, beta, (int)ACT_NONE, remap_cin, remap_taps, 0,
                    vec ? vec_len : 0};
    return reduce_submit(a, st);
}
