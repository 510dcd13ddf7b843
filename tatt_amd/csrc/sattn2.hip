// Score-free multi-head self-attention of the TBSRN FeatureEnhancer (reference model/tbsrn.py:96-151) on the bf16 matrix cores with split
// operands -- round 6.  Same entry points, arguments, dropout masks and saved log-sum-exp as sattn.hip (exact fp32 MFMA, kept for
// tatt_amd.set_arithmetic("fp32")); here every product is three v_mfma_f32_32x32x16_bf16 of hi / lo halves (a = hi + lo, a b ~ hi hi +
// hi lo + lo hi, fp32 accumulation: 2^-16 relative per product, the arithmetic of the convolution / token-GEMM families).
//
// The fp32 kernels are matrix-core-bound on the slow pipe (0.57 of the fp32 MFMA peak, 1.41 ms per FeatureEnhancer at B = 48, 52 % of the
// TBSRN step).  On the bf16 pipe the same FLOPs cost a fifth; what is left is the softmax arithmetic (exp + the dropout hash: ~25 VALU
// per score), so the organisation aims at keeping the VALU busy: 256-thread work-groups, two per CU.
//
// Layouts.  A (sample, head) tensor is (P rows, 32 channels).  Tiles of 64 rows are staged once per work-group as bf16 hi / lo IMAGES in
// LDS in one or both of two pitches: "row" images (80 B per row: conflict-free 16-byte reads of 8 consecutive channels of a row = the
// operand whose contraction runs over CHANNELS) and "tr" images (64 B per row: ds_read_b64_tr_b16 hands a lane 4 consecutive ROWS of one
// channel = the operand whose contraction runs over rows; the four rows of a read fall into the four bank quarters).
// Every score block is computed with its 32 x 32 accumulator laid out so that a lane owns ONE query (forward, dQ) or ONE key (dK / dV)
// and 16 of the 32 keys / queries of the block: the softmax statistics are per-lane scalars + one exchange with lane ^ 32, and the block
// becomes the B operand of the next product after a v_permlane32_swap of register quads (the lane pair trades the halves it is missing).
#include "common.h"

typedef __bf16 s2_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 s2_bf16x2 __attribute__((ext_vector_type(2)));
typedef float s2_f32x2 __attribute__((ext_vector_type(2)));
typedef float s2_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned s2_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned s2_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 s2_bf16x4_t;
typedef __attribute__((address_space(3))) s2_bf16x4_t* s2_lds_b64;

#define S2_D 32
#define S2_T 64                           // rows per staged tile
#define S2_PR 80                          // bytes per row of a row image
#define S2_PT 64                          // bytes per row of a tr image
#define S2_RIMG (S2_T * S2_PR)            // 5120 B
#define S2_TIMG (S2_T * S2_PT)            // 4096 B

struct SAttn2P {                           // (SAttnP of sattn.hip + bits)
    const float* Q; const float* K; const float* V;
    float* O; float* lse;
    const float* dO; const float* Dv;
    float* dQ; float* dK; float* dV;
    int B, P, h, E;
    float scale, pdrop; const unsigned long long* seed; unsigned site;
    unsigned* bits;                        // keep bits of the dropout masks (MODE 2), see below
};

__device__ __forceinline__ f32x4 s2_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// (a, b) -> packed bf16 hi pair (round to nearest even) and packed bf16 lo pair of the exact remainders.  (v_dot2c_f32_bf16 with the
// constant (-1, 0) would give a remainder in one instruction instead of two; tried: the copies it needs -- its accumulator is its
// destination -- eat the saving, 547 vs 537 us for the B = 48 forward + backward.)
__device__ __forceinline__ void s2_split2(float a, float b, unsigned& hi, unsigned& lo) {
    const s2_bf16x2 h = __builtin_convertvector((s2_f32x2){a, b}, s2_bf16x2);
    const unsigned u = __builtin_bit_cast(unsigned, h);
    const float ra = a - __builtin_bit_cast(float, u << 16), rb = b - __builtin_bit_cast(float, u & 0xffff0000u);
    hi = u;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector((s2_f32x2){ra, rb}, s2_bf16x2));
}
// v_permlane32_swap_b32 a, b: a <- [a.lo | b.lo], b <- [a.hi | b.hi] (gru.hip: the builtin of ROCm 7.2 loses one result)
__device__ __forceinline__ void s2_swap(unsigned& a, unsigned& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float s2_pair_max(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    s2_swap(a, b);
    return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
__device__ __forceinline__ float s2_pair_sum(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    s2_swap(a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ s2_f32x16 s2_mfma(s2_u32x4 a, s2_u32x4 b, s2_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s2_bf16x8, a), __builtin_bit_cast(s2_bf16x8, b), c, 0, 0, 0);
}
// 8 consecutive ROWS (k0 .. k0 + 7, k0 = 8 kb of the caller's 16-row step) of this lane's channel from a tr image: `p0` = the lane's piece
// address for row offset 0 (row a16 / 4, channel quad a16 % 4 of channels 16 g16 ..), `off` = byte offset of the step's first row
__device__ __forceinline__ s2_u32x4 s2_tr8(const char* p0, int off) {
    const s2_bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((s2_lds_b64)(p0 + off));
    const s2_bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((s2_lds_b64)(p0 + off + 4 * S2_PT));
    const s2_u32x2 ua = __builtin_bit_cast(s2_u32x2, a), ub = __builtin_bit_cast(s2_u32x2, b);
    return (s2_u32x4){ua[0], ua[1], ub[0], ub[1]};
}
#define S2_LOG2E 1.4426950408889634f
#define S2_LN2 0.6931471805599453f
__device__ __forceinline__ float s2_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// dropout_keep (common.h) for an element index below 2^32, keys hoisted
__device__ __forceinline__ bool s2_keep(uint32_t k0, uint32_t k1, uint32_t idx, uint32_t th) {
    uint32_t h = idx ^ k0;
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h += k1;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h >= th;
}
// x where the lane's bit of the 64-bit wave mask is set, else 0: the mask goes in as the SGPR pair it was loaded into
__device__ __forceinline__ float s2_sel(float x, unsigned long long mask) {
    float r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(mask));
    return r;
}
// lanes `lane`, `lane + 1` of w <- the halves of the wave-uniform 64-bit x (ROCm 7.2's clang has no __builtin_amdgcn_writelane).  The
// s_nop is needed: a v_writelane right behind the v_cmp that produced its SGPR source read the OLD register on gfx950 (measured: the one
// ballot of 16 that the scheduler had placed back to back came out wrong); the compiler's hazard recogniser does not look into the asm.
__device__ __forceinline__ unsigned s2_writelane2(unsigned w, unsigned long long x, int lane) {
    asm("s_nop 3\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4" : "+v"(w) : "s"((unsigned)x), "s"((unsigned)(x >> 32)), "n"(lane), "n"(lane + 1));
    return w;
}
// Work-group -> (row block, sample x head).  The gridDim.x row blocks of one (sample, head) read the same K / V (Q / dO) tiles, but
// consecutive work-groups of a launch go to different XCDs (8 L2 caches): in launch order every XCD would stream every tile.  When the
// number of (sample, head) pairs is a multiple of 8, XCD c instead owns the pairs c, c + 8, ..: all their row blocks run on it.
__device__ __forceinline__ void s2_block(int& xb, int& bh) {
    const unsigned nx = gridDim.x, ny = gridDim.y;
    xb = blockIdx.x; bh = blockIdx.y;
    if ((ny & 7) == 0) {
        const unsigned L = blockIdx.x + nx * blockIdx.y, c = L & 7, r = L >> 3;
        xb = (int)(r % nx);
        bh = (int)(c + 8 * (r / nx));
    }
}
// staging: thread tid of 256 moves rows tid >> 3 and (tid >> 3) + 32, channels 4 (tid & 7) .. + 3 of a 64 x 32 tile
__device__ __forceinline__ void s2_fetch(const float* __restrict__ X, long base_row, int E, int hoff, int tid, f32x4& a, f32x4& b) {
    const int r0 = tid >> 3, c4 = (tid & 7) * 4;
    a = s2_ld4(X + (base_row + r0) * E + hoff + c4);
    b = s2_ld4(X + (base_row + r0 + 32) * E + hoff + c4);
}
// row image pair (hi | lo, S2_RIMG apart) at `rimg` and / or tr image pair (S2_TIMG apart) at `timg`
template <bool ROW, bool TR>
__device__ __forceinline__ void s2_put(char* rimg, char* timg, int tid, f32x4 a, f32x4 b) {
    const int r0 = tid >> 3, c4 = (tid & 7) * 4;
    unsigned h0, l0, h1, l1;
    s2_split2(a[0], a[1], h0, l0); s2_split2(a[2], a[3], h1, l1);
    if (ROW) {
        *reinterpret_cast<uint2*>(rimg + r0 * S2_PR + c4 * 2) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(rimg + S2_RIMG + r0 * S2_PR + c4 * 2) = make_uint2(l0, l1);
    }
    if (TR) {
        *reinterpret_cast<uint2*>(timg + r0 * S2_PT + c4 * 2) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(timg + S2_TIMG + r0 * S2_PT + c4 * 2) = make_uint2(l0, l1);
    }
    s2_split2(b[0], b[1], h0, l0); s2_split2(b[2], b[3], h1, l1);
    if (ROW) {
        *reinterpret_cast<uint2*>(rimg + (r0 + 32) * S2_PR + c4 * 2) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(rimg + S2_RIMG + (r0 + 32) * S2_PR + c4 * 2) = make_uint2(l0, l1);
    }
    if (TR) {
        *reinterpret_cast<uint2*>(timg + (r0 + 32) * S2_PT + c4 * 2) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(timg + S2_TIMG + (r0 + 32) * S2_PT + c4 * 2) = make_uint2(l0, l1);
    }
}
// the six MFMAs of a 32 x 32 block whose contraction runs over the 32 CHANNELS: A = rows 32 sub .. + 31 of a row image pair, B = a row operand
__device__ __forceinline__ s2_f32x16 s2_rows_x_operand(const char* rimg, int sub, int lj, int kb, const s2_u32x4 (&bh)[2], const s2_u32x4 (&bl)[2]) {
    s2_f32x16 c;
#pragma unroll
    for (int v = 0; v < 16; ++v) c[v] = 0.f;
    const char* r = rimg + (32 * sub + lj) * S2_PR + 16 * kb;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const s2_u32x4 ah = *reinterpret_cast<const s2_u32x4*>(r + 32 * s), al = *reinterpret_cast<const s2_u32x4*>(r + 32 * s + S2_RIMG);
        c = s2_mfma(ah, bh[s], c);
        c = s2_mfma(ah, bl[s], c);
        c = s2_mfma(al, bh[s], c);
    }
    return c;
}
// the six MFMAs of acc[channel][column] += X^T[channel][row] B[row][column] over rows 32 sub .. + 31 of a tr image pair, B = a block operand
__device__ __forceinline__ s2_f32x16 s2_tr_x_block(const char* timg_lane, int sub, const s2_u32x4 (&bh)[2], const s2_u32x4 (&bl)[2], s2_f32x16 c) {
    const char* t = timg_lane + 32 * sub * S2_PT;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const s2_u32x4 ah = s2_tr8(t, 16 * s * S2_PT), al = s2_tr8(t + S2_TIMG, 16 * s * S2_PT);
        c = s2_mfma(ah, bh[s], c);
        c = s2_mfma(ah, bl[s], c);
        c = s2_mfma(al, bh[s], c);
    }
    return c;
}
// this lane's 8 channels 16 s + 8 kb .. + 7 of row `row` of a (B, P, E) tensor, times `mul`, split: the B operand of a product that
// contracts over channels with this lane's row as the column
__device__ __forceinline__ void s2_row_operand(const float* __restrict__ X, long row, int E, int hoff, int kb, float mul, s2_u32x4 (&hi)[2], s2_u32x4 (&lo)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float* q = X + row * E + hoff + 16 * s + 8 * kb;
        const f32x4 v0 = s2_ld4(q), v1 = s2_ld4(q + 4);
        unsigned h, l;
        s2_split2(v0[0] * mul, v0[1] * mul, h, l); hi[s][0] = h; lo[s][0] = l;
        s2_split2(v0[2] * mul, v0[3] * mul, h, l); hi[s][1] = h; lo[s][1] = l;
        s2_split2(v1[0] * mul, v1[1] * mul, h, l); hi[s][2] = h; lo[s][2] = l;
        s2_split2(v1[2] * mul, v1[3] * mul, h, l); hi[s][3] = h; lo[s][3] = l;
    }
}
// A 32 x 32 block held as accumulator registers (register 4 g + r of lane (lj, kb) = row 8 g + 4 kb + r, column lj) becomes the B operand
// of a product contracting over its ROWS: for the 16-row step s the lane needs rows 16 s + 8 kb .. + 7 of its column -- it owns half of
// them (register quads 2 s and 2 s + 1 are rows 16 s + 4 kb .. + 3 and 16 s + 8 + 4 kb .. + 3), lane ^ 32 the other half: split, swap.
__device__ __forceinline__ void s2_block_operand(const float (&e)[16], s2_u32x4 (&hi)[2], s2_u32x4 (&lo)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        unsigned ah0, al0, ah1, al1, bh0, bl0, bh1, bl1;
        s2_split2(e[8 * s + 0], e[8 * s + 1], ah0, al0); s2_split2(e[8 * s + 2], e[8 * s + 3], ah1, al1);
        s2_split2(e[8 * s + 4], e[8 * s + 5], bh0, bl0); s2_split2(e[8 * s + 6], e[8 * s + 7], bh1, bl1);
        s2_swap(ah0, bh0); s2_swap(ah1, bh1); s2_swap(al0, bl0); s2_swap(al1, bl1);
        hi[s] = (s2_u32x4){ah0, ah1, bh0, bh1};
        lo[s] = (s2_u32x4){al0, al1, bl0, bl1};
    }
}

// Dropout.  The keep decision of score (b, head, q, key) is dropout_keep(seed, site, ((b h + head) P + q) P + key) as in sattn.hip and the
// materialised path -- 19 of the ~30 VALU issue slots a score costs in the forward.  MODE 1 recomputes it in all three kernels; MODE 2
// computes it in the forward only, which leaves the decisions behind as BITS (B h P^2 / 8 bytes: 25 MB at B = 48): per 32 x 32 block
// (query block, key block) 32 words, word 2 v + half = the 32 queries' bits of key (v & 3) + 8 (v >> 2) + 4 half -- the two words of a v
// are the forward's wave-wide compare result of accumulator register v, so the dQ kernel (same lane layout) loads them with scalar loads
// straight into the mask operand of v_cndmask, and the dK / dV kernel (lane = key) loads its key's word and tests 16 of its bits.
// MODE 0: no dropout.
//
// forward: one work-group = 128 queries of one (sample, head), wave = 32 queries (lane = query lj, key half kb); K / V stream through
// LDS in tiles of 64 keys (K: row image, V: tr image); per 32-key block: S^T = K Q^T (6 MFMAs), online softmax (base 2: log2 e is folded
// into the scale of Q), dropout, O^T += V^T P^T (6); 1 / (1 - p) is applied to O at the end
template <int MODE>
__global__ __launch_bounds__(256, 2) void sattn2_fwd_kernel(SAttn2P p) {
    __shared__ __attribute__((aligned(16))) char KR[2][2 * S2_RIMG];      // K tile: row image hi | lo
    __shared__ __attribute__((aligned(16))) char VT[2][2 * S2_TIMG];      // V tile: tr image hi | lo
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lj = lane & 31, kb = lane >> 5, a16 = lane & 15, g16 = (lane >> 4) & 1;
    int xb, bh;
    s2_block(xb, bh);
    const int b = bh / p.h, head = bh - b * p.h, hoff = head * S2_D;
    const int q = xb * 128 + wave * 32 + lj;
    const long brow = (long)b * p.P;
    s2_u32x4 qh[2], ql[2];
    s2_row_operand(p.Q, brow + q, p.E, hoff, kb, p.scale * S2_LOG2E, qh, ql);
    const uint64_t sd = MODE ? p.seed[0] : 0ull;
    const uint32_t th = dropout_thresh(p.pdrop);
    const uint32_t k0 = (uint32_t)sd ^ (p.site * 0x9E3779B9u), k1 = (uint32_t)(sd >> 32) + p.site * 0x85EBCA77u;
    const uint32_t rowidx = ((uint32_t)bh * p.P + q) * (uint32_t)p.P + 4 * kb;      // flat index of (b, head, q, key 4 kb)
    const int nb = p.P / 32;
    unsigned* bits = MODE == 2 ? p.bits + ((size_t)bh * nb + xb * 4 + wave) * nb * 32 : nullptr;
    s2_f32x16 o;
#pragma unroll
    for (int v = 0; v < 16; ++v) o[v] = 0.f;
    float m = -INFINITY, lpart = 0.f;
    const int nt = p.P / S2_T;
    const int trp = (8 * kb + (a16 >> 2)) * S2_PT + 32 * g16 + 8 * (a16 & 3);      // this lane's piece of the [4 rows][16 channels] blocks
    f32x4 ka, kb4, va, vb;
    s2_fetch(p.K, brow, p.E, hoff, tid, ka, kb4);
    s2_fetch(p.V, brow, p.E, hoff, tid, va, vb);
    s2_put<true, false>(KR[0], nullptr, tid, ka, kb4);
    s2_put<false, true>(nullptr, VT[0], tid, va, vb);
    __syncthreads();
    for (int kt = 0; kt < nt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nt) {
            s2_fetch(p.K, brow + (long)(kt + 1) * S2_T, p.E, hoff, tid, ka, kb4);
            s2_fetch(p.V, brow + (long)(kt + 1) * S2_T, p.E, hoff, tid, va, vb);
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            // S^T block: register 4 g + r = key 32 sub + 8 g + 4 kb + r of query lj
            const s2_f32x16 st = s2_rows_x_operand(KR[cur], sub, lj, kb, qh, ql);
            float mt = st[0];
#pragma unroll
            for (int v = 1; v < 16; ++v) mt = fmaxf(mt, st[v]);
            mt = s2_pair_max(mt);
            const float mn = fmaxf(m, mt);
            const float alpha = s2_exp2(m - mn);
            float e[16], ls = 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) { e[v] = s2_exp2(st[v] - mn); ls += e[v]; }
            lpart = lpart * alpha + ls;                          // (this lane's 16 keys of every block; the pair is joined at the end)
            m = mn;
            if (MODE) {
                unsigned w = 0;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const bool keep = s2_keep(k0, k1, rowidx + (uint32_t)(kt * S2_T + 32 * sub + 8 * (v >> 2) + (v & 3)), th);
                    e[v] = keep ? e[v] : 0.f;
                    if (MODE == 2) {
                        const unsigned long long bl = __builtin_amdgcn_ballot_w64(keep);
                        w = s2_writelane2(w, bl, 2 * v);
                    }
                }
                if (MODE == 2 && lane < 32) bits[(size_t)(2 * kt + sub) * 32 + lane] = w;
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) o[v] *= alpha;
            // O^T[d][q] += V^T[d][key] P^T[key][q]
            s2_u32x4 ph[2], pl[2];
            s2_block_operand(e, ph, pl);
            o = s2_tr_x_block(VT[cur] + trp, sub, ph, pl, o);
        }
        if (kt + 1 < nt) {
            s2_put<true, false>(KR[cur ^ 1], nullptr, tid, ka, kb4);
            s2_put<false, true>(nullptr, VT[cur ^ 1], tid, va, vb);
        }
        __syncthreads();
    }
    const float l = s2_pair_sum(lpart);
    const float inv = (MODE ? 1.f / (1.f - p.pdrop) : 1.f) / l;
    // register 4 g + r = channel 8 g + 4 kb + r of query lj
    float* op = p.O + (brow + q) * p.E + hoff + 4 * kb;
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(op + 8 * g) = (f32x4){o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv};
    if (kb == 0) p.lse[(long)bh * p.P + q] = (m + __builtin_amdgcn_logf(l)) * S2_LN2;
}

// ---------------------------------------------------------------------------------------------------------------------------
// dK, dV: one work-group = 128 keys of one (sample, head), wave = 32 keys (lane = key lj, query half kb); Q / dO stream through LDS in
// tiles of 64 queries (row images for S = Q K^T and dP = dO V^T, tr images for dK^T += Q^T dS and dV^T += dO^T P); dO is staged times
// 1 / (1 - p).  Per 32-query block 24 MFMAs.
#define S2_KV_BUF (4 * S2_RIMG + 4 * S2_TIMG + 512)      // Q row | dO row | Q tr | dO tr | L[64] | D[64]
template <int MODE>
__global__ __launch_bounds__(256, 2) void sattn2_bwd_kv_kernel(SAttn2P p) {
    extern __shared__ __attribute__((aligned(16))) char s2_dyn[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lj = lane & 31, kb = lane >> 5, a16 = lane & 15, g16 = (lane >> 4) & 1;
    int xb, bh;
    s2_block(xb, bh);
    const int b = bh / p.h, head = bh - b * p.h, hoff = head * S2_D;
    const int key = xb * 128 + wave * 32 + lj;
    const long brow = (long)b * p.P;
    s2_u32x4 kh[2], kl[2], vh[2], vl[2];
    s2_row_operand(p.K, brow + key, p.E, hoff, kb, p.scale * S2_LOG2E, kh, kl);
    s2_row_operand(p.V, brow + key, p.E, hoff, kb, 1.f, vh, vl);
    const uint64_t sd = MODE == 1 ? p.seed[0] : 0ull;
    const uint32_t th = dropout_thresh(p.pdrop);
    const uint32_t k0 = (uint32_t)sd ^ (p.site * 0x9E3779B9u), k1 = (uint32_t)(sd >> 32) + p.site * 0x85EBCA77u;
    const float sc = MODE ? 1.f / (1.f - p.pdrop) : 1.f;
    const uint32_t colidx = (uint32_t)bh * p.P * (uint32_t)p.P + (uint32_t)(4 * kb) * p.P + key;   // flat index of (b, head, query 4 kb, key)
    const int nb = p.P / 32;
    // this lane's word of a block's 32: its key is register v = 4 (lj >> 3) + (lj & 3), half (lj >> 2) & 1 of the forward
    const unsigned* bits = MODE == 2 ? p.bits + (size_t)bh * nb * nb * 32 + (size_t)(xb * 4 + wave) * 32 + 2 * (4 * (lj >> 3) + (lj & 3)) + ((lj >> 2) & 1) : nullptr;
    s2_f32x16 accK, accV;
#pragma unroll
    for (int v = 0; v < 16; ++v) { accK[v] = 0.f; accV[v] = 0.f; }
    const int nt = p.P / S2_T;
    const int trp = (8 * kb + (a16 >> 2)) * S2_PT + 32 * g16 + 8 * (a16 & 3);
    f32x4 qa, qb, ga, gb;
    float lq = 0.f, dq = 0.f;
    unsigned wn[2] = {0u, 0u};
    auto fetch = [&](int qt) {
        s2_fetch(p.Q, brow + (long)qt * S2_T, p.E, hoff, tid, qa, qb);
        s2_fetch(p.dO, brow + (long)qt * S2_T, p.E, hoff, tid, ga, gb);
        ga *= sc; gb *= sc;
        if (tid < S2_T) { lq = p.lse[(long)bh * p.P + qt * S2_T + tid] * S2_LOG2E; dq = p.Dv[(long)bh * p.P + qt * S2_T + tid]; }
        if (MODE == 2) { wn[0] = bits[(size_t)(2 * qt) * nb * 32]; wn[1] = bits[(size_t)(2 * qt + 1) * nb * 32]; }
    };
    auto stash = [&](char* buf) {
        s2_put<true, true>(buf, buf + 4 * S2_RIMG, tid, qa, qb);
        s2_put<true, true>(buf + 2 * S2_RIMG, buf + 4 * S2_RIMG + 2 * S2_TIMG, tid, ga, gb);
        if (tid < S2_T) { reinterpret_cast<float*>(buf + 4 * S2_RIMG + 4 * S2_TIMG)[tid] = lq; reinterpret_cast<float*>(buf + 4 * S2_RIMG + 4 * S2_TIMG)[64 + tid] = dq; }
    };
    fetch(0);
    stash(s2_dyn);
    __syncthreads();
    for (int qt = 0; qt < nt; ++qt) {
        const char* buf = s2_dyn + (qt & 1) * S2_KV_BUF;
        const unsigned wc[2] = {wn[0], wn[1]};
        if (qt + 1 < nt) fetch(qt + 1);
        const float* Ls = reinterpret_cast<const float*>(buf + 4 * S2_RIMG + 4 * S2_TIMG);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            // S and dP blocks: register 4 g + r = query 32 sub + 8 g + 4 kb + r, column = key lj
            const s2_f32x16 st = s2_rows_x_operand(buf, sub, lj, kb, kh, kl);
            const s2_f32x16 dp = s2_rows_x_operand(buf + 2 * S2_RIMG, sub, lj, kb, vh, vl);
            const unsigned wsh = wc[sub] >> (4 * kb);
            float pd[16], ds[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 L4 = *reinterpret_cast<const f32x4*>(Ls + 32 * sub + 8 * g + 4 * kb), D4 = *reinterpret_cast<const f32x4*>(Ls + 64 + 32 * sub + 8 * g + 4 * kb);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int v = 4 * g + r;
                    const float pr = s2_exp2(st[v] - L4[r]);
                    float x = dp[v];
                    pd[v] = pr;
                    if (MODE == 1) {
                        const bool keep = s2_keep(k0, k1, colidx + (uint32_t)(qt * S2_T + 32 * sub + 8 * g + r) * (uint32_t)p.P, th);
                        pd[v] = keep ? pr : 0.f;
                        x = keep ? x : 0.f;
                    }
                    if (MODE == 2) {
                        const int mk = (int)(wsh << (31 - (8 * g + r))) >> 31;           // all ones = kept
                        pd[v] = __builtin_bit_cast(float, __builtin_bit_cast(int, pr) & mk);
                        x = __builtin_bit_cast(float, __builtin_bit_cast(int, x) & mk);
                    }
                    ds[v] = pr * (x - D4[r]);
                }
            }
            s2_u32x4 oh[2], ol[2];
            s2_block_operand(pd, oh, ol);
            accV = s2_tr_x_block(buf + 4 * S2_RIMG + 2 * S2_TIMG + trp, sub, oh, ol, accV);
            s2_block_operand(ds, oh, ol);
            accK = s2_tr_x_block(buf + 4 * S2_RIMG + trp, sub, oh, ol, accK);
        }
        if (qt + 1 < nt) stash(s2_dyn + ((qt + 1) & 1) * S2_KV_BUF);
        __syncthreads();
    }
    // register 4 g + r = channel 8 g + 4 kb + r of key lj
    const long o0 = (brow + key) * p.E + hoff + 4 * kb;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<f32x4*>(p.dK + o0 + 8 * g) = (f32x4){accK[4 * g] * p.scale, accK[4 * g + 1] * p.scale, accK[4 * g + 2] * p.scale, accK[4 * g + 3] * p.scale};
        *reinterpret_cast<f32x4*>(p.dV + o0 + 8 * g) = (f32x4){accV[4 * g], accV[4 * g + 1], accV[4 * g + 2], accV[4 * g + 3]};
    }
}

// dQ: one work-group = 128 queries of one (sample, head), wave = 32 queries (lane = query lj, key half kb: the forward's layout); K / V
// stream through LDS (row images for S^T = K Q^T and dP^T = V dO^T, K tr image for dQ^T += K^T dS^T).  Per 32-key block 18 MFMAs.
template <int MODE>
__global__ __launch_bounds__(256, 2) void sattn2_bwd_q_kernel(SAttn2P p) {
    __shared__ __attribute__((aligned(16))) char KR[2][2 * S2_RIMG];
    __shared__ __attribute__((aligned(16))) char VR[2][2 * S2_RIMG];
    __shared__ __attribute__((aligned(16))) char KT[2][2 * S2_TIMG];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lj = lane & 31, kb = lane >> 5, a16 = lane & 15, g16 = (lane >> 4) & 1;
    int xb, bh;
    s2_block(xb, bh);
    const int b = bh / p.h, head = bh - b * p.h, hoff = head * S2_D;
    const int q = xb * 128 + wave * 32 + lj;
    const long brow = (long)b * p.P;
    const uint64_t sd = MODE == 1 ? p.seed[0] : 0ull;
    const uint32_t th = dropout_thresh(p.pdrop);
    const uint32_t k0 = (uint32_t)sd ^ (p.site * 0x9E3779B9u), k1 = (uint32_t)(sd >> 32) + p.site * 0x85EBCA77u;
    const float sc = MODE ? 1.f / (1.f - p.pdrop) : 1.f;
    s2_u32x4 qh[2], ql[2], gh[2], gl[2];
    s2_row_operand(p.Q, brow + q, p.E, hoff, kb, p.scale * S2_LOG2E, qh, ql);
    s2_row_operand(p.dO, brow + q, p.E, hoff, kb, sc, gh, gl);
    const float L2 = p.lse[(long)bh * p.P + q] * S2_LOG2E, Dq = p.Dv[(long)bh * p.P + q];
    const uint32_t rowidx = ((uint32_t)bh * p.P + q) * (uint32_t)p.P + 4 * kb;
    const int nb = p.P / 32;
    // (constant address space: written by the forward launch, never by this one -- uniform loads from it are scalar loads)
    typedef const __attribute__((address_space(4))) unsigned long long* s2_cmask;
    const s2_cmask bits = MODE == 2 ? (s2_cmask)(p.bits) + ((size_t)bh * nb + xb * 4 + wave) * nb * 16 : (s2_cmask)nullptr;
    s2_f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    const int nt = p.P / S2_T;
    const int trp = (8 * kb + (a16 >> 2)) * S2_PT + 32 * g16 + 8 * (a16 & 3);
    f32x4 ka, kb4, va, vb;
    s2_fetch(p.K, brow, p.E, hoff, tid, ka, kb4);
    s2_fetch(p.V, brow, p.E, hoff, tid, va, vb);
    s2_put<true, true>(KR[0], KT[0], tid, ka, kb4);
    s2_put<true, false>(VR[0], nullptr, tid, va, vb);
    __syncthreads();
    for (int kt = 0; kt < nt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nt) {
            s2_fetch(p.K, brow + (long)(kt + 1) * S2_T, p.E, hoff, tid, ka, kb4);
            s2_fetch(p.V, brow + (long)(kt + 1) * S2_T, p.E, hoff, tid, va, vb);
        }
        unsigned long long mk[32];                               // the tile's 32 masks: scalar loads, issued ahead of the products
        if (MODE == 2) {
#pragma unroll
            for (int v = 0; v < 32; ++v) mk[v] = bits[(size_t)kt * 32 + v];
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            // S^T and dP^T blocks: register 4 g + r = key 32 sub + 8 g + 4 kb + r of query lj
            const s2_f32x16 st = s2_rows_x_operand(KR[cur], sub, lj, kb, qh, ql);
            const s2_f32x16 dp = s2_rows_x_operand(VR[cur], sub, lj, kb, gh, gl);
            float ds[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float pr = s2_exp2(st[v] - L2);
                if (MODE == 2) {
                    // (the asm select must not read an MFMA or v_exp result directly: the compiler's hazard recogniser does not
                    // see into it -- it gets the product, written by a plain VALU instruction)
                    ds[v] = s2_sel(pr * dp[v], mk[16 * sub + v]) - pr * Dq;
                } else {
                    float x = dp[v];
                    if (MODE == 1) x = s2_keep(k0, k1, rowidx + (uint32_t)(kt * S2_T + 32 * sub + 8 * (v >> 2) + (v & 3)), th) ? x : 0.f;
                    ds[v] = pr * (x - Dq);
                }
            }
            s2_u32x4 dh[2], dl[2];
            s2_block_operand(ds, dh, dl);
            acc = s2_tr_x_block(KT[cur] + trp, sub, dh, dl, acc);
        }
        if (kt + 1 < nt) {
            s2_put<true, true>(KR[cur ^ 1], KT[cur ^ 1], tid, ka, kb4);
            s2_put<true, false>(VR[cur ^ 1], nullptr, tid, va, vb);
        }
        __syncthreads();
    }
    float* op = p.dQ + (brow + q) * p.E + hoff + 4 * kb;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(op + 8 * g) = (f32x4){acc[4 * g] * p.scale, acc[4 * g + 1] * p.scale, acc[4 * g + 2] * p.scale, acc[4 * g + 3] * p.scale};
}

static int sattn_generation = 2;                 // tatt_sattn_generation: 1 = the exact-fp32 kernels of sattn.hip
TATT_API int tatt_sattn_generation(int gen) {
    const int old = sattn_generation;
    if (gen == 1 || gen == 2) sattn_generation = gen;
    return old;
}
// geometry the split-bf16 kernels take: 128-row work-groups, 32-bit dropout indices
static bool sattn2_takes(int B, int P, int h) { return sattn_generation == 2 && P % 128 == 0 && (double)B * h * P * P < 4.0e9; }
static int sattn2_mode(float pdrop, const unsigned* bits) { return pdrop > 0.f ? (bits ? 2 : 1) : 0; }
// -1 = not taken (the caller runs the fp32 kernels, which recompute the masks and never touch `bits`)
int sattn2_fwd(const float* Q, const float* K, const float* V, float* O, float* lse, int B, int P, int h, float scale, float pdrop,
               const unsigned long long* seed, unsigned site, unsigned* bits, hipStream_t st) {
    if (!sattn2_takes(B, P, h)) return -1;
    SAttn2P p = {Q, K, V, O, lse, nullptr, nullptr, nullptr, nullptr, nullptr, B, P, h, h * S2_D, scale, pdrop, seed, site, bits};
    const dim3 grid(P / 128, B * h);
    switch (sattn2_mode(pdrop, bits)) {
        case 0: hipLaunchKernelGGL(sattn2_fwd_kernel<0>, grid, dim3(256), 0, st, p); break;
        case 1: hipLaunchKernelGGL(sattn2_fwd_kernel<1>, grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL(sattn2_fwd_kernel<2>, grid, dim3(256), 0, st, p); break;
    }
    return LAUNCH_CHECK();
}
// (the caller has launched the D = rowsum(dO * O) kernel)
int sattn2_bwd(const float* Q, const float* K, const float* V, const float* lse, const float* dO, const float* Dws, float* dQ, float* dK,
               float* dV, int B, int P, int h, float scale, float pdrop, const unsigned long long* seed, unsigned site, const unsigned* bits,
               hipStream_t st) {
    if (!sattn2_takes(B, P, h)) return -1;
    SAttn2P p = {Q, K, V, nullptr, const_cast<float*>(lse), dO, Dws, dQ, dK, dV, B, P, h, h * S2_D, scale, pdrop, seed, site, const_cast<unsigned*>(bits)};
    const dim3 grid(P / 128, B * h);
    const int lds = 2 * S2_KV_BUF;
    static TattPerDevice once;
    tatt_per_device(once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sattn2_bwd_kv_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sattn2_bwd_kv_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sattn2_bwd_kv_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    });
    switch (sattn2_mode(pdrop, bits)) {
        case 0:
            hipLaunchKernelGGL(sattn2_bwd_kv_kernel<0>, grid, dim3(256), lds, st, p);
            hipLaunchKernelGGL(sattn2_bwd_q_kernel<0>, grid, dim3(256), 0, st, p);
            break;
        case 1:
            hipLaunchKernelGGL(sattn2_bwd_kv_kernel<1>, grid, dim3(256), lds, st, p);
            hipLaunchKernelGGL(sattn2_bwd_q_kernel<1>, grid, dim3(256), 0, st, p);
            break;
        default:
            hipLaunchKernelGGL(sattn2_bwd_kv_kernel<2>, grid, dim3(256), lds, st, p);
            hipLaunchKernelGGL(sattn2_bwd_q_kernel<2>, grid, dim3(256), 0, st, p);
            break;
    }
    return LAUNCH_CHECK();
}
