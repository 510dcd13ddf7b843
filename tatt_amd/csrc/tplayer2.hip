// Backward of one TP-interpreter transformer layer, second generation (round 5).  Same mathematics, inputs and outputs as
// tplayer_kernel<true> (tplayer.hip; reference model/transformer_v2.py:806-833, 380-390, 470-484): recompute the layer from
// x, qpos, K, V and walk it in reverse.  What changed is the organisation -- the first generation spent 53 % of its wave
// cycles parked (17 work-group barriers per 32-token tile, two waves per SIMD arriving together), 11 VALU per MFMA, 31 % of
// its LDS cycles in bank conflicts and spilled ~50 registers (profiles/r03_pmc_tplayer_bwd.txt):
//
//   * A WAVE owns 16 tokens and ALL 64 channels and runs the whole chain on them (one wave per SIMD, up to 512 registers);
//     the four waves of a work-group meet only where a product contracts over TOKENS and wants more of them than a wave has
//     (the four weight gradients: 4 barriers each per round of 64 tokens, phases of thousands of cycles in between).
//   * Every product is computed TRANSPOSED: D^T[n][t] = sum_k W[n][k] X[t][k], weights as the MFMA A operand, tokens as
//     columns.  In the C layout of v_mfma_f32_16x16x* a lane (am = lane & 15, kq = lane >> 4) then holds, for its token am,
//     channels 16 nb + 4 kq + r (nb, r = 0..3) -- and that IS a legal B operand of the next product if the contraction
//     index is enumerated in the same order (k-slot j of k-step ks of lane group kq <-> channel 16 (2 ks + (j >> 2)) + 4 kq
//     + (j & 3); the weight images are packed in that order by tplayer2_prep_kernel).  A chain of products (x+qpos -> Q ->
//     scores -> P -> ctx -> out-projection -> LayerNorm -> FFN ...) therefore never leaves the registers: no LDS staging,
//     no layout change, the softmax over the keys and the LayerNorm over the channels are in-lane sums + two lane exchanges.
//   * Arithmetic: split bf16 on the bf16 matrix cores, a = hi + lo, a b ~ hi hi + hi lo + lo hi with fp32 accumulation
//     (2^-16 relative per product, the arithmetic of conv3_c64_sb / tokgemm_sb; tatt_amd.set_arithmetic("fp32") selects the
//     exact-fp32 first generation).  A tensor is split ONCE, when it is produced, into packed words that serve both as the
//     next product's B operand and -- written to the wave's LDS image [channel][token] -- as the operands of the products
//     that contract over tokens.
//   * Weight gradients: wave w accumulates rows 16 w .. 16 w + 15 of all four 64 x 64 gradients over the tokens of ALL four
//     waves (64 accumulator registers instead of 256: with 256 + dK / dV the kernel spilled 583 registers): every wave
//     writes the image of its G = dY tile, reads its row block of all four images, writes the image of its X tile, reads
//     all of them; K = 32 tokens (two waves' tiles) per v_mfma_f32_16x16x32_bf16.  Bias gradients ride along as an all-ones
//     B column.  dK / dV of the wave's current sample stay wave-private accumulators (v_mfma_f32_16x16x16_bf16, K = the
//     wave's 16 tokens) and are merged through LDS at the end: <= 3 records per work-group.
//   * Dropout masks: the hash of the stand-alone kernels (same seed word, site, flat index), evaluated once per element in
//     the recomputed forward and kept as bit masks for the backward half of the tile.
//
// LDS (158,208 bytes, one work-group per CU): eight weight images (fwd and transposed, hi and lo: 128 KB) in MFMA-fragment
// order -- a wave's 16-byte fragment load is 1 KB of consecutive LDS, conflict-free --, ten parameter vectors, one 6 KB
// image buffer per wave (row pitch 48 bytes: ds_read_b64 of 16 rows x 4 words covers the 64 banks exactly once).
// K and V of the sample arrive as MFMA fragments from global memory (tplayer2_prep_kernel packs four forms per sample,
// 32 KB, L2-resident), so no per-sample LDS state ties the waves together.
#include "common.h"
#include <mutex>

typedef __bf16 t2_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 t2_bf16x2 __attribute__((ext_vector_type(2)));
typedef short t2_s16x4 __attribute__((ext_vector_type(4)));
typedef float t2_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned t2_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned t2_u32x2 __attribute__((ext_vector_type(2)));

#define T2_NT 256
#define T2_WIMG_WORDS (8 * 4096)                 // 8 images x [4 blk][2 ks][2 hl][64 lanes][4 words]
#define T2_KVF_WORDS 8192                        // per sample: KF1, VF1 (2048 words each), VTF, KTF (2048 each)
#define T2_VEC_OFF (T2_WIMG_WORDS * 4)           // bytes
#define T2_TB_OFF (T2_VEC_OFF + 10 * 64 * 4)
#define T2_TB_BYTES 6144                         // per wave: 2 planes x 64 rows x 48 bytes; or 16 x 68 floats
#define T2_LDS_BYTES (T2_TB_OFF + 4 * T2_TB_BYTES)
#define T2_PREC (4 * 4096 + 10 * 64)             // = TL_PREC of tplayer.hip: the parameter reducer is shared
#define T2_KVREC 4096

struct T2P {
    const float* x; const float* qpos; long qbs;
    const unsigned* wimg; const unsigned* kvf;
    const float* bv[4]; const float* lnw[3]; const float* lnb[3];
    float fin_scale; int fin_both;
    int B, L, S, tps, ntiles, nper;
    float p_attn, p_res, p_ffn; const unsigned long long* seed; unsigned site0; float eps;
    const float* dxout; const float* dfin; const float* dwavg; const float* dqacc;
    float* dx; float* dqpos; float* kvpart; float* ppart; int* kvflags;
    const unsigned long long* hmask;                 // relu-and-kept bits of the forward launch (tatt_tplayer_fwd_m), nullable
};

struct T2Pack { t2_u32x4 h[2], l[2]; };          // a (16 tokens x 64 channels) tensor split into bf16 hi / lo, B-operand order

__device__ __forceinline__ void t2_split2(float a, float b, unsigned& hi, unsigned& lo) {
    const t2_f32x2 v = (t2_f32x2){a, b};
    const t2_bf16x2 h = __builtin_convertvector(v, t2_bf16x2);
    const t2_bf16x2 l = __builtin_convertvector(v - __builtin_convertvector(h, t2_f32x2), t2_bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// word e of k-step ks: block nb = 2 ks + (e >> 1), elements r = 2 (e & 1), 2 (e & 1) + 1
__device__ __forceinline__ void t2_pack16(const f32x4 (&v)[4], T2Pack& o) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 s = v[2 * ks + (e >> 1)];
            unsigned hi, lo;
            t2_split2(s[2 * (e & 1)], s[2 * (e & 1) + 1], hi, lo);
            o.h[ks][e] = hi; o.l[ks][e] = lo;
        }
}
__device__ __forceinline__ t2_u32x2 t2_sub4(const t2_u32x4 (&w)[2], int nb) {      // the 4 elements of block nb
    return (t2_u32x2){w[nb >> 1][2 * (nb & 1)], w[nb >> 1][2 * (nb & 1) + 1]};
}
__device__ __forceinline__ f32x4 t2_mfma32(t2_u32x4 a, t2_u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(t2_bf16x8, a), __builtin_bit_cast(t2_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 t2_mfma16(t2_u32x2 a, t2_u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(t2_s16x4, a), __builtin_bit_cast(t2_s16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 t2_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void t2_st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ float t2_sum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
// sum over the four lane groups kq of a token (lanes am, am + 16, am + 32, am + 48), result in all of them
__device__ __forceinline__ float t2_allkq(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float t2_maxkq(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v; }
// dropout_keep of common.h for flat indices below 2^32 (the launcher checks): identical masks
__device__ __forceinline__ bool t2_keep(uint32_t k0, uint32_t k1, uint32_t idx, uint32_t thresh) {
    uint32_t h = idx ^ k0;
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h += k1;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h >= thresh;
}

__global__ __launch_bounds__(T2_NT, 1) void tplayer2_bwd_kernel(T2P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char t2_smem[];
    const t2_u32x4* const Wl = reinterpret_cast<const t2_u32x4*>(t2_smem);
    float* const Vec = reinterpret_cast<float*>(t2_smem + T2_VEC_OFF);
    const int tid = threadIdx.x, wave = tid >> 6;
    // lane, am, kq are re-derived from an opaque copy at the top of every round: otherwise the compiler hoists ~100 lane-dependent
    // addresses (LDS fragment offsets, 64-bit row pointers of every global array) out of the tile loop and spills them (258 spilled
    // registers, 144 scratch loads per round; with the copy: 7)
    int lane = tid & 63;
    int am = lane & 15, kq = lane >> 4;
    unsigned char* const TB = t2_smem + T2_TB_OFF + wave * T2_TB_BYTES;
    const bool fin_on = p.lnw[2] != nullptr;
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- resident operands: weight images (straight copy of the packed form), parameter vectors -----------------------------
    {
        const t2_u32x4* src = reinterpret_cast<const t2_u32x4*>(p.wimg);
        t2_u32x4* dst = reinterpret_cast<t2_u32x4*>(t2_smem);
#pragma unroll 8
        for (int i = tid; i < T2_WIMG_WORDS / 4; i += T2_NT) dst[i] = src[i];
        for (int i = tid; i < 10 * 64; i += T2_NT) {
            const int v = i >> 6, c = i & 63;
            const float* s = v < 4 ? p.bv[v] : (((v - 4) & 1) ? p.lnb[(v - 4) >> 1] : p.lnw[(v - 4) >> 1]);
            Vec[i] = s ? s[c] : 0.f;
        }
    }
    __syncthreads();
    auto vec4 = [&](int v, int nb) __attribute__((always_inline)) -> f32x4 { return t2_ld4(Vec + v * 64 + 16 * nb + 4 * kq); };
    auto wfrag = [&](int im, int blk, int ks, int hl) __attribute__((always_inline)) -> t2_u32x4 { return Wl[(((im * 4 + blk) * 2 + ks) * 2 + hl) * 64 + lane]; };
    // D^T = W X^T: acc[nb] rows = output channels 16 nb .., columns = the wave's 16 tokens
    auto gemm16 = [&](int im, const T2Pack& b, f32x4 (&acc)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] = z4;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            t2_u32x4 ah[4], al[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) { ah[nb] = wfrag(im, nb, ks, 0); al[nb] = wfrag(im, nb, ks, 1); }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = t2_mfma32(ah[nb], b.h[ks], acc[nb]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = t2_mfma32(ah[nb], b.l[ks], acc[nb]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = t2_mfma32(al[nb], b.h[ks], acc[nb]);
        }
    };
    // ---- wave-private transposition buffer: image [row = channel][token] of bf16, hi plane then lo plane, pitch 48 bytes ------
    auto tb_write = [&](const t2_u32x4 (&H)[2], const t2_u32x4 (&Lo)[2]) __attribute__((always_inline)) {
        unsigned short* const hp = reinterpret_cast<unsigned short*>(TB);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 16 * (2 * ks + (e >> 1)) + 4 * kq + 2 * (e & 1);
                unsigned short* d = hp + row * 24 + am;
                const unsigned wh = H[ks][e], wl = Lo[ks][e];
                d[0] = (unsigned short)wh; d[24] = (unsigned short)(wh >> 16);
                d[1536] = (unsigned short)wl; d[1536 + 24] = (unsigned short)(wl >> 16);
            }
    };
    // operand fragment of rows 16 blk .. 16 blk + 15: lane (am, kq) <- row 16 blk + am, tokens 4 kq .. 4 kq + 3
    auto tb_frag = [&](int blk, int plane) __attribute__((always_inline)) -> t2_u32x2 {
        return *reinterpret_cast<const t2_u32x2*>(TB + plane * 3072 + (16 * blk + am) * 48 + 8 * kq);
    };
    // sum over the tile's 16 tokens of a tensor held in the C layout -> the value of channel `lane` (fp32 through LDS)
    auto tok_sum = [&](const f32x4 (&e)[4]) __attribute__((always_inline)) -> float {
        float* const tf = reinterpret_cast<float*>(TB);
        wave_lds_sync();
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) t2_st4(tf + am * 68 + 16 * nb + 4 * kq, e[nb]);
        wave_lds_sync();
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int t = 0; t < 16; t += 2) { s0 += tf[t * 68 + lane]; s1 += tf[(t + 1) * 68 + lane]; }
        wave_lds_sync();
        return s0 + s1;
    };
    // LayerNorm over the 64 channels of the lane's token
    auto ln_fwd = [&](const f32x4 (&v)[4], f32x4 (&xh)[4], float& rstd) __attribute__((always_inline)) {
        const float mean = t2_allkq((t2_sum4(v[0]) + t2_sum4(v[1])) + (t2_sum4(v[2]) + t2_sum4(v[3]))) * (1.f / 64.f);
        float q = 0.f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) { xh[nb] = v[nb] - mean; q += t2_sum4(xh[nb] * xh[nb]); }
        rstd = __builtin_amdgcn_rsqf(t2_allkq(q) * (1.f / 64.f) + p.eps);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) xh[nb] = xh[nb] * rstd;
    };
    auto ln_bwd = [&](const f32x4 (&g)[4], const f32x4 (&xh)[4], int gv, float rstd, f32x4 (&o)[4]) __attribute__((always_inline)) {
        f32x4 gg[4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) { gg[nb] = g[nb] * vec4(gv, nb); s1 += t2_sum4(gg[nb]); s2 += t2_sum4(gg[nb] * xh[nb]); }
        s1 = t2_allkq(s1) * (1.f / 64.f); s2 = t2_allkq(s2) * (1.f / 64.f);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) o[nb] = (gg[nb] - s1 - xh[nb] * s2) * rstd;
    };

    // ---- dropout constants -------------------------------------------------------------------------------------------------------
    const bool any_drop = p.p_attn > 0.f || p.p_res > 0.f || p.p_ffn > 0.f;
    const uint64_t sd = any_drop ? p.seed[0] : 0ull;
    const uint32_t th_attn = dropout_thresh(p.p_attn), th_res = dropout_thresh(p.p_res), th_ffn = dropout_thresh(p.p_ffn);
    const float sc_attn = p.p_attn > 0.f ? 1.f / (1.f - p.p_attn) : 1.f;
    const float sc_res = p.p_res > 0.f ? 1.f / (1.f - p.p_res) : 1.f;
    const float sc_ffn = p.p_ffn > 0.f ? 1.f / (1.f - p.p_ffn) : 1.f;
    auto key0 = [&](unsigned site) __attribute__((always_inline)) -> uint32_t { return (uint32_t)sd ^ (site * 0x9E3779B9u); };
    auto key1 = [&](unsigned site) __attribute__((always_inline)) -> uint32_t { return (uint32_t)(sd >> 32) + site * 0x85EBCA77u; };
    const uint32_t ka0 = key0(p.site0), ka1 = key1(p.site0);
    const uint32_t kr0 = key0(p.site0 + 1), kr1 = key1(p.site0 + 1);
    const uint32_t kf0 = key0(p.site0 + 2), kf1 = key1(p.site0 + 2);
    const uint32_t ks0 = key0(p.site0 + 3), ks1 = key1(p.site0 + 3);

    // ---- accumulators that live across the tiles of this wave --------------------------------------------------------------------
    f32x4 dWm[4][4];                                         // [matrix: Wq, Wo, W1, W2][column block kb]: rows 16 wave .. of the gradient
    f32x4 accK[4][2], accV[4][2];                            // [head][key block] of the wave's current sample
    f32x4 accB = z4;                                         // bias gradients: column m = matrix m, rows 16 wave ..
    float dgA = 0.f, dbA = 0.f, dgB = 0.f, dbB = 0.f, dgF = 0.f, dbF = 0.f;     // LayerNorm gradients of channel `lane`
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dWm[i][j] = z4;
        accK[i][0] = z4; accK[i][1] = z4; accV[i][0] = z4; accV[i][1] = z4;
    }
    const unsigned one2 = 0x3F803F80u;                       // two bf16 ones
    // fragment of another wave's image: rows 16 blk .. of wave `src`
    auto tb_frag_of = [&](int src, int blk, int plane) __attribute__((always_inline)) -> t2_u32x2 {
        return *reinterpret_cast<const t2_u32x2*>(t2_smem + T2_TB_OFF + src * T2_TB_BYTES + plane * 3072 + (16 * blk + am) * 48 + 8 * kq);
    };
    // rows 16 wave .. of dW[m] += G^T X over the 64 tokens of the work-group's four tiles; the bias gradient rides along as column m
    // of accB.  G, X: this wave's packed tensors.  Every wave of the work-group calls it at the same point (4 barriers).
    auto wgrad = [&](f32x4 (&dW)[4], int m, const T2Pack& G, const T2Pack& X) __attribute__((always_inline)) {
        tb_write(G.h, G.l);
        __syncthreads();
        t2_u32x2 gh[4], gl[4];
#pragma unroll
        for (int src = 0; src < 4; ++src) { gh[src] = tb_frag_of(src, wave, 0); gl[src] = tb_frag_of(src, wave, 1); }
        __syncthreads();
        tb_write(X.h, X.l);
        __syncthreads();
        const unsigned o = am == m ? one2 : 0u;
        const t2_u32x4 ones = (t2_u32x4){o, o, o, o};
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {                     // two source tiles per k-step: slots 0-3 tile 2 pr, slots 4-7 tile 2 pr + 1
            const t2_u32x4 ah = (t2_u32x4){gh[2 * pr][0], gh[2 * pr][1], gh[2 * pr + 1][0], gh[2 * pr + 1][1]};
            const t2_u32x4 al = (t2_u32x4){gl[2 * pr][0], gl[2 * pr][1], gl[2 * pr + 1][0], gl[2 * pr + 1][1]};
            t2_u32x4 bh[4], bl[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const t2_u32x2 h0 = tb_frag_of(2 * pr, kb, 0), h1 = tb_frag_of(2 * pr + 1, kb, 0);
                const t2_u32x2 l0 = tb_frag_of(2 * pr, kb, 1), l1 = tb_frag_of(2 * pr + 1, kb, 1);
                bh[kb] = (t2_u32x4){h0[0], h0[1], h1[0], h1[1]};
                bl[kb] = (t2_u32x4){l0[0], l0[1], l1[0], l1[1]};
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) dW[kb] = t2_mfma32(ah, bh[kb], dW[kb]);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) dW[kb] = t2_mfma32(ah, bl[kb], dW[kb]);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) dW[kb] = t2_mfma32(al, bh[kb], dW[kb]);
            accB = t2_mfma32(ah, ones, accB);
            accB = t2_mfma32(al, ones, accB);
        }
        __syncthreads();                                     // the images are private again
    };

    // ---- tiles ------------------------------------------------------------------------------------------------------------------------
    // Round rd of the work-group = tiles wg0 + 4 rd .. + 3, one per wave.  tps is a multiple of 4 (t2_takes), so the four tiles of a
    // round lie in ONE sample and the number of tiles is a multiple of 4: every round is whole, and a work-group's rounds fall into at
    // most two segments by sample (4 nper <= tps).  dK / dV of the first segment leave when it ends; the last one is merged at the end.
    const int wg0 = blockIdx.x * 4 * p.nper;
    const int nrd = min(p.nper, (p.ntiles - wg0) >> 2);
    const int b0 = wg0 / p.tps;
    const int split = min(nrd, ((b0 + 1) * p.tps - wg0) >> 2);                // rounds [0, split): sample b0, [split, nrd): b0 + 1
    float* const kv_wg = p.kvpart + (long)blockIdx.x * 5 * T2_KVREC;          // records 0-3: first segment per wave, 4: last segment
    // element (kv, key s = 16 sb + 4 kq + r, channel 16 h + am) of a dK / dV record
    auto kv_store = [&](float* rec) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = (16 * sb + 4 * kq + r) * 64 + 16 * h + am;
                    rec[i] = accK[h][sb][r];
                    rec[2048 + i] = accV[h][sb][r];
                }
    };

    for (int seg = 0; seg < 2; ++seg) {
    const int rd0 = seg ? split : 0, rd1 = seg ? nrd : split;
    for (int rd = rd0; rd < rd1; ++rd) {
        asm volatile("" : "+v"(lane));
        am = lane & 15; kq = lane >> 4;
        const int tile = wg0 + 4 * rd + wave;
        const int b = b0 + seg, tok0 = (tile - b * p.tps) * 16;
        const long row = (long)b * p.L + tok0 + am;          // global token row of this lane
        const float* const xrow = p.x + row * 64 + 4 * kq;
        const float* const qrow = p.qpos + (long)b * p.qbs + (long)(tok0 + am) * 64 + 4 * kq;
        const t2_u32x2* const kvf2 = reinterpret_cast<const t2_u32x2*>(p.kvf + (long)b * T2_KVF_WORDS);
        const t2_u32x4* const kvf4 = reinterpret_cast<const t2_u32x4*>(p.kvf + (long)b * T2_KVF_WORDS);
        const uint32_t ridx = (uint32_t)(row * 64) + 4 * kq;                                      // + 16 nb + r
        const uint32_t aidx = (uint32_t)(((long)b * 4 * p.L + tok0 + am) * p.S) + 4 * kq;         // + (h L S) + 16 sb + r
        const uint32_t hstep = (uint32_t)p.L * (uint32_t)p.S;
        unsigned kb_attn = 0, kb1 = 0, kb3 = 0;              // keep bits: attention (8 h + 4 sb + r), residual 1, residual 2 (4 nb + r)

        // ================================ stage A: forward to x1 = LN_A(x + drop(attention)) ======================================
        // (written head by head, operands loaded when they are needed: the kernel's register budget is what the accumulators leave)
        f32x4 xh1[4];
        float rstd1;
        T2Pack QP, CX;                                       // the scaled query projection and ctx, split: kept for the backward
        f32x4 P[4][2];                                       // un-dropped probabilities: [head][key block], keys 16 sb + 4 kq + r
        {
            f32x4 acc[4];
            {
                f32x4 xq[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) xq[nb] = t2_ld4(xrow + 16 * nb) + t2_ld4(qrow + 16 * nb);
                T2Pack XQ;
                t2_pack16(xq, XQ);
                gemm16(0, XQ, acc);
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] = (acc[nb] + vec4(0, nb)) * 0.25f;
            t2_pack16(acc, QP);
            f32x4 ctx[4];
            // the key / value fragments come from L2: those of head h + 1 are requested before head h is computed
            t2_u32x2 kfn[2][2];
            t2_u32x4 vtn[2];
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) { kfn[sb][0] = kvf2[(sb * 2 + 0) * 64 + lane]; kfn[sb][1] = kvf2[(sb * 2 + 1) * 64 + lane]; }
            vtn[0] = kvf4[1024 + lane]; vtn[1] = kvf4[1024 + 64 + lane];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                t2_u32x2 kf[2][2];
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) { kf[sb][0] = kfn[sb][0]; kf[sb][1] = kfn[sb][1]; }
                const t2_u32x4 vt0 = vtn[0], vt1 = vtn[1];
                if (h < 3) {
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb) { kfn[sb][0] = kvf2[(((h + 1) * 2 + sb) * 2 + 0) * 64 + lane]; kfn[sb][1] = kvf2[(((h + 1) * 2 + sb) * 2 + 1) * 64 + lane]; }
                    vtn[0] = kvf4[1024 + ((h + 1) * 2 + 0) * 64 + lane]; vtn[1] = kvf4[1024 + ((h + 1) * 2 + 1) * 64 + lane];
                }
                const t2_u32x2 qh = t2_sub4(QP.h, h), ql = t2_sub4(QP.l, h);
                f32x4 sc[2];
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    f32x4 a = t2_mfma16(kf[sb][0], qh, z4);
                    a = t2_mfma16(kf[sb][0], ql, a);
                    sc[sb] = t2_mfma16(kf[sb][1], qh, a);
                }
                float mx = -INFINITY;
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool live = 16 * sb + 4 * kq + r < p.S;
                        sc[sb][r] = live ? sc[sb][r] : -INFINITY;
                        mx = fmaxf(mx, sc[sb][r]);
                    }
                mx = t2_maxkq(mx);
                float sum = 0.f;
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = (16 * sb + 4 * kq + r < p.S) ? __expf(sc[sb][r] - mx) : 0.f;
                        P[h][sb][r] = e; sum += e;
                    }
                const float inv = __builtin_amdgcn_rcpf(t2_allkq(sum));
                f32x4 pd[2];
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pr = P[h][sb][r] * inv;
                        P[h][sb][r] = pr;
                        float d = pr;
                        if (p.p_attn > 0.f) {
                            const bool k = t2_keep(ka0, ka1, aidx + (uint32_t)h * hstep + 16 * sb + r, th_attn);
                            kb_attn |= k ? (1u << (8 * h + 4 * sb + r)) : 0u;
                            d = k ? pr * sc_attn : 0.f;
                        }
                        pd[sb][r] = d;
                    }
                unsigned w0, w1, w2, w3, l0, l1, l2, l3;
                t2_split2(pd[0][0], pd[0][1], w0, l0); t2_split2(pd[0][2], pd[0][3], w1, l1);
                t2_split2(pd[1][0], pd[1][1], w2, l2); t2_split2(pd[1][2], pd[1][3], w3, l3);
                const t2_u32x4 ph = (t2_u32x4){w0, w1, w2, w3}, pl = (t2_u32x4){l0, l1, l2, l3};
                f32x4 a = t2_mfma32(vt0, ph, z4);
                a = t2_mfma32(vt0, pl, a);
                ctx[h] = t2_mfma32(vt1, ph, a);
            }
            t2_pack16(ctx, CX);
            gemm16(1, CX, acc);
            f32x4 y1[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                f32x4 a = acc[nb] + vec4(1, nb);
                if (p.p_res > 0.f) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool k = t2_keep(kr0, kr1, ridx + 16 * nb + r, th_res);
                        kb1 |= k ? (1u << (4 * nb + r)) : 0u;
                        a[r] = k ? a[r] * sc_res : 0.f;
                    }
                }
                y1[nb] = t2_ld4(xrow + 16 * nb) + a;         // (x again: L1-resident, cheaper than 16 registers held across the attention)
            }
            ln_fwd(y1, xh1, rstd1);
        }
        // ================================ stage B: FFN forward and backward, LN_B / LN_F / LN_A backward ===============================
        f32x4 dxr[4];                                        // gradient reaching x through the attention block's residual
        T2Pack DA;                                           // da = dropout'(dxr), split
        {
            T2Pack X1, HD;
            f32x4 acc[4];
            {
                f32x4 x1[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) x1[nb] = xh1[nb] * vec4(4, nb) + vec4(5, nb);
                t2_pack16(x1, X1);
            }
            gemm16(2, X1, acc);
            // bit 4 nb + r: relu active and kept.  Taken from the FORWARD launch (tplayer2_fwd_kernel leaves them): a forward and a
            // recomputation that differ in the last bits (first version: exact-fp32 forward, 2^-16 products here) can disagree on the sign of a
            // pre-activation within ~1e-5 of zero, and a flipped relu moves that token's gradients by O(1) (measured: 4 of 49,152 tokens).
            unsigned hmask = 0;
            if (p.hmask) {
                const unsigned long long* hm = p.hmask + (long)tile * 16;                    // word [tile][nb][r], bit `lane` (tplayer2_fwd_kernel)
#pragma unroll
                for (int i = 0; i < 16; ++i) hmask |= ((unsigned)(hm[i] >> lane) & 1u) << i;
            }
            {
                f32x4 hd[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const f32x4 bj = vec4(2, nb);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float h = acc[nb][r] + bj[r];
                        if (p.hmask) {
                            h = ((hmask >> (4 * nb + r)) & 1u) ? h * sc_ffn : 0.f;
                        } else {
                            h = fmaxf(h, 0.f);
                            if (p.p_ffn > 0.f) h = t2_keep(kf0, kf1, ridx + 16 * nb + r, th_ffn) ? h * sc_ffn : 0.f;
                            hmask |= h > 0.f ? (1u << (4 * nb + r)) : 0u;
                        }
                        hd[nb][r] = h;
                    }
                }
                t2_pack16(hd, HD);
            }
            gemm16(3, HD, acc);
            f32x4 xh2[4];
            float rstd2;
            {
                f32x4 y2[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    f32x4 f = acc[nb] + vec4(3, nb);
                    if (p.p_res > 0.f) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool k = t2_keep(ks0, ks1, ridx + 16 * nb + r, th_res);
                            kb3 |= k ? (1u << (4 * nb + r)) : 0u;
                            f[r] = k ? f[r] * sc_res : 0.f;
                        }
                    }
                    y2[nb] = (xh1[nb] * vec4(4, nb) + vec4(5, nb)) + f;       // x1 again
                }
                ln_fwd(y2, xh2, rstd2);
            }
            f32x4 g2[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) g2[nb] = p.dxout ? t2_ld4(p.dxout + row * 64 + 4 * kq + 16 * nb) : z4;
            if (fin_on) {
                f32x4 dfr[4], xf[4];
                float rf;
                {
                    f32x4 x2[4];
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        dfr[nb] = t2_ld4(p.dfin + row * 64 + 4 * kq + 16 * nb) * p.fin_scale;
                        x2[nb] = xh2[nb] * vec4(6, nb) + vec4(7, nb);
                    }
                    ln_fwd(x2, xf, rf);
                }
                {
                    f32x4 e[4];
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) e[nb] = dfr[nb] * xf[nb];
                    dgF += tok_sum(e);
                }
                dbF += tok_sum(dfr);
                f32x4 o[4];
                ln_bwd(dfr, xf, 8, rf, o);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) g2[nb] += o[nb];
            }
            {
                f32x4 e[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) e[nb] = g2[nb] * xh2[nb];
                dgB += tok_sum(e);
            }
            dbB += tok_sum(g2);
            f32x4 dx1r[4];                                   // gradient reaching x1 through the FFN block's residual
            ln_bwd(g2, xh2, 6, rstd2, dx1r);
            T2Pack DF;
            {
                f32x4 df[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        df[nb][r] = (p.p_res > 0.f) ? ((kb3 >> (4 * nb + r)) & 1u ? dx1r[nb][r] * sc_res : 0.f) : dx1r[nb][r];
                t2_pack16(df, DF);
            }
            wgrad(dWm[3], 3, DF, HD);                        // dW2 += df^T hd, db2
            gemm16(7, DF, acc);                              // df W2
            T2Pack DH;
            {
                f32x4 dhp[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dhp[nb][r] = ((hmask >> (4 * nb + r)) & 1u) ? acc[nb][r] * sc_ffn : 0.f;
                t2_pack16(dhp, DH);
            }
            wgrad(dWm[2], 2, DH, X1);                        // dW1 += dhp^T x1, db1
            gemm16(6, DH, acc);                              // dhp W1
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] += dx1r[nb];                // g1: everything that reaches x1
            {
                f32x4 e[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) e[nb] = acc[nb] * xh1[nb];
                dgA += tok_sum(e);
            }
            dbA += tok_sum(acc);
            ln_bwd(acc, xh1, 4, rstd1, dxr);
            {
                f32x4 da[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        da[nb][r] = (p.p_res > 0.f) ? ((kb1 >> (4 * nb + r)) & 1u ? dxr[nb][r] * sc_res : 0.f) : dxr[nb][r];
                t2_pack16(da, DA);
            }
        }
        // ================================ stage C: attention backward ================================================================
        {
            wgrad(dWm[1], 1, DA, CX);                        // dWo += da^T ctx, dbo
            T2Pack DC;
            {
                f32x4 dctx[4];
                gemm16(5, DA, dctx);                         // da Wo
                t2_pack16(dctx, DC);
            }
            // operands of the token-contracting products dV += Pd^T dctx, dK += dS^T Q: the images of dctx and Q, read as B operands
            // (rows 16 h .. = the channels of head h), then the buffer serves the per-head-pair images of Pd and dS
            t2_u32x2 dth[4], dtl[4], qth[4], qtl[4];
            wave_lds_sync();
            tb_write(DC.h, DC.l);
            wave_lds_sync();
#pragma unroll
            for (int h = 0; h < 4; ++h) { dth[h] = tb_frag(h, 0); dtl[h] = tb_frag(h, 1); }
            wave_lds_sync();
            tb_write(QP.h, QP.l);
            wave_lds_sync();
#pragma unroll
            for (int h = 0; h < 4; ++h) { qth[h] = tb_frag(h, 0); qtl[h] = tb_frag(h, 1); }
            f32x4 dq[4];
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
                f32x4 pd[2][2], ds[2][2];                    // [head of the pair][key block]
#pragma unroll
                for (int hs = 0; hs < 2; ++hs) {
                    const int h = 2 * hp + hs;
                    t2_u32x2 vf[2][2];
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb) { vf[sb][0] = kvf2[1024 + ((h * 2 + sb) * 2 + 0) * 64 + lane]; vf[sb][1] = kvf2[1024 + ((h * 2 + sb) * 2 + 1) * 64 + lane]; }
                    const t2_u32x4 kt0 = kvf4[1536 + (h * 2 + 0) * 64 + lane], kt1 = kvf4[1536 + (h * 2 + 1) * 64 + lane];
                    // dPd^T[s][t] = sum_c V[s][c] dctx[t][c]
                    const t2_u32x2 ch = t2_sub4(DC.h, h), cl = t2_sub4(DC.l, h);
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb) {
                        f32x4 a = t2_mfma16(vf[sb][0], ch, z4);
                        a = t2_mfma16(vf[sb][0], cl, a);
                        ds[hs][sb] = t2_mfma16(vf[sb][1], ch, a);
                    }
                    // score gradients; the dropped probabilities (what multiplied V) for dV
                    float dot = 0.f;
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int s = 16 * sb + 4 * kq + r;
                            const bool live = s < p.S;
                            float d = ds[hs][sb][r];
                            if (p.dwavg && live) d += 0.25f * p.dwavg[row * p.S + s];
                            const bool keep = !(p.p_attn > 0.f) || ((kb_attn >> (8 * h + 4 * sb + r)) & 1u);
                            const float pr = P[h][sb][r];
                            pd[hs][sb][r] = (live && keep) ? pr * sc_attn : 0.f;
                            d = (live && keep) ? d * sc_attn : 0.f;
                            ds[hs][sb][r] = d;
                            dot = fmaf(pr, d, dot);
                        }
                    dot = t2_allkq(dot);
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) ds[hs][sb][r] = P[h][sb][r] * (ds[hs][sb][r] - dot);
                    // dq^T[d][t] = sum_s K[s][d] dS[t][s], times the 1/4 of the query scaling
                    unsigned w0, w1, w2, w3, l0, l1, l2, l3;
                    t2_split2(ds[hs][0][0], ds[hs][0][1], w0, l0); t2_split2(ds[hs][0][2], ds[hs][0][3], w1, l1);
                    t2_split2(ds[hs][1][0], ds[hs][1][1], w2, l2); t2_split2(ds[hs][1][2], ds[hs][1][3], w3, l3);
                    const t2_u32x4 sh = (t2_u32x4){w0, w1, w2, w3}, sl = (t2_u32x4){l0, l1, l2, l3};
                    f32x4 a = t2_mfma32(kt0, sh, z4);
                    a = t2_mfma32(kt0, sl, a);
                    dq[h] = t2_mfma32(kt1, sh, a) * 0.25f;
                }
                // images of the pair: rows 32 hs + key (= blocks 2 hs + sb)
#pragma unroll
                for (int which = 0; which < 2; ++which) {    // 0: Pd -> dV with dctx; 1: dS -> dK with Q
                    const f32x4 v4[4] = {which ? ds[0][0] : pd[0][0], which ? ds[0][1] : pd[0][1], which ? ds[1][0] : pd[1][0], which ? ds[1][1] : pd[1][1]};
                    T2Pack PP;
                    t2_pack16(v4, PP);
                    wave_lds_sync();
                    tb_write(PP.h, PP.l);
                    wave_lds_sync();
#pragma unroll
                    for (int hs = 0; hs < 2; ++hs)
#pragma unroll
                        for (int sb = 0; sb < 2; ++sb) {
                            const int h = 2 * hp + hs;
                            const t2_u32x2 ah = tb_frag(2 * hs + sb, 0), al = tb_frag(2 * hs + sb, 1);
                            const t2_u32x2 bh = which ? qth[h] : dth[h], bl = which ? qtl[h] : dtl[h];
                            f32x4 a = which ? accK[h][sb] : accV[h][sb];
                            a = t2_mfma16(ah, bh, a);
                            a = t2_mfma16(ah, bl, a);
                            a = t2_mfma16(al, bh, a);
                            if (which) accK[h][sb] = a; else accV[h][sb] = a;
                        }
                }
            }
            wave_lds_sync();
            T2Pack DQ;
            t2_pack16(dq, DQ);
            {
                f32x4 xq[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) xq[nb] = t2_ld4(xrow + 16 * nb) + t2_ld4(qrow + 16 * nb);
                T2Pack XQ;
                t2_pack16(xq, XQ);
                wgrad(dWm[0], 0, DQ, XQ);                    // dWq += dq^T (x + qpos), dbq
            }
            f32x4 dxq[4];
            gemm16(4, DQ, dxq);                              // dq Wq
            if (fin_on && p.fin_both) {                      // gradient reaching x through LN_F(x) (last decoder layer)
                f32x4 dfr[4], xh[4];
                float rs;
                {
                    f32x4 xv[4];
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        xv[nb] = t2_ld4(xrow + 16 * nb);
                        dfr[nb] = t2_ld4(p.dfin + row * 64 + 4 * kq + 16 * nb) * p.fin_scale;
                    }
                    ln_fwd(xv, xh, rs);
                }
                {
                    f32x4 e[4];
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) e[nb] = dfr[nb] * xh[nb];
                    dgF += tok_sum(e);
                }
                dbF += tok_sum(dfr);
                f32x4 o[4];
                ln_bwd(dfr, xh, 8, rs, o);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) dxr[nb] += o[nb];
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                t2_st4(p.dx + row * 64 + 4 * kq + 16 * nb, dxr[nb] + dxq[nb]);
                if (p.dqpos) {
                    f32x4 o = dxq[nb];
                    if (p.dqacc) o += t2_ld4(p.dqacc + row * 64 + 4 * kq + 16 * nb);
                    t2_st4(p.dqpos + row * 64 + 4 * kq + 16 * nb, o);
                }
            }
        }
    }
    if (seg == 0 && split < nrd) {                           // the work-group changes sample: this wave's dK / dV of sample b0 leave now
        kv_store(kv_wg + wave * T2_KVREC);
#pragma unroll
        for (int h = 0; h < 4; ++h) { accK[h][0] = z4; accK[h][1] = z4; accV[h][0] = z4; accV[h][1] = z4; }
    }
    }

    // ==================================== the work-group's records =========================================================================
    // (the weight images are dead from here on: LDS is reused from offset 0 once every wave has left the tile loop -- the
    // barrier below)
    float* const R = reinterpret_cast<float*>(t2_smem);      // four regions of 4096 floats, one per wave
    float* const Rv = R + 4 * 4096;                          // [wave][6][64] LayerNorm-gradient partials
    float* const rec = p.ppart + (long)blockIdx.x * T2_PREC;
    __syncthreads();                                         // every wave has finished its last product on the weight images
    // weight gradients: this wave's rows of the four matrices; bias gradients: lanes am == m hold rows 16 wave + 4 kq + r of matrix m
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) rec[m * 4096 + (16 * wave + 4 * kq + r) * 64 + 16 * kb + am] = dWm[m][kb][r];
    if (am < 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) rec[4 * 4096 + am * 64 + 16 * wave + 4 * kq + r] = accB[r];
    }
    Rv[wave * 384 + 0 * 64 + lane] = dgA; Rv[wave * 384 + 1 * 64 + lane] = dbA;
    Rv[wave * 384 + 2 * 64 + lane] = dgB; Rv[wave * 384 + 3 * 64 + lane] = dbB;
    Rv[wave * 384 + 4 * 64 + lane] = dgF; Rv[wave * 384 + 5 * 64 + lane] = dbF;
    kv_store(R + wave * 4096);
    __syncthreads();
    for (int i = tid; i < 384; i += T2_NT) rec[4 * 4096 + 4 * 64 + i] = (Rv[i] + Rv[384 + i]) + (Rv[2 * 384 + i] + Rv[3 * 384 + i]);
    // dK / dV of the last segment: the four waves' accumulators, added in wave order
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = 4 * (tid + T2_NT * q);
        t2_st4(kv_wg + 4 * T2_KVREC + i, (t2_ld4(R + i) + t2_ld4(R + 4096 + i)) + (t2_ld4(R + 2 * 4096 + i) + t2_ld4(R + 3 * 4096 + i)));
    }
    // which sample the records belong to: [0] records 0-3 (-1: not written), [1] record 4
    if (tid == 0) { p.kvflags[blockIdx.x * 2] = split < nrd ? b0 : -1; p.kvflags[blockIdx.x * 2 + 1] = split < nrd ? b0 + 1 : b0; }
}


// ---- forward of the layer in the same organisation, EXACT fp32 (training mode: it leaves the FFN's relu bits for the backward above) -------
// A wave owns 16 tokens and chains Q projection -> scores -> softmax -> PV -> out projection -> LN_A -> FFN -> LN_B (-> final norm) in
// registers: v_mfma_f32_16x16x4_f32 takes ONE f32 per lane as its B operand, k = lane group kq, so step c of a product consumes the lane's
// own value (block c >> 2, element c & 3) when the contraction index is enumerated as channel 16 (c >> 2) + 4 kq + (c & 3) -- the weight
// images are packed in that order.  Nothing is shared between waves: no barrier after the prologue.  Eight waves per work-group (two per
// SIMD); the four fp32 weight images (64 KB) are the only LDS.  A work-group takes T consecutive tiles, wave w the tiles w, w + 8, ...: with
// T = 12 (B = 48) every SIMD hosts one wave with two tiles and one with one.
// Why not split-bf16 like the backward: it was built first (22.6 us per decoder layer, the same as this one -- the kernel is bound by its
// vector-ALU epilogues, not by the matrix pipe) and agrees with the exact forward to 1e-5, but a forward that differs from the reference's
// fp32 arithmetic by 1e-5 decides a handful of relus (pre-activation within 1e-5 of zero: ~4 of 49,152 tokens per layer) the other way, and
// the gradients of those tokens then differ by O(1) from the reference's.  Products in fp32 keep the forward inside fp32 round-off of the
// reference; the backward takes the relu decisions from here (hmask), so ITS 2^-16 products cannot flip one.
struct T2F {
    const float* x; const float* qpos; long qbs;
    const float* wimg32; const float* kvf32;
    const float* bv[4]; const float* lnw[3]; const float* lnb[3];
    float fin_scale; int fin_both;
    int B, L, S, tps, ntiles, T;
    float p_attn, p_res, p_ffn; const unsigned long long* seed; unsigned site0; float eps;
    float* xout; float* fin; float* wavg; unsigned long long* hmask;
};
#define T2F_NT 512
#define T2F_WIMG_FLOATS (4 * 4096)               // 4 images x [4 nb][4 c4][64 lanes][4 floats]
#define T2F_KVF_FLOATS 4096                      // per sample: KF1f [4 h][2 sb][64][4], VTFf [4 h][2 c4][64][4]
#define T2F_LDS_BYTES (T2F_WIMG_FLOATS * 4 + 10 * 64 * 4)

__global__ __launch_bounds__(T2F_NT) void tplayer2_fwd_kernel(T2F p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char t2_smem[];
    const f32x4* const Wl = reinterpret_cast<const f32x4*>(t2_smem);
    float* const Vec = reinterpret_cast<float*>(t2_smem + T2F_WIMG_FLOATS * 4);
    const int tid = threadIdx.x, wave = tid >> 6;
    int lane = tid & 63;
    int am = lane & 15, kq = lane >> 4;
    const bool fin_on = p.lnw[2] != nullptr;
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.wimg32);
        f32x4* dst = reinterpret_cast<f32x4*>(t2_smem);
#pragma unroll 8
        for (int i = tid; i < T2F_WIMG_FLOATS / 4; i += T2F_NT) dst[i] = src[i];
        for (int i = tid; i < 10 * 64; i += T2F_NT) {
            const int v = i >> 6, c = i & 63;
            const float* s = v < 4 ? p.bv[v] : (((v - 4) & 1) ? p.lnb[(v - 4) >> 1] : p.lnw[(v - 4) >> 1]);
            Vec[i] = s ? s[c] : 0.f;
        }
    }
    __syncthreads();
    auto vec4 = [&](int v, int nb) __attribute__((always_inline)) -> f32x4 { return t2_ld4(Vec + v * 64 + 16 * nb + 4 * kq); };
    // D^T = W X^T, exact fp32: acc[nb] rows = output channels 16 nb .., columns = the wave's 16 tokens; X: the lane's 16 values of its token
    auto gemm16 = [&](int im, const f32x4 (&X)[4], f32x4 (&acc)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] = z4;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            f32x4 a[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) a[nb] = Wl[((im * 4 + nb) * 4 + c4) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nb][u], X[c4][u], acc[nb], 0, 0, 0);
        }
    };
    auto ln_fwd = [&](const f32x4 (&v)[4], f32x4 (&xh)[4], float& rstd) __attribute__((always_inline)) {
        const float mean = t2_allkq((t2_sum4(v[0]) + t2_sum4(v[1])) + (t2_sum4(v[2]) + t2_sum4(v[3]))) * (1.f / 64.f);
        float q = 0.f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) { xh[nb] = v[nb] - mean; q += t2_sum4(xh[nb] * xh[nb]); }
        rstd = __builtin_amdgcn_rsqf(t2_allkq(q) * (1.f / 64.f) + p.eps);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) xh[nb] = xh[nb] * rstd;
    };
    const bool any_drop = p.p_attn > 0.f || p.p_res > 0.f || p.p_ffn > 0.f;
    const uint64_t sd = any_drop ? p.seed[0] : 0ull;
    const uint32_t th_attn = dropout_thresh(p.p_attn), th_res = dropout_thresh(p.p_res), th_ffn = dropout_thresh(p.p_ffn);
    const float sc_attn = p.p_attn > 0.f ? 1.f / (1.f - p.p_attn) : 1.f;
    const float sc_res = p.p_res > 0.f ? 1.f / (1.f - p.p_res) : 1.f;
    const float sc_ffn = p.p_ffn > 0.f ? 1.f / (1.f - p.p_ffn) : 1.f;
    auto key0 = [&](unsigned site) __attribute__((always_inline)) -> uint32_t { return (uint32_t)sd ^ (site * 0x9E3779B9u); };
    auto key1 = [&](unsigned site) __attribute__((always_inline)) -> uint32_t { return (uint32_t)(sd >> 32) + site * 0x85EBCA77u; };
    const uint32_t ka0 = key0(p.site0), ka1 = key1(p.site0);
    const uint32_t kr0 = key0(p.site0 + 1), kr1 = key1(p.site0 + 1);
    const uint32_t kf0 = key0(p.site0 + 2), kf1 = key1(p.site0 + 2);
    const uint32_t ks0 = key0(p.site0 + 3), ks1 = key1(p.site0 + 3);

    const int wg0 = blockIdx.x * p.T, wg1 = min(p.ntiles, wg0 + p.T);
    for (int tile = wg0 + wave; tile < wg1; tile += 8) {
        asm volatile("" : "+v"(lane));                       // (see tplayer2_bwd_kernel: keeps lane-dependent addresses out of loop-invariant registers)
        am = lane & 15; kq = lane >> 4;
        const int b = tile / p.tps, tok0 = (tile - b * p.tps) * 16;
        const long row = (long)b * p.L + tok0 + am;
        const float* const xrow = p.x + row * 64 + 4 * kq;
        const float* const qrow = p.qpos + (long)b * p.qbs + (long)(tok0 + am) * 64 + 4 * kq;
        const f32x4* const kvf = reinterpret_cast<const f32x4*>(p.kvf32 + (long)b * T2F_KVF_FLOATS);
        const uint32_t ridx = (uint32_t)(row * 64) + 4 * kq;
        const uint32_t aidx = (uint32_t)(((long)b * 4 * p.L + tok0 + am) * p.S) + 4 * kq;
        const uint32_t hstep = (uint32_t)p.L * (uint32_t)p.S;
        f32x4 acc[4], Q[4];
        {
            f32x4 xq[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) xq[nb] = t2_ld4(xrow + 16 * nb) + t2_ld4(qrow + 16 * nb);
            gemm16(0, xq, acc);
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) Q[nb] = (acc[nb] + vec4(0, nb)) * 0.25f;
        f32x4 ctx[4], wsum[2] = {z4, z4};                    // wsum: head sum of the dropped probabilities (the layer's attention-weight output)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const f32x4 kf0v = kvf[(h * 2 + 0) * 64 + lane], kf1v = kvf[(h * 2 + 1) * 64 + lane];        // K[16 sb + am][16 h + 4 kq + j]
            const f32x4 vt0 = kvf[512 + (h * 2 + 0) * 64 + lane], vt1 = kvf[512 + (h * 2 + 1) * 64 + lane];   // V[16 c4 + 4 kq + u][16 h + am]
            f32x4 sc[2] = {z4, z4};
#pragma unroll
            for (int j = 0; j < 4; ++j) {                    // head dims 4 kq + j
                sc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf0v[j], Q[h][j], sc[0], 0, 0, 0);
                sc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf1v[j], Q[h][j], sc[1], 0, 0, 0);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool live = 16 * sb + 4 * kq + r < p.S;
                    sc[sb][r] = live ? sc[sb][r] : -INFINITY;
                    mx = fmaxf(mx, sc[sb][r]);
                }
            mx = t2_maxkq(mx);
            float sum = 0.f;
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = (16 * sb + 4 * kq + r < p.S) ? __expf(sc[sb][r] - mx) : 0.f;
                    sc[sb][r] = e; sum += e;
                }
            const float inv = __builtin_amdgcn_rcpf(t2_allkq(sum));
            f32x4 pd[2];
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float d = sc[sb][r] * inv;
                    if (p.p_attn > 0.f) d = t2_keep(ka0, ka1, aidx + (uint32_t)h * hstep + 16 * sb + r, th_attn) ? d * sc_attn : 0.f;
                    pd[sb][r] = d;
                }
            wsum[0] += pd[0]; wsum[1] += pd[1];
            f32x4 a0 = z4, a1 = z4;                          // two accumulator chains: keys 0-15 and 16-31
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vt0[u], pd[0][u], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vt1[u], pd[1][u], a1, 0, 0, 0);
            }
            ctx[h] = a0 + a1;
        }
        if (p.wavg) {
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int s = 16 * sb + 4 * kq + r;
                    if (s < p.S) p.wavg[row * p.S + s] = 0.25f * wsum[sb][r];
                }
        }
        f32x4 xh1[4];
        float rstd1;
        {
            gemm16(1, ctx, acc);
            f32x4 y1[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                f32x4 a = acc[nb] + vec4(1, nb);
                if (p.p_res > 0.f) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] = t2_keep(kr0, kr1, ridx + 16 * nb + r, th_res) ? a[r] * sc_res : 0.f;
                }
                y1[nb] = t2_ld4(xrow + 16 * nb) + a;
            }
            ln_fwd(y1, xh1, rstd1);
        }
        f32x4 x1[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) x1[nb] = xh1[nb] * vec4(4, nb) + vec4(5, nb);
        gemm16(2, x1, acc);
        f32x4 hd[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const f32x4 bj = vec4(2, nb);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float h = fmaxf(acc[nb][r] + bj[r], 0.f);
                if (p.p_ffn > 0.f) h = t2_keep(kf0, kf1, ridx + 16 * nb + r, th_ffn) ? h * sc_ffn : 0.f;
                hd[nb][r] = h;
                if (p.hmask) {                               // word [tile][nb][r]: bit `lane` = relu active and kept for (token am, channel 16 nb + 4 kq + r)
                    const unsigned long long m = __ballot(h > 0.f);
                    if (lane == 0) p.hmask[(long)tile * 16 + 4 * nb + r] = m;
                }
            }
        }
        gemm16(3, hd, acc);
        f32x4 xh2[4];
        float rstd2;
        {
            f32x4 y2[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                f32x4 f = acc[nb] + vec4(3, nb);
                if (p.p_res > 0.f) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) f[r] = t2_keep(ks0, ks1, ridx + 16 * nb + r, th_res) ? f[r] * sc_res : 0.f;
                }
                y2[nb] = x1[nb] + f;
            }
            ln_fwd(y2, xh2, rstd2);
        }
        f32x4 x2[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            x2[nb] = xh2[nb] * vec4(6, nb) + vec4(7, nb);
            if (p.xout) t2_st4(p.xout + row * 64 + 4 * kq + 16 * nb, x2[nb]);
        }
        if (fin_on && p.fin) {
            f32x4 xf[4], o[4];
            float rf;
            ln_fwd(x2, xf, rf);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) o[nb] = xf[nb] * vec4(8, nb) + vec4(9, nb);
            if (p.fin_both) {
                f32x4 xv[4], xh[4];
                float rs;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) xv[nb] = t2_ld4(xrow + 16 * nb);
                ln_fwd(xv, xh, rs);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) o[nb] += xh[nb] * vec4(8, nb) + vec4(9, nb);
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) t2_st4(p.fin + row * 64 + 4 * kq + 16 * nb, o[nb] * p.fin_scale);
        }
    }
}

// ---- packed operands: weight images in k-slot order, K / V as MFMA fragments -------------------------------------------------------------
// k-slot order of the 16x16x32 products: slot j of k-step ks in lane group kq <-> index 16 (2 ks + (j >> 2)) + 4 kq + (j & 3)
__device__ __forceinline__ int t2_kslot(int ks, int kq, int j) { return 16 * (2 * ks + (j >> 2)) + 4 * kq + (j & 3); }
struct T2Prep { const float* W[4]; const float* K; const float* V; unsigned* wimg; unsigned* kvf; float* wimg32; float* kvf32; int B, S; };
__global__ __launch_bounds__(256) void tplayer2_prep_kernel(T2Prep p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long nb16 = T2_WIMG_WORDS + (long)p.B * T2_KVF_WORDS;
    if (idx >= nb16) {
        // ---- the forward's exact-fp32 operands (tplayer2_fwd_kernel): one float per lane and MFMA step ----------------------------------
        long j = idx - nb16;
        if (j < T2F_WIMG_FLOATS) {
            // [im][nb][c4][lane][u]: W[16 nb + am][16 c4 + 4 kq + u]   (step c = 4 c4 + u of v_mfma_f32_16x16x4_f32, k = kq)
            const int u = j & 3, lane = (j >> 2) & 63, c4 = (j >> 8) & 3, nb = (j >> 10) & 3, im = (int)(j >> 12);
            p.wimg32[j] = p.W[im][(16 * nb + (lane & 15)) * 64 + 16 * c4 + 4 * (lane >> 4) + u];
            return;
        }
        j -= T2F_WIMG_FLOATS;
        if (j >= (long)p.B * T2F_KVF_FLOATS) return;
        const int b = (int)(j / T2F_KVF_FLOATS), w = (int)(j % T2F_KVF_FLOATS);
        const int u = w & 3, lane = (w >> 2) & 63, i2 = (w >> 8) & 1, h = (w >> 9) & 3, form = w >> 11;
        const int am = lane & 15, kq = lane >> 4;
        int s_, c_;
        if (form == 0) { s_ = 16 * i2 + am; c_ = 16 * h + 4 * kq + u; }               // KF1f [h][sb][lane][j]: K[16 sb + am][16 h + 4 kq + j]
        else { s_ = 16 * i2 + 4 * kq + u; c_ = 16 * h + am; }                            // VTFf [h][c4][lane][u]: V[16 c4 + 4 kq + u][16 h + am]
        const float* src = (form == 0 ? p.K : p.V) + (long)b * p.S * 64;
        p.kvf32[j] = s_ < p.S ? src[s_ * 64 + c_] : 0.f;
        return;
    }
    float v0, v1;
    unsigned* dst;
    int hl;
    if (idx < T2_WIMG_WORDS) {
        // word [im][blk][ks][hl][lane][e]: fwd image (im < 4): W[16 blk + am][kslot]; transposed image: W[kslot][16 blk + am]
        const int e = idx & 3, lane = (idx >> 2) & 63, ks = (idx >> 9) & 1, blk = (idx >> 10) & 3, im = (int)(idx >> 12);
        hl = (idx >> 8) & 1;
        const int am = lane & 15, kq = lane >> 4;
        const float* W = p.W[im & 3];
        const int k0 = t2_kslot(ks, kq, 2 * e), k1 = t2_kslot(ks, kq, 2 * e + 1), n = 16 * blk + am;
        if (im < 4) { v0 = W[n * 64 + k0]; v1 = W[n * 64 + k1]; }
        else { v0 = W[k0 * 64 + n]; v1 = W[k1 * 64 + n]; }
        dst = p.wimg + idx;
    } else {
        const long j = idx - T2_WIMG_WORDS;
        const int b = (int)(j / T2_KVF_WORDS), w = (int)(j % T2_KVF_WORDS);
        const int form = w >> 11, u = w & 2047;              // 0: KF1, 1: VF1, 2: VTF, 3: KTF
        const float* src = ((form == 0 || form == 3) ? p.K : p.V) + (long)b * p.S * 64;
        int s0, s1, c0, c1;
        if (form < 2) {
            // [h][sb][hl][lane][e2]: row key 16 sb + am, k-slots: head dims 4 kq + 2 e2 .. (16x16x16: 4 slots per lane)
            const int e2 = u & 1, lane = (u >> 1) & 63, sb = (u >> 8) & 1, h = u >> 9;
            hl = (u >> 7) & 1;
            const int am = lane & 15, kq = lane >> 4;
            s0 = s1 = 16 * sb + am;
            c0 = 16 * h + 4 * kq + 2 * e2; c1 = c0 + 1;
        } else {
            // [h][hl][lane][e]: row channel 16 h + am, k-slots: keys 16 (j >> 2) + 4 kq + (j & 3), j = 2 e, 2 e + 1
            const int e = u & 3, lane = (u >> 2) & 63, h = u >> 9;
            hl = (u >> 8) & 1;
            const int am = lane & 15, kq = lane >> 4;
            c0 = c1 = 16 * h + am;
            s0 = 16 * ((2 * e) >> 2) + 4 * kq + ((2 * e) & 3); s1 = s0 + 1;
        }
        v0 = s0 < p.S ? src[s0 * 64 + c0] : 0.f;
        v1 = s1 < p.S ? src[s1 * 64 + c1] : 0.f;
        dst = p.kvf + j;
    }
    unsigned hi, lo;
    t2_split2(v0, v1, hi, lo);
    *dst = hl ? lo : hi;
}

// ---- the same packing with the key / value projections folded in: ONE launch for all (one or two) decoder layers over a memory --------------
// Replaces, per training forward: kin = mem + pos (1 launch), K = kin Wk^T + bk and V = mem Wv^T + bv per layer (4 launches), the packing
// above per layer (2 launches).  A work-group per (layer, sample) computes the sample's 26 x 64 keys and values in fp32 (64 k-ordered FMAs
// per element: the arithmetic of the GEMM kernel it replaces, another summation order) into LDS and writes the fragment forms from there --
// K and V themselves never reach memory; 192 more work-groups per layer pack the weight images.
struct T2KVPrep {
    const float* mem; const float* pos; long pos_bs;         // (B,S,64); pos (B,S,64) [pos_bs = S*64] or (S,64) [0]
    const float* in_w[2]; const float* in_b[2]; const float* W[2][4];
    unsigned* wimg[2]; unsigned* kvf[2]; float* wimg32[2]; float* kvf32[2];
    float* kin;                                              // (B,S,64) = mem + pos, for the backward's weight gradients (nullable)
    int B, S, nl;
};
#define T2KV_WBLOCKS ((T2_WIMG_WORDS + T2F_WIMG_FLOATS) / 256)
__global__ __launch_bounds__(256) void tplayer2_kvprep_kernel(T2KVPrep p) {
    __shared__ float Min[32][64], Kin[32][64], Kc[32][64], Vc[32][64];
    const int per = p.B + T2KV_WBLOCKS;
    const int l = blockIdx.x / per, j = blockIdx.x - l * per, tid = threadIdx.x;
    if (j >= p.B) {                                          // ---- weight images of layer l (as tplayer2_prep_kernel) ----
        const long idx = (long)(j - p.B) * 256 + tid;
        if (idx < T2_WIMG_WORDS) {
            const int e = idx & 3, lane = (idx >> 2) & 63, hl = (idx >> 8) & 1, ks = (idx >> 9) & 1, blk = (idx >> 10) & 3, im = (int)(idx >> 12);
            const int am = lane & 15, kq = lane >> 4;
            const float* W = p.W[l][im & 3];
            const int k0 = t2_kslot(ks, kq, 2 * e), k1 = t2_kslot(ks, kq, 2 * e + 1), n = 16 * blk + am;
            unsigned hi, lo;
            if (im < 4) t2_split2(W[n * 64 + k0], W[n * 64 + k1], hi, lo);
            else t2_split2(W[k0 * 64 + n], W[k1 * 64 + n], hi, lo);
            p.wimg[l][idx] = hl ? lo : hi;
        } else {
            const long q = idx - T2_WIMG_WORDS;
            const int u = q & 3, lane = (q >> 2) & 63, c4 = (q >> 8) & 3, nb = (q >> 10) & 3, im = (int)(q >> 12);
            p.wimg32[l][q] = p.W[l][im][(16 * nb + (lane & 15)) * 64 + 16 * c4 + 4 * (lane >> 4) + u];
        }
        return;
    }
    // ---- keys and values of sample j for layer l ----
    const int b = j;
    for (int i = tid; i < 32 * 64; i += 256) {
        const int s_ = i >> 6, c = i & 63;
        float m = 0.f, k = 0.f;
        if (s_ < p.S) {
            m = p.mem[((long)b * p.S + s_) * 64 + c];
            k = m + p.pos[(long)b * p.pos_bs + s_ * 64 + c];
            if (l == 0 && p.kin) p.kin[((long)b * p.S + s_) * 64 + c] = k;
        }
        Min[s_][c] = m; Kin[s_][c] = k;
    }
    __syncthreads();
    {
        const int c = tid & 63, sg = tid >> 6;               // output channel c, rows sg, sg + 4, ...
        const float* wk = p.in_w[l] + (64 + c) * 64;
        const float* wv = p.in_w[l] + (128 + c) * 64;
        const float bk = p.in_b[l][64 + c], bv = p.in_b[l][128 + c];
        f32x4 w[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) w[q] = t2_ld4(wk + 4 * q);
        for (int s_ = sg; s_ < 32; s_ += 4) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(&Kin[s_][4 * q]);
                a = fmaf(x[0], w[q][0], a); a = fmaf(x[1], w[q][1], a); a = fmaf(x[2], w[q][2], a); a = fmaf(x[3], w[q][3], a);
            }
            Kc[s_][c] = s_ < p.S ? a + bk : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) w[q] = t2_ld4(wv + 4 * q);
        for (int s_ = sg; s_ < 32; s_ += 4) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(&Min[s_][4 * q]);
                a = fmaf(x[0], w[q][0], a); a = fmaf(x[1], w[q][1], a); a = fmaf(x[2], w[q][2], a); a = fmaf(x[3], w[q][3], a);
            }
            Vc[s_][c] = s_ < p.S ? a + bv : 0.f;
        }
    }
    __syncthreads();
    unsigned* const kvf = p.kvf[l] + (long)b * T2_KVF_WORDS;
    for (int w_ = tid; w_ < T2_KVF_WORDS; w_ += 256) {       // the backward's four bf16 forms (layout: tplayer2_prep_kernel)
        const int form = w_ >> 11, u = w_ & 2047;
        int s0, s1, c0, c1, hl;
        if (form < 2) {
            const int e2 = u & 1, lane = (u >> 1) & 63, sb = (u >> 8) & 1, h = u >> 9;
            hl = (u >> 7) & 1;
            s0 = s1 = 16 * sb + (lane & 15);
            c0 = 16 * h + 4 * (lane >> 4) + 2 * e2; c1 = c0 + 1;
        } else {
            const int e = u & 3, lane = (u >> 2) & 63, h = u >> 9;
            hl = (u >> 8) & 1;
            c0 = c1 = 16 * h + (lane & 15);
            s0 = 16 * ((2 * e) >> 2) + 4 * (lane >> 4) + ((2 * e) & 3); s1 = s0 + 1;
        }
        const bool useK = form == 0 || form == 3;
        unsigned hi, lo;
        t2_split2(useK ? Kc[s0][c0] : Vc[s0][c0], useK ? Kc[s1][c1] : Vc[s1][c1], hi, lo);
        kvf[w_] = hl ? lo : hi;
    }
    float* const kvf32 = p.kvf32[l] + (long)b * T2F_KVF_FLOATS;
    for (int w_ = tid; w_ < T2F_KVF_FLOATS; w_ += 256) {     // the forward's two fp32 forms
        const int u = w_ & 3, lane = (w_ >> 2) & 63, i2 = (w_ >> 8) & 1, h = (w_ >> 9) & 3, form = w_ >> 11;
        const int am = lane & 15, kq = lane >> 4;
        kvf32[w_] = form == 0 ? Kc[16 * i2 + am][16 * h + 4 * kq + u] : Vc[16 * i2 + 4 * kq + u][16 * h + am];
    }
}
// mem (B,S,64); pos (B,S,64) [pos_bs = S*64] or (S,64) [pos_bs = 0]; per layer l < nl (nl = 1 or 2): in_w (192,64), in_b (192), out_w, w1, w2
// (64,64) and the four destinations of tatt_tplayer2_prep; kin (B,S,64, nullable) receives mem + pos.  Pointer arrays are HOST arrays.
TATT_API int tatt_tplayer2_kvprep(const float* mem, const float* pos, long pos_bs, const float* const* in_w, const float* const* in_b,
                                  const float* const* out_w, const float* const* w1, const float* const* w2, unsigned* const* wimg,
                                  unsigned* const* kvf, float* const* wimg32, float* const* kvf32, float* kin, int B, int S, int nl,
                                  hipStream_t st) {
    if (B < 1 || S < 1 || S > 32 || nl < 1 || nl > 2) return 1;
    T2KVPrep p = {};
    p.mem = mem; p.pos = pos; p.pos_bs = pos_bs; p.kin = kin; p.B = B; p.S = S; p.nl = nl;
    for (int l = 0; l < nl; ++l) {
        p.in_w[l] = in_w[l]; p.in_b[l] = in_b[l];
        p.W[l][0] = in_w[l]; p.W[l][1] = out_w[l]; p.W[l][2] = w1[l]; p.W[l][3] = w2[l];
        p.wimg[l] = wimg[l]; p.kvf[l] = kvf[l]; p.wimg32[l] = wimg32[l]; p.kvf32[l] = kvf32[l];
    }
    hipLaunchKernelGGL(tplayer2_kvprep_kernel, dim3(nl * (B + T2KV_WBLOCKS)), dim3(256), 0, st, p);
    return LAUNCH_CHECK();
}

// ---- geometry --------------------------------------------------------------------------------------------------------------------------------
struct T2Geom { int tps, ntiles, G, nper; };
static inline T2Geom t2_geom(int B, int L) {
    T2Geom g;
    g.tps = L / 16;
    g.ntiles = B * g.tps;
    const int G0 = cdiv(g.ntiles, 4) < 256 ? cdiv(g.ntiles, 4) : 256;
    g.nper = cdiv(g.ntiles, 4 * G0);
    g.G = cdiv(g.ntiles, 4 * g.nper);
    return g;
}
// 1 when the second-generation backward takes the geometry: whole rounds of four 16-token tiles inside one sample, a work-group's
// tiles inside two samples at most, flat dropout indices below 2^32
static inline int t2_takes(int B, int L, int S) {
    if (B < 1 || L < 64 || L % 64 || S < 1 || S > 32) return 0;
    const T2Geom g = t2_geom(B, L);
    if (4 * g.nper > g.tps) return 0;
    if ((double)B * 4.0 * L * S >= 4294967296.0 || (double)B * L * 64.0 >= 4294967296.0) return 0;
    return 1;
}
// out[0] = 1 if the geometry is taken, out[1] = work-groups, out[2] = floats of kvpart, out[3] = floats of ppart, out[4] = ints of
// kvflags, out[5] = words of wimg, out[6] = words of kvf, out[7] = floats of wimg32, out[8] = floats of kvf32
TATT_API int tatt_tplayer2_geom(int B, int L, int S, int* out) {
    out[0] = t2_takes(B, L, S);
    const T2Geom g = out[0] ? t2_geom(B, L) : T2Geom{0, 0, 0, 0};
    out[1] = g.G; out[2] = g.G * 5 * T2_KVREC; out[3] = g.G * T2_PREC; out[4] = g.G * 2;
    out[5] = T2_WIMG_WORDS; out[6] = B * T2_KVF_WORDS; out[7] = T2F_WIMG_FLOATS; out[8] = B * T2F_KVF_FLOATS;
    return 0;
}

// in_w: the packed in-projection (192, 64) -- its first 64 rows are the query projection
TATT_API int tatt_tplayer2_prep(const float* in_w, const float* out_w, const float* w1, const float* w2, const float* K, const float* V,
                                unsigned* wimg, unsigned* kvf, float* wimg32, float* kvf32, int B, int S, hipStream_t st) {
    if (B < 1 || S < 1 || S > 32) return 1;
    T2Prep p = {{in_w, out_w, w1, w2}, K, V, wimg, kvf, wimg32, kvf32, B, S};
    const long n = T2_WIMG_WORDS + (long)B * T2_KVF_WORDS + T2F_WIMG_FLOATS + (long)B * T2F_KVF_FLOATS;
    hipLaunchKernelGGL(tplayer2_prep_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, p);
    return LAUNCH_CHECK();
}

TATT_API int tatt_tplayer2_bwd(const float* x, const float* qpos, long qbs, const unsigned* wimg, const unsigned* kvf, const float* in_b,
                               const float* out_b, const float* b1, const float* b2, const float* lnA_w, const float* lnA_b,
                               const float* lnB_w, const float* lnB_b, const float* lnF_w, const float* lnF_b, float fin_scale,
                               int fin_both, const float* dxout, const float* dfin, const float* dwavg, const float* dqacc, float* dx,
                               float* dqpos, float* kvpart, float* ppart, int* kvflags, const unsigned long long* hmask, int B, int L, int S,
                               float p_attn, float p_res, float p_ffn, const unsigned long long* seed, unsigned site0, float eps,
                               hipStream_t st) {
    if (!t2_takes(B, L, S)) return 1;
    if ((p_attn > 0.f || p_res > 0.f || p_ffn > 0.f) && !seed) return 2;
    if (lnF_w && !dfin) return 3;
    T2P p = {};
    p.x = x; p.qpos = qpos; p.qbs = qbs; p.wimg = wimg; p.kvf = kvf;
    p.bv[0] = in_b; p.bv[1] = out_b; p.bv[2] = b1; p.bv[3] = b2;
    p.lnw[0] = lnA_w; p.lnw[1] = lnB_w; p.lnw[2] = lnF_w;
    p.lnb[0] = lnA_b; p.lnb[1] = lnB_b; p.lnb[2] = lnF_b;
    p.fin_scale = fin_scale; p.fin_both = fin_both;
    const T2Geom g = t2_geom(B, L);
    p.B = B; p.L = L; p.S = S; p.tps = g.tps; p.ntiles = g.ntiles; p.nper = g.nper;
    p.p_attn = p_attn; p.p_res = p_res; p.p_ffn = p_ffn; p.seed = seed; p.site0 = site0; p.eps = eps;
    p.dxout = dxout; p.dfin = dfin; p.dwavg = dwavg; p.dqacc = dqacc;
    p.dx = dx; p.dqpos = dqpos; p.kvpart = kvpart; p.ppart = ppart; p.kvflags = kvflags; p.hmask = hmask;
    static TattPerDevice attr;
    tatt_per_device(attr, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tplayer2_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS_BYTES);
    });
    hipLaunchKernelGGL(tplayer2_bwd_kernel, dim3(g.G), dim3(T2_NT), T2_LDS_BYTES, st, p);
    return LAUNCH_CHECK();
}


// Forward of the layer (training), exact fp32: arguments as tatt_tplayer_fwd with the matrices / K / V replaced by the fp32 operand images of
// tatt_tplayer2_prep (wimg32, kvf32); hmask (B L 64-bit words, nullable) receives the relu-and-kept bits of the FFN's hidden layer
TATT_API int tatt_tplayer2_fwd(const float* x, const float* qpos, long qbs, const float* wimg32, const float* kvf32, const float* in_b,
                               const float* out_b, const float* b1, const float* b2, const float* lnA_w, const float* lnA_b,
                               const float* lnB_w, const float* lnB_b, const float* lnF_w, const float* lnF_b, float fin_scale,
                               int fin_both, float* xout, float* fin, float* wavg, unsigned long long* hmask, int B, int L, int S,
                               float p_attn, float p_res, float p_ffn, const unsigned long long* seed, unsigned site0, float eps,
                               hipStream_t st) {
    if (!t2_takes(B, L, S)) return 1;
    if ((p_attn > 0.f || p_res > 0.f || p_ffn > 0.f) && !seed) return 2;
    T2F p = {};
    p.x = x; p.qpos = qpos; p.qbs = qbs; p.wimg32 = wimg32; p.kvf32 = kvf32;
    p.bv[0] = in_b; p.bv[1] = out_b; p.bv[2] = b1; p.bv[3] = b2;
    p.lnw[0] = lnA_w; p.lnw[1] = lnB_w; p.lnw[2] = lnF_w;
    p.lnb[0] = lnA_b; p.lnb[1] = lnB_b; p.lnb[2] = lnF_b;
    p.fin_scale = fin_scale; p.fin_both = fin_both;
    const int tps = L / 16, ntiles = B * tps;
    const int G0 = cdiv(ntiles, 8) < 256 ? cdiv(ntiles, 8) : 256;
    const int T = cdiv(ntiles, G0), G = cdiv(ntiles, T);
    p.B = B; p.L = L; p.S = S; p.tps = tps; p.ntiles = ntiles; p.T = T;
    p.p_attn = p_attn; p.p_res = p_res; p.p_ffn = p_ffn; p.seed = seed; p.site0 = site0; p.eps = eps;
    p.xout = xout; p.fin = fin; p.wavg = wavg; p.hmask = hmask;
    static TattPerDevice attr;
    tatt_per_device(attr, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tplayer2_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, T2F_LDS_BYTES);
    });
    hipLaunchKernelGGL(tplayer2_fwd_kernel, dim3(G), dim3(T2F_NT), T2F_LDS_BYTES, st, p);
    return LAUNCH_CHECK();
}

// ---- dK / dV: sum the records of the work-groups that walked sample b, in work-group order ----------------------------------------------------
__global__ __launch_bounds__(256) void tplayer2_reduce_kv_kernel(const float* __restrict__ part, const int* __restrict__ flags,
                                                                 float* __restrict__ dK, float* __restrict__ dV, int B, int S, T2Geom g) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int n = S * 64;
    if (idx >= (long)B * n) return;
    const int b = (int)(idx / n), i = (int)(idx % n);
    const int per = 4 * g.nper;
    const int w_lo = (b * g.tps) / per, w_hi = min(g.G - 1, ((b + 1) * g.tps - 1) / per);
    float sk = 0.f, sv = 0.f;
    for (int w = w_lo; w <= w_hi; ++w) {
        const float* rec = part + (long)w * 5 * T2_KVREC;
        if (flags[w * 2] == b) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { sk += rec[q * T2_KVREC + i]; sv += rec[q * T2_KVREC + 2048 + i]; }
        }
        if (flags[w * 2 + 1] == b) { sk += rec[4 * T2_KVREC + i]; sv += rec[4 * T2_KVREC + 2048 + i]; }
    }
    dK[idx] = sk; dV[idx] = sv;
}
TATT_API int tatt_tplayer2_reduce_kv(const float* kvpart, const int* kvflags, float* dK, float* dV, int B, int L, int S, hipStream_t st) {
    if (!t2_takes(B, L, S)) return 1;
    const T2Geom g = t2_geom(B, L);
    hipLaunchKernelGGL(tplayer2_reduce_kv_kernel, dim3(cdiv((long)B * S * 64, 256)), dim3(256), 0, st, kvpart, kvflags, dK, dV, B, S, g);
    return LAUNCH_CHECK();
}
