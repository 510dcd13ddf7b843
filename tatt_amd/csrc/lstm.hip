// Bidirectional LSTM recurrences of the CRNN text-prior generator (gfx950): reference model/crnn/crnn.py:5-26
// (BidirectionalLSTM = nn.LSTM(nIn, 256, bidirectional=True) + nn.Linear), sequence length 26, batch = images, hidden 256.
//
// The input projection gi = x W_ih^T + b_ih of all time steps is one GEMM done beforehand (tatt_gemm); these kernels run the
// latency-bound recurrence, ONE LAUNCH PER TIME STEP FOR BOTH DIRECTIONS (blockIdx.z): h W_hh^T on v_mfma_f32_16x16x4_f32
// with K split over the 4 waves of a work-group, the cell update fused in the epilogue.  Same organisation as the
// query-GRU step kernels of gru.hip.  Layout: time-major (T, Bt, *) row-major; direction d handles time t = s (d = 0) or
// T-1-s (d = 1) at step s; `out` (T, Bt, 2H) = [h_fwd | h_rev] is both the result and the h_{t-1} operand of the next step.
// Gate order (i, f, g, o) as in torch.nn.LSTM.
#include "common.h"

struct LstmFwdP {
    const float* gi;            // (T, Bt, 8H): [dir][gate][H] along the last axis, b_ih included
    const float* whh[2];        // (4H, H)
    const float* bhh[2];        // (4H)
    float* out;                 // (T, Bt, 2H)
    float* cseq;                // (2, T, Bt, H)
    float* gsave;               // (2, T, Bt, 4, H): activated gates i, f, g, o
    int T, Bt, H, s;
};
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(LstmFwdP p) {
    __shared__ float red[4][4][16][17];
    const int d = blockIdx.z;
    const int m0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int H = p.H, T = p.T, Bt = p.Bt;
    const int tt = d ? T - 1 - p.s : p.s, tp = d ? tt + 1 : tt - 1;
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.s > 0) {
        const float* hprev = p.out + (long)tp * Bt * 2 * H + d * H;          // row pitch 2H
        const float* whh = p.whh[d];
        const int i = lane & 15, q = lane >> 4;
        const int kspan = H / 4, kbeg = wave * kspan;
        const int arow = min(m0 + i, Bt - 1);
        for (int kb = kbeg; kb < kbeg + kspan; kb += 64) {
            f32x4 a[4], b[4][4];
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
                a[ss] = *reinterpret_cast<const f32x4*>(hprev + (long)arow * 2 * H + kb + 16 * ss + 4 * q);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    b[g][ss] = *reinterpret_cast<const f32x4*>(whh + ((long)g * H + j0 + i) * H + kb + 16 * ss + 4 * q);
            }
#pragma unroll
            for (int ss = 0; ss < 4; ++ss)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ss][u], b[g][ss][u], acc[g], 0, 0, 0);
        }
    }
    {
        const int col = lane & 15, rb = (lane >> 4) * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][g][rb + r][col] = acc[g][r];
    }
    __syncthreads();
    const int m = t >> 4, j = t & 15;
    if (m0 + m >= Bt) return;
    const long row = m0 + m;
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        pre[g] = (red[0][g][m][j] + red[1][g][m][j]) + (red[2][g][m][j] + red[3][g][m][j]) + p.bhh[d][g * H + j0 + j] +
                 p.gi[((long)tt * Bt + row) * 8 * H + (d * 4 + g) * H + j0 + j];
    const float ig = sigmoid_f(pre[0]), fg = sigmoid_f(pre[1]), gg = tanhf(pre[2]), og = sigmoid_f(pre[3]);
    const long plane = (long)T * Bt * H;
    const long e = ((long)tt * Bt + row) * H + j0 + j;
    const float cp = p.s > 0 ? p.cseq[d * plane + ((long)tp * Bt + row) * H + j0 + j] : 0.f;
    const float c = fg * cp + ig * gg;
    p.cseq[d * plane + e] = c;
    p.out[((long)tt * Bt + row) * 2 * H + d * H + j0 + j] = og * tanhf(c);
    float* gs = p.gsave + (d * plane + ((long)tt * Bt + row) * H) * 4 + j0 + j;       // [row][4][H]
    gs[0] = ig; gs[H] = fg; gs[2 * H] = gg; gs[3 * H] = og;
}
TATT_API int tatt_lstm_fwd_step(const float* gi, const float* whh_f, const float* whh_r, const float* bhh_f, const float* bhh_r,
                                float* out, float* cseq, float* gsave, int T, int Bt, int H, int s, hipStream_t st) {
    if (H % 256) return 1;          // each of the 4 waves reduces H/4 columns in trips of 64
    LstmFwdP p = {gi, {whh_f, whh_r}, {bhh_f, bhh_r}, out, cseq, gsave, T, Bt, H, s};
    hipLaunchKernelGGL(lstm_fwd_step_kernel, dim3(cdiv(Bt, 16), H / 16, 2), dim3(256), 0, st, p);
    return LAUNCH_CHECK();
}

// Backward step s (s = T-1 .. 0), both directions: on a (16 rows x 16 hidden units) tile
//   dh = dout[t] + dgates[t_next] @ W_hh     (t_next = the time processed by the previous backward step; absent for s = T-1)
//   dc = dc_carry + dh o (1 - tanh(c)^2);  dgates[t] = [dc g i(1-i), dc c_prev f(1-f), dc i (1-g^2), dh tanh(c) o(1-o)];  dc_carry = dc f
// whhT = W_hh^T (H, 4H) so that the B operand is read along the contraction axis.
struct LstmBwdP {
    const float* dout;          // (T, Bt, 2H)
    const float* whhT[2];       // (H, 4H)
    const float* cseq;          // (2, T, Bt, H)
    const float* gsave;         // (2, T, Bt, 4, H)
    float* dgates;              // (2, T, Bt, 4H)
    float* dccarry;             // (2, Bt, H)
    int T, Bt, H, s;
};
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(LstmBwdP p) {
    __shared__ float red[4][16][17];
    const int d = blockIdx.z;
    const int m0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int H = p.H, T = p.T, Bt = p.Bt, K = 4 * p.H;
    const int tt = d ? T - 1 - p.s : p.s;                 // time of this step
    const int tn = d ? tt - 1 : tt + 1;                   // time of forward step s+1 (processed just before in this backward sweep)
    const int tp = d ? tt + 1 : tt - 1;                   // time of forward step s-1 (c_prev)
    const long plane = (long)T * Bt * H;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.s < T - 1) {
        const float* dg = p.dgates + (d * plane + (long)tn * Bt * H) * 4;
        const float* whh = p.whhT[d];
        const int i = lane & 15, q = lane >> 4;
        const int kspan = K / 4, kbeg = wave * kspan;
        const int arow = min(m0 + i, Bt - 1);
        for (int kb = kbeg; kb < kbeg + kspan; kb += 64) {
            f32x4 a[4], b[4];
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
                a[ss] = *reinterpret_cast<const f32x4*>(dg + (long)arow * K + kb + 16 * ss + 4 * q);
                b[ss] = *reinterpret_cast<const f32x4*>(whh + (long)(j0 + i) * K + kb + 16 * ss + 4 * q);
            }
#pragma unroll
            for (int ss = 0; ss < 4; ++ss)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ss][u], b[ss][u], acc, 0, 0, 0);
        }
    }
    {
        const int col = lane & 15, rb = (lane >> 4) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][rb + r][col] = acc[r];
    }
    __syncthreads();
    const int m = t >> 4, j = t & 15;
    if (m0 + m >= Bt) return;
    const long row = m0 + m;
    const float dh = p.dout[((long)tt * Bt + row) * 2 * H + d * H + j0 + j] +
                     ((red[0][m][j] + red[1][m][j]) + (red[2][m][j] + red[3][m][j]));
    const long e = ((long)tt * Bt + row) * H + j0 + j;
    const float* gs = p.gsave + (d * plane + ((long)tt * Bt + row) * H) * 4 + j0 + j;
    const float ig = gs[0], fg = gs[H], gg = gs[2 * H], og = gs[3 * H];
    const float c = p.cseq[d * plane + e];
    const float cp = p.s > 0 ? p.cseq[d * plane + ((long)tp * Bt + row) * H + j0 + j] : 0.f;
    const float tc = tanhf(c);
    float* dcc = p.dccarry + ((long)d * Bt + row) * H + j0 + j;
    const float dc = (p.s < T - 1 ? *dcc : 0.f) + dh * og * (1.f - tc * tc);
    float* dg = p.dgates + (d * plane + ((long)tt * Bt + row) * H) * 4 + j0 + j;
    dg[0] = dc * gg * ig * (1.f - ig);
    dg[H] = dc * cp * fg * (1.f - fg);
    dg[2 * H] = dc * ig * (1.f - gg * gg);
    dg[3 * H] = dh * tc * og * (1.f - og);
    *dcc = dc * fg;
}
TATT_API int tatt_lstm_bwd_step(const float* dout, const float* whhT_f, const float* whhT_r, const float* cseq, const float* gsave,
                                float* dgates, float* dccarry, int T, int Bt, int H, int s, hipStream_t st) {
    if (H % 64) return 1;           // each wave reduces 4H/4 gate columns in trips of 64
    LstmBwdP p = {dout, {whhT_f, whhT_r}, cseq, gsave, dgates, dccarry, T, Bt, H, s};
    hipLaunchKernelGGL(lstm_bwd_step_kernel, dim3(cdiv(Bt, 16), H / 16, 2), dim3(256), 0, st, p);
    return LAUNCH_CHECK();
}

// ---- bicubic resize + luminance (reference interfaces/base.py:797-815: F.interpolate(img[:, :3], (32, W'), 'bicubic') then
// 0.299 R + 0.587 G + 0.114 B) -- ATen upsample_bicubic2d semantics: source = (o + 0.5) * in/out - 0.5, A = -0.75, border
// taps clamped.  img (B, C >= 3, H, W) by element strides; out (B, OH, OW) contiguous (= NHWC with one channel). ----
__device__ __forceinline__ void cubic_w(float t, float w[4]) {
    const float a = -0.75f;
    const float x0 = t + 1.f, x3 = 2.f - t, x1 = t, x2 = 1.f - t;
    w[0] = ((a * x0 - 5.f * a) * x0 + 8.f * a) * x0 - 4.f * a;
    w[1] = ((a + 2.f) * x1 - (a + 3.f)) * x1 * x1 + 1.f;
    w[2] = ((a + 2.f) * x2 - (a + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((a * x3 - 5.f * a) * x3 + 8.f * a) * x3 - 4.f * a;
}
__global__ void bicubic_luma_kernel(const float* __restrict__ img, long sn, long sc, long sh, long sw, float* __restrict__ out,
                                    int B, int H, int W, int OH, int OW) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * OH * OW) return;
    const int ow = idx % OW; long r = idx / OW;
    const int oh = r % OH; const int n = r / OH;
    const float sy = (oh + 0.5f) * ((float)H / OH) - 0.5f, sx = (ow + 0.5f) * ((float)W / OW) - 0.5f;
    const float fy = floorf(sy), fx = floorf(sx);
    float wy[4], wx[4];
    cubic_w(sy - fy, wy);
    cubic_w(sx - fx, wx);
    const float lum[3] = {0.299f, 0.587f, 0.114f};
    float acc = 0.f;
    for (int c = 0; c < 3; ++c) {
        const float* pl = img + n * sn + c * sc;
        float v = 0.f;
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max((int)fy - 1 + a, 0), H - 1);
            float rowv = 0.f;
            for (int b = 0; b < 4; ++b) {
                const int xx = min(max((int)fx - 1 + b, 0), W - 1);
                rowv += wx[b] * pl[yy * sh + xx * sw];
            }
            v += wy[a] * rowv;
        }
        acc += lum[c] * v;
    }
    out[idx] = acc;
}
TATT_API int tatt_bicubic_luma(const float* img, long sn, long sc, long sh, long sw, float* out, int B, int H, int W, int OH,
                               int OW, hipStream_t st) {
    const long total = (long)B * OH * OW;
    hipLaunchKernelGGL(bicubic_luma_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, img, sn, sc, sh, sw, out, B, H, W, OH, OW);
    return LAUNCH_CHECK();
}
