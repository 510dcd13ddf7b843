#!/usr/bin/env python3
"""CPU emulation of csrc/conv3w.hip's data movement (fragment gather, MFMA operand semantics, tile ownership, partial layout) for one
work-group on a tiny problem, against a direct weight gradient -- checks the index algebra, not the HIP code generation."""
import numpy as np

rng = np.random.default_rng(0)
B, H, W, Cin, Cout = 2, 3, 64, 64, 64
x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
dy = rng.standard_normal((B, H, W, Cout)).astype(np.float32)

# direct: dW[tap][ci][co] = sum x[b, h+ky-1, w+kx-1, ci] * dy[b, h, w, co]
ref = np.zeros((9, Cin, Cout))
xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0))).astype(np.float64)
for ky in range(3):
    for kx in range(3):
        ref[ky * 3 + kx] = np.einsum("bhwi,bhwo->io", xp[:, ky:ky + H, kx:kx + W], dy.astype(np.float64))
refdb = dy.astype(np.float64).sum((0, 1, 2))


def mfma(A, Bm):
    """v_mfma_f32_16x16x32: lane (i = l & 15, kq = l >> 4) holds A[i][8 kq + e] and B[8 kq + e][j = l & 15]; C[4 (l >> 4) + r][l & 15]."""
    Af = np.zeros((16, 32)); Bf = np.zeros((32, 16))
    for l in range(64):
        i, kq = l & 15, l >> 4
        Af[i, 8 * kq:8 * kq + 8] = A[l]
        Bf[8 * kq:8 * kq + 8, i] = Bm[l]
    return Af @ Bf            # [row][col]; lane l reads rows 4 (l >> 4) + r, col l & 15


CP, HW = 66, 66
acc = np.zeros((8, 9, 2, 16, 16))          # [wave][tap][c] tile (row = ci within tile, col = co within tile)
accdb = np.zeros((8, 2, 16, 16))
for s in range(B * H):                     # one work-group walks all segments (W = 64: one segment per row)
    n, h = divmod(s, H)
    IMG = np.zeros(3 * HW * CP + 64 * CP)
    for rr in range(3):
        for px in range(HW):
            hh, ww = h + rr - 1, px - 1
            if 0 <= hh < H and 0 <= ww < W:
                IMG[(rr * HW + px) * CP:(rr * HW + px) * CP + 64] = x[n, hh, ww]
    HALO = 3 * HW * CP
    for px in range(64):
        IMG[HALO + px * CP:HALO + px * CP + 64] = dy[n, h, px]
    F = {}
    for ky in range(3):
        for wave in range(8):
            xt, ks = (wave >> 1) & 3, wave & 1
            for kx in range(3):
                frag = np.zeros((64, 8))
                for l in range(64):
                    li, kq = l & 15, l >> 4
                    base = (ky * HW + 32 * ks + 8 * kq) * CP + 16 * xt + li
                    v = [IMG[base + e * CP] for e in range(10)]
                    frag[l] = v[kx:kx + 8]
                F[(kx * 4 + xt) * 2 + ks] = frag
            if ky == 0:
                frag = np.zeros((64, 8))
                for l in range(64):
                    li, kq = l & 15, l >> 4
                    base = HALO + (32 * ks + 8 * kq) * CP + 16 * (wave >> 1) + li
                    frag[l] = [IMG[base + e * CP] for e in range(8)]
                F[24 + (wave >> 1) * 2 + ks] = frag
        for wave in range(8):
            ct, cg = wave & 3, wave >> 2
            for kx in range(3):
                for ks in range(2):
                    A = F[(kx * 4 + ct) * 2 + ks]
                    for c in range(2):
                        Bm = F[24 + (2 * cg + c) * 2 + ks]
                        acc[wave, ky * 3 + kx, c] += mfma(A, Bm)
            if ky == 0 and ct == 0:
                for c in range(2):
                    for ks in range(2):
                        accdb[wave, c] += mfma(np.ones((64, 8)), F[24 + (2 * cg + c) * 2 + ks])

P = np.zeros((9 * Cin, Cout)); pdb = np.zeros(Cout)
for wave in range(8):
    ct, cg = wave & 3, wave >> 2
    for tap in range(9):
        for c in range(2):
            for l in range(64):
                li, kq = l & 15, l >> 4
                for r in range(4):
                    P[tap * Cin + 16 * ct + 4 * kq + r, 16 * (2 * cg + c) + li] = acc[wave, tap, c][4 * kq + r, li]
    if ct == 0:
        for c in range(2):
            for li in range(16):
                pdb[16 * (2 * cg + c) + li] = accdb[wave, c][0, li]
err = np.abs(P.reshape(9, Cin, Cout) - ref).max() / np.abs(ref).max()
errb = np.abs(pdb - refdb).max() / np.abs(refdb).max()
print("dW rel err %.2e, db rel err %.2e" % (err, errb))
assert err < 1e-12 and errb < 1e-12
