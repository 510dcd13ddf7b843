"""CPU emulation of the INDEX ALGEBRA between tatt_gru32_bwd2 (fragment emission, csrc/gru.hip) and tatt_gru_wgrad_frag
(fragment consumption, csrc/gruwgrad.hip): slots, octets, K-steps, lane order, the token of every x row, the partial-slab rows.
Arithmetic in float64 (the hi / lo split is not modelled: hi carries the value, lo is zero).  Exact match with the direct weight
gradients = the two kernels agree on the layout.  Run: python tools/emulate_gru_frag.py"""
import numpy as np


def seq_geom(B, H, W, vertical):
    return (B * W, H, W, H * W, 1, W) if vertical else (B * H, W, 1, W, 0, 1)


def emit(dgi, dgh, hprev, geom):
    """what the lanes of gru32_bwd2<FRAGS> store: frag[c][slot][hl][lane'][8 elements]"""
    nseq, T, s_in, shi, slo, st = geom
    nK = nseq * (T // 8) // 4
    frag = np.zeros((nK, 20, 2, 64, 8))
    for seq in range(nseq):
        base = (seq // s_in) * shi + (seq % s_in) * slo
        for d in range(2):
            for j in range(32):
                for s_hi in range(T - 1, -1, -8):                    # the main loop's groups of 8 steps, descending
                    stg = np.zeros((5, 8))
                    for step in range(s_hi, s_hi - 8, -1):
                        tok0 = base + ((T - 1) * st if d else 0)
                        tok = tok0 + step * (-st if d else st)
                        e = ((T - 1 - step) if d else step) & 7
                        stg[0, e] = dgi[tok, d * 96 + j]             # drp
                        stg[1, e] = dgi[tok, d * 96 + 32 + j]        # dzp
                        stg[2, e] = dgi[tok, d * 96 + 64 + j]        # dnp
                        stg[3, e] = dgh[tok, d * 96 + 64 + j]        # dghn
                        stg[4, e] = hprev[tok, d * 32 + j]
                    tt0 = (T - 1 - s_hi) if d else s_hi - 7
                    o = seq * (T >> 3) + (tt0 >> 3)
                    c, kq = o >> 2, o & 3
                    for q in range(5):
                        slot = d * 8 + q * 2 + (j >> 4) if q < 4 else 16 + d * 2 + (j >> 4)
                        frag[c, slot, 0, kq * 16 + (j & 15)] = stg[q]
    return frag


def mfma(A, Bm):
    """v_mfma_f32_16x16x32: A[lane][8] with lane = 16 kq + i -> A[i][8 kq + e]; B[lane][8] with lane = 16 kq + j -> B[8 kq + e][j]"""
    a = np.zeros((16, 32)); b = np.zeros((32, 16))
    for lane in range(64):
        i, kq = lane & 15, lane >> 4
        a[i, 8 * kq:8 * kq + 8] = A[lane]
        b[8 * kq:8 * kq + 8, i] = Bm[lane]
    return a @ b


def consume(frag, x, xb, geom, G):
    nseq, T, s_in, shi, slo, st = geom
    T8 = T // 8
    nK = frag.shape[0]
    K = 128 if xb is not None else 64
    NT = K // 16
    p1 = np.zeros((G, 192, K)); s1 = np.zeros((G, 192)); p2 = np.zeros((G, 192, 32)); s2 = np.zeros((G, 192))
    for g in range(G):
        for c in range(g, nK, G):
            img = np.zeros((32, K))
            for row in range(32):
                o = 4 * c + (row >> 3)
                s, w = o // T8, o % T8
                tok = (s // s_in) * shi + (s % s_in) * slo + (8 * w + (row & 7)) * st
                img[row, :64] = x[tok]
                if xb is not None:
                    img[row, 64:] = xb[tok]
            F = np.zeros((NT, 64, 8))
            for tile in range(NT):
                for lane in range(64):
                    li, kq = lane & 15, lane >> 4
                    F[tile, lane] = img[8 * kq:8 * kq + 8, 16 * tile + li]
            ones = np.ones((64, 8))
            for wave in range(6):
                d, gate = wave // 3, wave % 3
                a_slot = d * 8 + gate * 2
                ah_slot = d * 8 + 6 if gate == 2 else a_slot
                bh_slot = 16 + d * 2
                for m in range(2):
                    A = frag[c, a_slot + m, 0]; AH = frag[c, ah_slot + m, 0]
                    rows = slice(32 * wave + 16 * m, 32 * wave + 16 * m + 16)
                    s1[g, rows] += mfma(A, ones)[:, 0]
                    s2[g, rows] += mfma(AH, ones)[:, 0]
                    for n in range(NT):
                        p1[g, rows, 16 * n:16 * n + 16] += mfma(A, F[n])
                    for n in range(2):
                        p2[g, rows, 16 * n:16 * n + 16] += mfma(AH, frag[c, bh_slot + n, 0])
    return p1.sum(0), s1.sum(0), p2.sum(0), s2.sum(0)


def main():
    rng = np.random.default_rng(0)
    for (B, H, W, vertical, cat, G) in [(1, 16, 8, True, True, 3), (2, 8, 16, False, False, 2), (1, 8, 64, False, True, 5), (3, 16, 4, True, False, 4)]:
        geom = seq_geom(B, H, W, vertical)
        M = B * H * W
        dgi, dgh, hp = rng.standard_normal((M, 192)), rng.standard_normal((M, 192)), rng.standard_normal((M, 64))
        dgh[:, :64] = dgi[:, :64]; dgh[:, 96:160] = dgi[:, 96:160]           # the r / z gate gradients are shared
        x, xb = rng.standard_normal((M, 64)), (rng.standard_normal((M, 64)) if cat else None)
        frag = emit(dgi, dgh, hp, geom)
        dWp, dbp, dWhh, dbhh = consume(frag, x, xb, geom, G)
        xx = np.concatenate([x, xb], 1) if cat else x
        full = dgh.T @ hp                                                 # (192, 64): diagonal blocks wanted
        want_hh = np.concatenate([full[:96, :32], full[96:, 32:]], 0)
        errs = [np.abs(dWp - dgi.T @ xx).max(), np.abs(dbp - dgi.sum(0)).max(), np.abs(dWhh - want_hh).max(), np.abs(dbhh - dgh.sum(0)).max()]
        print((B, H, W, vertical, cat, G), ["%.1e" % e for e in errs])
        assert max(errs) < 1e-9


if __name__ == "__main__":
    main()
    print("fragment emission and consumption agree")
