"""Host-side geometry of the fused TP-interpreter layer (csrc/tplayer.hip): runs without a GPU (the library only has to load)."""


def test_fused_layer_raw_abi_geometry():
    """tatt_tplayer_geom: records cover every (work-group, sample) pair for a spread of batch / length combinations."""
    from tatt_amd import ops
    for B, L in [(1, 1), (48, 1024), (16, 4096), (48, 26), (300, 26), (7, 33), (2, 100000)]:
        G, span, nkv, npp = ops.tplayer_geom(B, L)
        tps = (L + 31) // 32
        ntiles = B * tps
        nper = (ntiles + G - 1) // G
        assert 1 <= G <= 256 and G * nper >= ntiles and (G - 1) * nper < ntiles
        for w in range(G):
            b0, b1 = (w * nper) // tps, (min(ntiles, (w + 1) * nper) - 1) // tps
            assert b1 - b0 + 1 <= span, (B, L, w, b0, b1, span)
        assert nkv == G * span * 2 * 32 * 64 and npp == G * (4 * 4096 + 640)
