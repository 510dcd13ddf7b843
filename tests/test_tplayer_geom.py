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


def test_second_generation_geometry_and_fallback_rule():
    """tatt_tplayer2_geom (csrc/tplayer2.hip): the second generation takes whole rounds of four 16-token tiles inside one sample
    (L % 64 == 0), a work-group's tiles inside two samples at most, at most 32 keys; every tile is covered exactly once by
    (work-group, round, wave); everything else is left to the first generation."""
    from tatt_amd import ops
    for B, L, S, want in [(48, 1024, 26, 1), (16, 4096, 26, 1), (2, 1024, 26, 1), (2, 64, 26, 1), (3, 128, 5, 1), (48, 26, 26, 0),
                          (3, 80, 26, 0), (1, 32, 26, 0), (2, 1024, 33, 0), (300, 64, 26, 0), (64, 64, 26, 1)]:
        g = ops.tplayer2_geom(B, L, S)
        assert g[0] == want, (B, L, S, g)
        if not want:
            continue
        G, nkv, npp, nfl = g[1:5]
        tps, ntiles = L // 16, B * (L // 16)
        nper = -(-ntiles // (4 * G))
        assert 1 <= G <= 256 and 4 * nper <= tps and 4 * nper * G >= ntiles > 4 * nper * (G - 1)
        assert nkv == G * 5 * 4096 and npp == G * (4 * 4096 + 640) and nfl == 2 * G
        seen = set()
        for w in range(G):
            wg0 = w * 4 * nper
            nrd = min(nper, (ntiles - wg0) // 4)
            b0 = wg0 // tps
            split = min(nrd, ((b0 + 1) * tps - wg0) // 4)
            assert nrd >= 1 and split >= 1
            for rd in range(nrd):
                b = b0 + (0 if rd < split else 1)
                for wave in range(4):
                    t = wg0 + 4 * rd + wave
                    assert t // tps == b, (B, L, w, rd, wave)        # the four tiles of a round lie in ONE sample
                    seen.add(t)
        assert seen == set(range(ntiles))
        assert g[5] == 8 * 4096 and g[6] == B * 8192 and g[7] == 4 * 4096 and g[8] == B * 4096


def test_contraction_split_rules_of_the_round5_kernels():
    """Host-side split selection (no GPU): tatt_qgru_wgrad_sb must give every split at least one 32-token chunk for any token count;
    the generic convolution kernel's split rule gives the STN head's deep layers the partial-map counts the folded BatchNorm launches
    were measured with (functional.STN_FOLD_MAX: up to 9 are folded, the 18-map ones keep their own reduction launch)."""
    import torch
    from tatt_amd import ops
    for chunks in list(range(1, 40)) + [96, 192, 97, 1000]:
        S = ops.qgru_wgrad_splits(32 * chunks)
        per = -(-chunks // S)
        assert 1 <= S <= ops.QGRU_WGRAD_SPLIT and (S - 1) * per < chunks <= S * per, (chunks, S)
    B = 48

    def split(H, W, Cin, Cout):
        return ops.conv_split(torch.empty(B, H, W, Cin), Cout, 3, 3)
    # forward convolutions of the head at B = 48 (model/stn_head.py:33-49): 4->32 @16x64, 32->64 @8x32, 64->128 @4x16, 128->256 @2x8, 256->256 @1x4, @1x2
    assert [split(16, 64, 4, 32), split(8, 32, 32, 64), split(4, 16, 64, 128), split(2, 8, 128, 256), split(1, 4, 256, 256),
            split(1, 2, 256, 256)] == [1, 1, 4, 9, 18, 18]
    # their data gradients, last layer first: 256->256 @1x2, @1x4, 256->128 @2x8, 128->64 @4x16, 64->32 @8x32
    assert [split(1, 2, 256, 256), split(1, 4, 256, 256), split(2, 8, 256, 128), split(4, 16, 128, 64), split(8, 32, 64, 32)] == [18, 18, 18, 9, 4]
