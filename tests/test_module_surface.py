"""CPU: host logic -- the drop-in nn.Module surface and the C ABI."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle.fixtures import summarize

STD = dict(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,cls,nkeys", [("kat", "TSRN_TL_TRANS", 304), ("kat_tsrn", "TSRN", 239)])
def test_state_dict_keys_and_default_init_match_reference(name, cls, nkeys):
    """Same keys, shapes and -- under torch.manual_seed(1234) -- the same initial values as the reference module
    (fingerprints captured from the reference import by tools/gen_golden.py)."""
    import tatt_amd
    z = np.load("tests/golden/%s.npz" % name)
    torch.manual_seed(1234)
    sd = getattr(tatt_amd, cls)(**STD).state_dict()
    assert list(sd.keys()) == z["sd_keys"].tolist()
    assert len(sd) == nkeys
    for k, ref in zip(sd, z["sd_summary"]):
        assert np.abs(summarize(sd[k].float()) - ref).max() < 1e-6, k


def test_tbsrn_state_dict_matches_reference():
    import tatt_amd
    z = np.load("tests/golden/kat_tbsrn.npz")
    torch.manual_seed(1234)
    m = tatt_amd.TBSRN(scale_factor=2, width=512, height=32, STN=True, mask=True, input_channel=4)
    sd = m.state_dict()
    assert list(sd.keys()) == z["sd_keys"].tolist() and len(sd) == 346
    assert sum(p.numel() for p in m.parameters()) == 3221019
    for k, ref in zip(sd, z["sd_summary"]):
        assert np.abs(summarize(sd[k].float()) - ref).max() < 1e-6, k
    with pytest.raises(RuntimeError, match="GPU"):
        m.eval()(torch.rand(1, 4, 16, 256))


def test_param_count_and_shapes():
    import tatt_amd
    m = tatt_amd.TSRN_TL_TRANS(**STD)
    assert sum(p.numel() for p in m.parameters()) == 7608334
    sd = m.state_dict()
    assert tuple(sd["block1.0.weight"].shape) == (64, 4, 9, 9)
    assert tuple(sd["block2.gru1.conv1.weight"].shape) == (64, 128, 1, 1)
    assert tuple(sd["infoGen.transformer.gru_encoding.weight_ih_l0"].shape) == (1536, 1024)
    assert tuple(sd["block8.0.conv.weight"].shape) == (256, 64, 3, 3)
    assert tuple(sd["tps.target_coordinate_repr"].shape) == (1024, 23)
    big = tatt_amd.TSRN_TL_TRANS(scale_factor=2, width=256, height=64, STN=False)
    assert sum(p.numel() for p in big.parameters()) == 20111814


def test_load_state_dict_roundtrip_and_modes():
    import tatt_amd
    a, b = tatt_amd.TSRN_TL_TRANS(**STD), tatt_amd.TSRN_TL_TRANS(**STD)
    b.load_state_dict(a.state_dict(), strict=True)
    for (k1, v1), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    a.train(); assert a.training and a.block2.bn1.training
    a.eval(); assert not a.block2.bn1.training
    for p in a.parameters():
        p.requires_grad = False


def test_product_path_has_no_cpu_fallback():
    import tatt_amd
    m = tatt_amd.TSRN_TL_TRANS(**STD).eval()
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.rand(1, 4, 16, 64), torch.rand(1, 37, 1, 26))
    with pytest.raises(RuntimeError):
        tatt_amd.TSRN(**STD).eval()(torch.rand(1, 4, 16, 64))


def test_round6_operators_have_no_cpu_fallback():
    """The operators added in round 6 (TBSRN's composite sub-layers, the score-free attention, the token weight gradient) refuse CPU
    tensors like everything else on the path: nothing silently computes on the host."""
    import tatt_amd
    from tatt_amd import functional as Fh, ops
    from tatt_amd.tbsrn import MultiHeadedAttention, PositionwiseFeedForward
    x = torch.randn(1, 128, 128)
    ga, be = torch.ones(128), torch.zeros(128)
    pff, mh = PositionwiseFeedForward(128, 128), MultiHeadedAttention(4, 128)
    Fh.linear_prepack([pff.w_1, pff.w_2] + list(mh.linears))              # CPU weights: nothing is packed, the table stays empty
    assert not Fh._PKL.table
    with pytest.raises(RuntimeError, match="GPU"):
        Fh.feed_forward_ln(x, pff.w_1, pff.w_2, ga, be, 1e-6, 1, 0.1, True, 1)
    with pytest.raises(RuntimeError, match="GPU"):
        Fh.attention_ln(x, mh, ga, be, 1e-6, 1, 0.0, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Fh.SelfAttnFlashFn.apply(x, x, x, 4, 0.0, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Fh._tokgemm_ex(torch.randn(64, 128), torch.randn(128 * 128), None, 128, 128)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.tok_wgrad_sb(torch.randn(64, 128), torch.randn(64, 128), torch.empty(128, 128))
    with pytest.raises(RuntimeError):
        tatt_amd.TBSRN(scale_factor=2, width=128, height=32, STN=False, mask=True).eval()(torch.rand(1, 4, 16, 64))


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "tatt_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_c_abi_exports_every_declared_symbol():
    from tatt_amd._lib import LIB, LIB_PATH, parse_header
    protos = parse_header()
    assert len(protos) >= 37
    dll = ctypes.CDLL(LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), name
    LIB.load()
    # every TATT_API definition in the sources is declared in the header
    defined = set()
    for f in os.listdir(os.path.join(ROOT, "tatt_amd", "csrc")):
        if f.endswith(".hip"):
            defined |= set(re.findall(r"TATT_API\s+int\s+(\w+)", open(os.path.join(ROOT, "tatt_amd", "csrc", f)).read()))
    assert defined == set(protos), defined ^ set(protos)


def test_packed_filter_sizes_come_from_the_library():
    """tatt_repack_words (host-only) is the one definition of the packed-layout sizes: the Python allocation asks it, so a layout
    change in csrc/gemm.hip cannot leave a buffer short (the split-bf16 3x3 operands once shared the 9x9 Toeplitz size by accident)."""
    from tatt_amd._lib import LIB
    from tatt_amd.ops import _packed_numel
    for mode in (0, 1, 2, 3, 6, 7):
        assert LIB.tatt_repack_words(64, 64, 3, 3, mode) == 64 * 64 * 9
    assert LIB.tatt_repack_words(4, 64, 9, 9, 8) == LIB.tatt_repack_words(64, 4, 9, 9, 9) == 9 * 4 * 2 * 6 * 64 * 4
    assert LIB.tatt_repack_words(256, 64, 3, 3, 10) == LIB.tatt_repack_words(256, 64, 3, 3, 11) == 256 * 64 * 9
    assert LIB.tatt_repack_words(64, 4, 9, 9, 0) == 64 * 4 * 81
    assert LIB.tatt_repack_words(4, 64, 9, 9, 12) == LIB.tatt_repack_words(64, 4, 9, 9, 13) == 9 * 4 * 6 * 2 * 64 * 4
    assert LIB.tatt_repack_words(128, 64, 3, 3, 14) == LIB.tatt_repack_words(64, 128, 3, 3, 15) == 128 * 64 * 9      # generations 3 / 4
    for mode in (-1, 4, 5, 16):
        assert LIB.tatt_repack_words(64, 64, 3, 3, mode) == -1
        with pytest.raises(RuntimeError):
            _packed_numel((64, 64, 3, 3), mode)
    assert _packed_numel((256, 64, 3, 3), 10) == 256 * 64 * 9


def test_set_arithmetic_switches_every_split_kernel_family():
    """`tatt_amd.set_arithmetic` is the documented switch between the split-bf16 default and exact fp32 products (INTEGRATION.md
    "Arithmetic"): it must move all four kernel families together and refuse anything else."""
    import tatt_amd
    from tatt_amd import functional as Fh, ops
    assert tatt_amd.get_arithmetic() == "split_bf16"
    try:
        tatt_amd.set_arithmetic("fp32")
        assert not (ops.CONV3_SB or ops.CONV3_WGRAD_SB or Fh.TOKGEMM_SB or Fh.GRU_WGRAD_SB or Fh.QGRU_CHAIN_SB)
        assert tatt_amd.get_arithmetic() == "fp32"
        ops.CONV3_SB = True
        assert tatt_amd.get_arithmetic() == "mixed"
        with pytest.raises(ValueError):
            tatt_amd.set_arithmetic("bf16")
    finally:
        tatt_amd.set_arithmetic("split_bf16")
    assert ops.CONV3_SB and ops.CONV3_WGRAD_SB and Fh.TOKGEMM_SB and Fh.GRU_WGRAD_SB and Fh.QGRU_CHAIN_SB


def test_bench_cpu_baseline_leg_runs_without_a_gpu():
    """`bench.py --cpu-baseline-only` (the child process of the cpu_baseline leg): bounded thread count, JSON contract."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-only", "--arch", "tsrn", "--cpu-batch", "2"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert set(d) == {"value", "unit", "cores", "kind", "sample"}
    assert d["kind"] == "port" and d["unit"] == "LR images/s" and d["value"] > 0 and 1 <= d["cores"] <= 32


def test_sync_kernels_fit_beside_each_other():
    """The persistent query-GRU launches occupy EVERY CU (256 work-groups of 8 waves) while the STN head's launches -- which also wait
    for their own work-groups in flight -- run on the other lane.  A kernel of one family whose waves do not fit into the registers a
    resident work-group of the other leaves on a SIMD (512 VGPRs per lane, allocated in blocks of 8) would have to wait for it: the lane
    stalls for the length of the recurrence, and two half-resident launches could wait for each other until their wall-clock bound
    expires.  Checked at build time from hipcc's resource report: 2 waves/SIMD of a chain kernel + the waves/SIMD of each STN kernel."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def vgprs(src):
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
                              "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(root, "tatt_amd", "csrc", src), "-o", "/dev/null"],
                             capture_output=True, text=True).stderr
        res, name = {}, None
        for line in out.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r" VGPRs: (\d+)", line)
            if m and name:
                res[name] = max(res.get(name, 0), int(m.group(1)))
        return res
    up8 = lambda v: (v + 7) // 8 * 8
    gru, stn = vgprs("gru.hip"), vgprs("stnhead.hip")
    chains = {k: v for k, v in gru.items() if "chain_kernel" in k}
    assert len(chains) == 4 and len(stn) >= 8, (chains, stn)
    for k, v in stn.items():
        direction = "fwd" if "fwd" in k else "bwd"                  # the head's forward runs beside the forward recurrence, ...
        chain = max(2 * up8(c) for n, c in chains.items() if direction + "_chain" in n)      # 512 threads = 2 waves per SIMD
        per_simd = (2 if "_fc_" in k else 1) * up8(v)               # the fc launches have 512 threads, the map launches 256
        assert chain + per_simd <= 512, (k, v, chains)


def test_fused_path_predicates_choose_the_fallbacks():
    """Host logic of the launches that only take the published geometry: the STN head's fused launches (B <= 64, 16-row input,
    BatchNorms in training mode with a momentum) and the persistent query-GRU launches (hidden 512, rows % 16 == 0, <= 256 tiles);
    everything else must be routed to the operator-by-operator / per-step paths."""
    import tatt_amd.tsrn as T
    from tatt_amd import functional as Fh
    stn = T.STNHead(4, 20, "none").train()
    x = torch.zeros(3, 16, 64, 4)                                  # NHWC-indexed view shape
    assert Fh.stn_head_fusable(x, stn, 3) and Fh.stn_head_fusable(x, stn, 64)
    assert not Fh.stn_head_fusable(x, stn, 65)                     # the fully connected launch holds <= 64 samples
    assert not Fh.stn_head_fusable(torch.zeros(3, 32, 128, 4), stn, 3)
    stn.stn_convnet[4][1].momentum = None                          # cumulative-average BatchNorm: not handled by the fused launch
    assert not Fh.stn_head_fusable(x, stn, 3)
    stn.stn_convnet[4][1].momentum = 0.1
    stn.eval()
    assert not Fh.stn_head_fusable(x, stn, 3)
    assert Fh._qgru_chain_takes(64, 512)
    assert not Fh._qgru_chain_takes(128, 1024) and not Fh._qgru_chain_takes(60, 512) and not Fh._qgru_chain_takes(144, 512)


def test_residency_gate_routes_a_partitioned_device_to_the_fallbacks(monkeypatch):
    """The launches that synchronise their work-groups in flight need their whole grid resident (256 work-groups for the query-GRU chains
    at W = 64, up to 128 / 32 for the STN head).  With the occupancy query reporting less -- a partitioned or CU-masked GPU -- the
    predicates must choose the per-step / operator-chain paths (nothing on the host checked this)."""
    import tatt_amd.tsrn as T
    from tatt_amd import functional as Fh

    class _X:                                                       # a "device tensor" for the predicate: shape + is_cuda + device
        shape, is_cuda, device = (3, 16, 64, 4), True, torch.device("cpu")
    stn = T.STNHead(4, 20, "none").train()
    monkeypatch.setattr(Fh, "_has_gpu", lambda: True)
    monkeypatch.setattr(Fh, "sync_capacity", lambda device=None: (256, 256, 256, 256, 512, 512))     # a whole MI355X
    assert Fh._qgru_chain_takes(64, 512) and Fh._qgru_chain_takes(64, 512, True) and Fh.stn_head_fusable(_X, stn, 3)
    monkeypatch.setattr(Fh, "sync_capacity", lambda device=None: (128, 128, 128, 128, 256, 256))     # half the CUs
    assert not Fh._qgru_chain_takes(64, 512) and not Fh._qgru_chain_takes(64, 512, True)
    assert Fh._qgru_chain_takes(32, 512)                           # 128 tiles still fit
    assert Fh.stn_head_fusable(_X, stn, 3)
    monkeypatch.setattr(Fh, "sync_capacity", lambda device=None: (256, 256, 256, 128, 96, 24))       # the backward fp32 chain and the STN launches do not fit
    assert Fh._qgru_chain_takes(64, 512) and Fh._qgru_chain_takes(64, 512, True)                    # (split-bf16 is the default form)
    monkeypatch.setattr(Fh, "QGRU_CHAIN_SB", False)
    assert Fh._qgru_chain_takes(64, 512) and not Fh._qgru_chain_takes(64, 512, True)
    assert not Fh.stn_head_fusable(_X, stn, 3)
