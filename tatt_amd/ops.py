"""Raw (non-autograd) launchers of the libtatt_hip.so kernels on torch CUDA(HIP) tensors.

torch is used here only as device-memory / stream plumbing: every function marshals `data_ptr()`s,
sizes and strides into the C ABI declared in include/tatt_hip.h and launches on torch's current HIP
stream (so the calls are hipGraph-capturable).  There is no CPU fallback: a non-HIP tensor raises.
"""
from __future__ import annotations

import ctypes
import weakref
from typing import Optional

import os

import torch

from ._lib import LIB

ACT_NONE, ACT_RELU, ACT_MISH, ACT_TANH = 0, 1, 2, 3
_vp = ctypes.c_void_p


def _check_dev(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("tatt_amd kernels need tensors on an AMD GPU (HIP device); got %s. "
                           "There is no CPU fallback in the product path." % t.device)
    if t.dtype != torch.float32:
        raise RuntimeError("tatt_amd kernels are fp32; got %s" % t.dtype)


def P(t: Optional[torch.Tensor]):
    if t is None:
        return None
    return _vp(t.data_ptr())


def stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def call(name: str, *args):
    rc = getattr(LIB, name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (name, rc))


def cdiv(a, b):
    return (a + b - 1) // b


def new(ref: torch.Tensor, *shape, dtype=torch.float32):
    return torch.empty(shape, device=ref.device, dtype=dtype)


# --------------------------------------------------------------------------------------------------
# GEMM family
# --------------------------------------------------------------------------------------------------
_REDUCE_KEEP = {}          # stream handle -> split-K workspaces registered on that stream (present <=> the stream is deferring)


def _skey():
    return torch.cuda.current_stream().cuda_stream


def reduce_defer(on: bool):
    """Split-K reductions behind the GEMMs / convolution weight gradients issued ON THE CURRENT STREAM from now on are batched into
    one launch per 36 (tatt_reduce_defer); `reduce_defer(False)` (or `reduce_flush`) runs them.  Until then their outputs are
    undefined.  The state is per stream, here and in the library: lanes / trainers on other streams are not affected."""
    call("tatt_reduce_defer", int(bool(on)), stream())
    if on:
        _REDUCE_KEEP.setdefault(_skey(), [])
    else:
        _REDUCE_KEEP.pop(_skey(), None)


def reduce_flush():
    call("tatt_reduce_flush", stream())
    if _skey() in _REDUCE_KEEP:
        _REDUCE_KEEP[_skey()] = []


def _split_ws(t):
    """A split-K workspace: must outlive a deferred reduction."""
    keep = _REDUCE_KEEP.get(_skey())
    if keep is not None and t is not None:
        keep.append(t)
    return t


def gemm(A, sam, sak, B, sbk, sbn, C, scm, scn, M, N, K, *, A2=None, sa2m=0, sa2k=0, K1=0, bias=None,
         Z=1, bsA=0, bsA2=0, bsB=0, bsC=0, bsBias=0, alpha=1.0, beta=0.0, act=ACT_NONE, splitk=1, rowsum=None):
    """C = act(alpha*(A@B + bias)) + beta*C with explicit element strides (see tatt_gemm)."""
    _check_dev(C)
    ws = None
    if splitk > 1:
        nchunks = cdiv(K, 16)
        splitk = max(1, min(splitk, nchunks))
        if splitk > 1:
            ws = _split_ws(new(C, Z * splitk * M * N + (splitk * M if rowsum is not None else 0)))
    call("tatt_gemm", P(A), sam, sak, P(A2), sa2m, sa2k, K1, P(B), sbk, sbn, P(bias), P(C), scm, scn,
         M, N, K, Z, bsA, bsA2, bsB, bsC, bsBias, alpha, beta, act, splitk, P(ws), P(rowsum), stream())
    return C


def _auto_split(M_out, N_out, Kred, cap=128):
    tiles = cdiv(M_out, 64) * cdiv(N_out, 64)
    nchunks = cdiv(Kred, 16)
    want = max(1, min(cap, 512 // tiles))
    return max(1, min(want, nchunks // 8 if nchunks >= 16 else 1))


def linear_fwd(x2, W, b=None, *, act=ACT_NONE, alpha=1.0, x2b=None, out=None):
    """y (M,N) = act(alpha*(x2 @ W^T + b)).  x2 (M,K) row-major (last stride 1); W (N,K) contiguous.
    x2b: optional second source for the K-columns beyond x2.shape[1] (K-concatenation)."""
    _check_dev(x2)
    M, K1 = x2.shape
    N, K = W.shape
    y = out if out is not None else new(x2, M, N)
    assert K1 + (0 if x2b is None else x2b.shape[1]) == K
    if x2b is None:
        assert K1 == K
        gemm(x2, x2.stride(0), x2.stride(1), W, 1, K, y, y.stride(0), 1, M, N, K, bias=b, alpha=alpha, act=act)
    else:
        assert K1 + x2b.shape[1] == K
        gemm(x2, x2.stride(0), x2.stride(1), W, 1, K, y, y.stride(0), 1, M, N, K, A2=x2b, sa2m=x2b.stride(0),
             sa2k=x2b.stride(1), K1=K1, bias=b, alpha=alpha, act=act)
    return y


def linear_bwd_input(dy, W, *, col0=0, ncols=None, alpha=1.0, out=None, beta=0.0):
    """dx (M,ncols) = alpha * dy (M,N) @ W[:, col0:col0+ncols]   (W (N,K) contiguous)."""
    M, N = dy.shape
    K = W.shape[1]
    ncols = K if ncols is None else ncols
    dx = out if out is not None else new(dy, M, ncols)
    Wv = W.reshape(-1)[col0:] if col0 else W
    # few output tiles over a deep contraction (the query GRU's input gradient: 64 x 1024 outputs over 1536 gate columns = 16
    # work-groups walking 96 K-chunks, 61 us): split the contraction over the idle CUs
    tiles = cdiv(M, 64) * cdiv(ncols, 64)
    splitk = max(1, min(256 // tiles, N // 96)) if (tiles <= 32 and N >= 768) else 1
    gemm(dy, dy.stride(0), dy.stride(1), Wv, K, 1, dx, dx.stride(0), 1, M, ncols, N, alpha=alpha, beta=beta, splitk=splitk)
    return dx


def linear_bwd_input_halves(dy, W):
    """(dy @ W[:, :K/2], dy @ W[:, K/2:]) as two contiguous (M, K/2) tensors from ONE launch (a 2-batch GEMM that shares A): the
    input gradients of a linear layer over the concatenation of two equally wide sources (GruBlock over cat[x, tp_map])."""
    M, N = dy.shape
    K = W.shape[1]
    h = K // 2
    assert K == 2 * h
    buf = new(dy, 2, M, h)
    gemm(dy, dy.stride(0), dy.stride(1), W, K, 1, buf, h, 1, M, h, N, Z=2, bsA=0, bsB=h, bsC=M * h)
    return buf[0], buf[1]


def linear_bwd_weight(dy, x2, *, alpha=1.0, out=None, out_ld=None, beta=0.0, rowsum=None):
    """dW (N,K) = alpha * dy^T (N,M) @ x2 (M,K); reduction over the M tokens (split-K, deterministic).
    rowsum: optional (N,) tensor that receives alpha * dy.sum(0) -- the bias gradient -- from the same pass."""
    M, N = dy.shape
    K = x2.shape[1]
    dW = out if out is not None else new(dy, N, K)
    ld = out_ld if out_ld is not None else dW.stride(0)
    gemm(dy, dy.stride(1), dy.stride(0), x2, x2.stride(0), x2.stride(1), dW, ld, 1, N, K, M, alpha=alpha, beta=beta,
         splitk=_auto_split(N, K, M), rowsum=rowsum)
    return dW


GRU_WGRAD_GROUPS = 128        # persistent work-groups (= partial slabs) of tatt_gru_wgrad_sb
CONV3_WGRAD_SB = True          # test / A-B hook: False -> the fp32-MFMA weight-gradient kernel (conv3.hip)
CONV3_WGRAD_GROUPS = 128      # persistent work-groups of the 3x3 weight-gradient kernels (all 64x64 channel blocks together); round 4, split-bf16 kernel: 128 measured 0.15 ms per step better than 256 (half the partial slabs to write and reduce, half the CUs left to the main lane)


def gru_wgrad_fusable(dgi, dgh, x2, xb2, hprev):
    """The fused split-bf16 weight-gradient pass of a GruBlock takes contiguous (M, 192) gate gradients and (M, 64) inputs."""
    M = dgi.shape[0]
    ok = lambda t, n: t.dim() == 2 and t.shape[0] == M and t.shape[1] == n and t.is_contiguous()
    return (M % 32 == 0 and M >= 32 and ok(dgi, 192) and ok(dgh, 192) and ok(x2, 64) and ok(hprev, 64)
            and (xb2 is None or ok(xb2, 64)))


def gru_wgrad_sb(dgi, dgh, x2, xb2, hprev, dWp, dWhh, dbp, dbhh):
    """dWp (192, K) = dgi^T [x2 | xb2], dbp = dgi.sum(0), dWhh (192, 64) = dgh^T hprev, dbhh = dgh.sum(0): one pass over the tokens
    (tatt_gru_wgrad_sb) + two (deferrable) split-K reductions."""
    _check_dev(dgi)
    M = dgi.shape[0]
    K = 128 if xb2 is not None else 64
    G = max(1, min(M // 32, GRU_WGRAD_GROUPS, 256))
    ws1 = _split_ws(new(dgi, G * 192 * K + G * 192))
    ws2 = _split_ws(new(dgi, G * 192 * 64 + G * 192))
    call("tatt_gru_wgrad_sb", P(dgi), P(dgh), P(x2), P(xb2), P(hprev), P(ws1), P(ws2), M, G, stream())
    call("tatt_splitk_reduce", P(ws1), P(dWp), 192, K, G, 0, 0, 0.0, P(dbp), 192, stream())
    call("tatt_splitk_reduce", P(ws2), P(dWhh), 192, 64, G, 0, 0, 0.0, P(dbhh), 192, stream())


TOK_WGRAD_SPLITS = 128        # token splits of tatt_tok_wgrad_sb (one work-group each)
TOK_WGRAD_CHUNK = 32          # tokens per staged chunk (TW_TOK of csrc/tokwgrad.hip)


def tok_wgrad_takes(dy, x):
    """tatt_tok_wgrad_sb's geometry: contiguous (M, N) / (M, K), M % 32 == 0, N and K in {64, 128}"""
    return (dy.dim() == 2 and x.dim() == 2 and dy.shape[0] == x.shape[0] and dy.is_contiguous() and x.is_contiguous()
            and dy.shape[0] % TOK_WGRAD_CHUNK == 0 and dy.shape[0] > 0 and dy.shape[1] in (64, 128) and x.shape[1] in (64, 128))


def tok_wgrad_sb(dy, x, dW, db=None):
    """dW (N, K) = dy^T x, db = dy.sum(0) (optional) in one split-bf16 pass over the tokens (tatt_tok_wgrad_sb) + a (deferrable)
    split-K reduction."""
    _check_dev(dy)
    M, N = dy.shape
    K = x.shape[1]
    chunks = M // TOK_WGRAD_CHUNK
    S = max(1, min(TOK_WGRAD_SPLITS, chunks))
    while (S - 1) * cdiv(chunks, S) >= chunks:
        S -= 1
    ws = _split_ws(new(dy, S * N * K + S * N))
    call("tatt_tok_wgrad_sb", P(dy), P(x), P(ws), M, N, K, S, stream())
    call("tatt_splitk_reduce", P(ws), P(dW), N, K, S, 0, 0, 0.0, P(db), N, stream())


QGRU_WGRAD_SPLIT = 8          # contraction splits of tatt_qgru_wgrad_sb (96 output tiles x S work-groups for both directions; 4: 81 us, 6: 85, 8: 77, 12: 82)


def qgru_wgrad_takes(A, B):
    """tatt_qgru_wgrad_sb's geometry: contiguous (M, N) / (M, K) with M % 32 == 0, N % 128 == 0, K % 128 == 0"""
    return (A.dim() == 2 and B.dim() == 2 and A.shape[0] == B.shape[0] and A.is_contiguous() and B.is_contiguous()
            and A.shape[0] % 32 == 0 and A.shape[0] > 0 and A.shape[1] % 128 == 0 and B.shape[1] % 128 == 0)


def qgru_wgrad_splits(M):
    """contraction splits tatt_qgru_wgrad_sb runs M tokens with: at most QGRU_WGRAD_SPLIT, every split owning at least one 32-token chunk"""
    chunks = M // 32
    S = max(1, min(QGRU_WGRAD_SPLIT, chunks))
    while (S - 1) * cdiv(chunks, S) >= chunks:
        S -= 1
    return S


def qgru_wgrad_sb(A0, A1, B0, B1):
    """-> (dW0, db0, dW1, db1): dW_d (N, K) = A_d^T B_d, db_d = A_d.sum(0) for both directions of the query GRU in one split-bf16
    launch (tatt_qgru_wgrad_sb) + two (deferrable) split-K reductions."""
    _check_dev(A0)
    M, N = A0.shape
    K = B0.shape[1]
    S = qgru_wgrad_splits(M)
    ws = [_split_ws(new(A0, S * N * K + S * N)) for _ in range(2)]
    call("tatt_qgru_wgrad_sb", P(A0), P(A1), P(B0), P(B1), P(ws[0]), P(ws[1]), M, N, K, S, stream())
    out = []
    for d in range(2):
        dW, db = new(A0, N, K), new(A0, N)
        call("tatt_splitk_reduce", P(ws[d]), P(dW), N, K, S, 0, 0, 0.0, P(db), N, stream())
        out += [dW, db]
    return tuple(out)


def colsum(x2, *, out=None, scale=1.0, beta=0.0):
    """out[c] = scale * sum_m x2[m, c]  (+ beta*out)."""
    _check_dev(x2)
    M, C = x2.shape
    out = out if out is not None else new(x2, C)
    ws = new(x2, 256 * C, dtype=torch.float64)
    call("tatt_colsum", P(x2), x2.stride(0), M, C, P(out), scale, beta, P(ws), stream())
    return out


# --------------------------------------------------------------------------------------------------
# convolution (x given as a (B,H,W,C)-indexed tensor with arbitrary strides)
# --------------------------------------------------------------------------------------------------
def _packed_numel(shape, mode):
    """Words of the packed layout -- asked from the library (tatt_repack_words), the one place the sizes are defined."""
    Cout, Cin, KH, KW = shape
    n = LIB.tatt_repack_words(int(Cout), int(Cin), int(KH), int(KW), int(mode))
    if n <= 0:
        raise RuntimeError("tatt_repack_words: unknown filter packing mode %d" % mode)
    return n


class _PackedHolder:
    """Packed layouts of ONE filter: {mode: (buffer, torch version counter at packing time)}.  Owned by the parameter object."""
    __slots__ = ("bufs", "key", "owner", "__weakref__")

    def __init__(self):
        self.bufs, self.key, self.owner = {}, None, None      # owner: weak reference to the parameter object


class _PackedFilters:
    """Packed copies of convolution filters (the layouts the conv kernels want, see tatt_repack_conv_weight), kept per
    (parameter, layout) instead of being rebuilt at every use -- 39 launches of a training step.

    Lifetime: the buffers of a filter live in a holder that the PARAMETER OBJECT owns (attribute `_tatt_packed`); this registry
    maps (address, shape, device) -> weak reference to the holder, so that aliases of the parameter (the tensors autograd hands
    back from `saved_tensors` are new Python objects over the same storage) find them.  When a model is released its holders go
    with it -- nothing is ever evicted while its parameter is alive (buffer pointers are baked into captured hipGraphs), and a
    long-lived process that keeps building models does not accumulate them.
    An entry is valid while the parameter's torch version counter is unchanged; in-place updates torch does not see (the Trainer's
    Adam kernel writes through raw pointers, a broadcast into the flat buffer) are followed by `refresh()`: ONE launch that rebuilds
    every live entry (captured into the optimiser's hipGraph; the buffers are persistent, so the captured pointers stay valid)."""

    def __init__(self):
        self.reg = {}

    def _holder(self, w):
        key = (w.data_ptr(), tuple(w.shape), str(w.device))
        r = self.reg.get(key)
        h = r() if r is not None else None
        if h is None:
            h = getattr(w, "_tatt_packed", None)         # re-homed parameter (FlatParams moves .data): same owner, new address
            if h is None:
                h = _PackedHolder()
                w._tatt_packed = h
            elif h.key is not None and h.key != key:
                self.reg.pop(h.key, None)                # the old address is no longer this filter's
                h.bufs = {}
            h.key = key
            h.owner = weakref.ref(w)
            if len(self.reg) >= 4096:
                self.reg = {k: v for k, v in self.reg.items() if v() is not None}
            self.reg[key] = weakref.ref(h)
        return h

    def get(self, w, mode):
        h = self._holder(w)
        e = h.bufs.get(mode)
        if e is not None and e[1] == w._version:
            return e[0]
        Cout, Cin, KH, KW = w.shape
        out = e[0] if e is not None else torch.empty(_packed_numel(w.shape, mode), device=w.device, dtype=torch.float32)
        call("tatt_repack_conv_weight", P(w), P(out), Cout, Cin, KH, KW, mode, stream())
        h.bufs[mode] = (out, w._version)
        return out

    def _live(self, device=None, skip_frozen=False):
        """Entries whose filter still lives where the key says.  The weights are read through the raw address in the key (one batched
        launch), so an entry whose owner is gone or has been re-homed since (FlatParams moved `.data`, `.to()`, a `.data =`
        assignment) must NOT be refreshed from that address -- the block may have been freed.  Such entries are dropped: the next
        `get()` re-registers the filter under its new address and packs it afresh."""
        for key, r in list(self.reg.items()):
            h = r()
            w = h.owner() if (h is not None and h.owner is not None) else None
            if h is None or w is None or (w.data_ptr(), tuple(w.shape), str(w.device)) != key:
                del self.reg[key]
                if h is not None and h.key == key:
                    h.bufs, h.key = {}, None
            elif device is None or key[2] == str(device):
                if skip_frozen and not w.requires_grad and not getattr(w, "_tatt_in_flat", False):
                    continue                             # nobody updates it behind torch's back (a frozen teacher recogniser): its layouts stay valid
                for mode, (out, _) in h.bufs.items():
                    yield key, mode, out

    def refresh(self, device=None):
        """Rebuild every cached layout from the current weights (one launch per 96 entries).  Filters of parameters that do not require
        a gradient and do not live in a flat parameter buffer are left alone (in-place updates torch sees bump their version counter and
        are caught by get())."""
        ent = list(self._live(device, skip_frozen=True))
        if not ent:
            return
        n = len(ent)
        ws = (ctypes.c_void_p * n)(*[k[0] for k, _, _ in ent])
        outs = (ctypes.c_void_p * n)(*[o.data_ptr() for _, _, o in ent])
        dims = (ctypes.c_int * (5 * n))(*[v for k, m, _ in ent for v in (k[1][0], k[1][1], k[1][2], k[1][3], m)])
        call("tatt_repack_conv_weight_batch", ws, outs, dims, n, stream())

    def clear(self):
        for r in self.reg.values():
            h = r()
            if h is not None:
                h.bufs = {}
        self.reg = {}

    @property
    def entries(self):                                   # (diagnostics / tests)
        return {(k[0], k[1], m, k[2]): o for k, m, o in self._live()}


PACKED = _PackedFilters()


def repack_weight(w_oihw, mode, cache=True):
    """Packed layout `mode` of an OIHW filter.  Parameters (leaf tensors) go through the cache; anything else is packed on the spot."""
    if cache and w_oihw.is_leaf and w_oihw.is_contiguous():
        return PACKED.get(w_oihw, mode)
    Cout, Cin, KH, KW = w_oihw.shape
    out = new(w_oihw, _packed_numel(w_oihw.shape, mode))
    call("tatt_repack_conv_weight", P(w_oihw), P(out), Cout, Cin, KH, KW, mode, stream())
    return out


# generic implicit-GEMM convolution: fewer output tiles than CONV_SPLIT_TILES and a deep contraction -> the contraction is split over
# up to CONV_SPLIT_WGS work-groups (deterministic second-stage sum).  Round 5: 128 / 256 -> 256 / 768 (the data gradient of the STN
# head's second convolution, 192 tiles x 36 K-chunks on the main lane of the last pass: 45 -> 30 us)
CONV_SPLIT_TILES, CONV_SPLIT_WGS = 256, 768


def conv_fwd(x_bhwc, wpacked, bias, Cout, KH, KW, *, act=ACT_NONE, out=None, beta=0.0):
    """Stride-1 'same' convolution from a mode-0/1 packed filter: 9x9 64k->4 goes to the vector-ALU kernel, everything else to
    the generic implicit-GEMM MFMA kernel (the specialised 3x3 kernels take their own packings: conv2d_forward / conv2d_dgrad)."""
    _check_dev(x_bhwc)
    B, H, W, Cin = x_bhwc.shape
    sn, sh, sw, sc = x_bhwc.stride()
    y = out if out is not None else new(x_bhwc, B, H, W, Cout)
    contig = x_bhwc.is_contiguous()
    if contig and KH == 9 and KW == 9 and Cout == 4 and Cin % 16 == 0 and H % 8 == 0 and W % 32 == 0 \
            and act == ACT_NONE and beta == 0.0:
        call("tatt_conv9_c64_to_c4", P(x_bhwc), P(wpacked), P(bias), P(y), B, H, W, Cin, stream())
        return y
    if contig and KH == 9 and KW == 9 and Cout == 64 and Cin == 4 and H % 4 == 0 and W % 64 == 0 and beta == 0.0 \
            and y.is_contiguous():
        call("tatt_conv9_c4_to_c64_sb" if CONV9_SB and CONV9_SB_C4 else "tatt_conv9_c4_to_c64", P(x_bhwc), P(wpacked), P(bias), P(y), B, H, W, act, stream())
        return y
    M = B * H * W
    splitk, ws = conv_split(x_bhwc, Cout, KH, KW), None
    if splitk > 1:
        ws = _split_ws(new(x_bhwc, splitk * M * Cout))           # (must outlive a deferred reduction, like every split-K slab)
    call("tatt_conv2d_fwd", P(x_bhwc), sn, sh, sw, sc, P(wpacked), P(bias), P(y), Cout, B, H, W, Cin, Cout, KH, KW,
         act, beta, splitk, P(ws), stream())
    return y


def conv_split(x_bhwc, Cout, KH, KW):
    """contraction splits the generic convolution kernel runs this shape with (1: none)"""
    B, H, W, Cin = x_bhwc.shape
    tiles, nchunks = cdiv(B * H * W, 64) * cdiv(Cout, 64), cdiv(KH * KW * Cin, 16)
    if tiles < CONV_SPLIT_TILES and nchunks >= 32:
        return max(1, min(CONV_SPLIT_WGS // tiles, nchunks // 8))
    return 1


def conv_partials(x_bhwc, wpacked, Cout, KH, KW):
    """conv_fwd's generic kernel without bias / activation, its split contraction left UNSUMMED -> (parts, S): S partial maps
    (S, B, H, W, Cout) for a consumer that adds them as it loads them (the STN head's BatchNorm launches).  S = 1: the finished map."""
    _check_dev(x_bhwc)
    B, H, W, Cin = x_bhwc.shape
    sn, sh, sw, sc = x_bhwc.stride()
    splitk = conv_split(x_bhwc, Cout, KH, KW)
    ws = new(x_bhwc, splitk, B, H, W, Cout)
    got = ctypes.c_int(0)
    call("tatt_conv2d_fwd_partials", P(x_bhwc), sn, sh, sw, sc, P(wpacked), B, H, W, Cin, Cout, KH, KW, splitk, P(ws),
         ctypes.byref(got), stream())
    return ws, int(got.value)


CONV3_SB_NARROW_MAPS = False    # True: maps with H % 4 == 0, W % 16 == 0 (not only W % 64 == 0) take the split-bf16 3x3 kernels too.  Off: the
                                # only such maps of the models are the STN head's, whose operator-chain path stays on exact-fp32 products


def _conv3_geom_ok(x_bhwc, any_width=False):
    """map sizes the 3x3 64-channel kernels tile: 64-pixel row segments, or (split-bf16 kernels of round 6, on request: `any_width` of a
    caller or the module hook) 4 x 16-pixel tiles -- the generation-4 kernel cuts the last tile column of a ragged map"""
    H, W = x_bhwc.shape[1], x_bhwc.shape[2]
    return W % 64 == 0 or (CONV3_SB and H % 4 == 0 and ((CONV3_SB_NARROW_MAPS and W % 16 == 0) or (any_width and CONV3_SB_GENERATION == 4)))


def _conv3_fast_ok(x_bhwc, Cin, Cout, KH, KW, any_width=False):
    return (x_bhwc.is_contiguous() and KH == 3 and KW == 3 and Cin % 64 == 0 and Cout % 64 == 0
            and _conv3_geom_ok(x_bhwc, any_width))


# exact-fp32 3x3 kernels (used when CONV3_SB is off): weight-stationary ws16 kernel for 64 input channels, filter packings 6 / 7
_WS_ENTRY, _WS_FWD_MODE, _WS_DGRAD_MODE = "tatt_conv3_c64_fwd_ws16", 6, 7


# The 3x3 convolutions between multiples of 64 channels on the bf16 matrix cores by operand splitting (tatt_conv3_c64_fwd_sb: three
# bf16 products per fp32 product, fp32 accumulation; measured effect on SR 1e-6, profiles/r03_split_bf16_probe.txt).  Test / A-B hook:
# False -> the exact-fp32 MFMA kernels.
CONV3_SB = True
CONV3_SB_GENERATION = 4        # test / A-B hook: which split-bf16 3x3 forward / data-gradient kernel runs (tatt_conv3_sb_generation)


def _conv3_sb(x_bhwc, w_oihw, mode, bias, act=ACT_NONE, in_scale=None, in_shift=None, in_act=ACT_NONE, stats=None):
    """3x3 'same' convolution through tatt_conv3_c64_fwd_sb; mode 10: forward (Cin = w.shape[1]), mode 11: data gradient of w
    (input channels = w.shape[0], output channels = w.shape[1]).  Contractions wider than 64 channels are chunked (beta = 1)."""
    B, H, W, cin = x_bhwc.shape
    cout = w_oihw.shape[0] if mode == 10 else w_oihw.shape[1]
    assert cin == (w_oihw.shape[1] if mode == 10 else w_oihw.shape[0])
    if LIB.tatt_conv3_sb_generation(0) != CONV3_SB_GENERATION:
        LIB.tatt_conv3_sb_generation(int(CONV3_SB_GENERATION))
    wl = repack_weight(w_oihw, LIB.tatt_conv3_sb_packing(B, H, W, cin, cout, int(act), ACT_NONE) + (mode - 10))
    y = new(x_bhwc, B, H, W, cout)
    nchunk = cin // 64
    assert nchunk == 1 or (act in (ACT_NONE, ACT_RELU) and stats is None)
    for c in range(nchunk):                                  # (an output activation belongs to the last chunk)
        call("tatt_conv3_c64_fwd_sb", P(x_bhwc), cin, 64 * c, P(wl[c * cout * 576:]), P(bias) if c == 0 else None, P(y), B, H, W, cout,
             int(act) if c == nchunk - 1 else ACT_NONE, 0.0 if c == 0 else 1.0, P(in_scale), P(in_shift), int(in_act), P(stats), stream())
    return y


# The 9x9 convolutions 64 -> 4 (output convolution forward, block1's data gradient: tatt_conv9_c64_to_c4_sb) and 4 -> 64 (block1
# forward, the output convolution's data gradient: tatt_conv9_c4_to_c64_sb) and their weight gradients (tatt_conv9_*_wgrad_sb) on the
# bf16 matrix cores with split operands.
# Test / A-B hook: False -> the exact-fp32 MFMA kernels.
CONV9_SB = True
CONV9_SB_C4 = True             # (finer hooks under CONV9_SB: the 4 -> 64 direction; the two weight gradients)
CONV9_SB_WGRAD = True


def _conv9_mfma_ok(x_bhwc):
    return x_bhwc.is_contiguous() and x_bhwc.shape[1] % 4 == 0 and x_bhwc.shape[2] % 64 == 0 and x_bhwc.shape[3] == 64


def conv2d_forward(x_bhwc, weight_oihw, bias, act=ACT_NONE, any_width=False):
    """y = act(conv(x, W) + b) from the reference-layout (OIHW) filter: picks the kernel and the filter packing it wants.
    any_width: 3x3 convolutions between multiples of 64 channels on maps whose height is a multiple of 4 take the split-bf16 kernel
    whatever their width (the CRNN's maps; off for the STN head's operator chain, which keeps exact fp32 products)."""
    Cout, Cin, KH, KW = weight_oihw.shape
    if KH == 9 and KW == 9 and Cout == 4 and Cin == 64 and act == ACT_NONE and _conv9_mfma_ok(x_bhwc):
        B, H, W, _ = x_bhwc.shape
        y = new(x_bhwc, B, H, W, 4)
        if CONV9_SB:
            call("tatt_conv9_c64_to_c4_sb", P(x_bhwc), P(repack_weight(weight_oihw, 12)), P(bias), P(y), B, H, W, stream())
        else:
            call("tatt_conv9_c64_to_c4_mfma", P(x_bhwc), P(repack_weight(weight_oihw, 8)), P(bias), P(y), B, H, W, stream())
        return y
    if _conv3_fast_ok(x_bhwc, Cin, Cout, KH, KW, any_width):
        B, H, W, _ = x_bhwc.shape
        gen4 = CONV3_SB_GENERATION == 4 and H % 4 == 0       # (generation 4: any width, ReLU on the output, chunked contractions)
        if CONV3_SB and ((gen4 and act in (ACT_NONE, ACT_RELU)) or act == ACT_NONE or (Cin == 64 and W % 64 == 0)):
            return _conv3_sb(x_bhwc, weight_oihw, 10, bias, act)
        if W % 64 == 0:                                          # (the exact-fp32 kernels walk 64-pixel row segments)
            y = new(x_bhwc, B, H, W, Cout)
            if Cin == 64:                                        # weight-stationary kernel: the filter lives in registers
                wl = repack_weight(weight_oihw, _WS_FWD_MODE)
                call(_WS_ENTRY, P(x_bhwc), P(wl), P(bias), P(y), B, H, W, Cout, act, 0.0, stream())
                return y
            wt = repack_weight(weight_oihw, 2)                   # [9][Cout][Cin]
            call("tatt_conv3_c64_fwd_t", P(x_bhwc), P(wt), P(bias), P(y), B, H, W, Cin, Cout, act, 0.0, stream())
            return y
    return conv_fwd(x_bhwc, repack_weight(weight_oihw, 0), bias, Cout, KH, KW, act=act)


def conv3_bn_fusable(x_bhwc, weight_oihw, bn=None):
    """The 3x3 64 -> 64 convolutions of the residual blocks / block7 on a contiguous NHWC map whose width is a multiple of 64: the
    weight-stationary kernel can fold the producer's BatchNorm into its input staging and emit this layer's batch statistics.
    `bn`: the nn.BatchNorm2d holder that follows -- the folded path implements the reference's configuration (affine, running
    statistics with a fixed momentum, model/tsrn.py:878,886); cumulative averaging (momentum=None), affine=False or
    track_running_stats=False take the unfused operator chain."""
    if bn is not None and (bn.momentum is None or not bn.affine or not bn.track_running_stats or bn.running_mean is None):
        return False
    return (tuple(weight_oihw.shape) == (64, 64, 3, 3) and x_bhwc.is_contiguous()
            and x_bhwc.shape[3] == 64 and _conv3_geom_ok(x_bhwc))


def conv3_bn_forward(x_bhwc, weight_oihw, bias, in_scale=None, in_shift=None, in_act=ACT_NONE, want_stats=True):
    """y = conv3x3(in_act(x * in_scale + in_shift)) + bias (no transform without in_scale) and, with want_stats, the stage-1
    partials of y's per-channel batch statistics ([G][2][64] doubles) -> (y, part, G)."""
    _check_dev(x_bhwc)
    B, H, W, _ = x_bhwc.shape
    G = min(256, B * H * W // 64)
    part = new(x_bhwc, G * 128, dtype=torch.float64) if want_stats else None
    if CONV3_SB:
        return _conv3_sb(x_bhwc, weight_oihw, 10, bias, ACT_NONE, in_scale, in_shift, in_act, part), part, G
    y = new(x_bhwc, B, H, W, 64)
    call("tatt_conv3_c64_fwd_ws16_bn", P(x_bhwc), P(repack_weight(weight_oihw, 6)), P(bias), P(y), B, H, W, 64, ACT_NONE, 0.0,
         P(in_scale), P(in_shift), int(in_act), P(part), stream())
    return y, part, G


def bn_stats_finish(part, G, C, M, eps, momentum, gamma, beta, running_mean, running_var):
    """stage-1 partials -> mean, rstd, and the folded affine map scale = gamma * rstd, shift = beta - mean * scale; updates the
    running statistics (reference nn.BatchNorm2d in train mode)."""
    mean, rstd, scale, shift = new(gamma, C), new(gamma, C), new(gamma, C), new(gamma, C)
    call("tatt_bn_stats_finish", P(part), G, C, M, float(eps), float(momentum), P(gamma), P(beta), P(mean), P(rstd),
         P(running_mean), P(running_var), P(scale), P(shift), stream())
    return mean, rstd, scale, shift


def conv2d_dgrad(dy_bhwc, weight_oihw, any_width=False):
    """dx = conv(dy, flip(W)^T): the data gradient as a forward convolution with Cout input / Cin output channels."""
    Cout, Cin, KH, KW = weight_oihw.shape
    if KH == 9 and KW == 9 and Cout == 64 and Cin == 4 and _conv9_mfma_ok(dy_bhwc):
        B, H, W, _ = dy_bhwc.shape
        dx = new(dy_bhwc, B, H, W, 4)
        if CONV9_SB:
            call("tatt_conv9_c64_to_c4_sb", P(dy_bhwc), P(repack_weight(weight_oihw, 13)), None, P(dx), B, H, W, stream())
        else:
            call("tatt_conv9_c64_to_c4_mfma", P(dy_bhwc), P(repack_weight(weight_oihw, 9)), None, P(dx), B, H, W, stream())
        return dx
    if _conv3_fast_ok(dy_bhwc, Cout, Cin, KH, KW, any_width):
        B, H, W, _ = dy_bhwc.shape
        if CONV3_SB:
            return _conv3_sb(dy_bhwc, weight_oihw, 11, None)
        dx = new(dy_bhwc, B, H, W, Cin)
        if Cout == 64:
            wl = repack_weight(weight_oihw, _WS_DGRAD_MODE)
            call(_WS_ENTRY, P(dy_bhwc), P(wl), None, P(dx), B, H, W, Cin, ACT_NONE, 0.0, stream())
            return dx
        wt = repack_weight(weight_oihw, 3)                       # [9][Cin][Cout], taps flipped
        call("tatt_conv3_c64_fwd_t", P(dy_bhwc), P(wt), None, P(dx), B, H, W, Cout, Cin, ACT_NONE, 0.0, stream())
        return dx
    return conv_fwd(dy_bhwc, repack_weight(weight_oihw, 1), None, Cin, KH, KW)


def conv_wgrad(x_bhwc, dy_bhwc, Cout, KH, KW, want_db=False, any_width=False):
    """-> dw (OIHW), or (dw, db) with want_db: the bias gradient (column sums of dy) comes out of the 3x3 64-channel kernel for
    free (it streams dy anyway); the other paths add a column-sum pass."""
    B, H, W, Cin = x_bhwc.shape
    sn, sh, sw, sc = x_bhwc.stride()
    dw = new(x_bhwc, Cout, Cin, KH, KW)
    contig = x_bhwc.is_contiguous() and dy_bhwc.is_contiguous()
    if contig and KH == 3 and KW == 3 and Cin % 64 == 0 and Cout % 64 == 0 and (W % 64 == 0 or (CONV3_WGRAD_SB and H % 4 == 0 and ((CONV3_SB_NARROW_MAPS and W % 16 == 0) or any_width))):
        nseg = B * (H // 4) * ((W + 15) // 16) if W % 64 else B * H * W // 64
        nblk = (Cin // 64) * (Cout // 64)
        G = min(nseg, max(1, (CONV3_WGRAD_GROUPS if nblk == 1 else 256) // nblk))     # (several channel blocks: one group per CU together)
        n = G * 9 * Cin * Cout
        part = _split_ws(new(x_bhwc, n + (G * Cout if want_db else 0)))
        db = new(x_bhwc, Cout) if want_db else None
        call("tatt_conv3_c64_wgrad_partial_sb" if CONV3_WGRAD_SB else "tatt_conv3_c64_wgrad_partial", P(x_bhwc), P(dy_bhwc), P(part),
             P(part[n:]) if want_db else None, B, H, W, Cin, Cout, G, stream())
        call("tatt_splitk_reduce", P(part), P(dw), 9 * Cin, Cout, G, Cin, 9, 0.0, P(db), Cout, stream())
        return (dw, db) if want_db else dw
    if want_db:
        return _conv_wgrad_general(x_bhwc, dy_bhwc, dw, Cout, KH, KW, contig), colsum(dy_bhwc.reshape(-1, Cout))
    return _conv_wgrad_general(x_bhwc, dy_bhwc, dw, Cout, KH, KW, contig)


# generic weight gradient: at most this many contraction splits (a one-tile output over 49,152 pixels -- the STN head's first
# convolution -- is 128 work-groups of 24 K-chunks each with 128: 37 us; 256: 26 us, profiles/r05_kernel_microbench.txt)
CONV_WGRAD_SPLIT_CAP = 256


def _conv_wgrad_general(x_bhwc, dy_bhwc, dw, Cout, KH, KW, contig):
    B, H, W, Cin = x_bhwc.shape
    sn, sh, sw, sc = x_bhwc.stride()
    if contig and KH == 9 and KW == 9 and Cout == 4 and Cin == 64 and H % 4 == 0 and W % 64 == 0:
        G = min(B * (H // 4) * (W // 64), 256)
        part = new(x_bhwc, G * 64 * 336)
        call("tatt_conv9_c64_c4_wgrad_sb" if CONV9_SB and CONV9_SB_WGRAD else "tatt_conv9_c64_c4_wgrad", P(x_bhwc), P(dy_bhwc), P(dw), P(part), B, H, W, stream())
        return dw
    if contig and KH == 9 and KW == 9 and Cout == 64 and Cin == 4 and H % 4 == 0 and W % 64 == 0:
        G = min(B * (H // 4) * (W // 64), 256)
        part = new(x_bhwc, G * 64 * 336)
        call("tatt_conv9_c4_c64_wgrad_sb" if CONV9_SB and CONV9_SB_WGRAD else "tatt_conv9_c4_c64_wgrad", P(x_bhwc), P(dy_bhwc), P(dw), P(part), B, H, W, stream())
        return dw
    Mo, Kred = KH * KW * Cin, B * H * W
    splitk = max(2, _auto_split(Mo, Cout, Kred, cap=CONV_WGRAD_SPLIT_CAP))
    ws = _split_ws(new(x_bhwc, (splitk + 1) * Mo * Cout))
    call("tatt_conv2d_wgrad", P(x_bhwc), sn, sh, sw, sc, P(dy_bhwc), Cout, P(dw), B, H, W, Cin, Cout, KH, KW, 0.0,
         splitk, P(ws), stream())
    return dw


# --------------------------------------------------------------------------------------------------
# normalisation
# --------------------------------------------------------------------------------------------------
def bn_stats(x2, eps, momentum, running_mean, running_var):
    M, C = x2.shape
    mean, rstd = new(x2, C), new(x2, C)
    ws = new(x2, 256 * 2 * C, dtype=torch.float64)
    call("tatt_bn_stats", P(x2), x2.stride(0), M, C, eps, momentum, P(mean), P(rstd), P(running_mean), P(running_var),
         P(ws), stream())
    return mean, rstd


def bn_rstd(var, eps):
    rstd = torch.empty_like(var)
    call("tatt_bn_rstd", P(var), P(rstd), var.numel(), eps, stream())
    return rstd


def bn_apply(x2, mean, rstd, gamma, beta, act, out=None):
    M, C = x2.shape
    y = out if out is not None else new(x2, M, C)
    call("tatt_bn_apply", P(x2), x2.stride(0), P(y), y.stride(0), M, C, P(mean), P(rstd), P(gamma), P(beta), act, stream())
    return y


def bn_bwd(x2, dy2, mean, rstd, gamma, beta, act, training):
    M, C = x2.shape
    dx = new(x2, M, C)
    dgamma, dbeta, sums = new(x2, C), new(x2, C), new(x2, 2 * C)
    ws = new(x2, 256 * 2 * C, dtype=torch.float64)
    call("tatt_bn_bwd", P(x2), x2.stride(0), P(dy2), dy2.stride(0), P(dx), C, M, C, P(mean), P(rstd), P(gamma), P(beta),
         act, int(training), P(dgamma), P(dbeta), P(sums), P(ws), stream())
    return dx, dgamma, dbeta


def bn_bwd_partials(x2, dy2, mean, rstd, gamma, beta, act):
    """Stage 1 of the train-mode BatchNorm backward alone: (part [G][2][C] doubles, G) of du = dy act'(gamma xhat + beta), du xhat."""
    M, C = x2.shape
    G = LIB.tatt_bn_bwd_groups(int(M))
    part = new(x2, G * 2 * C, dtype=torch.float64)
    call("tatt_bn_bwd_partials", P(x2), x2.stride(0), P(dy2), dy2.stride(0), M, C, P(mean), P(rstd), P(gamma), P(beta), int(act),
         P(part), stream())
    return part, G


def bn_bwd_finish(part, G, C, M, mean, rstd, gamma):
    """partials -> (dgamma, dbeta, coef (3, C)): dx = coef[0] du + coef[1] x + coef[2] is the BatchNorm backward, to be applied by the
    consumer while it stages its input (conv3_dgrad_bn) or materialised by bn_bwd_affine."""
    dgamma, dbeta, coef = new(gamma, C), new(gamma, C), new(gamma, 3, C)
    call("tatt_bn_bwd_finish", P(part), G, C, M, P(mean), P(rstd), P(gamma), P(dgamma), P(dbeta), P(coef), stream())
    return dgamma, dbeta, coef


def bn_bwd_affine(x2, du2, coef):
    M, C = x2.shape
    dx = new(x2, M, C)
    call("tatt_bn_bwd_affine", P(x2), P(du2), P(dx), M, C, P(coef), stream())
    return dx


def conv3_dgrad_bn(du_bhwc, weight_oihw, x2_bhwc=None, coef=None, ep=None):
    """Data gradient of a 64 -> 64 3x3 convolution (split-bf16 kernel) with the BatchNorm backward folded in: the gradient entering is
    coef[0] du + coef[1] x2 + coef[2] (x2 / coef None: du as it is); ep = (y_below, mean, rstd, gamma, beta, act): the result is
    multiplied by act'(bn(y_below)) and the stage-1 partials of that BatchNorm's backward come back -> (out, part or None, G)."""
    _check_dev(du_bhwc)
    B, H, W, _ = du_bhwc.shape
    if LIB.tatt_conv3_sb_generation(0) != CONV3_SB_GENERATION:
        LIB.tatt_conv3_sb_generation(int(CONV3_SB_GENERATION))
    wl = repack_weight(weight_oihw, LIB.tatt_conv3_sb_packing(B, H, W, 64, 64, ACT_NONE, int(ep[5]) if ep is not None else ACT_NONE) + 1)
    out = new(du_bhwc, B, H, W, 64)
    G = min(256, B * H * W // 64)
    part = new(du_bhwc, G * 128, dtype=torch.float64) if ep is not None else None
    a = b = c = None
    if x2_bhwc is not None:
        a, b, c = coef[0], coef[1], coef[2]
    e = ep if ep is not None else (None,) * 5 + (ACT_NONE,)
    call("tatt_conv3_c64_dgrad_bn_sb", P(du_bhwc), P(x2_bhwc), P(a), P(b), P(c), P(wl), P(out), B, H, W, P(e[0]), P(e[1]), P(e[2]),
         P(e[3]), P(e[4]), int(e[5]), P(part), stream())
    return out, part, G


def ln_fwd(a2, b2, gamma, beta, eps=1e-5, mode=0, pdrop=0.0, seed=None, site=0):
    """LayerNorm(a2 + b2), or LayerNorm(a2 + Dropout_pdrop(b2)) with the mask of `dropout(b2, pdrop, seed, site)`."""
    M, C = a2.shape
    y, stats = new(a2, M, C), new(a2, M, 2)
    call("tatt_ln_fwd", P(a2), P(b2), P(y), P(stats), M, C, P(gamma), P(beta), eps, mode, float(pdrop), P(seed), int(site),
         stream())
    return y, stats


def ln_bwd(a2, b2, dy2, stats, gamma, eps=1e-5, mode=0, pdrop=0.0, seed=None, site=0):
    """-> dx (gradient of a2), db2 (gradient of b2: dx itself without dropout, None if there is no b2), dgamma, dbeta"""
    M, C = a2.shape
    dx, dgb = new(a2, M, C), new(a2, 2 * C)
    db2 = new(a2, M, C) if (pdrop > 0.0 and b2 is not None) else None
    dgamma, dbeta = dgb[:C], dgb[C:]              # adjacent: the kernel finishes both with one launch
    G = cdiv(M, 64)
    part = new(a2, G * 2 * C)
    ws = new(a2, 256 * 2 * C, dtype=torch.float64)
    call("tatt_ln_bwd", P(a2), P(b2), P(dy2), P(stats), P(dx), P(db2), M, C, P(gamma), P(dgamma), P(dbeta), P(part), P(ws),
         eps, mode, float(pdrop), P(seed), int(site), stream())
    if db2 is None and b2 is not None:
        db2 = dx
    return dx, db2, dgamma, dbeta


# --------------------------------------------------------------------------------------------------
# element-wise
# --------------------------------------------------------------------------------------------------
def prelu_fwd(x, alpha):
    _check_dev(x)
    y = torch.empty_like(x)
    call("tatt_prelu_fwd", P(x), P(y), P(alpha), x.numel(), stream())
    return y


def prelu_bwd(x, dy, alpha):
    dx = torch.empty_like(x)
    G = cdiv(x.numel(), 256)
    part = new(x, G, 1)
    call("tatt_prelu_bwd", P(x), P(dy), P(dx), P(alpha), x.numel(), P(part), stream())
    return dx, colsum(part)


def act_fwd(x, act):
    y = torch.empty_like(x)
    call("tatt_act_fwd", P(x), P(y), x.numel(), act, stream())
    return y


def act_bwd(ref, dy, act, from_output):
    dx = torch.empty_like(dy)
    call("tatt_act_bwd", P(ref), P(dy), P(dx), dy.numel(), act, int(from_output), stream())
    return dx


def axpby(a, b, alpha=1.0, beta=1.0):
    _check_dev(a)
    y = torch.empty_like(a)
    call("tatt_axpby", P(a), P(b), P(y), alpha, beta, a.numel(), stream())
    return y


def add_n(ts):
    """((t0 + t1) + t2) + ... in one launch (2 <= len(ts) <= 8, equal shapes)."""
    ts = [t if t.is_contiguous() else t.contiguous() for t in ts]
    _check_dev(ts[0])
    y = torch.empty_like(ts[0])
    srcs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    call("tatt_add_n", srcs, len(ts), P(y), y.numel(), stream())
    return y


def gather_grads(dst_flat, entries):
    """dst_flat[off : off + n] <- src (zeros where src is None) for every (src, off, n) of `entries`: one launch per 112 entries
    (tatt_gather_grads: the bucket gather of the data-parallel flat gradient buffer without framework kernels)."""
    m = len(entries)
    if not m:
        return
    srcs = (ctypes.c_void_p * m)(*[None if s is None else s.data_ptr() for s, _, _ in entries])
    offs = (ctypes.c_long * m)(*[int(o) for _, o, _ in entries])
    ns = (ctypes.c_int * m)(*[int(n) for _, _, n in entries])
    call("tatt_gather_grads", srcs, offs, ns, m, P(dst_flat), stream())


def inc_i64(t):
    """t += 1 for an int64 tensor (contiguous), one small launch."""
    assert t.dtype == torch.int64 and t.is_contiguous()
    call("tatt_inc_i64", P(t), t.numel(), stream())


def zero_f32(t):
    assert t.dtype == torch.float32 and t.is_contiguous()
    call("tatt_zero_f32", P(t), t.numel(), stream())


def add_rowbcast(a2, b2, period):
    y = torch.empty_like(a2)
    rows = a2.numel() // a2.shape[-1]
    call("tatt_add_rowbcast", P(a2), P(b2), P(y), rows, a2.shape[-1], period, stream())
    return y


def pixel_shuffle_fwd(x, act):
    B, H, W, C4 = x.shape
    y = new(x, B, 2 * H, 2 * W, C4 // 4)
    call("tatt_pixel_shuffle_fwd", P(x), P(y), B, H, W, C4 // 4, act, stream())
    return y


def pixel_shuffle_bwd(x, dout, act):
    B, H, W, C4 = x.shape
    dx = torch.empty_like(x)
    call("tatt_pixel_shuffle_bwd", P(x), P(dout), P(dx), B, H, W, C4 // 4, act, stream())
    return dx


def maxpool_fwd(x, kh, kw, sh=None, sw=None, ph=0, pw=0):
    B, H, W, C = x.shape
    sh, sw = sh or kh, sw or kw
    y = new(x, B, (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1, C)
    call("tatt_maxpool_fwd", P(x), P(y), B, H, W, C, kh, kw, sh, sw, ph, pw, stream())
    return y


def maxpool_bwd(x, dout, kh, kw, sh=None, sw=None, ph=0, pw=0):
    B, H, W, C = x.shape
    dx = torch.empty_like(x)
    call("tatt_maxpool_bwd", P(x), P(dout), P(dx), B, H, W, C, kh, kw, sh or kh, sw or kw, ph, pw, stream())
    return dx


def dropout(x, p, seed, site):
    y = torch.empty_like(x)
    call("tatt_dropout", P(x), P(y), x.numel(), p, P(seed), site, stream())
    return y


def bump_seed(seed, snap=None):
    call("tatt_bump_seed", P(seed), P(snap), stream())


def copy4d(src, dst, sizes, sstr, dstr, beta=0.0):
    n = list(sizes)
    call("tatt_copy4d", P(src), P(dst), n[0], n[1], n[2], n[3], sstr[0], sstr[1], sstr[2], sstr[3], dstr[0], dstr[1],
         dstr[2], dstr[3], beta, stream())
    return dst


def to_contiguous(x4):
    """Materialise a 4-D strided view as a contiguous tensor (e.g. NCHW input viewed as NHWC)."""
    _check_dev(x4)
    out = new(x4, *x4.shape)
    copy4d(x4, out, x4.shape, x4.stride(), out.stride())
    return out


# --------------------------------------------------------------------------------------------------
# GRU / attention / TPS
# --------------------------------------------------------------------------------------------------
def seq_geom(B, H, W, vertical):
    """(nseq, T, s_in, stride_hi, stride_lo, stride_t) on the (B,H,W) token grid."""
    if vertical:
        return B * W, H, W, H * W, 1, W
    return B * H, W, 1, W, 0, 1


GRU32_V2 = True             # test / A-B hook: False -> the first-generation recurrences (a 32-lane group per (sequence, direction))
GRU32_FWD_V2_MIN_T = 32     # forward recurrences shorter than this stay on the first generation (test hook: 0 = always the second)


def gru32_fwd(gi, whh_f, bhh_f, whh_r, bhh_r, geom, save=False):
    """-> (out [tok][64], gates [tok][256] or None).  save: keep r, z, n and W_hn h + b_hn of every step for gru32_bwd."""
    out = new(gi, gi.shape[0], 64)
    gates = new(gi, gi.shape[0], 256) if save else None
    # forward: the one-wave-per-(sequence, direction) kernel wins where the recurrence is long (T = 64: 30.7 vs 40.8 us); the 16-step
    # vertical scans are HBM-bound and the first generation is a little ahead there (26.5 vs 29.1 us, profiles/r04_d_ubench_gru.txt)
    v2 = GRU32_V2 and geom[1] >= GRU32_FWD_V2_MIN_T
    call("tatt_gru32_fwd2" if v2 else "tatt_gru32_fwd", P(gi), P(whh_f), P(bhh_f), P(whh_r), P(bhh_r), P(out), P(gates),
         *geom, stream())
    return out, gates


def gru32_bwd(gates, out, dout, whh_f, whh_r, geom):
    dgi, dgh, hprev = new(out, out.shape[0], 192), new(out, out.shape[0], 192), torch.empty_like(out)
    if GRU32_V2:
        call("tatt_gru32_bwd2", P(gates), P(out), P(dout), P(whh_f), P(whh_r), P(dgi), P(dgh), P(hprev), None, *geom, stream())
    else:
        call("tatt_gru32_bwd", P(gates), P(out), P(dout), P(whh_f), P(whh_r), P(dgi), P(dgh), P(hprev), *geom, stream())
    return dgi, dgh, hprev


def gru_frag_ok(geom):
    """Can tatt_gru32_bwd2 leave MFMA operand fragments for this sequence geometry?  (windows of 8 steps, K-steps of 4 windows)"""
    nseq, T = geom[0], geom[1]
    return GRU32_V2 and T % 8 == 0 and (nseq * (T // 8)) % 4 == 0


def gru32_bwd_frag(gates, out, dout, whh_f, whh_r, geom):
    """BPTT that leaves dgi (M, 192) and, instead of dgh / hprev, the weight-gradient pass's operands in fragment order."""
    nseq, T = geom[0], geom[1]
    dgi = new(out, out.shape[0], 192)
    frag = new(out, nseq * T // 32 * 10240)
    call("tatt_gru32_bwd2", P(gates), P(out), P(dout), P(whh_f), P(whh_r), P(dgi), None, None, P(frag), *geom, stream())
    return dgi, frag


GRU_WGRAD_FRAG_GROUPS = 128   # persistent work-groups (= partial slabs) of tatt_gru_wgrad_frag


def gru_wgrad_frag(frag, x2, xb2, geom, dWp, dWhh_c, dbp, dbhh):
    """dWp (192, K) = dgi^T [x2 | xb2], dbp = dgi.sum(0), dWhh_c (192, 32) = [dW_hh fwd; dW_hh rev], dbhh = dgh.sum(0) from the
    fragment stream of gru32_bwd_frag: one streaming pass (tatt_gru_wgrad_frag) + two (deferrable) split-K reductions."""
    _check_dev(frag)
    nseq, T = geom[0], geom[1]
    K = 128 if xb2 is not None else 64
    G = max(1, min(nseq * T // 32, GRU_WGRAD_FRAG_GROUPS, 256))
    ws1 = _split_ws(new(frag, G * 192 * K + G * 192))
    ws2 = _split_ws(new(frag, G * 192 * 32 + G * 192))
    call("tatt_gru_wgrad_frag", P(frag), P(x2), P(xb2), P(ws1), P(ws2), *geom, G, stream())
    call("tatt_splitk_reduce", P(ws1), P(dWp), 192, K, G, 0, 0, 0.0, P(dbp), 192, stream())
    call("tatt_splitk_reduce", P(ws2), P(dWhh_c), 192, 32, G, 0, 0, 0.0, P(dbhh), 192, stream())


def attn_fwd(Q, K, V, pdrop, seed, site, need_weights=True):
    B, Lq, E = Q.shape
    S = K.shape[1]
    assert E == 64
    ctx = torch.empty_like(Q)
    wavg = new(Q, B, Lq, S) if need_weights else None
    call("tatt_attn_fwd", P(Q), P(K), P(V), P(ctx), P(wavg), B, Lq, S, pdrop, P(seed), site, stream())
    return ctx, wavg


def attn_bwd(Q, K, V, dctx, dwavg, pdrop, seed, site):
    B, Lq, E = Q.shape
    S = K.shape[1]
    dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
    part = new(Q, B * cdiv(Lq, 64) * 2 * S * 64)
    call("tatt_attn_bwd", P(Q), P(K), P(V), P(dctx), P(dwavg), P(dQ), P(dK), P(dV), P(part), B, Lq, S, pdrop, P(seed),
         site, stream())
    return dQ, dK, dV


# --------------------------------------------------------------------------------------------------
# fused TP-interpreter layer (csrc/tplayer.hip)
# --------------------------------------------------------------------------------------------------
def tplayer_geom(B, L):
    """-> (work-groups, dK/dV records per work-group, floats of kvpart, floats of ppart) of a fused-layer launch."""
    out = (ctypes.c_int * 4)()
    call("tatt_tplayer_geom", int(B), int(L), out)
    return out[0], out[1], out[2], out[3]


def _tpl_weights(lp, lnF):
    """Pointer list shared by tatt_tplayer_fwd / _bwd: lp = (in_w, in_b, out_w, out_b, w1, b1, w2, b2, lnA_w, lnA_b, lnB_w, lnB_b)."""
    return [P(t) for t in lp] + [P(lnF[0]) if lnF else None, P(lnF[1]) if lnF else None]


def tplayer_fwd(x, qpos, K, V, lp, lnF, fin_scale, fin_both, p_attn, p_res, p_ffn, seed, site0, eps, want_xout, want_wavg):
    """One fused transformer layer (see tatt_tplayer_fwd).  x (B,L,64); qpos (B,L,64) or (L,64); K, V (B,S,64).
    -> (xout or None, fin or None, wavg or None)"""
    _check_dev(x)
    B, L, E = x.shape
    S = K.shape[1]
    assert E == 64 and K.shape == (B, S, 64) and V.shape == K.shape and x.is_contiguous() and qpos.is_contiguous()
    qbs = L * 64 if qpos.dim() == 3 else 0
    xout = torch.empty_like(x) if want_xout else None
    fin = torch.empty_like(x) if lnF else None
    wavg = new(x, B, L, S) if want_wavg else None
    call("tatt_tplayer_fwd", P(x), P(qpos), qbs, P(K), P(V), *_tpl_weights(lp, lnF), float(fin_scale), int(fin_both), P(xout), P(fin),
         P(wavg), B, L, S, float(p_attn), float(p_res), float(p_ffn), P(seed), int(site0), float(eps), stream())
    return xout, fin, wavg


def tplayer_bwd(x, qpos, K, V, lp, lnF, fin_scale, fin_both, p_attn, p_res, p_ffn, seed, site0, eps, dxout, dfin, dwavg, dqacc,
                want_dqpos):
    """-> dx, dqpos (or None), kvpart, ppart (partial records for tplayer_reduce_kv / tplayer_reduce_params)"""
    B, L, E = x.shape
    S = K.shape[1]
    qbs = L * 64 if qpos.dim() == 3 else 0
    _, _, nkv, npp = tplayer_geom(B, L)
    dx = torch.empty_like(x)
    dqpos = torch.empty_like(x) if want_dqpos else None
    kvpart, ppart = new(x, nkv), new(x, npp)
    call("tatt_tplayer_bwd", P(x), P(qpos), qbs, P(K), P(V), *_tpl_weights(lp, lnF), float(fin_scale), int(fin_both), P(dxout), P(dfin),
         P(dwavg), P(dqacc), P(dx), P(dqpos), P(kvpart), P(ppart), B, L, S, float(p_attn), float(p_res), float(p_ffn), P(seed),
         int(site0), float(eps), stream())
    return dx, dqpos, kvpart, ppart


def tplayer_reduce_kv(kvpart, B, L, S):
    dK, dV = new(kvpart, B, S, 64), new(kvpart, B, S, 64)
    call("tatt_tplayer_reduce_kv", P(kvpart), P(dK), P(dV), B, L, S, stream())
    return dK, dV


def tplayer_reduce_params(ppart, B, L, dsts, betaF=0.0):
    """dsts: 14 tensors or None in the order of tatt_tplayer_reduce_params."""
    call("tatt_tplayer_reduce_params", P(ppart), B, L, *[P(t) for t in dsts], float(betaF), stream())


# second generation of the fused layer in TRAINING (csrc/tplayer2.hip: forward and backward, split-bf16 on the bf16 matrix cores, a wave
# owns 16 tokens).  False (tatt_amd.set_arithmetic("fp32"), tests) keeps the exact-fp32 first generation for every geometry; evaluation
# always runs the first generation's forward.
TPLAYER_BWD2 = True


def tplayer2_geom(B, L, S):
    """-> (taken, work-groups, floats of kvpart, floats of ppart, ints of kvflags, words of wimg, words of kvf, floats of wimg32,
    floats of kvf32)"""
    out = (ctypes.c_int * 9)()
    call("tatt_tplayer2_geom", int(B), int(L), int(S), out)
    return tuple(out[i] for i in range(9))


def tplayer2_prep(lp, K, V):
    """tatt_tplayer2_prep: the layer's four 64x64 matrices and K, V (B,S,64) as MFMA operand fragments -> (wimg, kvf, wimg32, kvf32):
    the backward's split-bf16 images (int32 tensors) and the forward's exact-fp32 ones"""
    B, S = K.shape[0], K.shape[1]
    g = tplayer2_geom(B, 64, S)
    wimg = torch.empty(g[5], dtype=torch.int32, device=K.device)
    kvf = torch.empty(g[6], dtype=torch.int32, device=K.device)
    wimg32, kvf32 = new(K, g[7]), new(K, g[8])
    call("tatt_tplayer2_prep", P(lp[0]), P(lp[2]), P(lp[4]), P(lp[6]), P(K), P(V), P(wimg), P(kvf), P(wimg32), P(kvf32), B, S, stream())
    return wimg, kvf, wimg32, kvf32


def tplayer2_kvprep(mem, pos, lps):
    """tatt_tplayer2_kvprep: ONE launch for all layers `lps` over the memory `mem` (B,S,64): mem + pos, the key / value projections of every
    layer and the packing of tplayer2_prep.  -> (kin (B,S,64), [(wimg, kvf, wimg32, kvf32) per layer])"""
    _check_dev(mem)
    B, S, E = mem.shape
    nl = len(lps)
    assert E == 64 and 1 <= nl <= 2 and mem.is_contiguous() and pos.is_contiguous()
    g = tplayer2_geom(B, 64, S)
    packs = [(torch.empty(g[5], dtype=torch.int32, device=mem.device), torch.empty(g[6], dtype=torch.int32, device=mem.device),
              new(mem, g[7]), new(mem, g[8])) for _ in range(nl)]
    kin = torch.empty_like(mem)
    arr = lambda ts: (ctypes.c_void_p * nl)(*[t.data_ptr() for t in ts])
    call("tatt_tplayer2_kvprep", P(mem), P(pos), S * 64 if pos.dim() == 3 else 0, arr([lp[0] for lp in lps]), arr([lp[1] for lp in lps]),
         arr([lp[2] for lp in lps]), arr([lp[4] for lp in lps]), arr([lp[6] for lp in lps]), arr([pk[0] for pk in packs]),
         arr([pk[1] for pk in packs]), arr([pk[2] for pk in packs]), arr([pk[3] for pk in packs]), P(kin), B, S, nl, stream())
    return kin, packs


def tplayer2_fwd(x, qpos, packed, lp, lnF, fin_scale, fin_both, p_attn, p_res, p_ffn, seed, site0, eps, want_xout, want_wavg, S,
                 want_bits=True):
    """The layer's forward, second generation (csrc/tplayer2.hip; training mode; exact fp32 products).  packed = tplayer2_prep(...).
    -> (xout or None, fin or None, wavg or None, hmask or None)"""
    _check_dev(x)
    B, L, E = x.shape
    assert E == 64 and x.is_contiguous() and qpos.is_contiguous()
    qbs = L * 64 if qpos.dim() == 3 else 0
    wimg32, kvf32 = packed[2], packed[3]
    in_w, in_b, out_w, out_b, w1, b1, w2, b2, lnA_w, lnA_b, lnB_w, lnB_b = lp
    xout = torch.empty_like(x) if want_xout else None
    fin = torch.empty_like(x) if lnF else None
    wavg = new(x, B, L, S) if want_wavg else None
    hmask = torch.empty(B * L, dtype=torch.int64, device=x.device) if want_bits else None
    call("tatt_tplayer2_fwd", P(x), P(qpos), qbs, P(wimg32), P(kvf32), P(in_b), P(out_b), P(b1), P(b2), P(lnA_w), P(lnA_b), P(lnB_w), P(lnB_b),
         P(lnF[0]) if lnF else None, P(lnF[1]) if lnF else None, float(fin_scale), int(fin_both), P(xout), P(fin), P(wavg), P(hmask),
         B, L, S, float(p_attn), float(p_res), float(p_ffn), P(seed), int(site0), float(eps), stream())
    return xout, fin, wavg, hmask


def tplayer2_bwd(x, qpos, packed, lp, lnF, fin_scale, fin_both, p_attn, p_res, p_ffn, seed, site0, eps, dxout, dfin, dwavg, dqacc,
                 want_dqpos, S, hmask=None):
    """tatt_tplayer2_bwd -> dx, dqpos (or None), kvpart, kvflags, ppart, work-groups.  packed = tplayer2_prep(...) (the forward's);
    hmask: the relu bits tplayer2_fwd left (None: the recomputation decides the relu itself -- may flip kinks, see csrc/tplayer2.hip)"""
    B, L, E = x.shape
    qbs = L * 64 if qpos.dim() == 3 else 0
    taken, G, nkv, npp, nfl = tplayer2_geom(B, L, S)[:5]
    assert taken
    wimg, kvf = packed[0], packed[1]
    in_w, in_b, out_w, out_b, w1, b1, w2, b2, lnA_w, lnA_b, lnB_w, lnB_b = lp
    dx = torch.empty_like(x)
    dqpos = torch.empty_like(x) if want_dqpos else None
    kvpart, ppart = new(x, nkv), new(x, npp)
    kvflags = torch.empty(nfl, dtype=torch.int32, device=x.device)
    call("tatt_tplayer2_bwd", P(x), P(qpos), qbs, P(wimg), P(kvf), P(in_b), P(out_b), P(b1), P(b2), P(lnA_w), P(lnA_b), P(lnB_w),
         P(lnB_b), P(lnF[0]) if lnF else None, P(lnF[1]) if lnF else None, float(fin_scale), int(fin_both), P(dxout), P(dfin),
         P(dwavg), P(dqacc), P(dx), P(dqpos), P(kvpart), P(ppart), P(kvflags), P(hmask), B, L, S, float(p_attn), float(p_res),
         float(p_ffn), P(seed), int(site0), float(eps), stream())
    return dx, dqpos, kvpart, kvflags, ppart, G


def tplayer2_reduce_kv(kvpart, kvflags, B, L, S):
    dK, dV = new(kvpart, B, S, 64), new(kvpart, B, S, 64)
    call("tatt_tplayer2_reduce_kv", P(kvpart), P(kvflags), P(dK), P(dV), B, L, S, stream())
    return dK, dV


def tplayer_reduce_params_g(ppart, G, dsts, betaF=0.0):
    """dsts: 14 tensors or None in the order of tatt_tplayer_reduce_params; G: records in ppart."""
    call("tatt_tplayer_reduce_params_g", P(ppart), int(G), *[P(t) for t in dsts], float(betaF), stream())


def tps_grid_fwd(ctrl, inv, pad, repr_):
    B, N, _ = ctrl.shape
    Pn = repr_.shape[0]
    src = new(ctrl, B, Pn, 2)
    call("tatt_tps_grid_fwd", P(ctrl), P(inv), P(pad), P(repr_), P(src), B, N, Pn, stream())
    return src


def tps_grid_bwd(dsrc, inv, repr_, N):
    B, Pn, _ = dsrc.shape
    dctrl = new(dsrc, B, N, 2)
    call("tatt_tps_grid_bwd", P(dsrc), P(inv), P(repr_), P(dctrl), B, N, Pn, stream())
    return dctrl


def grid_sample_fwd(x_nchw, src):
    B, C, H, W = x_nchw.shape
    sn, sc, sh, sw = x_nchw.stride()
    out = new(x_nchw, B, H, W, C)
    call("tatt_grid_sample_fwd", P(x_nchw), sn, sc, sh, sw, P(src), P(out), B, C, H, W, stream())
    return out


def grid_sample_bwd(x_nchw, src, dout):
    B, C, H, W = x_nchw.shape
    sn, sc, sh, sw = x_nchw.stride()
    dsrc = torch.empty_like(src)
    call("tatt_grid_sample_bwd", P(x_nchw), sn, sc, sh, sw, P(src), P(dout), P(dsrc), B, C, H, W, stream())
    return dsrc
