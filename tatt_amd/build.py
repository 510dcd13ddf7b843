"""Build libtatt_hip.so (hipcc, --offload-arch=gfx950) in-tree.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import fcntl
import hashlib
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libtatt_hip.so")
SOURCES = ["gemm.hip", "conv3.hip", "conv3w.hip", "conv9.hip", "norm.hip", "elementwise.hip", "gru.hip", "attn.hip", "sattn.hip", "sattn2.hip", "tplayer.hip", "tplayer2.hip", "tokgemm.hip", "tokwgrad.hip", "gruwgrad.hip", "tps.hip", "loss.hip", "lstm.hip", "ssim.hip", "stnhead.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=fast"]
# per-source additions.  conv3.hip: the staging waves of the 3x3 kernels run beside MFMA waves on the same SIMD, and packed fp32 VALU forms
# (what SLP vectorisation makes of adjacent scalar adds / fmas) take issue time from the matrix pipe (profiles/r06_conv3_sb4_roles.txt)
EXTRA_FLAGS = {"conv3.hip": ["-fno-slp-vectorize"], "conv3w.hip": ["-fno-slp-vectorize"], "sattn2.hip": ["-fno-slp-vectorize"]}


def _digest(paths, extra=()) -> str:
    h = hashlib.sha256(" ".join(FLAGS + list(extra)).encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _manifest_path(objdir):
    return os.path.join(objdir, "manifest.json")


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile what changed and link.  Staleness is decided by CONTENT hashes (a snapshot copied to another machine keeps
    the prebuilt objects valid whatever happened to the mtimes); concurrent callers (one process per GPU under
    torch.distributed.run) serialise on a file lock and the library is replaced atomically."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    common = os.path.join(CSRC, "common.h")
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            try:
                with open(_manifest_path(objdir)) as f:
                    manifest = json.load(f)
            except (OSError, ValueError):
                manifest = {}
            jobs, digests = [], {}
            for s in SOURCES:
                src = os.path.join(CSRC, s)
                obj = os.path.join(objdir, s.replace(".hip", ".o"))
                digests[s] = _digest([src, common], EXTRA_FLAGS.get(s, ()))
                if force or not os.path.exists(obj) or manifest.get(s) != digests[s]:
                    jobs.append((src, obj))

            def cc(job):
                src, obj = job
                cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
                r = subprocess.run(cmd, capture_output=True, text=True)
                if r.returncode != 0:
                    raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
                return src

            if jobs:
                if verbose:
                    print("[tatt_amd.build] compiling %d HIP source(s) for gfx950" % len(jobs), file=sys.stderr)
                with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
                    list(ex.map(cc, jobs))
            objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
            if force or jobs or not os.path.exists(LIB) or manifest.get("__lib__") != _digest(objs):
                tmp = LIB + ".tmp.%d" % os.getpid()
                cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
                r = subprocess.run(cmd, capture_output=True, text=True)
                if r.returncode != 0:
                    raise RuntimeError("link failed:\n%s" % r.stderr)
                os.replace(tmp, LIB)
                digests["__lib__"] = _digest(objs)
                with open(_manifest_path(objdir), "w") as f:
                    json.dump(digests, f, indent=0, sort_keys=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
