"""ctypes binding of libtatt_hip.so.  The signatures are parsed from include/tatt_hip.h, the single
source of truth for the C ABI, so argument marshalling cannot drift from the header."""
from __future__ import annotations

import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "tatt_hip.h")
LIB_PATH = os.path.join(HERE, "lib", "libtatt_hip.so")

_CT = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "unsigned": ctypes.c_uint,
       "hipStream_t": ctypes.c_void_p}


def parse_header(path: str = HEADER):
    """-> {name: [(ctype, argname), ...]} for every `int tatt_*(...)` prototype."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(tatt_\w+)\s*\(([^)]*)\)\s*;", txt):
        args = []
        for a in m.group(2).split(","):
            a = " ".join(a.split())
            if "*" in a:
                args.append((ctypes.c_void_p, a.split("*")[-1].strip()))
            else:
                ty, nm = a.rsplit(" ", 1)
                args.append((_CT[ty], nm))
        protos[m.group(1)] = args
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self.protos = parse_header()

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    "tatt_amd: %s is missing -- run `python -m tatt_amd.build` (hipcc, gfx950). "
                    "There is no CPU / PyTorch fallback for the product path." % LIB_PATH)
            dll = ctypes.CDLL(LIB_PATH)
            for name, args in self.protos.items():
                fn = getattr(dll, name)          # AttributeError if the symbol is not exported
                fn.restype = ctypes.c_int
                fn.argtypes = [t for t, _ in args]
            self._dll = dll
        return self._dll

    def __getattr__(self, name):
        if name.startswith("tatt_"):
            return getattr(self.load(), name)
        raise AttributeError(name)


LIB = _Lib()
