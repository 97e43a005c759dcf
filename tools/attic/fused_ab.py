"""A/B of the fused-chain kernel's knobs (NP_FUSED_U, NP_FUSED_FULL are read once per process, so
each configuration runs in its own subprocess).  Usage: python tools/fused_ab.py"""
import ctypes as C
import json
import os
os.environ.setdefault("NP_HIP_USE_TUNING_BUILD", "1")   # needs `python -m numpower_amd.build --tuning`
import subprocess
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))


def one(chain):
    from numpower_amd import _lib
    from numpower_amd._lib import BINARY_OPS, UNARY_OPS, FusedOp
    lib = _lib.load()
    _lib.check(lib.np_init(0))
    n = 100_000_000
    bufs = [_lib.DeviceBuffer(4 * n) for _ in range(4)]
    import numpy as np
    from numpower_amd import synth
    for i, b in enumerate(bufs[:3]):          # random data: constant fills read ~4 % faster
        h = synth.uniform((n,), 40 + i, 0.25, 1.75)
        _lib.check(lib.np_memcpy_h2d(b.ptr, h.ctypes.data, 4 * n))
    two = C.c_float(2.0)
    if chain in ("exp_mul_add", "exp_mul_add_sum"):
        inputs = [bufs[0].ptr, bufs[1].ptr, C.addressof(two)]
        kinds = [0, 0, 4]
        ops = [FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0), FusedOp(1, BINARY_OPS["multiply"], 1, 0, 0, 0, 1, n // 8 * 8),
               FusedOp(1, BINARY_OPS["add"], 2, 0, 0, 0, 0, 0)]
        nbytes = 12 * n
    elif chain in ("exp_row", "exp_col"):     # exp(X) + r on 25000 x 4000 (BASELINE C3c in one pass): 8 B/elem
        inputs = [bufs[0].ptr, bufs[1].ptr]
        kinds = [0, 2 if chain == "exp_row" else 3]
        ops = [FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0), FusedOp(1, BINARY_OPS["add"], 1, 0, 0, 0, 0, 0)]
        nbytes = 8 * n
    elif chain in ("fma3", "fma3_sum"):       # a*b+c : 3 arrays in, 1 out
        inputs = [bufs[0].ptr, bufs[1].ptr, bufs[2].ptr]
        kinds = [0, 0, 0]
        ops = [FusedOp(1, BINARY_OPS["multiply"], 1, 0, 0, 0, 1, n // 8 * 8), FusedOp(1, BINARY_OPS["add"], 2, 0, 0, 0, 0, 0)]
        nbytes = 16 * n
    else:                        # long unary chain: 1 in 1 out, 6 ops
        inputs = [bufs[0].ptr]
        kinds = [0]
        ops = [FusedOp(0, UNARY_OPS[u], 0, 0, 0, 0, 0, 0) for u in ("exp", "log1p", "sqrt", "tanh", "abs", "sin")]
        nbytes = 8 * n
    reduce = chain.endswith("_sum")
    if reduce:
        nbytes -= 4 * n
    arr = (C.c_void_p * len(inputs))(*inputs)
    k = (C.c_int * len(kinds))(*kinds)
    o = (FusedOp * len(ops))(*ops)
    t = _lib.Timer()
    res = C.c_float(0.0)

    def launch():
        if reduce:
            _lib.check(lib.np_fused_chain_reduce(arr, k, len(inputs), o, len(ops), 0, 1, n, C.byref(res)))
        else:
            rows, cols = (25000, 4000) if chain in ("exp_row", "exp_col") else (1, n)
            _lib.check(lib.np_fused_chain(arr, k, len(inputs), o, len(ops), bufs[3].ptr, rows, cols))
    for _ in range(5):
        launch()
    t.start()
    reps = 30
    for _ in range(reps):
        launch()
    t.stop()
    _lib.check(lib.np_sync())
    ms = t.elapsed_ms() / reps
    print(json.dumps({"chain": chain, "U": os.environ.get("NP_FUSED_U"), "FULL": os.environ.get("NP_FUSED_FULL"), "RBPC": os.environ.get("NP_FUSED_RBPC"),
                      "ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 1)}))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(sys.argv[1])
    else:
        if os.environ.get("FUSED_AB_BCAST"):
            for rnd in range(2):
                for chain in ("exp_row", "exp_col", "exp_mul_add"):
                    for u in ("1", "2"):
                        subprocess.run([sys.executable, __file__, chain], env=dict(os.environ, NP_FUSED_U=u), check=False)
            sys.exit(0)
        if os.environ.get("FUSED_AB_REDUCE"):
            for chain in ("exp_mul_add_sum", "fma3_sum"):
                for u in ("1", "2"):
                    for rbpc in os.environ.get("FUSED_AB_RBPCS", "4,8,16,32,64").split(","):
                        env = dict(os.environ, NP_FUSED_U=u, NP_FUSED_RBPC=rbpc)
                        subprocess.run([sys.executable, __file__, chain], env=env, check=False)
            sys.exit(0)
        for chain in ("exp_mul_add", "fma3", "unary6"):
            for u in ("1", "2"):
                for full in (None, "1"):
                    env = dict(os.environ, NP_FUSED_U=u)
                    if full:
                        env["NP_FUSED_FULL"] = full
                    subprocess.run([sys.executable, __file__, chain], env=env, check=False)
