"""Random-shape fuzz of the mid-size and small-tile GEMM kernels: random M, N, K (1 .. 1500, a third of them multiples of 16 / 32 /
64); a third of the cases force sgemm_dmas_kernel with a random tile shape and split S, a third force sgemm_kq_kernel (any of its 7 tile
shapes; K and N multiples of 4 in most cases — the float4 form — and anything in the others, the dword form, operands 16-byte aligned — the rest falls through to the planner, which is part of the
test), a third take the default planner; operands at random 4-byte offsets inside NaN-filled allocations, C inside a canary
frame: within 1e-6 |A|.|B| of the fp64 product, nothing written outside C, the same bits on a second run.
Usage: python tools/gemm_mid_fuzz.py [cases = 300] [seed = 1]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D
from numpower_amd._lib import check, load

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
D.init(0)
lib = load()
bad = 0
for case in range(cases):
    def dim():
        r = rng.random()
        if r < 0.33:
            return int(rng.choice([16, 32, 64, 128, 256, 512, 1024])) * int(rng.integers(1, 3))
        return int(rng.integers(1, 1500))
    m, n, k = dim(), max(4, dim()), max(4, dim())
    shape, S = int(rng.choice([0, 1, 2, 3, 5])), int(rng.choice([1, 1, 2, 4, 8, 16]))
    oa, ob, oc = (int(x) for x in rng.integers(0, 4, 3))
    form = ["dmas", "kq", "plan"][case % 3]
    if form == "kq":
        shape = int(rng.integers(0, 7))
        if rng.random() < 0.85:
            k = max(4, k // 4 * 4)
            n = max(4, n // 4 * 4)
            oa, ob = (int(x) * 4 for x in rng.integers(0, 2, 2))
    A = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    B = rng.uniform(-1, 1, (k, n)).astype(np.float32)
    ha = np.full(m * k + 8, np.nan, np.float32); ha[oa:oa + m * k] = A.reshape(-1)
    hb = np.full(k * n + 8, np.nan, np.float32); hb[ob:ob + k * n] = B.reshape(-1)
    hc = np.full(m * n + 64, -777.0, np.float32)
    da, db, dc = D.DeviceArray.from_host(ha), D.DeviceArray.from_host(hb), D.DeviceArray.from_host(hc)
    check(lib.np_sgemm_set_variant(-999))
    if form == "dmas":
        check(lib.np_sgemm_set_variant(-(1000 + 100 * shape + S)))
    elif form == "kq":
        check(lib.np_sgemm_set_variant(-(2000 + shape)))
    runs = []
    for _ in range(2):
        check(lib.np_memcpy_h2d(dc.ptr, hc.ctypes.data, hc.nbytes))
        check(lib.np_sgemm(m, n, k, da.ptr + 4 * oa, db.ptr + 4 * ob, dc.ptr + 4 * (32 + oc)))
        runs.append(dc.to_host().copy())
    check(lib.np_sgemm_set_variant(-999))
    got = runs[0][32 + oc:32 + oc + m * n].reshape(m, n).astype(np.float64)
    want = A.astype(np.float64) @ B.astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64) + 1e-30
    frame_ok = (runs[0][:32 + oc] == -777.0).all() and (runs[0][32 + oc + m * n:] == -777.0).all()
    ok = (not np.isnan(got).any()) and (np.abs(got - want) <= 1e-6 * scale).all() and frame_ok and \
        (runs[0].view(np.uint32) == runs[1].view(np.uint32)).all()
    if not ok:
        bad += 1
        print("MISMATCH case %d (%s): %d x %d x %d shape %d S %d offsets %d %d %d frame %s" % (case, form, m, n, k, shape, S, oa, ob, oc, frame_ok), flush=True)
    for d in (da, db, dc):
        d.free()
rc = lib.np_sync()
print("%d cases, %d mismatches, np_sync rc %d" % (cases, bad, rc))
sys.exit(1 if bad or rc else 0)
