"""First-contact GPU run: quick correctness of every kernel family against numpy, then timing
sweeps over the tuning variants.  Writes gpurun_out/first.json.  (Dev tool, not a test.)"""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from numpower_amd import device as D
from numpower_amd._lib import load, check, Timer

res = {"checks": {}, "timing": {}}
D.init(0)
lib = load()
rng = np.random.default_rng(0)

def relerr(x, ref):
    ref = np.asarray(ref, dtype=np.float64); x = np.asarray(x, dtype=np.float64)
    d = np.abs(x - ref); s = np.maximum(np.abs(ref), 1e-30)
    m = np.isfinite(ref)
    return float(np.max(d[m] / s[m])) if m.any() else 0.0

# ---- correctness -------------------------------------------------------------------------
def chk(name, got, ref, tol=1e-5):
    e = relerr(got, ref)
    res["checks"][name] = {"relerr": e, "ok": bool(e <= tol)}
    print(name, e, "OK" if e <= tol else "FAIL", flush=True)

for shape in [(2, 2), (1000, 1000), (257, 1001), (64, 4096)]:
    a = rng.uniform(-2, 2, shape).astype(np.float32); b = rng.uniform(0.5, 2, shape).astype(np.float32)
    da, db = D.DeviceArray.from_host(a), D.DeviceArray.from_host(b)
    r, c = shape
    chk(f"add{shape}", D.binary("add", da, "full", db, "full", r, c).to_host(), a + b, 0)
    chk(f"sub{shape}", D.binary("subtract", da, "full", db, "full", r, c).to_host(), a - b, 0)
    chk(f"mul{shape}", D.binary("multiply", da, "full", db, "full", r, c).to_host(), a * b, 0)
    chk(f"div{shape}", D.binary("divide", da, "full", db, "full", r, c).to_host(), a / b, 0)
    chk(f"mod{shape}", D.binary("mod", da, "full", db, "full", r, c).to_host(), np.fmod(a, b), 0)
    chk(f"pow{shape}", D.binary("pow", db, "full", da, "full", r, c).to_host(), np.power(b.astype(np.float64), a.astype(np.float64)), 1e-5)
    row = b[0].copy(); col = b[:, 0].copy()
    drow, dcol = D.DeviceArray.from_host(row), D.DeviceArray.from_host(col)
    ds = D.DeviceArray.from_host(np.float32([1.5]))
    chk(f"add_row{shape}", D.binary("add", da, "full", drow, "row", r, c).to_host(), a + row[None, :], 0)
    chk(f"add_col{shape}", D.binary("add", da, "full", dcol, "col", r, c).to_host(), a + col[:, None], 0)
    chk(f"row_sub{shape}", D.binary("subtract", drow, "row", da, "full", r, c).to_host(), row[None, :] - a, 0)
    chk(f"col_div{shape}", D.binary("divide", dcol, "col", da, "full", r, c).to_host(), col[:, None] / a, 0)
    chk(f"add_scalar{shape}", D.binary("add", da, "full", ds, "scalar", r, c).to_host(), a + np.float32(1.5), 0)
    chk(f"scalar_sub{shape}", D.binary("subtract", ds, "scalar", da, "full", r, c).to_host(), np.float32(1.5) - a, 0)
    for op, f in [("exp", np.exp), ("log", lambda x: np.log(np.abs(x) + 0.1)), ("sin", np.sin), ("tanh", np.tanh), ("sqrt", lambda x: np.sqrt(np.abs(x)))]:
        x = a if op not in ("log", "sqrt") else (np.abs(a) + np.float32(0.1) if op == "log" else np.abs(a))
        ref = {"exp": np.exp, "log": np.log, "sin": np.sin, "tanh": np.tanh, "sqrt": np.sqrt}[op](x.astype(np.float64))
        chk(f"{op}{shape}", D.unary(op, D.DeviceArray.from_host(x)).to_host(), ref, 2e-6)
    chk(f"sum{shape}", D.reduce_all("sum", da), a.astype(np.float64).sum(), 1e-5)
    chk(f"max{shape}", D.reduce_all("max", da), a.max(), 0)
    chk(f"min{shape}", D.reduce_all("min", da), a.min(), 0)
    chk(f"mean{shape}", D.reduce_all("mean", db), b.astype(np.float64).mean(), 1e-5)
    chk(f"sum0{shape}", D.reduce_axis("sum", db, 0).to_host(), b.astype(np.float64).sum(0), 1e-5)
    chk(f"sum1{shape}", D.reduce_axis("sum", db, 1).to_host(), b.astype(np.float64).sum(1), 1e-5)
    chk(f"max0{shape}", D.reduce_axis("max", da, 0).to_host(), a.max(0), 0)
    chk(f"min1{shape}", D.reduce_axis("min", da, 1).to_host(), a.min(1), 0)
    chk(f"mean0{shape}", D.reduce_axis("mean", db, 0).to_host(), b.astype(np.float64).mean(0), 1e-5)

x3 = rng.uniform(0.5, 1.5, (7, 300, 64)).astype(np.float32)
d3 = D.DeviceArray.from_host(x3)
for ax in range(3):
    chk(f"sum3d_ax{ax}", D.reduce_axis("sum", d3, ax).to_host(), x3.astype(np.float64).sum(ax), 1e-5)
    chk(f"prod3d_ax{ax}", D.reduce_axis("prod", d3, ax).to_host(), x3.astype(np.float64).prod(ax), 1e-4)
big = rng.uniform(0, 1, (8192, 4096)).astype(np.float32)
dbig = D.DeviceArray.from_host(big)
chk("sum0_big", D.reduce_axis("sum", dbig, 0).to_host(), big.astype(np.float64).sum(0), 1e-5)
chk("sum1_big", D.reduce_axis("sum", dbig, 1).to_host(), big.astype(np.float64).sum(1), 1e-5)
chk("sum_big", D.reduce_all("sum", dbig), big.astype(np.float64).sum(), 1e-5)

for (m, n, k) in [(2, 2, 2), (2, 1, 2), (64, 64, 64), (128, 128, 128), (100, 90, 70), (257, 129, 65), (512, 512, 512), (1000, 1000, 1000), (1024, 1024, 1024)]:
    A = rng.uniform(-1, 1, (m, k)).astype(np.float32); B = rng.uniform(-1, 1, (k, n)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    absref = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    for var in (0, 1, 2, 4, 31):
        lib.np_sgemm_set_variant(var)
        got = D.sgemm(D.DeviceArray.from_host(A), D.DeviceArray.from_host(B)).to_host()
        e = float(np.max(np.abs(got - ref) / absref))
        res["checks"][f"sgemm{(m,n,k)}v{var}"] = {"relerr": e, "ok": bool(e < 1e-6)}
        print("sgemm", (m, n, k), var, e, flush=True)
lib.np_sgemm_set_variant(0)
A = rng.uniform(-1, 1, (5, 130, 70)).astype(np.float32); B = rng.uniform(-1, 1, (5, 70, 200)).astype(np.float32)
got = D.sgemm_batched(D.DeviceArray.from_host(A), D.DeviceArray.from_host(B)).to_host()
chk("sgemm_batched", got, np.einsum("bmk,bkn->bmn", A.astype(np.float64), B.astype(np.float64)), 1e-4)
A = rng.uniform(-1, 1, (300, 1000)).astype(np.float32); xv = rng.uniform(-1, 1, 1000).astype(np.float32)
chk("sgemv", D.sgemv(D.DeviceArray.from_host(A), D.DeviceArray.from_host(xv)).to_host(), A.astype(np.float64) @ xv.astype(np.float64), 1e-4)
json.dump(res, open("gpurun_out/first.json", "w"), indent=1)

# ---- timing ---------------------------------------------------------------------------------
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    D.sync()
    ts = []
    for _ in range(iters):
        t = Timer(); t.start(); fn(); t.stop(); ts.append(t.elapsed_ms())
    ts.sort()
    return ts[len(ts) // 2], ts[0]

N = 100_000_000
a = D.DeviceArray((N,)); b = D.DeviceArray((N,)); o = D.DeviceArray((N,))
D.fill(a, 1.25); D.fill(b, 2.5); D.sync()
for var in [0, 101, 102, 104, 108, 114, 124, 144, 184, 1, 2, 4, 8, 14, 24, 44, 84, 122, 142, 128, 148]:
    lib.np_elementwise_set_variant(var)
    med, mn = timeit(lambda: D.binary("add", a, "full", b, "full", 1, N, out=o))
    res["timing"][f"add1e8_v{var}"] = {"ms_med": med, "ms_min": mn, "GBps_med": 1.2e9 / med / 1e6}
    print("add var", var, med, mn, 1.2 / med * 1e3, "GB/s", flush=True)
    json.dump(res, open("gpurun_out/first.json", "w"), indent=1)
lib.np_elementwise_set_variant(0)
for op in ["exp", "log", "sin", "tanh", "sqrt", "abs"]:
    med, mn = timeit(lambda: D.unary(op, a, out=o))
    res["timing"][f"{op}1e8"] = {"ms_med": med, "ms_min": mn, "GBps_med": 0.8e9 / med / 1e6}
    print(op, med, 0.8 / med * 1e3, "GB/s", flush=True)
for op in ["multiply", "divide", "mod", "pow"]:
    med, mn = timeit(lambda: D.binary(op, a, "full", b, "full", 1, N, out=o), iters=10)
    res["timing"][f"{op}1e8"] = {"ms_med": med, "ms_min": mn, "GBps_med": 1.2e9 / med / 1e6}
    print(op, med, 1.2 / med * 1e3, "GB/s", flush=True)
R, Cc = 25000, 4000
row = D.DeviceArray((Cc,)); col = D.DeviceArray((R,)); D.fill(row, 1.0); D.fill(col, 2.0)
med, mn = timeit(lambda: D.binary("add", a, "full", row, "row", R, Cc, out=o))
res["timing"]["add_row_25000x4000"] = {"ms_med": med, "GBps_med": 0.8e9 / med / 1e6}
print("add_row", med, 0.8 / med * 1e3, flush=True)
med, mn = timeit(lambda: D.binary("add", a, "full", col, "col", R, Cc, out=o))
res["timing"]["add_col_25000x4000"] = {"ms_med": med, "GBps_med": 0.8e9 / med / 1e6}
print("add_col", med, 0.8 / med * 1e3, flush=True)
med, mn = timeit(lambda: D.reduce_all("sum", a))
res["timing"]["sum_all_1e8"] = {"ms_med": med, "GBps_med": 0.4e9 / med / 1e6}
print("sum_all", med, 0.4 / med * 1e3, flush=True)
a.free(); b.free(); o.free()
X = D.DeviceArray((65536, 4096)); D.fill(X, 0.5); out = D.DeviceArray((4096,)); out1 = D.DeviceArray((65536,))
med, mn = timeit(lambda: D.reduce_axis("sum", X, 0, out=out))
res["timing"]["sum_axis0_65536x4096"] = {"ms_med": med, "ms_min": mn, "GBps_med": 1.0737e9 / med / 1e6}
print("sum_axis0", med, 1.0737 / med * 1e3, flush=True)
med, mn = timeit(lambda: D.reduce_axis("sum", X, 1, out=out1))
res["timing"]["sum_axis1_65536x4096"] = {"ms_med": med, "ms_min": mn, "GBps_med": 1.0737e9 / med / 1e6}
print("sum_axis1", med, 1.0737 / med * 1e3, flush=True)
X.free()
json.dump(res, open("gpurun_out/first.json", "w"), indent=1)

n = 4096
A = D.DeviceArray.from_host(rng.uniform(-1, 1, (n, n)).astype(np.float32))
B = D.DeviceArray.from_host(rng.uniform(-1, 1, (n, n)).astype(np.float32))
Cm = D.DeviceArray((n, n))
for var in [1, 2, 3, 11, 21, 41, 81, 42, 4]:
    lib.np_sgemm_set_variant(var)
    med, mn = timeit(lambda: D.sgemm(A, B, out=Cm), iters=10, warm=2)
    res["timing"][f"sgemm4096_v{var}"] = {"ms_med": med, "ms_min": mn, "TFLOPs_med": 2 * n**3 / med / 1e9}
    print("sgemm var", var, med, 2 * n**3 / med / 1e9, "TF", flush=True)
    json.dump(res, open("gpurun_out/first.json", "w"), indent=1)
lib.np_sgemm_set_variant(0)
Ab = D.DeviceArray((64, 1024, 1024)); Bb = D.DeviceArray((64, 1024, 1024)); Cb = D.DeviceArray((64, 1024, 1024))
D.fill(Ab, 0.5); D.fill(Bb, 0.25)
med, mn = timeit(lambda: D.sgemm_batched(Ab, Bb, out=Cb), iters=10, warm=2)
res["timing"]["sgemm_batched_64x1024"] = {"ms_med": med, "TFLOPs_med": 64 * 2 * 1024**3 / med / 1e9}
print("batched", med, 64 * 2 * 1024**3 / med / 1e9, flush=True)
json.dump(res, open("gpurun_out/first.json", "w"), indent=1)
bad = [k for k, v in res["checks"].items() if not v["ok"]]
print("FAILED CHECKS:", bad)
