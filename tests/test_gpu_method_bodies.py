"""ext/method_bodies.c on the GPU: the device branch of the reference's PHP_METHODs (numpower.c), written
in C against include/numpower_host.h + ext/hip_math.h with the reference's own call expressions —
`NDArrayMathGPU_ElementWise(nda, cuda_float_sin)` with the function POINTER, `NDArray_Add_Float`,
`reduce(nda, &axis_i, NDArray_Add_Float)`, `NDArray_Matmul` ... — compiled by gcc -Werror, run as a
process of its own (no Python, no ctypes in it) and checked here against the oracle.

Bars as everywhere: bit-exact for exact ops, 1e-5 relative for libm-class ops / sums / GEMM."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from tests.test_gpu_parity import EXACT_UNARY, REL_TOL, assert_bit_equal, assert_close

ROOT = Path(__file__).resolve().parent.parent
EXE = ROOT / "numpower_amd" / "lib" / "method_bodies"
ROWS, COLS = 257, 255


def c_input(rows, cols, seed, lo, hi):
    """ext/method_bodies.c input(): x[i] = lo + (hi - lo) * frac(i * 0.6180339887 + seed * 0.37), in double."""
    n = (rows or 1) * cols
    t = np.arange(n, dtype=np.float64) * 0.6180339887 + float(seed) * 0.37
    t -= np.trunc(t)
    lo64, hi64 = float(np.float32(lo)), float(np.float32(hi))
    x = (lo64 + (hi64 - lo64) * t).astype(np.float32)
    return x.reshape(rows, cols) if rows else x


def read_records(path):
    raw = Path(path).read_bytes()
    out, pos = {}, 0
    while pos < len(raw):
        name = raw[pos:pos + 32].split(b"\0", 1)[0].decode()
        head = np.frombuffer(raw, dtype=np.int32, count=5, offset=pos + 32)
        ndim, dims = int(head[0]), [int(d) for d in head[1:1 + int(head[0])]]
        n = int(np.prod(dims)) if ndim else 1
        out[name] = np.frombuffer(raw, dtype=np.float32, count=n, offset=pos + 52).reshape(dims)
        pos += 52 + 4 * n
    return out


UNARY = [("sin", -10, 10), ("cos", -10, 10), ("tan", -1.4, 1.4), ("arcsin", -1, 1), ("arccos", -1, 1),
         ("arctan", -10, 10), ("sinh", -8, 8), ("cosh", -8, 8), ("tanh", -8, 8), ("arcsinh", -10, 10),
         ("arccosh", 1, 20), ("arctanh", -0.95, 0.95), ("exp", -10, 10), ("expm1", -5, 5), ("log", 0.01, 100),
         ("log2", 0.01, 100), ("log10", 0.01, 100), ("log1p", -0.9, 50), ("logb", 0.01, 100), ("sqrt", 0, 100),
         ("reciprocal", 0.1, 10), ("negate", -10, 10), ("positive", -10, 10), ("sign", -10, 10),
         ("floor", -10, 10), ("ceil", -10, 10), ("trunc", -10, 10), ("fix", -10, 10), ("rint", -10, 10),
         ("radians", -360, 360), ("degrees", -7, 7), ("sinc", -5, 5), ("rsqrt", 0.01, 100), ("exp2", -10, 10)]


@pytest.fixture(scope="module")
def records(tmp_path_factory):
    out = tmp_path_factory.mktemp("method_bodies") / "results.bin"
    proc = subprocess.run([str(EXE), str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert proc.returncode == 0, "method_bodies failed (%d): %s" % (proc.returncode, proc.stderr)
    return read_records(out)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(UNARY)), ids=[u[0] for u in UNARY])
def test_unary_method(i, records, oracle):
    name, lo, hi = UNARY[i]
    x = c_input(ROWS, COLS, i, lo, hi)
    got = records[name]
    ref = oracle.unary(name, x)
    if name in EXACT_UNARY:
        assert_bit_equal(got, ref, name)
    else:
        assert_close(got, ref, name)


@pytest.mark.gpu
def test_clip_round_arctan2_methods(records, oracle):
    x, y = c_input(ROWS, COLS, 101, -50, 50), c_input(ROWS, COLS, 102, -50, 50)
    assert_bit_equal(records["clip"], oracle.unary("clip", x, -7.25, 11.5), "clip")
    assert_bit_equal(records["round"], oracle.unary("round", x, 2.0), "round")
    assert_close(records["arctan2"], np.arctan2(x.astype(np.float64), y.astype(np.float64)), "arctan2")


@pytest.mark.gpu
@pytest.mark.parametrize("op", ["add", "subtract", "multiply", "divide", "mod"])
def test_binary_methods(op, records, oracle):
    x, y = c_input(ROWS, COLS, 101, -50, 50), c_input(ROWS, COLS, 102, -50, 50)
    row = c_input(0, COLS, 103, 0.5, 4)
    assert_bit_equal(records[op], oracle.binary(op, x, y), op)
    assert_bit_equal(records[op + "_row"], oracle.binary(op, x, row), op + " row")
    assert_bit_equal(records[op + "_scalar"], oracle.binary(op, x, np.float32(2.5)), op + " scalar")


@pytest.mark.gpu
def test_pow_reduce_matmul_methods(records, oracle):
    x = c_input(ROWS, COLS, 101, -50, 50)
    p, row = c_input(ROWS, COLS, 104, 0.25, 4), c_input(0, COLS, 103, 0.5, 4)
    assert_close(records["pow_row"], oracle.binary("pow", p, row), "pow row")
    x64 = x.astype(np.float64)
    mag = np.abs(x64).sum()
    assert abs(float(records["sum"]) - x64.sum()) <= REL_TOL * mag
    for axis in (0, 1):
        err = np.abs(records["sum_axis%d" % axis].astype(np.float64) - x64.sum(axis))
        assert (err <= REL_TOL * np.abs(x64).sum(axis)).all(), "sum axis %d" % axis
    assert float(records["min"]) == x.min() and float(records["max"]) == x.max()
    b = c_input(COLS, 129, 105, -1, 1)
    scale = np.abs(x64) @ np.abs(b.astype(np.float64))
    ref = oracle.matmul(x, b)
    for name in ("matmul", "dot"):
        assert records[name].shape == (ROWS, 129)
        assert (np.abs(records[name] - ref) / scale).max() <= REL_TOL, name
        assert (np.abs(records[name] - x64 @ b.astype(np.float64)) / scale).max() <= 1e-6, name
    assert_close(records["exp_cpu"], oracle.unary("exp", np.clip(x, -50, 50)), "exp -> cpu()")


@pytest.mark.gpu
def test_sharded_batched_matmul_method(records, oracle):
    """NDArray_ShardedBatchedMatmul from C, a world of one rank: kept / gathered / 3 overlapped pieces are the same
    bits, and every matrix equals the oracle's loop of 2-D matmuls (linalg.c:239-242 rejects ndim > 2: the loop IS
    the reference form)."""
    a = c_input(6 * 33, 47, 106, -1, 1).reshape(6, 33, 47)
    b = c_input(6 * 47, 29, 107, -1, 1).reshape(6, 47, 29)
    keep = records["sharded_keep"]
    assert keep.shape == (6, 33, 29)
    for name in ("sharded_gather", "sharded_overlap3", "sharded_overlap_auto"):
        assert_bit_equal(records[name], keep, name)
    for i in range(6):
        ref = oracle.matmul(a[i], b[i])
        scale = np.abs(a[i]).astype(np.float64) @ np.abs(b[i]).astype(np.float64)
        assert (np.abs(keep[i] - ref) / scale).max() <= REL_TOL
        assert (np.abs(keep[i] - a[i].astype(np.float64) @ b[i].astype(np.float64)) / scale).max() <= 1e-6


def test_method_bodies_builds_and_refuses_to_run_without_a_device(tmp_path):
    """CPU tier: the program exists (the build compiled the reference's call expressions with -Werror)
    and, with no GPU, stops at gpu() with the reference's message instead of computing anywhere else."""
    import torch
    assert EXE.exists()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the -m gpu tests run the program")
    proc = subprocess.run([str(EXE), str(tmp_path / "o.bin")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert proc.returncode == 1
    assert "gpu() failed" in proc.stderr


# The unary PHP_METHODs whose device branch is `rtn = NDArrayMathGPU_ElementWise(nda, cuda_float_<name>);` in the reference's
# numpower.c.  Data: the call sites a drop-in has to serve (abs goes through NDArray_Abs, rsqrt / exp2 have no working GPU
# branch there).  test_method_table_matches_the_reference_call_sites re-derives it from the reference when that is present.
REFERENCE_UNARY_CALL_SITES = sorted(
    "arccos arccosh arcsin arcsinh arctan arctanh ceil cos cosh degrees exp expm1 fix floor log log10 log1p log2 logb negate "
    "positive radians reciprocal rint sign sin sinc sinh sqrt tan tanh trunc".split())


def _method_table_names():
    import re
    src = (ROOT / "ext" / "method_bodies.c").read_text()
    table = src[src.index("static const UnaryMethod kUnary[]"):]
    table = table[:table.index("};")]
    return sorted(re.findall(r'\{"(\w+)",\s*cuda_float_(\w+),', table))


# the two unary methods tools/apply_with_hip.py re-points at entry points cuda_math.h does not have (hip_math.h)
PATCHED_UNARY_CALL_SITES = ["exp2", "rsqrt"]


def test_method_table_covers_every_unary_call_site():
    """CPU tier: ext/method_bodies.c hands the driver the same cuda_float_* pointer under every name the reference does
    (+ the two methods the --with-hip patch re-points: rsqrt, exp2)."""
    pairs = _method_table_names()
    assert all(label == sym for label, sym in pairs)
    assert [label for label, _ in pairs] == sorted(REFERENCE_UNARY_CALL_SITES + PATCHED_UNARY_CALL_SITES)
    assert [u[0] for u in sorted(UNARY)] == sorted(REFERENCE_UNARY_CALL_SITES + PATCHED_UNARY_CALL_SITES)   # and the GPU test checks every one of them


def test_method_table_matches_the_reference_call_sites():
    import re
    ref = Path("/root/reference/numpower.c")
    if not ref.exists():
        pytest.skip("reference tree not present on this box")
    names = set(re.findall(r"NDArrayMathGPU_ElementWise\(\s*nda\s*,\s*cuda_float_(\w+)\s*\)", ref.read_text(errors="ignore")))
    assert sorted(names) == REFERENCE_UNARY_CALL_SITES
