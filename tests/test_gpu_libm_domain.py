"""Whole-domain parity of the libm-class unary ops against the oracle (glibc through the reference's
float_* kernels, double_math.c:27-198) — VERDICT r01 "What's weak" #2 / next-round item 3.

test_gpu_parity.py keeps to comfortable domains ([-10, 10] for exp and trig, [1e-3, 1e3] for log); the
reference has no such restriction, and device libm (ocml) and glibc differ most where range
reduction, overflow / underflow and denormals come in.  Per op, sweeps that cover those regions:

  sin cos tan arctan tanh      |x| log-uniform over [1e-38, 1e38], both signs
  exp exp2 expm1               +-[80, 90] and +-[100, 130] (in ln units for exp / expm1, scaled by
                               1 / ln 2 for exp2): the overflow threshold, the denormal-output band,
                               underflow to zero; plus a dense sweep of the threshold neighbourhood
  log log2 log10               denormal inputs, [1e-38, 1e38], 1 +- k ulp
  log1p                        denormals, +-[1e-38, 1e-3], -1 + k ulp, up to 1e38
  arcsin arccos                +-(1 - k ulp), +-[1e-38, 1]
  sinc                         |x| log-uniform over [1e-38, 1e38]

Bar: the non-finite pattern (NaN / +inf / -inf positions) matches the oracle's bit for bit; every finite
value is within 1e-5 relative (north_star) — or within one denormal spacing (2^-149) when the result
itself is denormal, where one ulp is already a large relative step for both libraries.
"""
import numpy as np
import pytest

from numpower_amd import synth

pytestmark = pytest.mark.gpu

N = 120_000
ULP_DENORM = 2.0 ** -149
LN2 = float(np.log(2.0))


def log_sweep(lo, hi, n, seed):
    u = synth.uniform((n,), seed, 0.0, 1.0).astype(np.float64)
    return np.exp(np.log(lo) + u * (np.log(hi) - np.log(lo))).astype(np.float32)


def lin_sweep(lo, hi, n, seed):
    return synth.uniform((n,), seed, lo, hi)


def ulps_from(x0, ks):
    """x0 stepped by k float32 ulps, k in ks (may be negative)."""
    base = np.full(len(ks), x0, dtype=np.float32).view(np.int32)
    return (base + np.asarray(ks, dtype=np.int32)).view(np.float32)


DENORMALS = np.concatenate([np.arange(1, 2049, dtype=np.int32), np.arange(0x007FF800, 0x00800000, dtype=np.int32),
                            (synth.uniform((20_000,), 77, 1.0, float(0x007FFFFF))).astype(np.int32)]).view(np.float32)
SPECIALS = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 3.4028235e38, -3.4028235e38, 1.1754944e-38,
                     -1.1754944e-38, 1e-45, -1e-45, 0.5, -0.5, 2.0, -2.0], dtype=np.float32)


def both_signs(parts):
    return list(parts) + [-p for p in parts]


def sweeps(op):
    if op in ("sin", "cos", "tan", "arctan", "tanh", "sinc"):
        return both_signs([log_sweep(1e-38, 1e-3, N, 1), log_sweep(1e-3, 10.0, N, 2), log_sweep(10.0, 1e5, N, 3),
                           log_sweep(1e5, 1e12, N, 4), log_sweep(1e12, 1e30, N, 5), log_sweep(1e30, 3e38, N, 6),
                           DENORMALS])
    if op in ("exp", "expm1", "exp2"):
        s = 1.0 / LN2 if op == "exp2" else 1.0
        parts = both_signs([lin_sweep(80.0 * s, 90.0 * s, N, 1), lin_sweep(100.0 * s, 130.0 * s, N, 2),
                            lin_sweep(0.0, 80.0 * s, N, 3), log_sweep(1e-38, 1e-3, N, 4), log_sweep(130.0 * s, 3e38, N // 4, 5)])
        # the thresholds themselves, ulp by ulp: overflow (88.7228 / 128), first denormal output
        # (-87.3365 / -126), last non-zero output (-103.972 / -149), expm1's saturation at -1 (~ -17.3)
        for t in (88.72284 * s if op != "exp2" else 128.0, -87.33655 * s if op != "exp2" else -126.0,
                  -103.97208 * s if op != "exp2" else -149.0, -17.32868, -16.63553 * s):
            parts.append(ulps_from(np.float32(t), np.arange(-3000, 3001)))
        parts.append(DENORMALS)
        parts.append(-DENORMALS)
        return parts
    if op in ("log", "log2", "log10"):
        return [DENORMALS, log_sweep(1e-38, 1e-3, N, 1), log_sweep(1e-3, 1e3, N, 2), log_sweep(1e3, 3e38, N, 3),
                ulps_from(np.float32(1.0), np.arange(-20000, 20001)), lin_sweep(0.5, 2.0, N, 4),
                -log_sweep(1e-38, 1e3, 1000, 5)]            # negative inputs: NaN on both sides
    if op == "log1p":
        return both_signs([DENORMALS, log_sweep(1e-38, 1e-3, N, 1)]) + [
            log_sweep(1e-3, 1.0, N, 2), log_sweep(1.0, 3e38, N, 3), -log_sweep(1e-3, 0.999, N, 4),
            ulps_from(np.float32(-1.0), -np.arange(0, 20001)),       # -1, then -1 + k ulp (towards zero)
            ulps_from(np.float32(-1.0), np.arange(1, 2001))]         # below -1: NaN
    if op in ("arcsin", "arccos"):
        return both_signs([ulps_from(np.float32(1.0), -np.arange(0, 40001)), log_sweep(1e-38, 1e-3, N, 1),
                           log_sweep(1e-3, 1.0, N, 2), lin_sweep(0.9, 1.0, N, 3), DENORMALS])
    raise KeyError(op)


OPS = ["sin", "cos", "tan", "arctan", "tanh", "sinc", "exp", "exp2", "expm1", "log", "log2", "log10", "log1p",
       "arcsin", "arccos"]


def compare(op, x, got, ref):
    """-> None or a failure description"""
    got = np.asarray(got, dtype=np.float32)
    ref = np.asarray(ref, dtype=np.float32)
    for what, fn in (("NaN", np.isnan), ("+inf", np.isposinf), ("-inf", np.isneginf)):
        bad = fn(got) != fn(ref)
        if bad.any():
            i = int(np.argmax(bad))
            return "%s: %s pattern differs at %d inputs, first x = %r (%s): got %r, oracle %r" % (
                op, what, int(bad.sum()), x[i], hex(int(x[i:i + 1].view(np.uint32)[0])), got[i], ref[i])
    fin = np.isfinite(ref)
    g, r = got[fin].astype(np.float64), ref[fin].astype(np.float64)
    err = np.abs(g - r)
    bound = np.where(np.abs(r) < 1.1754944e-38, np.maximum(1e-5 * np.abs(r), ULP_DENORM), 1e-5 * np.abs(r))
    bad = err > bound
    if bad.any():
        i = int(np.argmax(err / np.maximum(bound, 1e-300)))
        return "%s: %d of %d finite results off by more than 1e-5 relative; worst x = %r: got %r, oracle %r" % (
            op, int(bad.sum()), int(fin.sum()), x[fin][i], g[i], r[i])
    return None


@pytest.mark.parametrize("op", OPS)
def test_libm_whole_domain(op, hip, oracle):
    x = np.concatenate(sweeps(op) + [SPECIALS]).astype(np.float32)
    dx = hip.DeviceArray.from_host(x)
    got = hip.unary(op, dx).to_host()
    with np.errstate(all="ignore"):
        ref = oracle.unary(op, x)
    failure = compare(op, x, got, ref)
    assert failure is None, failure


def test_arctan2_and_pow_wide_domain(hip, oracle):
    """The two libm-class BINARY ops over 80 binades of both operands (arctan2: all four quadrants and
    wildly different magnitudes; pow: bases over [1e-30, 1e30] against exponents that reach overflow,
    underflow and denormal results)."""
    n = 300_000
    sign = lambda seed: np.where(synth.uniform((n,), seed, 0.0, 1.0) < 0.5, np.float32(-1), np.float32(1))
    y = log_sweep(1e-38, 1e38, n, 1) * sign(2)
    xx = log_sweep(1e-38, 1e38, n, 3) * sign(4)
    got = hip.binary("arctan2", hip.DeviceArray.from_host(y), "full", hip.DeviceArray.from_host(xx), "full", 1, n).to_host()
    ref = np.arctan2(y.astype(np.float64), xx.astype(np.float64)).astype(np.float32)
    failure = compare("arctan2", y, got.reshape(-1), ref)
    assert failure is None, failure
    base = log_sweep(1e-30, 1e30, n, 5)
    expo = synth.uniform((n,), 6, -8.0, 8.0)
    got = hip.binary("pow", hip.DeviceArray.from_host(base), "full", hip.DeviceArray.from_host(expo), "full", 1, n).to_host()
    with np.errstate(all="ignore"):
        ref = oracle.binary("pow", base, expo)
    failure = compare("pow", base, got.reshape(-1), ref.reshape(-1))
    assert failure is None, failure
