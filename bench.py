#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X hot path (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-extras]

A "step" is one pass of the hot path over one batch of synthetic input: a batch of 16 independent
nd::matmul products of two 4096 x 4096 fp32 matrices each (BASELINE config 2, the configuration the
metric is quoted on), one np_sgemm launch per product, 4 distinct operand sets.  (Until the last line
of round 6 a step was ONE product: W = 5 such steps are 5 ms of matrix work, the device needs ~50 ms
to raise the matrix cores' clock, and a timed region right behind them reported that ramp —
0.82-0.85 of the MFMA peak for a kernel that runs at 0.91.  bench_matmul's docstring; the figure of
the old step definition is still in the line: roofline.frac_launches_5_24; NP_BENCH_STEP_PRODUCTS=1
runs the old step.)
Inputs are resident in HBM before the timed region.  K steps are launched back to back between
a barrier + device sync on both sides; `value` = all ranks' FLOPs / max-over-ranks wall time.
With N > 1 every rank (one per GPU) multiplies its own independent matrices.  The ranks come
either from an external launcher (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE) or —
when `--gpus N` is given and RANK is NOT in the environment — from this script itself: it spawns
N copies of itself on 127.0.0.1 with those variables set, relays rank 0's JSON line and exits
with rank 0's status (spawn_ranks).  Every rank multiplies its own
independent matrices — the path shards over independent arrays with no data-path collective
("weak" scaling); BASELINE config 5 (batched matmul sharded over the ranks + one RCCL
all-gather) is measured separately and reported under "extras", twice: with torch.distributed's
all_gather_into_tensor and with the library's own np_allgather (RCCL behind the C ABI).  NP_COMM=abi
runs the whole multi-rank bench without importing torch (rendezvous, barrier, max over ranks and the
gather through np_comm_*).  NP_BENCH_SHARE_DEVICE=1 puts every spawned rank on device 0 — possible only over
a collective library that allows ranks to share a device (the tests' stand-in does, RCCL does not): the line then
says "valid": false; it is how the N > 1 path runs on a one-GPU box, not a measurement.

The JSON line also carries
  roofline      achieved vs peak for the dominant kernel (fp32 MFMA GEMM), from HIP events
                recorded on the kernel's own stream around the timed launches
  cpu_baseline  the oracle (CPU restatement of the reference: cblas_sgemm from the OpenBLAS found
                on this host) timed on a bounded sample of the same workload
  extras        the second half of the metric (elementwise add on 1e8 floats, GB/s) and the other
                BASELINE configs (exp/log, broadcast, axis-0 sum), each with its HBM roofline
                fraction and CPU baseline, plus a parity verdict against the oracle.
"""
from __future__ import annotations

import argparse
import ctypes as C
import datetime
import json
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from numpower_amd import device as D          # noqa: E402
from numpower_amd import synth                 # noqa: E402
from numpower_amd._lib import Timer, load      # noqa: E402
from numpower_amd._lib import check as lib_check   # noqa: E402

METRIC = "GFLOP/s nd::matmul 4096² fp32; GB/s elementwise add 10^8 fp32 @1 MI355X"
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: 256 CU x 256 FLOP/clk x 2.4 GHz
PEAK_HBM_GBPS = 8000.0          # HBM3E spec; ~6.3 TB/s is what a float4 copy reaches


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def pmc_traffic():
    """HBM bytes per launch measured with rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in their own
    runs, gfx950 correction applied — see the _note inside the file) and committed under profiles/;
    the newest round's file wins.  {} when none is there: `traffic` is then null."""
    files = sorted((ROOT / "profiles").glob("r*/pmc_traffic.json"))
    if not files:
        return {}, None
    try:
        return json.loads(files[-1].read_text()), str(files[-1].relative_to(ROOT))
    except Exception:
        return {}, None


class _stdout_to_devnull:
    """Temporarily point fd 1 at /dev/null and flush C stdio into it (library banners)."""

    def __enter__(self):
        sys.stdout.flush()
        self._libc = C.CDLL(None)
        self._libc.fflush(None)
        self._saved = os.dup(1)
        self._null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self._null, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        os.close(self._null)
        return False


DRYRUN = os.environ.get("NP_BENCH_DRYRUN") == "1"   # launcher / rendezvous logic only: gloo, no GPU, no kernels


class Dist:
    """torch.distributed plumbing (only imported when N > 1, or when NP_BENCH_FORCE_DIST=1 asks for
    the same code path on one GPU: world size 1, RCCL initialised, kernels on torch's stream).
    NP_BENCH_DRYRUN=1 swaps RCCL for gloo and never touches a device: it exists so that the
    self-launch path (spawn_ranks -> rendezvous -> barrier -> max over ranks -> one JSON line from
    rank 0) can be exercised on a box without a GPU (tests/test_bench_launcher_cpu.py)."""

    def __init__(self, n):
        self.n = n
        self.rank = 0
        self.local_rank = 0
        self.torch = None
        self.use_torch = n > 1 or os.environ.get("NP_BENCH_FORCE_DIST") == "1"
        # NP_COMM=abi: no torch anywhere — rendezvous, barrier, max-over-ranks and config 5's all-gather all go
        # through the library's own np_comm_* entry points (RCCL behind the C ABI: what a PHP host would call)
        self.abi = self.use_torch and os.environ.get("NP_COMM") == "abi" and not DRYRUN
        if self.abi:
            self.use_torch = False
            self.rank = int(os.environ.get("RANK", "0"))
            self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
            world = int(os.environ.get("WORLD_SIZE", str(n)))
            if world != n:
                raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (n, world))
            D.init(self.local_rank)
            from numpower_amd._lib import check
            # not MASTER_PORT itself: under torch.distributed.run the launcher's own store is listening there
            endpoint = "tcp://%s:%d" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29531")) + 29)
            with _stdout_to_devnull():      # RCCL's banner
                check(load().np_comm_init(self.rank, n, endpoint.encode()))
                check(load().np_comm_barrier())
            return
        if self.use_torch:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            self.rank = int(os.environ.get("RANK", "0"))
            self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", str(n))
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if int(os.environ["WORLD_SIZE"]) != n:
                raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s" % (n, os.environ["WORLD_SIZE"]))
            if DRYRUN:
                with _stdout_to_devnull():      # gloo announces its peers on stdout
                    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
                return
            torch.cuda.set_device(self.local_rank)
            # RCCL prints a version banner through C stdio on stdout when the communicator comes up;
            # stdout must carry exactly one JSON line, so the banner is flushed into /dev/null.
            with _stdout_to_devnull():
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank),
                                        timeout=datetime.timedelta(seconds=600))
                warm = torch.zeros(1, device="cuda")
                dist.all_reduce(warm)
                torch.cuda.synchronize()
            D.init(self.local_rank)
            # run our kernels on torch's current stream so that RCCL and torch see one order.  The
            # legacy default stream has handle 0, which np_set_stream reads as "library-owned
            # stream", so make a real stream current first.
            from numpower_amd._lib import check
            self.stream = torch.cuda.Stream(device=self.local_rank)
            torch.cuda.set_stream(self.stream)
            assert self.stream.cuda_stream != 0
            check(load().np_set_stream(self.stream.cuda_stream))
        elif not DRYRUN:
            D.init(0)

    def barrier_sync(self):
        if self.abi:
            from numpower_amd._lib import check
            check(load().np_comm_barrier())
            D.sync()
            return
        if self.use_torch:
            self.dist.barrier()
            if not DRYRUN:
                self.torch.cuda.synchronize()
        if not DRYRUN:
            D.sync()

    def max_over_ranks(self, x: float) -> float:
        if self.abi:
            from numpower_amd._lib import check
            m = C.c_float(0.0)
            check(load().np_comm_max(float(x), C.byref(m)))
            return float(m.value)
        if not self.use_torch:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cpu" if DRYRUN else "cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def describe(self):
        """Who took part and through what: {"ranks_seen", "collective"} for the JSON line (VERDICT r05 next #6a).  A collective:
        every rank calls it."""
        seen = int(self.max_over_ranks(float(self.rank))) + 1
        if DRYRUN:
            how = "gloo (dry run: no device, no RCCL)"
        elif self.abi or not self.use_torch:
            v = C.c_int(0)
            ok = load().np_comm_rccl_version(C.byref(v)) == 0
            how = ("np_comm_* (RCCL %d.%d.%d behind the C ABI)" % (v.value // 10000, v.value // 100 % 100, v.value % 100)) if ok else "np_comm_* (RCCL version unknown)"
            if ok and v.value >= 90000:     # no RCCL release: the loader was pointed at something else (LD_LIBRARY_PATH)
                how = "np_comm_* over a stand-in for librccl.so.1 (ncclGetVersion says %d: not RCCL)" % v.value
            if not self.abi:
                how = "none (one rank, no communicator); the library would load " + how
        else:
            try:
                how = "torch.distributed nccl backend = RCCL %s" % ".".join(str(x) for x in self.torch.cuda.nccl.version())
            except Exception as e:      # noqa: BLE001 — a version string must never take the line down
                how = "torch.distributed nccl backend (version unavailable: %r)" % (e,)
        return {"ranks_seen": seen, "collective": how}

    def close(self):
        if self.abi:
            with _stdout_to_devnull():
                load().np_comm_destroy()
        if self.use_torch:
            with _stdout_to_devnull():
                self.dist.destroy_process_group()


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int, argv, timeout_s: float = 1500.0) -> int:
    """`python bench.py --gpus N` without an external launcher: start N copies of this script, one
    per GPU, with the variables torch.distributed.run would set (rendezvous on 127.0.0.1, a free
    port), relay rank 0's stdout (the one JSON line) to ours, leave every rank's stderr attached,
    and return rank 0's exit status.  A rank that dies takes the job down: the survivors are
    terminated (by the exact PIDs started here) instead of sitting in the rendezvous timeout."""
    import subprocess
    port = _free_port()
    procs = []
    share = os.environ.get("NP_BENCH_SHARE_DEVICE") == "1"     # every rank on device 0 (see main(): a code-path run)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if share else str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + timeout_s
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            failed = (-1, "timeout after %.0f s" % timeout_s)
            break
        time.sleep(0.05)
    if failed is not None:
        log("bench.py: rank %s failed (%s); stopping the other ranks" % failed)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    if out0 and out0[0]:
        sys.stdout.write(out0[0].decode())
        sys.stdout.flush()
    if failed is not None:
        return failed[1] if isinstance(failed[1], int) and failed[1] > 0 else 1
    return 0


def timed(dist: Dist, fn, steps: int, warmup: int):
    """-> (wall seconds max over ranks, HIP-event ms on this rank's stream) for `steps` calls."""
    for _ in range(warmup):
        fn()
    dist.barrier_sync()
    ev = None if DRYRUN else Timer()
    t0 = time.perf_counter()
    if ev:
        ev.start()
    for _ in range(steps):
        fn()
    if ev:
        ev.stop()
    dist.barrier_sync()
    wall = time.perf_counter() - t0
    return dist.max_over_ranks(wall), (ev.elapsed_ms() if ev else wall * 1e3)


def cpu_time(fn, budget_s=8.0, max_iters=5):
    """Median time of a CPU callable over a bounded number of runs."""
    fn()   # warm (page in, thread pool)
    ts = []
    t_start = time.perf_counter()
    while len(ts) < max_iters and (time.perf_counter() - t_start < budget_s or not ts):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), len(ts)


# ---------------------------------------------------------------------------------------------

STEP_PRODUCTS = max(1, int(os.environ.get("NP_BENCH_STEP_PRODUCTS", "16")))
PRODUCT_SETS = 4


def bench_matmul(dist: Dist, steps, warmup, do_cpu):
    """The headline.  A STEP is one batch of STEP_PRODUCTS independent 4096^3 products, one np_sgemm launch each, cycling
    over PRODUCT_SETS distinct (A_i, B_i, C_i): A_i = A * 2^-i (exact), B_i a copy of B — so every C_i must equal
    C_0 * 2^-i BIT FOR BIT, which is checked over all elements next to the fp64 check of C_0.
    Why a batch per step: the device raises the matrix cores' clock over the first ~50 ms of matrix work
    (profiles/r06/headline_ramp_from_kernel_trace.txt: 1.10-1.21 ms per product at launch 0-4, 0.945 from launch 50 on), and
    W warm-up steps of ONE 1 ms product each (the steps of rounds 1-6's earlier lines) end long before that; a timed region
    behind them reported the ramp, 0.82-0.85, for a kernel that runs at 0.91.  The W warm-up steps are launched with an event
    pair around every product (they are untimed anyway): `ramp` in the result is what those launches took, and
    `frac_launches_5_24` the figure the one-product steps of the earlier lines would have given in this very run."""
    n = 4096
    seed = 3 + 100 * dist.rank
    A = synth.uniform((n, n), seed, -1.0, 1.0)
    B = synth.uniform((n, n), seed + 1, -1.0, 1.0)
    dA, dB = D.DeviceArray.from_host(A), D.DeviceArray.from_host(B)
    sets = [(dA, dB, D.DeviceArray((n, n)))]
    for i in range(1, PRODUCT_SETS):
        scale = D.DeviceArray.from_host(np.float32([2.0 ** -i]))
        dAi = D.binary("multiply", dA, "full", scale, "scalar", 1, n * n, out=D.DeviceArray((n, n)))
        dBi = D.DeviceArray((n, n))
        lib_check(load().np_memcpy_d2d(dBi.ptr, dB.ptr, 4 * n * n))
        sets.append((dAi, dBi, D.DeviceArray((n, n))))
        D.sync()
        scale.free()
    flop = 2.0 * n ** 3
    P = STEP_PRODUCTS
    ramp_timers = []

    def product(j):
        a, b, c = sets[j % PRODUCT_SETS]
        D.sgemm(a, b, out=c)

    def warm_step():
        for j in range(P):
            t = Timer()
            t.start()
            product(j)
            t.stop()
            ramp_timers.append(t)

    def step():
        for j in range(P):
            product(j)

    for _ in range(warmup):      # the W untimed warm-up steps (timed()'s own warm-up count is 0 below)
        warm_step()
    wall, ev_ms = timed(dist, step, steps, 0)
    ramp_ms = [t.elapsed_ms() for t in ramp_timers]
    ramp = None
    if ramp_ms:
        tf = lambda xs: flop / (sum(xs) / len(xs)) / 1e9 if xs else None      # noqa: E731
        ramp = {"first_launch_ms": ramp_ms[0], "launches_5_24_ms": (sum(ramp_ms[5:25]) / len(ramp_ms[5:25])) if len(ramp_ms) >= 25 else None,
                "last_16_warmup_launches_ms": sum(ramp_ms[-16:]) / len(ramp_ms[-16:]), "warmup_launches": len(ramp_ms),
                "TFLOPs_launches_5_24": tf(ramp_ms[5:25]) if len(ramp_ms) >= 25 else None,
                "note": "every product of the W warm-up steps bracketed by its own event pair, from a cold device; launches 5-24 are "
                        "what a timed region of K = 20 one-product steps behind W = 5 covers"}
    # the spread behind the average (941-1228 us inside one run): `steps` launches
    # again, each bracketed by its own event pair on the kernel's stream (after, not inside, the timed region)
    timers = [Timer() for _ in range(steps)]
    for j, t in enumerate(timers):
        t.start()
        product(j)
        t.stop()
    per = sorted(t.elapsed_ms() for t in timers)
    launch_ms = {"min": per[0], "median": per[len(per) // 2], "max": per[-1],
                 "note": "%d individually timed launches after the timed region" % steps}
    # parity: sampled rows of C_0 against an fp64 product (1e-5 relative to |A|.|B|); every other product of the batch
    # against C_0 through linearity: A_i = A * 2^-i exactly, so C_i = C_0 * 2^-i bit for bit
    rows = [0, 1, 1234, 4095]
    C0 = sets[0][2].to_host()
    got = C0[rows].astype(np.float64)
    ref = A[rows].astype(np.float64) @ B.astype(np.float64)
    scale = np.abs(A[rows]).astype(np.float64) @ np.abs(B).astype(np.float64)
    err = float((np.abs(got - ref) / scale).max())
    batch_ok = True
    for i in range(1, PRODUCT_SETS):
        Ci = sets[i][2].to_host()
        batch_ok = batch_ok and bool((Ci.view(np.uint32) == (C0 * np.float32(2.0 ** -i)).view(np.uint32)).all())
        del Ci
    del C0
    out = {
        "wall_s": wall, "event_ms": ev_ms, "flop_per_step": flop * P, "flop_per_launch": flop, "products_per_step": P,
        "launch_ms": launch_ms, "ramp": ramp,
        "parity_max_norm_err_vs_fp64": err, "parity_ok": bool(err <= 1e-6) and batch_ok, "batch_linearity_bit_exact": batch_ok,
    }
    if do_cpu:
        from oracle import oracle
        info = oracle.use_openblas()
        Cc = np.empty((n, n), dtype=np.float32)
        t, iters = cpu_time(lambda: Cc.__setitem__(slice(None), oracle.matmul(A, B)), budget_s=10.0, max_iters=3)
        out["cpu"] = {"value": flop / t / 1e9, "unit": "GFLOP/s", "cores": info.get("threads", 1),
                      "kind": "port",
                      "sample": "%d x full 4096^2 sgemm via %s" % (iters, info.get("kind")),
                      "blas": info}
        cerr = float((np.abs(Cc[rows].astype(np.float64) - ref) / scale).max())
        out["gpu_vs_cpu_max_norm_err"] = float((np.abs(got - Cc[rows]) / scale).max())
        out["cpu_vs_fp64_max_norm_err"] = cerr
    for trio in sets:
        for d in trio:
            d.free()
    return out


def hbm_case(name, bytes_per_launch, launch, steps, warmup, dist):
    """`ms_per_launch` / `frac` = the timed region (K launches back to back behind W warm-ups).  The same K launches
    again, each bracketed by its own event pair (after, not inside, the timed region), give the spread behind that
    average (`launch_ms`: min / median / max; an individually bracketed launch includes its launch gap, so its median
    sits a few per cent ABOVE the back-to-back average for these 70-400 us kernels — unlike the GEMM, the HBM-bound
    kernels show no ramp: profiles/r03/bench_r03.json)."""
    wall, ev_ms = timed(dist, launch, steps, warmup)
    per_launch_ms = ev_ms / steps
    gbps = bytes_per_launch / per_launch_ms / 1e6
    out = {"name": name, "ms_per_launch": per_launch_ms, "GBps": gbps,
           "algorithmic_bytes": bytes_per_launch,
           "roofline": {"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": gbps / PEAK_HBM_GBPS, "traffic": None}}
    timers = [Timer() for _ in range(steps)]
    for t in timers:
        t.start()
        launch()
        t.stop()
    per = sorted(t.elapsed_ms() for t in timers)
    out["launch_ms"] = {"min": per[0], "median": per[len(per) // 2], "max": per[-1]}
    out["roofline"]["frac_median_launch"] = bytes_per_launch / per[len(per) // 2] / 1e6 / PEAK_HBM_GBPS
    return out


def bench_c1(dist: Dist, steps, warmup):
    """BASELINE config 1 ("parity + CPU time"): nd::add and nd::sum on 1000 x 1000 fp32.  The
    reference runs it on the CPU (arithmetics.c:160-278 AVX2 add, :58-71 sequential sum): the oracle's restatement is timed
    here as that path; beside it the device kernels on resident buffers, and the whole `$a->gpu()` -> op -> `->cpu()`
    round trip a PHP caller pays for an array this small."""
    from oracle import oracle
    from numpower_amd.ndarray import NDArray
    R = 1000
    a = synth.uniform((R, R), 1, 0.0, 1.0)
    b = synth.uniform((R, R), 2, 0.0, 1.0)
    t_add, _ = cpu_time(lambda: oracle.binary("add", a, b), budget_s=1.5, max_iters=21)
    t_sum, _ = cpu_time(lambda: oracle.reduce_all("sum", a), budget_s=1.5, max_iters=21)
    da, db, do = D.DeviceArray.from_host(a), D.DeviceArray.from_host(b), D.DeviceArray((R, R))
    add = hbm_case("add 1000x1000 (C1)", 12.0 * R * R, lambda: D.binary("add", da, "full", db, "full", 1, R * R, out=do),
                   steps, warmup, dist)
    got_add = do.to_host()
    # the call as a C / PHP host makes it — np_reduce_all(op, ptr, n, &value) with nothing allocated per call (the Python
    # convenience wrapper D.reduce_all builds a ctypes float and a reference every time: ~1-2 us that are not the library's)
    from numpower_amd._lib import REDUCE_OPS
    lib_c1, val_c1 = load(), C.c_float()
    ref_c1, op_c1, ptr_c1 = C.byref(val_c1), REDUCE_OPS["sum"], da.ptr
    call_sum = lib_c1.np_reduce_all

    def sum_call():
        if call_sum(op_c1, ptr_c1, R * R, ref_c1) != 0:
            raise RuntimeError(lib_c1.np_last_error().decode())

    ssum = hbm_case("sum 1000x1000 (C1)", 4.0 * R * R, sum_call, steps, warmup, dist)
    got_sum = D.reduce_all("sum", da)
    want_add = oracle.binary("add", a, b)
    want64 = float(a.astype(np.float64).sum())
    ha, hb = NDArray.array(a), NDArray.array(b)
    (ha.gpu() + hb.gpu()).cpu()             # once untimed: first-use costs are not the round trip
    t0 = time.perf_counter()
    res = (ha.gpu() + hb.gpu()).cpu()
    e2e_add = time.perf_counter() - t0
    t0 = time.perf_counter()
    NDArray.sum(ha.gpu())
    e2e_sum = time.perf_counter() - t0
    ok = bool((got_add.view(np.uint32) == want_add.view(np.uint32)).all() and (res.numpy() == want_add).all()
              and abs(got_sum - want64) <= 1e-5 * want64)
    for d in (da, db, do):
        d.free()
    return {"workload": "nd::add + nd::sum 1000x1000 fp32 (BASELINE config 1)",
            "cpu_add_ms": t_add * 1e3, "cpu_sum_ms": t_sum * 1e3, "cpu_kind": "port (oracle restatement, AVX2, 1 thread)",
            "gpu_add_kernel_us": add["launch_ms"]["median"] * 1e3, "gpu_sum_call_us": ssum["launch_ms"]["median"] * 1e3,
            "gpu_add_back_to_back_us": add["ms_per_launch"] * 1e3, "gpu_sum_back_to_back_us": ssum["ms_per_launch"] * 1e3,
            "end_to_end_gpu_add_cpu_ms": e2e_add * 1e3, "end_to_end_gpu_sum_ms": e2e_sum * 1e3,
            "sum_rel_err_vs_fp64": abs(got_sum - want64) / want64,
            "cpu_sum_rel_err_vs_fp64": abs(float(oracle.reduce_all("sum", a)) - want64) / want64, "parity_ok": ok}


def bench_mid(dist: Dist, steps, warmup):
    """What a PHP caller multiplies and reshapes more often than 4096^3: nd::matmul at 768^3, 1000^3
    and 1024^3 (round 4: LDS-DMA tiles of 64 x 64, np_sgemm.hip sgemm_dmas_kernel), a transpose whose rows are off the
    128-byte line grid, an NHWC-like permute.  Each: K launches back to back, parity against fp64 / numpy."""
    from numpower_amd._lib import check
    lib = load()
    out = {}
    # ... and (round 5) a deep-K product of a few tiles, 100 x 100 x 100000: K-chunks on the k-quartered tiles
    for (m, n, k) in ((768,) * 3, (1000,) * 3, (1024,) * 3, (100, 100, 100000)):
        A = synth.uniform((m, k), 31, -1.0, 1.0)
        B = synth.uniform((k, n), 32, -1.0, 1.0)
        dA, dB, dC = D.DeviceArray.from_host(A), D.DeviceArray.from_host(B), D.DeviceArray((m, n))
        # a launch is ~20 us: K of them after an upload are over before the clock has come up (1024^3: 24.5 us for the first
        # hundred, 22.7 a moment later, profiles/r04/gemm_mid_sweep_forced2.log).  Both are reported: `us_first_launches` (what a
        # caller's first product costs) and `us_per_launch` after 0.25 s of the same product (what the 500th costs).
        _, ev_ms = timed(dist, lambda: D.sgemm(dA, dB, out=dC), steps * 2, warmup * 2)
        us_first = ev_ms / (steps * 2) * 1e3
        if not DRYRUN:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.25:
                for _ in range(200):
                    D.sgemm(dA, dB, out=dC)
                D.sync()
        _, ev_ms = timed(dist, lambda: D.sgemm(dA, dB, out=dC), steps * 10, warmup)
        us = ev_ms / (steps * 10) * 1e3
        got = dC.to_host().astype(np.float64)
        ref = A.astype(np.float64) @ B.astype(np.float64)
        scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
        err = float((np.abs(got - ref) / scale).max())
        tf = 2.0 * m * n * k / us / 1e6
        out["matmul_%d" % n if m == n == k else "matmul_%dx%dx%d" % (m, n, k)] = {"us_per_launch": us, "us_first_launches": us_first, "TFLOPs": tf, "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_FP32_MFMA_TFLOPS,
                                                                              "unit": "TFLOP/s", "frac": tf / PEAK_FP32_MFMA_TFLOPS, "traffic": None},
                                "parity_max_norm_err_vs_fp64": err, "parity_ok": bool(err <= 1e-6)}
        for d in (dA, dB, dC):
            d.free()
    rows, cols = 8191, 8193
    X = synth.uniform((rows, cols), 33, -1.0, 1.0)
    dX, dT = D.DeviceArray.from_host(X), D.DeviceArray((cols, rows))
    r = hbm_case("transpose 8191x8193", 8.0 * rows * cols, lambda: check(lib.np_transpose2d(dX.ptr, dT.ptr, 1, rows, cols)), steps, warmup, dist)
    r["parity_ok"] = bool((dT.to_host() == X.T).all())
    out["transpose_8191x8193"] = r
    shape, perm = (60, 128, 1024, 8), (0, 2, 1, 3)      # (fits inside X's buffer)
    n = int(np.prod(shape))
    sh = (C.c_int * 4)(*shape)
    pm = (C.c_int * 4)(*perm)
    dP = D.DeviceArray((n,))
    r = hbm_case("permute (60,128,1024,8) (0,2,1,3)", 8.0 * n, lambda: check(lib.np_permute(dX.ptr, dP.ptr, 4, sh, pm)), steps, warmup, dist)
    r["parity_ok"] = bool((dP.to_host().reshape(-1) == np.ascontiguousarray(X.reshape(-1)[:n].reshape(shape).transpose(perm)).reshape(-1)).all())
    out["permute_nhwc_like"] = r
    for d in (dX, dT, dP):
        d.free()
    return out


def bench_extras(dist: Dist, steps, warmup):
    """The HBM-bound configs (C3a/b/c, C4) at BASELINE.json's sizes, N = 1 only."""
    from oracle import oracle
    ex = {}
    N = 100_000_000
    a = synth.uniform((N,), 5, 0.0, 1.0)
    b = synth.uniform((N,), 6, 0.0, 1.0)
    da, db, do = D.DeviceArray.from_host(a), D.DeviceArray.from_host(b), D.DeviceArray((N,))
    # What THIS box's HBM gives the three access mixes, measured in the same run through the library's own
    # streaming kernels on the same buffers ("measure achievable with a copy kernel and
    # report both"): a float4 copy (np_memcpy_d2d, 8 B/elem), a read-only stream (np_reduce_all_dev: 4 B/elem)
    # and a write-only stream (np_fill: 4 B/elem).  `frac` stays against the 8 TB/s spec; frac_of_copy says how
    # the add kernel (2 reads + 1 write) compares with the copy (1 read + 1 write) on this box.
    lib0 = load()
    from numpower_amd._lib import check as _check
    sink = D.DeviceArray((4,))
    ceil = {}
    for key, nbytes, fn in (
            ("copy", 8.0 * N, lambda: _check(lib0.np_memcpy_d2d(do.ptr, da.ptr, 4 * N))),
            ("read", 4.0 * N, lambda: _check(lib0.np_reduce_all_dev(0, da.ptr, N, sink.ptr))),
            ("write", 4.0 * N, lambda: _check(lib0.np_fill(do.ptr, 1.5, N)))):
        c = hbm_case(key, nbytes, fn, steps, warmup, dist)
        ceil[key + "_GBps"] = c["GBps"]
    sink.free()
    r = hbm_case("add 1e8 fp32 (C3a)", 12.0 * N, lambda: D.binary("add", da, "full", db, "full", 1, N, out=do),
                 steps, warmup, dist)
    if os.environ.get("NP_BENCH_DIAG") == "1":
        again = [hbm_case("add", 12.0 * N, lambda: D.binary("add", da, "full", db, "full", 1, N, out=do), steps, warmup, dist)["GBps"]
                 for _ in range(4)]
        log("[diag] add 1e8 (bench buffers, random data) %s then %s GB/s  a=%#x b=%#x o=%#x" % (
            "%.0f" % r["GBps"], " ".join("%.0f" % x for x in again), da.ptr, db.ptr, do.ptr))
    ceil["frac_of_copy"] = r["GBps"] / ceil["copy_GBps"]
    ceil["note"] = ("same run, same buffers: np_memcpy_d2d (float4 copy), np_reduce_all_dev sum (read-only), "
                    "np_fill (write-only); MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy")
    r["roofline"]["ceiling"] = ceil
    got = do.to_host()
    t, it = cpu_time(lambda: oracle.binary("add", a, b), budget_s=6.0, max_iters=3)
    ref = oracle.binary("add", a, b)
    r["parity_ok"] = bool((got.view(np.uint32) == ref.view(np.uint32)).all())
    r["cpu_baseline"] = {"value": 12.0 * N / t / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
                         "sample": "%d x full 1e8-element NDArray_Add_Float restatement (AVX2, 1 thread)" % it}
    ex["add_1e8"] = r
    del ref, got
    # end to end through the boundary's placement calls, once: gpu() x2 -> add -> cpu() on pageable
    # host buffers (NDArray_ToGPU / NDArray_ToCPU); PCIe-bound, reported beside the kernel number
    from numpower_amd.ndarray import NDArray
    ha, hb = NDArray.array(a), NDArray.array(b)
    t0 = time.perf_counter()
    res = (ha.gpu() + hb.gpu()).cpu()
    e2e = time.perf_counter() - t0
    r["end_to_end_gpu_add_cpu"] = {"seconds": e2e, "GBps_over_pcie": 12.0 * N / e2e / 1e9,
                                   "note": "2 x H2D 0.4 GB + kernel + D2H 0.4 GB, pageable memory"}
    del ha, hb, res
    # comparison elementwise (same kernel template, 12 B/elem)
    r = hbm_case("greater 1e8 fp32 (logic.c, §8f)", 12.0 * N,
                 lambda: D.binary("greater", da, "full", db, "full", 1, N, out=do), steps, warmup, dist)
    r["parity_ok"] = bool((do.to_host().reshape(-1) == (a > b).astype(np.float32)).all())
    ex["greater_1e8"] = r
    # §8(f) row 1: NDArray_ArrayEqual / AllClose as one streaming reduction, 8 B/elem, result on the host
    flag = C.c_int(0)
    r = hbm_case("allclose(a, a') 1e8 fp32 (logic.c:719-772, §8f)", 8.0 * N,
                 lambda: _check(lib0.np_count_mismatch(1, da.ptr, db.ptr, N, 1e-5, 1e-8, C.byref(flag))),
                 steps, warmup, dist)
    r["parity_ok"] = bool(flag.value == 1)      # a, b are independent uniforms: not close
    r["note"] = "includes the 4-byte D2H of the verdict per call"
    ex["allclose_1e8"] = r
    # pow on the path's six binary ops is the one with real arithmetic: fp64 2^(y log2 x) per element
    r = hbm_case("pow 1e8 fp32 (arithmetics.c:825)", 12.0 * N,
                 lambda: D.binary("pow", da, "full", db, "full", 1, N, out=do), steps, warmup, dist)
    sample = slice(0, 2_000_000)
    with np.errstate(all="ignore"):
        ref = np.power(a[sample].astype(np.float64), b[sample].astype(np.float64))
    gotp = do.to_host().reshape(-1)[sample].astype(np.float64)
    r["parity_max_rel_err_vs_fp64"] = float((np.abs(gotp - ref) / np.maximum(ref, 1e-300)).max())
    r["parity_ok"] = bool(r["parity_max_rel_err_vs_fp64"] <= 1e-5)
    ex["pow_1e8"] = r
    # median of 1e8 floats: np_order_stat.  A selection has to read every element once: 4 B/elem is the algorithmic
    # figure.  The bracket path (sample -> one filtering pass -> radix passes over ~1 % copied keys) reads ~4.1 B/elem;
    # the plain three-pass radix select it falls back to (and round 1 shipped) reads 12 B/elem and is timed beside it.
    two_f = (C.c_float * 2)()
    r = hbm_case("median 1e8 fp32 (bracketed radix select; arithmetics.c:111-158 sorts a host copy)", 4.0 * N,
                 lambda: _check(lib0.np_order_stat(da.ptr, N, N // 2 - 1, two_f)), steps, warmup, dist)
    part = np.partition(a, [N // 2 - 1, N // 2])
    r["parity_ok"] = bool(two_f[0] == part[N // 2 - 1] and two_f[1] == part[N // 2])
    path = C.c_int(-1)
    _check(lib0.np_select_last_path(C.byref(path)))
    r["bracket_path_taken"] = bool(path.value == 1)
    _check(lib0.np_select_set_variant(0))
    r3 = hbm_case("three-pass radix select", 12.0 * N,
                  lambda: _check(lib0.np_order_stat(da.ptr, N, N // 2 - 1, two_f)), steps, warmup, dist)
    _check(lib0.np_select_set_variant(1))
    r["parity_ok"] = bool(r["parity_ok"] and two_f[0] == part[N // 2 - 1] and two_f[1] == part[N // 2])
    r["three_pass"] = {"ms_per_launch": r3["ms_per_launch"], "bytes_per_elem": 12.0, "GBps": r3["GBps"]}
    r["note"] = "includes the 8-byte D2H of the two order statistics per call; 4 B/elem = one read of the array"
    ex["median_1e8"] = r
    del part
    # exp(a) * b + 2 as ONE fused kernel (12 B/elem) vs three launches
    from numpower_amd._lib import BINARY_OPS, UNARY_OPS, FusedOp, check
    prog = (FusedOp * 3)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0),
                         FusedOp(1, BINARY_OPS["multiply"], 1, 0, 0, 0, 0, 0),      # flags 0: what numpower_amd/lazy.py emits
                         FusedOp(1, BINARY_OPS["add"], 2, 0, 0, 0, 0, 0))
    two = C.c_float(2.0)
    ptrs = (C.c_void_p * 3)(da.ptr, db.ptr, C.cast(C.pointer(two), C.c_void_p))
    kinds = (C.c_int * 3)(0, 0, 4)
    lib = load()
    r = hbm_case("exp(a)*b+2 fused, 1e8 (§8f row 4)", 12.0 * N,
                 lambda: check(lib.np_fused_chain(ptrs, kinds, 3, prog, 3, do.ptr, 1, N)), steps, warmup, dist)
    fused_out = do.to_host().reshape(-1)
    tmp, tmp2 = D.DeviceArray((N,)), D.DeviceArray((N,))
    stwo = D.DeviceArray.from_host(np.float32([2.0]))

    def unfused():   # what three PHP-level ops cost: three launches, two temporaries
        D.unary("exp", da, out=tmp)
        D.binary("multiply", tmp, "full", db, "full", 1, N, out=tmp2)
        D.binary("add", tmp2, "full", stwo, "scalar", 1, N, out=tmp)

    _, ev_ms = timed(dist, unfused, steps, warmup)
    r["unfused_ms_per_chain"] = ev_ms / steps
    r["speedup_vs_unfused"] = (ev_ms / steps) / r["ms_per_launch"]
    r["parity_ok"] = bool((fused_out.view(np.uint32) == tmp.to_host().reshape(-1).view(np.uint32)).all())
    ex["fused_chain_1e8"] = r
    # ... and with the reduction fused in as well: sum(exp(a) * b + 2) reads 8 B/elem, nothing written
    # The roofline figure is the kernel's: np_fused_chain_reduce_dev leaves the sum on the device, calls run back to back
    # like every other kernel here.  The host-result form (np_fused_chain_reduce: what nd::sum() of a lazy chain costs a
    # PHP caller, one host round trip per call) is timed beside it.
    out = C.c_float(0.0)
    dsum = D.DeviceArray((1,))
    r = hbm_case("sum(exp(a)*b+2) fused, 1e8 (§8f row 4)", 8.0 * N,
                 lambda: check(lib.np_fused_chain_reduce_dev(ptrs, kinds, 3, prog, 3, 0, 1, N, dsum.ptr)),
                 steps, warmup, dist)
    want = float(tmp.to_host().reshape(-1).astype(np.float64).sum())
    got_dev = float(dsum.to_host()[0])
    rh = hbm_case("host result", 8.0 * N,
                  lambda: check(lib.np_fused_chain_reduce(ptrs, kinds, 3, prog, 3, 0, 1, N, C.byref(out))),
                  steps, warmup, dist)
    r["host_result_call"] = {"ms_per_call": rh["ms_per_launch"], "GBps": rh["GBps"], "frac": rh["roofline"]["frac"]}
    r["parity_rel_err_vs_fp64"] = max(abs(out.value - want), abs(got_dev - want)) / abs(want)
    r["parity_ok"] = bool(r["parity_rel_err_vs_fp64"] <= 1e-5)
    dsum.free()
    ex["fused_chain_sum_1e8"] = r
    tmp.free()
    tmp2.free()

    for op, seed, lo, hi in (("exp", 7, -10.0, 10.0), ("log", 8, 1e-3, 1e3)):
        x = synth.uniform((N,), seed, lo, hi)
        check_n = 4_000_000
        dx = D.DeviceArray.from_host(x)
        r = hbm_case("%s 1e8 fp32 (C3b)" % op, 8.0 * N, lambda: D.unary(op, dx, out=do), steps, warmup, dist)
        got = do.to_host()[:check_n].astype(np.float64)
        t, it = cpu_time(lambda: oracle.unary(op, x[:20_000_000]), budget_s=5.0, max_iters=2)
        ref = oracle.unary(op, x[:check_n]).astype(np.float64)
        rel = float((np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)).max())
        r["parity_max_rel_err"] = rel
        r["parity_ok"] = bool(rel <= 1e-5)
        r["cpu_baseline"] = {"value": 8.0 * 20_000_000 / t / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
                             "sample": "%d x 2e7-element NDArray_Map(float_%s) restatement (libm, 1 thread)" % (it, op)}
        ex["%s_1e8" % op] = r
        dx.free()
        del x

    # C3c: broadcast forms on 25000 x 4000, no materialised temporary: 8 B/elem + the vector
    R, Cc = 25000, 4000
    row = synth.uniform((Cc,), 9, 0.0, 1.0)
    col = synth.uniform((R,), 10, 0.0, 1.0)
    drow, dcol = D.DeviceArray.from_host(row), D.DeviceArray.from_host(col)
    r = hbm_case("X + row, 25000x4000 (C3c)", 8.0 * N + 4.0 * Cc,
                 lambda: D.binary("add", da, "full", drow, "row", R, Cc, out=do), steps, warmup, dist)
    got = do.to_host().reshape(R, Cc)
    r["parity_ok"] = bool((got == (a.reshape(R, Cc) + row[None, :])).all())
    ex["add_row_broadcast"] = r
    r = hbm_case("X + col, 25000x4000 (C3c)", 8.0 * N + 4.0 * R,
                 lambda: D.binary("add", da, "full", dcol, "col", R, Cc, out=do), steps, warmup, dist)
    got = do.to_host().reshape(R, Cc)
    r["parity_ok"] = bool((got == (a.reshape(R, Cc) + col[:, None])).all())
    ex["add_col_broadcast"] = r
    # C3c as BASELINE.json words it — "exp/log with broadcast": exp(X) + r in ONE pass (8 B/elem)
    # through the fused chain, against the two launches (16 B/elem + a temporary) the op-by-op API costs
    prog2 = (FusedOp * 2)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0),
                          FusedOp(1, BINARY_OPS["add"], 1, 0, 0, 0, 0, 0))
    tmp = D.DeviceArray((N,))
    for label, dvec, kind, nvec in (("row", drow, 2, Cc), ("col", dcol, 3, R)):
        ptrs2 = (C.c_void_p * 2)(da.ptr, dvec.ptr)
        kinds2 = (C.c_int * 2)(0, kind)
        r = hbm_case("exp(X) + %s fused, 25000x4000 (C3c)" % label, 8.0 * N + 4.0 * nvec,
                     lambda: check(lib.np_fused_chain(ptrs2, kinds2, 2, prog2, 2, do.ptr, R, Cc)),
                     steps, warmup, dist)
        fused_out = do.to_host().reshape(-1)

        def two_launches():
            D.unary("exp", da, out=tmp)
            D.binary("add", tmp, "full", dvec, label, R, Cc, out=do)

        _, ev_ms = timed(dist, two_launches, steps, warmup)
        r["unfused_ms_per_chain"] = ev_ms / steps
        r["speedup_vs_unfused"] = (ev_ms / steps) / r["ms_per_launch"]
        r["parity_ok"] = bool((fused_out.view(np.uint32) == do.to_host().reshape(-1).view(np.uint32)).all())
        ex["exp_plus_%s_fused" % label] = r
    # ... and with a reduction over an axis as the chain's last step: sum(exp(X), axis) reads X once (4 B/elem)
    prog1 = (FusedOp * 1)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0))
    ptrs1 = (C.c_void_p * 1)(da.ptr)
    kinds1 = (C.c_int * 1)(0)
    e64 = np.exp(a.reshape(R, Cc).astype(np.float64))
    for axis, nout in ((1, R), (0, Cc)):
        dred = D.DeviceArray((nout,))
        r = hbm_case("sum(exp(X), axis %d) fused, 25000x4000" % axis, 4.0 * N + 4.0 * nout,
                     lambda: check(lib.np_fused_chain_reduce_axis(ptrs1, kinds1, 1, prog1, 1, 0, R, Cc, axis, dred.ptr)),
                     steps, warmup, dist)
        ref = e64.sum(axis=axis)
        r["parity_max_rel_err_vs_fp64"] = float((np.abs(dred.to_host().astype(np.float64) - ref) / ref).max())
        r["parity_ok"] = bool(r["parity_max_rel_err_vs_fp64"] <= 1e-5)

        def exp_then_reduce():
            D.unary("exp", da, out=tmp)
            check(lib.np_reduce_axis(0, tmp.ptr, R if axis == 1 else 1, Cc if axis == 1 else R, 1 if axis == 1 else Cc, dred.ptr, 0))

        _, ev_ms = timed(dist, exp_then_reduce, steps, warmup)
        r["unfused_ms_per_chain"] = ev_ms / steps
        r["speedup_vs_unfused"] = (ev_ms / steps) / r["ms_per_launch"]
        ex["sum_exp_axis%d_fused" % axis] = r
        dred.free()
    del e64
    tmp.free()
    # SURVEY.md section 8(f) rows 2 and 4 at the sizes of tools/misc_sweep.py: argmax of the flat 1e8 array
    # (4 B/elem, calculation.c:73-194), variance in one read (np_moments: 4 B/elem, statistics.c:88-130), the weighted average's two sums (8 B/elem, :131-154) and
    # dot(matrix, vector) with few long rows (10 x 1e7: 4 (M K + K + M) bytes, linalg.c:367-386)
    idx = D.DeviceArray((1,))
    r = hbm_case("argmax flat 1e8 (8f row 2)", 4.0 * N, lambda: check(lib.np_argreduce(1, da.ptr, 1, N, 1, idx.ptr)), steps, warmup, dist)
    r["parity_ok"] = bool(idx.to_host()[0] == np.float32(np.argmax(a)))
    ex["argmax_1e8"] = r
    mean, m2 = C.c_float(), C.c_float()
    r = hbm_case("moments (variance) 1e8, one read (8f row 2)", 4.0 * N,
                 lambda: check(lib.np_moments(da.ptr, N, C.byref(mean), C.byref(m2))), steps, warmup, dist)
    a64 = a.astype(np.float64)
    var64 = float(((a64 - a64.mean()) ** 2).sum())
    r["parity_rel_err_vs_fp64"] = abs(m2.value - var64) / var64
    r["parity_ok"] = bool(r["parity_rel_err_vs_fp64"] <= 1e-5 and abs(mean.value - a64.mean()) <= 1e-5 * a64.mean())
    # the same values moved to 1e4 (large mean, small spread) on the device and read back: fp64 of the very array the device holds
    dsc = D.DeviceArray.from_host(np.array([1e4], dtype=np.float32))
    D.binary("add", da, "full", dsc, "scalar", 1, N, out=do)
    dsc.free()
    check(lib.np_moments(do.ptr, N, C.byref(mean), C.byref(m2)))
    s64 = do.to_host().astype(np.float64)
    svar = float(((s64 - s64.mean()) ** 2).sum())
    r["large_mean_rel_err_vs_fp64"] = abs(m2.value - svar) / svar
    r["parity_ok"] = bool(r["parity_ok"] and r["large_mean_rel_err_vs_fp64"] <= 1e-5)
    del s64
    ex["moments_1e8"] = r
    saw, sw = C.c_float(), C.c_float()
    r = hbm_case("weighted sums (average) 1e8, one read of each (8f row 2)", 8.0 * N,
                 lambda: check(lib.np_weighted_sums(da.ptr, db.ptr, N, C.byref(saw), C.byref(sw))), steps, warmup, dist)
    b64 = b.astype(np.float64)
    r["parity_rel_err_vs_fp64"] = max(abs(saw.value - float((a64 * b64).sum())) / float((a64 * b64).sum()),
                                      abs(sw.value - float(b64.sum())) / float(b64.sum()))
    r["parity_ok"] = bool(r["parity_rel_err_vs_fp64"] <= 1e-5)
    ex["weighted_sums_1e8"] = r
    del a64, b64
    Mv, Kv = 10, 10_000_000
    yv = D.DeviceArray((Mv,))
    r = hbm_case("sgemv 10 x 1e7 (8f row 4)", 4.0 * (Mv * Kv + Kv + Mv), lambda: check(lib.np_sgemv(Mv, Kv, da.ptr, db.ptr, yv.ptr)),
                 steps, warmup, dist)
    refv = a.reshape(Mv, Kv).astype(np.float64) @ b[:Kv].astype(np.float64)
    r["parity_max_rel_err_vs_fp64"] = float((np.abs(yv.to_host().astype(np.float64) - refv) / np.abs(refv)).max())
    r["parity_ok"] = bool(r["parity_max_rel_err_vs_fp64"] <= 1e-5)
    ex["sgemv_10x1e7"] = r
    yv.free()
    idx.free()
    for d in (da, db, do, drow, dcol):
        d.free()
    del a, b, got

    # C4: sum(axis 0) of 65536 x 4096
    rows, cols = 65536, 4096
    X = synth.uniform((rows, cols), 11, 0.0, 1.0)
    dX, dout = D.DeviceArray.from_host(X), D.DeviceArray((cols,))
    r = hbm_case("sum(axis 0) 65536x4096 (C4)", 4.0 * rows * cols + 4.0 * cols,
                 lambda: D.reduce_axis("sum", dX, 0, out=dout), steps, warmup, dist)
    got = dout.to_host().astype(np.float64)
    ref64 = X.sum(axis=0, dtype=np.float64)
    rel = float((np.abs(got - ref64) / ref64).max())
    r["parity_max_rel_err_vs_fp64"] = rel
    r["parity_ok"] = bool(rel <= 1e-5)
    sub = X[:8192]
    t, it = cpu_time(lambda: oracle.reduce_axis("sum", sub, 0), budget_s=5.0, max_iters=3)
    ref_seq = oracle.reduce_axis("sum", sub, 0).astype(np.float64)
    r["reference_order_rel_err_vs_fp64_8192rows"] = float(
        (np.abs(ref_seq - sub.sum(axis=0, dtype=np.float64)) / sub.sum(axis=0, dtype=np.float64)).max())
    r["cpu_baseline"] = {"value": 4.0 * sub.size / t / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
                         "sample": "%d x reduce(axis 0) restatement on the first 8192 rows" % it}
    ex["sum_axis0"] = r
    dout.free()
    # device transpose of the same 65536 x 4096 array, 8 B/elem
    dT = D.DeviceArray((cols, rows))
    r = hbm_case("transpose 65536x4096 (manipulation.c, §8f)", 8.0 * rows * cols,
                 lambda: D.transpose2d(dX, out=dT), steps, warmup, dist)
    r["parity_ok"] = bool((dT.to_host()[:, :4096] == X[:4096].T).all())
    ex["transpose_65536x4096"] = r
    dT.free()
    # argmax over the last axis of 65536 x 1024 (the first quarter of the same buffer, contiguous): one wave per row
    ar, ac = 65536, 1024
    didx = D.DeviceArray((ar,))
    r = hbm_case("argmax(axis 1) 65536x1024 (8f row 2)", 4.0 * ar * ac + 4.0 * ar,
                 lambda: check(lib.np_argreduce(1, dX.ptr, ar, ac, 1, didx.ptr)), steps, warmup, dist)
    r["parity_ok"] = bool((didx.to_host() == np.argmax(X.reshape(-1)[:ar * ac].reshape(ar, ac), axis=1).astype(np.float32)).all())
    ex["argmax_axis1_65536x1024"] = r
    didx.free()
    dX.free()
    return ex


XGMI_LINK_GBPS = 153.0   # MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU; a peer's slab arrives over that peer's own link


def interleaved_legs(dist, legs, steps, rounds=5, prewarm=None, prewarm_s=0.25):
    """Times several forms of the same step so that none of them owns the cold (or the hot) side of the clock ramp
    (measured one after the other, `compute_only` came out SLOWER than compute + gather).  First
    `prewarm` runs for >= prewarm_s on every rank (local work, no collective); then `rounds` rounds, each timing every
    leg once (1 untimed call + `steps` timed calls between barriers, max over ranks) in an order rotated by one per
    round.  -> {name: {"median": s/step, "min": s/step, "samples": [...]}}"""
    names = list(legs)
    if prewarm is not None and not DRYRUN:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < prewarm_s:
            for _ in range(10):
                prewarm()
            D.sync()
    samples = {k: [] for k in names}
    for r in range(rounds):
        k0 = r % len(names)
        for k in names[k0:] + names[:k0]:
            fn = legs[k]
            if getattr(fn, "before", None):      # a switch that synchronises (np_comm_set_variant) stays outside the clock
                fn.before()
            samples[k].append(timed(dist, fn, steps, 1)[0] / steps)
            if getattr(fn, "after", None):
                fn.after()
    return {k: {"median": float(np.median(v)), "min": float(min(v)), "samples": v} for k, v in samples.items()}


def _config5_report(dist, per, n, legs, slab_bytes, parity, how, steps):
    """The common shape of config 5's two reports.  legs: interleaved_legs() output (wall seconds per step, max over
    ranks).  Every derived number uses the per-leg MEDIAN over the rounds; the minimum is printed beside it.  Nothing is
    clamped: a gathered form that comes out faster than compute alone is reported as what it is — an inconsistent
    measurement — and no "gather alone" figure is derived from it."""
    total = per * dist.n
    flop = 2.0 * total * n ** 3
    med = {k: v["median"] for k, v in legs.items()}
    base = med["compute_only"]
    out = {"workload": "%d x (%dx%d) fp32 batched matmul, %d slab(s) of %d, %s" % (total, n, n, dist.n, per, how),
           "scaling": "strong", "allgather_bytes_per_rank": slab_bytes,
           "protocol": "%d rounds x %d steps per leg, legs interleaved in rotating order behind a 0.25 s pre-warm; median (min)"
                       % (len(next(iter(legs.values()))["samples"]), steps),
           "ms_per_step": {k: round(v * 1e3, 4) for k, v in med.items()},
           "ms_per_step_min": {k: round(v["min"] * 1e3, 4) for k, v in legs.items()},
           "compute_only_GFLOPs": flop / base / 1e9,
           "gathered_GFLOPs": flop / med["gathered"] / 1e9,
           "vs_compute_only": {k: round(v / base, 4) for k, v in med.items() if k != "compute_only"},
           "parity_max_norm_err_vs_fp64": parity, "parity_ok": bool(max(parity.values()) <= 1e-6)}
    best = min((v, k) for k, v in med.items() if k != "compute_only")
    out["best_gathered_form"] = best[1]
    out["best_gathered_GFLOPs"] = flop / best[0] / 1e9
    # a form that computes AND moves cannot beat computing alone by more than the noise of the measurement
    slack = 0.98
    bad = sorted(k for k, v in med.items() if v < slack * base)
    out["consistent"] = not bad
    if bad:
        out["inconsistent_legs"] = bad
    if dist.n > 1:
        # every rank receives (n - 1) slabs, each over its own link: the rate one link has to sustain
        exposed = med["gathered"] - base
        x = {"model_link_GBps": XGMI_LINK_GBPS, "model_gather_ms": slab_bytes / XGMI_LINK_GBPS / 1e6,
             "link_GBps_best_form_whole_step": slab_bytes / best[0] / 1e9}
        if exposed > 0:
            x["gather_alone_ms"] = exposed * 1e3                  # = gathered - compute_only (medians)
            x["link_GBps_gather_alone"] = slab_bytes / exposed / 1e9
        else:
            x["gather_alone_ms"] = None
            x["link_GBps_gather_alone"] = None
            x["inconsistent"] = True
        out["xgmi"] = x
    return out


def _peer_matrix_err(got, j, n):
    Ah = synth.uniform((n, n), 12_000 + j, -1.0, 1.0).astype(np.float64)
    Bh = synth.uniform((n, n), 13_000 + j, -1.0, 1.0).astype(np.float64)
    return float((np.abs(got.astype(np.float64) - Ah @ Bh) / (np.abs(Ah) @ np.abs(Bh))).max())


def bench_config5(dist: Dist, steps, rounds=5):
    """BASELINE config 5 with torch.distributed as the plumbing: 512 x (1024 x 1024) batched matmul, batch sharded
    over the ranks in contiguous slabs (strong scaling).  Legs: compute only; compute + ONE all-gather of the result
    slabs behind it; and the overlapped pipeline of numpower_amd.parallel (slab in 2 / 4 / 8 pieces, each piece's
    point-to-point exchange on the process group's stream while the next piece computes).  Timed by interleaved_legs."""
    from numpower_amd import parallel
    torch = dist.torch
    # NP_BENCH_DRYRUN: the same legs, the same collectives (gloo), the same report — 16 x (32 x 32) on CPU tensors with torch.bmm
    # standing in for the GEMM launch: the N > 1 path of this function end to end on a box without a GPU
    total, n = (8 * dist.n, 32) if DRYRUN else (512, 1024)
    per = total // dist.n
    lo = dist.rank * per
    dev = torch.device("cpu") if DRYRUN else torch.device("cuda", dist.local_rank)
    # inputs of this rank's slab, generated per matrix so every rank holds exactly its own
    A = torch.empty((per, n, n), dtype=torch.float32, device=dev)
    B = torch.empty((per, n, n), dtype=torch.float32, device=dev)
    for i in range(per):
        A[i].copy_(torch.from_numpy(synth.uniform((n, n), 12_000 + lo + i, -1.0, 1.0)))
        B[i].copy_(torch.from_numpy(synth.uniform((n, n), 13_000 + lo + i, -1.0, 1.0)))
    Cfull = torch.empty((total, n, n), dtype=torch.float32, device=dev)
    mine = Cfull[lo:lo + per]
    lib = None if DRYRUN else load()
    from numpower_amd._lib import check
    bad_leg = os.environ.get("NP_BENCH_DRYRUN_BAD_LEG") if DRYRUN else None   # test hook: this leg leaves a wrong result

    def gemm(a, b, out):
        if DRYRUN:
            torch.bmm(a, b, out=out)
            return
        check(lib.np_sgemm_strided_batched(a.shape[0], n, n, n, a.data_ptr(), n * n, b.data_ptr(), n * n,
                                           out.data_ptr(), n * n))

    def compute():
        gemm(A, B, mine)

    def compute_and_gather():
        compute()
        dist.dist.all_gather_into_tensor(Cfull.view(-1), mine.reshape(-1))
        if bad_leg == "gathered":
            Cfull.mul_(1.5)

    def overlapped(chunks):
        def step():
            handles = []
            for plo, cnt in parallel.pieces_of(per, chunks):
                gemm(A[plo:plo + cnt], B[plo:plo + cnt], mine[plo:plo + cnt])
                handles.extend(parallel.exchange_piece(dist.dist, Cfull, per, plo, cnt))
            for h in handles:
                h.wait()
            if bad_leg == "overlapped_%d" % chunks:
                Cfull.mul_(1.5)
        return step

    forms = {"compute_only": compute, "gathered": compute_and_gather}
    for chunks in (2, 4, 8):
        if chunks <= per:
            forms["overlapped_%d" % chunks] = overlapped(chunks)
    legs = interleaved_legs(dist, forms, steps, rounds, prewarm=compute)
    # parity: one matrix of a PEER's slab as each gathering form leaves it in a zeroed result, against fp64
    j = ((dist.rank + 1) % dist.n) * per + per - 1
    parity = {}
    for name, fn in forms.items():
        if name == "compute_only":
            continue
        Cfull.zero_()
        fn()
        if not DRYRUN:
            torch.cuda.synchronize()
        parity[name] = _peer_matrix_err(Cfull[j].cpu().numpy(), j, n)
    return _config5_report(dist, per, n, legs, per * n * n * 4, parity,
                           "torch.distributed (gloo, dry run)" if DRYRUN else "torch.distributed (RCCL) collectives", steps)


def bench_config5_abi(dist: Dist, steps, rounds=5, own_comm_port=None, world1=False, secured=None, full_batch=False):
    """BASELINE config 5 the way a C / PHP host writes it — no torch tensor, no torch collective: the rank's
    slab of the batch is written in place into the full result buffer by np_sgemm_strided_batched, and
      gathered         ONE np_allgather behind it on the same stream (no overlap possible)
      two_stream       np_sgemm_strided_batched_allgather(chunks = 1): the same all-gather on the communication stream
      p2p_1            ... chunks = 1, moved as one grouped send/recv exchange instead of ncclAllGather
      overlapped_k     ... the slab in k pieces, piece c's exchange travelling while piece c + 1 is computed (the
                       default issue form: one progress-reporting launch without peers, one launch per piece with peers)
      single_launch_k  (world > 1 only) ... the same with ONE progress-reporting GEMM launch per slab
                       (np_comm_set_variant(3): opt-in with peers until a multi-GPU run has validated it — this leg,
                       with the parity check behind it, is that run)
    Timed by interleaved_legs (pre-warm, rotating order, median / min over the rounds).
    own_comm_port: bring up a communicator just for this leg (torch mode: the job's collectives belong to
    torch.distributed).  world1: a one-rank communicator on a single GPU — nothing travels, but the whole mechanism
    (second stream, device-side flags, progress counters) runs, so `overlapped_k` / `compute_only` is what the pipeline
    itself costs.  full_batch (with world1): the WHOLE of config 5 — all 512 matrices, 2 GiB each of A, B and C — on the one
    GPU: the workload an 8-GPU node shards, unsharded (VERDICT r04 missing #3); three forms only."""
    from numpower_amd._lib import check
    lib = load()
    total, n = (64, 1024) if (world1 and not full_batch) else (512, 1024)
    per = total // dist.n
    lo = dist.rank * per
    if own_comm_port is not None:
        with _stdout_to_devnull():
            check(lib.np_comm_init(dist.rank, dist.n, ("tcp://127.0.0.1:%d" % own_comm_port).encode()))
    try:
        A, B = D.DeviceArray((per, n, n)), D.DeviceArray((per, n, n))
        for dst, seed0 in ((A, 12_000), (B, 13_000)):      # one seed per matrix: every rank generates exactly its own slab
            for i, h in enumerate(synth.uniform_many((n, n), range(seed0 + lo, seed0 + lo + per), -1.0, 1.0)):
                check(lib.np_memcpy_h2d(dst.ptr + i * n * n * 4, h.ctypes.data, n * n * 4))
        full = D.DeviceArray((total, n, n))
        mine = full.ptr + lo * n * n * 4
        slab_bytes = per * n * n * 4

        def compute():
            check(lib.np_sgemm_strided_batched(per, n, n, n, A.ptr, n * n, B.ptr, n * n, mine, n * n))

        def compute_and_gather():
            compute()
            check(lib.np_allgather(mine, full.ptr, slab_bytes))

        def pipelined(chunks, mode, variant=0):
            def step():
                check(lib.np_sgemm_strided_batched_allgather(per, n, n, n, A.ptr, n * n, B.ptr, n * n, full.ptr, chunks, mode))
            if variant:
                step.before = lambda: check(lib.np_comm_set_variant(variant))
                step.after = lambda: check(lib.np_comm_set_variant(0))
            return step

        forms = {"compute_only": compute, "gathered": compute_and_gather,
                 "two_stream": pipelined(1, 1), "p2p_1": pipelined(1, 2)}
        for chunks in (2, 4, 8):
            if chunks <= per:
                forms["overlapped_%d" % chunks] = pipelined(chunks, 0)
        model_pick = C.c_int(0)
        check(lib.np_comm_debug_model(dist.n, per, n, n, n, 0, C.byref(model_pick), None))
        if dist.n > 1:
            forms["overlapped_auto"] = pipelined(0, 0)     # chunks = 0: the library's step model picks the piece count
        if full_batch:
            forms = {k: forms[k] for k in ("compute_only", "gathered", "two_stream", "overlapped_8")}
        j = ((dist.rank + 1) % dist.n) * per + per - 1  # one matrix of a PEER's slab, as each gathering form leaves it
        got = np.empty((n, n), dtype=np.float32)

        def measure(fs):
            lg = interleaved_legs(dist, fs, steps, rounds, prewarm=compute)
            par = {}
            for name, fn in fs.items():
                if name == "compute_only":
                    continue
                check(lib.np_memset0(full.ptr, total * n * n * 4))
                if getattr(fn, "before", None):
                    fn.before()
                fn()
                check(lib.np_memcpy_d2h(got.ctypes.data, full.ptr + j * n * n * 4, n * n * 4))
                if getattr(fn, "after", None):
                    fn.after()
                par[name] = _peer_matrix_err(got, j, n)
            return lg, par

        legs, parity = measure(forms)
        rep = _config5_report(dist, per, n, legs, slab_bytes, parity, "np_comm_* (RCCL behind the C ABI)", steps)
        rep["model_piece_count"] = model_pick.value            # what chunks = 0 resolves to for this world / slab
        if world1 and not full_batch:
            rep["workload"] = ("64 x (1024x1024) fp32 batched matmul = ONE rank's slab of config 5 on a one-rank communicator: "
                               "nothing travels, the two-stream pipeline itself is what is measured")
        if full_batch:
            rep["workload"] = ("512 x (1024x1024) fp32 batched matmul = the WHOLE of config 5 on one GPU (one-rank communicator: "
                               "nothing travels)")
            # every matrix of the pipelined form against the plain launch, bit for bit, on the device
            ref = D.DeviceArray((total, n, n))
            check(lib.np_sgemm_strided_batched(per, n, n, n, A.ptr, n * n, B.ptr, n * n, ref.ptr, n * n))
            forms["overlapped_8"]()
            differs = C.c_int(1)
            check(lib.np_count_mismatch(0, full.ptr, ref.ptr, total * n * n, 0.0, 0.0, C.byref(differs)))
            rep["all_512_matrices_bit_identical_to_plain_launch"] = differs.value == 0
            rep["parity_ok"] = bool(rep["parity_ok"] and differs.value == 0)
            tf = rep["compute_only_GFLOPs"] / 1e3
            rep["roofline"] = {"bound": "mfma", "achieved": tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": tf / PEAK_FP32_MFMA_TFLOPS, "algorithmic_flop_per_launch": 2.0 * total * n ** 3}
            ref.free()
        if secured is not None:
            secured["config5_batched_matmul_allgather_c_abi"] = rep     # what the watchdog prints if the next phase never returns
        if dist.n > 1:
            # second phase, after the report of the proven forms is in hand: ONE progress-reporting GEMM launch per slab with
            # real peers (np_comm_set_variant(3)) — a form that has only ever run on one GPU.  Its own legs, its own parity.
            trial = {"compute_only": compute}
            for chunks in (4, 8):
                if chunks <= per:
                    trial["single_launch_%d" % chunks] = pipelined(chunks, 0, variant=3)
            try:
                lg2, par2 = measure(trial)
                base = lg2["compute_only"]["median"]
                rep["single_launch_with_peers"] = {
                    "ms_per_step": {k: round(v["median"] * 1e3, 4) for k, v in lg2.items()},
                    "vs_compute_only": {k: round(v["median"] / base, 4) for k, v in lg2.items() if k != "compute_only"},
                    "parity_max_norm_err_vs_fp64": par2, "parity_ok": bool(max(par2.values()) <= 1e-6)}
            except Exception as e:
                rep["single_launch_with_peers"] = {"error": repr(e)}
        for d in (A, B, full):
            d.free()
    finally:
        if own_comm_port is not None:
            with _stdout_to_devnull():
                lib.np_comm_destroy()
    return rep


def _compact(x):
    """The printed line must fit every BASELINE config into the tail a log keeps: prose (`note`, `name`) stays in this
    file's docstrings, floats are cut to 6 significant digits, per-round sample lists are dropped."""
    if isinstance(x, dict):
        return {k: _compact(v) for k, v in x.items() if k not in ("note", "name", "samples")}
    if isinstance(x, (list, tuple)):
        return [_compact(v) for v in x]
    if isinstance(x, float):
        return float("%.6g" % x)
    return x


def _summary(result, extras):
    """One short object with the figure of every BASELINE config (C1 .. C5), placed last on the line."""
    def frac(key):
        e = extras.get(key)
        return _compact(e["roofline"]["frac"]) if isinstance(e, dict) and "roofline" in e else None

    def ok(key):
        e = extras.get(key)
        return e.get("parity_ok") if isinstance(e, dict) else None

    out = {"c2_matmul_4096_frac_mfma": _compact(result["roofline"]["frac"]),
           "c2_matmul_4096_TFLOPs": _compact(result["roofline"]["achieved"]),
           "c3a_add_1e8_frac_hbm": frac("add_1e8"),
           "c3a_add_1e8_GBps": _compact(extras["add_1e8"]["GBps"]) if "add_1e8" in extras else None,
           "c3b_exp_1e8_frac_hbm": frac("exp_1e8"), "c3b_log_1e8_frac_hbm": frac("log_1e8"),
           "c3c_add_row_frac_hbm": frac("add_row_broadcast"), "c3c_add_col_frac_hbm": frac("add_col_broadcast"),
           "c3c_exp_plus_row_fused_frac_hbm": frac("exp_plus_row_fused"),
           "c3c_exp_plus_col_fused_frac_hbm": frac("exp_plus_col_fused"),
           "c4_sum_axis0_frac_hbm": frac("sum_axis0")}
    for key in ("matmul_768", "matmul_1000", "matmul_1024", "matmul_100x100x100000"):
        e = extras.get(key)
        if isinstance(e, dict) and "TFLOPs" in e:
            out[key + "_TFLOPs"] = _compact(e["TFLOPs"])
    out["transpose_8191x8193_frac_hbm"], out["permute_nhwc_like_frac_hbm"] = frac("transpose_8191x8193"), frac("permute_nhwc_like")
    out["argmax_1e8_frac_hbm"], out["argmax_axis1_65536x1024_frac_hbm"] = frac("argmax_1e8"), frac("argmax_axis1_65536x1024")
    out["moments_1e8_frac_hbm"], out["sgemv_10x1e7_frac_hbm"] = frac("moments_1e8"), frac("sgemv_10x1e7")
    out["weighted_sums_1e8_frac_hbm"] = frac("weighted_sums_1e8")
    c1 = extras.get("c1")
    if isinstance(c1, dict) and "cpu_add_ms" in c1:
        out["c1_cpu_add_ms"], out["c1_cpu_sum_ms"] = _compact(c1["cpu_add_ms"]), _compact(c1["cpu_sum_ms"])
        out["c1_gpu_add_us"], out["c1_gpu_sum_us"] = _compact(c1["gpu_add_kernel_us"]), _compact(c1["gpu_sum_call_us"])
        out["c1_end_to_end_add_ms"] = _compact(c1["end_to_end_gpu_add_cpu_ms"])
    for key in ("config5_one_rank_slab_c_abi", "config5_full_batch_world1", "config5_batched_matmul_allgather_c_abi",
                "config5_batched_matmul_allgather"):
        c5 = extras.get(key)
        if isinstance(c5, dict) and "ms_per_step" in c5:
            out["c5_" + key[8:]] = {"ms_per_step": c5["ms_per_step"], "consistent": c5["consistent"],
                                    "compute_only_GFLOPs": _compact(c5["compute_only_GFLOPs"]),
                                    "best_gathered_GFLOPs": _compact(c5["best_gathered_GFLOPs"]),
                                    "best_gathered_form": c5["best_gathered_form"], "parity_ok": c5["parity_ok"]}
            if "roofline" in c5:
                out["c5_" + key[8:]]["frac_mfma"] = _compact(c5["roofline"]["frac"])
    out["parity_all_ok"] = bool(result["parity"]["ok"] and all(
        e.get("parity_ok", True) for e in extras.values() if isinstance(e, dict)))
    return out


def _diag_add(dist, label):
    """NP_BENCH_DIAG=1: the add-1e8 rate at different points of the run (is it the kernel or the context?)."""
    if os.environ.get("NP_BENCH_DIAG") != "1":
        return
    N = 100_000_000
    a, b, o = D.DeviceArray((N,)), D.DeviceArray((N,)), D.DeviceArray((N,))
    D.fill(a, 0.25)
    D.fill(b, 0.5)
    rates = []
    for _ in range(4):
        _, ev_ms = timed(dist, lambda: D.binary("add", a, "full", b, "full", 1, N, out=o), 25, 5)
        rates.append(12.0 * N / (ev_ms / 25) / 1e6)
    log("[diag] add 1e8 (constant data) %-28s %s GB/s  a=%#x b=%#x o=%#x" % (label, " ".join("%.0f" % r for r in rates), a.ptr, b.ptr, o.ptr))
    for d in (a, b, o):
        d.free()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-extras", action="store_true", help="headline only")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # no external launcher: this process becomes the launcher (one rank per GPU, see spawn_ranks)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    if DRYRUN and os.environ.get("NP_BENCH_DRYRUN_FAIL_RANK") == os.environ.get("RANK", ""):
        sys.exit(3)     # test hook: this rank dies before the rendezvous (tests/test_bench_launcher_cpu.py)
    dist = Dist(args.gpus)
    rank0 = dist.rank == 0
    if DRYRUN:
        # launcher smoke without a GPU: the timed region is K sleeps; no throughput claim is made
        wall, _ = timed(dist, lambda: time.sleep(0.001), args.steps, args.warmup)
        who = dist.describe()                                            # collective: every rank calls it
        line = {"metric": METRIC, "value": 0.0, "unit": "GFLOP/s", "n_gpus": args.gpus,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
                "dry_run": True, **who}
        if dist.use_torch and args.gpus > 1 and not args.no_extras:
            # config 5's N > 1 report on gloo: every leg timed and checked, whatever one leg's parity says
            try:
                line["extras"] = {"config5_batched_matmul_allgather": _compact(bench_config5(dist, max(2, args.steps // 2), 2))}
            except Exception as e:      # noqa: BLE001
                line["extras"] = {"error": repr(e)}
        if rank0:
            print(json.dumps(line), flush=True)
        dist.close()
        return
    _diag_add(dist, "before anything")
    mm = bench_matmul(dist, args.steps, args.warmup, do_cpu=rank0 and args.gpus == 1 and os.environ.get("NP_BENCH_DIAG_NOCPU") != "1")
    _diag_add(dist, "after matmul + cpu baseline")
    flop = mm["flop_per_launch"]                                            # one 4096^3 product = one launch
    P = mm["products_per_step"]
    value = mm["flop_per_step"] * args.steps * args.gpus / mm["wall_s"] / 1e9   # GFLOP/s, whole job
    kernel_tflops = flop / (mm["event_ms"] / (args.steps * P)) / 1e9        # this rank's kernel: average launch of the timed region
    ramp = mm.get("ramp") or {}
    result = {
        "metric": METRIC, "value": value, "unit": "GFLOP/s", "n_gpus": args.gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": mm["wall_s"] / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "nd::matmul 4096x4096 . 4096x4096 fp32 (BASELINE config 2); a step = one batch of %d independent "
                               "products (one launch each, %d distinct operand sets) per GPU" % (P, PRODUCT_SETS),
                   "products_per_step": P,
                   "parallelism": "replicas x%d (independent arrays, no collective)" % args.gpus},
        "roofline": {"bound": "mfma", "achieved": kernel_tflops, "peak": PEAK_FP32_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": kernel_tflops / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                     "kernel": "sgemm_dma_kernel 256x128x16 (v_mfma_f32_32x32x2_f32, LDS-DMA staging)",
                     "algorithmic_flop_per_launch": flop, "launches_per_step": P, "launch_ms": mm["launch_ms"],
                     "frac_best_launch": flop / mm["launch_ms"]["min"] / 1e9 / PEAK_FP32_MFMA_TFLOPS,
                     # what the one-product steps of the lines before this one reported, measured in THIS run on the warm-up launches
                     "frac_launches_5_24": (ramp["TFLOPs_launches_5_24"] / PEAK_FP32_MFMA_TFLOPS) if ramp.get("TFLOPs_launches_5_24") else None,
                     "ramp": ramp or None},
        "parity": {"matmul_max_norm_err_vs_fp64": mm["parity_max_norm_err_vs_fp64"], "ok": mm["parity_ok"],
                   "batch_linearity_bit_exact": mm["batch_linearity_bit_exact"]},
    }
    result.update(dist.describe())      # ranks_seen + the collective library and its version (a collective: every rank calls it)
    if "cpu" in mm:
        result["cpu_baseline"] = mm["cpu"]
        result["parity"]["gpu_vs_cpu_reference_max_norm_err"] = mm["gpu_vs_cpu_max_norm_err"]
    extras = {}
    secured = {}      # config-5 reports already complete when a later phase hangs (sharded_watchdog prints them)

    def sharded_watchdog():
        # the sharded extra has collectives in it: if a peer dies or a transfer never completes there, the headline
        # measured above must still be reported — after 200 s rank 0 prints it without the extra and every rank leaves
        def bail():
            if rank0:
                result["extras"] = dict(_compact(secured), error="config 5 (sharded batched matmul + all-gather) did not finish in 200 s; "
                                                                 "what had been measured by then is above")
                print(json.dumps(result), flush=True)
            os._exit(0)
        t = threading.Timer(200.0, bail)
        t.daemon = True
        t.start()
        return t

    if not args.no_extras:
        if dist.abi:
            watchdog = sharded_watchdog()
            try:
                extras = {"config5_batched_matmul_allgather_c_abi": bench_config5_abi(dist, max(5, args.steps // 5), 5, secured=secured)}
            except Exception as e:
                extras = {"error": repr(e)}
            watchdog.cancel()
        elif not dist.use_torch:
            try:
                extras = bench_extras(dist, max(10, args.steps // 2), args.warmup)
                try:
                    extras["c1"] = bench_c1(dist, max(10, args.steps // 2), args.warmup)
                except Exception as e:
                    extras["c1"] = {"error": repr(e)}
                try:
                    extras.update(bench_mid(dist, max(10, args.steps // 2), args.warmup))
                except Exception as e:
                    extras["mid_sizes"] = {"error": repr(e)}
                try:
                    extras["config5_one_rank_slab_c_abi"] = bench_config5_abi(dist, max(10, args.steps // 2), 7,
                                                                                own_comm_port=_free_port(), world1=True)
                except Exception as e:
                    extras["config5_one_rank_slab_c_abi"] = {"error": repr(e)}
                try:
                    extras["config5_full_batch_world1"] = bench_config5_abi(dist, 5, 5, own_comm_port=_free_port(), world1=True,
                                                                            full_batch=True)
                except Exception as e:
                    extras["config5_full_batch_world1"] = {"error": repr(e)}
                add = extras["add_1e8"]
                result["secondary"] = {"metric": "GB/s elementwise add 10^8 fp32", "value": add["GBps"],
                                       "unit": "GB/s", "roofline": add["roofline"],
                                       "cpu_baseline": add["cpu_baseline"], "parity_ok": add["parity_ok"]}
            except Exception as e:   # extras must never take the headline down
                extras = {"error": repr(e)}
        else:
            watchdog = sharded_watchdog()
            try:
                extras = {"config5_batched_matmul_allgather": bench_config5(dist, max(5, args.steps // 5), 5)}
                secured.update(extras)
            except Exception as e:
                extras = {"error": repr(e)}
            # the same workload with the collective issued through the C ABI (np_allgather) on its own communicator
            try:
                port = int(os.environ.get("MASTER_PORT", "29531")) + 23
                extras["config5_batched_matmul_allgather_c_abi"] = bench_config5_abi(dist, max(5, args.steps // 5), 5,
                                                                                      own_comm_port=port, secured=secured)
            except Exception as e:
                extras["config5_batched_matmul_allgather_c_abi"] = {"error": repr(e)}
            watchdog.cancel()
    # MFMA utilisation of the headline kernel from the committed rocprofv3 SQ counter pass
    # (profiles/rNN/gemm_pmc.json: SQ_VALU_MFMA_BUSY_CYCLES over 4 SIMDs x SQ_BUSY_CU_CYCLES)
    pmc_files = sorted((ROOT / "profiles").glob("r*/gemm_pmc.json"))
    if pmc_files:
        try:
            g = json.loads(pmc_files[-1].read_text())
            result["roofline"]["mfma_busy"] = g.get("mfma_busy")
            result["roofline"]["mfma_busy_source"] = str(pmc_files[-1].relative_to(ROOT))
        except Exception:
            pass
    # HBM traffic per launch from the committed PMC passes (profiles/rNN/pmc_traffic.json)
    traffic, src = pmc_traffic()
    if traffic:
        result["roofline"]["traffic"] = traffic.get("sgemm_dma_kernel", {}).get("hbm_bytes")
        result["roofline"]["traffic_source"] = src
        # the counters were taken by rocprofv3 in their own passes (they cannot be read in-process): say whether the committed
        # file was measured on THESE kernel sources (tools/source_stamp.py; stamped by tools/gpu_lease.sh since round 5)
        try:
            sys.path.insert(0, str(ROOT / "tools"))
            from source_stamp import stamp
            result["roofline"]["traffic_kernel_sha16"] = traffic.get("kernel_sha16")
            result["roofline"]["traffic_measured_on_these_kernels"] = bool(traffic.get("kernel_sha16") == stamp()["kernel_sha16"])
        except Exception:
            pass
        for key, entry in extras.items():
            if isinstance(entry, dict) and key in traffic and "roofline" in entry:
                entry["roofline"]["traffic"] = traffic[key].get("hbm_bytes")
        if "secondary" in result and "add_1e8" in traffic:
            result["secondary"]["roofline"]["traffic"] = traffic["add_1e8"].get("hbm_bytes")
    # The second half of BASELINE.json's metric — GB/s of the elementwise add on 1e8 floats — inside the two objects the
    # driver's record keeps (`roofline`, `cpu_baseline`): once nested, once as flat scalars (VERDICT r03 missing #2).
    if "secondary" in result:
        sec = result["secondary"]
        r2, c2 = sec["roofline"], sec["cpu_baseline"]
        result["roofline"]["secondary"] = {"metric": sec["metric"], "kernel": "binary_vec_kernel<add> (float4, non-temporal)",
                                           "bound": "hbm", "achieved": r2["achieved"], "peak": r2["peak"], "unit": "GB/s",
                                           "frac": r2["frac"], "traffic": r2.get("traffic"),
                                           "algorithmic_bytes_per_launch": 1.2e9,
                                           "frac_median_launch": r2.get("frac_median_launch"),
                                           "frac_of_copy": r2.get("ceiling", {}).get("frac_of_copy"),
                                           "copy_GBps": r2.get("ceiling", {}).get("copy_GBps")}
        for k, v in result["roofline"]["secondary"].items():
            result["roofline"]["secondary_" + k] = v
        # The driver's record keeps the first ~20 scalars of `roofline` (BENCH_r05: everything behind secondary_traffic was cut,
        # frac_of_copy among it — VERDICT r05 weak #5): the figures that explain the line come first, prose and provenance last.
        first = ("bound", "achieved", "peak", "unit", "frac", "traffic", "secondary_achieved", "secondary_frac", "secondary_frac_of_copy",
                 "secondary_copy_GBps", "secondary_traffic", "secondary_unit", "secondary_peak", "secondary_bound", "frac_best_launch",
                 "frac_launches_5_24", "mfma_busy", "traffic_measured_on_these_kernels", "secondary_algorithmic_bytes_per_launch", "algorithmic_flop_per_launch")
        rl = result["roofline"]
        result["roofline"] = {**{k: rl[k] for k in first if k in rl}, **{k: v for k, v in rl.items() if k not in first}}
        result["cpu_baseline"]["secondary"] = dict(c2, metric=sec["metric"])
        for k, v in result["cpu_baseline"]["secondary"].items():
            result["cpu_baseline"]["secondary_" + k] = v
    if extras:
        result["extras"] = _compact(extras)
        result["summary"] = _summary(result, extras)       # LAST key: the tail of the line shows every BASELINE config
    if os.environ.get("NP_BENCH_SHARE_DEVICE") == "1" and args.gpus > 1:
        # N ranks on ONE device (possible only over a collective library that allows it — RCCL does not): every line of the
        # N > 1 path runs on hardware, the ranks take the GPU from each other, and the figures are not a measurement of anything
        result = {**result, "valid": False, "shared_device": True,
                  "note": "NP_BENCH_SHARE_DEVICE=1: %d ranks share device 0 — a code-path run of the N > 1 bench, not a measurement" % args.gpus}
    if rank0:
        print(json.dumps(result), flush=True)
    dist.close()
    # nothing but the JSON line may reach stdout: whatever a library still holds in C stdio buffers
    # (RCCL's banner) is sent to /dev/null at exit
    sys.stdout.flush()
    os.dup2(os.open(os.devnull, os.O_WRONLY), 1)


if __name__ == "__main__":
    main()
