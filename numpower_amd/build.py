"""Build the native libraries in-tree.

  numpower_amd/lib/libnp_hip.so         HIP kernels + C ABI (include/np_hip.h), hipcc, gfx950
  numpower_amd/lib/libnumpower_host.so  C++ host mirror of the reference's NDArray L2/L3
                                        interface for the hot path (include/numpower_host.h)
  oracle/lib/libnp_oracle.so            CPU restatement of the reference (test infrastructure)

hipcc cross-compiles for gfx950 without a GPU.  Objects are rebuilt only when a source or
header is newer than the object, translation units compile in parallel.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "numpower_amd" / "csrc"
HOST = ROOT / "numpower_amd" / "host"
INCLUDE = ROOT / "include"
LIBDIR = ROOT / "numpower_amd" / "lib"
OBJDIR = ROOT / "build" / "obj"

HIP_SOURCES = ["np_runtime.hip", "np_elementwise.hip", "np_reduce.hip", "np_sgemm.hip", "np_layout.hip", "np_select.hip"]
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{INCLUDE}", f"-I{CSRC}"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libnp_hip.so)")


def _newer(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def _run(cmd):
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(map(str, cmd)), proc.stdout))
    return proc.stdout


def build_hip(force: bool = False, verbose: bool = False) -> Path:
    """Compile the HIP translation units for gfx950 and link libnp_hip.so."""
    hipcc = _hipcc()
    OBJDIR.mkdir(parents=True, exist_ok=True)
    LIBDIR.mkdir(parents=True, exist_ok=True)
    headers = list(INCLUDE.glob("*.h")) + list(CSRC.glob("*.h"))
    jobs = []
    objs = []
    for src in HIP_SOURCES:
        s = CSRC / src
        o = OBJDIR / (s.stem + ".o")
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([hipcc, *HIP_FLAGS, "-c", str(s), "-o", str(o)])
    if jobs:
        if verbose:
            print("[build] compiling %d HIP translation unit(s) for gfx950" % len(jobs), flush=True)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(_run, jobs))
    lib = LIBDIR / "libnp_hip.so"
    if force or jobs or _newer(lib, objs):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(lib)])
    return lib


def build_host(force: bool = False, verbose: bool = False) -> Path:
    """Compile the C++ host layer (NDArray struct + the reference's L2 entry points)."""
    lib = LIBDIR / "libnumpower_host.so"
    srcs = sorted(HOST.glob("*.cpp"))
    if not srcs:
        return lib
    headers = list(INCLUDE.glob("*.h")) + list(HOST.glob("*.h"))
    if force or _newer(lib, srcs + headers + [LIBDIR / "libnp_hip.so"]):
        if verbose:
            print("[build] compiling host layer", flush=True)
        cxx = shutil.which("g++") or "g++"
        _run([cxx, "-O2", "-std=c++17", "-fPIC", "-shared", f"-I{INCLUDE}", f"-I{HOST}",
              *map(str, srcs), "-o", str(lib), f"-L{LIBDIR}", "-lnp_hip",
              "-Wl,-rpath,$ORIGIN"])
    return lib


def build_oracle(force: bool = False, verbose: bool = False) -> Path:
    """Compile oracle/'s C restatement (test infrastructure; never loaded by the product)."""
    odir = ROOT / "oracle"
    lib = odir / "lib" / "libnp_oracle.so"
    srcs = sorted(odir.glob("*.c"))
    if not srcs:
        return lib
    (odir / "lib").mkdir(exist_ok=True)
    if force or _newer(lib, srcs + list(odir.glob("*.h"))):
        if verbose:
            print("[build] compiling oracle", flush=True)
        cc = shutil.which("gcc") or "gcc"
        # -mavx2 -mfma: what the reference's `-mavx2 -march=native` (config.m4:36,50) amounts to on
        # any FMA-capable x86-64 host; decides how gcc contracts a - floor(a/b)*b and the rsqrt
        # Newton step (see oracle/np_oracle.c).
        _run([cc, "-O2", "-mavx2", "-mfma", "-fPIC", "-shared", *map(str, srcs), "-o", str(lib),
              "-lm", "-ldl"])
    return lib


def build_all(force: bool = False, verbose: bool = False):
    hip = build_hip(force, verbose)
    host = build_host(force, verbose)
    oracle = build_oracle(force, verbose)
    return hip, host, oracle


if __name__ == "__main__":
    libs = build_all(force="--force" in sys.argv, verbose=True)
    for lib in libs:
        print(lib, "ok" if Path(lib).exists() else "(not built)")
