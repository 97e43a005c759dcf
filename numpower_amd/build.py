"""Build the native libraries in-tree.

  numpower_amd/lib/libnp_hip.so         HIP kernels + C ABI (include/np_hip.h), hipcc, gfx950
  numpower_amd/lib/libnumpower_host.so  C++ host mirror of the reference's NDArray L2/L3
                                        interface for the hot path (include/numpower_host.h) + the
                                        --with-hip glue of ext/ (cuda_* / vmalloc / drivers)
  numpower_amd/lib/libnp_hipmath.so     the ext/ glue alone (reference's cuda_math.h + gpu_alloc.h
                                        symbols over libnp_hip.so, no host layer)
  oracle/lib/libnp_oracle.so            CPU restatement of the reference (test infrastructure)
  tests/loopback_rccl/lib/librccl.so.1  a stand-in for RCCL that lets several ranks share ONE GPU (test infrastructure: only
                                        worker processes of tests/test_gpu_comm_loopback_peers.py get it, via LD_LIBRARY_PATH)

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

HIP_SOURCES = ["np_runtime.hip", "np_elementwise.hip", "np_fused_static.hip", "np_reduce.hip", "np_sgemm.hip", "np_layout.hip",
               "np_select.hip", "np_comm.hip"]
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


def build_hip(force: bool = False, verbose: bool = False, tuning: bool = False) -> Path:
    """Compile the HIP translation units for gfx950 and link libnp_hip.so.

    tuning=True builds libnp_hip_tuning.so instead, with -DNP_TUNING: the plan-forcing environment
    variables (NP_SGEMM_PLAN, NP_SGEMM_PLAN_DEBUG) and the deliberately wrong timing-ablation GEMM
    variants exist only there (tools/gemm_plan_sweep.py, tools/gemm_ab.py); the shipped library reads
    no environment variable."""
    hipcc = _hipcc()
    objdir = OBJDIR.parent / "obj_tuning" if tuning else OBJDIR
    flags = HIP_FLAGS + (["-DNP_TUNING"] if tuning else [])
    objdir.mkdir(parents=True, exist_ok=True)
    LIBDIR.mkdir(parents=True, exist_ok=True)
    headers = list(INCLUDE.glob("*.h")) + list(CSRC.glob("*.h"))
    jobs = []
    objs = []
    for src in HIP_SOURCES:
        s = CSRC / src
        o = objdir / (s.stem + ".o")
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([hipcc, *flags, "-c", str(s), "-o", str(o)])
    if jobs:
        if verbose:
            print("[build] compiling %d HIP translation unit(s) for gfx950" % len(jobs), flush=True)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(_run, jobs))
    lib = LIBDIR / ("libnp_hip_tuning.so" if tuning else "libnp_hip.so")
    if force or jobs or _newer(lib, objs):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(lib), "-ldl"])
    return lib


EXT = ROOT / "ext"
# the --with-hip glue (plain C over np_hip.h; ext/README.md).  zend_hooks.c needs PHP's headers and is
# compiled only inside a PHP extension build; standalone_hooks.c is for the glue-only test library.
EXT_GLUE = ["hip_math.c", "gpu_alloc_hip.c"]
EXT_CFLAGS = ["-O2", "-std=c99", "-fPIC", "-Wall", "-Wextra", "-Werror", f"-I{INCLUDE}", f"-I{EXT}"]


def _ext_objects(names, force, verbose):
    """gcc -c the given ext/*.c files -> build/obj/ext_<name>.o (rebuilt when a source / header is newer)."""
    OBJDIR.mkdir(parents=True, exist_ok=True)
    cc = shutil.which("gcc") or "gcc"
    headers = list(INCLUDE.glob("*.h")) + list(EXT.glob("*.h"))
    objs = []
    for name in names:
        src = EXT / name
        obj = OBJDIR / ("ext_" + src.stem + ".o")
        if force or _newer(obj, [src] + headers):
            if verbose:
                print("[build] compiling ext/%s" % name, flush=True)
            _run([cc, *EXT_CFLAGS, "-c", str(src), "-o", str(obj)])
        objs.append(obj)
    return objs


def build_ext_glue(force: bool = False, verbose: bool = False) -> Path:
    """libnp_hipmath.so: ext/hip_math.c + ext/gpu_alloc_hip.c with the stand-alone hooks — the symbols
    of the reference's cuda_math.h / gpu_alloc.h over libnp_hip.so and NOTHING of the host layer: proof
    (and test vehicle) that the inner boundary of INTEGRATION.md section 2a is self-contained."""
    lib = LIBDIR / "libnp_hipmath.so"
    objs = _ext_objects(EXT_GLUE + ["standalone_hooks.c"], force, verbose)
    if force or _newer(lib, objs + [LIBDIR / "libnp_hip.so"]):
        cc = shutil.which("gcc") or "gcc"
        _run([cc, "-shared", "-fPIC", *map(str, objs), "-o", str(lib), f"-L{LIBDIR}", "-lnp_hip",
              "-Wl,-rpath,$ORIGIN", "-Wl,--no-undefined"])
    return lib


def build_host(force: bool = False, verbose: bool = False) -> Path:
    """Compile the C++ host layer (NDArray struct + the reference's L2 entry points) and link the
    --with-hip glue into it (cuda_* / vmalloc ... / NDArrayMathGPU_ElementWise* with the reference's
    signatures, ext/): one library carries the whole outer boundary."""
    lib = LIBDIR / "libnumpower_host.so"
    srcs = sorted(HOST.glob("*.cpp"))
    if not srcs:
        return lib
    headers = list(INCLUDE.glob("*.h")) + list(HOST.glob("*.h"))
    ext_objs = _ext_objects(EXT_GLUE + ["hip_math_drivers.c", "hip_fast.c", "hip_lazy.c"], force, verbose)
    if force or _newer(lib, srcs + headers + ext_objs + [LIBDIR / "libnp_hip.so"]):
        if verbose:
            print("[build] compiling host layer", flush=True)
        cxx = shutil.which("g++") or "g++"
        _run([cxx, "-O2", "-std=c++17", "-fPIC", "-shared", f"-I{INCLUDE}", f"-I{HOST}", f"-I{EXT}",
              *map(str, srcs), *map(str, ext_objs), "-o", str(lib), f"-L{LIBDIR}", "-lnp_hip",
              "-Wl,-rpath,$ORIGIN", "-Wl,--no-undefined"])
    return lib


def build_method_bodies(force: bool = False, verbose: bool = False) -> Path:
    """ext/method_bodies.c -> numpower_amd/lib/method_bodies: the device branch of the reference's
    PHP_METHODs as a C99 program over numpower_host.h + hip_math.h (-Wall -Wextra -Werror: a
    signature that drifts from the reference's call sites fails here)."""
    exe = LIBDIR / "method_bodies"
    src = EXT / "method_bodies.c"
    deps = [src, LIBDIR / "libnumpower_host.so"] + list(INCLUDE.glob("*.h")) + list(EXT.glob("*.h"))
    if force or _newer(exe, deps):
        if verbose:
            print("[build] compiling ext/method_bodies.c", flush=True)
        cc = shutil.which("gcc") or "gcc"
        flags = [f for f in EXT_CFLAGS if f != "-fPIC"]
        _run([cc, *flags, str(src), "-o", str(exe), f"-L{LIBDIR}", "-lnumpower_host", "-lnp_hip",
              "-Wl,-rpath,$ORIGIN"])
    return exe


def build_fast_path_bodies(force: bool = False, verbose: bool = False) -> Path:
    """The text tools/apply_with_hip.py inserts into the reference's device-dispatching L2 functions,
    wrapped into a C99 program by the tool itself (fast_path_program_source) -> build/gen/fast_path_bodies.c ->
    numpower_amd/lib/fast_path_bodies; -Wall -Wextra -Werror against numpower_host.h + hip_fast.h."""
    exe = LIBDIR / "fast_path_bodies"
    tool = ROOT / "tools" / "apply_with_hip.py"
    gen = ROOT / "build" / "gen" / "fast_path_bodies.c"
    deps = [tool, LIBDIR / "libnumpower_host.so"] + list(INCLUDE.glob("*.h")) + list(EXT.glob("*.h"))
    if force or _newer(exe, deps):
        if verbose:
            print("[build] generating + compiling fast_path_bodies", flush=True)
        import importlib.util
        spec = importlib.util.spec_from_file_location("np_apply_with_hip", tool)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = mod          # dataclasses looks the module up while the class body runs
        spec.loader.exec_module(mod)
        gen.parent.mkdir(parents=True, exist_ok=True)
        gen.write_text(mod.fast_path_program_source())
        cc = shutil.which("gcc") or "gcc"
        flags = [f for f in EXT_CFLAGS if f != "-fPIC"]
        _run([cc, *flags, str(gen), "-o", str(exe), f"-L{LIBDIR}", "-lnumpower_host", "-lnp_hip",
              "-Wl,-rpath,$ORIGIN"])
    return exe


def build_lazy_bodies(force: bool = False, verbose: bool = False) -> Path:
    """The text tools/apply_with_hip.py inserts for the pending chains (buffer_get's flush, the appenders of
    numpower.c's operator handler, static arithmetic methods and unary methods), wrapped by the tool itself into a C99 program
    around a stand-in for the Zend object table (lazy_program_source) -> build/gen/lazy_bodies.c -> numpower_amd/lib/lazy_bodies."""
    exe = LIBDIR / "lazy_bodies"
    tool = ROOT / "tools" / "apply_with_hip.py"
    gen = ROOT / "build" / "gen" / "lazy_bodies.c"
    deps = [tool, LIBDIR / "libnumpower_host.so"] + list(INCLUDE.glob("*.h")) + list(EXT.glob("*.h"))
    if force or _newer(exe, deps):
        if verbose:
            print("[build] generating + compiling lazy_bodies", flush=True)
        import importlib.util
        spec = importlib.util.spec_from_file_location("np_apply_with_hip", tool)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = mod
        spec.loader.exec_module(mod)
        gen.parent.mkdir(parents=True, exist_ok=True)
        gen.write_text(mod.lazy_program_source())
        cc = shutil.which("gcc") or "gcc"
        flags = [f for f in EXT_CFLAGS if f != "-fPIC"]
        _run([cc, *flags, str(gen), "-o", str(exe), f"-L{LIBDIR}", "-lnumpower_host", "-lnp_hip",
              "-Wl,-rpath,$ORIGIN"])
    return exe


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


def build_loopback_rccl(force: bool = False, verbose: bool = False) -> Path:
    """Compile tests/loopback_rccl (test infrastructure; the product never names it: np_comm.hip dlopens "librccl.so.1" and a
    test's worker processes are started with LD_LIBRARY_PATH pointing here)."""
    tdir = ROOT / "tests" / "loopback_rccl"
    src, lib = tdir / "loopback_rccl.hip", tdir / "lib" / "librccl.so.1"
    if not src.exists():
        return lib
    (tdir / "lib").mkdir(exist_ok=True)
    if force or _newer(lib, [src]):
        if verbose:
            print("[build] compiling tests/loopback_rccl", flush=True)
        _run([_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror", str(src),
              "-o", str(lib), "-Wl,-soname,librccl.so.1", "-lrt"])
    return lib


def build_all(force: bool = False, verbose: bool = False):
    hip = build_hip(force, verbose)
    host = build_host(force, verbose)
    build_ext_glue(force, verbose)
    build_method_bodies(force, verbose)
    build_fast_path_bodies(force, verbose)
    build_lazy_bodies(force, verbose)
    oracle = build_oracle(force, verbose)
    build_loopback_rccl(force, verbose)
    return hip, host, oracle


if __name__ == "__main__":
    if "--tuning" in sys.argv:
        print(build_hip(force="--force" in sys.argv, verbose=True, tuning=True), "ok")
        sys.exit(0)
    libs = build_all(force="--force" in sys.argv, verbose=True)
    for lib in libs:
        print(lib, "ok" if Path(lib).exists() else "(not built)")
