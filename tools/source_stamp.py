"""A stamp of the sources a measurement was taken on, computable on the GPU box (the snapshot carries no .git):
sha256 over the kernel, ABI, glue, host-layer and bench sources, in sorted path order.  tools/gpu_lease.sh writes it next to
every evidence file of a lease (stamp.json, and the `kernel_sha16` / `source_sha16` fields of pmc_traffic.json / gemm_pmc.json); in the
repository `python tools/source_stamp.py` at the commit named in profiles/rNN/README.md prints the same 16 hex digits.
Usage: python tools/source_stamp.py [--json]"""
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
KERNEL_GLOBS = ("numpower_amd/csrc/*.hip", "numpower_amd/csrc/*.h", "include/*.h")     # what libnp_hip.so is built from
GLOBS = KERNEL_GLOBS + ("numpower_amd/host/*.cpp", "ext/*.c", "ext/*.h", "numpower_amd/*.py", "bench.py", "tools/prof_kernels.py",
                        "tools/prof_counters.py")


def _sha(globs):
    files = sorted({p for g in globs for p in ROOT.glob(g)})
    h = hashlib.sha256()
    for p in files:
        h.update(p.relative_to(ROOT).as_posix().encode() + b"\0")
        h.update(p.read_bytes())
        h.update(b"\0")
    return h.hexdigest()[:16], len(files)


def stamp():
    """kernel_sha16: the device library's sources only (what a counter or a kernel time depends on); source_sha16: those plus the
    host layer, the glue, the Python binding, bench.py and the profiling drivers."""
    k, nk = _sha(KERNEL_GLOBS)
    a, na = _sha(GLOBS)
    return {"kernel_sha16": k, "source_sha16": a, "kernel_files": nk, "files": na}


if __name__ == "__main__":
    s = stamp()
    print(json.dumps(s) if "--json" in sys.argv else "kernels %s  all sources %s" % (s["kernel_sha16"], s["source_sha16"]))
