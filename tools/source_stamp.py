"""A stamp of the sources a measurement was taken on, computable on the GPU box (the snapshot carries no .git):
sha256 over the kernel, ABI, glue, host-layer and bench sources, in sorted path order.  tools/gpu_lease.sh writes it next to
every evidence file of a lease (stamp.json, and the `source_sha16` field of pmc_traffic.json / gemm_pmc.json); in the
repository `python tools/source_stamp.py` at the commit named in profiles/rNN/README.md prints the same 16 hex digits.
Usage: python tools/source_stamp.py [--json]"""
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
GLOBS = ("numpower_amd/csrc/*.hip", "numpower_amd/csrc/*.h", "numpower_amd/host/*.cpp", "include/*.h", "ext/*.c", "ext/*.h",
         "numpower_amd/*.py", "bench.py", "tools/prof_kernels.py", "tools/prof_counters.py")


def stamp():
    files = sorted({p for g in GLOBS for p in ROOT.glob(g)})
    h = hashlib.sha256()
    for p in files:
        h.update(p.relative_to(ROOT).as_posix().encode() + b"\0")
        h.update(p.read_bytes())
        h.update(b"\0")
    return {"source_sha16": h.hexdigest()[:16], "files": len(files)}


if __name__ == "__main__":
    s = stamp()
    print(json.dumps(s) if "--json" in sys.argv else s["source_sha16"])
