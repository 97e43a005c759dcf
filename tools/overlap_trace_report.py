"""Reads a rocprofv3 --kernel-trace CSV of tools/overlap_trace.py and reports, per kernel name, on which
stream / queue it ran and how much of the non-GEMM kernels' time fell inside a GEMM interval on ANOTHER queue.

    python tools/overlap_trace_report.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
if not rows:
    sys.exit("empty trace")
key = lambda r, *names: next((r[n] for n in names if n in r and r[n] != ""), "?")   # noqa: E731
ev = []
for r in rows:
    ev.append({"name": key(r, "Kernel_Name", "kernel_name"), "q": key(r, "Stream_Id", "Queue_Id", "queue_id"),
               "queue": key(r, "Queue_Id", "queue_id"),
               "t0": int(key(r, "Start_Timestamp", "start_timestamp")), "t1": int(key(r, "End_Timestamp", "end_timestamp"))})
gemm = [e for e in ev if "sgemm" in e["name"]]
other = [e for e in ev if "sgemm" not in e["name"]]
by = defaultdict(lambda: [0, 0, set(), set()])
for e in ev:
    b = by[e["name"][:70]]
    b[0] += 1
    b[1] += e["t1"] - e["t0"]
    b[2].add(e["q"])
    b[3].add(e["queue"])
print("%-72s %6s %12s  streams / queues" % ("kernel", "calls", "total us"))
for name, (calls, ns, qs, queues) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("%-72s %6d %12.1f  %s / %s" % (name, calls, ns / 1e3, sorted(qs), sorted(queues)))
gemm.sort(key=lambda e: e["t0"])
inside = total = 0
n_overlapping = 0
for e in other:
    if e["t1"] - e["t0"] < 20_000:       # fills etc. are too short to matter
        continue
    total += e["t1"] - e["t0"]
    got = 0
    for g in gemm:
        if g["q"] == e["q"] and g["queue"] == e["queue"]:
            continue
        lo, hi = max(e["t0"], g["t0"]), min(e["t1"], g["t1"])
        if hi > lo:
            got += hi - lo
    got = min(got, e["t1"] - e["t0"])
    inside += got
    n_overlapping += got > 0
print()
print("non-GEMM kernels >= 20 us: %.1f us in total, of which %.1f us (%.0f %%) ran while a GEMM was executing on "
      "another stream (%d kernels overlapped)" % (total / 1e3, inside / 1e3, 100.0 * inside / max(total, 1), n_overlapping))

# timeline of the last progress-reporting GEMM and what ran on the other streams while it did
prog = [g for g in gemm if g["t1"] - g["t0"] > 500_000]
if prog:
    g = prog[-1]
    print()
    print("last whole-slab GEMM launch: %.1f us on stream %s; other kernels that started inside it (us after its start):" % ((g["t1"] - g["t0"]) / 1e3, g["q"]))
    for e in sorted(other, key=lambda e: e["t0"]):
        if g["t0"] <= e["t0"] <= g["t1"] + 300_000:
            print("  +%8.1f .. +%8.1f  stream %-3s %s" % ((e["t0"] - g["t0"]) / 1e3, (e["t1"] - g["t0"]) / 1e3, e["q"], e["name"][:60]))
