"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean over dispatches).
Usage: python tools/pmc_summary.py gpurun_out/prof_*/ *_counter_collection.csv ..."""
import collections, csv, re, sys
for path in sys.argv[1:]:
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        m = re.search(r"::(\w+)<([^>]*)>", name)
        k = (m.group(1) + "<" + m.group(2) + ">") if m else name[:40]
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", path)
    for k, cs in d.items():
        print("  %-62s n=%d " % (k, len(next(iter(cs.values())))) + " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
