"""Build profiles/rNN/pmc_traffic.json — HBM bytes per CALL of every workload bench.py reports — from the two rocprofv3 --pmc
passes over tools/prof_kernels.py (FETCH_SIZE and WRITE_SIZE counter_collection CSVs) and the section list that run wrote.
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE reports half of a wide coalesced read stream
on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE calibrates 1:1 (add writes its 0.4 GB).
Round 6: the CSV rows are ordered by dispatch and cut at prof_kernels.py's marker launches (`raise_error_kernel`), so a workload's
figure is the SUM over every kernel one call launched (np_moments' pass + fold, the median's six kernels, a K-chunked product's
fold) divided by the number of calls — no entry of the bench is left without a counter figure because it is not one kernel.
The 2x on FETCH_SIZE is calibrated on 16 B/lane streaming reads; kernels that read with narrower accesses (the strided side of
a skinny transpose) may be over-stated by it, never under-stated.
Usage: python tools/pmc_traffic.py fetch.csv write.csv out.json [sections.json]"""
import collections, csv, json, re, sys

MARKER = "raise_error_kernel"


def rows(path, counter):
    rs = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    if rs and "Dispatch_Id" in rs[0]:
        rs.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rs


def by_section(path, counter, n_sections):
    """-> [{"total": sum of the counter, "kernels": {short kernel name: [launches, sum]}}] per section, cut at the markers."""
    out, cur = [], None
    for r in rows(path, counter):
        name = r["Kernel_Name"]
        if MARKER in name:
            cur = {"total": 0.0, "kernels": collections.OrderedDict()}
            out.append(cur)
            continue
        if cur is None:
            continue            # uploads' fill / copy kernels before the first marker
        v = float(r["Counter_Value"])
        cur["total"] += v
        m = re.search(r"(\w+)(<.*>)?\(", name.replace("(anonymous namespace)::", ""))   # void ns::kernel<...>(args) -> kernel
        short = m.group(1) if m else name[:60]
        k = cur["kernels"].setdefault(short, [0, 0.0])
        k[0] += 1
        k[1] += v
    if len(out) != n_sections:
        raise SystemExit("%s: %d marker launches for %d sections (was the CSV written by another prof_kernels.py?)" % (path, len(out), n_sections))
    return out


def main(argv):
    fetch_csv, write_csv, out_json = argv[1], argv[2], argv[3]
    sections = json.load(open(argv[4])) if len(argv) > 4 else None
    if sections is None:
        raise SystemExit(__doc__)
    f = by_section(fetch_csv, "FETCH_SIZE", len(sections))
    w = by_section(write_csv, "WRITE_SIZE", len(sections))
    out = {"_note": __doc__.split("Usage")[0].strip() + "  (the non-workload keys: _note, and source_sha16 / kernel_sha16 written by tools/gpu_lease.sh)"}
    for (label, calls), fs, ws in zip(sections, f, w):
        if label == "end" or label.startswith("_") or calls <= 0:
            continue
        kernels = {}
        for name, (launches, total) in fs["kernels"].items():
            kernels[name] = {"launches_per_call": launches / calls, "FETCH_SIZE_KB_per_call": total / calls}
        for name, (launches, total) in ws["kernels"].items():
            kernels.setdefault(name, {"launches_per_call": launches / calls})["WRITE_SIZE_KB_per_call"] = total / calls
        out[label] = {"calls": calls, "FETCH_SIZE_KB": fs["total"] / calls, "WRITE_SIZE_KB": ws["total"] / calls,
                      "hbm_bytes": (2 * fs["total"] + ws["total"]) * 1024 / calls, "kernels": kernels}
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes"] / 1e6, 1) for k, v in out.items() if isinstance(v, dict)}))


if __name__ == "__main__":
    main(sys.argv)
