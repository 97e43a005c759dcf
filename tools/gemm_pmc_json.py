"""Build profiles/rNN/gemm_pmc.json (MFMA utilisation of the headline GEMM kernel) from the rocprofv3 --pmc
counter_collection CSVs of tools/prof_counters.py (two passes: SQ_BUSY / MFMA counters and SQ_INSTS / LDS counters).
Usage: python tools/gemm_pmc_json.py out.json pass1.csv [pass2.csv ...]"""
import collections, csv, json, sys

vals = collections.defaultdict(list)
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if "sgemm_dma_kernel" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in vals.items()}
out = {"kernel": "sgemm_dma_kernel<false,false,true> 4096^3 (PRIO: priority alternation, period 16)",
       "launches_averaged": len(next(iter(vals.values()))) if vals else 0, "counters": c}
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c:
    out["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"])
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
    out["mfma_busy_of_elapsed"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
out["_note"] = ("rocprofv3 --pmc passes of tools/prof_counters.py (tools/gpu_lease.sh evidence), means over the launches seen. "
                "SQ_VALU_MFMA_BUSY_CYCLES sums busy cycles over all 1024 SIMDs (= 64 cycles x the number of v_mfma_f32_32x32x2_f32); "
                "mfma_busy = that / (4 SIMDs x SQ_BUSY_CU_CYCLES); mfma_busy_of_elapsed divides by 1024 SIMDs x (GRBM_GUI_ACTIVE / 8 XCDs) "
                "instead, i.e. includes launch ramp and tail.")
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: out.get(k) for k in ("mfma_busy", "mfma_busy_of_elapsed", "launches_averaged")}))
