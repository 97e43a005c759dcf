"""Build profiles/rNN/pmc_traffic.json (HBM bytes per launch of the benchmarked kernels) from the two
rocprofv3 --pmc passes over tools/prof_kernels.py (FETCH_SIZE and WRITE_SIZE counter_collection CSVs).
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE reports half of a wide coalesced read stream
on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE calibrates 1:1 (add writes its 0.4 GB).
Usage: python tools/pmc_traffic.py fetch.csv write.csv out.json"""
import collections, csv, json, sys

KEYS = [  # bench key, substring(s) identifying the kernel
    ("sgemm_dma_kernel", ["sgemm_dma_kernel"]),
    ("add_1e8", ["binary_vec_kernel<0, 0, 0,"]),
    ("pow_1e8", ["binary_vec_kernel<5, 0, 0,"]),
    ("exp_1e8", ["unary_vec_kernel<2,"]),
    ("log_1e8", ["unary_vec_kernel<5,"]),
    ("add_row_broadcast", ["binary_vec_kernel<0, 0, 2,"]),
    ("add_col_broadcast", ["binary_vec_kernel<0, 0, 3,"]),
    ("sum_axis0", ["reduce_axis_cols<0, false"]),
    ("fused_chain_1e8", [">, -1>", "fused_chain_kernel"]),          # cchain_flat_kernel<CChain<...>, -1> (store)
    ("fused_chain_sum_1e8", [">, 0>(", ">, 0>"]),                   # cchain_flat_kernel<CChain<...>, 0> (sum)
    ("sum_exp_axis0_fused", ["cchain_cols_kernel", "fused_chain_cols_kernel"]),
    ("sum_exp_axis1_fused", ["cchain_rows_kernel", "fused_chain_rows_kernel"]),
    ("transpose_65536x4096", ["transpose_tile_kernel"]),
    ("argmax_1e8", ["argreduce_rows_kernel<true"]),
    ("argmax_axis1_65536x1024", ["argreduce_rows_wave<true"]),
    ("sgemv_10x1e7", ["sgemv_fewrows_chunks_kernel"]),
    ("moments_second_pass_1e8", ["reduce_xform_pass1<1"]),
]


def means(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


def pick(table, pats):
    for name, v in table.items():
        if any(p in name for p in pats):
            return v
    return None


fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
out = {"_note": __doc__.split("Usage")[0].strip() + "  (the non-kernel keys: _note, and source_sha16 / kernel_sha16 written by tools/gpu_lease.sh)"}
for key, pats in KEYS:
    f, w = pick(fetch, pats), pick(write, pats)
    if f is not None and w is not None:
        out[key] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes": (2 * f + w) * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes"] / 1e6) for k, v in out.items() if isinstance(v, dict)}))
