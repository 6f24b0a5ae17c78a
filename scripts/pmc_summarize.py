"""Turn the two rocprofv3 PMC passes over scripts/pmc_workload.py into profiles/<name>.json.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 scripts/pmc_workload.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 scripts/pmc_workload.py
    python scripts/pmc_summarize.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01e_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1 KB = 1024 B?  No: rocprofv3 reports them in
KB of 1024 bytes on this stack; the calibration kernels (known byte counts) pin both the unit
and the gfx950 read under-count (MI355X_MICROARCH.md, HBM section) in the same run.
"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash          # ties the stored traffic to the kernel sources

fetch_dir, write_dir, out = sys.argv[1:4]
size_note = sys.argv[4] if len(sys.argv) > 4 else "n=1e6, m=1e5"
commit = sys.argv[5] if len(sys.argv) > 5 else "?"


def per_kernel(d, counter):
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True),
            key=os.path.getmtime)                      # newest pass in that directory
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")


def find(acc, *keys):
    hits = [k for k in acc if all(s in k for s in keys)]
    assert hits, keys
    return acc[max(hits, key=lambda k: len(acc[k]))]   # several instantiations: the loop's


nc = 8000000
cal = {}
for label, key in (("8B_per_lane_k_map_mul", "OpMul"), ("16B_per_lane_k_axpby2", "k_axpby2")):
    fv = [v for v in find(fetch, key) if v > 0.5 * max(find(fetch, key))]      # the nc-sized launches
    wv = [v for v in find(write, key) if v > 0.5 * max(find(write, key))]
    f_kb, w_kb = sum(fv) / len(fv), sum(wv) / len(wv)
    cal[label] = {"known_read_bytes": 16.0 * nc, "FETCH_SIZE_KB": f_kb,
                  "read_factor": 16.0 * nc / (f_kb * 1024), "known_write_bytes": 8.0 * nc,
                  "WRITE_SIZE_KB": w_kb, "write_factor": 8.0 * nc / (w_kb * 1024)}
rf = cal["8B_per_lane_k_map_mul"]["read_factor"]
kernels = {}
names = {"spmv_A_r": ("k_csr_spmv<false, false, false>",),
         "spmv_r_minus_Atv": ("k_csr_spmv<false, true, true>",),
         "spmv_H_p": ("k_csr_spmv<true, false, true>",),
         "step1": ("k_cg_step1(",), "step2": ("k_cg_step2(",),
         "step1_spmv_A_r": ("k_cg_step1_ar",), "step2_spmv_H_p": ("k_cg_step2_hp",),
         "banded_solve_with_residual": ("k_solve_decoupled",),
         "banded_solve_pcr": ("k_solve_pcr",)}
for label, keys in names.items():
    if not any(all(s_ in k for s_ in keys) for k in fetch):
        continue                      # kernel not in this build's loop (fused / unfused)
    fv, wv = find(fetch, *keys), find(write, *keys)
    # launches of the loop only: the modal size class (priming launches differ)
    f_kb = sorted(fv)[len(fv) // 2]
    w_kb = sorted(wv)[len(wv) // 2]
    kernels[label] = {"launches": len(fv), "FETCH_SIZE_KB_median": f_kb, "WRITE_SIZE_KB_median": w_kb,
                      "hbm_read_bytes": f_kb * 1024 * rf, "hbm_write_bytes": w_kb * 1024,
                      "hbm_bytes_per_launch": f_kb * 1024 * rf + w_kb * 1024}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                     "scripts/pmc_workload.py (40 CG iterations, %s), MI355X; "
                     "summarised by scripts/pmc_summarize.py" % size_note,
           "source_commit": commit,
           "kernel_source_hash": kernel_source_hash(),
           "correction": "FETCH_SIZE under-reports streaming reads 2x on gfx950 "
                         "(MI355X_MICROARCH.md section HBM); calibrated here on kernels with known "
                         "byte counts: read_factor below; WRITE_SIZE is exact",
           "calibration": cal, "kernels": kernels}, open(out, "w"), indent=1)
print(json.dumps(kernels, indent=1))
