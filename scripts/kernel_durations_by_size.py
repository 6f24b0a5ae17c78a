"""Per-kernel launch durations grouped by grid size from a rocprofv3 kernel trace of bench.py
(the bench runs the loop at n=1e6 and n=4e6: the same kernel appears with two grids).
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/benchprof -- python3 bench.py
    python scripts/kernel_durations_by_size.py gpurun_out/benchprof profiles/<name>.json"""
import collections, csv, glob, json, os, statistics, sys
d, out = sys.argv[1:3]
f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if not short.startswith(("k_cg_step", "k_solve_pcr", "k_csr_spmv", "k_compact", "k_pairs", "k_solve_dec")):
        continue
    grid = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
    acc[(short[:40], grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for (k, g), v in sorted(acc.items()):
    if len(v) < 100:
        continue
    rows.append({"kernel": k, "grid_threads": g, "launches": len(v), "median_us": round(statistics.median(v), 3),
                 "mean_us": round(sum(v) / len(v), 3), "min_us": round(min(v), 3), "max_us": round(max(v), 3)})
json.dump(rows, open(out, "w"), indent=1)
print(json.dumps(rows, indent=1))
