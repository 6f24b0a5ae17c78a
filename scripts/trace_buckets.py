"""Mean duration of the named kernels per tenth of a trace's time span (dev tool):
    python scripts/trace_buckets.py <rocprofv3 output dir> [substring ...]"""
import csv, glob, os, sys
d = sys.argv[1]
names = sys.argv[2:] or ["k_cg_step1_box", "k_solve_pcr", "k_cg_step2_hp"]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the solve = the last long stretch: keep everything after the last gap of >= 1 s
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 1e9:
        cut = i
rows = rows[cut:]
t0, t1 = rows[0][0], rows[-1][1]
print("kernels %d, span %.2f s" % (len(rows), (t1 - t0) / 1e9))
B = 10
for nm in names:
    acc = [[0, 0.0] for _ in range(B)]
    for s, e, k in rows:
        if nm in k:
            b = min(B - 1, int((s - t0) * B / (t1 - t0)))
            acc[b][0] += 1
            acc[b][1] += e - s
    print(nm, " ".join("%.1f(%d)" % (t / c / 1e3 if c else 0.0, c) for c, t in acc))
