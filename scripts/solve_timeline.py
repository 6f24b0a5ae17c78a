"""GPU timeline of ONE warm config-3 solve (dev tool).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python3 scripts/solve_timeline.py run
    python3 scripts/solve_timeline.py summarize /tmp/tl [out.txt]

`run`: three warm solves, the last one behind a 60 ms pause (the gap the summary looks for).
`summarize`: kernels of the last solve from the trace -- launches, busy time, span, idle gaps,
per-kernel totals.
"""
import csv
import glob
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run():
    sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
    import torch
    import ipsolver
    from ipsolver.synthetic import CenteredBandedNLP, DeviceCallbacks, LeanDeviceCallbacks
    if os.environ.get("LEAN", "1") == "1":        # (LEAN=0: the plain torch callbacks)
        DeviceCallbacks = LeanDeviceCallbacks
    warnings.simplefilter("ignore")
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    prob = CenteredBandedNLP(n, n // 10, eps=1e-3)
    dc = DeviceCallbacks(prob)

    def solve():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess,
                                            dc.constraints(ipsolver), method="tr_interior_point")
        torch.cuda.synchronize()
        return res, time.perf_counter() - t0
    for _ in range(3):
        res, dt = solve()
    time.sleep(0.06)
    res, dt = solve()
    print("solve: %.2f ms, status %d, %d outer / %d CG" % (1e3 * dt, res.status, res.niter,
                                                           res.cg_niter))


def summarize(d, out=None):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the last gap of >= 40 ms starts the measured solve
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - rows[i - 1][1] > 40e6:
            cut = i
    rows = rows[cut:]
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    gaps = [rows[i][0] - rows[i - 1][1] for i in range(1, len(rows))]
    lines = []
    lines.append("kernels %d, span %.3f ms, busy %.3f ms (%.0f %%), idle %.3f ms" %
                 (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
    for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 100), (100, 1e9)):
        g = [x for x in gaps if lo * 1e3 <= x < hi * 1e3]
        lines.append("  gaps %4g-%-5g us: %4d, %.3f ms" % (lo, hi if hi < 1e9 else float("inf"),
                                                          len(g), sum(g) / 1e6))
    by = {}
    for s, e, k in rows:
        k = k.replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0][:100]
        c = by.setdefault(k, [0, 0])
        c[0] += 1
        c[1] += e - s
    lines.append("per kernel (count, total us, mean us):")
    for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        lines.append("  %5d %9.1f %7.2f  %s" % (c, t / 1e3, t / 1e3 / c, k))
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")
        # (the launches in order: offset, duration, gap before, grid, name)
        with open(out + ".sequence", "w") as fh:
            for i, (s, e, k) in enumerate(rows):
                k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
                fh.write("%9.1f %6.1f %6.1f  %s\n" % ((s - rows[0][0]) / 1e3, (e - s) / 1e3,
                                                     (s - rows[i - 1][1]) / 1e3 if i else 0.0, k))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        summarize(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
