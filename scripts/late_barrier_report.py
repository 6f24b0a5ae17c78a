"""Deviation report of the product against the late-barrier goldens (tests/golden/
late_barrier_n*.npz): per recorded projected_cg call of the reference's barrier run, the
iteration counts and the relative deviation of the returned step and of the first iterates."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ip-nonlinear-solver_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_gpu_late_barrier as T  # noqa: E402

for n in (400, 12000):
    gold = T._gold(n)
    for j in range(T.SIZES[n]):
        d = T._pieces(gold, j)
        x, info = T._solve_single(d)
        xh = x.to_host()
        err = np.max(np.abs(xh - d["x_out"])) / np.max(np.abs(d["x_out"]))
        xg, ig = T._solve_single(d, return_all=True)
        st = d["stride"]
        devs = [np.max(np.abs(ig["allvecs"][k].to_host()[::st] - w)) / max(np.max(np.abs(w)), 1e-300)
                for k, w in enumerate(d["allvecs"]) if k < len(ig["allvecs"])]
        print("n=%d call %d mu=%.2e radius=%.3e  niter got %d / general %d / ref %d  stop %d/%d  "
              "x rel dev %.2e  first iterates max dev %.2e" % (
                  n, j, d["mu"], d["radius"], info["niter"], ig["niter"], d["info"][0],
                  info["stop_cond"], d["info"][1], err, max(devs) if devs else 0.0), flush=True)
