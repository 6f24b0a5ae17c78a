"""How far the product is from the REFERENCE's own one-ulp sensitivity on the late-barrier
goldens: per recorded call, (projection error) / (the reference's projection error),
(deviation of x) / (its one-ulp movement), iteration-count difference against its spread."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ip-nonlinear-solver_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_gpu_late_barrier as T  # noqa: E402
for n in (400, 12000):
    gold = T._gold(n)
    for j in range(T.SIZES[n]):
        d = T._pieces(gold, j)
        sens = gold["c%d_sens" % j]
        z = T._solve_single(d, projection_only=True)
        zs = np.max(np.abs(d["z_true"]))
        zerr = np.max(np.abs(z - d["z_true"])) / zs
        x, info = T._solve_single(d)
        xerr = np.max(np.abs(x.to_host() - d["x_out"])) / np.max(np.abs(d["x_out"]))
        spread = int(np.max(np.abs(sens[:, 0] - d["info"][0])))
        print("n=%5d call %d mu=%.1e |c|/|Zc|=%.1e  proj err %.2e (ref %.2e, ratio %.1f)   x dev %.2e (sens %.2e, ratio %.1f)  "
              "niter %d vs %d (spread %d)" % (n, j, d["mu"], np.max(np.abs(d["c"])) / zs, zerr, d["ref_proj_err"],
                                             zerr / max(d["ref_proj_err"], 1e-300), xerr, sens[:, 3].max(),
                                             xerr / max(sens[:, 3].max(), 1e-300), info["niter"], d["info"][0], spread), flush=True)
