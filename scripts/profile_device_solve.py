"""Where the wall-clock of a device-callback solve goes (dev tool).

    python scripts/profile_device_solve.py [n m]        cProfile of a warm config-3 solve
Run the same file under `rocprofv3 --kernel-trace --stats` for the per-kernel side.
"""
import cProfile
import io
import os
import pstats
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import torch
import ipsolver
from ipsolver.synthetic import CenteredBandedNLP, DeviceCallbacks, LeanDeviceCallbacks
if os.environ.get("LEAN", "1") == "1":        # (LEAN=0: the plain torch callbacks)
    DeviceCallbacks = LeanDeviceCallbacks

warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
prob = CenteredBandedNLP(n, m, eps=1e-3)
dc = DeviceCallbacks(prob)


def solve():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, dc.constraints(ipsolver),
                                        method="tr_interior_point")
    torch.cuda.synchronize()
    return res, time.perf_counter() - t0


for k in range(3):
    res, dt = solve()
    print("run %d: %.4f s  status %d niter %d cg %d" % (k, dt, res.status, res.niter, res.cg_niter))
pr = cProfile.Profile()
pr.enable()
res, dt = solve()
pr.disable()
print("profiled run: %.4f s" % dt)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(60)
print(s.getvalue()[:12000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
print(s.getvalue()[:6000])
from ipsolver import cg_fused
print("loop pool:", cg_fused.POOL_STATS, "cg stats:", cg_fused.STATS)
