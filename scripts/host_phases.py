"""Where the wall clock of a warm config-3 solve goes, by phase of the outer loop (dev tool):
wraps the user callbacks and the backend's subproblem entry points with host timers (no extra
synchronisation: a phase that ends in a blocking read carries the GPU time it waited for).

    python scripts/host_phases.py [n m]
"""
import os
import sys
import time
import warnings
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import torch
import ipsolver
from ipsolver import backend_hip, device as dv, projector, sqp
from ipsolver.synthetic import CenteredBandedNLP, DeviceCallbacks, LeanDeviceCallbacks
if os.environ.get("LEAN", "1") == "1":        # (LEAN=0: the plain torch callbacks)
    DeviceCallbacks = LeanDeviceCallbacks

warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
prob = CenteredBandedNLP(n, m, eps=1e-3)
dc = DeviceCallbacks(prob)
acc, cnt = defaultdict(float), defaultdict(int)


def timed(name, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] += time.perf_counter() - t0
            cnt[name] += 1
    return wrapper


for name in ("fun", "grad", "hess"):
    setattr(dc, name, timed("user " + name, getattr(dc, name)))
dc.constr_fun = timed("user constr", dc.constr_fun)
dc.constr_jac = timed("user jac", dc.constr_jac)
backend_hip.projections = timed("projections (factorization)", backend_hip.projections)
backend_hip.modified_dogleg = timed("modified_dogleg", backend_hip.modified_dogleg)
backend_hip.projected_cg = timed("projected_cg", backend_hip.projected_cg)
backend_hip.hessian_operator = timed("hessian_operator", backend_hip.hessian_operator)
dv.ScalarPack.read = timed("pack.read (blocking)", dv.ScalarPack.read)
dv._read = timed("device._read (blocking)", dv._read)


def solve():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, dc.constraints(ipsolver),
                                        method="tr_interior_point")
    torch.cuda.synchronize()
    return res, time.perf_counter() - t0


for k in range(3):
    res, dt = solve()
acc.clear()
cnt.clear()
R = 5
tot = 0.0
for k in range(R):
    res, dt = solve()
    tot += dt
print("warm solve: %.2f ms  (status %d, %d outer / %d CG)" % (1e3 * tot / R, res.status, res.niter,
                                                              res.cg_niter))
for name in sorted(acc, key=acc.get, reverse=True):
    print("  %-32s %7.3f ms  %4d calls" % (name, 1e3 * acc[name] / R, cnt[name] // R))
print("  (phases nest: projected_cg / modified_dogleg / projections contain blocking reads)")
