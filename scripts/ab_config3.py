"""A/B of one switch on warm config-3 solves (dev tool): alternating groups of solves with the
switch on / off, medians.   python scripts/ab_config3.py [module.attr] [n]"""
import os, statistics, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import importlib
import torch
import ipsolver
from ipsolver.synthetic import CenteredBandedNLP, LeanDeviceCallbacks
warnings.simplefilter("ignore")
what = sys.argv[1] if len(sys.argv) > 1 else "ipsolver.sqp.EVALUATE_BEHIND_THE_CHAIN"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
mod, attr = what.rsplit(".", 1)
mod = importlib.import_module(mod)
dc = LeanDeviceCallbacks(CenteredBandedNLP(n, n // 10, eps=1e-3))


def solve():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, dc.constraints(ipsolver),
                                        method="tr_interior_point")
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0), res


for _ in range(3):
    solve()
times = {True: [], False: []}
for rep in range(6):
    for on in (True, False):
        setattr(mod, attr, on)
        for _ in range(5):
            dt, res = solve()
            times[on].append(dt)
for on in (True, False):
    t = sorted(times[on])
    print("%s = %-5s  median %.2f ms  min %.2f  max %.2f  (%d solves; %d outer / %d CG)" % (
        what, on, statistics.median(t), t[0], t[-1], len(t), res.niter, res.cg_niter))
