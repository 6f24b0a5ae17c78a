"""Several seeds of the banded NLP, device-callback mode against host-callback mode (dev tool)."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, torch
import ipsolver
from ipsolver.synthetic import CenteredBandedNLP, DeviceCallbacks
warnings.simplefilter("ignore")
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(800, 30000)); m = n // int(rng.integers(5, 12))
    for kind, extra in ((("equals", 0), ()), (("less", 0.0), ("box",))):
        prob = CenteredBandedNLP(n, m, seed=seed, eps=1e-3 if kind[0] == "equals" else 1.0)
        dc = DeviceCallbacks(prob)
        cons_d = [dc.constraints(ipsolver, kind)]
        cons_h = [prob.constraints(ipsolver, kind)]
        if extra:
            cons_d.append(ipsolver.BoxConstraint(("interval", -0.8, 0.8)))
            cons_h.append(ipsolver.BoxConstraint(("interval", -0.8, 0.8)))
        rd = ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, cons_d, max_iter=400)
        rh = ipsolver.minimize_constrained(prob.fun, prob.x0, prob.grad, prob.hess, cons_h, max_iter=400)
        xd = rd.x.cpu().numpy()
        dx = np.max(np.abs(xd - rh.x)) / np.max(np.abs(rh.x))
        ok = rd.status == rh.status == 1 and dx < 1e-5 and abs(rd.fun - rh.fun) <= 1e-8 * max(1, abs(rh.fun))
        bad += not ok
        print("seed %d n=%d m=%d %-7s dev: st %d it %d cg %d | host: st %d it %d cg %d | dx %.1e %s"
              % (seed, n, m, kind[0], rd.status, rd.niter, rd.cg_niter, rh.status, rh.niter, rh.cg_niter, dx, "" if ok else "<-- CHECK"))
print("flagged:", bad)
