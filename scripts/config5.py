"""BASELINE config-5 style run: banded NLP with nonlinear inequalities + box on
every variable, tr_interior_point, device-callback mode.
    python scripts/config5.py [n] [m] [max_iter]"""
import json, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, torch
import ipsolver
from ipsolver.synthetic import CenteredBandedNLP, DeviceCallbacks
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
m = int(sys.argv[2]) if len(sys.argv) > 2 else n // 10
max_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
warnings.simplefilter("ignore")
prob = CenteredBandedNLP(n, m, eps=1.0)
dc = DeviceCallbacks(prob)
cons = (dc.constraints(ipsolver, ("less", 0.0)), ipsolver.BoxConstraint(("interval", -0.8, 0.8)))
t0 = time.time()
res = ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, cons, max_iter=max_iter)
torch.cuda.synchronize()
wall = time.time() - t0
x = res.x.cpu().numpy()
print(json.dumps({"n": n, "m": m, "status": int(res.status), "niter": int(res.niter),
                  "cg_niter": int(res.cg_niter), "nfev": int(res.nfev),
                  "optimality": float(res.optimality),
                  "constr_violation": float(res.constr_violation),
                  "barrier_parameter": float(res.barrier_parameter), "wall_s": wall,
                  "cg_it_per_s": res.cg_niter / wall,
                  "active_bounds": int(np.sum(np.abs(np.abs(x) - 0.8) < 1e-6)),
                  "fun": float(res.fun),
                  "fused_cg_stats": __import__("ipsolver.cg_fused", fromlist=["STATS"]).STATS}))
