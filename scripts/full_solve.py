"""Wall-clock of minimize_constrained to gtol=1e-8 on the BASELINE configs
(dev/measurement tool; bench.py reports the headline it/s).

    python scripts/full_solve.py config3 [n m]     sparse banded NLP, tr_interior_point
    python scripts/full_solve.py config2 [n m]     dense equality QP, equality_constrained_sqp
"""
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np
import ipsolver
from ipsolver.synthetic import CenteredBandedNLP

cfg = sys.argv[1] if len(sys.argv) > 1 else "config3"
warnings.simplefilter("ignore")
trace = []


def cb(st):
    trace.append((int(st.niter), int(st.cg_niter), float(st.trust_radius), float(st.optimality)))
    return False


if cfg == "config3":
    nums = [a for a in sys.argv[2:] if not a.startswith("--")]
    n = int(nums[0]) if len(nums) > 0 else 1000000
    m = int(nums[1]) if len(nums) > 1 else 100000
    prob = CenteredBandedNLP(n, m, eps=1e-3)
    if "--device" in sys.argv:          # device-callback mode: x0 and callbacks on the GPU
        import torch
        from ipsolver.synthetic import DeviceCallbacks
        dc = DeviceCallbacks(prob)
        torch.cuda.synchronize()
        t0 = time.time()
        res = ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess,
                                            dc.constraints(ipsolver), method="tr_interior_point",
                                            callback=cb)
        cfg = "config3-device-callbacks"
    else:
        t0 = time.time()
        res = ipsolver.minimize_constrained(prob.fun, prob.x0, prob.grad, prob.hess,
                                            prob.constraints(ipsolver),
                                            method="tr_interior_point", callback=cb)
else:
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    rng = np.random.default_rng(0)
    A = rng.standard_normal((m, n))
    G = rng.standard_normal((n, n)) / np.sqrt(n)
    H = G.dot(G.T) + np.eye(n)
    c = rng.standard_normal(n)
    b = A.dot(rng.standard_normal(n))
    t0 = time.time()
    res = ipsolver.minimize_constrained(lambda x: 0.5 * x.dot(H.dot(x)) + c.dot(x), np.zeros(n),
                                        lambda x: H.dot(x) + c, lambda x: H,
                                        ipsolver.LinearConstraint(A, ("equals", b)),
                                        method="equality_constrained_sqp", callback=cb)
wall = time.time() - t0
print(json.dumps({"config": cfg, "n": n, "m": m, "status": int(res.status), "niter": int(res.niter),
                  "cg_niter": int(res.cg_niter), "nfev": int(res.nfev),
                  "optimality": float(res.optimality),
                  "constr_violation": float(res.constr_violation),
                  "wall_s": wall, "solver_s": float(res.execution_time),
                  "trace_every_2": trace[::2]}))
