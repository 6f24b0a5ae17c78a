import cProfile, pstats, io, os, sys, time, warnings
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np
import ipsolver
from ipsolver.synthetic import CenteredBandedNLP
warnings.simplefilter("ignore")
prob = CenteredBandedNLP(1000000, 100000, eps=1e-3)
def solve():
    t0 = time.time()
    res = ipsolver.minimize_constrained(prob.fun, prob.x0, prob.grad, prob.hess, prob.constraints(ipsolver), method="tr_interior_point")
    return res, time.time() - t0
res, dt = solve(); print("first", dt, res.status, res.niter, res.cg_niter)
pr = cProfile.Profile(); pr.enable(); res, dt = solve(); pr.disable()
print("second", dt)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30); print(s.getvalue()[:5000])
