"""cProfile of a config-5 solve, second of two (host side; the device loop shows up as the
blocking state reads)."""
import cProfile, io, os, pstats, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, torch
import ipsolver
from ipsolver.synthetic import CenteredBandedNLP, DeviceCallbacks
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
m = n // 10
warnings.simplefilter("ignore")
prob = CenteredBandedNLP(n, m, eps=1.0)
dc = DeviceCallbacks(prob)
cons = (dc.constraints(ipsolver, ("less", 0.0)), ipsolver.BoxConstraint(("interval", -0.8, 0.8)))
ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, cons)      # symbolic set-up
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
res = ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, cons)
torch.cuda.synchronize()
pr.disable()
print("wall", time.time() - t0, "niter", res.niter, "cg", res.cg_niter)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[-6000:])
