"""cProfile of the first outer iterations of config 5 (box + nonlinear inequalities, n = 5e5,
device callbacks), second run of two: where the host spends the ~5 ms per outer iteration
that are not the CG loop (dev tool).   python scripts/profile_config5.py [n m max_iter]"""
import cProfile, os, pstats, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import torch
import ipsolver
from ipsolver import cg_fused
from ipsolver.synthetic import CenteredBandedNLP, DeviceCallbacks
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
m = int(sys.argv[2]) if len(sys.argv) > 2 else n // 10
max_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 30
warnings.simplefilter("ignore")
prob = CenteredBandedNLP(n, m, eps=1.0)
dc = DeviceCallbacks(prob)
cons = (dc.constraints(ipsolver, ("less", 0.0)), ipsolver.BoxConstraint(("interval", -0.8, 0.8)))
def solve():
    torch.cuda.synchronize(); t0 = time.time()
    res = ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, cons, max_iter=max_iter)
    torch.cuda.synchronize()
    return res, time.time() - t0
res, dt = solve()
print("first run %.3f s" % dt)
pr = cProfile.Profile(); pr.enable(); res, dt = solve(); pr.disable()
print("second run %.3f s: %d outer / %d CG iterations (%.1f us of wall per CG iteration)" % (dt, res.niter, res.cg_niter, 1e6 * dt / max(res.cg_niter, 1)))
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumulative").print_stats(40)
