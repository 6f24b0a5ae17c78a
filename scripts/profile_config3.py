"""cProfile of the whole config-3 solve (n = 1e6, device callbacks): where the host spends the
~1.2 ms per outer iteration around ~0.15 ms of GPU work."""
import cProfile, os, pstats, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd")); sys.path.insert(0, ROOT)
import torch
import ipsolver
from ipsolver.synthetic import CenteredBandedNLP, DeviceCallbacks, LeanDeviceCallbacks
if os.environ.get("LEAN", "1") == "1":
    DeviceCallbacks = LeanDeviceCallbacks
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
full = CenteredBandedNLP(n, n // 10, eps=1e-3)
dc = DeviceCallbacks(full)
def solve():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, dc.constraints(ipsolver),
                                             method="tr_interior_point")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.time(); res = solve(); torch.cuda.synchronize()
    print("wall %.4f s, status %d, %d outer / %d CG" % (time.time() - t0, res.status, res.niter, res.cg_niter))
pr = cProfile.Profile(); pr.enable(); solve(); torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
st.sort_stats("cumulative").print_stats(45)
