import cProfile, os, pstats, sys, time, warnings, io
sys.path.insert(0, "/root/repo/ip-nonlinear-solver_amd")
import torch, ipsolver
from ipsolver.synthetic import DenseDeviceCallbacks
warnings.simplefilter("ignore")
cbn = DenseDeviceCallbacks.on_device(10000, 2000)
def solve():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ipsolver.minimize_constrained(cbn.fun, cbn.x0, cbn.grad, cbn.hess, cbn.constraints(ipsolver), method="equality_constrained_sqp")
    torch.cuda.synchronize(); return r, time.perf_counter() - t0
for _ in range(4):
    r, dt = solve(); print("%.1f ms  %d/%d status %d" % (1e3*dt, r.niter, r.cg_niter, r.status))
pr = cProfile.Profile(); pr.enable(); solve(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25); print(s.getvalue()[:5000])
