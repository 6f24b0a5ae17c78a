"""Where the fixed cost of one public ``ipsolver.qp.projected_cg`` call goes (n = 1e6 problem of
bench.py, tol = 0, max_iter = 20): cProfile of 50 calls + wall clock per call."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from ipsolver import device as dv, projector, qp, cg_fused
from ipsolver.operators import DeviceHessian
from ipsolver.synthetic import CenteredBandedNLP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = n // 10
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
prob = CenteredBandedNLP(n, m, seed=0)
x = prob.x0
v = 0.1 * np.random.default_rng(7).standard_normal(m)
A = dv.DeviceCSR.from_scipy(prob.constr_jac(x))
H = DeviceHessian(n, csr=dv.DeviceCSR.from_scipy(prob.hess(x)), diag=dv.DVec.from_host(prob.kappa * prob.Wt.dot(v)))
c = dv.DVec.from_host(prob.grad(x)); b = dv.DVec.zeros(m)
Z, LS, Y = projector.projections(A)
for _ in range(3):
    qp.projected_cg(H, c, Z, Y, b, trust_radius=1e300, tol=0, max_iter=K)
torch.cuda.synchronize()
for label, bb in (("b given", b), ("b=None", None)):
    t0 = time.perf_counter()
    R = 50
    for _ in range(R):
        xo, info = qp.projected_cg(H, c, Z, Y, bb, trust_radius=1e300, tol=0, max_iter=K)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    print("%s: %.1f us per call of %d iterations (%.1f us per iteration)" % (label, 1e6 * dt, K, 1e6 * dt / K))
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    qp.projected_cg(H, c, Z, Y, b, trust_radius=1e300, tol=0, max_iter=K)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
