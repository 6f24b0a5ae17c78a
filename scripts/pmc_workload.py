"""Workload for the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE):
calibration kernels with known byte counts, then 40 iterations of the fused
CG loop at n (default 1e6), m = n/10."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, torch
from ipsolver import _hip, cg_fused, projector, device as dv
from ipsolver.operators import DeviceHessian
from ipsolver.synthetic import CenteredBandedNLP
lib = _hip.load(); st = dv.stream_ptr()
# ---- calibration: 8-byte-per-lane streaming (k_map<OpMul>: reads 2n, writes n doubles)
#      and 16-byte-per-lane streaming (k_axpby2: reads 2n, writes n doubles)
nc = 8000000
a = torch.randn(nc, dtype=torch.float64, device="cuda"); b = torch.randn_like(a); o = torch.empty_like(a)
for _ in range(10):
    lib.ipx_mul(nc, dv._p(a), dv._p(b), dv._p(o), st)
for _ in range(10):
    lib.ipx_axpby(nc, 1.5, dv._p(a), 0.5, dv._p(b), dv._p(o), st)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = n // 10
prob = CenteredBandedNLP(n, m)
x = prob.x0
v = 0.1 * np.random.default_rng(7).standard_normal(m)
A = dv.DeviceCSR.from_scipy(prob.constr_jac(x))
H = DeviceHessian(n, dv.DeviceCSR.from_scipy(prob.hess(x)), dv.DVec.from_host(prob.kappa * prob.Wt.dot(v)))
c = dv.DVec.from_host(prob.grad(x)); bz = dv.DVec.zeros(m)
Z, LS, Y = projector.projections(A); P = Z.projector
x0 = Y.dot(-bz); r0 = Z.dot(H.dot(x0) + c); g0 = Z.dot(r0); rt_g = g0.sumsq_amax()[0]
L = cg_fused._Loop(H, P, None, None)
L.args.no_radius = 0          # as bench.py's headline: a finite trust radius that is never reached
L.x.copy_(x0.t); L.r.copy_(r0.t)
_hip.call("ipx_axpby", n, -1.0, dv._p(g0.t), 0.0, None, dv._p(L.p), st)
init = np.zeros(L.state.numel()); init[0] = rt_g; init[3] = 1e300; init[9] = P.orth_tol * P.norm_A
L.state.copy_(torch.from_numpy(init))
lib.ipx_cg_hp(L.ref(), st)
lib.ipx_cg_iterate(L.ref(), 0, 40, st)
torch.cuda.synchronize()
print("done", L.state.tolist()[13])
