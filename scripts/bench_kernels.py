"""Standalone (back-to-back, same kernel) timing of the CG loop's kernels (dev tool)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, torch
from ipsolver import _hip, cg_fused, projector, device as dv
from ipsolver.operators import DeviceHessian
from ipsolver.synthetic import CenteredBandedNLP
n, m = 1000000, 100000
prob = CenteredBandedNLP(n, m)
x = prob.x0
v = 0.1 * np.random.default_rng(7).standard_normal(m)
A = dv.DeviceCSR.from_scipy(prob.constr_jac(x))
H = DeviceHessian(n, dv.DeviceCSR.from_scipy(prob.hess(x)), dv.DVec.from_host(prob.kappa * prob.Wt.dot(v)))
c = dv.DVec.from_host(prob.grad(x)); b = dv.DVec.zeros(m)
Z, LS, Y = projector.projections(A); P = Z.projector
lib = _hip.load(); st = dv.stream_ptr()
x0 = Y.dot(-b); r0 = Z.dot(H.dot(x0) + c); g0 = Z.dot(r0); rt_g = g0.sumsq_amax()[0]
L = cg_fused._Loop(H, P, None, None)
L.x.copy_(x0.t); L.r.copy_(r0.t)
_hip.call("ipx_axpby", n, -1.0, dv._p(g0.t), 0.0, None, dv._p(L.p), st)
init = np.zeros(L.state.numel()); init[0] = rt_g; init[1] = rt_g; init[3] = np.inf; init[9] = 0.0
L.state.copy_(torch.from_numpy(init))
lib.ipx_cg_hp(L.ref(), st)
lib.ipx_cg_iterate(L.ref(), 0, 4, st)
torch.cuda.synchronize()

def timeit(name, fn, N=300):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): fn()
    torch.cuda.synchronize()
    print("%-28s %7.2f us" % (name, (time.perf_counter() - t0) / N * 1e6))

a = L.args
At = A.T
Hc = H.csr
timeit("fused step2+Hp (mode 3)", lambda: lib.ipx_cg_resume(L.ref(), 0, 3, st))
timeit("H.p spmv (ipx_cg_hp)", lambda: lib.ipx_cg_hp(L.ref(), st))
X = dv.DVec(L.x); Pv = dv.DVec(L.p); R = dv.DVec(L.r); W = dv.DVec(L.w); V = dv.DVec(L.v)
timeit("A.r spmv", lambda: A.spmv(R, out=W))
timeit("r - A'v spmv (reduce)", lambda: At.spmv(V, alpha=-1.0, beta=1.0, yin=R, out=R, reduce=False))
timeit("H.p plain spmv", lambda: Hc.spmv(Pv, out=dv.DVec(L.Hp)))
timeit("banded solve", lambda: lib.ipx_banded_solve(ctypes.c_void_p(P.solver.handle), dv._p(L.w), dv._p(L.v), st))
grid = int(lib.ipx_cg_vec_grid(n))
timeit("step1", lambda: lib.ipx_cg_step1(n, dv._p(L.state), 0, dv._p(L.part1), Hc.pattern.ntiles, dv._p(L.x), dv._p(L.p), dv._p(L.r), dv._p(L.Hp), None, None, dv._p(L.part2), grid, st))
