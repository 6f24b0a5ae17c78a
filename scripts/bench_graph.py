"""Eager launches vs hipGraph replay of the fused CG loop (dev tool)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, torch
from ipsolver import _hip, cg_fused, projector, device as dv
from ipsolver.operators import DeviceHessian
from ipsolver.synthetic import CenteredBandedNLP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = n // 10
prob = CenteredBandedNLP(n, m)
x = prob.x0
v = 0.1 * np.random.default_rng(7).standard_normal(m)
A = dv.DeviceCSR.from_scipy(prob.constr_jac(x))
H = DeviceHessian(n, dv.DeviceCSR.from_scipy(prob.hess(x)), dv.DVec.from_host(prob.kappa * prob.Wt.dot(v)))
c = dv.DVec.from_host(prob.grad(x)); b = dv.DVec.zeros(m)
Z, LS, Y = projector.projections(A); P = Z.projector
lib = _hip.load()
side = torch.cuda.Stream()
torch.cuda.set_stream(side)          # graphs cannot be captured on the legacy default stream
st = dv.stream_ptr()
x0 = Y.dot(-b); r0 = Z.dot(H.dot(x0) + c); g0 = Z.dot(r0); rt_g = g0.sumsq_amax()[0]
L = cg_fused._Loop(H, P, None, None)
L.x.copy_(x0.t); L.r.copy_(r0.t)
_hip.call("ipx_axpby", n, -1.0, dv._p(g0.t), 0.0, None, dv._p(L.p), st)
init = np.zeros(L.state.numel()); init[0] = rt_g; init[3] = np.inf; init[9] = P.orth_tol * P.norm_A
L.state.copy_(torch.from_numpy(init))
lib.ipx_cg_hp(L.ref(), st)
lib.ipx_cg_iterate(L.ref(), 0, 20, st)
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter(); lib.ipx_cg_iterate(L.ref(), 20, 20 + K, st); th = time.perf_counter() - t0; torch.cuda.synchronize()
print("n=%d eager  us/iter %.2f  (host enqueue %.2f)" % (n, (time.perf_counter() - t0) / K * 1e6, th / K * 1e6))
g = lib.ipx_cg_graph_create(L.ref(), st)
torch.cuda.synchronize()
assert g, "graph capture failed"
lib.ipx_cg_graph_launch(ctypes.c_void_p(g), 5, st); torch.cuda.synchronize()
t0 = time.perf_counter(); lib.ipx_cg_graph_launch(ctypes.c_void_p(g), K // 2, st); torch.cuda.synchronize()
print("graph  us/iter %.2f" % ((time.perf_counter() - t0) / K * 1e6))
print("state", L.state.tolist()[4:7], L.state.tolist()[13])
