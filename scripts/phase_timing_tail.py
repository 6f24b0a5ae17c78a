"""In-kernel phase timing (workgroup 0) of the cyclic-reduction solve WITH its g = r - A'v tail,
as the CG loop launches it.  Needs: make -C ip-nonlinear-solver_amd/csrc phase-timing (dev tool)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, torch
from ipsolver import _hip
_hip.LIB_PATH = os.path.join(ROOT, "ip-nonlinear-solver_amd", "lib_dbg", "libipx.so")
from ipsolver import projector, device as dv, cg_fused
from ipsolver.operators import DeviceHessian
from ipsolver.synthetic import CenteredBandedNLP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = n // 10
prob = CenteredBandedNLP(n, m)
x = prob.x0
v = 0.1 * np.random.default_rng(7).standard_normal(m)
A = dv.DeviceCSR.from_scipy(prob.constr_jac(x))
H = DeviceHessian(n, dv.DeviceCSR.from_scipy(prob.hess(x)), dv.DVec.from_host(prob.kappa * prob.Wt.dot(v)))
c = dv.DVec.from_host(prob.grad(x)); bz = dv.DVec.zeros(m)
Z, LS, Y = projector.projections(A); P = Z.projector
lib = _hip.load()
x0 = Y.dot(-bz); r0 = Z.dot(H.dot(x0) + c); g0 = Z.dot(r0); rt_g = g0.sumsq_amax()[0]
L = cg_fused._Loop(H, P, None, None)
L.args.no_radius = 1
st = dv.stream_ptr()
init = np.zeros(L.state.numel()); init[cg_fused.ST_RTG0] = rt_g; init[cg_fused.ST_RADIUS] = np.inf
init[cg_fused.ST_ORTH_RHS] = P.orth_tol * P.norm_A
L.x.copy_(x0.t); L.r.copy_(r0.t)
_hip.call("ipx_axpby", n, -1.0, dv._p(g0.t), 0.0, None, dv._p(L.p), st)
L.state.copy_(torch.from_numpy(init))
_hip.check(lib.ipx_cg_hp(L.ref(), st), "hp")
acc = np.zeros(16); R = 40
for it in range(R + 5):
    _hip.check(lib.ipx_cg_iterate(L.ref(), it, it + 1, st), "it")
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    lib.ipx_debug_stamps(out)
    t = np.array(list(out), dtype=np.float64)
    if it >= 5:
        acc += t - t[0]
names = {1: "loads issued (+stop test)", 2: "level-0 copies", 3: "reduction levels done",
         6: "x stored, barrier", 7: "tail + residual done"}
ts = acc / R * 0.01
print("stamps (us after kernel start, workgroup 0):")
for k in (1, 2, 3, 6, 7):
    if ts[k] > 0:
        print("  %5.2f  %s" % (ts[k], names[k]))
print("qv", L.args.At_qv, "fused tail", bool(L.args.At_qv))
