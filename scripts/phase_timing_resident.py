"""In-kernel phase timing of the resident loop kernel (csrc/resident.hip), workgroup 0, last
iteration of a batch.  Needs the diagnostic build: make -C ip-nonlinear-solver_amd/csrc phase-timing"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, torch
from ipsolver import _hip
_hip.LIB_PATH = os.path.join(ROOT, "ip-nonlinear-solver_amd", os.environ.get("IPX_DBG_LIB", "lib_dbg"), "libipx.so")
from ipsolver import projector, cg_fused, device as dv
from ipsolver.operators import DeviceHessian
from ipsolver.synthetic import CenteredBandedNLP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
m = n // 10
prob = CenteredBandedNLP(n, m)
x = prob.x0
A = dv.DeviceCSR.from_scipy(prob.constr_jac(x))
vv = 0.1 * np.random.default_rng(7).standard_normal(m)
H = DeviceHessian(n, dv.DeviceCSR.from_scipy(prob.hess(x)), dv.DVec.from_host(prob.kappa * prob.Wt.dot(vv)))
Z, LS, Y = projector.projections(A)
lib = _hip.load()
c = dv.DVec.from_host(prob.grad(x)); b = dv.DVec.zeros(m)
P = Z.projector
st = dv.stream_ptr()
x0 = Y.dot(-b); r0 = Z.dot(H.dot(x0) + c); g0 = Z.dot(r0); rt_g = g0.sumsq_amax()[0]
L = cg_fused._Loop(H, P, None, None)
assert L.args.resident
L.x.copy_(x0.t); L.r.copy_(r0.t)
_hip.call("ipx_axpby", n, -1.0, dv._p(g0.t), 0.0, None, dv._p(L.p), st)
init = np.zeros(L.state.numel()); init[0] = rt_g; init[3] = 1e300; init[9] = P.orth_tol * P.norm_A
L.state.copy_(torch.from_numpy(init))
lib.ipx_cg_hp(L.ref(), st)
names = ["r_next on span", "w = A r (row sums)", "cyclic reduction", "t = A'v, g", "residual + block sums",
         "hop 2 (publish + wait)", "fold + branches", "x, p update", "Hp rows + block sum", "hop 1 (publish + wait)", "fold"]
acc = np.zeros(11)
skew = np.zeros(2)
R = 40
for rep in range(R):
    lib.ipx_cg_iterate(L.ref(), 4 * rep, 4 * rep + 4, st)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)(); lib.ipx_debug_stamps_res(out)
    t = np.array(list(out), dtype=np.float64)
    acc += np.diff(t[:12])
    skew += [t[12] - t[6], t[13] - t[10]]
print("n=%d  resident kernel, an interior workgroup, last iteration of 4 (wall_clock64, 10 ns ticks)" % n)
for k in range(11):
    print("  %-28s %6.2f us" % (names[k], acc[k] / R * 0.01))
print("  total %.2f us" % (acc.sum() / R * 0.01))
print("  of the two folds, waiting for the workgroup's other waves to leave the hop: %.2f / %.2f us"
      % (skew[0] / R * 0.01, skew[1] / R * 0.01))
