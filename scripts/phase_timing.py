"""In-kernel phase timing (workgroup 0) of the banded solve, the H.p SpMV and step2.
Needs the diagnostic build:  make -C ip-nonlinear-solver_amd/csrc phase-timing   (dev tool)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, torch
from ipsolver import _hip
_hip.LIB_PATH = os.path.join(ROOT, "ip-nonlinear-solver_amd", os.environ.get("IPX_DBG_LIB", "lib_dbg"), "libipx.so")
from ipsolver import projector, device as dv
from ipsolver.synthetic import CenteredBandedNLP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = n // 10
prob = CenteredBandedNLP(n, m)
A = dv.DeviceCSR.from_scipy(prob.constr_jac(prob.x0))
Z, LS, Y = projector.projections(A)
S = Z.projector.solver
lib = _hip.load()
w = torch.randn(m, dtype=torch.float64, device="cuda"); v = torch.empty_like(w)
part = torch.zeros(m // 256 + 2, dtype=torch.float64, device="cuda")
npart = ctypes.c_int32(0)
names = ["loads issued", "loads landed", "lds stored+sync", "chunk_solve(lane0)", "sync", "xs+sync", "corrected+sync", "residual+reduce"]
acc = np.zeros(7)
R = 50
for rep in range(R + 5):
    lib.ipx_banded_solve_resid(ctypes.c_void_p(S.handle), dv._p(w), dv._p(v), dv._p(part), ctypes.byref(npart), None, dv.stream_ptr())
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    lib.ipx_debug_stamps(out)
    t = np.array(list(out)[:8], dtype=np.float64)
    if rep >= 5:
        acc += np.diff(t)
print("n=%d m=%d decoupled=%d  (wall_clock64 ticks of 10 ns)" % (n, m, lib.ipx_banded_decoupled(ctypes.c_void_p(S.handle))))
for k in range(7):
    print("  %-22s %6.2f us" % (names[k + 1], acc[k] / R * 0.01))
print("  total in-kernel        %6.2f us" % (acc.sum() / R * 0.01))

# ---- the loop's last H.p SpMV and last step2 (workgroup 0) -------------------
from ipsolver import cg_fused
from ipsolver.operators import DeviceHessian
x = prob.x0
vv = 0.1 * np.random.default_rng(7).standard_normal(m)
H = DeviceHessian(n, dv.DeviceCSR.from_scipy(prob.hess(x)), dv.DVec.from_host(prob.kappa * prob.Wt.dot(vv)))
c = dv.DVec.from_host(prob.grad(x)); b = dv.DVec.zeros(m)
P = Z.projector
st = dv.stream_ptr()
x0 = Y.dot(-b); r0 = Z.dot(H.dot(x0) + c); g0 = Z.dot(r0); rt_g = g0.sumsq_amax()[0]
L = cg_fused._Loop(H, P, None, None)
L.x.copy_(x0.t); L.r.copy_(r0.t)
_hip.call("ipx_axpby", n, -1.0, dv._p(g0.t), 0.0, None, dv._p(L.p), st)
init = np.zeros(L.state.numel()); init[0] = rt_g; init[3] = np.inf; init[9] = P.orth_tol * P.norm_A
L.state.copy_(torch.from_numpy(init))
lib.ipx_cg_hp(L.ref(), st)
acc_s, acc_c, acc_b = np.zeros(7), np.zeros(3), np.zeros(7)
for rep in range(40):
    lib.ipx_cg_iterate(L.ref(), 2 * rep, 2 * rep + 2, st)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)(); lib.ipx_debug_stamps(out)
    acc_b += np.diff(np.array(list(out)[:8], dtype=np.float64))
    out = (ctypes.c_ulonglong * 16)(); lib.ipx_debug_stamps_spmv(out)
    acc_s += np.diff(np.array(list(out)[:8], dtype=np.float64))
    out = (ctypes.c_ulonglong * 16)(); lib.ipx_debug_stamps_cg(out)
    tt = np.array(list(out), dtype=np.float64)
    acc_c2 = globals().setdefault("acc_c2", np.zeros(7)); acc_c2 += np.diff(tt[:8])
    acc_c1 = globals().setdefault("acc_c1", np.zeros(6)); acc_c1 += np.diff(tt[8:15])
print("H.p SpMV, workgroup 0:")
for k, nm in enumerate(["tile info + guard", "batch-1 loads landed (colidx,val,...)", "gathers issued", "gathers landed + LDS stores", "barrier", "row sums + y stores", "reduce + partial"]):
    print("  %-40s %6.2f us" % (nm, acc_s[k] / 40 * 0.01))
print("  total %.2f us" % (acc_s.sum() / 40 * 0.01))
print("step2, workgroup 0:")
for k, nm in enumerate(["operand + partial + state loads", "fold", "update + stores"]):
    print("  %-40s %6.2f us" % (nm, acc_c[k] / 40 * 0.01))
print("  total %.2f us" % (acc_c.sum() / 40 * 0.01))
print("banded solve in the loop (with the g = r - A'v tail when fused), workgroup 0:")
for k in range(7):
    print("  %-22s %6.2f us" % (names[k + 1], acc_b[k] / 40 * 0.01))
print("  total %.2f us" % (acc_b.sum() / 40 * 0.01))
print("fused step2 + H.p (k_cg_step2_hp), workgroup 0:")
for k, nm in enumerate(["loads issued, state landed", "fold", "p_next span + x/p stores issued", "barrier", "products", "rows", "reduce + stores"]):
    print("  %-40s %6.2f us" % (nm, acc_c2[k] / 40 * 0.01))
print("  total %.2f us" % (acc_c2.sum() / 40 * 0.01))
print("fused step1 + A.r (k_cg_step1_ar), workgroup 0:")
for k, nm in enumerate(["loads issued, state landed", "fold", "r_next span + stores issued", "barrier", "products + rows", "reduce + stores"]):
    print("  %-40s %6.2f us" % (nm, acc_c1[k] / 40 * 0.01))
print("  total %.2f us" % (acc_c1.sum() / 40 * 0.01))
