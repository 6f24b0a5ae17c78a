"""In-kernel phase timing of k_solve_decoupled (diagnostic build: compile csrc with
-DIPX_PHASE_TIMING into ip-nonlinear-solver_amd/lib_dbg/libipx.so; dev tool)."""
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
