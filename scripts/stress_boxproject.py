"""Random barrier-shaped Jacobians: ipx_boxschur_project (box rows per group) against the SpMV
form with the same solver and against a direct sparse solve (dev tool)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.sparse as sps, scipy.sparse.linalg as spla, torch
from ipsolver import _hip, device as dv, projector
from ipsolver.boxschur import BoxSchurNormalSolver
from banded_setup import BandedInstance
lib = _hip.load()
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(600, 30000)); m = max(20, n // int(rng.integers(6, 14)))
    J = BandedInstance(n, m).A
    kind = rng.integers(0, 4, n) if seed % 3 else np.full(n, 3)
    L, U = np.flatnonzero(kind & 1), np.flatnonzero(kind & 2)
    I = sps.eye(n, format="csr")
    # slacks as in a barrier run: active constraints ~1e-8, but never both bounds of a variable
    s_nl = 10.0 ** rng.uniform(-6, 0.3, m)
    side = rng.integers(0, 2, n)                    # which bound of a variable may be active
    small = 10.0 ** rng.uniform(-8, 0.3, n)
    s_lb = np.where(side == 0, small, rng.uniform(0.5, 1.6, n))[L]
    s_ub = np.where(side == 1, small, rng.uniform(0.5, 1.6, n))[U]
    s = np.concatenate((s_nl, s_lb, s_ub))
    blocks = [[J, sps.diags(s[:m]), None, None]]
    if len(L): blocks.append([-I[L], None, sps.diags(s[m:m + len(L)]), None])
    if len(U): blocks.append([I[U], None, None, sps.diags(s[m + len(L):])])
    A = sps.bmat(blocks, format="csr"); A.sort_indices()
    N, M = A.shape[1], A.shape[0]
    Ad = dv.DeviceCSR.from_scipy(A)
    Z, LS, Y = projector.projections(Ad)
    S = Z.projector.solver
    if not isinstance(S, BoxSchurNormalSolver) or S.c_args() is None:
        print(seed, "solver", type(S).__name__, "(skipped)"); continue
    args = S.c_args()
    r = rng.standard_normal(N); rd = dv.DVec.from_host(r)
    nblk = lib.ipx_boxschur_project_count(ctypes.byref(args))
    g = torch.empty(N, dtype=torch.float64, device="cuda")
    pg = torch.zeros(2 * nblk + 2, dtype=torch.float64, device="cuda")
    pres = torch.zeros(M // 64 + 4, dtype=torch.float64, device="cuda")
    n3, n4 = ctypes.c_int32(0), ctypes.c_int32(0)
    _hip.call("ipx_boxschur_project", ctypes.byref(args), dv._p(rd.t), dv._p(g), dv._p(pg), ctypes.byref(n3),
              dv._p(pres), ctypes.byref(n4), None, dv.stream_ptr())
    got = g.cpu().numpy()
    ref_dev = Ad.rmatvec_sub(S.solve(Ad.dot(rd)), rd).to_host()
    # (A A' has condition ~1e16 with active slacks: the direct sparse solve of the normal
    # equations is itself only good to ~1e-3 there; the null-space property is the check)
    e1 = np.max(np.abs(got - ref_dev)) / np.max(np.abs(r))
    e2 = np.max(np.abs(A @ got)) / (np.max(np.abs(r)) * np.sqrt(A.shape[1]))
    flag = "" if e1 < 1e-10 and e2 < 1e-8 else "  <-- BAD"
    bad += bool(flag)
    print("%2d n=%d m=%d L=%d U=%d  vs SpMV form %.1e  |A g| %.1e%s" % (seed, n, m, len(L), len(U), e1, e2, flag))
print("bad:", bad)
