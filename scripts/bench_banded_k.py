"""Banded (A A')^-1 solve and factorization time by half bandwidth k at m = 1e5 (dev tool)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, scipy.sparse as sps, torch
from ipsolver import _hip, device as dv
from ipsolver.projector import BandedNormalSolver
lib = _hip.load()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for k in (1, 2, 3, 4, 5, 6, 7, 8, 9):
    rng = np.random.default_rng(k)
    starts = 4 * np.arange(m)
    cols = (starts[:, None] + np.arange(4 * k)[None, :]).ravel()
    A = sps.csr_matrix((rng.standard_normal(len(cols)), (np.repeat(np.arange(m), 4 * k), cols)),
                       shape=(m, 4 * m + 4 * k))
    Ad = dv.DeviceCSR.from_scipy(A)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    try:
        s = BandedNormalSolver(Ad)
    except NotImplementedError as e:
        print("k=%d: %s" % (k, e)); continue
    torch.cuda.synchronize(); t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(5):
        s2 = BandedNormalSolver(Ad)
    torch.cuda.synchronize(); t_fac = (time.perf_counter() - t0) / 5
    w = dv.DVec.from_host(rng.standard_normal(m))
    out = torch.empty(m, dtype=torch.float64, device="cuda")
    h = ctypes.c_void_p(s.handle)
    for _ in range(3):
        lib.ipx_banded_solve(h, dv._p(w.t), dv._p(out), dv.stream_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        lib.ipx_banded_solve(h, dv._p(w.t), dv._p(out), dv.stream_ptr())
    torch.cuda.synchronize(); t_solve = (time.perf_counter() - t0) / 20
    from ipsolver.projector import IterativeNormalSolver, normal_solver_for
    it = IterativeNormalSolver(Ad)
    v = it.solve(w)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        v = it.solve(w)
    torch.cuda.synchronize(); t_pcg = (time.perf_counter() - t0) / 3
    err = float(torch.max(torch.abs(v.t - out)) / torch.max(torch.abs(out)))
    print("      device PCG: %.2f ms per solve (%d iterations), vs banded %.1e; projections() picks %s"
          % (1e3 * t_pcg, it.stats["iterations"] // it.stats["solves"], err,
             type(normal_solver_for(Ad)).__name__))
    eta = ctypes.c_double(0.0)
    steps = lib.ipx_banded_refine_steps(h, ctypes.byref(eta))
    import scipy.sparse.linalg as spla
    S = (A @ A.T).tocsc()
    ref = spla.splu(S).solve(w.to_host())
    err_d = float(np.max(np.abs(out.cpu().numpy() - ref)) / np.max(np.abs(ref)))
    print("k=%d (solver k=%d) levels %d decoupled %d pcr %d refine steps %d (eta %.2e): factor %.2f ms, "
          "solve %.1f us, vs sparse LU %.1e"
          % (k, s.k, lib.ipx_banded_levels(h), lib.ipx_banded_decoupled(h), lib.ipx_banded_pcr_level(h),
             steps, eta.value, 1e3 * t_fac, 1e6 * t_solve, err_d))
