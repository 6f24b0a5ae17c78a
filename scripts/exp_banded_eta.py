import ctypes, os, sys
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, scipy.sparse as sps, scipy.sparse.linalg as spla, torch
from ipsolver import _hip, device as dv
from ipsolver.projector import BandedNormalSolver, BandedNotDecoupled
lib = _hip.load()
m, k = 20000, 6
rng = np.random.default_rng(6)
for eps in (0.3, 0.1, 0.05, 0.03, 0.01):
    cols = (np.arange(m)[:, None] + np.arange(k + 1)[None, :]).ravel()
    rows = np.repeat(np.arange(m), k + 1)
    vals = (1.0 + 0.01 * rng.standard_normal((m, k + 1))).ravel()
    A = sps.csr_matrix((np.concatenate((vals, np.full(m, eps))),
                        (np.concatenate((rows, np.arange(m))), np.concatenate((cols, m + k + np.arange(m))))),
                       shape=(m, 2 * m + k))
    Ad = dv.DeviceCSR.from_scipy(A)
    try:
        s = BandedNormalSolver(Ad)
    except BandedNotDecoupled as e:
        print(eps, "not decoupled:", e); continue
    h = ctypes.c_void_p(s.handle)
    eta = ctypes.c_double(0)
    steps = lib.ipx_banded_refine_steps(h, ctypes.byref(eta))
    w = rng.standard_normal(m)
    v = s.solve(dv.DVec.from_host(w)).to_host()
    ref = spla.splu((A @ A.T).tocsc()).solve(w)
    torch.cuda.synchronize()
    import time
    W = dv.DVec.from_host(w); t0 = time.perf_counter()
    for _ in range(10): s.solve(W)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(eps, "k", s.k, "steps", steps, "eta", eta.value, "err", np.max(np.abs(v - ref)) / np.max(np.abs(ref)), "solve us", dt * 1e6)
