"""Banded (A A')^-1 against a sparse LU over a grid of sizes and bandwidths (dev tool):
edges of the planning rules (single chunk / chunks, wide levels, defect correction)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, scipy.sparse as sps, scipy.sparse.linalg as spla, torch
from ipsolver import _hip, device as dv
from ipsolver.projector import BandedNormalSolver, BandedNotDecoupled
lib = _hip.load()
worst = 0.0
for m in (1, 2, 7, 63, 64, 65, 66, 129, 300, 1000, 2047, 2048, 2049, 2100, 3000, 4097, 8193, 20011, 65537):
    for kA in (1, 2, 3, 4, 5, 6, 8, 9):
        rng = np.random.default_rng(m * 31 + kA)
        cols = (4 * np.arange(m)[:, None] + np.arange(4 * kA)[None, :]).ravel()
        A = sps.csr_matrix((rng.standard_normal(len(cols)), (np.repeat(np.arange(m), 4 * kA), cols)),
                           shape=(m, 4 * m + 4 * kA))
        try:
            s = BandedNormalSolver(dv.DeviceCSR.from_scipy(A))
        except BandedNotDecoupled as e:
            print("m=%d kA=%d: not decoupled (fallback)" % (m, kA)); continue
        w = rng.standard_normal(m)
        v = s.solve(dv.DVec.from_host(w)).to_host()
        S = (A @ A.T).tocsc()
        ref = spla.splu(S).solve(w) if m > 1 else w / S[0, 0]
        err = float(np.max(np.abs(v - ref)) / np.max(np.abs(ref)))
        h = ctypes.c_void_p(s.handle)
        tag = "k=%d lev=%d dec=%d pcr=%d ref=%d" % (s.k, lib.ipx_banded_levels(h), lib.ipx_banded_decoupled(h),
                                                    lib.ipx_banded_pcr_level(h), lib.ipx_banded_refine_steps(h, None))
        worst = max(worst, err)
        if err > 1e-10:
            print("BAD m=%d kA=%d %s err %.2e" % (m, kA, tag, err))
print("worst relative error", worst)
