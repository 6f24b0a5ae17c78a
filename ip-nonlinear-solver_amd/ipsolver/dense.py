"""Dense Jacobian path (BASELINE config 2): device matrix + Gram/Cholesky
normal-equation solver.  Kernels: csrc/dense.hip.

The reference factors dense Jacobians with LAPACK pivoted QR
(projections.py:175-233).  Here ``G = A A'`` is formed with fp64 MFMA,
factored by a blocked Cholesky, and ``G^-1`` is kept explicitly so that each
``(AA')^-1`` application inside the CG loop is one dense matvec (6 us instead of two
latency-bound triangular sweeps of 63 dependent tile steps each).  Applying the explicit
inverse is as accurate as two triangular solves with the factor -- both are limited by
the normal equations, cond(A)^2 eps: measured 3.0e-5 vs 1.1e-5 at cond(A) = 1e6,
scripts/exp_ill_conditioned_dense.py and tests/test_gpu_qp.py::test_dense_ill_conditioned --
and what the normal equations lose against the reference's QR is recovered by refinement
steps (``refine_steps``, enabled from the measured pivot loss).  The
operators are the same (QR vs Gram-Cholesky iterates
agree to 8e-16 on well-conditioned problems); a rank-deficient Jacobian is
reported through the pivot flag.
"""
import ctypes
import warnings

import numpy as np
import torch

from . import _hip
from . import device as dv
from .device import DVec, _p, stream_ptr, ctx

_F64 = torch.float64


class DeviceDense:
    """Row-major dense fp64 matrix in HBM with ``dot`` / ``T.dot`` (the
    transpose is materialised once: both products are row-streaming matvecs)."""

    def __init__(self, t):
        assert t.dtype == _F64 and t.dim() == 2 and t.is_cuda
        self.t = t if t.is_contiguous() else t.contiguous()
        self.shape = (int(t.shape[0]), int(t.shape[1]))
        self._T = None

    @staticmethod
    def from_host(a):
        dv._require_gpu()
        a = np.ascontiguousarray(np.atleast_2d(np.asarray(a, dtype=np.float64)))
        with warnings.catch_warnings():        # (a read-only host array is only read here)
            warnings.simplefilter("ignore", UserWarning)
            return DeviceDense(torch.from_numpy(a).to(ctx().device))

    def to_host(self):
        return self.t.cpu().numpy()

    @property
    def T(self):
        if self._T is None:
            self._T = DeviceDense(self.t.t().contiguous())     # layout copy, no arithmetic
            self._T._T = self
        return self._T

    def gemv(self, x, alpha=1.0, diag=None, beta=0.0, yin=None, reduce=False, slot=0):
        m, n = self.shape
        assert len(x) == n, (len(x), n)
        out = DVec(dv._empty(m))
        c = ctx()
        _hip.call("ipx_dense_gemv", m, n, _p(self.t), n, _p(x.t), float(alpha),
                  _p(diag.t) if diag is not None else None, float(beta),
                  _p(yin.t) if yin is not None else None, _p(out.t),
                  ctypes.c_void_p(c.out.data_ptr() + 16 * slot) if reduce else None,
                  _p(c.ws), stream_ptr())
        return out

    spmv = gemv

    def dot(self, x):
        return self.gemv(x if isinstance(x, DVec) else DVec.from_host(x))

    matvec = dot

    def matvec_sumsq(self, x, slot=0):
        out = self.gemv(x, reduce=True, slot=slot)
        return out, dv.read_slots(2 * slot + 1)[2 * slot]

    def rmatvec_sub(self, v, x, reduce=False, slot=0):
        return self.T.gemv(v, alpha=-1.0, beta=1.0, yin=x, reduce=reduce, slot=slot)

    def frobenius_norm(self):
        return dv.norm(DVec(self.t.reshape(-1)))


class DenseNormalSolver:
    """(A A')^-1 through a dense Cholesky: MFMA Gram for a DeviceDense A, sparse
    row products for a DeviceCSR A whose A A' is too wide for the banded
    solver; blocked Cholesky; explicit inverse applied as one matvec."""

    MAX_ROWS_FROM_SPARSE = 16384     # 2 GiB of G^-1

    def __init__(self, A):
        lib = _hip.load()
        m, n = A.shape
        self.m = m
        M = int(lib.ipx_dense_padded(m))
        dev = ctx().device
        G = torch.empty((M, M), dtype=_F64, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        st = stream_ptr()
        if isinstance(A, DeviceDense):
            splits = int(lib.ipx_gram_splits(m, n))
            ws = torch.empty(int(lib.ipx_gram_ws_doubles(m, splits)), dtype=_F64, device=dev) \
                if splits > 1 else None
            _hip.call("ipx_gram_f64_mfma_split", m, n, _p(A.t), n, _p(G), _p(ws), splits, st)
        else:
            p = A.pattern
            _hip.call("ipx_aat_dense", m, _p(p.indptr), _p(p.indices), _p(A.val), _p(G), st)
        work = torch.empty(M + 1, dtype=_F64, device=dev)
        _hip.call("ipx_chol_factor", M, _p(G), _p(flag), _p(work), st)
        fl = int(flag.item())
        if fl & 4:
            raise np.linalg.LinAlgError("Singular Jacobian matrix: A A' is not positive definite")
        # bit 1 alone: every pivot positive, one lost 43 bits (numerically rank deficient):
        # ``projections`` takes the SVD exit when the matrix is small enough, else keeps this
        # factorization with its refinement steps (pivot_ratio below)
        self.ill_conditioned = bool(fl & 1)
        # Digits lost by the factorization (~1/cond(A)^2).  The reference's pivoted QR
        # (projections.py:175-233) loses cond(A), not cond(A)^2: for an ill-conditioned
        # Jacobian the least-squares and row-space operators get that back by refinement
        # steps against their own residuals (the null-space operator has the reference's
        # orthogonality-driven loop already).  None at the conditioning of the benchmarks.
        self.pivot_ratio = float(work[M].item())
        r = self.pivot_ratio
        self.refine_steps = 0 if r > 1e-3 else (1 if r > 1e-8 else (2 if r > 1e-11 else 3))
        X = torch.empty((M, M), dtype=_F64, device=dev)
        _hip.call("ipx_chol_inverse", M, _p(G), _p(X), st)
        self.M = M
        self.Ginv = DeviceDense(X)
        self._pad = torch.zeros(M, dtype=_F64, device=dev) if M != m else None

    def solve(self, w):
        if self._pad is None:
            return self.Ginv.gemv(w)
        self._pad[:self.m].copy_(w.t)
        return self.Ginv.gemv(DVec(self._pad))[:self.m].copy()
