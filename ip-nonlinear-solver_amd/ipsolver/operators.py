"""Device operator types of the seam besides plain matrices."""

import os

from . import _hip
from .device import DVec, DeviceCSR, _p, stream_ptr


class DiagonalOperator:
    """x -> d * x (scaling S = diag(1, ..., s) of tr_interior_point.py:102-115,
    diagonal constraint Hessians)."""

    def __init__(self, d):
        self.d = d
        self.shape = (len(d), len(d))

    def dot(self, x):
        return self.d * x

    matvec = dot

    @property
    def T(self):
        return self


class DeviceHessian:
    """The assembled Lagrangian Hessian ``p -> sum_h h.dot(p)`` of
    _canonical_constraint.py:119-139 / tr_interior_point.py:222-241.

    Terms are kept in ``hess_list`` order.  When they are one CSR matrix plus
    diagonal terms the product is a single fused kernel (CSR row sum, then
    ``+ diag*p``) and the device-resident CG loop can use it; any other term
    (dense block, user callback) is applied one after the other like the
    reference does.
    """

    def __init__(self, n, csr=None, diag=None, others=(), merge=True):
        self.n = n
        self.shape = (n, n)
        if csr is not None and diag is not None and merge:
            # a CSR term that has every diagonal entry takes the diagonal terms into its values
            # (one scatter-add per Hessian, on a copy: the caller's matrix is not touched): the
            # product then reads no separate diagonal vector -- 8 n bytes less in every CG
            # iteration -- and is the ONE matrix SURVEY.md 8(d) counts (nnz ~ 3n for the
            # benchmark's tridiagonal + diagonal Hessian).  Same operator; the diagonal
            # products are rounded with their row's entry instead of after the row sum.
            # (``merge=False``: the copy + scatter costs two passes over the values, ~20 us at
            # nnz = 3e6 -- more than the diagonal's 8 n bytes cost a Hessian that is multiplied
            # a dozen times before the next one replaces it: backend_hip.hessian_operator)
            pos = csr.pattern.diagonal_positions()
            if pos is not None and len(diag) == n:
                val = csr.val.clone()
                _hip.call("ipx_scatter_add", n, _p(diag.t), _p(pos), _p(val), stream_ptr())
                csr, diag = DeviceCSR(csr.pattern, val), None
        self.csr, self.diag, self.others = csr, diag, tuple(others)

    def dot(self, p):
        if self.csr is not None:
            out = self.csr.spmv(p, diag=self.diag)
        elif self.diag is not None:
            out = self.diag * p
        else:
            out = DVec.zeros(self.n)
        for h in self.others:
            out = out + h.dot(p)
        return out

    matvec = dot
