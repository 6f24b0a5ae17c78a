"""Device operator types of the seam besides plain matrices."""

from .device import DVec


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

    def __init__(self, n, csr=None, diag=None, others=()):
        self.n = n
        self.shape = (n, n)
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
