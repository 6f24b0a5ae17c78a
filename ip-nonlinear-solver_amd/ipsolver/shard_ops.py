"""Local arithmetic of the row-sharded solver on one GPU (ipsolver/sharded.py): ``HipOps`` wraps
device.DVec / DeviceCSR and the ipx kernels -- the product path; tests/test_sharded_gloo.py
swaps in the numpy twin of oracle/numpy_local.py to run the same orchestration on CPUs."""
import numpy as np
import scipy.sparse as sps
import torch


class HipOps:
    """Local arithmetic of the sharded solver on one GPU: device.DVec / DeviceCSR and the
    ipx kernels (the product path; fails loudly without the library or a GPU)."""
    name = "hip"
    fused = True

    def __init__(self):
        from . import device as dv
        self.dv = dv
        dv.ctx()

    def from_host(self, a):
        return self.dv.DVec.from_host(a)

    def to_host(self, v):
        return v.to_host()

    def tensor(self, v):
        return v.t

    def zeros(self, n):
        return self.dv.DVec.zeros(n)

    def full(self, n, value):
        return self.dv.DVec.full(n, value)

    def copy(self, v):
        return v.copy()

    def add_scaled(self, x, o, a):
        return x.add_scaled(o, a)

    def scaled_sub(self, x, a, o):
        return x.scaled_sub(a, o)

    def sumsq_amax(self, v):
        return v.sumsq_amax() if len(v) else [0.0, 0.0]

    def dot(self, a, b):
        return a.dot(b) if len(a) else 0.0

    def clip(self, x, lb, ub):
        return self.dv.clip(x, lb, ub)

    def count_outside_box(self, x, lb, ub):
        return self.dv.count_outside_box(x, lb, ub) if len(x) else 0.0

    def box_sphere_reduce(self, z, d, dscale, lb, ub):
        return self.dv.box_sphere_reduce(z, d, dscale, lb, ub)

    def index(self, idx):
        """A host index array as the operand of ``take``."""
        return torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int32)).to(self.dv.ctx().device)

    def take(self, v, idx):
        """v[idx] (a gather kernel)."""
        from . import _hip
        out = self.dv._empty(idx.numel())
        if idx.numel():
            _hip.call("ipx_gather", idx.numel(), self.dv._p(v.t), self.dv._p(idx), None, None,
                      self.dv._p(out), self.dv.stream_ptr())
        return self.dv.DVec(out)

    def csr(self, M, row_breaks=None, col_breaks=None):
        """Local block on the device; row tiles (and those of the stored transpose) are cut
        at the own / halo boundaries so per-tile partial sums can be taken over own tiles."""
        A = self.dv.DeviceCSR.from_scipy(sps.csr_matrix(M), row_breaks=row_breaks)
        if col_breaks is not None:
            A.pattern.transpose(row_breaks=col_breaks)
        return A

    def refresh(self, A, data):
        """Same pattern, new values (host array)."""
        val = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float64)).to(A.val.device)
        return self.dv.DeviceCSR(A.pattern, val)

    def rmatvec(self, A, v):
        return A.T.dot(v)

    def pack(self):
        return self.dv.ScalarPack()

    def from_tensor(self, t):
        return self.dv.DVec(t.contiguous())

    def row_sumsq(self, A):
        """sum_j A_ij^2 per local row (diag of A A')."""
        sq = self.dv.DVec(A.val) * self.dv.DVec(A.val)
        return self.dv.DeviceCSR(A.pattern, sq.t).dot(self.dv.DVec.full(A.shape[1], 1.0))

    def hessian(self, n, H_csr, diag):
        from .operators import DeviceHessian
        return DeviceHessian(n, csr=H_csr, diag=diag)

    def normal_solver(self, A):
        from .projector import BandedNormalSolver
        return BandedNormalSolver(A)

    def frob_sq_rows(self, A, r0, r1):
        ip = A.pattern.indptr_h
        v = self.dv.DVec(A.val[int(ip[r0]):int(ip[r1])])
        return v.sumsq_amax()[0] if len(v) else 0.0

    # -- barrier problems (z = [x; s])
    def concat(self, parts):
        return self.dv.hstack(parts)

    def maximum(self, v, c):
        from . import backend_hip
        return backend_hip.maximum(v, c)

    def where_positive(self, v, a, c):
        from . import backend_hip
        return backend_hip.where_positive(v, a, c)

    def sum_log(self, s):
        """(sum of log s_i over s_i > 0, number of s_i <= 0)"""
        from . import _hip
        if len(s) == 0:
            return 0.0, 0.0
        c = self.dv.ctx()
        _hip.call("ipx_sum_log", len(s), self.dv._p(s.t), self.dv._p(c.out), self.dv._p(c.ws),
                  self.dv.stream_ptr())
        total, bad = self.dv.read_slots(2)
        return total, bad

    def assign_negated_where(self, s, mask, c):
        """s[mask != 0] = -c[mask != 0] in place (tr_interior_point.py:92)."""
        from . import _hip
        if len(s):
            _hip.call("ipx_assign_negated_where", len(s), self.dv._p(s.t), self.dv._p(mask.t),
                      self.dv._p(c.t), self.dv.stream_ptr())

    def augmented_box(self, J, s_nl, s_lb, s_ub, col_breaks=None):
        """Local block of the barrier problem's augmented Jacobian for nonlinear inequality
        rows + a box on every variable (tr_interior_point.py:141-194 on the canonical rows of
        _canonical_constraint.py:350-355: nonlinear rows, all lower bounds, all upper bounds):

            [ J   diag(s_nl)      0           0      ]
            [ -I      0       diag(s_lb)      0      ]
            [ +I      0           0       diag(s_ub) ]

        on a pattern built once per Jacobian pattern; a refresh is four scatters.
        ``col_breaks``: row-tile boundaries of the stored transpose (the own / halo
        boundaries of the z segments, for per-tile partial sums over own entries)."""
        from . import _hip
        dvm = self.dv
        pat = J.pattern
        cache = getattr(pat, "_ipx_aug_box", None)
        mE, nX = pat.shape
        if cache is None:
            ip = pat.indptr_h.astype(np.int64)
            cnt = np.diff(ip)
            rows_nl = ip + np.arange(mE + 1)                       # one slack entry per row
            nnz_nl = int(rows_nl[-1])
            indptr = np.concatenate((rows_nl, nnz_nl + 2 * np.arange(1, 2 * nX + 1)))
            nnz = int(indptr[-1])
            indices = np.empty(nnz, dtype=np.int32)
            template = np.zeros(nnz)
            pos_J = (np.arange(pat.nnz) + np.repeat(np.arange(mE), cnt)).astype(np.int64)
            indices[pos_J] = pat.indices_h
            pos_snl = rows_nl[1:] - 1
            indices[pos_snl] = nX + np.arange(mE)
            base = nnz_nl + 2 * np.arange(nX)
            indices[base], template[base] = np.arange(nX), -1.0            # -I
            pos_slb = base + 1
            indices[pos_slb] = nX + mE + np.arange(nX)
            base2 = nnz_nl + 2 * nX + 2 * np.arange(nX)
            indices[base2], template[base2] = np.arange(nX), 1.0           # +I
            pos_sub = base2 + 1
            indices[pos_sub] = nX + mE + nX + np.arange(nX)
            apat = dvm.CSRPattern(indptr.astype(np.int32), indices, (mE + 2 * nX, nX + mE + 2 * nX))
            if col_breaks is not None:
                apat.transpose(row_breaks=col_breaks)
            dev = dvm.ctx().device
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
            cache = pat._ipx_aug_box = (apat, torch.from_numpy(template).to(dev), t(pos_J), t(pos_snl),
                                        t(pos_slb), t(pos_sub))
        apat, template, pos_J, pos_snl, pos_slb, pos_sub = cache
        val = template.clone()
        st = dvm.stream_ptr()
        for src, idx in ((J.val, pos_J), (s_nl.t, pos_snl), (s_lb.t, pos_slb), (s_ub.t, pos_sub)):
            if idx.numel():
                _hip.call("ipx_scatter", idx.numel(), dvm._p(src), dvm._p(idx), dvm._p(val), st)
        return dvm.DeviceCSR(apat, val)

    def hessian_z(self, Hx, slack_block, breaks=None):
        """[[Hx, 0], [0, diag(slack_block)]] for the local x-space operator ``Hx`` (row tiles
        cut at ``breaks``)."""
        from . import backend_hip
        from .operators import DeviceHessian
        n_x = Hx.shape[0] if hasattr(Hx, "shape") else Hx.n
        n_tot = n_x + len(slack_block)
        csr = self.dv.DeviceCSR(backend_hip._extend_pattern(Hx.csr.pattern, n_tot, breaks), Hx.csr.val)
        xdiag = Hx.diag if Hx.diag is not None else self.dv.DVec.zeros(n_x)
        return DeviceHessian(n_tot, csr, self.dv.hstack((xdiag, slack_block)))

    def any_normal_solver(self, A):
        """(A A')^-1 for a local block of any supported structure: the selection of
        ``projector.projections`` (banded, box rows eliminated analytically, dense)."""
        from . import projector
        return projector.normal_solver_for(A)


