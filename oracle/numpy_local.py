"""Oracle (test infrastructure): numpy twin of ``ipsolver.sharded.HipOps``.

Lets tests/test_sharded_gloo.py run the product's row-partitioned solver (layout, halo
exchange, all-reduces, distributed vectors, and -- through ``ipsolver.qp`` /
``ipsolver.sqp`` -- the reference's algorithms on them) over gloo on CPUs.  Each method
restates the kernel / device class of the same name with numpy / scipy.  Never imported
by the product.
"""
import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla
import torch


class _LocalCSR:
    def __init__(self, M):
        self.M = sps.csr_matrix(M)
        self.shape = self.M.shape

    def dot(self, x):
        return self.M.dot(x)


class _LocalHessian:
    def __init__(self, H, diag):
        self.H, self.diag = H, diag

    def dot(self, p):
        y = self.H.M.dot(p)
        return y + self.diag * p if self.diag is not None else y


class _LocalSolver:
    """(A A')^-1 on the local extended rows: sparse LU of A A'."""

    def __init__(self, A):
        S = sps.csc_matrix(A.M.dot(A.M.T))
        self.lu = spla.splu(S)

    def solve(self, w):
        return self.lu.solve(w)


class _LocalPack:
    """Twin of ipsolver.device.ScalarPack on numpy arrays (immediate values)."""

    def __init__(self):
        self.vals = []

    def _add(self, v):
        self.vals.append(float(v))
        return len(self.vals) - 1

    def dot(self, a, b):
        return self._add(a.dot(b))

    def sumsq(self, v):
        return self._add(v.dot(v))

    def norm_inf(self, v):
        return self._add(np.abs(v).max() if len(v) else 0.0)

    def read(self):
        return list(self.vals)


class NumpyOps:
    name = "numpy-oracle"
    fused = False

    def pack(self):
        return _LocalPack()

    def from_tensor(self, t):
        return t.numpy().copy()

    def row_sumsq(self, A):
        return np.asarray(A.M.multiply(A.M).sum(axis=1)).ravel()

    def from_host(self, a):
        return np.array(a, dtype=np.float64)

    def to_host(self, v):
        return np.array(v)

    def tensor(self, v):
        return torch.from_numpy(v)

    def zeros(self, n):
        return np.zeros(int(n))

    def full(self, n, value):
        return np.full(int(n), float(value))

    def copy(self, v):
        return v.copy()

    def add_scaled(self, x, o, a):
        return x + a * o

    def scaled_sub(self, x, a, o):
        return a * x - o

    def sumsq_amax(self, v):
        return [float(v.dot(v)), float(np.abs(v).max()) if len(v) else 0.0]

    def dot(self, a, b):
        return float(a.dot(b))

    def clip(self, x, lb, ub):
        return np.minimum(np.maximum(x, lb), ub)

    def count_outside_box(self, x, lb, ub):
        return float(np.count_nonzero(~((lb <= x) & (x <= ub))))

    def box_sphere_reduce(self, z, d, dscale, lb, ub):          # csrc/vec.hip RedBoxSphere
        d = dscale * d
        lo = lb if lb is not None else np.full(len(z), -np.inf)
        hi = ub if ub is not None else np.full(len(z), np.inf)
        nz = d != 0
        zero_out = float(np.count_nonzero((~nz) & ((z < lo) | (z > hi))))
        ta, tb = -np.inf, np.inf
        if nz.any():
            with np.errstate(invalid="ignore", divide="ignore"):
                tl, tu = (lo[nz] - z[nz]) / d[nz], (hi[nz] - z[nz]) / d[nz]
            ta, tb = float(np.max(np.minimum(tl, tu))), float(np.min(np.maximum(tl, tu)))
        return [float(d.dot(d)), float(z.dot(d)), float(z.dot(z)), ta, tb, zero_out,
                float(np.count_nonzero(nz))]

    def index(self, idx):
        return np.asarray(idx, dtype=np.int64)

    def take(self, v, idx):                                      # ipx_gather
        return np.asarray(v)[idx]

    def csr(self, M, row_breaks=None, col_breaks=None):
        return _LocalCSR(M)

    def refresh(self, A, data):
        M = A.M.copy()
        M.data = np.array(data, dtype=float)
        return _LocalCSR(M)

    def rmatvec(self, A, v):
        return A.M.T.dot(v)

    def hessian(self, n, H_csr, diag):
        return _LocalHessian(H_csr, diag)

    def normal_solver(self, A):
        return _LocalSolver(A)

    def frob_sq_rows(self, A, r0, r1):
        ip = A.M.indptr
        d = A.M.data[ip[r0]:ip[r1]]
        return float(d.dot(d))

    # -- barrier problems (z = [x; s])
    def concat(self, parts):
        return np.concatenate(parts)

    def maximum(self, v, c):
        return np.maximum(v, c)

    def where_positive(self, v, a, c):
        return np.where(v > 0, a, c)

    def sum_log(self, s):
        pos = s > 0
        return float(np.sum(np.log(s[pos]))), float(np.count_nonzero(~pos))

    def assign_negated_where(self, s, mask, c):                 # tr_interior_point.py:92
        sel = mask != 0
        s[sel] = -c[sel]

    def augmented_box(self, J, s_nl, s_lb, s_ub, col_breaks=None):                  # tr_interior_point.py:141-194
        mE, nX = J.M.shape
        I = sps.identity(nX, format="csr")
        return _LocalCSR(sps.bmat([[J.M, sps.diags(s_nl), None, None],
                                   [-I, None, sps.diags(s_lb), None],
                                   [I, None, None, sps.diags(s_ub)]], format="csr"))

    def hessian_z(self, Hx, slack_block, breaks=None):
        class _Z:
            def __init__(self, Hx, sb):
                self.Hx, self.sb, self.nx = Hx, sb, Hx.H.shape[0]

            def dot(self, p):
                return np.concatenate((self.Hx.dot(p[:self.nx]), self.sb * p[self.nx:]))
        return _Z(Hx, slack_block)

    def any_normal_solver(self, A):
        return _LocalSolver(A)
