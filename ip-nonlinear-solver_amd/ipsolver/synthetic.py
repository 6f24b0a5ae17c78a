"""Synthetic benchmark inputs -- own code, no reference
counterpart (the reference ships no benchmark).

``CenteredBandedNLP(n, m)`` is the seeded sparse banded NLP used for BASELINE
configs 3-5 and for the projected-CG microbench:

* ``A0``: m x n CSR, ``bw`` contiguous nonzeros per row starting at
  ``clip(i*stride + stride//2 - 7, 0, n-bw)``, ``stride = n // m``.
* ``Q = tridiag(-e, 2+d, -e)`` (SPD), objective
  ``f = 1/2 dl'Q dl - eps q'dl + rho/4 sum(dl^4)``, ``dl = x - x_feas``.
* constraint ``c(x) = A0 x + kappa/2 W (x*x) - b`` with ``W = A0*A0`` and ``b``
  chosen so that ``c(x_feas) = 0``; Jacobian has A0's pattern (value refresh
  only); constraint Hessian ``diag(kappa W'v)``.

Everything here is host-side numpy/scipy; ``device_callbacks`` re-expresses the
callbacks on device buffers for the GPU configs (user-land code: it may use
torch freely, the library's own arithmetic never does).
"""
import numpy as np
import scipy.sparse as sps


class CenteredBandedNLP:
    def __init__(self, n, m, bw=15, seed=0, kappa=0.1, rho=0.1, eps=1e-3):
        rng = np.random.default_rng(seed)
        a = rng.standard_normal(m * bw)
        d = rng.uniform(0, 1, n)
        e = rng.uniform(0, 1, n - 1)
        q = rng.standard_normal(n)
        x_feas = rng.uniform(-1, 1, n)
        self.x0 = x_feas + 0.1 * np.random.default_rng(seed + 12345) \
            .standard_normal(n)

        stride = n // m
        start = np.clip(np.arange(m) * stride + stride // 2 - 7, 0, n - bw)
        indptr = (np.arange(m + 1) * bw).astype(np.int32)
        indices = (start[:, None] + np.arange(bw)[None, :]) \
            .astype(np.int32).ravel()
        self.A0 = sps.csr_matrix((a, indices, indptr), shape=(m, n))
        self.W = sps.csr_matrix((a * a, indices, indptr), shape=(m, n))
        self.Wt = sps.csr_matrix(self.W.T)
        self.Q = sps.diags([-e, 2 + d, -e], [-1, 0, 1], format='csr')
        self.Q.indices = self.Q.indices.astype(np.int32)
        self.Q.indptr = self.Q.indptr.astype(np.int32)
        self.q, self.x_feas = q, x_feas
        self.n, self.m, self.bw = n, m, bw
        self.kappa, self.rho, self.eps = kappa, rho, eps
        self.b = self.A0.dot(x_feas) + 0.5 * kappa * self.W.dot(x_feas ** 2)
        # position of the diagonal entry inside each CSR row of Q
        self._qdiag = np.flatnonzero(
            self.Q.indices == np.repeat(np.arange(n), np.diff(self.Q.indptr)))

    # ---- objective ------------------------------------------------------
    def fun(self, x):
        dl = x - self.x_feas
        return (0.5 * dl.dot(self.Q.dot(dl)) - self.eps * self.q.dot(dl)
                + 0.25 * self.rho * np.sum(dl ** 4))

    def grad(self, x):
        dl = x - self.x_feas
        return self.Q.dot(dl) - self.eps * self.q + self.rho * dl ** 3

    def hess(self, x):
        dl = x - self.x_feas
        H = self.Q.copy()
        H.data[self._qdiag] += 3 * self.rho * dl ** 2
        return H

    # ---- constraint -----------------------------------------------------
    def constr_fun(self, x):
        return self.A0.dot(x) + 0.5 * self.kappa * self.W.dot(x * x) - self.b

    def constr_jac(self, x):
        data = self.A0.data + self.kappa * self.W.data * x[self.A0.indices]
        return sps.csr_matrix((data, self.A0.indices, self.A0.indptr),
                              shape=self.A0.shape)

    def constr_hess(self, x, v):
        return sps.diags(self.kappa * self.Wt.dot(v), format='csr')

    # ---- assembled pieces used by the microbench ------------------------
    def lagrangian_hessian_matrix(self, x, v):
        """hess(x) + constr_hess(x, v) as one CSR matrix (nnz = 3n-2)."""
        dl = x - self.x_feas
        H = self.Q.copy()
        H.data[self._qdiag] += 3 * self.rho * dl ** 2 \
            + self.kappa * self.Wt.dot(v)
        return H

    def constraints(self, ns, kind=('equals', 0)):
        """Constraint objects built from namespace ``ns`` (this package, or
        the reference when generating golden vectors)."""
        return ns.NonlinearConstraint(self.constr_fun, kind, self.constr_jac,
                                      self.constr_hess)


class CenteredDenseNLP:
    """Dense NONLINEAR equality constraints: the case in which the Jacobian -- and with it the
    factorization of the projections -- changes at every accepted step (the reference spends
    76 % of such a run in its pivoted QR, projections.py:179; BASELINE config 2 itself has a
    constant Jacobian and factors once).  ``A ~ N(0, 1)`` m x n, ``H = G G' / n + I``,
    objective ``1/2 dl'H dl - eps q'dl`` with ``dl = x - x_feas`` (centred like the banded
    generator: |f| stays small next to gtol), constraint ``c(x) = A x + kappa/2 W (x*x) - b``
    with ``W = A*A`` and ``c(x_feas) = 0``: Jacobian ``A + kappa W diag(x)`` (dense, new
    values every step), constraint Hessian ``diag(kappa W'v)``."""

    def __init__(self, n, m, seed=0, kappa=0.1, eps=1e-3):
        rng = np.random.default_rng(seed)
        self.A = rng.standard_normal((m, n))
        G = rng.standard_normal((n, n)) / np.sqrt(n)
        self.H = G.dot(G.T) + np.eye(n)
        self.q = rng.standard_normal(n)
        self.x_feas = rng.uniform(-1, 1, n)
        self.x0 = self.x_feas + 0.1 * np.random.default_rng(seed + 12345).standard_normal(n)
        self.W = self.A * self.A
        self.n, self.m, self.kappa, self.eps = n, m, kappa, eps
        self.b = self.A.dot(self.x_feas) + 0.5 * kappa * self.W.dot(self.x_feas ** 2)

    def fun(self, x):
        dl = x - self.x_feas
        return 0.5 * dl.dot(self.H.dot(dl)) - self.eps * self.q.dot(dl)

    def grad(self, x):
        return self.H.dot(x - self.x_feas) - self.eps * self.q

    def hess(self, x):
        return self.H

    def constr_fun(self, x):
        return self.A.dot(x) + 0.5 * self.kappa * self.W.dot(x * x) - self.b

    def constr_jac(self, x):
        return self.A + self.kappa * self.W * x[None, :]

    def constr_hess(self, x, v):
        return sps.diags(self.kappa * self.W.T.dot(v), format='csr')

    def constraints(self, ns, kind=('equals', 0)):
        return ns.NonlinearConstraint(self.constr_fun, kind, self.constr_jac, self.constr_hess)


class DenseDeviceCallbacks:
    """``CenteredDenseNLP`` with every callback on the GPU (device-callback mode: 2-D CUDA
    tensors for the Jacobian and the Hessian; user-land code, torch freely)."""

    def __init__(self, prob):
        import torch
        self.torch = torch
        dev = torch.device("cuda", torch.cuda.current_device())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        self.p = prob
        self.A, self.W, self.H = t(prob.A), t(prob.W), t(prob.H)
        self.Wt = self.W.t().contiguous()
        self.q, self.x_feas, self.b, self.x0 = t(prob.q), t(prob.x_feas), t(prob.b), t(prob.x0)

    @classmethod
    def on_device(cls, n, m, seed=0, kappa=0.1, eps=1e-3):
        """The same family at a size whose host generation would take longer than the solve
        (BASELINE config 2's n = 10000, m = 2000: bench.py): A and the vectors from the numpy
        generator, ``H = G G' / n + I`` formed on the device from a seeded torch generator."""
        import torch
        from types import SimpleNamespace
        self = cls.__new__(cls)
        self.torch = torch
        dev = torch.device("cuda", torch.cuda.current_device())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        rng = np.random.default_rng(seed)
        self.A = t(rng.standard_normal((m, n)))
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        G = torch.randn((n, n), dtype=torch.float64, device=dev, generator=gen) / np.sqrt(n)
        self.H = G @ G.t() + torch.eye(n, dtype=torch.float64, device=dev)
        del G
        self.q = t(rng.standard_normal(n))
        self.x_feas = t(rng.uniform(-1, 1, n))
        self.x0 = self.x_feas + 0.1 * t(np.random.default_rng(seed + 12345).standard_normal(n))
        self.W = self.A * self.A
        self.Wt = self.W.t().contiguous()
        self.b = self.A @ self.x_feas + 0.5 * kappa * (self.W @ (self.x_feas * self.x_feas))
        self.p = SimpleNamespace(n=n, m=m, kappa=kappa, eps=eps)
        return self

    def fun(self, x):
        dl = x - self.x_feas
        return float(0.5 * dl.dot(self.H @ dl) - self.p.eps * self.q.dot(dl))

    def grad(self, x):
        return self.H @ (x - self.x_feas) - self.p.eps * self.q

    def hess(self, x):
        return self.H

    def constr_fun(self, x):
        return self.A @ x + 0.5 * self.p.kappa * (self.W @ (x * x)) - self.b

    def constr_jac(self, x):
        return self.A + self.p.kappa * self.W * x[None, :]

    def constr_hess(self, x, v):
        return self.p.kappa * (self.Wt @ v)                      # diagonal

    def constraints(self, ns, kind=('equals', 0)):
        return ns.NonlinearConstraint(self.constr_fun, kind, self.constr_jac, self.constr_hess)


class DeviceCallbacks:
    """The same NLP with every callback on the GPU (device-callback mode of
    ``minimize_constrained``): user-land code, so it uses torch elementwise ops
    freely; matrix-vector products go through the library's DeviceCSR."""

    def __init__(self, prob):
        import torch
        from .device import DeviceCSR, DVec
        self.torch, self.DVec, self.DeviceCSR = torch, DVec, DeviceCSR
        dev = torch.device("cuda", torch.cuda.current_device())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        self.p = prob
        self.Q = DeviceCSR.from_scipy(prob.Q)
        self.A0 = DeviceCSR.from_scipy(prob.A0)
        self.W = DeviceCSR(self.A0.pattern, t(prob.W.data))
        self.Wt = self.W.T
        self.col = self.A0.pattern.indices.long()
        self.qdiag = torch.from_numpy(prob._qdiag).to(dev)
        self.q, self.x_feas, self.b = t(prob.q), t(prob.x_feas), t(prob.b)
        self.x0 = t(prob.x0)

    def fun(self, x):
        dl = x - self.x_feas
        qd = self.Q.dot(self.DVec(dl)).t
        return float(0.5 * dl.dot(qd) - self.p.eps * self.q.dot(dl)
                     + 0.25 * self.p.rho * (dl ** 4).sum())

    def grad(self, x):
        dl = x - self.x_feas
        return self.Q.dot(self.DVec(dl)).t - self.p.eps * self.q + self.p.rho * dl ** 3

    def hess(self, x):
        dl = x - self.x_feas
        val = self.Q.val.clone()
        val[self.qdiag] += 3 * self.p.rho * dl ** 2
        return self.DeviceCSR(self.Q.pattern, val)

    def constr_fun(self, x):
        return (self.A0.dot(self.DVec(x)).t
                + 0.5 * self.p.kappa * self.W.dot(self.DVec(x * x)).t - self.b)

    def constr_jac(self, x):
        return self.DeviceCSR(self.A0.pattern,
                              self.A0.val + self.p.kappa * self.W.val * x[self.col])

    def constr_hess(self, x, v):
        return self.p.kappa * self.Wt.dot(self.DVec(v)).t        # diagonal

    def constraints(self, ns, kind=('equals', 0)):
        return ns.NonlinearConstraint(self.constr_fun, kind, self.constr_jac, self.constr_hess)


class DistributedCallbacks:
    """The same NLP for ``minimize_constrained`` on the row-sharded backend with DEVICE
    callbacks (``minimize._minimize_distributed``): the reference's callback set -- ``fun``,
    ``grad``, ``hess`` and a ``NonlinearConstraint(fun, kind, jac, hess)`` -- over distributed
    vectors; every rank holds its own rows and variables (+ halos), nothing is gathered."""

    def __init__(self, prob, sh):
        self._c = ShardedCallbacks(prob, sh)
        self.p, self.sh, self.x0 = prob, sh, self._c.x0
        self.fun, self.grad = self._c.fun, self._c.grad
        self.constr_fun, self.constr_jac = self._c.constr_fun, self._c.constr_jac
        from . import sharded
        self._Q = sharded.ShardHessian(sh, sh.ops.hessian(self._c._nloc, self._c._Q_local, None))
        self._Q._csr = self._c._Q_local

    def hess(self, x):
        dl = x - self._c.x_feas
        return self._Q, 3 * self.p.rho * (dl * dl)        # (matrix term, diagonal term)

    def constr_hess(self, x, v):
        return self.p.kappa * self._c.W.T.dot(v)          # diagonal

    def constraints(self, ns, kind=('equals', 0)):
        return ns.NonlinearConstraint(self.constr_fun, kind, self.constr_jac, self.constr_hess)


class LeanDeviceCallbacks(DeviceCallbacks):
    """The same callbacks written the way a user who watches the solve's wall clock writes them
    (user-land all the same: torch for the elementwise work, the library's public device types
    for products and dot products).  What changes against ``DeviceCallbacks`` is launches, not
    mathematics: ``x - x_feas``, ``Q (x - x_feas)`` and its square are formed once per point
    (``fun`` at a trial point and ``grad`` after its acceptance get the SAME tensor), the three
    dot products of ``fun`` are enqueued into one ``ScalarPack`` and read together, the
    products' scalings ride in ``spmv``'s alpha / beta, ``hess`` returns the constant matrix
    and its point-dependent DIAGONAL as two terms (device-callback mode accepts a tuple of
    terms) instead of copying the matrix and scattering into it.  ~20 launches per outer
    iteration instead of ~60."""

    def __init__(self, prob):
        super().__init__(prob)
        self._x = None

    def _point(self, x):
        # (recognised by its storage and version counter, the tensor kept alive so that the
        # address cannot be handed out again: the solver passes views of one iterate)
        seen = self._x
        if seen is None or seen.data_ptr() != x.data_ptr() or seen.numel() != x.numel() \
                or self._ver != x._version:
            dl = x - self.x_feas
            self._pt = (dl, self.Q.dot(self.DVec(dl)).t, dl * dl)
            self._x, self._ver = x, x._version
        return self._pt

    def fun(self, x):
        from .device import ScalarPack
        dl, qd, d2 = self._point(x)
        D = self.DVec
        pk = ScalarPack()
        h = (pk.dot(D(dl), D(qd)), pk.dot(D(self.q), D(dl)), pk.dot(D(d2), D(d2)))
        # 0.5 dl'Q dl - eps q'dl + 0.25 rho sum dl^4, left on the device: the step's verdict
        # consumes it there (the value a host expression over the three sums would have)
        return pk.combine(h, (0.5, -self.p.eps, 0.25 * self.p.rho))

    def grad(self, x):
        dl, qd, d2 = self._point(x)
        g = self.torch.add(qd, self.q, alpha=-self.p.eps)
        return g.addcmul_(d2, dl, value=self.p.rho)

    def hess(self, x):
        _, _, d2 = self._point(x)
        return self.Q, d2 * (3 * self.p.rho)          # (matrix term, diagonal term)

    def constr_fun(self, x):
        D = self.DVec
        lin = self.A0.dot(D(x))
        out = self.W.spmv(D(x * x), alpha=0.5 * self.p.kappa, beta=1.0, yin=lin)
        return out.t.sub_(self.b)

    def constr_jac(self, x):
        return self.DeviceCSR(self.A0.pattern,
                              self.torch.addcmul(self.A0.val, self.W.val, x[self.col],
                                                 value=self.p.kappa))

    def constr_hess(self, x, v):
        return self.Wt.spmv(self.DVec(v), alpha=self.p.kappa).t      # diagonal


class ShardedCallbacks:
    """The same NLP on row-sharded data (``ipsolver.sharded``): every callback works on the
    rank's own + halo entries and returns distributed objects.  User-land code written with
    the distributed vectors' own arithmetic, so it runs on whatever local backend the
    ``Sharding`` context carries (HIP kernels; the numpy twin in the CPU tests)."""

    def __init__(self, prob, sh):
        from . import sharded
        self.p, self.sh, self.S = prob, sh, sharded
        self.Q = sharded.ShardHessian.from_global(sh, prob.Q)
        self.A0 = sharded.ShardCSR.from_global(sh, prob.A0)
        self.W = self.A0.with_values(self._local_data(prob.W))
        self.q = sh.from_global(prob.q, "col")
        self.x_feas = sh.from_global(prob.x_feas, "col")
        self.b = sh.from_global(prob.b, "row")
        self.x0 = sh.from_global(prob.x0, "col")
        d = sh.lay.me
        loc = sps.csr_matrix(prob.A0[d["E0"]:d["E1"], d["x0"]:d["x1"]])
        self._a0 = loc.data.copy()
        self._w = self._local_data(prob.W)
        self._cols = loc.indices.astype(np.int64)
        Ql = sps.csr_matrix(prob.Q[d["x0"]:d["x1"], d["x0"]:d["x1"]])
        _, _, lo, hi = sh.lay.geom("col")
        self._Q_local = sh.ops.csr(Ql, row_breaks=[lo, hi])
        self._nloc = Ql.shape[0]
        self._torch = None
        if hasattr(self.x0.loc, "t"):             # device arrays: gather with torch (user-land)
            import torch
            self._torch = torch
            dev = self.x0.loc.t.device
            self._a0_d = torch.from_numpy(self._a0).to(dev)
            self._w_d = torch.from_numpy(self._w).to(dev)
            self._cols_d = torch.from_numpy(self._cols).to(dev)

    def _local_data(self, M):
        d = self.sh.lay.me
        return sps.csr_matrix(M[d["E0"]:d["E1"], d["x0"]:d["x1"]]).data.copy()

    def fun(self, x):
        dl = x - self.x_feas
        d2 = dl * dl
        return (0.5 * dl.dot(self.Q.dot(dl)) - self.p.eps * self.q.dot(dl)
                + 0.25 * self.p.rho * d2.dot(d2))

    def grad(self, x):
        dl = x - self.x_feas
        return self.Q.dot(dl) - self.p.eps * self.q + self.p.rho * (dl * dl * dl)

    def constr_fun(self, x):
        return self.A0.dot(x) + 0.5 * self.p.kappa * self.W.dot(x * x) - self.b

    def constr_jac(self, x):
        if self._torch is not None:
            val = self._a0_d + self.p.kappa * self._w_d * x.loc.t[self._cols_d]
            from .device import DeviceCSR
            return self.S.ShardCSR(self.sh, DeviceCSR(self.A0.local.pattern, val))
        return self.A0.with_values(self._a0 + self.p.kappa * self._w * x.loc[self._cols])

    def lagr_hess(self, x, v):
        dl = x - self.x_feas
        diag = 3 * self.p.rho * (dl * dl) + self.p.kappa * self.W.T.dot(v)
        return self.S.ShardHessian(self.sh, self.sh.ops.hessian(self._nloc, self._Q_local,
                                                                diag.loc))
