"""Row-sharded projections and projected CG for Jacobians of ANY sparsity (SURVEY.md section
8(e), the non-banded case; reference projections.py:93-172 accepts any sparse ``A``).

``ipsolver.sharded`` partitions banded problems with neighbour halos.  Without a band there
is no neighbourhood: a rank's constraint rows may touch any variable.  The partition here is
the plain one of SURVEY 8(e):

* constraint rows and variables are cut into contiguous, (almost) equal blocks; rank g keeps
  rows ``[R0, R1)`` of ``A`` COMPLETE (all n columns) and owns variables ``[c0, c1)``;
* ``A x``: all-gather of ``x`` (8 n bytes per rank and product), then the local rows;
* ``A'v``: every rank forms the partial n-vector of its rows, a reduce-scatter leaves each
  rank the sum on its own variables;
* ``H p``: rows ``[c0, c1)`` of H, all-gather of ``p``;
* ``(A A')^-1 w``: no factorization -- conjugate gradients on ``A (A' v) = w`` preconditioned
  by the diagonal of ``A A'`` (one all-gather + one reduce-scatter + two scalar all-reduces per
  inner iteration), to the floor of fp64, under the projector's usual refinement loop.

Vectors are ``sharded.ShardVec`` on a layout without halos, so ``qp.projected_cg`` /
``modified_dogleg`` / the intersections (the general, host-driven driver) and the outer loops
run unchanged.  This is the fallback that keeps every sparse problem solvable on N ranks; the
banded path with its device-resident loop is the fast one.  Local arithmetic: ``HipOps`` or
the numpy twin (tests/test_sharded_gloo.py).
"""
import numpy as np
import scipy.sparse as sps
import torch
import torch.distributed as dist

from .sharded import ShardVec, Sharding, ShardComm, ShardedBackend, _ShardOp

__all__ = ["GeneralLayout", "GeneralCSR", "GeneralHessian", "GeneralBackend", "projections",
           "general_sharding"]


class GeneralLayout:
    """Contiguous equal blocks of rows and of variables, no halos (the surface
    ``sharded.Sharding`` needs of a layout)."""

    def __init__(self, shape, world, rank):
        m, n = int(shape[0]), int(shape[1])
        if m < world or n < world:
            raise ValueError("%d x %d is too small for %d ranks" % (m, n, world))
        self.m, self.n, self.world, self.rank = m, n, world, rank
        self.row_cuts = [(r * m) // world for r in range(world)] + [m]
        self.col_cuts = [(r * n) // world for r in range(world)] + [n]
        self.row_block, self.halo_rows = 0, 0
        self.ranks = [dict(R0=self.row_cuts[r], R1=self.row_cuts[r + 1], E0=self.row_cuts[r],
                           E1=self.row_cuts[r + 1], c0=self.col_cuts[r], c1=self.col_cuts[r + 1],
                           x0=self.col_cuts[r], x1=self.col_cuts[r + 1]) for r in range(world)]
        self.me = self.ranks[rank]

    def geom(self, kind, rank=None):
        d = self.ranks[self.rank if rank is None else rank]
        if kind == "col":
            return d["c0"], d["c1"] - d["c0"], 0, d["c1"] - d["c0"]
        return d["R0"], d["R1"] - d["R0"], 0, d["R1"] - d["R0"]

    def sends(self, kind):
        return 0, 0

    def global_len(self, kind):
        return self.n if kind == "col" else self.m


def general_sharding(shape, ops, comm=None):
    comm = comm if comm is not None else ShardComm()
    return Sharding(GeneralLayout(shape, comm.world, comm.rank), comm, ops)


def _cuts(sh, kind):
    return sh.lay.col_cuts if kind == "col" else sh.lay.row_cuts


def _gather_full(sh, v):
    """The global vector behind a distributed one as a local array of ``ops``: an all-gather
    of the own blocks (padded to the widest; device to device under RCCL, staged through the
    host under gloo -- the test-only combination)."""
    comm, kind = sh.comm, v.kind
    if comm.world == 1:
        return v.loc
    t = sh.ops.tensor(v.loc)
    cuts = _cuts(sh, kind)
    width = int(max(np.diff(cuts)))
    stage = t.is_cuda and comm.backend != "nccl"
    mine = torch.zeros(width, dtype=t.dtype, device="cpu" if stage else t.device)
    mine[:t.numel()] = t.cpu() if stage else t
    parts = [torch.empty_like(mine) for _ in range(comm.world)]
    comm.stats["all_reduce"] += 1                      # (counted with the collectives)
    comm.stats["all_reduce_bytes"] += 8 * width * comm.world
    dist.all_gather(parts, mine, group=comm.group)
    full = torch.cat([p[:cuts[r + 1] - cuts[r]] for r, p in enumerate(parts)])
    return sh.ops.from_tensor(full.to(t.device) if stage else full)


def _reduce_scatter(sh, partial, kind):
    """Sum of every rank's full-length partial vector, own block kept (collective).  RCCL has
    the primitive; gloo (the CPU tests) does not: there it is an all-reduce + slice."""
    comm = sh.comm
    if comm.world == 1:
        return ShardVec(partial, sh, kind)
    g0, ln, _, _ = sh.lay.geom(kind)
    t = sh.ops.tensor(partial)
    comm.stats["all_reduce"] += 1
    comm.stats["all_reduce_bytes"] += t.numel() * 8
    if comm.backend == "nccl":
        cuts = _cuts(sh, kind)
        width = int(max(np.diff(cuts)))
        padded = torch.zeros(comm.world * width, dtype=t.dtype, device=t.device)
        for r in range(comm.world):
            padded[r * width:r * width + cuts[r + 1] - cuts[r]] = t[cuts[r]:cuts[r + 1]]
        out = torch.empty(width, dtype=t.dtype, device=t.device)
        dist.reduce_scatter_tensor(out, padded, group=comm.group)
        own = out[:ln].clone()
    else:
        h = t.cpu() if t.is_cuda else t.clone()
        dist.all_reduce(h, group=comm.group)
        own = h[g0:g0 + ln].clone()
        own = own.to(t.device) if t.is_cuda else own
    return ShardVec(sh.ops.from_tensor(own), sh, kind)


class GeneralCSR:
    """Rows ``[R0, R1)`` of a sparse matrix, all columns: ``dot`` gathers x, ``T.dot``
    reduce-scatters the partial products."""

    def __init__(self, sh, local, transposed=False, other=None):
        self.sh, self.local, self.transposed, self._T = sh, local, transposed, other
        m, n = sh.lay.m, sh.lay.n
        self.shape = (n, m) if transposed else (m, n)
        self.row_kind, self.col_kind = "row", "col"

    @staticmethod
    def from_global(sh, A):
        d = sh.lay.me
        return GeneralCSR(sh, sh.ops.csr(sps.csr_matrix(sps.csr_matrix(A)[d["R0"]:d["R1"], :])))

    @property
    def T(self):
        if self._T is None:
            self._T = GeneralCSR(self.sh, self.local, not self.transposed, self)
        return self._T

    def dot(self, x):
        sh = self.sh
        if not self.transposed:
            assert x.kind == "col"
            return ShardVec(self.local.dot(_gather_full(sh, x)), sh, "row")
        assert x.kind == "row"
        return _reduce_scatter(sh, sh.ops.rmatvec(self.local, x.loc), "col")

    matvec = dot

    def frobenius_norm(self):
        m_loc = self.local.shape[0]
        tot = self.sh.ops.frob_sq_rows(self.local, 0, m_loc)
        return float(np.sqrt(self.sh.comm.reduce_floats([tot])[0]))

    def row_sumsq(self):
        """diag(A A') on the own rows (the Jacobi preconditioner of the inner solve)."""
        return ShardVec(self.sh.ops.row_sumsq(self.local), self.sh, "row")


class GeneralHessian:
    """Rows ``[c0, c1)`` of the Hessian (+ diagonal term on the own variables)."""

    def __init__(self, sh, local, diag=None):
        self.sh, self.local, self.diag, self.kind = sh, local, diag, "col"
        self.shape = (sh.lay.n, sh.lay.n)

    @staticmethod
    def from_global(sh, H, hdiag=None):
        d = sh.lay.me
        loc = sps.csr_matrix(sps.csr_matrix(H)[d["c0"]:d["c1"], :])
        diag = sh.from_global(np.asarray(hdiag, dtype=float), "col") if hdiag is not None else None
        return GeneralHessian(sh, sh.ops.csr(loc), diag)

    def dot(self, p):
        y = ShardVec(self.local.dot(_gather_full(self.sh, p)), self.sh, "col")
        return y + self.diag * p if self.diag is not None else y

    matvec = dot


class GeneralProjector:
    """Z, LS, Y through the normal equations (projections.py:58-90) with a matrix-free inner
    solve: Jacobi-preconditioned CG on ``A (A' v) = w`` over the ranks."""

    RTOL, MAXIT = 1e-15, 2000

    def __init__(self, A, orth_tol=1e-12, max_refin=3):
        self.A, self.sh = A, A.sh
        self.orth_tol, self.max_refin = orth_tol, max_refin
        d = A.row_sumsq()
        d_h = self.sh.ops.to_host(d.loc)
        dmin = self.sh.comm.reduce_mixed(mins=[float(np.min(d_h)) if len(d_h) else np.inf])[2][0]
        if not dmin > 0:
            raise np.linalg.LinAlgError("Singular Jacobian matrix: a row of A is zero")
        self.dinv = ShardVec(self.sh.ops.from_host(1.0 / d_h), self.sh, "row")
        self.norm_A = A.frobenius_norm()
        self.stats = {"solves": 0, "refinements": 0, "inner_iterations": 0}
        self.fused_sharded = False

    def _apply_inv(self, w):
        """v = (A A')^-1 w by preconditioned CG (the recurrences of csrc/pcg.hip)."""
        self.stats["solves"] += 1
        A = self.A
        v = w.zeros_like()
        norm_w = float(np.sqrt(w.sumsq_amax()[0]))
        if norm_w == 0:
            return v
        r = w.copy()
        z = self.dinv * r
        p = z.copy()
        rz = r.dot(z)
        best, stall = np.inf, 0
        for it in range(self.MAXIT):
            Sp = A.dot(A.T.dot(p))
            pSp = p.dot(Sp)
            if not pSp > 0:
                raise np.linalg.LinAlgError("Singular Jacobian matrix: A A' is not positive "
                                            "definite")
            alpha = rz / pSp
            v = v.add_scaled(p, alpha)
            r = r.add_scaled(Sp, -alpha)
            self.stats["inner_iterations"] += 1
            res = float(np.sqrt(r.sumsq_amax()[0]))
            if res <= self.RTOL * norm_w:
                break
            stall = stall + 1 if res >= best else 0       # fp64 floor: no progress 5 times
            best = min(best, res)
            if stall >= 5:
                break
            z = self.dinv * r
            rz_new = r.dot(z)
            p = z.add_scaled(p, rz_new / rz)
            rz = rz_new
        return v

    def orthogonality(self, z):
        norm_z = np.sqrt(z.sumsq_amax()[0])
        if norm_z == 0 or self.norm_A == 0:
            return 0.0, None
        Az = self.A.dot(z)
        return float(np.sqrt(Az.sumsq_amax()[0]) / (self.norm_A * norm_z)), Az

    def null_space(self, x):                           # projections.py:65-80
        z = x - self.A.T.dot(self._apply_inv(self.A.dot(x)))
        k = 0
        while True:
            orth, Az = self.orthogonality(z)
            if not orth > self.orth_tol or k >= self.max_refin:
                break
            z = z - self.A.T.dot(self._apply_inv(Az))
            k += 1
            self.stats["refinements"] += 1
        return z

    def least_squares(self, x):
        return self._apply_inv(self.A.dot(x))

    def row_space(self, x):
        return self.A.T.dot(self._apply_inv(x))

    def operators(self):
        m, n = self.A.shape
        return (_ShardOp((n, n), self.null_space, self), _ShardOp((m, n), self.least_squares, self),
                _ShardOp((n, m), self.row_space, self))


def projections(A, method=None, orth_tol=1e-12, max_refin=3, tol=1e-15):
    """``projections`` (reference projections.py:290-406) for a ``GeneralCSR``."""
    if method not in (None, "NormalEquation", "AugmentedSystem"):
        raise ValueError("Method not allowed for sparse matrix.")
    return GeneralProjector(A, orth_tol, max_refin).operators()


class GeneralBackend(ShardedBackend):
    """The backend of the outer loops (``sqp.py`` / ``barrier.py``) on the plain block
    partition: what ``minimize_constrained`` dispatches to when the Jacobian has no band for
    ``sharded.ShardLayout`` to follow.  Equality-constrained problems (both outer methods)."""
    name = "sharded-general"

    def matrix(self, J, key="jac"):
        if J is None or isinstance(J, GeneralCSR):
            return J
        if not sps.issparse(J):
            raise NotImplementedError("sharded backend: dense Jacobians are not distributed")
        return GeneralCSR.from_global(self.sh, J)

    def augmented_jacobian(self, J_eq, J_ineq, s, n_vars, n_eq, n_ineq):
        raise NotImplementedError("row-sharded solve without a banded Jacobian: equality "
                                  "constraints only")

    def hessian_operator(self, terms, n_vars, slack_block):
        if slack_block is not None:
            raise NotImplementedError("row-sharded solve without a banded Jacobian: equality "
                                      "constraints only")
        if isinstance(terms, GeneralHessian):
            return terms
        from .canonical import HessianSum
        flat = terms.flat_terms() if isinstance(terms, HessianSum) else list(terms)
        total = None
        for h in flat:
            if isinstance(h, np.ndarray) and h.ndim == 2:
                h = sps.csr_matrix(h)
            if not sps.issparse(h):
                raise NotImplementedError("sharded backend: Hessian terms must be sparse matrices")
            total = sps.csr_matrix(h) if total is None else total + sps.csr_matrix(h)
        if total is None:
            total = sps.csr_matrix((self.sh.lay.n, self.sh.lay.n))
        return GeneralHessian.from_global(self.sh, total)

    def projections(self, A, method=None):
        return projections(A, method)
