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
run unchanged.  Problems WITH inequality rows -- any mix of equality, inequality and box
constraints the reference accepts -- run the barrier method on the same partition: the slack
space is one more block-partitioned space, ``z = [x; s]`` and the rows ``[c_eq; c_ineq + s]`` are
stacked vectors, the augmented Jacobian ``[[J_eq, 0], [J_ineq, diag(s)]]``
(tr_interior_point.py:141-194) is assembled from the global host matrices every accepted step
and cut into the ranks' rows.  This is the fallback that keeps every sparse problem solvable on
N ranks; the banded path with its device-resident loop is the fast one.  Local arithmetic: ``HipOps`` or
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
    """Contiguous equal blocks of every space, no halos (the surface ``sharded.Sharding`` needs
    of a layout).  Spaces: "row" (constraint rows of the partitioned Jacobian), "col"
    (variables) and any further named ones (``extra``: the barrier problem's inequality rows /
    slacks)."""

    def __init__(self, shape, world, rank, extra=None):
        m, n = int(shape[0]), int(shape[1])
        self.sizes = {"row": m, "col": n}
        self.sizes.update({k: int(v) for k, v in (extra or {}).items()})
        for k, size in self.sizes.items():
            if 0 < size < world or (size == 0 and k == "col"):
                raise ValueError("space %r of %d entries is too small for %d ranks"
                                 % (k, size, world))
        self.m, self.n, self.world, self.rank = m, n, world, rank
        self.cuts = {k: [(r * size) // world for r in range(world)] + [size]
                     for k, size in self.sizes.items()}
        self.row_cuts, self.col_cuts = self.cuts["row"], self.cuts["col"]
        self.row_block, self.halo_rows = 0, 0
        self.ranks = [dict(R0=self.row_cuts[r], R1=self.row_cuts[r + 1], E0=self.row_cuts[r],
                           E1=self.row_cuts[r + 1], c0=self.col_cuts[r], c1=self.col_cuts[r + 1],
                           x0=self.col_cuts[r], x1=self.col_cuts[r + 1]) for r in range(world)]
        self.me = self.ranks[rank]

    def geom(self, kind, rank=None):
        c = self.cuts[kind]
        r = self.rank if rank is None else rank
        return c[r], c[r + 1] - c[r], 0, c[r + 1] - c[r]

    def sends(self, kind):
        return 0, 0

    def global_len(self, kind):
        return self.sizes[kind]


def general_sharding(shape, ops, comm=None, extra=None):
    comm = comm if comm is not None else ShardComm()
    return Sharding(GeneralLayout(shape, comm.world, comm.rank, extra), comm, ops)


def _gather_segment(sh, t, kind):
    """All-gather of one space's own blocks (padded to the widest; device to device under
    RCCL, staged through the host under gloo -- the test-only combination)."""
    comm = sh.comm
    cuts = sh.lay.cuts[kind]
    width = int(max(np.diff(cuts)))
    stage = t.is_cuda and comm.backend != "nccl"
    mine = torch.zeros(width, dtype=t.dtype, device="cpu" if stage else t.device)
    mine[:t.numel()] = t.cpu() if stage else t
    parts = [torch.empty_like(mine) for _ in range(comm.world)]
    comm.stats["all_reduce"] += 1                      # (counted with the collectives)
    comm.stats["all_reduce_bytes"] += 8 * width * comm.world
    dist.all_gather(parts, mine, group=comm.group)
    full = torch.cat([p[:cuts[r + 1] - cuts[r]] for r, p in enumerate(parts)])
    return full.to(t.device) if stage else full


def _gather_full(sh, v):
    """The global vector behind a distributed one as a local array of ``ops`` (the segments of
    a stacked vector in their global order)."""
    if sh.comm.world == 1:
        return v.loc
    t = sh.ops.tensor(v.loc)
    parts = [_gather_segment(sh, t[off:off + ln], k)
             for k, off, ln, *_ in sh.segments(v.kind) if sh.lay.global_len(k)]
    return sh.ops.from_tensor(parts[0] if len(parts) == 1 else torch.cat(parts))


def _reduce_scatter(sh, partial, kind):
    """Sum of every rank's full-length partial vector, own blocks kept (collective).  RCCL has
    the primitive; gloo (the CPU tests) does not: there it is an all-reduce + slice."""
    comm = sh.comm
    if comm.world == 1:
        return ShardVec(partial, sh, kind)
    t = sh.ops.tensor(partial)
    comm.stats["all_reduce"] += 1
    comm.stats["all_reduce_bytes"] += t.numel() * 8
    segs = [sg for sg in sh.segments(kind) if sh.lay.global_len(sg[0])]
    own = []
    if comm.backend == "nccl":
        for k, _, ln, _, _, _, _, goff, glen in segs:
            cuts = sh.lay.cuts[k]
            width = int(max(np.diff(cuts)))
            padded = torch.zeros(comm.world * width, dtype=t.dtype, device=t.device)
            for r in range(comm.world):
                padded[r * width:r * width + cuts[r + 1] - cuts[r]] = \
                    t[goff + cuts[r]:goff + cuts[r + 1]]
            out = torch.empty(width, dtype=t.dtype, device=t.device)
            dist.reduce_scatter_tensor(out, padded, group=comm.group)
            own.append(out[:ln].clone())
    else:
        h = t.cpu() if t.is_cuda else t.clone()
        dist.all_reduce(h, group=comm.group)
        for k, _, ln, _, _, _, _, goff, _ in segs:
            g0 = sh.lay.geom(k)[0]
            piece = h[goff + g0:goff + g0 + ln].clone()
            own.append(piece.to(t.device) if t.is_cuda else piece)
    return ShardVec(sh.ops.from_tensor(own[0] if len(own) == 1 else torch.cat(own)), sh, kind)


class GeneralCSR:
    """The rank's rows of a sparse matrix, all columns: ``dot`` gathers x, ``T.dot``
    reduce-scatters the partial products.  Row and column spaces may be stacked (the barrier
    problem's augmented Jacobian maps z = ("col", "ineq") to the rows ("row", "ineq"))."""

    def __init__(self, sh, local, transposed=False, other=None, row_kind="row", col_kind="col"):
        self.sh, self.local, self.transposed, self._T = sh, local, transposed, other
        self.row_kind, self.col_kind = row_kind, col_kind
        self.full = None             # the whole matrix (host, replicated) when built from one
        m, n = sh.global_len(row_kind), sh.global_len(col_kind)
        self.shape = (n, m) if transposed else (m, n)

    @staticmethod
    def from_global(sh, A, row_kind="row", col_kind="col"):
        A = sps.csr_matrix(A)
        assert A.shape == (sh.global_len(row_kind), sh.global_len(col_kind)), A.shape
        blocks = []
        for k, _, ln, _, _, _, _, goff, _ in sh.segments(row_kind):
            g0 = sh.lay.geom(k)[0]
            blocks.append(A[goff + g0:goff + g0 + ln, :])
        loc = blocks[0] if len(blocks) == 1 else sps.vstack(blocks, format="csr")
        out = GeneralCSR(sh, sh.ops.csr(sps.csr_matrix(loc)), row_kind=row_kind,
                         col_kind=col_kind)
        out.full = A
        return out

    @property
    def T(self):
        if self._T is None:
            self._T = GeneralCSR(self.sh, self.local, not self.transposed, self, self.row_kind,
                                 self.col_kind)
        return self._T

    def dot(self, x):
        sh = self.sh
        if not self.transposed:
            assert _same_kind(x.kind, self.col_kind), (x.kind, self.col_kind)
            return ShardVec(self.local.dot(_gather_full(sh, x)), sh, self.row_kind)
        assert _same_kind(x.kind, self.row_kind), (x.kind, self.row_kind)
        return _reduce_scatter(sh, sh.ops.rmatvec(self.local, x.loc), self.col_kind)

    matvec = dot

    def frobenius_norm(self):
        m_loc = self.local.shape[0]
        tot = self.sh.ops.frob_sq_rows(self.local, 0, m_loc)
        return float(np.sqrt(self.sh.comm.reduce_floats([tot])[0]))

    def row_sumsq(self):
        """diag(A A') on the own rows (the Jacobi preconditioner of the inner solve)."""
        return ShardVec(self.sh.ops.row_sumsq(self.local), self.sh, self.row_kind)


def _same_kind(a, b):
    from .sharded import _kinds
    return _kinds(a) == _kinds(b)


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


class GeneralHessianZ:
    """``[Hx p_x ; slack_block * p_s]`` on z = [x; s] (tr_interior_point.py:222-241): the rows of
    the x-space Hessian the rank owns, the slack block elementwise on its own slacks."""

    def __init__(self, Hx, slack_block, n_vars):
        self.Hx, self.slack, self.n_vars, self.sh = Hx, slack_block, n_vars, Hx.sh
        n = n_vars + len(slack_block)
        self.shape = (n, n)

    def dot(self, p):
        from .sharded import _kinds
        y_x = self.Hx.dot(p[:self.n_vars])
        y_s = self.slack * p[self.n_vars:]
        kinds = _kinds(y_x.kind) + _kinds(y_s.kind)
        return ShardVec(self.sh.ops.concat([y_x.loc, y_s.loc]), self.sh, kinds)

    matvec = dot


class GeneralProjector:
    """Z, LS, Y through the normal equations (projections.py:58-90).  ``(A A')^-1``: matrix-free,
    Jacobi-preconditioned CG on ``A (A' v) = w`` over the ranks -- or, ``replicate``, the
    option SURVEY.md 8(e) names for the constraint-space solve: all-gather the right-hand side
    and solve on every rank with a factorization of the WHOLE ``A A'`` (the local arithmetic's
    own solver: the device factorizations of ``projector.normal_solver_for`` for ``HipOps``).
    The barrier subproblems take the second form -- their ``A A'`` loses its conditioning with
    the slacks, which the diagonal preconditioner does not follow -- and need the matrix
    replicated on the host, which the host-callback route gives."""

    RTOL, MAXIT = 1e-15, 2000

    def __init__(self, A, orth_tol=1e-12, max_refin=3, replicate=False):
        self.A, self.sh = A, A.sh
        self.orth_tol, self.max_refin = orth_tol, max_refin
        self.stats = {"solves": 0, "refinements": 0, "inner_iterations": 0}
        self.fused_sharded = False
        self.solver = None
        if replicate:
            if A.full is None:
                raise NotImplementedError("replicated constraint-space solve: the whole matrix "
                                          "is not at hand")
            self.solver = self.sh.ops.any_normal_solver(self.sh.ops.csr(sps.csr_matrix(A.full)))
            self.norm_A = A.frobenius_norm()
            return
        d = A.row_sumsq()
        d_h = self.sh.ops.to_host(d.loc)
        dmin = self.sh.comm.reduce_mixed(mins=[float(np.min(d_h)) if len(d_h) else np.inf])[2][0]
        if not dmin > 0:
            raise np.linalg.LinAlgError("Singular Jacobian matrix: a row of A is zero")
        self.dinv = ShardVec(self.sh.ops.from_host(1.0 / d_h), self.sh, A.row_kind)
        self.norm_A = A.frobenius_norm()

    def _apply_inv(self, w):
        """v = (A A')^-1 w: the replicated factorization on the gathered right-hand side, or
        preconditioned CG over the ranks (the recurrences of csrc/pcg.hip)."""
        self.stats["solves"] += 1
        A = self.A
        if self.solver is not None:
            full = self.solver.solve(_gather_full(self.sh, w))
            parts = []
            for k, _, ln, _, _, _, _, goff, _ in self.sh.segments(w.kind):
                g0 = self.sh.lay.geom(k)[0]
                parts.append(full[goff + g0:goff + g0 + ln])
            return ShardVec(self.sh.ops.concat(parts) if len(parts) > 1 else
                            self.sh.ops.copy(parts[0]), self.sh, w.kind)
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


def projections(A, method=None, orth_tol=1e-12, max_refin=3, tol=1e-15, replicate=None):
    """``projections`` (reference projections.py:290-406) for a ``GeneralCSR``; ``replicate``:
    see ``GeneralProjector`` (default: for the barrier problem's stacked row space)."""
    if method not in (None, "NormalEquation", "AugmentedSystem"):
        raise ValueError("Method not allowed for sparse matrix.")
    if replicate is None:
        replicate = A.row_kind != "row" and A.full is not None
    return GeneralProjector(A, orth_tol, max_refin, replicate).operators()


class GeneralBackend(ShardedBackend):
    """The backend of the outer loops (``sqp.py`` / ``barrier.py``) on the plain block
    partition: what ``minimize_constrained`` dispatches to when the problem is not one of the
    two shapes ``sharded.ShardLayout`` follows (a banded equality Jacobian; banded nonlinear
    inequalities + an interval box on every variable).  Any mix of equality and inequality
    rows: the spaces "col" (variables), "row" (equality rows) and "ineq" (inequality rows /
    slacks) are block-partitioned, z = ("col", "ineq") and the rows of the barrier subproblem
    ("row", "ineq") are stacked vectors."""
    name = "sharded-general"

    def __init__(self, sh, n_ineq=0):
        super().__init__(sh)
        self.n_ineq = int(n_ineq)
        n_eq = sh.lay.sizes["row"]
        self.INEQ = "ineq"
        self.Z = ("col", "ineq")
        self.ROWS = ("row", "ineq") if n_eq else "ineq"
        # the callers name the space of every vector they create (asvec / full: "x", "eq",
        # "ineq", "z"), so spaces of equal size -- a one-sided bound on every variable makes
        # n_ineq = n_vars -- stay apart; a bare length is looked up only where it is unambiguous
        self._by_name = {"x": "col", "eq": "row", "ineq": "ineq",
                         "z": self.Z if self.n_ineq else "col"}
        if self.n_ineq:
            seen = {}
            for kind in ("col", "ineq", self.Z) + (("row", self.ROWS) if n_eq else ()):
                ln = sh.global_len(kind)
                if ln in seen:
                    sh.spaces.pop(ln, None)
                    self._ambiguous = getattr(self, "_ambiguous", set()) | {ln}
                elif ln not in getattr(self, "_ambiguous", ()):
                    seen[ln] = kind
                    sh.register(kind)

    def _kind(self, n, space=None):
        if space is not None:
            kind = self._by_name[space]
            assert self.sh.global_len(kind) == n, (space, n, self.sh.global_len(kind))
            return kind
        if n in getattr(self, "_ambiguous", ()):
            raise NotImplementedError("row-sharded solve on the general partition: two spaces "
                                      "have %d entries and the caller did not name one" % n)
        return self.sh.kind_of_len(n)

    def matrix(self, J, key="jac"):
        if J is None or isinstance(J, GeneralCSR):
            return J
        # (a dense Jacobian is the rows of a full CSR here)
        return GeneralCSR.from_global(self.sh, sps.csr_matrix(J))

    def augmented_jacobian(self, J_eq, J_ineq, s, n_vars, n_eq, n_ineq):
        """[[J_eq, 0], [J_ineq, diag(s)]] (tr_interior_point.py:141-194) from the global host
        matrices and the gathered slacks, cut into the ranks' rows."""
        s_h = self.tohost(s)
        blocks = [[sps.csr_matrix(J_ineq), sps.diags(s_h)]]
        if n_eq:
            blocks.insert(0, [sps.csr_matrix(J_eq), None])
        A = sps.bmat(blocks, format="csr")
        return GeneralCSR.from_global(self.sh, A, row_kind=self.ROWS, col_kind=self.Z)

    def _x_hessian(self, terms):
        if isinstance(terms, GeneralHessian) or hasattr(terms, "parts"):
            return terms
        from .canonical import HessianSum
        from .sharded import HostOperatorTerm, OperatorSum, _is_host_operator
        flat = terms.flat_terms() if isinstance(terms, HessianSum) else list(terms)
        total, host_ops = None, []
        for h in flat:
            if isinstance(h, np.ndarray) and h.ndim == 2:
                h = sps.csr_matrix(h)
            if _is_host_operator(h):              # finite differences, LinearOperator terms
                host_ops.append(HostOperatorTerm(self.sh, h))
                continue
            if not sps.issparse(h):
                raise NotImplementedError("sharded backend: a Hessian term is neither a sparse "
                                          "matrix nor an operator")
            total = sps.csr_matrix(h) if total is None else total + sps.csr_matrix(h)
        if total is None and host_ops:
            return host_ops[0] if len(host_ops) == 1 else OperatorSum(self.sh, host_ops)
        if total is None:
            total = sps.csr_matrix((self.sh.lay.n, self.sh.lay.n))
        H = GeneralHessian.from_global(self.sh, total)
        return OperatorSum(self.sh, [H] + host_ops) if host_ops else H

    def hessian_operator(self, terms, n_vars, slack_block):
        Hx = self._x_hessian(terms)
        if slack_block is None:
            return Hx
        return GeneralHessianZ(Hx, slack_block, n_vars)

    def projections(self, A, method=None):
        # through minimize_constrained the matrices come from replicated host callbacks, so the
        # whole A is at hand on every rank: the constraint-space solve is the replicated
        # factorization (robust for ill-conditioned A A', one collective per application); the
        # distributed CG stays for matrices that exist only as their ranks' rows
        return projections(A, method, replicate=A.full is not None)
