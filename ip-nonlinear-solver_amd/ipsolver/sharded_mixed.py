"""Equality AND inequality rows together on the banded (halo) partition of the row-sharded solver.

The reference puts every constraint mix through one path: the canonical form stacks
``[c_eq; c_ineq + s]`` and the barrier method factors ``[[J_eq, 0], [J_ineq, diag(s)]]``
(_canonical_constraint.py:363-480, tr_interior_point.py:141-194).  Stacked like that a banded
Jacobian whose equality and inequality rows INTERLEAVE is not banded any more -- but
``A A' = J J' + diag(0, s^2)`` is, once the rows are taken in the order of the band.  Round 4
made that order the one the single-GPU factorization works in (projector.projections: rows
permuted before the factorization); here it is the order of the PARTITION:

* the rows of ``[J_eq; J_ineq]`` are merged into the order of their first columns; ``ShardLayout``
  cuts THAT matrix into blocks of 260 rows with their variables and halos, exactly as for an
  equality-only problem;
* the equality rows and the inequality rows (= the slacks) are two further distributed spaces,
  subsets of the merged rows: a rank owns the members among its own rows (``ShardLayout``
  ``subsets``).  z = [x; s] is the stacked space ("col", "rin"), the rows of the barrier
  subproblem ("req", "rin") -- what ``barrier.py`` / ``sqp.py`` slice and stack;
* a rank's block of the augmented Jacobian is kept in the MERGED row order
  (``MergedRowsCSR``): its ``A A'`` is tridiagonal like the equality-only problem's, the local
  factorization is the plain banded one, and the device-resident loop of ``sharded.py`` runs
  on it unchanged (two segments of z: the vector kernels with own ranges).
  Only ``A.dot`` / ``A.T.dot`` / ``(A A')^-1`` permute between the stacked order of the outer
  loops' vectors and the merged one, at their boundary (one gather each).

What the band cannot follow -- rows whose kinds do not appear in the order of the band, boxes,
operator Hessians -- stays on the plain partition (``sharded_general``).
"""
import numpy as np
import scipy.sparse as sps

from .sharded import (ShardCSR, ShardHessian, ShardLayout, ShardVec, ShardedBackend, Sharding,
                      _Empty)

__all__ = ["MergedRowsCSR", "MixedBandedBackend", "merged_order", "try_backend"]

STATS = {"backends": 0}        # problems this process put on the merged-order banded partition


def merged_order(J_eq, J_ineq):
    """Order of the rows of ``[J_eq; J_ineq]`` in which the stacked matrix is banded (first and
    last column non-decreasing from row to row), or None: some row is empty, no such order, or
    the rows of one kind would change their relative order (a distributed vector of the kind's
    space is cut along it)."""
    J = sps.vstack([sps.csr_matrix(J_eq), sps.csr_matrix(J_ineq)], format="csr")
    J.sort_indices()
    if J.shape[0] == 0 or np.any(np.diff(J.indptr) == 0):
        return None
    first = J.indices[J.indptr[:-1]].astype(np.int64)
    last = J.indices[J.indptr[1:] - 1].astype(np.int64)
    order = np.lexsort((last, first))                      # (stable: ties keep the stacked order)
    if np.any(np.diff(first[order]) < 0) or np.any(np.diff(last[order]) < 0):
        return None
    n_eq = J_eq.shape[0]
    is_eq = order < n_eq
    if np.any(np.diff(order[is_eq]) < 0) or np.any(np.diff(order[~is_eq]) < 0):
        return None
    return order


class MergedRowsCSR(ShardCSR):
    """The augmented Jacobian of the barrier subproblem on one rank: rows = its (own + halo)
    merged rows, in the merged order; columns = its x entries, then its slack entries.  To the
    outer loops it is a matrix from z = ("col", "rin") to the stacked rows ("req", "rin")."""
    merged = True

    def __init__(self, sh, local, rows_kind, col_kind, to_stacked, to_banded, transposed=False,
                 other=None):
        ShardCSR.__init__(self, sh, local, transposed, other, rows_kind, col_kind)
        self._to_stacked, self._to_banded = to_stacked, to_banded

    def to_stacked(self, loc):
        """a local constraint-space array in the merged order -> in the stacked order"""
        return self.sh.ops.take(loc, self._to_stacked)

    def to_banded(self, loc):
        return self.sh.ops.take(loc, self._to_banded)

    @property
    def T(self):
        if self._T is None:
            self._T = MergedRowsCSR(self.sh, self.local, self.row_kind, self.col_kind,
                                    self._to_stacked, self._to_banded, not self.transposed, self)
        return self._T

    def dot(self, x):
        sh = self.sh
        if not self.transposed:
            assert x.kind == self.col_kind, (x.kind, self.col_kind)
            return ShardVec(self.to_stacked(self.local.dot(x.loc)), sh, self.row_kind)
        assert x.kind == self.row_kind, (x.kind, self.row_kind)
        return sh.sync(ShardVec(sh.ops.rmatvec(self.local, self.to_banded(x.loc)), sh,
                                self.col_kind))

    matvec = dot

    def frobenius_norm(self):
        _, _, lo, hi = self.sh.lay.geom("row")             # (own rows of the merged order)
        tot = self.sh.ops.frob_sq_rows(self.local, lo, hi)
        return float(np.sqrt(self.sh.comm.reduce_floats([tot])[0]))


class MixedBandedBackend(ShardedBackend):
    """``ShardedBackend`` for equality rows + inequality rows of one banded Jacobian (module
    docstring).  Spaces by the callers' names: "x" -> "col", "eq" -> "req", "ineq" -> "rin",
    "z" -> ("col", "rin")."""
    name = "sharded-mixed"
    INEQ = "rin"
    Z = ("col", "rin")
    ROWS = ("req", "rin")

    def __init__(self, sh, order, n_eq, n_ineq):
        ShardedBackend.__init__(self, sh)
        self.order = np.asarray(order, dtype=np.int64)
        self.n_eq, self.n_ineq = int(n_eq), int(n_ineq)
        self._by_name = {"x": "col", "eq": "req", "ineq": "rin", "z": self.Z}
        # (the merged rows' own space "row" is internal; the stacked rows have its length)
        seen, self._ambiguous = {}, set()
        sh.spaces.clear()
        for kind in ("col", "rin", self.Z, "req", self.ROWS):
            ln = sh.global_len(kind)
            if ln in seen:
                sh.spaces.pop(ln, None)
                self._ambiguous.add(ln)
            elif ln not in self._ambiguous:
                seen[ln] = kind
                sh.register(kind)
        lay, ops = sh.lay, sh.ops
        # stacked order of the local rows = [members of "req" among them; members of "rin"]
        E0, mE, _, _ = lay.geom("row")
        is_eq = self.order[E0:E0 + mE] < self.n_eq
        stacked = np.concatenate((np.flatnonzero(is_eq), np.flatnonzero(~is_eq)))
        inv = np.empty(mE, dtype=np.int64)
        inv[stacked] = np.arange(mE)
        self._to_stacked, self._to_banded = ops.index(stacked), ops.index(inv)
        # local slack column of every local merged row (-1: an equality row)
        self._slack_col = np.where(is_eq, -1, np.cumsum(~is_eq) - 1)

    def _kind(self, n, space=None):
        if space is not None:
            kind = self._by_name[space]
            assert self.sh.global_len(kind) == n, (space, n, self.sh.global_len(kind))
            return kind
        if n in self._ambiguous:
            raise NotImplementedError("row-sharded solve (equality + inequality rows on the "
                                      "banded partition): two spaces have %d entries and the "
                                      "caller did not name one" % n)
        return self.sh.kind_of_len(n)

    def zeros(self, n):
        return _Empty() if n == 0 else self.sh.zeros(self._kind(n))

    def matrix(self, J, key="jac"):
        if J is None or isinstance(J, ShardCSR):
            return J
        raise NotImplementedError("sharded backend (mixed rows): a Jacobian outside the barrier "
                                  "subproblem's augmented matrix")

    def augmented_jacobian(self, J_eq, J_ineq, s, n_vars, n_eq, n_ineq):
        """[[J_eq, 0], [J_ineq, diag(s)]] (tr_interior_point.py:141-194): this rank's merged
        rows over its x and slack entries, assembled from the global host matrices (the
        replicated callbacks' values) and the local slacks."""
        sh, lay = self.sh, self.sh.lay
        d = lay.me
        Jm = sps.vstack([sps.csr_matrix(J_eq), sps.csr_matrix(J_ineq)], format="csr")[self.order]
        Jloc = sps.csr_matrix(Jm[d["E0"]:d["E1"], d["x0"]:d["x1"]])
        mE = Jloc.shape[0]
        n_s = lay.geom("rin")[1]
        assert s.kind == "rin", s.kind
        s_h = np.asarray(sh.ops.to_host(s.loc), dtype=float)
        rows = np.flatnonzero(self._slack_col >= 0)
        S = sps.csr_matrix((s_h[self._slack_col[rows]], (rows, self._slack_col[rows])),
                           shape=(mE, n_s))
        loc = sps.hstack([Jloc, S], format="csr")
        loc.sort_indices()
        sig = (loc.shape, loc.nnz, hash(loc.indptr.tobytes()), hash(loc.indices.tobytes()))
        like = self._like.get("aug")
        if like is not None and like[0] == sig:
            local = sh.ops.refresh(like[1], loc.data)
        else:
            _, _, rlo, rhi = lay.geom("row")
            local = sh.ops.csr(loc, row_breaks=[rlo, rhi], col_breaks=self._z_breaks())
        self._like["aug"] = (sig, local)
        return MergedRowsCSR(sh, local, self.ROWS, self.Z, self._to_stacked, self._to_banded)

    def hessian_operator(self, terms, n_vars, slack_block):
        if not isinstance(terms, ShardHessian):
            terms = self._host_hessian(terms)
        if not isinstance(terms, ShardHessian):
            raise NotImplementedError("sharded backend (mixed rows): operator Hessian terms")
        if slack_block is None:
            return terms
        return ShardHessian(self.sh, self.sh.ops.hessian_z(terms.local, slack_block.loc,
                                                          breaks=self._z_breaks()), self.Z)


def try_backend(J_eq, J_ineq, n_vars, ops, comm):
    """The backend for a problem with equality and inequality rows, or None when the banded
    partition cannot follow it (``minimize._sharded_backend`` then takes the plain one).
    Collective: every rank arrives at the same answer (the inputs are replicated, the numerical
    check of the halo truncation is an all-reduce)."""
    order = merged_order(J_eq, J_ineq)
    if order is None:
        return None
    n_eq, n_ineq = J_eq.shape[0], J_ineq.shape[0]
    Jm = sps.vstack([sps.csr_matrix(J_eq), sps.csr_matrix(J_ineq)], format="csr")[order]
    Jm.sort_indices()
    pos = np.arange(len(order))
    try:
        lay = ShardLayout(Jm.indptr, Jm.indices, Jm.shape, comm.world, comm.rank,
                          subsets={"req": pos[order < n_eq], "rin": pos[order >= n_eq]})
    except (NotImplementedError, ValueError):
        return None
    # every rank needs slack entries to own and at least one row of each kind is not required,
    # but an EMPTY local space would make its segment vanish from the stacked vectors
    for r in range(comm.world):
        if lay.geom("rin", r)[1] == 0 or lay.geom("req", r)[1] == 0:
            return None
    sh = Sharding(lay, comm, ops)
    xp = MixedBandedBackend(sh, order, n_eq, n_ineq)
    # the halo partition rests on (A A')^-1 decaying across one block of rows, which the
    # projector measures on the numbers: ask once, at the initial Jacobian with unit slacks
    try:
        xp.projections(xp.augmented_jacobian(J_eq, J_ineq, sh.full("rin", 1.0), n_vars, n_eq,
                                             n_ineq))
    except NotImplementedError:
        return None
    STATS["backends"] += 1
    return xp
