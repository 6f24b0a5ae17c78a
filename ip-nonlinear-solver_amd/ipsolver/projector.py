"""Projection operators Z, LS, Y on the device (reference projections.py).

Formulation: normal equations, the reference's ``NormalEquation`` method
(projections.py:58-90) -- ``v = (AA')^-1 A x``; ``z = x - A'v`` with the same
orthogonality-driven refinement loop (:69-78) and the same measure
``||A z|| / (||A||_F ||z||)`` (:23-55).  The reference's default sparse method
(AugmentedSystem, SuperLU) and dense method (pivoted QR) compute the same
three operators; SURVEY.md section 7 records agreement to <= 4e-16 on the
benchmark problems.

``(AA')^-1`` on MI355X:
  * sparse A whose ``AA'`` is banded (after a bandwidth-reducing row order found
    once per sparsity pattern on the host -- symbolic work only): partitioned
    banded LDL' kernels (csrc/banded.hip);
  * dense A: Gram matrix + dense Cholesky kernels (csrc/dense.hip).
All numeric work of the full-rank path runs on the GPU.  A numerically rank-deficient
Jacobian (a pivot of the device factorization lost against its diagonal) takes the
reference's own exit: a warning and the SVD projections (projections.py:101-108,
181-187,236-287) -- the thin SVD of the (small, dense) matrix is LAPACK on the host as in
the reference, its factors are uploaded once and every operator application is device
matvecs (``SVDProjector``).
"""
from warnings import warn

import ctypes

import numpy as np
import torch

from . import _hip
from . import device as dv
from .device import DVec, DeviceCSR, _p, stream_ptr, ctx

_F64 = torch.float64


def half_bandwidth_of_aat(pattern):
    """Half bandwidth of ``A A'`` from the pattern of ``A`` alone, O(nnz): rows i and j couple
    iff they share a column, so it is the widest (last row - first row) of a column.  (Forming
    the product pattern for this, and trying a reordering of it, costs 0.1 s on the barrier
    problem's 1e6-row augmented Jacobian, whose band no ordering makes narrow.)"""
    k = getattr(pattern, "_ipx_aat_half_bw", None)
    if k is None:
        m, n = pattern.shape
        rows = np.repeat(np.arange(m, dtype=np.int64), np.diff(pattern.indptr_h))
        cols = pattern.indices_h
        first, last = np.full(n, m, dtype=np.int64), np.full(n, -1, dtype=np.int64)
        last[cols] = rows                      # rows ascend: the last write is the last row
        first[cols[::-1]] = rows[::-1]         # ... and reversed, the first
        used = last >= 0
        k = int(np.max(last[used] - first[used])) if used.any() else 0
        pattern._ipx_aat_half_bw = k
    return k


class _Symbolic:
    """Pattern-level analysis of S = A A' (host, once per pattern)."""

    def __init__(self, pattern):
        import scipy.sparse as sps
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        m, n = pattern.shape
        ones = np.ones(pattern.nnz, dtype=np.float32)
        B = sps.csr_matrix((ones, pattern.indices_h, pattern.indptr_h), shape=(m, n))
        S = sps.csr_matrix(B.dot(B.T))
        S.sort_indices()

        def half_bw(M):
            coo = M.tocoo()
            return int(np.max(np.abs(coo.row - coo.col))) if coo.nnz else 0

        self.k = half_bw(S)
        assert self.k == half_bandwidth_of_aat(pattern), (self.k, half_bandwidth_of_aat(pattern))
        self.perm = None
        if self.k > 1 and m > 2:
            perm = np.ascontiguousarray(reverse_cuthill_mckee(S, symmetric_mode=True),
                                        dtype=np.int32)
            k2 = half_bw(S[perm][:, perm])
            if k2 < self.k:
                self.k, self.perm = k2, perm
        self.m = m

    def __del__(self):
        # handles parked by BandedNormalSolver for the next factorization on this pattern
        for _, handle, _ in self.__dict__.get("_handle_pool", []):
            try:
                _hip.load().ipx_banded_destroy(ctypes.c_void_p(handle))
            except Exception:
                pass


_SYMBOLIC_ATTR = "_ipx_aat_symbolic"


def _symbolic_for(pattern):
    sym = getattr(pattern, _SYMBOLIC_ATTR, None)
    if sym is None:
        sym = _Symbolic(pattern)
        setattr(pattern, _SYMBOLIC_ATTR, sym)
    return sym


HANDLE_STATS = {"created": 0, "pooled": 0, "deferred": 0}     # banded handles (diagnostics)


class BandedNotDecoupled(NotImplementedError):
    """Half bandwidth 5..8 on a long band whose separator blocks do not decouple numerically
    (csrc/banded.hip ipx_banded_create: there is no compiled separator level for them)."""


class BandedNormalSolver:
    """(A A')^-1 for sparse A with banded A A' (half bandwidth <= kmax)."""

    def __init__(self, A, chunk=64, col_weights=None, deferred=None):
        """``col_weights`` (device tensor, one per column of A) factors
        ``A diag(w) A'`` instead (Schur complements, boxschur.py).  ``deferred`` (an object with
        a device tensor ``verdict``: the outer iteration's chain, sqp_chain.py): the blocking
        read that ends a factorization is left out when the handle's previous factorization was
        clean -- the same verdict is assumed, a kernel checks it on the device
        (``ipx_banded_status_deferred``) and the caller reads ``verdict`` with its next block
        (``self.pending`` until then; ``confirm`` reads it on its own)."""
        sym = _symbolic_for(A.pattern)
        kmax = _hip.load().ipx_banded_kmax()
        if sym.k > kmax:
            raise NotImplementedError(
                "A A' has half bandwidth %d after reordering; the device banded "
                "solver handles <= %d (no host fallback)" % (sym.k, kmax))
        self.m = sym.m
        self.k = max(sym.k, 1)
        dev = ctx().device
        self.perm = None
        if sym.perm is not None:
            self.perm = torch.from_numpy(sym.perm).to(dev)          # new row i = old row perm[i]
            inv = np.empty_like(sym.perm)
            inv[sym.perm] = np.arange(self.m, dtype=np.int32)
            self.iperm = torch.from_numpy(inv).to(dev)
        # handle + band storage are recycled per pattern: creating / destroying a handle is a
        # dozen hipMalloc / hipFree calls (~0.5 ms), more than the numeric refresh itself
        lib = _hip.load()
        self._sym = sym          # the pool's handles are destroyed with `sym`: keep it alive
        self._pool = sym.__dict__.setdefault("_handle_pool", [])
        self._key = (self.m, self.k, int(chunk))
        self.handle, self.band = None, None
        for i, (key, handle, band) in enumerate(self._pool):
            if key == self._key:
                self.handle, self.band = handle, band
                del self._pool[i]
                break
        HANDLE_STATS["pooled" if self.handle is not None else "created"] += 1
        if self.handle is None:
            self.band = torch.empty((self.k + 1) * self.m, dtype=_F64, device=dev)
            self.handle = lib.ipx_banded_create(self.m, self.k, int(chunk))
            if not self.handle:
                raise _hip.IpxError("ipx_banded_create failed (m=%d, k=%d)" % (self.m, self.k))
        p = A.pattern
        self.pending, self.ill_conditioned = False, False
        if deferred is not None and self.perm is None and self.k == 1:
            # the whole refresh behind one entry and in three launches where the handle
            # qualifies (csrc/banded.hip ipx_banded_refactor); 0: it does not, nothing enqueued
            rc = lib.ipx_banded_refactor(ctypes.c_void_p(self.handle), self.m, self.k,
                                         _p(p.indptr), _p(p.indices), _p(A.val), _p(col_weights),
                                         _p(self.band), deferred.verdict.data_ptr(), stream_ptr())
            if rc < 0:
                _hip.check(rc, "ipx_banded_refactor")
            if rc == 1:
                self.pending, self._verdict = True, deferred.verdict
                HANDLE_STATS["deferred"] += 1
                return
        _hip.call("ipx_aat_band_w", self.m, self.k, _p(p.indptr), _p(p.indices), _p(A.val),
                  _p(self.perm), _p(col_weights), _p(self.band), stream_ptr())
        _hip.call("ipx_banded_factor", ctypes.c_void_p(self.handle), _p(self.band), stream_ptr())
        if deferred is not None and lib.ipx_banded_status_deferred(
                ctypes.c_void_p(self.handle), deferred.verdict.data_ptr(), stream_ptr()) == 0:
            self.pending, self._verdict = True, deferred.verdict
            HANDLE_STATS["deferred"] += 1
            return
        rc = lib.ipx_banded_status(ctypes.c_void_p(self.handle), stream_ptr())
        if rc == -3:
            raise np.linalg.LinAlgError("Singular Jacobian matrix: A A' is not positive definite")
        # -6: every pivot positive but one lost 43 bits against its diagonal entry: numerically
        # rank deficient.  ``projections`` takes the reference's SVD exit when the matrix is
        # small enough for a dense SVD, and otherwise keeps this factorization (the reference's
        # sparse LU only bails on exact singularity) under the orthogonality-driven refinement
        self.ill_conditioned = rc == -6
        if rc == -6:
            rc = 0
        if rc == -5:
            raise BandedNotDecoupled("A A' (m=%d, half bandwidth %d): separator blocks of the "
                                     "partitioned factorization are coupled" % (self.m, self.k))
        _hip.check(rc, "ipx_banded_status")

    POOL_MAX = 4

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if not h:
            return
        pool = getattr(self, "_pool", None)
        try:
            if pool is not None and len(pool) < self.POOL_MAX:
                pool.append((self._key, h, self.band))       # next factorization on this pattern
            else:
                _hip.load().ipx_banded_destroy(ctypes.c_void_p(h))
        except Exception:
            pass

    def _gather(self, x, idx):
        out = dv._empty(len(x))
        _hip.call("ipx_gather", len(x), _p(x.t), _p(idx), None, None, _p(out), stream_ptr())
        return DVec(out)

    def solve(self, w):
        """v = (A A')^-1 w, in the caller's (unpermuted) row order."""
        if self.perm is not None:
            w = self._gather(w, self.perm)
        out = dv._empty(self.m)
        _hip.call("ipx_banded_solve", ctypes.c_void_p(self.handle), _p(w.t), _p(out), stream_ptr())
        v = DVec(out)
        if self.perm is not None:
            v = self._gather(v, self.iperm)
        return v


class _Op:
    """Operator object returned by ``projections`` (duck type of scipy's
    LinearOperator as used by the reference: shape / dot / matvec)."""

    def __init__(self, shape, fn):
        self.shape = shape
        self._fn = fn

    def dot(self, x):
        return self._fn(x if isinstance(x, DVec) else DVec.from_host(x))

    matvec = dot


class NormalEquationProjector:
    """Z, LS, Y for a device matrix ``A`` with an ``(AA')^-1`` solver object.

    ``stats`` counts solves / refinement steps (SURVEY.md 3.3 measured 0
    refinements on the benchmark; the counters let tests assert the same).
    """

    def __init__(self, A, solver, orth_tol=1e-12, max_refin=3, row_perm=None, lazy_norm=False):
        """``row_perm`` (host ints, or None): ``A`` is the caller's matrix with its rows taken in
        this order (``A = A_caller[row_perm]``: the order in which A A' is banded, see
        ``projections``).  Z does not see the order of the rows; LS returns its multipliers in
        the caller's order and Y takes its right-hand side in it (``rows_in`` for callers that
        hand constraint-space vectors to kernels themselves: cg_fused's priming)."""
        self.A, self.solver = A, solver
        self.orth_tol, self.max_refin = orth_tol, max_refin
        self.m, self.n = A.shape
        self.row_perm = None
        if row_perm is not None:
            inv = np.empty(len(row_perm), dtype=np.int32)
            inv[np.asarray(row_perm)] = np.arange(len(row_perm), dtype=np.int32)
            dev = ctx().device
            self.row_perm = torch.from_numpy(np.ascontiguousarray(row_perm, dtype=np.int32)).to(dev)
            self._row_iperm = torch.from_numpy(inv).to(dev)
        # ||A||_F: computed on first use (a blocking read) -- or handed over by the outer
        # iteration's chain, which folds it into the block it reads anyway (norm_partials)
        self._norm_A = None if (lazy_norm and self.m > 0) else \
            (A.frobenius_norm() if self.m > 0 else 0.0)
        self.stats = {"solves": 0, "refinements": 0, "cancellation_steps": 0}

    @property
    def norm_A(self):
        if self._norm_A is None:
            self._norm_A = self.A.frobenius_norm()
        return self._norm_A

    @norm_A.setter
    def norm_A(self, value):
        self._norm_A = value

    def norm_partials(self):
        """First stage of ``A.frobenius_norm()`` enqueued: (partials, count) -- the fold
        (ipx_sum_partials' order: the bits of the blocking form) is the reader's."""
        val = self.A.val
        g = dv._reduce_grid(val.numel())
        part = dv._empty(2 * g)
        _hip.call("ipx_norms_partials", val.numel(), _p(val), _p(part), stream_ptr())
        return part, g

    # -- reference projections.py:23-55, with ||A z|| from the fused SpMV epilogue
    def _orthogonality(self, z):
        """z's ||z||^2 was left in slot 0 by the SpMV that produced it; A z
        goes to slot 1 so one read-back serves both norms."""
        Az = self.A.spmv(z, reduce=True, slot=1)
        red = self._red = dv.read_slots(7)           # (slot 6: ||x||^2 of null_space's input)
        norm_z, norm_Az = float(np.sqrt(red[0])), float(np.sqrt(red[2]))
        if norm_z == 0 or self.norm_A == 0:
            return 0.0, Az
        return norm_Az / (self.norm_A * norm_z), Az

    def _apply_inv(self, w):
        self.stats["solves"] += 1
        return self.solver.solve(w)

    # x lost this many bits to the row space in z = x - A'v before a correction step is added
    CANCELLATION = 2.0 ** -10

    def null_space(self, x):
        if self.m == 0:
            return x.copy()
        c = dv.ctx()
        # ||x||^2 -> slot 6 (read below together with the orthogonality measure's norms)
        _hip.call("ipx_norms", len(x), dv._p(x.t), ctypes.c_void_p(c.out.data_ptr() + 8 * 6),
                  dv._p(c.ws), dv.stream_ptr())
        v = self._apply_inv(self.A.dot(x))
        z = self.A.rmatvec_sub(v, x, reduce=True)    # x - A'v, ||z||^2 -> slot 0
        k = 0
        while True:                                  # projections.py:72-78
            orth, Az = self._orthogonality(z)
            if k == 0:
                red = self._red
                # Accuracy beyond the reference's loop.  When x lies almost entirely in the row
                # space of A (late barrier subproblems: |Z c| = 1e-6 |c|) the ONE subtraction
                # x - A'v cancels most of x and leaves an error of eps |x| -- 1e-10 of |z| --
                # that the orthogonality measure does not see (it is 1e-13 of ||A||_F |z|):
                # measured 7-9x the reference's augmented-system error on its own late-barrier
                # calls (tests/golden/late_barrier_*.npz).  One correction step on z itself,
                # whose own rounding is eps |z|, removes it: z <- z - A'(A A')^-1 (A z).
                cancelled = red[0] < (self.CANCELLATION ** 2) * red[6]
            else:
                cancelled = False
            if k >= self.max_refin or not (orth > self.orth_tol or cancelled):
                break
            v = self._apply_inv(Az)
            z = self.A.rmatvec_sub(v, z, reduce=True)
            k += 1
            self.stats["refinements" if orth > self.orth_tol else "cancellation_steps"] += 1
        return z

    def null_space_enqueue(self, x, base):
        """``null_space`` without its read-back: z = x - A'(A A')^-1 A x enqueued, and
        ||x||^2, ||z||^2, ||A z||^2 left at doubles ``base + 4``, ``base``, ``base + 2`` of the
        context's reduction block -- the caller (cg_fused.projected_cg's priming) has the
        DEVICE decide from them whether the refinement / cancellation steps above are needed,
        and falls back to ``null_space`` when they are.  (slot = base / 2: 16 bytes each.)"""
        c = dv.ctx()
        _hip.call("ipx_norms", len(x), dv._p(x.t), ctypes.c_void_p(c.out.data_ptr() + 8 * (base + 4)),
                  dv._p(c.ws), dv.stream_ptr())
        v = self._apply_inv(self.A.dot(x))
        z = self.A.rmatvec_sub(v, x, reduce=True, slot=base // 2)
        self.A.spmv(z, reduce=True, slot=base // 2 + 1)
        return z

    def _gather_rows(self, v, idx):
        out = dv._empty(idx.numel())
        _hip.call("ipx_gather", idx.numel(), dv._p(v.t), dv._p(idx), None, None, dv._p(out),
                  dv.stream_ptr())
        return DVec(out)

    def rows_in(self, b):
        """A constraint-space vector of the caller in this projector's row order."""
        return b if self.row_perm is None else self._gather_rows(b, self.row_perm)

    def least_squares(self, x):
        if self.m == 0:
            return DVec.zeros(0)
        v = self._apply_inv(self.A.dot(x))
        for _ in range(getattr(self.solver, "refine_steps", 0)):     # ill-conditioned dense A
            v = v + self._apply_inv(self.A.dot(self.A.rmatvec_sub(v, x)))
        return v if self.row_perm is None else self._gather_rows(v, self._row_iperm)

    def row_space(self, x):
        if self.m == 0:
            return DVec.zeros(self.n)
        x = self.rows_in(x)
        y = self.A.T.dot(self._apply_inv(x))
        for _ in range(getattr(self.solver, "refine_steps", 0)):
            y = y + self.A.T.dot(self._apply_inv(x - self.A.dot(y)))
        return y

    def operators(self):
        ops = (_Op((self.n, self.n), self.null_space),
               _Op((self.m, self.n), self.least_squares),
               _Op((self.n, self.m), self.row_space))
        for op in ops:
            op.projector = self      # lets the fused CG loop recognise its own operators
        return ops


class SVDProjector:
    """Z, LS, Y from a thin SVD ``A = U diag(s) Vt`` with the singular values ``<= tol``
    dropped -- the reference's ``svd_factorization_projections`` (projections.py:236-287)
    statement by statement, on device vectors.  Reached for rank-deficient Jacobians (with
    the reference's warning) and for ``method='SVDFactorization'``."""

    MAX_ELEMENTS = 1 << 25      # dense copy of A on the host: 256 MiB

    def __init__(self, A, orth_tol, max_refin, tol):
        import scipy.linalg
        from .dense import DeviceDense
        self.A = A
        self.m, self.n = A.shape
        if self.m * self.n > self.MAX_ELEMENTS:
            raise np.linalg.LinAlgError(
                "Singular Jacobian matrix (%d x %d): too large for the dense SVD fallback"
                % (self.m, self.n))
        Ah = A.to_scipy().toarray() if isinstance(A, DeviceCSR) else A.to_host()
        U, sv, Vt = scipy.linalg.svd(Ah, full_matrices=False)       # :240
        keep = sv > tol                                             # :243-245
        self.rank = int(np.count_nonzero(keep))
        self.U = DeviceDense.from_host(U[:, keep])
        self.Vt = DeviceDense.from_host(Vt[keep, :])
        self.inv_s = DVec.from_host(1.0 / sv[keep])
        self.orth_tol, self.max_refin = orth_tol, max_refin
        self.norm_A = A.frobenius_norm() if self.m > 0 else 0.0
        self.stats = {"solves": 0, "refinements": 0}

    def _apply_inv(self, x):
        """v = U 1/s V' x = pinv(A') x  (:250-253)"""
        self.stats["solves"] += 1
        if self.rank == 0:
            return DVec.zeros(self.m)
        return self.U.dot(self.inv_s * self.Vt.dot(x))

    def _orth(self, z):
        norm_z = dv.norm(z)
        if norm_z == 0 or self.norm_A == 0:
            return 0.0
        return dv.norm(self.A.dot(z)) / (self.norm_A * norm_z)

    def null_space(self, x):                                        # :248-270
        z = self.A.rmatvec_sub(self._apply_inv(x), x)
        k = 0
        while self._orth(z) > self.orth_tol:
            if k >= self.max_refin:
                break
            z = self.A.rmatvec_sub(self._apply_inv(z), z)
            k += 1
            self.stats["refinements"] += 1
        return z

    def least_squares(self, x):                                     # :273-278
        return self._apply_inv(x)

    def row_space(self, x):                                         # :281-286
        if self.rank == 0:
            return DVec.zeros(self.n)
        return self.Vt.T.dot(self.inv_s * self.U.T.dot(x))

    def operators(self):
        ops = (_Op((self.n, self.n), self.null_space),
               _Op((self.m, self.n), self.least_squares),
               _Op((self.n, self.m), self.row_space))
        for op in ops:
            op.projector = self
        return ops


def orthogonality(A, g):
    """``||A g|| / (||A||_F ||g||)`` (reference projections.py:23-55)."""
    A = as_device_matrix(A)
    g = g if isinstance(g, DVec) else DVec.from_host(g)
    norm_g = dv.norm(g)
    norm_A = A.frobenius_norm()
    if norm_g == 0 or norm_A == 0:
        return 0
    return dv.norm(A.dot(g)) / (norm_A * norm_g)


def _box_schur_applies(A, kmax, any_sparsity=False):
    """Pattern test: enough box-like rows, and the general rows alone are banded -- or, with
    ``any_sparsity``, of any pattern (the Schur complement then goes to the dense or the
    iterative solver: boxschur.BoxSchurNormalSolver)."""
    from .boxschur import analysis_for
    from .device_mode import RowSelection
    an = analysis_for(A.pattern)
    if not an.worthwhile:
        return False
    if any_sparsity:
        return len(an.general) > 0
    cache = getattr(A.pattern, "_ipx_box_general_pattern", None)
    if cache is None:
        cache = A.pattern._ipx_box_general_pattern = RowSelection(A.pattern, an.general, None)
    return _symbolic_for(cache.pattern).k <= kmax


class PcgArgs(ctypes.Structure):
    """Mirror of ipx_pcg_args (include/ipx.h)."""
    _P, _I = ctypes.c_void_p, ctypes.c_int64
    _fields_ = [("m", _I), ("n", _I),
                ("A_rowptr", _P), ("A_colidx", _P), ("A_val", _P), ("A_tiles", _P), ("A_ntiles", _I),
                ("At_rowptr", _P), ("At_colidx", _P), ("At_val", _P), ("At_tiles", _P),
                ("At_ntiles", _I), ("dinv", _P), ("v", _P), ("r", _P), ("p", _P), ("Sp", _P),
                ("t", _P), ("state", _P), ("part1", _P), ("part2", _P), ("grid", _I),
                ("binv", _P), ("border", _P), ("nblk", _I), ("z", _P), ("part3", _P)]


class IterativeNormalSolver:
    """``(A A')^-1`` without a factorization, for sparse Jacobians whose ``A A'`` is neither
    banded (after reordering) nor small enough for the dense device Cholesky: preconditioned
    conjugate gradients on ``A (A' v) = w``, device resident (csrc/pcg.hip):
    one C call enqueues a batch of iterations, convergence and stall tests are taken on the
    device, the host reads one state block per batch.  The reference factors any sparse A
    with SuperLU (projections.py:93-172); this keeps such problems solvable here (at the
    speed of an iterative solve) instead of refusing them.  The inner solve runs to the
    floor of fp64; the projector's orthogonality-driven refinement (projections.py:72-78)
    sits on top of it as usual.

    Preconditioner (``precond``): "block" (default) -- block Jacobi: the diagonal 32 x 32 blocks
    of ``A A'`` with the rows taken in the bandwidth-reducing order of the symbolic analysis
    (reverse Cuthill-McKee of the pattern of ``A A'``), formed, Cholesky-factored and inverted
    on the device once per factorization; "jacobi" -- the diagonal (round 2)."""

    perm = None
    RTOL, MAXIT = 1e-15, 2000
    WARN_RELRES = 1e-10          # a solve that ends above this says so (warning)
    PS_RZ0, PS_BEST0, PS_DONE, PS_ITERS, PS_NORM_W, PS_RTOL, PS_STALL_FAR = 0, 2, 6, 7, 8, 9, 11
    STALL_FAR = 30               # iterations without a new smallest residual that end a solve
                                 # whose residual is still above 1e-9 ||w|| (5 below that)
    BLOCK = 32

    def __init__(self, A, precond="block"):
        lib = _hip.load()
        self.precond = precond
        self.A, self.At = A, A.T
        self.m, n = A.shape
        sq = DVec(A.val) * DVec(A.val)
        rowsq = DeviceCSR(A.pattern, sq.t).dot(DVec.full(n, 1.0))
        d = rowsq.to_host()
        if not np.all(d > 0):
            raise np.linalg.LinAlgError("Singular Jacobian matrix: a row of A is zero")
        self.dinv = DVec.from_host(1.0 / d)
        dev, m = ctx().device, self.m
        z = lambda k: torch.zeros(int(k), dtype=_F64, device=dev)
        self.r, self.p, self.Sp, self.t = z(m), z(m), z(m), z(n)
        self.state = z(lib.ipx_pcg_state_size())
        self.grid = int(lib.ipx_cg_vec_grid(max(m, 1)))
        self.part1, self.part2 = z(2 * A.pattern.ntiles), z(2 * self.grid)
        a = self.args = PcgArgs()
        a.m, a.n = m, n
        for pre, M in (("A", A), ("At", self.At)):
            pat = M.pattern
            setattr(a, pre + "_rowptr", pat.indptr.data_ptr())
            setattr(a, pre + "_colidx", pat.indices.data_ptr())
            setattr(a, pre + "_val", M.val.data_ptr())
            setattr(a, pre + "_tiles", pat.tiles.data_ptr())
            setattr(a, pre + "_ntiles", pat.ntiles)
        a.dinv = self.dinv.t.data_ptr()
        a.r, a.p, a.Sp, a.t = (t.data_ptr() for t in (self.r, self.p, self.Sp, self.t))
        a.state, a.part1, a.part2 = (t.data_ptr() for t in (self.state, self.part1, self.part2))
        a.grid = self.grid
        if precond == "block":
            # rows in the order of the symbolic analysis, padded to whole blocks
            order = _symbolic_for(A.pattern).perm
            order = np.arange(m, dtype=np.int32) if order is None else np.asarray(order, np.int32)
            nblk = (m + self.BLOCK - 1) // self.BLOCK
            padded = np.full(nblk * self.BLOCK, -1, dtype=np.int32)
            padded[:m] = order
            self.border = torch.from_numpy(padded).to(dev)
            self.binv = torch.empty(nblk * self.BLOCK * self.BLOCK, dtype=_F64, device=dev)
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            pat = A.pattern
            _hip.call("ipx_blockjacobi_build", nblk, _p(pat.indptr), _p(pat.indices), _p(A.val),
                      _p(self.border), _p(self.binv), _p(flag), stream_ptr())
            if int(flag.item()) != 0:
                raise np.linalg.LinAlgError("Singular Jacobian matrix: a diagonal block of A A' "
                                            "is not positive definite")
            self.z, self.part3 = z(m), z(nblk // 8 + 2)
            self.dinv = DVec.zeros(m)             # (r'z comes from the block kernel)
            a.dinv = self.dinv.t.data_ptr()
            a.binv, a.border, a.nblk = self.binv.data_ptr(), self.border.data_ptr(), nblk
            a.z, a.part3 = self.z.data_ptr(), self.part3.data_ptr()
        elif precond != "jacobi":
            raise ValueError("precond must be 'block' or 'jacobi'")
        self.stats = {"solves": 0, "iterations": 0, "batches": 0}

    def solve(self, w):
        lib = _hip.load()
        v = DVec.zeros(self.m)
        norm_w = dv.norm(w)
        if norm_w == 0:
            return v
        self.r.copy_(w.t)
        init = np.zeros(self.state.numel())
        if self.precond == "block":
            self.state.zero_()
            _hip.call("ipx_blockjacobi_apply", self.m, self.args.nblk, _p(self.border),
                      _p(self.binv), _p(w.t), _p(self.z), _p(self.part3), _p(self.state),
                      stream_ptr())
            z0 = DVec(self.z)
        else:
            z0 = self.dinv * w
        self.p.copy_(z0.t)
        init[self.PS_RZ0] = w.dot(z0)
        init[self.PS_BEST0] = np.inf
        init[self.PS_NORM_W], init[self.PS_RTOL] = norm_w, self.RTOL
        init[self.PS_STALL_FAR] = self.STALL_FAR
        self.state.copy_(torch.from_numpy(init))
        self.args.v = v.t.data_ptr()
        it, batch = 0, 8
        while it < self.MAXIT:
            end = min(self.MAXIT, it + batch)
            _hip.check(lib.ipx_pcg_iterate(ctypes.byref(self.args), it, end, stream_ptr()),
                       "ipx_pcg_iterate")
            s = self.state.tolist()               # one blocking read per batch
            self.stats["batches"] += 1
            if s[self.PS_DONE] != 0:
                if s[self.PS_DONE] == 3:
                    raise np.linalg.LinAlgError("Singular Jacobian matrix: A A' is not positive "
                                                "definite")
                break
            it, batch = end, min(2 * batch, 64)
        self.stats["solves"] += 1
        self.stats["iterations"] += int(s[self.PS_ITERS])
        # neither converged nor down at the floor of fp64 (MAXIT reached, or no progress far
        # above it): the reference's direct factorization would have been accurate here --
        # say so instead of returning a poor solve silently
        relres = min(s[self.PS_BEST0], s[self.PS_BEST0 + 1]) / norm_w
        self.stats["worst_relres"] = max(self.stats.get("worst_relres", 0.0), relres)
        if s[self.PS_DONE] != 1 and relres > self.WARN_RELRES:
            from warnings import warn
            warn("IterativeNormalSolver: the preconditioned CG on A A' stopped at a relative "
                 "residual of %.1e after %d iterations (ill-conditioned Jacobian; m = %d): "
                 "projections with this factorization are only that accurate"
                 % (relres, int(s[self.PS_ITERS]), self.m))
        return v


def as_device_matrix(A):
    """Upload a scipy sparse matrix / ndarray (device matrices pass through).
    An empty matrix is forced to the sparse representation like the
    reference does (projections.py:368-369)."""
    import scipy.sparse as sps
    from .dense import DeviceDense
    if isinstance(A, DeviceCSR):
        return A.with_sorted_indices()
    if isinstance(A, DeviceDense):
        return A
    m, n = np.shape(A)
    if sps.issparse(A) or m * n == 0:
        return DeviceCSR.from_scipy(sps.csr_matrix(A))
    return DeviceDense.from_host(np.asarray(A, dtype=float))


def normal_solver_for(A, deferred=None):
    """The ``(A A')^-1`` solver ``projections`` picks for a full-row-rank device matrix."""
    from .dense import DenseNormalSolver, DeviceDense
    if isinstance(A, DeviceDense):
        return DenseNormalSolver(A)
    kmax = _hip.load().ipx_banded_kmax()
    m = A.shape[0]
    if half_bandwidth_of_aat(A.pattern) > kmax and _box_schur_applies(A, kmax):
        # the barrier problem's augmented Jacobian: a general row couples with the two bound
        # rows of each of its variables, no reordering of A A' is narrow -- skip the attempt
        from .boxschur import BoxSchurNormalSolver
        try:
            return BoxSchurNormalSolver(A)
        except BandedNotDecoupled:
            pass
    k = _symbolic_for(A.pattern).k
    if k <= kmax:
        try:
            return BandedNormalSolver(A, deferred=deferred)
        except BandedNotDecoupled:
            # Half bandwidths 5-8 on a long band run the single-launch decoupled solve (the
            # separator system, half bandwidth 2k-1, is only formed to test that its blocks
            # decouple).  When they do not, a serial sweep would take 80-100 ms per solve at
            # m = 1e5: the device-resident preconditioned CG is 20x faster and as accurate
            # after the projector's refinement (profiles/r02_banded_by_bandwidth.txt).
            if m > DenseNormalSolver.MAX_ROWS_FROM_SPARSE:
                return IterativeNormalSolver(A)
    if _box_schur_applies(A, kmax):
        from .boxschur import BoxSchurNormalSolver
        try:
            return BoxSchurNormalSolver(A)      # bound rows eliminated analytically
        except BandedNotDecoupled:
            # the Schur complement of the general rows is such a coupled wide band: the
            # elimination buys nothing, solve with A itself by the paths below
            pass
    if m <= DenseNormalSolver.MAX_ROWS_FROM_SPARSE:
        return DenseNormalSolver(A)             # wide band: dense Cholesky of A A'
    if _box_schur_applies(A, kmax, any_sparsity=True):
        # barrier problem with a Jacobian of general sparsity (the reference factors any
        # pattern with SuperLU, projections.py:93-172): the bound rows -- two thirds of the
        # matrix, and the ones whose slacks ruin the conditioning of A A' late in the barrier
        # run -- are still eliminated in closed form; what is left to the dense / iterative
        # solver is the Schur complement of the general rows, J (I - W) J' + S^2
        from .boxschur import BoxSchurNormalSolver
        return BoxSchurNormalSolver(A, any_sparsity=True)
    return IterativeNormalSolver(A)             # general sparsity: matrix-free solve


def _banded_row_order(A):
    """The row order in which A A' is banded when the natural one is not and the banded solver
    would be the choice of ``normal_solver_for`` (not the box-Schur elimination, which has its
    own row bookkeeping); None otherwise."""
    kmax = _hip.load().ipx_banded_kmax()
    if half_bandwidth_of_aat(A.pattern) <= kmax:
        return None
    if _box_schur_applies(A, kmax):
        return None
    sym = _symbolic_for(A.pattern)
    return sym.perm if (sym.k <= kmax and sym.perm is not None) else None


def _rows_in_order(A, perm):
    """``A[perm]`` as a value refresh on a pattern derived once from A's (cached on it)."""
    from .device_mode import RowSelection
    sel = getattr(A.pattern, "_ipx_rows_in_order", None)
    if sel is None:
        sel = A.pattern._ipx_rows_in_order = RowSelection(A.pattern, np.asarray(perm, dtype=np.int64), None)
    return sel.apply(A)


def projections(A, method=None, orth_tol=1e-12, max_refin=3, tol=1e-15, deferred=None):
    """Device counterpart of ``projections`` (projections.py:290-406).

    ``A`` is a DeviceCSR / DeviceDense (or a scipy / numpy matrix, uploaded).
    ``method`` accepts the reference's names; every one maps to the device
    normal-equation solve (the operators are the same; see module docstring).
    """
    from .dense import DenseNormalSolver
    A = as_device_matrix(A)
    # The SAME matrix object with unchanged values: the factorization made for it is returned.
    # Values declared immutable (``constant``: linear constraints, N1) are factored once per
    # solve; any other matrix is recognised by its value tensor's version counter -- the
    # barrier method hands the Jacobian of its last accepted point to the next subproblem
    # (tr_interior_point.py:338-340: the reference refactors it, 11 of config 3's 25
    # factorizations).
    #
    # What the version counter does NOT see: writes through raw pointers (a user kernel, or this
    # library's own ipx_* entry points called on ``A.val`` directly).  A callback that refills a
    # preallocated value tensor that way must say so -- ``projector.invalidate(A)`` -- or go
    # through torch ops (``A.val.copy_(...)``), which bump the counter.  The entry keeps a
    # reference to the tensor it was made for: a reassigned ``A.val`` can then never collide
    # with a collected tensor's id.
    t, version = _values_version(A)
    key = (method, orth_tol, max_refin, tol, version)
    cached = getattr(A, "_ipx_projections", None)
    if cached is not None and version is not None and cached[0] == key and cached[2] is t:
        # (a caller that cannot take a pending verdict along gets it settled first)
        if deferred is not None or confirm(A, cached[1]):
            return cached[1]
    out = _projections(A, method, orth_tol, max_refin, tol, deferred)
    try:
        A._ipx_projections = (key, out, t)
    except AttributeError:
        pass
    return out


def confirm(A, ops):
    """Settle a factorization whose verdict is still pending (``BandedNormalSolver.pending``)
    by a blocking read of its own; a verdict that differs from the assumed one forgets the
    factorization (the next ``projections(A)`` factors again, blocking).  True: ``ops`` stand."""
    P = getattr(ops[0], "projector", None)
    solver = getattr(P, "solver", None)
    if solver is None or not getattr(solver, "pending", False):
        return True
    bad = dv.read_doubles(solver._verdict, 1)[0] != 0
    solver.pending = False
    if bad:
        invalidate(A)
    return not bad


def invalidate(A):
    """Forget the factorization cached on a device matrix (its values were rewritten in place
    behind torch's back: see ``projections``)."""
    try:
        A._ipx_projections = None
    except AttributeError:
        pass


def _values_version(A):
    """(tensor that holds a device matrix's values, its version counter); (None, None) for an
    unknown type: never cached."""
    t = getattr(A, "val", None)
    if t is None:
        t = getattr(A, "t", None)
    return t, getattr(t, "_version", None)


def _projections(A, method, orth_tol, max_refin, tol, deferred=None):
    from .dense import DenseNormalSolver
    sparse = isinstance(A, DeviceCSR)
    if sparse:
        if method not in (None, "NormalEquation", "AugmentedSystem"):
            raise ValueError("Method not allowed for sparse matrix.")
    else:
        if method not in (None, "QRFactorization", "SVDFactorization"):
            raise ValueError("Method not allowed for dense array.")
    m, n = A.shape
    if method == "SVDFactorization":
        return SVDProjector(A, orth_tol, max_refin, tol).operators()
    A_caller, row_perm = A, None
    try:
        if sparse and m > 0:
            # A A' banded only after a reordering of the rows (equality rows stacked on
            # inequality rows: _canonical_constraint.py:169-360): factor the matrix WITH ITS ROWS
            # IN THAT ORDER instead of permuting inside every solve -- the projections do not see
            # the order of the rows, and the device-resident CG loop (cg_fused) takes any matrix
            # whose natural order is banded
            row_perm = _banded_row_order(A)
            if row_perm is not None:
                A = _rows_in_order(A, row_perm)
        solver = None if m == 0 else (normal_solver_for(A) if deferred is None
                                      else normal_solver_for(A, deferred))
        inner = getattr(solver, "inner", solver)           # (box-Schur: its banded Schur solve)
        if getattr(inner, "ill_conditioned", False):
            if m * n <= SVDProjector.MAX_ELEMENTS:
                raise np.linalg.LinAlgError("numerically rank deficient")
            warn("Ill-conditioned Jacobian matrix (a pivot of A A' lost 43 bits); too large for "
                 "the dense SVD fallback: keeping the factorization under iterative refinement.")
    except np.linalg.LinAlgError:
        # the reference's exits: projections.py:101-108 (sparse), :181-187 (dense)
        warn("Singular Jacobian matrix. Using dense SVD decomposition to perform the "
             "factorizations." if sparse else
             "Singular Jacobian matrix. Using SVD decomposition to perform the "
             "factorizations.")
        return SVDProjector(A_caller, orth_tol, max_refin, tol).operators()
    return NormalEquationProjector(A, solver, orth_tol, max_refin, row_perm=row_perm,
                                   lazy_norm=deferred is not None).operators()
