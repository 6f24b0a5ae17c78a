"""Row-partitioned trust-region subproblem solver across the GPUs of one node
(BASELINE configs 4 and 5).

One process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI).

Partition.  The constraint rows are cut into blocks of ``row_block`` rows (the rows one
workgroup of the single-launch banded solve owns, csrc/banded.hip) and every rank gets a
contiguous run of blocks together with the variables those rows bring in: rank g owns the
rows ``[R0, R1)`` and the variables whose FIRST constraint is one of them -- BOTH spaces are
partitioned, nothing is replicated.
Next to what it owns a rank keeps HALO copies: one block of rows on either side and the
variables those rows touch.  Its local problem -- rows ``E = own + halo rows`` of the
Jacobian (complete rows), the matching rows of ``A'`` and of the Hessian -- is an ordinary
banded problem, so every local operation is the single-GPU kernel (or, in the tests, its
numpy restatement) on the extended local arrays:

* elementwise operations act on own + halo entries alike and keep the copies consistent;
* ``A x`` is exact on every local row (rows are complete);
* ``A'v``, ``H p`` and ``(A A')^-1 w`` are exact on the own entries and wrong only near the
  far ends of the halo: ``(A A')^-1`` of the banded benchmark decays geometrically (what the
  decoupled single-launch solve relies on, checked numerically at every factorization,
  DESIGN.md section 4), so truncating the system one block (260 rows) away changes an own
  entry by less than 2^-56 of its size; the wrong halo entries are then overwritten by the
  owners' values in a neighbour exchange.

Communication per projected-CG iteration of the device-resident loop
(``FusedShardedCG``): TWO small all-reduces (``p'Hp``; then ``||x+ap||^2``, the box
violation count, ``||g||^2``, ``||A g||^2`` packed in 4 doubles) and ONE neighbour
exchange of the halo of ``g`` (point-to-point, at most a block's columns per side,
overlapping the second all-reduce).  The reduced scalars are bit-identical on every rank,
so the device-side branches of csrc/cg.hip go the same way everywhere.

``ShardVec`` has the surface of ``device.DVec`` (arithmetic, ``dot``, ``sumsq_amax`` ...),
with reductions summed over the ranks, so the reference's algorithms in ``qp.py``
(projected CG with its box / trust-region / negative-curvature logic, modified dogleg,
intersections: qp_subproblem.py:66-643) and the outer loops ``sqp.py`` / ``barrier.py``
run on distributed vectors unchanged (``backend_sharded``).

Local arithmetic goes through a small ``ops`` object: ``HipOps`` (ipx kernels; the product)
or the numpy twin in ``oracle/numpy_local.py`` that lets tests/test_sharded_gloo.py run the
same orchestration over gloo on CPUs.
"""
import ctypes
import os

import numpy as np
import scipy.sparse as sps
import torch
import torch.distributed as dist

from .shard_layout import ROW_BLOCK, ShardLayout                      # noqa: F401  (re-exported)
from .shard_comm import ShardComm, PeerMailbox, _device_id, _host_id   # noqa: F401
from .shard_ops import HipOps                                          # noqa: F401


# --------------------------------------------------------------------------- vectors
class _Empty:
    """A distributed vector of global length 0 (the slack / inequality-multiplier slices of a
    problem without inequalities)."""
    kind = None

    def __len__(self):
        return 0


# --------------------------------------------------------------------------- context
def _kinds(kind):
    return (kind,) if isinstance(kind, str) else tuple(kind)


class Sharding:
    """Layout + communicator + local arithmetic of one sharded problem.

    A distributed vector lives in a SPACE: "col" (one entry per variable), "row" (one per
    constraint row of the partitioned Jacobian) or a tuple of those -- the segments of a
    stacked vector such as the barrier problem's z = [x; s_nl; s_lb; s_ub] = ("col", "row",
    "col", "col").  Locally a vector is the concatenation of its segments' extended arrays
    (own + halo entries each); globally the concatenation of the segments' global vectors."""

    def __init__(self, layout, comm, ops):
        self.lay, self.comm, self.ops = layout, comm, ops
        self._segs = {}
        self._mailbox = None
        self.spaces = {}                 # global length -> space (what xp.zeros(n) means)
        if layout.n != layout.m:
            self.spaces[layout.n], self.spaces[layout.m] = "col", "row"

    def register(self, kind):
        self.spaces[self.global_len(kind)] = kind if isinstance(kind, str) else tuple(kind)

    def mailbox(self):
        """The peer mailboxes of this group (``PeerMailbox``; created collectively on first use)
        or None: one rank, the numpy twin, ``IPX_SHARD_TRANSPORT=dist``, or hipIpc not available
        between the ranks -- the device loop then reduces through ``torch.distributed``."""
        if self._mailbox is None:
            self._mailbox = False
            if self.comm.world > 1 and getattr(self.ops, "fused", False) \
                    and os.environ.get("IPX_SHARD_TRANSPORT", "ipc") == "ipc":
                # room for the halo of the largest stacked vector (the barrier problem's z)
                segs = self.segments(("col", "row", "col", "col"))
                cap = max(sum(sg[3] for sg in segs), sum(sg[2] - sg[4] for sg in segs)) + 64
                mb = PeerMailbox(self.comm, cap)
                self.mailbox_error = mb.error
                if mb.ok:
                    self._mailbox = mb
                    # (the outer loops' small collectives take the same road from here on)
                    if os.environ.get("IPX_SHARD_OUTER", "ipc") == "ipc":
                        self.comm.mbox = mb
                        self.comm.on_mailbox_drop = self._drop_mailbox
        return self._mailbox or None

    def _drop_mailbox(self):
        os.environ["IPX_SHARD_TRANSPORT"] = "dist"
        self._mailbox = False
        self.comm.mbox = None

    @property
    def transport(self):
        return "ipc" if self.mailbox() is not None else "dist"

    def segments(self, kind):
        """Per segment: (kind, local offset, local length, own_lo, own_hi, send_left,
        send_right, global offset, global length)."""
        key = _kinds(kind)
        t = self._segs.get(key)
        if t is None:
            t, off, goff = [], 0, 0
            for k in key:
                _, ln, lo, hi = self.lay.geom(k)
                sl, sr = self.lay.sends(k)
                t.append((k, off, ln, lo, hi, sl, sr, goff, self.lay.global_len(k)))
                off += ln
                goff += self.lay.global_len(k)
            self._segs[key] = t
        return t

    def local_len(self, kind):
        return sum(seg[2] for seg in self.segments(kind))

    def global_len(self, kind):
        return sum(seg[8] for seg in self.segments(kind))

    # -- construction of distributed vectors
    def from_global(self, a, kind):
        a = np.asarray(a, dtype=float)
        assert a.shape == (self.global_len(kind),), (a.shape, kind)
        parts = []
        for k, _, ln, _, _, _, _, goff, _ in self.segments(kind):
            g0 = self.lay.geom(k)[0]
            parts.append(a[goff + g0:goff + g0 + ln])
        return ShardVec(self.ops.from_host(np.concatenate(parts)), self, kind)

    def zeros(self, kind):
        return ShardVec(self.ops.zeros(self.local_len(kind)), self, kind)

    def full(self, kind, value):
        return ShardVec(self.ops.full(self.local_len(kind), value), self, kind)

    def kind_of_len(self, k):
        if k in self.spaces:
            return self.spaces[k]
        raise ValueError("no distributed space of length %d is registered" % k)

    def sync(self, v):
        """Overwrite the halo entries of ``v`` with their owners' values."""
        t = self.ops.tensor(v.loc)
        segs = self.segments(v.kind)
        if len(segs) == 1:
            _, off, ln, lo, hi, sl, sr, _, _ = segs[0]
            self.comm.exchange(t[off:off + ln], lo, hi, sl, sr)
        else:
            self.comm.exchange_many([(t[off:off + ln], lo, hi, sl, sr)
                                     for _, off, ln, lo, hi, sl, sr, _, _ in segs], whole=t)
        return v


class ShardVec:
    """Distributed fp64 vector: the local extended array (own + halo entries of every
    segment) with the arithmetic surface of ``device.DVec``; reductions run over the own
    entries and are summed over the ranks."""
    __slots__ = ("loc", "sh", "kind")
    __array_priority__ = 1000

    def __init__(self, loc, sh, kind):
        self.loc, self.sh, self.kind = loc, sh, kind if isinstance(kind, str) else tuple(kind)

    def _new(self, loc):
        return ShardVec(loc, self.sh, self.kind)

    def owns(self):
        """The own part of every segment (views)."""
        return [self.loc[off + lo:off + hi]
                for _, off, _, lo, hi, _, _, _, _ in self.sh.segments(self.kind)]

    def own(self):
        parts = self.owns()
        assert len(parts) == 1
        return parts[0]

    def __len__(self):
        return self.sh.global_len(self.kind)

    @property
    def shape(self):
        return (len(self),)

    def copy(self):
        return self._new(self.sh.ops.copy(self.loc))

    def zeros_like(self):
        return self.sh.zeros(self.kind)

    def full_like(self, value):
        return self.sh.full(self.kind, value)

    def __getitem__(self, key):
        """Slices along segment boundaries (``z[:n_vars]``, ``z[n_vars:]``): a VIEW on the
        selected segments -- barrier.py writes the slacks through it
        (tr_interior_point.py:62-63,92)."""
        if not isinstance(key, slice) or key.step not in (None, 1):
            raise TypeError("ShardVec supports slices along segment boundaries only")
        n = len(self)
        start, stop, _ = key.indices(n)
        if stop <= start:
            return _Empty()
        segs = self.sh.segments(self.kind)
        first = [i for i, sg in enumerate(segs) if sg[7] == start]
        last = [i for i, sg in enumerate(segs) if sg[7] + sg[8] == stop]
        if not first or not last:
            raise NotImplementedError("slice [%d:%d] does not follow the segments of %r"
                                      % (start, stop, self.kind))
        i0, i1 = first[0], last[0]
        kinds = tuple(sg[0] for sg in segs[i0:i1 + 1])
        lo, hi = segs[i0][1], segs[i1][1] + segs[i1][2]
        return ShardVec(self.loc[lo:hi], self.sh, kinds[0] if len(kinds) == 1 else kinds)

    def to_host(self):
        """The global vector on every rank (collective)."""
        sh = self.sh
        out = []
        for (k, off, _, lo, hi, _, _, _, _) in sh.segments(self.kind):
            own = np.ascontiguousarray(sh.ops.to_host(self.loc[off + lo:off + hi]))
            if sh.comm.world == 1:
                out.append(own)
                continue
            named = getattr(sh.lay, "cuts", None)       # (the general partition's named spaces)
            cuts = named[k] if named is not None else \
                (sh.lay.col_cuts if k == "col" else sh.lay.row_cuts)
            sizes = np.diff(cuts)
            pad = np.zeros(int(sizes.max()))
            pad[:len(own)] = own
            mine = torch.from_numpy(pad)
            if sh.comm.backend == "nccl":
                mine = mine.to(torch.device("cuda", torch.cuda.current_device()))
            parts = [torch.empty_like(mine) for _ in range(sh.comm.world)]
            dist.all_gather(parts, mine, group=sh.comm.group)
            out.append(np.concatenate([p.cpu().numpy()[:kk] for p, kk in zip(parts, sizes)]))
        return np.concatenate(out)

    # -- elementwise (own and halo alike: copies stay consistent)
    def _other(self, o):
        assert isinstance(o, ShardVec) and o.kind == self.kind, \
            "mismatched distributed vectors: %r vs %r" % (self.kind, getattr(o, "kind", o))
        return o.loc

    def add_scaled(self, o, a):
        return self._new(self.sh.ops.add_scaled(self.loc, self._other(o), a))

    def scaled_sub(self, a, o):
        return self._new(self.sh.ops.scaled_sub(self.loc, a, self._other(o)))

    def __add__(self, o):
        return self._new(self.loc + (self._other(o) if isinstance(o, ShardVec) else float(o)))

    __radd__ = __add__

    def __sub__(self, o):
        return self._new(self.loc - (self._other(o) if isinstance(o, ShardVec) else float(o)))

    def __rsub__(self, o):
        return self._new(float(o) - self.loc)

    def __neg__(self):
        return self._new(-self.loc)

    def __mul__(self, o):
        return self._new(self.loc * (self._other(o) if isinstance(o, ShardVec) else float(o)))

    __rmul__ = __mul__

    # -- reductions (own entries of every segment, then over the ranks)
    def dot(self, o):
        ops = self.sh.ops
        v = sum(ops.dot(a, b) for a, b in zip(self.owns(), o.owns()))
        return self.sh.comm.reduce_floats([v])[0]

    def sumsq_amax(self):
        ss, am = 0.0, 0.0
        for a in self.owns():
            s1, a1 = self.sh.ops.sumsq_amax(a)
            ss, am = ss + s1, max(am, a1)
        if self.sh.comm.world == 1:
            return [ss, am]
        sums, maxs, _ = self.sh.comm.reduce_mixed([ss], [am])
        return [sums[0], maxs[0]]

    def _clip(self, lb, ub):
        return self._new(self.sh.ops.clip(self.loc, lb.loc, ub.loc))

    def _count_outside_box(self, lb, ub):
        v = sum(self.sh.ops.count_outside_box(a, l, u)
                for a, l, u in zip(self.owns(), lb.owns(), ub.owns()))
        return self.sh.comm.reduce_floats([v])[0]

    def _box_sphere_reduce(self, d, dscale, lb, ub):
        """The 7 quantities of device.box_sphere_reduce over the segments and the ranks: d.d,
        z.d, z.z and the count of zero-direction coordinates outside the box are sums,
        ta = max, tb = min (qp_subproblem.py:215-216)."""
        tot = [0.0, 0.0, 0.0, -np.inf, np.inf, 0.0, 0.0]
        lbs = lb.owns() if lb is not None else [None] * len(self.owns())
        ubs = ub.owns() if ub is not None else [None] * len(self.owns())
        for z, dd, l, u in zip(self.owns(), d.owns(), lbs, ubs):
            if len(z) == 0:
                continue
            r = self.sh.ops.box_sphere_reduce(z, dd, dscale, l, u)
            for i in (0, 1, 2, 5, 6):
                tot[i] += r[i]
            tot[3], tot[4] = max(tot[3], r[3]), min(tot[4], r[4])
        c = self.sh.comm
        if c.world == 1:
            return tot
        s, ta, tb = c.reduce_mixed([tot[0], tot[1], tot[2], tot[5], tot[6]], [tot[3]], [tot[4]])
        return [s[0], s[1], s[2], ta[0], tb[0], s[3], s[4]]


# --------------------------------------------------------------------------- operators
class ShardCSR:
    """A sparse matrix whose rows follow a distributed space and whose columns another (the
    Jacobian: rows = constraints, columns = variables; the barrier problem's augmented
    Jacobian: rows = inequality rows, columns = z).  The local block holds COMPLETE rows over
    the rank's extended column range, so ``dot`` is exact on every local row; ``T.dot`` is
    exact on the own entries, its halo is synchronised."""

    def __init__(self, sh, local, transposed=False, other=None, row_kind="row", col_kind="col"):
        self.sh, self.local, self.transposed = sh, local, transposed
        self._T = other
        self.row_kind, self.col_kind = row_kind, col_kind
        m, n = sh.global_len(row_kind), sh.global_len(col_kind)
        self.shape = (n, m) if transposed else (m, n)

    @staticmethod
    def from_global(sh, A, like=None):
        """The rank's block (complete rows E0..E1 over its extended columns) of a global
        matrix.  ``like``: a ShardCSR made from a matrix of the same pattern -- its local
        pattern (tiles, transpose, symbolic factorization) is reused, only values move."""
        d = sh.lay.me
        A = sps.csr_matrix(A)
        loc = sps.csr_matrix(A[d["E0"]:d["E1"], d["x0"]:d["x1"]])
        if not loc.has_sorted_indices:
            loc.sort_indices()
        sig = (loc.shape, loc.nnz, hash(loc.indptr.tobytes()), hash(loc.indices.tobytes()))
        if like is not None and getattr(like, "_sig", None) == sig:
            out = like.with_values(loc.data)
        else:
            _, _, lo, hi = sh.lay.geom("row")
            _, _, clo, chi = sh.lay.geom("col")
            out = ShardCSR(sh, sh.ops.csr(loc, row_breaks=[lo, hi], col_breaks=[clo, chi]))
        out._sig = sig
        return out

    @property
    def T(self):
        if self._T is None:
            self._T = ShardCSR(self.sh, self.local, not self.transposed, self, self.row_kind,
                               self.col_kind)
        return self._T

    def with_values(self, data):
        """Same pattern, new local values (value refresh of a Jacobian)."""
        return ShardCSR(self.sh, self.sh.ops.refresh(self.local, data), row_kind=self.row_kind,
                        col_kind=self.col_kind)

    def dot(self, x):
        sh = self.sh
        if not self.transposed:
            assert x.kind == self.col_kind, (x.kind, self.col_kind)
            return ShardVec(self.local.dot(x.loc), sh, self.row_kind)
        assert x.kind == self.row_kind, (x.kind, self.row_kind)
        return sh.sync(ShardVec(sh.ops.rmatvec(self.local, x.loc), sh, self.col_kind))

    matvec = dot

    def frobenius_norm(self):
        ops, tot, r0 = self.sh.ops, 0.0, 0
        for _, off, ln, lo, hi, _, _, _, _ in self.sh.segments(self.row_kind):
            tot += ops.frob_sq_rows(self.local, off + lo, off + hi)
        return float(np.sqrt(self.sh.comm.reduce_floats([tot])[0]))


class ShardHessian:
    """Rows ``H[X, X]`` of a banded Hessian (+ optional diagonal term) on a rank's variables:
    exact on the own entries, halo synchronised."""

    def __init__(self, sh, local, kind="col"):
        self.sh, self.local, self.kind = sh, local, kind
        self.shape = (sh.global_len(kind),) * 2

    @staticmethod
    def from_global(sh, H, hdiag=None, like=None):
        """Rows / columns X of a global banded Hessian (+ diagonal term).  ``like``: a
        ShardHessian made from a matrix of the same pattern (local pattern reused)."""
        d = sh.lay.me
        H = sps.csr_matrix(H)
        loc = sps.csr_matrix(H[d["x0"]:d["x1"], d["x0"]:d["x1"]])
        if not loc.has_sorted_indices:
            loc.sort_indices()
        _, ln, lo, hi = sh.lay.geom("col")
        sig = (loc.shape, loc.nnz, hash(loc.indptr.tobytes()), hash(loc.indices.tobytes()))
        if like is not None and getattr(like, "_sig", None) == sig:
            csr = sh.ops.refresh(like._csr, loc.data)
        else:
            csr = sh.ops.csr(loc, row_breaks=[lo, hi])
        diag = sh.ops.from_host(np.asarray(hdiag, dtype=float)[d["x0"]:d["x1"]]) \
            if hdiag is not None else None
        out = ShardHessian(sh, sh.ops.hessian(ln, csr, diag))
        out._sig, out._csr = sig, csr
        return out

    def dot(self, p):
        assert p.kind == self.kind, (p.kind, self.kind)
        return self.sh.sync(ShardVec(self.local.dot(p.loc), self.sh, self.kind))

    matvec = dot


class HostOperatorTerm:
    """A Hessian term that exists only as a HOST operator -- finite differences of the user's
    gradient (the reference's default ``hess='2-point'``, _numdiff.py:403-441), BFGS-like
    updates, any ``LinearOperator`` (_canonical_constraint.py:119-139) -- on a distributed
    vector: the vector is gathered, the operator applied on every rank (the user's callbacks
    are replicated host code), the result cut back into the ranks' blocks.  One collective and
    one host call per product: what a host operator costs on one GPU too (vectors cross to the
    host for it), times the gather."""

    def __init__(self, sh, op, kind="col"):
        self.sh, self.op, self.kind = sh, op, kind
        n = sh.global_len(kind)
        self.shape = (n, n)

    def dot(self, p):
        fn = getattr(self.op, "dot", None) or self.op.matvec
        return self.sh.from_global(np.asarray(fn(p.to_host()), dtype=float).ravel(), self.kind)

    matvec = dot


class OperatorSum:
    """Sum of distributed Hessian terms (a ``ShardHessian`` / ``GeneralHessian`` part and host
    operators): no matrix to hand to the device-resident loop, so the general driver runs."""

    def __init__(self, sh, parts, kind="col"):
        self.sh, self.parts, self.kind = sh, list(parts), kind
        self.shape = self.parts[0].shape

    def dot(self, p):
        y = self.parts[0].dot(p)
        for h in self.parts[1:]:
            y = y + h.dot(p)
        return y

    matvec = dot


def _is_host_operator(h):
    return (not sps.issparse(h) and not isinstance(h, np.ndarray)
            and (hasattr(h, "dot") or hasattr(h, "matvec")) and hasattr(h, "shape"))


class _ShardOp:
    def __init__(self, shape, fn, projector):
        self.shape, self._fn, self.projector = shape, fn, projector

    def dot(self, x):
        return self._fn(x)

    matvec = dot


class ShardProjector:
    """Z, LS, Y of the distributed Jacobian through the normal equations
    (projections.py:58-90, refinement loop :69-78, orthogonality :23-55): the local banded
    solve on the extended rows is exact on the own rows (module docstring)."""

    def __init__(self, A, orth_tol=1e-12, max_refin=3):
        self.A, self.sh = A, A.sh
        self.orth_tol, self.max_refin = orth_tol, max_refin
        plain = A.row_kind == "row" and A.col_kind == "col"
        # (``merged``: the local block is kept in the order in which A A' is banded, another one
        # than the stacked spaces of its rows -- sharded_mixed.MergedRowsCSR; the solve runs in
        # that order, ``_apply_inv`` permutes at its boundary)
        self.solver = (self.sh.ops.normal_solver(A.local) if plain or getattr(A, "merged", False)
                       else self.sh.ops.any_normal_solver(A.local))
        self.norm_A = A.frobenius_norm()
        self.stats = {"solves": 0, "refinements": 0, "cancellation_steps": 0}
        self.fused_sharded = bool(getattr(self.sh.ops, "fused", False))
        self.plain = plain
        self._check_truncation()

    def _check_truncation(self):
        """The local solve on own + halo rows stands for the global one on the own rows only
        when ``(A A')^-1`` decays across the halo (module docstring).  Every operator of this
        projector relies on it (Z, LS, Y; the fused loop and the general driver alike), so it is
        established here, once per factorization: on the device from the decoupling measured by
        the factorization itself (csrc/banded.hip: separator couplings below 2^-56), with the
        numpy twin by the residual of one probe solve on the own rows."""
        sh = self.sh
        if sh.comm.world == 1:
            return
        if self.fused_sharded:
            from . import _hip
            banded = _banded_of(self)
            geo = (ctypes.c_int32 * 2)()
            bad = banded is None or not _hip.load().ipx_banded_decoupled_geometry(
                ctypes.c_void_p(banded.handle), geo)
            # (the ranks decide together: a refusal on one rank only would leave the others
            # waiting in the next collective)
            if sh.comm.reduce_floats([1.0 if bad else 0.0])[0] > 0:
                raise NotImplementedError(
                    "row-sharded projections need a banded (A A')^-1 whose partitioned "
                    "factorization decouples numerically (DESIGN.md section 5): the local solve "
                    "truncated to own + halo rows would not be the global one")
            return
        w = sh.full(self.A.row_kind, 1.0)
        v = self._apply_inv(w)
        self.stats["solves"] -= 1
        res = self.A.dot(self.A.T.dot(v)) - w
        err = np.sqrt(res.sumsq_amax()[0] / max(len(w), 1))
        if not err <= 1e-9:
            raise NotImplementedError(
                "row-sharded projections: the local solve truncated to own + halo rows misses "
                "the global one by %.1e on the own rows (slowly decaying (A A')^-1: fewer "
                "ranks or a wider halo)" % err)

    def _apply_inv(self, w):
        self.stats["solves"] += 1
        A = self.A
        if getattr(A, "merged", False):
            loc = A.to_stacked(self.solver.solve(A.to_banded(w.loc)))
        else:
            loc = self.solver.solve(w.loc)
        return self.sh.sync(ShardVec(loc, self.sh, A.row_kind))

    def orthogonality(self, z):
        norm_z = np.sqrt(z.sumsq_amax()[0])
        if norm_z == 0 or self.norm_A == 0:
            return 0.0, None
        Az = self.A.dot(z)
        return float(np.sqrt(Az.sumsq_amax()[0]) / (self.norm_A * norm_z)), Az

    def null_space(self, x):
        v = self._apply_inv(self.A.dot(x))
        z = x - self.A.T.dot(v)
        k = 0
        while True:
            orth, Az = self.orthogonality(z)
            # (one correction step when the subtraction above cancelled most of x -- late
            # barrier subproblems --, as projector.NormalEquationProjector.null_space)
            cancelled = False
            if k == 0 and Az is not None:
                from .projector import NormalEquationProjector as _NEP
                cancelled = z.sumsq_amax()[0] < (_NEP.CANCELLATION ** 2) * x.sumsq_amax()[0]
            if k >= self.max_refin or not (orth > self.orth_tol or cancelled):
                break
            z = z - self.A.T.dot(self._apply_inv(Az))
            k += 1
            self.stats["refinements" if orth > self.orth_tol else "cancellation_steps"] += 1
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
    """Sharded counterpart of ``projector.projections`` for a ``ShardCSR`` Jacobian."""
    if method not in (None, "NormalEquation", "AugmentedSystem"):
        raise ValueError("Method not allowed for sparse matrix.")
    return ShardProjector(A, orth_tol, max_refin).operators()


# --------------------------------------------------------------------------- fused loop
# counters over the life of the process (tests assert the device-resident path was taken)
STATS = {"fused_calls": 0, "iterations": 0, "batches": 0, "box_events": 0, "refine_events": 0,
         "resident_batches": 0,       # batches that ran as ONE resident launch per rank
         "resident_halo_syncs": 0}    # ... and the halo synchronisations the host did after them

ST_RTG0, ST_RTG1, ST_TOL, ST_RADIUS, ST_ALPHA, ST_STOP, ST_NITER, ST_BETA = range(8)
ST_PTHP, ST_ORTH_RHS, ST_XNORM2, ST_VIOL, ST_ORTH, ST_IT_DONE = 8, 9, 10, 11, 12, 13
_TINY = 1e-25


class Shard2Ext(ctypes.Structure):
    """Mirror of ipx_shard2_ext (include/ipx.h)."""
    _A4 = ctypes.c_int64 * 4
    _fields_ = [("s1", ctypes.c_void_p), ("pack", ctypes.c_void_p), ("nseg", ctypes.c_int64),
                ("own_lo", _A4), ("own_hi", _A4), ("p1_lo", _A4), ("p1_hi", _A4),
                ("p3_lo", _A4), ("p3_hi", _A4), ("p2_lo", ctypes.c_int64),
                ("p2_hi", ctypes.c_int64), ("p4_lo", ctypes.c_int64), ("p4_hi", ctypes.c_int64),
                ("peer", ctypes.c_void_p), ("seg_lo", _A4), ("seg_hi", _A4),
                ("send_left", _A4), ("send_right", _A4), ("fuse_comm", ctypes.c_int64),
                ("res_wg0", ctypes.c_int64), ("res_nwg", ctypes.c_int64),
                ("res_gwg0", ctypes.c_int64), ("res_gnwg", ctypes.c_int64)]


def _banded_of(P):
    """The banded factorization whose workgroups follow the partition of the "row" space:
    the solver itself (x-space problems) or the Schur complement of the nonlinear rows (the
    barrier problem: box rows eliminated analytically), or None."""
    from .projector import BandedNormalSolver
    from .boxschur import BoxSchurNormalSolver
    sv = P.solver
    if P.plain or getattr(P.A, "merged", False):
        return sv if isinstance(sv, BandedNormalSolver) and sv.perm is None else None
    if P.A.row_kind != ShardedBackend.INEQ or P.A.col_kind != ShardedBackend.Z:
        return None
    if not isinstance(sv, BoxSchurNormalSolver) or sv.c_args() is None:
        return None
    # the rows left to the banded solve must be exactly the nonlinear rows, in order
    mE = P.sh.lay.geom("row")[1]
    if len(sv.an.general) != mE or not np.array_equal(sv.an.general, np.arange(mE)):
        return None
    return sv.inner


def fused_supports(H, Z, Y):
    P = getattr(Z, "projector", None)
    if not isinstance(P, ShardProjector) or getattr(Y, "projector", None) is not P:
        return False
    if not isinstance(H, ShardHessian) or not P.fused_sharded or H.kind != P.A.col_kind:
        return False
    from . import cg_fused
    return _banded_of(P) is not None and cg_fused._hessian_parts(H.local) is not None \
        and _loop_geometry_ok(P)


def _loop_geometry_ok(P):
    """The device-resident loop sums per-workgroup partials of the banded solve over the rank's
    OWN rows, so the solve's workgroups must follow the layout's row blocks (260 rows for a
    tridiagonal A A'; a wider band gives the factorization another chunk size).  Where they do
    not, the general driver runs on the same distributed vectors instead (``qp.projected_cg``)."""
    ok = getattr(P, "_loop_geometry", None)
    if ok is None:
        from . import _hip
        sh = P.sh
        banded = _banded_of(P)
        geo = (ctypes.c_int32 * 2)()
        _, _, rlo, rhi = sh.lay.geom("row")
        dec = _hip.load().ipx_banded_decoupled_geometry(ctypes.c_void_p(banded.handle), geo)
        ok = bool(dec) and geo[0] == sh.lay.row_block and rlo % geo[0] == 0 and \
            (rhi % geo[0] == 0 or sh.lay.me["R1"] == sh.lay.m)
        flags = sh.comm.reduce_floats([0.0 if ok else 1.0])      # the ranks decide together
        ok = P._loop_geometry = flags[0] == 0.0
    return ok


class FusedShardedCG:
    """Device-resident projected CG on one rank's extended local problem: the kernels and
    the argument block of the single-GPU loop (``cg_fused._Loop``), driven segment by
    segment between the two all-reduces and the halo exchange of an iteration."""

    def __init__(self, H, P, lb, ub, transport=None):
        """``transport``: None = the group's peer mailboxes when it has them ("ipc"), else
        ``torch.distributed``; "dist" forces the latter (A/B measurements)."""
        from . import _hip, cg_fused
        from . import device as dv
        from .projector import NormalEquationProjector
        self._hip, self.dv = _hip, dv
        self.lib = _hip.load()
        sh = self.sh = P.sh
        A_loc = P.A.local
        # the local problem as the single-GPU loop sees it
        local_P = getattr(P, "_local_projector", None)
        if local_P is None:
            local_P = P._local_projector = NormalEquationProjector.__new__(NormalEquationProjector)
            local_P.A, local_P.solver = A_loc, P.solver
            local_P.orth_tol, local_P.max_refin = P.orth_tol, P.max_refin
            local_P.m, local_P.n = A_loc.shape
            local_P.norm_A = P.norm_A
            local_P.stats = P.stats
            local_P.row_perm = None
        self.P = P
        self.kind = H.kind
        # (the rank-local loop keeps the separate launches: the own-range partial sums of
        # ipx_cg_shard2_segment follow their workgroups / row tiles)
        # (resident=None: the tables of the resident kernel are built when the local problem
        # qualifies; whether the GROUP runs it is decided below)
        self.L = L = cg_fused._Loop(H.local, local_P, lb.loc if lb is not None else None,
                                    ub.loc if ub is not None else None,
                                    resident=None if transport != "dist" else False)
        L.args.resident = 0          # (the single-GPU launch never runs on a rank's local problem)
        a = L.args
        dev = dv.ctx().device
        self.s1 = torch.zeros(2, dtype=torch.float64, device=dev)
        self.pack = torch.zeros(4, dtype=torch.float64, device=dev)
        e = self.ext = Shard2Ext()
        e.s1, e.pack = self.s1.data_ptr(), self.pack.data_ptr()
        segs = sh.segments(self.kind)
        _, _, rlo, rhi = sh.lay.geom("row")
        e.nseg = len(segs)
        Hc, _ = cg_fused._hessian_parts(H.local)
        for k, (_, off, _, lo, hi, *_rest) in enumerate(segs):
            e.own_lo[k], e.own_hi[k] = off + lo, off + hi
            e.p1_lo[k], e.p1_hi[k] = Hc.pattern.tile_range(off + lo, off + hi)
        if a.A_span:                       # step1 inside A.r: partials per row tile of A
            if getattr(L, "own_tiles", None) is not A_loc.pattern.tiles:
                raise _hip.IpxError("sharded loop: the fused step1 must use the pattern's row tiles")
            e.p2_lo, e.p2_hi = A_loc.pattern.tile_range(rlo, rhi)
        else:                              # vector kernel, masked to the own entries
            e.p2_lo, e.p2_hi = 0, int(a.vec_grid)
        geo = (ctypes.c_int32 * 2)()
        banded = _banded_of(P)
        decoupled = self.lib.ipx_banded_decoupled_geometry(ctypes.c_void_p(banded.handle), geo)
        if not decoupled or geo[0] != sh.lay.row_block or rlo % geo[0] or \
                (rhi % geo[0] and sh.lay.me["R1"] != sh.lay.m):
            raise NotImplementedError(
                "row-sharded projected CG needs the numerically decoupled single-launch banded "
                "solve with %d rows per workgroup (DESIGN.md section 5); this factorization "
                "reports decoupled=%d, rows per workgroup %d" % (sh.lay.row_block, decoupled,
                                                                  geo[0]))
        w0, w1 = rlo // geo[0], (rhi + geo[0] - 1) // geo[0]
        e.p4_lo, e.p4_hi = w0, w1
        if a.At_qv:                        # g = r - A'v rides in the solve: partials per workgroup
            e.p3_lo[0], e.p3_hi[0] = w0, w1
        else:
            Atp = A_loc.T.pattern
            for k in range(len(segs)):
                e.p3_lo[k], e.p3_hi[k] = Atp.tile_range(e.own_lo[k], e.own_hi[k])
        for k, (_, off, ln, lo, hi, sl, sr, _, _) in enumerate(segs):
            e.seg_lo[k], e.seg_hi[k] = off, off + ln
            e.send_left[k], e.send_right[k] = sl, sr
        self.mailbox = sh.mailbox() if transport != "dist" else None
        self.resident, self.resident_dirty = False, False
        if self.mailbox is not None:
            # the loop's kernels all-reduce the scalars and move the halo of g themselves: in
            # the prologues of the kernels that consume them (3 launches per iteration; the C
            # side decides per argument block -- one segment, no box, 16-bit index forms) or
            # in pack kernels of their own (5 launches; IPX_DEBUG_FORMS=pack-comm forces that)
            e.peer = self.mailbox.handle
            e.fuse_comm = self._agree_on_fused_comm(Hc, A_loc, lb is not None)
            self._exchange_g = None
            self.resident = self._agree_on_resident(Hc, A_loc, w0, w1)
        else:
            self._exchange_g = sh.comm.prepare_exchange_many(
                [(self.L.r[off:off + ln], lo, hi, sl, sr)
                 for _, off, ln, lo, hi, sl, sr, _, _ in segs])
        # (own range and send counts of the first segment: bench.py measures the exchange's floor)
        self.col_geom = (segs[0][3], segs[0][4], segs[0][5], segs[0][6])

    def _agree_on_fused_comm(self, Hc, A_loc, has_box):
        """1 when EVERY rank's argument block allows the collectives in the prologues of the
        loop's own kernels (csrc/cg.hip ipx_cg_shard2_fusable: 16-bit index tables, the solve's
        fused tail -- functions of each rank's own slice of A and H), else 0 on every rank: the
        two forms order an iteration's collectives differently and share sequence numbers and
        slots, so a group in which one rank took the pack kernels and another the prologues
        would read each other's words as the wrong quantity.  One all-reduce (MIN)
        per pair of patterns, cached on the sharding (the tables are symbolic)."""
        sh = self.sh
        asked = 0 if self._hip.debug_form("pack-comm") else 1
        key = (id(Hc.pattern), Hc.pattern.nnz, id(A_loc.pattern), A_loc.pattern.nnz, has_box, asked)
        cache = sh.__dict__.setdefault("_fuse_comm_agreed", {})
        if key not in cache:
            mine = asked and int(self.lib.ipx_cg_shard2_fusable(self.L.ref(), ctypes.byref(self.ext)))
            # every workgroup of the prologue kernels spins until its peers' words arrive.  Ranks
            # on their own GPUs cannot be in each other's way; ranks that SHARE a device (the
            # one-GPU rehearsals) can: the workgroups of one rank's next kernel take the slots
            # the other rank's last workgroups are waiting for and never give them back --
            # measured on config 5 at full size, two ranks on one MI355X: ~890 row tiles of H per
            # rank deadlock after a few hundred iterations (10 s timeout, fall-back), ~620 run
            # 50 000 iterations.  So a group that shares a device takes the pack kernels (small
            # grids) when its row tiles together exceed five per compute unit.
            cus = ctypes.c_int(0)
            self._hip.call("ipx_device_info", ctypes.byref(cus), None, None, 0)
            info = [None] * sh.comm.world
            dist.all_gather_object(info, (int(mine), _device_id(), int(Hc.pattern.ntiles),
                                          int(cus.value)), group=sh.comm.group)
            ok = all(i[0] for i in info)
            for dev in {i[1] for i in info}:
                on = [i for i in info if i[1] == dev]
                if len(on) > 1:
                    ok = ok and sum(i[2] for i in on) <= 5 * min(i[3] for i in on)
            cache[key] = int(ok)
            # (the patterns are kept alive with the decision: their ids stay theirs)
            cache[key, "keep"] = (Hc.pattern, A_loc.pattern)
        return cache[key]

    def _agree_on_resident(self, Hc, A_loc, w0, w1):
        """True when the GROUP runs a batch as one resident launch per rank (csrc/resident.hip,
        PEER form): every rank's local problem has the kernel's tables and fits its budgets, all
        own blocks of all ranks together are within the kernel's record arrays, the hand-off
        buffers are mapped.  The ranks exchange (tables?, own blocks, longest halo) once per pair
        of patterns -- the halo capacity and every rank's first global workgroup follow from
        it -- and take the minimum of their verdicts."""
        sh, L, e = self.sh, self.L, self.ext
        self.resident_dirty = False
        asked = not self._hip.debug_form("no-resident") and e.nseg == 1
        key = ("res", id(Hc.pattern), Hc.pattern.nnz, id(A_loc.pattern), A_loc.pattern.nnz,
               getattr(L, "pcr_L", None), asked)
        cache = sh.__dict__.setdefault("_resident_agreed", {})
        if key not in cache:
            have = asked and getattr(L, "proj_tabs", None) is not None
            if have:
                # the kernel writes the variables its workgroups own (cg_fused.fuse_vown); they
                # must be the layout's own range, which the halo synchronisation starts from
                vown = L.vown.cpu().numpy()
                have = int(vown[w0]) == int(e.own_lo[0]) and int(vown[w1]) == int(e.own_hi[0])
            cus = ctypes.c_int(0)
            self._hip.call("ipx_device_info", ctypes.byref(cus), None, None, 0)
            mine = (bool(have), int(w1 - w0), int(L.args.R_hw) if have else 0, _device_id(),
                    int(cus.value))
            info = [None] * sh.comm.world
            dist.all_gather_object(info, mine, group=sh.comm.group)
            ok = all(i[0] for i in info) and \
                sum(i[1] for i in info) <= int(self.lib.ipx_cg_resident_max_global())
            # every workgroup of every rank must be running at once, one per compute unit: ranks
            # that share a device (the one-GPU rehearsal of the tests) share its compute units
            for dev in {i[3] for i in info}:
                on = [i for i in info if i[3] == dev]
                ok = ok and sum(i[1] for i in on) <= min(i[4] for i in on)
            ok = ok and bool(self.mailbox.attach_resident())      # (collective: every rank or none)
            verdict = None
            if ok:
                verdict = dict(hw=max(i[2] for i in info), gwg0=sum(i[1] for i in info[:sh.comm.rank]),
                               gnwg=sum(i[1] for i in info))
            cache[key] = verdict
            cache[key, "keep"] = (Hc.pattern, A_loc.pattern)
            if verdict is not None:
                self._set_resident(verdict, w0, w1)
                mine_ok = int(self.lib.ipx_cg_shard2_resident_ok(L.ref(), ctypes.byref(e)))
                if int(sh.comm.reduce_floats([float(mine_ok)], op="min")[0]) == 0:
                    cache[key] = None
        verdict = cache[key]
        if verdict is None:
            e.res_nwg = 0
            return False
        self._set_resident(verdict, w0, w1)
        return True

    def _set_resident(self, verdict, w0, w1):
        e = self.ext
        self.L.args.R_hw = verdict["hw"]
        e.res_wg0, e.res_nwg = int(w0), int(w1 - w0)
        e.res_gwg0, e.res_gnwg = verdict["gwg0"], verdict["gnwg"]

    def sync_halos(self):
        """After resident batches the local x, p, r, Hp hold this rank's OWN entries only:
        overwrite their halo copies with the owners' values (one batched neighbour exchange) and
        re-derive what the separate launches keep next to p.  Called before the host touches the
        vectors: an event, the end of the loop."""
        if not getattr(self, "resident_dirty", False):
            return
        self.resident_dirty = False
        STATS["resident_halo_syncs"] += 1
        L = self.L
        segs = self.sh.segments(self.kind)
        self.sh.comm.exchange_many([(t[off:off + ln], lo, hi, sl, sr)
                                    for t in (L.x, L.p, L.r, L.Hp)
                                    for _, off, ln, lo, hi, sl, sr, _, _ in segs])
        self._hip.call("ipx_cg_save_pb", L.ref(), self.dv.stream_ptr())

    def _segment(self, phase, it, mode=0):
        self._hip.call("ipx_cg_shard2_segment", self.L.ref(), ctypes.byref(self.ext), int(phase),
                       int(it), int(mode), self.dv.stream_ptr())

    def prime(self, x0, r0, g0, rt_g, tol, trust_radius):
        L, n = self.L, self.L.n
        st = self.dv.stream_ptr()
        L.x.copy_(x0.loc.t)
        L.r.copy_(r0.loc.t)
        self._hip.call("ipx_axpby", n, -1.0, self.dv._p(g0.loc.t), 0.0, None, self.dv._p(L.p), st)
        key = (rt_g, tol, trust_radius)
        if getattr(self, "_init_key", None) != key:      # (a restart re-uses the device copy)
            init = np.zeros(L.state.numel())
            init[ST_RTG0], init[ST_TOL], init[ST_RADIUS] = rt_g, tol, trust_radius
            init[ST_ORTH_RHS] = self.P.orth_tol * self.P.norm_A
            self._init_dev = torch.from_numpy(init).to(L.state.device)
            self._init_key = key
        L.state.copy_(self._init_dev)
        self._hip.check(self.lib.ipx_cg_hp(L.ref(), st), "ipx_cg_hp")
        self._hip.call("ipx_cg_shard2_fold_hp", L.ref(), ctypes.byref(self.ext), st)

    def iterate(self, it_begin, it_end):
        """Enqueue iterations [it_begin, it_end): per iteration two all-reduces and one halo
        exchange, no host synchronisation.  On the peer mailboxes this is ONE C call for the
        whole batch (the reductions and the exchange happen inside the loop's launches);
        through ``torch.distributed`` the host issues three collectives per iteration."""
        comm, exchange_g = self.sh.comm, self._exchange_g
        if self.mailbox is not None:
            if self.resident:
                # ONE launch per rank for the whole batch; the workgroups of all ranks hand their
                # scalars and halos to each other (two hops per iteration)
                self._hip.call("ipx_cg_shard2_resident", self.L.ref(), ctypes.byref(self.ext),
                               int(it_begin), int(it_end), self.dv.stream_ptr())
                self.resident_dirty = True
                STATS["resident_batches"] += 1
            else:
                self._hip.call("ipx_cg_shard2_iterate", self.L.ref(), ctypes.byref(self.ext),
                               int(it_begin), int(it_end), self.dv.stream_ptr())
            comm.stats["ipc_batches"] += 1
            comm.stats["ipc_iterations"] += it_end - it_begin
            return
        for it in range(it_begin, it_end):
            comm.all_reduce(self.s1)                                   # p'Hp
            self._segment(0, it)
            comm.all_reduce(self.pack)          # ||x+ap||^2, #violations, ||g||^2, ||A g||^2
            exchange_g()                                               # halo of g
            self._segment(1, it, 0)

    def resume(self, it, mode):
        """Finish iteration ``it`` after the host handled a box / refinement event."""
        self.sync_halos()
        self.L.state[ST_STOP] = 0.0
        self._segment(1, it, mode)
        from . import device as dv
        return dv.read_doubles(self.L.state, self.L.state.numel())


def fused_projected_cg(H, c, Z, Y, b, trust_radius=np.inf, lb=None, ub=None, tol=None,
                       max_iter=None, max_infeasible_iter=None, batch=None):
    """qp_subproblem.py:416-643 on distributed vectors with the device-resident loop; the
    rare events (trust-region exit, negative curvature, box-infeasible iterate, projection
    refinement) are finished on the host with the distributed forms of the reference's
    helper routines, exactly as ``cg_fused.projected_cg`` does on one GPU."""
    from . import qp
    from . import device as dv
    P = Z.projector
    sh = P.sh
    n, m = len(c), len(b)
    has_box = lb is not None or ub is not None
    if has_box:
        lb = lb if lb is not None else c.full_like(-np.inf)
        ub = ub if ub is not None else c.full_like(np.inf)

    x0 = Y.dot(-b)                                       # :502-512
    r0 = Z.dot(H.dot(x0) + c)
    g0 = Z.dot(r0)
    rt_g = g0.sumsq_amax()[0]
    tr_distance = trust_radius - dv.norm(x0)
    if tr_distance < 0:
        raise ValueError("Trust region problem does not have a solution.")
    if tr_distance < _TINY:
        return x0, {'niter': 0, 'stop_cond': 2, 'hits_boundary': True}
    if tol is None:
        tol = max(min(0.01 * np.sqrt(rt_g), 0.1 * rt_g), _TINY)
    if max_iter is None:
        max_iter = n - m
    max_iter = min(max_iter, n - m)
    if max_infeasible_iter is None:
        max_infeasible_iter = n - m

    return _run_fused(H, P, c, x0, r0, g0, rt_g, tol, trust_radius, lb, ub, has_box, max_iter,
                      max_infeasible_iter, batch, retry=True)


def _run_fused(H, P, c, x0, r0, g0, rt_g, tol, trust_radius, lb, ub, has_box, max_iter,
               max_infeasible_iter, batch, retry):
    from . import _hip
    from . import device as dv
    sh = P.sh
    F = FusedShardedCG(H, P, lb if has_box else None, ub if has_box else None)
    try:
        return _drive_fused(F, c, x0, r0, g0, rt_g, tol, trust_radius, lb, ub, max_iter,
                            max_infeasible_iter, batch)
    except _hip.IpxError as exc:
        # a wait on the peer mailboxes timed out (stop code 7): every rank's waits time out
        # together, so every rank arrives here.  The mailbox words are in an unknown state:
        # the group gives the transport up for good and solves this subproblem again through
        # torch.distributed (nothing of the call's inputs has been overwritten)
        if not (retry and F.mailbox is not None and "timed out" in str(exc)):
            raise
        from warnings import warn
        warn("row-sharded projected CG: %s -- falling back to the torch.distributed transport "
             "for the rest of this process" % exc)
        os.environ["IPX_SHARD_TRANSPORT"] = "dist"
        sh._mailbox = False
        sh.comm.mbox = None              # (the outer loops' collectives leave it as well)
        return _run_fused(H, P, c, x0, r0, g0, rt_g, tol, trust_radius, lb, ub, has_box, max_iter,
                          max_infeasible_iter, batch, retry=False)


def _drive_fused(F, c, x0, r0, g0, rt_g, tol, trust_radius, lb, ub, max_iter,
                 max_infeasible_iter, batch):
    from . import device as dv
    sh = F.sh
    F.prime(x0, r0, g0, rt_g, tol, trust_radius)
    L = F.L
    DV = dv.DVec

    class Driver:
        """The sharded loop behind ``cg_fused.run_device_loop`` (the same event handling as on
        one GPU, over distributed vectors)."""
        first_batch, batch_cap = 2, 64

        def iterate(self, it, end):
            self.end = end
            F.iterate(it, end)

        def read_state(self):
            # (the read that ends the loop -- out of iterations on a continuing state -- is
            # agreed on here, before the host's first collective on the result: the halo
            # synchronisation after resident batches)
            return agreed(dv.read_doubles(L.state, L.state.numel()), final=self.end >= max_iter)

        def X(self):
            F.sync_halos()
            return ShardVec(DV(L.x), sh, F.kind)

        def Pv(self):
            F.sync_halos()
            return ShardVec(DV(L.p), sh, F.kind)

        def set_x(self, v):
            L.x = v.loc.t

        def zeros(self):
            return c.zeros_like()

        def resume(self, it_stop, mode):
            return agreed(F.resume(it_stop, mode))

        def refine(self, it_stop):
            _refine_sharded(F)

    last = {"stop": 0, "final": False}

    def agreed(state, final=False):
        """Before the host ACTS on a state block read on the mailbox transport -- an event to
        handle with torch.distributed collectives, or the end of the subproblem -- the ranks
        establish together whether any of them saw a wait time out (stop code 7; ADVICE r3:
        the last communicating launch before a read can time out on one rank and complete on
        a slower one, which would then pair its next collectives with the other's restarted
        subproblem).  One all-reduce (MAX) per non-continuing read; a rank that merely
        continues (stop 0) does not take part: its next batch finds no partner, times out and
        joins the agreement its peers are waiting in."""
        last["stop"] = stop = int(state[ST_STOP])
        if F.mailbox is not None and (stop != 0 or final):
            last["final"] = last["final"] or stop == 0
            sh.comm.stats["agreements"] = sh.comm.stats.get("agreements", 0) + 1
            if sh.comm.reduce_floats([1.0 if stop == 7 else 0.0], op="max")[0] > 0.0:
                state = list(state)
                state[ST_STOP] = 7.0
        return state

    from . import cg_fused
    x, niter, stop_cond, hits_boundary = cg_fused.run_device_loop(
        Driver(), STATS, lb, ub, trust_radius, max_iter, max_infeasible_iter, batch)
    if last["stop"] == 0 and not last["final"]:
        # the loop ran out of iterations on a continuing state without a read (max_iter 0):
        # this rank has not taken part in an agreement its peers may be waiting in (they cannot
        # have seen anything but 0 or 7 in the same read)
        if int(agreed([0.0] * ST_STOP + [-1.0])[ST_STOP]) == 7:
            raise F._hip.IpxError("sharded projected CG: a wait on the peer mailboxes timed out "
                                  "on another rank")
    STATS["fused_calls"] += 1
    STATS["iterations"] += niter
    return x, {'niter': niter, 'stop_cond': stop_cond, 'hits_boundary': hits_boundary}


def _refine_sharded(F):
    """Iterative refinement of g = Z r (projections.py:69-78) on the buffers of the fused
    loop (L.r holds g, own + synchronised halo); the refined ||g||^2 replaces the packed
    value step2 derives beta from."""
    P, sh, L = F.P, F.sh, F.L
    F.sync_halos()
    g = ShardVec(F.dv.DVec(L.r), sh, F.kind)
    k = 0
    while k < P.max_refin:
        orth, Az = P.orthogonality(g)
        if k > 0 and not orth > P.orth_tol:
            break
        g = g - P.A.T.dot(P._apply_inv(Az))
        k += 1
        P.stats["refinements"] += 1
    L.r.copy_(g.loc.t)
    F.pack[2] = g.sumsq_amax()[0]


# --------------------------------------------------------------------------- outer loops
class _DiagOp:
    def __init__(self, d):
        self.d = d

    def dot(self, x):
        return self.d * x


class BoxInequalityJacobian:
    """Jacobian of the canonical inequality rows ``[c_nl(x); lb - x; x - ub] <= 0``
    (_canonical_constraint.py:169-360: the nonlinear rows, then all lower bounds, then all
    upper bounds): the distributed nonlinear block; the box blocks -I / +I stay symbolic."""

    def __init__(self, J_nl):
        self.J_nl = J_nl


class _ShardPack:
    """Scalar pack over distributed vectors: the local parts of every quantity are enqueued
    into ONE local pack (one blocking read), then summed / maximised over the ranks in ONE
    collective -- instead of a read and a collective per norm or dot product."""

    def __init__(self, sh):
        self.sh, self.loc = sh, sh.ops.pack()
        self.items = []            # per handle: ("sum" | "max" | "zero", local handles, sqrt?)

    def dot(self, a, b):
        if not len(a):
            self.items.append(("zero", [], False))
        else:
            self.items.append(("sum", [self.loc.dot(x, y) for x, y in zip(a.owns(), b.owns())
                                       if len(x)], False))
        return len(self.items) - 1

    def norm(self, v):
        if not len(v):
            self.items.append(("zero", [], False))
        else:
            self.items.append(("sum", [self.loc.sumsq(x) for x in v.owns() if len(x)], True))
        return len(self.items) - 1

    def norm_inf(self, v):
        if not len(v):
            self.items.append(("zero", [], False))
        else:
            self.items.append(("max", [self.loc.norm_inf(x) for x in v.owns() if len(x)], False))
        return len(self.items) - 1

    def read(self):
        vals = self.loc.read()
        sums, maxs = [], []
        for kind, hs, _ in self.items:
            if kind == "sum":
                sums.append(sum(vals[h] for h in hs))
            elif kind == "max":
                maxs.append(max([vals[h] for h in hs] + [0.0]))
        if self.sh.comm.world > 1 and (sums or maxs):
            sums, maxs, _ = self.sh.comm.reduce_mixed(sums, maxs)
        out, si, mi = [], 0, 0
        for kind, _, root in self.items:
            if kind == "sum":
                out.append(float(np.sqrt(sums[si])) if root else sums[si])
                si += 1
            elif kind == "max":
                out.append(maxs[mi])
                mi += 1
            else:
                out.append(0.0)
        return out


class ShardedBackend:
    """The vector / matrix / operator factory ``sqp.py`` and ``barrier.py`` are written
    against (cf. backend_hip), for one sharded problem: every vector is a ``ShardVec``, the
    Jacobian a ``ShardCSR``, the Hessian a ``ShardHessian``; scalars (norms, dot products)
    are all-reduced.  Equality-constrained problems (BASELINE configs 3 / 4: both outer
    methods) and nonlinear inequalities + a box on every variable (config 5: barrier method,
    z = [x; s_nl; s_lb; s_ub] with every segment partitioned like the space it belongs to)."""
    name = "sharded"
    INEQ = ("row", "col", "col")
    Z = ("col", "row", "col", "col")

    def __init__(self, sh):
        self.sh = sh
        self._like = {}              # last distributed matrix per role (pattern reuse)
        self._masks = {}

    def _z_breaks(self):
        """Own / halo boundaries of the z segments in the local z vector."""
        return [off + b for _, off, _, lo, hi, *_ in self.sh.segments(self.Z) for b in (lo, hi)]

    def _kind(self, n, space=None):
        """The distributed space of a vector of global length n; ``space`` is the caller's name
        for it ("x", "eq", "ineq", "z") where the backend keeps spaces a length cannot tell
        apart (the general partition), else the length decides."""
        return self.sh.kind_of_len(n)

    def asvec(self, a, space=None):
        if isinstance(a, (ShardVec, _Empty)):
            return a
        a = np.asarray(a, dtype=float)
        if a.size == 0:
            return _Empty()
        return self.sh.from_global(a, self._kind(len(a), space))

    def tohost(self, v):
        if isinstance(v, _Empty):
            return np.zeros(0)
        return v.to_host() if isinstance(v, ShardVec) else np.asarray(v)

    def zeros(self, n):
        return _Empty() if n == 0 else self.sh.zeros(self.sh.kind_of_len(n))

    def full(self, n, value, space=None):
        return self.sh.full(self._kind(n, space), value)

    def copy(self, v):
        return v.copy()

    def pack(self):
        return _ShardPack(self.sh)

    def hstack(self, parts):
        parts = [p for p in parts if len(p)]
        if len(parts) == 1:
            return parts[0]
        kinds = sum((_kinds(p.kind) for p in parts), ())
        return ShardVec(self.sh.ops.concat([p.loc for p in parts]), self.sh, kinds)

    def norm(self, v):
        return float(np.sqrt(v.sumsq_amax()[0])) if len(v) else 0.0

    def norm_inf(self, v):
        return v.sumsq_amax()[1] if len(v) else 0.0

    def dot(self, a, b):
        return a.dot(b)

    def maximum(self, v, c):
        return v._new(self.sh.ops.maximum(v.loc, c))

    def where_positive(self, v, a, c):
        return v._new(self.sh.ops.where_positive(v.loc, a.loc, c))

    def sum_log(self, s):
        """sum(log s_i), -inf when any s_i <= 0 (tr_interior_point.py:93-95)."""
        if not len(s):
            return 0.0
        tot, bad = 0.0, 0.0
        for part in s.owns():
            t, b = self.sh.ops.sum_log(part)
            tot, bad = tot + t, bad + b
        tot, bad = self.sh.comm.reduce_floats([tot, bad])
        return -np.inf if bad > 0 else tot

    def assign_negated_where(self, s, mask, c):
        """s[mask] = -c[mask] in place through the view of z (tr_interior_point.py:92):
        elementwise on own and halo entries alike."""
        mask = np.asarray(mask)
        if not mask.any():
            return
        key = (s.kind, mask.tobytes())
        m = self._masks.get(key)
        if m is None:
            m = self._masks[key] = self.sh.from_global(mask.astype(float), s.kind)
        self.sh.ops.assign_negated_where(s.loc, m.loc, c.loc)

    def diagonal_operator(self, d):
        return _DiagOp(d)

    def matrix(self, J, key="jac"):
        """A global host matrix (scipy sparse: what the reference's constraint classes hand
        to the solver, replicated on every rank) becomes the rank's block of it."""
        if J is None or isinstance(J, (ShardCSR, BoxInequalityJacobian)):
            return J
        if not sps.issparse(J):
            raise NotImplementedError("sharded backend: dense Jacobians are not distributed "
                                      "(sparse_jacobian=True)")
        A = self._like[key] = ShardCSR.from_global(self.sh, J, like=self._like.get(key))
        return A

    def _box_inequality(self, J_ineq, n_vars):
        """The canonical inequality Jacobian of `nonlinear rows + a box on every variable`
        (_canonical_constraint.py:350-355: all lower bounds, then all upper bounds) as a
        BoxInequalityJacobian; anything else is refused."""
        J = sps.csr_matrix(J_ineq)
        m = J.shape[0] - 2 * n_vars
        sig = (J.shape, J.nnz, hash(J.indptr.tobytes()), hash(J.indices.tobytes()))
        if self._like.get("box_sig") != sig:
            ok = m == self.sh.lay.m and J.shape[1] == n_vars
            if ok:
                eye = sps.identity(n_vars, format="csr")
                ok = (J[m:] != sps.vstack([-eye, eye], format="csr")).nnz == 0
            if not ok:
                raise NotImplementedError(
                    "sharded backend: inequality constraints must be nonlinear rows followed by "
                    "an interval BoxConstraint on every variable (BASELINE config 5); got a "
                    "%d x %d inequality Jacobian for %d partitioned rows"
                    % (J.shape[0], J.shape[1], self.sh.lay.m))
            self._like["box_sig"] = sig
        return BoxInequalityJacobian(self.matrix(J[:m], "jac_nl"))

    def mark_constant(self, A):
        return A

    def augmented_jacobian(self, J_eq, J_ineq, s, n_vars, n_eq, n_ineq):
        """[[J_eq, 0], [J_ineq, diag(s)]] (tr_interior_point.py:141-194) as a distributed
        matrix from the inequality rows' space to z-space."""
        if n_eq:
            raise NotImplementedError("sharded backend: equality rows next to inequality rows")
        if not isinstance(J_ineq, BoxInequalityJacobian):
            J_ineq = self._box_inequality(J_ineq, n_vars)
        sh = self.sh
        segs = sh.segments(self.INEQ)
        parts = [s.loc[off:off + ln] for _, off, ln, *_ in segs]
        local = sh.ops.augmented_box(J_ineq.J_nl.local, *parts, col_breaks=self._z_breaks())
        return ShardCSR(sh, local, row_kind=self.INEQ, col_kind=self.Z)

    def _host_hessian(self, terms):
        """Host Hessian terms (canonical.HessianSum: scipy sparse matrices, replicated on every
        rank) as a ShardHessian: diagonal-pattern terms form the diagonal part, the others are
        summed into the CSR part -- the split backend_hip.hessian_operator makes, so the local
        products round the same way as on one GPU."""
        from .canonical import HessianSum
        flat = terms.flat_terms() if isinstance(terms, HessianSum) else list(terms)
        n = self.sh.lay.n
        csr, diag, host_ops = None, None, []
        for h in flat:
            if isinstance(h, np.ndarray) and h.ndim == 2:
                h = sps.csr_matrix(h)
            if _is_host_operator(h):
                host_ops.append(HostOperatorTerm(self.sh, h))
                continue
            if not sps.issparse(h):
                raise NotImplementedError("sharded backend: a Hessian term is neither a sparse "
                                          "matrix nor an operator")
            h = sps.csr_matrix(h)
            if not h.has_canonical_format:
                h = h.copy()
                h.sum_duplicates()
            rows = np.repeat(np.arange(h.shape[0], dtype=np.int32), np.diff(h.indptr))
            if np.array_equal(rows, h.indices):          # purely diagonal pattern
                d = h.diagonal() if h.nnz < h.shape[0] else h.data
                diag = d if diag is None else diag + d
            else:
                csr = h if csr is None else csr + h
        if host_ops and csr is None and diag is None:
            return host_ops[0] if len(host_ops) == 1 else OperatorSum(self.sh, host_ops)
        if csr is None:
            csr = sps.csr_matrix((n, n))
        H = self._like["hess"] = ShardHessian.from_global(self.sh, csr, diag,
                                                          like=self._like.get("hess"))
        return OperatorSum(self.sh, [H] + host_ops) if host_ops else H

    def hessian_operator(self, terms, n_vars, slack_block):
        if not isinstance(terms, ShardHessian):
            terms = self._host_hessian(terms)
        if slack_block is None:
            return terms
        if not isinstance(terms, ShardHessian):
            raise NotImplementedError("sharded backend: operator Hessian terms next to the box "
                                      "form of the barrier problem (BASELINE config 5 shape)")
        return ShardHessian(self.sh, self.sh.ops.hessian_z(terms.local, slack_block.loc,
                                                          breaks=self._z_breaks()), self.Z)

    def projections(self, A, method=None):
        return projections(A, method)

    def modified_dogleg(self, A, Y, b, trust_radius, lb, ub, norm_out=None):
        from . import qp
        return qp.modified_dogleg(A, Y, b, trust_radius, lb, ub, norm_out)

    def projected_cg(self, H, c, Z, Y, b, trust_radius, lb, ub):
        from . import qp
        return qp.projected_cg(H, c, Z, Y, b, trust_radius, lb, ub)

    def box_intersections(self, z, d, lb, ub, entire_line=False):
        from . import qp
        return qp.box_intersections(z, d, lb, ub, entire_line)


def minimize_equality_constrained(sh, fun, grad, lagr_hess, constr, jac, x0, method=None,
                                  xtol=1e-8, gtol=1e-8, max_iter=1000, callback=None, **options):
    """``minimize_constrained`` for an equality-constrained NLP on distributed data (the full
    solve of BASELINE config 4): the same outer loops (``sqp.equality_constrained_sqp`` /
    ``barrier.tr_interior_point``, reference _minimize_constrained.py:441-565) over the
    sharded backend.  Callbacks take and return distributed objects:

        fun(x) -> float (already summed over the ranks)      grad(x) -> ShardVec
        constr(x) -> ShardVec (rows)                          jac(x) -> ShardCSR
        lagr_hess(x, v) -> ShardHessian
    """
    import time
    from scipy.optimize import OptimizeResult
    from .barrier import tr_interior_point
    from .minimize import TERMINATION_MESSAGES, _METHODS, _make_stop_criteria
    from .sqp import equality_constrained_sqp
    xp = ShardedBackend(sh)
    method = _METHODS[method or 'equality_constrained_sqp']
    state = OptimizeResult(niter=0, nfev=1, ngev=1, ncev=1, njev=1, nhev=0, cg_niter=0,
                           cg_info={})
    stop_criteria = _make_stop_criteria(method, gtol, xtol, max_iter, 1e-8, callback, 0,
                                        lambda s: s)
    n, m = sh.lay.n, sh.lay.m
    f0, g0, c0, J0 = fun(x0), grad(x0), constr(x0), jac(x0)
    start = time.time()
    if method == 'equality_constrained_sqp':
        result = equality_constrained_sqp(
            lambda x: (fun(x), constr(x)), lambda x: (grad(x), jac(x)), lagr_hess,
            x0, f0, g0, c0, J0, stop_criteria, state, xp, **options)
    else:
        result = tr_interior_point(
            fun, grad, lambda x, v_eq, v_ineq: lagr_hess(x, v_eq), n, 0, m,
            lambda x: (_Empty(), constr(x)), lambda x: (None, jac(x)), x0, f0, g0,
            _Empty(), None, c0, J0, stop_criteria, None, xtol, state, xp, **options)
    result.execution_time = time.time() - start
    result.method = method
    result.message = TERMINATION_MESSAGES[result.status]
    return result


def minimize_box_inequality(sh, fun, grad, lagr_hess, constr_nl, jac_nl, x0, lb, ub, xtol=1e-8,
                            gtol=1e-8, max_iter=1000, callback=None, **options):
    """``minimize_constrained(..., constraints=(NonlinearConstraint(c, ('less', 0)),
    BoxConstraint(('interval', lb, ub))))`` on distributed data -- BASELINE config 5: the
    barrier method ``barrier.tr_interior_point`` (reference tr_interior_point.py:254-355) over
    the sharded backend, with the canonical inequality rows ``[c_nl(x); lb - x; x - ub]`` and
    their slacks partitioned like the rows / variables they belong to.  Callbacks:

        fun(x) -> float      grad(x) -> ShardVec      constr_nl(x) -> ShardVec (rows)
        jac_nl(x) -> ShardCSR      lagr_hess(x, v_nl) -> ShardHessian   (v_nl: row multipliers)
    """
    import time
    from scipy.optimize import OptimizeResult
    from .barrier import tr_interior_point
    from .minimize import TERMINATION_MESSAGES, _make_stop_criteria
    xp = ShardedBackend(sh)
    n, m = sh.lay.n, sh.lay.m
    n_ineq = m + 2 * n
    sh.register(xp.INEQ)
    sh.register(xp.Z)
    state = OptimizeResult(niter=0, nfev=1, ngev=1, ncev=1, njev=1, nhev=0, cg_niter=0,
                           cg_info={})
    stop_criteria = _make_stop_criteria('tr_interior_point', gtol, xtol, max_iter, 1e-8, callback,
                                        0, lambda s: s)

    def constr(x):
        return xp.hstack((constr_nl(x), lb - x, x - ub)), _Empty()

    def jac(x):
        return BoxInequalityJacobian(jac_nl(x)), None

    def hess(x, v_eq, v_ineq):
        return lagr_hess(x, v_ineq[:m])            # box rows have no curvature

    f0, g0 = fun(x0), grad(x0)
    c0, _ = constr(x0)
    J0, _ = jac(x0)
    start = time.time()
    result = tr_interior_point(fun, grad, hess, n, n_ineq, 0, constr, jac, x0, f0, g0, c0, J0,
                               _Empty(), None, stop_criteria, np.zeros(n_ineq, dtype=bool), xtol,
                               state, xp, **options)
    result.execution_time = time.time() - start
    result.method = 'tr_interior_point'
    result.message = TERMINATION_MESSAGES[result.status]
    return result
