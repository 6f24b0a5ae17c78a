"""Device-resident vectors and matrices: the operand types of the operator seam.

``DVec`` / ``DeviceCSR`` / ``DeviceDense`` own HBM through torch tensors
(plumbing only: allocation, streams, host<->device copies) and do ALL
arithmetic through the hand-written kernels in ``libipx.so``.  They expose the
duck-typed surface the reference's hot path uses (``.dot``, ``.T.dot``,
``+ - *`` on vectors; qp_subproblem.py:376-410,502-634), so the host control
flow reads like the reference's.
"""
import ctypes
import weakref

import numpy as np
import torch

from . import _hip

_F64 = torch.float64
_I32 = torch.int32


def _require_gpu():
    if not torch.cuda.is_available():
        raise _hip.IpxError("no HIP device visible: the ipsolver product path "
                            "is GPU-only (there is no CPU fallback)")


PACK_SLOTS = 128
PACK_BASE = 32        # the pack's slots start here: [0, 32) serve the one-off reductions
PARTS_ARENA = 1 << 18   # doubles: 64 pending reductions of two quantities at the largest grid
FOLD_MAX = 32           # include/ipx.h IPX_FOLD_MAX
SUM, MAX, MIN = 0, 1, 2


class _Context:
    """Per-process scratch: reduction workspace and scalar read-back slots."""
    _inst = None

    def __init__(self):
        _require_gpu()
        _hip.load()
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.ws = torch.empty(_hip.WS_DOUBLES, dtype=_F64, device=self.device)
        self.out = torch.zeros(PACK_BASE + PACK_SLOTS, dtype=_F64, device=self.device)
        self.open_pack = None          # the ScalarPack with enqueued, unread slots (one at a time)
        # partial sums of reductions whose results only the host wants (dot / norms): their
        # second stage runs inside the read-back (``read_folded``); a bump arena, emptied by
        # the read that ends a decision point
        self.parts = torch.empty(PARTS_ARENA, dtype=_F64, device=self.device)
        self.parts_used = 0

    def partials(self, doubles):
        """Offset of ``doubles`` free entries of the partial arena, or None when it is full
        (the caller then takes the two-launch reduction)."""
        off = self.parts_used
        if off + doubles > PARTS_ARENA:
            return None
        self.parts_used = off + doubles
        return off

    @classmethod
    def get(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst


def ctx():
    return _Context.get()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """Raw hipStream_t of torch's current stream on the context's device, as the plain integer
    ctypes converts for a ``void *`` parameter (every entry point declares its argtypes)."""
    if _raw_stream is not None:            # ~0.3 us; the Stream object below costs ~8 us
        inst = _Context._inst
        return _raw_stream((inst if inst is not None else ctx()).device.index)
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    """Device address of a tensor for a ``void *`` parameter (an integer: a ctypes.c_void_p
    object per argument was a fifth of a launch's host cost)."""
    return None if t is None else t.data_ptr()


def _empty(n):
    inst = _Context._inst
    return torch.empty(int(n), dtype=_F64, device=(inst if inst is not None else ctx()).device)


def _wrap(t):
    """A DVec around a tensor this module allocated itself (fp64, 1-D, contiguous, on the
    device: the checks of ``DVec.__init__`` are for tensors that come from outside)."""
    v = DVec.__new__(DVec)
    v.t = t
    return v


import threading


class _ReadBuffers(threading.local):
    """Host staging of the blocking reads, one set per Python thread (the C side keeps its
    pinned buffer and tag per host thread too, and polls with the GIL released: two threads
    sharing one buffer would overwrite each other's results or descriptors -- ADVICE r5)."""

    def __init__(self):
        self.buf = (ctypes.c_double * 512)()
        self.ptr = ctypes.addressof(self.buf)
        self.fold = None            # (_FoldDesc * FOLD_MAX)(), made on first use
        self.fold_ptr = None


_RB = _ReadBuffers()


def read_doubles(t, k, offset=0):
    """The k doubles t[offset : offset + k] of a device tensor as a list, behind everything
    queued on the current stream: csrc/misc.hip ipx_read_doubles (a publish kernel into pinned
    host memory + a polled word -- 15 us behind a small kernel where ``tensor.tolist()``, a copy
    into pageable memory with two runtime synchronisations, takes 21)."""
    if k <= 0:
        return []
    if k > 512:
        return t[offset:offset + k].tolist()
    rb = _RB
    _hip.call("ipx_read_doubles", t.data_ptr() + 8 * offset, int(k), rb.ptr, stream_ptr())
    return rb.buf[:k]


class _FoldDesc(ctypes.Structure):
    _fields_ = [("part", ctypes.c_void_p), ("count", ctypes.c_int32), ("op", ctypes.c_int32)]




def read_folded(descs):
    """Blocking read of scalars that are still partial sums on the device: ``descs`` = (offset
    into the context's partial arena, count, SUM / MAX / MIN) per scalar; the fold runs inside
    the read-back's kernel in the order of the reductions' own second launch (same bits)."""
    c = ctx()
    base = c.parts.data_ptr()
    rb = _RB
    if rb.fold is None:
        rb.fold = (_FoldDesc * FOLD_MAX)()
        rb.fold_ptr = ctypes.addressof(rb.fold)
    out = []
    for i in range(0, len(descs), FOLD_MAX):
        chunk = descs[i:i + FOLD_MAX]
        for q, (off, count, op) in enumerate(chunk):
            d = rb.fold[q]
            d.part, d.count, d.op = base + 8 * off, count, op
        _hip.call("ipx_read_folded", len(chunk), rb.fold_ptr, rb.ptr, stream_ptr())
        out.extend(rb.buf[:len(chunk)])
    return out


def _reduce_grid(n):
    g = (n + 1023) // 1024            # csrc/vec.hip launch_reduce: ipx_grid_for(n, 4 * 256)
    return 1 if g < 1 else (_hip.VEC_GRID_CAP if g > _hip.VEC_GRID_CAP else g)


def _read(k):
    """Read back the first k reduction outputs (one blocking read)."""
    return read_doubles(ctx().out, k)


read_slots = _read


# ---------------------------------------------------------------------------
class DVec:
    """fp64 device vector with numpy-like arithmetic routed to ipx kernels."""
    __slots__ = ("t",)
    __array_priority__ = 1000

    def __init__(self, t):
        assert t.dtype == _F64 and t.dim() == 1 and t.is_cuda
        self.t = t if t.is_contiguous() else t.contiguous()

    # -- construction / conversion
    @staticmethod
    def from_host(a):
        _require_gpu()
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
        return DVec(torch.from_numpy(a).to(ctx().device))

    @staticmethod
    def zeros(n):
        return DVec.full(n, 0.0)

    @staticmethod
    def full(n, value):
        out = _empty(n)
        _hip.call("ipx_fill", int(n), float(value), _p(out), stream_ptr())
        return _wrap(out)

    def to_host(self):
        return self.t.cpu().numpy()

    def copy(self):
        return DVec(self.t.clone())

    def zeros_like(self):
        return DVec.zeros(len(self))

    def full_like(self, value):
        return DVec.full(len(self), value)

    def __len__(self):
        return self.t.numel()

    @property
    def shape(self):
        return (self.t.numel(),)

    @property
    def size(self):
        return self.t.numel()

    def __getitem__(self, key):
        if isinstance(key, slice):
            if key.indices(len(self)) == (0, len(self), 1):
                return self       # (the whole vector: a view is the same storage anyway, and the
                #                    same OBJECT lets callers recognise a point they have seen)
            return DVec(self.t[key])
        raise TypeError("DVec supports slice views only")

    # -- arithmetic (each one kernel launch)
    def _axpby(self, a, other, b):
        out = _empty(len(self))
        _hip.call("ipx_axpby", len(self), float(a), _p(self.t), float(b),
                  _p(other.t) if other is not None else None, _p(out), stream_ptr())
        return _wrap(out)

    def add_scaled(self, o, a):
        """self + a*o in one pass (1.0*x is exact, so this equals x + a*o)."""
        return self._axpby(1.0, o, a)

    def scaled_sub(self, a, o):
        """a*self - o in one pass."""
        return self._axpby(a, o, -1.0)

    def __add__(self, o):
        if isinstance(o, DVec):
            return self._axpby(1.0, o, 1.0)
        out = _empty(len(self))
        _hip.call("ipx_affine", len(self), 1.0, _p(self.t), float(o), _p(out), stream_ptr())
        return _wrap(out)

    __radd__ = __add__

    def __sub__(self, o):
        if isinstance(o, DVec):
            return self._axpby(1.0, o, -1.0)
        return self.__add__(-float(o))

    def __rsub__(self, o):          # scalar - vec
        out = _empty(len(self))
        _hip.call("ipx_affine", len(self), -1.0, _p(self.t), float(o), _p(out), stream_ptr())
        return _wrap(out)

    def __neg__(self):
        return self._axpby(-1.0, None, 0.0)

    def __mul__(self, o):
        if isinstance(o, DVec):
            out = _empty(len(self))
            _hip.call("ipx_mul", len(self), _p(self.t), _p(o.t), _p(out), stream_ptr())
            return _wrap(out)
        return self._axpby(float(o), None, 0.0)

    __rmul__ = __mul__

    def dot(self, o):
        c = ctx()
        n = len(self)
        g = _reduce_grid(n)
        off = c.partials(g) if n > 0 else None
        if off is None:
            _hip.call("ipx_dot", n, _p(self.t), _p(o.t), _p(c.out), _p(c.ws), stream_ptr())
            return _read(1)[0]
        _hip.call("ipx_dot_partials", n, _p(self.t), _p(o.t), c.parts.data_ptr() + 8 * off,
                  stream_ptr())
        val = read_folded([(off, g, SUM)])[0]
        if c.open_pack is None or c.open_pack() is None:
            c.parts_used = 0
        return val

    def sumsq_amax(self):
        c = ctx()
        n = len(self)
        g = _reduce_grid(n)
        off = c.partials(2 * g) if n > 0 else None
        if off is None:
            _hip.call("ipx_norms", n, _p(self.t), _p(c.out), _p(c.ws), stream_ptr())
            return _read(2)
        _hip.call("ipx_norms_partials", n, _p(self.t), c.parts.data_ptr() + 8 * off, stream_ptr())
        vals = read_folded([(off, g, SUM), (off + g, g, MAX)])
        if c.open_pack is None or c.open_pack() is None:
            c.parts_used = 0
        return vals


class DeviceScalar:
    """A double the device holds (``t``: a tensor of one element) and the host may not have read
    yet: ``float()`` reads it once (a blocking read) and remembers; a kernel that consumes it
    takes ``t`` and the host learns the value from whatever it reads next anyway
    (``known``) -- the objective value of a trial point between the user's ``fun`` and the
    verdict on the step (csrc/sqp.hip ipx_sqp_judge: f_next_dev)."""
    __slots__ = ("t", "_v")

    def __init__(self, t, value=None):
        self.t, self._v = t, value

    def __float__(self):
        if self._v is None:
            self._v = read_doubles(self.t, 1)[0]
        return self._v

    def known(self, value):
        self._v = float(value)

    @property
    def is_known(self):
        return self._v is not None


class ScalarPack:
    """Reductions enqueued into consecutive slots of the context's output block and read back
    TOGETHER: one blocking D2H copy serves a whole decision point of the outer loops (the
    reference reads every norm / dot product as it goes: equality_constrained_sqp.py:138-169).
    ``dot`` / ``norm`` / ``norm_inf`` return a handle; ``read()`` returns the values, each
    computed exactly as the unpacked ``DVec.dot`` / ``norm`` / ``norm_inf`` would."""

    def __init__(self):
        self.c = ctx()
        self.k = 0
        self.how = []          # per handle: (slot | ("fold", index into folds) | None, post-processing)
        self.folds = []        # (offset, count, op) in the context's partial arena

    def _claim(self):
        # (as for the slots: two packs with unread partials at the same time are refused --
        # the read that ends one empties the arena)
        other = self.c.open_pack() if self.c.open_pack is not None else None
        if other is not None and other is not self:
            raise _hip.IpxError("ScalarPack: another pack has unread slots")
        self.c.open_pack = weakref.ref(self)

    def _slot(self, n):
        # the pack's slots are a region of their own ([PACK_BASE, ...)): a one-off reduction
        # issued between two enqueues (an operator's own norm, say) cannot overwrite them; two
        # packs with unread slots at the same time would, so that is refused
        other = self.c.open_pack() if self.c.open_pack is not None else None
        if other is not None and other is not self:
            raise _hip.IpxError("ScalarPack: another pack has unread slots")
        self.c.open_pack = weakref.ref(self)
        base = self.k
        self.k += n
        if self.k > PACK_SLOTS:
            raise _hip.IpxError("ScalarPack: more than %d slots" % PACK_SLOTS)
        return ctypes.c_void_p(self.c.out.data_ptr() + 8 * (PACK_BASE + base)), base

    def dot(self, a, b):
        if len(a) == 0:
            self.how.append((None, None))
            return len(self.how) - 1
        g = _reduce_grid(len(a))
        off = self.c.partials(g)
        if off is not None:                 # (one launch: the fold rides in the read-back)
            self._claim()
            _hip.call("ipx_dot_partials", len(a), _p(a.t), _p(b.t),
                      self.c.parts.data_ptr() + 8 * off, stream_ptr())
            self.how.append((("fold", len(self.folds)), None))
            self.folds.append((off, g, SUM))
        else:
            ptr, base = self._slot(1)
            _hip.call("ipx_dot", len(a), _p(a.t), _p(b.t), ptr, _p(self.c.ws), stream_ptr())
            self.how.append((base, None))
        return len(self.how) - 1

    def _norms(self, v, which):
        if len(v) == 0:
            self.how.append((None, None))
            return len(self.how) - 1
        g = _reduce_grid(len(v))
        off = self.c.partials(2 * g)
        if off is not None:
            self._claim()
            _hip.call("ipx_norms_partials", len(v), _p(v.t), self.c.parts.data_ptr() + 8 * off,
                      stream_ptr())
            self.how.append((("fold", len(self.folds)), "sqrt" if which == 0 else None))
            self.folds.append((off, g, SUM) if which == 0 else (off + g, g, MAX))
        else:
            ptr, base = self._slot(2)
            _hip.call("ipx_norms", len(v), _p(v.t), ptr, _p(self.c.ws), stream_ptr())
            self.how.append((base + which, "sqrt" if which == 0 else None))
        return len(self.how) - 1

    def norm(self, v):
        return self._norms(v, 0)

    def norm_inf(self, v):
        return self._norms(v, 1)

    def sumsq(self, v):
        h = self._norms(v, 0)
        self.how[h] = (self.how[h][0], None)
        return h

    def combine(self, handles, weights):
        """``((w0 v[h0] + w1 v[h1]) + w2 v[h2]) + ...`` as a ``DeviceScalar`` WITHOUT reading the
        pack (one launch folds the partial sums and forms the expression on the device, in that
        order: the bits of the same expression written on the host over ``read()``'s values).
        Ends the pack like ``read()``.  Reductions that did not leave partial sums (a full
        arena) are read and the result is a known scalar."""
        import torch
        folds = [self.how[h][0] for h in handles]
        if self.k or not all(isinstance(f, tuple) for f in folds) \
                or any(self.how[h][1] is not None for h in handles) or len(handles) > FOLD_MAX:
            v = self.read()
            acc = None
            for h, w in zip(handles, weights):
                term = float(w) * v[h]
                acc = term if acc is None else acc + term
            return DeviceScalar(None, acc)
        rb = _RB
        if rb.fold is None:
            rb.fold = (_FoldDesc * FOLD_MAX)()
            rb.fold_ptr = ctypes.addressof(rb.fold)
        base = self.c.parts.data_ptr()
        for q, f in enumerate(folds):
            off, count, op = self.folds[f[1]]
            d = rb.fold[q]
            d.part, d.count, d.op = base + 8 * off, count, op
        w = (ctypes.c_double * len(handles))(*[float(x) for x in weights])
        out = torch.empty(1, dtype=torch.float64, device=self.c.device)
        _hip.call("ipx_fold_combine", len(handles), rb.fold_ptr, ctypes.addressof(w),
                  out.data_ptr(), stream_ptr())
        if self.c.open_pack is not None and self.c.open_pack() is self:
            self.c.open_pack = None
            self.c.parts_used = 0
        return DeviceScalar(out)

    def read(self):
        vals = read_doubles(self.c.out, self.k, PACK_BASE)
        folded = read_folded(self.folds) if self.folds else []
        if self.c.open_pack is not None and self.c.open_pack() is self:
            self.c.open_pack = None
            self.c.parts_used = 0
        out = []
        for slot, post in self.how:
            if slot is None:
                v = 0.0
            elif isinstance(slot, tuple):
                v = folded[slot[1]]
            else:
                v = vals[slot]
            out.append(float(np.sqrt(v)) if post == "sqrt" else v)
        return out


def norm(v):
    if len(v) == 0:
        return 0.0
    return float(np.sqrt(v.sumsq_amax()[0]))


def norm_inf(v):
    if len(v) == 0:
        return 0.0
    return v.sumsq_amax()[1]


def clip(x, lb, ub):
    if not isinstance(x, DVec):          # distributed vectors (sharded.ShardVec) bring their own
        return x._clip(lb, ub)
    out = _empty(len(x))
    _hip.call("ipx_clip", len(x), _p(x.t), _p(lb.t), _p(ub.t), _p(out), stream_ptr())
    return DVec(out)


def count_outside_box(x, lb, ub):
    if not isinstance(x, DVec):
        return x._count_outside_box(lb, ub)
    c = ctx()
    _hip.call("ipx_box_inside", len(x), _p(x.t), _p(lb.t), _p(ub.t), _p(c.out), _p(c.ws),
              stream_ptr())
    return _read(1)[0]


def box_sphere_reduce(z, d, dscale, lb, ub):
    if not isinstance(z, DVec):
        return z._box_sphere_reduce(d, dscale, lb, ub)
    c = ctx()
    _hip.call("ipx_box_sphere_reduce", len(z), _p(z.t), _p(d.t), float(dscale),
              _p(lb.t) if lb is not None else None, _p(ub.t) if ub is not None else None,
              _p(c.out), _p(c.ws), stream_ptr())
    return _read(7)


def hstack(parts):
    return DVec(torch.cat([p.t for p in parts]))


# ---------------------------------------------------------------------------
def _tiles_for(rowptr_host, tile_nnz=_hip.SPMV_TILE_NNZ, max_rows=1024, row_breaks=None):
    """SpMV row tiles: row boundaries, then rowptr at them.  ``row_breaks`` forces tile
    boundaries at the given rows (the sharded loop sums per-tile partials over a rank's own
    rows only, so own / halo rows must not share a tile)."""
    if row_breaks is not None and len(row_breaks):
        rp = np.ascontiguousarray(rowptr_host, dtype=np.int64)
        cuts = sorted(set([0, len(rp) - 1] + [int(b) for b in row_breaks if 0 < b < len(rp) - 1]))
        bounds = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            t = _tiles_for((rp[a:b + 1] - rp[a]).astype(np.int32), tile_nnz, max_rows)
            k = len(t) // 2
            bounds.extend(int(v) + a for v in t[:k - 1])
        bounds.append(len(rp) - 1)
        rows = np.array(bounds, dtype=np.int64)
        return np.concatenate((rows, rp[rows])).astype(np.int32)
    nrows = len(rowptr_host) - 1
    cap = 2 * (nrows + 2)
    tiles = np.empty(cap, dtype=np.int32)
    rp = np.ascontiguousarray(rowptr_host, dtype=np.int32)
    nt = _hip.call("ipx_csr_tiles_host", nrows, rp.ctypes.data_as(ctypes.c_void_p),
                   int(tile_nnz), int(max_rows), tiles.ctypes.data_as(ctypes.c_void_p), cap)
    return tiles[:2 * (nt + 1)].copy()      # row boundaries, then rowptr at them


class CSRPattern:
    """Sparsity pattern (device + host copies) with its SpMV row tiles and a
    lazily built transpose.  Patterns are immutable and shared by every value
    refresh of a Jacobian/Hessian (hard part 7)."""

    def __init__(self, indptr, indices, shape, row_breaks=None):
        _require_gpu()
        self.shape = (int(shape[0]), int(shape[1]))
        self.indptr_h = np.ascontiguousarray(indptr, dtype=np.int32)
        self.indices_h = np.ascontiguousarray(indices, dtype=np.int32)
        dev = ctx().device
        self.indptr = torch.from_numpy(self.indptr_h).to(dev)
        self.indices = torch.from_numpy(self.indices_h).to(dev)
        tiles = _tiles_for(self.indptr_h, row_breaks=row_breaks)
        self.ntiles = len(tiles) // 2 - 1
        self.tiles_h = tiles
        self.tiles = torch.from_numpy(tiles).to(dev)
        self.nnz = int(self.indptr_h[-1])
        self._transpose = None

    def diagonal_positions(self):
        """Positions (into the value array) of the entries (i, i), one per row, as a device
        int32 tensor -- or None when the matrix is not square or some row has no diagonal
        entry.  Symbolic, once per pattern."""
        if not hasattr(self, "_diag_pos"):
            pos = None
            m, n = self.shape
            if m == n and self.nnz >= m:
                rows = np.repeat(np.arange(m, dtype=np.int64), np.diff(self.indptr_h))
                hit = np.flatnonzero(self.indices_h == rows)
                if len(hit) == m and np.array_equal(rows[hit], np.arange(m)):
                    pos = torch.from_numpy(hit.astype(np.int32)).to(ctx().device)
            self._diag_pos = pos
        return self._diag_pos

    def same_as(self, indptr, indices):
        return (len(indptr) == len(self.indptr_h) and len(indices) == len(self.indices_h)
                and np.array_equal(indptr, self.indptr_h)
                and np.array_equal(indices, self.indices_h))

    def tile_range(self, row_lo, row_hi):
        """Tiles covering exactly rows [row_lo, row_hi) (both must be tile boundaries)."""
        b = self.tiles_h[:self.ntiles + 1]
        t0, t1 = int(np.searchsorted(b, row_lo)), int(np.searchsorted(b, row_hi))
        if b[t0] != row_lo or b[t1] != row_hi:
            raise ValueError("rows [%d, %d) do not start / end on tile boundaries"
                             % (row_lo, row_hi))
        return t0, t1

    def transpose(self, row_breaks=None):
        """(pattern of A', permutation with valT = val[perm]) -- symbolic, once.  ``row_breaks``
        (tile boundaries of the transpose) must be given by the first caller: a later request
        for other boundaries is refused instead of silently keeping the cached tiles."""
        breaks = None if row_breaks is None else tuple(int(b) for b in row_breaks)
        if self._transpose is not None and breaks is not None \
                and breaks != getattr(self, "_transpose_breaks", None):
            raise _hip.IpxError("CSRPattern.transpose: the transpose of this pattern was already "
                                "built with row breaks %r, now asked for %r"
                                % (getattr(self, "_transpose_breaks", None), breaks))
        if self._transpose is None:
            self._transpose_breaks = breaks
            import scipy.sparse as sps
            m, n = self.shape
            tag = sps.csr_matrix((np.arange(1, self.nnz + 1, dtype=np.float64),
                                  self.indices_h, self.indptr_h), shape=(m, n))
            t = sps.csr_matrix(tag.T)
            t.sort_indices()
            perm = (t.data - 1).astype(np.int64)
            pat = CSRPattern(t.indptr, t.indices, (n, m), row_breaks=row_breaks)
            self._transpose = (pat, torch.from_numpy(perm).to(ctx().device))
        return self._transpose


class DeviceCSR:
    """CSR matrix in HBM: ``dot`` and ``T.dot`` are owner-computes SpMVs (the
    transpose is stored as a second CSR, so there are no atomics)."""

    def __init__(self, pattern, val):
        self.pattern = pattern
        self.val = val
        self.shape = pattern.shape
        self._T = None

    @staticmethod
    def from_scipy(M, pattern=None, row_breaks=None):
        import scipy.sparse as sps
        M = sps.csr_matrix(M)
        if not M.has_sorted_indices:
            M = M.sorted_indices()
        if pattern is None or not pattern.same_as(M.indptr, M.indices):
            pattern = CSRPattern(M.indptr, M.indices, M.shape, row_breaks=row_breaks)
        val = torch.from_numpy(np.ascontiguousarray(M.data, dtype=np.float64)).to(ctx().device)
        return DeviceCSR(pattern, val)

    def to_scipy(self):
        import scipy.sparse as sps
        p = self.pattern
        return sps.csr_matrix((self.val.cpu().numpy(), p.indices_h, p.indptr_h), shape=p.shape)

    def with_sorted_indices(self):
        """This matrix with the column indices of every row in increasing order (the
        normal-equation assembly merges sorted rows).  Matrices built by ``from_scipy`` are
        sorted already; a hand-made pattern is checked once and, if need be, re-ordered
        through a cached permutation of the values."""
        pat = self.pattern
        info = getattr(pat, "_ipx_sorted", None)
        if info is None:
            rows = np.repeat(np.arange(pat.shape[0], dtype=np.int64), np.diff(pat.indptr_h))
            key = rows * max(pat.shape[1], 1) + pat.indices_h.astype(np.int64)
            if np.all(np.diff(key) >= 0):
                info = (None, None)
            else:
                order = np.argsort(key, kind="stable")
                spat = CSRPattern(pat.indptr_h, pat.indices_h[order], pat.shape)
                info = (spat, torch.from_numpy(order).to(ctx().device))
            pat._ipx_sorted = info
        if info[0] is None:
            return self
        return DeviceCSR(info[0], self.val[info[1]])

    @property
    def T(self):
        if self._T is None:
            pat, perm = self.pattern.transpose()
            self._T = DeviceCSR(pat, self.val[perm])
            self._T._T = self
        return self._T

    def spmv(self, x, alpha=1.0, diag=None, beta=0.0, yin=None, out=None, reduce=False,
             slot=0):
        """out = alpha*A x [+ diag*x] [+ beta*yin]; with ``reduce`` leaves
        (sum out^2, sum x*out) in scalar slots [2*slot, 2*slot+1] of the context
        (read them back with ``read_slots``)."""
        p = self.pattern
        m, n = p.shape
        assert len(x) == n, (len(x), n)
        if out is None:
            out = _wrap(_empty(m))
        c = ctx()
        _hip.call("ipx_csr_spmv", m, n, _p(p.indptr), _p(p.indices), _p(self.val),
                  _p(p.tiles), p.ntiles, _p(x.t), float(alpha),
                  _p(diag.t) if diag is not None else None, float(beta),
                  _p(yin.t) if yin is not None else None, _p(out.t),
                  1 if m == n else 0,
                  ctypes.c_void_p(c.out.data_ptr() + 16 * slot) if reduce else None,
                  _p(c.ws), stream_ptr())
        return out

    def matvec_sumsq(self, x, slot=0):
        """(A x, ||A x||^2) -- one blocking read."""
        out = self.spmv(x, reduce=True, slot=slot)
        return out, read_slots(2 * slot + 1)[2 * slot]

    def rmatvec_sub(self, v, x, reduce=False, slot=0):
        """x - A'v (optionally leaving ||.||^2 in the slot)."""
        return self.T.spmv(v, alpha=-1.0, beta=1.0, yin=x, reduce=reduce, slot=slot)

    def dot(self, x):
        return self.spmv(x)

    matvec = dot

    def frobenius_norm(self):
        return norm(DVec(self.val))
