"""HIP backend of the outer loops: the vector / matrix / operator factory that
``sqp.py`` and ``barrier.py`` are written against.  Every array lives in HBM
(``DVec`` / ``DeviceCSR`` / ``DeviceDense``) and every arithmetic operation is
an ipx kernel.  This is the ONLY backend the package ships; tests may inject
the CPU oracle's backend to exercise the host logic without a GPU.
"""
import numpy as np
import scipy.sparse as sps
import torch

from . import _hip
from . import device as dv
from . import qp, projector
from .canonical import HessianSum
from .dense import DeviceDense
from .device import DVec, DeviceCSR, CSRPattern, _p, stream_ptr, ctx
from .operators import DeviceHessian, DiagonalOperator

name = "hip"


# ---- vectors ---------------------------------------------------------------
def asvec(a, space=None):
    # (space: which vector space the caller means -- "x", "eq", "ineq", "z"; only the
    # distributed backends, where a length does not always tell, look at it)
    return a if isinstance(a, DVec) else DVec.from_host(a)


def tohost(v):
    return v.to_host() if isinstance(v, DVec) else np.asarray(v)


zeros = DVec.zeros


def full(n, value, space=None):
    return DVec.full(n, value)

hstack = dv.hstack
norm = dv.norm
norm_inf = dv.norm_inf


def copy(v):
    return v.copy()


def pack():
    """Scalar pack: enqueue norms / dot products, read them back together."""
    return dv.ScalarPack()


def dot(a, b):
    return a.dot(b)


def maximum(v, c):
    out = dv._empty(len(v))
    _hip.call("ipx_max_scalar", len(v), _p(v.t), float(c), _p(out), stream_ptr())
    return DVec(out)


def where_positive(v, a, c):
    out = dv._empty(len(v))
    _hip.call("ipx_where_positive", len(v), _p(v.t), _p(a.t), float(c), _p(out), stream_ptr())
    return DVec(out)


def sum_log(s):
    """sum(log s_i), -inf when any s_i <= 0 (tr_interior_point.py:93-95)."""
    if len(s) == 0:
        return 0.0
    c = ctx()
    _hip.call("ipx_sum_log", len(s), _p(s.t), _p(c.out), _p(c.ws), stream_ptr())
    total, bad = dv.read_slots(2)
    return -np.inf if bad > 0 else total


def assign_negated_where(s, mask, c):
    """s[mask] = -c[mask] in place (s is a view of z)."""
    m = DVec.from_host(np.asarray(mask, dtype=np.float64))
    _hip.call("ipx_assign_negated_where", len(s), _p(s.t), _p(m.t), _p(c.t), stream_ptr())


# ---- matrices / operators ----------------------------------------------------
_pattern_cache = {}


def _upload_csr(M, key):
    """scipy sparse -> DeviceCSR, re-using the device pattern (tiles, transpose,
    symbolic factorization) of the previous matrix uploaded under ``key`` when
    the sparsity pattern is unchanged -- the usual case for a Jacobian."""
    M = sps.csr_matrix(M)
    if not M.has_sorted_indices:
        M = M.sorted_indices()
    A = DeviceCSR.from_scipy(M, _pattern_cache.get(key))
    _pattern_cache[key] = A.pattern
    return A


def matrix(J, key="jac"):
    if isinstance(J, (DeviceCSR, DeviceDense)):
        return J
    if sps.issparse(J) or 0 in np.shape(J):
        return _upload_csr(J, (key, np.shape(J)))
    return DeviceDense.from_host(J)


def mark_constant(A):
    """Declare that A's values never change: ``projections(A)`` is then computed
    once and re-used."""
    A.constant = True
    return A


def diagonal_operator(d):
    return DiagonalOperator(d)


class HostCallbackOperator:
    """User operator that only lives on the host (e.g. a finite-difference
    Hessian): the vector crosses PCIe for the call.  Outside the accelerated
    path by construction (_numdiff row)."""

    def __init__(self, op):
        self.op = op
        self.shape = op.shape

    def dot(self, p):
        return DVec.from_host(self.op.dot(p.to_host()))


class PaddedOperator:
    """[[H, 0], [0, 0]] acting on z = [x; s] for an x-space operator H."""

    def __init__(self, op, n_vars, n_total):
        self.op, self.n_vars, self.n_total = op, n_vars, n_total

    def dot(self, p):
        return dv.hstack((self.op.dot(p[:self.n_vars]),
                          DVec.zeros(self.n_total - self.n_vars)))


def _extend_pattern(pattern, n_total, row_breaks=None):
    """CSR pattern of [[H, 0], [0, 0]] (n_total x n_total) sharing H's columns (row tiles cut
    at ``row_breaks``: the sharded solver's own / halo boundaries)."""
    cache = getattr(pattern, "_ipx_extended", None)
    if cache is None:
        cache = pattern._ipx_extended = {}
    key = n_total if row_breaks is None else (n_total, tuple(int(b) for b in row_breaks))
    if key not in cache:
        extra = n_total - pattern.shape[0]
        indptr = np.concatenate((pattern.indptr_h,
                                 np.full(extra, pattern.indptr_h[-1], dtype=np.int32)))
        cache[key] = CSRPattern(indptr, pattern.indices_h, (n_total, n_total),
                                row_breaks=row_breaks)
    return cache[key]


_merge_cache = {}


def _merge_sparse_terms(terms):
    """Sum of scipy CSR matrices as a DeviceCSR on the union of their patterns.
    The union pattern and the position of every term's nonzeros in it are
    symbolic work (host, cached per combination of patterns); the values are
    uploaded and added by ipx_scatter_add."""
    shape = terms[0].shape
    key = tuple((t.shape, t.nnz, hash(t.indptr.tobytes()), hash(t.indices.tobytes()))
                for t in terms)
    hit = _merge_cache.get(key)
    if hit is None:
        ncols = shape[1]
        keys = []
        for t in terms:
            rows = np.repeat(np.arange(shape[0], dtype=np.int64), np.diff(t.indptr))
            keys.append(rows * ncols + t.indices.astype(np.int64))
        union = np.unique(np.concatenate(keys))                      # sorted (row, col) pairs
        urows = (union // ncols).astype(np.int64)
        indptr = np.concatenate(([0], np.cumsum(np.bincount(urows, minlength=shape[0]))))
        pattern = CSRPattern(indptr.astype(np.int32), (union % ncols).astype(np.int32), shape)
        maps = [torch.from_numpy(np.searchsorted(union, k).astype(np.int32)).to(ctx().device)
                for k in keys]
        hit = _merge_cache[key] = (pattern, maps)
        if len(_merge_cache) > 16:
            _merge_cache.pop(next(iter(_merge_cache)))
    pattern, maps = hit
    val = torch.zeros(pattern.nnz, dtype=torch.float64, device=ctx().device)
    for t, idx in zip(terms, maps):
        data = torch.from_numpy(np.ascontiguousarray(t.data, dtype=np.float64)).to(ctx().device)
        _hip.call("ipx_scatter_add", t.nnz, _p(data), _p(idx), _p(val), stream_ptr())
    return DeviceCSR(pattern, val)


_dense_cache = {}


def _immutable(a):
    """A numpy array nobody can write through: read-only itself and down its chain of bases."""
    while isinstance(a, np.ndarray):
        if a.flags.writeable:
            return False
        a = a.base
    return a is None


FINGERPRINT_MIN_BYTES = 1 << 24      # arrays below 16 MB are simply uploaded again


def _fingerprint(h):
    """~1e5 probed entries of a large C-contiguous matrix as bytes: every (size // 65536)-th
    entry, the diagonal, the first and the last row."""
    flat = h.reshape(-1)
    step = max(1, flat.size // 65536)
    parts = [flat[::step], h[0], h[-1]]
    if h.ndim == 2:
        parts.append(np.diagonal(h))
    return np.concatenate([np.ascontiguousarray(p).ravel() for p in parts]).tobytes()


def _upload_dense_cached(h):
    """A dense Hessian term returned by a host callback.  The reference wraps ``hess(x)`` anew
    every iteration (_minimize_constrained.py:395-407); here that is an 800 MB upload per outer
    iteration for BASELINE config 2 (0.9 of its 1.6 s), although a quadratic objective returns
    the SAME array every time.  Arrays that CANNOT change -- marked read-only
    (``H.setflags(write=False)``) or declared constant (``constant_hessian``) -- are uploaded
    once.  A WRITABLE array at the same address, of the same shape, is recognised by a
    fingerprint of ~1e5 probed entries (an exact comparison reads as many bytes as the upload
    moves and is slower than it: 2.8 s instead of 1.6 s, measured): a callback that returns
    the same matrix gets the device copy, one that computes a new matrix or refills its buffer
    changes nearly every entry and is uploaded.  What the probe cannot see is an IN-PLACE
    change of a few off-diagonal entries of a large matrix that touches none of the probed ones
    -- a callback that does that must return a new array (any new address is uploaded)."""
    const, orig = getattr(h, "_ipx_constant", False), h
    h = np.asarray(h, dtype=np.float64)
    if h.size == 0 or not h.flags.c_contiguous:
        return DeviceDense.from_host(h)
    fixed = const or _immutable(h)
    if not fixed and h.nbytes < FINGERPRINT_MIN_BYTES:
        return DeviceDense.from_host(h)
    ident = orig if const else h
    key = (h.__array_interface__["data"][0], h.shape)
    hit = _dense_cache.get(key)
    if hit is not None:
        if fixed and hit[0] is ident:
            return hit[1]
        if not fixed and hit[2] is not None and hit[2] == _fingerprint(h):
            DENSE_CACHE_STATS["fingerprint_hits"] += 1
            return hit[1]
    if len(_dense_cache) > 2:
        _dense_cache.clear()
    D = DeviceDense.from_host(h)
    DENSE_CACHE_STATS["uploads"] += 1
    # (the entry keeps the array alive: the address stays its own)
    _dense_cache[key] = (ident, D, None if fixed else _fingerprint(h))
    return D


DENSE_CACHE_STATS = {"uploads": 0, "fingerprint_hits": 0}


# CG iterations of the last tangential step (the outer loops report them: note_cg_length).  A
# Hessian assembled between two SHORT solves keeps its diagonal terms as a vector (every
# product then reads 8 n bytes more); before long ones they are merged into a copy of the CSR
# values (operators.DeviceHessian).
_cg_length = {"last": 0}
MERGE_DIAGONAL_FROM = 12


def note_cg_length(niter):
    _cg_length["last"] = int(niter)


def hessian_operator(terms, n_vars, slack_block):
    """Device operator for the Lagrangian Hessian terms (HessianSum from
    canonical.lagrangian_hessian) and, in barrier problems, the diagonal slack
    block: ``[Hx p_x ; slack_block * p_s]`` (tr_interior_point.py:222-241)."""
    flat = terms.flat_terms() if isinstance(terms, HessianSum) else list(terms)
    n_total = n_vars + (len(slack_block) if slack_block is not None else 0)
    csr, diag, others = None, None, []
    sparse_terms = []
    for h in flat:
        if sps.issparse(h):
            h = sps.csr_matrix(h)
            if not h.has_canonical_format:
                h = h.copy()
                h.sum_duplicates()
            rows = np.repeat(np.arange(h.shape[0], dtype=np.int32), np.diff(h.indptr))
            if np.array_equal(rows, h.indices):          # purely diagonal pattern
                d = DVec.from_host(h.diagonal() if h.nnz < h.shape[0] else h.data)
                diag = d if diag is None else diag + d
            else:
                sparse_terms.append(h)
        elif isinstance(h, DeviceCSR) and csr is None:
            csr = h
        elif isinstance(h, DVec):
            diag = h if diag is None else diag + h
        elif isinstance(h, (DeviceCSR, DeviceHessian, DeviceDense)) \
                or getattr(h, "device_operator", False):
            others.append(h)
        elif isinstance(h, np.ndarray):
            others.append(_upload_dense_cached(h))
        else:
            others.append(HostCallbackOperator(h))
    if sparse_terms:
        # several sparse terms become ONE matrix on their union pattern (symbolic
        # on the host, values summed on the device), so the product stays one
        # fused SpMV; differs from term-by-term products by rounding only
        up = (_upload_csr(sparse_terms[0], ("hess", sparse_terms[0].shape))
              if len(sparse_terms) == 1 else _merge_sparse_terms(sparse_terms))
        if csr is None:
            csr = up
        else:
            others.append(up)
    if slack_block is None:
        return DeviceHessian(n_vars, csr, diag, others,
                             merge=_cg_length["last"] >= MERGE_DIAGONAL_FROM)
    # z-space: extend the CSR block with empty slack rows, put the slack block
    # on the diagonal, pad any other x-space term
    if csr is not None:
        csr = DeviceCSR(_extend_pattern(csr.pattern, n_total), csr.val)
    xdiag = diag if diag is not None else DVec.zeros(n_vars)
    zdiag = dv.hstack((xdiag, slack_block))
    return DeviceHessian(n_total, csr, zdiag,
                         [PaddedOperator(h, n_vars, n_total) for h in others])


def augmented_jacobian(J_eq, J_ineq, s, n_vars, n_eq, n_ineq):
    """[[J_eq, 0], [J_ineq, diag(s)]] (tr_interior_point.py:141-194) with the
    slack entries written on the device."""
    if isinstance(J_eq, DeviceCSR) and isinstance(J_ineq, DeviceCSR):
        from . import device_mode
        return device_mode.augmented_jacobian(J_eq, J_ineq, s, n_vars, n_eq, n_ineq)
    if sps.issparse(J_eq) or sps.issparse(J_ineq):
        J_eq, J_ineq = sps.csr_matrix(J_eq), sps.csr_matrix(J_ineq)
        J = sps.vstack([J_eq, J_ineq], format="csr")
        J.sort_indices()
        # one extra entry at the end of every inequality row (largest column)
        shift = np.concatenate((np.zeros(n_eq + 1, dtype=np.int64),
                                np.arange(1, n_ineq + 1, dtype=np.int64)))
        indptr = J.indptr.astype(np.int64) + shift
        nnz = int(indptr[-1])
        slots = (indptr[n_eq + 1:] - 1).astype(np.int64)
        mask = np.zeros(nnz, dtype=bool)
        mask[slots] = True
        indices = np.empty(nnz, dtype=np.int32)
        data = np.zeros(nnz)
        indices[mask] = n_vars + np.arange(n_ineq)
        indices[~mask] = J.indices
        data[~mask] = J.data
        A = _upload_csr(sps.csr_matrix((data, indices, indptr.astype(np.int32)),
                                       shape=(n_eq + n_ineq, n_vars + n_ineq)),
                        ("augjac", n_vars, n_eq, n_ineq))
        idx = torch.from_numpy(slots.astype(np.int32)).to(ctx().device)
        _hip.call("ipx_scatter", n_ineq, _p(s.t), _p(idx), _p(A.val), stream_ptr())
        return A
    s_h = s.to_host()
    top = np.hstack((np.atleast_2d(J_eq).reshape(n_eq, n_vars), np.zeros((n_eq, n_ineq))))
    bot = np.hstack((np.atleast_2d(J_ineq).reshape(n_ineq, n_vars), np.diag(s_h)))
    return DeviceDense.from_host(np.vstack((top, bot)))


# ---- the trust-region subproblem (HIP kernels) ---------------------------
def projections(A, method=None):
    return projector.projections(A, method)


modified_dogleg = qp.modified_dogleg
projected_cg = qp.projected_cg
box_intersections = qp.box_intersections
