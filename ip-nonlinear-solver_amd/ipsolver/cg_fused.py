"""Device-resident projected CG: host side of csrc/cg.hip.

Same algorithm and results as ``qp.projected_cg`` (reference
qp_subproblem.py:416-643); the per-iteration scalars stay on the GPU and the
host reads one 128-byte state block per *batch* of iterations instead of three
scalars per iteration.  Rare events (box-infeasible iterate, projection
refinement) hand the iteration back to the host, which finishes it with the
general kernels and resumes the loop.
"""
import ctypes
import os

import numpy as np
import torch

from . import _hip
from . import device as dv
from .device import DVec, DeviceCSR, _p, stream_ptr, ctx

_TINY = 1e-25
_P = ctypes.c_void_p
_I64 = ctypes.c_int64

# state block indices (csrc/cg.hip)
ST_RTG0, ST_RTG1, ST_TOL, ST_RADIUS, ST_ALPHA, ST_STOP, ST_NITER, ST_BETA = range(8)
ST_PTHP, ST_ORTH_RHS, ST_XNORM2, ST_VIOL, ST_ORTH, ST_IT_DONE, ST_MARGIN = 8, 9, 10, 11, 12, 13, 14
ST_PRIME_STEPS = 15


class CgArgs(ctypes.Structure):
    _fields_ = [(name, typ) for name, typ in (
        ("n", _I64), ("m", _I64),
        ("A_rowptr", _P), ("A_colidx", _P), ("A_val", _P), ("A_tiles", _P), ("A_ntiles", _I64),
        ("At_rowptr", _P), ("At_colidx", _P), ("At_val", _P), ("At_tiles", _P), ("At_ntiles", _I64),
        ("H_rowptr", _P), ("H_colidx", _P), ("H_val", _P), ("H_tiles", _P), ("H_ntiles", _I64),
        ("H_diag", _P), ("banded", _P),
        ("x", _P), ("p", _P), ("r", _P), ("Hp", _P),
        ("w", _P), ("v", _P), ("t", _P),
        ("lb", _P), ("ub", _P), ("state", _P),
        ("part1", _P), ("part2", _P), ("part3", _P), ("part4", _P),
        ("vec_grid", _I64), ("solver_kind", _I64), ("pb", _P), ("H_hmax", _I64), ("H_tile_rows", _I64),
        ("r_next", _P), ("A_own", _P), ("A_span", _I64), ("fold_ws", _P),
        ("At_vown", _P), ("At_qv", _I64), ("A_tile_nnz", _I64),
        ("At_ell_row", _P), ("At_ell_val", _P),
        ("H_col16", _P), ("H_rowlen", _P), ("A_col16", _P), ("no_radius", _I64),
        ("A_off16", _P), ("A_rowfirst", _P), ("A_rl", _I64), ("P_win", _P), ("P_nspan", _I64),
        ("P_navn", _I64), ("H_operator", _I64),
        ("resident", _I64), ("R_ll", _P), ("R_hw", _I64), ("R_seq", _P))]


# Counters over the life of the process (diagnostics: how often the device loop
# had to hand an iteration back to the host).
STATS = {"calls": 0, "iterations": 0, "batches": 0, "box_events": 0, "refine_events": 0,
         "primed_on_device": 0,   # calls whose priming read nothing back (ipx_cg_prime_state)
         "prime_retries": 0,      # ... of which the device sent back to the host-driven priming
         "host_primed_directly": 0,   # calls that skipped the device priming after such a retry
         "resident_calls": 0,     # solves whose batches ran as resident launches (csrc/resident.hip)
         "resident_fallbacks": 0, # batches repeated on the separate launches (stop code 8)
         "operator_calls": 0}     # solves whose Hessian was an operator applied by the host


def _hessian_parts(H):
    """(csr, diag) for the Hessian operator types the fused loop understands."""
    from .operators import DeviceHessian
    if isinstance(H, DeviceCSR) and H.shape[0] == H.shape[1]:
        return H, None
    if isinstance(H, DeviceHessian) and H.csr is not None and not H.others:
        return H.csr, H.diag
    return None


def _dense_hessian(H):
    """The dense matrix behind a Hessian operator that is nothing else (a DeviceDense, or the
    DeviceHessian backend_hip builds from one ndarray term), or None."""
    from .dense import DeviceDense
    from .operators import DeviceHessian
    if isinstance(H, DeviceDense) and H.shape[0] == H.shape[1]:
        return H
    if isinstance(H, DeviceHessian) and H.csr is None and H.diag is None \
            and len(H.others) == 1 and isinstance(H.others[0], DeviceDense):
        return H.others[0]
    return None


FUSE_HMAX = 64          # csrc/cg.hip FUSE_HMAX


def fuse_halo(pattern):
    """Halo width for the fused step2 + H.p kernel (csrc/cg.hip k_cg_step2_hp), or 0
    when the pattern does not qualify: square, every row tile on the SpMV's fast
    path, columns of a tile within ``hmax <= 64`` of its row range,
    every tile at least ``hmax`` rows long.  Symbolic; cached on the pattern."""
    cached = getattr(pattern, "_ipx_fuse_halo", None)
    if cached is not None:
        return cached
    hmax = 0
    n = pattern.shape[0]
    nt = pattern.ntiles
    if pattern.shape[0] == pattern.shape[1] and nt > 0 and pattern.nnz > 0:
        t = pattern.tiles_h
        r0, r1 = t[:nt].astype(np.int64), t[1:nt + 1].astype(np.int64)
        s, e = t[nt + 1:2 * nt + 1].astype(np.int64), t[nt + 2:2 * nt + 2].astype(np.int64)
        ok = np.all(e - s <= _hip.SPMV_TILE_NNZ) and np.all(r1 - r0 <= 1024)
        if ok:
            # tiles without nonzeros (the empty slack rows of a z-space Hessian) only
            # update p and x; the halo is set by the others
            full = np.flatnonzero(e > s)
            idx = pattern.indices_h
            cmin = np.minimum.reduceat(idx, s[full])
            cmax = np.maximum.reduceat(idx, s[full])
            h = int(max(np.max(r0[full] - cmin), np.max(cmax + 1 - r1[full]), 1))
            if h <= FUSE_HMAX and h <= int(np.min(r1 - r0)):
                hmax = h
    pattern._ipx_fuse_halo = hmax
    return hmax


def compact_columns(pattern, hmax):
    """Compact index form of a banded square pattern for the fused step2 + H.p kernel
    (csrc/cg.hip C16): ``(col16, rowlen)`` -- per nonzero its column as a 16-bit offset into
    the row tile's span (``col - max(first row of the tile - hmax, 0)``), per tile the common
    length of its rows or -1.  Symbolic; cached on the pattern."""
    key = ("_ipx_col16", int(hmax))
    cache = getattr(pattern, "_ipx_col16", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    nt = pattern.ntiles
    t = pattern.tiles_h
    r0 = t[:nt].astype(np.int64)
    s, e = t[nt + 1:2 * nt + 1].astype(np.int64), t[nt + 2:2 * nt + 2].astype(np.int64)
    c_lo = np.maximum(r0 - hmax, 0)
    off = pattern.indices_h.astype(np.int64) - np.repeat(c_lo, e - s)
    assert off.size == 0 or (off.min() >= 0 and off.max() < 65536)
    lens = np.diff(pattern.indptr_h.astype(np.int64))
    rows = np.diff(t[:nt + 1].astype(np.int64))
    first = lens[np.minimum(r0, len(lens) - 1)]
    uniform = (e - s) == rows * first
    # (equal totals do not make equal rows: check min == max per tile)
    nonempty = rows > 0
    lo = np.minimum.reduceat(lens, r0[nonempty]) if nonempty.any() else np.array([])
    hi = np.maximum.reduceat(lens, r0[nonempty]) if nonempty.any() else np.array([])
    same = np.zeros(nt, dtype=bool)
    same[nonempty] = lo == hi
    rowlen = np.where(uniform & same, first, -1).astype(np.int32)
    dev = ctx().device
    out = (torch.from_numpy(off.astype(np.uint16).view(np.int16)).to(dev),
           torch.from_numpy(rowlen).to(dev))
    pattern._ipx_col16 = (key, out)
    return out


def fuse_own(pattern, tile_nnz=None):
    """Column ownership for the fused step1 + A.r kernel (csrc/cg.hip k_cg_step1_ar):
    ``(own, span, tiles, ntiles)`` or None.  Qualifies when every row tile is on the
    SpMV's fast path and non-empty and the tiles sweep the columns monotonically (first
    column touched non-decreasing from tile to tile: every banded Jacobian); tile t then
    owns the columns from its first touched column up to the next tile's.  ``tile_nnz``
    = 1024 cuts its own, finer row tiles for this kernel (twice the workgroups, half the
    registers); None uses the pattern's SpMV tiles.  Symbolic; cached."""
    attr = "_ipx_fuse_own_%s" % (tile_nnz or "std")
    cached = getattr(pattern, attr, False)
    if cached is not False:
        return cached
    out = None
    n = pattern.shape[1]
    if tile_nnz:
        t = dv._tiles_for(pattern.indptr_h, tile_nnz, 1024)
        nt = len(t) // 2 - 1
    else:
        t, nt = pattern.tiles_h, pattern.ntiles
    if nt > 0 and pattern.nnz > 0:
        r0, r1 = t[:nt].astype(np.int64), t[1:nt + 1].astype(np.int64)
        s, e = t[nt + 1:2 * nt + 1].astype(np.int64), t[nt + 2:2 * nt + 2].astype(np.int64)
        if np.all(e - s <= (tile_nnz or _hip.SPMV_TILE_NNZ)) and np.all(r1 - r0 <= 1024):
            # tiles without nonzeros (rows of a column block that belong to other ranks'
            # constraints) own no columns
            full = np.flatnonzero(e > s)
            idx = pattern.indices_h
            cmin_f = np.minimum.reduceat(idx, s[full]).astype(np.int64)
            cmax_f = np.maximum.reduceat(idx, s[full]).astype(np.int64)
            if np.all(np.diff(cmin_f) >= 0):
                # first own column per tile: a full tile starts at its first touched column
                # (the first full tile at 0), an empty tile where the next full one does
                start = np.full(nt + 1, n, dtype=np.int64)
                start[full] = cmin_f
                start[full[0]] = 0
                own = np.minimum.accumulate(start[::-1])[::-1]
                c_hi = own[1:].copy()
                c_hi[full] = np.maximum(cmax_f + 1, own[1:][full])
                span = int(np.max(c_hi - own[:-1]))
                if span <= 2048:
                    table = np.concatenate((own, c_hi, [n])).astype(np.int32)
                    dev = ctx().device
                    tiles = torch.from_numpy(np.ascontiguousarray(t)).to(dev) if tile_nnz \
                        else pattern.tiles
                    out = (torch.from_numpy(table).to(dev), span, tiles, nt)
    setattr(pattern, attr, out)
    return out


def own_columns16(pattern):
    """Column indices of a pattern that qualifies for ``fuse_own`` (standard tiles) as 16-bit
    offsets from the first column its row tile owns (csrc/cg.hip k_cg_step1_ar C16).
    Symbolic; cached."""
    cached = getattr(pattern, "_ipx_own_col16", None)
    if cached is None:
        own = fuse_own(pattern)[0].cpu().numpy().astype(np.int64)
        nt = pattern.ntiles
        t = pattern.tiles_h
        s, e = t[nt + 1:2 * nt + 1].astype(np.int64), t[nt + 2:2 * nt + 2].astype(np.int64)
        off = pattern.indices_h.astype(np.int64) - np.repeat(own[:nt], e - s)
        assert off.size == 0 or (off.min() >= 0 and off.max() < 65536)
        cached = pattern._ipx_own_col16 = torch.from_numpy(
            off.astype(np.uint16).view(np.int16)).to(ctx().device)
    return cached


def fuse_vown(At_pattern, rows_per_wg, nwg, k=1):
    """Variables owned by each workgroup of the single-launch decoupled solve, for the
    fused g = r - A'v tail (csrc/banded.hip AtvJob): ``(table, qv, row_rel)`` or None
    (``row_rel``, host array: every variable's first constraint as an offset from its owner's
    first row -- the index of the ELL(2) form, ``ell_rows``).  A variable
    belongs to the workgroup whose constraint rows contain the first constraint that
    touches it.  Qualifies when every variable sees at most ``k + 1`` constraints, all within
    ``k`` rows of the first (``k`` = half bandwidth of A A', <= 4: the kernel keeps the
    solution of its own rows and ``k`` rows either side in LDS), the owners are
    non-decreasing along the variables, and no workgroup gets more than 4096 of them.
    Symbolic; cached per geometry."""
    key = ("_ipx_fuse_vown", int(rows_per_wg), int(nwg), int(k))
    cache = getattr(At_pattern, "_ipx_fuse_vown", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    out = None
    n = At_pattern.shape[0]
    ip, idx = At_pattern.indptr_h.astype(np.int64), At_pattern.indices_h.astype(np.int64)
    lens = np.diff(ip)
    if At_pattern.nnz > 0 and lens.max() <= k + 1 and 1 <= k <= 4:
        nonempty = lens > 0
        first = np.full(n, -1, dtype=np.int64)
        last = first.copy()
        starts = ip[:-1][nonempty]
        first[nonempty] = np.minimum.reduceat(idx, starts)      # entries in any order
        last[nonempty] = np.maximum.reduceat(idx, starts)
        if np.all(last - first <= k):
            # rows without entries go with the next variable that has some
            pos = np.where(nonempty, np.arange(n), n)
            nxt_idx = np.minimum.accumulate(pos[::-1])[::-1]
            filled = np.where(nxt_idx < n, first[np.minimum(nxt_idx, n - 1)], idx.max())
            owner = filled // rows_per_wg
            if np.all(np.diff(owner) >= 0) and owner.max() < nwg:
                vown = np.searchsorted(owner, np.arange(nwg + 1), side="left")
                vmax = int(np.max(np.diff(vown)))
                qv = (vmax + 191) // 192          # 192 lanes per workgroup work on the tail
                if 0 < qv <= 16:
                    out = (torch.from_numpy(vown.astype(np.int32)).to(ctx().device), qv,
                           filled - owner * rows_per_wg)
    At_pattern._ipx_fuse_vown = (key, out)
    return out


def ell_rows(At, row_rel):
    """The rows of ``At`` (every one with at most two entries, in adjacent columns: checked by
    ``fuse_vown`` with k = 1, whose ``row_rel`` this takes) in ELL(2) form for the tail of the
    cyclic-reduction solve: ``(row16, val)`` with ``val[t * n + j]`` = the entry of variable j
    in its first constraint (t = 0) / in the row after it (t = 1), 0 where absent, and
    ``row16[j]`` = that first constraint as a 16-bit offset from the first row of the owning
    workgroup (csrc/banded.hip reads pairs of variables with 16- and 4-byte loads: n must be
    even -- asserted by the caller -- and the buffers 16-byte aligned, which torch's are).
    18 bytes per variable instead of the 24 of two full column indices.  The row table and the
    gather map are symbolic (cached on the pattern); the values are one gather per call."""
    done = getattr(At, "_ipx_ell_done", None)        # same matrix object: same values
    if done is not None and done[0] is row_rel:
        return done[1], done[2]
    pat = At.pattern
    cache = getattr(pat, "_ipx_ell2", None)
    if cache is None or cache[0] is not row_rel:
        n = pat.shape[0]
        ip = pat.indptr_h.astype(np.int64)
        idx = pat.indices_h.astype(np.int64)
        lens = np.diff(ip)
        last = max(pat.nnz - 1, 0)
        ea = np.minimum(ip[:-1], last)                       # rows without entries: any valid one
        eb = np.where(lens > 1, ip[:-1] + 1, ea)
        swap = idx[ea] > idx[eb] if pat.nnz else np.zeros(n, dtype=bool)   # (entries in any order)
        e0, e1 = np.where(swap, eb, ea), np.where(swap, ea, eb)
        src = np.concatenate((e0, e1))
        mask = np.concatenate((lens > 0, lens > 1)).astype(np.float64)
        dev = ctx().device
        cache = pat._ipx_ell2 = (row_rel,
                                 torch.from_numpy(row_rel.astype(np.uint16).view(np.int16)).to(dev),
                                 torch.from_numpy(src.astype(np.int32)).to(dev),
                                 torch.from_numpy(mask).to(dev))
    _, row16, src, mask = cache
    val = torch.empty(src.numel(), dtype=torch.float64, device=row16.device)
    _hip.call("ipx_gather", src.numel(), _p(At.val), _p(src), _p(mask), None, _p(val), stream_ptr())
    At._ipx_ell_done = (row_rel, row16, val)
    return row16, val


_RES_LIMITS = None


def resident_limits():
    """The budgets of csrc/resident.hip, asked of the library (``ipx_cg_resident_limits``): a
    dict with max_wg (workgroups of one launch), threads, span / own variables / window rows per
    workgroup, entries per row of A / of H, halo entries either side."""
    global _RES_LIMITS
    if _RES_LIMITS is None:
        out = (ctypes.c_int32 * 8)()
        _hip.load().ipx_cg_resident_limits(out)
        _RES_LIMITS = dict(zip(("max_wg", "threads", "span", "own", "rows", "row_A", "row_H", "halo"),
                               (int(v) for v in out)))
    return _RES_LIMITS


def fuse_project(pattern, vown_h, rows_wg, nwg, H):
    """Window tables of the resident loop kernel (csrc/resident.hip) for a Jacobian
    pattern and the geometry of the cyclic-reduction solve, or None: every row has the same
    number ``rl`` of entries; per entry its column as a 16-bit offset from the row's first
    column; per workgroup of the solve the span of columns it needs (its window's rows -- own
    rows + H either side -- and its own variables, ``vown_h``); all within the kernel's
    per-lane budgets.  Symbolic; cached per geometry."""
    key = ("_ipx_project", int(rows_wg), int(nwg), int(H))
    cache = getattr(pattern, "_ipx_project", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    out = None
    m, n = pattern.shape
    ip, idx = pattern.indptr_h.astype(np.int64), pattern.indices_h.astype(np.int64)
    lens = np.diff(ip)
    if m > 0 and pattern.nnz > 0 and lens.min() == lens.max() and lens[0] >= 1:
        rl = int(lens[0])
        rows = idx.reshape(m, rl)
        first, last = rows.min(axis=1), rows.max(axis=1)
        off = rows - first[:, None]
        R = rows_wg + 2 * H
        lim = resident_limits()
        if off.max() < 65536 and rl <= lim["row_A"] and R <= lim["rows"]:
            b = np.arange(nwg, dtype=np.int64)
            rlo = np.maximum(b * rows_wg - H, 0)
            rhi = np.minimum((b + 1) * rows_wg + H, m)
            ok = np.all(rhi > rlo)
            if ok:
                # (running min / max: rows need not sweep the columns monotonically)
                pairs = np.stack((rlo, rhi), 1).ravel()
                cmin = np.minimum.reduceat(np.append(first, n), pairs)[::2]
                cmax = np.maximum.reduceat(np.append(last, 0), pairs)[::2]
                c_lo = np.minimum(cmin, vown_h[:-1])
                c_hi = np.maximum(cmax + 1, vown_h[1:])
                nspan = int(np.max(c_hi - c_lo))
                avn = int(np.max(np.diff(vown_h)))
                # a workgroup's span may reach into its two neighbours' variables only (the
                # halos travel between neighbours): hw = the longest halo, either side
                nl, nr = vown_h[:-1] - c_lo, c_hi - vown_h[1:]
                near = np.all(c_lo[1:] >= vown_h[:-2]) and np.all(c_hi[:-1] <= vown_h[2:])
                hw = int(max(nl.max(), nr.max(), 1))
                # (the launch's workgroup count is the library's check: ipx_cg_resident_ok; a
                # sharded rank launches its own blocks only)
                if nspan <= lim["span"] and 0 < avn <= lim["own"] and near and hw <= lim["halo"]:
                    dev = ctx().device
                    win = np.stack((c_lo, c_hi), 1).ravel().astype(np.int32)
                    # (inner boundaries: the smallest halo, which must cover the Hessian's)
                    inner = int(min(nl[1:].min(), nr[:-1].min())) if nwg > 1 else 1 << 30
                    out = (torch.from_numpy(off.ravel().astype(np.uint16).view(np.int16)).to(dev),
                           torch.from_numpy(first.astype(np.int32)).to(dev), rl,
                           torch.from_numpy(win).to(dev), nspan, avn, hw, inner)
    pattern._ipx_project = (key, out)
    return out


def _solver_kind(solver):
    """0: banded handle, 1: box-Schur argument block, None: not usable here."""
    from .projector import BandedNormalSolver
    from .boxschur import BoxSchurNormalSolver
    if isinstance(solver, BandedNormalSolver):
        return 0 if solver.perm is None else None
    if isinstance(solver, BoxSchurNormalSolver):
        return 1 if solver.c_args() is not None else None
    return None


def supports(H, Z, Y):
    from .projector import NormalEquationProjector
    P = getattr(Z, "projector", None)
    if P is None or getattr(Y, "projector", None) is not P:
        return False
    if not isinstance(P, NormalEquationProjector):
        return False
    from .dense import DeviceDense, DenseNormalSolver
    if isinstance(P.A, DeviceDense):          # dense Jacobian (config 2): dense or CSR Hessian
        return (P.m > 0 and isinstance(P.solver, DenseNormalSolver)
                and not getattr(P.solver, "refine_steps", 0)
                and (_dense_hessian(H) is not None or _hessian_parts(H) is not None))
    if not isinstance(P.A, DeviceCSR):
        return False
    if P.m == 0 or _solver_kind(P.solver) is None:
        return False
    # a CSR (+ diagonal) Hessian rides inside the loop's launches; any other operator with
    # ``dot`` over device vectors (finite differences, user callbacks, dense or padded terms:
    # _canonical_constraint.py:119-139) is applied between two iterations, the scalar branches
    # stay on the device all the same
    return _hessian_parts(H) is not None or hasattr(H, "dot")


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_POOL = {}          # signature -> idle _Loop (buffers + argument block), see _loop_for
POOL_STATS = {"built": 0, "reused": 0}


def _signature(H, P, lb, ub):
    """What a pooled loop object must share with a new call to be reused: the sparsity
    patterns (tiles, fused-kernel tables and every buffer size follow from them), the kind of
    solver and which optional operands exist.  None: this call is not pooled."""
    from .dense import DeviceDense
    if isinstance(P.A, DeviceDense):
        return None
    if _hessian_parts(H) is None:
        return None                      # operator Hessians: not pooled
    Hc, Hd = _hessian_parts(H)
    flags = os.environ.get("IPX_DEBUG_FORMS", "")
    return (id(Hc.pattern), Hd is None, id(P.A.pattern), lb is None, ub is None,
            _solver_kind(P.solver), int(getattr(P.solver, "k", 0)), flags)


def _loop_for(H, P, lb, ub):
    """A loop object for this call: an idle one built for the same patterns with its value
    pointers re-bound (a projected_cg call otherwise allocates ~15 device buffers and rebuilds
    the argument block: ~0.15 ms, as much as the kernels of a short solve), else a new one."""
    key = _signature(H, P, lb, ub)
    L = _POOL.pop(key, None) if key is not None else None
    if L is not None and L.rebind(H, P, lb, ub):
        POOL_STATS["reused"] += 1
        L.enqueued = None
        if key in _NO_RESIDENT:
            L.args.resident = 0
        return L, key
    POOL_STATS["built"] += 1
    L = _Loop(H, P, lb, ub, resident=False if key in _NO_RESIDENT else None)
    L.enqueued = None            # (the batch ipx_cg_prime enqueued behind the priming, if any)
    return L, key


def _release(L, key):
    """Park the loop object for the next call on the same patterns; the iterate it returned
    stays the caller's (a fresh buffer takes its place)."""
    if key is None:
        return
    if len(_POOL) >= 4:
        _POOL.pop(next(iter(_POOL)))
    _POOL[key] = L


class _Loop:
    """Buffers + argument block for one projected_cg call."""

    def rebind(self, H, P, lb, ub):
        """Point the argument block at the values of a new call on the same patterns.  False
        when the new factorization takes another solve path than the one the block was laid
        out for (the decoupling is numerical): the caller then builds a new object."""
        lib = _hip.load()
        a = self.args
        A = P.A
        At = A.T
        Hc, Hd = _hessian_parts(H)
        if a.solver_kind == 0:
            geo = (ctypes.c_int32 * 2)()
            ok = bool(lib.ipx_banded_decoupled_geometry(ctypes.c_void_p(P.solver.handle), geo))
            if (ok, geo[0], geo[1]) != self.geometry:
                return False
            if self.pcr_L is not None and \
                    int(lib.ipx_banded_pcr_level(ctypes.c_void_p(P.solver.handle))) != self.pcr_L:
                return False            # the resident kernel's windows follow 2^L
            a.banded = ctypes.c_void_p(P.solver.handle)
        else:
            a.banded = ctypes.cast(ctypes.pointer(P.solver.c_args()), ctypes.c_void_p)
        a.A_val, a.At_val, a.H_val = _ptr(A.val), _ptr(At.val), _ptr(Hc.val)
        a.H_diag = _ptr(Hd.t) if Hd is not None else None
        a.lb = _ptr(lb.t) if lb is not None else None
        a.ub = _ptr(ub.t) if ub is not None else None
        if a.At_ell_val:
            self.ell_row, self.ell_val = ell_rows(At, self.row_rel)
            a.At_ell_row, a.At_ell_val = _ptr(self.ell_row), _ptr(self.ell_val)
        a.no_radius = 0
        self.x = torch.empty(self.n, dtype=torch.float64, device=self.state.device)
        a.x = _ptr(self.x)
        self.keep = (A, At, Hc, Hd, lb, ub, P)
        return True

    def __init__(self, H, P, lb, ub, resident=None):
        from .dense import DeviceDense
        self.geometry, self.pcr_L, self.operator = None, None, None
        if isinstance(P.A, DeviceDense):
            self._init_dense(H, P, lb, ub)
            return
        lib = _hip.load()
        A = P.A
        At = A.T
        parts = _hessian_parts(H)
        self.operator = None if parts is not None else H
        Hc, Hd = parts if parts is not None else (None, None)
        self.n, self.m = P.n, P.m
        n, m = self.n, self.m
        dev = ctx().device
        f64 = torch.float64
        self.x = torch.empty(n, dtype=f64, device=dev)
        self.p = torch.empty(n, dtype=f64, device=dev)
        self.r = torch.empty(n, dtype=f64, device=dev)
        self.Hp = torch.empty(n, dtype=f64, device=dev)
        self.w = torch.empty(m, dtype=f64, device=dev)
        self.v = torch.empty(m, dtype=f64, device=dev)
        self.t = torch.empty(m, dtype=f64, device=dev)
        self.state = torch.zeros(lib.ipx_cg_state_size(), dtype=f64, device=dev)
        grid = lib.ipx_cg_vec_grid(n)
        self.part1 = torch.zeros(2 * (Hc.pattern.ntiles if Hc is not None else 1), dtype=f64,
                                 device=dev)
        # (box-Schur projection: step1 writes one partial per >= 1280 elements, csrc/cg.hip SB_RMIN)
        self.part2 = torch.zeros(2 * max(grid, A.pattern.ntiles, n // 1024 + 2), dtype=f64,
                                 device=dev)
        self.part3 = torch.zeros(2 * max(At.pattern.ntiles, (m + 255) // 256 + 1, n // 1024 + 2),
                                 dtype=f64, device=dev)
        self.part4 = torch.zeros((m + 255) // 256 + 1, dtype=f64, device=dev)   # ||w-(AA')v||^2 partials
        self.keep = (A, At, Hc, Hd, lb, ub, P)
        a = CgArgs()
        a.n, a.m = n, m
        for pre, M in (("A", A), ("At", At), ("H", Hc)):
            if M is None:                 # operator Hessian: applied by the host (H_operator)
                a.H_ntiles, a.H_operator = 1, 1
                continue
            pat = M.pattern
            setattr(a, pre + "_rowptr", _ptr(pat.indptr))
            setattr(a, pre + "_colidx", _ptr(pat.indices))
            setattr(a, pre + "_val", _ptr(M.val))
            setattr(a, pre + "_tiles", _ptr(pat.tiles))
            setattr(a, pre + "_ntiles", pat.ntiles)
        a.H_diag = _ptr(Hd.t) if Hd is not None else None
        a.solver_kind = _solver_kind(P.solver)
        if a.solver_kind == 1:
            a.banded = ctypes.cast(ctypes.pointer(P.solver.c_args()), ctypes.c_void_p)
        else:
            a.banded = ctypes.c_void_p(P.solver.handle)
        for name in ("x", "p", "r", "Hp", "w", "v", "t", "state",
                     "part1", "part2", "part3", "part4"):
            setattr(a, name, _ptr(getattr(self, name)))
        a.lb = _ptr(lb.t) if lb is not None else None
        a.ub = _ptr(ub.t) if ub is not None else None
        a.vec_grid = grid
        self.fold_ws = torch.zeros(16384, dtype=f64, device=dev)      # IPX_FOLD_WS_DOUBLES
        a.fold_ws = _ptr(self.fold_ws)
        # banded Hessian: step2 rides inside the H.p SpMV (one launch less per iteration)
        no_fuse = _hip.debug_form("no-fuse")
        hmax = 0 if (no_fuse or Hc is None) else fuse_halo(Hc.pattern)
        if hmax > 0:
            self.pb = torch.zeros(2 * Hc.pattern.ntiles * 2 * hmax, dtype=f64, device=dev)
            a.pb, a.H_hmax = _ptr(self.pb), hmax
            th = Hc.pattern.tiles_h
            a.H_tile_rows = int(np.max(np.diff(th[:Hc.pattern.ntiles + 1])))
            self.H_col16, self.H_rowlen = compact_columns(Hc.pattern, hmax)
            a.H_col16, a.H_rowlen = _ptr(self.H_col16), _ptr(self.H_rowlen)
        # banded Jacobian, no box: step1 rides inside the A.r SpMV
        own = None if (no_fuse or lb is not None or m == 0) else fuse_own(A.pattern)
        if own is not None:
            self.r_next = torch.empty(n, dtype=f64, device=dev)
            self.own, self.own_tiles = own[0], own[2]
            a.r_next, a.A_own, a.A_span = _ptr(self.r_next), _ptr(self.own), own[1]
            a.A_tiles, a.A_ntiles, a.A_tile_nnz = _ptr(self.own_tiles), own[3], 0
            if self.part2.numel() < 2 * own[3]:
                self.part2 = torch.zeros(2 * own[3], dtype=f64, device=dev)
                a.part2 = _ptr(self.part2)
            self.A_col16 = own_columns16(A.pattern)
            a.A_col16 = _ptr(self.A_col16)
        # tridiagonal A A' on the single-launch solve: g = r - A'v rides in that launch
        if a.solver_kind == 0 and not no_fuse:
            geo = (ctypes.c_int32 * 2)()
            ok = bool(lib.ipx_banded_decoupled_geometry(ctypes.c_void_p(P.solver.handle), geo))
            self.geometry = (ok, geo[0], geo[1])
            if ok:
                kS = int(getattr(P.solver, "k", 1))
                vown = fuse_vown(At.pattern, geo[0], geo[1], kS) if kS <= 4 else None
                if vown is not None:
                    self.vown = vown[0]
                    a.At_vown, a.At_qv = _ptr(self.vown), vown[1]
                    # tridiagonal A A': ELL(2) rows with one 16-bit row offset per variable, pairs of
                    # variables per 16-byte load
                    if n % 2 == 0 and kS == 1:
                        self.row_rel = vown[2]
                        self.ell_row, self.ell_val = ell_rows(At, self.row_rel)
                        a.At_ell_row, a.At_ell_val = _ptr(self.ell_row), _ptr(self.ell_val)
                    # small problems (one CU per workgroup of the cyclic-reduction solve: the
                    # per-rank sizes of a multi-GPU run), uniform rows, no box, short Hessian
                    # rows: a whole batch of iterations as ONE resident launch
                    # (csrc/resident.hip) -- 19.6 -> 16.0 us per iteration at n = 1.25e5
                    # (profiles/r04_per_rank_sweep.json).  resident=False: the separate launches.
                    L_pcr = int(lib.ipx_banded_pcr_level(ctypes.c_void_p(P.solver.handle)))
                    if resident is None and _hip.debug_form("no-resident"):
                        resident = False
                    if kS == 1 and L_pcr > 0 and lb is None and hmax > 0 and resident is not False \
                            and int(np.max(np.diff(Hc.pattern.indptr_h))) <= resident_limits()["row_H"]:
                        pj = fuse_project(A.pattern, self.vown.cpu().numpy().astype(np.int64),
                                          geo[0], geo[1], 1 << L_pcr)
                        if pj is not None and hmax <= pj[7]:
                            self.proj_tabs = pj
                            a.A_off16, a.A_rowfirst, a.A_rl = _ptr(pj[0]), _ptr(pj[1]), pj[2]
                            a.P_win, a.P_nspan, a.P_navn = _ptr(pj[3]), pj[4], pj[5]
                            self.pcr_L = L_pcr
                            words = int(lib.ipx_cg_resident_ll_words(geo[1], pj[6]))
                            self.ll = torch.zeros(words, dtype=torch.int64, device=dev)
                            self.ll_seq = ctypes.c_int64(0)
                            a.R_ll, a.R_hw = _ptr(self.ll), pj[6]
                            a.R_seq = ctypes.cast(ctypes.pointer(self.ll_seq), ctypes.c_void_p)
                            a.resident = 1
                            if not lib.ipx_cg_resident_ok(ctypes.byref(a)):
                                a.resident = 0
        self.args = a

    def _init_dense(self, H, P, lb, ub):
        """Argument block for a dense Jacobian (csrc/cg.hip cg_iterate_dense)."""
        from .dense import DeviceDense
        lib = _hip.load()
        A, At = P.A, P.A.T
        self.n, self.m = n, m = P.n, P.m
        M = P.solver.M
        dev, f64 = ctx().device, torch.float64
        z = lambda k: torch.zeros(int(k), dtype=f64, device=dev)
        self.x, self.p, self.r, self.Hp = z(n), z(n), z(n), z(n)
        self.w, self.v, self.t = z(M), z(M), z(m)
        self.state = z(lib.ipx_cg_state_size())
        grid = lib.ipx_cg_vec_grid(n)
        self.part1, self.part3, self.part4 = z(2 * 2048), z(2 * 2048), z(2 * 2048)
        self.part2 = z(2 * grid)
        a = CgArgs()
        a.n, a.m = n, m
        a.A_val, a.At_val = _ptr(A.t), _ptr(At.t)
        Hd_ = _dense_hessian(H)
        if Hd_ is not None:
            a.H_val = _ptr(Hd_.t)
            self.keep = (A, At, Hd_, lb, ub, P)
        else:
            Hc, Hd = _hessian_parts(H)
            pat = Hc.pattern
            a.H_rowptr, a.H_colidx, a.H_val = _ptr(pat.indptr), _ptr(pat.indices), _ptr(Hc.val)
            a.H_tiles, a.H_ntiles = _ptr(pat.tiles), pat.ntiles
            a.H_diag = _ptr(Hd.t) if Hd is not None else None
            self.part1 = z(2 * max(pat.ntiles, 1))
            self.keep = (A, At, Hc, Hd, lb, ub, P)
        a.banded = ctypes.c_void_p(P.solver.Ginv.t.data_ptr())
        a.solver_kind = 2
        for name in ("x", "p", "r", "Hp", "w", "v", "t", "state",
                     "part1", "part2", "part3", "part4"):
            setattr(a, name, _ptr(getattr(self, name)))
        a.lb = _ptr(lb.t) if lb is not None else None
        a.ub = _ptr(ub.t) if ub is not None else None
        a.vec_grid = grid
        self.fold_ws = None
        self.args = a

    def ref(self):
        return ctypes.byref(self.args)

    def apply_operator(self):
        """Operator Hessians: Hp = H p by the operator's own ``dot`` and p'Hp into the slot the
        next step1 folds (part1[1]); enqueued, no host synchronisation of its own."""
        Hp = self.operator.dot(DVec(self.p))
        self.Hp.copy_(Hp.t)
        _hip.call("ipx_dot", self.n, _p(self.p), _p(self.Hp),
                  ctypes.c_void_p(self.part1.data_ptr() + 8), _p(ctx().ws), stream_ptr())

    def g_tensor(self, it):
        """The buffer that holds g (= the next r) after iteration ``it``."""
        return self.r


class _PrimeRetry(Exception):
    """The device found that the call's priming needs the host (stop code 9)."""


class _ResidentGaveUp(Exception):
    """A resident launch timed out (stop code 8): the call starts over on the separate
    launches (the loop object has ``resident`` cleared)."""


_NO_RESIDENT = set()       # pool signatures whose resident launches timed out once
_INJECT_RESIDENT_TIMEOUT = []      # tests append one item: the next resident batch "times out"


def projected_cg(H, c, Z, Y, b, trust_radius=np.inf, lb=None, ub=None, tol=None,
                 max_iter=None, max_infeasible_iter=None, batch=None, stats=None, b_zero=False):
    """qp_subproblem.py:416-643 on the device-resident loop.  The call first tries a priming
    that reads nothing back (``_prime_without_reads``): the scalars of :502-542 -- rt_g, the
    default tolerance, the distance to the trust-region boundary, the orthogonality measures of
    the two initial projections -- are reduced and tested on the device, which writes the
    loop's state block itself; the first host read of the call is the state block after the
    first batch of iterations.  When the device finds that a projection needs a refinement or
    cancellation step, or that the start sits on the trust-region boundary, it says so in that
    block (stop code 9) and the call starts over on the host-driven priming below."""
    P = Z.projector
    from .projector import NormalEquationProjector
    # Near convergence Z c is tiny next to c and EVERY call needs the cancellation step: after a
    # call whose device priming was turned down, calls on the same problem (same patterns: the
    # loop pool's signature) go to the host priming directly -- until one of them gets by
    # without refinement or cancellation steps, which the host priming counts anyway.
    try:
        return _projected_cg_once(H, c, Z, Y, b, trust_radius, lb, ub, tol, max_iter,
                                  max_infeasible_iter, batch, stats, b_zero)
    except _ResidentGaveUp:
        # (once per pattern: the retry and every later call run the separate launches)
        return _projected_cg_once(H, c, Z, Y, b, trust_radius, lb, ub, tol, max_iter,
                                  max_infeasible_iter, batch, stats, b_zero)


def _projected_cg_once(H, c, Z, Y, b, trust_radius, lb, ub, tol, max_iter, max_infeasible_iter,
                       batch, stats, b_zero):
    P = Z.projector
    from .projector import NormalEquationProjector
    key = _signature(H, P, lb, ub) if isinstance(P, NormalEquationProjector) else None
    if isinstance(P, NormalEquationProjector) and P.m > 0 and len(c) - len(b) >= 1 \
            and (max_iter is None or max_iter >= 1) and trust_radius >= 0 \
            and not (key is not None and key in _HOST_PRIMED):
        try:
            return _projected_cg(H, c, Z, Y, b, trust_radius, lb, ub, tol, max_iter,
                                 max_infeasible_iter, batch, stats, b_zero, fast=True)
        except _PrimeRetry:
            STATS["prime_retries"] += 1
            if b_zero and key is not None and key not in _STEP_PRIMED:
                # once more on the device, with the projections' correction steps armed (the
                # same arithmetic as the host's steps; what they do not settle is the host's)
                if len(_STEP_PRIMED) >= 16:
                    _STEP_PRIMED.clear()
                _STEP_PRIMED.add(key)
                try:
                    return _projected_cg(H, c, Z, Y, b, trust_radius, lb, ub, tol, max_iter,
                                         max_infeasible_iter, batch, stats, b_zero, fast=True)
                except _PrimeRetry:
                    STATS["prime_retries"] += 1
            if key is not None:
                if len(_HOST_PRIMED) >= 16:
                    _HOST_PRIMED.clear()
                _HOST_PRIMED.add(key)
    steps = None
    if key is not None and key in _HOST_PRIMED:
        steps = P.stats["refinements"] + P.stats["cancellation_steps"]
        STATS["host_primed_directly"] += 1
    out = _projected_cg(H, c, Z, Y, b, trust_radius, lb, ub, tol, max_iter,
                        max_infeasible_iter, batch, stats, b_zero, fast=False)
    if steps is not None and P.stats["refinements"] + P.stats["cancellation_steps"] == steps:
        _HOST_PRIMED.discard(key)        # (a clean priming: the next call tries the device again)
    return out


_HOST_PRIMED = set()       # pool signatures whose last call needed the host's priming
_STEP_PRIMED = set()       # ... whose primings carry the projections' correction steps (device)


_PRIME_IDX = (ctypes.c_int32 * 7)(12, 4, 0, 2, 10, 6, 8)
_PRIME_IDX_B0 = (ctypes.c_int32 * 7)(-1, 4, 0, 2, 10, 6, 8)


def _prime_without_reads(L, H, c, Z, Y, b, b_zero, P, tol, trust_radius):
    """x0, r0 = Z(H x0 + c), g0 = Z r0 enqueued with their norms left in the context's reduction
    block (doubles 0..13), then the state block written by ``ipx_cg_prime_state``."""
    n = len(c)
    ctx_ = ctx()
    if b_zero:
        x0 = DVec.zeros(n)
        t = c
    else:
        x0 = Y.dot(-b)
        _hip.call("ipx_norms", n, _p(x0.t), ctypes.c_void_p(ctx_.out.data_ptr() + 8 * 12),
                  _p(ctx_.ws), stream_ptr())
        t = H.dot(x0) + c
    r0 = P.null_space_enqueue(t, 0)          # ||t||^2 -> 4, ||r0||^2 -> 0, ||A r0||^2 -> 2
    g0 = P.null_space_enqueue(r0, 6)         # ||r0||^2 -> 10, ||g0||^2 -> 6, ||A g0||^2 -> 8
    _hip.call("ipx_cg_prime_state", _p(L.state), _p(ctx_.out),
              _PRIME_IDX_B0 if b_zero else _PRIME_IDX,
              float("nan") if tol is None else float(tol), float(trust_radius),
              float(P.orth_tol), float(P.norm_A), float(P.CANCELLATION), stream_ptr())
    return x0, r0, g0


def _projected_cg(H, c, Z, Y, b, trust_radius, lb, ub, tol, max_iter, max_infeasible_iter,
                  batch, stats, b_zero, fast):
    from . import qp
    lib = _hip.load()
    P = Z.projector
    n, m = len(c), len(b)
    has_box = lb is not None or ub is not None
    ub_given = ub is not None
    if has_box:
        lb = lb if lb is not None else DVec.full(n, -np.inf)
        ub = ub if ub is not None else DVec.full(n, np.inf)
    if max_iter is None:
        max_iter = n - m
    max_iter = min(max_iter, n - m)
    if max_infeasible_iter is None:
        max_infeasible_iter = n - m
    if fast:
        L, pool_key = _loop_for(H, P, lb if has_box else None,
                                ub if has_box and ub_given else None)
        if L.operator is not None:            # (an operator Hessian is applied by the host)
            _release(L, pool_key)
            raise _PrimeRetry()
        if np.isinf(trust_radius) and trust_radius > 0 and not has_box \
                and not _hip.debug_form("keep-xn2"):
            L.args.no_radius = 1
        st = stream_ptr()
        a = L.args
        if isinstance(P.A, DeviceCSR) and a.solver_kind in (0, 1) and getattr(P.solver, "perm", None) is None \
                and a.banded and not getattr(P.solver, "refine_steps", 0) \
                and lib.ipx_cg_prime_ws_doubles(L.ref(), P.A.pattern.ntiles) <= 65536:   # (IPX_WS_DOUBLES)
            # the whole priming behind one C call, into the loop's own buffers
            ctx_ = ctx()
            pat = P.A.pattern
            # ... and the call's first batch behind it (the host's way from here to its own
            # ipx_cg_iterate call is ~50 us of idle GPU otherwise)
            first_end = min(max_iter, batch if batch else _first_batch(max_iter, False))
            b_rows = None if b_zero else P.rows_in(b)      # (the projector's row order)
            # (correction steps of the two projections on the device when the last primings on
            # these patterns needed them or came close: _STEP_PRIMED)
            L.stepped = bool(b_zero and pool_key is not None and pool_key in _STEP_PRIMED)
            _hip.call("ipx_cg_prime", L.ref(), _p(pat.tiles), pat.ntiles, _p(c.t),
                      None if b_zero else _p(b_rows.t), _p(ctx_.out), _p(ctx_.ws),
                      float("nan") if tol is None else float(tol), float(trust_radius),
                      float(P.orth_tol), float(P.norm_A), float(P.CANCELLATION),
                      max(first_end, 0), 1 if L.stepped else 0, st)
            L.enqueued = (0, first_end) if first_end > 0 else None
            P.stats["solves"] += 2 if b_zero else 3
        else:
            x0, r0, g0 = _prime_without_reads(L, H, c, Z, Y, b, b_zero, P, tol, trust_radius)
            L.x.copy_(x0.t)
            L.r.copy_(r0.t)
            _hip.call("ipx_axpby", n, -1.0, _p(g0.t), 0.0, None, _p(L.p), st)      # p = -g
            _hip.check(lib.ipx_cg_hp(L.ref(), st), "ipx_cg_hp")
        STATS["primed_on_device"] += 1
        return _run_loop(L, pool_key, P, lib, st, n, lb, ub, trust_radius, max_iter,
                         max_infeasible_iter, batch, stats, fast=True)

    # ---- initial point, residual, direction (qp_subproblem.py:502-512)
    if b_zero:
        # b = 0 by construction (the SQP's call): x0 = Y.dot(-0) = 0, H.dot(0) + c = c
        x0 = DVec.zeros(n)
        r0 = Z.dot(c)
        norm_x0 = 0.0
    else:
        x0 = Y.dot(-b)
        r0 = Z.dot(H.dot(x0) + c)
        norm_x0 = None
    g0 = Z.dot(r0)
    # norm(g)**2: the projection's last pass left ||g0||^2 in its reduction block, read with the
    # orthogonality measure -- the number the device priming starts from (same kernel, same
    # fold), so the two primings start the loop from the same bits, and one read less
    red = getattr(P, "_red", None)
    rt_g = float(red[0]) if red is not None and P.m > 0 else g0.sumsq_amax()[0]
    tr_distance = trust_radius - (dv.norm(x0) if norm_x0 is None else norm_x0)
    if tr_distance < 0:
        raise ValueError("Trust region problem does not have a solution.")
    if tr_distance < _TINY:
        return x0, {'niter': 0, 'stop_cond': 2, 'hits_boundary': True}

    if tol is None:                       # :529-542
        tol = max(min(0.01 * np.sqrt(rt_g), 0.1 * rt_g), _TINY)

    # (no upper bounds given: the loop's kernels do not read a vector of +inf)
    L, pool_key = _loop_for(H, P, lb if has_box else None, ub if has_box and ub_given else None)
    # Unbounded trust region and no box (the reference's default trust_radius=np.inf): the
    # test norm(x_next) >= trust_radius of qp_subproblem.py:583 is always False, so the norm
    # is not formed (the fused step1 + A.r kernel then reads neither x nor p)
    if np.isinf(trust_radius) and trust_radius > 0 and not has_box \
            and not _hip.debug_form("keep-xn2"):
        L.args.no_radius = 1
    st = stream_ptr()
    L.x.copy_(x0.t)
    L.r.copy_(r0.t)
    _hip.call("ipx_axpby", n, -1.0, _p(g0.t), 0.0, None, _p(L.p), st)      # p = -g
    init = np.zeros(L.state.numel())
    init[ST_RTG0] = rt_g
    init[ST_TOL] = tol
    init[ST_RADIUS] = trust_radius
    init[ST_ORTH_RHS] = P.orth_tol * P.norm_A
    L.state.copy_(torch.from_numpy(init))
    _hip.check(lib.ipx_cg_hp(L.ref(), st), "ipx_cg_hp")
    if L.operator is not None:
        L.apply_operator()
        STATS["operator_calls"] += 1
    return _run_loop(L, pool_key, P, lib, st, n, lb, ub, trust_radius, max_iter,
                     max_infeasible_iter, batch, stats, fast=False)


def _first_batch(max_iter, operator):
    """Iterations of a call's first batch: a caller that asks for at most a few dozen gets them
    in ONE batch (one state read for the call); the open-ended calls of the SQP start with
    four (two with an operator Hessian, which the host applies once per enqueued iteration)."""
    if operator:
        return 2
    return max_iter if 4 < max_iter <= 32 else 4


def _run_loop(L, pool_key, P, lib, st, n, lb, ub, trust_radius, max_iter, max_infeasible_iter,
              batch, stats, fast, primed_state=None, release=True, first_batch=None):
    """``primed_state``: the state block as somebody already read it behind the batch recorded
    in ``L.enqueued`` (the outer iteration's chain, sqp_chain.py: its one read carries the
    loop's block) -- the first ``read_state`` then costs nothing.  ``release=False``: the loop
    object stays the caller's."""
    pending = [primed_state]

    _fb = first_batch if first_batch else _first_batch(max_iter, L.operator is not None)

    class Driver:
        """The single-GPU loop behind ``run_device_loop``."""
        first_batch = _fb
        batch_cap = 64 if L.operator is None else 8   # (an operator is applied once per
                                                      #  enqueued iteration, stopped or not)
        def iterate(self, it, end):
            self.last = (it, end)
            if getattr(L, "enqueued", None) is not None:
                done, L.enqueued = L.enqueued, None
                if done == (it, end):        # (enqueued with the priming: ipx_cg_prime)
                    return
                raise _hip.IpxError("device loop: the batch enqueued with the priming %r is not "
                                    "the one the driver asks for %r" % (done, (it, end)))
            if L.operator is None:
                _hip.check(lib.ipx_cg_iterate(L.ref(), it, end, st), "ipx_cg_iterate")
            else:
                # the operator is applied between the iterations (after a stop: on unchanged
                # p, harmless); the branches of the iterations were still taken on the device
                for k in range(it, end):
                    _hip.check(lib.ipx_cg_iterate(L.ref(), k, k + 1, st), "ipx_cg_iterate")
                    L.apply_operator()

        def read_state(self):
            if pending[0] is not None:
                s, pending[0] = list(pending[0]), None
            else:
                s = dv.read_doubles(L.state, L.state.numel())
            if fast and int(s[ST_STOP]) == 9:
                # the device's verdict on the priming: the host must do it (nothing of the
                # call's inputs was overwritten; the loop's launches were no-ops)
                if release:
                    _release(L, pool_key)
                raise _PrimeRetry()
            if fast and pool_key is not None and getattr(L, "stepped", None) is not None:
                # arm / disarm the device's correction steps for the next primings on these
                # patterns: taken now, or within a factor 64 of the cancellation test
                if s[ST_PRIME_STEPS] > 0 or s[ST_MARGIN] < 64.0 * P.CANCELLATION ** 2:
                    _STEP_PRIMED.add(pool_key)
                else:
                    _STEP_PRIMED.discard(pool_key)
                L.stepped = None
            if _INJECT_RESIDENT_TIMEOUT and L.args.resident and int(s[ST_STOP]) in (0, 4):
                # (test hook, one shot: what a resident launch leaves when a late workgroup
                # commits alone -- stop code 8 and HALF of x advanced)
                _INJECT_RESIDENT_TIMEOUT.pop()
                L.x[: L.n // 2] += 1.0
                L.state[ST_STOP] = 8.0
                s = list(s)
                s[ST_STOP] = 8.0
            if int(s[ST_STOP]) == 8 and L.args.resident:
                # a hand-off of the resident launch timed out (a workgroup that never became
                # resident: the GPU shared with another process's kernels).  The workgroups
                # that saw every record of the commit hop before THEIR deadline have written
                # their part of x, p, r, Hp back, one that arrived past it has not (or the
                # other way round: a late one that then finds all records commits alone): the
                # loop's vectors cannot be trusted.  The CALL's inputs can -- the subproblem is
                # solved again from its priming, on the separate launches, and this pattern
                # stays on them (the row-sharded loop does the same on its stop code 7).
                STATS["resident_fallbacks"] += 1
                if pool_key is not None:
                    _NO_RESIDENT.add(pool_key)
                L.args.resident = 0
                L.state[ST_STOP] = 0.0
                raise _ResidentGaveUp()
            return s

        def X(self):
            return DVec(L.x)

        def Pv(self):
            return DVec(L.p)

        def set_x(self, v):
            L.x = v.t

        def zeros(self):
            return DVec.zeros(n)

        def resume(self, it_stop, mode):
            return _resume(lib, L, it_stop, mode, st)

        def refine(self, it_stop):
            _refine(P, L, DVec(L.g_tensor(it_stop)))

    x, niter, stop_cond, hits_boundary = run_device_loop(
        Driver(), STATS, lb, ub, trust_radius, max_iter, max_infeasible_iter, batch, stats)
    STATS["calls"] += 1
    STATS["iterations"] += niter
    STATS["resident_calls"] += 1 if L.args.resident else 0
    if release:
        _release(L, pool_key)
    return x, {'niter': niter, 'stop_cond': stop_cond, 'hits_boundary': hits_boundary}


def run_device_loop(D, counters, lb, ub, trust_radius, max_iter, max_infeasible_iter,
                    batch=None, stats=None):
    """The host side of the device-resident loop, ONE routine for the single-GPU and the
    row-sharded drivers (``D``: iterate / read_state / X / Pv / set_x / zeros / resume /
    refine over their own vector types).  Batches of iterations are enqueued, the state block
    is read once per batch, and the rare events are finished here with the reference's helper
    routines -- qp_subproblem.py:551 tolerance, :558-576 negative curvature, :583-596
    trust-region exit, :599-616 box-infeasible iterates with the ``counter`` /
    ``last_feasible_x`` bookkeeping, projections.py:72-78 refinement (stop codes 4, 3, 2, 5, 6
    of csrc/cg.hip).  Returns ``(x, niter, stop_cond, hits_boundary)``."""
    from . import qp
    has_box = lb is not None or ub is not None
    hits_boundary, stop_cond = False, 1
    counter, last_viol_it = 0, -2
    last_feasible_x = D.zeros() if has_box else None     # (only a box makes iterates infeasible)
    it = 0
    nbatch = batch if batch else D.first_batch
    s = None
    while it < max_iter:
        end = min(max_iter, it + nbatch)
        D.iterate(it, end)
        s = D.read_state()               # one blocking read per batch
        counters["batches"] += 1
        if stats is not None:
            stats["batches"] = stats.get("batches", 0) + 1
        stop = int(s[ST_STOP])
        if stop == 0:
            it = end
            if not batch:
                nbatch = min(2 * nbatch, D.batch_cap)
            continue
        it_stop = int(s[ST_IT_DONE])     # index of the iteration that raised the flag
        alpha = s[ST_ALPHA]
        if stop == 4:                     # :551
            stop_cond = 4
            break
        if stop == 3:                     # :558-576
            if np.isinf(trust_radius):
                raise ValueError("Negative curvature not allowed "
                                 "for unrestrited problems.")
            X, Pv = D.X(), D.Pv()
            _, al, hit = qp.box_sphere_intersections(X, Pv, lb, ub, trust_radius,
                                                     entire_line=True)
            xf = X.add_scaled(Pv, al) if hit else X
            D.set_x(qp.reinforce_box_boundaries(xf, lb, ub))
            stop_cond, hits_boundary = 3, True
            break
        if stop == 2:                     # :583-596
            X, Pv = D.X(), D.Pv()
            _, theta, hit = qp.box_sphere_intersections(X, Pv, lb, ub, trust_radius,
                                                        dscale=alpha)
            xf = X.add_scaled(Pv, theta * alpha) if hit else X
            D.set_x(qp.reinforce_box_boundaries(xf, lb, ub))
            stop_cond, hits_boundary = 2, True
            break
        mode = 0
        if stop == 5:                     # :599-616 x_next outside the box
            counters["box_events"] += 1
            if last_viol_it != it_stop - 1:
                counter = 0
            counter += 1
            last_viol_it = it_stop
            X, Pv = D.X(), D.Pv()
            _, theta, hit = qp.box_sphere_intersections(X, Pv, lb, ub, trust_radius,
                                                        dscale=alpha)
            if hit:
                last_feasible_x = qp.reinforce_box_boundaries(
                    X.add_scaled(Pv, theta * alpha), lb, ub)
                counter = 0
                last_viol_it = -2
            if counter > max_infeasible_iter:
                break
            mode = 1
            # the orthogonality check has not run yet for this iteration
            s = D.resume(it_stop, mode)
            if int(s[ST_STOP]) == 6:
                stop, mode = 6, 1
            else:
                it = it_stop + 1
                continue
        if stop == 6:                     # projections.py:72-78 refinement
            counters["refine_events"] += 1
            D.refine(it_stop)
            s = D.resume(it_stop, mode | 2)
            it = it_stop + 1
            continue
        if stop == 7:
            raise _hip.IpxError("sharded projected CG: a wait on the peer mailboxes timed out "
                                "(a rank of the group died or fell out of step; iteration %d, "
                                "wait code %s)" % (it_stop, s[ST_VIOL]))
        raise _hip.IpxError("unexpected CG stop code %d" % stop)

    x = D.X()
    if has_box and not qp.inside_box_boundaries(x, lb, ub):     # :636-638
        x = last_feasible_x
        hits_boundary = True
    # (the state block read last is current: nothing was enqueued after it on every exit)
    niter = int(s[ST_NITER]) if s is not None else 0
    return x, niter, stop_cond, hits_boundary


def _resume(lib, L, it_stop, mode, st):
    """Clear the stop flag, finish iteration ``it_stop`` (step2 + Hp)."""
    L.state[ST_STOP] = 0.0
    _hip.check(lib.ipx_cg_resume(L.ref(), it_stop, mode, st), "ipx_cg_resume")
    if L.operator is not None:
        L.apply_operator()
    return dv.read_doubles(L.state, L.state.numel())


def _refine(P, L, R):
    """Iterative refinement of g = Z r (projections.py:69-78) on the buffers of
    the fused loop: R is the view of the buffer that holds g (``_Loop.g_tensor``)."""
    Az = P.A.dot(R)          # the loop only formed ||A g||^2 (as a constraint-space residual)
    k = 0
    while k < P.max_refin:
        v = P._apply_inv(Az)
        z = P.A.rmatvec_sub(v, R, reduce=True)       # ||z||^2 -> slot 0
        R.t.copy_(z.t)
        k += 1
        P.stats["refinements"] += 1
        orth, Az = P._orthogonality(R)
        L.t.copy_(Az.t)
        if not orth > P.orth_tol:
            break
    if k > 0:
        # step2 derives beta from the ||g||^2 partials: replace them by the
        # refined value (one non-zero entry, the rest adds exact zeros)
        gg = dv.read_slots(1)[0]
        L.part3.zero_()
        L.part3[0] = gg
