"""``(A A')^-1`` with the box-like rows of ``A`` eliminated analytically
(kernels in csrc/boxschur.hip).

Barrier problems with bounds on the variables (BASELINE config 5) put two rows
``-e_j' + s_lb``, ``+e_j' + s_ub`` per bounded variable into the augmented
Jacobian.  They make ``A A'`` wide (half bandwidth ~40 for the banded
benchmark, 1e6 rows) although they only couple pairwise and through one
column.  Here such *simple* rows -- exactly one entry in a column shared with
other rows plus at most one entry in a private column -- are grouped per
shared column (one or two rows per group) and eliminated in closed form; what
remains is ``Sigma = A_R diag(w) A_R'`` over the general rows, which keeps the
band of ``A_R A_R'`` and goes to the partitioned banded solver.

The classification is symbolic (sparsity pattern only, once per pattern on the
host); all arithmetic is on the device.
"""
import ctypes
import os

import numpy as np
import torch

from . import _hip
from . import device as dv
from .device import DVec, _p, stream_ptr, ctx

_F64 = torch.float64


def _i32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(ctx().device)


class BoxRowAnalysis:
    """Pattern-level split of the rows of A into simple (grouped) and general."""

    def __init__(self, pattern):
        m, n = pattern.shape
        indptr, indices = pattern.indptr_h.astype(np.int64), pattern.indices_h
        nnz_row = np.diff(indptr)
        col_count = np.bincount(indices, minlength=n)
        private = col_count[indices] == 1                     # per nonzero
        csum = np.concatenate(([0], np.cumsum(~private)))
        shared_in_row = csum[indptr[1:]] - csum[indptr[:-1]]
        simple = (nnz_row >= 1) & (nnz_row <= 2) & (shared_in_row == 1)
        # position of the shared / private entry of every simple row
        pos_a = np.full(m, -1, dtype=np.int64)
        pos_s = np.full(m, -1, dtype=np.int64)
        rows_of_nz = np.repeat(np.arange(m), nnz_row)
        nz = np.arange(len(indices))
        sel = simple[rows_of_nz]
        pos_a[rows_of_nz[sel & ~private]] = nz[sel & ~private]
        pos_s[rows_of_nz[sel & private]] = nz[sel & private]
        shared_col = np.where(simple, indices[np.maximum(pos_a, 0)], -1)
        # groups: shared columns with one or two simple rows; more -> general rows
        srows = np.flatnonzero(simple)
        order = np.argsort(shared_col[srows], kind="stable")
        srows = srows[order]
        cols, start, count = np.unique(shared_col[srows], return_index=True, return_counts=True)
        ok = count <= 2
        demoted = np.concatenate([srows[s:s + c] for s, c in zip(start[~ok], count[~ok])]) \
            if (~ok).any() else np.empty(0, dtype=np.int64)
        simple[demoted] = False
        self.col = cols[ok].astype(np.int32)
        self.rowp = srows[start[ok]].astype(np.int32)
        self.rowq = np.where(count[ok] == 2, srows[np.minimum(start[ok] + 1, len(srows) - 1)],
                             -1).astype(np.int32)
        self.pos_a, self.pos_s = pos_a.astype(np.int32), pos_s.astype(np.int32)
        self.general = np.flatnonzero(~simple).astype(np.int64)
        self.n_simple = int(simple.sum())
        self.m, self.n = m, n

    @property
    def worthwhile(self):
        return self.n_simple >= max(8, self.m // 4) and len(self.general) > 0


_ANALYSIS_ATTR = "_ipx_box_analysis"


def analysis_for(pattern):
    a = getattr(pattern, _ANALYSIS_ATTR, None)
    if a is None:
        a = BoxRowAnalysis(pattern)
        setattr(pattern, _ANALYSIS_ATTR, a)
    return a


_P, _I64 = ctypes.c_void_p, ctypes.c_int64


class BoxSchurArgs(ctypes.Structure):
    """Mirror of ipx_boxschur_args (include/ipx.h)."""
    _fields_ = [(k, _I64) for k in ("m", "n", "ng", "mR")] + \
               [(k, _P) for k in ("rowp", "rowq", "col", "general", "inv", "alpha")] + \
               [("AR_rowptr", _P), ("AR_colidx", _P), ("AR_val", _P), ("AR_tiles", _P),
                ("AR_ntiles", _I64),
                ("ARt_rowptr", _P), ("ARt_colidx", _P), ("ARt_val", _P), ("ARt_tiles", _P),
                ("ARt_ntiles", _I64), ("inner", _P)] + \
               [(k, _P) for k in ("t", "u", "wR", "rhs", "vR", "y")] + \
               [("gcol", _P), ("grp", _P), ("gen_cols", _P), ("ngen", _I64), ("ny", _I64),
                ("up", _P), ("grp2", _P), ("yell_col", _P), ("yell_val", _P)] + \
               [(k, _I64) for k in ("gaffine", "gc0", "gdp", "gdq", "gen0", "AR_rowlen")] + \
               [("post_own_g", _P), ("post_own_e", _P), ("post_rows_wg", _I64),
                ("post_reach", _I64)]


class BoxSchurNormalSolver:
    """(A A')^-1 through per-variable elimination of the simple rows and a
    banded solve on the Schur complement of the general rows."""

    def __init__(self, A, any_sparsity=False):
        from .device_mode import RowSelection
        from .projector import BandedNormalSolver
        an = analysis_for(A.pattern)
        self.an, self.m, self.n = an, an.m, an.n
        pat = A.pattern
        cache = getattr(pat, "_ipx_box_device", None)
        if cache is None:
            sel = getattr(pat, "_ipx_box_general_pattern", None) or RowSelection(pat, an.general,
                                                                                None)
            pat._ipx_box_general_pattern = sel
            # tables of ipx_boxschur_project: per group the shared column and the rows' private
            # columns; the columns outside every group
            idx = pat.indices_h
            priv = lambda rows: np.where((rows >= 0) & (an.pos_s[np.maximum(rows, 0)] >= 0),
                                         idx[np.maximum(an.pos_s[np.maximum(rows, 0)], 0)], -1)
            gcol = np.stack((an.col, priv(an.rowp), np.where(an.rowq >= 0, priv(an.rowq), -2)),
                            axis=1).astype(np.int32)        # (-2: single-row group)
            in_group = np.zeros(an.n, dtype=bool)
            in_group[gcol[gcol >= 0]] = True
            # every variable bounded on both sides: the tables are arithmetic progressions
            gen = np.flatnonzero(~in_group)
            affine = None
            if len(gcol) and (gcol >= 0).all() and \
                    (gcol[:, 0] == gcol[0, 0] + np.arange(len(gcol))).all() and \
                    (gcol[:, 1] - gcol[:, 0] == gcol[0, 1] - gcol[0, 0]).all() and \
                    (gcol[:, 2] - gcol[:, 0] == gcol[0, 2] - gcol[0, 0]).all() and \
                    (len(gen) == 0 or (gen == gen[0] + np.arange(len(gen))).all()):
                affine = (int(gcol[0, 0]), int(gcol[0, 1] - gcol[0, 0]),
                          int(gcol[0, 2] - gcol[0, 0]), int(gen[0]) if len(gen) else 0)
            cache = pat._ipx_box_device = {
                "rowp": _i32(an.rowp), "rowq": _i32(an.rowq), "col": _i32(an.col),
                "pos_a": _i32(an.pos_a), "pos_s": _i32(an.pos_s),
                "general": _i32(an.general),
                "gcol": _i32(gcol.ravel()), "gen_cols": _i32(gen), "affine": affine,
                "sel": sel}
        self.c = cache
        self.ng = len(an.col)
        dev = ctx().device
        self.alpha = torch.zeros(self.m, dtype=_F64, device=dev)
        self.inv = torch.empty(3 * max(self.ng, 1), dtype=_F64, device=dev)
        self.wcol = torch.ones(self.n, dtype=_F64, device=dev)
        self.grp = torch.empty(4 * max(self.ng, 1), dtype=_F64, device=dev)
        # the same table in half the bytes for rows of the box form (entries +-1, slack >= 0):
        # what the CG loop's two group kernels read when the factor kernel found it exact
        self.grp2 = torch.empty(2 * max(self.ng, 1), dtype=_F64, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        st = stream_ptr()
        _hip.call("ipx_pairs_factor", self.ng, _p(cache["rowp"]), _p(cache["rowq"]),
                  _p(cache["pos_a"]), _p(cache["pos_s"]), _p(A.val), _p(cache["col"]),
                  _p(self.alpha), _p(self.inv), _p(self.wcol), _p(flag), _p(self.grp),
                  _p(self.grp2), st)
        self.A_R = cache["sel"].apply(A)                 # general rows (value gather)
        if any_sparsity:
            # Sigma = A_R W A_R' = B B' with B = A_R diag(sqrt(w)) (0 < w <= 1): a scaled copy of
            # the general rows' values goes to the solver of any sparsity -- dense Cholesky up to
            # 16384 rows, else the device-resident preconditioned CG (projector.py section 4)
            from .projector import IterativeNormalSolver
            from .dense import DenseNormalSolver
            sw = torch.sqrt(self.wcol)
            idx = getattr(self.A_R.pattern, "_ipx_col_index64", None)
            if idx is None:
                idx = self.A_R.pattern._ipx_col_index64 = self.A_R.pattern.indices.to(torch.int64)
            B = dv.DeviceCSR(self.A_R.pattern, self.A_R.val * sw[idx])
            self.B = B
            mR = B.shape[0]
            self.inner = DenseNormalSolver(B) if mR <= DenseNormalSolver.MAX_ROWS_FROM_SPARSE \
                else IterativeNormalSolver(B)
        else:
            self.inner = BandedNormalSolver(self.A_R, col_weights=self.wcol)   # Sigma = A_R W A_R'
        bits = int(flag.item())
        if bits & 1:
            raise np.linalg.LinAlgError("Singular Jacobian matrix: A A' is not positive definite")
        if bits & 2 or _hip.debug_form("no-compact-groups"):
            self.grp2 = None
        self.u = torch.zeros(self.n, dtype=_F64, device=dev)      # only grouped columns are written
        self.t = torch.zeros(self.m, dtype=_F64, device=dev)
        self._args = None

    perm = None        # rows are taken in the caller's order (cg_fused.supports)

    def c_args(self):
        """Argument block for ipx_boxschur_solve (the device-resident CG loop);
        None when the Schur system needs a row permutation."""
        if getattr(self.inner, "perm", None) is not None or not hasattr(self.inner, "band"):
            return None                       # (a dense / iterative Schur solve: host-driven)
        if self._args is None:
            mR, dev = len(self.an.general), ctx().device
            self._scratch = [torch.zeros(k, dtype=_F64, device=dev) for k in (mR, mR, mR, self.n)]
            a = BoxSchurArgs()
            a.m, a.n, a.ng, a.mR = self.m, self.n, self.ng, mR
            c = self.c
            a.rowp, a.rowq, a.col = (c[k].data_ptr() for k in ("rowp", "rowq", "col"))
            a.general = c["general"].data_ptr()
            a.inv, a.alpha = self.inv.data_ptr(), self.alpha.data_ptr()
            ARt = self.A_R.T
            self._keep = ARt
            for pre, M in (("AR", self.A_R), ("ARt", ARt)):
                pat = M.pattern
                setattr(a, pre + "_rowptr", pat.indptr.data_ptr())
                setattr(a, pre + "_colidx", pat.indices.data_ptr())
                setattr(a, pre + "_val", M.val.data_ptr() if M.val.numel() else None)
                setattr(a, pre + "_tiles", pat.tiles.data_ptr())
                setattr(a, pre + "_ntiles", pat.ntiles)
            a.inner = self.inner.handle
            a.t, a.u = self.t.data_ptr(), self.u.data_ptr()
            a.wR, a.rhs, a.vR, a.y = (t.data_ptr() for t in self._scratch)
            # the fused projection (ipx_boxschur_project): group tables; ny = one past the last
            # column A_R touches (the slack columns of the box rows lie behind it)
            a.gcol, a.grp = c["gcol"].data_ptr(), self.grp.data_ptr()
            a.gen_cols, a.ngen = c["gen_cols"].data_ptr(), c["gen_cols"].numel()
            a.ny = int(self.A_R.pattern.indices_h.max()) + 1 if self.A_R.pattern.nnz else 0
            self._up = torch.zeros(self.n, dtype=_F64, device=dev)
            a.up = self._up.data_ptr()
            a.grp2 = self.grp2.data_ptr() if self.grp2 is not None else None
            if self.grp2 is not None and c["affine"] and not _hip.debug_form("no-affine-groups"):
                a.gaffine = 1
                a.gc0, a.gdp, a.gdq, a.gen0 = c["affine"]
            lens = np.diff(self.A_R.pattern.indptr_h)
            if len(lens) and lens[0] in (2, 4, 8, 16) and (lens == lens[0]).all() \
                    and not _hip.debug_form("no-compact-groups"):
                a.AR_rowlen = int(lens[0])     # A_R u formed inside the Schur solve's kernel
            self._yell = self._item_columns(ARt)
            if self._yell is not None:
                a.yell_col, a.yell_val = c["yell"][2].data_ptr(), self._yell.data_ptr()
            post = self._post_tables() if (self._yell is not None and a.gaffine and a.AR_rowlen) \
                else None
            if post is not None:
                a.post_own_g, a.post_own_e = post[0].data_ptr(), post[1].data_ptr()
                a.post_rows_wg, a.post_reach = post[2], post[3]
            self._args = a
        return self._args

    POST_PG, POST_TB = 6, 512            # csrc/banded.hip POST_PG, PCR_TB
    POST_ITEMS_PER_PART = 1024           # csrc/boxschur.hip ipx_boxschur_project_count

    def _post_tables(self):
        """Item ranges per workgroup of the Schur solve for the fused back substitution
        (ipx_boxschur_args.post_own_g / post_own_e), or None: an item belongs to the workgroup
        whose rows contain the FIRST general row of its column; needs every item to have one,
        first rows non-decreasing along the groups and along the other columns, the second row
        within reach, and no workgroup with more items than the kernel's lanes take.  Symbolic;
        cached per geometry of the solve."""
        if _hip.debug_form("no-post-tail"):
            return None
        lib = _hip.load()
        geo = (ctypes.c_int32 * 2)()
        if not lib.ipx_banded_decoupled_geometry(ctypes.c_void_p(self.inner.handle), geo):
            return None
        rows_wg, nwg = int(geo[0]), int(geo[1])
        L = int(lib.ipx_banded_pcr_level(ctypes.c_void_p(self.inner.handle)))
        if L <= 0:
            return None
        c = self.c
        key = ("post", rows_wg, nwg)
        if key not in c:
            c[key] = None
            pat = self._keep.pattern                       # A_R' (rows = columns of A_R)
            indptr, indices = pat.indptr_h.astype(np.int64), pat.indices_h.astype(np.int64)
            ng = self.ng
            cols = np.concatenate((self.an.col.astype(np.int64),
                                   c["gen_cols"].cpu().numpy().astype(np.int64)))
            cnt = indptr[cols + 1] - indptr[cols]
            if len(cols) and cnt.min() >= 1 and cnt.max() <= 2:
                first = indices[indptr[cols]]
                second = np.where(cnt > 1, indices[np.minimum(indptr[cols] + 1, len(indices) - 1)],
                                  first)
                lo, hi = np.minimum(first, second), np.maximum(first, second)
                ok = np.all(np.diff(lo[:ng]) >= 0) and np.all(np.diff(lo[ng:]) >= 0)
                owner = lo // rows_wg
                if ok and owner.max() < nwg:
                    own_g = np.searchsorted(owner[:ng], np.arange(nwg + 1), side="left")
                    own_e = np.searchsorted(owner[ng:], np.arange(nwg + 1), side="left")
                    if np.diff(own_g).max() <= self.POST_PG * self.POST_TB and \
                            np.diff(own_e).max() <= self.POST_TB:
                        c[key] = (_i32(own_g), _i32(own_e), rows_wg, int((hi - lo).max()))
        post = c[key]
        # (reach <= 1: only the first halo row of the solve's window is exact; nwg <= the count
        # of ||g||^2 partials per half the consumer folds -- csrc/banded.hip
        # ipx_banded_solve_rows_launch refuses the same two cases)
        if post is None or post[3] > 1 or post[3] > (1 << L):
            return None
        if nwg > (self.ng + int(c["gen_cols"].numel()) + self.POST_ITEMS_PER_PART - 1) \
                // self.POST_ITEMS_PER_PART:
            return None
        return post

    def _item_columns(self, ARt):
        """Values of the columns of A_R per item of the projection, ELL(2) (see
        ipx_boxschur_args.yell_col in include/ipx.h); the positions are found once per pattern.
        None when a column holds more than two entries."""
        c = self.c
        if "yell" not in c:
            c["yell"] = None
            pat = ARt.pattern
            indptr, indices = pat.indptr_h.astype(np.int64), pat.indices_h
            cols = np.concatenate((self.an.col.astype(np.int64),
                                   c["gen_cols"].cpu().numpy().astype(np.int64)))
            first, cnt = indptr[cols], indptr[cols + 1] - indptr[cols]
            if len(cols) and len(indices) and cnt.max() <= 2 and \
                    not _hip.debug_form("no-compact-groups"):
                have = np.stack((cnt >= 1, cnt >= 2))
                pos = np.where(have, np.stack((first, first + 1)), 0)
                col = np.where(have, indices[pos], 0)
                c["yell"] = (_i32(pos.ravel()),
                             torch.from_numpy(have.ravel().astype(np.float64)).to(ctx().device),
                             _i32(col.ravel()))
        if c["yell"] is None:
            return None
        pos, mask, _ = c["yell"]
        out = dv._empty(pos.numel())
        _hip.call("ipx_gather", pos.numel(), _p(ARt.val), _p(pos), _p(mask), None, _p(out),
                  stream_ptr())
        return out

    def solve(self, w):
        c, st = self.c, stream_ptr()
        args = self.c_args()
        if args is not None:                 # whole application behind one ABI call
            v = dv._empty(self.m)
            _hip.call("ipx_boxschur_solve", ctypes.byref(args), _p(w.t), _p(v), None, None, None,
                      st)
            return DVec(v)
        _hip.call("ipx_pairs_tsolve", self.ng, _p(c["rowp"]), _p(c["rowq"]), _p(self.inv),
                  _p(self.alpha), _p(w.t), _p(self.t), _p(c["col"]), _p(self.u), st)
        mR = len(self.an.general)
        w_R = dv._empty(mR)
        _hip.call("ipx_gather", mR, _p(w.t), _p(c["general"]), None, None, _p(w_R), st)
        rhs = self.A_R.spmv(DVec(self.u), alpha=-1.0, beta=1.0, yin=DVec(w_R))
        v_R = self.inner.solve(rhs)
        y = self.A_R.T.dot(v_R)
        v = dv._empty(self.m)
        _hip.call("ipx_pairs_vsolve", self.ng, _p(c["rowp"]), _p(c["rowq"]), _p(self.inv),
                  _p(self.alpha), _p(self.t), _p(y.t), _p(c["col"]), _p(v), st)
        _hip.call("ipx_scatter", mR, _p(v_R.t), _p(c["general"]), _p(v), st)
        return DVec(v)
