"""Host side of csrc/sqp.hip: one outer iteration of the trust-region SQP method as three
chains of launches (``ipx_sqp_front`` / ``ipx_sqp_judge`` / ``ipx_sqp_refresh``) whose decisions
are taken on the device -- the host reads one block of scalars per chain and only to learn what
it must know before it calls the user's callbacks.  ``sqp.py`` drives either this object or its
host-driven twin (``sqp.HostStages``) through the same four calls; both work on the same block
layout and the same decision arithmetic (``ipx_sqp_*_host`` is the kernels' own code compiled for
the host), so a solve may change from one to the other between two iterations.

What qualifies (``StepChain.fits``): the HIP backend's own types -- CSR Jacobian with a banded or
box-Schur ``(A A')^-1`` in the caller's row order, CSR (+ diagonal) Lagrangian Hessian, diagonal
or no scaling.  Everything else (dense Jacobians, operator Hessians, the SVD exit) takes the
host-driven stages.
"""
import ctypes

import numpy as np
import torch

from . import _hip
from . import cg_fused
from . import device as dv
from .device import DVec, DeviceCSR, _p, stream_ptr, ctx

_P, _I64, _F64c = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double

# ---- the block (csrc/sqp.hip SQ_*) -------------------------------------------------------
(RADIUS, PENALTY, F, NORM_B, NORM_DN, RADIUS_T, NORMAL_KIND, NVIOL, HDD, CD, LIN, NORM_D, NORM_DT,
 QMODEL, VPRED, PREV_PENALTY, PRED, MERIT, F_NEXT, NORM_B_NEXT, ACTUAL, RATIO, SOC, ACCEPT, OPT,
 VIOL, NORM_A2, FACTOR_BAD, EXIT_TAU, EXIT_DONE, X_OUTSIDE, PRIME_STEPS) = range(32)
CG = 32
SIZE = 64

TR_FACTOR = 0.8            # equality_constrained_sqp.py:59-60
BOX_FACTOR = 0.5


class ChainArgs(ctypes.Structure):
    """Mirror of ipx_sqp_args (include/ipx.h)."""
    _fields_ = [(name, typ) for name, typ in (
        ("n", _I64), ("m", _I64), ("cg", _P), ("A_tiles", _P), ("A_ntiles", _I64), ("q", _P),
        ("x", _P), ("c", _P), ("b", _P), ("lb", _P), ("ub", _P), ("scale", _P),
        ("dn", _P), ("ct", _P), ("lbt", _P), ("ubt", _P), ("d", _P), ("Hd", _P), ("x_next", _P),
        ("Ad", _P), ("v_out", _P), ("part", _P), ("red", _P), ("ws", _P),
        ("orth_tol", _F64c), ("cancellation", _F64c), ("verdict", _P),
        ("A_norm_part", _P), ("A_norm_grid", _I64), ("host_block", _P))]


STATS = {"fronts": 0, "host_doglegs": 0, "host_cg": 0, "prime_retries": 0, "refreshes": 0,
         "deferred_factorizations": 0, "verdict_misses": 0, "host_iterations": 0,
         "settles_reused": 0}


def _block_from(values):
    return (ctypes.c_double * SIZE)(*values)


def model_host(q):
    _hip.load().ipx_sqp_model_host(q)


def ratio_host(q):
    _hip.load().ipx_sqp_ratio_host(q)


def radius_host(q):
    _hip.load().ipx_sqp_radius_host(q)


def new_block():
    return (ctypes.c_double * SIZE)()


class StepChain:
    """Workspace of the device-side outer iteration for one (n, m, which bounds exist)."""

    _cache = {}

    @classmethod
    def get(cls, n, m, has_lb, has_ub):
        key = (n, m, has_lb, has_ub, ctx().device.index)
        hit = cls._cache.get(key)
        if hit is None:
            if len(cls._cache) >= 4:
                cls._cache.pop(next(iter(cls._cache)))
            hit = cls._cache[key] = cls(n, m, has_lb, has_ub)
        return hit

    def __init__(self, n, m, has_lb, has_ub):
        dev, f64 = ctx().device, torch.float64
        z = lambda k: torch.zeros(int(k), dtype=f64, device=dev)
        self.n, self.m = n, m
        self.dn, self.ct, self.d, self.Hd = z(n), z(n), z(n), z(n)
        self.lbt = z(n) if has_lb else None
        self.ubt = z(n) if has_ub else None
        self.Ad = z(m)
        self.q = z(SIZE)
        self.red = z(24)
        self.verdict = z(2)
        self.part = None
        self.anorm = None
        self.rargs = None
        self.last_niter = (3, 3)     # (CG iterations of the last two calls: size the next first batch)
        self.expect_dogleg = True    # (the last normal step was not the Newton point)
        self.expect_steps = False    # (the last priming's projections needed a correction step)
        self.args = ChainArgs()
        a = self.args
        a.n, a.m = n, m
        a.q, a.red, a.ws = self.q.data_ptr(), self.red.data_ptr(), ctx().ws.data_ptr()
        a.dn, a.ct, a.d, a.Hd, a.Ad = (t.data_ptr() for t in (self.dn, self.ct, self.d, self.Hd,
                                                                  self.Ad))
        a.lbt = self.lbt.data_ptr() if has_lb else None
        a.ubt = self.ubt.data_ptr() if has_ub else None
        self.lbt_vec = DVec(self.lbt) if has_lb else None
        self.ubt_vec = DVec(self.ubt) if has_ub else None
        self.keep = None
        # every chain returns with the block in here: its last workgroup publishes it
        self.block = (ctypes.c_double * SIZE)()
        a.host_block = ctypes.addressof(self.block)

    # ---- what qualifies
    @staticmethod
    def fits(A, Z, Y, H, scaling_op, vectors):
        from .projector import NormalEquationProjector
        from .operators import DiagonalOperator
        P = getattr(Z, "projector", None)
        if not isinstance(A, DeviceCSR) or not isinstance(P, NormalEquationProjector):
            return False
        if P.A is not A or P.row_perm is not None or P.m == 0 or P.n - P.m < 1:
            return False
        if getattr(Y, "projector", None) is not P:
            return False
        if cg_fused._solver_kind(P.solver) is None or getattr(P.solver, "refine_steps", 0) \
                or getattr(P.solver, "perm", None) is not None:
            return False
        if H is not None and cg_fused._hessian_parts(H) is None:
            return False
        if scaling_op is not None and not isinstance(scaling_op, DiagonalOperator):
            return False
        return all(isinstance(v, DVec) for v in vectors if v is not None)

    # ---- binding to an iteration's operands
    def bind(self, L, P, x, c, b, lb, ub, scale, x_next=None, v_out=None):
        lib = _hip.load()
        a = self.args
        a.cg = ctypes.cast(ctypes.pointer(L.args), _P)
        pat = P.A.pattern
        a.A_tiles, a.A_ntiles = pat.tiles.data_ptr(), pat.ntiles
        a.x, a.c, a.b = x.t.data_ptr(), c.t.data_ptr(), b.t.data_ptr()
        a.lb = lb.t.data_ptr() if lb is not None else None
        a.ub = ub.t.data_ptr() if ub is not None else None
        a.scale = scale.t.data_ptr() if scale is not None else None
        a.x_next = x_next.data_ptr() if x_next is not None else None
        a.v_out = v_out.data_ptr() if v_out is not None else None
        a.orth_tol, a.cancellation = float(P.orth_tol), float(P.CANCELLATION)
        need = int(lib.ipx_sqp_part_doubles(ctypes.byref(a)))
        if self.part is None or self.part.numel() < need:
            self.part = torch.zeros(need, dtype=torch.float64, device=ctx().device)
        a.part = self.part.data_ptr()
        self.keep = (L, P, x, c, b, lb, ub, scale, x_next, v_out)

    def read(self):
        """The block of the chain that just returned (a host copy the decision functions can work
        on): the chain's last kernel handed it over, the entry point waited for it -- ONE blocking
        read per chain and no launch of its own."""
        return type(self.block).from_buffer_copy(self.block)

    def bind_refresh(self, P, c, b, v_out):
        """Operands of ``ipx_sqp_refresh``: the Jacobian, its transpose and the solver of the
        projector, in a CG argument block of this object's own (no Hessian yet at this point
        of the outer iteration)."""
        lib = _hip.load()
        if self.rargs is None:
            self.rargs = cg_fused.CgArgs()
            dev, f64 = ctx().device, torch.float64
            self.rw = torch.zeros(self.m, dtype=f64, device=dev)
            self.rv = torch.zeros(self.m, dtype=f64, device=dev)
        r = self.rargs
        A, At = P.A, P.A.T
        r.n, r.m = self.n, self.m
        for pre, M in (("A", A), ("At", At)):
            pat = M.pattern
            setattr(r, pre + "_rowptr", pat.indptr.data_ptr())
            setattr(r, pre + "_colidx", pat.indices.data_ptr())
            setattr(r, pre + "_val", M.val.data_ptr())
            setattr(r, pre + "_tiles", pat.tiles.data_ptr())
            setattr(r, pre + "_ntiles", pat.ntiles)
        r.H_ntiles = 0
        r.solver_kind = cg_fused._solver_kind(P.solver)
        if r.solver_kind == 1:
            r.banded = ctypes.cast(ctypes.pointer(P.solver.c_args()), _P)
        else:
            r.banded = ctypes.c_void_p(P.solver.handle)
        r.w, r.v = self.rw.data_ptr(), self.rv.data_ptr()
        a = self.args
        a.cg = ctypes.cast(ctypes.pointer(r), _P)
        a.A_tiles, a.A_ntiles = A.pattern.tiles.data_ptr(), A.pattern.ntiles
        a.c, a.b, a.v_out = c.t.data_ptr(), b.t.data_ptr(), v_out.data_ptr()
        need = int(lib.ipx_sqp_part_doubles(ctypes.byref(a)))
        if self.part is None or self.part.numel() < need:
            self.part = torch.zeros(need, dtype=torch.float64, device=ctx().device)
        a.part = self.part.data_ptr()
        self.keep = (P, A, At, c, b, v_out)

    def front(self, have_dn, with_dogleg, radius, penalty, f, norm_b, norm_A, first_end,
              quiet=False):
        """``quiet``: return behind the chain's last launch, the block stays on the device --
        the next chain's block carries all of it (the caller enqueues the callbacks at the
        trial point and the verdict behind this one and reads once)."""
        a = self.args
        keep = a.host_block
        if quiet:
            a.host_block = None
        # (||A||_F is the host's by now -- the refresh's read brought it: by value)
        try:
            _hip.call("ipx_sqp_front", ctypes.byref(a), int(have_dn), int(with_dogleg),
                      int(self.expect_steps), float(radius), float(penalty),
                      float(f), float(norm_b), TR_FACTOR, BOX_FACTOR, float("nan"), float(norm_A),
                      int(first_end), stream_ptr())
        finally:
            a.host_block = keep
        STATS["fronts"] += 1

    def model(self, penalty, f, norm_b, host_cg):
        _hip.call("ipx_sqp_model", ctypes.byref(self.args), float(penalty), float(f),
                  float(norm_b), 1 if host_cg else 0, stream_ptr())

    def judge(self, b_next, f_next):
        fdev = None
        if torch.is_tensor(f_next):
            self.keep_f = f_next
            fdev, f_next = f_next.data_ptr(), 0.0
        _hip.call("ipx_sqp_judge", ctypes.byref(self.args), _p(b_next.t), float(f_next), fdev,
                  stream_ptr())

    def refresh(self, A_new_norm, verdict):
        a = self.args
        if A_new_norm is not None:
            part, grid = A_new_norm
            a.A_norm_part, a.A_norm_grid = part.data_ptr(), grid
            self.anorm = part
        else:
            a.A_norm_part, a.A_norm_grid = None, 0
        a.verdict = self.verdict.data_ptr() if verdict else None
        _hip.call("ipx_sqp_refresh", ctypes.byref(a), stream_ptr())
        STATS["refreshes"] += 1
