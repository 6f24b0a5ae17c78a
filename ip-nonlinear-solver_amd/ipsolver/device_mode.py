"""Device-callback mode of ``minimize_constrained``.

When ``x0`` is a CUDA tensor the user's callbacks are called with CUDA
tensors and return device objects, so nothing crosses PCIe between two
iterations:

    fun(x) -> float | 0-d tensor | device.DeviceScalar (e.g. ScalarPack.combine: no read)
    grad(x) -> 1-D tensor
    hess(x) -> DeviceCSR | 1-D tensor (diagonal) | 2-D tensor / DeviceDense (dense) |
               DeviceHessian | None | a tuple of such terms (their sum)
    NonlinearConstraint.fun(x) -> 1-D tensor
    NonlinearConstraint.jac(x) -> DeviceCSR (a fixed CSRPattern, values refreshed)
    NonlinearConstraint.hess(x, v) -> DeviceCSR | 1-D tensor (diagonal) | None
    LinearConstraint(A)  with A a scipy sparse matrix, a DeviceCSR, or -- dense Jacobian,
                         BASELINE config 2: every row an equality -- a 2-D CUDA tensor /
                         DeviceDense
    BoxConstraint(kind)  unchanged

Order of evaluations: with the device chains of the outer iteration (sqp.ChainStages) ``fun``
and the constraints' ``fun`` are enqueued at the trial point BEHIND the kernels that compute it
and before the host has looked at their outcome; a trial step the host then has to finish (a CG
loop that needs more iterations, a box event, the dogleg proper) is evaluated again in a new
vector.  Such a provisional call is not counted (``nfev`` / ``ncev`` are the reference's), but
the callback was called at a point the reference would not have shown it;
``ipsolver.sqp.EVALUATE_BEHIND_THE_CHAIN = False`` restores the reference's order exactly.

The canonical form (row selection, sign flips, stacking of several
constraints, multiplier re-signing; reference _canonical_constraint.py:
169-480) and the barrier's augmented Jacobian (tr_interior_point.py:165-194)
are value refreshes on patterns built once: gather / scatter kernels over
index maps computed on the host at initialisation (symbolic work only).
Finite-difference Hessians are not available in this mode (they evaluate host
callbacks by construction).
"""
import numpy as np
import scipy.sparse as sps
import torch

from . import _hip
from . import device as dv
from .canonical import parse_constraint, HessianSum
from .constraints import (NonlinearConstraint, LinearConstraint, BoxConstraint, check_kind,
                          check_enforce_feasibility, is_feasible, reinforce_box, _INFEASIBLE)
from .device import DVec, DeviceCSR, CSRPattern, _p, stream_ptr, ctx

_F64 = torch.float64


def is_device_vector(x):
    return torch.is_tensor(x) and x.is_cuda


def as_dvec(t):
    if isinstance(t, DVec):
        return t
    if not torch.is_tensor(t):
        raise TypeError("device-callback mode: expected a CUDA tensor, got %r" % type(t))
    return DVec(t.to(_F64).reshape(-1).contiguous())


def objective_value(f):
    """What a device-mode ``fun`` returned, for the outer loops: a float, or -- a value still on
    the device (``device.DeviceScalar``, a 0-d / one-element CUDA tensor) -- a DeviceScalar that
    the step's verdict consumes on the device; the host reads it only where it needs it."""
    if isinstance(f, dv.DeviceScalar):
        return f
    if torch.is_tensor(f) and f.is_cuda and f.numel() == 1:
        return dv.DeviceScalar(f.detach().to(_F64).reshape(1))
    return float(f)


def _idx(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(ctx().device)


def _vec(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(ctx().device)


def gather(x, idx, sign=None, shift=None, out=None):
    n = idx.numel()
    out = dv._empty(n) if out is None else out
    _hip.call("ipx_gather", n, _p(x), _p(idx), _p(sign), _p(shift), _p(out), stream_ptr())
    return out


def scatter(x, idx, out):
    _hip.call("ipx_scatter", idx.numel(), _p(x), _p(idx), _p(out), stream_ptr())


class RowSelection:
    """J[rows, :] scaled by ``sign`` per row, as a value refresh on a pattern
    derived once from J's pattern (_canonical_constraint.py:251-265)."""

    def __init__(self, pattern, rows, sign):
        indptr, indices = pattern.indptr_h.astype(np.int64), pattern.indices_h
        counts = indptr[rows + 1] - indptr[rows]
        new_indptr = np.concatenate(([0], np.cumsum(counts))).astype(np.int32)
        self.identity = (len(rows) == pattern.shape[0]
                         and np.array_equal(rows, np.arange(pattern.shape[0]))
                         and (sign is None or np.all(sign == 1)))
        # source position of every selected nonzero: start of its row + offset inside the row
        total = int(new_indptr[-1])
        src = (np.repeat(indptr[rows] - new_indptr[:-1], counts) + np.arange(total)
               if total and not self.identity else np.empty(0, dtype=np.int64))
        self.pattern = pattern if self.identity else CSRPattern(
            new_indptr, indices[src] if len(src) else np.empty(0, np.int32),
            (len(rows), pattern.shape[1]))
        self.src = _idx(src)
        self.sign = None if sign is None or np.all(sign == 1) else _vec(np.repeat(sign, counts))

    def apply(self, J):
        if self.identity:
            return J
        if self.src.numel() == 0:
            return DeviceCSR(self.pattern, dv._empty(0))
        return DeviceCSR(self.pattern, gather(J.val, self.src, self.sign))


class DeviceRowMap:
    """Device twin of canonical._RowMap."""

    def __init__(self, kind, n_vars):
        (self.eq, self.ineq, val_eq, val_ineq, self.sign_h,
         self.fun_len) = parse_constraint(kind)
        self.n_eq, self.n_ineq, self.n_vars = len(self.eq), len(self.ineq), n_vars
        self.eq_idx, self.ineq_idx = _idx(self.eq), _idx(self.ineq)
        self.val_eq, self.val_ineq = _vec(val_eq), _vec(val_ineq)
        self.sign = _vec(self.sign_h)
        self._sel = {}
        # multipliers back in the user's row order (reference :210-218), as
        # three gathers from [v; 0]
        def inverse(index_sets, total):
            out = np.full(self.fun_len, total, dtype=np.int64)      # -> the appended zero
            for pos, rows in index_sets:
                out[rows] = pos
            return _idx(out)
        up = np.flatnonzero(self.sign_h == 1)
        lo = np.flatnonzero(self.sign_h == -1)
        self.m_eq = inverse([(np.arange(self.n_eq), self.eq)], self.n_eq)
        self.m_up = inverse([(up, self.ineq[up])], self.n_ineq)
        self.m_lo = inverse([(lo, self.ineq[lo])], self.n_ineq)
        # every row an equality with right-hand side 0, in the caller's order (``('equals', 0)``):
        # the canonical values ARE the callback's, the multipliers the canonical ones -- no
        # gather, no copy (c - 0 and 1 * v are the same bits)
        self.all_eq = (self.n_ineq == 0 and self.n_eq == self.fun_len
                       and np.array_equal(self.eq, np.arange(self.fun_len))
                       and not np.any(val_eq))
        self._none = DVec.zeros(0)

    def values(self, c):
        if self.all_eq:
            return self._none, c
        c_eq = DVec(gather(c.t, self.eq_idx, None, self.val_eq)) if self.n_eq else DVec.zeros(0)
        c_ineq = DVec(gather(c.t, self.ineq_idx, self.sign, self.val_ineq)) if self.n_ineq \
            else DVec.zeros(0)
        return c_ineq, c_eq

    def jac(self, J):
        from .dense import DeviceDense
        if isinstance(J, DeviceDense):
            # a dense Jacobian passes through whole: rows all equalities, in order
            if self.n_ineq or not np.array_equal(self.eq, np.arange(J.shape[0])):
                raise NotImplementedError("device-callback mode: a dense constraint Jacobian "
                                          "must consist of equality rows only (kind 'equals')")
            return self._no_rows(J.shape[1]), J
        if self.all_eq:
            return self._no_rows(J.shape[1]), J
        key = id(J.pattern)
        if key not in self._sel:
            if len(self._sel) >= 4:          # (a row map may outlive a solve: _SPEC_CACHE)
                self._sel.pop(next(iter(self._sel)))
            self._sel[key] = (RowSelection(J.pattern, self.ineq, self.sign_h),
                              RowSelection(J.pattern, self.eq, None), J.pattern)
        sel_ineq, sel_eq, _ = self._sel[key]
        return sel_ineq.apply(J), sel_eq.apply(J)

    _NO_ROWS = {}

    @staticmethod
    def _no_rows(n_vars):
        key = (n_vars, ctx().device.index)
        hit = DeviceRowMap._NO_ROWS.get(key)
        if hit is None:
            if len(DeviceRowMap._NO_ROWS) >= 8:
                DeviceRowMap._NO_ROWS.clear()
            hit = DeviceRowMap._NO_ROWS[key] = DeviceCSR(
                CSRPattern(np.zeros(1, np.int32), np.empty(0, np.int32), (0, n_vars)),
                dv._empty(0))
        return hit

    def multipliers(self, v_eq, v_ineq):
        if self.all_eq:
            return v_eq
        zero = torch.zeros(1, dtype=_F64, device=ctx().device)
        ve = torch.cat((v_eq.t, zero))
        vi = torch.cat((v_ineq.t, zero))
        v = DVec(gather(ve, self.m_eq))
        return v + DVec(gather(vi, self.m_up)) - DVec(gather(vi, self.m_lo))


def _identity_csr(n):
    pat = CSRPattern(np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), (n, n))
    return DeviceCSR(pat, torch.ones(n, dtype=_F64, device=ctx().device))


_SPEC_CACHE = {}


def _spec_key(kind, enforce, m, n):
    """(kind, enforce_feasibility, rows, variables, device) when the specification consists of
    a keyword and Python scalars (then it IS its value); None for arrays."""
    if isinstance(kind, str):
        kind = (kind,)
    if not isinstance(kind, (tuple, list)) or not isinstance(enforce, (bool, np.bool_)):
        return None
    flat = []
    for k in kind:
        if isinstance(k, str):
            flat.append(k)
        elif isinstance(k, (int, float, np.integer, np.floating)) and not isinstance(k, bool):
            flat.append(float(k))
        else:
            return None
    return (tuple(flat), bool(enforce), int(m), int(n), ctx().device.index)


class _DeviceConstraint:
    """One user constraint evaluated on the device + its canonical row map."""

    def __init__(self, user, x0):
        n = len(x0)
        self.constant_jac = isinstance(user, (BoxConstraint, LinearConstraint))
        if isinstance(user, BoxConstraint):
            J = _identity_csr(n)
            self.fun = lambda x: x
            self.jac = lambda x: J
            self.hess = None
            f0 = x0
        elif isinstance(user, LinearConstraint):
            from .dense import DeviceDense
            if isinstance(user.A, (DeviceCSR, DeviceDense)):
                A = user.A
            elif torch.is_tensor(user.A) and user.A.dim() == 2:
                if not user.A.is_cuda:
                    raise TypeError("device-callback mode: a dense LinearConstraint matrix must "
                                    "be a CUDA tensor")
                A = DeviceDense(user.A.to(_F64))       # dense Jacobian resident in HBM
            else:
                A = DeviceCSR.from_scipy(sps.csr_matrix(user.A))
            self.fun = lambda x: A.dot(x)
            self.jac = lambda x: A
            self.hess = None
            f0 = A.dot(x0)
        elif isinstance(user, NonlinearConstraint):
            ufun, ujac, uhess = user._fun, user._jac, user._hess
            self.fun = lambda x: as_dvec(ufun(x.t))
            self.jac = lambda x: _check_jac(ujac(x.t))
            if uhess in ('2-point', '3-point', 'cs'):
                # d/dx [J(x)' v] by differences (reference _constraints.py:136-146), on the device
                from .fd import DeviceFiniteDifferenceOperator
                self.hess = lambda x, v: DeviceFiniteDifferenceOperator(
                    lambda xt: _check_jac(ujac(xt)).T.dot(v), x, uhess)
            else:
                self.hess = None if uhess is None else (lambda x, v: uhess(x.t, v.t))
            f0 = self.fun(x0)
        else:
            raise ValueError("Unknown Constraint type.")
        m = len(f0)
        # kind / enforce_feasibility broadcast to the rows and the canonical row map derived from
        # them (index maps on the host, their device copies, per Jacobian pattern the row
        # selections) depend on the SPECIFICATION only: kept across calls for specifications made
        # of scalars (0.5 + 0.5 ms of a 13 ms config-3 solve went into rebuilding them)
        key = _spec_key(user.kind, user.enforce_feasibility, m, n)
        hit = _SPEC_CACHE.get(key) if key is not None else None
        if hit is None:
            self.kind = check_kind(user.kind, m)
            self.enforce = check_enforce_feasibility(user.enforce_feasibility, m)
        else:
            self.kind, self.enforce, self.rows = hit
        self.x0 = x0
        if self.enforce.any():
            f0_h = f0.to_host()
            if not is_feasible(self.kind, self.enforce, f0_h):
                if isinstance(user, BoxConstraint):     # reference _constraints.py:332-340
                    from warnings import warn
                    warn("The initial point was changed in order to stay inside box "
                         "constraints.")
                    self.x0 = DVec.from_host(reinforce_box(self.kind, self.enforce, f0_h))
                    f0 = self.x0
                else:
                    raise ValueError(_INFEASIBLE)
        if hit is None:
            self.rows = DeviceRowMap(self.kind, n)
            if key is not None:
                if len(_SPEC_CACHE) >= 8:
                    _SPEC_CACHE.pop(next(iter(_SPEC_CACHE)))
                _SPEC_CACHE[key] = (self.kind, self.enforce, self.rows)
        self.f0 = f0
        self.J0 = self.jac(self.x0)

    @property
    def n_eq(self):
        return self.rows.n_eq

    @property
    def n_ineq(self):
        return self.rows.n_ineq


def _check_jac(J):
    """What a device-mode ``jac`` callback may return: a DeviceCSR, or -- dense Jacobians,
    equality rows only (``DeviceRowMap.jac``) -- a DeviceDense / a 2-D CUDA tensor."""
    from .dense import DeviceDense
    if torch.is_tensor(J) and J.dim() == 2 and J.is_cuda:
        J = DeviceDense(J.to(_F64).contiguous())
    if not isinstance(J, (DeviceCSR, DeviceDense)):
        raise TypeError("device-callback mode: `jac` must return an ipsolver.device.DeviceCSR "
                        "(or, for a dense Jacobian of equality rows, a 2-D CUDA tensor)")
    return J


_vstack_cache = {}


def vstack_csr(parts, n_cols):
    """Stack DeviceCSR blocks vertically; the stacked pattern is built once per
    combination of part patterns, values are concatenated."""
    parts = [p for p in parts]
    if len(parts) == 1:
        return parts[0]
    key = tuple(id(p.pattern) for p in parts)
    hit = _vstack_cache.get(key)
    if hit is None:
        indptr = [np.zeros(1, dtype=np.int64)]
        off = 0
        for p in parts:
            indptr.append(p.pattern.indptr_h[1:].astype(np.int64) + off)
            off += p.pattern.nnz
        pat = CSRPattern(np.concatenate(indptr).astype(np.int32),
                         np.concatenate([p.pattern.indices_h for p in parts]),
                         (sum(p.shape[0] for p in parts), n_cols))
        hit = _vstack_cache[key] = (pat, [p.pattern for p in parts])   # keep parts alive
    return DeviceCSR(hit[0], torch.cat([p.val for p in parts]))


class DeviceCanonical:
    """Device twin of canonical.CanonicalConstraint for a list of constraints."""

    def __init__(self, constraints, x0):
        self.parts = []
        for c in constraints:
            part = _DeviceConstraint(c, x0)
            x0 = part.x0
            self.parts.append(part)
        for part in self.parts:       # reference _canonical_constraint.py:382-383
            if part.x0 is not x0 and not torch.equal(part.x0.t, x0.t):
                raise RuntimeError("Unmatching initial point.")
        self.x0 = x0
        self.n_vars = len(x0)
        self.n_eq = sum(p.n_eq for p in self.parts)
        self.n_ineq = sum(p.n_ineq for p in self.parts)
        self.enforce_feasibility = np.hstack(
            [p.enforce[p.rows.ineq] if p.n_ineq else np.empty(0, dtype=bool)
             for p in self.parts]) if self.parts else np.empty(0, dtype=bool)
        self._empty = DeviceRowMap._no_rows(self.n_vars)
        vals = [p.rows.values(p.f0) for p in self.parts]
        self.c_ineq0, self.c_eq0 = self._stack_values(vals)
        self.J_ineq0, self.J_eq0 = self._stack_jacs([p.rows.jac(p.J0) for p in self.parts])
        self.hess = self._hess if any(p.hess is not None for p in self.parts) else None
        # linear / box constraints only: jac(x) is the same pair of matrices for every x
        # (canonical.CanonicalConstraint.constant_jac)
        self.constant_jac = all(p.constant_jac for p in self.parts)

    def _stack_values(self, pairs):
        ineq = [a for a, _ in pairs if len(a)]
        eq = [b for _, b in pairs if len(b)]
        one = lambda parts: parts[0] if len(parts) == 1 else dv.hstack(parts)   # (no copy of one)
        return (one(ineq) if ineq else DVec.zeros(0), one(eq) if eq else DVec.zeros(0))

    def _stack_jacs(self, pairs):
        ineq = [a for a, _ in pairs if a.shape[0]]
        eq = [b for _, b in pairs if b.shape[0]]
        return (vstack_csr(ineq, self.n_vars) if ineq else self._empty,
                vstack_csr(eq, self.n_vars) if eq else self._empty)

    def constr(self, x):
        return self._stack_values([p.rows.values(p.fun(x)) for p in self.parts])

    def jac(self, x):
        if self.constant_jac:
            return self.J_ineq0, self.J_eq0
        return self._stack_jacs([p.rows.jac(p.jac(x)) for p in self.parts])

    def _hess(self, x, v_eq, v_ineq):
        terms, i_eq, i_ineq = [], 0, 0
        for p in self.parts:
            if p.hess is not None:
                v = p.rows.multipliers(v_eq[i_eq:i_eq + p.n_eq], v_ineq[i_ineq:i_ineq + p.n_ineq])
                terms.extend(_as_terms(p.hess(x, v)))
            i_eq += p.n_eq
            i_ineq += p.n_ineq
        return terms


def _as_terms(h):
    """A Hessian callback's return value as a list of device terms: one term, or a tuple /
    list of them (a constant matrix and a point-dependent diagonal, say -- summed like the
    terms of different callbacks are, _canonical_constraint.py:131-137)."""
    if isinstance(h, (tuple, list)):
        return [_as_term(t) for t in h]
    return [_as_term(h)]


def _as_term(h):
    """Normalise a Hessian callback's return value to a device term."""
    from .dense import DeviceDense
    from .operators import DeviceHessian
    if h is None or isinstance(h, (DeviceCSR, DVec, DeviceHessian, DeviceDense)) \
            or getattr(h, "device_operator", False):
        return h
    if torch.is_tensor(h) and h.dim() == 1:
        return as_dvec(h)                      # diagonal
    if torch.is_tensor(h) and h.dim() == 2 and h.is_cuda:
        return _dense_term(h)                  # dense Hessian resident in HBM
    raise TypeError("device-callback mode: a Hessian callback must return a DeviceCSR, a 1-D "
                    "CUDA tensor (diagonal), a 2-D CUDA tensor / DeviceDense (dense) or a "
                    "DeviceHessian, got %r" % type(h))


_dense_terms = {}


def _dense_term(t):
    """DeviceDense wrapper of a 2-D CUDA tensor, one per storage (a callback returning the same
    resident matrix every iteration gets the same wrapper and its cached transpose)."""
    from .dense import DeviceDense
    if t.dtype != _F64 or not t.is_contiguous():
        # the conversion copies: the caller's tensor is not kept alive by the wrapper, and a
        # callback returning a FRESH tensor per iteration gets the same address and version
        # back from the caching allocator -- the cache would serve the previous iteration's
        # matrix.  Convert, do not cache.
        return DeviceDense(t.to(_F64).contiguous())
    key = (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version)
    hit = _dense_terms.get(key)
    if hit is None:
        if len(_dense_terms) > 4:
            _dense_terms.clear()
        hit = _dense_terms[key] = (t, DeviceDense(t))       # (the entry pins t: aliased, alive)
    return hit[1]


def lagrangian_hessian(canonical, hess):
    """Device twin of canonical.lagrangian_hessian (terms in hess_list order)."""
    def lagr_hess(x, v_eq=None, v_ineq=None):
        terms = []
        if hess is not None:
            terms.extend(_as_terms(hess(x.t)))
        if canonical.hess is not None:
            terms.extend(canonical.hess(x, v_eq if v_eq is not None else DVec.zeros(0),
                                        v_ineq if v_ineq is not None else DVec.zeros(0)))
        return HessianSum(len(x), [t for t in terms if t is not None])
    return lagr_hess


_aug_cache = {}


def augmented_jacobian(J_eq, J_ineq, s, n_vars, n_eq, n_ineq):
    """[[J_eq, 0], [J_ineq, diag(s)]] for DeviceCSR blocks: the pattern and the
    destination index of every value are built once; a refresh is three scatters."""
    key = (id(J_eq.pattern), id(J_ineq.pattern))
    hit = _aug_cache.get(key)
    if hit is None:
        pe, pi = J_eq.pattern, J_ineq.pattern
        ip_e, ip_i = pe.indptr_h.astype(np.int64), pi.indptr_h.astype(np.int64)
        indptr = np.concatenate((ip_e, ip_e[-1] + ip_i[1:] + np.arange(1, n_ineq + 1)))
        nnz = int(indptr[-1])
        slots = indptr[n_eq + 1:] - 1
        dst_i = (np.arange(pi.nnz) + ip_e[-1]
                 + np.repeat(np.arange(n_ineq), np.diff(ip_i)))
        indices = np.empty(nnz, dtype=np.int32)
        indices[:pe.nnz] = pe.indices_h
        indices[dst_i] = pi.indices_h
        indices[slots] = n_vars + np.arange(n_ineq)
        pat = CSRPattern(indptr.astype(np.int32), indices, (n_eq + n_ineq, n_vars + n_ineq))
        hit = _aug_cache[key] = (pat, _idx(np.arange(pe.nnz)), _idx(dst_i), _idx(slots), pe, pi)
    pat, dst_e, dst_i, slots = hit[:4]
    val = dv._empty(pat.nnz)
    if dst_e.numel():
        scatter(J_eq.val, dst_e, val)
    if dst_i.numel():
        scatter(J_ineq.val, dst_i, val)
    scatter(s.t, slots, val)
    return DeviceCSR(pat, val)
