"""``minimize_constrained``: the public driver of the drop-in boundary.

Signature, defaults, method names, result fields, counters, status codes and
exceptions follow the reference's ``ipsolver/_minimize_constrained.py:96-565``
(cited below).  The user's callbacks are evaluated on the host with numpy
arrays, as in the reference; everything between two callback evaluations --
the trust-region subproblem solves, projections, merit-function algebra --
runs on the GPU through ``backend_hip``.  Results are returned as numpy
arrays.
"""
import os
import time
from copy import deepcopy
from warnings import warn

import numpy as np
from scipy.optimize import OptimizeResult

from . import backend as _backend
from .barrier import tr_interior_point
from .canonical import lagrangian_hessian, to_canonical, empty_canonical_constraint
from .constraints import (NonlinearConstraint, LinearConstraint, BoxConstraint, wrap_hessian)
from .fd import FiniteDifferenceOperator, FD_METHODS
from .sqp import equality_constrained_sqp

__all__ = ['minimize_constrained']

TERMINATION_MESSAGES = {
    0: "The maximum number of function evaluations is exceeded.",
    1: "`gtol` termination condition is satisfied.",
    2: "`xtol` termination condition is satisfied.",
    3: "`callback` function requested termination"
}

_METHODS = {'equality_constrained_sqp': 'equality_constrained_sqp',
            'equality-constrained-sqp': 'equality_constrained_sqp',   # docstring spelling
            'tr_interior_point': 'tr_interior_point',
            'tr-interior-point': 'tr_interior_point'}

_VECTOR_FIELDS = ("x", "v", "s", "grad", "constr")


class _Memoize:
    """Gradient cache used with finite-difference Hessians
    (_minimize_constrained.py:28-40; keeps a reference to x, not a copy)."""

    def __init__(self, fun, x0, f0):
        self.fun, self._x, self._f = fun, x0, f0

    def __call__(self, x):
        if not np.array_equal(x, self._x):
            self._x = x
            self._f = self.fun(x)
        return self._f


def _print_header(method):
    if method == 'equality_constrained_sqp':
        cols = ("niter", "f evals", "CG iter", "tr radius", "penalty", "opt", "c viol")
        widths = (7, 7, 7, 10, 10, 10, 10)
    else:
        cols = ("niter", "f evals", "CG iter", "barrier param", "tr radius", "penalty", "opt",
                "c viol")
        widths = (7, 7, 7, 13, 10, 10, 10, 10)
    print("|" + "|".join("{0:^{1}}".format(c, w) for c, w in zip(cols, widths)) + "|")
    print("|" + "|".join("-" * (w - 1) + ":" if w == 7 else ":" + "-" * (w - 2) + ":"
                         for w in widths) + "|")


def _print_iter(method, state):
    head = "|{0:>7}|{1:>7}|{2:>7}|".format(state.niter, state.nfev, state.cg_niter)
    vals = [state.trust_radius, state.penalty, state.optimality, state.constr_violation]
    if method == 'tr_interior_point':
        head += "   {0:^1.2e}  |".format(state.barrier_parameter)
    print(head + "".join(" {0:^1.2e} |".format(v) for v in vals))


def _is_cuda_tensor(x):
    try:
        import torch
    except ImportError:
        return False
    return torch.is_tensor(x) and x.is_cuda


def _make_stop_criteria(method, gtol, xtol, max_iter, barrier_tol, callback, verbose, view):
    """The two closures of _minimize_constrained.py:460-502 (status codes 0-3)."""
    interior = method == 'tr_interior_point'

    def stop_criteria(state):
        if verbose >= 2:
            _print_iter(method, state)
        state.status = None
        if callback is not None and callback(view(state)):
            state.status = 3
        elif state.optimality < gtol and state.constr_violation < gtol:
            state.status = 1
        elif state.trust_radius < xtol and (not interior
                                            or state.barrier_parameter < barrier_tol):
            state.status = 2
        elif state.niter > max_iter:
            state.status = 0
        return state.status in (0, 1, 2, 3)
    return stop_criteria


def _minimize_device(fun, x0, grad, hess, constraints, method, xtol, gtol, options, callback,
                     max_iter, verbose, xp):
    """Device-callback mode: ``x0`` is a CUDA tensor, callbacks take and return
    device objects (see device_mode.py); results carry CUDA tensors."""
    from . import device_mode as dm
    if xp.name != "hip":
        raise RuntimeError("device-callback mode needs the HIP backend")
    if hess in FD_METHODS:           # N4: differences of the device gradient callback
        from .fd import DeviceFiniteDifferenceOperator
        fd_method = hess
        hess = lambda xt: DeviceFiniteDifferenceOperator(grad, dm.as_dvec(xt), fd_method)
    x0_dev = dm.as_dvec(x0.detach().clone())
    n_vars = len(x0_dev)
    f0 = float(fun(x0_dev.t))
    g0 = dm.as_dvec(grad(x0_dev.t))
    if isinstance(constraints, (NonlinearConstraint, LinearConstraint, BoxConstraint)):
        constraints = [constraints]
    canon = dm.DeviceCanonical(list(constraints), x0_dev)
    x0_dev = canon.x0
    lagr = dm.lagrangian_hessian(canon, hess if callable(hess) else None)

    state = OptimizeResult(niter=0, nfev=1, ngev=1, ncev=1, njev=1, nhev=0, cg_niter=0,
                           cg_info={})
    options = dict(options)
    return_all = options.get("return_all", False)
    if return_all:
        state.allvecs, state.allmult = [], []
    if method is None:
        method = 'equality_constrained_sqp' if canon.n_ineq == 0 else 'tr_interior_point'
    if method not in _METHODS:
        raise ValueError("Unknown optimization ``method``.")
    method = _METHODS[method]
    barrier_tol = options.pop("barrier_tol", 1e-8)

    def tensors(state):
        view = OptimizeResult(state)
        for k in _VECTOR_FIELDS:
            if k in view and hasattr(view[k], "t"):
                view[k] = view[k].t
        return view
    stop_criteria = _make_stop_criteria(method, gtol, xtol, max_iter, barrier_tol, callback,
                                        verbose, tensors)
    if verbose >= 2:
        _print_header(method)
    start_time = time.time()
    from . import sqp as _sqp
    on_device, _sqp._CALLBACKS_ON_DEVICE = _sqp._CALLBACKS_ON_DEVICE, True
    try:
        result = _run_device_method(method, canon, fun, grad, lagr, n_vars, x0_dev, f0, g0,
                                    stop_criteria, state, xp, xtol, options, dm)
    finally:
        _sqp._CALLBACKS_ON_DEVICE = on_device
    result.execution_time = time.time() - start_time
    result.method = method
    result.message = TERMINATION_MESSAGES[result.status]
    for k in _VECTOR_FIELDS:
        if k in result and hasattr(result[k], "t"):
            result[k] = result[k].t
    if return_all:
        for k in ("allvecs", "allmult", "allslack"):
            if k in result:
                result[k] = [t.t for t in result[k]]
    if verbose >= 1:
        print(result.message)
    return result


def _run_device_method(method, canon, fun, grad, lagr, n_vars, x0_dev, f0, g0, stop_criteria,
                       state, xp, xtol, options, dm):
    if method == 'equality_constrained_sqp':
        if canon.n_ineq > 0:
            raise ValueError("'equality_constrained_sqp' does not support "
                             "inequality constraints.")
        if canon.constant_jac:
            xp.mark_constant(canon.J_eq0)      # one factorization for the whole run (N1)
        result = equality_constrained_sqp(
            lambda x: (dm.objective_value(fun(x.t)), canon.constr(x)[1]),
            lambda x: (dm.as_dvec(grad(x.t)), canon.jac(x)[1]),
            lambda x, v: xp.hessian_operator(lagr(x, v), n_vars, None),
            x0_dev, f0, g0, canon.c_eq0, canon.J_eq0, stop_criteria, state, xp, **options)
    else:
        if canon.n_ineq == 0:
            warn("The problem only has equality constraints. The solver "
                 "'equality_constrained_sqp' is a better choice for those situations.")
        result = tr_interior_point(
            lambda x: dm.objective_value(fun(x.t)), lambda x: dm.as_dvec(grad(x.t)), lagr,
            n_vars,
            canon.n_ineq, canon.n_eq, canon.constr, canon.jac, x0_dev, f0, g0, canon.c_ineq0,
            canon.J_ineq0, canon.c_eq0, canon.J_eq0, stop_criteria, canon.enforce_feasibility,
            xtol, state, xp, **options)
    return result


def _minimize_distributed(fun, x0, grad, hess, constraints, method, xtol, gtol, options,
                          callback, max_iter):
    """``minimize_constrained`` with every callback on DISTRIBUTED data (one process per GPU):
    ``x0`` is a ``sharded.ShardVec`` -- the rank's variables plus halo copies of a block either
    side, on its device --, and so is every vector the callbacks see or return:

        fun(x) -> float (the global value: ShardVec.dot reduces over the ranks)
        grad(x) -> ShardVec
        hess(x) -> ShardHessian | ShardVec (a diagonal) | a tuple of such terms
        NonlinearConstraint(fun, kind, jac, hess):  fun(x) -> ShardVec over the constraint rows,
            jac(x) -> ShardCSR,  hess(x, v) -> as ``hess`` (v: the rows' multipliers)
        BoxConstraint(('interval', lb, ub)) with scalar bounds

    Two shapes, as on the halo partition of the host-callback route (``_sharded_backend``):
    one nonlinear constraint of kind ``('equals', 0)`` (BASELINE configs 3 / 4, either method)
    and one of kind ``('less', 0)`` with an interval box on every variable (config 5).  Nothing
    is gathered to a host or to one rank (reference _minimize_constrained.py:374-565 evaluates
    ``fun(x)`` on the whole vector; here x never exists in one place)."""
    from . import sharded
    sh = x0.sh
    if isinstance(constraints, (NonlinearConstraint, LinearConstraint, BoxConstraint)):
        constraints = [constraints]
    nl = [c for c in constraints if isinstance(c, NonlinearConstraint)]
    boxes = [c for c in constraints if isinstance(c, BoxConstraint)]
    if len(nl) != 1 or len(nl) + len(boxes) != len(list(constraints)) or len(boxes) > 1:
        raise NotImplementedError("distributed callbacks: one NonlinearConstraint, optionally "
                                  "with one BoxConstraint")
    con = nl[0]
    if not callable(hess) or not callable(con._hess):
        raise NotImplementedError("distributed callbacks: exact Hessian callbacks (finite "
                                  "differences evaluate host callbacks)")
    nloc = sh.local_len("col")

    def terms_of(h):
        return [t for t in (h if isinstance(h, (tuple, list)) else [h]) if t is not None]

    def lagr_hess(x, v):
        terms = terms_of(hess(x)) + terms_of(con._hess(x, v))
        mats = [t for t in terms if isinstance(t, sharded.ShardHessian)]
        diags = [t for t in terms if isinstance(t, sharded.ShardVec)]
        if len(mats) == 1 and len(mats) + len(diags) == len(terms) and hasattr(mats[0], "_csr"):
            # ONE local matrix + diagonal terms: one operator the device-resident loop takes
            d = None
            for t in diags:
                d = t if d is None else d + t
            return sharded.ShardHessian(sh, sh.ops.hessian(nloc, mats[0]._csr,
                                                           d.loc if d is not None else None))
        parts = [t if not isinstance(t, sharded.ShardVec) else sharded._DiagOp(t) for t in terms]
        return sharded.OperatorSum(sh, parts)

    kind = con.kind
    options = dict(options)
    options.pop("barrier_tol", None)
    if not boxes:
        if not (isinstance(kind, (tuple, list)) and kind[0] == "equals" and np.all(np.asarray(kind[1]) == 0)):
            raise NotImplementedError("distributed callbacks: kind ('equals', 0) without a box")
        return sharded.minimize_equality_constrained(
            sh, fun, grad, lagr_hess, con._fun, con._jac, x0, method=method, xtol=xtol, gtol=gtol,
            max_iter=max_iter, callback=callback, **options)
    bk = boxes[0].kind
    if not (isinstance(kind, (tuple, list)) and kind[0] == "less" and np.all(np.asarray(kind[1]) == 0)
            and isinstance(bk, (tuple, list)) and bk[0] == "interval"
            and np.ndim(bk[1]) == 0 and np.ndim(bk[2]) == 0):
        raise NotImplementedError("distributed callbacks: kind ('less', 0) with a BoxConstraint("
                                  "('interval', lb, ub)) of scalar bounds")
    if _METHODS.get(method or 'tr_interior_point') != 'tr_interior_point':
        raise ValueError("'equality_constrained_sqp' does not support inequality constraints.")
    return sharded.minimize_box_inequality(
        sh, fun, grad, lagr_hess, con._fun, con._jac, x0, sh.full("col", float(bk[1])),
        sh.full("col", float(bk[2])), xtol=xtol, gtol=gtol, max_iter=max_iter, callback=callback,
        **options)


class _ConstantArray(np.ndarray):
    """The value of a Hessian callback the caller declared constant (``options=
    {'constant_hessian': True}``): backend_hip keeps ONE device copy of it."""
    _ipx_constant = True


def _shard_request(options):
    """``options['shard']``: True / False, or the local-arithmetic object of the row-sharded
    backend (tests pass the numpy twin).  Sharding is OPT-IN (the option, or ``IPX_SHARD=1`` in
    the environment of every rank): a process group that merely exists does not turn a call
    into a collective one -- a torchrun job running one independent solve per GPU (device
    callbacks, rank-dependent problems) keeps working.  The sharded call is collective: every
    rank of the default group must make it with identical arguments."""
    shard = options.pop("shard", None)
    if shard is None:
        shard = os.environ.get("IPX_SHARD", "0") not in ("", "0")
    return shard


def _sharded_backend(shard, constr, n_vars, operator_hessian=False):
    """The row-sharded backend (one process per GPU) for this problem.  Two shapes run on the
    banded partition with its device-resident loop (ipsolver/sharded.py): equality rows only,
    partitioned along their banded Jacobian (BASELINE configs 3 / 4), and nonlinear inequality
    rows + an interval box on every variable, partitioned along the nonlinear rows (config 5);
    a third -- equality and inequality rows interleaved along one band -- on the same partition
    in the merged row order (ipsolver/sharded_mixed.py).
    Everything else the reference accepts with a sparse Jacobian -- no band, rows whose kinds
    do not follow the band, ragged or no boxes -- runs on the plain block partition with
    all-gather / reduce-scatter products (ipsolver/sharded_general.py).  ``operator_hessian``:
    some Hessian term is a host operator (finite differences, LinearOperator); the box form of
    the barrier problem has no operator for those, the plain partition has."""
    import scipy.sparse as sps
    from . import sharded, sharded_general
    ops = shard if hasattr(shard, "from_host") else sharded.HipOps()
    comm = sharded.ShardComm()
    n_eq, n_ineq = constr.n_eq, constr.n_ineq
    if n_eq + n_ineq == 0:
        raise NotImplementedError("row-sharded solve needs constraint rows to partition")
    dense = not (sps.issparse(constr.J_eq0) and sps.issparse(constr.J_ineq0))

    def general():
        try:
            sh = sharded_general.general_sharding((n_eq, n_vars), ops, comm,
                                                  {"ineq": n_ineq} if n_ineq else None)
        except ValueError as exc:
            raise NotImplementedError("row-sharded solve: %s" % exc)
        return sharded_general.GeneralBackend(sh, n_ineq)

    J, boxed = None, False
    if dense:
        pass           # dense Jacobians: rows of a full CSR on the plain partition
    elif n_ineq == 0:
        J = constr.J_eq0
    elif n_eq == 0 and n_ineq > 2 * n_vars:
        # nonlinear rows followed by all lower bounds, then all upper bounds
        # (_canonical_constraint.py:350-355) of an interval box on every variable?
        m_nl = n_ineq - 2 * n_vars
        eye = sps.identity(n_vars, format="csr")
        tail = sps.csr_matrix(constr.J_ineq0)[m_nl:]
        if tail.shape[1] == n_vars and not operator_hessian \
                and (tail != sps.vstack([-eye, eye], format="csr")).nnz == 0:
            J, boxed = sps.csr_matrix(constr.J_ineq0)[:m_nl], True
    if J is None and not dense and n_eq > 0 and n_ineq > 0 and not operator_hessian:
        # equality and inequality rows of ONE banded Jacobian (the kinds interleaved along the
        # band): the banded partition in the merged row order (ipsolver/sharded_mixed.py)
        from . import sharded_mixed
        xp = sharded_mixed.try_backend(sps.csr_matrix(constr.J_eq0), sps.csr_matrix(constr.J_ineq0),
                                       n_vars, ops, comm)
        if xp is not None:
            return xp
    if J is None:
        return general()
    J = sps.csr_matrix(J)
    J.sort_indices()
    try:
        lay = sharded.ShardLayout(J.indptr, J.indices, J.shape, comm.world, comm.rank)
    except (NotImplementedError, ValueError):
        return general()         # no band to follow, or too few row blocks for the ranks
    sh = sharded.Sharding(lay, comm, ops)
    xp = sharded.ShardedBackend(sh)
    if boxed:
        sh.register(xp.INEQ)
        sh.register(xp.Z)
    else:
        # the halo partition rests on (A A')^-1 decaying across one block of rows, which the
        # projector measures on the numbers (sharded.ShardProjector._check_truncation): ask once
        # at the initial Jacobian and take the plain partition if it refuses (every rank
        # arrives at the same answer: the check is collective)
        try:
            xp.projections(xp.matrix(J))
        except NotImplementedError:
            return general()
    return xp


def minimize_constrained(fun, x0, grad, hess='2-point', constraints=(), method=None,
                         xtol=1e-8, gtol=1e-8, sparse_jacobian=None, options={},
                         callback=None, max_iter=1000, verbose=0):
    """Minimize a scalar function subject to constraints (see the reference's
    docstring, _minimize_constrained.py:101-372, for the full parameter list).

    ``method`` is ``'equality_constrained_sqp'`` or ``'tr_interior_point'``
    (hyphenated spellings are accepted too); ``None`` picks by constraint type.
    Returns a ``scipy.optimize.OptimizeResult`` with the reference's fields.

    Launched as one process per GPU with ``options={'shard': True}`` (or ``IPX_SHARD=1``; a
    collective call: identical arguments on every rank of the initialised ``torch.distributed``
    group) the same call runs on the row-sharded backend: the user's
    callbacks are evaluated on the host with global numpy arrays exactly as here (replicated
    on every rank), every vector, the Jacobian and the Hessian between two evaluations are
    partitioned over the ranks (ipsolver/sharded.py), the result carries global arrays.
    """
    options = dict(options)
    shard = _shard_request(options)
    # (an ADDITIVE option, see below; popped before any dispatch so that it never reaches the
    # outer loops' keyword arguments: device-callback mode keeps its Hessians on the device
    # anyway, and a non-callable ``hess`` -- finite differences, quasi-Newton -- has no constant
    # value to keep)
    constant_hessian = bool(options.pop("constant_hessian", False))
    if hasattr(x0, "sh") and hasattr(x0, "owns"):
        # a DISTRIBUTED start vector (sharded.ShardVec): device-callback mode on the row-sharded
        # backend -- the callbacks take and return distributed objects, nothing is gathered
        return _minimize_distributed(fun, x0, grad, hess, constraints, method, xtol, gtol,
                                     options, callback, max_iter)
    if _is_cuda_tensor(x0):
        if shard:
            raise NotImplementedError(
                "row-sharded solve with device callbacks: pass the start vector as a distributed "
                "vector (ipsolver.sharded.ShardVec: its layout tells every callback which rows "
                "and variables are this rank's) -- see minimize._minimize_distributed; a plain "
                "CUDA tensor carries no partition")
        return _minimize_device(fun, x0, grad, hess, constraints, method, xtol, gtol, options,
                                callback, max_iter, verbose, _backend.get())
    xp = None if shard else _backend.get()
    x0 = np.atleast_1d(x0).astype(float)                     # :374-379
    n_vars = np.size(x0)
    f0 = fun(x0)
    g0 = np.atleast_1d(grad(x0))

    def plain_grad(x):
        return np.atleast_1d(grad(x))
    grad_wrapped = _Memoize(plain_grad, x0, g0) if hess in FD_METHODS else plain_grad

    if callable(hess) and constant_hessian:
        # ADDITIVE option (the reference has none; its signature is unchanged): the objective's
        # Hessian does not depend on x -- a quadratic objective.  ``hess`` is called ONCE (the
        # reference calls it at every accepted step, _minimize_constrained.py:395-407; ``nhev``
        # still counts those evaluations) and its value kept as an immutable array, so the
        # backend uploads a dense Hessian once instead of once per outer iteration (BASELINE
        # config 2: fifteen uploads of 800 MB, 0.9 of the solve's 1.6 s).
        H0 = wrap_hessian(hess, hess(x0), 1)(x0)
        if isinstance(H0, np.ndarray):
            H0 = H0.view(_ConstantArray)         # (a marked VIEW: the caller's array is untouched)

        def hess_wrapped(x, _H0=H0):
            return _H0
    elif callable(hess):                                     # :395-422
        hess_wrapped = wrap_hessian(hess, hess(x0), 1)
    elif hess in FD_METHODS:
        def hess_wrapped(x):
            return FiniteDifferenceOperator(grad_wrapped, x, hess)
    else:
        hess_wrapped = hess

    if isinstance(constraints, (NonlinearConstraint, LinearConstraint, BoxConstraint)):
        constraints = [constraints]                          # :425-437
    copied = [deepcopy(c) for c in constraints]
    for c in copied:
        x0 = c.evaluate_and_initialize(x0, sparse_jacobian)
    constr = (empty_canonical_constraint(x0, n_vars, sparse_jacobian) if len(copied) == 0
              else to_canonical(copied))
    host_lagr_hess = lagrangian_hessian(constr, hess_wrapped)
    if shard:
        from .constraints import _is_operator
        op_hess = hess in FD_METHODS or (callable(hess) and _is_operator(hess(x0))) \
            or any(isinstance(c, NonlinearConstraint) and isinstance(c._hess, str)
                   and c._hess in FD_METHODS for c in copied)
        xp = _sharded_backend(shard, constr, n_vars, op_hess)

    state = OptimizeResult(niter=0, nfev=1, ngev=1, ncev=1, njev=1, nhev=0,
                           cg_niter=0, cg_info={})           # :443-450
    return_all = options.get("return_all", False)
    if return_all:
        state.allvecs, state.allmult = [], []

    if method is None:                                       # :453-457
        method = 'equality_constrained_sqp' if constr.n_ineq == 0 else 'tr_interior_point'
    if method not in _METHODS:
        raise ValueError("Unknown optimization ``method``.")
    method = _METHODS[method]
    interior = method == 'tr_interior_point'
    barrier_tol = options.get("barrier_tol", 1e-8)

    def user_view(state):
        """State as the user's callback sees it: host arrays."""
        view = OptimizeResult(state)
        for k in _VECTOR_FIELDS:
            if k in view:
                view[k] = xp.tohost(view[k])
        return view

    def stop_criteria(state):                                # :460-502
        if verbose >= 2:
            _print_iter(method, state)
        state.status = None
        if callback is not None and callback(user_view(state)):
            state.status = 3
        elif state.optimality < gtol and state.constr_violation < gtol:
            state.status = 1
        elif state.trust_radius < xtol and (not interior
                                            or state.barrier_parameter < barrier_tol):
            state.status = 2
        elif state.niter > max_iter:
            state.status = 0
        return state.status in (0, 1, 2, 3)

    # ---- host callbacks seen from the device loops --------------------------
    def fun_dev(x):
        return fun(xp.tohost(x))

    def grad_dev(x):
        return xp.asvec(grad_wrapped(xp.tohost(x)), space="x")

    def constr_dev(x):
        c_ineq, c_eq = constr.constr(xp.tohost(x))
        return xp.asvec(c_ineq, space="ineq"), xp.asvec(c_eq, space="eq")

    def jac_host(x):
        return constr.jac(xp.tohost(x))

    if verbose >= 2:
        _print_header(method)
    start_time = time.time()
    x0_dev, g0_dev = xp.asvec(x0, space="x"), xp.asvec(g0, space="x")
    if not interior:                                         # :512-530
        if constr.n_ineq > 0:
            raise ValueError("'equality_constrained_sqp' does not support "
                             "inequality constraints.")

        def fun_and_constr(x):
            xh = xp.tohost(x)
            _, c_eq = constr.constr(xh)
            return fun(xh), xp.asvec(c_eq, space="eq")

        A0 = xp.matrix(constr.J_eq0)
        if constr.constant_jac:
            xp.mark_constant(A0)       # one upload, one factorization for the whole run (N1)

        def grad_and_jac(x):
            xh = xp.tohost(x)
            if constr.constant_jac:
                return xp.asvec(grad_wrapped(xh), space="x"), A0
            _, J_eq = constr.jac(xh)
            return xp.asvec(grad_wrapped(xh), space="x"), xp.matrix(J_eq)

        def lagr_hess(x, v):
            terms = host_lagr_hess(xp.tohost(x), xp.tohost(v))
            return xp.hessian_operator(terms, n_vars, None)

        result = equality_constrained_sqp(
            fun_and_constr, grad_and_jac, lagr_hess, x0_dev, f0, g0_dev,
            xp.asvec(constr.c_eq0, space="eq"), A0, stop_criteria, state, xp,
            **options)
    else:                                                    # :532-544
        if constr.n_ineq == 0:
            warn("The problem only has equality constraints. The solver "
                 "'equality_constrained_sqp' is a better choice for those situations.")

        def lagr_hess_terms(x, v_eq, v_ineq):
            return host_lagr_hess(xp.tohost(x), xp.tohost(v_eq), xp.tohost(v_ineq))

        options.pop("barrier_tol", None)
        result = tr_interior_point(
            fun_dev, grad_dev, lagr_hess_terms, n_vars, constr.n_ineq, constr.n_eq,
            constr_dev, jac_host, x0_dev, f0, g0_dev, xp.asvec(constr.c_ineq0, space="ineq"),
            constr.J_ineq0, xp.asvec(constr.c_eq0, space="eq"), constr.J_eq0, stop_criteria,
            constr.enforce_feasibility, xtol, state, xp, **options)

    result.execution_time = time.time() - start_time         # :548-564
    result.method = method
    result.message = TERMINATION_MESSAGES[result.status]
    for k in _VECTOR_FIELDS:
        if k in result:
            result[k] = xp.tohost(result[k])
    if "jac" in result and hasattr(result.jac, "to_scipy"):
        result.jac = result.jac.to_scipy()
    elif "jac" in result and hasattr(result.jac, "to_host"):
        result.jac = result.jac.to_host()
    if return_all:
        for k in ("allvecs", "allmult", "allslack"):
            if k in result:
                result[k] = [xp.tohost(t) for t in result[k]]
    if verbose >= 2:
        print("")
        print((7 * 3 + 10 * 4 + (8 if not interior else 22)) * "-")
        print("")
    if verbose >= 1:
        print(result.message)
        print("Number of iteractions: {0}, function evaluations: {1}, "
              "CG iterations: {2}, optimality: {3:.2e}, "
              "constraint violation: {4:.2e}, execution time: {5:4.2} s."
              .format(result.niter, result.nfev, result.cg_niter, result.optimality,
                      result.constr_violation, result.execution_time))
    return result
