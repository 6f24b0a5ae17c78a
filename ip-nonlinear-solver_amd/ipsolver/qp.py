"""Trust-region QP subproblem solvers on device vectors.

Same functions, argument meaning, return values and error behaviour as the
reference's ``_large_scale_constrained/qp_subproblem.py``; vectors are
``DVec`` (lists / ndarrays are uploaded), operators are anything with
``.dot`` over ``DVec``.  Every O(n) operation is an ipx kernel; the scalar
branch logic stays on the host exactly as written in the reference, fed by
fused single-pass reductions (one read-back per decision point).

``projected_cg`` here is the general, operator-agnostic driver (arbitrary
``H``/``Z``/``Y`` callables, ``return_all``, box handling).  When the
operators are the library's own device types the call is routed to the
device-resident fused loop in ``cg_fused.py`` (no per-iteration host
round-trip); both produce the same iterates.
"""
from math import copysign, sqrt

import numpy as np

from . import device as dv
from .device import DVec

__all__ = ['sphere_intersections', 'box_intersections', 'box_sphere_intersections',
           'inside_box_boundaries', 'reinforce_box_boundaries', 'modified_dogleg',
           'projected_cg']

_TINY = 1e-25   # CLOSE_TO_ZERO, qp_subproblem.py:496
_INF = float("inf")


def _vec(x):
    # DVec, or a distributed vector with the same surface (sharded.ShardVec)
    return x if hasattr(x, "sumsq_amax") else DVec.from_host(x)


def _optvec(x):
    return None if x is None else _vec(x)


# ---- scalar tails of the intersection routines --------------------------
def _sphere_from_scalars(dd, zd, zz, trust_radius, entire_line):
    """qp_subproblem.py:99-149 given d.d, z.d, z.z."""
    if dd == 0:
        return 0, 0, False
    # (a finite radius whose square overflows -- 1e300, "no trust region" in some callers -- is a
    # sphere no double-precision step reaches: the reference's ``trust_radius**2`` raises
    # OverflowError there; here it is the infinite sphere)
    r2 = float(trust_radius) * float(trust_radius)
    if np.isinf(r2):
        return (-_INF, _INF, True) if entire_line else (0, 1, True)
    a = dd
    b = 2 * zd
    c = zz - r2
    disc = b * b - 4 * a * c
    if disc < 0:
        return 0, 0, False
    aux = b + copysign(sqrt(disc), b)
    with np.errstate(divide='ignore', invalid='ignore'):
        ta, tb = sorted([float(-np.float64(aux) / (2 * a)),
                         float(-2 * c / np.float64(aux))])
    if entire_line:
        return ta, tb, True
    if tb < 0 or ta > 1:
        return 0, 0, False
    return max(0, ta), min(1, tb), True


def _box_from_scalars(dd, ta, tb, zero_d_outside, entire_line):
    """qp_subproblem.py:194-234 given max-of-min / min-of-max."""
    if dd == 0:
        return 0, 0, False
    if zero_d_outside > 0:
        return 0, 0, False
    intersect = bool(ta <= tb)
    if not entire_line:
        if tb < 0 or ta > 1:
            return 0, 0, False
        ta, tb = max(0, ta), min(1, tb)
    return ta, tb, intersect


def _reduce(z, d, dscale, lb, ub):
    return dv.box_sphere_reduce(z, d, dscale, lb, ub)


def sphere_intersections(z, d, trust_radius, entire_line=False):
    """Reference: qp_subproblem.py:66-149."""
    r = _reduce(_vec(z), _vec(d), 1.0, None, None)
    return _sphere_from_scalars(r[0], r[1], r[2], trust_radius, entire_line)


def box_intersections(z, d, lb, ub, entire_line=False):
    """Reference: qp_subproblem.py:152-234."""
    r = _reduce(_vec(z), _vec(d), 1.0, _vec(lb), _vec(ub))
    return _box_from_scalars(r[0], r[3], r[4], r[5], entire_line)


def box_sphere_intersections(z, d, lb, ub, trust_radius, entire_line=False,
                             extra_info=False, dscale=1.0):
    """Reference: qp_subproblem.py:237-303.  ``dscale`` (extension) evaluates
    the direction ``dscale*d`` without materialising it (the reference passes
    ``alpha*p``, :585,605)."""
    r = _reduce(_vec(z), _vec(d), dscale, _optvec(lb), _optvec(ub))
    ta_b, tb_b, hit_b = _box_from_scalars(r[0], r[3], r[4], r[5], entire_line)
    ta_s, tb_s, hit_s = _sphere_from_scalars(r[0], r[1], r[2], trust_radius, entire_line)
    ta = np.maximum(ta_b, ta_s)
    tb = np.minimum(tb_b, tb_s)
    intersect = bool(hit_b and hit_s and ta <= tb)
    if extra_info:
        return (ta, tb, intersect,
                {'ta': ta_s, 'tb': tb_s, 'intersect': hit_s},
                {'ta': ta_b, 'tb': tb_b, 'intersect': hit_b})
    return ta, tb, intersect


def inside_box_boundaries(x, lb, ub):
    """Reference: qp_subproblem.py:306-308."""
    if lb is None and ub is None:
        return True
    return dv.count_outside_box(_vec(x), _vec(lb), _vec(ub)) == 0


def reinforce_box_boundaries(x, lb, ub):
    """Reference: qp_subproblem.py:310-317."""
    if lb is None and ub is None:
        return x
    return dv.clip(_vec(x), _vec(lb), _vec(ub))


def _full(n, value):
    return DVec.full(n, value)


def modified_dogleg(A, Y, b, trust_radius, lb, ub, norm_out=None):
    """Reference: qp_subproblem.py:320-413.  ``norm_out`` (extension: a list) receives the
    norm of the returned step when the routine has it anyway (the accepted Newton point), so
    the caller need not read it back a second time."""
    b = _vec(b)
    lb, ub = _optvec(lb), _optvec(ub)       # None = no bound on that side
    newton = -Y.dot(b)
    if (lb is None) != (ub is None):
        lb = lb if lb is not None else newton.full_like(-_INF)
        ub = ub if ub is not None else newton.full_like(_INF)
    if inside_box_boundaries(newton, lb, ub):
        norm_newton = dv.norm(newton)
        if norm_newton <= trust_radius:
            if norm_out is not None:
                norm_out.append(norm_newton)
            return newton

    g = A.T.dot(b)
    Ag = A.dot(g)
    cauchy = (-g.dot(g) / Ag.dot(Ag)) * g
    origin = cauchy.zeros_like()

    step = newton - cauchy
    _, alpha, hit = box_sphere_intersections(cauchy, step, lb, ub, trust_radius)
    if hit:
        x1 = cauchy + alpha * step
    else:
        _, alpha, _ = box_sphere_intersections(origin, cauchy, lb, ub, trust_radius)
        x1 = origin + alpha * cauchy

    _, alpha, _ = box_sphere_intersections(origin, newton, lb, ub, trust_radius)
    x2 = origin + alpha * newton

    if dv.norm(A.dot(x1) + b) < dv.norm(A.dot(x2) + b):
        return x1
    return x2


def projected_cg(H, c, Z, Y, b, trust_radius=np.inf, lb=None, ub=None, tol=None,
                 max_iter=None, max_infeasible_iter=None, return_all=False):
    """Reference: qp_subproblem.py:416-643 (same stop codes and info dict).  ``b=None``
    (extension) stands for b = 0, the SQP's call (equality_constrained_sqp.py:126): the
    starting point ``Y.dot(-b)`` is then the zero vector by construction and its solve, its
    Hessian product and its norm are not formed (same values: 0, and 0 + c = c exactly)."""
    from . import cg_fused
    b_zero = b is None
    if not b_zero and not hasattr(b, "sumsq_amax"):
        # a HOST right-hand side (the reference's calling convention: numpy arrays) that is all
        # zero is the same call: one pass over m doubles on the host instead of a solve, two
        # products and an upload on the device.  (A device vector is taken as it comes: whether
        # it is zero is not known without reading it back.)
        bh = np.asarray(b, dtype=float)
        if bh.ndim == 1 and not bh.any():
            b_zero = True
    c = _vec(c)
    fused = not return_all and cg_fused.supports(H, Z, Y)
    if b_zero:
        P = getattr(Z, "projector", None)
        if hasattr(P, "sh"):
            b = P.sh.zeros(P.A.row_kind)
        elif fused:
            b = DVec(dv._empty(Y.shape[1]))      # (its length is all the fused path asks of it)
        else:
            b = DVec.zeros(Y.shape[1])
    b = _vec(b)
    lb, ub = _optvec(lb), _optvec(ub)
    if fused:
        return cg_fused.projected_cg(H, c, Z, Y, b, trust_radius, lb, ub, tol,
                                     max_iter, max_infeasible_iter, b_zero=b_zero)
    if not return_all and getattr(getattr(Z, "projector", None), "fused_sharded", False):
        from . import sharded            # the device-resident loop of the row-sharded solver
        if sharded.fused_supports(H, Z, Y):
            return sharded.fused_projected_cg(H, c, Z, Y, b, trust_radius, lb, ub, tol,
                                              max_iter, max_infeasible_iter)
    n, m = len(c), len(b)
    has_box = lb is not None or ub is not None
    if has_box:
        lb = lb if lb is not None else c.full_like(-_INF)
        ub = ub if ub is not None else c.full_like(_INF)

    x = Y.dot(-b)                                        # :502-505
    r = Z.dot(H.dot(x) + c)
    g = Z.dot(r)
    p = -g
    allvecs = [x] if return_all else None
    H_p = H.dot(p)                                       # :511-512
    rt_g = dv.norm(g) ** 2

    tr_distance = trust_radius - dv.norm(x)              # :515-526
    if tr_distance < 0:
        raise ValueError("Trust region problem does not have a solution.")
    if tr_distance < _TINY:
        info = {'niter': 0, 'stop_cond': 2, 'hits_boundary': True}
        if return_all:
            allvecs.append(x)
            info['allvecs'] = allvecs
        return x, info

    if tol is None:                                      # :529-542
        tol = max(min(0.01 * np.sqrt(rt_g), 0.1 * rt_g), _TINY)
    if max_iter is None:
        max_iter = n - m
    max_iter = min(max_iter, n - m)
    if max_infeasible_iter is None:
        max_infeasible_iter = n - m

    hits_boundary = False
    stop_cond = 1
    counter = 0
    last_feasible_x = c.zeros_like()    # reference: np.empty_like (:547)
    k = 0
    for _ in range(max_iter):
        if rt_g < tol:                                   # :551
            stop_cond = 4
            break
        k += 1
        pt_H_p = H_p.dot(p)                              # :556
        if pt_H_p <= 0:                                  # :558-576
            if np.isinf(trust_radius):
                raise ValueError("Negative curvature not allowed "
                                 "for unrestrited problems.")
            _, alpha, hit = box_sphere_intersections(x, p, lb, ub, trust_radius,
                                                     entire_line=True)
            if hit:
                x = x.add_scaled(p, alpha)
            x = reinforce_box_boundaries(x, lb, ub)
            stop_cond = 3
            hits_boundary = True
            break

        alpha = rt_g / pt_H_p                            # :579-580
        x_next = x.add_scaled(p, alpha)

        if dv.norm(x_next) >= trust_radius:              # :583-596
            _, theta, hit = box_sphere_intersections(x, p, lb, ub, trust_radius,
                                                     dscale=alpha)
            if hit:
                x = x.add_scaled(p, theta * alpha)
            x = reinforce_box_boundaries(x, lb, ub)
            stop_cond = 2
            hits_boundary = True
            break

        if inside_box_boundaries(x_next, lb, ub):        # :599-616
            counter = 0
        else:
            counter += 1
        if counter > 0:
            _, theta, hit = box_sphere_intersections(x, p, lb, ub, trust_radius,
                                                     dscale=alpha)
            if hit:
                last_feasible_x = reinforce_box_boundaries(
                    x.add_scaled(p, theta * alpha), lb, ub)
                counter = 0
        if counter > max_infeasible_iter:
            break
        if return_all:
            allvecs.append(x_next)

        r_next = r.add_scaled(H_p, alpha)                        # :622
        g_next = Z.dot(r_next)                           # :624
        rt_g_next = dv.norm(g_next) ** 2                 # :626
        beta = rt_g_next / rt_g
        p = p.scaled_sub(beta, g_next)                       # :628
        x = x_next
        g = g_next
        r = g_next                                       # sic, :632
        rt_g = rt_g_next                                 # :633 recomputes the same value
        H_p = H.dot(p)                                   # :634

    if not inside_box_boundaries(x, lb, ub):             # :636-638
        x = last_feasible_x
        hits_boundary = True
    info = {'niter': k, 'stop_cond': stop_cond, 'hits_boundary': hits_boundary}
    if return_all:
        info['allvecs'] = allvecs
    return x, info
