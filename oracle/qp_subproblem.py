"""Oracle (test infrastructure): numpy restatement of the trust-region QP helpers.

Follows ``ipsolver/_large_scale_constrained/qp_subproblem.py`` of the
reference; each function names the lines it restates.  Operators ``H, Z, Y,
A`` are duck typed (``.dot``; ``A`` also ``.T.dot``) exactly as there.
"""
import math

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla

_TINY = 1e-25  # CLOSE_TO_ZERO, qp_subproblem.py:496


def eqp_kktfact(H, c, A, b):
    """Direct KKT solve of min 1/2 x'Hx + c'x s.t. Ax + b = 0.

    Reference: qp_subproblem.py:21-63 (test helper only).  Returns
    ``(x, multipliers)`` with the multiplier sign flipped as there (:61).
    """
    n, m = np.shape(c)[0], np.shape(b)[0]
    K = sps.csc_matrix(sps.bmat([[H, A.T], [A, None]]))
    sol = spla.splu(K).solve(np.hstack([-np.asarray(c), -np.asarray(b)]))
    return sol[:n], -sol[n:n + m]


def sphere_intersections(z, d, trust_radius, entire_line=False):
    """Interval of t with ||z + t d|| <= trust_radius.

    Reference: qp_subproblem.py:66-149.  Degenerate direction -> (0, 0,
    False) (:99-100); infinite radius shortcut (:102-110); cancellation-free
    quadratic roots through copysign (:112-131); segment clamp (:133-147).
    """
    if np.linalg.norm(d) == 0:
        return 0, 0, False
    if np.isinf(trust_radius):
        return (-np.inf, np.inf, True) if entire_line else (0, 1, True)

    qa = np.dot(d, d)
    qb = 2 * np.dot(z, d)
    qc = np.dot(z, z) - trust_radius ** 2
    disc = qb * qb - 4 * qa * qc
    if disc < 0:
        return 0, 0, False
    root = np.sqrt(disc)
    aux = qb + math.copysign(root, qb)
    ta, tb = sorted([-aux / (2 * qa), -2 * qc / aux])

    if entire_line:
        return ta, tb, True
    if tb < 0 or ta > 1:
        return 0, 0, False
    return max(0, ta), min(1, tb), True


def box_intersections(z, d, lb, ub, entire_line=False):
    """Interval of t with lb <= z + t d <= ub.

    Reference: qp_subproblem.py:152-234.  Coordinates with d == 0 only veto
    (:198-203); the rest give per-coordinate intervals whose intersection is
    max-of-min / min-of-max (:211-216).
    """
    z, d, lb, ub = (np.asarray(v) for v in (z, d, lb, ub))
    if np.linalg.norm(d) == 0:
        return 0, 0, False

    still = (d == 0)
    if (z[still] < lb[still]).any() or (z[still] > ub[still]).any():
        return 0, 0, False
    mv = ~still
    z, d, lb, ub = z[mv], d[mv], lb[mv], ub[mv]

    t_lo = (lb - z) / d
    t_hi = (ub - z) / d
    ta = np.max(np.minimum(t_lo, t_hi))
    tb = np.min(np.maximum(t_lo, t_hi))

    intersect = bool(ta <= tb)
    if not entire_line:
        if tb < 0 or ta > 1:
            return 0, 0, False
        ta, tb = max(0, ta), min(1, tb)
    return ta, tb, intersect


def box_sphere_intersections(z, d, lb, ub, trust_radius,
                             entire_line=False, extra_info=False):
    """Intersection of the box and ball intervals.

    Reference: qp_subproblem.py:237-303.
    """
    ta_b, tb_b, hit_b = box_intersections(z, d, lb, ub, entire_line)
    ta_s, tb_s, hit_s = sphere_intersections(z, d, trust_radius, entire_line)
    ta = np.maximum(ta_b, ta_s)
    tb = np.minimum(tb_b, tb_s)
    intersect = bool(hit_b and hit_s and ta <= tb)
    if extra_info:
        return (ta, tb, intersect,
                {'ta': ta_s, 'tb': tb_s, 'intersect': hit_s},
                {'ta': ta_b, 'tb': tb_b, 'intersect': hit_b})
    return ta, tb, intersect


def inside_box_boundaries(x, lb, ub):
    """qp_subproblem.py:306-308."""
    return bool((lb <= x).all() and (x <= ub).all())


def reinforce_box_boundaries(x, lb, ub):
    """qp_subproblem.py:310-317 (clip)."""
    return np.minimum(np.maximum(x, lb), ub)


def modified_dogleg(A, Y, b, trust_radius, lb, ub):
    """Normal step: approximately minimise ||A x + b|| in the box and ball.

    Reference: qp_subproblem.py:320-413.  Newton point accepted when interior
    (:368-373); otherwise Cauchy point (:376-380) and three segment searches
    (:386-407); ties go to the origin->Newton candidate (:410-413).
    """
    newton = -Y.dot(b)
    if inside_box_boundaries(newton, lb, ub) \
            and np.linalg.norm(newton) <= trust_radius:
        return newton

    g = A.T.dot(b)
    Ag = A.dot(g)
    cauchy = -np.dot(g, g) / np.dot(Ag, Ag) * g
    origin = np.zeros_like(cauchy)

    _, alpha, hit = box_sphere_intersections(cauchy, newton - cauchy,
                                             lb, ub, trust_radius)
    if hit:
        x1 = cauchy + alpha * (newton - cauchy)
    else:
        _, alpha, _ = box_sphere_intersections(origin, cauchy,
                                               lb, ub, trust_radius)
        x1 = origin + alpha * cauchy

    _, alpha, _ = box_sphere_intersections(origin, newton,
                                           lb, ub, trust_radius)
    x2 = origin + alpha * newton

    if np.linalg.norm(A.dot(x1) + b) < np.linalg.norm(A.dot(x2) + b):
        return x1
    return x2


def projected_cg(H, c, Z, Y, b, trust_radius=np.inf, lb=None, ub=None,
                 tol=None, max_iter=None, max_infeasible_iter=None,
                 return_all=False, trace=None):
    """Steihaug-Toint projected CG (Gould-Hribar-Nocedal Alg. 6.2).

    Reference: qp_subproblem.py:416-643.  Stop codes (:470-475): 1 iteration
    limit, 2 trust-region boundary, 3 negative curvature, 4 tolerance.
    ``trace`` (oracle-only extra) collects per-iteration scalars
    ``(rt_g, pt_H_p, alpha, beta)`` for the golden traces.
    """
    n, m = np.shape(c)[0], np.shape(b)[0]

    # :502-512 initial point, residual, direction
    x = Y.dot(-b)
    r = Z.dot(H.dot(x) + c)
    g = Z.dot(r)
    p = -g
    allvecs = [x] if return_all else None
    H_p = H.dot(p)
    rt_g = np.linalg.norm(g) ** 2

    # :515-526 feasibility of the trust region
    tr_distance = trust_radius - np.linalg.norm(x)
    if tr_distance < 0:
        raise ValueError("Trust region problem does not have a solution.")
    if tr_distance < _TINY:
        info = {'niter': 0, 'stop_cond': 2, 'hits_boundary': True}
        if return_all:
            allvecs.append(x)
            info['allvecs'] = allvecs
        return x, info

    # :529-542 defaults
    if tol is None:
        tol = max(min(0.01 * np.sqrt(rt_g), 0.1 * rt_g), _TINY)
    if lb is None:
        lb = np.full(n, -np.inf)
    if ub is None:
        ub = np.full(n, np.inf)
    if max_iter is None:
        max_iter = n - m
    max_iter = min(max_iter, n - m)
    if max_infeasible_iter is None:
        max_infeasible_iter = n - m

    hits_boundary = False
    stop_cond = 1
    counter = 0
    last_feasible_x = np.empty_like(x)  # deliberately uninitialised (:547)
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
            _, alpha, hit = box_sphere_intersections(
                x, p, lb, ub, trust_radius, entire_line=True)
            if hit:
                x = x + alpha * p
            x = reinforce_box_boundaries(x, lb, ub)
            stop_cond = 3
            hits_boundary = True
            break

        alpha = rt_g / pt_H_p                            # :579-580
        x_next = x + alpha * p

        if np.linalg.norm(x_next) >= trust_radius:       # :583-596
            _, theta, hit = box_sphere_intersections(
                x, alpha * p, lb, ub, trust_radius)
            if hit:
                x = x + theta * alpha * p
            x = reinforce_box_boundaries(x, lb, ub)
            stop_cond = 2
            hits_boundary = True
            break

        if inside_box_boundaries(x_next, lb, ub):        # :599-616
            counter = 0
        else:
            counter += 1
        if counter > 0:
            _, theta, hit = box_sphere_intersections(
                x, alpha * p, lb, ub, trust_radius)
            if hit:
                last_feasible_x = reinforce_box_boundaries(
                    x + theta * alpha * p, lb, ub)
                counter = 0
        if counter > max_infeasible_iter:
            break
        if return_all:
            allvecs.append(x_next)

        r_next = r + alpha * H_p                         # :622
        g_next = Z.dot(r_next)                           # :624
        rt_g_next = np.linalg.norm(g_next) ** 2          # :626
        beta = rt_g_next / rt_g
        p = -g_next + beta * p                           # :628
        if trace is not None:
            trace.append((rt_g, pt_H_p, alpha, beta))
        x = x_next
        g = g_next
        r = g_next                                       # sic, :632
        rt_g = np.linalg.norm(g) ** 2                    # :633
        H_p = H.dot(p)                                   # :634

    if not inside_box_boundaries(x, lb, ub):             # :636-638
        x = last_feasible_x
        hits_boundary = True
    info = {'niter': k, 'stop_cond': stop_cond,
            'hits_boundary': hits_boundary}
    if return_all:
        info['allvecs'] = allvecs
    return x, info
