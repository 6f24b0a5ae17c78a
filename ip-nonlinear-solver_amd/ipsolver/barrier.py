"""Trust-region interior-point outer loop (barrier subproblems) over backend
vectors.  Follows ``_large_scale_constrained/tr_interior_point.py`` of the
reference (cited per method): slack initialisation, fraction-to-the-boundary
bounds, barrier / tolerance decay, scaled augmented Jacobian and the
primal / primal-dual slack Hessian block.  z = [x; s] is ONE backend vector;
``s`` is a view of it, and the enforce-feasibility slack reset writes through
that view exactly like the reference does (tr_interior_point.py:62-63,92).
"""
import numpy as np

from .sqp import equality_constrained_sqp

__all__ = ['tr_interior_point', 'BarrierSubproblem']


class BarrierSubproblem:
    """min fun(x) - mu*sum(log s)  s.t. c_eq(x) = 0, c_ineq(x) + s = 0
    (tr_interior_point.py:24-251)."""

    def __init__(self, xp, x0, s0, fun, grad, lagr_hess, n_vars, n_ineq, n_eq, constr, jac,
                 barrier_parameter, tolerance, enforce_feasibility, global_stop_criteria,
                 xtol, fun0, grad0, constr_ineq0, jac_ineq0, constr_eq0, jac_eq0):
        self.xp = xp
        self.n_vars, self.n_ineq, self.n_eq = n_vars, n_ineq, n_eq
        self.fun, self.grad, self.lagr_hess = fun, grad, lagr_hess
        self.constr, self.jac = constr, jac
        self.barrier_parameter, self.tolerance = barrier_parameter, tolerance
        self.enforce_feasibility = np.asarray(enforce_feasibility, dtype=bool)
        self.any_enforced = bool(self.enforce_feasibility.any())
        self.global_stop_criteria = global_stop_criteria
        self.xtol = xtol
        self.fun0 = self._compute_function(fun0, constr_ineq0, s0)       # :53-56
        self.grad0 = self._compute_gradient(grad0)
        self.constr0 = self._compute_constr(constr_ineq0, constr_eq0, s0)
        self.jac0 = self._compute_jacobian(jac_eq0, jac_ineq0, s0)

    def update(self, barrier_parameter, tolerance):
        self.barrier_parameter, self.tolerance = barrier_parameter, tolerance

    def get_slack(self, z):
        return z[self.n_vars:self.n_vars + self.n_ineq]

    def get_variables(self, z):
        return z[:self.n_vars]

    # The method re-evaluates the problem at the point a subproblem ended on before it starts
    # the next one (tr_interior_point.py:338-340); that point is the vector OBJECT the last
    # accepted step evaluated, unchanged since.  What the user's callbacks returned for it is
    # kept (the last few evaluations, recognised by object identity + the tensor's version
    # counter; backends without one never hit) and only the terms that carry the barrier
    # parameter are formed again: 11 of config 3's 26 evaluations of fun / constr / grad / jac,
    # and -- the Jacobian being the same object -- of its factorizations (projector.projections).
    MEMO = 3

    @staticmethod
    def _memo_key(z):
        t = getattr(getattr(z, "loc", z), "t", None)          # DVec / ShardVec
        v = getattr(t, "_version", None)
        return None if v is None else (z, v)

    def _memo_get(self, name, z):
        key = self._memo_key(z)
        if key is not None:
            for (obj, ver), val in getattr(self, name, ()):
                if obj is key[0] and ver == key[1]:
                    return val
        return None

    def _memo_put(self, name, z, val):
        key = self._memo_key(z)
        if key is not None:
            setattr(self, name, ([(key, val)] + list(getattr(self, name, ())))[:self.MEMO])

    def function_and_constraints(self, z):                                # :68-86
        x, s = self.get_variables(z), self.get_slack(z)
        hit = self._memo_get("_memo_f", z)
        if hit is None:
            f = self.fun(x)
            c_ineq, c_eq = self.constr(x)
        else:
            f, c_ineq, c_eq = hit
        out = (self._compute_function(f, c_ineq, s),
               self._compute_constr(c_ineq, c_eq, s))
        if hit is None:
            self._memo_put("_memo_f", z, (f, c_ineq, c_eq))   # (after the slack reset, if any)
        return out

    def _compute_function(self, f, c_ineq, s):                            # :88-95
        if self.n_ineq == 0:
            return f          # (f - mu * 0.0: the same number; f may still be on the device)
        f = float(f)
        if self.any_enforced:
            # s[enforce] = -c_ineq[enforce], in place through the view of z
            self.xp.assign_negated_where(s, self.enforce_feasibility, c_ineq)
        return f - self.barrier_parameter * self.xp.sum_log(s)

    def _compute_constr(self, c_ineq, c_eq, s):                           # :97-100
        if self.n_ineq == 0:
            return c_eq
        return self.xp.hstack((c_eq, c_ineq + s))

    def scaling(self, z):                                                 # :102-115
        if self.n_ineq == 0:
            return None                    # identity
        xp = self.xp
        return xp.diagonal_operator(xp.hstack((xp.full(self.n_vars, 1.0, space="x"),
                                               self.get_slack(z))))

    def gradient_and_jacobian(self, z):                                   # :117-136
        x, s = self.get_variables(z), self.get_slack(z)
        hit = self._memo_get("_memo_g", z)
        if hit is not None:
            g, A = hit
            return self._compute_gradient(g), A
        g = self.grad(x)
        J_ineq, J_eq = self.jac(x)
        A = self._compute_jacobian(J_eq, J_ineq, s)
        self._memo_put("_memo_g", z, (g, A))
        return self._compute_gradient(g), A

    def _compute_gradient(self, g):                                       # :138-139
        if self.n_ineq == 0:
            return g
        return self.xp.hstack((g, self.xp.full(self.n_ineq, -self.barrier_parameter,
                                               space="ineq")))

    def _compute_jacobian(self, J_eq, J_ineq, s):                         # :141-194
        if self.n_ineq == 0:
            return self.xp.matrix(J_eq)
        return self.xp.augmented_jacobian(J_eq, J_ineq, s, self.n_vars, self.n_eq, self.n_ineq)

    def lagrangian_hessian(self, z, v):                                   # :196-241
        xp = self.xp
        x = self.get_variables(z)
        v_eq = v[:self.n_eq]
        v_ineq = v[self.n_eq:self.n_eq + self.n_ineq]
        Hx_terms = self.lagr_hess(x, v_eq, v_ineq)
        if self.n_ineq == 0:
            return xp.hessian_operator(Hx_terms, self.n_vars, None)
        s = self.get_slack(z)
        # S Hs S: primal-dual (v*s) where v_ineq > 0, primal (mu) elsewhere (:206-220)
        slack_block = xp.where_positive(v_ineq, v_ineq * s, self.barrier_parameter)
        return xp.hessian_operator(Hx_terms, self.n_vars, slack_block)

    def stop_criteria(self, state):                                       # :243-251
        return ((state.optimality < self.tolerance
                 and state.constr_violation < self.tolerance)
                or self.global_stop_criteria(state)
                or state.trust_radius < self.xtol)


def tr_interior_point(fun, grad, lagr_hess, n_vars, n_ineq, n_eq, constr, jac, x0, fun0, grad0,
                      constr_ineq0, jac_ineq0, constr_eq0, jac_eq0, stop_criteria,
                      enforce_feasibility, xtol, state, xp,
                      initial_barrier_parameter=0.1, initial_tolerance=0.1,
                      initial_penalty=1.0, initial_trust_radius=1.0, return_all=False,
                      factorization_method=None):
    """Reference tr_interior_point.py:254-355."""
    BOUNDARY_PARAMETER = 0.995
    BARRIER_DECAY_RATIO = 0.2
    TRUST_ENLARGEMENT = 5

    if enforce_feasibility is None:
        enforce_feasibility = np.zeros(n_ineq, bool)
    state.barrier_parameter = initial_barrier_parameter       # :287-292
    state.tolerance = initial_tolerance
    state.trust_radius = initial_trust_radius
    state.penalty = initial_penalty
    state.optimality = np.inf
    state.constr_violation = np.inf
    s0 = xp.maximum(-1.5 * constr_ineq0, 1.0) if n_ineq > 0 else xp.zeros(0)   # :294
    subprob = BarrierSubproblem(
        xp, x0, s0, fun, grad, lagr_hess, n_vars, n_ineq, n_eq, constr, jac,
        state.barrier_parameter, state.tolerance, enforce_feasibility, stop_criteria, xtol,
        fun0, grad0, constr_ineq0, jac_ineq0, constr_eq0, jac_eq0)
    z = xp.hstack((x0, s0)) if n_ineq > 0 else xp.copy(x0)   # :302
    fun0_sub, constr0_sub = subprob.fun0, subprob.constr0
    grad0_sub, jac0_sub = subprob.grad0, subprob.jac0
    if n_ineq > 0:                                            # :306-308
        trust_lb = xp.hstack((xp.full(n_vars, -np.inf, space="x"),
                              xp.full(n_ineq, -BOUNDARY_PARAMETER, space="ineq")))
        trust_ub = None                    # no upper bounds (sqp keeps that side off the kernels)
    else:
        trust_lb = trust_ub = None         # all-infinite box: skipped by the kernels

    first = True
    while True:                                               # :313-340
        if not first:
            state.trust_radius = max(initial_trust_radius, TRUST_ENLARGEMENT * state.trust_radius)
            state.barrier_parameter *= BARRIER_DECAY_RATIO
            state.tolerance *= BARRIER_DECAY_RATIO
        first = False
        subprob.update(state.barrier_parameter, state.tolerance)
        state = equality_constrained_sqp(
            subprob.function_and_constraints, subprob.gradient_and_jacobian,
            subprob.lagrangian_hessian, z, fun0_sub, grad0_sub, constr0_sub, jac0_sub,
            subprob.stop_criteria, state, xp, trust_lb, trust_ub, initial_penalty,
            state.trust_radius, subprob.scaling, return_all, factorization_method)
        z = state.x
        if stop_criteria(state):
            break
        fun0_sub, constr0_sub = subprob.function_and_constraints(z)
        grad0_sub, jac0_sub = subprob.gradient_and_jacobian(z)

    state.x = subprob.get_variables(z)                        # :343-353
    state.s = subprob.get_slack(z)
    if return_all:
        zs = state.allvecs
        state.allvecs = [subprob.get_variables(t) for t in zs]
        state.allslack = [subprob.get_slack(t) for t in zs]
    return state
