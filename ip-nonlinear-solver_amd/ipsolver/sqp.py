"""Byrd-Omojokun trust-region SQP outer loop over backend vectors.

The algorithm, its constants, counters and accept / reject rules are those of the
reference's ``_large_scale_constrained/equality_constrained_sqp.py:18-252`` (cited below);
what is organised differently is WHERE the scalars come from.  Every vector lives where
the backend ``xp`` keeps it (HBM for ``backend_hip``, partitioned over the ranks for the
sharded backend) and a norm or a dot product read back on its own is a blocking device-to-
host copy (plus, sharded, a collective).  The reference takes ~18 of them per outer
iteration one by one (``norm(dn)`` three times); here an iteration is cut at its DECISION
POINTS and the scalars of each are enqueued into one pack (``xp.pack()``) and read together:

    A  after the normal step          ||dn||                        -> the tangential radius
    B  after the tangential step      (Hd).d, c.d, ||A d + b||,
                                      ||d||, ||dt||                 -> model, penalty, prediction
    C  after the trial evaluation     ||b_next||                    -> accept / reject
    D  after an accepted step         ||c + A'v||_inf, ||b||_inf,
                                      ||b||                         -> state, next iteration

(A is free when the normal step is the Newton point: ``modified_dogleg`` had to measure it.)
The values are the ones the unpacked calls would return, so the iterate sequence is that of
the reference.  The trust-region subproblem itself (normal step, tangential step,
projections) is delegated to ``xp``, i.e. to the HIP kernels.
"""
import numpy as np

__all__ = ['equality_constrained_sqp']


def equality_constrained_sqp(fun_and_constr, grad_and_jac, lagr_hess, x0, fun0, grad0,
                             constr0, jac0, stop_criteria, state, xp,
                             trust_lb=None, trust_ub=None, initial_penalty=1.0,
                             initial_trust_radius=1.0, scaling=None, return_all=False,
                             factorization_method=None):
    PENALTY_FACTOR = 0.3               # :50-60
    LARGE_REDUCTION_RATIO = 0.9
    INTERMEDIARY_REDUCTION_RATIO = 0.3
    SUFFICIENT_REDUCTION_RATIO = 1e-8
    TRUST_ENLARGEMENT_FACTOR_L = 7.0
    TRUST_ENLARGEMENT_FACTOR_S = 2.0
    MAX_TRUST_REDUCTION = 0.5
    MIN_TRUST_REDUCTION = 0.1
    SOC_THRESHOLD = 0.1
    TR_FACTOR = 0.8
    BOX_FACTOR = 0.5

    n = len(x0)
    # No box at all (pure equality SQP): keep None so the kernels skip the
    # bound passes; the reference materialises +-inf vectors (:65-68).
    boxed = trust_lb is not None or trust_ub is not None
    # one-sided boxes (the barrier problem: only lower bounds, on the slacks) keep the missing
    # side None towards projected_cg, whose kernels then skip that bound vector
    lb_free, ub_free = trust_lb is None, trust_ub is None
    if boxed:
        trust_lb = trust_lb if trust_lb is not None else xp.full(n, -np.inf, space="z")
        trust_ub = trust_ub if trust_ub is not None else xp.full(n, np.inf, space="z")
        half_lb, half_ub = BOX_FACTOR * trust_lb, BOX_FACTOR * trust_ub
    else:
        half_lb = half_ub = None

    def measure(c, A, v, b):
        """Decision point D: optimality, constraint violation (:86-87,238-239) and ||b|| for
        the next iterations' merit function, one read."""
        pk = xp.pack()
        h_opt = pk.norm_inf(c + A.T.dot(v))
        h_viol, h_nb = pk.norm_inf(b), pk.norm(b)
        vals = pk.read()
        return vals[h_opt], vals[h_viol], vals[h_nb]

    x = xp.copy(x0)                                           # :71-83
    trust_radius = initial_trust_radius
    penalty = initial_penalty
    f, c, b, A = fun0, grad0, constr0, jac0
    S = scaling(x) if scaling is not None else None
    Z, LS, Y = xp.projections(A, factorization_method)
    v = -LS.dot(c)

    state.optimality, state.constr_violation, norm_b = measure(c, A, v, b)   # :86-99
    state.niter += 1
    state.x, state.v, state.fun, state.grad = x, v, f, c
    state.constr, state.jac = b, A
    state.trust_radius, state.penalty = trust_radius, penalty
    if return_all:
        state.allvecs += [xp.copy(x)]
        state.allmult += [xp.copy(v)]

    compute_hess = True
    while not stop_criteria(state):                           # :102
        if compute_hess:
            H = lagr_hess(x, v)
            state.nhev += 1

        # ---- normal step (:113-116); decision point A: its norm
        known = []
        dn = xp.modified_dogleg(A, Y, b, TR_FACTOR * trust_radius, half_lb, half_ub, known)
        if known:
            norm_dn = known[0]
        else:
            pk = xp.pack()
            h = pk.norm(dn)
            norm_dn = pk.read()[h]

        # ---- tangential step (:125-132): b_t = 0 (None: projected_cg then starts from the
        # zero vector without solving for it)
        c_t = H.dot(dn) + c
        trust_radius_t = np.sqrt(trust_radius ** 2 - norm_dn ** 2)
        lb_t = trust_lb - dn if boxed and not lb_free else None
        ub_t = trust_ub - dn if boxed and not ub_free else None
        dt, info_cg = xp.projected_cg(H, c_t, Z, Y, None, trust_radius_t, lb_t, ub_t)

        # ---- decision point B: quadratic model, linearised constraints, step lengths
        d = dn + dt                                           # :135-153
        pk = xp.pack()
        h_hd, h_cd = pk.dot(H.dot(d), d), pk.dot(c, d)
        h_lin, h_d, h_dt = pk.norm(A.dot(d) + b), pk.norm(d), pk.norm(dt)
        vals = pk.read()
        quadratic_model = 1 / 2 * vals[h_hd] + vals[h_cd]
        norm_d, norm_dt = vals[h_d], vals[h_dt]
        vpred = max(1e-16, norm_b - vals[h_lin])
        previous_penalty = penalty
        if quadratic_model > 0:
            penalty = max(penalty, quadratic_model / ((1 - PENALTY_FACTOR) * vpred))
        predicted_reduction = -quadratic_model + penalty * vpred

        # ---- trial point; decision point C: its constraint norm
        merit_function = f + penalty * norm_b                 # :156-169
        x_next = x + (S.dot(d) if S is not None else d)
        f_next, b_next = fun_and_constr(x_next)
        state.nfev += 1
        state.ncev += 1
        norm_b_next = xp.norm(b_next)
        actual_reduction = merit_function - (f_next + penalty * norm_b_next)
        reduction_ratio = actual_reduction / predicted_reduction

        if reduction_ratio < SUFFICIENT_REDUCTION_RATIO and \
                norm_dn <= SOC_THRESHOLD * norm_dt:           # :172-193 second-order correction
            y = -Y.dot(b_next)
            if boxed:
                _, t, intersect = xp.box_intersections(d, y, trust_lb, trust_ub)
            else:
                # an unbounded box never clips the segment (ta, tb = 0, 1) unless y == 0
                intersect = xp.norm(y) != 0
                t = 1 if intersect else 0
            step = d + t * y
            x_soc = x + (S.dot(step) if S is not None else step)
            f_soc, b_soc = fun_and_constr(x_soc)
            state.nfev += 1
            state.ncev += 1
            norm_b_soc = xp.norm(b_soc)
            ratio_soc = (merit_function - (f_soc + penalty * norm_b_soc)) / predicted_reduction
            if intersect and ratio_soc >= SUFFICIENT_REDUCTION_RATIO:
                x_next, f_next, b_next, norm_b_next = x_soc, f_soc, b_soc, norm_b_soc
                reduction_ratio = ratio_soc

        if reduction_ratio >= LARGE_REDUCTION_RATIO:          # :196-212
            trust_radius = max(TRUST_ENLARGEMENT_FACTOR_L * norm_d, trust_radius)
        elif reduction_ratio >= INTERMEDIARY_REDUCTION_RATIO:
            trust_radius = max(TRUST_ENLARGEMENT_FACTOR_S * norm_d, trust_radius)
        elif reduction_ratio < SUFFICIENT_REDUCTION_RATIO:
            trust_reduction = (1 - SUFFICIENT_REDUCTION_RATIO) / (1 - reduction_ratio)
            new_trust_radius = trust_reduction * norm_d
            if new_trust_radius >= MAX_TRUST_REDUCTION * trust_radius:
                trust_radius *= MAX_TRUST_REDUCTION
            elif new_trust_radius >= MIN_TRUST_REDUCTION * trust_radius:
                trust_radius = new_trust_radius
            else:
                trust_radius *= MIN_TRUST_REDUCTION

        state.niter += 1                                      # :215-242
        if reduction_ratio >= SUFFICIENT_REDUCTION_RATIO:
            x = x_next
            f, b = f_next, b_next
            c, A = grad_and_jac(x)
            S = scaling(x) if scaling is not None else None
            state.ngev += 1
            state.njev += 1
            Z, LS, Y = xp.projections(A, None)                # method only honoured at entry (:225)
            v = -LS.dot(c)
            compute_hess = True
            state.x, state.v, state.fun, state.grad = x, v, f, c
            state.constr, state.jac = b, A
            state.optimality, state.constr_violation, norm_b = measure(c, A, v, b)
        else:
            penalty = previous_penalty
            compute_hess = False
        state.trust_radius = trust_radius                     # :244-250
        state.penalty = penalty
        state.cg_niter += info_cg["niter"]
        state.cg_info = info_cg
        if return_all:
            state.allvecs.append(xp.copy(x))
            state.allmult.append(xp.copy(v))
    return state
