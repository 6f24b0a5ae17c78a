"""Byrd-Omojokun trust-region SQP outer loop (reference
``_large_scale_constrained/equality_constrained_sqp.py:18-252``) as a sequence of STAGES.

An outer iteration is cut where the host must act -- the user's callbacks -- and nowhere else:

    settle    a new iterate's factorization, multipliers ``v = -LS c`` and its two measures
              (optimality, constraint violation) + ``||b||``               (:81-87, 225-239)
    propose   normal step, tangential step, the quadratic model / penalty / predicted
              reduction and the trial point ``x + S d``                      (:113-153, 158)
    judge     after ``fun_and_constr(x_next)``: actual / predicted reduction, is the
              second-order correction due, trust-radius ladder, accept / reject
                                                                             (:159-169, 196-242)

Every stage works on ONE block of scalars (layout: ``sqp_chain``; csrc/sqp.hip ``SQ_*``) and the
decisions' arithmetic is the library's (``ipx_sqp_model_host`` / ``_ratio_host`` /
``_radius_host``: the code the decide kernels run, compiled for the host) -- this module holds
no formula of the method, only the order of the stages and the bookkeeping of the reference's
``state``.  Two providers of the stages:

``ChainStages``  (HIP backend, CSR Jacobian + CSR Hessian): each stage is one C entry point
                 whose launches end in a decide kernel; one blocking read of the block per stage
                 (``sqp_chain.py``, csrc/sqp.hip).
``HostStages``   (every other backend and operator type -- dense Jacobians, operator Hessians,
                 the row-sharded solver, the CPU oracle's backend in the tests): the stage's
                 vectors through the backend ``xp``, its scalars enqueued into one pack and read
                 together, the same decision functions on a host block.

The second-order correction (:172-193), rare, is host-driven for both.
"""
import time

import numpy as np

from . import sqp_chain as sc
from .sqp_chain import (RADIUS, PENALTY, F, NORM_B, NORM_DN, RADIUS_T, HDD, CD, LIN, NORM_D,
                        NORM_DT, PRED, MERIT, F_NEXT, NORM_B_NEXT, RATIO, SOC, ACCEPT,
                        TR_FACTOR, BOX_FACTOR)

__all__ = ['equality_constrained_sqp']

_SUFFICIENT = 1e-8         # SUFFICIENT_REDUCTION_RATIO (:53): the correction's own acceptance test

# host seconds spent finishing projected-CG calls whose first batch (enqueued by the front chain)
# did not end them -- with the GPU time of the chains' own CG part (ipx_sqp_cg_timing) the time
# "inside projected_cg" of a solve (bench.py)
TIMERS = {"host_cg_seconds": 0.0}


class _Point:
    """An iterate with everything the stages derive for it."""
    __slots__ = ("x", "f", "c", "b", "A", "S", "Z", "LS", "Y", "v", "opt", "viol", "norm_b")


class _Trial:
    __slots__ = ("x_next", "d", "cg_info", "on_chain", "finish", "q")

    def __init__(self):
        self.finish = self.q = None


class _Box:
    """The step's bounds.  ``lb`` / ``ub`` as given (None = that side is free: the kernels then
    skip the bound vector); ``lb_full`` / ``ub_full`` with the free side materialised for the
    routines that want both (the reference materialises both, :65-68)."""

    def __init__(self, xp, n, trust_lb, trust_ub):
        self.lb, self.ub = trust_lb, trust_ub
        self.any = trust_lb is not None or trust_ub is not None
        if self.any:
            self.lb_full = trust_lb if trust_lb is not None else xp.full(n, -np.inf, space="z")
            self.ub_full = trust_ub if trust_ub is not None else xp.full(n, np.inf, space="z")
            self.half_lb, self.half_ub = BOX_FACTOR * self.lb_full, BOX_FACTOR * self.ub_full
        else:
            self.lb_full = self.ub_full = self.half_lb = self.half_ub = None


class HostStages:
    """The stages through the backend's vector operations; scalars of a stage read together."""

    def __init__(self, xp, box):
        self.xp, self.box = xp, box

    # -- settle
    def settle(self, pt, method):
        self.settle_with(pt, self.xp.projections(pt.A, method))

    def settle_with(self, pt, ops):
        xp = self.xp
        pt.Z, pt.LS, pt.Y = ops
        pt.v = -pt.LS.dot(pt.c)
        pk = xp.pack()
        h_opt = pk.norm_inf(pt.c + pt.A.T.dot(pt.v))
        h_viol, h_nb = pk.norm_inf(pt.b), pk.norm(pt.b)
        vals = pk.read()
        pt.opt, pt.viol, pt.norm_b = vals[h_opt], vals[h_viol], vals[h_nb]

    # -- propose
    def propose(self, pt, H, radius, penalty):
        xp, box = self.xp, self.box
        q = sc.new_block()
        q[RADIUS], q[PENALTY], q[F], q[NORM_B] = radius, penalty, float(pt.f), pt.norm_b
        known = []
        dn = xp.modified_dogleg(pt.A, pt.Y, pt.b, TR_FACTOR * radius, box.half_lb, box.half_ub,
                                known)
        if known:                      # (the accepted Newton point: the dogleg measured it)
            q[NORM_DN] = known[0]
        else:
            pk = xp.pack()
            h = pk.norm(dn)
            q[NORM_DN] = pk.read()[h]
        q[RADIUS_T] = np.sqrt(radius ** 2 - q[NORM_DN] ** 2)
        # b_t = 0 (None: projected_cg then starts from the zero vector without solving for it)
        c_t = H.dot(dn) + pt.c
        lb_t = box.lb - dn if box.lb is not None else None
        ub_t = box.ub - dn if box.ub is not None else None
        dt, info = xp.projected_cg(H, c_t, pt.Z, pt.Y, None, q[RADIUS_T], lb_t, ub_t)
        if hasattr(xp, "note_cg_length"):
            xp.note_cg_length(info['niter'])
        d = dn + dt
        pk = xp.pack()
        h_hd, h_cd = pk.dot(H.dot(d), d), pk.dot(pt.c, d)
        h_lin, h_d, h_dt = pk.norm(pt.A.dot(d) + pt.b), pk.norm(d), pk.norm(dt)
        vals = pk.read()
        q[HDD], q[CD], q[LIN] = vals[h_hd], vals[h_cd], vals[h_lin]
        q[NORM_D], q[NORM_DT] = vals[h_d], vals[h_dt]
        sc.model_host(q)
        t = _Trial()
        t.d, t.cg_info, t.on_chain = d, info, False
        t.x_next = pt.x + (pt.S.dot(d) if pt.S is not None else d)
        return q, t

    # -- judge
    def judge(self, q, trial, f_next, b_next):
        q[F_NEXT], q[NORM_B_NEXT] = float(f_next), self.xp.norm(b_next)
        sc.ratio_host(q)
        if not q[SOC]:
            sc.radius_host(q)
        return q


# ChainStages.propose returns with its chain still running and the caller evaluates the callbacks
# at the trial point behind it (False: the block is waited for first -- A/B measurements, and
# the order in which the reference shows its callbacks the points, exactly)
EVALUATE_BEHIND_THE_CHAIN = True
# ... in device-callback mode only (minimize._minimize_device sets this for its call): a host
# callback starts with a copy of the trial point to the host, which waits for the chain anyway --
# nothing to hide, and a step the host finishes would cost a second evaluation for nothing
_CALLBACKS_ON_DEVICE = False


class ChainStages:
    """The stages as device chains (sqp_chain.StepChain); a stage whose operands do not qualify
    goes to the host-driven twin for that call."""

    # CG iterations enqueued behind the priming: as many as the longer of the last two calls on
    # this problem took, within these bounds -- an iteration enqueued behind a stopped loop is
    # three launches that do nothing, ~17 us; a loop that is not done when the block is read
    # costs a second read.  (The launch that would SEE the tolerance met is not enqueued: the
    # host applies that test -- qp_subproblem.py:551, the first thing an iteration does -- to
    # the state block itself, _finish.)
    BATCH_MIN, BATCH_MAX = 2, 16

    def __init__(self, xp, box, host, n, m):
        self.xp, self.box, self.host = xp, box, host
        self.n, self.m = n, m
        self.chain = sc.StepChain.get(n, m, box.lb is not None, box.ub is not None)

    # -- settle
    def settle(self, pt, method):
        import torch
        from . import projector
        from .device import DVec, DeviceCSR
        chain = self.chain
        if not isinstance(pt.A, DeviceCSR) or pt.A.shape != (self.m, self.n):
            return self.host.settle(pt, method)
        # the same matrix, gradient and constraint value as the last time this matrix was
        # settled (objects and version counters): the same multipliers and measures -- the
        # barrier method hands the point a subproblem ended on to the next one unchanged when
        # the barrier terms do not touch them (tr_interior_point.py:338-340)
        memo_key = _settle_key(pt, method)
        memo = getattr(pt.A, "_ipx_settled", None)
        if memo_key is not None and memo is not None and _same_key(memo[0], memo_key):
            pt.Z, pt.LS, pt.Y, pt.v, pt.opt, pt.viol, pt.norm_b = memo[1]
            sc.STATS["settles_reused"] += 1
            return
        ops = projector.projections(pt.A, method, deferred=chain)
        P = getattr(ops[0], "projector", None)
        if not sc.StepChain.fits(ops[0].projector.A if P is not None else None, ops[0], ops[2],
                                 None, pt.S, (pt.x, pt.c, pt.b)) or P.A is not pt.A:
            projector.confirm(pt.A, ops)
            return self.host.settle_with(pt, projector.projections(pt.A, method))
        pending = bool(getattr(P.solver, "pending", False))
        sc.STATS["deferred_factorizations"] += 1 if pending else 0
        v_out = torch.empty(self.m, dtype=torch.float64, device=pt.c.t.device)
        chain.bind_refresh(P, pt.c, pt.b, v_out)
        chain.refresh(P.norm_partials() if P._norm_A is None else None, pending)
        P.stats["solves"] += 1
        q = chain.read()
        if pending:
            if q[sc.FACTOR_BAD] != 0:
                # the factorization did not end like the one before it (another reduction level,
                # a pivot finding): what was enqueued under the assumed verdict is void
                sc.STATS["verdict_misses"] += 1
                projector.invalidate(pt.A)
                return self.host.settle(pt, method)
            P.solver.pending = False
        if P._norm_A is None:
            P._norm_A = float(np.sqrt(q[sc.NORM_A2]))
        pt.Z, pt.LS, pt.Y = ops
        pt.v = DVec(v_out)
        pt.opt, pt.viol, pt.norm_b = q[sc.OPT], q[sc.VIOL], q[NORM_B]
        if memo_key is not None:
            try:
                pt.A._ipx_settled = (memo_key, (pt.Z, pt.LS, pt.Y, pt.v, pt.opt, pt.viol,
                                                pt.norm_b))
            except AttributeError:
                pass

    # -- propose
    def propose(self, pt, H, radius, penalty):
        import torch
        from . import _hip, cg_fused
        from .device import DVec, stream_ptr
        chain, box, n, m = self.chain, self.box, self.n, self.m
        P = getattr(pt.Z, "projector", None)
        if not sc.StepChain.fits(pt.A, pt.Z, pt.Y, H, pt.S, (pt.x, pt.c, pt.b, box.lb, box.ub)):
            sc.STATS["host_iterations"] += 1
            return self.host.propose(pt, H, radius, penalty)
        lib = _hip.load()
        L, key = cg_fused._loop_for(H, P, chain.lbt_vec, chain.ubt_vec)
        a = L.args
        if L.operator is not None or a.solver_kind not in (0, 1) or not a.banded or \
                lib.ipx_cg_prime_ws_doubles(L.ref(), P.A.pattern.ntiles) > 65536:
            cg_fused._release(L, key)
            sc.STATS["host_iterations"] += 1
            return self.host.propose(pt, H, radius, penalty)
        a.no_radius = 0
        x_next = torch.empty(n, dtype=torch.float64, device=pt.x.t.device)
        scale = pt.S.d if pt.S is not None else None
        chain.bind(L, P, pt.x, pt.c, pt.b, box.lb, box.ub, scale, x_next)
        max_iter = n - m
        first_end = min(max_iter, max(self.BATCH_MIN, min(max(chain.last_niter),
                                                           self.BATCH_MAX)))
        # (the dogleg proper rides along when the last normal step needed it: nine launches that
        # do nothing otherwise)
        t = _Trial()
        t.x_next, t.d, t.on_chain = DVec(x_next), DVec(chain.d), True
        finish = lambda q: self._finish(q, t, pt, H, radius, penalty, L, key, P, scale, first_end,
                                        max_iter)
        if not (EVALUATE_BEHIND_THE_CHAIN and _CALLBACKS_ON_DEVICE):
            chain.front(0, chain.expect_dogleg, radius, penalty, pt.f, pt.norm_b, P.norm_A,
                        first_end)
            q, _ = finish(chain.read())
            return q, t
        # The chain is enqueued and NOT waited for, its block stays on the device: the caller
        # evaluates the user's objective and constraints at the trial point right behind it,
        # ``judge`` enqueues the verdict behind those -- the host's call overhead for all of it
        # runs while the chain's ~300 us execute -- and reads ONE block that carries the
        # chain's entries with the verdict's.  In the usual case the block says the step stood
        # and the verdict is the one the method needs; a step the host has to finish (the dogleg
        # proper, a longer CG, a box event, a refinement) lands in a NEW trial vector and
        # evaluation and verdict are repeated there -- the provisional evaluation is not
        # counted; the callbacks saw a point the reference would not have shown them.
        chain.front(0, chain.expect_dogleg, radius, penalty, pt.f, pt.norm_b, P.norm_A, first_end,
                    quiet=True)
        t.finish = finish
        return None, t

    def _finish(self, q, t, pt, H, radius, penalty, L, key, P, scale, first_end, max_iter):
        """The rest of ``propose`` once a block with the chain's entries (``q``) is in:
        (block, redo) -- redo: the step was finished by the host into a new ``t.x_next``;
        callbacks and verdict are due there."""
        import torch
        from . import _hip, cg_fused
        from .device import DVec, stream_ptr
        chain, box, n, m = self.chain, self.box, self.n, self.m
        lib = _hip.load()
        redo = [False]

        def fresh():
            # (the provisional evaluation is remembered by OBJECT, barrier.py's memo and the
            # callbacks' own: a step written again gets a tensor nobody has seen)
            x_new = torch.empty(n, dtype=torch.float64, device=pt.x.t.device)
            chain.bind(L, P, pt.x, pt.c, pt.b, box.lb, box.ub, scale, x_new)
            t.x_next = DVec(x_new)
            redo[0] = True
            return x_new
        x_next = t.x_next.t
        chain.expect_dogleg = q[sc.NORMAL_KIND] != 1
        if q[sc.NORMAL_KIND] == 0:
            # the Newton point leaves the box or the 0.8-radius ball: the dogleg proper, then the
            # rest of the chain on its step
            sc.STATS["host_doglegs"] += 1
            dn = self.xp.modified_dogleg(pt.A, pt.Y, pt.b, TR_FACTOR * radius, box.half_lb,
                                         box.half_ub)
            chain.dn.copy_(dn.t)
            x_next = fresh()
            chain.front(1, 0, radius, penalty, pt.f, pt.norm_b, P.norm_A, first_end)
            q = chain.read()
        P.stats["solves"] += 3
        st = q[sc.CG:sc.CG + 16]
        stop = int(st[cg_fused.ST_STOP])
        if stop == 9 and not chain.expect_steps:
            # a projection of the priming needs its correction step and none was armed: once
            # more on the device with the steps (the normal step stands: chain.dn)
            sc.STATS["prime_rearmed"] = sc.STATS.get("prime_rearmed", 0) + 1
            chain.expect_steps = True
            x_next = fresh()
            chain.front(1, 0, radius, penalty, pt.f, pt.norm_b, P.norm_A, first_end)
            q = chain.read()
            st = q[sc.CG:sc.CG + 16]
            stop = int(st[cg_fused.ST_STOP])
        steps_taken = st[cg_fused.ST_PRIME_STEPS]
        L.enqueued = (0, first_end) if first_end > 0 else None
        if stop == 0 and int(st[cg_fused.ST_NITER]) == first_end and \
                st[cg_fused.ST_RTG0 + (first_end & 1)] < st[cg_fused.ST_TOL]:
            # every enqueued iteration ran and the next one's first test (:551, rt_g < tol on
            # the state's own numbers: csrc/cg.hip k_cg_step1) ends the loop before it touches
            # anything -- the iterate on the device is the call's result
            stop = 4
        on_device = stop == 4 or (stop in (2, 3) and q[sc.EXIT_DONE] != 0) \
            or (stop == 0 and first_end >= max_iter)
        outside = box.any and q[sc.X_OUTSIDE] > 0
        if on_device and not outside:
            info = {'niter': int(st[cg_fused.ST_NITER]), 'stop_cond': {4: 4, 2: 2, 3: 3, 0: 1}[stop],
                    'hits_boundary': stop in (2, 3)}
            cg_fused.STATS["calls"] += 1
            cg_fused.STATS["iterations"] += info['niter']
            L.enqueued = None
        else:
            # the loop needs the host: more iterations, a box event, a refinement, a priming the
            # device turned down -- finish it with the general driver, then the model again
            sc.STATS["host_cg"] += 1
            t_host = time.perf_counter()
            why = "host_cg_stop_%d" % stop if not on_device else "host_cg_outside_box"
            sc.STATS[why] = sc.STATS.get(why, 0) + 1
            c_t, radius_t = DVec(chain.ct), q[RADIUS_T]
            lb_t, ub_t = chain.lbt_vec, chain.ubt_vec
            if on_device:                  # (:636-638 with no feasible iterate recorded)
                dt, info = DVec.zeros(n), {'niter': int(st[cg_fused.ST_NITER]),
                                           'stop_cond': {4: 4, 2: 2, 3: 3, 0: 1}[stop],
                                           'hits_boundary': True}
                L.enqueued = None
            elif stop == 9:
                sc.STATS["prime_retries"] += 1
                chain.expect_steps = True      # (the next primings carry their correction steps)
                L.enqueued = None
                cg_fused._release(L, key)
                # (the host's priming straight away: the device just turned this one down)
                dt, info = cg_fused._projected_cg(H, c_t, pt.Z, pt.Y, DVec.zeros(m), radius_t,
                                                  lb_t, ub_t, None, None, None, None, None, True,
                                                  fast=False)
                L, key = cg_fused._loop_for(H, P, lb_t, ub_t)
            elif stop == 8:
                # a resident launch timed out: the loop's vectors are void, the call's inputs
                # are not -- the tangential step again from its priming, separate launches
                L.enqueued = None
                cg_fused.STATS["resident_fallbacks"] += 1
                if key is not None:
                    cg_fused._NO_RESIDENT.add(key)
                dt, info = self.xp.projected_cg(H, c_t, pt.Z, pt.Y, None, radius_t, lb_t, ub_t)
                L, key = cg_fused._loop_for(H, P, lb_t, ub_t)
            else:
                lbf = lb_t if lb_t is not None or not box.any else DVec.full(n, -np.inf)
                ubf = ub_t if ub_t is not None or not box.any else DVec.full(n, np.inf)
                dt, info = cg_fused._run_loop(L, key, P, lib, stream_ptr(), n, lbf, ubf, radius_t,
                                              max_iter, max_iter, None, None, fast=False,
                                              primed_state=st, release=False,
                                              first_batch=first_end)
            TIMERS["host_cg_seconds"] += time.perf_counter() - t_host
            L.args.x = dt.t.data_ptr()
            x_next = fresh()
            # (a verdict enqueued behind the chain before this block was looked at has moved the
            # block's trust radius along its ladder: the one this step was computed for)
            chain.q[RADIUS:RADIUS + 1].fill_(radius)
            chain.model(penalty, pt.f, pt.norm_b, host_cg=True)
            q = chain.read()
            chain.keep = chain.keep + (dt,)
        cg_fused._release(L, key)
        # the next priming carries its correction steps when this one took some, or came within
        # a factor 64 of needing the cancellation step (the projected gradient shrinks against
        # the gradient from one outer iteration to the next: the margin announces the first call
        # that needs one; a call that needs one and has none ends in stop code 9 -- host)
        if stop != 9:
            chain.expect_steps = steps_taken > 0 or \
                st[cg_fused.ST_MARGIN] < 64.0 * P.CANCELLATION ** 2
        chain.last_niter = (chain.last_niter[1], info['niter'])
        self.xp.note_cg_length(info['niter'])
        t.cg_info = info
        return q, redo[0]

    # -- judge
    def judge(self, q, trial, f_next, b_next):
        """The verdict's block -- or None: the proposing chain, examined only now, had left its
        step to the host; ``trial`` holds the finished step (``trial.q`` its block): evaluate
        the callbacks at the new ``trial.x_next`` and call again."""
        if not trial.on_chain:
            return self.host.judge(q, trial, f_next, b_next)
        # (an objective value still on the device goes to the verdict's kernel as it is and
        # comes back in the block: no read of its own)
        lazy = hasattr(f_next, "known") and not f_next.is_known
        self.chain.judge(b_next, f_next.t if lazy else float(f_next))
        q = self.chain.read()
        finish, trial.finish = trial.finish, None
        if finish is not None:
            front_q, redo = finish(q)
            if redo:
                trial.q = front_q
                return None
        if lazy:
            f_next.known(q[F_NEXT])
        return q


def _settle_key(pt, method):
    """What ``settle`` depends on, as (object, version counter) pairs; None: unknown types."""
    tens = (getattr(pt.A, "val", None), getattr(pt.c, "t", None), getattr(pt.b, "t", None))
    if any(t is None or not hasattr(t, "_version") for t in tens):
        return None
    return (method,) + tuple((t, t._version) for t in tens)


def _same_key(a, b):
    return a[0] == b[0] and all(x[0] is y[0] and x[1] == y[1] for x, y in zip(a[1:], b[1:]))


def _stages(xp, n, m, x0, box):
    host = HostStages(xp, box)
    if getattr(xp, "name", None) == "hip" and m > 0 and hasattr(x0, "t") \
            and not _chain_disabled():
        return ChainStages(xp, box, host, n, m)
    return host


def _chain_disabled():
    from . import _hip
    return _hip.debug_form("no-step-chain")


def equality_constrained_sqp(fun_and_constr, grad_and_jac, lagr_hess, x0, fun0, grad0,
                             constr0, jac0, stop_criteria, state, xp,
                             trust_lb=None, trust_ub=None, initial_penalty=1.0,
                             initial_trust_radius=1.0, scaling=None, return_all=False,
                             factorization_method=None):
    n, m = len(x0), len(constr0)
    box = _Box(xp, n, trust_lb, trust_ub)
    stages = _stages(xp, n, m, x0, box)
    if state.niter == 0:
        # a new solve starts from the same launch plan whatever ran before it: its first normal
        # step is expected long, its first Hessian keeps its diagonal apart (backend_hip
        # .hessian_operator: that choice rounds differently, so it must not depend on history)
        if hasattr(stages, "chain"):
            stages.chain.expect_dogleg = True
            stages.chain.expect_steps = False
            stages.chain.last_niter = (3, 3)
        if hasattr(xp, "note_cg_length"):
            xp.note_cg_length(0)

    def publish(pt):
        state.x, state.v, state.fun, state.grad = pt.x, pt.v, float(pt.f), pt.c
        state.constr, state.jac = pt.b, pt.A
        state.optimality, state.constr_violation = pt.opt, pt.viol

    pt = _Point()
    pt.x, pt.f, pt.c, pt.b, pt.A = xp.copy(x0), fun0, grad0, constr0, jac0
    pt.S = scaling(pt.x) if scaling is not None else None
    stages.settle(pt, factorization_method)          # (the method is honoured here only, :81/:225)
    radius, penalty = initial_trust_radius, initial_penalty
    publish(pt)
    state.niter += 1
    state.trust_radius, state.penalty = radius, penalty
    if return_all:
        state.allvecs += [xp.copy(pt.x)]
        state.allmult += [xp.copy(pt.v)]

    H, fresh = None, True
    while not stop_criteria(state):
        if fresh:                          # (not recomputed after a rejected step, :101-106)
            H = lagr_hess(pt.x, pt.v)
            state.nhev += 1
        q, trial = stages.propose(pt, H, radius, penalty)
        f_next, b_next = fun_and_constr(trial.x_next)
        state.nfev += 1
        state.ncev += 1
        q = stages.judge(q, trial, f_next, b_next)
        if q is None:
            # (the proposing chain had left its step to the host: ChainStages.propose)
            f_next, b_next = fun_and_constr(trial.x_next)
            q = stages.judge(trial.q, trial, f_next, b_next)
        if q[SOC]:
            f_next, b_next = _second_order_correction(xp, box, pt, trial, q, f_next, b_next,
                                                      fun_and_constr, state)
            sc.radius_host(q)
        radius, penalty = q[RADIUS], q[PENALTY]
        state.niter += 1
        if q[ACCEPT]:
            nxt = _Point()
            nxt.x, nxt.f, nxt.b = trial.x_next, f_next, b_next
            nxt.c, nxt.A = grad_and_jac(nxt.x)
            nxt.S = scaling(nxt.x) if scaling is not None else None
            state.ngev += 1
            state.njev += 1
            stages.settle(nxt, None)
            _retire(pt, nxt)
            pt, fresh = nxt, True
            publish(pt)
        else:
            fresh = False
        state.trust_radius, state.penalty = radius, penalty
        state.cg_niter += trial.cg_info["niter"]
        state.cg_info = trial.cg_info
        if return_all:
            state.allvecs.append(xp.copy(pt.x))
            state.allmult.append(xp.copy(pt.v))
    return state


def _retire(old, new):
    """The iterate an accepted step leaves behind: what was cached ON its Jacobian (the
    factorization, the settled multipliers) refers back to that matrix -- a reference cycle
    that only the cyclic collector would free, some solves later, and with it the factorization
    handle and its device buffers, which the next factorization then cannot recycle (measured
    before this: 12 of a config-3 solve's 15 factorizations created a new handle, a dozen
    hipMalloc / hipFree each).  Dropped here, the old factorization dies with the step."""
    A = old.A
    if A is new.A:
        return                       # (a constant Jacobian: one factorization for the whole run)
    for attr in ("_ipx_projections", "_ipx_settled"):
        if getattr(A, attr, None) is not None:
            try:
                setattr(A, attr, None)
            except AttributeError:
                pass


def _second_order_correction(xp, box, pt, trial, q, f_next, b_next, fun_and_constr, state):
    """:172-193.  Tries ``d + t y`` with ``y = -Y b_next`` cut to the box; takes it when it is
    inside and reduces the merit function enough.  Leaves the ratio, ``f_next`` and
    ``||b_next||`` of whichever point stands in the block."""
    d = trial.d
    y = -pt.Y.dot(b_next)
    if box.any:
        _, t, intersect = xp.box_intersections(d, y, box.lb_full, box.ub_full)
    else:
        # an unbounded box never clips the segment (ta, tb = 0, 1) unless y == 0
        intersect = xp.norm(y) != 0
        t = 1 if intersect else 0
    step = d + t * y
    x_soc = pt.x + (pt.S.dot(step) if pt.S is not None else step)
    f_soc, b_soc = fun_and_constr(x_soc)
    f_soc = float(f_soc)
    state.nfev += 1
    state.ncev += 1
    norm_b_soc = xp.norm(b_soc)
    ratio_soc = (q[MERIT] - (f_soc + q[PENALTY] * norm_b_soc)) / q[PRED]
    if intersect and ratio_soc >= _SUFFICIENT:
        trial.x_next = x_soc
        q[RATIO], q[F_NEXT], q[NORM_B_NEXT] = ratio_soc, f_soc, norm_b_soc
        return f_soc, b_soc
    return f_next, b_next
