"""the strong-scaling ceiling measured where it can be -- on ONE GPU at the
per-rank sizes of an N-GPU run.  For N = 1, 2, 4, 8 the device-resident projected-CG loop runs
on the config-3 subproblem at n = 1e6 / N, m = 1e5 / N (no communication: what one rank would
have to do between its collectives), in every form of the loop the library has; the implied
bound on the speed-up is t(1e6) / t(1e6 / N).

    python scripts/per_rank_sweep.py [out.json]        (profiles/r04_per_rank_sweep.json)

Forms: "three_launches" (step1 + A.r | cyclic reduction + g | step2 + H.p), "resident" (the
whole iteration in one resident launch with hand-rolled grid hand-offs, when the library has
it and the size fits).  Finite trust radius that is never reached (the SQP's usage) and
trust_radius = inf."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from ipsolver import _hip, cg_fused, projector
from ipsolver import device as dv
from ipsolver.operators import DeviceHessian
from ipsolver.synthetic import CenteredBandedNLP

K, W, REGIONS, SEG = 200, 40, 9, 200


def measure(n, m, form):
    prob = CenteredBandedNLP(n, m, seed=0)
    x = prob.x0
    v = 0.1 * np.random.default_rng(7).standard_normal(m)
    A = dv.DeviceCSR.from_scipy(prob.constr_jac(x))
    H = DeviceHessian(n, csr=dv.DeviceCSR.from_scipy(prob.hess(x)),
                      diag=dv.DVec.from_host(prob.kappa * prob.Wt.dot(v)))
    c = dv.DVec.from_host(prob.grad(x))
    b = dv.DVec.zeros(m)
    Z, LS, Y = projector.projections(A)
    P = Z.projector
    x0 = Y.dot(-b)
    r0 = Z.dot(H.dot(x0) + c)
    g0 = Z.dot(r0)
    rt_g = g0.sumsq_amax()[0]
    lib = _hip.load()
    for k, val in form.get("env", {}).items():
        os.environ[k] = val
    try:
        L = cg_fused._Loop(H, P, None, None, **form.get("loop_kw", {}))
    finally:
        for k in form.get("env", {}):
            os.environ.pop(k, None)
    if form.get("need") and not form["need"](L):
        return None
    st = dv.stream_ptr()
    out = {}
    for label, radius in (("finite_radius", 1e300), ("radius_inf", np.inf)):
        init = np.zeros(L.state.numel())
        init[cg_fused.ST_RTG0], init[cg_fused.ST_TOL] = rt_g, 0.0
        init[cg_fused.ST_RADIUS] = radius
        init[cg_fused.ST_ORTH_RHS] = P.orth_tol * P.norm_A
        init_d = torch.from_numpy(init).to(L.state.device)
        L.args.no_radius = 0 if np.isfinite(radius) else 1

        def prime():
            L.x.copy_(x0.t)
            L.r.copy_(r0.t)
            _hip.call("ipx_axpby", n, -1.0, dv._p(g0.t), 0.0, None, dv._p(L.p), st)
            L.state.copy_(init_d)
            _hip.check(lib.ipx_cg_hp(L.ref(), st), "ipx_cg_hp")

        def run(k):
            it = 0
            while it < k:
                if it % SEG == 0:
                    prime()
                end = min(k, it - it % SEG + SEG)
                _hip.check(lib.ipx_cg_iterate(L.ref(), it % SEG, it % SEG + end - it, st), "iterate")
                it = end

        run(W)
        times = []
        for _ in range(REGIONS):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(K)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) / K)
        s = L.state.tolist()
        if int(s[cg_fused.ST_STOP]) != 0 or int(s[cg_fused.ST_IT_DONE]) != K:
            raise SystemExit("%s n=%d: stop=%s done=%s" % (form["name"], n, s[cg_fused.ST_STOP],
                                                           s[cg_fused.ST_IT_DONE]))
        times.sort()
        out[label] = {"us_per_iteration_median": 1e6 * times[len(times) // 2],
                      "us_per_iteration_min": 1e6 * times[0]}
        xs = L.x.clone()
        out[label]["x_checksum"] = float(xs.double().abs().sum().item())
        out[label]["_x"] = xs
    return out


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles",
                                                                  "r04_per_rank_sweep.json")
    forms = [{"name": "three_launches", "loop_kw": {"resident": False}},
             {"name": "resident", "loop_kw": {"resident": True},
              "need": lambda L: bool(getattr(L.args, "resident", 0))}]
    rows = []
    for N in (1, 2, 4, 8):
        n, m = 1000000 // N, 100000 // N
        row = {"N": N, "n": n, "m": m}
        ref_x = {}
        for form in forms:
            r = measure(n, m, form)
            if r is None:
                continue
            for label in ("finite_radius", "radius_inf"):
                xs = r[label].pop("_x")
                if label in ref_x:
                    r[label]["max_abs_diff_vs_three_launches"] = float((xs - ref_x[label]).abs().max().item())
                else:
                    ref_x[label] = xs
            row[form["name"]] = r
        rows.append(row)
        print(json.dumps(row), flush=True)
    base = rows[0]
    for row in rows:
        names = [f["name"] for f in forms]
        for name in names:
            if name in row:
                best1 = min(base[f]["finite_radius"]["us_per_iteration_median"]
                            for f in names if f in base)
                row[name]["implied_speedup_bound_vs_best_N1"] = \
                    best1 / row[name]["finite_radius"]["us_per_iteration_median"]
    doc = {"what": "device-resident projected-CG loop on ONE MI355X at the per-rank sizes of an "
                   "N-GPU run of the n=1e6 / m=1e5 problem (no communication); %d regions of %d "
                   "iterations, host clock around stream-synchronised regions" % (REGIONS, K),
           "rows": rows}
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
    print("written", out_path)


if __name__ == "__main__":
    main()
