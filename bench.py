#!/usr/bin/env python3
"""Headline benchmark: projected-CG iterations/s (fp64) at n=1e6, m=1e5.

A "step" is ONE projected-CG iteration (reference qp_subproblem.py:549-634) on
the synthetic sparse banded problem of BASELINE.json config 3 (SURVEY.md
Appendix C: CSR Jacobian 1e5 x 1e6 with 15 nnz/row, tridiagonal-plus-diagonal
Lagrangian Hessian).  `value` times the PRODUCT FUNCTION: one call
``ipsolver.qp.projected_cg(H, c, Z, Y, b, trust_radius=1e300, tol=0,
max_iter=K)`` -- its priming (qp_subproblem.py:502-512), the batched device
loop, its state reads, the result -- with a finite trust radius that is never
reached (how the SQP calls it: the norm test of qp_subproblem.py:583 is formed
and taken every iteration), so exactly K iterations execute.  The bare device
loop (one C call, no priming: the headline of rounds 1-3) is reported as
`device_loop_only`, the same loop with trust_radius=inf (the norm test cannot
trigger and is not formed) as `unbounded_trust_region`.  All inputs are
resident in HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 1000000] [--m 100000]

N > 1 runs one rank per GPU: under torch.distributed.run (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment), or -- started directly, the way
the driver starts N = 1 -- bench.py spawns its N ranks itself as fresh child
processes from a parent that never touches the GPU (``spawn_ranks``).  The SAME
n=1e6 / m=1e5 problem is row-partitioned over the ranks (BASELINE config 4,
ipsolver/sharded.py: constraint rows and variables both partitioned) with two
small RCCL all-reduces per iteration (p'Hp; the packed norms) and a neighbour
exchange of the halo of g; value = iterations of the shared problem / max
time over ranks ("strong" scaling).  The JSON also carries the measured
latency floor of those collectives, a weak-scaling point (n = N * 1e6) and the
whole config-4 solve on the sharded backend.

Prints ONE JSON line (see the driver contract in the task statement) with
`roofline` for the dominant kernel (step2 fused into the H.p CSR SpMV),
`roofline_out_of_cache` (the same measurement at n=4e6, past the Infinity
Cache), `loop_kernels` (per kernel: algorithmic bytes, PMC bytes moved, their
ratio), `public_api` (the product function over several calls, both radii),
`config5` / `config2` (the other
single-GPU BASELINE configs to gtol) and `cpu_baseline` (the oracle =
numpy/scipy restatement of the reference path, the better of 1 / all host
threads).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_MFMA_PEAK_TFLOPS = 78.6   # AMD datasheet, dense fp64 matrix
PMC_PROFILE = "r06_pmc_traffic_n1e6.json"      # see roofline.traffic_source
PMC_PROFILE_BIG = "r06_pmc_traffic_n16e6.json"


def kernel_source_hash():
    """Hash of the sources of the loop's kernels: a stored PMC traffic figure is attached to
    the bench line only when it was measured on kernels built from exactly these files."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "ip-nonlinear-solver_amd", "csrc")
    # (every source of the library + the ABI header: VERDICT r5 item 5)
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))
                       + [os.path.join(ROOT, "include", "ipx.h")]):
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode())
            h.update(f.read())
    return h.hexdigest()[:16]


def stored_traffic(profile, kernel):
    """(bytes per launch, provenance) from profiles/<profile>, or (None, reason)."""
    path = os.path.join(ROOT, "profiles", profile)
    if not os.path.exists(path):
        return None, "no PMC profile for this build (profiles/%s missing)" % profile
    with open(path) as f:
        pmc = json.load(f)
    have, want = pmc.get("kernel_source_hash"), kernel_source_hash()
    if have != want:
        return None, ("profiles/%s was measured on other kernel sources (hash %s, this build %s): "
                      "not attached" % (profile, have, want))
    alias = {"banded_solve_residual_r_minus_Atv": "banded_solve_pcr"}     # (the summary's name)
    val = (pmc["kernels"].get(kernel) or pmc["kernels"].get(alias.get(kernel, ""), {})) \
        .get("hbm_bytes_per_launch")
    return val, ("STORED, not measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                 "(separate passes, gfx950 FETCH correction calibrated in-run on kernels of known "
                 "byte count) over scripts/pmc_workload.py, profiles/%s, same kernel sources "
                 "(hash %s)" % (profile, want))


def traffic_table(profile, algo):
    """Per loop kernel: algorithmic bytes, the STORED PMC bytes (same-source profile only) and
    their ratio -- above 1 means bytes moved twice (window halos, a second copy of a matrix)."""
    out = {}
    for name, nbytes in algo.items():
        moved, _ = stored_traffic(profile, name)
        out[name] = {"algorithmic_bytes": nbytes, "pmc_bytes_moved": moved,
                     "traffic_over_algorithmic": None if not moved else moved / nbytes}
    return out


def spmv_bytes(nnz, rows, cols, extra_row_vectors=0):
    """Algorithmic HBM bytes of one CSR SpMV launch."""
    return 12 * nnz + 4 * (rows + 1) + 8 * rows + 8 * cols + 8 * rows * extra_row_vectors


def loop_bytes(n, m, nnzA, nnzH, fused_step1=True, fused_step2=True, fused_tail=True,
               diag_separate=False):
    """THE byte ledger of one projected-CG iteration (qp_subproblem.py:549-634) on the banded
    problem, per kernel, as SURVEY.md 8(d) counts: CSR SpMV = 12 nnz + 4 (rows + 1) + 8 rows +
    8 cols (+ 8 rows per further row vector), a vector pass = 8 bytes per element and direction.
    Returns (per-kernel dict, total).  Every figure quoted for "algorithmic bytes per iteration"
    -- this file's roofline blocks, DESIGN.md, README.md -- is this function's."""
    algo = {
        "spmv_H_p": spmv_bytes(nnzH, n, n, extra_row_vectors=1 if diag_separate else 0),
        "spmv_A_r": spmv_bytes(nnzA, m, n),
        "spmv_r_minus_Atv": spmv_bytes(nnzA, n, m, extra_row_vectors=1),
        "step1": 5 * 8 * n,     # read x,p,r,Hp; write r
        "step2": 5 * 8 * n,     # read x,p,g;   write x,p
    }
    if fused_step2:     # p is read once instead of twice
        algo["step2_spmv_H_p"] = algo.pop("spmv_H_p") + algo.pop("step2") - 8 * n
    if fused_step1:     # r_next is not read back by the SpMV
        algo["step1_spmv_A_r"] = algo.pop("spmv_A_r") + algo.pop("step1") - 8 * n
    if fused_tail:      # v is not read back by the SpMV
        algo["banded_solve_residual_r_minus_Atv"] = algo.pop("spmv_r_minus_Atv") - 8 * m
    # (+ the cyclic-reduction solve's own traffic: w in, v out, the band: 4 m-vector passes)
    return algo, sum(algo.values()) + 4 * 8 * m


def single_gpu_measure(A_h, H_h, hdiag_h, c_h, n, m, K, W, repeats=20):
    """Upload one subproblem, factor, prime the device loop like projected_cg does, then
    W warm-up + EXACTLY K timed iterations (the contract's region), `repeats` further K-step
    regions (median / min / max), HIP-event attribution per kernel, and the dominant kernel
    on its own.  The timed workload has a FINITE trust radius that is never reached (1e300):
    ||x + alpha p||^2 is formed and tested every iteration, as in every call the SQP makes;
    the trust_radius=inf variant (test not formed) is measured after it."""
    import numpy as np
    import torch
    from ipsolver import _hip, cg_fused, projector
    from ipsolver import device as dv
    from ipsolver.operators import DeviceHessian
    lib = _hip.load()
    A = dv.DeviceCSR.from_scipy(A_h)
    H = DeviceHessian(n, csr=dv.DeviceCSR.from_scipy(H_h), diag=dv.DVec.from_host(hdiag_h))
    c = dv.DVec.from_host(c_h)
    b = dv.DVec.zeros(m)
    torch.cuda.synchronize()
    t0 = time.time()
    Z, LS, Y = projector.projections(A)
    torch.cuda.synchronize()
    t_factor = time.time() - t0
    P = Z.projector

    # ---- prime the loop exactly like projected_cg does (untimed)
    x0 = Y.dot(-b)
    r0 = Z.dot(H.dot(x0) + c)
    g0 = Z.dot(r0)
    rt_g = g0.sumsq_amax()[0]
    L = cg_fused._Loop(H, P, None, None)
    st = dv.stream_ptr()

    def state_for(radius):
        init = np.zeros(L.state.numel())
        init[cg_fused.ST_RTG0] = rt_g
        init[cg_fused.ST_TOL] = 0.0
        init[cg_fused.ST_RADIUS] = radius
        init[cg_fused.ST_ORTH_RHS] = P.orth_tol * P.norm_A
        return torch.from_numpy(init).to(L.state.device)
    init_finite, init_inf = state_for(1e300), state_for(np.inf)
    mode = {"init": init_finite}
    L.args.no_radius = 0

    def prime():
        """x = x0, r = Z(H x0 + c), p = -g, state reset, Hp = H p: device copies and one
        SpMV, no host synchronisation."""
        L.x.copy_(x0.t)
        L.r.copy_(r0.t)
        _hip.call("ipx_axpby", n, -1.0, dv._p(g0.t), 0.0, None, dv._p(L.p), st)
        L.state.copy_(mode["init"])
        _hip.check(lib.ipx_cg_hp(L.ref(), st), "ipx_cg_hp")

    # With tol = 0 the CG would run into an exactly zero residual after ~500 iterations
    # (p'Hp = 0 ends it).  Whatever K is asked for, the loop is therefore restarted from
    # the same subproblem every SEG iterations (five device launches, <0.3 % of a segment).
    SEG = 200

    def run(it0, it1, what):
        it = it0
        while it < it1:
            j = it % SEG
            if j == 0:
                prime()
            end = min(it1, it - j + SEG)
            _hip.check(lib.ipx_cg_iterate(L.ref(), j, j + (end - it), st), what)
            it = end

    def check_ran(total):
        s = L.state.tolist()
        expected_done = (total - 1) % SEG + 1 if total > 0 else 0
        if int(s[cg_fused.ST_STOP]) != 0 or int(s[cg_fused.ST_IT_DONE]) != expected_done:
            raise SystemExit("timed region did not run its iterations: stop=%s done=%s (expected "
                             "%d in the last segment)" % (s[cg_fused.ST_STOP],
                                                          s[cg_fused.ST_IT_DONE], expected_done))

    # ---- warmup, then EXACTLY K timed iterations
    run(0, W, "warmup")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(W, W + K, "timed")
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    check_ran(W + K)
    # ---- the same K-step region again, `repeats` times (spread of the measurement)
    rates = []
    done = W + K
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(done, done + K, "repeat")
        torch.cuda.synchronize()
        rates.append(K / (time.perf_counter() - t0))
        done += K
    check_ran(done)
    rates.sort()
    repeat = {"regions": repeats, "steps_each": K,
              "iterations_per_s": {"median": rates[len(rates) // 2], "min": rates[0],
                                   "max": rates[-1]}} if rates else None

    fused1, fused2, fused3 = bool(L.args.A_span), bool(L.args.H_hmax), bool(L.args.At_qv)

    # ---- the same loop with trust_radius = inf (the reference's default argument): the test
    # norm(x_next) >= inf of qp_subproblem.py:583 cannot trigger, the norm is not formed and the
    # fused step1 + A.r kernel reads neither x nor p
    unbounded = None
    if repeats > 0:
        L.args.no_radius = 1
        mode["init"] = init_inf
        fr = []
        run(0, W, "unbounded warmup")
        done_f = W
        for _ in range(min(5, repeats)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(done_f, done_f + K, "unbounded")
            torch.cuda.synchronize()
            fr.append(K / (time.perf_counter() - t0))
            done_f += K
        check_ran(done_f)
        fr.sort()
        unbounded = {"trust_radius": "inf", "iterations_per_s": fr[len(fr) // 2],
                     "regions": len(fr), "steps_each": K,
                     "algorithmic_bytes_less_per_iteration": 2 * 8 * n if fused1 else 0}
        L.args.no_radius = 0
        mode["init"] = init_finite

    # ---- the dominant kernel on its own: back-to-back launches (same arguments as in the
    # loop) between two HIP events on the launch stream, in `repeats` groups of K.  This is
    # the kernel's own duration, the quantity rocprofv3 --kernel-trace reports.
    # With a banded Hessian step2 rides inside the H.p SpMV (k_cg_step2_hp): the
    # fused kernel is then the dominant one (mode 3 = no stop tests).
    def dominant():
        if fused2:
            return lib.ipx_cg_step2_hp(L.ref(), 0, 3, st)
        return lib.ipx_cg_hp(L.ref(), st)
    for _ in range(20):
        dominant()
    groups = []
    for _ in range(max(1, repeats)):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(K):
            dominant()
        ev1.record()
        torch.cuda.synchronize()
        groups.append(1e3 * ev0.elapsed_time(ev1) / K)
    groups.sort()
    hp_us = groups[len(groups) // 2]

    nnzA, nnzH = A.pattern.nnz, H.csr.pattern.nnz
    # (the diagonal term rides in the CSR values when the pattern has every diagonal entry:
    # SURVEY.md 8(d)'s "tridiagonal + diagonal Hessian, nnz ~ 3e6" as ONE matrix)
    algo, iter_bytes = loop_bytes(n, m, nnzA, nnzH, fused1, fused2, fused3,
                                  diag_separate=H.diag is not None)
    dom = "step2_spmv_H_p" if fused2 else "spmv_H_p"
    dom_label = ("k_cg_step2_hp (step2 fused into the H.p SpMV, p'Hp epilogue)" if fused2
                 else "k_csr_spmv (H.p with p'Hp epilogue)")
    achieved = algo[dom] / (hp_us * 1e-6) / 1e9
    # rounds 1-2 read the Hessian's diagonal term as a vector of its own and counted its 8n
    # bytes; merged into the CSR values (SURVEY.md 8(d)'s count: ONE matrix) they are neither
    # moved nor counted.  The same launch priced on the older count, for comparison across rounds:
    merged_diag = H.diag is None and hdiag_h is not None
    older = (algo[dom] + 8 * n) / (hp_us * 1e-6) / 1e9 if merged_diag else None
    med_rate = repeat["iterations_per_s"]["median"] if repeat else K / elapsed
    return {
        "A": A, "H": H, "c": c, "b": b, "Z": Z, "Y": Y, "elapsed": elapsed, "repeat": repeat,
        "unbounded": unbounded,
        "t_factor": t_factor, "nnzA": nnzA, "nnzH": nnzH, "dom": dom,
        "roofline": {"bound": "hbm", "kernel": dom_label,
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "algorithmic_bytes_per_launch": algo[dom],
                     "avg_launch_us": hp_us,
                     "avg_launch_us_min_max": [groups[0], groups[-1]],
                     "method": "median of %d groups of %d back-to-back launches, each between "
                               "two HIP events on the launch stream" % (len(groups), K),
                     "on_the_byte_count_of_rounds_1_2": None if older is None else {
                         "algorithmic_bytes_per_launch": algo[dom] + 8 * n, "achieved": older,
                         "frac": older / HBM_PEAK_GBS,
                         "note": "the Hessian's diagonal term counted as a separate 8n-byte "
                                 "vector, as it was read until round 3 merged it into the CSR "
                                 "values: same operator, same launch duration"}},
        "working_set_bytes": (12 * (2 * nnzA + nnzH) + 18 * n + 4 * (2 * n + m) + 8 * (5 * n + 5 * m)),
        "whole_iteration": {"algorithmic_bytes": iter_bytes,
                            "algorithmic_bytes_per_kernel": algo,
                            "achieved_GBs": iter_bytes * med_rate / 1e9,
                            "frac_of_hbm_peak": iter_bytes * med_rate / 1e9 / HBM_PEAK_GBS,
                            "at": "median of the repeated regions"},
    }


def per_rank_sweep_leg(K=200, W=40, regions=5):
    """The strong-scaling ceiling measured where one GPU can measure it (VERDICT r4 item 1a, r5
    item 4): the loop at the PER-RANK sizes of an N-GPU run (N = 1, 2, 4, 8; no communication)
    of the n = 1e6 / m = 1e5 problem AND of problems 4 and 16 times its size, in its two forms
    -- the three launches per iteration and the resident kernel (one launch per batch,
    csrc/resident.hip; the form the sharded loop takes where it fits: its cross-rank hand-offs
    are the same tagged words, sent over xGMI).  Finite trust radius that is never reached.
    bound(n, N) = t(n, best form) / t(n / N, best form): what N GPUs can gain at most on a
    problem of n variables before a byte crosses xGMI."""
    import numpy as np
    import torch
    from ipsolver import _hip, cg_fused, projector
    from ipsolver import device as dv
    from ipsolver.operators import DeviceHessian
    from ipsolver.synthetic import CenteredBandedNLP
    lib = _hip.load()
    st = dv.stream_ptr()
    SEG = 200
    totals = (1000000, 4000000, 16000000)
    sizes = sorted({t // N for t in totals for N in (1, 2, 4, 8)})
    by_n = {}
    for n in sizes:
        m = n // 10
        prob = CenteredBandedNLP(n, m, seed=0)
        x = prob.x0
        v = 0.1 * np.random.default_rng(7).standard_normal(m)
        A = dv.DeviceCSR.from_scipy(prob.constr_jac(x))
        H = DeviceHessian(n, csr=dv.DeviceCSR.from_scipy(prob.hess(x)),
                          diag=dv.DVec.from_host(prob.kappa * prob.Wt.dot(v)))
        c = dv.DVec.from_host(prob.grad(x))
        del prob
        Z, LS, Y = projector.projections(A)
        P = Z.projector
        x0 = Y.dot(-dv.DVec.zeros(m))
        r0 = Z.dot(H.dot(x0) + c)
        g0 = Z.dot(r0)
        rt_g = g0.sumsq_amax()[0]
        row = {"n": n, "m": m}
        reg = regions if n <= 1000000 else 3
        for form, kw in (("three_launches", {"resident": False}), ("resident", {"resident": True})):
            L = cg_fused._Loop(H, P, None, None, **kw)
            if form == "resident" and not L.args.resident:
                row[form] = None               # (does not fit: more blocks than compute units)
                continue
            init = np.zeros(L.state.numel())
            init[cg_fused.ST_RTG0], init[cg_fused.ST_RADIUS] = rt_g, 1e300
            init[cg_fused.ST_ORTH_RHS] = P.orth_tol * P.norm_A
            init_d = torch.from_numpy(init).to(L.state.device)

            def run(k):
                it = 0
                while it < k:
                    if it % SEG == 0:
                        L.x.copy_(x0.t)
                        L.r.copy_(r0.t)
                        _hip.call("ipx_axpby", n, -1.0, dv._p(g0.t), 0.0, None, dv._p(L.p), st)
                        L.state.copy_(init_d)
                        _hip.check(lib.ipx_cg_hp(L.ref(), st), "ipx_cg_hp")
                    end = min(k, it - it % SEG + SEG)
                    _hip.check(lib.ipx_cg_iterate(L.ref(), it % SEG, it % SEG + end - it, st), "iterate")
                    it = end
            run(W)
            times = []
            for _ in range(reg):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(K)
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t0) / K)
            sstate = L.state.tolist()
            if int(sstate[cg_fused.ST_STOP]) != 0 or int(sstate[cg_fused.ST_IT_DONE]) != K:
                raise RuntimeError("per-rank sweep, %s at n=%d: stop=%s done=%s"
                                   % (form, n, sstate[cg_fused.ST_STOP], sstate[cg_fused.ST_IT_DONE]))
            row[form] = {"us_per_iteration": 1e6 * sorted(times)[len(times) // 2]}
            del L
        row["best_us_per_iteration"] = min(r["us_per_iteration"]
                                           for r in (row["three_launches"], row["resident"]) if r)
        by_n[n] = row
        del A, H, Z, Y, LS, P, c, x0, r0, g0
        torch.cuda.empty_cache()
    tables = []
    for total in totals:
        t1 = by_n[total]["best_us_per_iteration"]
        tables.append({"n_total": total, "us_per_iteration_on_one_gpu": t1,
                       "bound_by_N": {str(N): t1 / by_n[total // N]["best_us_per_iteration"]
                                      for N in (2, 4, 8)}})
    # the rows of the n = 1e6 problem in the shape of rounds 4-5 (N, n, the two forms)
    rows = []
    t1 = by_n[1000000]["best_us_per_iteration"]
    for N in (1, 2, 4, 8):
        r = dict(by_n[1000000 // N], N=N)
        for form in ("three_launches", "resident"):
            if r.get(form):
                r[form] = dict(r[form], speedup_bound=t1 / r[form]["us_per_iteration"])
        rows.append(r)
    res8 = rows[-1]["resident"]
    return {"what": "device loop on ONE GPU at the per-rank sizes of an N-GPU run (n_total / N), "
                    "no communication: median of %d (3 beyond n = 1e6) regions of %d iterations"
                    % (regions, K),
            "rows": rows,
            "per_rank_size": [by_n[n] for n in sizes],
            "bound_by_problem_size": tables,
            "reading": "bound_by_N = what N GPUs can gain at most (communication free) on a problem "
                       "of n_total variables; the >= 6x target at N = 8 needs the per-rank loop to "
                       "stay on the HBM roofline, i.e. a problem large enough that n_total / 8 still "
                       "fills a GPU",
            "analytic_floor": {
                "formula": "t_iteration(N) >= t_resident(1e6 / N) + 2 * (t_hop_xgmi - t_hop_on_chip): "
                           "an iteration has two dependent all-to-all hand-offs (p'Hp; the packed "
                           "norms + the halo of g) that the resident kernel's time already contains "
                           "at their ON-CHIP cost; between GPUs each becomes a store over xGMI + a "
                           "poll of local memory",
                "t_hop_on_chip_us": 2.5,
                "t_hop_on_chip_source": "profiles/r04_resident_phase_timing.txt (hop 1 incl. fold)",
                "t_hop_xgmi_us": None,
                "t_hop_xgmi_source": "bench.py --gpus N preflight: mailbox_pingpong_us / 2 (needs "
                                     "two GPUs; never measured in this build)",
                "bound_at_N8_with_free_communication": None if not res8 else res8["speedup_bound"],
                "target": 6.0}}


def measured_stream():
    """What plain streaming kernels reach on this part in THIS run ("report the
    achieved fraction against 8.0 TB/s and also against a same-run measured copy kernel"):
    the library's own out = a x + b y (2 reads + 1 write per element, 16-byte lanes) and the
    runtime's device-to-device copy, on 512 MiB arrays (past the Infinity Cache), 20 launches
    between two HIP events."""
    import torch
    from ipsolver import _hip
    from ipsolver import device as dv
    n = 1 << 26
    dev = dv.ctx().device
    x = torch.ones(n, dtype=torch.float64, device=dev)
    y = torch.ones(n, dtype=torch.float64, device=dev)
    o = torch.empty(n, dtype=torch.float64, device=dev)
    st = dv.stream_ptr()
    triad = lambda: _hip.call("ipx_axpby", n, 1.5, dv._p(x), 0.5, dv._p(y), dv._p(o), st)
    copy = lambda: o.copy_(x)
    out = {}
    for name, fn, nbytes in (("axpby_2r1w", triad, 24 * n), ("device_copy_1r1w", copy, 16 * n)):
        for _ in range(3):
            fn()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(20):
            fn()
        ev1.record()
        torch.cuda.synchronize()
        out[name] = nbytes / (ev0.elapsed_time(ev1) / 20 * 1e-3) / 1e9
    del x, y, o
    out.update(unit="GB/s", bytes_per_array=8 * n,
               note="20 launches between HIP events on 512 MiB arrays")
    return out


def public_api_leg(H, c, Z, Y, b, K, reps=5):
    """The product function as a user calls it:
    ``ipsolver.qp.projected_cg(H, c, Z, Y, b, tol=0, max_iter=K)`` -- initial projections,
    buffer set-up, the batched device loop with its state polls, the result -- wall clock
    around the call, for trust_radius = inf and for a finite radius that is never reached."""
    import numpy as np
    import torch
    from ipsolver import qp, cg_fused
    out = {"call": "ipsolver.qp.projected_cg(H, c, Z, Y, b, trust_radius, tol=0, max_iter=%d)" % K}
    for name, radius in (("trust_radius_finite", 1e300), ("trust_radius_inf", np.inf)):
        qp.projected_cg(H, c, Z, Y, b, trust_radius=radius, tol=0, max_iter=K)      # warm
        rates, batches = [], 0
        for _ in range(reps):
            b0 = cg_fused.STATS["batches"]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            x, info = qp.projected_cg(H, c, Z, Y, b, trust_radius=radius, tol=0, max_iter=K)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if info["niter"] != K:
                raise SystemExit("public_api leg: %d iterations instead of %d" % (info["niter"], K))
            rates.append(K / dt)
            batches = cg_fused.STATS["batches"] - b0
        rates.sort()
        out[name] = {"iterations_per_s": rates[len(rates) // 2], "min": rates[0],
                     "max": rates[-1], "calls": reps, "state_reads_per_call": batches}
    return out


def config3_leg(n, m, repeats=5):
    """BASELINE config 3 to gtol on one GPU (minimize_constrained, tr_interior_point,
    device-callback mode: nothing crosses PCIe): the wall clock (median of ``repeats`` warm
    solves), how often the host blocked on the device and how many kernels the library
    launched in one solve, and the projected-CG rate INSIDE the solve (SURVEY.md 8(d)(i):
    cg_niter / time in projected_cg -- here the GPU time between the first launch of a call's
    priming and the last launch of its iterations, HIP events inside ipx_sqp_front, plus the
    host-timed continuations of the calls whose first batch was too short)."""
    import ctypes
    import warnings
    import torch
    import ipsolver
    from ipsolver import _hip, sqp, sqp_chain
    from ipsolver.synthetic import CenteredBandedNLP, DeviceCallbacks, LeanDeviceCallbacks
    prob = CenteredBandedNLP(n, m, eps=1e-3)
    lib = _hip.load()

    def solve(dc):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess,
                                            dc.constraints(ipsolver), method="tr_interior_point")
        torch.cuda.synchronize()
        return res, time.perf_counter() - t0

    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, cls in (("lean", LeanDeviceCallbacks), ("torch", DeviceCallbacks)):
            dc = cls(prob)
            solve(dc)                               # (pays the one-off symbolic set-up)
            walls = []
            for _ in range(repeats):
                res, dt = solve(dc)
                walls.append(dt)
            walls.sort()
            out[name] = (res, walls, dc)
        res, walls, dc = out["lean"]
        # the same solves with the chain waited for before the callbacks at the trial point are
        # called (the reference's order of evaluations exactly; three reads per iteration)
        sqp.EVALUATE_BEHIND_THE_CHAIN = False
        try:
            reads0 = int(lib.ipx_read_count())
            res_w, _ = solve(dc)
            reads_w = int(lib.ipx_read_count()) - reads0
            walls_w = sorted(solve(dc)[1] for _ in range(repeats))
        finally:
            sqp.EVALUATE_BEHIND_THE_CHAIN = True
        # one more solve with the calls counted, one with the CG bracketed by events
        launches0, reads0 = int(lib.ipx_launch_count()), int(lib.ipx_read_count())
        solve(dc)
        launches = int(lib.ipx_launch_count()) - launches0
        counts = {"reads": int(lib.ipx_read_count()) - reads0}
        before = dict(sqp_chain.STATS)
        sqp.TIMERS["host_cg_seconds"] = 0.0
        lib.ipx_sqp_cg_timing(1, None, None)
        res_t, _ = solve(dc)
        ms, calls = ctypes.c_double(0.0), ctypes.c_int(0)
        lib.ipx_sqp_cg_timing(0, ctypes.byref(ms), ctypes.byref(calls))
        cg_seconds = 1e-3 * ms.value + sqp.TIMERS["host_cg_seconds"]
        chain = {k: sqp_chain.STATS[k] - before.get(k, 0) for k in sqp_chain.STATS
                 if sqp_chain.STATS[k] - before.get(k, 0)}
    res_p, walls_p, _ = out["torch"]
    return {
        "seconds": walls[len(walls) // 2], "seconds_min": walls[0], "seconds_max": walls[-1],
        "repeats": repeats, "status": int(res.status), "niter": int(res.niter),
        "cg_niter": int(res.cg_niter), "nfev": int(res.nfev),
        "optimality": float(res.optimality), "constr_violation": float(res.constr_violation),
        "blocking_reads_per_solve": counts["reads"],
        "library_launches_per_solve": launches,
        "cg_iterations_per_s_in_solve": res_t.cg_niter / cg_seconds if cg_seconds else None,
        "seconds_in_projected_cg": cg_seconds, "projected_cg_calls": int(calls.value),
        "chain": chain,
        "callbacks": "synthetic.LeanDeviceCallbacks (user-land: torch elementwise ops + the "
                     "library's DeviceCSR / ScalarPack; ~20 launches per outer iteration)",
        "waiting_for_the_chain_first": {
            "seconds": walls_w[len(walls_w) // 2], "blocking_reads_per_solve": reads_w,
            "status": int(res_w.status), "niter": int(res_w.niter), "cg_niter": int(res_w.cg_niter),
            "note": "ipsolver.sqp.EVALUATE_BEHIND_THE_CHAIN = False: the proposing chain's block "
                    "is read before the objective / constraints are called at the trial point "
                    "(default: they and the verdict are enqueued behind the chain, one block "
                    "for both; a step the host has to finish is then evaluated again)"},
        "with_plain_torch_callbacks": {
            "seconds": walls_p[len(walls_p) // 2], "status": int(res_p.status),
            "niter": int(res_p.niter), "cg_niter": int(res_p.cg_niter),
            "callbacks": "synthetic.DeviceCallbacks (rounds 1-5: ~60 launches per outer "
                         "iteration, float(f) by a torch synchronisation)"},
        "note": "config 3 (eps=1e-3), gtol=xtol=1e-8; the reference reaches status 1 in 25 "
                "outer / 34 CG iterations (SURVEY.md Appendix B: 103 s on the survey host).  "
                "blocking reads: every wait of the library for device results (ipx_read_count: "
                "the chains' blocks, ipx_read_doubles / ipx_read_folded; the lean callbacks' "
                "objective leaves its value on the device for the step's verdict); "
                "launches: the library's own (ipx_launch_count), the callbacks' torch kernels "
                "not included"}


def config5_leg(run_twice=True):
    """BASELINE config 5 on one GPU: n=5e5 variables, box on every variable + 5e4 nonlinear
    inequalities (N = 1.55e6 with slacks), tr_interior_point to gtol with device callbacks;
    wall clock and the in-solve projected-CG rate (cg_niter / time inside projected_cg)."""
    import warnings
    import torch
    import ipsolver
    from ipsolver import backend_hip
    from ipsolver.synthetic import CenteredBandedNLP, DeviceCallbacks
    n, m = 500000, 50000
    prob = CenteredBandedNLP(n, m, eps=1.0)
    dc = DeviceCallbacks(prob)
    cons = (dc.constraints(ipsolver, ("less", 0.0)), ipsolver.BoxConstraint(("interval", -0.8, 0.8)))
    # time inside projected_cg: calls through the backend (host-driven stages) by a host timer
    # around them (they return after a blocking read of the loop's state); calls that start
    # inside the outer iteration's front chain by the GPU time of their priming + first batch
    # (HIP events, ipx_sqp_cg_timing) + the host-timed continuation (sqp.TIMERS)
    import ctypes
    from ipsolver import _hip, sqp
    lib = _hip.load()
    timer = {"t": 0.0, "calls": 0}
    plain = backend_hip.projected_cg

    def timed(*a, **k):
        t0 = time.perf_counter()
        out = plain(*a, **k)           # returns after a blocking read of the loop's state
        timer["t"] += time.perf_counter() - t0
        timer["calls"] += 1
        return out

    def clock_start():
        timer["t"], timer["calls"] = 0.0, 0
        sqp.TIMERS["host_cg_seconds"] = 0.0
        lib.ipx_sqp_cg_timing(1, None, None)

    def clock_stop():
        ms, calls = ctypes.c_double(0.0), ctypes.c_int(0)
        lib.ipx_sqp_cg_timing(0, ctypes.byref(ms), ctypes.byref(calls))
        timer["t"] += 1e-3 * ms.value + sqp.TIMERS["host_cg_seconds"]
        timer["calls"] += int(calls.value)
    backend_hip.projected_cg = timed
    other = None
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for attempt in range(2 if run_twice else 1):     # first call pays symbolic set-up
                clock_start()
                torch.cuda.synchronize()
                t0 = time.time()
                res = ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, cons)
                torch.cuda.synchronize()
                wall = time.time() - t0
                clock_stop()
            main = (timer["t"], timer["calls"])
            # the same solve with the back substitution as a launch of its own (round 4's four
            # launches): g is bit-identical, ||g||^2 is summed in another order -- and the run
            # ends on another barrier level (profiles/r05_config5_levels.json)
            try:
                os.environ["IPX_DEBUG_FORMS"] = "no-post-tail"
                clock_start()
                torch.cuda.synchronize()
                t0 = time.time()
                res4 = ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, cons)
                torch.cuda.synchronize()
                wall4 = time.time() - t0
                clock_stop()
                other = {"seconds": wall4, "status": int(res4.status),
                         "niter": int(res4.niter), "cg_niter": int(res4.cg_niter),
                         "cg_iterations_per_s_in_solve": res4.cg_niter / timer["t"] if timer["t"] else None,
                         "launches_per_cg_iteration": 4,
                         "active_bounds": int((res4.x.abs() > 0.8 - 1e-6).sum().item())}
            finally:
                os.environ.pop("IPX_DEBUG_FORMS", None)
            timer["t"], timer["calls"] = main
    finally:
        backend_hip.projected_cg = plain
    x = res.x
    return {"workload": "config5: n=5e5, box + 5e4 nonlinear inequalities, tr_interior_point, "
                        "device callbacks, gtol=xtol=1e-8",
            "seconds": wall, "status": int(res.status), "niter": int(res.niter),
            "cg_niter": int(res.cg_niter), "optimality": float(res.optimality),
            "constr_violation": float(res.constr_violation),
            "projected_cg_calls": timer["calls"], "seconds_in_projected_cg": timer["t"],
            "cg_iterations_per_s_in_solve": res.cg_niter / timer["t"] if timer["t"] else None,
            "launches_per_cg_iteration": 3,
            "launches_note": "k_cg_step1_box, k_solve_pcr (which forms A_R u itself and, since "
                             "round 5, does the per-item back substitution as its tail), "
                             "k_cg_step2_hp by construction (csrc/cg.hip cg_iterate, box_project "
                             "branch); rocprofv3 launch counts: profiles/r05_config5_kernel_stats.csv",
            "with_the_back_substitution_as_its_own_launch": other,
            "trajectory_note": "the two forms compute the same g bit for bit and sum ||g||^2 in "
                               "different orders (one ulp in beta): level by level their CG counts "
                               "agree to 2-7 %, but they pass the last level's stopping test on "
                               "different sides and end one barrier level apart -- compare the "
                               "per-iteration rates, not the wall clocks",
            "active_bounds": int((x.abs() > 0.8 - 1e-6).sum().item())}


def config2_leg():
    """BASELINE config 2: dense random equality-constrained QP n=10000, m=2000,
    equality_constrained_sqp, numpy callbacks (the drop-in usage); wall clock to gtol and the
    fp64 MFMA Gram kernel G = A A' on its own against the matrix-core peak."""
    import warnings
    import numpy as np
    import torch
    import ipsolver
    from ipsolver import _hip
    from ipsolver import device as dv
    from ipsolver.dense import DeviceDense
    n, m = 10000, 2000
    rng = np.random.default_rng(0)                 # the generator's order
    A = rng.standard_normal((m, n))
    G = rng.standard_normal((n, n)) / np.sqrt(n)
    Hd = G.dot(G.T) + np.eye(n)
    c = rng.standard_normal(n)
    bq = A.dot(rng.standard_normal(n))
    del G
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for attempt in range(2):
            torch.cuda.synchronize()
            t0 = time.time()
            res = ipsolver.minimize_constrained(
                lambda x: 0.5 * x.dot(Hd.dot(x)) + c.dot(x), np.zeros(n), lambda x: Hd.dot(x) + c,
                lambda x: Hd, ipsolver.LinearConstraint(A, ("equals", bq)),
                method="equality_constrained_sqp")
            torch.cuda.synchronize()
            wall = time.time() - t0
        # the same call with the Hessian declared constant (additive option; or, equivalently,
        # the array marked H.setflags(write=False)): its device copy is uploaded once, not once
        # per outer iteration
        for attempt in range(2):
            torch.cuda.synchronize()
            t0 = time.time()
            res_ro = ipsolver.minimize_constrained(
                lambda x: 0.5 * x.dot(Hd.dot(x)) + c.dot(x), np.zeros(n), lambda x: Hd.dot(x) + c,
                lambda x: Hd, ipsolver.LinearConstraint(A, ("equals", bq)),
                method="equality_constrained_sqp", options={"constant_hessian": True})
            torch.cuda.synchronize()
            wall_ro = time.time() - t0
    # the same solve with everything resident in HBM (device-callback mode)
    dev = torch.device("cuda", torch.cuda.current_device())
    At, Ht, ct = (torch.from_numpy(a).to(dev) for a in (A, Hd, c))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for attempt in range(2):
            torch.cuda.synchronize()
            t0 = time.time()
            res_d = ipsolver.minimize_constrained(
                lambda x: 0.5 * torch.dot(x, Ht @ x) + torch.dot(ct, x),
                torch.zeros(n, dtype=torch.float64, device=dev), lambda x: Ht @ x + ct,
                lambda x: Ht, ipsolver.LinearConstraint(At, ("equals", bq)),
                method="equality_constrained_sqp")
            torch.cuda.synchronize()
            wall_d = time.time() - t0
    del At, Ht
    # dense NONLINEAR equality constraints at the same size: the Jacobian changes at every
    # accepted step, every one of them pays Gram + Cholesky + inverse (the reference: a pivoted
    # QR per step, 76 % of such a run: projections.py:179)
    dense_nl = None
    try:
        import ipsolver.dense as _dense
        from ipsolver.synthetic import DenseDeviceCallbacks
        cbn = DenseDeviceCallbacks.on_device(n, m)
        built = {"n": 0, "s": 0.0}
        real_init = _dense.DenseNormalSolver.__init__

        def timed_init(self, Amat):
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            real_init(self, Amat)
            torch.cuda.synchronize()
            built["n"] += 1
            built["s"] += time.perf_counter() - t_
        _dense.DenseNormalSolver.__init__ = timed_init
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for attempt in range(4):          # (the last of four: 105, then 78 ms each)
                    built["n"], built["s"] = 0, 0.0
                    torch.cuda.synchronize()
                    t0 = time.time()
                    res_n = ipsolver.minimize_constrained(
                        cbn.fun, cbn.x0, cbn.grad, cbn.hess, cbn.constraints(ipsolver),
                        method="equality_constrained_sqp")
                    torch.cuda.synchronize()
                    wall_n = time.time() - t0
        finally:
            _dense.DenseNormalSolver.__init__ = real_init
        dense_nl = {"workload": "dense nonlinear equality constraints c(x) = A x + kappa/2 W (x*x) "
                                "- b, n=10000, m=2000 (synthetic.DenseDeviceCallbacks.on_device), "
                                "equality_constrained_sqp, device callbacks",
                    "seconds": wall_n, "status": int(res_n.status), "niter": int(res_n.niter),
                    "cg_niter": int(res_n.cg_niter), "optimality": float(res_n.optimality),
                    "constr_violation": float(res_n.constr_violation),
                    "factorizations": built["n"],
                    "ms_per_factorization_gram_cholesky_inverse": 1e3 * built["s"] / max(built["n"], 1)}
        del cbn
        torch.cuda.empty_cache()
    except Exception as exc:                      # never lose the block over the extra problem
        dense_nl = {"error": repr(exc)}
    # the Gram kernel alone
    Ad = DeviceDense.from_host(A)
    lib = _hip.load()
    M = lib.ipx_dense_padded(m)
    Gd = torch.zeros(M * M, dtype=torch.float64, device=Ad.t.device)
    st = dv.stream_ptr()
    splits = int(lib.ipx_gram_splits(m, n))      # K-splits that even the tiles out over the CUs
    ws = torch.empty(max(int(lib.ipx_gram_ws_doubles(m, splits)), 1), dtype=torch.float64,
                     device=Ad.t.device)
    gram = lambda: _hip.call("ipx_gram_f64_mfma_split", m, n, dv._p(Ad.t), n, dv._p(Gd),
                             dv._p(ws), splits, st)          # what DenseNormalSolver launches
    for _ in range(3):
        gram()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        gram()
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 10
    # executed flop: the kernel forms the 64 x 64 tiles on and below the diagonal only (G is
    # symmetric): nt (nt + 1) / 2 tiles of 2 * 64 * 64 * n flop each -- priced against those,
    # not against the 2 m^2 n of a full product
    nt = (M + 63) // 64
    flop = nt * (nt + 1) // 2 * 2.0 * 64 * 64 * n
    tf = flop / (ms * 1e-3) / 1e12
    return {"workload": "config2: dense equality QP n=10000, m=2000, equality_constrained_sqp, "
                        "numpy callbacks, gtol=xtol=1e-8",
            "seconds": wall, "status": int(res.status), "niter": int(res.niter),
            "cg_niter": int(res.cg_niter), "optimality": float(res.optimality),
            "constr_violation": float(res.constr_violation),
            "reference_trace": "status 1, 15 outer / 28 CG (tests/golden/config2.json)",
            "constant_hessian": {"seconds": wall_ro, "status": int(res_ro.status),
                                 "niter": int(res_ro.niter), "cg_niter": int(res_ro.cg_niter),
                                 "optimality": float(res_ro.optimality),
                                 "note": "same numpy callbacks, options={'constant_hessian': True} "
                                         "(or the array marked read-only): the 800 MB Hessian is "
                                         "uploaded once; the rest is the user's own numpy "
                                         "callbacks (dense 10000^2 matvecs on the host)"},
            "device_callbacks": {"seconds": wall_d, "status": int(res_d.status),
                                 "niter": int(res_d.niter), "cg_niter": int(res_d.cg_niter),
                                 "optimality": float(res_d.optimality),
                                 "note": "A and H resident in HBM (2-D CUDA tensors): nothing "
                                         "crosses PCIe between two iterations.  NOT a wall clock "
                                         "to gtol when status is 2: the end game of this QP sits "
                                         "on the merit function's rounding floor and the run "
                                         "stops on xtol at an optimality of a few 1e-8 (the "
                                         "iterate agrees with the reference's)"},
            "dense_nonlinear": dense_nl,
            "gram_mfma": {"kernel": "k_gram_mfma (v_mfma_f64_16x16x4_f64) in %d K-splits + "
                                    "k_gram_reduce" % splits, "ms": ms,
                          "flop_executed": flop, "tiles": [nt * (nt + 1) // 2, nt * nt],
                          "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS,
                          "unit": "TFLOP/s", "frac": tf / FP64_MFMA_PEAK_TFLOPS, "bound": "mfma"}}


def _sharded_setup(A_h, H_h, hdiag_h, c_h, transports=(None,)):
    """Partition one subproblem over the ranks and prime the device-resident sharded loop
    (tol = 0, infinite radius) exactly like projected_cg does; one loop object per requested
    transport (None = the default: peer mailboxes when available)."""
    import numpy as np
    from ipsolver import sharded
    world, rank = sharded.ShardComm().world, sharded.ShardComm().rank
    A_h = A_h.tocsr()
    lay = sharded.ShardLayout(A_h.indptr, A_h.indices, A_h.shape, world, rank)
    sh = sharded.Sharding(lay, sharded.ShardComm(), sharded.HipOps())
    A = sharded.ShardCSR.from_global(sh, A_h)
    H = sharded.ShardHessian.from_global(sh, H_h, hdiag_h)
    Z, LS, Y = sharded.projections(A)
    c = sh.from_global(c_h, "col")
    b = sh.zeros("row")
    x0 = Y.dot(-b)
    r0 = Z.dot(H.dot(x0) + c)
    g0 = Z.dot(r0)
    rt_g = g0.sumsq_amax()[0]
    Fs = [sharded.FusedShardedCG(H, Z.projector, None, None, transport=t) for t in transports]
    return sh, (Fs[0] if len(Fs) == 1 else Fs), (x0, r0, g0, rt_g, 0.0, np.inf)


def _sharded_run(F, primed, K, W, dist, torch):
    """W warm-up + K timed iterations (restart from the primed state every SEG iterations:
    device copies and one SpMV, no collective); max over ranks of the elapsed time."""
    from ipsolver.sharded import ST_STOP, ST_IT_DONE
    SEG = 200

    def run(it0, it1):
        it = it0
        while it < it1:
            j = it % SEG
            if j == 0:
                F.prime(*primed)
            end = min(it1, it - j + SEG)
            F.iterate(j, j + (end - it))
            it = end

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    run(0, W)
    barrier()
    t0 = time.perf_counter()
    run(W, W + K)
    barrier()
    elapsed = time.perf_counter() - t0
    s = F.L.state.tolist()
    last_segment = (W + K - 1) % SEG + 1
    if int(s[ST_STOP]) != 0 or int(s[ST_IT_DONE]) != last_segment:
        raise RuntimeError("timed region did not run %d iterations: stop=%s done=%s (expected %d "
                           "in the last segment)" % (K, s[ST_STOP], s[ST_IT_DONE], last_segment))
    tt = torch.tensor([elapsed], dtype=torch.float64,
                      device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item()), last_segment


def sharded_leg(args, world, rank, A_h, H_h, hdiag_h, c_h, n, m, K, W):
    """N > 1: the SAME n=1e6 / m=1e5 subproblem row-partitioned over the ranks (BASELINE
    config 4, strong scaling; ipsolver/sharded.py): both spaces partitioned, per iteration
    two all-reduces (2 + 4 doubles) and one neighbour exchange of the halo of g -- through the
    peer mailboxes (hipIpc-mapped HBM written over xGMI inside the loop's own launches; one C
    call per batch) when the ranks can map each other, through torch.distributed (RCCL)
    otherwise.  `value` is the default transport's; both are measured and reported."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from ipsolver import sharded
    from ipsolver.synthetic import CenteredBandedNLP

    devices = [None] * world
    dist.all_gather_object(devices, int(torch.cuda.current_device()))
    sh, (F, F_dist), primed = _sharded_setup(A_h, H_h, hdiag_h, c_h, transports=(None, "dist"))
    transport = "ipc" if F.mailbox is not None else "dist"
    # ---- preflight, before any timing (first contact with a real node: VERDICT r4 item 7): one
    # all-reduce on the torch.distributed backend ("nccl" = RCCL), the hipIpc mapping of the
    # peers' buffers (done by the constructors above: distinct devices on a real node), one
    # tagged-word ping-pong per neighbour pair.  A failure of the backend itself is fatal (the
    # failure line, non-zero exit of this fresh child); a mailbox that cannot be mapped is a
    # reported fall-back to torch.distributed.
    backend = dist.get_backend()
    ones = torch.ones(1, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(ones)
    seen = int(round(float(ones.item())))
    if seen != world:
        raise RuntimeError("preflight: an all-reduce over %s saw %d of %d ranks" % (backend, seen, world))
    pingpong, pingpong_error = None, None
    if F.mailbox is not None:
        try:
            mine = F.mailbox.pingpong(200)
        except Exception as exc:
            mine, pingpong_error = None, repr(exc)
        allp = [None] * world
        dist.all_gather_object(allp, mine)
        if all(pp is not None for pp in allp):
            pingpong = {"%d-%d" % (r, q): us for r, pp in enumerate(allp) for q, us in pp.items() if r < q}
    preflight = {
        "torch_distributed_backend": backend,
        "rccl_ranks_seen": seen if backend == "nccl" else None,
        "ranks_seen": seen,
        "devices_by_rank": devices,
        "distinct_devices": len(set(devices)) == world and torch.cuda.device_count() >= world,
        "mailbox_mapped": F.mailbox is not None,
        "mailbox_error": getattr(sh, "mailbox_error", None),
        "mailbox_pingpong_us": pingpong,
        "mailbox_pingpong_error": pingpong_error,
        "mailbox_pingpong_note": "round trip of one tagged 16-byte word between neighbour ranks, "
                                 "200 round trips inside one kernel per pair (half of it = one "
                                 "cross-GPU hand-off of the resident loop kernel)",
        "resident_form": bool(F.resident),
        "resident_buffers_error": getattr(F.mailbox, "resident_error", None) if F.mailbox else None}
    before = dict(sh.comm.stats)
    ipc_error = None
    try:
        elapsed, last_segment = _sharded_run(F, primed, K, W, dist, torch)
    except (Exception, SystemExit) as exc:
        # the mailbox path has only ever run between processes that share one GPU (no multi-GPU
        # box in the build): if it fails here -- on every rank alike, the waits time out
        # together -- the measurement falls back to the torch.distributed transport and says so
        if transport != "ipc":
            raise
        ipc_error, transport, F = repr(exc), "dist", F_dist
        os.environ["IPX_SHARD_TRANSPORT"] = "dist"      # (the later legs of this run too)
        sh._mailbox = False
        before = dict(sh.comm.stats)
        elapsed, last_segment = _sharded_run(F, primed, K, W, dist, torch)
    calls = {k: v - before[k] for k, v in sh.comm.stats.items()}
    x_sharded = sharded.ShardVec(sharded.HipOps().dv.DVec(F.L.x.clone()), sh, "col").to_host()
    ab = {transport: {"iterations_per_s": K / elapsed, "ms_per_step": 1e3 * elapsed / K}}
    if ipc_error is not None:
        ab["ipc"] = {"error": ipc_error}
    resident = transport == "ipc" and bool(F.resident)
    if resident:                            # A/B: the same loop on the three launches per iteration
        ab["ipc"]["form"] = "one resident launch per rank and batch (csrc/resident.hip, PEER)"
        try:
            F.resident = False
            t3 = sorted(_sharded_run(F, primed, K, W, dist, torch)[0] for _ in range(3))[1]
            ab["ipc_three_launches"] = {"iterations_per_s": K / t3, "ms_per_step": 1e3 * t3 / K,
                                        "launches_per_iteration": 3 if F.ext.fuse_comm else 5}
        except Exception as exc:
            ab["ipc_three_launches"] = {"error": repr(exc)}
        finally:
            F.resident = True
    launches = 3 if (transport == "ipc" and (F.ext.fuse_comm or F.mailbox.fused_launches() > 0)) else 5
    if transport == "ipc":
        ab["ipc"]["launches_per_iteration"] = ("1 per batch" if resident else launches)
        if launches == 3 and not resident:      # A/B: the collectives in pack kernels of their own
            # (alternating, three regions each: ranks that share one GPU interleave differently
            # from region to region, a single pair of numbers says little)
            try:
                t3, t5 = [], []
                for _ in range(3):
                    F.ext.fuse_comm = 0
                    t5.append(_sharded_run(F, primed, K, W, dist, torch)[0])
                    F.ext.fuse_comm = 1
                    t3.append(_sharded_run(F, primed, K, W, dist, torch)[0])
                ab["ipc"]["ms_per_step_regions"] = [1e3 * t / K for t in [elapsed] + t3]
                ab["ipc_pack_kernels"] = {"iterations_per_s": K / sorted(t5)[1],
                                          "ms_per_step": 1e3 * sorted(t5)[1] / K,
                                          "ms_per_step_regions": [1e3 * t / K for t in t5],
                                          "launches_per_iteration": 5}
            except Exception as exc:
                ab["ipc_pack_kernels"] = {"error": repr(exc)}
            finally:
                F.ext.fuse_comm = 1
        try:                                    # A/B: the same loop through torch.distributed
            e2, _ = _sharded_run(F_dist, primed, K, W, dist, torch)
            ab["dist"] = {"iterations_per_s": K / e2, "ms_per_step": 1e3 * e2 / K}
        except Exception as exc:
            ab["dist"] = {"error": repr(exc)}

    # ---- untimed: the sharded iterate against the single-GPU fused loop (rank 0)
    parity = None
    if rank == 0:
        try:
            from ipsolver import device as dv, projector, qp
            from ipsolver.operators import DeviceHessian
            H1 = DeviceHessian(n, csr=dv.DeviceCSR.from_scipy(H_h), diag=dv.DVec.from_host(hdiag_h))
            Z1, _, Y1 = projector.projections(dv.DeviceCSR.from_scipy(A_h))
            x1, info1 = qp.projected_cg(H1, c_h, Z1, Y1, np.zeros(m), tol=0.0,
                                        max_iter=last_segment)
            x1 = x1.to_host()
            parity = {"iterations": int(info1["niter"]),
                      "max_rel_diff": float(np.max(np.abs(x_sharded - x1)) / np.max(np.abs(x1)))}
            del H1, Z1, Y1
        except Exception as exc:                   # never lose the measurement over the check
            parity = {"error": repr(exc)}

    # ---- latency floor of the two collectives the loop uses, measured here: back-to-back
    # all-reduces of the 4-double pack / exchanges of the halo of g, stream-synchronised once
    def floor(fn, reps=200):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t0) / reps
    clo, chi, sl, sr = F.col_geom
    scratch = torch.zeros(4, dtype=torch.float64, device="cuda")
    lat_allreduce = floor(lambda: sh.comm.all_reduce(scratch))
    halo_buf = F.L.r.clone()
    lat_exchange = floor(lambda: sh.comm.exchange(halo_buf, clo, chi, sl, sr))
    # the mailbox path's floor: 200 one-workgroup kernels back to back, each storing 4 doubles
    # into every peer's mailbox and spinning until the peers' have arrived (launch + flag wait)
    lat_mailbox = None
    if F.mailbox is not None:
        F.mailbox.allreduce([1.0, 2.0, 3.0, 4.0], reps=20)
        dist.barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        F.mailbox.allreduce([1.0, 2.0, 3.0, 4.0], reps=200)
        ev1.record()
        torch.cuda.synchronize()
        lat_mailbox = 1e3 * ev0.elapsed_time(ev1) / 200

    # ---- a weak-scaling point in the same run: n = world * 1e6 (per-GPU work fixed)
    weak = None
    shared_gpu = torch.cuda.device_count() < world
    if shared_gpu and not args.no_weak:
        # ranks that share one GPU cannot keep n = 1e6 each resident side by side: a rank's
        # workgroups would fill the chip and spin on words the other rank's kernels, waiting
        # behind them, can never write (the 3 s deadline, stop code 7).  Not a measurement
        # the rehearsal can make.
        weak = {"rehearsal": "skipped", "reason": "%d ranks on %d GPU(s): the weak-scaling point "
                "needs one GPU per rank" % (world, torch.cuda.device_count())}
    elif not args.no_weak:
        try:
            nw, mw = n * world, m * world
            probw = CenteredBandedNLP(nw, mw, seed=0)
            xw = probw.x0
            vw = 0.1 * np.random.default_rng(7).standard_normal(mw)
            shw, Fw, primedw = _sharded_setup(probw.constr_jac(xw), probw.hess(xw),
                                              probw.kappa * probw.Wt.dot(vw), probw.grad(xw))
            ew, _ = _sharded_run(Fw, primedw, K, W, dist, torch)
            weak = {"n": nw, "m": mw, "iterations_per_s": K / ew, "ms_per_step": 1e3 * ew / K,
                    "transport": "ipc" if Fw.mailbox is not None else "dist",
                    "note": "per-GPU work = the 1-GPU benchmark's (n=1e6 per rank); compare "
                            "with the n_gpus=1 value for weak-scaling efficiency"}
            del shw, Fw, primedw, probw
        except Exception as exc:
            weak = {"error": repr(exc)}

    # ---- ranks that share one GPU cannot keep n = 1e6 resident side by side (the group refuses
    # the resident form, preflight.resident_form false): the SAME code path at a size where every
    # rank's workgroups fit the chip together, against the three launches on the same size -- the
    # evidence this build has for the resident PEER kernel on hardware
    resident_rehearsal = None
    if shared_gpu and transport == "ipc" and not resident:
        try:
            nr, mr = 400000, 40000
            probr = CenteredBandedNLP(nr, mr, seed=0)
            xr = probr.x0
            vr = 0.1 * np.random.default_rng(7).standard_normal(mr)
            shr, Fr, primedr = _sharded_setup(probr.constr_jac(xr), probr.hess(xr),
                                              probr.kappa * probr.Wt.dot(vr), probr.grad(xr))
            resident_rehearsal = {"n": nr, "m": mr, "resident_form": bool(Fr.resident)}
            if Fr.resident:
                tr = sorted(_sharded_run(Fr, primedr, K, W, dist, torch)[0] for _ in range(3))[1]
                xres = Fr.L.x.clone()
                Fr.resident = False
                t3 = sorted(_sharded_run(Fr, primedr, K, W, dist, torch)[0] for _ in range(3))[1]
                Fr.resident = True
                lo_, hi_ = shr.lay.geom("col")[2:]
                diff = float((xres[lo_:hi_] - Fr.L.x[lo_:hi_]).abs().max().item())
                resident_rehearsal.update({
                    "resident_iterations_per_s": K / tr, "three_launches_iterations_per_s": K / t3,
                    "own_entries_max_abs_diff_between_the_forms": diff,
                    "note": "%d ranks on one GPU, every workgroup of both resident at once; the "
                            "hand-offs between the ranks go through the same hipIpc-mapped "
                            "buffers as between GPUs" % world})
            del shr, Fr, primedr, probr
        except Exception as exc:
            resident_rehearsal = {"error": repr(exc)}

    # ---- the whole config-4 solve to gtol on the sharded backend (second half of the metric)
    full_solve = None
    try:
        import warnings
        from ipsolver.synthetic import ShardedCallbacks
        cb = ShardedCallbacks(CenteredBandedNLP(n, m, eps=1e-3), sh)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for attempt in range(2):              # first call pays one-off symbolic set-up
                torch.cuda.synchronize()
                dist.barrier()
                t0 = time.time()
                res = sharded.minimize_equality_constrained(
                    sh, cb.fun, cb.grad, cb.lagr_hess, cb.constr_fun, cb.constr_jac, cb.x0,
                    method="tr_interior_point")
                torch.cuda.synchronize()
                wall = time.time() - t0
        full_solve = {"seconds": wall, "status": int(res.status), "niter": int(res.niter),
                      "cg_niter": int(res.cg_niter), "optimality": float(res.optimality),
                      "constr_violation": float(res.constr_violation),
                      "note": "config 4 = config 3 (eps=1e-3) sharded; the reference reaches "
                              "status 1 in 25 outer / 34 CG iterations"}
    except Exception as exc:
        full_solve = {"error": repr(exc)}

    nnzA, nnzH = A_h.nnz, H_h.nnz
    _, iter_bytes = loop_bytes(n, m, nnzA, nnzH)       # (the single-GPU loop's ledger)
    per_it = {k: v / max(1, (W + K)) for k, v in calls.items()}
    lo, hi = sh.lay.geom("col")[2:]
    result = {
        "metric": "projected-CG iters/sec (fp64) at n=1e6,m=1e5",
        "value": K / elapsed, "unit": "iterations/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "timed_region_s": elapsed,
        "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "config4: the config-3 subproblem row-partitioned over %d GPUs: "
                               "constraint rows AND variables partitioned (nothing replicated), "
                               "per iteration 2 all-reduces (2 and 4 doubles) + 1 neighbour "
                               "exchange of the halo of g; transport of the timed region: %s"
                               % (world, "peer buffers (hipIpc-mapped HBM, writes over xGMI): ONE "
                                         "resident launch per rank and batch, the workgroups of "
                                         "all ranks hand scalars and halos to each other"
                                  if resident else
                                  "peer mailboxes (hipIpc-mapped HBM, writes over xGMI "
                                  "inside the loop's own %d launches per iteration, one "
                                  "C call per batch)" % launches
                                  if transport == "ipc" else
                                  "torch.distributed (%s), three calls per iteration from the "
                                  "host%s" % (dist.get_backend(),
                                              "; mailboxes unavailable: %s"
                                              % getattr(sh, "mailbox_error", None)
                                              if os.environ.get("IPX_SHARD_TRANSPORT", "ipc") == "ipc"
                                              else "")),
                   "n": n, "m": m, "nnz_A": int(nnzA), "nnz_H": int(nnzH),
                   "parallelism": "rows and variables sharded x%d, halo = 1 block of %d rows"
                                  % (world, sh.lay.row_block)},
        "roofline": {"bound": "hbm", "kernel": "whole iteration (collective-latency bound at "
                                               "this size, see DESIGN.md section 5)",
                     "achieved": iter_bytes / (elapsed / K) / 1e9, "peak": HBM_PEAK_GBS * world,
                     "unit": "GB/s",
                     "frac": iter_bytes / (elapsed / K) / 1e9 / (HBM_PEAK_GBS * world),
                     "traffic": None},
        "parity_vs_single_gpu": parity,
        "backend": {"torch_distributed": dist.get_backend(), "world_size_seen": dist.get_world_size(),
                    "devices_by_rank": devices, "visible_gpus": torch.cuda.device_count()},
        "transport": transport,
        "preflight": preflight,
        "rccl_ranks_seen": preflight["rccl_ranks_seen"],
        "mailbox_pingpong_us": preflight["mailbox_pingpong_us"],
        "transport_fallback_reason": (ipc_error if ipc_error is not None
                                      else getattr(sh, "mailbox_error", None)
                                      if transport != "ipc" else None),
        "transport_ab": ab,
        "host_calls_per_iteration_in_the_timed_region": {
            "torch_distributed_all_reduce": per_it["all_reduce"],
            "torch_distributed_exchange": per_it["exchange"],
            "c_calls": (per_it["ipc_batches"] if transport == "ipc" else 2.0),
            "note": "ipc: one C call per restart segment of <= 200 iterations; "
                    + ("the whole segment is one resident launch per rank" if resident else
                       "the reductions and the halo exchange happen inside its %d launches per "
                       "iteration%s" % (launches, " (in the prologues of the kernels that consume "
                                                  "them)" if launches == 3 else ""))},
        "exchanged_per_iteration": {
            "all_reduce_doubles": [2, 4],
            "halo_columns_rank0": [int(lo), int(sh.lay.geom("col")[1] - hi)]},
        "collective_latency_floor_us": {"mailbox_all_reduce_4_doubles_launch_plus_flag_wait":
                                        lat_mailbox,
                                        "torch_distributed_all_reduce_4_doubles": lat_allreduce,
                                        "torch_distributed_halo_exchange": lat_exchange,
                                        "method": "200 back-to-back operations (HIP events for the "
                                                  "mailbox kernels; host clock + one stream "
                                                  "synchronise for torch.distributed over %s)"
                                                  % dist.get_backend()},
        "weak_scaling_point": weak,
        "resident_rehearsal": resident_rehearsal,
        "wall_clock_to_gtol": full_solve,
    }
    if rank == 0:
        print(json.dumps(result))
    dist.destroy_process_group()


def failure_line(n_gpus, K, W, error, **more):
    """The contract's keys with value 0 and a readable ``error``: a failed N > 1 run must still
    leave one JSON line behind (and a non-zero exit code)."""
    d = {"metric": "projected-CG iters/sec (fp64) at n=1e6,m=1e5", "value": 0.0,
         "unit": "iterations/s", "n_gpus": n_gpus, "steps": K, "warmup": W,
         "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
         "vs_baseline": None, "dtype": "f64", "data": "synthetic",
         "config": {"workload": "config4 sharded over %d GPUs: FAILED" % n_gpus},
         "error": error}
    d.update(more)
    return d


def spawn_ranks(n_gpus, argv, K, W):
    """``python bench.py --gpus N`` without a launcher: start the N ranks as FRESH child
    processes of this parent (one per GPU: RANK = LOCAL_RANK = device, WORLD_SIZE, MASTER_ADDR =
    127.0.0.1 and a free MASTER_PORT in their environment -- what torch.distributed.run would
    set), which itself never touches the GPU and is not replaced by anything (no exec); relay
    rank 0's JSON line.  Any rank failing = a non-zero exit code here and one ``error`` line;
    the other ranks get 60 s to notice (a collective that lost its peer) before they are
    terminated by PID."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = dict(os.environ, IPX_BENCH_SPAWNED="1", WORLD_SIZE=str(n_gpus),
                LOCAL_WORLD_SIZE=str(n_gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # (what torch.distributed.run sets too: with one thread pool of all cores per rank the tiny
    # host-side tensor ops of the outer loop's collectives take milliseconds each -- measured:
    # the sharded config-4 solve 3.1 s instead of 0.13 s)
    base.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs, errs = [], []
    for r in range(n_gpus):
        err = tempfile.TemporaryFile(mode="w+")
        errs.append(err)
        procs.append(subprocess.Popen(
            cmd, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=err, text=True,
            env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), cwd=ROOT))
    out = ""
    first_failure = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if first_failure is None and any(c not in (None, 0) for c in codes):
            first_failure = time.time()
        if first_failure is not None and time.time() - first_failure > 60:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            time.sleep(5)
            for p in procs:
                if p.poll() is None:
                    p.kill()
        if procs[0].poll() is None:
            try:                                   # (drain rank 0's pipe while waiting)
                o, _ = procs[0].communicate(timeout=1.0)
                out += o or ""
            except subprocess.TimeoutExpired:
                pass
        else:
            time.sleep(0.5)
    if procs[0].stdout is not None and not procs[0].stdout.closed:
        out += procs[0].stdout.read() or ""
    codes = [p.returncode for p in procs]
    tails = []
    for r, err in enumerate(errs):
        err.seek(0)
        t = err.read()[-1500:]
        err.close()
        if codes[r] != 0 and t:
            tails.append("[rank %d, exit code %s] %s" % (r, codes[r], t))
    err_tail = "\n".join(tails)[-4000:]
    rc = next((c for c in codes if c != 0), 0)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    for ln in out.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        try:
            d = json.loads(lines[-1])
        except ValueError:
            d = None
        if d is not None:
            d["launcher"] = ("bench.py spawned %d ranks itself (fresh child processes, RANK / "
                             "LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment; the parent "
                             "made no GPU call)" % n_gpus)
            if rc != 0 and "error" not in d:
                d["error"] = "rank exit codes %s after the line was printed" % codes
            if rc != 0:
                d["stderr_tail"] = err_tail
            print(json.dumps(d))
            sys.stdout.flush()
            if rc != 0 or "error" in d:
                sys.stderr.write(err_tail)
                raise SystemExit(rc or 1)
            return
    print(json.dumps(failure_line(n_gpus, K, W, "the ranks exited with codes %s without a JSON "
                                  "line" % codes, stderr_tail=err_tail)))
    sys.stdout.flush()
    sys.stderr.write(err_tail)
    raise SystemExit(rc or 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--m", type=int, default=100000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-iters", type=int, default=250)
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the config-5 and config-2 full solves")
    ap.add_argument("--repeats", type=int, default=20,
                    help="further K-step regions after the contract's one (median/min/max)")
    ap.add_argument("--no-weak", action="store_true",
                    help="N > 1: skip the weak-scaling point (n = N * 1e6)")
    ap.add_argument("--no-big", action="store_true",
                    help="skip the out-of-Infinity-Cache measurement (n=1.6e7, m=1.6e6)")
    ap.add_argument("--no-sweep", action="store_true",
                    help="skip the per-rank size sweep (n = 1e6 / N on this GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started directly (``python bench.py --gpus N``, the driver's N = 1 form): this process
        # becomes the launcher.  It has made no GPU call and imports nothing that could.
        return spawn_ranks(args.gpus, sys.argv[1:], args.steps, args.warmup)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        msg = ("--gpus %d but the launcher started %d ranks (WORLD_SIZE=%d)"
               % (args.gpus, world, world))
        if rank == 0:
            print(json.dumps(failure_line(args.gpus, args.steps, args.warmup, msg)))
            sys.stdout.flush()
        raise SystemExit(msg)
    # IPX_BENCH_BACKEND=gloo lets the N > 1 leg be exercised on a one-GPU box (all ranks share
    # cuda:0, collectives staged through the host); the measured configuration is nccl = RCCL.
    backend = os.environ.get("IPX_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    from ipsolver import _hip, cg_fused, projector
    from ipsolver import device as dv
    from ipsolver.operators import DeviceHessian
    from ipsolver.synthetic import CenteredBandedNLP

    n, m = args.n, args.m
    K, W = args.steps, args.warmup
    lib = _hip.load()

    # ---- synthetic instance (same seed on every rank: one shared problem)
    t0 = time.time()
    prob = CenteredBandedNLP(n, m, seed=0)
    x = prob.x0
    v = 0.1 * np.random.default_rng(7).standard_normal(m)
    A_h = prob.constr_jac(x)
    H_h = prob.hess(x)
    hdiag_h = prob.kappa * prob.Wt.dot(v)
    c_h = prob.grad(x)
    t_gen = time.time() - t0

    if world > 1:
        try:
            return sharded_leg(args, world, rank, A_h, H_h, hdiag_h, c_h, n, m, K, W)
        except Exception as exc:         # a failure must be readable in the bench line
            import traceback
            if rank == 0:
                print(json.dumps(failure_line(world, K, W, repr(exc),
                                              traceback=traceback.format_exc()[-1500:])))
                sys.stdout.flush()
            raise

    r1 = single_gpu_measure(A_h, H_h, hdiag_h, c_h, n, m, K, W, repeats=args.repeats)
    A, H, c, b, Z, Y = (r1.pop(k) for k in ("A", "H", "c", "b", "Z", "Y"))
    traffic, traffic_src = None, None
    if (n, m) == (1000000, 100000):
        traffic, traffic_src = stored_traffic(PMC_PROFILE, r1["dom"])
    # ---- the headline: the PRODUCT FUNCTION as a user calls it -- ONE call of
    # ipsolver.qp.projected_cg(H, c, Z, Y, b, trust_radius, tol=0, max_iter=K), W warm-up
    # iterations in a call before it, the timed region = the call (synchronised either side):
    # its priming (x0 = Y(-b), r0 = Z(H x0 + c), g0 = Z r0: qp_subproblem.py:502-512), buffer
    # set-up, the batched device loop with its state reads, the result.  The bare device loop
    # (one C call enqueuing K iterations; the former headline) is `device_loop_only`.
    from ipsolver import qp as _qp
    b_dev = b
    if W > 0:
        _qp.projected_cg(H, c, Z, Y, b, trust_radius=1e300, tol=0, max_iter=W)
    # (with tol = 0 the CG reaches an exactly zero residual after ~500 iterations: a K beyond
    # 200 is run as consecutive calls of <= 200 iterations on the same subproblem)
    chunks = [200] * (K // 200) + ([K % 200] if K % 200 else [])
    # FIVE timed regions of exactly K steps each (a region at the driver's K = 20 lasts 1 ms:
    # one sample says little -- VERDICT r4); `value` is their median, every one is reported
    regions = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done = 0
        for kc in chunks:
            _xk, _info = _qp.projected_cg(H, c, Z, Y, b, trust_radius=1e300, tol=0, max_iter=kc)
            done += _info["niter"]
        torch.cuda.synchronize()
        regions.append(time.perf_counter() - t0)
        if done != K:
            raise SystemExit("headline call(s) ran %d iterations instead of %d" % (done, K))
    elapsed = sorted(regions)[len(regions) // 2]
    result = {
        "metric": "projected-CG iters/sec (fp64) at n=1e6,m=1e5",
        "value": K / elapsed,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": 1e3 * elapsed / K,
        "timed_region_s": elapsed,
        "timed_regions": {"what": "value = the median of these regions of exactly `steps` "
                                  "iterations each (barrier + synchronise either side of each)",
                          "iterations_per_s": [K / t for t in regions]},
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "config3: sparse banded NLP subproblem, CSR Jacobian "
                               "bandwidth 15: ONE call ipsolver.qp.projected_cg(H, c, Z, Y, b, "
                               "trust_radius=1e300, tol=0, max_iter=steps), b a device vector "
                               "(zeros; x0 = Y(-b) is computed: whether a device vector is zero "
                               "is not known without reading it) -- priming included; "
                               "finite trust radius that is never reached (the norm test of "
                               "qp_subproblem.py:583 formed every iteration, as in the SQP's "
                               "calls; trust_radius=inf: see unbounded_trust_region)",
                   "n": n, "m": m, "nnz_A": r1["nnzA"], "nnz_H": r1["nnzH"],
                   "parallelism": "single GPU"},
        "device_loop_only": {
            "what": "the bare device loop: ipx_cg_iterate enqueues exactly `steps` iterations of "
                    "the primed subproblem in one C call (rounds 1-3's headline); no priming, no "
                    "state reads inside the region",
            "iterations_per_s": K / r1["elapsed"], "ms_per_step": 1e3 * r1["elapsed"] / K,
            "repeat": r1["repeat"]},
        "unbounded_trust_region": r1["unbounded"],
        "roofline": dict(r1["roofline"], traffic=traffic, traffic_source=traffic_src,
                         frac_on_moved_bytes=None if not traffic else
                         traffic / (r1["roofline"]["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS),
        "loop_kernels": traffic_table(PMC_PROFILE, r1["whole_iteration"]["algorithmic_bytes_per_kernel"])
        if (n, m) == (1000000, 100000) else None,
        "whole_iteration": r1["whole_iteration"],
        "kernel_source_hash": kernel_source_hash(),
        "setup_s": {"generate_host": t_gen, "factor_device": r1["t_factor"]},
    }
    # ---- the product function as a user calls it
    result["public_api"] = public_api_leg(H, c, Z, Y, b, K)
    # (b = None -- the extension the SQP's own call uses, equality_constrained_sqp.py:126 passes
    # zeros -- starts from x0 = 0 without the solve and the two products of Y(-b), H x0 + c)
    result["public_api"]["with_b_none"] = {
        k: v["iterations_per_s"] for k, v in public_api_leg(H, c, Z, Y, None, K, reps=3).items()
        if isinstance(v, dict)}
    # ---- the strong-scaling ceiling at the per-rank sizes, measured in this run
    if (n, m) == (1000000, 100000) and not args.no_sweep:
        try:
            result["per_rank_sweep"] = per_rank_sweep_leg()
        except Exception as exc:                    # never lose the line over a side measurement
            result["per_rank_sweep"] = {"error": repr(exc)}
    # ---- the same-run streaming reference, and the dominant kernel against it
    try:
        stream = measured_stream()
        result["measured_stream"] = stream
        result["roofline"]["frac_of_measured_stream"] = \
            result["roofline"]["achieved"] / stream["axpby_2r1w"]
    except Exception as exc:                        # never lose the line over the side measurement
        result["measured_stream"] = {"error": repr(exc)}

    # ---- the same measurement well past the 256 MiB Infinity Cache: n = 1.6e7 (matrices and
    # vectors of the loop ~2.3 GB, nine times the cache).  The n=1e6 working set (~145 MB) is
    # cache resident, so its "HBM" fraction is partly an Infinity-Cache fraction; rounds 2-4
    # measured n = 4e6 (~580 MB) here, which is still ~45 % resident and ran ABOVE the same
    # run's streaming kernels -- not an out-of-cache point.
    if (n, m) == (1000000, 100000) and not args.no_big:
        nb, mb = 16000000, 1600000
        probb = CenteredBandedNLP(nb, mb, seed=0)
        xb = probb.x0
        vb = 0.1 * np.random.default_rng(7).standard_normal(mb)
        rb = single_gpu_measure(probb.constr_jac(xb), probb.hess(xb), probb.kappa * probb.Wt.dot(vb),
                                probb.grad(xb), nb, mb, K, W, repeats=min(args.repeats, 5))
        traffic_b, src_b = stored_traffic(PMC_PROFILE_BIG, rb["dom"])
        result["roofline_out_of_cache"] = dict(
            rb["roofline"], traffic=traffic_b, traffic_source=src_b, n=nb, m=mb,
            working_set_bytes=rb["working_set_bytes"],
            iterations_per_s=K / rb["elapsed"], ms_per_step=1e3 * rb["elapsed"] / K,
            repeat=rb["repeat"], unbounded_trust_region=rb["unbounded"],
            whole_iteration=rb["whole_iteration"])
        if isinstance(result.get("measured_stream"), dict) and "axpby_2r1w" in result["measured_stream"]:
            result["roofline_out_of_cache"]["frac_of_measured_stream"] = \
                result["roofline_out_of_cache"]["achieved"] / result["measured_stream"]["axpby_2r1w"]
        del rb, probb
        torch.cuda.empty_cache()

    # ---- second half of the metric: wall-clock of the whole config-3 solve to
    # gtol = 1e-8 (minimize_constrained, tr_interior_point, device-callback mode:
    # objective / constraint callbacks evaluate on the GPU, nothing crosses PCIe)
    if (n, m) == (1000000, 100000) or os.environ.get("IPX_BENCH_FULL_SOLVE"):
        result["wall_clock_to_gtol"] = config3_leg(n, m)

    # ---- the other single-GPU BASELINE configs to gtol (full sizes)
    if (n, m) == (1000000, 100000) and not args.no_configs:
        for key, leg in (("config5", config5_leg), ("config2", config2_leg)):
            try:
                torch.cuda.empty_cache()
                result[key] = leg()
            except Exception as exc:             # never lose the headline over a side leg
                result[key] = {"error": repr(exc)}

    # ---- CPU baseline: the oracle (numpy/scipy restatement), rank 0, N=1 only: the better of
    # one thread and all host cores (the path is SuperLU substitutions + short
    # BLAS-1 calls, effectively single-threaded)
    if rank == 0 and world == 1 and not args.no_cpu:
        import oracle
        from threadpoolctl import threadpool_limits, threadpool_info
        ncores = os.cpu_count() or 1
        blas = sorted({"%s %s" % (d.get("internal_api"), d.get("version")) for d in threadpool_info()
                       if d.get("user_api") == "blas"})
        t0 = time.time()
        Zo, _, Yo = oracle.projections(A_h)        # AugmentedSystem (SuperLU), as the reference
        t_fac = time.time() - t0
        H_full = H_h + __import__("scipy.sparse", fromlist=["diags"]).diags(hdiag_h)
        kc = args.cpu_iters
        runs = {}
        for threads in sorted({1, ncores}):
            torch.set_num_threads(threads)
            with threadpool_limits(limits=threads):
                t0 = time.time()
                xo, info = oracle.projected_cg(H_full, c_h, Zo, Yo, np.zeros(m), tol=0, max_iter=kc)
                runs[threads] = info["niter"] / (time.time() - t0)
        best = max(runs, key=runs.get)
        result["cpu_baseline"] = {
            "value": runs[best], "unit": "iterations/s", "cores": best, "kind": "port",
            "host_cpu_count": ncores, "blas": blas,
            "iterations_per_s_by_threads": {str(k): v for k, v in runs.items()},
            "sample": "%d projected-CG iterations of the same n=%d, m=%d instance through "
                      "oracle/ (scipy SuperLU augmented-system projections, numpy vectors) per "
                      "thread count; factorization %.1f s not included" % (kc, n, m, t_fac)}
        # parity of the GPU iterates with the CPU oracle after the same number of iterations
        xg, ginfo = cg_fused.projected_cg(H, c, Z, Y, b_dev, tol=0, max_iter=kc)
        err = np.max(np.abs(xg.to_host() - xo)) / np.max(np.abs(xo))
        result["parity_vs_oracle"] = {"iterations": kc, "max_rel_err": float(err)}

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
