"""Kernel timeline of ONE public projected_cg call (n = 1e6, max_iter = K): run under
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scripts/trace_public_call.py run [K]
then  python3 scripts/trace_public_call.py show DIR  prints the last call's launches with start
offsets, durations and the idle gaps between them."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "run":
    sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd")); sys.path.insert(0, ROOT)
    import numpy as np, torch
    from ipsolver import device as dv, projector, qp
    from ipsolver.operators import DeviceHessian
    from ipsolver.synthetic import CenteredBandedNLP
    n, K = 1000000, int(sys.argv[2]) if len(sys.argv) > 2 else 20
    m = n // 10
    prob = CenteredBandedNLP(n, m, seed=0)
    x = prob.x0
    v = 0.1 * np.random.default_rng(7).standard_normal(m)
    A = dv.DeviceCSR.from_scipy(prob.constr_jac(x))
    H = DeviceHessian(n, csr=dv.DeviceCSR.from_scipy(prob.hess(x)), diag=dv.DVec.from_host(prob.kappa * prob.Wt.dot(v)))
    c = dv.DVec.from_host(prob.grad(x)); b = dv.DVec.zeros(m)
    Z, LS, Y = projector.projections(A)
    for _ in range(6):
        qp.projected_cg(H, c, Z, Y, b, trust_radius=1e300, tol=0, max_iter=K)
        torch.cuda.synchronize()
    # marker: a recognisable kernel before the call that is shown
    torch.zeros(12345, device="cuda").cos_()
    torch.cuda.synchronize()
    qp.projected_cg(H, c, Z, Y, b, trust_radius=1e300, tol=0, max_iter=K)
    torch.cuda.synchronize()
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    k = max(i for i, r in enumerate(rows) if "cos" in r["Kernel_Name"])
    rows = rows[k + 1:]
    t0 = int(rows[0]["Start_Timestamp"]); prev = t0
    busy = 0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:58]
        print("%8.1f  +%5.1f gap  %6.1f us  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, name))
        prev = e; busy += e - s
    print("launches %d, busy %.1f us, span %.1f us" % (len(rows), busy / 1e3, (prev - t0) / 1e3))
