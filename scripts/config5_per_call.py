"""Per projected_cg call of a config-5 solve: iterations, GPU time (HIP events around the call,
host-driven stages so that every call goes through the backend's entry), us per iteration,
and what the subproblem looked like (barrier parameter, trust radius, active slack bounds)
-- does the loop's time per iteration depend on WHEN in the solve it runs or on WHAT it runs on?
    IPX_DEBUG_FORMS=no-step-chain python scripts/config5_per_call.py [out.json]"""
import json, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
os.environ.setdefault("IPX_DEBUG_FORMS", "no-step-chain")
import torch
import ipsolver
from ipsolver import backend_hip
from ipsolver.synthetic import CenteredBandedNLP, DeviceCallbacks
warnings.simplefilter("ignore")
n = 500000
prob = CenteredBandedNLP(n, n // 10, eps=1.0)
dc = DeviceCallbacks(prob)
cons = (dc.constraints(ipsolver, ("less", 0.0)), ipsolver.BoxConstraint(("interval", -0.8, 0.8)))
calls = []
plain = backend_hip.projected_cg


def timed(H, c, Z, Y, b, radius, lb, ub):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    x, info = plain(H, c, Z, Y, b, radius, lb, ub)
    e1.record()
    near = int((lb.t[-(len(c) - n):] > -0.995 * 0 - 1e300).sum().item()) if False else None
    calls.append((e0, e1, info["niter"], info["stop_cond"], float(radius), time.perf_counter() - t0))
    return x, info


backend_hip.projected_cg = timed
ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, cons, max_iter=3)
calls.clear()
res = ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, cons)
torch.cuda.synchronize()
rows = []
for k, (e0, e1, niter, stop, radius, host_s) in enumerate(calls):
    ms = e0.elapsed_time(e1)
    rows.append({"call": k, "niter": niter, "stop_cond": stop, "radius": radius, "gpu_ms": ms,
                 "host_ms": 1e3 * host_s, "us_per_iteration": 1e3 * ms / max(niter, 1)})
print("status %d, %d outer / %d CG, %d calls" % (res.status, res.niter, res.cg_niter, len(rows)))
for r in rows:
    print("%3d  niter %6d  stop %d  radius %.3g  gpu %8.2f ms  %.1f us/it" %
          (r["call"], r["niter"], r["stop_cond"], r["radius"], r["gpu_ms"], r["us_per_iteration"]))
if len(sys.argv) > 1:
    json.dump({"status": int(res.status), "niter": int(res.niter), "cg_niter": int(res.cg_niter),
               "calls": rows}, open(sys.argv[1], "w"), indent=1)
