"""Host timeline of the START of one warm config-3 solve (dev tool): when each set-up step of
``minimize_constrained`` begins and ends relative to the call, up to the first proposed step."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import torch
import ipsolver
from ipsolver import device_mode as dm, sqp, barrier, minimize, projector
from ipsolver.synthetic import CenteredBandedNLP, LeanDeviceCallbacks

warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dc = LeanDeviceCallbacks(CenteredBandedNLP(n, n // 10, eps=1e-3))
T0, LOG, ON = [0.0], [], [False]


def mark(name):
    if ON[0]:
        LOG.append((1e6 * (time.perf_counter() - T0[0]), name))


def wrap(obj, attr, label=None, limit=3):
    f = getattr(obj, attr)
    label = label or attr
    seen = [0]

    def g(*a, **k):
        seen[0] += 1
        if seen[0] <= limit or not ON[0]:
            mark(label + " >")
        r = f(*a, **k)
        if seen[0] <= limit or not ON[0]:
            mark(label + " <")
        return r
    g.reset = lambda: seen.__setitem__(0, 0)
    setattr(obj, attr, g)
    return g


W = [wrap(dm.DeviceCanonical, "__init__", "DeviceCanonical"),
     wrap(dm._DeviceConstraint, "__init__", "_DeviceConstraint"),
     wrap(dm.DeviceRowMap, "__init__", "DeviceRowMap"),
     wrap(dm, "lagrangian_hessian"),
     wrap(minimize, "tr_interior_point"),
     wrap(barrier, "equality_constrained_sqp") if hasattr(barrier, "equality_constrained_sqp") else None,
     wrap(sqp.ChainStages, "settle", limit=2), wrap(sqp.ChainStages, "propose", limit=2),
     wrap(projector, "projections", limit=2)]
from ipsolver import sqp_chain, cg_fused
W += [wrap(sqp_chain.StepChain, "front", "chain.front", limit=2),
      wrap(sqp_chain.StepChain, "bind", "chain.bind", limit=2),
      wrap(sqp_chain.StepChain, "judge", "chain.judge", limit=2),
      wrap(sqp_chain.StepChain, "refresh", "chain.refresh", limit=2),
      wrap(sqp_chain.StepChain, "bind_refresh", "chain.bind_refresh", limit=2),
      wrap(cg_fused, "_loop_for", limit=2), wrap(cg_fused, "_release", limit=2),
      wrap(sqp.ChainStages, "judge", "stage.judge", limit=2)]
for name in ("fun", "grad", "hess", "constr_fun", "constr_jac", "constr_hess"):
    W.append(wrap(dc, name, "user." + name, limit=2))


def solve():
    for w in W:
        if w is not None:
            w.reset()
    del LOG[:]
    torch.cuda.synchronize()
    T0[0] = time.perf_counter()
    res = ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, dc.constraints(ipsolver),
                                        method="tr_interior_point")
    torch.cuda.synchronize()
    return res, time.perf_counter() - T0[0]


for _ in range(3):
    solve()
ON[0] = True
res, dt = solve()
print("solve %.2f ms, %d outer / %d CG" % (1e3 * dt, res.niter, res.cg_niter))
for t, name in LOG:
    print("%9.1f us  %s" % (t, name))
print({k: v for k, v in sqp_chain.STATS.items() if v})
