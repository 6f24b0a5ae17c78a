"""Which library entry points a warm config-3 solve calls, and how often (dev tool): the host
side of a solve is ~35 calls per outer iteration at ~7 us each."""
import collections, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import torch
import ipsolver
from ipsolver import _hip
from ipsolver.synthetic import CenteredBandedNLP, DeviceCallbacks, LeanDeviceCallbacks
if os.environ.get("LEAN", "1") == "1":        # (LEAN=0: the plain torch callbacks)
    DeviceCallbacks = LeanDeviceCallbacks
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
prob = CenteredBandedNLP(n, n // 10, eps=1e-3)
dc = DeviceCallbacks(prob)
warnings.simplefilter("ignore")
def solve():
    return ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, dc.constraints(ipsolver),
                                         method="tr_interior_point")
for _ in range(2):
    solve()
counts = collections.Counter()
orig = _hip.call
def counting(name, *a):
    counts[name] += 1
    return orig(name, *a)
_hip.call = counting
for mod in list(sys.modules.values()):
    if mod is not None and getattr(mod, "__name__", "").startswith("ipsolver") and hasattr(mod, "_hip"):
        pass
res = solve()
torch.cuda.synchronize()
print("status %d, %d outer / %d CG; %d calls through _hip.call" % (res.status, res.niter, res.cg_niter, sum(counts.values())))
for name, c in counts.most_common():
    print("  %5d  %s" % (c, name))
reads = counts["ipx_read_doubles"] + counts["ipx_read_folded"]
print("blocking reads through the library: %d" % reads)
from ipsolver import sqp_chain, cg_fused
print("chain:", dict(sqp_chain.STATS))
print("cg:", {k: v for k, v in cg_fused.STATS.items() if v})
from ipsolver import projector
print("banded handles:", projector.HANDLE_STATS)
