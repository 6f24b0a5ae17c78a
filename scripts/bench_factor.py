"""Banded refactorization (A A' band, factor, checks, status read) back to back at the
benchmark's size (dev tool):  python scripts/bench_factor.py [n m] -- IPX_LIB_DIR=<dir> takes
another build of the library (A/B)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import torch
from ipsolver import _hip
if os.environ.get("IPX_LIB_DIR"):
    _hip.LIB_PATH = os.path.join(os.environ["IPX_LIB_DIR"], "libipx.so")
from ipsolver import device as dv, projector
from ipsolver.synthetic import CenteredBandedNLP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = int(sys.argv[2]) if len(sys.argv) > 2 else n // 10
prob = CenteredBandedNLP(n, m)
A = dv.DeviceCSR.from_scipy(prob.constr_jac(prob.x0))
for _ in range(5):
    S = projector.BandedNormalSolver(A); del S
torch.cuda.synchronize()
ts = []
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(100):
        S = projector.BandedNormalSolver(A); del S
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 100 * 1e6)
print("lib %s: refactorization %.1f us (min of 5 x 100: %.1f)" % (_hip.LIB_PATH if hasattr(_hip, "LIB_PATH") else "-", sorted(ts)[2], min(ts)))
