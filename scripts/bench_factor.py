import os, sys, time
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ipsolver import device as dv, projector
from ipsolver.synthetic import CenteredBandedNLP
prob = CenteredBandedNLP(1000000, 100000)
A = dv.DeviceCSR.from_scipy(prob.constr_jac(prob.x0))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        s = projector.BandedNormalSolver(A)
    torch.cuda.synchronize(); print("factorization (pooled handle): %.1f us" % (1e6 * (time.perf_counter() - t0) / 50))
