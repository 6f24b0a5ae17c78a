"""Micro-benchmark of the banded (AA')^-1 solve alone (dev tool)."""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, torch
from ipsolver import device as dv, projector, _hip
from ipsolver.synthetic import CenteredBandedNLP

n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, int(sys.argv[2]) if len(sys.argv) > 2 else 100000
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 64
prob = CenteredBandedNLP(n, m)
A = dv.DeviceCSR.from_scipy(prob.constr_jac(prob.x0))
solver = projector.BandedNormalSolver(A, chunk=chunk)
w = dv.DVec.from_host(np.random.default_rng(0).standard_normal(m))
out = torch.empty(m, dtype=torch.float64, device="cuda")
lib = _hip.load()
st = dv.stream_ptr()
for name in ("ipx_banded_solve", "ipx_banded_solve_multilaunch"):
    fn = getattr(lib, name)
    for _ in range(20):
        fn(ctypes.c_void_p(solver.handle), dv._p(w.t), dv._p(out), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 200
    for _ in range(N):
        fn(ctypes.c_void_p(solver.handle), dv._p(w.t), dv._p(out), st)
    torch.cuda.synchronize()
    print(name, "decoupled", lib.ipx_banded_decoupled(ctypes.c_void_p(solver.handle)), "chunk", chunk, "levels", lib.ipx_banded_levels(ctypes.c_void_p(solver.handle)),
          "us/solve %.2f" % ((time.perf_counter() - t0) / N * 1e6))
