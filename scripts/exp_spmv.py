"""Experiment: H.p SpMV time with the gathers / the row-sum phase removed (results are wrong on
purpose; dev tool).  IPX_LIB_DIR selects the library build, e.g.

    cd ip-nonlinear-solver_amd/csrc && mkdir -p ../lib_exp_NOGATHER && \
      hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DIPX_EXP_NOGATHER \
            -I../../include -shared -o ../lib_exp_NOGATHER/libipx.so *.hip
    IPX_LIB_DIR=lib_exp_NOGATHER python scripts/exp_spmv.py
"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, torch
from ipsolver import _hip
_hip.LIB_PATH = os.path.join(ROOT, "ip-nonlinear-solver_amd", os.environ.get("IPX_LIB_DIR", "lib"), "libipx.so")
from ipsolver import device as dv
from ipsolver.synthetic import CenteredBandedNLP
n = 1000000; m = n // 10
prob = CenteredBandedNLP(n, m)
H = dv.DeviceCSR.from_scipy(prob.hess(prob.x0)); A = dv.DeviceCSR.from_scipy(prob.constr_jac(prob.x0)); At = A.T
d = dv.DVec.from_host(np.ones(n)); x = dv.DVec.from_host(np.random.default_rng(0).standard_normal(n))
w = dv.DVec.from_host(np.random.default_rng(1).standard_normal(m))
y = dv.DVec.zeros(n); wm = dv.DVec.zeros(m)
def timeit(name, fn, N=300):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-10s %-26s %7.2f us" % (os.environ.get("IPX_LIB_DIR", "lib"), name, e0.elapsed_time(e1) / N * 1e3))
timeit("H.p (diag, reduce)", lambda: H.spmv(x, diag=d, out=y, reduce=True))
timeit("A.r", lambda: A.spmv(x, out=wm))
timeit("r - A'v", lambda: At.spmv(w, alpha=-1.0, beta=1.0, yin=x, out=y, reduce=True))
