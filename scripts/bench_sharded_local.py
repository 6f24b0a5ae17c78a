"""Sharded CG driver on ONE rank (no collectives): host + launch overhead of the segmented loop
next to the fused single-GPU loop (dev tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, torch
from ipsolver.sharded import ShardedProjectedCG, HipEngine, SegmentsByKernel
from ipsolver.synthetic import CenteredBandedNLP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = n // 10
prob = CenteredBandedNLP(n, m, seed=0)
x = prob.x0
v = 0.1 * np.random.default_rng(7).standard_normal(m)
A, H, hd, c = prob.constr_jac(x), prob.hess(x), prob.kappa * prob.Wt.dot(v), prob.grad(x)
for label, seg in (("one C call per segment", None), ("kernel by kernel from Python", SegmentsByKernel.segment)):
    eng = HipEngine()
    if seg is not None:
        eng.segment = lambda cg, ph, it, _e=eng: seg(_e, cg, ph, it)
    cg = ShardedProjectedCG(eng, A, H, hd)
    cg.prime(c, 0.0, np.inf)
    cg.iterate(0, 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cg.iterate(20, 220)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    s = cg.read_state()
    print("%-30s %7.1f us/iter (host enqueue %6.1f us/iter)  done=%d stop=%d" % (label, t / 200 * 1e6, t_host / 200 * 1e6, s[13], s[5]))
