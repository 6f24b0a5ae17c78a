"""Dense-Jacobian factorization kernels at BASELINE config 2's size (A: 2000 x 10000):
Gram (fp64 MFMA), Cholesky, inverse, gemv -- HIP-event times and rates.
    python scripts/bench_dense.py [m] [n]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, torch
from ipsolver import _hip, device as dv
from ipsolver.dense import DeviceDense
m = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
lib = _hip.load()
rng = np.random.default_rng(0)
A_h = rng.standard_normal((m, n))
A = DeviceDense.from_host(A_h)
M = int(lib.ipx_dense_padded(m))
G = torch.empty((M, M), dtype=torch.float64, device="cuda")
X = torch.empty((M, M), dtype=torch.float64, device="cuda")
flag = torch.zeros(1, dtype=torch.int32, device="cuda")
work = torch.zeros(M + 1, dtype=torch.float64, device="cuda")
st = dv.stream_ptr()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {"m": m, "n": n, "M": M}
nt = (M + 63) // 64
flop = nt * (nt + 1) // 2 * 2.0 * 64 * 64 * n          # executed: tiles on / below the diagonal
auto = int(lib.ipx_gram_splits(m, n))
out["gram_splits_auto"] = auto
out["gram_ms_by_splits"] = {}
for S in sorted({1, 2, 3, 4, 5, 6, 8, auto}):
    ws = torch.empty(max(int(lib.ipx_gram_ws_doubles(m, S)), 1), dtype=torch.float64, device="cuda")
    t = timed(lambda: _hip.call("ipx_gram_f64_mfma_split", m, n, dv._p(A.t), n, dv._p(G),
                                dv._p(ws), S, st))
    out["gram_ms_by_splits"][S] = t
    if S == auto:
        out["gram_ms"] = t
        out["gram_TFs_executed"] = flop / (t * 1e-3) / 1e12
        out["gram_frac_of_78.6TF"] = out["gram_TFs_executed"] / 78.6
ws = torch.empty(max(int(lib.ipx_gram_ws_doubles(m, auto)), 1), dtype=torch.float64, device="cuda")
_hip.call("ipx_gram_f64_mfma_split", m, n, dv._p(A.t), n, dv._p(G), dv._p(ws), auto, st)
Gh = G.cpu().numpy()[:m, :m]
ref = A_h @ A_h.T
out["gram_rel_err"] = float(np.max(np.abs(Gh - ref)) / np.max(np.abs(ref)))
G0 = G.clone()


def chol():
    G.copy_(G0)
    _hip.call("ipx_chol_factor", M, dv._p(G), dv._p(flag), dv._p(work), st)


t_copy = timed(lambda: G.copy_(G0))
out["chol_ms"] = timed(chol) - t_copy
chol()
L0 = G.clone()


def inverse():                  # (the inverse works in G's storage: every repetition from L)
    G.copy_(L0)
    _hip.call("ipx_chol_inverse", M, dv._p(G), dv._p(X), st)


out["inverse_ms"] = timed(inverse, reps=3) - t_copy
Xh = X.cpu().numpy()[:m, :m]
out["inverse_resid"] = float(np.max(np.abs(Xh @ ref - np.eye(m))))
x = dv.DVec.from_host(rng.standard_normal(n))
t = timed(lambda: A.gemv(x), reps=50)
out["gemv_A_us"] = 1e3 * t
out["gemv_A_GBs"] = 8.0 * m * n / (t * 1e-3) / 1e9
print(json.dumps(out, indent=1))
