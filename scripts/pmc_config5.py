"""Workload for PMC passes over the barrier problem's CG loop (rocprofv3 --pmc FETCH_SIZE /
WRITE_SIZE, separate passes): the calibration kernels of scripts/pmc_workload.py, then a few
outer iterations of BASELINE config 5 at full size.
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc5_f -- python3 scripts/pmc_config5.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc5_w -- python3 scripts/pmc_config5.py
    python scripts/pmc_config5.py --summarize /tmp/pmc5_f /tmp/pmc5_w profiles/<name>.json"""
import collections, csv, glob, json, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
NC = 8000000
# (round 5: the per-item back substitution -- k_pairs_post, 46.0 MB algorithmic -- is the tail of
# k_solve_pcr's launch; with IPX_DEBUG_FORMS=no-post-tail it is a launch of its own again)
KERNELS = {"k_cg_step1_box": 86.8e6, "k_solve_pcr": 19.4e6 + 46.0e6, "k_cg_step2_hp": 112.0e6}
if os.environ.get("IPX_DEBUG_FORMS", "").find("no-post-tail") >= 0:
    KERNELS = {"k_cg_step1_box": 86.8e6, "k_pairs_post": 46.0e6, "k_solve_pcr": None,
               "k_cg_step2_hp": 112.0e6}


def summarize(fetch_dir, write_dir, out):
    sys.path.insert(0, ROOT)
    from bench import kernel_source_hash

    def per_kernel(d, counter):
        f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True),
                key=os.path.getmtime)
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        return acc
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")

    def find(acc, key):
        hits = [k for k in acc if key in k]
        assert hits, key
        return acc[max(hits, key=lambda k: len(acc[k]))]
    fv = [v for v in find(fetch, "OpMul") if v > 0.5 * max(find(fetch, "OpMul"))]
    rf = 16.0 * NC / (sum(fv) / len(fv) * 1024)        # gfx950 FETCH_SIZE correction, in-run
    res = {}
    for key, alg in KERNELS.items():
        f, w = sorted(find(fetch, key)), sorted(find(write, key))
        # the loop's launches: the upper half of the size distribution (stopped launches of a
        # batch return at once and move nothing)
        f, w = f[len(f) // 2:], w[len(w) // 2:]
        fb, wb = f[len(f) // 2] * 1024 * rf, w[len(w) // 2] * 1024
        res[key] = {"launches": len(find(fetch, key)), "hbm_read_bytes": fb, "hbm_write_bytes": wb,
                    "hbm_bytes_per_launch": fb + wb, "algorithmic_bytes": alg}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over "
                         "scripts/pmc_config5.py (config 5 at full size, 30 outer iterations), "
                         "MI355X; FETCH calibrated in-run on k_map<OpMul> (read_factor %.3f)" % rf,
               "kernel_source_hash": kernel_source_hash(), "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if len(sys.argv) > 1 and sys.argv[1] == "--summarize":
    summarize(*sys.argv[2:5])
    sys.exit(0)
import torch
import ipsolver
from ipsolver import _hip, device as dv
from ipsolver.synthetic import CenteredBandedNLP, DeviceCallbacks
lib = _hip.load(); st = dv.stream_ptr()
a = torch.randn(NC, dtype=torch.float64, device="cuda"); b = torch.randn_like(a); o = torch.empty_like(a)
for _ in range(10):
    lib.ipx_mul(NC, dv._p(a), dv._p(b), dv._p(o), st)
torch.cuda.synchronize()
warnings.simplefilter("ignore")
prob = CenteredBandedNLP(500000, 50000, eps=1.0)
dc = DeviceCallbacks(prob)
cons = (dc.constraints(ipsolver, ("less", 0.0)), ipsolver.BoxConstraint(("interval", -0.8, 0.8)))
res = ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, cons, max_iter=int(os.environ.get("PMC5_ITERS", "30")))
torch.cuda.synchronize()
print("niter", res.niter, "cg", res.cg_niter)
