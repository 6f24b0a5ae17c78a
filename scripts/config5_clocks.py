"""Clocks, power and the projected-CG rate over a config-5 solve (is the drift
of the loop's kernels from 43 to 56 us per iteration over a 6 s solve the chip's sustained
clocks?).  A side thread samples the GPU's sysfs files (engine / memory clock, power,
temperature; rocm-smi as a fall-back) every 50 ms and the solver's iteration counter; the
summary compares the first and the last second of the solve.

    python scripts/config5_clocks.py [out.json] [n]
"""
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))


def _current(path):
    """The active level of a pp_dpm_* file ('2: 2100Mhz *') in MHz."""
    try:
        for line in open(path):
            if "*" in line:
                return float(re.search(r"([0-9.]+)\s*[Mm][Hh]z", line).group(1))
    except Exception:
        return None
    return None


def _number(path, scale=1.0):
    try:
        return float(open(path).read().strip()) * scale
    except Exception:
        return None


class Sampler(threading.Thread):
    def __init__(self, counter, period=0.05):
        super().__init__(daemon=True)
        self.counter, self.period, self.rows, self.stop = counter, period, [], False
        devs = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        self.dev = self._mine([os.path.dirname(d) for d in devs])
        hw = sorted(glob.glob(os.path.join(self.dev, "hwmon", "hwmon*"))) if self.dev else []
        self.hw = hw[0] if hw else None
        self.source = "sysfs" if self.dev else "rocm-smi"

    @staticmethod
    def _mine(devs):
        """The sysfs device of THIS process's GPU among the node's: by PCI address (torch's
        device properties), else the one that gets busy under a short burst of work."""
        if not devs:
            return None
        import torch
        try:
            pr = torch.cuda.get_device_properties(torch.cuda.current_device())
            want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            for d in devs:
                if os.path.basename(os.path.realpath(d)).lower().startswith(want):
                    return d
        except Exception:
            pass
        x = torch.ones(1 << 26, device="cuda")
        t0 = time.time()
        while time.time() - t0 < 0.5:
            x.mul_(1.0001)
        busy = [(_number(os.path.join(d, "gpu_busy_percent")) or 0.0, d) for d in devs]
        torch.cuda.synchronize()
        return max(busy)[1]

    def sample(self):
        if self.dev:
            pw = None
            for name in ("power1_average", "power1_input"):
                if self.hw and pw is None:
                    pw = _number(os.path.join(self.hw, name), 1e-6)
            return {"sclk_mhz": _current(os.path.join(self.dev, "pp_dpm_sclk")),
                    "mclk_mhz": _current(os.path.join(self.dev, "pp_dpm_mclk")),
                    "fclk_mhz": _current(os.path.join(self.dev, "pp_dpm_fclk")),
                    "power_w": pw,
                    "temp_c": _number(os.path.join(self.hw, "temp1_input"), 1e-3) if self.hw else None,
                    "busy_pct": _number(os.path.join(self.dev, "gpu_busy_percent"))}
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"],
                                 capture_output=True, text=True, timeout=5).stdout
            card = next(iter(json.loads(out).values()))
            num = lambda k: next((float(re.search(r"([0-9.]+)", str(v)).group(1))
                                  for kk, v in card.items() if k in kk.lower()), None)
            return {"sclk_mhz": num("sclk"), "mclk_mhz": num("mclk"), "power_w": num("power")}
        except Exception as exc:
            return {"error": repr(exc)}

    def run(self):
        t0 = time.perf_counter()
        while not self.stop:
            row = self.sample()
            row["t"] = time.perf_counter() - t0
            row["cg_iterations"] = self.counter()
            self.rows.append(row)
            time.sleep(self.period)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
    import torch
    import ipsolver
    from ipsolver import cg_fused
    from ipsolver.synthetic import CenteredBandedNLP, DeviceCallbacks
    warnings.simplefilter("ignore")
    prob = CenteredBandedNLP(n, n // 10, eps=1.0)
    dc = DeviceCallbacks(prob)
    cons = (dc.constraints(ipsolver, ("less", 0.0)), ipsolver.BoxConstraint(("interval", -0.8, 0.8)))
    # (the iteration counter moves once per projected_cg call: ~70 times in the solve)
    ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, cons, max_iter=3)   # warm
    torch.cuda.synchronize()
    time.sleep(2.0)                       # (the chip idles: the solve starts from idle clocks)
    base = cg_fused.STATS["iterations"]
    smp = Sampler(lambda: cg_fused.STATS["iterations"] - base)
    idle = smp.sample()
    smp.start()
    t0 = time.perf_counter()
    res = ipsolver.minimize_constrained(dc.fun, dc.x0, dc.grad, dc.hess, cons)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    smp.stop = True
    smp.join()
    rows = smp.rows

    def window(lo, hi):
        sel = [r for r in rows if lo <= r["t"] < hi]
        if len(sel) < 2:
            return None
        mean = lambda k: (lambda v: sum(v) / len(v) if v else None)([r[k] for r in sel if r.get(k) is not None])
        its = sel[-1]["cg_iterations"] - sel[0]["cg_iterations"]
        dt = sel[-1]["t"] - sel[0]["t"]
        return {"from_s": lo, "to_s": hi, "sclk_mhz": mean("sclk_mhz"), "mclk_mhz": mean("mclk_mhz"),
                "power_w": mean("power_w"), "temp_c": mean("temp_c"),
                "cg_iterations_per_s": its / dt if dt > 0 else None,
                "us_per_cg_iteration_wall": 1e6 * dt / its if its else None}
    summary = {"n": n, "wall_s": wall, "status": int(res.status), "niter": int(res.niter),
               "cg_niter": int(res.cg_niter), "source": smp.source, "sysfs_device": smp.dev,
               "samples": len(rows),
               "idle_before": idle,
               "windows": [w for w in (window(a, a + 1.0) for a in range(0, int(wall) + 1)) if w],
               "note": "one-second windows of the solve: mean clocks / power of the samples in the "
                       "window, CG iterations finished in it (counted per projected_cg call) over "
                       "its wall time -- outer-loop work included, so the per-iteration figure is "
                       "an upper bound of the loop's own"}
    print(json.dumps(summary, indent=1))
    if out:
        with open(out, "w") as f:
            json.dump(dict(summary, rows=rows[::4]), f, indent=1)


if __name__ == "__main__":
    main()
