"""Run bench.py (same arguments) with a watchdog: after HANG_AFTER seconds (default 90) every
thread's Python stack is written to stderr and the process exits -- where a multi-rank run sits
when it does not come back (dev tool).
    python -m torch.distributed.run --nproc-per-node 2 scripts/hang_probe.py --gpus 2 ..."""
import faulthandler, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
faulthandler.dump_traceback_later(int(os.environ.get("HANG_AFTER", "90")), exit=True)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
sys.path.insert(0, ROOT)
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
