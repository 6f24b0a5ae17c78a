"""cProfile of scripts/full_solve.py (dev tool):  python scripts/profile_full_solve.py config2 [n m]"""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = ["full_solve.py"] + (sys.argv[1:] or ["config3"])
pr = cProfile.Profile()
pr.enable()
exec(open(os.path.join(ROOT, "scripts", "full_solve.py")).read())
pr.disable()
for key, k in (("cumulative", 45), ("tottime", 20)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(k)
    print(s.getvalue()[:9000])
