import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = ["full_solve.py", "config3"]
pr = cProfile.Profile()
pr.enable()
exec(open(os.path.join(ROOT, "scripts", "full_solve.py")).read())
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
