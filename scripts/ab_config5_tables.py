"""A/B of the compact tables of the barrier problem's projection (ipx_boxschur_args.grp2,
yell_*; IPX_DEBUG_FORMS=no-compact-groups = the full tables) on BASELINE config 5: bench.config5_leg
in a child process per setting, alternating, [rounds] times.
    python scripts/ab_config5_tables.py [rounds]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = ("import json, sys; sys.path.insert(0, %r); import bench; "
         "print('LEG ' + json.dumps(bench.config5_leg()))" % ROOT)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
out = {"compact": [], "full": []}
for _ in range(rounds):
    for name, flag in (("compact", ""), ("full", "1")):
        env = dict(os.environ, IPX_DEBUG_FORMS="no-compact-groups" if flag else "")
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True,
                           timeout=600)
        legs = [l[4:] for l in p.stdout.splitlines() if l.startswith("LEG ")]
        if not legs:
            print(p.stdout[-2000:], p.stderr[-2000:])
            raise SystemExit(1)
        leg = json.loads(legs[-1])
        out[name].append({k: leg.get(k) for k in ("seconds", "seconds_in_projected_cg", "niter",
                                                   "cg_niter", "cg_iterations_per_s_in_solve")})
print(json.dumps(out, indent=1))
