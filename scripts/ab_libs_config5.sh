#!/bin/bash
# A/B of kernel variants on config 5: for each directory under exp_libs/ (experimental builds of
# libipx.so, not tracked) and the tree's own library, rocprofv3 kernel stats of
# scripts/config5.py 500000 50000; prints the five kernels of the CG iteration.
#   bash scripts/ab_libs_config5.sh [variant ...]
cd /tmp && export TMPDIR=/tmp
LIB=/root/repo/ip-nonlinear-solver_amd/lib/libipx.so
cp $LIB /tmp/libipx_base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/libipx_base.so $LIB; else cp /root/repo/exp_libs/$v/libipx.so $LIB; fi
  rm -rf /tmp/p_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$v -o c5 -- \
    python3 /root/repo/scripts/config5.py 500000 50000 > /tmp/c5_$v.json 2>/tmp/c5_$v.err
  echo "== $v: $(grep '^{' /tmp/c5_$v.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["wall_s"], d["niter"], d["cg_niter"], d["fun"])')"
  python3 - <<PY
import csv
rows = list(csv.DictReader(open('/tmp/p_$v/c5_kernel_stats.csv')))
tot = 0.0
for r in rows[:5]:
    print('   ', r['Name'][28:80].ljust(52), r['Calls'], round(float(r['AverageNs']) / 1e3, 2))
    tot += float(r['AverageNs']) / 1e3
print('    sum of the five:', round(tot, 2), 'us')
PY
done
cp /tmp/libipx_base.so $LIB
