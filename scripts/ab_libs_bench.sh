#!/bin/bash
# A/B of kernel variants on the headline loop: for the tree's library and each directory under
# exp_libs/ (experimental builds of libipx.so, not tracked), the short form of bench.py.
#   bash scripts/ab_libs_bench.sh [variant ...]
cd /root/repo
LIB=/root/repo/ip-nonlinear-solver_amd/lib/libipx.so
cp $LIB /tmp/libipx_base.so
for v in base "$@" base; do
  if [ "$v" = base ]; then cp /tmp/libipx_base.so $LIB; else cp /root/repo/exp_libs/$v/libipx.so $LIB; fi
  timeout 300 python3 bench.py --no-cpu --no-configs --no-weak ${BENCH_FLAGS:---no-big} 2>/tmp/ab_$v.err | grep '^{' | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d.get('repeat', {}).get('iterations_per_s', {})
print('$v', round(d['value']), 'median', round(r.get('median', 0)), 'per kernel us', {k: round(x, 2) for k, x in d['per_kernel_us'].items()}, 'unbounded', round(d['unbounded_trust_region']['iterations_per_s']), 'big', round(d.get('roofline_out_of_cache', {}).get('iterations_per_s', 0)))
"
done
cp /tmp/libipx_base.so $LIB
