#!/bin/bash
# Box-side refresh of the evidence under profiles/ (run through gpurun from the repo root):
# traces go to /tmp, only the summaries to gpurun_out/final/ (copy them into profiles/ by hand).
#   bash scripts/refresh_profiles.sh <commit> [steps: pmc attach bench bench20 trace pmc5 config5]
COMMIT=${1:-?}; shift
STEPS=${@:-bench trace pmc config5}
R=/root/repo; OUT=$R/gpurun_out/final; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for s in $STEPS; do case $s in
bench)
  (cd $R && timeout 900 python3 bench.py 2>/tmp/bench.err | grep '^{' > $OUT/bench_line.json); tail -c 600 $OUT/bench_line.json; echo;;
trace)
  rm -rf /tmp/bp; timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp -o b -- python3 $R/bench.py --no-cpu > /tmp/bench_trace.out 2>/tmp/bench_trace.err
  grep '^{' /tmp/bench_trace.out > $OUT/bench_under_trace.json
  cp $(find /tmp/bp -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv
  (cd $R && python3 scripts/kernel_durations_by_size.py /tmp/bp $OUT/bench_kernel_durations_by_size.json > /dev/null);;
pmc)
  for n in 1000000 16000000; do
    tag=n1e6; note="n=1e6, m=1e5"; [ $n = 16000000 ] && tag=n16e6 && note="n=1.6e7, m=1.6e6"
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pmc_$c; timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/scripts/pmc_workload.py $n > /dev/null 2>/tmp/pmc_$c.err
    done
    (cd $R && python3 scripts/pmc_summarize.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $OUT/pmc_traffic_$tag.json "$note" $COMMIT | tail -5)
  done;;
attach)   # (on the box: the bench line's `traffic` reads profiles/, guarded by the source hash)
  cp $OUT/pmc_traffic_n1e6.json $R/profiles/r06_pmc_traffic_n1e6.json; cp $OUT/pmc_traffic_n16e6.json $R/profiles/r06_pmc_traffic_n16e6.json;;
bench20)
  (cd $R && timeout 600 python3 bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' > $OUT/bench_line_steps20.json);;
pmc5)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc5_$c; timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc5_$c -- python3 $R/scripts/pmc_config5.py > /dev/null 2>/tmp/pmc5_$c.err
  done
  (cd $R && python3 scripts/pmc_config5.py --summarize /tmp/pmc5_FETCH_SIZE /tmp/pmc5_WRITE_SIZE $OUT/pmc_traffic_config5.json | tail -8);;
config5)
  rm -rf /tmp/p5; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5 -o c5 -- python3 $R/scripts/config5.py 500000 50000 > /tmp/c5.out 2>/tmp/c5.err
  grep '^{' /tmp/c5.out > $OUT/config5_under_trace.json
  cp $(find /tmp/p5 -name '*kernel_stats.csv' | head -1) $OUT/config5_kernel_stats.csv;;
esac; done
ls -la $OUT
