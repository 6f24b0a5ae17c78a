R=/root/repo; cd /tmp && export TMPDIR=/tmp
for t in 512 1024 512 1024; do
  rm -rf /tmp/p5_$t; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5_$t -o c5 -- python3 $R/scripts/_dbg5.py $t 500000 50000 > /tmp/c5_$t.out 2>/tmp/c5_$t.err
  echo "== threads=$t"; grep -E "k_solve_pcr<0, [12], true" $(find /tmp/p5_$t -name '*kernel_stats.csv' | head -1) | sed -e 's/(int,.*)"/"/' | cut -c1-200; grep '^{' /tmp/c5_$t.out | cut -c1-200
done
