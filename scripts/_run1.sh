mkdir -p /root/repo/gpurun_out/final
true
R=/root/repo; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tp; rocprofv3 --kernel-trace --output-format csv -d /tmp/tp -- python3 $R/scripts/trace_public_call.py run 20 > /dev/null 2>/tmp/tp.err
python3 $R/scripts/trace_public_call.py show /tmp/tp > $R/gpurun_out/final/public_call_timeline.txt 2>&1
sed -n 10,16p $R/gpurun_out/final/public_call_timeline.txt; tail -1 $R/gpurun_out/final/public_call_timeline.txt
