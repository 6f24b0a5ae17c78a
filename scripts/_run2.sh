bash scripts/refresh_profiles.sh final trace pmc config5 2>&1 | tail -4
cp gpurun_out/final/pmc_traffic_n1e6.json profiles/r04_pmc_traffic_n1e6.json
cp gpurun_out/final/pmc_traffic_n4e6.json profiles/r04_pmc_traffic_n4e6.json
python bench.py 2>/dev/null | grep '^{' > gpurun_out/final/bench_line.json
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep '^{' > gpurun_out/final/bench_line_steps20.json
make -C ip-nonlinear-solver_amd/csrc phase-timing > /dev/null 2>&1
python scripts/phase_timing_tail.py 2>&1 | tail -7 > gpurun_out/final/pcr_tail_phase_timing.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tp; rocprofv3 --kernel-trace --output-format csv -d /tmp/tp -- python3 /root/repo/scripts/trace_public_call.py run 20 > /dev/null 2>/tmp/tp.err
python3 /root/repo/scripts/trace_public_call.py show /tmp/tp > /root/repo/gpurun_out/final/public_call_timeline.txt 2>&1
