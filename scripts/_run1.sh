mkdir -p gpurun_out/final
python bench.py 2>/dev/null | grep '^{' > gpurun_out/final/bench_line.json
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep '^{' > gpurun_out/final/bench_line_steps20.json
python scripts/phase_timing_tail.py 2>&1 | tail -7 > gpurun_out/final/pcr_tail_phase_timing.txt
tail -c 300 gpurun_out/final/bench_line.json
