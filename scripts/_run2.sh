timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash scripts/refresh_profiles.sh final trace pmc 2>&1 | tail -3
cp gpurun_out/final/pmc_traffic_n1e6.json profiles/r04_pmc_traffic_n1e6.json
cp gpurun_out/final/pmc_traffic_n4e6.json profiles/r04_pmc_traffic_n4e6.json
python bench.py 2>/dev/null | grep '^{' > gpurun_out/final/bench_line.json
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep '^{' > gpurun_out/final/bench_line_steps20.json
