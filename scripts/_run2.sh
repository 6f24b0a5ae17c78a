timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
mkdir -p gpurun_out/final
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep '^{' > gpurun_out/final/bench_line_steps20.json
python bench.py 2>/dev/null | grep '^{' > gpurun_out/final/bench_line.json
python - <<'P'
import json
for f in ('gpurun_out/final/bench_line_steps20.json','gpurun_out/final/bench_line.json'):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['device_loop_only']['iterations_per_s'], d['wall_clock_to_gtol']['seconds'], d['config5']['seconds'], d['config2']['seconds'], json.dumps(d['public_api']['trust_radius_finite']))
P
