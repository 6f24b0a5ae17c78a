mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -12
python bench.py --no-configs --cpu-iters 40 --steps 20 --warmup 5 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_quick.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','steps')}, d['device_loop_only']['iterations_per_s'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
PY
