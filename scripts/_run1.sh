mkdir -p gpurun_out/final
python bench.py --no-cpu 2>/dev/null | grep '^{' > gpurun_out/final/bench_tb512.json
python - <<'P'
import json
d=json.load(open('gpurun_out/final/bench_tb512.json')); print(d['value'], d['ms_per_step'], d['device_loop_only']['iterations_per_s'], d['device_loop_only']['repeat']['iterations_per_s'], d['wall_clock_to_gtol']['seconds'], d['config5']['seconds'], d['roofline_out_of_cache']['iterations_per_s'])
P
