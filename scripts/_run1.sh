mkdir -p gpurun_out/final
IPX_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 2>/tmp/b2.err | grep '^{' > gpurun_out/final/bench_2rank.json
tail -3 /tmp/b2.err | cut -c1-300
python - <<'P'
import json
d=json.load(open('gpurun_out/final/bench_2rank.json')); print(d['value'], d['n_gpus'], d['transport'], d.get('parity_vs_single_gpu'), json.dumps(d.get('transport_ab'))[:400], json.dumps(d.get('wall_clock_to_gtol'))[:300])
P
