mkdir -p gpurun_out/final
IPX_BENCH_BACKEND=gloo python bench.py --gpus 2 2>/tmp/b2.err | grep '^{' > gpurun_out/final/bench_2rank.json; tail -c 200 gpurun_out/final/bench_2rank.json; echo; tail -3 /tmp/b2.err
python bench.py 2>/tmp/b.err | grep '^{' > gpurun_out/final/bench_line.json; tail -c 200 gpurun_out/final/bench_line.json; echo
