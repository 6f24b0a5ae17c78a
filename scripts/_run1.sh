timeout 900 python -m pytest tests/test_gpu_qp.py -x -q -k "random_shapes or random_band" 2>&1 | tail -5
