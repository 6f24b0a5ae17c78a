timeout 900 python -m pytest tests/test_gpu_qp.py -x -q -k "random_band_shapes" 2>&1 | tail -5
