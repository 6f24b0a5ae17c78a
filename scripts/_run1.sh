timeout 900 python -m pytest tests/test_gpu_qp.py -x -q -k "random_band_shapes or random_structures or general_sparsity" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "general_sparsity" 2>&1 | tail -4
