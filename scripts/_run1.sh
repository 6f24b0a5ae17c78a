timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "random_constraint_mixes" --durations=2 2>&1 | tail -5
