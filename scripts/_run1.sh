timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "sharded_random_constraint" --durations=2 2>&1 | tail -5
