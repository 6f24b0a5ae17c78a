timeout 900 python -m pytest tests/test_gpu_qp.py -x -q -k "priming_forms" 2>&1 | tail -5
