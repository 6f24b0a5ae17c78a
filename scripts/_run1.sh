timeout 1500 python -m pytest tests/test_gpu_e2e.py -x -q -s -k "general_sparsity_barrier_at_scale or complex_step or finite_difference_hessians" 2>&1 | grep -v "^$" | tail -12
