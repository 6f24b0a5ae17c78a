python scripts/profile_public_call.py 1000000 20 2>&1 | grep -v amdgpu | head -4
python scripts/profile_public_call.py 125000 20 2>&1 | grep -v amdgpu | head -3
timeout 1800 python -m pytest tests/test_gpu_qp.py tests/test_gpu_late_barrier.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -6
timeout 1800 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -6
