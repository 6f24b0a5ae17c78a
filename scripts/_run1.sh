timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "config2_device_callbacks" 2>&1 | grep -v "^$" | tail -25
python scripts/profile_public_call.py 1000000 20 2>&1 | grep -v amdgpu | head -50
