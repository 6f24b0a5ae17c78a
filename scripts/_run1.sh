python scripts/_dbg.py 2>&1 | grep -v amdgpu | tail -6 | cut -c1-300
