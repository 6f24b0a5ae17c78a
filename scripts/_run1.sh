python scripts/_dbg.py 2>&1 | grep -v amdgpu | tail -30 | cut -c1-260
