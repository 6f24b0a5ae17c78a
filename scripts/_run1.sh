python scripts/_dbg_sharded.py 2>&1 | grep -v "amdgpu\|Gloo" | head -60
