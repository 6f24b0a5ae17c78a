python scripts/_dbg.py 2>&1 | tail -8
