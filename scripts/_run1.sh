timeout 900 python - <<'P' 2>&1 | grep -v amdgpu | tail -14 | cut -c1-220
import sys; sys.path.insert(0,'tests')
import fuzz_projections
print("worst", fuzz_projections.run(150, 51, verbose=True, max_m=1500))
P
