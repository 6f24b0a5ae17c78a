timeout 900 python - <<'P' 2>&1 | tail -45 | cut -c1-200
import sys; sys.path.insert(0,'tests')
import fuzz_projections
print(fuzz_projections.run(40, 2, verbose=True, max_m=1500))
P
