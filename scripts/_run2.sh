mkdir -p gpurun_out/fuzz
( timeout 900 python tests/fuzz_fused_loop.py 300 11 2>&1 | tail -3 ) > gpurun_out/fuzz/fused.txt
( timeout 1500 python tests/fuzz_minimize.py 150 13 2>&1 | tail -3 ) > gpurun_out/fuzz/minimize.txt
( timeout 900 python tests/fuzz_banded_nlp.py 60 14 2>&1 | tail -3 ) > gpurun_out/fuzz/banded.txt
( timeout 600 python - <<'P' 2>&1 | tail -3
import sys; sys.path.insert(0,'tests')
import fuzz_projections
print("worst", fuzz_projections.run(300, 17, verbose=False, max_m=1500))
P
) > gpurun_out/fuzz/projections.txt
for f in gpurun_out/fuzz/fused.txt gpurun_out/fuzz/minimize.txt gpurun_out/fuzz/banded.txt gpurun_out/fuzz/projections.txt; do echo "== $f"; cut -c1-330 $f; done
