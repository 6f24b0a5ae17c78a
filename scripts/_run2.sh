mkdir -p gpurun_out/fuzz
( timeout 900 python tests/fuzz_box_schur.py 300 31 2>&1 | tail -2 ) > gpurun_out/fuzz/box.txt
( timeout 1500 python tests/fuzz_minimize.py 200 33 2>&1 | tail -2 ) > gpurun_out/fuzz/minimize.txt
( timeout 900 python tests/fuzz_banded_nlp.py 80 34 2>&1 | tail -2 ) > gpurun_out/fuzz/banded.txt
( timeout 900 python tests/fuzz_sharded.py 2 60 35 2>&1 | grep -v "Gloo\|c10d\|amdgpu" | tail -2 ) > gpurun_out/fuzz/sharded2.txt
( timeout 900 python tests/fuzz_sharded.py 2 16 36 solves 2>&1 | grep -v "Gloo\|c10d\|amdgpu" | tail -2 ) > gpurun_out/fuzz/sharded2_solves.txt
( timeout 600 python - <<'P' 2>&1 | tail -3
import sys; sys.path.insert(0,'tests')
import fuzz_projections
print("worst", fuzz_projections.run(400, 37, verbose=False, max_m=1500))
P
) > gpurun_out/fuzz/projections.txt
for f in gpurun_out/fuzz/box.txt gpurun_out/fuzz/minimize.txt gpurun_out/fuzz/banded.txt gpurun_out/fuzz/sharded2.txt gpurun_out/fuzz/sharded2_solves.txt gpurun_out/fuzz/projections.txt; do echo "== $f"; cut -c1-330 $f; done
