mkdir -p gpurun_out
make -C ip-nonlinear-solver_amd/csrc phase-timing > gpurun_out/make.log 2>&1; tail -1 gpurun_out/make.log
python scripts/phase_timing_resident.py 125000 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_qp.py -x -q -k "resident_loop" 2>&1 | tail -5
python scripts/per_rank_sweep.py gpurun_out/per_rank_sweep.json > gpurun_out/sweep.log 2>&1
grep -o '"N": [0-9]*\|"resident": {"finite_radius": {"us_per_iteration_median": [0-9.]*\|max_abs_diff_vs_three_launches": [0-9.e-]*' gpurun_out/sweep.log
