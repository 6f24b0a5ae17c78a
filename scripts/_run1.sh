mkdir -p gpurun_out/final
timeout 900 python scripts/per_rank_sweep.py gpurun_out/final/per_rank_sweep.json 2>&1 | grep -v amdgpu | tail -6 | cut -c1-600
