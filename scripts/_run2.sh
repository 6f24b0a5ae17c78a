timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
bash scripts/refresh_profiles.sh wip bench trace pmc config5 2>&1 | tail -30
