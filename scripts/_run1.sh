python scripts/profile_config3.py 2>&1 | cut -c1-160 | head -120
