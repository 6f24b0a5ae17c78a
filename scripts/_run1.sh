timeout 1700 python tests/fuzz_banded_nlp.py 24 1 2>&1 | tail -30 | cut -c1-330
