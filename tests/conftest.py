import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ip-nonlinear-solver_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def unjson(v):
    """Inverse of make_golden.jf."""
    if isinstance(v, str):
        return {"inf": np.inf, "-inf": -np.inf, "nan": np.nan}[v]
    if isinstance(v, list):
        return [unjson(t) for t in v]
    return v


@pytest.fixture(scope="session")
def qp_small():
    with open(os.path.join(GOLDEN, "qp_small.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def e2e_golden():
    with open(os.path.join(GOLDEN, "e2e.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def banded2000():
    return dict(np.load(os.path.join(GOLDEN, "banded_n2000.npz")))


@pytest.fixture(scope="session")
def banded20000():
    return dict(np.load(os.path.join(GOLDEN, "banded_n20000.npz")))
