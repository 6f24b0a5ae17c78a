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


def pytest_collection_modifyitems(config, items):
    """A test that hangs (a rank of a multi-process test waiting for a peer that died) must
    end, not burn the run: 15 minutes per test unless it says otherwise (pytest-timeout, when
    it is installed)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900))


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


@pytest.fixture(scope="session")
def qp_extra():
    with open(os.path.join(GOLDEN, "qp_extra.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def banded_refine2000():
    return dict(np.load(os.path.join(GOLDEN, "banded_refine_n2000.npz")))


@pytest.fixture(scope="session")
def config2_golden():
    with open(os.path.join(GOLDEN, "config2.json")) as f:
        return json.load(f)


def host(v):
    return v.to_host() if hasattr(v, "to_host") else np.asarray(v, dtype=float)


def rel_err(a, b, zero_scale=None):
    """max|a - b| / max|b| over the finite entries of the expected vector ``b`` (a truly
    relative, norm-wise measure: BASELINE.json's "1e-10 relative").  Non-finite entries must
    match exactly.  An all-zero ``b`` is measured against ``zero_scale`` (the scale of the
    inputs the zero was computed from), or absolutely when none is given."""
    a, b = host(a), np.asarray(b, dtype=float)
    assert a.shape == b.shape, (a.shape, b.shape)
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin)
    assert np.array_equal(a[~fin], b[~fin])
    if not fin.any():
        return 0.0
    scale = np.max(np.abs(b[fin]))
    if scale == 0:
        scale = 1.0 if zero_scale is None else zero_scale
    return float(np.max(np.abs(a[fin] - b[fin])) / scale)


def close(a, b, tol, zero_scale=None):
    err = rel_err(a, b, zero_scale)
    assert err <= tol, "relative error %.3e > %.1e" % (err, tol)


def close_projection(z, want, p, tol):
    """``z = Z p`` against the reference's value.  Relative to ``max|want|`` -- unless the
    projection cancels (almost) all of ``p`` (the reference's test point ``row 3 of A +
    1e-10 e_8``): what is left is then the rounding of ``p`` itself, on which the reference's
    own factorizations differ by 2e-5, and the error is measured against ``max|p|``."""
    want = np.asarray(want, dtype=float)
    pmax = float(np.max(np.abs(p)))
    if np.max(np.abs(want)) < 1e-6 * pmax:
        assert np.max(np.abs(host(z) - want)) <= tol * pmax
    else:
        close(z, want, tol)
