import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fn-ssl_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (ROCm device); run with -m gpu on the GPU box")


def rs_randn(seed, shape, scale=1.0):
    """Same input generator as tests/golden/make_golden.py."""
    return (np.random.RandomState(int(seed)).standard_normal(size=tuple(int(s) for s in shape)) * scale
            ).astype(np.float32)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def assert_close(got, want, rtol, atol, what=""):
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, "%s shape %s vs %s" % (what, got.shape, want.shape)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    bad = err > tol
    if bad.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError("%s: %d/%d out of tol (rtol %g atol %g); worst at %s got %r want %r; max abs err %g"
                             % (what, int(bad.sum()), bad.size, rtol, atol, i, got[i], want[i], float(err.max())))


@pytest.fixture(autouse=True)
def _fnssl_tuning_follows_env(monkeypatch):
    """The library reads no environment variable (include/fnssl.h: fnssl_tuning); fnssl/_lib.py parses FNSSL_<KNOB>
    variables when it loads the library and on ``refresh_tuning()``.  The tests flip knobs with ``monkeypatch.setenv`` in
    the middle of a test, so: re-read the environment at the start of every test (the previous test's monkeypatch has been
    undone by then) and after every setenv / delenv of an FNSSL_* variable."""
    from fnssl import _lib

    def sync():
        if _lib._lib is not None:
            _lib.refresh_tuning()

    sync()
    orig_set, orig_del = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, prepend=None):
        orig_set(name, value, prepend)
        if name.startswith("FNSSL_"):
            sync()

    def delenv(name, raising=True):
        orig_del(name, raising)
        if name.startswith("FNSSL_"):
            sync()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
    monkeypatch.setenv, monkeypatch.delenv = orig_set, orig_del
