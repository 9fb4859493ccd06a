"""Shared fixtures.  `-m gpu` tests need a MI355X; everything else runs on CPU."""
import ctypes
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "liboracle.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def _record_allclose():
    """Every `np.allclose(gpu, oracle, rtol, atol)` of the GPU tests also lands in the ledger (measured maxima next to the
    asserted bar, keyed by file:line): no call site has to change to be audited."""
    import inspect

    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _ledger
    orig = np.allclose

    def allclose(a, b, rtol=1e-05, atol=1e-08, equal_nan=False):
        try:
            fr = inspect.currentframe().f_back
            fn = os.path.basename(fr.f_code.co_filename)
            if fn.startswith("test_"):
                x, y = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
                x, y = np.broadcast_arrays(x, y)
                if x.size:
                    err = np.abs(x - y)
                    scale = np.maximum(np.abs(x), np.abs(y))
                    tol = atol + rtol * np.abs(y)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        rel = np.where(scale > 0, err / np.maximum(scale, 1e-300), 0.0)
                        used = np.where(tol > 0, err / np.where(tol > 0, tol, 1.0), np.where(err > 0, np.inf, 0.0))
                    _ledger.record(f"np.allclose at {fn}:{fr.f_lineno}", np.nanmax(err), np.nanmax(rel), np.nanmax(used), rtol, atol)
        except Exception:
            pass
        return orig(a, b, rtol=rtol, atol=atol, equal_nan=equal_nan)
    np.allclose = allclose


def pytest_sessionstart(session):
    if "gpu" in (session.config.getoption("-m") or "") and "not gpu" not in (session.config.getoption("-m") or ""):
        _record_allclose()


def pytest_sessionfinish(session, exitstatus):
    """The measured parity errors of this session (tests/_ledger.py) -> gpurun_out/r06_parity_errors.json."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _ledger
    path = _ledger.dump()
    if path:
        print(f"\n[ledger] measured parity errors written to {path}")


@pytest.fixture(scope="session")
def A():
    return graft.load_package()


@pytest.fixture(scope="session")
def P(A):
    return importlib.import_module("altro_cpp_amd.problems")


@pytest.fixture(scope="session")
def S(A):
    return importlib.import_module("altro_cpp_amd.sharding")


@pytest.fixture(scope="session")
def oracle_lib():
    if not os.path.exists(ORACLE_LIB):
        graft.build_oracle()
    return ctypes.CDLL(ORACLE_LIB)


@pytest.fixture(scope="session")
def oracle_make(A, oracle_lib):
    """Factory of oracle-backed solvers (CPU restatement, test infrastructure only)."""
    def make(n, m, N, batch, dtype):
        return A.BatchSolver(n, m, N, batch, dtype, _lib=oracle_lib, _prefix="oracle_")
    return make


@pytest.fixture(scope="session")
def hip_make(A):
    """Factory of product solvers (libaltro_hip.so through the C-ABI)."""
    def make(n, m, N, batch, dtype):
        return A.BatchSolver(n, m, N, batch, dtype)
    return make
