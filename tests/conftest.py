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
