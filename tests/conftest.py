import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def tx():
    """the product's scene module (GPU tests)"""
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from texir_code_amd import scene as S
    return S


def rel_l2(a, b):
    import numpy as np
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(autouse=True)
def _oracle_uses_all_cores():
    """the runners call torch.set_num_threads(1) like the reference's, which lowers the process-wide OpenMP default; the CPU oracle
    (test infrastructure) should keep using every core whatever ran before"""
    try:
        from oracle import oracle as O
        O.set_num_threads(os.cpu_count() or 1)
    except Exception:
        pass                    # oracle library not built: tests that need it fail on their own
    yield


@pytest.fixture
def monkeypatch(monkeypatch):
    """the stock fixture, plus: libtexir_hip.so parses its TEXIR_* switches once at load (csrc/env.h), so every change a test makes to such
    a variable is followed by texir_reload_env() -- and once more after the test's changes have been undone"""
    from texir_code_amd import _lib

    class Patched:
        def __getattr__(self, name):
            return getattr(monkeypatch, name)

        def setenv(self, name, value, prepend=None):
            monkeypatch.setenv(name, value, prepend)
            if name.startswith("TEXIR_"):
                _lib.reload_env()

        def delenv(self, name, raising=True):
            monkeypatch.delenv(name, raising)
            if name.startswith("TEXIR_"):
                _lib.reload_env()

    yield Patched()
    monkeypatch.undo()
    _lib.reload_env()
