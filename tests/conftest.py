import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _w4_thresholds_follow_the_environment(monkeypatch):
    """``ops`` caches the ADYOLO_W4_* dispatch thresholds at import (``ops.reload_thresholds``): tests move them with
    ``monkeypatch.setenv``, so re-read them whenever such a variable is set and once more when the test's environment is
    restored."""
    real_setenv = monkeypatch.setenv

    def setenv(name, value, *a, **kw):
        real_setenv(name, value, *a, **kw)
        if name.startswith("ADYOLO_W4") and "adyolo_amd" in sys.modules:       # (ADYOLO_W4_* and ADYOLO_W4W_*)
            from adyolo_amd import ops
            ops.reload_thresholds()
    monkeypatch.setenv = setenv
    yield
    monkeypatch.undo()
    if "adyolo_amd" in sys.modules:
        try:
            from adyolo_amd import ops
            ops.reload_thresholds()
        except Exception:                                  # noqa: BLE001  (CPU box without the built library)
            pass
