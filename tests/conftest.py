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
def _switch_table_follows_the_environment(monkeypatch):
    """``ops`` caches the dispatch switches at import (``ops.reload_thresholds``: the ADYOLO_W4_* / ADYOLO_W4W_* thresholds,
    ADYOLO_W4_PERSIST / _NARROW, ADYOLO_WINO1D, ADYOLO_WGRAD_ALGO): tests move them with ``monkeypatch.setenv`` / ``delenv``, so
    re-read the table whenever such a variable is set or deleted and once more when the test's environment is restored.  (Code
    that writes ``os.environ`` directly calls ``ops.reload_thresholds()`` itself: README "Switches".)"""
    real_setenv, real_delenv = monkeypatch.setenv, monkeypatch.delenv

    def reload(name):
        if name.startswith("ADYOLO_") and "adyolo_amd" in sys.modules:
            from adyolo_amd import ops
            ops.reload_thresholds()

    def setenv(name, value, *a, **kw):
        real_setenv(name, value, *a, **kw)
        reload(name)

    def delenv(name, *a, **kw):
        real_delenv(name, *a, **kw)
        reload(name)
    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
    monkeypatch.undo()
    if "adyolo_amd" in sys.modules:
        try:
            from adyolo_amd import ops
            ops.reload_thresholds()
        except Exception:                                  # noqa: BLE001  (CPU box without the built library)
            pass
