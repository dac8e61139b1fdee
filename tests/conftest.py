import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Plain `pytest tests` on a machine without a GPU (or without the built library) skips the gpu-marked tests
    instead of failing them; `-m gpu` on the GPU box runs them."""
    if not any("gpu" in item.keywords for item in items):
        return
    try:
        from caretta_amd import engine
        have = engine.device_count() > 0
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no MI355X visible (caretta_amd has no CPU path)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def oracle_libm():
    from oracle.pyoracle import Oracle
    return Oracle(libm_exp=True)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(GOLDEN / name, allow_pickle=False)
        return cache[name]

    return load


def _reload_library_config():
    """The library reads its CARETTA_* calibration switches once, when it is loaded (caretta_amd/csrc/cr_config.h); a test
    that changes one has the library read them again."""
    from caretta_amd import _capi
    if _capi._lib is not None:
        _capi._lib.cr_config_reload()


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch, with setenv / delenv of a CARETTA_* variable followed by cr_config_reload(), and the library
    told again once the environment has been restored."""
    set_env, del_env = monkeypatch.setenv, monkeypatch.delenv
    touched = []

    def setenv(name, value, prepend=None):
        set_env(name, value, prepend)
        if name.startswith("CARETTA_"):
            touched.append(name)
            _reload_library_config()

    def delenv(name, raising=True):
        del_env(name, raising)
        if name.startswith("CARETTA_"):
            touched.append(name)
            _reload_library_config()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield monkeypatch
    monkeypatch.undo()
    if touched:
        _reload_library_config()
