import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The .so files are git-ignored build artefacts: (re)build the HIP library like __graft_entry__.build() whenever it
    is missing or older than any source under csrc/ or include/mvn.h, so that the ABI and GPU tests never run against a
    stale library.  build_hip() only invokes hipcc when something changed; a box without hipcc and with a stale library
    fails the session instead of testing code that was never compiled."""
    import __graft_entry__ as ge

    if not os.environ.get("MVN_LIB_PATH"):
        ge.build_hip()


def _reload_switches():
    """libmvn_hip.so reads its MVN_* switches once per process; tell it when a test has changed them."""
    mvn = sys.modules.get("meta_viterbinet_amd")
    if mvn is not None and mvn._lib._lib is not None:
        mvn._lib._lib.mvn_reload_switches()


@pytest.fixture(autouse=True)
def _mvn_switches(monkeypatch):
    """Every monkeypatch.setenv / delenv of an MVN_* switch takes effect at once (the library re-reads its switches), and
    a test starts from the environment as it is (whatever the previous test's teardown restored)."""
    _reload_switches()
    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def setenv_reload(name, value, *a, **k):
        setenv(name, value, *a, **k)
        if name.startswith("MVN_"):
            _reload_switches()

    def delenv_reload(name, *a, **k):
        delenv(name, *a, **k)
        if name.startswith("MVN_"):
            _reload_switches()

    monkeypatch.setenv, monkeypatch.delenv = setenv_reload, delenv_reload
    yield
    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]

    return load


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (tests are one of the three places allowed to use it)."""
    import oracle as _oracle

    _oracle.build()
    return _oracle
