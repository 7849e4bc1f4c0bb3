import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    _spread_cpu_suite(config)


def _spread_cpu_suite(config):
    """`pytest -m "not gpu"` (the CPU gate: oracle, host logic and the kernels under the fiber emulator — ~1 h 45 min of single-threaded work) is spread over the
    host's cores with pytest-xdist when the caller did not ask for a distribution mode itself: the same tests, one process per core (8 at most).  Never for
    `-m gpu` (one GPU: the full-size tests plan tens of GB each) nor when VCAD_TEST_SERIAL=1.  Done here rather than in an ini file because it must depend on -m."""
    if hasattr(config, "workerinput") or os.environ.get("VCAD_TEST_SERIAL") == "1":
        return
    opt = config.option
    if (getattr(opt, "markexpr", "") or "").strip() != "not gpu" or not config.pluginmanager.hasplugin("xdist"):
        return
    if getattr(opt, "numprocesses", None) or getattr(opt, "dist", "no") != "no" or getattr(opt, "tx", None):
        return
    n = max(1, min(8, os.cpu_count() or 1))
    if n > 1:
        opt.numprocesses, opt.dist, opt.tx = n, "load", ["popen"] * n


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
