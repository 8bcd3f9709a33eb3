import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(1, HERE)          # test modules share helpers (wrap_ghosts, check_against_reference)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only)."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(autouse=True)
def _strip_paths_not_agglomerated_by_default(monkeypatch):
    """The small strips of the test cases would run every coarse multigrid depth agglomerated (suhmo_agg.hip, default threshold 100000
    cells per strip) and leave the strips' own coarse-depth paths (tile kernel on strips, locally computed right-hand sides, halo
    bookkeeping) untested: off by default here; the tests of the agglomeration set their own threshold."""
    if "SUHMO_AGG_MIN_CELLS" not in os.environ:
        monkeypatch.setenv("SUHMO_AGG_MIN_CELLS", "0")
