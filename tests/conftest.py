import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _install_oracle_cache():
    """The GPU suite's CPU-oracle CSVs come from tests/golden/oracle_cache when the inputs are byte-identical (tests/oracle_cache.py)."""
    import oracle_cache
    from oracle import pipeline as op
    if not hasattr(op.run_video, "__wrapped__"):
        op.run_video = oracle_cache.cached_run_video(op.run_video)


sys.path.insert(0, os.path.join(ROOT, "tests"))
_install_oracle_cache()
