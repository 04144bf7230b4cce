import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kat1():
    with open(os.path.join(GOLDEN, "rans_kat1.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def katrec():
    with open(os.path.join(GOLDEN, "records_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import orc as _orc
    _orc.build()
    return _orc
