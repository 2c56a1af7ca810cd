import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle binding (test infrastructure; built on demand with g++)."""
    import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def goldens():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "playthroughs.json"), encoding="utf-8") as f:
        return json.load(f)
