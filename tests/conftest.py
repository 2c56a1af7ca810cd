import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: over a minute of reference CPU work (still part of -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle binding (test infrastructure; built on demand with g++)."""
    import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def reference():
    """The GENUINE reference build (oracle/_ref/libspiel_ref.so: the reference's own .cc files
    compiled by oracle/Makefile.ref), bound to the same calls as the oracle.  Built here when
    /root/reference is present; otherwise the prebuilt library is used, and tests that need it
    are skipped when neither exists."""
    import reference_py
    if reference_py.sources_present():
        reference_py.build()
    if not reference_py.available():
        pytest.skip("oracle/_ref/libspiel_ref.so not built (needs the reference sources)")
    return reference_py


@pytest.fixture(scope="session")
def goldens():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "playthroughs.json"), encoding="utf-8") as f:
        return json.load(f)
