import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the real reference build in oracle/_ref")


@pytest.fixture(scope="session")
def ref_libs():
    from checkers import RefLib
    if not RefLib.available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    return RefLib.get(0), RefLib.get(1)
