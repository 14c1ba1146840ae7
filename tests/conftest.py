import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
# GF(2^8) hard decoding of n = 255 codes takes the bit-plane chain from ~4e5 frame-syndromes per call on (below that one
# wavefront per frame is faster); the parity tests want the chain at EVERY size (layout boundaries at 1, 31 .. 4161
# frames), so the suite runs with the threshold at 0.  tests/test_gpu_bitslice.py::test_small_calls_at_default_settings
# runs the default in a process of its own.
os.environ.setdefault("CC_AMD_PLANES_MIN_WORK", "0")
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
