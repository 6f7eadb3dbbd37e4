import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The folded ConvTranspose + concat + conv op (include/biu.h: biu_foldt_*) is taken only where a decoder level is large enough for it to pay;
# the test networks are small: take it wherever the kernels serve the shapes, so that every parity test exercises it (the library reads the
# variable once, at its first biu_foldt_ok call; the size rule itself is what bench.py and a default process run)
os.environ.setdefault("BIU_FOLDT", "always")
# The rolling-window convolution with register-resident weights (biu_conv_roll.hip) is taken by size too (narrow 3-D bf16 layers whose volume
# fills the chip): the tests take it wherever it serves the shapes; BIU_DISABLE=croll is the A/B switch (tests/test_gpu_variants.py)
os.environ.setdefault("BIU_ROLL", "always")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
