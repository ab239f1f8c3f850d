import os
import subprocess
import sys

import pytest

os.environ.setdefault("LZ4FLEX_TEST_HOOKS", "1")      # the library's fault-injection keys ("debug_*") only exist for processes that opt in
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The CPU oracle is test infrastructure: build it on demand (gcc, a second or two)."""
    so = os.path.join(ROOT, "oracle", "liblz4flex_oracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in
            ("lz4flex_block.c", "lz4flex_frame.c", "lz4flex_bench.c", "lz4flex_oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    yield


@pytest.fixture
def exact_encoder():
    """GPU tests that compare encoder bytes with the oracle (= lz4_flex's bytes) run the reference-exact encoder; the
    default is the throughput encoder, whose contract is a valid block (tests/test_gpu_wave_encoder.py)."""
    from lz4_flex_amd import block
    block.set_compress_mode("exact")
    yield
    block.set_compress_mode("fast")
