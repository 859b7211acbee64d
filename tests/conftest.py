import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
PARAMS = os.path.join(ROOT, "tests", "params")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built libraries (they are git-ignored): build them once (hipcc cross-compiles without a GPU)
    libs = [os.path.join(ROOT, "gandalf_amd", "csrc", "libgandalf_hip.so"), os.path.join(ROOT, "gandalf_amd", "host", "libgandalf_host.so"),
            os.path.join(ROOT, "oracle", "libgandalf_oracle.so")]
    if not all(os.path.exists(f) for f in libs):
        import __graft_entry__
        __graft_entry__.build()


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def golden():
    return load_golden
