import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mav-detection_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built library (the .so is git-ignored): build it once, as __graft_entry__.build() does
    so = os.path.join(ROOT, "mav-detection_amd", "mavflow", "libmavflow.so")
    if not os.path.exists(so) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "mav-detection_amd", "csrc")], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "foe_chain.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def fb_oracle():
    """The compiled C restatement of Farneback (oracle/farneback_oracle.c), built on demand."""
    from oracle import fb_oracle as mod
    return mod.load()


@pytest.fixture(scope="session")
def mav():
    """The product library through its ctypes loader; fails loudly when libmavflow.so is missing."""
    import mavflow
    return mavflow
