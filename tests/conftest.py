import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def split_state(fix, prefix):
    return {k[len(prefix):]: v for k, v in fix.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def ngan():
    """The product package (directory `neuron-gan_amd/`, imported as `neuron_gan_amd`)."""
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(params=["f32", "bf16x3"])
def conv_precision(request, ngan):
    """Run a GPU parity test under both conv arithmetic modes: exact fp32 MFMA and split-bf16 (same tolerances)."""
    ngan.ops.set_conv_precision(request.param)
    yield request.param
    ngan.ops.set_conv_precision("f32")
