import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver's GPU tier)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden(object):
    """npz with '/'-separated keys -> nested access helpers."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name))

    def __getitem__(self, k):
        return self.z[k]

    def t(self, k):
        return torch.from_numpy(np.array(self.z[k]))

    def sub(self, prefix):
        p = prefix + "/"
        return {k[len(p):]: self.z[k] for k in self.z.files if k.startswith(p)}


@pytest.fixture(scope="session")
def golden_modules():
    return Golden("modules.npz")


@pytest.fixture(scope="session")
def golden_steps():
    return Golden("steps.npz")


@pytest.fixture(scope="session")
def golden_steps_f2():
    return Golden("steps_f2.npz")


@pytest.fixture(scope="session")
def golden_steps_f4():
    return Golden("steps_f4.npz")


@pytest.fixture(scope="session")
def golden_sdf():
    return Golden("sdfnet_examples.npz")


@pytest.fixture(scope="session")
def chairs_state():
    z = np.load(os.path.join(GOLDEN, "sdfnet_chairs_weights.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def summarize(t):
    a = t.detach().double().cpu().reshape(-1)
    return np.concatenate([[a.sum().item(), a.abs().sum().item()], a[:4].numpy(), a[-4:].numpy()])


def assert_summary_close(t, ref, rtol, atol, what=""):
    """Compares a tensor against the (sum, abs-sum, first 4, last 4) summary stored in a golden file."""
    got = summarize(t)
    scale = max(float(ref[1]) / max(t.numel(), 1), 1e-30)  # mean |x|
    # the signed sum of n terms of typical size `scale` carries ~sqrt(n)*eps*scale rounding noise
    assert abs(got[1] - ref[1]) <= rtol * abs(ref[1]) + atol * t.numel(), (what, "abs-sum", got[1], ref[1])
    assert abs(got[0] - ref[0]) <= rtol * abs(ref[1]) + atol * t.numel(), (what, "sum", got[0], ref[0])
    np.testing.assert_allclose(got[2:], ref[2:], rtol=rtol * 10, atol=max(atol, scale * rtol * 10), err_msg=what)
