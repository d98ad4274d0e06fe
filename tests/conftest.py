import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Return (tensors, cfg, batch) of tests/golden/<name>.safetensors."""
    from safetensors import safe_open
    path = os.path.join(GOLDEN_DIR, f"{name}.safetensors")
    tensors = {}
    with safe_open(path, framework="pt", device="cpu") as f:
        meta = f.metadata()
        for k in f.keys():
            tensors[k] = f.get_tensor(k)
    return tensors, json.loads(meta["config"]), int(meta["batch"])


def sub(tensors, prefix):
    """Entries under `prefix/` with the prefix stripped."""
    n = len(prefix) + 1
    return {k[n:]: v for k, v in tensors.items() if k.startswith(prefix + "/")}


def noise_keys(tensors):
    """Parameters whose exact gradient is zero: conv / temb-projection biases that feed a
    train-mode BatchNorm (the mean subtraction cancels them) and the key-projection bias of the
    attention block (it shifts every score of a softmax row equally).  Their computed gradient is
    rounding noise, ~1e-9 in the golden vectors."""
    return {k[len("grad/"):] for k in tensors
            if k.startswith("grad/") and k.endswith((".block1.0.2.bias", ".linear_temb.bias", ".k_proj.bias"))}


@pytest.fixture(scope="session", params=["t0", "t1", "t0_drop", "t0_pool"])
def golden(request):
    return (request.param,) + load_golden(request.param)


@pytest.fixture(scope="session", autouse=True)
def _conv_scratch_for_every_descriptor():
    """VAMD_TEST_SCRATCH=1: every ConvDesc the tests build carries a scratch buffer, so that the kernels that can
    split tiles between workgroups do so in all kernel tests (otherwise only the tests that ask for it)."""
    if os.environ.get("VAMD_TEST_SCRATCH") != "1":
        yield
        return
    import torch
    from vision_amd import _lib
    buf = torch.zeros(_lib.CONV_SCRATCH_BYTES, dtype=torch.uint8, device="cuda:0")
    _lib.set_default_scratch(buf)
    yield
    _lib.set_default_scratch(None)


@pytest.fixture
def no_splitk():
    """Whole-tile convolution launches only: whether a layer's reduction is split depends on its batch size, and tests
    that compare batch sizes bit for bit need one summation order."""
    from vision_amd._lib import lib
    old = lib.vamd_conv_splitk(0)
    yield
    lib.vamd_conv_splitk(old)
