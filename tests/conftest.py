import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a host without a GPU skips the gpu-marked tests (instead of erroring in their HipModel fixtures)."""
    try:
        import torch

        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a GPU (MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def cfg():
    from stylish_tts_amd.config import load_model_config

    return load_model_config()


@pytest.fixture(scope="session")
def weights(cfg):
    """Name-keyed synthetic weights for the five inference modules (same as the golden generator used)."""
    from stylish_tts_amd import params

    return {m: params.synth_state_dict(params.module_spec(m, cfg), 0, prefix=m + ".") for m in params.MODULE_SPECS}


def load_golden(name):
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
