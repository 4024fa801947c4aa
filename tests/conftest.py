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


@pytest.fixture(scope="session", autouse=True)
def _native_library():
    """Build (or refresh) tal_asrd_amd/libtal_asrd_hip.so before any test touches it."""
    import __graft_entry__ as g
    g.build()


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def sd_weights():
    """Synthetic SDModel weights keyed by the reference's state_dict names."""
    import json
    from tal_asrd_amd import synth
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))["SDModel"]
    return synth.fill_state_dict({k: tuple(s) for k, s in keys})


@pytest.fixture(scope="session")
def asr_weights():
    import json
    from tal_asrd_amd import synth
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))["ASRModel_2x_spk"]
    return synth.fill_state_dict({k: tuple(s) for k, s in keys})
