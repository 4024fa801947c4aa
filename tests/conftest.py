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
    config.addinivalue_line("markers", "range_fallback: the test drives the encoder out of the fp16 range on purpose")


@pytest.fixture(autouse=True)
def _no_silent_range_fallback(request):
    """The fp16-range guard re-runs an encoder call on the exact fp32 kernels when a kernel raises the status word.  On the
    O(1) data of these tests that must never happen: a fallback would hide a false positive of the guard behind slightly
    different (and slower) results."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from tal_asrd_amd import ops
    before = ops.range_fallbacks
    yield
    if request.node.get_closest_marker("range_fallback") is None:
        assert ops.range_fallbacks == before, "the fp16-range guard fired on in-range data (%d re-runs)" % (ops.range_fallbacks - before)


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
