"""csrc/dft20.h on the host (the header is plain C++ there): the 20-point prime-factor transform against numpy's FFT, and the
way csrc/logmel.hip's logmel_fft_kernel combines it -- two real frames per complex 400 = 20 x 20 transform, window in the time
domain, twiddles rebuilt from W^q and W^4q -- against numpy's rfft of each frame.  No GPU."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("dft20") / "dft20_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(HERE, "_dft20_harness.cpp"), "-o", exe], check=True)
    return exe


def test_dft20_matches_numpy(harness):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((50, 20)) + 1j * rng.standard_normal((50, 20))
    x[0] = 0
    x[1] = np.eye(20)[3]                                  # one impulse: a pure phase ramp, catches an index-map slip
    x[2] = np.exp(-2j * np.pi * 7 * np.arange(20) / 20)   # one tone: a single output bin
    text = "\n".join("%.17g %.17g" % (v.real, v.imag) for v in x.ravel())
    out = subprocess.run([harness], input=text, capture_output=True, text=True, check=True).stdout.split()
    got = np.array(out, dtype=np.float64).reshape(-1, 2)
    got = (got[:, 0] + 1j * got[:, 1]).reshape(50, 20)
    np.testing.assert_allclose(got, np.fft.fft(x, axis=1), atol=2e-14, rtol=0)


@pytest.mark.parametrize("seed", [1, 2])
def test_two_frames_per_transform_400(harness, seed):
    rng = np.random.default_rng(seed)
    a, b = rng.standard_normal(400), 1e-4 * rng.standard_normal(400)      # a loud and a quiet frame sharing one transform
    w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(400) / 400)).astype(np.float32).astype(np.float64)
    if seed == 2:
        w = w * (1.0 + 0.3 * np.arange(400) / 400.0)                     # not symmetric
    text = "\n".join("%.17g" % v for v in np.concatenate([a, b, w]))
    out = subprocess.run([harness, "400"], input=text, capture_output=True, text=True, check=True).stdout.split()
    got = np.array(out, dtype=np.float64).reshape(201, 2)
    pa, pb = np.abs(np.fft.rfft(a * w)) ** 2, np.abs(np.fft.rfft(b * w)) ** 2
    np.testing.assert_allclose(got[:, 0], pa, rtol=1e-11, atol=1e-12 * pa.max())
    # the quiet frame rides on the loud one's transform: its absolute error scales with the LOUD frame's magnitude (1e-16 of it)
    np.testing.assert_allclose(got[:, 1], pb, rtol=1e-6, atol=1e-13 * pa.max())
