"""Randomly drawn render configurations (scene, frame size, sample budget and batch, depth, light samples,
tolerance, bounce accumulation, hemisphere sampling, specular mode, look-ahead policy of the round loop, paths per
launch, shadow rays in a launch of their own):
the HIP path through ptamd_render against the CPU oracle, by the criteria of test_render_matches_oracle.
A fixed seed, so the cases are the same on every run; tools/fuzz_parity.py draws as many more as asked."""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import fuzz_parity  # noqa: E402

pytestmark = pytest.mark.gpu

_rng = random.Random(20240611)
_CASES = [fuzz_parity.draw(_rng) + (_rng.randrange(1 << 30),) for _ in range(36)]


@pytest.mark.parametrize("name,w,h,kw,per_pixel,shape,seed", _CASES)
def test_random_configuration_matches_oracle(device, name, w, h, kw, per_pixel, shape, seed):
    ok, same, close = fuzz_parity.check(device, name, w, h, kw, per_pixel, seed, shape)
    assert ok, "sample counts equal on %.4f of the pixels, radiances close on %.4f" % (same, close)
