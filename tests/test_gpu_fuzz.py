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


# Parameters far from the benchmark table's: many light samples per vertex (also with the shadow rays in launches of
# their own and rounds cut into launches of 3000 paths), path depths at which only Russian roulette ends a path, a
# long sample budget on a frame of sixteen pixels under either look-ahead policy, a batch size that divides nothing.
_EXTREMES = [
    ("CBspheres_lambertian", 32, 24, dict(ns_aa=8, max_ray_depth=3, ns_area_light=64), None, (3000, True)),
    ("CBspheres_lambertian", 32, 24, dict(ns_aa=16, max_ray_depth=100, ns_area_light=1), None, (0, False)),
    ("CBbunny", 24, 18, dict(ns_aa=16, max_ray_depth=60, ns_area_light=2), None, (0, True)),
    ("CBspheres_lambertian", 4, 4, dict(ns_aa=4096, max_ray_depth=3, ns_area_light=2, samples_per_batch=64), None, (0, False)),
    ("CBspheres_lambertian", 4, 4, dict(ns_aa=4096, max_ray_depth=3, ns_area_light=2, samples_per_batch=64), "0", (0, False)),
    ("CBgems", 16, 12, dict(ns_aa=2048, max_ray_depth=4, ns_area_light=1, max_tolerance=0.01), None, (20000, False)),
    ("banana", 32, 24, dict(ns_aa=64, max_ray_depth=5, ns_area_light=17, samples_per_batch=7), "1", (0, True)),
    ("CBspheres", 32, 24, dict(ns_aa=64, max_ray_depth=30, ns_area_light=3, specular=True), None, (0, False)),
    ("CBempty", 32, 24, dict(ns_aa=32, max_ray_depth=8, ns_area_light=200), None, (0, False)),
]


@pytest.mark.parametrize("name,w,h,kw,per_pixel,shape", _EXTREMES)
def test_extreme_parameters_match_oracle(device, name, w, h, kw, per_pixel, shape):
    ok, same, close = fuzz_parity.check(device, name, w, h, kw, per_pixel, 99, shape)
    assert ok, "sample counts equal on %.4f of the pixels, radiances close on %.4f" % (same, close)
