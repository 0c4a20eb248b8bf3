#!/usr/bin/env python3
"""Random render configurations, the HIP path against the CPU oracle (same scenes, cameras, parameters and
random-number keys; the criteria of tests/test_gpu_parity.py::test_render_matches_oracle).
usage: fuzz_parity.py [seed [count]]      (needs a GPU; a few configurations per second)"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCENES = ["CBspheres_lambertian", "CBspheres", "CBgems", "CBbunny", "cube", "banana", "plane", "CBempty", "CBcoil"]
SIZES = [(40, 30), (33, 17), (64, 48), (8, 8), (1, 1), (97, 3)]


def draw(rng):
    """One configuration: (scene, w, h, render keywords, PTAMD_PER_PIXEL or None, launch shape). The launch shape
    (paths per launch, or 0 for the default; whether the context sends shadow-ray lists of more than 2000 rays
    to a launch of their own) makes small frames take the routes that only big ones take by default: rounds cut
    into several launches, and k_shadow behind k_trace2."""
    name = rng.choice(SCENES)
    w, h = rng.choice(SIZES)
    kw = dict(ns_aa=rng.choice([1, 2, 7, 16, 31, 32, 33, 64, 96, 130]), max_ray_depth=rng.choice([0, 1, 2, 3, 4, 5, 6, 9, 13]),
              ns_area_light=rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16]), samples_per_batch=rng.choice([1, 2, 4, 5, 16, 32]),
              max_tolerance=rng.choice([0.02, 0.05, 0.2, 0.5]), accumulate_bounces=rng.random() < 0.75,
              hemisphere=rng.random() < 0.2, specular=rng.random() < 0.2)
    if rng.random() < 0.25:  # a cell of the frame only (the reference's -p option)
        xa, xb = sorted((rng.randrange(w + 1), rng.randrange(w + 1)))
        ya, yb = sorted((rng.randrange(h + 1), rng.randrange(h + 1)))
        kw["rect"] = (xa, ya, xb, yb)
    return name, w, h, kw, rng.choice([None, "0", "1"]), (rng.choice([0, 0, 700, 5000, 65536]), rng.random() < 0.3)


_split_dev = []


def split_device():
    """A context that traces shadow-ray lists of more than 2000 rays in a launch of their own (k_shadow)."""
    import pathtracer_amd as pt
    if not _split_dev:
        old = os.environ.get("PTAMD_FUSE_BELOW")
        os.environ["PTAMD_FUSE_BELOW"] = "2000"  # read when a context is created
        try:
            _split_dev.append(pt.Device(0))
        finally:
            if old is None:
                os.environ.pop("PTAMD_FUSE_BELOW", None)
            else:
                os.environ["PTAMD_FUSE_BELOW"] = old
    return _split_dev[0]


def check(dev, name, w, h, kw, per_pixel, seed, shape=(0, False)):
    """Renders on the GPU and with the oracle; returns (ok, share of equal sample counts, share of close pixels)."""
    import pathtracer_amd as pt
    from pathtracer_amd import cabi
    from conftest import scene_path
    import oracle_lib as O
    if per_pixel is None:
        os.environ.pop("PTAMD_PER_PIXEL", None)
    else:
        os.environ["PTAMD_PER_PIXEL"] = per_pixel
    if shape[1]:
        dev = split_device()
    try:
        sc = pt.Scene(scene_path(name))
        arr, cam = sc.arrays(), sc.camera(w, h)
        dev.upload(sc)
        dev.set_camera(cam)
        dev.set_max_paths(shape[0])
        p = cabi.make_params(w, h, seed=seed, **kw)
        hdr, cnt = dev.render(p)
        rh, rc, _ = O.render(arr, cam, p, threads=16)
    finally:
        os.environ.pop("PTAMD_PER_PIXEL", None)
        dev.set_max_paths(0)
    # float sinf / cosf in the hemisphere samplers (sampler.cpp:35-49): directions agree to ~1e-7 only
    float_trig = bool(kw.get("hemisphere")) or bool((arr.a["light_kind"] == 1).any())
    rtol, share = (2e-5, 0.97) if float_trig else (1e-9, 0.999)
    close = np.isclose(hdr, rh, rtol=rtol, atol=1e-12).all(axis=2)
    same = cnt == rc
    if not cnt.any() and not rc.any():  # an empty cell
        return True, 1.0, 1.0
    # (a frame of a few pixels: one pixel whose decision a last-bit difference flips is within the criteria)
    ok = (same.mean() >= share or (~same).sum() <= 1) and (close.mean() >= share or (~close).sum() <= 1)
    return ok, float(same.mean()), float(close.mean())


def main():
    import pathtracer_amd as pt
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = pt.Device(0)
    bad = 0
    for _ in range(count):
        name, w, h, kw, pp, shape = draw(rng)
        ok, same, close = check(dev, name, w, h, kw, pp, rng.randrange(1 << 30), shape)
        bad += not ok
        print("ok  " if ok else "FAIL", name, w, h, kw, "per_pixel", pp, "max_paths %d split %d" % shape,
              "same_cnt %.4f close %.4f" % (same, close), flush=True)
    print("failures:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
