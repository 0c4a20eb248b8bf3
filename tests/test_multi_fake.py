"""The several-GPU driver of pathtracer_amd/csrc/multi_gpu.cpp (one host thread per GPU, pixels dealt round-robin,
one point-to-point gather on the first GPU, a meeting before the exchange) EXECUTED with more than one rank.

Real RCCL refuses two ranks on one device and the test box has one GPU, so this runs the TEST build of the same
source file (tests/_build/libptamd_multi_fake.so, `make -C pathtracer_amd/csrc test_multi`): its RCCL calls go to the
in-process stand-in of tests/fake_rccl/fake_rccl.cpp (same-device copies ordered by events), one physical GPU may be
listed as several logical ranks, and one rank's part of a frame can be made to fail. Everything else (the deal, the
per-rank adaptive loops on their own streams and contexts, pack / send / receive / unpack, the frame resolve) is the
product's code path. What this cannot show is RCCL itself moving bytes between devices."""
import ctypes as C
import os

import numpy as np
import pytest

import pathtracer_amd as pt
from pathtracer_amd import api, cabi
from conftest import ROOT, scene_path

FAKE = os.path.join(ROOT, "tests", "_build", "libptamd_multi_fake.so")


def _fake():
    if not os.path.exists(FAKE):
        pytest.fail("%s is missing: build it with `make -C pathtracer_amd/csrc test_multi`" % FAKE)
    api.lib()  # libptamd.so first (the test library links it)
    M = api._declare_multi(C.CDLL(FAKE))
    M.ptamd_multi_test_fail_rank.argtypes = [C.c_int32]
    M.ptamd_multi_test_fail_rank.restype = None
    M.ptamd_multi_test_rendezvous.argtypes = [C.c_int32] * 4
    M.ptamd_multi_test_rendezvous.restype = C.c_int32
    return M


def test_test_library_exports_the_driver_and_its_hooks():
    M = _fake()
    for name in api.EXPORTS_MULTI + ("ptamd_multi_test_fail_rank", "ptamd_multi_test_rendezvous"):
        getattr(M, name)
    # the product library has the driver and none of the hooks
    import subprocess
    syms = subprocess.run(["nm", "-D", os.path.join(ROOT, "pathtracer_amd", "libptamd_multi.so")], capture_output=True, text=True).stdout
    assert "ptamd_multi_render" in syms and "ptamd_multi_test" not in syms and "ptfake" not in syms


@pytest.mark.parametrize("n,generations,failing,failing_generation", [(1, 3, -1, 0), (2, 4, -1, 0), (8, 5, 3, 2), (3, 3, 0, 0), (5, 4, 4, 3)])
def test_rendezvous_releases_everybody_and_forgets_a_failure(n, generations, failing, failing_generation):
    """Rendezvous::arrive (the meeting before the exchange): every thread of a meeting gets that meeting's verdict,
    one thread's failure is everybody's, and the next meeting starts clean."""
    assert _fake().ptamd_multi_test_rendezvous(n, generations, failing, failing_generation) == 0


def test_shard_arithmetic():
    M = _fake()
    for n in (0, 1, 7, 8, 9, 4800):
        for world in (1, 2, 3, 8):
            sizes = [M.ptamd_multi_shard_size(n, r, world) for r in range(world)]
            assert sum(sizes) == n and max(sizes) == M.ptamd_multi_rows_per_rank(n, world)
            assert sizes == [len(range(r, n, world)) for r in range(world)]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 3, 8])
def test_logical_ranks_on_one_gpu_equal_single_gpu_render(device, n):
    """ptamd_multi_render with n logical ranks == ptamd_render, bit for bit (frame and sample counts), on a full
    frame and on a cell whose pixel count n does not divide; rank 0's and the last rank's shares add up."""
    M = _fake()
    sc = pt.Scene(scene_path("CBspheres_lambertian"))
    w, h = 96, 72
    cam = sc.camera(w, h)
    device.upload(sc)
    device.set_camera(cam)
    md = pt.MultiDevice([0] * n, library=M)
    try:
        md.upload(sc)
        md.set_camera(cam)
        md.set_profiling(counters=True)
        # (the last: 96 batches per pixel, so every rank's loop runs several rounds of per-pixel look-ahead)
        for kw in (dict(), dict(rect=(5, 7, 64, 50)), dict(samples_per_batch=1, max_tolerance=0.3)):
            p = cabi.make_params(w, h, ns_aa=96, max_ray_depth=4, ns_area_light=2, seed=11, **kw)
            a, ac = device.render(p)
            b, bc = md.render(p)
            assert np.array_equal(ac, bc)
            assert a.tobytes() == b.tobytes()
            parts = [md.stats(r) for r in range(n)]
            assert sum(s.samples for s in parts) == int(ac.sum())
            assert all(s.prim_tests > 0 for s in parts)  # the counters were switched on for every rank
    finally:
        md.close()


@pytest.mark.gpu
def test_a_failing_rank_releases_the_others_and_the_next_frame_works(device):
    M = _fake()
    sc = pt.Scene(scene_path("CBspheres_lambertian"))
    w, h = 64, 48
    cam = sc.camera(w, h)
    device.upload(sc)
    device.set_camera(cam)
    md = pt.MultiDevice([0, 0, 0, 0], library=M)
    try:
        md.upload(sc)
        md.set_camera(cam)
        p = cabi.make_params(w, h, ns_aa=64, max_ray_depth=3, ns_area_light=2, seed=3)
        for bad in (2, 0):
            M.ptamd_multi_test_fail_rank(bad)
            with pytest.raises(api.PtamdError) as e:
                md.render(p)  # returns (every thread leaves the meeting) instead of hanging in a send or receive
            assert "failure injected" in str(e.value) or "another GPU" in str(e.value)
            a, ac = device.render(p)
            b, bc = md.render(p)
            assert np.array_equal(ac, bc) and a.tobytes() == b.tobytes()
    finally:
        M.ptamd_multi_test_fail_rank(-1)
        md.close()
