"""GPU tests of the round-level C ABI that multi-GPU drivers use: lists whose
length lives on the device, sharded tracing, and the pack / unpack exchange.

Several ranks are emulated on ONE GPU: every virtual rank has its own copy of the
per-pixel state, traces its shard of each round's list, and the ranks' packed
blocks are concatenated exactly as an all-gather would lay them out."""
import os
import numpy as np
import pytest
import torch

import pathtracer_amd as pt
from pathtracer_amd import cabi, distributed
from conftest import scene_path

pytestmark = pytest.mark.gpu

W, H = 96, 72
KW = dict(ns_aa=96, max_ray_depth=4, ns_area_light=2, samples_per_batch=16, max_tolerance=0.15, seed=5)


def _setup(device):
    sc = pt.Scene(scene_path("CBspheres_lambertian"))
    device.upload(sc)
    device.set_camera(sc.camera(W, H))
    return cabi.make_params(W, H, **KW)


def test_render_frame_pipelined_equals_synchronous_equals_render(device):
    p = _setup(device)
    cuda = torch.device("cuda", 0)
    # the reference's cadence: one batch per round, every count read before it is used
    base_rounds = []
    hdr_b, cnt_b = distributed.render_frame(device, p, device=cuda, lag=0, max_batches=1, record_rounds=base_rounds,
                                            per_pixel=False)
    hdr_b, cnt_b = hdr_b.cpu().numpy(), cnt_b.cpu().numpy()
    assert W * H * 16 < cnt_b.sum() < W * H * 96, "the adaptive test must stop some pixels early"
    assert len(base_rounds) == 6
    # ptamd_render and every other schedule (late bounds, batches traced ahead of their tests)
    hdr_r, cnt_r = device.render(p)
    assert np.array_equal(hdr_r, hdr_b) and np.array_equal(cnt_r, cnt_b)
    assert device.stats().samples == int(cnt_b.sum()), "only the samples that were kept count"
    for kw in (dict(lag=1, max_batches=1, sync_below=0), dict(lag=2, max_batches=1, sync_below=0), dict(lag=1),
               dict(lag=0, max_batches=2),
               dict(lag=1, max_batches=4, target_paths=200000),
               # every pixel from its own count, traced ahead as far as its statistics predict
               dict(lag=1, per_pixel=True, sync_below=0), dict(lag=0, max_batches=2, per_pixel=True),
               dict(lag=2, max_batches=6, target_paths=400000, per_pixel=True),
               # how fast the look-ahead may grow from round to round, and when the survivor count is read at once
               dict(per_pixel=True, growth=1), dict(per_pixel=True, growth=0, sync_below=10 ** 9),
               dict(per_pixel=True, growth=16, lag=2, target_paths=400000, sync_below=0)):
        rounds = []
        hdr, cnt = distributed.render_frame(device, p, device=cuda, record_rounds=rounds, **kw)
        assert np.array_equal(hdr.cpu().numpy(), hdr_b) and np.array_equal(cnt.cpu().numpy(), cnt_b), repr(kw)
        bounds = [b for _, b, _ in rounds]
        assert bounds == sorted(bounds, reverse=True)
        st = device.stats()
        assert st.samples == int(cnt_b.sum())
        assert st.rays >= st.samples, "slots a round leaves unused are not rays"


def _virtual_ranks_frame(dev, p, world, lag, max_batches=1, target_paths=3 << 20, per_pixel=False):
    """render_frame's loop with `world` ranks played in turn on one GPU."""
    cuda = torch.device("cuda", 0)
    npix = p.w * p.h
    states = [torch.zeros(npix * 6, dtype=torch.float64, device=cuda) for _ in range(world)]
    bufs = [torch.arange(npix, dtype=torch.int32, device=cuda), torch.empty(npix, dtype=torch.int32, device=cuda)]
    n_dev = torch.full((2,), npix, dtype=torch.int64, device=cuda)
    m0 = distributed.PixelSchedule.padded(npix, world)
    recv = torch.full((world * m0, 6), float("nan"), dtype=torch.float64, device=cuda)
    dev.frame_begin(p)
    cur, done, bound, history = 0, 0, npix, []
    while done < p.ns_aa and bound > 0:
        if done > 0:
            whole = cabi.PixelList(bufs[cur].data_ptr(), n_dev[cur:].data_ptr(), 0, bound, 0, 1)
            # every rank holds the same state, so one filter stands for all of them
            dev.filter_list(whole, -1 if per_pixel else done, states[0].data_ptr(), bufs[cur ^ 1].data_ptr(),
                            n_dev[cur ^ 1:].data_ptr())
            cur ^= 1
            history.append(int(n_dev[cur].item()))
            if len(history) > lag:
                bound = min(bound, history[len(history) - 1 - lag])
            if bound == 0:
                break
        k = distributed.PixelSchedule.batches(bound, world, p.samples_per_batch, p.ns_aa - done, target_paths, max_batches)
        ns = k * p.samples_per_batch if per_pixel else min(k * p.samples_per_batch, p.ns_aa - done)
        whole = cabi.PixelList(bufs[cur].data_ptr(), n_dev[cur:].data_ptr(), 0, bound, 0, 1)
        m = distributed.PixelSchedule.padded(bound, world)
        blocks = recv[:world * m]
        for r in range(world):
            mine = cabi.PixelList(bufs[cur].data_ptr(), n_dev[cur:].data_ptr(), 0, bound, r, world)
            dev.trace_list(mine, -1 if per_pixel else done, ns, states[r].data_ptr())
            dev.pack_rows(mine, states[r].data_ptr(), blocks[r * m:].data_ptr())
        for r in range(world):
            dev.unpack_rows(whole, world, r, blocks.data_ptr(), m, states[r].data_ptr())
        done += p.samples_per_batch if per_pixel else ns
    torch.cuda.synchronize()
    for r in range(1, world):
        assert torch.equal(states[r], states[0]), "rank %d holds a different frame state" % r
    hdr = torch.empty(npix * 3, dtype=torch.float64, device=cuda)
    cnt = torch.empty(npix, dtype=torch.int32, device=cuda)
    dev.frame_resolve(states[0].data_ptr(), hdr.data_ptr(), cnt.data_ptr())
    return hdr.view(p.h, p.w, 3).cpu().numpy(), cnt.view(p.h, p.w).cpu().numpy()


@pytest.mark.parametrize("world,lag,max_batches,per_pixel", [(2, 0, 1, False), (3, 1, 1, False), (8, 1, 1, False),
                                                             (2, 1, 2, False), (8, 0, 16, False), (2, 1, 4, True),
                                                             (8, 1, 16, True)])
def test_virtual_ranks_reproduce_the_single_gpu_frame(device, world, lag, max_batches, per_pixel):
    """Sharding by (offset, stride) + pack / all-gather layout / unpack: bit-identical
    to one GPU for any number of ranks (RNG keys are per pixel and sample)."""
    p = _setup(device)
    hdr_r, cnt_r = device.render(p)
    hdr, cnt = _virtual_ranks_frame(device, p, world, lag, max_batches, per_pixel=per_pixel)
    assert np.array_equal(cnt, cnt_r) and np.array_equal(hdr, hdr_r)


@pytest.mark.parametrize("world,kw", [(2, dict(per_pixel=False)), (3, dict(per_pixel=True)),
                                      (8, dict(per_pixel=False, max_batches=16)), (8, dict(per_pixel=True, lag=0))])
def test_pixels_dealt_once_per_frame_reproduce_the_single_gpu_frame(device, world, kw):
    """exchange = "frame": rank r renders pixels r, r + world, ... through the whole adaptive loop
    on its own; the ranks' frames (played in turn on one GPU, the other pixels black) add up
    to the single-GPU frame, bit for bit."""
    p = _setup(device)
    cuda = torch.device("cuda", 0)
    hdr_r, cnt_r = device.render(p)
    hdr = np.zeros_like(hdr_r)
    cnt = np.zeros_like(cnt_r)
    for r in range(world):
        h_, c_ = distributed.render_frame(device, p, device=cuda, pixel_stride=world, pixel_offset=r, **kw)
        h_, c_ = h_.cpu().numpy(), c_.cpu().numpy()
        assert not (cnt[c_ > 0]).any(), "two ranks rendered the same pixel"
        hdr += h_
        cnt += c_
    assert np.array_equal(cnt, cnt_r) and np.array_equal(hdr, hdr_r)


def test_list_argument_errors(device):
    p = _setup(device)
    device.frame_begin(p)
    cuda = torch.device("cuda", 0)
    buf = torch.zeros(16, dtype=torch.int32, device=cuda)
    state = torch.zeros(W * H * 6, dtype=torch.float64, device=cuda)
    with pytest.raises(pt.PtamdError):  # stride 0
        device.trace_list(cabi.PixelList(buf.data_ptr(), 0, 16, 16, 0, 0), 0, 4, state.data_ptr())
    with pytest.raises(pt.PtamdError):  # host length above its own bound
        device.trace_list(cabi.PixelList(buf.data_ptr(), 0, 17, 16, 0, 1), 0, 4, state.data_ptr())
    with pytest.raises(pt.PtamdError):  # unpack needs the whole list
        device.unpack_rows(cabi.PixelList(buf.data_ptr(), 0, 16, 16, 1, 2), 2, 0, state.data_ptr(), 8, state.data_ptr())
    # an empty shard is not an error
    device.trace_list(cabi.PixelList(buf.data_ptr(), 0, 1, 1, 3, 4), 0, 4, state.data_ptr())


# (both tests below look at the default launch policy; a run that overrides it through the
# environment, as the A/B scripts do, has nothing to check here)
_POLICY_OVERRIDDEN = any(k in os.environ for k in ("PTAMD_FUSE_BELOW", "PTAMD_TRACE_TWO", "PTAMD_FUSE_RESOLVE"))


@pytest.mark.skipif(_POLICY_OVERRIDDEN, reason="launch policy overridden in the environment")
def test_fused_and_separate_traversal_launches_give_the_same_frame(device, monkeypatch):
    """k_trace2 (bounce rays of depth d + 1 and shadow rays of depth d in one launch) against one
    launch per kind (PTAMD_TRACE_TWO=0, read when a context is created): which lane traces a ray,
    and in which launch, must not show in the result."""
    p = cabi.make_params(W, H, ns_aa=64, max_ray_depth=5, ns_area_light=4, seed=11)
    sc = pt.Scene(scene_path("CBbunny"))
    device.upload(sc)
    device.set_camera(sc.camera(W, H))
    hdr_f, cnt_f = device.render(p)
    st_f = device.stats()
    assert st_f.launches[cabi.K_TRACE2] > 0 and st_f.two_ext_rays > 0 and st_f.two_shadow_rays > 0
    monkeypatch.setenv("PTAMD_TRACE_TWO", "0")
    other = pt.Device(0)
    try:
        other.upload(sc)
        other.set_camera(sc.camera(W, H))
        hdr_s, cnt_s = other.render(p)
        st_s = other.stats()
    finally:
        other.close()
    assert st_s.launches[cabi.K_TRACE2] == 0 and st_s.two_ext_rays == 0
    assert np.array_equal(cnt_f, cnt_s) and np.array_equal(hdr_f, hdr_s)
    assert (st_f.rays, st_f.shadow_rays, st_f.samples) == (st_s.rays, st_s.shadow_rays, st_s.samples)
    # hemisphere lighting: the second list's rays are closest-hit walks too
    ph = cabi.make_params(W, H, ns_aa=32, max_ray_depth=3, ns_area_light=2, seed=3, hemisphere=True)
    hdr_f, cnt_f = device.render(ph)
    monkeypatch.setenv("PTAMD_TRACE_TWO", "0")
    other = pt.Device(0)
    try:
        other.upload(sc)
        other.set_camera(sc.camera(W, H))
        hdr_s, cnt_s = other.render(ph)
    finally:
        other.close()
    assert np.array_equal(cnt_f, cnt_s) and np.array_equal(hdr_f, hdr_s)


@pytest.mark.skipif(_POLICY_OVERRIDDEN, reason="launch policy overridden in the environment")
def test_light_samples_settled_beside_the_next_depth_give_the_same_frame(device, monkeypatch):
    """k_light_resolve (a depth's light samples summed by workgroups of the launch that draws the next
    depth's samples, with the records the two share kept per depth) against k_light and k_resolve one
    after the other (PTAMD_FUSE_RESOLVE=0, read when a context is created): same frame, same counts, and
    the modes that cannot use it (bounces not accumulated, specular materials, hemisphere sampling) still
    launch k_resolve once per depth."""
    sc = pt.Scene(scene_path("CBbunny"))
    device.upload(sc)
    device.set_camera(sc.camera(W, H))
    monkeypatch.setenv("PTAMD_FUSE_RESOLVE", "0")
    other = pt.Device(0)
    try:
        other.upload(sc)
        other.set_camera(sc.camera(W, H))
        for kw in (dict(ns_aa=64, max_ray_depth=5, ns_area_light=4, seed=11), dict(ns_aa=32, max_ray_depth=2, ns_area_light=3, seed=2),
                   dict(ns_aa=32, max_ray_depth=7, ns_area_light=1, seed=4)):
            p = cabi.make_params(W, H, **kw)
            hdr_f, cnt_f = device.render(p)
            st_f = device.stats()
            hdr_s, cnt_s = other.render(p)
            st_s = other.stats()
            assert np.array_equal(cnt_f, cnt_s) and np.array_equal(hdr_f, hdr_s), repr(kw)
            assert (st_f.rays, st_f.shadow_rays, st_f.samples, st_f.path_vertices) == (st_s.rays, st_s.shadow_rays, st_s.samples, st_s.path_vertices)
            # one k_resolve per round is left (the last depth's), the others ride with k_light
            assert st_f.launches[cabi.K_RESOLVE] * kw["max_ray_depth"] == st_s.launches[cabi.K_RESOLVE] > 0, repr(kw)
            assert st_f.launches[cabi.K_LIGHT] == st_s.launches[cabi.K_LIGHT]
        for kw in (dict(accumulate_bounces=False), dict(hemisphere=True), dict(specular=True), dict(max_ray_depth=1)):
            p = cabi.make_params(W, H, **{**dict(ns_aa=32, max_ray_depth=3, ns_area_light=2, seed=3), **kw})
            hdr_f, cnt_f = device.render(p)
            st_f = device.stats()
            hdr_s, cnt_s = other.render(p)
            assert np.array_equal(cnt_f, cnt_s) and np.array_equal(hdr_f, hdr_s), repr(kw)
            assert st_f.launches[cabi.K_RESOLVE] == other.stats().launches[cabi.K_RESOLVE], repr(kw)
    finally:
        other.close()


def test_one_call_frame_is_the_same_under_either_look_ahead_policy(device, monkeypatch):
    """ptamd_render's own round loop: every pixel traced ahead from its own count as far as its statistics
    predict (the choice for frames of 16 batches per pixel or more) against all pixels in lockstep
    (PTAMD_PER_PIXEL, read by every call): same frame and counts bit for bit, and the per-pixel policy
    traces fewer samples that are then dropped. Also with a sample budget that is no multiple of the batch."""
    w, h = 256, 192  # enough pixels for rounds that do not trace every sample at once
    sc = pt.Scene(scene_path("CBbunny"))
    device.upload(sc)
    device.set_camera(sc.camera(w, h))
    for ns_aa, batch in ((512, 16), (530, 32)):
        p = cabi.make_params(w, h, ns_aa=ns_aa, max_ray_depth=3, ns_area_light=2, seed=5, samples_per_batch=batch)
        monkeypatch.delenv("PTAMD_PER_PIXEL", raising=False)
        hdr_d, cnt_d = device.render(p)
        rays_d = device.stats().rays
        monkeypatch.setenv("PTAMD_PER_PIXEL", "0")
        hdr_l, cnt_l = device.render(p)
        rays_l = device.stats().rays
        monkeypatch.setenv("PTAMD_PER_PIXEL", "1")
        hdr_p, cnt_p = device.render(p)
        rays_p = device.stats().rays
        assert np.array_equal(cnt_d, cnt_l) and np.array_equal(hdr_d, hdr_l)
        assert np.array_equal(cnt_d, cnt_p) and np.array_equal(hdr_d, hdr_p)
        assert rays_d == rays_p, "16 batches per pixel or more: per-pixel is the default"
        assert rays_p < rays_l
        assert cnt_d.min() < ns_aa and cnt_d.max() == ns_aa, "the frame must have pixels that stop early and pixels that do not"


@pytest.mark.skipif(_POLICY_OVERRIDDEN, reason="launch policy overridden in the environment")
def test_headline_frame_is_the_same_under_every_launch_policy(monkeypatch):
    """The benchmark's own frame (the 100k-triangle stand-in, 800x600, 1024 samples, depth 5, four light samples) at
    full size, where rounds fill the buffers and long shadow-ray lists get launches of their own: bit-identical with
    4 Mi and 1.5 Mi paths per launch instead of 16 Mi, with all pixels in lockstep, with k_light and k_resolve apart,
    with bounce and shadow rays never or always in one launch, and with three times the look-ahead margin."""
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_standin_scene
    make_standin_scene.ensure("dragon")
    sc = pt.Scene(make_standin_scene.standin_path("dragon"))
    p = cabi.make_params(800, 600, ns_aa=1024, max_ray_depth=5, ns_area_light=4, seed=42)

    def frame(dev, max_paths=0):
        dev.upload(sc)
        dev.set_camera(sc.camera(800, 600))
        dev.set_max_paths(max_paths)
        hdr, cnt = dev.render(p)
        dev.set_max_paths(0)
        return hashlib.sha256(hdr.tobytes() + cnt.tobytes()).hexdigest(), int(cnt.sum())

    for k in ("PTAMD_PER_PIXEL", "PTAMD_PLAN_MARGIN"):
        monkeypatch.delenv(k, raising=False)
    dev = pt.Device(0)
    try:
        want = frame(dev)
        assert 0.15 * 800 * 600 * 1024 < want[1] < 0.5 * 800 * 600 * 1024
        assert frame(dev, 4 << 20) == want
        assert frame(dev, 3 << 19) == want
        monkeypatch.setenv("PTAMD_PER_PIXEL", "0")
        assert frame(dev) == want
        monkeypatch.delenv("PTAMD_PER_PIXEL")
    finally:
        dev.close()
    for k, v in (("PTAMD_FUSE_RESOLVE", "0"), ("PTAMD_TRACE_TWO", "0"), ("PTAMD_FUSE_BELOW", str(1 << 30)), ("PTAMD_PLAN_MARGIN", "3")):
        monkeypatch.setenv(k, v)  # read when a context is created
        other = pt.Device(0)
        try:
            assert frame(other) == want, k
        finally:
            other.close()
            monkeypatch.delenv(k)


@pytest.mark.skipif(_POLICY_OVERRIDDEN, reason="launch policy overridden in the environment")
def test_event_classes_restrict_what_is_timed(device):
    p = _setup(device)
    device.set_profiling(events=True, counters=False)
    try:
        device.render(p)
        st = device.stats()
        assert all(st.kernel_ms[k] > 0 for k in range(12) if st.launches[k]), "every launched class is timed"
        device.set_event_classes(1 << cabi.K_TRACE2)
        device.render(p)
        st = device.stats()
        assert st.kernel_ms[cabi.K_TRACE2] > 0
        assert all(st.kernel_ms[k] == 0 for k in range(12) if k != cabi.K_TRACE2)
        assert st.launches[2] > 0, "launches are counted whether timed or not"
    finally:
        device.set_event_classes(0xffffffff)
        device.set_profiling(events=False, counters=False)
