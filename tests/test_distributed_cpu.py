"""The multi-GPU round scheduler on CPU: two gloo ranks run
pathtracer_amd.distributed.render_frame with a stand-in device that traces through
the CPU oracle (test infrastructure), and must produce exactly the frame one rank
produces, which in turn must equal oracle_render's own per-pixel loop."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pathtracer_amd import cabi, distributed, ptar
from conftest import GOLDEN
import oracle_lib as O


class OracleDevice:
    """Implements the round-level device API on host memory with the CPU oracle."""

    def __init__(self, scene, cam):
        self.scene, self.cam = scene, cam
        self.traced = 0

    def frame_begin(self, params):
        self.p = params

    @staticmethod
    def _view(ptr, n, dtype):
        ct = {np.float64: C.c_double, np.int32: C.c_int32, np.int64: C.c_int64}[dtype]
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,))

    def _entries(self, pl):
        """The shard's pixel indices, as the device kernels enumerate them."""
        n = int(self._view(pl.d_n, 1, np.int64)[0]) if pl.d_n else int(pl.n)
        assert 0 <= n <= pl.n_max, "the host's bound must cover the device-side length"
        if n == 0:
            return np.zeros(0, np.int32)
        return self._view(pl.d_pixels, n, np.int32)[pl.offset::pl.stride].copy()

    def trace_list(self, pl, sample_base, n_samples, d_state, stream=0):
        p = self.p
        pix = self._entries(pl)
        st = self._view(d_state, p.w * p.h * 6, np.float64).reshape(-1, 6)
        rgb = (C.c_double * 3)()
        for px in pix:
            x, y = int(px) % p.w, int(px) // p.w
            kept = 0
            # sample_base < 0: the pixel goes on from its own count (this stand-in plans
            # no further than the device's upper limit; the frame does not depend on the plan)
            first = sample_base if sample_base >= 0 else int(st[px, 5])
            for s in range(first, min(first + n_samples, p.ns_aa if sample_base < 0 else first + n_samples)):
                if s > first and s % p.samples_per_batch == 0 and not self._continues(st[px], s, p.max_tolerance):
                    break
                kept += 1
                O.lib().oracle_sample_radiance(C.byref(self.scene.desc), C.byref(self.cam), C.byref(p), x, y, s, rgb)
                # Vector3D::illum: double arithmetic, float result
                xl = float(np.float32(float(np.float32(0.2126)) * rgb[0] + float(np.float32(0.7152)) * rgb[1]
                                      + float(np.float32(0.0722)) * rgb[2]))
                st[px, 0] += xl
                st[px, 1] += xl * xl
                st[px, 2] += rgb[0]
                st[px, 3] += rgb[1]
                st[px, 4] += rgb[2]
            st[px, 5] += kept
            self.traced += kept

    @staticmethod
    def _continues(row, i, tol):
        s1, s2 = row[0], row[1]
        mean = s1 * (1.0 / i)
        var = (1.0 / (i - 1)) * (s2 - s1 * s1 * (1.0 / i))
        with np.errstate(invalid="ignore"):
            conf = 1.96 * np.sqrt(np.float64(var * (1.0 / i)))
        return not (conf <= tol * mean)

    def filter_list(self, pl, n_done, d_state, d_out, d_n_out, stream=0):
        p = self.p
        pin = self._entries(pl)
        st = self._view(d_state, p.w * p.h * 6, np.float64).reshape(-1, 6)
        out = self._view(d_out, max(pin.size, 1), np.int32)
        k = 0
        i = n_done
        for px in pin:
            if i >= 0:
                keep = st[px, 5] >= i and self._continues(st[px], i, p.max_tolerance)
            else:  # judged at the pixel's own count
                n = int(st[px, 5])
                keep = n < p.ns_aa and (n == 0 or n % p.samples_per_batch != 0 or self._continues(st[px], n, p.max_tolerance))
            if keep:
                out[k] = px
                k += 1
        self._view(d_n_out, 1, np.int64)[0] = k

    def pack_rows(self, pl, d_state, d_rows, stream=0):
        p = self.p
        pix = self._entries(pl)
        st = self._view(d_state, p.w * p.h * 6, np.float64).reshape(-1, 6)
        if pix.size:
            self._view(d_rows, pix.size * 6, np.float64).reshape(-1, 6)[:] = st[pix]

    def unpack_rows(self, pl, world, skip_rank, d_rows, rows_per_rank, d_state, stream=0):
        p = self.p
        pix = self._entries(pl)
        st = self._view(d_state, p.w * p.h * 6, np.float64).reshape(-1, 6)
        rows = self._view(d_rows, world * rows_per_rank * 6, np.float64).reshape(world, rows_per_rank, 6)
        for i, px in enumerate(pix):
            if i % world != skip_rank:
                st[px] = rows[i % world, i // world]

    def frame_resolve(self, d_state, d_hdr, d_count, stream=0):
        p = self.p
        st = self._view(d_state, p.w * p.h * 6, np.float64).reshape(-1, 6)
        hdr = self._view(d_hdr, p.w * p.h * 3, np.float64).reshape(-1, 3)
        cnt = self._view(d_count, p.w * p.h, np.int32)
        for i in range(p.w * p.h):
            nn = st[i, 5]
            hdr[i] = st[i, 2:5] * (1.0 / nn) if nn > 0 else 0.0
            cnt[i] = int(nn)


def _setup():
    a = ptar.load(os.path.join(GOLDEN, "scene_CBspheres_lambertian.ptar"))
    sc = cabi.SceneArrays.from_ptar(a)
    cam26 = a["camera"].copy()
    return sc, cabi.camera_from_dump(cam26)


W, H = 24, 18
KW = dict(ns_aa=48, ns_area_light=2, max_ray_depth=3, samples_per_batch=16, max_tolerance=0.3, seed=9)


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception as e:  # surface the failure instead of leaving the parent waiting
        q.put((rank, repr(e), None, None, None))
        raise


def _worker_body(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, cam = _setup()
    dev = OracleDevice(sc, cam)
    p = cabi.make_params(W, H, **KW)
    rounds = []
    cpu = torch.device("cpu")
    hdr, cnt = distributed.render_frame(dev, p, rank=rank, world=world, device=cpu, record_rounds=rounds, lag=0,
                                        max_batches=1, per_pixel=False, exchange="round")
    traced = dev.traced
    # the pipelined schedule (bounds one round old) must give the same frame, and so
    # must rounds that trace several batches ahead of their adaptive tests
    for kw in (dict(lag=1, max_batches=1, sync_below=0), dict(lag=1), dict(lag=0, max_batches=2),
               dict(lag=1, per_pixel=True, sync_below=0), dict(lag=1, per_pixel=True),
               dict(lag=0, max_batches=2, per_pixel=True)):
        kw.setdefault("per_pixel", False)
        for exchange in ("round", "frame"):
            # "frame": the pixels dealt once, no collective inside the frame, one all-gather at its end
            hdr_l, cnt_l = distributed.render_frame(dev, p, rank=rank, world=world, device=cpu, exchange=exchange, **kw)
            assert torch.equal(hdr_l, hdr) and torch.equal(cnt_l, cnt), "%r, exchange %s: frame differs" % (kw, exchange)
    q.put((rank, hdr.numpy().copy(), cnt.numpy().copy(), rounds, traced))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(600)
def test_two_ranks_equal_one_rank_bit_for_bit():
    sc, cam = _setup()
    p = cabi.make_params(W, H, **KW)
    one = OracleDevice(sc, cam)
    rounds1 = []
    hdr1, cnt1 = distributed.render_frame(one, p, device=torch.device("cpu"), record_rounds=rounds1, lag=0,
                                          max_batches=1, per_pixel=False)
    hdr1, cnt1 = hdr1.numpy(), cnt1.numpy()
    rounds_l = []
    hdr_l, cnt_l = distributed.render_frame(one, p, device=torch.device("cpu"), record_rounds=rounds_l, lag=1,
                                            max_batches=1, sync_below=0, per_pixel=False)  # (every bound lagged, however short the list)
    assert np.array_equal(hdr_l.numpy(), hdr1) and np.array_equal(cnt_l.numpy(), cnt1)
    # several batches per round, traced ahead of the adaptive tests: same frame, fewer rounds
    rounds_s = []
    hdr_s, cnt_s = distributed.render_frame(one, p, device=torch.device("cpu"), record_rounds=rounds_s, lag=0, per_pixel=False)
    assert np.array_equal(hdr_s.numpy(), hdr1) and np.array_equal(cnt_s.numpy(), cnt1)
    assert len(rounds_s) < len(rounds1)
    # with lag 1 the host's bound of round r is the exact count of round r - 1
    assert [b for _, b, _ in rounds_l[2:]] == [b for _, b, _ in rounds1[1:len(rounds_l) - 1]]
    # the round loop reproduces raytrace_pixel's adaptive loop exactly
    rh, rc, st = O.render(sc, cam, p, threads=2)
    assert np.array_equal(cnt1, rc) and np.array_equal(hdr1, rh)
    assert 16 * W * H < cnt1.sum() < 48 * W * H  # the adaptive test really stopped some pixels

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=500) for _ in procs]
    for r in res:
        assert not isinstance(r[1], str), "rank %d failed: %s" % (r[0], r[1])
    res = sorted(res, key=lambda t: t[0])
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    for rank, hdr, cnt, rounds, traced in res:
        assert np.array_equal(hdr, hdr1) and np.array_equal(cnt, cnt1), "rank %d frame differs" % rank
        assert [r[:2] for r in rounds] == [r[:2] for r in rounds1]
    # the work really was split, evenly, every round
    t0, t1 = res[0][4], res[1][4]
    assert t0 + t1 == int(cnt1.sum()) and abs(t0 - t1) <= 16 * len(rounds1)
    for r0, r1 in zip(res[0][3], res[1][3]):
        assert r0[2] + r1[2] == r0[1] and abs(r0[2] - r1[2]) <= 1


def test_batches_per_round_policy():
    B = distributed.PixelSchedule.batches
    assert B(480000, 1, 32, 1024, 3 << 20, 16) == 1          # a full frame fills the GPU
    assert B(38000, 1, 32, 700, 3 << 20, 16) == 2            # late rounds of the benchmark frame
    assert B(38000, 8, 32, 700, 3 << 20, 16) == 16           # the same on 8 GPUs: capped
    assert B(38000, 8, 32, 96, 3 << 20, 16) == 3             # never past the last sample
    assert B(100, 1, 32, 1024, 3 << 20, 1) == 1              # the reference's cadence on request
    assert B(0, 4, 32, 64, 3 << 20, 16) >= 1
    assert B(60000, 1, 32, 1024, 4 << 20, 32, first=True) == 1 and B(60000, 1, 32, 1024, 4 << 20, 32) == 2


def test_pixel_schedule_covers_every_pixel_once():
    for n in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            seen = sorted(i for r in range(world) for i in distributed.PixelSchedule.shard(n, r, world))
            assert seen == list(range(n))
            sizes = [distributed.PixelSchedule.shard_size(n, r, world) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1 and max(sizes, default=0) <= distributed.PixelSchedule.padded(n, world)
