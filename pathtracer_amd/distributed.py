"""Frame rendering in rounds, optionally sharded over the GPUs of one node.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI). The
scene is replicated and the pixels are dealt out, pixel ``list[k]`` going to rank
``k % world``: the reference's tile queue (src/util/work_queue.h:11-51,
src/pathtracer/raytraced_renderer.cpp:297-302, 644-658) balances by handing 32x32
tiles to idle threads; dealing single pixels round-robin spreads the expensive regions of
a frame evenly over the ranks to begin with. Two ways of doing it (`exchange`):

"frame" (default): the frame's pixels are dealt ONCE. Every rank runs the whole adaptive
  loop on its own pixels, with no communication inside the frame, and one all-gather of the
  per-pixel sums at the end assembles the frame on every rank. Neighbouring pixels cost
  about the same, so the ranks' loads agree to a percent or so without re-dealing.
"round": the still-active pixels are dealt again after every adaptive round and one
  all-gather per round replicates the updated sums, so every rank holds the whole frame
  state and computes the same next work list. Exact balance in every round, at the price of
  a collective (and a wait for the slowest rank) per round.

Nothing in the loop waits for the GPU: the list lengths live in device memory
and the kernels read them there (see render_frame).

Results do not depend on the number of ranks: the random numbers of a path are
keyed by (seed, pixel, sample) and a pixel's samples are summed in sample order.

torch is plumbing here (device memory, streams, collectives); all rendering
goes through the C ABI.
"""
import math

import numpy as np
import torch

from . import api, cabi  # noqa: F401


class PixelSchedule:
    """Pure host logic of the sharding, testable without a GPU."""

    @staticmethod
    def shard(n_active, rank, world):
        """Indices into the active list that `rank` traces this round."""
        return range(rank, n_active, world)

    @staticmethod
    def shard_size(n_active, rank, world):
        return len(range(rank, n_active, world))

    @staticmethod
    def padded(n_active, world):
        return math.ceil(n_active / world) if n_active else 0

    @staticmethod
    def batches(n_active_bound, world, batch, samples_left, target_paths, max_batches, first=False):
        """How many batches of `batch` samples a round traces per pixel.

        A launch of fewer than `target_paths` paths leaves most of the GPU idle and
        still takes as long as its slowest ray, so when a rank's share of the active
        pixels is that small the next batches of the same pixels are traced in the
        same launches, ahead of the adaptive tests that decide whether they are
        needed (the tests then run between the batches when the results are
        summed, and what was traced past a pixel's stop is dropped). The result
        depends only on counts every rank knows, so all ranks choose alike.
        """
        paths = PixelSchedule.padded(n_active_bound, world) * batch
        if first:
            # most pixels of a typical frame stop at their first test, so the first
            # round traces ahead only when it is really small
            target_paths //= 4
        k = target_paths // max(paths, 1)
        k = min(k, max_batches, -(-samples_left // batch))
        return max(int(k), 1)


def per_pixel_default(params):
    """Per-pixel look-ahead pays when a pixel may run for many batches (measured: 1024 spp in
    batches of 32: 1 to 17 % faster, more so the smaller a rank's share; 256 spp: 3 to 10 % slower)."""
    return params.ns_aa >= 16 * max(1, params.samples_per_batch)


# Pixel index lists by (frame geometry, stride, offset, device): building one takes half a dozen
# small launches and as many host round trips, 0.7 ms in front of a frame that a rank of 8 renders
# in 25 ms. Callers must not write to the lists.
_pixel_lists = {}


def _frame_pixels(w, x0, y0, x1, y1, offset, stride, device):
    key = (w, x0, y0, x1, y1, offset, stride, str(device))
    got = _pixel_lists.get(key)
    if got is None:
        ys = torch.arange(y0, y1, device=device, dtype=torch.int32)
        xs = torch.arange(x0, x1, device=device, dtype=torch.int32)
        got = (ys[:, None] * w + xs[None, :]).reshape(-1)[offset::max(1, stride)].contiguous()
        if len(_pixel_lists) >= 16:
            _pixel_lists.clear()
        _pixel_lists[key] = got
    return got


class _CountReader:
    """List lengths on their way back to the host: a small ring of pinned words and events, reused."""

    def __init__(self, device, slots=4):
        self.on_gpu = device.type == "cuda"
        self.host = torch.empty(slots, dtype=torch.int64, pin_memory=self.on_gpu)
        self.events = [torch.cuda.Event() for _ in range(slots)] if self.on_gpu else [None] * slots
        self.device, self.k, self.slots = device, 0, slots

    def post(self, src):
        """Queues the copy of the one-element tensor `src`; returns a ticket for read()."""
        k = self.k
        self.k = (k + 1) % self.slots
        self.host[k:k + 1].copy_(src, non_blocking=True)
        if self.on_gpu:
            self.events[k].record(torch.cuda.current_stream(self.device))
        return k

    def read(self, k):
        if self.on_gpu:
            self.events[k].synchronize()
        return int(self.host[k].item())


_count_readers = {}


def _all_gather_rows(dist, recv, send, group, on_gpu):
    if on_gpu and dist.get_backend(group) == "gloo":
        # rehearsal without RCCL (several ranks sharing one GPU): stage through the host
        got = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(got, send.cpu(), group=group)
        recv.copy_(got)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)


def render_frame(dev, params, rank=0, world=1, group=None, device=None, record_rounds=None, lag=1,
                 pixel_stride=1, target_paths=16 << 20, max_batches=32, per_pixel=None, sync_below=None,
                 exchange="frame", pixel_offset=0, growth=None):
    """Renders one frame; returns (hdr[h,w,3], count[h,w]) as tensors on `device`, on every rank.

    dev: api.Device with scene and camera set. With world > 1 the default process
    group (or `group`) must be initialised with a backend that can move tensors on
    `device` ("nccl" for GPUs).

    The number of still-active pixels stays on the device. To size launches and
    the exchange the host only needs an upper bound, and uses the count of `lag`
    rounds earlier (counts never grow), copied back asynchronously: with lag = 1
    the host queues round r while the GPU still runs round r - 1, so the GPU never
    idles between rounds. lag = 0 reads every count before using it. The bound is
    the same on every rank (it is a count, not a timing), so the ranks' collectives
    always agree in size. record_rounds, if a list, receives (samples_done,
    bound, this rank's share of the bound) per round.

    sync_below: bounds at or below this many pixels are not lagged (see the loop). Default: 200000.

    growth (per_pixel only; default 4, 0: no limit): a round traces no pixel further ahead than this many
    times the samples the rounds before it traced ahead. A prediction made from one or two batches
    is rough, and a round sized for the whole remaining budget on such statistics traces (and
    drops) up to a tenth more samples than it needs; with the limit the look-ahead grows
    geometrically and the unlagged count, which alone would make a small rank's frame slower,
    makes every frame faster.

    target_paths, max_batches: see PixelSchedule.batches (max_batches = 1 keeps to
    one batch per round, the reference's own cadence).

    exchange: "frame" or "round", see the module's description (only matters with world > 1).

    per_pixel (default: per_pixel_default(params)): every pixel continues from its own sample count and is traced ahead
    only as far as its running statistics predict it will need (trace_list with
    sample_base = -1), instead of all pixels of a round moving in lockstep. Same
    frame; fewer samples traced and then dropped when a rank's rounds are small.

    pixel_stride = K (with pixel_offset = r < K) renders only pixels r, r + K, r + 2K ... of the
    rectangle (the others stay black): one GPU then does the work rank r of K would do with
    exchange = "frame", which is how the strong-scaling behaviour is rehearsed on a single GPU.
    """
    import torch.distributed as dist
    w, h = params.w, params.h
    device = device or torch.device("cuda", torch.cuda.current_device())
    on_gpu = device.type == "cuda"
    stream = torch.cuda.current_stream(device).cuda_stream if on_gpu else 0
    npix = w * h
    state = torch.zeros(npix * 6, dtype=torch.float64, device=device)
    x0, y0 = max(0, params.x0), max(0, params.y0)
    x1, y1 = min(w, params.x1), min(h, params.y1)
    if per_pixel is None:
        per_pixel = per_pixel_default(params)
    if exchange not in ("frame", "round"):
        raise ValueError("exchange must be 'frame' or 'round'")
    all_pix = _frame_pixels(w, x0, y0, x1, y1, pixel_offset, pixel_stride, device)
    n_all = all_pix.numel()
    rkey = (str(device), stream)  # one ring per stream: frames on different streams may run from different threads
    reader = _count_readers.get(rkey)
    if reader is None or reader.slots < lag + 2:
        reader = _count_readers[rkey] = _CountReader(device, slots=max(4, lag + 2))
    deal_once = world > 1 and exchange == "frame"
    if deal_once:
        # this rank's pixels for the whole frame; the loop below then runs as a single rank would
        frame_rank, frame_world = rank, world
        first = _frame_pixels(w, x0, y0, x1, y1, pixel_offset + rank * max(1, pixel_stride), world * max(1, pixel_stride), device)
        rank, world = 0, 1
    else:
        first = all_pix
    # bufs[0] is the cached, read-only list of the frame's pixels; the survivor lists alternate between the other two
    bufs = [first, torch.empty_like(first), torch.empty_like(first)]
    n0 = first.numel()
    n_dev = torch.full((3,), n0, dtype=torch.int64, device=device)  # lengths of the three lists
    pending = []  # tickets of counts not yet read
    rows_send = rows_recv = None
    if world > 1:
        m0 = PixelSchedule.padded(n0, world)
        rows_send = torch.zeros((m0, 6), dtype=torch.float64, device=device)
        rows_recv = torch.empty((world * m0, 6), dtype=torch.float64, device=device)
    dev.frame_begin(params)
    cur, done, bound = 0, 0, n0
    ahead = 0
    batch = params.samples_per_batch
    if sync_below is None:
        sync_below = 200000
    if growth is None:
        growth = 4
    # per_pixel: `done` is the least count any listed pixel can have (a pixel gains at
    # least one batch per round), which bounds the number of rounds
    while done < params.ns_aa and bound > 0:
        if done > 0:
            # the adaptive test of raytrace_pixel at i = done (per_pixel: at each pixel's own
            # count); every rank holds the whole state, so every rank computes the same survivor list
            whole = cabi.PixelList(bufs[cur].data_ptr(), n_dev[cur:].data_ptr(), 0, bound, 0, 1)
            nxt = 2 if cur == 1 else 1
            dev.filter_list(whole, -1 if per_pixel else done, state.data_ptr(), bufs[nxt].data_ptr(),
                            n_dev[nxt:].data_ptr(), stream)
            cur = nxt
            pending.append(reader.post(n_dev[cur:cur + 1]))
            # (a list that is not long is read back at once: waiting for its count idles the GPU for tens of
            # microseconds; a round sized by a count that is one round old looks less far ahead than it could,
            # which costs rounds, and one queued for pixels that have all stopped costs a round of launches)
            while len(pending) > (lag if bound > sync_below else 0):
                bound = min(bound, reader.read(pending.pop(0)))
            if bound == 0:
                break
        ns = batch * PixelSchedule.batches(bound, world, batch, params.ns_aa - done, target_paths, max_batches,
                                           first=(done == 0))
        if not per_pixel:
            ns = min(ns, params.ns_aa - done)
        elif growth and ahead > 0:
            ns = max(batch, min(ns, growth * ahead // batch * batch))
        ahead += ns
        whole = cabi.PixelList(bufs[cur].data_ptr(), n_dev[cur:].data_ptr(), 0, bound, 0, 1)
        mine = cabi.PixelList(bufs[cur].data_ptr(), n_dev[cur:].data_ptr(), 0, bound, rank, world)
        if record_rounds is not None:
            record_rounds.append((done, bound, PixelSchedule.shard_size(bound, rank, world)))
        dev.trace_list(mine, -1 if per_pixel else done, ns, state.data_ptr(), stream)
        if world > 1:
            # every rank's updated rows to every rank: pack, one all-gather, unpack
            m = PixelSchedule.padded(bound, world)
            send, recv = rows_send[:m], rows_recv[:world * m]
            dev.pack_rows(mine, state.data_ptr(), send.data_ptr(), stream)
            _all_gather_rows(dist, recv, send, group, on_gpu)
            dev.unpack_rows(whole, world, rank, recv.data_ptr(), m, state.data_ptr(), stream)
        done += batch if per_pixel else ns
    if deal_once and n_all > 0:
        # the one exchange of the frame: every rank's finished rows to every rank
        n_all_dev = torch.full((1,), n_all, dtype=torch.int64, device=device)
        whole = cabi.PixelList(all_pix.data_ptr(), n_all_dev.data_ptr(), 0, n_all, 0, 1)
        mine = cabi.PixelList(all_pix.data_ptr(), n_all_dev.data_ptr(), 0, n_all, frame_rank, frame_world)
        m = PixelSchedule.padded(n_all, frame_world)
        send = torch.zeros((m, 6), dtype=torch.float64, device=device)
        recv = torch.empty((frame_world * m, 6), dtype=torch.float64, device=device)
        dev.pack_rows(mine, state.data_ptr(), send.data_ptr(), stream)
        _all_gather_rows(dist, recv, send, group, on_gpu)
        dev.unpack_rows(whole, frame_world, frame_rank, recv.data_ptr(), m, state.data_ptr(), stream)
    hdr = torch.empty(npix * 3, dtype=torch.float64, device=device)
    cnt = torch.empty(npix, dtype=torch.int32, device=device)
    dev.frame_resolve(state.data_ptr(), hdr.data_ptr(), cnt.data_ptr(), stream)
    return hdr.view(h, w, 3), cnt.view(h, w)


def reference_schedule_cpu(counts_per_round, world):
    """For tests: per-rank pixel counts of every round given the active counts."""
    return [[PixelSchedule.shard_size(n, r, world) for r in range(world)] for n in counts_per_round]
