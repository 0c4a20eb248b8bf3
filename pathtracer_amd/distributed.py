"""Frame rendering in rounds, optionally sharded over the GPUs of one node.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI). The
scene is replicated; every round the still-active pixels are dealt out again,
pixel ``active[k]`` going to rank ``k % world``. That is the dynamic load
balancing: the reference's tile queue (src/util/work_queue.h:11-51,
src/pathtracer/raytraced_renderer.cpp:297-302, 644-658) balances by handing
tiles to idle threads; here the work list itself is re-dealt at pixel
granularity after every adaptive-sampling round, so expensive pixels are spread
evenly and no rank idles while another finishes a hard tile.

After a round each rank owns the updated sums of its pixels; one all-gather
replicates them, so every rank holds the whole frame state (s1, s2, sum, n per
pixel) and computes the same next work list. The frame buffer is therefore
assembled incrementally by RCCL and needs no final gather.

Results do not depend on the number of ranks: the random numbers of a path are
keyed by (seed, pixel, sample) and a pixel's samples are summed in sample order.

torch is plumbing here (device memory, streams, collectives); all rendering
goes through the C ABI.
"""
import math

import numpy as np
import torch

from . import api


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


def _exchange(state6, active, mine_idx, world, group):
    """All-gather of the rows each rank updated; `state6` is [npix, 6]."""
    import torch.distributed as dist
    n = active.numel()
    m = PixelSchedule.padded(n, world)
    send = torch.zeros((m, 6), dtype=state6.dtype, device=state6.device)
    k = mine_idx.numel()
    if k:
        send[:k] = state6.index_select(0, mine_idx.long())
    # output = the ranks' blocks concatenated along dim 0 (the layout both RCCL and
    # gloo accept for all_gather_into_tensor)
    recv = torch.empty((world * m, 6), dtype=state6.dtype, device=state6.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, m, 6)
    # rank r's rows are active[r::world]
    for r in range(world):
        idx = active[r::world].long()
        if idx.numel():
            state6.index_copy_(0, idx, recv[r, :idx.numel()])


def render_frame(dev, params, rank=0, world=1, group=None, device=None, record_rounds=None):
    """Renders one frame; returns (hdr[h,w,3], count[h,w]) as CUDA tensors on every rank.

    dev: api.Device with scene and camera set. With world > 1 the default process
    group (or `group`) must be initialised with a backend that can move tensors on
    `device` ("nccl" for GPUs).
    """
    w, h = params.w, params.h
    device = device or torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
    npix = w * h
    state = torch.zeros(npix * 6, dtype=torch.float64, device=device)
    state6 = state.view(npix, 6)
    x0, y0 = max(0, params.x0), max(0, params.y0)
    x1, y1 = min(w, params.x1), min(h, params.y1)
    ys = torch.arange(y0, y1, device=device, dtype=torch.int32)
    xs = torch.arange(x0, x1, device=device, dtype=torch.int32)
    bufs = [(ys[:, None] * w + xs[None, :]).reshape(-1).contiguous(), None]
    bufs[1] = torch.empty_like(bufs[0])
    cur, n = 0, bufs[0].numel()
    n_out = torch.zeros(1, dtype=torch.int64, device=device)
    dev.frame_begin(params)
    done = 0
    while done < params.ns_aa and n > 0:
        if done > 0:
            # the adaptive test of raytrace_pixel at i = done; every rank holds the
            # whole state, so every rank computes the same survivor list
            dev.adaptive_filter(bufs[cur].data_ptr(), n, done, state.data_ptr(), bufs[cur ^ 1].data_ptr(),
                                n_out.data_ptr(), stream)
            n = int(n_out.item())
            cur ^= 1
            if n == 0:
                break
        active = bufs[cur][:n]
        ns = min(params.samples_per_batch, params.ns_aa - done)
        mine = active[rank::world].contiguous() if world > 1 else active
        if record_rounds is not None:
            record_rounds.append((done, n, int(mine.numel())))
        if mine.numel():
            dev.trace_pixels(mine.data_ptr(), mine.numel(), done, ns, state.data_ptr(), stream)
        if world > 1:
            _exchange(state6, active, mine, world, group)
        done += ns
    hdr = torch.empty(npix * 3, dtype=torch.float64, device=device)
    cnt = torch.empty(npix, dtype=torch.int32, device=device)
    dev.frame_resolve(state.data_ptr(), hdr.data_ptr(), cnt.data_ptr(), stream)
    return hdr.view(h, w, 3), cnt.view(h, w)


def reference_schedule_cpu(counts_per_round, world):
    """For tests: per-rank pixel counts of every round given the active counts."""
    return [[PixelSchedule.shard_size(n, r, world) for r in range(world)] for n in counts_per_round]
