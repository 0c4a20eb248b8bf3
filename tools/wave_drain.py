#!/usr/bin/env python3
"""How the traversal launches of a run drain: reads the file a PTAMD_WAVELOG build of the device
layer writes (make -C pathtracer_amd/csrc variant NAME=wavelog DEFS=-DPTAMD_WAVELOG; run with
PTAMD_LIB=.../libptamd_wavelog.so PTAMD_WAVELOG=<file>) and prints, per kernel, how much of the
launches' time passes with fewer than half, a tenth and a hundredth of the peak number of waves
resident. usage: wave_drain.py <file> [skip_launches] [-v]"""
import struct, sys
import numpy as np

KIND = {0: "k_extend", 1: "k_trace2", 2: "k_shadow"}
data = open(sys.argv[1], "rb").read()
args = [a for a in sys.argv[2:] if a != "-v"]
skip = int(args[0]) if args else 0
n = struct.unpack_from("<I", data, 0)[0]
launches = [struct.unpack_from("<III", data, 4 + 12 * i) for i in range(n)]
ticks = np.frombuffer(data, dtype=np.uint64, offset=4 + 12 * n).reshape(-1, 2)
TICK_US = 0.01  # wall_clock64: 100 MHz
agg = {}
rows = []
for li, (kind, base, waves) in enumerate(launches):
    if li < skip:
        continue
    t = ticks[base:base + waves]
    t = t[t[:, 0] > 0]
    if len(t) == 0:
        continue  # the launch returned at once (k_shadow behind a fused k_trace2)
    t0, t1 = int(t[:, 0].min()), int(t[:, 1].max())
    dur = (t1 - t0) * TICK_US
    if dur < 20:
        continue
    # resident waves over time from the start / end events
    ev = np.concatenate([np.stack([t[:, 0], np.ones(len(t), np.int64)], 1).astype(np.int64),
                         np.stack([t[:, 1], -np.ones(len(t), np.int64)], 1).astype(np.int64)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    conc = np.cumsum(ev[:, 1])
    dt = np.diff(np.append(ev[:, 0], t1)) * TICK_US
    peak = conc.max()
    busy = float((conc * dt).sum()) / peak  # microseconds at full residency that hold the same wave-time
    below = [float(dt[conc < peak * f].sum()) for f in (0.5, 0.1, 0.01)]
    rows.append((li, KIND[kind], len(t), dur, busy, *below))
    a = agg.setdefault(KIND[kind], [0, 0.0, 0.0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += dur; a[2] += busy
    for k in range(3):
        a[3 + k] += below[k]
if "-v" in sys.argv:
    for r in rows:
        print("launch %4d %-9s waves %6d  %8.1f us  full-residency equivalent %8.1f us  below 50%%: %7.1f  10%%: %7.1f  1%%: %7.1f" % r)
print("%-9s %8s %10s %14s %12s %12s %12s" % ("kernel", "launches", "total ms", "wave-time ms", "<50% ms", "<10% ms", "<1% ms"))
for k, a in agg.items():
    print("%-9s %8d %10.2f %14.2f %12.2f %12.2f %12.2f" % (k, a[0], a[1] / 1e3, a[2] / 1e3, a[3] / 1e3, a[4] / 1e3, a[5] / 1e3))
