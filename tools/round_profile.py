#!/usr/bin/env python3
"""Per-round active-pixel bounds, rays traced and wall time of one frame.
usage: round_profile.py [dragon|bunny] [pixel stride] [per_pixel 0|1] [target_paths] [max_batches] [lag]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import pathtracer_amd as pt
from pathtracer_amd import cabi, distributed
import make_standin_scene
arg = lambda i, d: type(d)(sys.argv[i]) if len(sys.argv) > i else d
name, stride, per_pixel = arg(1, "dragon"), arg(2, 1), arg(3, 0)
target, max_batches, lag = arg(4, 16 << 20), arg(5, 32), arg(6, 1)
path = make_standin_scene.ensure("dragon") if name == "dragon" else os.path.join(ROOT, "scenes", "CBbunny.dae")
spp = 1024 if name == "dragon" else 256
w, h = 800, 600
sc = pt.Scene(path)
dev = pt.Device(0); dev.upload(sc); dev.set_camera(sc.camera(w, h))
dev.set_profiling(events=True, counters=False)
p = cabi.make_params(w, h, ns_aa=spp, max_ray_depth=5, ns_area_light=4, seed=1)
cuda = torch.device("cuda", 0)
for rep in range(3):
    rounds = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    distributed.render_frame(dev, p, device=cuda, lag=lag, record_rounds=rounds, pixel_stride=stride,
                             per_pixel=bool(per_pixel), target_paths=target, max_batches=max_batches)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    st = dev.stats()
print("stride %d per_pixel %d target %d max_batches %d lag %d: frame %.2f ms, %d rounds, kernels %.2f ms" % (
    stride, per_pixel, target, max_batches, lag, 1000 * (t1 - t0), len(rounds), sum(st.kernel_ms)))
print(" bounds:", [b for _, b, _ in rounds])
print(" samples kept %d, rays %d (%.2f per kept sample), shadow %d, launches %d" % (
    st.samples, st.rays, st.rays / max(st.samples, 1), st.shadow_rays, sum(st.launches)))
print(" kernel ms:", {cabi.K_NAMES[k]: round(st.kernel_ms[k], 2) for k in range(12) if st.launches[k]})
