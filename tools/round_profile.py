#!/usr/bin/env python3
"""Per-round active-pixel bounds and wall time of one frame (synchronous rounds)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import pathtracer_amd as pt
from pathtracer_amd import cabi, distributed
import make_standin_scene
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 1
path = make_standin_scene.ensure("dragon") if name == "dragon" else os.path.join(ROOT, "scenes", "CBbunny.dae")
spp = 1024 if name == "dragon" else 256
w, h = 800, 600
sc = pt.Scene(path)
dev = pt.Device(0); dev.upload(sc); dev.set_camera(sc.camera(w, h))
p = cabi.make_params(w, h, ns_aa=spp, max_ray_depth=5, ns_area_light=4, seed=1)
cuda = torch.device("cuda", 0)
for rep in range(2):
    rounds = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    distributed.render_frame(dev, p, device=cuda, lag=0, record_rounds=rounds, pixel_stride=stride)
    torch.cuda.synchronize(); t1 = time.perf_counter()
print("frame %.1f ms, %d rounds" % (1000 * (t1 - t0), len(rounds)))
print("active pixels per round:", [b for _, b, _ in rounds])
