#!/usr/bin/env python3
"""Experiment: two render contexts on two streams of one GPU, each rendering the share of one
rank of 2K, against one context rendering the share of one rank of K.
usage: two_streams.py [dragon|bunny] K"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import pathtracer_amd as pt
from pathtracer_amd import cabi, distributed
import make_standin_scene
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
path = make_standin_scene.ensure("dragon") if name == "dragon" else os.path.join(ROOT, "scenes", "CBbunny.dae")
spp = 1024 if name == "dragon" else 256
w, h = 800, 600
sc = pt.Scene(path)
cuda = torch.device("cuda", 0)
devs = []
for i in range(2):
    d = pt.Device(0); d.upload(sc); d.set_camera(sc.camera(w, h)); devs.append(d)
p = cabi.make_params(w, h, ns_aa=spp, max_ray_depth=5, ns_area_light=4, seed=1)
streams = [torch.cuda.Stream(device=cuda) for _ in range(2)]

def one(dev, stride, stream, out, target):
    with torch.cuda.stream(stream):
        hdr, cnt = distributed.render_frame(dev, p, device=cuda, pixel_stride=stride, target_paths=target)
        out.append(cnt)

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1000 * (time.perf_counter() - t0) / reps

def single():
    one(devs[0], K, streams[0], [], 16 << 20)

for target in (16 << 20, 8 << 20):
    def double_seq():
        for i in range(2):
            one(devs[i], 2 * K, streams[i], [], target)
    def double_thr():
        ts = [threading.Thread(target=one, args=(devs[i], 2 * K, streams[i], [], target)) for i in range(2)]
        for t in ts: t.start()
        for t in ts: t.join()
    print("%s K=%d target %dMi: one context, rank of K: %.2f ms; two contexts, each rank of 2K: issued in turn %.2f ms, from two threads %.2f ms" % (
        name, K, target >> 20, timed(single), timed(double_seq), timed(double_thr)))
