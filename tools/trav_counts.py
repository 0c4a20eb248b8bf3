#!/usr/bin/env python3
"""Prints the GPU traversal's own per-ray work (4-wide nodes visited, primitive tests)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import pathtracer_amd as pt
from pathtracer_amd import cabi
import make_standin_scene
name = sys.argv[1] if len(sys.argv) > 1 else "bunny"
path = make_standin_scene.ensure("dragon") if name == "dragon" else os.path.join(ROOT, "scenes", "CBbunny.dae")
w, h = 800, 600
sc = pt.Scene(path)
dev = pt.Device(0); dev.upload(sc); dev.set_camera(sc.camera(w, h))
dev.set_profiling(events=True, counters=True)
p = cabi.make_params(w, h, ns_aa=64, max_ray_depth=5, ns_area_light=4, seed=1)
hdr, cnt = dev.render(p)
st = dev.stats()
print(name, "rays %d shadow %d samples %d" % (st.rays, st.shadow_rays, st.samples))
print(" box tests/ray %.2f (nodes/ray %.2f)  prim tests/ray %.2f" % (st.nodes_visited / st.rays, st.nodes_visited / 4 / st.rays, st.prim_tests / st.rays))
print(" kernel ms", {cabi.K_NAMES[k]: round(st.kernel_ms[k], 2) for k in range(8)})
