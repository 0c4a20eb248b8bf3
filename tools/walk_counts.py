import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
import pathtracer_amd as pt
from pathtracer_amd import cabi, distributed
import make_standin_scene
for name in ("dragon", "bunny"):
    path = make_standin_scene.standin_path("dragon") if name == "dragon" else "scenes/CBbunny.dae"
    sc = pt.Scene(path); dev = pt.Device(0); dev.upload(sc); dev.set_camera(sc.camera(800, 600))
    dev.set_profiling(events=False, counters=True)
    p = cabi.make_params(800, 600, ns_aa=64, max_ray_depth=5, ns_area_light=4, seed=77)
    distributed.render_frame(dev, p, device=torch.device("cuda", 0))
    st = dev.stats()
    print(name, "rays", st.rays, "nodes/ray", st.nodes_visited / 4 / st.rays, "tests/ray", st.prim_tests / st.rays,
          "exact/test", st.exact_tests / max(st.prim_tests, 1), "exact/ray", st.exact_tests / st.rays)
    dev.close()
