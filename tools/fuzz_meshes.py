#!/usr/bin/env python3
"""Randomly shaped tube meshes in the Cornell box (make_standin_scene's irregular generator with random knot type,
frequencies, amplitudes and tessellation, 384 to 16000 triangles): the host pipeline builds each scene, and the HIP path
is compared with the CPU oracle on 60000 random primary rays (primitive ids and t bit for bit) and on a small
render (the criteria of tests/test_gpu_parity.py::test_render_matches_oracle). Exercises the 4-wide tree builder
and the walks on trees of many shapes.
usage: fuzz_meshes.py <seed> <count>      (needs a GPU)"""
import sys, os, random, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt
from pathtracer_amd import cabi
import make_standin_scene as M
import oracle_lib as O
rng = random.Random(int(sys.argv[1]))
dev = pt.Device(0)
tmp = tempfile.mkdtemp()
bad = 0
def rand_mesh(nu, nv):
    s = np.linspace(0.0, 2 * np.pi, nu, endpoint=False)
    u = M.warp(M.warp(s, rng.uniform(0, 0.95), rng.randint(1, 7)), rng.uniform(0, 0.7), rng.randint(8, 20))
    p, q = rng.choice([(2, 3), (1, 2), (3, 4), (2, 5)])
    r = 0.30 + rng.uniform(0.02, 0.14) * np.cos(q * u)
    centre = np.stack([r * np.cos(p * u), rng.uniform(0.02, 0.2) * np.sin(q * u), r * np.sin(p * u)], axis=1)
    up1 = u + 1e-4
    r1 = 0.30 + 0.12 * np.cos(q * up1)
    tangent = np.stack([r1 * np.cos(p * up1), 0.14 * np.sin(q * up1), r1 * np.sin(p * up1)], axis=1) - centre
    tangent /= np.linalg.norm(tangent, axis=1, keepdims=True)
    n1 = np.cross(tangent, np.array([0.0, 1.0, 0.0])); n1 /= np.linalg.norm(n1, axis=1, keepdims=True)
    n2 = np.cross(tangent, n1)
    centre = centre + rng.uniform(0, 0.05) * np.sin(rng.randint(5, 40) * u)[:, None] * n1 + rng.uniform(0, 0.04) * np.cos(rng.randint(5, 40) * u)[:, None] * n2
    t = np.linspace(0.0, 2 * np.pi, nv, endpoint=False)
    v = M.warp(t, rng.uniform(0, 0.9), rng.randint(1, 4))
    uu, vv = np.meshgrid(u, v, indexing="ij")
    tube = rng.uniform(0.03, 0.1) * (1.0 + rng.uniform(0, 0.6) * np.sin(rng.randint(1, 9) * uu + rng.randint(1, 4) * vv) + rng.uniform(0, 0.3) * np.cos(rng.randint(1, 12) * vv - 4 * uu))
    pts = (centre[:, None, :] + tube[..., None] * (np.cos(vv)[..., None] * n1[:, None, :] + np.sin(vv)[..., None] * n2[:, None, :]))
    pts = pts.reshape(-1, 3) + np.array([0.0, 0.62, -0.05])
    idx = np.arange(nu * nv).reshape(nu, nv)
    a = idx; b = np.roll(idx, -1, axis=0); c = np.roll(np.roll(idx, -1, axis=0), -1, axis=1); d = np.roll(idx, -1, axis=1)
    tris = np.concatenate([np.stack([a, b, c], axis=-1).reshape(-1, 3), np.stack([a, c, d], axis=-1).reshape(-1, 3)])
    return pts.astype(np.float32), tris
M.folded_mesh = rand_mesh
for it in range(int(sys.argv[2])):
    nu, nv = rng.choice([(24, 8), (64, 16), (128, 24), (200, 40), (37, 11)])
    M.PRESETS["folded-tiny"] = (nu, nv)
    path = os.path.join(tmp, "m%d.dae" % it)
    M.write_scene("folded-tiny", path)
    try:
        sc = pt.Scene(path)
    except Exception as e:
        print("skip (loader):", str(e)[:80]); continue
    w, h = rng.choice([(200, 150), (97, 61)])
    arr, cam = sc.arrays(), sc.camera(w, h)
    dev.upload(sc); dev.set_camera(cam)
    n = 60000
    g = np.random.default_rng(it)
    sx, sy = g.random(n), g.random(n)
    ids, ts = dev.trace_primary(sx, sy)
    rid, rts = O.trace_primary(arr, cam, sx, sy)
    ok1 = np.array_equal(ids, rid) and ts.tobytes() == rts.tobytes()
    p = cabi.make_params(48, 36, ns_aa=32, max_ray_depth=4, ns_area_light=2, seed=it)
    cam2 = sc.camera(48, 36); dev.set_camera(cam2)
    hdr, cnt = dev.render(p)
    rh, rc, _ = O.render(arr, cam2, p, threads=16)
    close = np.isclose(hdr, rh, rtol=1e-9, atol=1e-12).all(axis=2)
    ok2 = (cnt == rc).mean() >= 0.999 and close.mean() >= 0.999
    bad += not (ok1 and ok2)
    print("ok  " if ok1 and ok2 else "FAIL", nu, nv, "tris", 2 * nu * nv, "primary", ok1, "render", ok2, "hits", int((ids >= 0).sum()), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
