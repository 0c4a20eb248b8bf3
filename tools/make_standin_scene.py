#!/usr/bin/env python3
"""Writes the documented stand-ins for the headline scenes that are absent from the
reference checkout (dae/sky/CBdragon.dae, dae/sky/CBlucy.dae).

A stand-in is the CBempty Cornell box (walls, light, camera: copied from
scenes/CBempty.dae), its light declared an AREA light as in every other CB scene of the
reference (CBbunny.dae, CBspheres.dae ...: CBempty.dae alone lacks the CGL <area> extra, which
makes its light a point light that takes one sample whatever -l says), plus one closed, manifold, procedurally generated diffuse
mesh of the requested size: the tube around a (2,3) torus knot with a smooth
radial ripple. The mesh is deterministic (no random numbers), so every machine
writes byte-identical files.

    python tools/make_standin_scene.py dragon   # ~100k triangles -> scenes/_generated/CBdragon_standin.dae
    python tools/make_standin_scene.py lucy     # ~133k triangles -> scenes/_generated/CBlucy_standin.dae
    python tools/make_standin_scene.py folded   # ~100k triangles -> scenes/_generated/CBfolded_standin.dae

"folded" is a second, IRREGULAR stand-in of the dragon's size, there to tell how much of the
traversal figures belongs to the knot's regularity: the same tube, but sampled non-uniformly along
and around (triangle areas span more than 1000 x), with deep lobes and a sideways wobble of the
tube's centre line so that the surface folds back over itself and sheets of it overlap: long thin
triangles next to tiny ones, overlapping boxes, a worse tree. Same box, same light.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRESETS = {"dragon": (512, 98), "lucy": (640, 104), "tiny": (48, 12), "folded": (512, 98), "folded-tiny": (64, 16)}


def knot_mesh(nu, nv):
    """Vertices (world space: x right, y up, z towards the camera) and triangles."""
    u = np.linspace(0.0, 2 * np.pi, nu, endpoint=False)
    p, q = 2, 3
    r = 0.30 + 0.12 * np.cos(q * u)
    centre = np.stack([r * np.cos(p * u), 0.14 * np.sin(q * u), r * np.sin(p * u)], axis=1)
    du = 2 * np.pi / nu
    tangent = (np.roll(centre, -1, axis=0) - np.roll(centre, 1, axis=0)) / (2 * du)
    tangent /= np.linalg.norm(tangent, axis=1, keepdims=True)
    up = np.array([0.0, 1.0, 0.0])
    n1 = np.cross(tangent, up)
    n1 /= np.linalg.norm(n1, axis=1, keepdims=True)
    n2 = np.cross(tangent, n1)
    v = np.linspace(0.0, 2 * np.pi, nv, endpoint=False)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    tube = 0.085 * (1.0 + 0.18 * np.sin(7 * uu + 3 * vv) + 0.10 * np.cos(11 * vv - 5 * uu))
    pts = (centre[:, None, :] + tube[..., None] * (np.cos(vv)[..., None] * n1[:, None, :] + np.sin(vv)[..., None] * n2[:, None, :]))
    pts = pts.reshape(-1, 3) + np.array([0.0, 0.62, -0.05])
    idx = np.arange(nu * nv).reshape(nu, nv)
    a = idx
    b = np.roll(idx, -1, axis=0)
    c = np.roll(np.roll(idx, -1, axis=0), -1, axis=1)
    d = np.roll(idx, -1, axis=1)
    tris = np.concatenate([np.stack([a, b, c], axis=-1).reshape(-1, 3), np.stack([a, c, d], axis=-1).reshape(-1, 3)])
    return pts.astype(np.float32), tris


def warp(t, depth, periods):
    """Monotone map of [0, 2 pi) onto itself whose derivative swings between 1 - depth and 1 + depth."""
    return t + depth * np.sin(periods * t) / periods


def folded_mesh(nu, nv):
    """The irregular stand-in: the knot's tube under non-uniform sampling, deep lobes and a wobbling centre line."""
    s = np.linspace(0.0, 2 * np.pi, nu, endpoint=False)
    u = warp(warp(s, 0.93, 5), 0.6, 17)  # spacing along the tube varies ~70 x
    p, q = 2, 3
    r = 0.30 + 0.12 * np.cos(q * u)
    centre = np.stack([r * np.cos(p * u), 0.14 * np.sin(q * u), r * np.sin(p * u)], axis=1)
    eps = 1e-4
    up1 = u + eps
    r1 = 0.30 + 0.12 * np.cos(q * up1)
    tangent = np.stack([r1 * np.cos(p * up1), 0.14 * np.sin(q * up1), r1 * np.sin(p * up1)], axis=1) - centre
    tangent /= np.linalg.norm(tangent, axis=1, keepdims=True)
    up = np.array([0.0, 1.0, 0.0])
    n1 = np.cross(tangent, up)
    n1 /= np.linalg.norm(n1, axis=1, keepdims=True)
    n2 = np.cross(tangent, n1)
    # the centre line wobbles sideways faster than the tube is thick: neighbouring stretches of surface overlap
    centre = centre + 0.035 * np.sin(23 * u)[:, None] * n1 + 0.03 * np.cos(31 * u)[:, None] * n2
    t = np.linspace(0.0, 2 * np.pi, nv, endpoint=False)
    v = warp(t, 0.9, 3)  # spacing around the tube varies ~19 x
    uu, vv = np.meshgrid(u, v, indexing="ij")
    tube = 0.085 * (1.0 + 0.55 * np.sin(5 * uu + 2 * vv) + 0.30 * np.cos(9 * vv - 4 * uu) + 0.12 * np.sin(29 * uu))
    pts = (centre[:, None, :] + tube[..., None] * (np.cos(vv)[..., None] * n1[:, None, :] + np.sin(vv)[..., None] * n2[:, None, :]))
    pts = pts.reshape(-1, 3) + np.array([0.0, 0.62, -0.05])
    idx = np.arange(nu * nv).reshape(nu, nv)
    a = idx
    b = np.roll(idx, -1, axis=0)
    c = np.roll(np.roll(idx, -1, axis=0), -1, axis=1)
    d = np.roll(idx, -1, axis=1)
    tris = np.concatenate([np.stack([a, b, c], axis=-1).reshape(-1, 3), np.stack([a, c, d], axis=-1).reshape(-1, 3)])
    return pts.astype(np.float32), tris


def triangle_area_spread(pts, tris):
    """Largest over smallest triangle area, 99.9th over 0.1st percentile (for the scene's description)."""
    p = pts.astype(np.float64)
    ar = 0.5 * np.linalg.norm(np.cross(p[tris[:, 1]] - p[tris[:, 0]], p[tris[:, 2]] - p[tris[:, 0]]), axis=1)
    return float(ar.max() / max(ar.min(), 1e-300)), float(np.percentile(ar, 99.9) / max(np.percentile(ar, 0.1), 1e-300))


def write_scene(kind, out_path, area_light=True):
    """area_light=False keeps CBempty.dae's own light element, a point light (the scene of the earliest
    measurements)."""
    nu, nv = PRESETS[kind]
    pts, tris = folded_mesh(nu, nv) if kind.startswith("folded") else knot_mesh(nu, nv)
    with open(os.path.join(ROOT, "scenes", "CBempty.dae")) as f:
        dae = f.read()
    # the file is Z_UP: the loader maps (x, y, z) -> (-x, z, y); invert that here
    dae_pts = np.stack([-pts[:, 0], pts[:, 2], pts[:, 1]], axis=1)
    pos_txt = " ".join("%.7g" % x for x in dae_pts.ravel())
    p_txt = " ".join(str(int(i)) for i in tris.ravel())
    vcount = "3 " * len(tris)
    effect = """    <effect id="standin-effect">
      <profile_COMMON>
        <technique sid="common">
          <phong>
            <diffuse>
              <color sid="diffuse">0.7 0.62 0.45 1</color>
            </diffuse>
          </phong>
        </technique>
      </profile_COMMON>
    </effect>
"""
    material = """    <material id="standin-material" name="standin">
      <instance_effect url="#standin-effect"/>
    </material>
"""
    geometry = """    <geometry id="standin-mesh" name="standin">
      <mesh>
        <source id="standin-mesh-positions">
          <float_array id="standin-mesh-positions-array" count="%d">%s</float_array>
          <technique_common>
            <accessor source="#standin-mesh-positions-array" count="%d" stride="3">
              <param name="X" type="float"/>
              <param name="Y" type="float"/>
              <param name="Z" type="float"/>
            </accessor>
          </technique_common>
        </source>
        <vertices id="standin-mesh-vertices">
          <input semantic="POSITION" source="#standin-mesh-positions"/>
        </vertices>
        <polylist material="standin-material" count="%d">
          <input semantic="VERTEX" source="#standin-mesh-vertices" offset="0"/>
          <vcount>%s</vcount>
          <p>%s</p>
        </polylist>
      </mesh>
    </geometry>
""" % (dae_pts.size, pos_txt, len(dae_pts), len(tris), vcount, p_txt)
    node = """      <node id="standin" name="standin" type="NODE">
        <matrix sid="transform">1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1</matrix>
        <instance_geometry url="#standin-mesh">
          <bind_material>
            <technique_common>
              <instance_material symbol="standin-material" target="#standin-material"/>
            </technique_common>
          </bind_material>
        </instance_geometry>
      </node>
"""
    # the light: CGL area extra and radiance of dae/sky/CBbunny.dae (its size comes from the node's matrix)
    point_light = """          <quadratic_attenuation>0.001599967</quadratic_attenuation>
        </point>
      </technique_common>
"""
    area_extra = """      <extra>
        <technique profile="CGL">
          <area>
            <color sid="color">10 10 10</color>
          </area>
        </technique>
      </extra>
"""
    if dae.count(point_light) != 1:
        raise SystemExit("scenes/CBempty.dae: light element not found")
    if area_light:
        dae = dae.replace(point_light, point_light + area_extra, 1)
    dae = dae.replace("  </library_effects>", effect + "  </library_effects>", 1)
    dae = dae.replace("  </library_materials>", material + "  </library_materials>", 1)
    dae = dae.replace("  </library_geometries>", geometry + "  </library_geometries>", 1)
    dae = dae.replace("    </visual_scene>", node + "    </visual_scene>", 1)
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    tmp = out_path + ".tmp%d" % os.getpid()
    with open(tmp, "w") as f:
        f.write(dae)
    os.replace(tmp, out_path)
    return len(tris)


def standin_path(kind, area_light=True):
    return os.path.join(ROOT, "scenes", "_generated", "CB%s_standin%s.dae" % (kind, "" if area_light else "_pointlight"))


def ensure(kind, area_light=True):
    path = standin_path(kind, area_light)
    stale = True
    if os.path.exists(path):
        with open(path) as f:
            # (a file written before the light became an area light is stale)
            stale = ("<area>" in f.read(4096)) != area_light
    if stale:
        write_scene(kind, path, area_light)
    return path


if __name__ == "__main__":
    k = sys.argv[1] if len(sys.argv) > 1 else "dragon"
    n = write_scene(k, standin_path(k))
    print("%s: %d triangles -> %s" % (k, n + 12, standin_path(k)))
    nu_, nv_ = PRESETS[k]
    print("triangle areas, largest / smallest and 99.9th / 0.1st percentile: %.0f, %.0f" % triangle_area_spread(
        *(folded_mesh(nu_, nv_) if k.startswith("folded") else knot_mesh(nu_, nv_))))
