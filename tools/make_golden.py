#!/usr/bin/env python3
"""Generates tests/golden/* by RUNNING THE REFERENCE ITSELF (oracle/_ref/ref_driver,
built from the reference's own sources by `make -C oracle ref`). Runs only where
/root/reference exists; the resulting small fixtures are committed.

  kat.ptar                     pure-function known answers (bit-exact targets)
  scene_<name>.ptar            full flattened scene + BVH + camera of small scenes
  scene_hashes.json            SHA-256 of every array of every scene (incl. large ones)
  primary_<name>_<w>x<h>.npz   primary-ray primitive ids (all pixels, 4 sub-pixel
                               offsets), SHA-256 of t, and t on a strided subset
  render_<tag>.npz             reference images (float32 mean radiance + int32 sample
                               counts) with their command lines
"""
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import ptar  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
DAE = "/root/reference/dae"
GOLD = os.path.join(ROOT, "tests", "golden")
TMP = "/tmp/ptamd_golden"

SCENE_ARRAYS = ["prim_type", "prim_material", "prim_pos", "prim_nrm", "mat_kind", "mat_param", "node_bbox",
                "node_left", "node_right", "node_start", "node_count", "root", "light_kind", "light_param",
                "camera", "jitter_table"]


def run(args):
    subprocess.run([REF] + args, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def scenes():
    out = {}
    for sub in ("sky", "simple", "keenan", "meshedit"):
        d = os.path.join(DAE, sub)
        for f in sorted(os.listdir(d)):
            if f.endswith(".dae"):
                out[f[:-4]] = os.path.join(d, f)
    return out


def main(only_renders=False):
    os.makedirs(GOLD, exist_ok=True)
    os.makedirs(TMP, exist_ok=True)
    sc = scenes()
    if not only_renders:
        scene_and_primary_goldens(sc)
    render_goldens(sc)


def scene_and_primary_goldens(sc):
    run(["kat", "-f", os.path.join(GOLD, "kat.ptar")])

    hashes = {}
    small = ["CBspheres_lambertian", "CBspheres", "CBempty", "CBgems", "cube", "plane"]
    for name, path in sc.items():
        for (w, h) in ((800, 600), (480, 360)):
            out = os.path.join(TMP, "s.ptar")
            run(["dump", "-r", str(w), str(h), "-f", out, path])
            a = ptar.load(out)
            # materials per primitive, so that numbering conventions do not matter
            a["prim_mat_kind"] = a["mat_kind"][a["prim_material"]]
            a["prim_mat_param"] = a["mat_param"].reshape(-1, 3)[a["prim_material"]].ravel()
            key = "%s@%dx%d" % (name, w, h)
            hashes[key] = {k: sha(a[k]) for k in a if k not in ("prim_material", "mat_kind", "mat_param")}
            hashes[key]["n_prims"] = int(a["prim_type"].size)
            hashes[key]["n_nodes"] = int(a["node_left"].size)
            if name in small and (w, h) == (480, 360):
                ptar.save(os.path.join(GOLD, "scene_%s.ptar" % name), {k: a[k] for k in SCENE_ARRAYS})
    json.dump(hashes, open(os.path.join(GOLD, "scene_hashes.json"), "w"), indent=0, sort_keys=True)

    primary_goldens(sc, (("CBspheres_lambertian", 480, 360), ("CBspheres", 240, 180), ("CBgems", 240, 180),
                         ("CBbunny", 200, 150), ("CBcoil", 160, 120), ("cube", 96, 72)))


def primary_goldens(sc, jobs):
    for name, w, h in jobs:
        out = os.path.join(TMP, "p.ptar")
        run(["primary", "-r", str(w), str(h), "-f", out, sc[name]])
        a = ptar.load(out)
        rec = {"offsets": a["primary_offsets"], "w": w, "h": h}
        for o in range(4):
            ids = a["primary_id_%d" % o]
            rec["ids_%d" % o] = ids.astype(np.int32)
            rec["t_sha256_%d" % o] = np.frombuffer(bytes.fromhex(sha(a["primary_t_%d" % o])), np.uint8)
            rec["t_sub_%d" % o] = a["primary_t_%d" % o][::97]
        np.savez_compressed(os.path.join(GOLD, "primary_%s_%dx%d.npz" % (name, w, h)), **rec)



def primary_goldens_full_size():
    """Primary hits at the sizes of the benchmark table's configurations 2, 3 and 4 (3 and 4 on the
    stand-ins for the absent CBdragon.dae / CBlucy.dae), every pixel, four sub-pixel offsets."""
    import make_standin_scene
    sc = dict(scenes())
    sc["CBdragon_standin"] = make_standin_scene.ensure("dragon")
    sc["CBlucy_standin"] = make_standin_scene.ensure("lucy")
    # the irregular stand-in (folded, non-uniformly tessellated), at a reduced size
    sc["CBfolded_standin"] = make_standin_scene.ensure("folded")
    primary_goldens(sc, (("CBbunny", 800, 600), ("CBdragon_standin", 800, 600), ("CBlucy_standin", 1920, 1080),
                         ("CBfolded_standin", 400, 300)))


def render_goldens(sc):
    renders = [
        # tag, scene, threads, args
        ("cbsl_120x90_s1024", "CBspheres_lambertian", 8, ["-s", "1024", "-l", "4", "-m", "5", "-r", "120", "90"]),
        ("cbsl_240x180_s1024", "CBspheres_lambertian", 8, ["-s", "1024", "-l", "4", "-m", "5", "-r", "240", "180"]),
        ("cbsl_64x48_m0", "CBspheres_lambertian", 1, ["-s", "64", "-l", "1", "-m", "0", "-r", "64", "48"]),
        ("cbsl_64x48_m1", "CBspheres_lambertian", 1, ["-s", "256", "-l", "2", "-m", "1", "-r", "64", "48"]),
        ("cbsl_64x48_m2", "CBspheres_lambertian", 1, ["-s", "512", "-l", "2", "-m", "2", "-r", "64", "48"]),
        ("cbsl_64x48_m3_o0", "CBspheres_lambertian", 1, ["-s", "512", "-l", "2", "-m", "3", "-o", "0", "-r", "64", "48"]),
        ("cbsl_64x48_H", "CBspheres_lambertian", 1, ["-s", "512", "-l", "4", "-m", "2", "-H", "-r", "64", "48"]),
        ("cbsl_64x48_a16", "CBspheres_lambertian", 1, ["-s", "256", "-l", "1", "-m", "4", "-a", "16", "0.2", "-r", "64", "48"]),
        ("cbspheres_96x72", "CBspheres", 4, ["-s", "256", "-l", "4", "-m", "5", "-r", "96", "72"]),
        ("cbgems_96x72", "CBgems", 4, ["-s", "256", "-l", "4", "-m", "5", "-r", "96", "72"]),
        ("cbbunny_100x75", "CBbunny", 8, ["-s", "256", "-l", "4", "-m", "5", "-r", "100", "75"]),
        # the light kinds the Cornell boxes do not have: a point light (cube), a directional light and
        # three ambient (infinite-hemisphere) lights (banana), a point and an ambient light (plane)
        ("cube_64x48", "cube", 1, ["-s", "256", "-l", "1", "-m", "3", "-r", "64", "48"]),
        ("banana_64x48", "banana", 1, ["-s", "256", "-l", "2", "-m", "3", "-r", "64", "48"]),
        ("plane_64x48", "plane", 1, ["-s", "256", "-l", "2", "-m", "2", "-r", "64", "48"]),
    ]
    if len(sys.argv) > 2 and sys.argv[1] == "renders":
        renders = [r for r in renders if sys.argv[2] in r[0]]
    for tag, name, threads, args in renders:
        out = os.path.join(TMP, "r.ptar")
        t0 = time.time()
        run(["render"] + args + ["-t", str(threads), "-f", out, sc[name]])
        a = ptar.load(out)
        np.savez_compressed(os.path.join(GOLD, "render_%s.npz" % tag), hdr=a["hdr"].astype(np.float32),
                            count=a["count"], stats=a["stats"], camera=a["camera"],
                            cmd=np.array(" ".join(["ref_driver", "render"] + args + ["-t", str(threads), name + ".dae"])))
        print("%s: %.1f s, %d samples" % (tag, time.time() - t0, int(a["count"].sum())), flush=True)


def blocks(img, b=8):
    h, w = img.shape[:2]
    c = img.shape[2] if img.ndim == 3 else 1
    v = img.reshape(h, w, c)[:h - h % b, :w - w % b].reshape(h // b, b, w // b, b, c).mean(axis=(1, 3))
    return v if img.ndim == 3 else v[..., 0]


def block_goldens(only=None):
    """BASELINE.json's five configurations rendered by the reference at full size (3 and 4: the
    stand-ins for the absent CBdragon.dae and CBlucy.dae, written by tools/make_standin_scene.py;
    5: the reference's mirror / glass BSDFs are empty, so its spheres are black); kept as
    8x8 block means (they keep any bias and little noise) so the fixtures stay small."""
    import make_standin_scene
    sc = dict(scenes())
    sc["CBdragon_standin"] = make_standin_scene.ensure("dragon")
    sc["CBlucy_standin"] = make_standin_scene.ensure("lucy")
    jobs = [
        ("config4_lucy_standin_1920x1080_s4096_l4_m8", "CBlucy_standin", ["-s", "4096", "-l", "4", "-m", "8", "-r", "1920", "1080"], 1920, 1080),
        ("config5_cbspheres_800x600_s1024_l4_m16", "CBspheres", ["-s", "1024", "-l", "4", "-m", "16", "-r", "800", "600"], 800, 600),
        ("config3_dragon_standin_800x600_s1024_l4_m5", "CBdragon_standin", ["-s", "1024", "-l", "4", "-m", "5", "-r", "800", "600"], 800, 600),
        ("config1_cbsl_480x360_s64_l16_m5", "CBspheres_lambertian", ["-s", "64", "-l", "16", "-m", "5", "-r", "480", "360"], 480, 360),
        ("config2_cbbunny_800x600_s256_l4_m5", "CBbunny", ["-s", "256", "-l", "4", "-m", "5", "-r", "800", "600"], 800, 600),
    ]
    for tag, name, args, w, h in jobs:
        if only and only not in tag:
            continue
        out = os.path.join(TMP, "b.ptar")
        t0 = time.time()
        run(["render"] + args + ["-t", "8", "-f", out, sc[name]])
        a = ptar.load(out)
        hdr = a["hdr"].reshape(h, w, 3)
        cnt = a["count"].reshape(h, w).astype(np.float64)
        np.savez_compressed(os.path.join(GOLD, "blocks_%s.npz" % tag), hdr_blocks=blocks(hdr).astype(np.float32),
                            count_blocks=blocks(cnt).astype(np.float32), total_samples=np.int64(a["count"].sum()),
                            mean_rgb=hdr.reshape(-1, 3).mean(axis=0),
                            cmd=np.array(" ".join(["ref_driver", "render"] + args + ["-t", "8", name + ".dae"])))
        print("%s: %.1f s, %d samples" % (tag, time.time() - t0, int(a["count"].sum())), flush=True)


if __name__ == "__main__":
    if not os.path.exists(REF):
        raise SystemExit("build the reference driver first: make -C oracle ref")
    if len(sys.argv) > 1 and sys.argv[1] == "renders":
        os.makedirs(TMP, exist_ok=True)
        main(only_renders=True)
    elif len(sys.argv) > 1 and sys.argv[1] == "primary_full":
        os.makedirs(TMP, exist_ok=True)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        primary_goldens_full_size()
    elif len(sys.argv) > 1 and sys.argv[1] == "blocks":
        os.makedirs(TMP, exist_ok=True)
        block_goldens(sys.argv[2] if len(sys.argv) > 2 else None)
    else:
        main()
        primary_goldens_full_size()
        block_goldens()
