#!/usr/bin/env python3
"""Benchmark of the render hot path: path samples per second.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one full frame of the workload (default: the CBdragon stand-in, 800x600,
1024 spp, 4 samples per area light, max depth 5, adaptive sampling 32 / 0.05, the
reference's defaults). samples = sum of per-pixel sample counts, as the reference's
sampleCountBuffer would hold them. Rank 0 prints ONE JSON line.

The line also carries
  roofline      for the dominant kernel (a BVH walk: bound by VALU instruction issue,
                not by HBM). Top level: VALU lane-operations per second against the card's
                full-rate ceiling, 1024 SIMDs x 32 lanes x 2.4 GHz = 78.6 T/s (a wave64
                v_fma_f32 issues every 2 cycles once a SIMD holds two or more waves: measured
                in shader cycles, profiles/r03_valu_rates.txt), from the committed rocprofv3
                counter passes of this workload and the kernel's live HIP-event time. The
                fraction factors as (cycles the VALU is busy) x (what the instruction mix
                allows: most non-FMA instructions issue every 4 cycles) x (lanes switched on),
                reported under `issue`.
                Beside it: `hbm_algorithmic` (bytes the REFERENCE's traversal would
                move for the same rays, counted by the CPU restatement, over the live
                kernel time against 8 TB/s: above 1, the GPU walk does far less work per
                ray), `hbm_measured` (FETCH_SIZE / WRITE_SIZE counters), `l2_walk` (the
                GPU walk's own node and primitive fetches against the L2 peak);
  cpu_baseline  the reference's own renderer (its sources, built by
                `make -C oracle ref`) timed on this host on a bounded cell of the
                same frame.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
L2_PEAK_GBS = 34500.0  # aggregate of the eight XCDs' L2s
# VALU ceiling: 256 CUs x 4 SIMDs, each retiring a 64-lane full-rate instruction (v_fma_f32, v_add_f32,
# v_mul_f32, 32-bit integer add / and / xor, v_mov_b32) every 2 cycles = 32 lanes per cycle, at the 2.4 GHz
# peak clock. Measured on the card in shader cycles (tools/ubench/valu_rates.hip, profiles/r03_valu_rates.txt:
# 2.25 - 2.3 cycles per wave64 v_fma_f32 from two waves per SIMD on, whatever the EXEC mask; 4.1 - 4.3 for
# compares, selects, conversions, min / max, shifts, all FP64 arithmetic and v_pk_fma_f32; 8 for v_rcp_f32;
# 16 for v_rcp_f64 / v_sqrt_f64; the loops ran at 2.0 - 2.4 GHz by s_memtime against s_memrealtime).
N_SIMD, SIMD_LANES, CLOCK_HZ = 1024, 32, 2.4e9
FULL_RATE_CYCLES = 2.0  # cycles a full-rate wave64 instruction occupies its SIMD
LANE_OPS_PEAK_T = N_SIMD * SIMD_LANES * CLOCK_HZ / 1e12


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="dragon", help="dragon (stand-in), dragon-point (the stand-in with CBempty.dae's own point light: the earlier headline workload), "
                         "folded (the irregular stand-in), lucy (stand-in; configuration 4 of the benchmark table with --width 1920 "
                         "--height 1080 --depth 8), bunny, spheres (configuration 5 with --depth 16), cbsl (CBspheres_lambertian: configuration 1 "
                         "with --width 480 --height 360 --light-samples 16), or a .dae path")
    ap.add_argument("--specular", action="store_true",
                    help="ideal mirror / glass instead of the reference's black stubs (the divergent-specular "
                         "stress: --scene spheres --specular --depth 16)")
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--light-samples", type=int, default=4)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--max-paths", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=0, help="samples per pixel of the CPU baseline's cell (0: the workload's own)")
    ap.add_argument("--cpu-cell", type=int, default=8, help="width in pixels of the CPU baseline's cell (full frame height)")
    ap.add_argument("--target-paths", type=int, default=16 << 20, help="see PixelSchedule.batches")
    ap.add_argument("--max-batches", type=int, default=32)
    ap.add_argument("--per-pixel", type=int, default=-1,
                    help="1: pixels are traced ahead only as far as their statistics predict (see render_frame); "
                         "0: all pixels of a round in lockstep; -1: the library's choice for the workload")
    ap.add_argument("--exchange", default="frame", help="frame: pixels dealt once, one all-gather per frame; "
                    "round: dealt again after every adaptive round, one all-gather per round")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL; the real thing) or gloo (rehearsal of the "
                    "multi-rank flow with several ranks on one GPU, rows staged through the host)")
    ap.add_argument("--sync-below", type=int, default=-1,
                    help="active-pixel bounds at or below this are read back before the next round is queued "
                         "(-1: render_frame's default)")
    ap.add_argument("--pixel-stride", type=int, default=1,
                    help="rehearsal: render every K-th pixel only, i.e. the work of one rank of K")
    return ap.parse_args()


def resolve_scene(name, rank):
    import make_standin_scene
    if name == "dragon":
        path = make_standin_scene.standin_path("dragon")
        if rank == 0:
            make_standin_scene.ensure("dragon")
        return path, ("CBdragon stand-in (CBempty box with the CB scenes' area light + 100k-triangle procedural knot; "
                      "CBdragon.dae is absent from the reference checkout)"), 1024
    if name == "dragon-point":
        # the headline workload of earlier measurements, kept for like-for-like comparisons: the same geometry with
        # CBempty.dae's own light, a point light (one shadow ray per vertex whatever -l says)
        path = make_standin_scene.standin_path("dragon", area_light=False)
        if rank == 0:
            make_standin_scene.ensure("dragon", area_light=False)
        return path, "CBdragon stand-in with CBempty.dae's own point light (the earlier headline workload)", 1024
    if name == "folded":
        # a second, irregular stand-in of the same size: the knot's tube sampled non-uniformly (triangle areas span
        # more than 1000 x), with deep lobes and a wobbling centre line so that sheets of surface overlap
        path = make_standin_scene.standin_path("folded")
        if rank == 0:
            make_standin_scene.ensure("folded")
        return path, "irregular stand-in (CB box and area light + 100k-triangle folded, non-uniformly tessellated tube)", 1024
    if name == "lucy":
        path = make_standin_scene.standin_path("lucy")
        if rank == 0:
            make_standin_scene.ensure("lucy")
        return path, ("CBlucy stand-in (CB box and area light + 133k-triangle procedural knot; CBlucy.dae is absent from "
                      "the reference checkout)"), 4096
    if name == "bunny":
        return os.path.join(ROOT, "scenes", "CBbunny.dae"), "CBbunny.dae", 256
    if name == "cbsl":
        # configuration 1 of the benchmark table with --width 480 --height 360 --light-samples 16
        return os.path.join(ROOT, "scenes", "CBspheres_lambertian.dae"), "CBspheres_lambertian.dae", 64
    if name == "spheres":
        return os.path.join(ROOT, "scenes", "CBspheres.dae"), "CBspheres.dae", 1024
    return name, os.path.basename(name), 1024


def cpu_leg(scene_path, args, spp, threads_port):
    """Reference renderer + CPU restatement on a bounded cell of the same frame."""
    import numpy as np
    import pathtracer_amd as pt
    from pathtracer_amd import cabi, ptar
    import oracle_lib as O
    w, h = args.width, args.height
    cw = max(1, min(args.cpu_cell, w))
    cx = w // 2 - cw // 2
    cpu_spp = args.cpu_spp or spp
    sample = "cell x=%d..%d, y=0..%d of the %dx%d frame at %d spp (-l %d -m %d, adaptive 32/0.05)" % (
        cx, cx + cw, h, w, h, cpu_spp, args.light_samples, args.depth)
    out = {"sample": sample, "host": host_description()}
    # the CPU restatement: counts the reference traversal's node visits / primitive tests
    sc = pt.Scene(scene_path)
    arr = sc.arrays()
    cam = sc.camera(w, h)
    p = cabi.make_params(w, h, ns_aa=cpu_spp, max_ray_depth=args.depth, ns_area_light=args.light_samples,
                         rect=(cx, 0, cx + cw, h), specular=args.specular)
    _, _, st = O.render(arr, cam, p, threads=threads_port)
    ext_rays = st.rays - st.shadow_rays
    out["port"] = {"value": st.samples / st.seconds, "cores": threads_port}
    out["counts"] = {
        "V_ext": (st.nodes_visited - st.shadow_nodes_visited) / max(ext_rays, 1),
        "T_ext": (st.prim_tests - st.shadow_prim_tests) / max(ext_rays, 1),
        "V_shadow": st.shadow_nodes_visited / max(st.shadow_rays, 1),
        "T_shadow": st.shadow_prim_tests / max(st.shadow_rays, 1),
        "rays_per_sample": st.rays / max(st.samples, 1),
        "vertices_per_sample": st.path_vertices / max(st.samples, 1),
        "shadow_fraction": st.shadow_rays / max(st.rays, 1),
    }
    # the reference itself
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if os.path.exists(ref) and not args.specular:  # the reference only has the black stubs
        tmp = os.path.join(ROOT, "gpurun_out", "bench_ref_cell.ptar")
        os.makedirs(os.path.dirname(tmp), exist_ok=True)
        threads_ref = 8  # the thread count of the reference's own benchmark command line
        cmd = [ref, "render", "-s", str(cpu_spp), "-l", str(args.light_samples), "-m", str(args.depth),
               "-t", str(threads_ref), "-r", str(w), str(h), "-p", str(cx), "0", str(cw), str(h), "-f", tmp, scene_path]
        r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
        if r.returncode == 0:
            s = ptar.load(tmp)["stats"]
            out["reference"] = {"value": float(s[3] / s[0]), "cores": threads_ref}
    return out


def host_description():
    """CPU model and core counts of this host (the CPU baseline's machine)."""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    return {"cpu_model": model, "logical_cpus": os.cpu_count(), "cpus_usable_by_this_process": usable}


# keys of a summary's `isa_mix` (tools/isa_mix.py: substrings of the mangled kernel names)
MIX_NAMES = {"extend": "k_extendILb0", "shadow": "k_shadowILb0ELb0", "shade": "k_shadeILb0", "resolve": "k_resolve",
             "trace2": "k_trace2ILb0ELb0", "light": "k_light_resolve", "light_alone": "k_lightE", "light_resolve": "k_light_resolve"}
# Kernels of a timing class (ptamd_stats.kernel_ms). The class "light" times every launch that draws light samples:
# k_light (a round's first depth) and k_light_resolve (a later depth's light sampling with the previous depth's
# k_resolve in the same launch); its counters are the sums over both. "resolve" is what is left of k_resolve, the
# last depth's launch.
KERNEL_NAMES = {"extend": ("k_extend<false>",), "shadow": ("k_shadow<false, false>",), "shade": ("k_shade<false>",),
                "resolve": ("k_resolve",), "trace2": ("k_trace2<false, false>",), "light": ("k_light", "k_light_resolve"),
                "light_alone": ("k_light",), "light_resolve": ("k_light_resolve",)}


def _profile_for(scene, workload):
    """The latest committed counter summary of `scene` whose run was of this very workload (same scene,
    frame size, sample counts and depth: the workload string of the line it recorded), or None."""
    import glob
    for path in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_summary.json" % scene)))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if workload is None or d.get("bench", {}).get("config", {}).get("workload") == workload:
            return path, d
    return None, None


def _pass_sum(d, pmc_pass, kernels, counter):
    """(sum of `counter` over the profiled launches of `kernels`, number of those launches) in one counter pass."""
    rows = [d[pmc_pass][k] for k in kernels if k in d[pmc_pass]]
    if not rows:
        raise KeyError(kernels)
    return sum(r[counter] for r in rows), sum(r["_dispatches"] for r in rows)


def profiled_counters(scene, kernel, workload=None):
    """Per-launch counter sums of the timing class `kernel` from the committed rocprofv3 passes of this workload
    (profiles/r*_<scene>_summary.json, written by tools/profile_round.sh): VALU lane-operations
    (SQ_THREAD_CYCLES_VALU: one count per active lane and VALU instruction), VALU instructions,
    cycles the SIMDs spent issuing them (SQ_ACTIVE_INST_VALU x 4), L2 requests, and the average
    duration of a launch in that profile's kernel trace."""
    path, d = _profile_for(scene, workload)
    if kernel not in KERNEL_NAMES or not path:
        return None
    try:
        ks = KERNEL_NAMES[kernel]
        per = lambda p, c: (lambda t: t[0] / t[1])(_pass_sum(d, p, ks, c))  # noqa: E731
        n = _pass_sum(d, "pmc_sq2", ks, "SQ_THREAD_CYCLES_VALU")[1]
        mix = (d.get("isa_mix") or {}).get(MIX_NAMES.get(kernel, ""), {})
        try:
            from summarize_prof import device_source_hash
            stale = d.get("source_sha256") != device_source_hash(ROOT)
        except Exception:
            stale = None
        traced = [d["kernels"][k] for k in ks if k in d["kernels"]]
        hit, miss = _pass_sum(d, "pmc_l2", ks, "TCC_HIT_sum")[0], _pass_sum(d, "pmc_l2", ks, "TCC_MISS_sum")[0]
        return {"source": os.path.relpath(path, ROOT), "launches": n, "stale": stale,
                "mean_cycles_per_inst": mix.get("mean_cycles"),
                "lane_ops_per_launch": per("pmc_sq2", "SQ_THREAD_CYCLES_VALU"),
                "valu_insts_per_launch": per("pmc_sq", "SQ_INSTS_VALU"),
                "valu_issue_cycles_per_launch": 4.0 * per("pmc_sq", "SQ_ACTIVE_INST_VALU"),
                "gpu_cycles_per_launch": per("pmc_grbm", "GRBM_GUI_ACTIVE") / 8.0,
                "l2_read_requests_per_launch": per("pmc_l1", "TCP_TCC_READ_REQ_sum"),
                "trace_avg_launch_ms": sum(t["total_ms"] for t in traced) / sum(t["calls"] for t in traced),
                "l2_hit": hit / max(hit + miss, 1),
                # rays per launch of the profiled run, when this kernel was the one it reported on
                "rays_per_launch": (d.get("bench", {}).get("roofline", {}).get("rays_per_launch")
                                    if d.get("bench", {}).get("roofline", {}).get("kernel") == kernel else None)}
    except Exception:
        return None


def profiled_traffic(scene, kernel, workload=None):
    """HBM-side bytes per launch of the timing class `kernel` from the committed rocprofv3 PMC passes of this
    workload (profiles/, separate FETCH_SIZE and WRITE_SIZE runs). FETCH_SIZE is in KiB and,
    on gfx950, counts 128-byte read requests as 64 bytes, hence the factor 2."""
    path, d = _profile_for(scene, workload)  # the latest state profiled
    if kernel not in KERNEL_NAMES or not path:
        return None, None
    try:
        f, nf = _pass_sum(d, "pmc_fetch", KERNEL_NAMES[kernel], "FETCH_SIZE")
        w, nw = _pass_sum(d, "pmc_write", KERNEL_NAMES[kernel], "WRITE_SIZE")
        per_launch = (2.0 * f * 1024 / nf) + (w * 1024 / nw)
        return per_launch, os.path.relpath(path, ROOT)
    except Exception:
        return None, None


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    import torch
    import torch.distributed as dist
    import pathtracer_amd as pt
    from pathtracer_amd import cabi, distributed

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    # one rank per GPU; if the launcher masks devices so that each rank sees only its own,
    # LOCAL_RANK no longer names a visible device
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    scene_path, workload_name, default_spp = resolve_scene(args.scene, rank)
    spp = args.spp or default_spp
    if world > 1:
        dist.barrier()
    sc = pt.Scene(scene_path)
    cam = sc.camera(args.width, args.height)
    dev = pt.Device(local_rank)
    dev.upload(sc)
    dev.set_camera(cam)
    if args.max_paths:
        dev.set_max_paths(args.max_paths)
    dev.set_profiling(events=True, counters=False)
    # torch loads a kernel's code object on first use (tens of ms): do that for the
    # reduction that counts a frame's samples before anything is timed
    int(torch.zeros(8, dtype=torch.int32, device=device).sum().item())

    per_pixel = None if args.per_pixel < 0 else bool(args.per_pixel)

    def frame(seed):
        p = cabi.make_params(args.width, args.height, ns_aa=spp, max_ray_depth=args.depth,
                             ns_area_light=args.light_samples, seed=seed, specular=args.specular)
        hdr, cnt = distributed.render_frame(dev, p, rank=rank, world=world, device=device,
                                            pixel_stride=args.pixel_stride, target_paths=args.target_paths,
                                            max_batches=args.max_batches, per_pixel=per_pixel, exchange=args.exchange,
                                            sync_below=None if args.sync_below < 0 else args.sync_below)
        return cnt

    # Warm-up frames time every kernel class with HIP events and so find the dominant one; the timed
    # steps then keep only that class's events (each event pair idles the GPU for ~10 us between two
    # launches, 5 % of a frame when a rank of 8 runs 40 launches per ms-long round).
    NK = len(cabi.K_NAMES)
    warm_ms, dom = None, cabi.K_TRACE2
    for i in range(args.warmup):
        int(frame(1000 + i).sum().item())  # warm-up covers everything a timed step does
        st = dev.stats()
        warm_ms = {cabi.K_NAMES[k]: st.kernel_ms[k] for k in range(NK) if st.launches[k]}
        dom = max(range(NK), key=lambda k: st.kernel_ms[k])
    if args.warmup:
        dev.set_event_classes(1 << dom)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total_samples = 0
    agg = {"ms": [0.0] * NK, "launches": [0] * NK, "rays": 0, "shadow": 0, "vertices": 0, "two_ext": 0, "two_shadow": 0,
           "unused": 0, "traced": 0}
    step_ms = []
    for i in range(args.steps):
        ts = time.perf_counter()
        cnt = frame(i)
        total_samples += int(cnt.sum().item())
        wall = 1000.0 * (time.perf_counter() - ts)
        st = dev.stats()  # this rank's share of the frame
        step_ms.append((wall, sum(st.kernel_ms)))
        for k in range(NK):
            agg["ms"][k] += st.kernel_ms[k]
            agg["launches"][k] += st.launches[k]
        agg["rays"] += st.rays
        agg["shadow"] += st.shadow_rays
        agg["vertices"] += st.path_vertices
        agg["two_ext"] += st.two_ext_rays
        agg["two_shadow"] += st.two_shadow_rays
        agg["unused"] += st.unused_slots
        agg["traced"] += st.samples
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        print("step ms (wall, kernels): " + " ".join("%.1f/%.1f" % t for t in step_ms), file=sys.stderr)
        value = total_samples / elapsed
        line = {
            "metric": "path samples/sec",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / max(args.steps, 1),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s %dx%d, %d spp, -l %d -m %d, adaptive 32/0.05%s" % (
                workload_name, args.width, args.height, spp, args.light_samples, args.depth,
                ", specular mode" if args.specular else ""),
                "samples_per_step": total_samples // max(args.steps, 1),
                # rank 0: samples its pixels kept, and path slots of its launches that held no sample
                "samples_kept_by_rank0_per_step": agg["traced"] // max(args.steps, 1),
                "unused_path_slots_per_step": agg["unused"] // max(args.steps, 1),
                "parallelism": ("pixels dealt round-robin over %d GPU(s), " % world) +
                               ("once per frame, one all-gather at its end" if args.exchange == "frame" else
                                "again after every adaptive round, one all-gather per round")},
        }
        dev.set_event_classes(0xffffffff)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_leg(scene_path, args, spp, threads_port=min(16, os.cpu_count() or 1))
            except Exception as e:  # the baseline must never take the GPU number down with it
                cpu = {"error": repr(e)}
        # ---- the GPU walk's own work per ray (untimed extra frame with the traversal counters on)
        own = None
        if world == 1:
            try:
                dev.set_profiling(events=False, counters=True)
                pc = cabi.make_params(args.width, args.height, ns_aa=min(spp, 64), max_ray_depth=args.depth,
                                      ns_area_light=args.light_samples, seed=77, specular=args.specular)
                distributed.render_frame(dev, pc, device=device, pixel_stride=args.pixel_stride,
                                         per_pixel=per_pixel)
                sc_ = dev.stats()
                if sc_.rays:
                    own = {"nodes_per_ray": sc_.nodes_visited / 4.0 / sc_.rays, "prims_per_ray": sc_.prim_tests / sc_.rays}
                    own["fp64_tests_per_prim_test"] = sc_.exact_tests / max(sc_.prim_tests, 1)
                    # its records: 64-byte 4-wide node, 48-byte primitive (80 more when the FP32 test is
                    # undecided), the same 92 bytes of ray and hit
                    own["bytes_per_ray"] = (64 * own["nodes_per_ray"] + (48 + 80 * own["fp64_tests_per_prim_test"]) * own["prims_per_ray"] + 92)
            except Exception as e:
                own = {"error": repr(e)}
        # ---- roofline of the dominant kernel (rank 0's launches)
        names = cabi.K_NAMES
        if not args.warmup:
            dom = max(range(NK), key=lambda k: agg["ms"][k])
        roof = {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None,
                "traffic": None, "kernel": names[dom],
                "kernel_ms": {names[k]: agg["ms"][k] for k in range(NK) if agg["ms"][k]},
                "launches": {names[k]: agg["launches"][k] for k in range(NK) if agg["launches"][k]},
                "kernel_ms_all_classes_warmup_frame": warm_ms}
        ms = agg["ms"][dom]
        n_launch = max(agg["launches"][dom], 1)
        roof["avg_launch_ms"] = ms / n_launch
        ext_rays = agg["rays"] - agg["shadow"]
        rays = None
        if dom in (1, 3, cabi.K_TRACE2):
            # rays the dominant traversal kernel traced: k_extend the bounce and camera rays that did
            # not go through k_trace2, k_shadow likewise, k_trace2 rays of both kinds in one launch
            if dom == 1:
                rays = ext_rays - agg["two_ext"]
            elif dom == 3:
                rays = agg["shadow"] - agg["two_shadow"]
            else:
                rays = agg["two_ext"] + agg["two_shadow"]
                roof["rays_per_launch_by_kind"] = {"bounce": agg["two_ext"] / n_launch, "shadow": agg["two_shadow"] / n_launch}
            roof["rays_per_launch"] = rays / n_launch
        # counters are taken from a committed profile only if it is of this very workload
        standard = not args.specular and os.path.sep not in args.scene
        wl = line["config"]["workload"]
        # ---- the binding ceiling of a BVH walk: VALU instruction issue. Lane-operations per launch come
        # from the committed counter passes of this same command line, the time per launch is measured
        # live (HIP events on the launch stream).
        pc = profiled_counters(args.scene, names[dom], wl) if standard else None
        if pc and (world > 1 or args.pixel_stride > 1):
            # the profile is of the whole frame on one GPU; a share of the frame has smaller launches. A
            # traversal kernel's work is proportional to its rays, which are counted live on both sides
            live, prof = roof.get("rays_per_launch"), pc.get("rays_per_launch")
            if live and prof:
                scale = live / prof
                for k in ("lane_ops_per_launch", "valu_insts_per_launch", "valu_issue_cycles_per_launch",
                          "gpu_cycles_per_launch", "l2_read_requests_per_launch"):
                    pc[k] *= scale
                pc["scaled_by_rays"] = scale
            else:
                pc = None
        if pc and ms > 0:
            t_launch = ms * 1e-3 / n_launch
            ach = pc["lane_ops_per_launch"] / t_launch / 1e12
            roof.update({"bound": "valu-issue", "achieved": ach, "peak": LANE_OPS_PEAK_T, "unit": "Tlane-op/s",
                         "frac": ach / LANE_OPS_PEAK_T})
            roof["issue"] = {
                "definition": "VALU lane-operations (SQ_THREAD_CYCLES_VALU) per launch / live time per launch, against "
                              "%d SIMDs x %d lanes x %.1f GHz (full-rate wave64 instruction every %g cycles, measured: "
                              "profiles/r03_valu_rates.txt)" % (N_SIMD, SIMD_LANES, CLOCK_HZ / 1e9, FULL_RATE_CYCLES),
                "lane_ops_per_launch": pc["lane_ops_per_launch"],
                # (per ray only where the kernel traces rays)
                "lane_ops_per_ray": pc["lane_ops_per_launch"] / roof["rays_per_launch"] if roof.get("rays_per_launch") else None,
                "valu_insts_per_ray": 64.0 * pc["valu_insts_per_launch"] / roof["rays_per_launch"] if roof.get("rays_per_launch") else None,
                # VALU instructions per SIMD cycle of the profiled launch (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs),
                # and of the 64 lanes of those instructions the share that was switched on
                "valu_insts_per_simd_cycle": pc["valu_insts_per_launch"] / (pc["gpu_cycles_per_launch"] * N_SIMD),
                "lanes_active": pc["lane_ops_per_launch"] / (64.0 * pc["valu_insts_per_launch"]),
                "profiled_avg_launch_ms": pc["trace_avg_launch_ms"], "source": pc["source"]}
            if pc.get("mean_cycles_per_inst"):
                # the kernel's instructions weighted by the cycles their issue class takes (static mix of the
                # compiled kernel, tools/isa_mix.py): the share of SIMD cycles the VALU was busy, and the share
                # of the full-rate peak this mix could reach with the VALU always busy and every lane on.
                # frac = valu_busy x mix_allows x lanes_active x (profiled clock / 2.4 GHz)
                mc = pc["mean_cycles_per_inst"]
                roof["issue"]["mean_cycles_per_valu_inst"] = mc
                roof["issue"]["valu_busy"] = roof["issue"]["valu_insts_per_simd_cycle"] * mc
                roof["issue"]["mix_allows"] = FULL_RATE_CYCLES / mc
            roof["issue"]["profiled_clock_ghz"] = pc["gpu_cycles_per_launch"] / (pc["trace_avg_launch_ms"] * 1e-3) / 1e9
            if pc.get("stale") is not None:
                # the counters were collected from other device sources than the ones this run was built from
                roof["profile_stale"] = bool(pc["stale"])
            if "scaled_by_rays" in pc:
                roof["issue"]["profile_scaled_by_rays_per_launch"] = pc["scaled_by_rays"]
            roof["l2_measured"] = {"read_requests_per_launch": pc["l2_read_requests_per_launch"], "hit_rate": pc["l2_hit"],
                                   "achieved_gbs_at_64B": pc["l2_read_requests_per_launch"] * 64 / t_launch / 1e9,
                                   "peak": L2_PEAK_GBS}
        if standard and world == 1 and args.pixel_stride == 1:
            tr, src = profiled_traffic(args.scene, names[dom], wl)
            roof["traffic"], roof["traffic_source"] = tr, src
            if tr and ms > 0:
                g = tr / (ms * 1e-3 / n_launch) / 1e9
                roof["hbm_measured"] = {"achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g / HBM_PEAK_GBS}
        if cpu and "counts" in cpu:
            c = cpu["counts"]
            b_ext = 32 * c["V_ext"] + 36 * c["T_ext"] + 92
            b_sh = 32 * c["V_shadow"] + 36 * c["T_shadow"] + 92
            if rays is not None and ms > 0:
                if dom == 1:
                    b = b_ext
                elif dom == 3:
                    b = b_sh
                else:
                    b = (agg["two_ext"] * b_ext + agg["two_shadow"] * b_sh) / max(rays, 1)
                ach = rays * b / (ms * 1e-3) / 1e9
                # the definition of the survey: bytes of the REFERENCE's traversal for these rays. Its
                # fraction passes 1 because the GPU walk visits far fewer nodes; not a utilisation.
                roof["hbm_algorithmic"] = {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                           "bytes_per_ray": b, "algorithmic_bytes_per_launch": rays * b / n_launch}
                if roof["bound"] is None:  # no counter passes for this workload: the survey's definition stands alone
                    roof.update({"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS})
            # whole-frame figure: samples/s x algorithmic bytes per sample, the per-sample ray
            # and vertex counts being the reference traversal's (rays the GPU traced ahead of an
            # adaptive test and then dropped do not count)
            sf = c["shadow_fraction"]
            b_sample = c["rays_per_sample"] * ((1 - sf) * b_ext + sf * b_sh) + 72 * c["vertices_per_sample"] + 16
            roof["bytes_per_sample"] = b_sample
            roof["frame_frac_of_hbm_peak"] = value * b_sample / 1e9 / (HBM_PEAK_GBS * world)
            roof["counts"] = c
        if own:
            if "bytes_per_ray" in own and rays is not None and ms > 0:
                # the same launches priced by what the GPU walk itself fetches, against the L2s that serve it
                g = rays * own["bytes_per_ray"] / (ms * 1e-3) / 1e9
                roof["l2_walk"] = {"achieved": g, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": g / L2_PEAK_GBS,
                                   "bytes_per_ray": own["bytes_per_ray"]}
            roof["gpu_walk"] = own
        line["roofline"] = roof
        if cpu and "reference" in cpu:
            line["cpu_baseline"] = {"value": cpu["reference"]["value"], "unit": "samples/s",
                                    "cores": cpu["reference"]["cores"], "kind": "reference", "sample": cpu["sample"],
                                    "port_value": cpu["port"]["value"], "port_cores": cpu["port"]["cores"],
                                    "host": cpu["host"]}
        elif cpu and "port" in cpu:
            line["cpu_baseline"] = {"value": cpu["port"]["value"], "unit": "samples/s", "cores": cpu["port"]["cores"],
                                    "kind": "port", "sample": cpu["sample"], "host": cpu["host"]}
        elif cpu:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
