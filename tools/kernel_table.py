#!/usr/bin/env python3
"""Per-kernel table of the committed counter summaries (profiles/r*_<workload>_summary.json, the latest of each
workload): time per frame, HBM-side traffic, L2 hit rate, VALU busy, lanes on, lane-operations per second.
usage: kernel_table.py [workload ...]   (default: dragon bunny lucy spheres folded cbsl)"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402

ROWS = (("k_trace2<false, false>", "k_trace2ILb0ELb0"), ("k_shadow<false, false>", "k_shadowILb0ELb0"), ("k_extend<false>", "k_extendILb0"),
        ("k_shade<false>", "k_shadeILb0"), ("k_light_resolve", "k_light_resolve"), ("k_light", "k_lightE"), ("k_resolve", "k_resolve"))


def main():
    for wl in (sys.argv[1:] or ["dragon", "bunny", "lucy", "spheres", "folded", "cbsl"]):
        found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_summary.json" % wl)))
        if not found:
            continue
        d = json.load(open(found[-1]))
        b = d.get("bench", {})
        frames = b.get("steps", 3) + b.get("warmup", 1)  # the profiled run's frames (its short counters frame adds little)
        # (the run's own roofline line is not quoted: it priced its launches with the counters of the summaries
        # present when it ran, i.e. of the set before; the rows below are this set's)
        print("== %s (%s): %.1f ms/frame %.4g samples/s; dominant class %s" % (
            wl, os.path.basename(found[-1]), b.get("ms_per_step", 0), b.get("value", 0), b.get("roofline", {}).get("kernel")))
        for k, mixk in ROWS:
            if k not in d["kernels"] or k not in d["pmc_sq"]:
                continue
            t_ms = d["kernels"][k]["total_ms"]
            cyc = d["pmc_grbm"][k]["GRBM_GUI_ACTIVE"] / 8.0
            insts = d["pmc_sq"][k]["SQ_INSTS_VALU"]
            mc = (d.get("isa_mix") or {}).get(mixk, {}).get("mean_cycles") or 0
            lane_ops = d["pmc_sq2"][k]["SQ_THREAD_CYCLES_VALU"]
            hbm = (2 * d["pmc_fetch"][k]["FETCH_SIZE"] + d["pmc_write"][k]["WRITE_SIZE"]) * 1024 / (t_ms * 1e-3) / 1e12
            hit = d["pmc_l2"][k]["TCC_HIT_sum"] / max(d["pmc_l2"][k]["TCC_HIT_sum"] + d["pmc_l2"][k]["TCC_MISS_sum"], 1)
            lo = lane_ops / (t_ms * 1e-3) / 1e12
            print("  %-24s %7.2f ms/frame  avg %7.1f us  HBM-side %5.2f TB/s (%.2f)  L2 hit %.2f  VALU busy %.2f  lanes on %.2f  lane-op %4.1f T/s (%.3f of %.1f)" % (
                k, t_ms / frames, d["kernels"][k]["avg_us"], hbm, hbm / 8.0, hit, insts / (cyc * bench.N_SIMD) * mc,
                lane_ops / (64.0 * insts), lo, lo / bench.LANE_OPS_PEAK_T, bench.LANE_OPS_PEAK_T))


if __name__ == "__main__":
    main()
