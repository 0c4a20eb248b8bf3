#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/pmc_passes.sh into one small JSON/markdown summary."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def kernel_stats(d):
    out = {}
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[short(r["Name"])] = {"calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
                                     "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])}
    return out


def counters(d, name):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for f in glob.glob(os.path.join(d, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    return {k: dict(v, _dispatches=max(n[(k, c)] for c in v)) for k, v in acc.items()}


def device_source_hash(root=None):
    """SHA-256 over the device layer's sources (pathtracer_amd/csrc/*.hip and *.h, by name): a counter
    summary is only valid for the kernels it was collected from, and bench.py compares this with the
    tree it runs in."""
    import hashlib
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "pathtracer_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(src)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(src, f), "rb").read())
    return h.hexdigest()


def main():
    d = sys.argv[1]
    res = {"kernels": kernel_stats(d), "source_sha256": device_source_hash()}
    try:  # static VALU instruction mix by issue class (tools/isa_mix.py --json <outdir>/isa_mix.json)
        res["isa_mix"] = json.load(open(os.path.join(d, "isa_mix.json")))
    except Exception:
        pass
    for p in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq", "pmc_grbm", "pmc_sq2", "pmc_l1"):
        res[p] = counters(d, p)
    try:
        res["bench"] = json.load(open(os.path.join(d, "trace.json")))
    except Exception:
        pass
    json.dump(res, open(os.path.join(d, "summary.json"), "w"), indent=1)
    print("kernel              calls   total_ms   avg_us    pct")
    for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["total_ms"])[:12]:
        print("%-34s %6d %10.2f %9.1f %6.2f" % (k, v["calls"], v["total_ms"], v["avg_us"], v["pct"]))
    for k in ("k_shadow<false, false>", "k_trace2<false, false>", "k_extend<false>", "k_shade<false>", "k_light_resolve", "k_light", "k_resolve"):
        row = {}
        for p in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq", "pmc_grbm", "pmc_sq2", "pmc_l1"):
            row.update({c: v for c, v in res[p].get(k, {}).items() if not c.startswith("_")})
        if row:
            hit = row.get("TCC_HIT_sum", 0) / max(row.get("TCC_HIT_sum", 0) + row.get("TCC_MISS_sum", 0), 1)
            print(k, {c: "%.4g" % v for c, v in row.items()}, "L2 hit %.3f" % hit)


if __name__ == "__main__":
    main()
