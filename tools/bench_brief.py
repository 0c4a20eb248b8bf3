#!/usr/bin/env python3
"""Prints the interesting part of a bench.py JSON line."""
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline", {})
print("%.4g samples/s  %.1f ms/step  [%s]" % (d["value"], d["ms_per_step"], d["config"]["workload"][:40]))
print(" kernel_ms:", {k: round(v, 1) for k, v in r.get("kernel_ms", {}).items()})
print(" dominant=%s bound=%s achieved=%s %s frac=%s" % (r.get("kernel"), r.get("bound"), r.get("achieved"), r.get("unit"), r.get("frac")))
if "cpu_baseline" in d:
    print(" cpu:", {k: v for k, v in d["cpu_baseline"].items() if k != "sample"})
