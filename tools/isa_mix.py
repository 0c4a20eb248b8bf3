#!/usr/bin/env python3
"""Static VALU instruction mix of the device layer's kernels, by issue class.

    python3 tools/isa_mix.py [--src unit.hip ...] [--json out.json] [kernel substrings...]

Compiles the device layer's kernel units for gfx950 with --save-temps (in a scratch directory) and counts, per
kernel, the VALU instructions of its body by the issue class the cycle-stamped microbenchmark
(tools/ubench/valu_rates.hip) found on MI355X at two or more waves per SIMD:

  full   2 cycles per wave64 instruction: v_fma_f32, v_fmac_f32, v_add_f32, v_sub_f32, v_mul_f32,
         v_add_u32, v_sub_u32, v_and_b32, v_or_b32, v_xor_b32, v_mov_b32
  half   4 cycles: every other 32-bit VALU instruction measured (v_max/min/min3/max3/med3_f32,
         v_cmp_*, v_cndmask_b32, v_cvt_*, shifts, v_bfe_u32, v_ldexp_f32, v_mad_u32_u24,
         v_mul_lo/hi_u32) and all FP64 arithmetic, v_pk_fma_f32, 64-bit integer adds
  quarter 8 cycles: v_rcp_f32 and the other FP32 transcendentals
  f64t   16 cycles: v_rcp_f64, v_rsq_f64, v_sqrt_f64

The count is static (every instruction of the body once), so it describes the kernel's loop only
as far as the loop dominates the body, which it does for the traversal kernels. `mean_cycles` is
the class-weighted cost of one instruction; 2 / mean_cycles is the share of the full-rate VALU
peak that a kernel of this mix could reach at most.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = {"v_fma_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_not_b32", "v_add_co_u32", "v_sub_co_u32"}
QUARTER = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}
F64T = {"v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"}
CYCLES = {"full": 2.0, "half": 4.0, "quarter": 8.0, "f64t": 16.0}


def issue_class(op):
    op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op in FULL:
        return "full"
    if op in QUARTER:
        return "quarter"
    if op in F64T:
        return "f64t"
    return "half"


def compile_asm(src):
    tmp = tempfile.mkdtemp(prefix="isa_mix_")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-fno-gpu-rdc", "-Wno-unused-parameter", "--save-temps", "-c", os.path.abspath(src), "-o", os.path.join(tmp, "dev.o")]
    subprocess.check_call(cmd, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in os.listdir(tmp):
        if f.endswith("gfx950.s"):
            return os.path.join(tmp, f)
    raise SystemExit("no gfx950 assembly produced")


def kernels(asm_path):
    lines = open(asm_path).read().split("\n")
    out = {}
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if m:
            j = i + 1
            while j < len(lines) and not lines[j].strip().startswith("s_endpgm") and not lines[j].startswith(".Lfunc_end"):
                j += 1
            out[m.group(1)] = lines[i + 1:j]
            i = j
        i += 1
    return out


def mix(body):
    ops = collections.Counter()
    for ln in body:
        m = re.match(r"\s+(v_\w+)", ln)
        if m:
            ops[m.group(1)] += 1
    cls = collections.Counter()
    for op, n in ops.items():
        cls[issue_class(op)] += n
    total = sum(cls.values())
    mean = sum(CYCLES[c] * n for c, n in cls.items()) / max(total, 1)
    f64 = sum(n for op, n in ops.items() if "f64" in op or "u64" in op or "i64" in op)
    return {"valu": total, "by_class": dict(cls), "f64_or_64bit": f64, "mean_cycles": mean,
            "share_of_full_rate_peak": 2.0 / mean if total else None, "top": ops.most_common(12)}


def main():
    ap = argparse.ArgumentParser()
    csrc = os.path.join(ROOT, "pathtracer_amd", "csrc")
    ap.add_argument("--src", nargs="*", default=[os.path.join(csrc, u) for u in ("trace_kernels.hip", "shade_kernels.hip")])
    ap.add_argument("--asm", nargs="*", default=None, help="existing gfx950 .s files instead of compiling")
    ap.add_argument("--json", default=None)
    ap.add_argument("names", nargs="*", default=["k_trace2ILb0ELb0", "k_shadowILb0ELb0", "k_extendILb0", "k_shadeILb0", "k_light_resolve", "k_lightE", "k_resolve"])
    a = ap.parse_args()
    ks = {}
    for f in (a.asm or [compile_asm(u) for u in a.src]):
        ks.update(kernels(f))
    res = {}
    for sub in a.names:
        for name, body in ks.items():
            if sub in name:
                short = re.search(r"(k_\w+?)(I[\w]*?E)?Ev", name)
                res[sub] = mix(body)
                r = res[sub]
                print("%-20s VALU %5d  %s  64-bit %d  mean %.2f cycles -> at most %.2f of the full-rate peak" % (
                    sub, r["valu"], r["by_class"], r["f64_or_64bit"], r["mean_cycles"], r["share_of_full_rate_peak"]))
                break
    if a.json:
        for r in res.values():
            r["top"] = [list(t) for t in r["top"]]
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
