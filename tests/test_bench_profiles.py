"""The benchmark's roofline figures are recomputed from the counter summaries committed under
profiles/: the files bench.py will pick up must hold what it reads, and the figure they give
must be a fraction. CPU only."""
import glob
import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.mark.parametrize("scene", ["dragon", "bunny", "lucy", "spheres", "folded"])
def test_latest_profile_summary_feeds_the_roofline(scene):
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_summary.json" % scene)))
    assert found, "no committed profile summary for %s" % scene
    d = json.load(open(found[-1]))
    assert len(d.get("source_sha256", "")) == 64, "the summary does not say which device sources it was collected from"
    for kernel in ("shadow", "trace2", "extend", "shade", "light", "light_resolve", "resolve"):
        pc = bench.profiled_counters(scene, kernel)
        assert isinstance(pc["stale"], bool) and pc["mean_cycles_per_inst"] and 2.0 <= pc["mean_cycles_per_inst"] <= 16.0
        assert pc is not None and pc["source"].endswith(os.path.basename(found[-1]))
        t = pc["trace_avg_launch_ms"] * 1e-3
        frac = pc["lane_ops_per_launch"] / t / 1e12 / bench.LANE_OPS_PEAK_T  # against 1024 SIMDs x 32 lanes x 2.4 GHz
        assert 0 < frac <= 1, (kernel, frac)
        # at most 64 lanes per instruction (the two counters come from separate runs: 1 % of slack)
        assert 0 < pc["lane_ops_per_launch"] <= 64 * 1.01 * pc["valu_insts_per_launch"]
        tr, src = bench.profiled_traffic(scene, kernel)
        assert tr and tr > 0 and src == pc["source"]
        # measured HBM traffic below the 8 TB/s the card can move in that time
        assert tr / t / 1e9 < bench.HBM_PEAK_GBS
    # the run the summary was made from reported the kernel the counters are scaled by
    assert d["bench"]["roofline"]["kernel"] in bench.KERNEL_NAMES
    assert d["bench"]["roofline"].get("rays_per_launch", 0) > 0 or d["bench"]["roofline"]["kernel"] in ("shade", "resolve", "light", "light_resolve")


def test_a_profile_of_another_workload_is_not_used():
    """Counters are scaled to the live run only when the committed profile is of the very same workload."""
    assert bench.profiled_counters("dragon", "trace2", "some other workload") is None
    d = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_dragon_summary.json")))[-1]))
    assert bench.profiled_counters("dragon", "trace2", d["bench"]["config"]["workload"]) is not None
