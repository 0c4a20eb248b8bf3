"""A process may use the library first and torch afterwards (or the other way round): both must end up on one
HIP runtime. Run in a fresh interpreter, because the order of the first loads is the point."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_LIBRARY_FIRST = """
import sys
sys.path.insert(0, %r)
import pathtracer_amd as pt
from pathtracer_amd import cabi
assert "torch" not in sys.modules
dev = pt.Device(0)
sc = pt.Scene(%r)
dev.upload(sc)
dev.set_camera(sc.camera(16, 12))
hdr, cnt = dev.render(cabi.make_params(16, 12, ns_aa=4, max_ray_depth=2))
import torch
assert torch.cuda.is_available(), "torch does not find the GPU after the library has used it"
x = torch.arange(8, device="cuda").sum().item()
hdr2, cnt2 = dev.render(cabi.make_params(16, 12, ns_aa=4, max_ray_depth=2))
assert x == 28 and (hdr == hdr2).all() and (cnt == cnt2).all()
print("both work")
"""


def test_library_before_torch_in_one_process():
    code = _LIBRARY_FIRST % (ROOT, os.path.join(ROOT, "scenes", "cube.dae"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "both work" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
