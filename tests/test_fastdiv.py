"""The exact division-by-invariant used to map a light-sample slot to its vertex
(pathtracer_amd/csrc/fastdiv.h), checked on the CPU against the hardware division."""
import os
import subprocess

from conftest import ROOT

SRC = r"""
#include <cstdio>
#include <cstdint>
#include <random>
#include "fastdiv.h"
int main() {
  std::mt19937_64 rng(1);
  const uint32_t lim = 0x7fffffffu;
  long checked = 0;
  for (uint32_t d = 1; d < 65536; d = d < 300 ? d + 1 : d + 1 + d / 7) {
    const ptdev::FastDiv f = ptdev::make_fastdiv(d);
    if (f.m < 0x80000000u) { printf("m too small for d=%u\n", d); return 1; }
    auto chk = [&](uint32_t n) {
      checked++;
      if (f.div(n) != n / d) { printf("d=%u n=%u: %u != %u\n", d, n, f.div(n), n / d); return false; }
      return true;
    };
    // multiples of d and their neighbours near the top of the range, the ends, random values
    const uint32_t top = (lim / d) * d;
    for (uint32_t k = 0; k < 64; k++) {
      const uint32_t base = top - k * d;
      if (!chk(base) || (base > 0 && !chk(base - 1)) || (base < lim && !chk(base + 1))) return 1;
    }
    if (!chk(0) || !chk(1) || !chk(lim) || !chk(lim - 1) || !chk(1u << 23) || !chk((1u << 24) + 1)) return 1;
    for (int k = 0; k < 2000; k++) if (!chk((uint32_t)(rng() & lim))) return 1;
  }
  // the case that a float quotient gets wrong
  if (ptdev::make_fastdiv(3).div(46112846u) != 15370948u) return 1;
  printf("ok %ld\n", checked);
  return 0;
}
"""


def test_fastdiv_is_exact(tmp_path):
    src = tmp_path / "fastdiv_check.cpp"
    src.write_text(SRC)
    exe = tmp_path / "fastdiv_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "pathtracer_amd", "csrc"),
                           str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert out.stdout.startswith("ok")
