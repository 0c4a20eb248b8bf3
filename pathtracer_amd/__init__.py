"""MI355X-native render hot path of the CS184 path tracer (ctypes binding of libptamd.so).

The package is a thin binding: the product is the C-ABI library built from
``pathtracer_amd/csrc`` (host scene pipeline in C++, wavefront path tracer in
HIP for gfx950). There is no CPU fallback: anything that renders raises
``PtamdError`` when the library or a GPU is missing.
"""
from .api import Device, MultiDevice, PtamdError, Scene, lib, lib_multi, make_params  # noqa: F401
from . import cabi, ptar  # noqa: F401

__all__ = ["Device", "MultiDevice", "lib_multi", "PtamdError", "Scene", "lib", "make_params", "cabi", "ptar"]
