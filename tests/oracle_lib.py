"""ctypes loader for the CPU oracle (oracle/_build/liboracle.so). Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from pathtracer_amd import cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


class OracleStats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("samples", C.c_uint64), ("rays", C.c_uint64),
                ("shadow_rays", C.c_uint64), ("path_vertices", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("prim_tests", C.c_uint64),
                ("shadow_nodes_visited", C.c_uint64), ("shadow_prim_tests", C.c_uint64), ("threads", C.c_int32)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
        _LIB = C.CDLL(path)
        pd = C.POINTER(C.c_double)
        pi = C.POINTER(C.c_int32)
        _LIB.oracle_render.argtypes = [C.POINTER(cabi.SceneDesc), C.POINTER(cabi.Camera), C.POINTER(cabi.RenderParams),
                                       C.c_int32, pd, pi, C.POINTER(OracleStats)]
        _LIB.oracle_trace_primary.argtypes = [C.POINTER(cabi.SceneDesc), C.POINTER(cabi.Camera), C.c_int64, pd, pd, pi, pd]
        _LIB.oracle_sample_radiance.argtypes = [C.POINTER(cabi.SceneDesc), C.POINTER(cabi.Camera),
                                                C.POINTER(cabi.RenderParams), C.c_int32, C.c_int32, C.c_int32, pd]
        for name in ("oracle_kat_bbox", "oracle_kat_triangle", "oracle_kat_sphere", "oracle_kat_frame"):
            getattr(_LIB, name).argtypes = [C.c_int64, pd, pd]
        _LIB.oracle_kat_tonemap.argtypes = [C.c_int64, pd, pi]
        _LIB.oracle_kat_specular.argtypes = [C.c_int64, pi, pd, pd, pd, pd, pd, pd, pd]
        _LIB.oracle_philox.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
    return _LIB


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def render(scene, cam, params, threads=1):
    """scene: cabi.SceneArrays. Returns (hdr[h,w,3], count[h,w], OracleStats)."""
    hdr = np.zeros(params.w * params.h * 3, np.float64)
    cnt = np.zeros(params.w * params.h, np.int32)
    st = OracleStats()
    rc = lib().oracle_render(C.byref(scene.desc), C.byref(cam), C.byref(params), threads, _pd(hdr), _pi(cnt), C.byref(st))
    assert rc == 0
    return hdr.reshape(params.h, params.w, 3), cnt.reshape(params.h, params.w), st


def trace_primary(scene, cam, sx, sy):
    sx = np.ascontiguousarray(sx, np.float64)
    sy = np.ascontiguousarray(sy, np.float64)
    ids = np.zeros(sx.size, np.int32)
    ts = np.zeros(sx.size, np.float64)
    rc = lib().oracle_trace_primary(C.byref(scene.desc), C.byref(cam), sx.size, _pd(sx), _pd(sy), _pi(ids), _pd(ts))
    assert rc == 0
    return ids, ts


def kat(name, inp, out_per, out_dtype=np.float64, in_per=None):
    inp = np.ascontiguousarray(inp, np.float64)
    n = inp.size // in_per
    out = np.zeros(n * out_per, out_dtype)
    fn = getattr(lib(), "oracle_kat_" + name)
    fn(n, _pd(inp), _pi(out) if out_dtype == np.int32 else _pd(out))
    return out


def philox(c, k):
    out = (C.c_uint32 * 4)()
    lib().oracle_philox(c[0], c[1], c[2], c[3], k[0], k[1], out)
    return list(out)


def kat_specular(kind, param3, extra4, wo, u):
    """The specular-mode BSDFs on scripted inputs: returns (wi[n,3], f[n,3], pdf[n])."""
    kind = np.ascontiguousarray(kind, np.int32)
    n = kind.size
    param3 = np.ascontiguousarray(param3, np.float64).reshape(n, 3)
    extra4 = np.ascontiguousarray(extra4, np.float64).reshape(n, 4)
    wo = np.ascontiguousarray(wo, np.float64).reshape(n, 3)
    u = np.ascontiguousarray(u, np.float64).reshape(n)
    wi = np.zeros((n, 3)); f = np.zeros((n, 3)); pdf = np.zeros(n)
    lib().oracle_kat_specular(n, _pi(kind), _pd(param3), _pd(extra4), _pd(wo), _pd(u), _pd(wi), _pd(f), _pd(pdf))
    return wi, f, pdf
