"""ctypes binding of the ptamd C ABI (include/ptamd.h)."""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

from . import cabi
from .cabi import Camera, PixelList, RenderParams, SceneDesc, Stats, make_params  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# every symbol include/ptamd.h declares
EXPORTS = (
    "ptamd_last_error", "ptamd_scene_load", "ptamd_scene_dump_camera_file", "ptamd_scene_free", "ptamd_scene_get_desc",
    "ptamd_scene_get_camera", "ptamd_scene_load_camera_file", "ptamd_scene_build_seconds",
    "ptamd_create", "ptamd_destroy", "ptamd_upload_scene", "ptamd_set_camera", "ptamd_set_max_paths",
    "ptamd_set_profiling", "ptamd_set_event_classes", "ptamd_render", "ptamd_trace_primary", "ptamd_frame_begin",
    "ptamd_trace_pixels", "ptamd_adaptive_filter", "ptamd_frame_resolve", "ptamd_get_stats",
    "ptamd_trace_list", "ptamd_filter_list", "ptamd_filter_list_static", "ptamd_render_list", "ptamd_pack_rows", "ptamd_unpack_rows",
    "ptamd_tonemap_rgba8", "ptamd_write_png", "ptamd_write_png_top_down", "ptamd_sampling_rate_rgba8",
)


class PtamdError(RuntimeError):
    pass


def lib_path():
    """PTAMD_LIB selects another build of the same library (kernel A/B runs)."""
    return os.environ.get("PTAMD_LIB") or os.path.join(_HERE, "libptamd.so")


def _share_torch_hip_runtime():
    """One HIP runtime per process, whatever the import order.

    torch's ROCm wheels carry their own libamdhip64.so.7 and libhsa-runtime64.so.1, found through the RPATH of
    torch's libraries; libptamd.so names the same sonames and finds ROCm's. The dynamic loader keeps whichever
    copy came first for both users, and torch does not find the GPU through ROCm's copy once libptamd.so has
    initialised it ("No HIP GPUs are available"), while libptamd.so works with either. So where torch is
    installed its copy is loaded first, without importing torch (PTAMD_OWN_HIP_RUNTIME=1 leaves that out, for
    processes that never use torch and want the HIP runtime libptamd.so was linked against)."""
    if "torch" in sys.modules or os.environ.get("PTAMD_OWN_HIP_RUNTIME"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib():
    """Loads libptamd.so; raises PtamdError if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise PtamdError("%s is missing: build it with `make -C pathtracer_amd/csrc` "
                         "(there is no CPU fallback for the render path)" % path)
    _share_torch_hip_runtime()
    L = C.CDLL(path)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    L.ptamd_last_error.restype = C.c_char_p
    L.ptamd_scene_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.ptamd_scene_free.argtypes = [vp]
    L.ptamd_scene_free.restype = None
    L.ptamd_scene_get_desc.argtypes = [vp, C.POINTER(SceneDesc)]
    L.ptamd_scene_get_camera.argtypes = [vp, i32, i32, C.POINTER(Camera)]
    L.ptamd_scene_load_camera_file.argtypes = [vp, C.c_char_p]
    L.ptamd_scene_dump_camera_file.argtypes = [vp, C.c_char_p]
    L.ptamd_scene_build_seconds.argtypes = [vp]
    L.ptamd_scene_build_seconds.restype = dbl
    L.ptamd_create.argtypes = [i32, C.POINTER(vp)]
    L.ptamd_destroy.argtypes = [vp]
    L.ptamd_destroy.restype = None
    L.ptamd_upload_scene.argtypes = [vp, C.POINTER(SceneDesc)]
    L.ptamd_set_camera.argtypes = [vp, C.POINTER(Camera)]
    L.ptamd_set_max_paths.argtypes = [vp, i64]
    L.ptamd_set_profiling.argtypes = [vp, i32, i32]
    L.ptamd_set_event_classes.argtypes = [vp, C.c_uint32]
    L.ptamd_render.argtypes = [vp, C.POINTER(RenderParams), pd, pi]
    L.ptamd_trace_primary.argtypes = [vp, i64, pd, pd, pi, pd]
    L.ptamd_frame_begin.argtypes = [vp, C.POINTER(RenderParams)]
    L.ptamd_trace_pixels.argtypes = [vp, vp, i64, i32, i32, vp, vp]
    L.ptamd_adaptive_filter.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp]
    L.ptamd_frame_resolve.argtypes = [vp, vp, vp, vp, vp]
    pl = C.POINTER(PixelList)
    L.ptamd_trace_list.argtypes = [vp, pl, i32, i32, vp, vp]
    L.ptamd_filter_list.argtypes = [vp, pl, i32, vp, vp, vp, vp]
    L.ptamd_filter_list_static.argtypes = [vp, pl, i32, vp, vp, vp, vp]
    L.ptamd_render_list.argtypes = [vp, C.POINTER(RenderParams), vp, i64, vp, vp]
    L.ptamd_pack_rows.argtypes = [vp, pl, vp, vp, vp]
    L.ptamd_unpack_rows.argtypes = [vp, pl, i32, i32, vp, i64, vp, vp]
    L.ptamd_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.ptamd_tonemap_rgba8.argtypes = [pd, i32, i32, C.POINTER(C.c_uint32)]
    L.ptamd_write_png.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), i32, i32]
    L.ptamd_write_png_top_down.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), i32, i32]
    L.ptamd_sampling_rate_rgba8.argtypes = [pi, i32, i32, i32, C.POINTER(C.c_uint32)]
    _LIB = L
    return L


EXPORTS_MULTI = (
    "ptamd_multi_create", "ptamd_multi_destroy", "ptamd_multi_size", "ptamd_multi_upload_scene",
    "ptamd_multi_set_camera", "ptamd_multi_set_profiling", "ptamd_multi_render", "ptamd_multi_get_stats",
    "ptamd_multi_shard_size", "ptamd_multi_rows_per_rank",
)
_LIB_MULTI = None


def _declare_multi(M):
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    M.ptamd_multi_create.argtypes = [C.POINTER(i32), i32, C.POINTER(vp)]
    M.ptamd_multi_destroy.argtypes = [vp]
    M.ptamd_multi_destroy.restype = None
    M.ptamd_multi_size.argtypes = [vp]
    M.ptamd_multi_upload_scene.argtypes = [vp, C.POINTER(SceneDesc)]
    M.ptamd_multi_set_camera.argtypes = [vp, C.POINTER(Camera)]
    M.ptamd_multi_set_profiling.argtypes = [vp, i32, i32]
    M.ptamd_multi_render.argtypes = [vp, C.POINTER(RenderParams), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    M.ptamd_multi_get_stats.argtypes = [vp, i32, C.POINTER(Stats)]
    M.ptamd_multi_shard_size.argtypes = [i64, i32, i32]
    M.ptamd_multi_shard_size.restype = i64
    M.ptamd_multi_rows_per_rank.argtypes = [i64, i32]
    M.ptamd_multi_rows_per_rank.restype = i64
    return M


def lib_multi():
    """libptamd_multi.so: several GPUs of one node from C++ (host threads + RCCL). Loaded on demand."""
    global _LIB_MULTI
    if _LIB_MULTI is not None:
        return _LIB_MULTI
    try:
        # torch ships its own copies of the ROCm libraries (RCCL among them); if it is going to live in
        # this process they must be the ones loaded first, or the two copies' exit handlers collide
        import torch  # noqa: F401
    except ImportError:
        pass
    lib()  # libptamd.so first (libptamd_multi.so links it)
    path = os.path.join(os.path.dirname(lib_path()), "libptamd_multi.so")
    if not os.path.exists(path):
        raise PtamdError("%s is missing: build it with `make -C pathtracer_amd/csrc`" % path)
    M = _declare_multi(C.CDLL(path, mode=C.RTLD_GLOBAL))
    _LIB_MULTI = M
    return M


def _check(rc):
    if rc != 0:
        raise PtamdError("ptamd error %d: %s" % (rc, lib().ptamd_last_error().decode(errors="replace")))


class Scene:
    """Host scene: COLLADA file -> static scene -> BVH -> flat arrays."""

    def __init__(self, dae_path):
        self._h = C.c_void_p()
        _check(lib().ptamd_scene_load(os.fsencode(dae_path), C.byref(self._h)))
        self.desc = SceneDesc()
        _check(lib().ptamd_scene_get_desc(self._h, C.byref(self.desc)))

    def camera(self, w=0, h=0):
        cam = Camera()
        _check(lib().ptamd_scene_get_camera(self._h, w, h, C.byref(cam)))
        return cam

    def load_camera_file(self, path):
        _check(lib().ptamd_scene_load_camera_file(self._h, os.fsencode(path)))

    def dump_camera_file(self, path):
        _check(lib().ptamd_scene_dump_camera_file(self._h, os.fsencode(path)))

    def arrays(self):
        """Deep numpy copy of the flattened scene (cabi.SceneArrays)."""
        return cabi.SceneArrays.from_desc(self.desc)

    @property
    def build_seconds(self):
        return lib().ptamd_scene_build_seconds(self._h)

    def close(self):
        if self._h:
            lib().ptamd_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiDevice:
    """Several GPUs of this node rendering a frame together (ptamd_multi_*, C++ threads + RCCL)."""

    def __init__(self, device_ids, library=None):
        """library: another build of libptamd_multi.so, already loaded and declared (the tests' build with an
        in-process stand-in for RCCL); default: libptamd_multi.so."""
        self._m = library or lib_multi()
        ids = (C.c_int32 * len(device_ids))(*device_ids)
        self._h = C.c_void_p()
        _check(self._m.ptamd_multi_create(ids, len(device_ids), C.byref(self._h)))
        self._keep = None

    def set_profiling(self, events=False, counters=False):
        _check(self._m.ptamd_multi_set_profiling(self._h, int(events), int(counters)))

    def upload(self, scene):
        desc = scene.desc if hasattr(scene, "desc") else scene
        self._keep = scene
        _check(self._m.ptamd_multi_upload_scene(self._h, C.byref(desc)))

    def set_camera(self, cam):
        _check(self._m.ptamd_multi_set_camera(self._h, C.byref(cam)))

    def render(self, params):
        hdr = np.zeros((params.h, params.w, 3), np.float64)
        cnt = np.zeros((params.h, params.w), np.int32)
        _check(self._m.ptamd_multi_render(self._h, C.byref(params), hdr.ctypes.data_as(C.POINTER(C.c_double)),
                                              cnt.ctypes.data_as(C.POINTER(C.c_int32))))
        return hdr, cnt

    def stats(self, rank=0):
        st = Stats()
        _check(self._m.ptamd_multi_get_stats(self._h, rank, C.byref(st)))
        return st

    def close(self):
        if self._h:
            self._m.ptamd_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Device:
    """One GPU's render context."""

    def __init__(self, device_id=0):
        self._h = C.c_void_p()
        _check(lib().ptamd_create(device_id, C.byref(self._h)))
        self._keep = None

    def upload(self, scene):
        """scene: Scene, cabi.SceneArrays or SceneDesc."""
        desc = scene.desc if hasattr(scene, "desc") else scene
        self._keep = scene
        _check(lib().ptamd_upload_scene(self._h, C.byref(desc)))

    def set_camera(self, cam):
        _check(lib().ptamd_set_camera(self._h, C.byref(cam)))

    def set_max_paths(self, n):
        _check(lib().ptamd_set_max_paths(self._h, n))

    def set_profiling(self, events=False, counters=False):
        _check(lib().ptamd_set_profiling(self._h, int(events), int(counters)))

    def set_event_classes(self, mask):
        """Bit k of mask: launches of kernel class k are timed with HIP events."""
        _check(lib().ptamd_set_event_classes(self._h, mask))

    def render(self, params):
        """Whole frame; returns (hdr[h,w,3] float64, count[h,w] int32), row 0 = bottom."""
        hdr = np.zeros(params.w * params.h * 3, np.float64)
        cnt = np.zeros(params.w * params.h, np.int32)
        _check(lib().ptamd_render(self._h, C.byref(params), hdr.ctypes.data_as(C.POINTER(C.c_double)),
                                  cnt.ctypes.data_as(C.POINTER(C.c_int32))))
        return hdr.reshape(params.h, params.w, 3), cnt.reshape(params.h, params.w)

    def trace_primary(self, sx, sy):
        sx = np.ascontiguousarray(sx, np.float64).ravel()
        sy = np.ascontiguousarray(sy, np.float64).ravel()
        if sx.size != sy.size:
            raise PtamdError("sx and sy differ in length")
        ids = np.zeros(sx.size, np.int32)
        ts = np.zeros(sx.size, np.float64)
        pd = C.POINTER(C.c_double)
        _check(lib().ptamd_trace_primary(self._h, sx.size, sx.ctypes.data_as(pd), sy.ctypes.data_as(pd),
                                         ids.ctypes.data_as(C.POINTER(C.c_int32)), ts.ctypes.data_as(pd)))
        return ids, ts

    # ---- round-level API: arguments are device pointers (ints) -------------
    def frame_begin(self, params):
        _check(lib().ptamd_frame_begin(self._h, C.byref(params)))

    def trace_pixels(self, d_pixels, n, sample_base, n_samples, d_state, stream=0):
        _check(lib().ptamd_trace_pixels(self._h, d_pixels, n, sample_base, n_samples, d_state, stream))

    def adaptive_filter(self, d_in, n, n_done, d_state, d_out, d_n_out, stream=0):
        _check(lib().ptamd_adaptive_filter(self._h, d_in, n, n_done, d_state, d_out, d_n_out, stream))

    def frame_resolve(self, d_state, d_hdr, d_count, stream=0):
        _check(lib().ptamd_frame_resolve(self._h, d_state, d_hdr, d_count, stream))

    # the same on lists whose length lives on the device (cabi.PixelList)
    def trace_list(self, plist, sample_base, n_samples, d_state, stream=0):
        _check(lib().ptamd_trace_list(self._h, C.byref(plist), sample_base, n_samples, d_state, stream))

    def filter_list(self, plist, n_done, d_state, d_out, d_n_out, stream=0):
        _check(lib().ptamd_filter_list(self._h, C.byref(plist), n_done, d_state, d_out, d_n_out, stream))

    def filter_list_static(self, plist, n_done, d_state, d_out, d_n_out, stream=0):
        """filter_list without any device allocation (static scratch; lists up to FILTER_STATIC_MAX_PIXELS)."""
        _check(lib().ptamd_filter_list_static(self._h, C.byref(plist), n_done, d_state, d_out, d_n_out, stream))

    def pack_rows(self, plist, d_state, d_rows, stream=0):
        _check(lib().ptamd_pack_rows(self._h, C.byref(plist), d_state, d_rows, stream))

    def unpack_rows(self, plist, world, skip_rank, d_rows, rows_per_rank, d_state, stream=0):
        _check(lib().ptamd_unpack_rows(self._h, C.byref(plist), world, skip_rank, d_rows, rows_per_rank, d_state, stream))

    def stats(self):
        st = Stats()
        _check(lib().ptamd_get_stats(self._h, C.byref(st)))
        return st

    def close(self):
        if self._h:
            lib().ptamd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tonemap(hdr):
    """hdr[h,w,3] -> uint32[h,w] RGBA8 (R low byte), reference tone curve."""
    h, w, _ = hdr.shape
    a = np.ascontiguousarray(hdr, np.float64)
    out = np.zeros(h * w, np.uint32)
    _check(lib().ptamd_tonemap_rgba8(a.ctypes.data_as(C.POINTER(C.c_double)), w, h,
                                     out.ctypes.data_as(C.POINTER(C.c_uint32))))
    return out.reshape(h, w)


def sampling_rate(count, ns_aa):
    """save_sampling_rate_image's colouring of a sample-count map (top row first), as uint32 RGBA."""
    h, w = count.shape
    c = np.ascontiguousarray(count, np.int32)
    out = np.zeros(h * w, np.uint32)
    _check(lib().ptamd_sampling_rate_rgba8(c.ctypes.data_as(C.POINTER(C.c_int32)), w, h, ns_aa,
                                           out.ctypes.data_as(C.POINTER(C.c_uint32))))
    return out.reshape(h, w)


def write_png(path, rgba_bottom_up):
    h, w = rgba_bottom_up.shape
    a = np.ascontiguousarray(rgba_bottom_up, np.uint32)
    _check(lib().ptamd_write_png(os.fsencode(path), a.ctypes.data_as(C.POINTER(C.c_uint32)), w, h))
