"""ctypes mirror of include/ptamd.h (struct layouts and kind constants)."""
import ctypes as C

import numpy as np

PRIM_TRIANGLE, PRIM_SPHERE = 0, 1
MAT_DIFFUSE, MAT_EMISSION, MAT_MIRROR, MAT_MICROFACET, MAT_REFRACTION, MAT_GLASS = range(6)
LIGHT_DIRECTIONAL, LIGHT_INF_HEMISPHERE, LIGHT_POINT, LIGHT_SPOT, LIGHT_AREA = range(5)
K_NAMES = ("generate", "extend", "shade", "shadow", "resolve", "accumulate", "filter", "other", "trace2",
           "light", "class10", "class11")
K_RESOLVE = 4
K_TRACE2 = 8
K_LIGHT = 9

_pi32 = C.POINTER(C.c_int32)
_pf64 = C.POINTER(C.c_double)


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_prims", C.c_int32), ("n_nodes", C.c_int32), ("n_materials", C.c_int32),
        ("n_lights", C.c_int32), ("root", C.c_int32), ("jitter_n", C.c_int32),
        ("prim_type", _pi32), ("prim_pos", _pf64), ("prim_nrm", _pf64), ("prim_material", _pi32),
        ("node_bbox", _pf64), ("node_left", _pi32), ("node_right", _pi32),
        ("node_start", _pi32), ("node_count", _pi32),
        ("mat_kind", _pi32), ("mat_param", _pf64),
        ("light_kind", _pi32), ("light_param", _pf64),
        ("jitter_table", _pf64), ("mat_extra", _pf64),
    ]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("c2w", C.c_double * 9), ("hFov", C.c_double),
                ("vFov", C.c_double), ("nClip", C.c_double), ("fClip", C.c_double)]


class RenderParams(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("x0", C.c_int32), ("y0", C.c_int32),
                ("x1", C.c_int32), ("y1", C.c_int32), ("ns_aa", C.c_int32),
                ("max_ray_depth", C.c_int32), ("accumulate_bounces", C.c_int32),
                ("ns_area_light", C.c_int32), ("samples_per_batch", C.c_int32),
                ("hemisphere", C.c_int32), ("max_tolerance", C.c_double), ("seed", C.c_uint64),
                ("specular", C.c_int32), ("reserved", C.c_int32)]


class PixelList(C.Structure):
    """ptamd_pixel_list: a shard (offset, stride) of a device-resident pixel list whose
    length is read on the device from d_n (or is n when d_n is 0); n_max bounds it."""
    _fields_ = [("d_pixels", C.c_void_p), ("d_n", C.c_void_p), ("n", C.c_int64), ("n_max", C.c_int64),
                ("offset", C.c_int32), ("stride", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("samples", C.c_uint64), ("rays", C.c_uint64),
                ("shadow_rays", C.c_uint64), ("path_vertices", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("prim_tests", C.c_uint64),
                ("launches", C.c_uint64 * 12), ("kernel_ms", C.c_double * 12),
                ("two_ext_rays", C.c_uint64), ("two_shadow_rays", C.c_uint64),
                ("exact_tests", C.c_uint64), ("unused_slots", C.c_uint64)]


def make_params(w, h, ns_aa=1, max_ray_depth=1, accumulate_bounces=True, ns_area_light=1,
                samples_per_batch=32, max_tolerance=0.05, hemisphere=False, seed=0, rect=None, specular=False):
    """Render parameters with the reference's defaults (application.h:45-66).

    max_tolerance goes through float, as the reference's AppConfig field does.
    """
    x0, y0, x1, y1 = rect if rect is not None else (0, 0, w, h)
    return RenderParams(w, h, x0, y0, x1, y1, ns_aa, max_ray_depth, int(bool(accumulate_bounces)),
                        ns_area_light, samples_per_batch, int(bool(hemisphere)),
                        float(np.float32(max_tolerance)), seed, int(bool(specular)), 0)


class SceneArrays:
    """Owns numpy arrays of a flattened scene and exposes them as a SceneDesc."""

    FIELDS = (("prim_type", np.int32), ("prim_pos", np.float64), ("prim_nrm", np.float64),
              ("prim_material", np.int32), ("node_bbox", np.float64), ("node_left", np.int32),
              ("node_right", np.int32), ("node_start", np.int32), ("node_count", np.int32),
              ("mat_kind", np.int32), ("mat_param", np.float64), ("light_kind", np.int32),
              ("light_param", np.float64), ("jitter_table", np.float64), ("mat_extra", np.float64))

    def __init__(self, arrays, root):
        arrays = dict(arrays)
        if "mat_extra" not in arrays:  # dumps of the reference's own scene graph do not carry it
            arrays["mat_extra"] = np.zeros(4 * np.asarray(arrays["mat_kind"]).size)
        self.a = {k: np.ascontiguousarray(arrays[k], dtype=dt).ravel() for k, dt in self.FIELDS}
        self.root = int(root)
        d = SceneDesc()
        d.n_prims = self.a["prim_type"].size
        d.n_nodes = self.a["node_left"].size
        d.n_materials = self.a["mat_kind"].size
        d.n_lights = self.a["light_kind"].size
        d.root = self.root
        d.jitter_n = int(round((self.a["jitter_table"].size // 2) ** 0.5))
        for k, dt in self.FIELDS:
            ptr_t = _pi32 if dt == np.int32 else _pf64
            setattr(d, k, self.a[k].ctypes.data_as(ptr_t))
        self.desc = d

    @classmethod
    def from_ptar(cls, arrays):
        """From a scene dump (reference driver or ptamd_dump layout)."""
        a = dict(arrays)
        if "prim_material" not in a:  # ptamd_dump stores per-primitive materials
            a["prim_material"] = np.arange(a["prim_type"].size, dtype=np.int32)
            a["mat_kind"] = a["prim_mat_kind"]
            a["mat_param"] = a["prim_mat_param"]
        return cls(a, a["root"][0])

    @classmethod
    def from_desc(cls, d):
        """Deep copy of a SceneDesc returned by the library."""
        def arr(p, n, dt):
            return np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
        a = dict(prim_type=arr(d.prim_type, d.n_prims, np.int32), prim_pos=arr(d.prim_pos, d.n_prims * 9, np.float64),
                 prim_nrm=arr(d.prim_nrm, d.n_prims * 9, np.float64), prim_material=arr(d.prim_material, d.n_prims, np.int32),
                 node_bbox=arr(d.node_bbox, d.n_nodes * 6, np.float64), node_left=arr(d.node_left, d.n_nodes, np.int32),
                 node_right=arr(d.node_right, d.n_nodes, np.int32), node_start=arr(d.node_start, d.n_nodes, np.int32),
                 node_count=arr(d.node_count, d.n_nodes, np.int32), mat_kind=arr(d.mat_kind, d.n_materials, np.int32),
                 mat_param=arr(d.mat_param, d.n_materials * 3, np.float64), light_kind=arr(d.light_kind, d.n_lights, np.int32),
                 light_param=arr(d.light_param, d.n_lights * 16, np.float64),
                 jitter_table=arr(d.jitter_table, d.jitter_n * d.jitter_n * 2, np.float64))
        if d.mat_extra:
            a["mat_extra"] = arr(d.mat_extra, d.n_materials * 4, np.float64)
        return cls(a, d.root)


def camera_from_dump(cam26):
    """Camera struct from the 26-double camera record of a scene dump."""
    c = Camera()
    for i in range(3):
        c.pos[i] = cam26[i]
    for i in range(9):
        c.c2w[i] = cam26[3 + i]
    c.hFov, c.vFov, c.nClip, c.fClip = (float(v) for v in cam26[12:16])
    return c
