"""Specular mode (ptamd_render_params.specular): ideal mirror / refraction / glass.

The reference declares these BSDFs and leaves them empty (advanced_bsdf.cpp:21-27,
99-106, 124-134, 146-169), so there is nothing to pin them against: PARITY UNPINNED.
What is checked instead: the CPU restatement against the physics it claims (law of
reflection, Snell's law, the critical angle, Schlick's Fresnel term, the weights
f |cos| / pdf), that the default mode still renders the reference's black stubs, and
(GPU) the HIP path against the CPU restatement with the same random numbers."""
import numpy as np
import pytest

import pathtracer_amd as pt
from pathtracer_amd import cabi
from conftest import scene_path
import oracle_lib as O

MIRROR, REFRACTION, GLASS = 2, 4, 5


def _dirs(n, seed):
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def test_mirror_reflects_about_the_normal_and_keeps_the_reflectance():
    n = 200
    wo = _dirs(n, 1)
    refl = np.tile([0.9, 0.5, 0.25], (n, 1))
    wi, f, pdf = O.kat_specular(np.full(n, MIRROR), refl, np.zeros((n, 4)), wo, np.zeros(n))
    assert np.array_equal(wi, wo * [-1, -1, 1])
    assert np.all(pdf == 1)
    # the estimator's weight f |cos| / pdf is the reflectance
    assert np.allclose(f * np.abs(wi[:, 2:3]), refl, rtol=1e-15)


def test_refraction_obeys_snell_and_the_critical_angle():
    n = 400
    ior = 1.45
    wo = _dirs(n, 2)
    tr = np.tile([0.7, 0.8, 0.9], (n, 1))
    extra = np.tile([0, 0, 0, ior], (n, 1))
    wi, f, pdf = O.kat_specular(np.full(n, REFRACTION), tr, extra, wo, np.zeros(n))
    sin_o = np.sqrt(1 - wo[:, 2] ** 2)
    inside = wo[:, 2] < 0                       # leaving the medium
    tir = inside & (sin_o * ior > 1)
    assert tir.sum() > 10 and (~tir).sum() > 100
    assert not wi[tir].any() and not f[tir].any()   # path ends
    ok = ~tir
    assert np.allclose(np.linalg.norm(wi[ok], axis=1), 1, atol=1e-14)
    sin_i = np.sqrt(wi[ok, 0] ** 2 + wi[ok, 1] ** 2)
    eta = np.where(inside[ok], ior, 1 / ior)
    assert np.allclose(sin_i, eta * sin_o[ok], atol=1e-14)           # Snell
    assert np.all(np.sign(wi[ok, 2]) == -np.sign(wo[ok, 2]))          # crosses the surface
    # in the plane of incidence, on the far side of the normal
    assert np.allclose(wi[ok, 0] * wo[ok, 1], wi[ok, 1] * wo[ok, 0], atol=1e-14)
    assert np.allclose(f[ok] * np.abs(wi[ok, 2:3]) * eta[:, None] ** 2, tr[ok], rtol=1e-14)
    assert np.all(pdf == 1)


def test_glass_splits_by_schlick_and_conserves_the_weights():
    ior = 1.5
    r0 = ((1 - ior) / (1 + ior)) ** 2
    # normal incidence: R = R0 exactly
    wo = np.array([[0.0, 0.0, 1.0], [0.0, 0.0, 1.0]])
    refl, tr = [1.0, 1.0, 1.0], [1.0, 1.0, 1.0]
    extra = np.tile(refl + [ior], (2, 1))
    wi, f, pdf = O.kat_specular([GLASS, GLASS], np.tile(tr, (2, 1)), extra, wo, [r0 * 0.999, r0 * 1.001])
    assert np.array_equal(wi[0], [0, 0, 1]) and np.isclose(pdf[0], r0, rtol=1e-15)       # reflected
    assert np.allclose(wi[1], [0, 0, -1]) and np.isclose(pdf[1], 1 - r0, rtol=1e-15)     # transmitted
    # any direction: weight = reflectance on the reflected branch, transmittance / eta^2 on the other,
    # branch probabilities add up to one, R rises to 1 at grazing incidence
    n = 300
    wo = _dirs(n, 3)
    wo[:, 2] = np.abs(wo[:, 2])      # from outside: no total internal reflection
    refl = np.tile([0.8, 0.9, 1.0], (n, 1))
    tr = np.tile([0.6, 0.7, 0.5], (n, 1))
    extra = np.hstack([refl, np.full((n, 1), ior)])
    wi_r, f_r, p_r = O.kat_specular(np.full(n, GLASS), tr, extra, wo, np.zeros(n))         # u = 0: reflect
    wi_t, f_t, p_t = O.kat_specular(np.full(n, GLASS), tr, extra, wo, np.full(n, 0.99999999))
    R = r0 + (1 - r0) * (1 - wo[:, 2]) ** 5
    took_reflection = p_t == p_r     # u = 0.99999999 < R only at extreme grazing angles
    assert took_reflection.sum() <= 1
    assert np.allclose(p_r, R, rtol=1e-14) and np.allclose((p_r + p_t)[~took_reflection], 1, rtol=1e-15)
    assert np.allclose(f_r * np.abs(wi_r[:, 2:3]) / p_r[:, None], refl, rtol=1e-14)
    m = ~took_reflection
    assert np.allclose(f_t[m] * np.abs(wi_t[m, 2:3]) / p_t[m, None], tr[m] * ior ** 2, rtol=1e-13)
    # from inside beyond the critical angle glass reflects everything
    wo_in = np.array([[0.9, 0.0, -np.sqrt(1 - 0.81)]])
    wi, f, pdf = O.kat_specular([GLASS], [tr[0]], [extra[0]], wo_in, [0.5])
    assert np.array_equal(wi[0], wo_in[0] * [-1, -1, 1]) and pdf[0] == 1
    assert np.allclose(f[0] * abs(wi[0, 2]), refl[0])


def _cbspheres(w, h):
    sc = pt.Scene(scene_path("CBspheres"))
    return sc, sc.arrays(), sc.camera(w, h)


def _specular_window(arr, cam, w, h):
    """(mask of pixels whose camera ray hits the mirror or the glass sphere, a rectangle around them).
    The field of view shrinks with the frame (the reference sizes it for 800x600), so the
    spheres are only in view at a few hundred pixels across; a rectangle keeps the test cheap."""
    xs, ys = np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h)
    ids, _ = O.trace_primary(arr, cam, xs, ys)
    ids = ids.reshape(h, w)
    kinds = arr.a["mat_kind"][arr.a["prim_material"]]
    mask = (ids >= 0) & np.isin(kinds[np.maximum(ids, 0)], (MIRROR, GLASS))
    yy, xx = np.nonzero(mask)
    # samples are jittered over the whole pixel: keep pixels whose neighbours are on a sphere too
    core = mask.copy()
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            core &= np.roll(np.roll(mask, dy, axis=0), dx, axis=1)
    mask = core
    rect = (max(int(xx.min()) - 4, 0), max(int(yy.min()) - 4, 0), min(int(xx.max()) + 5, w), min(int(yy.max()) + 5, h))
    return mask, ids, rect


def test_oracle_specular_mode_lights_up_the_spheres_and_leaves_default_mode_alone():
    """CBspheres holds a mirror and a glass sphere. Default mode: black (the reference's
    stubs). Specular mode: they show the room; the walls around them change only by the
    light the spheres now pass on."""
    w, h = 320, 240
    sc, arr, cam = _cbspheres(w, h)
    mask, ids, rect = _specular_window(arr, cam, w, h)
    assert mask.sum() > 1000
    assert arr.a["mat_extra"].reshape(-1, 4)[:, 3].max() > 1.0, "the loader must deliver the index of refraction"
    kw = dict(ns_aa=32, max_ray_depth=6, ns_area_light=2, max_tolerance=0.0, seed=3, rect=rect)
    black, _, _ = O.render(arr, cam, cabi.make_params(w, h, **kw), threads=8)
    lit, _, _ = O.render(arr, cam, cabi.make_params(w, h, specular=True, **kw), threads=8)
    assert black[mask].max() == 0.0
    assert (lit[mask].sum(axis=1) > 0).mean() > 0.9
    inside = np.zeros((h, w), bool)
    inside[rect[1]:rect[3], rect[0]:rect[2]] = True
    off = inside & (ids >= 0) & ~mask
    assert off.sum() > 500
    assert 0.9 < lit[off].mean() / black[off].mean() < 1.8


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(ns_aa=32, max_ray_depth=8, ns_area_light=2),
                                dict(ns_aa=32, max_ray_depth=5, ns_area_light=1, accumulate_bounces=False),
                                dict(ns_aa=16, max_ray_depth=16, ns_area_light=4, max_tolerance=0.1, samples_per_batch=8)])
def test_gpu_specular_mode_matches_the_cpu_restatement(device, kw):
    """Same random numbers on both sides, so the frames agree to rounding (tolerances as in
    test_gpu_parity.test_render_matches_oracle)."""
    w, h = 320, 240
    sc, arr, cam = _cbspheres(w, h)
    mask, ids, rect = _specular_window(arr, cam, w, h)
    device.upload(sc)
    device.set_camera(cam)
    kw = dict(kw, rect=rect)
    p = cabi.make_params(w, h, seed=11, specular=True, **kw)
    hdr, cnt = device.render(p)
    rh, rc, _ = O.render(arr, cam, p, threads=8)
    assert (cnt == rc).mean() >= 0.999
    close = np.isclose(hdr, rh, rtol=1e-9, atol=1e-12).all(axis=2)
    assert close.mean() >= 0.999, "%.3f%% of pixels differ beyond rounding" % (100 * (1 - close.mean()))
    # and the mode does something on the GPU too
    p0 = cabi.make_params(w, h, seed=11, **kw)
    hdr0, _ = device.render(p0)
    assert hdr0[mask].max() == 0.0 and (hdr[mask].sum(axis=1) > 0).mean() > 0.9
