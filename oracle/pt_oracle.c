/* TEST INFRASTRUCTURE — CPU restatement of the reference's render hot path.
 * See pt_oracle.h for scope, the random-number layout and the parity status.
 * Compile with -ffp-contract=off (no FMA): the reference is built without -mfma.
 * All `path:line` citations are relative to the reference tree. */
#define _POSIX_C_SOURCE 200809L
#include "pt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI_D (3.14159265358979323)  /* CGL/include/CGL/misc.h:11 */
#define EPS_D (0.00000000001)       /* CGL/include/CGL/misc.h:12 */
#define RRT 0.7                     /* src/pathtracer/pathtracer.cpp:158 */

/* ---- vectors: operation order of CGL/include/CGL/vector3D.h:96-262 -------- */
typedef struct { double x, y, z; } v3;
static inline v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, double c) { return V(a.x * c, a.y * c, a.z * c); }
static inline v3 vdivs(v3 a, double c) { double rc = 1.0 / c; return V(rc * a.x, rc * a.y, rc * a.z); }
static inline double vdot(v3 u, v3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
static inline v3 vcross(v3 u, v3 v) { return V(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x); }
static inline double vnorm2(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static inline double vnorm(v3 a) { return sqrt(vnorm2(a)); }
static inline v3 vunit(v3 a) { double r = 1. / vnorm(a); return vscale(a, r); }
typedef struct { v3 c[3]; } m3; /* column-major, CGL/src/matrix3x3.cpp:110-114 */
static inline v3 mmul(const m3 *m, v3 v) {
  return vadd(vadd(vscale(m->c[0], v.x), vscale(m->c[1], v.y)), vscale(m->c[2], v.z));
}
static inline m3 mtrans(const m3 *a) {
  m3 b;
  b.c[0] = V(a->c[0].x, a->c[1].x, a->c[2].x);
  b.c[1] = V(a->c[0].y, a->c[1].y, a->c[2].y);
  b.c[2] = V(a->c[0].z, a->c[1].z, a->c[2].z);
  return b;
}
static inline double dmin(double a, double b) { return (b < a) ? b : a; } /* std::min */
static inline double dmax(double a, double b) { return (a < b) ? b : a; } /* std::max */
static inline v3 ld3(const double *p) { return V(p[0], p[1], p[2]); }

/* ---- Philox-4x32-10 ------------------------------------------------------- */
void oracle_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* src/util/random_util.h:15-22: word / (2^32-1), clamped */
static inline double u01(uint32_t w) {
  const double rmax = 1.0 / 4294967295.0;
  double v = (double)w * rmax;
  return dmin(dmax(v, 0.0000001), 0.99999999);
}

typedef struct {
  uint32_t k0, k1, sample, seed_hi;
} rng_t;
static inline void rng_block(const rng_t *g, uint32_t block, double u[4]) {
  uint32_t w[4];
  oracle_philox(g->sample, block, g->seed_hi, 0u, g->k0, g->k1, w);
  for (int i = 0; i < 4; i++) u[i] = u01(w[i]);
}

/* ---- scene access ----------------------------------------------------------- */
typedef struct {
  v3 o, d, inv_d;
  double min_t, max_t; /* max_t is mutated by primitive hits, src/pathtracer/ray.h:26-27 */
  int depth;
  int via_delta; /* specular mode: the ray left a Dirac-delta BSDF */
} ray_t;
static inline ray_t make_ray(v3 o, v3 d) { /* src/pathtracer/ray.h:39-42 */
  ray_t r;
  r.o = o; r.d = d;
  r.inv_d = V(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
  r.min_t = 0.0; r.max_t = INFINITY; r.depth = 0; r.via_delta = 0;
  return r;
}
typedef struct {
  double t;
  int prim;
  v3 n;
  int material;
} isect_t;

typedef struct {
  const ptamd_scene_desc *s;
  const ptamd_camera *cam;
  const ptamd_render_params *p;
  int nls; /* light samples per vertex */
  int bpv; /* RNG blocks per vertex */
  uint64_t rays, shadow_rays, vertices, nodes, tests, shadow_nodes, shadow_tests;
  int in_shadow;
} ctx_t;

/* src/scene/bbox.cpp:10-47 */
static int bbox_intersect(const double *bb, const ray_t *r, double *t0, double *t1) {
  double tmin, tmax, timinx, timaxx, timiny, timaxy, timinz, timaxz;
  double r_rdx = r->inv_d.x, r_rdy = r->inv_d.y, r_rdz = r->inv_d.z;
  double minox = bb[0] - r->o.x, maxox = bb[3] - r->o.x;
  double minoy = bb[1] - r->o.y, maxoy = bb[4] - r->o.y;
  double minoz = bb[2] - r->o.z, maxoz = bb[5] - r->o.z;
  timinx = minox * r_rdx;
  timaxx = maxox * r_rdx;
  tmin = dmin(timinx, timaxx);
  tmax = dmax(timinx, timaxx);
  timiny = minoy * r_rdy;
  timaxy = maxoy * r_rdy;
  tmin = dmax(tmin, dmin(timiny, timaxy));
  tmax = dmin(tmax, dmax(timiny, timaxy));
  timinz = minoz * r_rdz;
  timaxz = maxoz * r_rdz;
  tmin = dmax(tmin, dmin(timinz, timaxz));
  tmax = dmin(tmax, dmax(timinz, timaxz));
  if (tmin > tmax || tmin > r->max_t || tmax < r->min_t || tmax < 0) return 0;
  *t0 = tmin;
  *t1 = tmax;
  return 1;
}

/* src/scene/triangle.cpp:25-70 */
static int triangle_intersect(const double *P, const double *N, ray_t *r, double *t_out, v3 *n_out) {
  v3 p1 = ld3(P), p2 = ld3(P + 3), p3 = ld3(P + 6);
  v3 e1 = vsub(p2, p1), e2 = vsub(p3, p1);
  v3 normal = vcross(e1, e2);
  if (vdot(normal, r->d) == 0) return 0;
  v3 s = vsub(r->o, p1), s1 = vcross(r->d, e2), s2 = vcross(s, e1);
  double rse1 = 1 / (vdot(s1, e1));
  double t = vdot(s2, e2) * rse1, u = vdot(s1, s) * rse1, v = vdot(s2, r->d) * rse1;
  if (t < r->min_t || t > r->max_t || u < 0 || v < 0 || u + v > 1) return 0;
  r->max_t = t;
  v3 n1 = ld3(N), n2 = ld3(N + 3), n3 = ld3(N + 6);
  /* (1 - u - v) * n1 + u * n2 + v * n3 */
  *n_out = vadd(vadd(vscale(n1, 1 - u - v), vscale(n2, u)), vscale(n3, v));
  *t_out = t;
  return 1;
}

/* src/scene/sphere.cpp:11-72, normal: src/scene/sphere.h:73-75 */
static int sphere_intersect(const double *P, ray_t *r, double *t_out, v3 *n_out) {
  v3 o = ld3(P);
  double r2 = P[4];
  v3 ominc = vsub(r->o, o);
  double a = vdot(r->d, r->d), b = 2 * vdot(ominc, r->d), c = vdot(ominc, ominc) - r2;
  double discr = b * b - 4 * a * c;
  if (discr < 0) return 0;
  double r2a = 1 / (2 * a);
  double b2a = -b * r2a, sqrt_discr_r2a = sqrt(discr) * r2a;
  double t1 = b2a - sqrt_discr_r2a;
  double t2 = b2a + sqrt_discr_r2a;
  double t;
  if (t1 <= r->max_t && t1 >= r->min_t) t = t1;
  else if (t2 <= r->max_t && t2 >= r->min_t) t = t2;
  else return 0;
  v3 p = vadd(r->o, vscale(r->d, t)); /* r.o + i->t * r.d */
  *n_out = vunit(vsub(p, o));
  *t_out = t;
  r->max_t = t;
  return 1;
}

/* src/scene/bvh.cpp:164-200 (wrapper src/scene/bvh.h:95-98). The explicit stack
 * pushes l then r, so the right child is visited first; a later hit replaces
 * the recorded one only when strictly closer. */
static int bvh_intersect(ctx_t *c, ray_t *ray, isect_t *i) {
  const ptamd_scene_desc *s = c->s;
  int hit = 0;
  double t0, t1;
  c->rays++;
  i->t = ray->max_t;
  int stack[128];
  int sp = 0;
  stack[sp++] = s->root;
  while (sp > 0) {
    int n = stack[--sp];
    c->nodes++;
    if (c->in_shadow) c->shadow_nodes++;
    if (!bbox_intersect(s->node_bbox + 6 * (size_t)n, ray, &t0, &t1)) continue;
    if (s->node_count[n] > 0) {
      int b = s->node_start[n], e = b + s->node_count[n];
      for (int p = b; p < e; p++) {
        c->tests++;
        if (c->in_shadow) c->shadow_tests++;
        double t;
        v3 nn;
        int h = (s->prim_type[p] == PTAMD_PRIM_TRIANGLE)
                    ? triangle_intersect(s->prim_pos + 9 * (size_t)p, s->prim_nrm + 9 * (size_t)p, ray, &t, &nn)
                    : sphere_intersect(s->prim_pos + 9 * (size_t)p, ray, &t, &nn);
        if (h && t < i->t) {
          hit = 1;
          i->t = t;
          i->prim = p;
          i->n = nn;
          i->material = s->prim_material[p];
        }
      }
    } else {
      if (sp + 2 > 128) continue; /* deeper than any tree the builder produces */
      stack[sp++] = s->node_left[n];
      stack[sp++] = s->node_right[n];
    }
  }
  return hit;
}

/* src/pathtracer/camera.cpp:190-203 */
static ray_t generate_ray(const ptamd_camera *cam, double x, double y) {
  v3 sensor = V((x - 0.5) * 2 * tan(cam->hFov * (PI_D / 180) * 0.5), (y - 0.5) * 2 * tan(cam->vFov * (PI_D / 180) * 0.5), -1);
  m3 c2w;
  c2w.c[0] = ld3(cam->c2w);
  c2w.c[1] = ld3(cam->c2w + 3);
  c2w.c[2] = ld3(cam->c2w + 6);
  v3 dir = mmul(&c2w, sensor);
  ray_t r = make_ray(ld3(cam->pos), vunit(dir));
  r.min_t = cam->nClip;
  r.max_t = cam->fClip;
  return r;
}

/* src/pathtracer/bsdf.cpp:21-41 */
static void make_coord_space(m3 *o2w, v3 n) {
  v3 z = n;
  v3 h = z;
  if (fabs(h.x) <= fabs(h.y) && fabs(h.x) <= fabs(h.z)) h.x = 1.0;
  else if (fabs(h.y) <= fabs(h.x) && fabs(h.y) <= fabs(h.z)) h.y = 1.0;
  else h.z = 1.0;
  z = vscale(z, 1. / vnorm(z)); /* normalize(): *= 1/norm */
  v3 y = vcross(h, z);
  y = vscale(y, 1. / vnorm(y));
  v3 x = vcross(z, y);
  x = vscale(x, 1. / vnorm(x));
  o2w->c[0] = x;
  o2w->c[1] = y;
  o2w->c[2] = z;
}

/* src/pathtracer/sampler.cpp:64-73 */
static v3 cosine_sample(double Xi1, double Xi2, double *pdf) {
  double r = sqrt(Xi1);
  double theta = 2. * PI_D * Xi2;
  *pdf = sqrt(1 - Xi1) / PI_D;
  return V(r * cos(theta), r * sin(theta), sqrt(1 - Xi1));
}

/* src/pathtracer/sampler.cpp:35-49 (float trig, as in the reference) */
static v3 uniform_hemisphere_sample(double Xi1, double Xi2) {
  double theta = acos(Xi1);
  double phi = 2.0 * PI_D * Xi2;
  double xs = sinf((float)theta) * cosf((float)phi);
  double ys = sinf((float)theta) * sinf((float)phi);
  double zs = cosf((float)theta);
  return V(xs, ys, zs);
}

/* BSDF::f for every kind: only DiffuseBSDF is non-zero
 * (src/pathtracer/bsdf.cpp:52-58, 87-89; advanced_bsdf.cpp:17-19, 62-65, 95-97, 120-122) */
static v3 bsdf_f(const ctx_t *c, int material) {
  if (c->s->mat_kind[material] == PTAMD_MAT_DIFFUSE) return vdivs(ld3(c->s->mat_param + 3 * (size_t)material), PI_D);
  return V(0, 0, 0);
}
static v3 bsdf_emission(const ctx_t *c, int material) { /* bsdf.h:133,278 */
  if (c->s->mat_kind[material] == PTAMD_MAT_EMISSION) return ld3(c->s->mat_param + 3 * (size_t)material);
  return V(0, 0, 0);
}
/* sample_f: Diffuse (bsdf.cpp:63-73), Emission (:94-98), Microfacet
 * (advanced_bsdf.cpp:67-80) sample a cosine direction; Mirror / Refraction /
 * Glass are stubs that leave wi = (0,0,0) (advanced_bsdf.cpp:21-27,99-106,124-134). */
static v3 bsdf_sample_f(const ctx_t *c, int material, double Xi1, double Xi2, v3 *wi, double *pdf) {
  int k = c->s->mat_kind[material];
  if (k == PTAMD_MAT_DIFFUSE || k == PTAMD_MAT_EMISSION || k == PTAMD_MAT_MICROFACET) {
    *wi = cosine_sample(Xi1, Xi2, pdf);
  } else {
    *wi = V(0, 0, 0);
    *pdf = 0; /* uninitialised in the reference; never used because the d = 0 ray hits nothing */
  }
  return bsdf_f(c, material);
}

/* ---- specular mode (ptamd_render_params.specular; include/ptamd.h states the model).
 * The reference declares these BSDFs and leaves them empty (advanced_bsdf.cpp:21-27,
 * 99-106, 124-134, 146-169), so nothing below has a counterpart to be checked against:
 * parity unpinned. The tests check the formulas analytically and the GPU against this. */
static int is_delta_kind(int k) { return k == PTAMD_MAT_MIRROR || k == PTAMD_MAT_REFRACTION || k == PTAMD_MAT_GLASS; }
static v3 reflect_z(v3 wo) { return V(-wo.x, -wo.y, wo.z); }
static int refract_z(v3 wo, double ior, v3 *wi, double *eta_out) {
  double eta = wo.z > 0 ? 1.0 / ior : ior;
  double c2 = 1.0 - eta * eta * (1.0 - wo.z * wo.z);
  *eta_out = eta;
  if (c2 < 0) return 0;
  double cz = sqrt(c2);
  *wi = V(-eta * wo.x, -eta * wo.y, wo.z > 0 ? -cz : cz);
  return 1;
}
/* Returns f; *wi = (0,0,0) and f = 0 when the path ends (refraction at total
 * internal reflection). u is the vertex's first BSDF random number. */
static v3 specular_sample_f(int kind, const double *param3, const double *extra4, v3 wo, double u, v3 *wi, double *pdf) {
  v3 a = ld3(param3);
  double ior = extra4 ? extra4[3] : 0.0;
  double eta;
  *pdf = 1.0;
  if (kind == PTAMD_MAT_MIRROR) {
    *wi = reflect_z(wo);
    return vdivs(a, fabs(wi->z));
  }
  if (kind == PTAMD_MAT_REFRACTION) {
    if (!refract_z(wo, ior, wi, &eta)) { *wi = V(0, 0, 0); return V(0, 0, 0); }
    return vdivs(vdivs(a, fabs(wi->z)), eta * eta);
  }
  /* glass: a = transmittance, extra = reflectance */
  v3 refl = extra4 ? ld3(extra4) : V(0, 0, 0);
  v3 wt;
  if (!refract_z(wo, ior, &wt, &eta)) {
    *wi = reflect_z(wo);
    return vdivs(refl, fabs(wi->z));
  }
  double r0 = (1.0 - ior) / (1.0 + ior);
  r0 = r0 * r0;
  double m = 1.0 - fabs(wo.z);
  double R = r0 + (1.0 - r0) * (m * m * m * m * m);
  if (u < R) {
    *wi = reflect_z(wo);
    *pdf = R;
    return vdivs(vscale(refl, R), fabs(wi->z));
  }
  *wi = wt;
  *pdf = 1.0 - R;
  return vdivs(vdivs(vscale(a, 1.0 - R), fabs(wi->z)), eta * eta);
}
void oracle_kat_specular(int64_t n, const int32_t *kind, const double *param3, const double *extra4, const double *wo,
                         const double *u, double *wi_out, double *f_out, double *pdf_out) {
  for (int64_t i = 0; i < n; i++) {
    v3 wi;
    double pdf;
    v3 f = specular_sample_f(kind[i], param3 + 3 * i, extra4 + 4 * i, ld3(wo + 3 * i), u[i], &wi, &pdf);
    wi_out[3 * i] = wi.x; wi_out[3 * i + 1] = wi.y; wi_out[3 * i + 2] = wi.z;
    f_out[3 * i] = f.x; f_out[3 * i + 1] = f.y; f_out[3 * i + 2] = f.z;
    pdf_out[i] = pdf;
  }
}

/* SceneLight::sample_L for every light kind (src/scene/light.cpp:17-93) */
static v3 light_sample_L(const ctx_t *c, int li, v3 p, double sx, double sy, v3 *wi, double *dist, double *pdf) {
  const double *q = c->s->light_param + 16 * (size_t)li;
  v3 radiance = ld3(q);
  switch (c->s->light_kind[li]) {
    case PTAMD_LIGHT_AREA: {
      v3 position = ld3(q + 3), direction = ld3(q + 6), dim_x = ld3(q + 9), dim_y = ld3(q + 12);
      double area = q[15];
      double ax = sx - 0.5f, ay = sy - 0.5f; /* sampler.get_sample() - Vector2D(0.5f, 0.5f) */
      v3 d = vsub(vadd(vadd(position, vscale(dim_x, ax)), vscale(dim_y, ay)), p);
      double cosTheta = vdot(d, direction);
      double sqDist = vnorm2(d);
      double dd = sqrt(sqDist);
      *wi = vdivs(d, dd);
      *dist = dd;
      *pdf = sqDist / (area * fabs(cosTheta));
      return cosTheta < 0 ? radiance : V(0, 0, 0);
    }
    case PTAMD_LIGHT_POINT: {
      v3 d = vsub(ld3(q + 3), p);
      *wi = vunit(d);
      *dist = vnorm(d);
      *pdf = 1.0;
      return radiance;
    }
    case PTAMD_LIGHT_DIRECTIONAL:
      *wi = ld3(q + 6);
      *dist = INFINITY;
      *pdf = 1.0;
      return radiance;
    case PTAMD_LIGHT_INF_HEMISPHERE: {
      v3 dir = uniform_hemisphere_sample(sx, sy);
      m3 s2w; /* light.cpp:27-31 */
      s2w.c[0] = V(1, 0, 0);
      s2w.c[1] = V(0, 0, -1);
      s2w.c[2] = V(0, 1, 0);
      *wi = mmul(&s2w, dir);
      *dist = INFINITY;
      *pdf = 1.0 / (2.0 * PI_D);
      return radiance;
    }
    default: /* SpotLight::sample_L is a stub (light.cpp:67-70): wi, dist, pdf untouched */
      *wi = V(0, 0, 0);
      *dist = 0;
      *pdf = 0;
      return V(0, 0, 0);
  }
}
static int light_is_delta(int kind) { /* src/scene/light.h:21,36,52,67,85 */
  return kind == PTAMD_LIGHT_DIRECTIONAL || kind == PTAMD_LIGHT_POINT || kind == PTAMD_LIGHT_SPOT;
}
static int light_samples_per_vertex(const ptamd_scene_desc *s, const ptamd_render_params *p) {
  if (p->hemisphere) return s->n_lights * p->ns_area_light; /* pathtracer.cpp:70 */
  int n = 0;
  for (int i = 0; i < s->n_lights; i++) n += light_is_delta(s->light_kind[i]) ? 1 : p->ns_area_light;
  return n;
}
static inline void light_draw(const rng_t *g, uint32_t base, int q, double *sx, double *sy) {
  double u[4];
  rng_block(g, base + 1u + (uint32_t)(q / 2), u);
  *sx = u[2 * (q % 2)];
  *sy = u[2 * (q % 2) + 1];
}

/* src/pathtracer/pathtracer.cpp:93-135 */
static v3 estimate_direct_lighting_importance(ctx_t *c, const rng_t *g, const ray_t *r, const isect_t *isect) {
  m3 o2w;
  make_coord_space(&o2w, isect->n);
  m3 w2o = mtrans(&o2w);
  const v3 hit_p = vadd(r->o, vscale(r->d, isect->t));
  const v3 w_out = mmul(&w2o, vneg(r->d));
  (void)w_out; /* DiffuseBSDF::f ignores its directions */
  v3 L_out = V(0, 0, 0);
  uint32_t base = 1u + (uint32_t)(r->depth - 1) * (uint32_t)c->bpv;
  size_t sample_count = 0;
  int q = 0;
  for (int li = 0; li < c->s->n_lights; li++) {
    int num_samples = light_is_delta(c->s->light_kind[li]) ? 1 : c->p->ns_area_light;
    sample_count += num_samples;
    for (int j = 0; j < num_samples; j++, q++) {
      double sx, sy, distToLight, pdf;
      v3 wi;
      light_draw(g, base, q, &sx, &sy);
      v3 radiance = light_sample_L(c, li, hit_p, sx, sy, &wi, &distToLight, &pdf);
      v3 wi_o = mmul(&w2o, wi);
      if (wi_o.z < 0 || (radiance.x == 0 && radiance.y == 0 && radiance.z == 0)) continue;
      ray_t shadow_ray = make_ray(hit_p, wi);
      shadow_ray.min_t = EPS_D;
      shadow_ray.max_t = distToLight;
      isect_t light_isect;
      c->shadow_rays++;
      c->in_shadow = 1;
      int occluded = bvh_intersect(c, &shadow_ray, &light_isect);
      c->in_shadow = 0;
      if (!occluded) {
        /* f * radiance * abs_cos_theta(wi_o) / pdf */
        v3 term = vdivs(vscale(vmul(bsdf_f(c, isect->material), radiance), fabs(wi_o.z)), pdf);
        L_out = vadd(L_out, term);
      }
    }
  }
  L_out = vscale(L_out, 1. / (double)sample_count); /* operator/= */
  return L_out;
}

/* src/pathtracer/pathtracer.cpp:47-91 */
static v3 estimate_direct_lighting_hemisphere(ctx_t *c, const rng_t *g, const ray_t *r, const isect_t *isect) {
  m3 o2w;
  make_coord_space(&o2w, isect->n);
  m3 w2o = mtrans(&o2w);
  const v3 hit_p = vadd(r->o, vscale(r->d, isect->t));
  int num_samples = c->s->n_lights * c->p->ns_area_light;
  v3 L_out = V(0, 0, 0);
  uint32_t base = 1u + (uint32_t)(r->depth - 1) * (uint32_t)c->bpv;
  for (int i = 0; i < num_samples; i++) {
    double sx, sy;
    light_draw(g, base, i, &sx, &sy);
    v3 wi = uniform_hemisphere_sample(sx, sy);
    ray_t shadow_ray = make_ray(hit_p, mmul(&o2w, wi));
    shadow_ray.min_t = EPS_D;
    isect_t light_isect;
    c->shadow_rays++;
    c->in_shadow = 1;
    int found = bvh_intersect(c, &shadow_ray, &light_isect);
    c->in_shadow = 0;
    if (found) {
      v3 light_L = bsdf_emission(c, light_isect.material);
      /* f * light_L * dot(wi, w2o*isect.n) * 2 * PI */
      double cs = vdot(wi, mmul(&w2o, isect->n));
      v3 term = vscale(vscale(vscale(vmul(bsdf_f(c, isect->material), light_L), cs), 2), PI_D);
      L_out = vadd(L_out, term);
    }
  }
  L_out = vscale(L_out, 1. / (double)num_samples);
  return L_out;
}

/* src/pathtracer/pathtracer.cpp:145-156 */
static v3 one_bounce_radiance(ctx_t *c, const rng_t *g, const ray_t *r, const isect_t *isect) {
  if (c->p->hemisphere) return estimate_direct_lighting_hemisphere(c, g, r, isect);
  return estimate_direct_lighting_importance(c, g, r, isect);
}

/* src/pathtracer/pathtracer.cpp:160-201 */
static v3 at_least_one_bounce_radiance(ctx_t *c, const rng_t *g, const ray_t *r, const isect_t *isect) {
  m3 o2w;
  make_coord_space(&o2w, isect->n);
  m3 w2o = mtrans(&o2w);
  v3 hit_p = vadd(r->o, vscale(r->d, isect->t));
  v3 w_out = mmul(&w2o, vneg(r->d));
  v3 L_out = V(0, 0, 0);
  c->vertices++;
  const int kind = c->s->mat_kind[isect->material];
  const int specular = c->p->specular;
  const int delta = specular && is_delta_kind(kind);
  if (specular && (kind == PTAMD_MAT_EMISSION || kind == PTAMD_MAT_MICROFACET)) {
    /* a BSDF that is identically zero ends the path; an emitter reached from a
     * delta vertex is seen directly (no light sample could have covered it) */
    return r->via_delta ? bsdf_emission(c, isect->material) : V(0, 0, 0);
  }
  int max_ray_depth = c->p->max_ray_depth;
  int isAccumBounces = c->p->accumulate_bounces;
  uint32_t base = 1u + (uint32_t)(r->depth - 1) * (uint32_t)c->bpv;
  double u[4];
  rng_block(g, base, u);
  int ignore_RRT = r->depth == 1 && max_ray_depth > 1;
  /* the coin is drawn only when the short-circuit reaches it; with a counter-
   * based generator an unread word changes nothing */
  int not_terminate = ignore_RRT || (r->depth < max_ray_depth && (!isAccumBounces || (u[0] < RRT)));
  /* a delta vertex takes no light samples: its BSDF is zero for every sampled direction */
  if (!not_terminate) return delta ? V(0, 0, 0) : one_bounce_radiance(c, g, r, isect);

  v3 wi;
  double pdf;
  v3 f;
  if (delta) {
    const double *extra = c->s->mat_extra ? c->s->mat_extra + 4 * (size_t)isect->material : NULL;
    f = specular_sample_f(kind, c->s->mat_param + 3 * (size_t)isect->material, extra, w_out, u[1], &wi, &pdf);
  } else {
    f = bsdf_sample_f(c, isect->material, u[1], u[2], &wi, &pdf);
  }
  ray_t bounce_ray = make_ray(hit_p, mmul(&o2w, wi));
  bounce_ray.min_t = EPS_D;
  bounce_ray.depth = r->depth + 1;
  bounce_ray.via_delta = delta;
  isect_t bounce_isect;
  if (bvh_intersect(c, &bounce_ray, &bounce_isect)) {
    v3 bounce_radiance = at_least_one_bounce_radiance(c, g, &bounce_ray, &bounce_isect);
    /* f * abs_cos_theta(wi) * bounce_radiance / pdf / RRT */
    v3 ind = vdivs(vdivs(vmul(vscale(f, fabs(wi.z)), bounce_radiance), pdf), RRT);
    if (isAccumBounces) L_out = delta ? ind : vadd(one_bounce_radiance(c, g, r, isect), ind);
    else L_out = ind;
  } else {
    L_out = delta ? V(0, 0, 0) : one_bounce_radiance(c, g, r, isect);
  }
  return L_out;
}

/* src/pathtracer/pathtracer.cpp:203-232 (the environment light is a stub -> 0) */
static v3 est_radiance_global_illumination(ctx_t *c, const rng_t *g, ray_t *r) {
  isect_t isect;
  v3 L_out = V(0, 0, 0);
  if (!bvh_intersect(c, r, &isect)) return L_out;
  if (c->p->max_ray_depth == 0) L_out = bsdf_emission(c, isect.material);
  else if (c->p->accumulate_bounces) L_out = vadd(bsdf_emission(c, isect.material), at_least_one_bounce_radiance(c, g, r, &isect));
  else L_out = at_least_one_bounce_radiance(c, g, r, &isect);
  return L_out;
}

static v3 sample_radiance(ctx_t *c, int x, int y, int sample) {
  const ptamd_render_params *p = c->p;
  rng_t g;
  g.k0 = (uint32_t)p->seed;
  g.k1 = (uint32_t)(x + y * p->w);
  g.sample = (uint32_t)sample;
  g.seed_hi = (uint32_t)(p->seed >> 32);
  double u[4];
  rng_block(&g, 0u, u);
  /* JitteredSampler2D::get_sample, src/pathtracer/sampler.cpp:14-18 */
  int n = c->s->jitter_n;
  size_t ji = (size_t)floor(u[0] * n);
  size_t jj = (size_t)floor(u[1] * n);
  const double *js = c->s->jitter_table + 2 * (ji * (size_t)n + jj);
  double sxp = (double)x + js[0], syp = (double)y + js[1]; /* origin + sample */
  ray_t r = generate_ray(c->cam, sxp / (double)p->w, syp / (double)p->h);
  r.depth = 1;
  return est_radiance_global_illumination(c, &g, &r);
}

/* Vector3D::illum, CGL/include/CGL/vector3D.h:231-233: evaluated in double,
 * returned as float */
static inline float illum(v3 v) { return (float)(0.2126f * v.x + 0.7152f * v.y + 0.0722f * v.z); }

/* src/pathtracer/pathtracer.cpp:234-272 */
static void raytrace_pixel(ctx_t *c, int x, int y, double *hdr, int32_t *count) {
  const ptamd_render_params *p = c->p;
  int num_samples = p->ns_aa;
  v3 pixel_sum = V(0, 0, 0);
  double s1 = 0, s2 = 0;
  int i = 0;
  for (; i < num_samples; i++) {
    if (i > 0 && i % p->samples_per_batch == 0) {
      double one_over_i = 1.0 / i, one_over_im1 = 1.0 / (i - 1);
      double mean = s1 * one_over_i;
      double var = one_over_im1 * (s2 - s1 * s1 * one_over_i);
      double I = 1.96 * sqrt(var * one_over_i);
      if (I <= p->max_tolerance * mean) break;
    }
    v3 radiance = sample_radiance(c, x, y, i);
    double xl = illum(radiance);
    s1 += xl;
    s2 += xl * xl;
    pixel_sum = vadd(pixel_sum, radiance);
  }
  pixel_sum = vscale(pixel_sum, 1. / (double)i);
  size_t idx = (size_t)x + (size_t)y * (size_t)p->w;
  hdr[idx * 3] = pixel_sum.x;
  hdr[idx * 3 + 1] = pixel_sum.y;
  hdr[idx * 3 + 2] = pixel_sum.z;
  count[idx] = i;
}

static void ctx_init(ctx_t *c, const ptamd_scene_desc *s, const ptamd_camera *cam, const ptamd_render_params *p) {
  memset(c, 0, sizeof *c);
  c->s = s;
  c->cam = cam;
  c->p = p;
  c->nls = light_samples_per_vertex(s, p);
  c->bpv = 1 + (c->nls + 1) / 2;
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int oracle_render(const ptamd_scene_desc *scene, const ptamd_camera *cam, const ptamd_render_params *p,
                  int32_t threads, double *hdr, int32_t *count, oracle_stats *stats) {
  if (!scene || !cam || !p || !hdr || !count || p->w <= 0 || p->h <= 0 || p->samples_per_batch <= 0) return 1;
  memset(hdr, 0, sizeof(double) * 3 * (size_t)p->w * p->h);
  memset(count, 0, sizeof(int32_t) * (size_t)p->w * p->h);
  /* 32x32 tiles in row-major order, src/pathtracer/raytraced_renderer.cpp:297-302 */
  const int T = 32;
  int x0 = p->x0, y0 = p->y0, x1 = p->x1, y1 = p->y1;
  if (x1 > p->w) x1 = p->w;
  if (y1 > p->h) y1 = p->h;
  int ntx = (x1 - x0 + T - 1) / T, nty = (y1 - y0 + T - 1) / T;
  if (ntx < 0) ntx = 0;
  if (nty < 0) nty = 0;
  if (threads < 1) threads = 1;
  uint64_t rays = 0, shadow = 0, verts = 0, nodes = 0, tests = 0, snodes = 0, stests = 0;
  double t_begin = now_s();
#ifdef _OPENMP
#pragma omp parallel num_threads(threads) reduction(+ : rays, shadow, verts, nodes, tests, snodes, stests)
#endif
  {
    ctx_t c;
    ctx_init(&c, scene, cam, p);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int tile = 0; tile < ntx * nty; tile++) {
      int tx = x0 + (tile % ntx) * T, ty = y0 + (tile / ntx) * T;
      int ex = tx + T < x1 ? tx + T : x1, ey = ty + T < y1 ? ty + T : y1;
      for (int y = ty; y < ey; y++)
        for (int x = tx; x < ex; x++) raytrace_pixel(&c, x, y, hdr, count);
    }
    rays += c.rays; shadow += c.shadow_rays; verts += c.vertices; nodes += c.nodes; tests += c.tests;
    snodes += c.shadow_nodes; stests += c.shadow_tests;
  }
  if (stats) {
    stats->seconds = now_s() - t_begin;
    stats->rays = rays; stats->shadow_rays = shadow; stats->path_vertices = verts;
    stats->nodes_visited = nodes; stats->prim_tests = tests; stats->threads = threads;
    stats->shadow_nodes_visited = snodes; stats->shadow_prim_tests = stests;
    uint64_t total = 0;
    for (size_t i = 0; i < (size_t)p->w * p->h; i++) total += (uint64_t)count[i];
    stats->samples = total;
  }
  return 0;
}

int oracle_trace_primary(const ptamd_scene_desc *scene, const ptamd_camera *cam, int64_t n, const double *sx,
                         const double *sy, int32_t *prim_out, double *t_out) {
  ptamd_render_params p;
  memset(&p, 0, sizeof p);
  ctx_t c;
  ctx_init(&c, scene, cam, &p);
  for (int64_t i = 0; i < n; i++) {
    ray_t r = generate_ray(cam, sx[i], sy[i]);
    r.depth = 1;
    isect_t is;
    int hit = bvh_intersect(&c, &r, &is);
    prim_out[i] = hit ? is.prim : -1;
    t_out[i] = hit ? is.t : -1.0;
  }
  return 0;
}

int oracle_sample_radiance(const ptamd_scene_desc *scene, const ptamd_camera *cam, const ptamd_render_params *p,
                           int32_t x, int32_t y, int32_t sample, double rgb_out[3]) {
  ctx_t c;
  ctx_init(&c, scene, cam, p);
  v3 L = sample_radiance(&c, x, y, sample);
  rgb_out[0] = L.x; rgb_out[1] = L.y; rgb_out[2] = L.z;
  return 0;
}

/* ---- pure-function entry points (layouts of the reference-generated vectors) */
void oracle_kat_bbox(int64_t n, const double *in, double *out) {
  for (int64_t i = 0; i < n; i++) {
    const double *q = in + 14 * i;
    ray_t r = make_ray(ld3(q + 6), ld3(q + 9));
    r.min_t = q[12];
    r.max_t = q[13];
    double t0 = -7, t1 = -7;
    int hit = bbox_intersect(q, &r, &t0, &t1);
    out[3 * i] = hit; out[3 * i + 1] = t0; out[3 * i + 2] = t1;
  }
}
void oracle_kat_triangle(int64_t n, const double *in, double *out) {
  for (int64_t i = 0; i < n; i++) {
    const double *q = in + 26 * i;
    ray_t r = make_ray(ld3(q + 18), ld3(q + 21));
    r.min_t = q[24];
    r.max_t = q[25];
    double t = 0;
    v3 nn = V(0, 0, 0);
    int hit = triangle_intersect(q, q + 9, &r, &t, &nn);
    double *o = out + 6 * i;
    o[0] = hit; o[1] = hit ? t : 0; o[2] = hit ? nn.x : 0; o[3] = hit ? nn.y : 0; o[4] = hit ? nn.z : 0; o[5] = r.max_t;
  }
}
void oracle_kat_sphere(int64_t n, const double *in, double *out) {
  for (int64_t i = 0; i < n; i++) {
    const double *q = in + 12 * i;
    double P[9] = {q[0], q[1], q[2], q[3], q[3] * q[3], 0, 0, 0, 0};
    ray_t r = make_ray(ld3(q + 4), ld3(q + 7));
    r.min_t = q[10];
    r.max_t = q[11];
    double t = 0;
    v3 nn = V(0, 0, 0);
    int hit = sphere_intersect(P, &r, &t, &nn);
    double *o = out + 6 * i;
    o[0] = hit; o[1] = hit ? t : 0; o[2] = hit ? nn.x : 0; o[3] = hit ? nn.y : 0; o[4] = hit ? nn.z : 0; o[5] = r.max_t;
  }
}
void oracle_kat_frame(int64_t n, const double *in, double *out) {
  for (int64_t i = 0; i < n; i++) {
    m3 m;
    make_coord_space(&m, ld3(in + 3 * i));
    for (int c = 0; c < 3; c++) {
      out[9 * i + 3 * c] = m.c[c].x; out[9 * i + 3 * c + 1] = m.c[c].y; out[9 * i + 3 * c + 2] = m.c[c].z;
    }
  }
}
/* src/util/image.h:183-198 and 53-62 */
void oracle_kat_tonemap(int64_t n, const double *in, int32_t *out) {
  float gamma = 2.2f, level = 1.0f;
  float one_over_gamma = 1.0f / gamma;
  float exposure = (float)sqrt(pow(2, level));
  for (int64_t i = 0; i < n; i++) {
    float ch[3];
    for (int k = 0; k < 3; k++) ch[k] = (float)dmax(0.0, dmin(pow(in[3 * i + k] * exposure, one_over_gamma), 1.0));
    uint32_t px = 0;
    /* clamp(0.f, 1.f, c) in the reference is min(max(0,1), c) = min(1, c) */
    float b = ch[2] < 1.f ? ch[2] : 1.f, g = ch[1] < 1.f ? ch[1] : 1.f, r = ch[0] < 1.f ? ch[0] : 1.f;
    px |= ((uint32_t)(b * 255)) << 16;
    px |= ((uint32_t)(g * 255)) << 8;
    px |= ((uint32_t)(r * 255));
    px |= 0xFF000000u;
    out[i] = (int32_t)px;
  }
}
