// Host half of the C ABI: error state, the scene pipeline handle, and the
// output stage (tonemap, sampling-rate map, PNG).
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptamd.h"
#include "error_state.h"
#include "scene.h"

namespace ptamd {
static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
const char *get_error() { return g_error.c_str(); }
}  // namespace ptamd

using namespace ptamd;

struct ptamd_scene {
  StaticScene S;
  // flattened views handed out through ptamd_scene_desc
  std::vector<int32_t> prim_type, prim_material, node_left, node_right, node_start, node_count, mat_kind, light_kind;
  std::vector<double> prim_pos, prim_nrm, node_bbox, mat_param, mat_extra, light_param, jitter;
};

extern "C" {

const char *ptamd_last_error(void) { return ptamd::get_error(); }

int ptamd_scene_load(const char *dae_path, ptamd_scene **out) {
  if (!dae_path || !out) { set_error("null argument"); return PTAMD_ERR_ARG; }
  SceneInfo info;
  std::string err;
  if (collada_load(dae_path, &info, &err)) { set_error(err); return PTAMD_ERR_IO; }
  ptamd_scene *s = new ptamd_scene();
  if (build_static_scene(info, &s->S, &err)) {
    set_error(err);
    delete s;
    return PTAMD_ERR_SCENE;
  }
  const StaticScene &S = s->S;
  if (S.prims.empty()) {
    set_error("scene has no primitives");
    delete s;
    return PTAMD_ERR_SCENE;
  }
  const size_t n = S.prims.size();
  s->prim_type.resize(n);
  s->prim_material.resize(n);
  s->prim_pos.assign(n * 9, 0.0);
  s->prim_nrm.assign(n * 9, 0.0);
  for (size_t i = 0; i < n; i++) {
    const Primitive &p = S.prims[i];
    s->prim_type[i] = p.type;
    s->prim_material[i] = p.material;
    double *P = &s->prim_pos[i * 9], *N = &s->prim_nrm[i * 9];
    if (p.type == PRIM_TRIANGLE) {
      const V3 *pv[3] = {&p.p1, &p.p2, &p.p3};
      const V3 *nv[3] = {&p.n1, &p.n2, &p.n3};
      for (int k = 0; k < 3; k++)
        for (int c = 0; c < 3; c++) {
          P[k * 3 + c] = (*pv[k])[c];
          N[k * 3 + c] = (*nv[k])[c];
        }
    } else {
      P[0] = p.p1.x; P[1] = p.p1.y; P[2] = p.p1.z; P[3] = p.p2.x; P[4] = p.p2.y;
    }
  }
  const size_t N = S.nodes.size();
  s->node_bbox.resize(N * 6);
  s->node_left.resize(N); s->node_right.resize(N); s->node_start.resize(N); s->node_count.resize(N);
  for (size_t i = 0; i < N; i++) {
    const BvhNode &nd = S.nodes[i];
    for (int c = 0; c < 3; c++) {
      s->node_bbox[i * 6 + c] = nd.bb_min[c];
      s->node_bbox[i * 6 + 3 + c] = nd.bb_max[c];
    }
    s->node_left[i] = nd.l; s->node_right[i] = nd.r; s->node_start[i] = nd.start; s->node_count[i] = nd.count;
  }
  for (const Material &m : S.materials) {
    s->mat_kind.push_back(m.kind);
    // the three doubles the renderer reads: reflectance / radiance / transmittance
    // (microfacet eta/k/alpha are parsed but unused: its BSDF evaluates to zero)
    s->mat_param.push_back(m.a.x); s->mat_param.push_back(m.a.y); s->mat_param.push_back(m.a.z);
    // what only the specular mode reads: glass reflectance, index of refraction
    s->mat_extra.push_back(m.b.x); s->mat_extra.push_back(m.b.y); s->mat_extra.push_back(m.b.z);
    s->mat_extra.push_back(m.ior);
  }
  for (const Light &l : S.lights) {
    s->light_kind.push_back(l.kind);
    const V3 *v[5] = {&l.radiance, &l.position, &l.direction, &l.dim_x, &l.dim_y};
    for (int k = 0; k < 5; k++)
      for (int c = 0; c < 3; c++) s->light_param.push_back((*v[k])[c]);
    s->light_param.push_back(l.area);
  }
  make_jitter_table(&s->jitter, 16);
  *out = s;
  return PTAMD_OK;
}

void ptamd_scene_free(ptamd_scene *scene) { delete scene; }

int ptamd_scene_get_desc(const ptamd_scene *s, ptamd_scene_desc *d) {
  if (!s || !d) { set_error("null argument"); return PTAMD_ERR_ARG; }
  d->n_prims = (int32_t)s->prim_type.size();
  d->n_nodes = (int32_t)s->node_left.size();
  d->n_materials = (int32_t)s->mat_kind.size();
  d->n_lights = (int32_t)s->light_kind.size();
  d->root = s->S.root;
  d->jitter_n = 16;
  d->prim_type = s->prim_type.data();
  d->prim_pos = s->prim_pos.data();
  d->prim_nrm = s->prim_nrm.data();
  d->prim_material = s->prim_material.data();
  d->node_bbox = s->node_bbox.data();
  d->node_left = s->node_left.data();
  d->node_right = s->node_right.data();
  d->node_start = s->node_start.data();
  d->node_count = s->node_count.data();
  d->mat_kind = s->mat_kind.data();
  d->mat_param = s->mat_param.data();
  d->mat_extra = s->mat_extra.data();
  d->light_kind = s->light_kind.data();
  d->light_param = s->light_param.data();
  d->jitter_table = s->jitter.data();
  return PTAMD_OK;
}

int ptamd_scene_get_camera(ptamd_scene *s, int32_t w, int32_t h, ptamd_camera *out) {
  if (!s || !out) { set_error("null argument"); return PTAMD_ERR_ARG; }
  if (w < 0 || h < 0 || (w == 0) != (h == 0)) { set_error("invalid frame size"); return PTAMD_ERR_ARG; }
  CameraState &cam = s->S.camera;
  if (w && h) cam.set_screen_size((size_t)w, (size_t)h);
  for (int i = 0; i < 3; i++) out->pos[i] = cam.pos[i];
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) out->c2w[3 * c + r] = cam.c2w.col[c][r];
  out->hFov = cam.hFov; out->vFov = cam.vFov; out->nClip = cam.nClip; out->fClip = cam.fClip;
  return PTAMD_OK;
}

int ptamd_scene_load_camera_file(ptamd_scene *s, const char *path) {
  if (!s || !path) { set_error("null argument"); return PTAMD_ERR_ARG; }
  if (!s->S.camera.load_settings(path)) { set_error(std::string("cannot read camera settings from ") + path); return PTAMD_ERR_IO; }
  return PTAMD_OK;
}

int ptamd_scene_dump_camera_file(ptamd_scene *s, const char *path) {
  if (!s || !path) { set_error("null argument"); return PTAMD_ERR_ARG; }
  if (!s->S.camera.dump_settings(path)) { set_error(std::string("cannot write camera settings to ") + path); return PTAMD_ERR_IO; }
  return PTAMD_OK;
}

double ptamd_scene_build_seconds(const ptamd_scene *s) { return s ? s->S.build_seconds : 0.0; }

// HDRImageBuffer::toColor + ImageBuffer::update_pixel, src/util/image.h:183-198, 53-62
int ptamd_tonemap_rgba8(const double *hdr, int32_t w, int32_t h, uint32_t *rgba_out) {
  if (!hdr || !rgba_out || w <= 0 || h <= 0) { set_error("invalid arguments"); return PTAMD_ERR_ARG; }
  float gamma = 2.2f, level = 1.0f;
  float one_over_gamma = 1.0f / gamma;
  float exposure = (float)sqrt(pow(2, level));
  const size_t n = (size_t)w * h;
  for (size_t i = 0; i < n; i++) {
    float ch[3];
    for (int k = 0; k < 3; k++) ch[k] = (float)std::max(0.0, std::min(pow(hdr[3 * i + k] * exposure, one_over_gamma), 1.0));
    uint32_t p = 0;
    // clamp(0.f, 1.f, c): the reference passes (x, lo, hi) = (0, 1, c), i.e. min(1, c)
    p |= ((uint32_t)(std::min(std::max(0.f, 1.f), ch[2]) * 255)) << 16;
    p |= ((uint32_t)(std::min(std::max(0.f, 1.f), ch[1]) * 255)) << 8;
    p |= ((uint32_t)(std::min(std::max(0.f, 1.f), ch[0]) * 255));
    p |= 0xFF000000u;
    rgba_out[i] = p;
  }
  return PTAMD_OK;
}

// save_sampling_rate_image, src/pathtracer/raytraced_renderer.cpp:720-751
int ptamd_sampling_rate_rgba8(const int32_t *count, int32_t w, int32_t h, int32_t ns_aa, uint32_t *out) {
  if (!count || !out || w <= 0 || h <= 0 || ns_aa <= 0) { set_error("invalid arguments"); return PTAMD_ERR_ARG; }
  for (int x = 0; x < w; x++)
    for (int y = 0; y < h; y++) {
      float samplingRate = count[(size_t)y * w + x] * 1.0f / ns_aa;
      float cr, cg, cb;
      if (samplingRate <= 0.5) {
        float r = (0.5 - samplingRate) / 0.5;
        float q = (float)(1.0 - r);  // Color * (1.0 - r): the double narrows to float at the call
        cr = 0.0f * r + 0.0f * q; cg = 0.0f * r + 1.0f * q; cb = 1.0f * r + 0.0f * q;
      } else {
        float r = (1.0 - samplingRate) / 0.5;
        float q = (float)(1.0 - r);
        cr = 0.0f * r + 1.0f * q; cg = 1.0f * r + 0.0f * q; cb = 0.0f * r + 0.0f * q;
      }
      uint32_t p = 0;
      p |= ((uint32_t)(std::min(1.f, cb) * 255)) << 16;
      p |= ((uint32_t)(std::min(1.f, cg) * 255)) << 8;
      p |= ((uint32_t)(std::min(1.f, cr) * 255));
      p |= 0xFF000000u;
      out[(size_t)(h - 1 - y) * w + x] = p;
    }
  return PTAMD_OK;
}

static void put_be32(std::vector<unsigned char> &v, uint32_t x) {
  v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}
static void png_chunk(std::vector<unsigned char> &png, const char *type, const std::vector<unsigned char> &data) {
  put_be32(png, (uint32_t)data.size());
  size_t start = png.size();
  png.insert(png.end(), type, type + 4);
  png.insert(png.end(), data.begin(), data.end());
  uint32_t crc = crc32(0L, png.data() + start, (uInt)(png.size() - start));
  put_be32(png, crc);
}
static int write_png_top_down(const char *path, const uint32_t *rgba, int32_t w, int32_t h) {
  std::vector<unsigned char> raw((size_t)h * ((size_t)w * 4 + 1));
  for (int y = 0; y < h; y++) {
    unsigned char *row = &raw[(size_t)y * ((size_t)w * 4 + 1)];
    row[0] = 0;
    memcpy(row + 1, rgba + (size_t)y * w, (size_t)w * 4);  // little-endian host: bytes are R,G,B,A
  }
  uLongf clen = compressBound((uLong)raw.size());
  std::vector<unsigned char> comp(clen);
  if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { set_error("zlib compression failed"); return PTAMD_ERR_IO; }
  comp.resize(clen);
  std::vector<unsigned char> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<unsigned char> ihdr;
  put_be32(ihdr, (uint32_t)w);
  put_be32(ihdr, (uint32_t)h);
  ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  png_chunk(png, "IHDR", ihdr);
  png_chunk(png, "IDAT", comp);
  png_chunk(png, "IEND", {});
  FILE *f = fopen(path, "wb");
  if (!f) { set_error(std::string("cannot write ") + path); return PTAMD_ERR_IO; }
  size_t wr = fwrite(png.data(), 1, png.size(), f);
  fclose(f);
  if (wr != png.size()) { set_error(std::string("short write to ") + path); return PTAMD_ERR_IO; }
  return PTAMD_OK;
}

// save_image, src/pathtracer/raytraced_renderer.cpp:680-718: rows flipped, alpha forced to 255
int ptamd_write_png(const char *path, const uint32_t *rgba_bottom_up, int32_t w, int32_t h) {
  if (!path || !rgba_bottom_up || w <= 0 || h <= 0) { set_error("invalid arguments"); return PTAMD_ERR_ARG; }
  std::vector<uint32_t> out((size_t)w * h);
  for (int i = 0; i < h; i++) memcpy(&out[(size_t)i * w], rgba_bottom_up + (size_t)(h - i - 1) * w, 4 * (size_t)w);
  for (auto &p : out) p |= 0xFF000000u;
  return write_png_top_down(path, out.data(), w, h);
}

int ptamd_write_png_top_down(const char *path, const uint32_t *rgba, int32_t w, int32_t h) {
  if (!path || !rgba || w <= 0 || h <= 0) { set_error("invalid arguments"); return PTAMD_ERR_ARG; }
  return write_png_top_down(path, rgba, w, h);
}

}  // extern "C"
