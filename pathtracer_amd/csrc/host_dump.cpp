// ptamd_dump: runs the host scene pipeline (COLLADA -> static scene -> BVH ->
// camera) and writes the resulting arrays as a PTAR file, under the same array
// names the parity fixtures use. A developer/diagnostic tool; no GPU needed.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "ptar.h"
#include "scene.h"

using namespace ptamd;

static void put3(std::vector<double> &v, const V3 &p) {
  v.push_back(p.x);
  v.push_back(p.y);
  v.push_back(p.z);
}

int main(int argc, char **argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: ptamd_dump scene.dae W H out.ptar\n");
    return 1;
  }
  SceneInfo info;
  std::string err;
  if (collada_load(argv[1], &info, &err)) {
    fprintf(stderr, "%s\n", err.c_str());
    return 2;
  }
  StaticScene S;
  if (build_static_scene(info, &S, &err)) {
    fprintf(stderr, "%s\n", err.c_str());
    return 3;
  }
  size_t W = atoi(argv[2]), H = atoi(argv[3]);
  if (W && H) S.camera.set_screen_size(W, H);
  FILE *f = ptar_open_write(argv[4]);
  if (!f) return 4;
  size_t n = S.prims.size();
  std::vector<int32_t> ptype(n), pmat(n);
  std::vector<double> pos(n * 9, 0.0), nrm(n * 9, 0.0);
  std::vector<int32_t> pkind(n);
  std::vector<double> pparam(n * 3);
  for (size_t i = 0; i < n; i++) {
    const Primitive &p = S.prims[i];
    ptype[i] = p.type;
    pmat[i] = p.material;
    const Material &m = S.materials[p.material];
    pkind[i] = m.kind;
    // microfacet parameters are not part of the shared fixture layout
    V3 a = (m.kind == MAT_MICROFACET) ? V3() : m.a;
    pparam[i * 3] = a.x; pparam[i * 3 + 1] = a.y; pparam[i * 3 + 2] = a.z;
    if (p.type == PRIM_TRIANGLE) {
      const V3 *pv[3] = {&p.p1, &p.p2, &p.p3};
      const V3 *nv[3] = {&p.n1, &p.n2, &p.n3};
      for (int k = 0; k < 3; k++)
        for (int c = 0; c < 3; c++) {
          pos[i * 9 + k * 3 + c] = (*pv[k])[c];
          nrm[i * 9 + k * 3 + c] = (*nv[k])[c];
        }
    } else {
      pos[i * 9 + 0] = p.p1.x; pos[i * 9 + 1] = p.p1.y; pos[i * 9 + 2] = p.p1.z;
      pos[i * 9 + 3] = p.p2.x; pos[i * 9 + 4] = p.p2.y;
    }
  }
  ptar_put(f, "prim_type", PTAR_I32, ptype.data(), n);
  ptar_put(f, "prim_mat_kind", PTAR_I32, pkind.data(), n);
  ptar_put(f, "prim_mat_param", PTAR_F64, pparam.data(), pparam.size());
  ptar_put(f, "prim_pos", PTAR_F64, pos.data(), pos.size());
  ptar_put(f, "prim_nrm", PTAR_F64, nrm.data(), nrm.size());
  size_t N = S.nodes.size();
  std::vector<double> nb(N * 6);
  std::vector<int32_t> nl(N), nr(N), ns(N), nc(N);
  for (size_t i = 0; i < N; i++) {
    const BvhNode &nd = S.nodes[i];
    for (int c = 0; c < 3; c++) {
      nb[i * 6 + c] = nd.bb_min[c];
      nb[i * 6 + 3 + c] = nd.bb_max[c];
    }
    nl[i] = nd.l; nr[i] = nd.r; ns[i] = nd.start; nc[i] = nd.count;
  }
  ptar_put(f, "node_bbox", PTAR_F64, nb.data(), nb.size());
  ptar_put(f, "node_left", PTAR_I32, nl.data(), N);
  ptar_put(f, "node_right", PTAR_I32, nr.data(), N);
  ptar_put(f, "node_start", PTAR_I32, ns.data(), N);
  ptar_put(f, "node_count", PTAR_I32, nc.data(), N);
  ptar_put(f, "root", PTAR_I32, &S.root, 1);
  size_t L = S.lights.size();
  std::vector<int32_t> lk(L);
  std::vector<double> lp(L * 16, 0.0);
  for (size_t i = 0; i < L; i++) {
    const Light &l = S.lights[i];
    lk[i] = l.kind;
    double *q = &lp[i * 16];
    const V3 *v[5] = {&l.radiance, &l.position, &l.direction, &l.dim_x, &l.dim_y};
    for (int k = 0; k < 5; k++)
      for (int c = 0; c < 3; c++) q[k * 3 + c] = (*v[k])[c];
    q[15] = l.area;
  }
  ptar_put(f, "light_kind", PTAR_I32, lk.data(), L);
  ptar_put(f, "light_param", PTAR_F64, lp.data(), lp.size());
  const CameraState &cam = S.camera;
  std::vector<double> c;
  put3(c, cam.pos);
  for (int k = 0; k < 3; k++) put3(c, cam.c2w.col[k]);
  c.push_back(cam.hFov); c.push_back(cam.vFov); c.push_back(cam.nClip); c.push_back(cam.fClip);
  c.push_back((double)cam.screenW); c.push_back((double)cam.screenH); c.push_back(cam.screenDist); c.push_back(cam.ar);
  put3(c, cam.targetPos);
  c.push_back(cam.phi); c.push_back(cam.theta); c.push_back(cam.r);
  ptar_put(f, "camera", PTAR_F64, c.data(), c.size());
  std::vector<double> jt;
  make_jitter_table(&jt, 16);
  ptar_put(f, "jitter_table", PTAR_F64, jt.data(), jt.size());
  fclose(f);
  fprintf(stderr, "ptamd_dump: %zu primitives, %zu nodes, %zu lights, BVH build %.3f s\n", n, N, L, S.build_seconds);
  return 0;
}
