// Host-side construction of what the traversal kernels walk: the reference's primitive visiting
// order, the primitives at the far end of the lights' shadow rays, the 4-wide tree collapsed from
// the caller's binary BVH with its quantised child boxes, and the primitive records in device
// order. No GPU code or headers: compiled by the host compiler and unit-tested on the CPU
// (tests/test_tree4.py).
#include "tree4.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <utility>

#include "error_state.h"

using namespace ptk;

namespace ptamd {

// Fills the quantised boxes of a 4-wide node: per axis a grid step 2^e and an
// origin on that grid such that every child plane lies within 255 steps and every
// decoded plane origin + q * 2^e is exactly representable in float. lo planes are
// rounded down, hi planes up (checked in double, where all of this is exact).
static bool quantize_node(DevNode &n, const double boxes[][6], int nk) {
  n.exps = 0;
  for (int a = 0; a < 3; a++) {
    double lo = INFINITY, hi = -INFINITY;
    for (int k = 0; k < nk; k++) { lo = std::min(lo, boxes[k][a]); hi = std::max(hi, boxes[k][3 + a]); }
    if (!(std::isfinite(lo) && std::isfinite(hi)) || hi < lo) return false;
    int e;
    {
      const double ext = std::max(hi - lo, 1e-30);
      (void)std::frexp(ext / 255.0, &e);  // ext / 255 < 2^e
      e = std::max(e, -100);
    }
    for (;; e++) {
      if (e > 100) return false;
      const double s = std::ldexp(1.0, e);
      const double k0 = std::floor(lo / s);
      if (std::fabs(k0) > 4194304.0) continue;  // origin + 255 steps must stay below 2^24 steps
      const double org = k0 * s;
      if (std::ceil((hi - org) / s) > 255.0) continue;
      bool ok = true;
      uint8_t q[4][2];
      for (int k = 0; k < nk && ok; k++) {
        double ql = std::floor((boxes[k][a] - org) / s), qh = std::ceil((boxes[k][3 + a] - org) / s);
        while (ql > 0 && org + ql * s > boxes[k][a]) ql -= 1;
        while (org + qh * s < boxes[k][3 + a]) qh += 1;
        if (ql < 0 || qh > 255 || org + ql * s > boxes[k][a]) { ok = false; break; }
        // the device evaluates fmaf(q, s, org) in float: must be exact
        if ((double)(float)(org + ql * s) != org + ql * s || (double)(float)(org + qh * s) != org + qh * s) { ok = false; break; }
        q[k][0] = (uint8_t)ql; q[k][1] = (uint8_t)qh;
      }
      if (!ok || (double)(float)org != org) continue;
      n.org[a] = (float)org;
      n.exps |= (uint32_t)(uint8_t)(int8_t)e << (8 * a);
      n.qlo[a] = n.qhi[a] = 0;
      for (int k = 0; k < nk; k++) {
        n.qlo[a] |= (uint32_t)q[k][0] << (8 * k);
        n.qhi[a] |= (uint32_t)q[k][1] << (8 * k);
      }
      break;
    }
  }
  return true;
}

int build_tree4(const ptamd_scene_desc *d, Tree4 &out) {
  // ---- validate the tree and find the reference's visiting order: depth-first,
  // right child first, leaf primitives in ascending order (bvh.cpp:177-196)
  const int N = d->n_nodes, NP = d->n_prims;
  std::vector<int32_t> visit_order;  // caller's primitive indices in visiting order
  visit_order.reserve(NP);
  std::vector<int32_t> leaf_new_start(N, -1);
  std::vector<char> seen(N, 0);
  std::vector<char> prim_seen(NP, 0);
  int max_depth = 0;
  {
    std::vector<std::pair<int32_t, int>> st;
    st.push_back({d->root, 1});
    while (!st.empty()) {
      auto [n, depth] = st.back();
      st.pop_back();
      if (n < 0 || n >= N || seen[n]) { ptamd::set_error("BVH node graph is not a tree"); return PTAMD_ERR_SCENE; }
      seen[n] = 1;
      max_depth = std::max(max_depth, depth);
      if (d->node_count[n] > 0) {
        const int s = d->node_start[n], cnt = d->node_count[n];
        if (s < 0 || s + cnt > NP) { ptamd::set_error("BVH leaf range outside the primitive array"); return PTAMD_ERR_SCENE; }
        if (cnt > 4) { ptamd::set_error("BVH leaves of more than 4 primitives are not supported (the reference builds with max_leaf_size = 4)"); return PTAMD_ERR_SCENE; }
        leaf_new_start[n] = (int32_t)visit_order.size();
        for (int p = s; p < s + cnt; p++) {
          if (prim_seen[p]) { ptamd::set_error("primitive referenced by two leaves"); return PTAMD_ERR_SCENE; }
          prim_seen[p] = 1;
          visit_order.push_back(p);
        }
      } else {
        st.push_back({d->node_left[n], depth + 1});   // popped second
        st.push_back({d->node_right[n], depth + 1});  // popped first
      }
    }
  }
  if ((int)visit_order.size() != NP) { ptamd::set_error("BVH leaves do not cover every primitive"); return PTAMD_ERR_SCENE; }
  for (int p = 0; p < NP; p++) {
    if (d->prim_material[p] < 0 || d->prim_material[p] >= d->n_materials) { ptamd::set_error("primitive material index out of range"); return PTAMD_ERR_SCENE; }
    if (d->prim_type[p] != PTAMD_PRIM_TRIANGLE && d->prim_type[p] != PTAMD_PRIM_SPHERE) { ptamd::set_error("unknown primitive type"); return PTAMD_ERR_SCENE; }
  }
  if (NP >= (int)PRIM_NONE) { ptamd::set_error("more than 2^24 - 2 primitives are not supported"); return PTAMD_ERR_SCENE; }
  // ---- collapse the binary tree into 4-wide nodes whose children are nodes or
  // single primitives. new_index[p] = position of the caller's primitive p in
  // visiting order (the device's primitive index).
  std::vector<int32_t> new_index(NP);
  for (int i = 0; i < NP; i++) new_index[visit_order[i]] = i;
  // ---- The primitives at the far end of a light's shadow rays. The CB scenes' light is a
  // rectangle that coincides with an emissive quad of the geometry (to within the 5e-8 the
  // quad's float coordinates are off), and the reference's shadow rays end ON the light
  // (max_t = distToLight, pathtracer.cpp:125): whether the quad occludes such a ray is decided
  // by the last bits of an FP64 test, which no float pre-test can settle, for every shadow ray
  // of the frame. So up to four triangles that lie in a light's plane and overlap its rectangle
  // are listed per light (the first seven area or point lights that have any): k_light runs the reference's test on them
  // when it makes a shadow ray, and the ray's walk skips the primitives that carry its light's
  // tag. Which primitives are listed only decides where they are tested, not the result: a
  // shadow ray is occluded iff some primitive's test hits strictly before its end, in any order.
  std::vector<int32_t> light_end(4 * (size_t)std::max(d->n_lights, 1), -1), light_tag(std::max(d->n_lights, 1), 0);
  std::vector<uint8_t> prim_tag(NP, 0);
  int has_end = 0;
  {
    uint32_t next_tag = 1;
    for (int li = 0; li < d->n_lights && next_tag <= MAX_LIGHT_TAG; li++) {
      // (a point light's rays end on the light as well, pathtracer.cpp:125 with light.cpp:50-57, and
      // CBempty.dae puts one in the middle of its emissive quad: a rectangle of no extent, any plane)
      const bool area = d->light_kind[li] == PTAMD_LIGHT_AREA;
      if (!area && d->light_kind[li] != PTAMD_LIGHT_POINT) continue;
      const double *L = d->light_param + 16 * (size_t)li;
      const double *c0 = L + 3, *nrm = L + 6, *dx = L + 9, *dy = L + 12;
      double lo[3], hi[3], scale = 0;
      for (int k = 0; k < 3; k++) {
        const double ex = area ? 0.5 * (std::fabs(dx[k]) + std::fabs(dy[k])) : 0.0;
        lo[k] = c0[k] - ex; hi[k] = c0[k] + ex;
        scale = std::max(scale, std::max(std::fabs(lo[k]), std::fabs(hi[k])));
      }
      const double eps = 1e-5 * (1.0 + scale);
      int found = 0;
      for (int i = 0; i < NP && found < 4; i++) {  // in the device's primitive order
        const int p = visit_order[i];
        if (d->prim_type[p] != PTAMD_PRIM_TRIANGLE || prim_tag[p]) continue;
        const double *P = d->prim_pos + 9 * (size_t)p;
        bool in_plane = true, overlaps = true;
        if (area) {
          for (int v = 0; v < 3 && in_plane; v++) {
            const double h = (P[3 * v] - c0[0]) * nrm[0] + (P[3 * v + 1] - c0[1]) * nrm[1] + (P[3 * v + 2] - c0[2]) * nrm[2];
            in_plane = std::fabs(h) <= eps;
          }
        } else {
          // the point's distance to the triangle's plane
          const double e1[3] = {P[3] - P[0], P[4] - P[1], P[5] - P[2]}, e2[3] = {P[6] - P[0], P[7] - P[1], P[8] - P[2]};
          const double nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
          const double len = std::sqrt(nx * nx + ny * ny + nz * nz);
          in_plane = len > 0 && std::fabs((c0[0] - P[0]) * nx + (c0[1] - P[1]) * ny + (c0[2] - P[2]) * nz) <= eps * len;
        }
        if (!in_plane) continue;
        for (int k = 0; k < 3; k++) {
          const double tlo = std::min(P[k], std::min(P[3 + k], P[6 + k])), thi = std::max(P[k], std::max(P[3 + k], P[6 + k]));
          if (thi < lo[k] - eps || tlo > hi[k] + eps) overlaps = false;
        }
        if (!overlaps) continue;
        light_end[4 * (size_t)li + found++] = i;
        prim_tag[p] = (uint8_t)next_tag;
      }
      if (found) {
        light_tag[li] = (int32_t)next_tag++;
        has_end = 1;
      }
    }
  }
  struct Item { int32_t node; int32_t prim; };  // node >= 0: a node of the caller's tree; else the caller's primitive
  auto prim_box = [&](int p, double bb[6]) {
    const double *P = d->prim_pos + 9 * (size_t)p;
    if (d->prim_type[p] == PTAMD_PRIM_TRIANGLE) {
      for (int k = 0; k < 3; k++) {
        bb[k] = std::min(P[k], std::min(P[3 + k], P[6 + k]));
        bb[3 + k] = std::max(P[k], std::max(P[3 + k], P[6 + k]));
      }
    } else {  // o -/+ Vector3D(r, r, r), src/scene/sphere.h:32-34
      for (int k = 0; k < 3; k++) { bb[k] = P[k] - P[3]; bb[3 + k] = P[k] + P[3]; }
    }
  };
  auto item_box = [&](const Item &it, double bb[6]) {
    if (it.node >= 0) memcpy(bb, d->node_bbox + 6 * (size_t)it.node, 48);
    else prim_box(it.prim, bb);
  };
  auto area = [](const double bb[6]) {
    const double ex = bb[3] - bb[0], ey = bb[4] - bb[1], ez = bb[5] - bb[2];
    return ex * ey + ey * ez + ez * ex;
  };
  auto expand = [&](const Item &it, std::vector<Item> &out) {
    const int n = it.node;
    if (d->node_count[n] > 0) {
      for (int p = d->node_start[n]; p < d->node_start[n] + d->node_count[n]; p++) out.push_back({-1, p});
    } else {
      out.push_back({d->node_left[n], -1});
      out.push_back({d->node_right[n], -1});
    }
  };
  auto expanded_size = [&](const Item &it) { return d->node_count[it.node] > 0 ? d->node_count[it.node] : 2; };
  std::vector<DevNode> nodes;
  nodes.reserve((size_t)NP / 2 + 4);
  int depth4 = 0;
  struct Frame { int32_t node4; int slot; Item item; int depth; };
  std::vector<Frame> work;
  // a single-primitive leaf is just that primitive
  auto normalize = [&](Item it) {
    if (it.node >= 0 && d->node_count[it.node] == 1) return Item{-1, d->node_start[it.node]};
    return it;
  };
  int32_t root_code;
  {
    DevNode rootn;
    memset(&rootn, 0, sizeof rootn);
    for (int k = 0; k < 4; k++) rootn.child[k] = CHILD_EMPTY;
    nodes.push_back(rootn);
    root_code = 0;
    work.push_back({0, -1, normalize(Item{d->root, -1}), 1});
  }
  // `work` entries with slot == -1 fill node `node4` with the children of `item`
  while (!work.empty()) {
    Frame f = work.back();
    work.pop_back();
    depth4 = std::max(depth4, f.depth);
    std::vector<Item> kids;
    if (f.item.node < 0) {
      kids.push_back(f.item);  // the whole scene is one primitive
    } else {
      expand(f.item, kids);
      for (auto &k : kids) k = normalize(k);
      // open the largest openable child while the result still fits four slots
      for (;;) {
        int best = -1;
        double best_area = -1;
        for (size_t k = 0; k < kids.size(); k++) {
          if (kids[k].node < 0) continue;
          if ((int)kids.size() - 1 + expanded_size(kids[k]) > 4) continue;
          double bb[6];
          item_box(kids[k], bb);
          const double a = area(bb);
          if (a > best_area) { best_area = a; best = (int)k; }
        }
        if (best < 0) break;
        const Item it = kids[best];
        kids.erase(kids.begin() + best);
        std::vector<Item> sub;
        expand(it, sub);
        for (auto &k : sub) kids.push_back(normalize(k));
      }
    }
    double boxes[4][6];
    for (size_t k = 0; k < kids.size(); k++) item_box(kids[k], boxes[k]);
    if (!quantize_node(nodes[f.node4], boxes, (int)kids.size())) {
      ptamd::set_error("BVH node box cannot be represented (non-finite or out of float range)");
      return PTAMD_ERR_SCENE;
    }
    {
      // the light tag of the node's tagged primitive children, if they agree on one
      uint32_t ntag = 0;
      bool agree = true;
      for (size_t k = 0; k < kids.size(); k++)
        if (kids[k].node < 0 && prim_tag[kids[k].prim]) {
          if (ntag && ntag != prim_tag[kids[k].prim]) agree = false;
          ntag = prim_tag[kids[k].prim];
        }
      nodes[f.node4].exps = (nodes[f.node4].exps & 0x00ffffffu) | ((agree ? ntag : 0u) << 24);
    }
    for (size_t k = 0; k < kids.size(); k++) {
      if (kids[k].node < 0) {
        nodes[f.node4].child[k] = (int32_t)~((uint32_t)new_index[kids[k].prim] | ((uint32_t)prim_tag[kids[k].prim] << PRIM_BITS));
      } else {
        DevNode nn;
        memset(&nn, 0, sizeof nn);
        for (int j = 0; j < 4; j++) nn.child[j] = CHILD_EMPTY;
        const int32_t id = (int32_t)nodes.size();
        nodes.push_back(nn);
        nodes[f.node4].child[k] = id;
        work.push_back({id, -1, kids[k], f.depth + 1});
      }
    }
  }
  // deepest stack: three siblings pushed per level
  max_depth = 3 * depth4;
  // ---- primitive records in visiting order
  std::vector<double> prim((size_t)NP * PRIM_STRIDE, 0.0), nrm((size_t)NP * 9);
  std::vector<int32_t> info(NP);
  std::vector<Tree4::Word4> prim32((size_t)NP * 3);
  bool coords_ok = true;
  double max_coord = 0;
  for (int i = 0; i < NP; i++) {
    const int p = visit_order[i];
    const double *P = d->prim_pos + 9 * (size_t)p;
    double *R = prim.data() + (size_t)i * PRIM_STRIDE;
    if (d->prim_type[p] == PTAMD_PRIM_TRIANGLE) {
      // e1 = p2 - p1, e2 = p3 - p1: one rounding each, identical to evaluating
      // them per ray
      R[0] = P[0]; R[1] = P[1]; R[2] = P[2];
      R[3] = P[3] - P[0]; R[4] = P[4] - P[1]; R[5] = P[5] - P[2];
      R[6] = P[6] - P[0]; R[7] = P[7] - P[1]; R[8] = P[8] - P[2];
      info[i] = d->prim_material[p] << 1;
    } else {
      R[0] = P[0]; R[1] = P[1]; R[2] = P[2]; R[3] = P[4];
      info[i] = (d->prim_material[p] << 1) | 1;
    }
    memcpy(R + 9, &info[i], 4);
    memcpy(nrm.data() + (size_t)i * 9, d->prim_nrm + 9 * (size_t)p, 72);
    // the FP32 pre-test's record
    unsigned char *Q = reinterpret_cast<unsigned char *>(prim32.data() + (size_t)i * 3);
    memcpy(Q, R, 24);
    float ef[6];
    if (d->prim_type[p] == PTAMD_PRIM_TRIANGLE) {
      for (int k = 0; k < 6; k++) ef[k] = (float)R[3 + k];
    } else {
      ef[0] = NAN;
      ef[1] = (float)R[3];
      for (int k = 2; k < 6; k++) ef[k] = 0.f;
    }
    memcpy(Q + 24, ef, 24);
    // prim32.h assumes coordinates (and so edge lengths) well inside the float range
    if (d->prim_type[p] == PTAMD_PRIM_TRIANGLE) {
      for (int k = 0; k < 9; k++) max_coord = std::max(max_coord, std::fabs(P[k]));
    } else {
      for (int k = 0; k < 3; k++) max_coord = std::max(max_coord, std::fabs(P[k]) + std::fabs(P[3]));
    }
  }
  coords_ok = max_coord < 1048576.0;  // also false for NaN
  if (!coords_ok) {
    ptamd::set_error("scene coordinates of 2^20 or more are not supported");
    return PTAMD_ERR_SCENE;
  }
  out.nodes = std::move(nodes);
  out.root_code = root_code;
  out.stack_cap = max_depth + 1;
  out.visit_order = std::move(visit_order);
  out.prim = std::move(prim);
  out.nrm = std::move(nrm);
  out.info = std::move(info);
  out.prim32 = std::move(prim32);
  out.light_end = std::move(light_end);
  out.light_tag = std::move(light_tag);
  out.has_end = has_end;
  out.max_coord = max_coord;
  return PTAMD_OK;
}

}  // namespace ptamd
