// How many node visits and primitive tests would a wider tree save? Host-only experiment.
//
//   g++ -std=c++17 -O2 -ffp-contract=off -I pathtracer_amd/csrc -o /tmp/wide_tree_counts tools/wide_tree_counts.cpp \
//       pathtracer_amd/csrc/{tree4,scene,collada_loader,xml_lite,capi_host,renderer}.cpp -lz -Wl,--unresolved-symbols=ignore-all
//   /tmp/wide_tree_counts scene.dae [w h]
//
// Collapses the scene's binary SAH tree (the reference's, src/scene/bvh.cpp:47-135) into W-wide nodes the
// way the device layer does for W = 4 (a node's children are opened, largest box first, while the result
// fits W slots; every primitive a child of its own), for W = 2, 4, 8 and 16, and walks each tree with the
// same rays: the camera rays of a w x h frame, one cosine-weighted bounce ray per hit, and four shadow
// rays from every hit towards the scene's first area light. Closest-hit rays take a node's accepted
// children nearest first and drop stacked entries that begin beyond the current hit; shadow rays take
// them in node order and stop at the first hit: the device walk's two policies. Boxes and primitive tests
// are FP64 here (the counts do not depend on the device's conservative FP32 forms to first order).
// Printed per width and ray kind: node visits, child boxes tested, primitive tests, and "trips" of the
// device schedule for one lane (two node steps then one primitive step per trip). Two more pairs of rows: 4- and
// 8-wide with the binary tree's leaves kept whole (one child, its primitives tested in one step), and 4- and 8-wide
// collapsed by a dynamic programme for the least summed node area instead of the device layer's greedy rule.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../include/ptamd.h"

struct V3 { double x, y, z; };
static V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static V3 cross(V3 u, V3 v) { return {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x}; }
static double dot(V3 u, V3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
static V3 unit(V3 a) { return a * (1.0 / std::sqrt(dot(a, a))); }

struct WNode { std::vector<int> child; std::vector<std::array<double, 6>> box; };  // child >= 0: node, else ~primitive
struct WTree { std::vector<WNode> nodes; };

static void prim_box(const ptamd_scene_desc &d, int p, double bb[6]) {
  const double *P = d.prim_pos + 9 * (size_t)p;
  if (d.prim_type[p] == PTAMD_PRIM_TRIANGLE)
    for (int k = 0; k < 3; k++) { bb[k] = std::min(P[k], std::min(P[3 + k], P[6 + k])); bb[3 + k] = std::max(P[k], std::max(P[3 + k], P[6 + k])); }
  else
    for (int k = 0; k < 3; k++) { bb[k] = P[k] - P[3]; bb[3 + k] = P[k] + P[3]; }
}

// leaves = true: a leaf of the binary tree (up to four primitives, src/scene/bvh.h:52) stays ONE child with the leaf's box, code ~(leaf node);
// its primitives are then tested in one step, without boxes of their own
static WTree collapse(const ptamd_scene_desc &d, int W, bool leaves = false) {
  struct Item { int node, prim; };
  auto norm = [&](Item it) {
    if (leaves) return (it.node >= 0 && d.node_count[it.node] > 0) ? Item{-1, it.node} : it;  // prim = the leaf NODE
    return (it.node >= 0 && d.node_count[it.node] == 1) ? Item{-1, d.node_start[it.node]} : it;
  };
  auto box_of = [&](Item it, double bb[6]) {
    if (it.node >= 0 || leaves) for (int k = 0; k < 6; k++) bb[k] = d.node_bbox[6 * (size_t)(it.node >= 0 ? it.node : it.prim) + k];
    else prim_box(d, it.prim, bb);
  };
  auto area = [](const double bb[6]) { const double ex = bb[3] - bb[0], ey = bb[4] - bb[1], ez = bb[5] - bb[2]; return ex * ey + ey * ez + ez * ex; };
  auto expand = [&](Item it, std::vector<Item> &out) {
    if (d.node_count[it.node] > 0) for (int p = d.node_start[it.node]; p < d.node_start[it.node] + d.node_count[it.node]; p++) out.push_back({-1, p});
    else { out.push_back({d.node_left[it.node], -1}); out.push_back({d.node_right[it.node], -1}); }
  };
  auto esize = [&](Item it) { return d.node_count[it.node] > 0 ? d.node_count[it.node] : 2; };
  WTree T;
  T.nodes.push_back({});
  std::vector<std::pair<int, Item>> work{{0, norm({d.root, -1})}};
  while (!work.empty()) {
    auto [id, item] = work.back();
    work.pop_back();
    std::vector<Item> kids;
    if (item.node < 0) kids.push_back(item);
    else {
      expand(item, kids);
      for (auto &k : kids) k = norm(k);
      for (;;) {
        int best = -1; double ba = -1;
        for (size_t k = 0; k < kids.size(); k++) {
          if (kids[k].node < 0 || (int)kids.size() - 1 + esize(kids[k]) > W) continue;
          double bb[6]; box_of(kids[k], bb);
          if (area(bb) > ba) { ba = area(bb); best = (int)k; }
        }
        if (best < 0) break;
        const Item it = kids[best];
        kids.erase(kids.begin() + best);
        std::vector<Item> sub; expand(it, sub);
        for (auto &k : sub) kids.push_back(norm(k));
      }
    }
    for (auto &k : kids) {
      std::array<double, 6> bb; box_of(k, bb.data());
      T.nodes[id].box.push_back(bb);
      if (k.node < 0) T.nodes[id].child.push_back(~k.prim);
      else { T.nodes[id].child.push_back((int)T.nodes.size()); work.push_back({(int)T.nodes.size(), k}); T.nodes.push_back({}); }
    }
  }
  return T;
}

// The collapse that minimises the summed surface area of the wide nodes (= the expected number of node visits of a
// random ray under the surface-area heuristic): a dynamic programme over the binary tree. f(n, i) = least summed area of
// the wide nodes made inside subtree n when n may take up to i child slots of the wide node above it: either n becomes a
// wide node itself (one slot; its own W slots are split between its two children) or, with i >= 2, it dissolves and
// its children share the i slots. A leaf of m primitives is a little subtree of single primitives (each its own child).
struct OptCollapse {
  const ptamd_scene_desc &d; int W;
  // virtual binary tree: ids < N are the scene's nodes; leaves with m > 1 primitives get m - 1 virtual inner nodes
  struct VN { int l, r, prim; double area; };  // prim >= 0: a single primitive
  std::vector<VN> v; std::vector<std::vector<double>> f; std::vector<std::vector<signed char>> choice;  // choice[n][i]: 0 = node, k > 0 = dissolve, left gets k
  std::vector<std::vector<signed char>> split_node;  // as a wide node: left gets k of W
  static double area6(const double *bb) { const double ex = bb[3] - bb[0], ey = bb[4] - bb[1], ez = bb[5] - bb[2]; return ex * ey + ey * ez + ez * ex; }
  int build(int n) {  // returns the virtual id of scene node n
    if (d.node_count[n] > 0) {
      int cur = -1;
      for (int p = d.node_start[n]; p < d.node_start[n] + d.node_count[n]; p++) {
        v.push_back({-1, -1, p, 0.0}); const int leaf = (int)v.size() - 1;
        if (cur < 0) cur = leaf;
        else {
          // a left-deep chain over the leaf's primitives; boxes of the partial unions
          double bb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
          for (int q = d.node_start[n]; q <= p; q++) { double pb[6]; prim_box(d, q, pb); for (int k = 0; k < 3; k++) { bb[k] = std::min(bb[k], pb[k]); bb[3 + k] = std::max(bb[3 + k], pb[3 + k]); } }
          v.push_back({cur, leaf, -1, area6(bb)}); cur = (int)v.size() - 1;
        }
      }
      return cur;
    }
    const int l = build(d.node_left[n]), r = build(d.node_right[n]);
    v.push_back({l, r, -1, area6(d.node_bbox + 6 * (size_t)n)});
    return (int)v.size() - 1;
  }
  void solve(int n) {
    f[n].assign(W + 1, 0.0); choice[n].assign(W + 1, 0); split_node[n].assign(1, 0);
    if (v[n].prim >= 0) return;
    solve(v[n].l); solve(v[n].r);
    auto g = [&](int j, int &bestk) { double best = INFINITY; bestk = 1; for (int k = 1; k < j; k++) { const double c = f[v[n].l][k] + f[v[n].r][j - k]; if (c < best) { best = c; bestk = k; } } return best; };
    int kn; const double as_node = v[n].area + g(W, kn); split_node[n][0] = (signed char)kn;
    f[n][1] = as_node; choice[n][1] = 0;
    for (int i = 2; i <= W; i++) {
      int kd; const double dis = g(i, kd);
      f[n][i] = as_node; choice[n][i] = 0;
      if (dis < f[n][i]) { f[n][i] = dis; choice[n][i] = (signed char)kd; }
      if (f[n][i - 1] < f[n][i]) { f[n][i] = f[n][i - 1]; choice[n][i] = choice[n][i - 1]; if (choice[n][i - 1] == 0) choice[n][i] = 0; else choice[n][i] = -(signed char)(i - 1); }
    }
  }
  // the child slots subtree n presents when given i slots
  void emit(int n, int i, std::vector<int> &slots) {
    if (v[n].prim >= 0) { slots.push_back(~v[n].prim); return; }
    int c = choice[n][i];
    while (c < 0) { i = -c; c = choice[n][i]; }  // "use only i - 1 slots"
    if (c == 0) { slots.push_back(n); return; }  // a wide node of its own
    emit(v[n].l, c, slots); emit(v[n].r, i - c, slots);
  }
  WTree run() {
    v.reserve(4 * (size_t)d.n_nodes);
    const int root = build(d.root);
    f.resize(v.size()); choice.resize(v.size()); split_node.resize(v.size());
    solve(root);
    WTree T; T.nodes.push_back({});
    std::vector<std::pair<int, int>> work{{0, root}};
    while (!work.empty()) {
      auto [id, n] = work.back(); work.pop_back();
      std::vector<int> slots;
      if (v[n].prim >= 0) slots.push_back(~v[n].prim);
      else { const int k = split_node[n][0]; emit(v[n].l, k, slots); emit(v[n].r, W - k, slots); }
      for (int sc : slots) {
        std::array<double, 6> bb;
        if (sc < 0) prim_box(d, ~sc, bb.data());
        else {
          // box of virtual node sc: union of the primitives below it
          for (int k = 0; k < 3; k++) { bb[k] = INFINITY; bb[3 + k] = -INFINITY; }
          std::vector<int> st{sc};
          while (!st.empty()) { const int x = st.back(); st.pop_back(); if (v[x].prim >= 0) { double pb[6]; prim_box(d, v[x].prim, pb); for (int k = 0; k < 3; k++) { bb[k] = std::min(bb[k], pb[k]); bb[3 + k] = std::max(bb[3 + k], pb[3 + k]); } } else { st.push_back(v[x].l); st.push_back(v[x].r); } }
        }
        T.nodes[id].box.push_back(bb);
        if (sc < 0) T.nodes[id].child.push_back(sc);
        else { T.nodes[id].child.push_back((int)T.nodes.size()); work.push_back({(int)T.nodes.size(), sc}); T.nodes.push_back({}); }
      }
    }
    return T;
  }
};

static bool box_hit(const std::array<double, 6> &b, V3 o, V3 inv, double tmin, double tmax, double &tn) {
  double t0 = tmin, t1 = tmax;
  const double oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
  for (int a = 0; a < 3; a++) {
    double ta = (b[a] - oo[a]) * ii[a], tb = (b[3 + a] - oo[a]) * ii[a];
    if (ta > tb) std::swap(ta, tb);
    if (ta > t0) t0 = ta;  // NaN (0 * inf) leaves the bounds alone
    if (tb < t1) t1 = tb;
  }
  tn = t0;
  return t0 <= t1;
}
static bool prim_hit(const ptamd_scene_desc &d, int p, V3 o, V3 dir, double tmin, double tmax, double &t) {
  const double *P = d.prim_pos + 9 * (size_t)p;
  if (d.prim_type[p] == PTAMD_PRIM_TRIANGLE) {
    const V3 p1{P[0], P[1], P[2]}, e1{P[3] - P[0], P[4] - P[1], P[5] - P[2]}, e2{P[6] - P[0], P[7] - P[1], P[8] - P[2]};
    const V3 s = o - p1, s1 = cross(dir, e2), s2 = cross(s, e1);
    const double D = dot(s1, e1);
    if (D == 0) return false;
    const double r = 1 / D, u = dot(s1, s) * r, v = dot(s2, dir) * r;
    t = dot(s2, e2) * r;
    return !(t < tmin || t > tmax || u < 0 || v < 0 || u + v > 1);
  }
  const V3 om = o - V3{P[0], P[1], P[2]};
  const double a = dot(dir, dir), b = 2 * dot(om, dir), c = dot(om, om) - P[4], q = b * b - 4 * a * c;
  if (q < 0) return false;
  const double t1 = (-b - std::sqrt(q)) / (2 * a), t2 = (-b + std::sqrt(q)) / (2 * a);
  if (t1 >= tmin && t1 <= tmax) { t = t1; return true; }
  if (t2 >= tmin && t2 <= tmax) { t = t2; return true; }
  return false;
}

struct Counts { double rays = 0, visits = 0, boxes = 0, tests = 0, trips = 0, leaf_steps = 0; };
// one lane of the device schedule: a trip = up to two node steps, then a primitive step if the lane is at a primitive
struct TripMeter {
  int slot = 0; long trips = 0;  // slot: 0, 1 = node slots, 2 = primitive slot of the running trip
  void node() { if (slot >= 2) { trips++; slot = 0; } if (slot == 0 && trips == 0) trips = 1; slot++; }
  void prim() { if (trips == 0) trips = 1; if (slot > 2) { trips++; } slot = 3; }
};

static int trace(const ptamd_scene_desc &d, const WTree &T, V3 o, V3 dir, double tmin, double tmax, bool first_hit, Counts &C, int skip_prim, bool leaves = false) {
  const V3 inv{1 / dir.x, 1 / dir.y, 1 / dir.z};
  struct E { int code; double t; };
  std::vector<E> st{{0, 0.0}};
  double best = tmax; int hit = -1;
  TripMeter tm;
  C.rays++;
  while (!st.empty()) {
    const E e = st.back(); st.pop_back();
    if (e.t > best) continue;
    if (e.code < 0 && leaves) {
      const int leaf = ~e.code;
      C.leaf_steps++; tm.prim();
      bool done = false;
      for (int p = d.node_start[leaf]; p < d.node_start[leaf] + d.node_count[leaf]; p++) {
        if (p == skip_prim) continue;
        C.tests++;
        double t;
        if (prim_hit(d, p, o, dir, tmin, best, t) && (t < best || hit < 0)) { best = t; hit = p; if (first_hit) { done = true; break; } }
      }
      if (done) break;
      continue;
    }
    if (e.code < 0) {
      const int p = ~e.code;
      if (p == skip_prim) continue;
      C.tests++; tm.prim();
      double t;
      if (prim_hit(d, p, o, dir, tmin, best, t) && (t < best || hit < 0)) { best = t; hit = p; if (first_hit) break; }
      continue;
    }
    C.visits++; tm.node();
    const WNode &n = T.nodes[e.code];
    std::vector<E> acc;
    for (size_t k = 0; k < n.child.size(); k++) {
      C.boxes++;
      double tn;
      if (box_hit(n.box[k], o, inv, tmin, best, tn)) acc.push_back({n.child[k], tn});
    }
    if (!first_hit) std::stable_sort(acc.begin(), acc.end(), [](const E &a, const E &b) { return a.t < b.t; });
    for (size_t k = acc.size(); k-- > 0;) st.push_back(acc[k]);  // the first accepted (or nearest) child on top
  }
  C.trips += tm.trips;
  return hit;
}

int main(int argc, char **argv) {
  if (argc < 2) { printf("usage: wide_tree_counts scene.dae [w h]\n"); return 1; }
  const int w = argc > 3 ? atoi(argv[2]) : 160, h = argc > 3 ? atoi(argv[3]) : 120;
  ptamd_scene *sc = nullptr;
  if (ptamd_scene_load(argv[1], &sc)) { printf("%s\n", ptamd_last_error()); return 1; }
  ptamd_scene_desc d; ptamd_scene_get_desc(sc, &d);
  ptamd_camera cam; ptamd_scene_get_camera(sc, 800, 600, &cam);
  int la = -1;
  for (int li = 0; li < d.n_lights; li++) if (d.light_kind[li] == PTAMD_LIGHT_AREA) { la = li; break; }
  printf("%s: %d primitives, %d binary nodes\n", argv[1], d.n_prims, d.n_nodes);
  for (int mode = 0; mode < 8; mode++) {
    const int W = mode < 4 ? (2 << mode) : (mode == 4 || mode == 6 ? 4 : 8);
    const bool leaves = mode == 4 || mode == 5, optimal = mode >= 6;
    const WTree T = optimal ? OptCollapse{d, W}.run() : collapse(d, W, leaves);
    size_t children = 0, prim_children = 0;
    for (auto &n : T.nodes) { children += n.child.size(); for (int c : n.child) prim_children += c < 0; }
    Counts cam_c, bounce_c, shadow_c;
    // shadow rays that the occluder of the shadow ray before them (in generation order) also occludes
    double sc_rays = 0, sc_occluded = 0, sc_cache_hit = 0, sc_trips_all = 0, sc_trips_hit = 0; int last_occ = -1;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(0, 1);
    const double th = std::tan(cam.hFov * M_PI / 360), tv = std::tan(cam.vFov * M_PI / 360);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        const double sx = (x + 0.5) / w, sy = (y + 0.5) / h;
        const double s[3] = {(sx - 0.5) * 2 * th, (sy - 0.5) * 2 * tv, -1};
        V3 dir{0, 0, 0};
        dir.x = cam.c2w[0] * s[0] + cam.c2w[3] * s[1] + cam.c2w[6] * s[2];
        dir.y = cam.c2w[1] * s[0] + cam.c2w[4] * s[1] + cam.c2w[7] * s[2];
        dir.z = cam.c2w[2] * s[0] + cam.c2w[5] * s[1] + cam.c2w[8] * s[2];
        dir = unit(dir);
        V3 o{cam.pos[0], cam.pos[1], cam.pos[2]};
        for (int depth = 0; depth < 3; depth++) {
          Counts &C = depth == 0 ? cam_c : bounce_c;
          // (a ray leaving a surface skips the primitive it starts on, as the device walk does)
          static int last = -1;
          if (depth == 0) last = -1;
          Counts tmp;
          const int p = trace(d, T, o, dir, depth == 0 ? cam.nClip : 1e-11, depth == 0 ? cam.fClip : INFINITY, false, tmp, last, leaves);
          C.rays += tmp.rays; C.visits += tmp.visits; C.boxes += tmp.boxes; C.tests += tmp.tests; C.trips += tmp.trips; C.leaf_steps += tmp.leaf_steps;
          if (p < 0) break;
          // the hit point and a geometric normal facing the ray
          double t; prim_hit(d, p, o, dir, depth == 0 ? cam.nClip : 1e-11, INFINITY, t);
          const V3 hp = o + dir * t;
          const double *P = d.prim_pos + 9 * (size_t)p;
          V3 n = d.prim_type[p] == PTAMD_PRIM_TRIANGLE ? unit(cross(V3{P[3] - P[0], P[4] - P[1], P[5] - P[2]}, V3{P[6] - P[0], P[7] - P[1], P[8] - P[2]}))
                                                       : unit(hp - V3{P[0], P[1], P[2]});
          if (dot(n, dir) > 0) n = n * -1.0;
          if (la >= 0) {
            const double *L = d.light_param + 16 * (size_t)la;
            for (int q = 0; q < 4; q++) {
              const double ax = U(rng) - 0.5, ay = U(rng) - 0.5;
              const V3 S{L[3] + L[9] * ax + L[12] * ay, L[4] + L[10] * ax + L[13] * ay, L[5] + L[11] * ax + L[14] * ay};
              const V3 dd = S - hp; const double dist = std::sqrt(dot(dd, dd));
              if (dot(dd, n) <= 0) continue;
              Counts one;
              const V3 sd = dd * (1 / dist);
              const int occ = trace(d, T, hp, sd, 1e-11, dist * (1 - 1e-6), true, one, p, leaves);
              shadow_c.rays += one.rays; shadow_c.visits += one.visits; shadow_c.boxes += one.boxes; shadow_c.tests += one.tests;
              shadow_c.trips += one.trips; shadow_c.leaf_steps += one.leaf_steps;
              sc_rays++; sc_trips_all += one.trips;
              if (occ >= 0) sc_occluded++;
              double tc;
              if (last_occ >= 0 && last_occ != p && prim_hit(d, last_occ, hp, sd, 1e-11, dist * (1 - 1e-6), tc)) { sc_cache_hit++; sc_trips_hit += one.trips; }
              if (occ >= 0) last_occ = occ;
            }
          }
          // cosine-weighted bounce
          const double r1 = U(rng), r2 = U(rng), rr = std::sqrt(r1), ph = 2 * M_PI * r2;
          V3 hlp = std::fabs(n.x) < 0.5 ? V3{1, 0, 0} : V3{0, 1, 0};
          const V3 bx = unit(cross(hlp, n)), by = cross(n, bx);
          dir = unit(bx * (rr * std::cos(ph)) + by * (rr * std::sin(ph)) + n * std::sqrt(1 - r1));
          o = hp; last = p;
        }
      }
    auto pr = [&](const char *k, const Counts &C) {
      printf("  W=%2d %-7s rays %8.0f  node visits %6.2f  boxes %6.2f  primitive tests %5.2f  lane trips %5.2f", W, k, C.rays, C.visits / C.rays,
             C.boxes / C.rays, C.tests / C.rays, C.trips / C.rays);
      if (leaves) printf("  leaf steps %5.2f", C.leaf_steps / C.rays);
      printf("\n");
    };
    printf(" W=%d%s%s: %zu nodes, %.2f children per node (%.2f of them %s)\n", W, leaves ? " with whole leaves" : "", optimal ? " collapsed for the least summed node area" : "", T.nodes.size(),
           (double)children / T.nodes.size(), (double)prim_children / T.nodes.size(), leaves ? "leaves" : "primitives");
    pr("camera", cam_c); pr("bounce", bounce_c); pr("shadow", shadow_c);
    if (mode == 1)
      printf("  W= 4 shadow rays occluded: %.1f %%; occluded by the occluder of the shadow ray before them: %.1f %% of all, holding %.1f %% of the shadow rays' trips\n",
             100 * sc_occluded / sc_rays, 100 * sc_cache_hit / sc_rays, 100 * sc_trips_hit / sc_trips_all);
  }
  ptamd_scene_free(sc);
  return 0;
}
