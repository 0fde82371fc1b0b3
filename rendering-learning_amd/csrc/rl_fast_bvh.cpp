// Host side of the wave kernel's FAST traversal (rl_rtiow_wave.h, LDS_SCENE = 4): the frame in which reject-only boxes are
// rigorous, and a surface-area-heuristic binary tree over the spheres.
//
// Why a second structure is legal.  The reference folds over its own BVH (bvh.rs:79-95) in stored order with a shrinking
// ray_t.max (hittable/mod.rs:88-111).  For one ray the RESULT of that fold is (t*, sphere*) with t* = the smallest accepted root
// over the spheres the fold reaches; which spheres it reaches BEFORE that only changes the counters.  Any traversal that
//   (1) never skips a sphere whose Sphere::hit (sphere.rs:32-75) could return a root <= the current closest   [boxes only reject,
//       and they reject with the margins below], and
//   (2) notices every situation in which the ORDER could matter — two roots closer than 1e-7 relative (the reference resolves exact
//       ties by stored order, and prunes by `tmin < closest` on ITS boxes), a grazing hit (chord below 1e-6 relative: the
//       reference's own unpadded leaf box may or may not be passed), a ray outside the binary32 filter's range, a sphere normal that
//       trips the reference's assert —
// and in those situations re-traces the ray with the reference's own fold, returns the reference's (t*, sphere*) for every ray.
// tests/test_gpu_timed_kernels.py compares the two kernels bit for bit; bench.py does so on the full frame after every run.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <numeric>

#include "rl_program.h"

namespace rl {

// A box may only reject a ray that Sphere::hit would ALSO report as a miss, and Sphere::hit decides on the sign of the ROUNDED
// discriminant half_b^2 - a*c.  For a ray whose line passes the centre at distance b the exact value is a*(r^2 - b^2); the rounded
// one differs by at most ~16u*a*|oc|^2 (products and sums of the dot products) + 4u*a*M*|oc| (the rounding of oc = o - centre; M =
// largest coordinate involved), u = 2^-53.  A ray that misses [c - r - pad, c + r + pad] has b >= r + pad, i.e. r^2 - b^2 <= -2*r*pad, so
//     pad >= (8u*L^2 + 2u*M*L) / r        (L >= |oc|)
// makes the rounded discriminant negative as well; guard_pad returns 8x that (and never less than 1e-9 relative).
// L and M follow from a FRAME: every ray origin is the camera (checked against `reach` per render) or a hit point on a sphere.
GuardFrame guard_frame(const rl_rtiow_scene_desc &d) {
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, rmin = INFINITY;
  bool finite = true;
  for (uint32_t i = 0; i < d.n_spheres; i++) {
    const rl_sphere &sp = d.spheres[i];
    double r = std::fabs(sp.radius);
    finite = finite && std::isfinite(sp.radius);
    rmin = std::fmin(rmin, r);
    for (int ax = 0; ax < 3; ax++) {
      double c0 = sp.center0[ax], c1 = sp.moving ? sp.center1[ax] : sp.center0[ax];
      finite = finite && std::isfinite(c0) && std::isfinite(c1);
      lo[ax] = std::fmin(lo[ax], std::fmin(c0, c1) - r), hi[ax] = std::fmax(hi[ax], std::fmax(c0, c1) + r);
    }
  }
  GuardFrame f{};
  double diag2 = 0.0, cmax = 0.0;
  for (int ax = 0; ax < 3; ax++) {
    if (!(lo[ax] <= hi[ax])) lo[ax] = hi[ax] = 0.0;  // no spheres (or NaN): an empty frame
    f.center[ax] = 0.5 * (lo[ax] + hi[ax]);
    diag2 += (hi[ax] - lo[ax]) * (hi[ax] - lo[ax]);
    cmax = std::fmax(cmax, std::fmax(std::fabs(lo[ax]), std::fabs(hi[ax])));
  }
  f.half = 0.5 * std::sqrt(diag2);
  f.reach = 2.0 * f.half + 1.0;  // a camera farther away than a scene diameter renders through the unguarded ops
  f.L = f.reach + f.half;        // origin within reach of the centre (camera) or inside the box (hit points), sphere centre inside the box
  f.M = cmax + f.reach;
  // Sphere::hit's outward normal (p - c) * (1/r) misses unit length by at most ~28u (|oc|/r)^2 + 4u (M + |oc|)/r + 8u (residual of
  // the rounded root, rounding of p = o + t d): with twice that below 1e-5 / 2 the assert at vec3.rs:219-222 (|n|^2 within 1e-5 of 1)
  // cannot fire for any sphere the reference accepts along the way
  const double u = 1.1102230246251565e-16;
  f.normals_safe = finite && d.n_spheres > 0 && rmin > 0.0 && 64.0 * u * (f.L / rmin) * (f.L / rmin) + 8.0 * u * (f.M + f.L) / rmin <= 5e-6;
  return f;
}

double guard_pad(const GuardFrame &f, double r) {
  const double u = 1.1102230246251565e-16;
  return 8.0 * (8.0 * u * f.L * f.L + 2.0 * u * f.M * f.L) / r;  // r = 0: inf (the caller turns a non-finite box into "never rejects")
}

namespace {
float round_down(double v) {
  float f = (float)v;
  return (double)f > v ? std::nextafterf(f, -INFINITY) : f;
}
float round_up(double v) {
  float f = (float)v;
  return (double)f < v ? std::nextafterf(f, INFINITY) : f;
}
struct Box {
  double lo[3], hi[3];
  void grow(const Box &b) {
    for (int k = 0; k < 3; k++) lo[k] = std::fmin(lo[k], b.lo[k]), hi[k] = std::fmax(hi[k], b.hi[k]);
  }
  double area() const {
    double e0 = hi[0] - lo[0], e1 = hi[1] - lo[1], e2 = hi[2] - lo[2];
    return 2.0 * (e0 * e1 + e1 * e2 + e0 * e2);
  }
};
const Box EMPTY{{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
}  // namespace

bool build_fast_bvh(const rl_rtiow_scene_desc &d, const RtiowProgram &rt, const GuardFrame &f, std::vector<FastNode> &nodes, uint32_t &root_entry) {
  nodes.clear();
  const uint32_t n = d.n_spheres;
  if (n == 0 || n > 511 || !f.normals_safe) return false;  // entry ids: (n - 1) inner nodes + n spheres < FAST_NONE
  if (rt.has_planars || rt.has_instances || rt.has_images || rt.has_noise) return false;
  // every sphere must be part of the world exactly once (the tree is built over ALL spheres of the descriptor)
  std::vector<uint8_t> seen(n, 0);
  for (const DevOp &op : rt.ops) {
    uint32_t kind = op.code & 0xFFu;
    if (kind != OP_BOX_SPH && kind != OP_SPHERE) continue;
    for (uint32_t payload : {op.a, op.b}) {
      if (payload == NONE) continue;
      uint32_t s = payload & SPH_INDEX;
      if (s >= n || seen[s]) return false;
      seen[s] = 1;
    }
  }
  for (uint8_t v : seen)
    if (!v) return false;
  std::vector<Box> boxes(n);
  for (uint32_t i = 0; i < n; i++) {
    const rl_sphere &sp = d.spheres[i];
    double r = std::fabs(sp.radius), pad0 = guard_pad(f, r);
    for (int ax = 0; ax < 3; ax++) {
      double c0 = sp.center0[ax], c1 = sp.moving ? sp.center1[ax] : sp.center0[ax];
      double lo = std::fmin(c0, c1) - r, hi = std::fmax(c0, c1) + r;
      double pad = std::fmax(1e-9 * (std::fabs(lo) + std::fabs(hi) + r), pad0);
      boxes[i].lo[ax] = lo - pad, boxes[i].hi[ax] = hi + pad;
      if (!(std::fabs(boxes[i].lo[ax]) <= 1e30 && std::fabs(boxes[i].hi[ax]) <= 1e30)) return false;
    }
  }
  const uint32_t n_inner = n - 1;
  nodes.assign(n_inner, FastNode{});
  uint32_t next = 0;
  auto put = [&](FastNode &nd, int side, const Box &b) {
    for (int ax = 0; ax < 3; ax++) nd.box[side][2 * ax] = round_down(b.lo[ax]), nd.box[side][2 * ax + 1] = round_up(b.hi[ax]);
  };
  // returns the entry id of the subtree over ids[lo, hi) and its box
  std::vector<uint32_t> ids(n);
  std::iota(ids.begin(), ids.end(), 0u);
  std::function<uint32_t(uint32_t, uint32_t, uint32_t, Box &)> build = [&](uint32_t lo, uint32_t hi, uint32_t budget, Box &out) -> uint32_t {
    const uint32_t m = hi - lo;
    if (m == 1) {
      out = boxes[ids[lo]];
      return n_inner + ids[lo];
    }
    const uint32_t self = next++;
    uint32_t need = 0;  // depth a balanced tree over m leaves needs
    while ((1u << need) < m) need++;
    uint32_t split = lo + m / 2;
    int best_axis = 0;
    if (need >= budget) {  // no slack left for the heuristic: median split along the longest axis keeps the depth cap
      Box all = EMPTY;
      for (uint32_t i = lo; i < hi; i++) all.grow(boxes[ids[i]]);
      double e[3] = {all.hi[0] - all.lo[0], all.hi[1] - all.lo[1], all.hi[2] - all.lo[2]};
      best_axis = e[0] >= e[1] ? (e[0] >= e[2] ? 0 : 2) : (e[1] >= e[2] ? 1 : 2);
      std::stable_sort(ids.begin() + lo, ids.begin() + hi,
                       [&](uint32_t a, uint32_t b) { return boxes[a].lo[best_axis] + boxes[a].hi[best_axis] < boxes[b].lo[best_axis] + boxes[b].hi[best_axis]; });
    } else {
      double best = INFINITY;
      std::vector<uint32_t> order(m), best_order;
      std::vector<double> left(m);
      const uint32_t cap = 1u << (budget - 1);  // either side must still fit the remaining depth
      for (int ax = 0; ax < 3; ax++) {
        std::copy(ids.begin() + lo, ids.begin() + hi, order.begin());
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return boxes[a].lo[ax] + boxes[a].hi[ax] < boxes[b].lo[ax] + boxes[b].hi[ax]; });
        Box acc = EMPTY;
        for (uint32_t i = 0; i + 1 < m; i++) acc.grow(boxes[order[i]]), left[i] = acc.area() * (double)(i + 1);
        acc = EMPTY;
        for (uint32_t i = m - 1; i >= 1; i--) {
          acc.grow(boxes[order[i]]);
          if (i > cap || m - i > cap) continue;
          double c = left[i - 1] + acc.area() * (double)(m - i);
          if (c < best) best = c, best_axis = ax, split = lo + i, best_order = order;
        }
      }
      if (!best_order.empty()) std::copy(best_order.begin(), best_order.end(), ids.begin() + lo);
    }
    Box ba, bb;
    uint32_t ea = build(lo, split, budget - 1, ba);
    uint32_t eb = build(split, hi, budget - 1, bb);
    FastNode &nd = nodes[self];
    put(nd, 0, ba), put(nd, 1, bb);
    nd.child = ea | (eb << 16);
    out = ba;
    out.grow(bb);
    return self;
  };
  Box all;
  root_entry = build(0, n, FAST_MAX_DEPTH, all);
  return true;
}

}  // namespace rl
