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
#include <cstdio>
#include <cstdlib>
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


// =====================================================================================================================
// General scenes: world-space tree over primitive occurrences.
//
// The same argument as above carries over with three additions.
//  * Instances.  The reference evaluates a primitive under Translate / Transform scopes in OBJECT space (transform.rs:143-164: the
//    ray is transformed, t is not), so the device does exactly that for every candidate (replay of the PUSH chain, then the
//    reference's Sphere::hit / Plane::hit_ab arithmetic); only the reject-only boxes live in world space: bounds of the
//    primitive's transformed vertices (sphere: of the eight corners of its object-space box), grown by a relative 1e-9 of the
//    coordinates involved plus, for spheres, guard_pad mapped through the chain's norm bounds.
//  * Planar hits near an edge (barycentric within 1e-9 of 0 / 1) join the order-sensitive cases: the reference's own leaf box is
//    the exact bound of the vertices, so an edge-grazing hit may or may not pass it.
//  * The panic sites that depend on WHICH hits are accepted (sphere from_normalized, interpolated triangle normal ~ 0, instance
//    normal ~ 0) must be unreachable for every ray that relies on pruning: spheres bound the ray origins for which that holds (r_safe,
//    checked per ray in binary32); a ray that starts farther away — e.g. inside a huge ground sphere after a self-intersection at
//    t ~ 1e-10, which the reference produces too — widens every box by the padding its distance needs, does not prune by the closest
//    hit, and is re-traced in the reference's order as soon as it numerically touches a sphere from more than ~5e4 radii away.
//    Triangles with vertex normals need pairwise positive dot products, matrices a bounded norm.
namespace {

struct Chain {          // accumulated bounds of one PUSH chain
  std::vector<uint32_t> pushes;  // outermost first
  double ninv = 1.0;    // bound on |M^-1 v| / |v| through the chain (world -> object)
  double shift = 0.0;   // bound on the translation part (world -> object), in object units
};

double mat_norm_inf(const double *m) {
  double n = 0.0;
  for (int r = 0; r < 3; r++) n = std::fmax(n, std::fabs(m[3 * r]) + std::fabs(m[3 * r + 1]) + std::fabs(m[3 * r + 2]));
  return n;
}

void to_world(const rl_rtiow_scene_desc &d, const std::vector<DevOp> &ops, const std::vector<uint32_t> &pushes, double p[3]) {
  for (size_t k = pushes.size(); k-- > 0;) {  // innermost first, as the reference's recursion unwinds
    const DevOp &op = ops[pushes[k]];
    if ((op.code & 0xFFu) == OP_PUSH_TRANSLATE) {
      const double *off = d.translates[op.a].offset;
      for (int i = 0; i < 3; i++) p[i] += off[i];
    } else {
      const double *m = d.transforms[op.a].m;
      double q[3];
      for (int r = 0; r < 3; r++) q[r] = m[3 * r] * p[0] + m[3 * r + 1] * p[1] + m[3 * r + 2] * p[2];
      p[0] = q[0], p[1] = q[1], p[2] = q[2];
    }
  }
}

struct GItem {
  uint32_t seg = 0;  // segment of the program the occurrence belongs to (FastGeneral::seg_roots)
  int medium = -1;   // >= 0: not a world item but part of that medium's boundary (only its padded box is used: FastGeneral::stage_roots)
  bool plane = false;  // an unbounded Plane: no box, in no tree — a stage of its own that every ray visits (FastGeneral::stage_roots)
  FastItem it;
  Box box;       // world-space bounds (unpadded)
  double r = 0;  // spheres: radius; else 0
  double ninv = 1.0, shift = 0.0, cobj = 0.0;  // spheres: chain bounds, |object-space centre|_inf
};

}  // namespace

// Squared distance from the origin to the triangle (a, b, c): vertices, clamped edge projections, and the plane projection when it falls inside
static double min_norm2_on_triangle(const double *a, const double *b, const double *c) {
  auto dot3 = [](const double *x, const double *y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
  double best = std::fmin(dot3(a, a), std::fmin(dot3(b, b), dot3(c, c)));
  const double *e[3][2] = {{a, b}, {b, c}, {c, a}};
  for (auto &pq : e) {
    double dq[3] = {pq[1][0] - pq[0][0], pq[1][1] - pq[0][1], pq[1][2] - pq[0][2]};
    double l2 = dot3(dq, dq);
    if (!(l2 > 0.0)) continue;
    double t = std::fmin(1.0, std::fmax(0.0, -dot3(pq[0], dq) / l2));
    double p[3] = {pq[0][0] + t * dq[0], pq[0][1] + t * dq[1], pq[0][2] + t * dq[2]};
    best = std::fmin(best, dot3(p, p));
  }
  double ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
  double nn[3] = {ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0]};
  double n2 = dot3(nn, nn);
  if (n2 > 0.0) {
    double k = dot3(a, nn) / n2;
    double p[3] = {nn[0] * k, nn[1] * k, nn[2] * k};  // projection of the origin onto the plane
    double ap[3] = {p[0] - a[0], p[1] - a[1], p[2] - a[2]};
    double d00 = dot3(ab, ab), d01 = dot3(ab, ac), d11 = dot3(ac, ac), d20 = dot3(ap, ab), d21 = dot3(ap, ac);
    double den = d00 * d11 - d01 * d01;
    if (den > 0.0) {
      double v = (d11 * d20 - d01 * d21) / den, w = (d00 * d21 - d01 * d20) / den;
      if (v >= 0.0 && w >= 0.0 && v + w <= 1.0) best = std::fmin(best, dot3(p, p));
    }
  }
  return best;
}

// the eight-wide quantised form is A/B material (measured slower): built only when the experimental library asks for it
static bool g_build_octo = false;
void set_build_octo(bool on) { g_build_octo = on; }

bool build_fast_general(const rl_rtiow_scene_desc &d, const RtiowProgram &rt, FastGeneral &out) {
  out = FastGeneral{};
  const std::vector<DevOp> &ops = rt.ops;
  // ---- matrices: finite, bounded norm (an instance normal M^-T n of a unit n then has |.|^2 >= 1 / (3 |M|_inf^2) >> 1e-16)
  for (uint32_t i = 0; i < d.n_transforms; i++) {
    const rl_transform &t = d.transforms[i];
    for (int k = 0; k < 9; k++)
      if (!std::isfinite(t.m[k]) || !std::isfinite(t.inv[k]) || !std::isfinite(t.inv_t[k])) return false;
    if (!(mat_norm_inf(t.m) <= 1e5 && mat_norm_inf(t.inv) <= 1e5)) return false;
  }
  for (uint32_t i = 0; i < d.n_translates; i++)
    for (int k = 0; k < 3; k++)
      if (!(std::fabs(d.translates[i].offset[k]) <= 1e30)) return false;
  // ---- items = primitive occurrences, in program order
  std::vector<GItem> items;
  std::vector<uint32_t> pushes;
  uint32_t cur_seg = 0;  // media seen so far = the segment the next primitive occurrence belongs to
  int in_medium = -1;    // inside the boundary ops of this medium
  uint32_t medium_end = 0;
  auto chain_of = [&](GItem &g) {
    g.it.chain = pushes.empty() ? NONE : pushes.back();
    double ninv = 1.0, shift = 0.0;
    for (uint32_t pc : pushes) {  // outermost first: o' = M^-1 o  or  o' = o - offset
      const DevOp &op = ops[pc];
      if ((op.code & 0xFFu) == OP_PUSH_TRANSLATE) {
        const double *off = d.translates[op.a].offset;
        shift += std::fmax(std::fabs(off[0]), std::fmax(std::fabs(off[1]), std::fabs(off[2]))) * 1.7320508075688772;
      } else {
        double n = mat_norm_inf(d.transforms[op.a].inv) * 1.7320508075688772;  // 2-norm <= sqrt(3) inf-norm
        ninv *= n, shift *= n;
      }
    }
    g.ninv = ninv, g.shift = shift;
  };
  auto add_points = [&](GItem &g, const double (*pts)[3], int n) {
    g.box = EMPTY;
    for (int k = 0; k < n; k++) {
      double p[3] = {pts[k][0], pts[k][1], pts[k][2]};
      to_world(d, ops, pushes, p);
      for (int ax = 0; ax < 3; ax++) g.box.lo[ax] = std::fmin(g.box.lo[ax], p[ax]), g.box.hi[ax] = std::fmax(g.box.hi[ax], p[ax]);
    }
  };
  auto add_sphere = [&](uint32_t payload, uint32_t pc) -> bool {
    const rl_sphere &sp = d.spheres[payload & SPH_INDEX];
    double r = std::fabs(sp.radius);
    if (!(std::isfinite(sp.radius) && r > 0.0)) return false;
    GItem g;
    g.it.kind = 0, g.it.payload = payload, g.it.op_pc = pc, g.seg = cur_seg, g.medium = in_medium;
    chain_of(g);
    double lo[3], hi[3], cmax = 0.0;
    for (int ax = 0; ax < 3; ax++) {
      double c0 = sp.center0[ax], c1 = sp.moving ? sp.center1[ax] : sp.center0[ax];
      if (!std::isfinite(c0) || !std::isfinite(c1)) return false;
      lo[ax] = std::fmin(c0, c1) - r, hi[ax] = std::fmax(c0, c1) + r;
      cmax = std::fmax(cmax, std::fmax(std::fabs(c0), std::fabs(c1)));
    }
    double pts[8][3];
    for (int k = 0; k < 8; k++)
      for (int ax = 0; ax < 3; ax++) pts[k][ax] = ((k >> ax) & 1) ? hi[ax] : lo[ax];
    add_points(g, pts, 8);
    g.r = r, g.cobj = cmax;
    items.push_back(g);
    return true;
  };
  auto add_planar = [&](uint32_t idx, uint32_t pc) -> bool {
    const rl_planar &pl = d.planars[idx];
    if (pl.kind != RL_PLANAR_QUAD && pl.kind != RL_PLANAR_TRIANGLE && pl.kind != RL_PLANAR_PLANE) return false;
    GItem g;
    g.it.kind = 1, g.it.payload = idx, g.it.op_pc = pc, g.seg = cur_seg, g.medium = in_medium;
    chain_of(g);
    if (pl.kind == RL_PLANAR_PLANE) {  // plane.rs:51-100 without the quad's interior test: unbounded, so no box and no place in a tree
      for (int ax = 0; ax < 3; ax++)
        if (!(std::isfinite(pl.q[ax]) && std::isfinite(pl.u[ax]) && std::isfinite(pl.v[ax]))) return false;
      g.plane = true, g.box = EMPTY;
      items.push_back(g);
      return true;
    }
    double pts[4][3];
    for (int ax = 0; ax < 3; ax++) {
      pts[0][ax] = pl.q[ax], pts[1][ax] = pl.q[ax] + pl.u[ax], pts[2][ax] = pl.q[ax] + pl.v[ax], pts[3][ax] = pl.q[ax] + pl.u[ax] + pl.v[ax];
      if (!std::isfinite(pts[3][ax])) return false;
    }
    add_points(g, pts, pl.kind == RL_PLANAR_QUAD ? 4 : 3);
    if (pl.kind == RL_PLANAR_TRIANGLE && pl.has_normals) {  // triangle.rs:78-83: unit(n2 a + n3 b + n1 (1-a-b)) must not come near zero
      const double *n = pl.normals;
      double lmax = 0.0;
      for (int i = 0; i < 3; i++) {
        double l2 = n[3 * i] * n[3 * i] + n[3 * i + 1] * n[3 * i + 1] + n[3 * i + 2] * n[3 * i + 2];
        if (!(l2 >= 1e-6 && l2 <= 1e6)) return false;
        lmax = std::fmax(lmax, l2);
      }
      // the interpolated normal ranges over the triangle spanned by the three vertex normals: its shortest vector must stay well away
      // from zero (at least 1 % of the longest vertex normal)
      if (!(min_norm2_on_triangle(n, n + 3, n + 6) >= 1e-4 * lmax)) return false;
    }
    items.push_back(g);
    return true;
  };
  for (uint32_t pc = 0; pc < ops.size(); pc++) {
    const DevOp &op = ops[pc];
    uint32_t kind = op.code & 0xFFu;
    bool ok = true;
    if (kind == OP_MEDIUM_BEGIN) {  // the boundary's own ops (up to the matching END) are not world primitives: the medium evaluates them itself;
      // their boxes bound the medium (a ray whose stretch inside that box misses [t_min, closest] cannot draw: constant_medium.rs:43-47)
      if (in_medium >= 0 || op.skip <= pc + 1 || op.skip > ops.size()) return false;
      FastMedium fm{pc, pushes.empty() ? NONE : pushes.back(), 0u, 0u, 0u, NONE};
      {  // classify the boundary: PUSH* (PLANAR+ | SPHERE) POP*
        uint32_t q = pc + 1, end = op.skip - 1, chain_in = fm.chain, depth = 0;
        auto code_at = [&](uint32_t i) { return ops[i].code & 0xFFu; };
        while (q < end && (code_at(q) == OP_PUSH_TRANSLATE || code_at(q) == OP_PUSH_TRANSFORM)) chain_in = q++, depth++;
        const uint32_t first = q;
        uint32_t shape = 0;
        if (q < end && code_at(q) == OP_SPHERE) q++, shape = 2;
        else {
          while (q < end && code_at(q) == OP_PLANAR && d.planars[ops[q].a].kind != RL_PLANAR_PLANE &&
                 !(d.planars[ops[q].a].kind == RL_PLANAR_TRIANGLE && d.planars[ops[q].a].has_normals))
            q++;
          if (q > first) shape = 1;
        }
        const uint32_t count = q - first;
        uint32_t pops = 0;
        while (q < end && (code_at(q) == OP_POP_TRANSLATE || code_at(q) == OP_POP_TRANSFORM)) q++, pops++;
        if (shape && q == end && pops == depth) fm.shape = shape, fm.first = first, fm.count = count, fm.chain_in = chain_in;
      }
      out.media.push_back(fm);
      in_medium = (int)out.media.size() - 1, medium_end = op.skip - 1;
      continue;
    }
    if (in_medium >= 0 && pc == medium_end) {  // the OP_MEDIUM_END
      in_medium = -1, cur_seg++;
      continue;
    }
    if (kind == OP_PUSH_TRANSLATE || kind == OP_PUSH_TRANSFORM) pushes.push_back(pc);
    else if (kind == OP_POP_TRANSLATE || kind == OP_POP_TRANSFORM) pushes.pop_back();
    else if (kind == OP_SPHERE) ok = add_sphere(op.a, pc);
    else if (kind == OP_PLANAR) ok = add_planar(op.a, pc);
    else if (kind == OP_BOX_SPH) ok = add_sphere(op.a, pc) && (op.b == NONE || add_sphere(op.b, pc));
    else if (kind == OP_BOX_PLANAR) ok = add_planar(op.a, pc) && (op.b == NONE || add_planar(op.b, pc));
    if (!ok) return false;
  }
  if (in_medium >= 0) return false;
  // world items first (program order), the media's boundary parts behind them
  std::stable_sort(items.begin(), items.end(), [](const GItem &a, const GItem &b) {  // (within a segment: its planes first, then the boxed items in program order)
    if ((a.medium >= 0) != (b.medium >= 0)) return a.medium < 0;
    if (a.medium >= 0) return false;
    if (a.seg != b.seg) return a.seg < b.seg;
    return a.plane && !b.plane;
  });
  const size_t n_all = items.size();
  size_t n = 0;
  while (n < n_all && items[n].medium < 0) n++;
  if (n_all >= 0x3FFFFFF0u || out.media.size() >= 0x10000u) return false;
  const size_t n_seg = out.media.size() + 1;
  if (n_all == 0) {
    out.ok = true, out.root = NONE, out.r_safe = 1e30f;
    out.seg_roots.assign(n_seg, NONE);
    out.stage_roots.assign(1, NONE);
    out.media_stage.assign(n_seg - 1, NONE);
    return true;
  }
  size_t n_boxed = 0;
  for (const GItem &g : items) {
    if (g.plane) continue;
    n_boxed++;
    for (int ax = 0; ax < 3; ax++)
      if (!(std::fabs(g.box.lo[ax]) <= 1e30 && std::fabs(g.box.hi[ax]) <= 1e30)) return false;
  }
  // ---- frame: centre = mean of the item centres; r_safe = (nearly) the largest radius for which every sphere's padding and
  // normal-check bounds hold for ray origins within r_safe of the centre
  double c[3] = {0, 0, 0};
  for (const GItem &g : items)
    for (int ax = 0; ax < 3; ax++)
      if (!g.plane) c[ax] += 0.5 * (g.box.lo[ax] + g.box.hi[ax]) / (double)n_boxed;
  const double u = 1.1102230246251565e-16;
  auto far_of = [&](const GItem &g) {  // farthest point of the item's world box from the centre
    double s = 0.0;
    for (int ax = 0; ax < 3; ax++) {
      double m = std::fmax(std::fabs(g.box.lo[ax] - c[ax]), std::fabs(g.box.hi[ax] - c[ax]));
      s += m * m;
    }
    return std::sqrt(s);
  };
  auto sphere_bounds = [&](const GItem &g, double R, double &Lobj, double &Mobj) {
    Lobj = (g.ninv * (R + far_of(g)) + g.shift) * 1.001 + 2.0 * g.r;  // |o' - c| + r in object space, origin within R (Euclidean) of the centre
    Mobj = g.cobj + g.r + Lobj;
  };
  // Sphere::hit's outward normal misses unit length by at most 24u ((|oc| + r) / r)^2 (residual of the rounded root in the rounded
  // quadratic) + 3.5u (|oc| + r + M) / r (rounding of p = o + t d) + 8u (the evaluation itself); the assert fires beyond 1e-5
  auto admissible = [&](double R) {
    for (const GItem &g : items) {
      if (g.it.kind != 0) continue;
      double L, M;
      sphere_bounds(g, R, L, M);
      if (!(32.0 * u * (L / g.r) * (L / g.r) + 8.0 * u * (M + L) / g.r + 16.0 * u <= 1e-5)) return false;
    }
    return true;
  };
  double r_safe = 0.0;
  for (int e = 60; e >= -20 && r_safe == 0.0; e--)
    if (admissible(std::ldexp(1.0, e))) r_safe = std::ldexp(1.0, e);
  if (r_safe == 0.0) return false;
  for (double step = 0.5 * r_safe; step > 1e-3 * r_safe; step *= 0.5)  // refine between r_safe and 2 r_safe
    if (admissible(r_safe + step)) r_safe += step;
  r_safe = std::fmin(r_safe, 1e18);
  // ---- padded binary32 boxes
  std::vector<Box> pb(n_all);
  double cabs = std::fmax(std::fabs(c[0]), std::fmax(std::fabs(c[1]), std::fabs(c[2])));
  for (size_t i = 0; i < n_all; i++) {
    const GItem &g = items[i];
    pb[i] = EMPTY;
    if (g.plane) continue;
    double extra = 0.0;
    if (g.it.kind == 0) {  // object-space guard pad, mapped to world space by the forward norm bound (<= 1e5 * sqrt(3) per level, folded into ninv's reciprocal is not available: use the world box growth factor)
      double L, M;
      sphere_bounds(g, r_safe, L, M);
      double pad_obj = 8.0 * (8.0 * u * L * L + 2.0 * u * M * L) / g.r;
      double ext = 0.0;
      for (int ax = 0; ax < 3; ax++) ext = std::fmax(ext, g.box.hi[ax] - g.box.lo[ax]);
      extra = pad_obj * (ext / (2.0 * g.r)) * 1.7320508075688772;  // world extent / object diameter bounds the chain's stretch
    }
    for (int ax = 0; ax < 3; ax++) {
      double lo = g.box.lo[ax], hi = g.box.hi[ax];
      double pad = 1e-9 * (std::fabs(lo) + std::fabs(hi) + (hi - lo) + cabs + far_of(g)) + extra + 1e-300;
      pb[i].lo[ax] = lo - pad, pb[i].hi[ax] = hi + pad;
    }
  }
  // ---- binned SAH build (16 bins per axis on the box centres), leaf = one item, depth <= FASTG_MAX_DEPTH
  out.items.resize(n);
  out.item_spheres.assign(n, DevSphere{});
  out.item_material.assign(n, 0u);
  for (size_t i = 0; i < n; i++) {
    out.items[i] = items[i].it;
    if (items[i].it.kind == 0) {
      const uint32_t si = items[i].it.payload & SPH_INDEX;
      out.item_spheres[i] = rt.spheres[si], out.item_material[i] = rt.sphere_material[si];
    } else out.item_material[i] = rt.planars[items[i].it.payload].material;
  }
  std::vector<uint32_t> ids(n);
  std::iota(ids.begin(), ids.end(), 0u);
  out.nodes.reserve(n);
  auto put = [&](FastNodeG &nd, int side, const Box &b) {
    for (int ax = 0; ax < 3; ax++) nd.box[side][2 * ax] = round_down(b.lo[ax]), nd.box[side][2 * ax + 1] = round_up(b.hi[ax]);
  };
  std::function<uint32_t(size_t, size_t, uint32_t, Box &)> build = [&](size_t lo, size_t hi, uint32_t budget, Box &ob) -> uint32_t {
    const size_t m = hi - lo;
    if (m == 1) {
      ob = pb[ids[lo]];
      return FASTG_LEAF | ids[lo];
    }
    const uint32_t self = (uint32_t)out.nodes.size();
    out.nodes.push_back(FastNodeG{});
    uint32_t need = 0;
    while (((size_t)1 << need) < m) need++;
    Box cb = EMPTY;  // bounds of the box centres
    for (size_t i = lo; i < hi; i++) {
      const Box &b = pb[ids[i]];
      for (int ax = 0; ax < 3; ax++) {
        double ctr = 0.5 * (b.lo[ax] + b.hi[ax]);
        cb.lo[ax] = std::fmin(cb.lo[ax], ctr), cb.hi[ax] = std::fmax(cb.hi[ax], ctr);
      }
    }
    size_t split = lo + m / 2;
    bool done = false;
    if (need < budget) {
      const int NB = 16;
      const size_t cap = budget - 1 >= 63 ? (size_t)-1 : ((size_t)1 << (budget - 1));
      double best = INFINITY;
      int best_ax = -1, best_bin = -1;
      for (int ax = 0; ax < 3; ax++) {
        double w = cb.hi[ax] - cb.lo[ax];
        if (!(w > 0.0)) continue;
        Box bins[NB];
        size_t cnt[NB];
        for (int b = 0; b < NB; b++) bins[b] = EMPTY, cnt[b] = 0;
        double scale = (double)NB / w;
        for (size_t i = lo; i < hi; i++) {
          const Box &b = pb[ids[i]];
          int k = (int)((0.5 * (b.lo[ax] + b.hi[ax]) - cb.lo[ax]) * scale);
          k = k < 0 ? 0 : (k >= NB ? NB - 1 : k);
          bins[k].grow(b), cnt[k]++;
        }
        double la[NB];
        size_t lc[NB];
        Box acc = EMPTY;
        size_t c0 = 0;
        for (int b = 0; b < NB - 1; b++) acc.grow(bins[b]), c0 += cnt[b], la[b] = c0 ? acc.area() * (double)c0 : 0.0, lc[b] = c0;
        acc = EMPTY, c0 = 0;
        for (int b = NB - 1; b >= 1; b--) {
          acc.grow(bins[b]), c0 += cnt[b];
          size_t left = lc[b - 1];
          if (left == 0 || c0 == 0 || left > cap || c0 > cap) continue;
          double cost = la[b - 1] + acc.area() * (double)c0;
          if (cost < best) best = cost, best_ax = ax, best_bin = b;
        }
      }
      if (best_ax >= 0) {
        double w = cb.hi[best_ax] - cb.lo[best_ax], scale = 16.0 / w;
        auto mid = std::partition(ids.begin() + (long)lo, ids.begin() + (long)hi, [&](uint32_t id) {
          const Box &b = pb[id];
          int k = (int)((0.5 * (b.lo[best_ax] + b.hi[best_ax]) - cb.lo[best_ax]) * scale);
          k = k < 0 ? 0 : (k >= 16 ? 15 : k);
          return k < best_bin;
        });
        split = (size_t)(mid - ids.begin());
        done = split > lo && split < hi;
      }
    }
    if (!done) {  // no slack left for the heuristic (or all centres equal): median split along the widest axis of the centres
      int ax = 0;
      double e[3] = {cb.hi[0] - cb.lo[0], cb.hi[1] - cb.lo[1], cb.hi[2] - cb.lo[2]};
      ax = e[0] >= e[1] ? (e[0] >= e[2] ? 0 : 2) : (e[1] >= e[2] ? 1 : 2);
      split = lo + m / 2;
      std::nth_element(ids.begin() + (long)lo, ids.begin() + (long)split, ids.begin() + (long)hi,
                       [&](uint32_t a, uint32_t b) { return pb[a].lo[ax] + pb[a].hi[ax] < pb[b].lo[ax] + pb[b].hi[ax]; });
    }
    Box ba, bb;
    uint32_t ea = build(lo, split, budget - 1, ba);
    uint32_t eb = build(split, hi, budget - 1, bb);
    FastNodeG &nd = out.nodes[self];
    put(nd, 0, ba), put(nd, 1, bb);
    nd.child[0] = ea, nd.child[1] = eb;
    ob = ba;
    ob.grow(bb);
    return self;
  };
  // one tree per segment (items are in program order, so a segment is a contiguous range of ids)
  std::vector<uint32_t> broots(n_seg, NONE);
  {
    size_t lo = 0;
    for (size_t sg = 0; sg < n_seg; sg++) {
      size_t hi = lo;
      while (hi < n && items[hi].seg == sg) hi++;
      size_t mid = lo;  // the segment's planes come first: stages of their own, not in the tree
      while (mid < hi && items[mid].plane) mid++;
      if (hi - mid == 1) broots[sg] = FASTG_LEAF | (uint32_t)mid;  // a single occurrence: rays start in LEAF, no node at all
      else if (hi > mid) {
        Box all;
        broots[sg] = build(mid, hi, FASTG_MAX_DEPTH, all);
      }
      lo = hi;
    }
    if (lo != n) return false;
  }
  out.root = broots[0];
  // rays from outside r_safe: guard_pad's bound 8 (8u L^2 + 2u M L) / r with L = |oc| <= distance + radius and M <= L + |centre|, for the
  // smallest sphere; planars need far less (a relative 1e-15 of the coordinates), so one constant serves all items
  {
    double rmin = INFINITY, radius = 0.0;
    for (const GItem &g : items) {
      if (g.plane) continue;
      radius = std::fmax(radius, far_of(g));
      if (g.it.kind == 0) rmin = std::fmin(rmin, g.r / g.ninv);  // object-space pad seen from world space: conservative through the chain norms
    }
    if (!(rmin < INFINITY)) rmin = 1.0;
    out.radius = round_up(radius + cabs);
    out.pad_k = round_up(8.0 * (8.0 * u + 2.0 * u) / rmin * 1.5 + 1e-12);
  }
  // ---- four-wide form: fold every second level (the child with the largest box is replaced by its own two children until four are held)
  out.qnodes.clear();
  out.seg_roots = broots;
  {
    struct Cand {
      float box[6];
      uint32_t e;
    };
    out.qnodes.reserve(out.nodes.size() / 2 + 1);
    std::function<uint32_t(uint32_t)> fold = [&](uint32_t n2) -> uint32_t {
      Cand cs[4];
      int nc = 0;
      auto add = [&](const FastNodeG &nd, int side) {
        std::memcpy(cs[nc].box, nd.box[side], sizeof cs[nc].box);
        cs[nc].e = nd.child[side];
        nc++;
      };
      add(out.nodes[n2], 0), add(out.nodes[n2], 1);
      while (nc < 4) {
        int pick = -1;
        double best = -1.0;
        for (int k = 0; k < nc; k++) {
          if (cs[k].e & FASTG_LEAF) continue;
          const float *b = cs[k].box;
          double ex = (double)b[1] - b[0], ey = (double)b[3] - b[2], ez = (double)b[5] - b[4];
          double area = ex * ey + ey * ez + ez * ex;
          if (!(area <= best)) best = area, pick = k;  // NaN / inf areas count as largest
        }
        if (pick < 0) break;
        const FastNodeG &nd = out.nodes[cs[pick].e];
        cs[pick] = cs[nc - 1];
        nc--;
        add(nd, 0), add(nd, 1);
      }
      const uint32_t self = (uint32_t)out.qnodes.size();
      out.qnodes.push_back(FastNodeQ{});
      uint32_t ch[4] = {NONE, NONE, NONE, NONE};
      for (int k = 0; k < nc; k++) ch[k] = (cs[k].e & FASTG_LEAF) ? cs[k].e : fold(cs[k].e);
      FastNodeQ &q = out.qnodes[self];
      for (int k = 0; k < 4; k++) {
        for (int ax = 0; ax < 3; ax++) q.lo[ax][k] = k < nc ? cs[k].box[2 * ax] : 0.0f, q.hi[ax][k] = k < nc ? cs[k].box[2 * ax + 1] : 0.0f;
        q.child[k] = ch[k];
      }
      return self;
    };
    for (size_t sg = 0; sg < n_seg; sg++)
      if (broots[sg] != NONE && !(broots[sg] & FASTG_LEAF)) out.seg_roots[sg] = fold(broots[sg]);
  }
  // ---- the top of the first tree in breadth-first order at the front of the array (FASTG_TOP_NODES of them: rl_rtiow_fastgen.h stages them in LDS)
  out.top_nodes = 0;
  if (out.seg_roots[0] != NONE && !(out.seg_roots[0] & FASTG_LEAF)) {
    std::vector<uint32_t> order;  // old ids, new order
    std::vector<uint32_t> newid(out.qnodes.size(), NONE);
    order.push_back(out.seg_roots[0]);
    newid[out.seg_roots[0]] = 0;
    for (size_t h = 0; h < order.size() && order.size() < FASTG_TOP_NODES; h++)
      for (int k = 0; k < 4; k++) {
        const uint32_t ch = out.qnodes[order[h]].child[k];
        if (ch == NONE || (ch & FASTG_LEAF) || order.size() >= FASTG_TOP_NODES) continue;
        newid[ch] = (uint32_t)order.size();
        order.push_back(ch);
      }
    out.top_nodes = (uint32_t)order.size();
    for (uint32_t i = 0; i < out.qnodes.size(); i++)
      if (newid[i] == NONE) newid[i] = (uint32_t)order.size(), order.push_back(i);
    std::vector<FastNodeQ> moved(out.qnodes.size());
    for (size_t i = 0; i < order.size(); i++) {
      FastNodeQ q = out.qnodes[order[i]];
      for (int k = 0; k < 4; k++)
        if (q.child[k] != NONE && !(q.child[k] & FASTG_LEAF)) q.child[k] = newid[q.child[k]];
      moved[i] = q;
    }
    out.qnodes.swap(moved);
    for (uint32_t &r : out.seg_roots)
      if (r != NONE && !(r & FASTG_LEAF)) r = newid[r];
  }
  out.qroot = out.seg_roots[0];
  // ---- stages of a ray's walk: segment 0's tree, medium 0, segment 1's tree, ...; a medium is a one-child node whose box is the union of
  // its boundary parts' padded boxes (a boundary without bounded parts can never be hit: no stage)
  out.stage_roots.clear();
  out.media_stage.assign(n_seg - 1, NONE);
  {
    size_t it = 0;
    for (size_t sg = 0; sg < n_seg; sg++) {
      for (; it < n && items[it].seg == sg; it++)
        if (items[it].plane) out.stage_roots.push_back(FASTG_LEAF | (uint32_t)it);  // every ray visits an unbounded Plane
      if (out.seg_roots[sg] != NONE) out.stage_roots.push_back(out.seg_roots[sg]);
      if (sg + 1 == n_seg) break;
      const size_t k = sg;
      Box mb = EMPTY;
      bool any = false, unbounded = false;
      for (size_t i = n; i < n_all; i++)
        if (items[i].medium == (int)k) {
          any = true;
          if (items[i].plane) unbounded = true;
          else mb.grow(pb[i]);
        }
      if (!any) continue;  // a boundary without parts can never be hit
      out.media_stage[k] = (uint32_t)out.stage_roots.size();
      if (unbounded) {  // a Plane in the boundary: no box, every ray evaluates the medium
        out.stage_roots.push_back(FASTG_LEAF | FASTG_MEDIUM | (uint32_t)k);
        continue;
      }
      FastNodeQ q{};
      for (int ax = 0; ax < 3; ax++) q.lo[ax][0] = round_down(mb.lo[ax]), q.hi[ax][0] = round_up(mb.hi[ax]);
      q.child[0] = FASTG_LEAF | FASTG_MEDIUM | (uint32_t)k, q.child[1] = q.child[2] = q.child[3] = NONE;
      out.stage_roots.push_back((uint32_t)out.qnodes.size());
      out.qnodes.push_back(q);
    }
    if (out.stage_roots.empty()) out.stage_roots.push_back(NONE);
  }
  // ---- eight-wide form with quantised boxes (FastNodeO): fold until eight children are held (largest box first), then put every child's
  // box on the node's 8-bit grid, rounded OUTWARDS — checked below in exact arithmetic (every term is a dyadic rational that binary64 holds)
  out.onodes.clear();
  out.oroot = out.root;
  bool octo_ok = true;
  if (g_build_octo && n_seg == 1 && out.root != NONE && !(out.root & FASTG_LEAF)) {
    struct Cand {
      float box[6];
      uint32_t e;
    };
    out.onodes.reserve(out.nodes.size() / 4 + 1);
    std::function<uint32_t(uint32_t)> fold8 = [&](uint32_t n2) -> uint32_t {
      Cand cs[8];
      int nc = 0;
      auto add = [&](const FastNodeG &nd, int side) {
        std::memcpy(cs[nc].box, nd.box[side], sizeof cs[nc].box);
        cs[nc].e = nd.child[side];
        nc++;
      };
      add(out.nodes[n2], 0), add(out.nodes[n2], 1);
      // the node's own extent fixes the grid pitch (extent / 255 per axis): a child is replaced by its two children only if those are
      // still large against the pitch — a box of extent e on a grid of pitch p grows by up to 2p, and a subtree that is tiny next to a
      // sibling (cfg 5: the 1000 x 1000 field of spheres next to the radius-1e6 ground) would otherwise be blown up to a grid cell
      double ulo[3] = {INFINITY, INFINITY, INFINITY}, uhi[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (int k = 0; k < 2; k++)
        for (int ax = 0; ax < 3; ax++) ulo[ax] = std::fmin(ulo[ax], (double)cs[k].box[2 * ax]), uhi[ax] = std::fmax(uhi[ax], (double)cs[k].box[2 * ax + 1]);
      const double pitch_max = std::fmax(std::fmax(uhi[0] - ulo[0], uhi[1] - ulo[1]), uhi[2] - ulo[2]) / 255.0;
      auto big_enough = [&](const float *b) {
        const double e = std::fmax(std::fmax((double)b[1] - b[0], (double)b[3] - b[2]), (double)b[5] - b[4]);
        return e >= 16.0 * pitch_max;  // grows by <= 1/8 of its largest extent
      };
      while (nc < 8) {
        int pick = -1;
        double best = -1.0;
        for (int k = 0; k < nc; k++) {
          if (cs[k].e & FASTG_LEAF) continue;
          const FastNodeG &cn = out.nodes[cs[k].e];
          if (!(big_enough(cn.box[0]) && big_enough(cn.box[1]))) continue;
          const float *b = cs[k].box;
          double ex = (double)b[1] - b[0], ey = (double)b[3] - b[2], ez = (double)b[5] - b[4];
          double area = ex * ey + ey * ez + ez * ex;
          if (!(area <= best)) best = area, pick = k;
        }
        if (pick < 0) break;
        const FastNodeG &nd = out.nodes[cs[pick].e];
        cs[pick] = cs[nc - 1];
        nc--;
        add(nd, 0), add(nd, 1);
      }
      const uint32_t self = (uint32_t)out.onodes.size();
      out.onodes.push_back(FastNodeO{});
      uint32_t ch[8];
      for (int k = 0; k < 8; k++) ch[k] = NONE;
      for (int k = 0; k < nc; k++) ch[k] = (cs[k].e & FASTG_LEAF) ? cs[k].e : fold8(cs[k].e);
      FastNodeO &q = out.onodes[self];
      q.exps = 0;
      for (int ax = 0; ax < 3; ax++) {
        double lo = INFINITY, hi = -INFINITY;
        for (int k = 0; k < nc; k++) lo = std::fmin(lo, (double)cs[k].box[2 * ax]), hi = std::fmax(hi, (double)cs[k].box[2 * ax + 1]);
        if (!(std::fabs(lo) <= 1e30 && std::fabs(hi) <= 1e30)) octo_ok = false, lo = 0.0, hi = 1.0;
        q.o[ax] = (float)lo;  // lo IS one of the children's float bounds: exact
        // pitch 2^p with 255 * 2^p >= hi - lo; exponent kept in [-40, 60] (a ray's |1/d| is within [1e-30, 1e30]: products stay normal numbers)
        int p = -40;
        while (std::ldexp(255.0, p) < hi - lo && p < 60) p++;
        if (std::ldexp(255.0, p) < hi - lo) octo_ok = false;
        const double pitch = std::ldexp(1.0, p);
        q.exps |= (uint32_t)(p + 127) << (8 * ax);
        for (int k = 0; k < 8; k++) {
          if (k >= nc) {
            q.qlo[ax][k] = 255, q.qhi[ax][k] = 0;  // empty slot: inverted (the kernel also checks child != NONE)
            continue;
          }
          const double l = (double)cs[k].box[2 * ax] - lo, h = (double)cs[k].box[2 * ax + 1] - lo;  // exact (binary32 operands in binary64)
          double ql = std::floor(l / pitch), qh = std::ceil(h / pitch);  // division by a power of two: exact
          ql = ql < 0.0 ? 0.0 : (ql > 255.0 ? 255.0 : ql), qh = qh < 0.0 ? 0.0 : (qh > 255.0 ? 255.0 : qh);
          q.qlo[ax][k] = (uint8_t)ql, q.qhi[ax][k] = (uint8_t)qh;
          // the stored planes, in exact arithmetic, must enclose the child's own box
          if (!(lo + ql * pitch <= (double)cs[k].box[2 * ax] && lo + qh * pitch >= (double)cs[k].box[2 * ax + 1])) octo_ok = false;
        }
      }
      for (int k = 0; k < 8; k++) q.child[k] = ch[k];
      return self;
    };
    out.oroot = fold8(out.root);
  }
  if (!octo_ok) out.onodes.clear(), out.oroot = NONE;  // (non-finite / absurd extents: the four-wide form is walked instead)
  for (int ax = 0; ax < 3; ax++) out.center[ax] = (float)c[ax];
  out.r_safe = round_down(r_safe * 0.999 - 1e-6 * cabs);  // the device compares binary32 roundings of o and centre
  out.ok = out.r_safe > 0.0f;
  if (std::getenv("RL_DEBUG_FG")) std::fprintf(stderr, "[fastg] items %zu centre (%g %g %g) r_safe %g radius %g pad_k %g\n", n, c[0], c[1], c[2], (double)out.r_safe, (double)out.radius, (double)out.pad_k);
  return out.ok;
}

}  // namespace rl
