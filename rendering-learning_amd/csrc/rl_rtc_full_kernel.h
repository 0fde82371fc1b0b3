// RTC full World::color_at on gfx950: every Object of the reference — triangles,
// unit sphere / xz plane / cube / cylinder / cone (object/*.rs), CSG (object/csg.rs), Group / Bounded /
// Transformed — with Color or Pattern surfaces (pattern/*.rs) and the recursive reflection / refraction of
// world.rs:57-159 (Schlick blend, containers walk of intersect.rs:72-99, depth max_reflection_depth).
//
// One lane per pixel.  A ray's intersections are collected into a per-lane list (scratch memory) in the
// reference's evaluation order; CSG scopes sort + filter their own range (csg.rs:50-108), Transformed scopes
// re-normalise the normals of theirs (transformed.rs:43-49); the world-level stable sort by t then gives exactly
// the reference's `xs`.  An entry is 40 bytes {t, leaf, enclosing ENTER, normal}: the surface colour (material.rs:13-20, a pure
// function of the local-space point) is evaluated for the HIT only, from the local ray its ENTER chain reproduces, and neither the
// containers walk nor the shadow test keeps a list of its own (both are answered by rescanning the sorted entries).  The reflection / refraction tree is walked with an explicit stack of weighted rays:
// colour is linear in the sub-rays' colours (reflectivity, transparency and the Schlick factor are scalars), so
// total = sum over tree nodes of weight * surface colour — a reassociation of the reference's sums (ulp-level).
#pragma once
#include "rl_rtc_kernel.h"

namespace rl {

struct RtcFullParams {
  RtcParams R;
  const rl_rtc_shape *shapes;
  const rl_rtc_csg *csgs;
  const rl_rtc_pattern *patterns;
  uint32_t n_tris;
  uint32_t max_reflection_depth;
};

#define RL_RTC_K 48  // intersections kept per ray (overflow is flagged)

struct Ent {  // one Intersection (intersect.rs:11-16) + the CSG side tag
  double t;
  uint32_t leaf;   // triangles first, then shapes: object identity; bit 31: CSG side tag (set = right child)
  uint32_t chain;  // pc of the innermost enclosing ROP_ENTER (NONE: world space): reproduces the local ray for the hit's colour
  D3 normal;
};
static const uint32_t ENT_RIGHT = 0x80000000u;

__device__ __forceinline__ long long f2i64(double f) { return (long long)f; }  // v_cvt saturates; NaN -> 0

__device__ inline D3 rtc_surface_color(const RtcFullParams &F, const rl_rtc_material &m, D3 p) {  // material.rs:13-20
  if (m.pattern == 0) return ld3(m.color);
  const rl_rtc_pattern &pt = F.patterns[m.pattern - 1];
  D3 q = mul_point(pt.inverse, p);
  D3 a = ld3(pt.a), b = ld3(pt.b);
  if (pt.kind == RL_PAT_STRIPE) return (f2i64(floor(q.x)) % 2 == 0) ? a : b;
  if (pt.kind == RL_PAT_RING) return (f2i64(floor(sqrt(q.x * q.x + q.z * q.z))) % 2 == 0) ? a : b;
  if (pt.kind == RL_PAT_GRADIENT) {
    D3 distance = b - a;
    double fraction = q.x - floor(q.x);
    return a + distance * fraction;
  }
  return (f2i64(floor(q.x) + floor(q.y) + floor(q.z)) % 2 == 0) ? a : b;
}

__device__ __forceinline__ uint32_t rtc_leaf_material(const RtcFullParams &F, const DevTri *tris, uint32_t leaf) {
  return leaf < F.n_tris ? tris[leaf].material : F.shapes[leaf - F.n_tris].material;
}

struct RtcFullCounters {
  unsigned long long rays, nodes, tris, spheres, enters, flagged;
};

__device__ __forceinline__ void ent_push(Ent *list, uint32_t &n, double t, uint32_t leaf, uint32_t chain, D3 normal, RtcFullCounters &cnt) {
  if (n < RL_RTC_K) {
    Ent e;
    e.t = t, e.leaf = leaf, e.chain = chain, e.normal = normal;
    list[n++] = e;
  } else
    cnt.flagged++;
}
__device__ __forceinline__ D3 nrm_or_flag(D3 v, RtcFullCounters &cnt) {
  D3 n;
  if (!norm(v, n)) {
    cnt.flagged++;
    return d3(0.0, 0.0, 0.0);
  }
  return n;
}
__device__ inline void ent_sort(Ent *list, uint32_t lo, uint32_t hi) {  // stable insertion sort by t
  for (uint32_t i = lo + 1; i < hi; i++) {
    Ent e = list[i];
    uint32_t j = i;
    while (j > lo && list[j - 1].t > e.t) {
      list[j] = list[j - 1];
      j--;
    }
    list[j] = e;
  }
}

// The ray inside the Transformed scope whose ROP_ENTER is at `chain` (NONE: the world ray): transformed.rs:38-41 applied outermost first
__device__ inline void rtc_local_ray(const RtcParams &P, const DevOp *ops, uint32_t chain, D3 wo, D3 wd, D3 &o, D3 &d) {
  o = wo, d = wd;
  uint32_t stack[8];
  int ns = 0;
  while (chain != NONE && ns < 8) {
    stack[ns++] = chain;
    chain = ops[chain].b;
  }
  for (int i = ns - 1; i >= 0; i--) {
    const rl_rtc_transformed &px = P.xforms[ops[stack[i]].a];
    D3 no = mul_point(px.inverse, o), nd = mul_vec(px.inverse, d);
    o = no, d = nd;
  }
}

// guard_reject32 (rl_rtc_kernel.h) without a parameter range: true only when the whole LINE certainly misses the box
__device__ __forceinline__ bool guard_reject32_line(const float *b, const RtcAux32 &ra) {
  float t0x = fmaf(b[0], ra.invx, -ra.oix), t1x = fmaf(b[1], ra.invx, -ra.oix);
  float t0y = fmaf(b[2], ra.invy, -ra.oiy), t1y = fmaf(b[3], ra.invy, -ra.oiy);
  float t0z = fmaf(b[4], ra.invz, -ra.oiz), t1z = fmaf(b[5], ra.invz, -ra.oiz);
  float tmin = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
  float tmax = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
  float thresh = fmaf(fabsf(tmin) + fabsf(tmax), 4.76837158203125e-07f, ra.slack);  // 8u(|tmin|+|tmax|) + slack
  return (tmin - tmax) > thresh;                                                     // false for NaN / inf arithmetic
}

// World::intersect (world.rs:46-55) into list[0..n): evaluation order, CSG-filtered, normals in world space, sorted by t
__device__ inline uint32_t rtc_intersect_all(const RtcFullParams &F, const DevOp *ops, const DevTri *tris, D3 wo, D3 wd, Ent *list,
                                            RtcFullCounters &cnt, unsigned long long mult) {
  const RtcParams &P = F.R;
  D3 o = wo, d = wd;
  uint32_t n = 0, pc = 0;
  uint32_t xf_start[8], csg_start[4], csg_mid[4];
  int xf_depth = 0, csg_depth = 0;
  uint32_t cur_enter = NONE;  // innermost enclosing ROP_ENTER
  for (;;) {
    const DevOp &op = ops[pc];
    uint32_t code = op.code;
    if (code == ROP_END) break;
    if (code == ROP_TRIS) {
      uint32_t first = op.a, count = op.b;
      auto test_triangle = [&](uint32_t ti) {  // triangle.rs:63-101
        const DevTri &t = tris[ti];
        D3 e1 = ld3(t.e1), e2 = ld3(t.e2);
        D3 dir_cross_e2 = cross(d, e2);
        double det = dot(e1, dir_cross_e2);
        if (fabs(det) < 1e-8) return;
        double f = 1.0 / det;
        D3 p1_to_origin = o - ld3(t.p1);
        double u = f * dot(p1_to_origin, dir_cross_e2);
        if (!(0.0 <= u && u <= 1.0)) return;
        D3 origin_cross_e1 = cross(p1_to_origin, e1);
        double v = f * dot(d, origin_cross_e1);
        if (v < 0.0 || (u + v) > 1.0) return;
        double tt = f * dot(e2, origin_cross_e1);
        D3 nn = t.smooth ? nrm_or_flag((ld3(t.n2) * u + ld3(t.n3) * v) + ld3(t.n1) * (1.0 - u - v), cnt) : ld3(t.n1);
        ent_push(list, n, tt, ti, cur_enter, nn, cnt);
      };
      cnt.tris += mult * count;  // the reference tests every triangle of the group (group.rs has no acceleration structure)
      if (P.guards && op.skip != 0u) {
        // reject-only box tree over the range (rl_render.hip build_rtc_guards; leaves in triangle order): a triangle is skipped only
        // when the ray's LINE certainly misses its padded box — every intersection, of either sign of t, is still collected in order
        const RtcAux32 ax = rtc_aux32(o, d);
        uint32_t g = op.skip - 1u;
        const uint32_t gend = P.guards[g].skip;
        while (g < gend) {
          const RtcGuard &nd = P.guards[g];
          const float bx[6] = {nd.box[0], nd.box[1], nd.box[2], nd.box[3], nd.box[4], nd.box[5]};
          const uint32_t nskip = nd.skip, ntri = nd.tri;
          if (guard_reject32_line(bx, ax)) g = nskip;
          else if (ntri == NONE) g++;
          else {
            test_triangle(ntri);
            g = nskip;
          }
        }
      } else {
        for (uint32_t k = 0; k < count; k++) test_triangle(first + k);
      }
      pc++;
      continue;
    }
    if (code == ROP_SHAPE) {
      const rl_rtc_shape &sh = F.shapes[op.a];
      uint32_t leaf = F.n_tris + op.a;
      const double EPS = 1e-8;
      double ts[4];
      int nt = 0;
      if (sh.kind == RL_O_SPHERE) {  // sphere.rs:36-60
        cnt.spheres += mult;
        double a = dot(d, d);
        double b = 2.0 * dot(d, o);
        double c = dot(o, o) - 1.0;
        double disc = b * b - 4.0 * a * c;
        if (!(disc < 0.0)) {
          double sq = sqrt(disc);
          ts[nt++] = (-b - sq) / (2.0 * a);
          ts[nt++] = (-b + sq) / (2.0 * a);
        }
      } else if (sh.kind == RL_O_PLANE) {  // plane.rs:27-42
        cnt.tris += mult;
        if (!(fabs(d.y) < 1e-8)) ts[nt++] = -o.y / d.y;
      } else if (sh.kind == RL_O_CUBE) {  // cube.rs:37-78
        cnt.tris += mult;
        double lo[3], hi[3];
        const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
#pragma unroll
        for (int a = 0; a < 3; a++) {
          double tmin = (-1.0 - oo[a]) / dd[a];
          double tmax = (1.0 - oo[a]) / dd[a];
          bool sw = tmin > tmax;
          lo[a] = sw ? tmax : tmin;
          hi[a] = sw ? tmin : tmax;
        }
        double tmin = fmax(fmax(lo[0], lo[1]), lo[2]);
        double tmax = fmin(fmin(hi[0], hi[1]), hi[2]);
        if (!(tmin > tmax)) ts[nt++] = tmin, ts[nt++] = tmax;
      } else {  // cylinder.rs:90-140 / cone.rs:84-140
        cnt.tris += mult;
        bool cone = sh.kind == RL_O_CONE;
        double a = cone ? (d.x * d.x - d.y * d.y + d.z * d.z) : (d.x * d.x + d.z * d.z);
        double b = cone ? (2.0 * o.x * d.x - 2.0 * o.y * d.y + 2.0 * o.z * d.z) : (2.0 * o.x * d.x + 2.0 * o.z * d.z);
        double c = cone ? (o.x * o.x - o.y * o.y + o.z * o.z) : (o.x * o.x + o.z * o.z - 1.0);
        bool a0 = fabs(a) < EPS, b0 = fabs(b) < EPS;
        auto in_bounds = [&](double y) {
          if (sh.has_minimum && sh.has_maximum) return y > sh.minimum && y < sh.maximum;
          if (sh.has_minimum) return y > sh.minimum;
          if (sh.has_maximum) return y < sh.maximum;
          return true;
        };
        if (cone && a0) {
          if (!b0) ts[nt++] = -c / (2.0 * b);
        } else if (!a0) {
          double disc = b * b - 4.0 * a * c;
          if (!(disc < 0.0)) {
            double t0 = (-b - sqrt(disc)) / (2.0 * a);
            double t1 = (-b + sqrt(disc)) / (2.0 * a);
            if (in_bounds(o.y + t0 * d.y)) ts[nt++] = t0;
            if (in_bounds(o.y + t1 * d.y)) ts[nt++] = t1;
          }
        }
        if (sh.closed && !(fabs(d.y) < EPS)) {
          if (sh.has_minimum) {
            double t = (sh.minimum - o.y) / d.y;
            double x = o.x + t * d.x, z = o.z + t * d.z;
            if (x * x + z * z <= (cone ? fabs(sh.minimum) : 1.0)) ts[nt++] = t;
          }
          if (sh.has_maximum) {
            double t = (sh.maximum - o.y) / d.y;
            double x = o.x + t * d.x, z = o.z + t * d.z;
            if (x * x + z * z <= (cone ? fabs(sh.maximum) : 1.0)) ts[nt++] = t;
          }
        }
      }
      for (int i = 0; i < nt; i++) {  // build_basic_intersection (object/mod.rs:20-32): local-space point
        double t = ts[i];
        D3 p = o + d * t;
        D3 nn;
        if (sh.kind == RL_O_SPHERE) nn = nrm_or_flag(p, cnt);
        else if (sh.kind == RL_O_PLANE) nn = d3(0.0, 1.0, 0.0);
        else if (sh.kind == RL_O_CUBE) {
          double ax = fabs(p.x), ay = fabs(p.y), az = fabs(p.z);
          double mc = fmax(fmax(ax, ay), az);
          nn = nrm_or_flag((mc == ax) ? d3(p.x, 0.0, 0.0) : (mc == ay) ? d3(0.0, p.y, 0.0) : d3(0.0, 0.0, p.z), cnt);
        } else if (sh.kind == RL_O_CYLINDER) {
          double dist2 = p.x * p.x + p.z * p.z;
          if (dist2 < 1.0 && sh.has_maximum && p.y >= sh.maximum - EPS) nn = d3(0.0, 1.0, 0.0);
          else if (dist2 < 1.0 && sh.has_minimum && p.y <= sh.minimum + EPS) nn = d3(0.0, -1.0, 0.0);
          else nn = nrm_or_flag(d3(p.x, 0.0, p.z), cnt);
        } else {
          double dist2 = p.x * p.x + p.z * p.z;
          if (sh.has_maximum && dist2 < sh.maximum * sh.maximum && p.y >= sh.maximum - EPS) nn = d3(0.0, 1.0, 0.0);
          else if (sh.has_minimum && dist2 < sh.minimum * sh.minimum && p.y <= sh.minimum + EPS) nn = d3(0.0, -1.0, 0.0);
          else {
            double y = sqrt(p.x * p.x + p.z * p.z);
            if (p.y > 0.0) y = -y;
            nn = nrm_or_flag(d3(p.x, y, p.z), cnt);
          }
        }
        ent_push(list, n, t, leaf, cur_enter, nn, cnt);
      }
      pc++;
      continue;
    }
    if (code == ROP_BOUNDS) {
      cnt.nodes += mult;
      double lo[3], hi[3];
      const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
#pragma unroll
      for (int a = 0; a < 3; a++) {
        double tmin = (op.box[2 * a] - oo[a]) / dd[a];
        double tmax = (op.box[2 * a + 1] - oo[a]) / dd[a];
        bool sw = tmin > tmax;
        lo[a] = sw ? tmax : tmin;
        hi[a] = sw ? tmin : tmax;
      }
      double tmin = fmax(fmax(lo[0], lo[1]), lo[2]);
      double tmax = fmin(fmin(hi[0], hi[1]), hi[2]);
      pc = (tmin <= tmax) ? pc + 1 : op.skip;
      continue;
    }
    if (code == ROP_ENTER) {
      cnt.enters += mult;
      const rl_rtc_transformed &x = P.xforms[op.a];
      D3 no = mul_point(x.inverse, o), nd = mul_vec(x.inverse, d);
      o = no, d = nd;
      if (xf_depth < 8) xf_start[xf_depth] = n;
      xf_depth++;
      cur_enter = pc;
      pc++;
      continue;
    }
    if (code == ROP_EXIT) {  // transformed.rs:43-49: every intersection found inside gets normal <- unit(inverse_transpose * n)
      xf_depth--;
      uint32_t start = xf_depth < 8 ? xf_start[xf_depth] : n;
      const rl_rtc_transformed &x = P.xforms[op.a];
      for (uint32_t i = start; i < n; i++) {
        D3 wn = mul_vec(x.inverse_transpose, list[i].normal), nn;
        if (!norm(wn, nn)) {
          cnt.flagged++;
          nn = list[i].normal;
        }
        list[i].normal = nn;
      }
      cur_enter = op.b;
      rtc_local_ray(P, ops, cur_enter, wo, wd, o, d);  // parent ray: replay the enclosing ENTER chain from the world ray
      pc++;
      continue;
    }
    if (code == ROP_CSG_BEGIN) {
      if (csg_depth < 4) csg_start[csg_depth] = n, csg_mid[csg_depth] = n;
      csg_depth++;
      pc++;
      continue;
    }
    if (code == ROP_CSG_MID) {
      if (csg_depth >= 1 && csg_depth <= 4) csg_mid[csg_depth - 1] = n;
      pc++;
      continue;
    }
    {  // ROP_CSG_END: csg.rs:78-108 — left + right intersections, stable sort by t, filter by the operation
      csg_depth--;
      uint32_t start = csg_depth < 4 ? csg_start[csg_depth] : n, mid = csg_depth < 4 ? csg_mid[csg_depth] : n;
      for (uint32_t i = start; i < n; i++) list[i].leaf = (list[i].leaf & ~ENT_RIGHT) | (i >= mid ? ENT_RIGHT : 0u);
      ent_sort(list, start, n);
      uint32_t operation = F.csgs[op.a].operation;
      bool in_l = false, in_r = false;
      uint32_t w = start;
      for (uint32_t i = start; i < n; i++) {
        bool left = (list[i].leaf & ENT_RIGHT) == 0;
        bool allowed;
        if (operation == RL_CSG_UNION) allowed = (left && !in_r) || (!left && !in_l);
        else if (operation == RL_CSG_INTERSECTION) allowed = (left && in_r) || (!left && in_l);
        else allowed = (left && !in_r) || (!left && in_l);
        if (left) in_l = !in_l;
        else in_r = !in_r;
        if (allowed) {
          if (w != i) list[w] = list[i];
          w++;
        }
      }
      n = w;
      pc++;
      continue;
    }
  }
  for (uint32_t i = 0; i < n; i++) list[i].leaf &= ~ENT_RIGHT;
  ent_sort(list, 0, n);  // intersect::sort at world level (stable)
  return n;
}

// Material::color_at for the hit (material.rs:13-20): the local-space point is o' + d' * t with (o', d') the ray inside the hit's
// innermost Transformed — the same sequence of products that produced it while intersecting, hence the same bits.
__device__ inline D3 rtc_hit_color(const RtcFullParams &F, const DevOp *ops, const rl_rtc_material &m, uint32_t chain, double t, D3 wo, D3 wd) {
  if (m.pattern == 0) return ld3(m.color);
  D3 o, d;
  rtc_local_ray(F.R, ops, chain, wo, wd, o, d);
  return rtc_surface_color(F, m, o + d * t);
}

// containers.last() after the containers walk of intersect.rs:72-99 has processed list[0 .. upto): an object is in `containers` iff it
// occurs an odd number of times in that prefix (second occurrence removes, third appends again ...), and the Vec's order is the order of
// the objects' most recent append = their LAST occurrence.  NONE: empty.
__device__ inline uint32_t rtc_last_container(const Ent *list, uint32_t upto) {
  for (uint32_t j = upto; j-- > 0;) {
    uint32_t leaf = list[j].leaf, count = 0;
    bool later = false;
    for (uint32_t k = 0; k < upto; k++)
      if (list[k].leaf == leaf) count++, later |= k > j;
    if (!later && (count & 1u)) return leaf;
  }
  return NONE;
}

__device__ __forceinline__ bool rtc_are_equal(double a, double b) {  // math/util.rs:4-22
  if (isnan(a) || isnan(b)) return false;
  if (isinf(a) && isinf(b)) return a == b;
  if (fabs(a - b) <= 2.220446049250313e-16 * 2.0) return true;
  unsigned long long au = (unsigned long long)__double_as_longlong(a), bu = (unsigned long long)__double_as_longlong(b);
  unsigned long long diff = au > bu ? au - bu : bu - au;
  return diff <= 8;
}

// Pending reflection / refraction rays of one camera ray, walked depth-first: a ray with `remaining` bounces left pushes at most
// two rays with remaining - 1, so max_reflection_depth + 1 slots are enough (rl_rtc_scene_create rejects deeper worlds with
// RL_E_UNSUPPORTED; the reference recurses without a cap, world.rs:128-159).
static const uint32_t RTC_MAX_PENDING = 8;
struct Pending {
  D3 o, d;
  double w;
  uint32_t remaining;
  unsigned long long mult;  // how many times the reference evaluates this ray (it recomputes reflected / refracted per light)
};

// NT threads per workgroup; REGS_FOR = the workgroup size the register budget is computed for (512 -> 256 registers, two waves per SIMD)
template <int NT, int REGS_FOR>
__global__ void RL_KERNEL_ALIGN __launch_bounds__(REGS_FOR) rtc_full_kernel(RtcFullParams F) {
  const RtcParams &P = F.R;
  const int tid = threadIdx.x;
  const DevOp *ops = P.ops;
  const DevTri *tris = P.tris;
  const rl_rtc_camera &cam = P.cam;
  const uint32_t W = cam.hsize;
  RtcFullCounters cnt{0, 0, 0, 0, 0, 0};
  const uint64_t total = (uint64_t)W * P.nrows;
  Ent list[RL_RTC_K];
  Pending stack[RTC_MAX_PENDING];
  for (uint64_t idx = (uint64_t)blockIdx.x * NT + tid; idx < total; idx += (uint64_t)gridDim.x * NT) {
    uint32_t r = (uint32_t)(idx / W), px = (uint32_t)(idx % W);
    uint32_t py = P.row_first + r * P.row_step;
    D3 acc = d3(0.0, 0.0, 0.0);
    bool have = false;
    for (uint32_t nx = 0; nx < P.aa; nx++)
      for (uint32_t ny = 0; ny < P.aa; ny++) {
        double sample_offset = 1.0 / (double)P.aa;
        double xoffset = ((double)px + sample_offset * ((double)nx + 0.5)) * cam.pixel_size;
        double yoffset = ((double)py + sample_offset * ((double)ny + 0.5)) * cam.pixel_size;
        D3 pixel = mul_point(cam.inverse, d3(cam.half_width - xoffset, cam.half_height - yoffset, -1.0));
        D3 origin = mul_point(cam.inverse, d3(0.0, 0.0, 0.0));
        D3 dir;
        if (!norm(pixel - origin, dir)) {
          cnt.flagged++;
          dir = d3(0.0, 0.0, 0.0);
        }
        D3 c = d3(0.0, 0.0, 0.0);
        int sp = 0;
        stack[sp++] = Pending{origin, dir, 1.0, F.max_reflection_depth, 1ull};
        while (sp > 0) {
          Pending cur = stack[--sp];
          cnt.rays += cur.mult;
          uint32_t n = rtc_intersect_all(F, ops, tris, cur.o, cur.d, list, cnt, cur.mult);
          int hi = -1;  // intersect.rs:159-168: lowest t >= 0, later wins ties
          for (uint32_t i = 0; i < n; i++)
            if (list[i].t >= 0.0 && (hi < 0 || !(list[hi].t < list[i].t))) hi = (int)i;
          if (hi < 0 || P.n_lights == 0) {
            c = c + ld3(P.void_color) * cur.w;
            continue;
          }
          // prepare_computations (intersect.rs:48-115)
          Ent h = list[hi];
          const rl_rtc_material &m = P.materials[rtc_leaf_material(F, tris, h.leaf)];
          D3 point = cur.o + cur.d * h.t;
          D3 eye_v;
          if (!norm(-cur.d, eye_v)) {
            cnt.flagged++;
            eye_v = -cur.d;
          }
          D3 normal_v = h.normal;
          if (dot(normal_v, eye_v) < 0.0) normal_v = -normal_v;
          D3 over_point = point + normal_v * 1e-5;
          D3 under_point = point - normal_v * 1e-5;
          D3 reflect_v;
          if (!norm(reflect(cur.d, normal_v), reflect_v)) {
            cnt.flagged++;
            reflect_v = cur.d;
          }
          double n1 = 1.0, n2 = 1.0;
          if (m.transparency != 0.0) {  // n1 / n2 feed only refracted_color and schlick, both of which need a transparent material
            uint32_t is = 0;
            while (is < n && !(rtc_are_equal(list[is].t, h.t) && list[is].leaf == h.leaf)) is++;
            if (is < n) {
              uint32_t c1 = rtc_last_container(list, is), c2 = rtc_last_container(list, is + 1);
              if (c1 != NONE) n1 = P.materials[rtc_leaf_material(F, tris, c1)].refractive_index;
              if (c2 != NONE) n2 = P.materials[rtc_leaf_material(F, tris, c2)].refractive_index;
            }
          }
          // shade_hit (world.rs:57-87); the list is reused for the shadow rays from here on
          D3 object_color = rtc_hit_color(F, ops, m, h.chain, h.t, cur.o, cur.d);
          D3 lsum = d3(0.0, 0.0, 0.0);
          for (uint32_t li = 0; li < P.n_lights; li++) {
            const rl_rtc_light &light = P.lights[li];
            D3 lpos = ld3(light.position), intensity = ld3(light.intensity);
            D3 v = lpos - over_point;  // shadow_attenuation (world.rs:104-126)
            double distance = mag(v);
            D3 sdir;
            double shadow_att = 1.0;
            if (norm(v, sdir)) {
              cnt.rays += cur.mult;
              uint32_t ns = rtc_intersect_all(F, ops, tris, over_point, sdir, list, cnt, cur.mult);
              for (uint32_t i = 0; i < ns; i++) {
                if (!(list[i].t > 0.0 && list[i].t < distance)) continue;
                bool dup = false;  // take_while(seen.insert): every earlier in-range entry is in `seen`
                for (uint32_t k = 0; k < i; k++) dup |= list[k].t > 0.0 && list[k].t < distance && list[k].leaf == list[i].leaf;
                if (dup) break;
                shadow_att = shadow_att * P.materials[rtc_leaf_material(F, tris, list[i].leaf)].transparency;
              }
            }
            D3 effective = object_color * intensity;  // lighting (material.rs:54-90)
            D3 lightv;
            if (!norm(lpos - point, lightv)) lightv = d3(0.0, 0.0, 0.0);
            D3 ambient = effective * m.ambient;
            double ldn = dot(lightv, normal_v);
            D3 diffuse = d3(0.0, 0.0, 0.0), specular = d3(0.0, 0.0, 0.0);
            if (!(ldn < 0.0)) {
              D3 diff = (effective * m.diffuse) * ldn;
              D3 reflectv = -reflect(lightv, normal_v);
              double rde = dot(reflectv, eye_v);
              diffuse = diff * shadow_att;
              if (!(rde <= 0.0)) {
                double factor = pow(rde, m.shininess);
                specular = intensity * (m.specular * factor * shadow_att);
              }
            }
            D3 surface = (ambient + diffuse) + specular;
            lsum = (li == 0) ? surface : lsum + surface;
          }
          c = c + lsum * cur.w;
          // reflected_color / refracted_color (world.rs:128-159), evaluated once and weighted by n_lights
          double wl = cur.w * (double)P.n_lights;
          bool both = m.reflectivity > 0.0 && m.transparency > 0.0;
          double reflectance = 1.0;
          if (both) {  // Precomputation::schlick (intersect.rs:139-156)
            double cosv = dot(eye_v, normal_v);
            double nn = n1 / n2;
            double sin2_t = nn * nn * (1.0 - cosv * cosv);
            double cos_t = sqrt(1.0 - sin2_t);
            double cos_adj = nn > 1.0 ? cos_t : cosv;
            if (sin2_t > 1.0 && nn > 1.0) reflectance = 1.0;
            else {
              double q = (n1 - n2) / (n1 + n2);
              double r0 = q * q;
              double x = 1.0 - cos_adj;
              double x2 = x * x;
              reflectance = r0 + (1.0 - r0) * (x * (x2 * x2));
            }
          }
          if (cur.remaining > 0 && m.transparency != 0.0) {
            double n_ratio = n1 / n2;
            double cos_i = dot(eye_v, normal_v);
            double sin2_t = n_ratio * n_ratio * (1.0 - cos_i * cos_i);
            if (!(sin2_t > 1.0) && sp >= (int)RTC_MAX_PENDING) cnt.flagged++;  // cannot happen for a world rl_rtc_scene_create accepted
            else if (!(sin2_t > 1.0)) {
              double cos_t = sqrt(1.0 - sin2_t);
              D3 direction = normal_v * (n_ratio * cos_i - cos_t) - eye_v * n_ratio;
              double wt = wl * m.transparency * (both ? (1.0 - reflectance) : 1.0);
              stack[sp++] = Pending{under_point, direction, wt, cur.remaining - 1, cur.mult * P.n_lights};
            }
          }
          if (cur.remaining > 0 && m.reflectivity != 0.0 && sp >= (int)RTC_MAX_PENDING) cnt.flagged++;
          else if (cur.remaining > 0 && m.reflectivity != 0.0) {
            double wr = wl * m.reflectivity * (both ? reflectance : 1.0);
            stack[sp++] = Pending{over_point, reflect_v, wr, cur.remaining - 1, cur.mult * P.n_lights};
          }
        }
        acc = have ? acc + c : c;
        have = true;
      }
    D3 res = acc * (1.0 / (double)((uint64_t)P.aa * P.aa));
    double *outp = P.out + idx * 3;
    outp[0] = res.x, outp[1] = res.y, outp[2] = res.z;
  }
  unsigned long long v;
  v = wave_sum(cnt.rays);
  if ((tid & 63) == 0) atomicAdd(&P.stats[0], v);
  v = wave_sum(cnt.nodes);
  if ((tid & 63) == 0) atomicAdd(&P.stats[1], v);
  v = wave_sum(cnt.spheres);
  if ((tid & 63) == 0) atomicAdd(&P.stats[2], v);
  v = wave_sum(cnt.tris);
  if ((tid & 63) == 0) atomicAdd(&P.stats[3], v);
  v = wave_sum(cnt.enters);
  if ((tid & 63) == 0) atomicAdd(&P.stats[4], v);
  v = wave_sum(cnt.flagged);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[6], v);
}

}  // namespace rl
