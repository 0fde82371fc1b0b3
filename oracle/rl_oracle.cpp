// rl_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT.
//
// CPU restatement of the reference's per-pixel / per-ray hot path, used ONLY as the checker by
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product path (the HIP library
// behind include/rl_render.h) never links, loads or calls anything in this directory.
//
// Parity status: PINNED.  The Rust reference cannot be compiled here (no cargo/rustc; crates not
// vendored), so this restatement is pinned by the reference's own golden images:
//   * ray-tracing-one-weekend/tests/expectations/test.ppm          (tests/ray_tracing_one_weekend.rs:77-95)
//   * ray-tracer-challenge/tests/expectations/test_obj_scene.ppm   (tests/ray_tracer.rs:242-275)
// byte-for-byte (tests/test_oracle_golden.py), plus the unit known-answers of SURVEY.md §4.
// Third-party arithmetic absent from /root/reference and restated from its published algorithm:
//   rand_chacha 0.3.1 / rand_core 0.6.4 (ChaCha8, seed_from_u64, set_stream), rand 0.8.5 (Standard f64,
//   Uniform<f64>), rand_distr 0.4.3 (UnitDisc, UnitSphere), float-cmp 0.9.0 (approx_eq) — all exercised
//   by the RTIOW golden.
//
// It consumes the same POD scene descriptors as the product (include/rl_render.h) but evaluates them
// the way the reference does: recursive Hittable::hit / Object::intersect, heap Vec + stable sort for
// RTC, recursive ray_color with the reference's unwind-multiply order.  Compile with
//   g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math   (see oracle/Makefile)
// Reference file:line cited at each function; paths relative to /root/reference/.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../include/rl_render.h"

#define RTIOW "ray-tracing-one-weekend/src/"
#define RTC "ray-tracer-challenge/src/"

namespace {

struct V3 {
  double x, y, z;
};
inline V3 v3(const double *p) { return V3{p[0], p[1], p[2]}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vec3.rs:44
inline double len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }        // vec3.rs:36
inline V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 div_s(V3 a, double s) { return a * (1.0 / s); }                     // vec3.rs:177-179
inline V3 normalize(V3 a) { return div_s(a, std::sqrt(len2(a))); }           // vec3.rs:56

// ============================================================== ChaCha8 stream RNG
// rand_chacha 0.3.1 ChaCha8Rng as used at RTIOW camera.rs:161-170.
struct ChaCha8 {
  uint32_t key[8];
  uint64_t stream = 0, pos = 0;  // pos = u32 word position
  uint32_t buf[16];
  uint64_t buf_ctr = ~0ull, buf_stream = ~0ull;
  bool buf_valid = false;
  uint64_t words_drawn = 0;

  static uint32_t rotr32(uint32_t x, unsigned r) { return (x >> (r & 31)) | (x << ((32 - r) & 31)); }
  static uint32_t rotl32(uint32_t x, unsigned r) { return (x << r) | (x >> (32 - r)); }
  // rand_core 0.6.4 SeedableRng::seed_from_u64: PCG32 fills the 32-byte seed
  void seed_from_u64(uint64_t state) {
    const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
    for (int k = 0; k < 8; k++) {
      state = state * MUL + INC;
      uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
      uint32_t rot = (uint32_t)(state >> 59);
      key[k] = rotr32(xorshifted, rot);
    }
    stream = 0, pos = 0, buf_valid = false;
  }
  static void block(const uint32_t key[8], uint64_t ctr, uint64_t stream, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      (uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t x[16];
    std::memcpy(x, s, sizeof x);
#define QR(a, b, c, d)                  \
  x[a] += x[b], x[d] ^= x[a], x[d] = rotl32(x[d], 16); \
  x[c] += x[d], x[b] ^= x[c], x[b] = rotl32(x[b], 12); \
  x[a] += x[b], x[d] ^= x[a], x[d] = rotl32(x[d], 8);  \
  x[c] += x[d], x[b] ^= x[c], x[b] = rotl32(x[b], 7);
    for (int r = 0; r < 4; r++) {
      QR(0, 4, 8, 12) QR(1, 5, 9, 13) QR(2, 6, 10, 14) QR(3, 7, 11, 15)
      QR(0, 5, 10, 15) QR(1, 6, 11, 12) QR(2, 7, 8, 13) QR(3, 4, 9, 14)
    }
#undef QR
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
  }
  void set_stream(uint64_t s) { stream = s; }  // keeps pos (camera.rs:170)
  uint32_t next_u32() {
    uint64_t ctr = pos >> 4;
    if (!buf_valid || buf_ctr != ctr || buf_stream != stream) {
      block(key, ctr, stream, buf);
      buf_ctr = ctr, buf_stream = stream, buf_valid = true;
    }
    uint32_t w = buf[pos & 15];
    pos++;
    words_drawn++;
    return w;
  }
  uint64_t next_u64() {
    uint64_t lo = next_u32();
    uint64_t hi = next_u32();
    return lo | (hi << 32);
  }
  double gen_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }  // rand 0.8.5 Standard
  double uniform_m1_1() {  // rand 0.8.5 Uniform::new(-1.0, 1.0).sample: (v12 - 1.0) * scale + low, scale = 2
    uint64_t bits = (next_u64() >> 12) | 0x3FF0000000000000ull;
    double v;
    std::memcpy(&v, &bits, 8);
    return (v - 1.0) * 2.0 + (-1.0);
  }
  void unit_disc(double &a, double &b) {  // rand_distr 0.4.3 UnitDisc (accept <= 1)
    for (;;) {
      a = uniform_m1_1();
      b = uniform_m1_1();
      if (a * a + b * b <= 1.0) return;
    }
  }
  V3 unit_sphere() {  // rand_distr 0.4.3 UnitSphere (Marsaglia; reject >= 1)
    for (;;) {
      double x1 = uniform_m1_1(), x2 = uniform_m1_1();
      double sum = x1 * x1 + x2 * x2;
      if (sum >= 1.0) continue;
      double factor = 2.0 * std::sqrt(1.0 - sum);
      return V3{x1 * factor, x2 * factor, 1.0 - 2.0 * sum};
    }
  }
};

// ============================================================== RTIOW
struct Ray {
  V3 o, d;
  double time;
  V3 at(double t) const { return o + d * t; }  // ray.rs:29
};
struct HitRecord {  // hittable/mod.rs:24-30
  V3 p, normal;
  double t, u, v;
  bool front;
  uint32_t mat;
};
struct Counters {
  uint64_t rays = 0, node_tests = 0, sphere_tests = 0, planar_tests = 0, instance_enters = 0, rng_words = 0, flagged = 0;
};
struct ChaCha8;
struct RtiowCtx {
  const rl_rtiow_scene_desc *d;
  Counters c;
  ChaCha8 *rng = nullptr;  // the pixel's stream: only ConstantMedium::hit (deterministic variant, rl_render.h rl_medium) draws while traversing
};
double medium_draw(RtiowCtx &cx);

// float-cmp approx_eq(a, b, F64Margin{epsilon, ulps}) == (a==b || |a-b|<=eps || ulps_diff<=ulps)
inline bool approx_eq_eps(double a, double b, double eps) { return a == b || std::fabs(a - b) <= eps; }

// NormalizedVec3::try_from (vec3.rs:236-247): None when |v|^2 ~ 0 within 1e-16
inline bool try_normalize(V3 v, V3 &out) {
  double m = len2(v);
  if (approx_eq_eps(m, 0.0, 1e-16)) return false;
  out = normalize(v);
  return true;
}

// aabb.rs:143-152
inline void intersect_axis(double mn, double mx, double origin, double speed, double &lo, double &hi) {
  double t0 = (mn - origin) / speed;
  double t1 = (mx - origin) / speed;
  if (t0 < t1) lo = t0, hi = t1;
  else lo = t1, hi = t0;
}
// aabb.rs:123-132 (f64::max/min ignore NaN == fmax/fmin)
inline bool aabb_hit(const double b[6], const Ray &r, double tmin_, double tmax_) {
  double xl, xh, yl, yh, zl, zh;
  intersect_axis(b[0], b[1], r.o.x, r.d.x, xl, xh);
  intersect_axis(b[2], b[3], r.o.y, r.d.y, yl, yh);
  intersect_axis(b[4], b[5], r.o.z, r.d.z, zl, zh);
  double tmin = std::fmax(std::fmax(std::fmax(xl, yl), zl), tmin_);
  double tmax = std::fmin(std::fmin(std::fmin(xh, yh), zh), tmax_);
  return tmin < tmax;
}

bool hit_href(RtiowCtx &cx, rl_href h, const Ray &r, double tmin, double tmax, HitRecord &rec);

// hittable/mod.rs:88-105: fold with shrinking max; a later hit replaces
bool hit_slice(RtiowCtx &cx, const rl_href *items, uint32_t n, const Ray &r, double tmin, double tmax, HitRecord &rec) {
  bool any = false;
  double closest = tmax;
  HitRecord tmp;
  for (uint32_t i = 0; i < n; i++)
    if (hit_href(cx, items[i], r, tmin, closest, tmp)) {
      any = true;
      closest = tmp.t;
      rec = tmp;
    }
  return any;
}

// hittable/mod.rs:32-38
inline void face_normal(const Ray &r, V3 outward, V3 &normal, bool &front) {
  if (dot(r.d, outward) <= 0.0) normal = outward, front = true;
  else normal = -outward, front = false;
}

void sphere_uv(V3 p, double &u, double &v) {  // sphere.rs:91-99 get_sphere_uv (host libm acos / atan2: consumed by Image textures only, colour-only)
  const double PI = 3.14159265358979323846;
  double theta = std::acos(-p.y);
  double phi = std::atan2(-p.z, p.x) + PI;
  u = phi / (2.0 * PI);
  v = theta / PI;
}

bool hit_sphere(RtiowCtx &cx, const rl_sphere &s, const Ray &r, double tmin, double tmax, HitRecord &rec) {  // sphere.rs:32-75
  cx.c.sphere_tests++;
  V3 c0 = v3(s.center0);
  V3 center = s.moving ? c0 + (v3(s.center1) - c0) * r.time : c0;  // sphere.rs:24-29
  V3 oc = r.o - center;
  double a = len2(r.d);
  double half_b = dot(oc, r.d);
  double c = len2(oc) - s.radius * s.radius;
  double disc = half_b * half_b - a * c;
  if (disc < 0.0) return false;
  double sq = std::sqrt(disc);
  double r_l = (-half_b - sq) / a;
  double r_u = (-half_b + sq) / a;
  double t;
  if (tmin <= r_l && r_l <= tmax) t = r_l;
  else if (tmin <= r_u && r_u <= tmax) t = r_u;
  else return false;
  V3 p = r.at(t);
  V3 outward = div_s(p - center, s.radius);
  // NormalizedVec3::from_normalized asserts |n|^2 ~ 1 (eps 1e-5, 4 ulps) — a panic site (vec3.rs:219-222)
  double l2 = len2(outward);
  if (!(l2 == 1.0 || std::fabs(l2 - 1.0) <= 1e-5)) cx.c.flagged++;
  face_normal(r, outward, rec.normal, rec.front);
  rec.p = p, rec.t = t, rec.mat = s.material;
  sphere_uv(outward, rec.u, rec.v);  // sphere.rs:70
  return true;
}

bool hit_planar(RtiowCtx &cx, const rl_planar &pl, const Ray &r, double tmin, double tmax, HitRecord &rec) {  // plane.rs:51-100
  cx.c.planar_tests++;
  V3 normal = v3(pl.normal);
  double denom = dot(normal, r.d);
  if (std::fabs(denom) < 1e-8) return false;
  double t = (pl.d - dot(normal, r.o)) / denom;
  if (!(tmin <= t && t <= tmax)) return false;
  V3 p = r.at(t);
  V3 hp = p - v3(pl.q);
  double alpha = dot(v3(pl.w), cross(hp, v3(pl.v)));
  double beta = dot(v3(pl.w), cross(v3(pl.u), hp));
  face_normal(r, normal, rec.normal, rec.front);
  rec.p = p, rec.t = t, rec.u = alpha, rec.v = beta, rec.mat = pl.material;
  if (pl.kind == RL_PLANAR_QUAD) {  // quad.rs:37-42
    return 0.0 <= alpha && alpha <= 1.0 && 0.0 <= beta && beta <= 1.0;
  } else if (pl.kind == RL_PLANAR_TRIANGLE) {  // triangle.rs:60-95
    if (!(0.0 <= alpha && 0.0 <= beta && alpha + beta <= 1.0)) return false;
    double frac2 = alpha, frac3 = beta, frac1 = 1.0 - alpha - beta;
    if (pl.has_normals) {
      V3 n = (v3(pl.normals + 3) * frac2 + v3(pl.normals + 6) * frac3) + v3(pl.normals) * frac1;
      V3 nn;
      if (!try_normalize(n, nn)) {
        cx.c.flagged++;
        nn = normal;
      }
      face_normal(r, nn, rec.normal, rec.front);
    }
    if (pl.has_uvs) {
      rec.u = pl.uvs[0] * frac1 + pl.uvs[2] * frac2 + pl.uvs[4] * frac3;
      rec.v = pl.uvs[1] * frac1 + pl.uvs[3] * frac2 + pl.uvs[5] * frac3;
    }
    return true;
  }
  return true;
}

inline V3 mat3_mul(const double m[9], V3 v) {  // matrix.rs:42-60 (accumulate from 0.0)
  double o[3];
  double vd[3] = {v.x, v.y, v.z};
  for (int n = 0; n < 3; n++) {
    double sum = 0.0;
    for (int k = 0; k < 3; k++) sum += m[3 * n + k] * vd[k];
    o[n] = sum;
  }
  return V3{o[0], o[1], o[2]};
}

bool hit_href(RtiowCtx &cx, rl_href h, const Ray &r, double tmin, double tmax, HitRecord &rec) {
  const rl_rtiow_scene_desc &d = *cx.d;
  switch (h.kind) {
    case RL_H_SPHERE:
      return hit_sphere(cx, d.spheres[h.index], r, tmin, tmax, rec);
    case RL_H_PLANAR:
      return hit_planar(cx, d.planars[h.index], r, tmin, tmax, rec);
    case RL_H_LIST: {
      const rl_list &l = d.lists[h.index];
      return hit_slice(cx, d.list_items + l.first, l.count, r, tmin, tmax, rec);
    }
    case RL_H_BVH: {  // bvh.rs:79-95
      const rl_bvh_node &n = d.bvh_nodes[h.index];
      cx.c.node_tests++;
      if (!aabb_hit(n.bbox, r, tmin, tmax)) return false;
      return hit_slice(cx, n.child, n.n_children, r, tmin, tmax, rec);
    }
    case RL_H_MEDIUM: {  // constant_medium.rs:27-80; the free path is drawn from the pixel's stream instead of rand::random (:55)
      if (h.index >= d.n_media || !cx.rng) return false;
      const rl_medium &m = d.media[h.index];
      const double INF = std::numeric_limits<double>::infinity();
      HitRecord rec1, rec2;
      if (!hit_href(cx, m.boundary, r, -INF, INF, rec1)) return false;              // boundary.hit(r, &Interval::universe())
      if (!hit_href(cx, m.boundary, r, rec1.t + 1e-4, INF, rec2)) return false;     // ... min: rec1.t + 1e-4
      double t1 = std::fmax(rec1.t, tmin), t2 = std::fmin(rec2.t, tmax);             // f64::max / f64::min
      if (t1 >= t2) return false;
      t1 = std::fmax(t1, 0.0);
      double ray_length = std::sqrt(len2(r.d));                                      // Vec3::length
      double distance_inside_boundary = (t2 - t1) * ray_length;
      double hit_distance = m.neg_inv_density * std::log(medium_draw(cx));
      if (hit_distance > distance_inside_boundary) return false;
      double t = t1 + hit_distance / ray_length;
      rec.t = t, rec.p = r.at(t), rec.normal = V3{1.0, 0.0, 0.0}, rec.u = 0.0, rec.v = 0.0, rec.front = true, rec.mat = m.material;
      return true;
    }
    case RL_H_TRANSLATE: {  // translate.rs:14-21
      const rl_translate &t = d.translates[h.index];
      cx.c.instance_enters++;
      Ray r2{r.o - v3(t.offset), r.d, r.time};
      if (!hit_href(cx, t.child, r2, tmin, tmax, rec)) return false;
      rec.p = rec.p + v3(t.offset);
      return true;
    }
    case RL_H_TRANSFORM: {  // transform.rs:143-164 (t and face are NOT recomputed)
      const rl_transform &t = d.transforms[h.index];
      cx.c.instance_enters++;
      Ray r2{mat3_mul(t.inv, r.o), mat3_mul(t.inv, r.d), r.time};
      if (!hit_href(cx, t.child, r2, tmin, tmax, rec)) return false;
      rec.p = mat3_mul(t.m, rec.p);
      V3 nn;
      if (!try_normalize(mat3_mul(t.inv_t, rec.normal), nn)) {
        cx.c.flagged++;
        nn = rec.normal;
      }
      rec.normal = nn;
      return true;
    }
  }
  return false;
}

int32_t f64_as_i32(double x) {  // Rust `as i32`: saturating, NaN -> 0
  if (std::isnan(x)) return 0;
  if (x >= 2147483647.0) return INT32_MAX;
  if (x <= -2147483648.0) return INT32_MIN;
  return (int32_t)x;
}

double perlin_noise(const rl_perlin &pn, V3 p) {  // perlin.rs:39-66 + perlin_interp :101-125
  double u = p.x - std::floor(p.x), v = p.y - std::floor(p.y), w = p.z - std::floor(p.z);
  uint32_t i = (uint32_t)f64_as_i32(std::floor(p.x)), j = (uint32_t)f64_as_i32(std::floor(p.y)), k = (uint32_t)f64_as_i32(std::floor(p.z));
  const double *c[2][2][2];
  for (uint32_t di = 0; di < 2; di++)
    for (uint32_t dj = 0; dj < 2; dj++)
      for (uint32_t dk = 0; dk < 2; dk++)
        c[di][dj][dk] = pn.randvec[(pn.perm_x[(i + di) & 255u] ^ pn.perm_y[(j + dj) & 255u] ^ pn.perm_z[(k + dk) & 255u]) & 255u];
  double uu = u * u * (3.0 - 2.0 * u);
  double vv = v * v * (3.0 - 2.0 * v);
  double ww = w * w * (3.0 - 2.0 * w);
  double accum = 0.0;
  for (int a = 0; a < 2; a++)
    for (int b = 0; b < 2; b++)
      for (int e = 0; e < 2; e++) {
        double i_f = (double)a, j_f = (double)b, k_f = (double)e;
        V3 weight_v{u - i_f, v - j_f, w - k_f};
        accum += (i_f * uu + (1.0 - i_f) * (1.0 - uu)) * (j_f * vv + (1.0 - j_f) * (1.0 - vv)) * (k_f * ww + (1.0 - k_f) * (1.0 - ww)) *
                 dot(v3(c[a][b][e]), weight_v);
      }
  return accum;
}

double perlin_turb(const rl_perlin &pn, V3 p, uint32_t depth) {  // perlin.rs:68-80
  double accum = 0.0, weight = 1.0;
  V3 temp_p = p;
  for (uint32_t it = 0; it < depth; it++) {
    accum += weight * perlin_noise(pn, temp_p);
    weight *= 0.5;
    temp_p = temp_p * 2.0;
  }
  return std::fabs(accum);
}

V3 texture_value(const rl_rtiow_scene_desc &d, uint32_t tex, double u, double v, V3 p) {  // texture.rs
  const rl_texture &t = d.textures[tex];
  switch (t.kind) {
    case RL_TEX_SOLID:
      return v3(t.color);
    case RL_TEX_CHECKER: {  // texture.rs:41-55; `as i64` saturates, Rust % keeps the sign
      auto to_i64 = [](double f) -> int64_t {
        if (std::isnan(f)) return 0;
        if (f >= 9223372036854775807.0) return INT64_MAX;
        if (f <= -9223372036854775808.0) return INT64_MIN;
        return (int64_t)f;
      };
      int64_t xi = to_i64(std::floor(p.x * t.inv_scale));
      int64_t yi = to_i64(std::floor(p.y * t.inv_scale));
      int64_t zi = to_i64(std::floor(p.z * t.inv_scale));
      int64_t sum = (int64_t)((uint64_t)xi + (uint64_t)yi + (uint64_t)zi);
      bool is_even = (sum % 2) == 0;
      return texture_value(d, is_even ? t.even : t.odd, u, v, p);
    }
    case RL_TEX_NOISE: {  // texture.rs:84-94
      double sv = 1.0 + std::sin(t.inv_scale * p.z + 10.0 * perlin_turb(d.perlins[t.image], p, 7));
      return V3{0.5 * sv, 0.5 * sv, 0.5 * sv};
    }
    case RL_TEX_IMAGE: {  // texture.rs:62-82
      const rl_image &im = d.images[t.image];
      double uu = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);  // f64::clamp (NaN stays NaN)
      double vc = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
      double vv = 1.0 - vc;
      auto to_u32 = [](double f) -> uint32_t {
        if (!(f > 0.0)) return 0;  // NaN and negatives -> 0 (saturating cast)
        if (f >= 4294967295.0) return 4294967295u;
        return (uint32_t)f;
      };
      uint32_t i = to_u32(uu * (double)(im.width - 1));
      uint32_t j = to_u32(vv * (double)(im.height - 1));
      const float *px = im.rgb + ((size_t)j * im.width + i) * 3;
      return V3{(double)px[0], (double)px[1], (double)px[2]};
    }
  }
  return V3{0, 0, 0};
}

// material.rs: scatter(); returns false when no scatter. emitted via emitted().
bool scatter(RtiowCtx &cx, ChaCha8 &rng, const rl_material &m, const Ray &ray, const HitRecord &rec, V3 &att, Ray &out) {
  switch (m.kind) {
    case RL_MAT_LAMBERTIAN: {  // material.rs:74-92
      V3 dir = rec.normal + rng.unit_sphere();
      bool near_zero = approx_eq_eps(dir.x, 0.0, 1e-8) && approx_eq_eps(dir.y, 0.0, 1e-8) && approx_eq_eps(dir.z, 0.0, 1e-8);  // vec3.rs:60-65
      if (near_zero) dir = rec.normal;
      out = Ray{rec.p, dir, ray.time};
      att = texture_value(*cx.d, m.texture, rec.u, rec.v, rec.p);
      return true;
    }
    case RL_MAT_METAL: {  // material.rs:105-122; reflect vec3.rs:67-69
      V3 reflected = ray.d - rec.normal * (2.0 * dot(ray.d, rec.normal));
      V3 fuzzed = normalize(reflected) + rng.unit_sphere() * m.fuzz;
      out = Ray{rec.p, fuzzed, ray.time};
      if (dot(fuzzed, rec.normal) > 0.0) {
        att = v3(m.albedo);
        return true;
      }
      return false;
    }
    case RL_MAT_DIELECTRIC: {  // material.rs:139-165
      double ri = rec.front ? 1.0 / m.ior : m.ior;
      V3 ud;
      if (!try_normalize(ray.d, ud)) {  // "How did the incident ray have magnitude 0?" (material.rs:151)
        cx.c.flagged++;
        ud = ray.d;
      }
      double cos_theta = std::fmin(dot(-ud, rec.normal), 1.0);
      double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
      bool cannot_refract = ri * sin_theta > 1.0;
      bool reflect;
      if (cannot_refract) reflect = true;
      else {  // reflectance material.rs:173-176; powi(2)=x*x, powi(5)=x*((x*x)*(x*x))
        double q = (1.0 - ri) / (1.0 + ri);
        double r0 = q * q;
        double x = 1.0 - cos_theta;
        double x2 = x * x;
        double refl = r0 + (1.0 - r0) * (x * (x2 * x2));
        reflect = refl > rng.gen_f64();
      }
      V3 dir;
      if (reflect) dir = ud - rec.normal * (2.0 * dot(ud, rec.normal));
      else {  // refract vec3.rs:224-230
        double c = std::fmin(dot(-ud, rec.normal), 1.0);
        V3 perp = (ud + rec.normal * c) * ri;
        V3 par = rec.normal * (-std::sqrt(std::fabs(1.0 - len2(perp))));
        dir = perp + par;
      }
      out = Ray{rec.p, dir, ray.time};
      att = V3{1.0, 1.0, 1.0};
      return true;
    }
    case RL_MAT_ISOTROPIC: {  // material.rs:201-214
      out = Ray{rec.p, rng.unit_sphere(), ray.time};
      att = texture_value(*cx.d, m.texture, rec.u, rec.v, rec.p);
      return true;
    }
    default:  // Flat, DiffuseLight: no scatter
      return false;
  }
}

V3 emitted(RtiowCtx &cx, const rl_material &m, const HitRecord &rec) {
  if (m.kind == RL_MAT_DIFFUSE_LIGHT) return texture_value(*cx.d, m.texture, rec.u, rec.v, rec.p);  // material.rs:192-194
  return V3{0.0, 0.0, 0.0};
}

// camera.rs:232-260 — recursive, same unwind order as the reference
V3 ray_color(RtiowCtx &cx, const rl_rtiow_camera &cam, ChaCha8 &rng, const Ray &r, uint32_t depth) {
  if (depth == 0) return V3{0.0, 0.0, 0.0};
  cx.c.rays++;
  HitRecord rec;
  if (hit_href(cx, cx.d->root, r, 1e-10, std::numeric_limits<double>::infinity(), rec)) {
    const rl_material &m = cx.d->materials[rec.mat];
    V3 em = emitted(cx, m, rec);
    V3 att;
    Ray sc;
    if (scatter(cx, rng, m, r, rec, att, sc)) {
      V3 from_scatter = att * ray_color(cx, cam, rng, sc, depth - 1);
      return em + from_scatter;
    }
    return em;
  }
  return v3(cam.background);
}

double medium_draw(RtiowCtx &cx) { return cx.rng->gen_f64(); }

void rtiow_pixel(RtiowCtx &cx, const rl_rtiow_camera &cam, uint64_t first_sample, uint32_t i, uint32_t j, double out[3]) {
  // camera.rs:160-175
  ChaCha8 rng;
  rng.seed_from_u64(cam.seed);
  cx.rng = &rng;
  const uint64_t W = cam.image_width, H = cam.image_height;
  V3 p00 = v3(cam.pixel_00), du = v3(cam.pixel_du), dv = v3(cam.pixel_dv);
  V3 sum{0.0, 0.0, 0.0};
  for (uint64_t n = 0; n < cam.samples_per_pixel; n++) {
    uint64_t sample_index = n + first_sample;
    uint64_t stream_index = sample_index * W * H + (uint64_t)i * W + (uint64_t)j;  // camera.rs:167-169 (i*W, sic)
    rng.set_stream(stream_index);
    // get_ray camera.rs:203-216
    V3 pixel_center = (p00 + du * (double)i) + dv * (double)j;
    double px = -0.5 + rng.gen_f64();
    double py = -0.5 + rng.gen_f64();
    V3 pixel_sample = pixel_center + (du * px + dv * py);
    V3 origin;
    if (cam.defocus_angle <= 0.0) origin = v3(cam.lookfrom);
    else {
      double a, b;
      rng.unit_disc(a, b);
      origin = (v3(cam.lookfrom) + v3(cam.defocus_disk_u) * a) + v3(cam.defocus_disk_v) * b;
    }
    V3 dir = pixel_sample - origin;
    double time = rng.gen_f64();
    Ray r{origin, dir, time};
    V3 c = ray_color(cx, cam, rng, r, cam.max_depth);
    sum = sum + c;
  }
  cx.c.rng_words += rng.words_drawn;
  cx.rng = nullptr;
  out[0] = sum.x, out[1] = sum.y, out[2] = sum.z;
}

// ============================================================== RTC
struct P3 {
  double x, y, z;
};
inline V3 sub(P3 a, P3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline P3 padd(P3 a, V3 b) { return P3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline P3 psub(P3 a, V3 b) { return P3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline double mag(V3 v) { return std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }  // math/vector.rs:32
inline bool norm(V3 v, V3 &out) {  // vector.rs:36-43 / 142-156: true division
  double m = mag(v);
  if (m == 0.0) return false;
  out = V3{v.x / m, v.y / m, v.z / m};
  return true;
}
inline V3 reflect(V3 v, V3 n) { return v - (n * 2.0) * dot(v, n); }  // vector.rs:57-59

struct RRay {
  P3 o;
  V3 d;
};
struct Isect {  // scene/intersect.rs:11-16
  double t;
  uint32_t object;  // triangle index: object identity
  V3 color, normal;
};
struct RtcCtx {
  const rl_rtc_scene_desc *d;
  Counters c;
};

inline P3 mul_point(const double m[16], P3 p) {  // point.rs:89-96, matrix.rs:192-210; w forced to 1
  double v[4] = {p.x, p.y, p.z, 1.0}, o[3];
  for (int n = 0; n < 3; n++) {
    double sum = 0.0;
    for (int i = 0; i < 4; i++) sum += m[4 * n + i] * v[i];
    o[n] = sum;
  }
  return P3{o[0], o[1], o[2]};
}
inline V3 mul_vec(const double m[16], V3 p) {  // vector.rs:115-122; w forced to 0
  double v[4] = {p.x, p.y, p.z, 0.0}, o[3];
  for (int n = 0; n < 3; n++) {
    double sum = 0.0;
    for (int i = 0; i < 4; i++) sum += m[4 * n + i] * v[i];
    o[n] = sum;
  }
  return V3{o[0], o[1], o[2]};
}

inline void sort_xs(std::vector<Isect> &xs) {  // intersect.rs:170-172 (stable)
  std::stable_sort(xs.begin(), xs.end(), [](const Isect &a, const Isect &b) { return a.t < b.t; });
}

void intersect_oref(RtcCtx &cx, rl_oref o, const RRay &ray, std::vector<Isect> &out);

// leaf identity: triangles first, then shapes (object identity in intersect.rs:72-99, world.rs:117)
inline uint32_t leaf_material(const rl_rtc_scene_desc &d, uint32_t leaf) {
  return leaf < d.n_triangles ? d.triangles[leaf].material : d.shapes[leaf - d.n_triangles].material;
}
inline int64_t f2i64(double f) {  // Rust `as i64`: saturating, NaN -> 0
  if (std::isnan(f)) return 0;
  if (f >= 9223372036854775807.0) return INT64_MAX;
  if (f <= -9223372036854775808.0) return INT64_MIN;
  return (int64_t)f;
}
// Surface::color_at (material.rs:13-20) -> Pattern::at (pattern/mod.rs:9-11) -> at_local
V3 surface_color_at(const rl_rtc_scene_desc &d, const rl_rtc_material &m, P3 p);

// build_basic_intersection (object/mod.rs:20-32): colour and normal are evaluated at the LOCAL-space point
void emit_shape_hit(RtcCtx &cx, const rl_rtc_shape &sh, uint32_t leaf, const RRay &ray, double t, V3 normal, std::vector<Isect> &out) {
  P3 p = padd(ray.o, ray.d * t);
  out.push_back(Isect{t, leaf, surface_color_at(*cx.d, cx.d->materials[sh.material], p), normal});
}
inline V3 nrm_or_flag(RtcCtx &cx, V3 v) {  // NormalizedVec3d::new(..).unwrap()
  V3 n;
  if (!norm(v, n)) {
    cx.c.flagged++;
    return V3{0, 0, 0};
  }
  return n;
}
inline bool in_bounds(const rl_rtc_shape &s, double y) {  // cylinder.rs:21-28 / cone.rs:21-28 (strict)
  if (s.has_minimum && s.has_maximum) return y > s.minimum && y < s.maximum;
  if (s.has_minimum) return y > s.minimum;
  if (s.has_maximum) return y < s.maximum;
  return true;
}

void intersect_shape(RtcCtx &cx, uint32_t index, const RRay &ray, std::vector<Isect> &out) {
  const rl_rtc_scene_desc &d = *cx.d;
  const rl_rtc_shape &sh = d.shapes[index];
  uint32_t leaf = d.n_triangles + index;
  const double EPS = 1e-8;
  double ox = ray.o.x, oy = ray.o.y, oz = ray.o.z, dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
  auto point_at = [&](double t) { return padd(ray.o, ray.d * t); };
  switch (sh.kind) {
    case RL_O_SPHERE: {  // sphere.rs:36-60
      cx.c.sphere_tests++;
      V3 s2r = sub(ray.o, P3{0.0, 0.0, 0.0});
      double a = dot(ray.d, ray.d);
      double b = 2.0 * dot(ray.d, s2r);
      double c = dot(s2r, s2r) - 1.0;
      double disc = b * b - 4.0 * a * c;
      if (disc < 0.0) return;
      double sq = std::sqrt(disc);
      double ts[2] = {(-b - sq) / (2.0 * a), (-b + sq) / (2.0 * a)};
      for (double t : ts) {
        P3 p = point_at(t);
        emit_shape_hit(cx, sh, leaf, ray, t, nrm_or_flag(cx, sub(p, P3{0.0, 0.0, 0.0})), out);
      }
      return;
    }
    case RL_O_PLANE: {  // plane.rs:27-42
      cx.c.planar_tests++;
      if (std::fabs(dy) < 1e-8) return;
      double t = -oy / dy;
      emit_shape_hit(cx, sh, leaf, ray, t, V3{0.0, 1.0, 0.0}, out);
      return;
    }
    case RL_O_CUBE: {  // cube.rs:37-78
      cx.c.planar_tests++;
      auto axis = [](double origin, double direction, double &lo, double &hi) {
        double tmin = (-1.0 - origin) / direction;
        double tmax = (1.0 - origin) / direction;
        if (tmin > tmax) lo = tmax, hi = tmin;
        else lo = tmin, hi = tmax;
      };
      double xl, xh, yl, yh, zl, zh;
      axis(ox, dx, xl, xh), axis(oy, dy, yl, yh), axis(oz, dz, zl, zh);
      double tmin = std::fmax(std::fmax(xl, yl), zl);
      double tmax = std::fmin(std::fmin(xh, yh), zh);
      if (tmin > tmax) return;
      double ts[2] = {tmin, tmax};
      for (double t : ts) {  // normal_at cube.rs:15-30
        P3 p = point_at(t);
        double ax = std::fabs(p.x), ay = std::fabs(p.y), az = std::fabs(p.z);
        double mc = std::fmax(std::fmax(ax, ay), az);
        V3 n = (mc == ax) ? V3{p.x, 0.0, 0.0} : (mc == ay) ? V3{0.0, p.y, 0.0} : V3{0.0, 0.0, p.z};
        emit_shape_hit(cx, sh, leaf, ray, t, nrm_or_flag(cx, n), out);
      }
      return;
    }
    case RL_O_CYLINDER: {  // cylinder.rs:30-140
      cx.c.planar_tests++;
      std::vector<double> ts;
      double a = dx * dx + dz * dz;
      if (!(std::fabs(a) < EPS)) {
        double b = 2.0 * ox * dx + 2.0 * oz * dz;
        double c = ox * ox + oz * oz - 1.0;
        double disc = b * b - 4.0 * a * c;
        if (!(disc < 0.0)) {
          double t0 = (-b - std::sqrt(disc)) / (2.0 * a);
          double t1 = (-b + std::sqrt(disc)) / (2.0 * a);
          double y0 = oy + t0 * dy;
          if (in_bounds(sh, y0)) ts.push_back(t0);
          double y1 = oy + t1 * dy;
          if (in_bounds(sh, y1)) ts.push_back(t1);
        }
      }
      if (sh.closed && !(std::fabs(dy) < EPS)) {
        auto check_cap = [&](double t) {
          double x = ox + t * dx, z = oz + t * dz;
          return x * x + z * z <= 1.0;
        };
        if (sh.has_minimum) {
          double t = (sh.minimum - oy) / dy;
          if (check_cap(t)) ts.push_back(t);
        }
        if (sh.has_maximum) {
          double t = (sh.maximum - oy) / dy;
          if (check_cap(t)) ts.push_back(t);
        }
      }
      for (double t : ts) {  // normal_at cylinder.rs:66-84
        P3 p = point_at(t);
        double dist2 = p.x * p.x + p.z * p.z;
        V3 n;
        if (dist2 < 1.0 && sh.has_maximum && p.y >= sh.maximum - EPS) n = V3{0.0, 1.0, 0.0};
        else if (dist2 < 1.0 && sh.has_minimum && p.y <= sh.minimum + EPS) n = V3{0.0, -1.0, 0.0};
        else n = V3{p.x, 0.0, p.z};
        emit_shape_hit(cx, sh, leaf, ray, t, nrm_or_flag(cx, n), out);
      }
      return;
    }
    case RL_O_CONE: {  // cone.rs:30-150
      cx.c.planar_tests++;
      std::vector<double> ts;
      double a = dx * dx - dy * dy + dz * dz;
      double b = 2.0 * ox * dx - 2.0 * oy * dy + 2.0 * oz * dz;
      double c = ox * ox - oy * oy + oz * oz;
      bool a0 = std::fabs(a) < EPS, b0 = std::fabs(b) < EPS;
      if (a0 && b0) {
      } else if (a0 && !b0) {
        ts.push_back(-c / (2.0 * b));
      } else {
        double disc = b * b - 4.0 * a * c;
        if (!(disc < 0.0)) {
          double t0 = (-b - std::sqrt(disc)) / (2.0 * a);
          double t1 = (-b + std::sqrt(disc)) / (2.0 * a);
          double y0 = oy + t0 * dy;
          if (in_bounds(sh, y0)) ts.push_back(t0);
          double y1 = oy + t1 * dy;
          if (in_bounds(sh, y1)) ts.push_back(t1);
        }
      }
      if (sh.closed && !(std::fabs(dy) < EPS)) {
        auto check_cap = [&](double y, double t) {
          double x = ox + t * dx, z = oz + t * dz;
          return x * x + z * z <= std::fabs(y);
        };
        if (sh.has_minimum) {
          double t = (sh.minimum - oy) / dy;
          if (check_cap(sh.minimum, t)) ts.push_back(t);
        }
        if (sh.has_maximum) {
          double t = (sh.maximum - oy) / dy;
          if (check_cap(sh.maximum, t)) ts.push_back(t);
        }
      }
      for (double t : ts) {  // normal_at cone.rs:62-80
        P3 p = point_at(t);
        double dist2 = p.x * p.x + p.z * p.z;
        V3 n;
        if (sh.has_maximum && dist2 < sh.maximum * sh.maximum && p.y >= sh.maximum - EPS) n = V3{0.0, 1.0, 0.0};
        else if (sh.has_minimum && dist2 < sh.minimum * sh.minimum && p.y <= sh.minimum + EPS) n = V3{0.0, -1.0, 0.0};
        else {
          double y = std::sqrt(p.x * p.x + p.z * p.z);
          if (p.y > 0.0) y = -y;
          n = V3{p.x, y, p.z};
        }
        emit_shape_hit(cx, sh, leaf, ray, t, nrm_or_flag(cx, n), out);
      }
      return;
    }
  }
}

inline void check_axis(double mn, double mx, double origin, double speed, double &lo, double &hi) {  // bounded.rs:127-139
  double tmin = (mn - origin) / speed;
  double tmax = (mx - origin) / speed;
  if (tmin > tmax) lo = tmax, hi = tmin;
  else lo = tmin, hi = tmax;
}

void intersect_oref(RtcCtx &cx, rl_oref o, const RRay &ray, std::vector<Isect> &out) {
  const rl_rtc_scene_desc &d = *cx.d;
  switch (o.kind) {
    case RL_O_TRIANGLE: {  // triangle.rs:63-101
      const rl_rtc_triangle &t = d.triangles[o.index];
      cx.c.planar_tests++;
      V3 e1 = v3(t.e1), e2 = v3(t.e2);
      V3 dir_cross_e2 = cross(ray.d, e2);
      double det = dot(e1, dir_cross_e2);
      if (std::fabs(det) < 1e-8) return;
      double f = 1.0 / det;
      V3 p1_to_origin = sub(ray.o, P3{t.p1[0], t.p1[1], t.p1[2]});
      double u = f * dot(p1_to_origin, dir_cross_e2);
      if (!(0.0 <= u && u <= 1.0)) return;
      V3 origin_cross_e1 = cross(p1_to_origin, e1);
      double v = f * dot(ray.d, origin_cross_e1);
      if (v < 0.0 || (u + v) > 1.0) return;
      double tt = f * dot(e2, origin_cross_e1);
      V3 normal;
      if (t.smooth) {
        V3 n = (v3(t.n2) * u + v3(t.n3) * v) + v3(t.n1) * (1.0 - u - v);
        if (!norm(n, normal)) {
          cx.c.flagged++;
          normal = V3{0, 0, 0};
        }
      } else
        normal = v3(t.n1);
      out.push_back(Isect{tt, o.index, surface_color_at(d, d.materials[t.material], padd(ray.o, ray.d * tt)), normal});
      return;
    }
    case RL_O_SPHERE:
    case RL_O_PLANE:
    case RL_O_CUBE:
    case RL_O_CYLINDER:
    case RL_O_CONE:
      intersect_shape(cx, o.index, ray, out);
      return;
    case RL_O_CSG: {  // csg.rs:78-108
      const rl_rtc_csg &c = d.csgs[o.index];
      std::vector<Isect> lx, rx;
      intersect_oref(cx, c.left, ray, lx);
      intersect_oref(cx, c.right, ray, rx);
      struct Sided {
        Isect i;
        bool left;
      };
      std::vector<Sided> all;
      for (auto &i : lx) all.push_back(Sided{i, true});
      for (auto &i : rx) all.push_back(Sided{i, false});
      std::stable_sort(all.begin(), all.end(), [](const Sided &a, const Sided &b) { return a.i.t < b.i.t; });
      bool in_l = false, in_r = false;
      for (auto &x : all) {  // filter_intersections csg.rs:50-74, intersection_allowed :16-28
        bool allowed;
        if (c.operation == RL_CSG_UNION) allowed = (x.left && !in_r) || (!x.left && !in_l);
        else if (c.operation == RL_CSG_INTERSECTION) allowed = (x.left && in_r) || (!x.left && in_l);
        else allowed = (x.left && !in_r) || (!x.left && in_l);
        if (x.left) in_l = !in_l;
        else in_r = !in_r;
        if (allowed) out.push_back(x.i);
      }
      return;
    }
    case RL_O_GROUP: {  // group.rs:29-42
      const rl_rtc_group &g = d.groups[o.index];
      std::vector<Isect> xs;
      for (uint32_t i = 0; i < g.count; i++) intersect_oref(cx, d.group_items[g.first + i], ray, xs);
      sort_xs(xs);
      out.insert(out.end(), xs.begin(), xs.end());
      return;
    }
    case RL_O_BOUNDED: {  // bounded.rs:100-124,146-152
      const rl_rtc_bounded &b = d.boundeds[o.index];
      cx.c.node_tests++;
      double xl, xh, yl, yh, zl, zh;
      check_axis(b.minimum[0], b.maximum[0], ray.o.x, ray.d.x, xl, xh);
      check_axis(b.minimum[1], b.maximum[1], ray.o.y, ray.d.y, yl, yh);
      check_axis(b.minimum[2], b.maximum[2], ray.o.z, ray.d.z, zl, zh);
      double tmin = std::fmax(std::fmax(xl, yl), zl);
      double tmax = std::fmin(std::fmin(xh, yh), zh);
      if (tmin <= tmax) intersect_oref(cx, b.child, ray, out);
      return;
    }
    case RL_O_TRANSFORMED: {  // transformed.rs:39-51
      const rl_rtc_transformed &t = d.transformeds[o.index];
      cx.c.instance_enters++;
      RRay local{mul_point(t.inverse, ray.o), mul_vec(t.inverse, ray.d)};
      std::vector<Isect> xs;
      intersect_oref(cx, t.child, local, xs);
      for (auto &x : xs) {
        V3 wn = mul_vec(t.inverse_transpose, x.normal), nn;
        if (!norm(wn, nn)) {
          cx.c.flagged++;
          nn = x.normal;
        }
        x.normal = nn;
      }
      out.insert(out.end(), xs.begin(), xs.end());
      return;
    }
  }
}

V3 surface_color_at(const rl_rtc_scene_desc &d, const rl_rtc_material &m, P3 p) {
  if (m.pattern == 0) return v3(m.color);
  const rl_rtc_pattern &pt = d.patterns[m.pattern - 1];
  P3 q = mul_point(pt.inverse, p);  // pattern/mod.rs:9-11
  V3 a = v3(pt.a), b = v3(pt.b);
  switch (pt.kind) {
    case RL_PAT_STRIPE:  // stripe.rs:20-26
      return (f2i64(std::floor(q.x)) % 2 == 0) ? a : b;
    case RL_PAT_RING: {  // ring.rs:19-27
      double radius = std::sqrt(q.x * q.x + q.z * q.z);
      return (f2i64(std::floor(radius)) % 2 == 0) ? a : b;
    }
    case RL_PAT_GRADIENT: {  // gradient.rs:19-24
      V3 distance = b - a;
      double fraction = q.x - std::floor(q.x);
      return a + distance * fraction;
    }
    default:  // checker3d.rs:19-25
      return (f2i64(std::floor(q.x) + std::floor(q.y) + std::floor(q.z)) % 2 == 0) ? a : b;
  }
}

std::vector<Isect> world_intersect(RtcCtx &cx, const RRay &ray) {  // world.rs:46-55
  std::vector<Isect> xs;
  for (uint32_t i = 0; i < cx.d->n_objects; i++) intersect_oref(cx, cx.d->objects[i], ray, xs);
  sort_xs(xs);
  return xs;
}

// intersect.rs:159-168: lowest t >= 0, later element wins ties
inline const Isect *hit(const std::vector<Isect> &xs) {
  const Isect *acc = nullptr;
  for (const auto &i : xs)
    if (i.t >= 0.0) {
      if (acc) acc = (acc->t < i.t) ? acc : &i;
      else acc = &i;
    }
  return acc;
}

inline bool are_equal(double a, double b) {  // math/util.rs:4-22
  if (std::isnan(a) || std::isnan(b)) return false;
  if (std::isinf(a) && std::isinf(b)) return a == b;
  double abs_diff = std::fabs(a - b);
  if (abs_diff <= 2.220446049250313e-16 * 2.0) return true;
  uint64_t au, bu;
  std::memcpy(&au, &a, 8);
  std::memcpy(&bu, &b, 8);
  uint64_t diff = au > bu ? au - bu : bu - au;
  return diff <= 8;
}

struct Comps {  // intersect.rs:123-136
  double t;
  uint32_t object;
  P3 point, over_point, under_point;
  V3 eye_v, normal_v, reflect_v;
  bool inside;
  double n1, n2;
  V3 object_color;
};

Comps prepare_computations(RtcCtx &cx, const Isect &isect, const RRay &ray, const std::vector<Isect> &xs) {  // intersect.rs:48-115
  Comps c;
  c.t = isect.t;
  c.object = isect.object;
  c.point = padd(ray.o, ray.d * isect.t);
  if (!norm(-ray.d, c.eye_v)) {
    cx.c.flagged++;
    c.eye_v = -ray.d;
  }
  V3 normal_v = isect.normal;
  double nde = dot(normal_v, c.eye_v);
  if (nde < 0.0) c.normal_v = -normal_v, c.inside = true;
  else c.normal_v = normal_v, c.inside = false;
  c.over_point = padd(c.point, c.normal_v * 1e-5);
  c.under_point = psub(c.point, c.normal_v * 1e-5);
  if (!norm(reflect(ray.d, c.normal_v), c.reflect_v)) {
    cx.c.flagged++;
    c.reflect_v = ray.d;
  }
  std::vector<uint32_t> containers;
  c.n1 = 1.0, c.n2 = 1.0;
  auto ri = [&](uint32_t leaf) { return cx.d->materials[leaf_material(*cx.d, leaf)].refractive_index; };
  for (const auto &i : xs) {
    bool same = are_equal(i.t, isect.t) && i.object == isect.object;
    if (same) c.n1 = containers.empty() ? 1.0 : ri(containers.back());
    auto it = std::find(containers.begin(), containers.end(), i.object);
    if (it != containers.end()) containers.erase(it);
    else containers.push_back(i.object);
    if (same) {
      c.n2 = containers.empty() ? 1.0 : ri(containers.back());
      break;
    }
  }
  c.object_color = isect.color;
  return c;
}

V3 color_at_internal(RtcCtx &cx, const RRay &ray, uint32_t remaining);

double shadow_attenuation(RtcCtx &cx, P3 point, const rl_rtc_light &light) {  // world.rs:104-126
  V3 v = sub(P3{light.position[0], light.position[1], light.position[2]}, point);
  double distance = mag(v);
  V3 dir;
  if (!norm(v, dir)) return 1.0;
  cx.c.rays++;
  RRay r{point, dir};
  std::vector<Isect> xs = world_intersect(cx, r);
  std::vector<uint32_t> seen;
  double prod = 1.0;
  for (const auto &i : xs) {
    if (!(i.t > 0.0 && i.t < distance)) continue;
    if (std::find(seen.begin(), seen.end(), i.object) != seen.end()) break;  // take_while(seen.insert)
    seen.push_back(i.object);
    prod = prod * cx.d->materials[leaf_material(*cx.d, i.object)].transparency;
  }
  return prod;
}

V3 lighting(const rl_rtc_material &m, P3 point, V3 object_color, const rl_rtc_light &light, V3 eyev, V3 normalv, double shadow_att) {  // material.rs:54-90
  V3 intensity = v3(light.intensity);
  V3 effective = object_color * intensity;
  V3 lightv;
  if (!norm(sub(P3{light.position[0], light.position[1], light.position[2]}, point), lightv)) lightv = V3{0, 0, 0};
  V3 ambient = effective * m.ambient;
  double ldn = dot(lightv, normalv);
  V3 diffuse{0, 0, 0}, specular{0, 0, 0};
  if (!(ldn < 0.0)) {
    V3 diff = (effective * m.diffuse) * ldn;
    V3 reflectv = -reflect(lightv, normalv);
    double rde = dot(reflectv, eyev);
    diffuse = diff * shadow_att;
    if (!(rde <= 0.0)) {
      double factor = std::pow(rde, m.shininess);
      specular = intensity * (m.specular * factor * shadow_att);
    }
  }
  return (ambient + diffuse) + specular;
}

double schlick(const Comps &c) {  // intersect.rs:139-156
  double cosv = dot(c.eye_v, c.normal_v);
  double n = c.n1 / c.n2;
  double sin2_t = n * n * (1.0 - cosv * cosv);
  double cos_t = std::sqrt(1.0 - sin2_t);
  double cos_adj = n > 1.0 ? cos_t : cosv;
  if (sin2_t > 1.0 && n > 1.0) return 1.0;
  double q = (c.n1 - c.n2) / (c.n1 + c.n2);
  double r0 = q * q;
  double x = 1.0 - cos_adj;
  double x2 = x * x;
  return r0 + (1.0 - r0) * (x * (x2 * x2));
}

bool shade_hit(RtcCtx &cx, const Comps &c, uint32_t remaining, V3 &out) {  // world.rs:57-87
  const rl_rtc_material &m = cx.d->materials[leaf_material(*cx.d, c.object)];
  bool have = false;
  V3 acc{0, 0, 0};
  for (uint32_t li = 0; li < cx.d->n_lights; li++) {
    const rl_rtc_light &light = cx.d->lights[li];
    double sa = shadow_attenuation(cx, c.over_point, light);
    V3 surface = lighting(m, c.point, c.object_color, light, c.eye_v, c.normal_v, sa);
    V3 reflected{0, 0, 0}, refracted{0, 0, 0};
    if (!(remaining == 0 || m.reflectivity == 0.0)) {  // world.rs:128-136
      RRay rr{c.over_point, c.reflect_v};
      reflected = color_at_internal(cx, rr, remaining - 1) * m.reflectivity;
    }
    if (!(remaining == 0 || m.transparency == 0.0)) {  // world.rs:138-159
      double n_ratio = c.n1 / c.n2;
      double cos_i = dot(c.eye_v, c.normal_v);
      double sin2_t = n_ratio * n_ratio * (1.0 - cos_i * cos_i);
      if (!(sin2_t > 1.0)) {
        double cos_t = std::sqrt(1.0 - sin2_t);
        V3 direction = c.normal_v * (n_ratio * cos_i - cos_t) - c.eye_v * n_ratio;
        RRay rr{c.under_point, direction};
        refracted = color_at_internal(cx, rr, remaining - 1) * m.transparency;
      }
    }
    V3 col;
    if (m.reflectivity > 0.0 && m.transparency > 0.0) {
      double reflectance = schlick(c);
      col = surface + (reflected * reflectance + refracted * (1.0 - reflectance));
    } else
      col = surface + (reflected + refracted);
    if (have) acc = acc + col;
    else acc = col, have = true;
  }
  out = acc;
  return have;
}

V3 color_at_internal(RtcCtx &cx, const RRay &ray, uint32_t remaining) {  // world.rs:89-98
  cx.c.rays++;
  std::vector<Isect> xs = world_intersect(cx, ray);
  const Isect *h = hit(xs);
  V3 out;
  if (h) {
    Comps c = prepare_computations(cx, *h, ray, xs);
    if (shade_hit(cx, c, remaining, out)) return out;
  }
  return v3(cx.d->void_color);
}

void rtc_pixel(RtcCtx &cx, const rl_rtc_camera &cam, uint32_t samples, uint32_t px, uint32_t py, double out[3]) {  // camera.rs:63-124
  V3 acc{0, 0, 0};
  bool have = false;
  for (uint32_t nx = 0; nx < samples; nx++)
    for (uint32_t ny = 0; ny < samples; ny++) {
      double sample_offset = 1.0 / (double)samples;
      double xoffset = ((double)px + sample_offset * ((double)nx + 0.5)) * cam.pixel_size;
      double yoffset = ((double)py + sample_offset * ((double)ny + 0.5)) * cam.pixel_size;
      double world_x = cam.half_width - xoffset;
      double world_y = cam.half_height - yoffset;
      P3 pixel = mul_point(cam.inverse, P3{world_x, world_y, -1.0});
      P3 origin = mul_point(cam.inverse, P3{0.0, 0.0, 0.0});
      V3 dir;
      if (!norm(sub(pixel, origin), dir)) {
        cx.c.flagged++;
        dir = V3{0, 0, 0};
      }
      V3 c = color_at_internal(cx, RRay{origin, dir}, cx.d->max_reflection_depth);
      if (have) acc = acc + c;
      else acc = c, have = true;
    }
  V3 r = acc * (1.0 / (double)((uint64_t)samples * samples));
  out[0] = r.x, out[1] = r.y, out[2] = r.z;
}

void add_counters(Counters &a, const Counters &b) {
  a.rays += b.rays, a.node_tests += b.node_tests, a.sphere_tests += b.sphere_tests, a.planar_tests += b.planar_tests;
  a.instance_enters += b.instance_enters, a.rng_words += b.rng_words, a.flagged += b.flagged;
}
void store_stats(rl_stats *s, const Counters &c, double ms) {
  if (!s) return;
  s->rays = c.rays, s->node_tests = c.node_tests, s->sphere_tests = c.sphere_tests, s->planar_tests = c.planar_tests;
  s->instance_enters = c.instance_enters, s->rng_words = c.rng_words, s->flagged = c.flagged, s->kernel_ms = ms;
}

template <class F>
void parallel_rows(uint32_t nrows, int threads, F f) {
  if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
  if (threads < 1) threads = 1;
  std::atomic<uint32_t> next{0};
  std::vector<std::thread> ts;
  for (int t = 0; t < threads; t++)
    ts.emplace_back([&, t] {
      for (;;) {
        uint32_t r = next.fetch_add(1);
        if (r >= nrows) break;
        f(r, t);
      }
    });
  for (auto &t : ts) t.join();
}

}  // namespace

extern "C" {

int rlo_hardware_threads() { return (int)std::thread::hardware_concurrency(); }

// CPU restatement of Camera::_render (RTIOW camera.rs:145-199) over rows row_first, row_first+row_step, ...
// out: compact rows, W*3 doubles each (sums). threads<=0: all hardware threads.
int rlo_rtiow_render_rows(const rl_rtiow_scene_desc *desc, const rl_rtiow_camera *cam, uint64_t first_sample, uint32_t row_first,
                          uint32_t row_step, int threads, double *out, rl_stats *stats) {
  if (!desc || !cam || !out || row_step == 0) return RL_E_INVALID;
  uint32_t H = cam->image_height, W = cam->image_width;
  uint32_t nrows = row_first < H ? (H - row_first + row_step - 1) / row_step : 0;
  if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
  std::vector<Counters> per(threads > 0 ? threads : 1);
  parallel_rows(nrows, threads, [&](uint32_t r, int t) {
    RtiowCtx cx{desc, Counters{}};
    uint32_t y = row_first + r * row_step;
    for (uint32_t x = 0; x < W; x++) rtiow_pixel(cx, *cam, first_sample, x, y, out + ((size_t)r * W + x) * 3);
    add_counters(per[t], cx.c);
  });
  Counters tot;
  for (auto &c : per) add_counters(tot, c);
  store_stats(stats, tot, 0.0);
  return tot.flagged ? RL_E_DEGENERATE : RL_OK;
}
// The same per-pixel loop over an explicit pixel list (xs[i], ys[i]), one task per PIXEL: bench.py's full-spp spot check of the
// timed frame and its cpu_baseline leg (a row is too coarse a task for 256 host threads).  out: n x 3 sums.
int rlo_rtiow_render_pixels(const rl_rtiow_scene_desc *desc, const rl_rtiow_camera *cam, uint64_t first_sample, const uint32_t *xs, const uint32_t *ys,
                            uint32_t n, int threads, double *out, rl_stats *stats) {
  if (!desc || !cam || !out || (n && (!xs || !ys))) return RL_E_INVALID;
  for (uint32_t i = 0; i < n; i++)
    if (xs[i] >= cam->image_width || ys[i] >= cam->image_height) return RL_E_INVALID;
  if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
  std::vector<Counters> per(threads > 0 ? threads : 1);
  parallel_rows(n, threads, [&](uint32_t i, int t) {
    RtiowCtx cx{desc, Counters{}};
    rtiow_pixel(cx, *cam, first_sample, xs[i], ys[i], out + (size_t)i * 3);
    add_counters(per[t], cx.c);
  });
  Counters tot;
  for (auto &c : per) add_counters(tot, c);
  store_stats(stats, tot, 0.0);
  return tot.flagged ? RL_E_DEGENERATE : RL_OK;
}
// Bvh::new (RTIOW bvh.rs:22-60) + find_longest_axis (bvh.rs:63-77) restated as the plain recursion the reference runs, on
// the hittables' bounding boxes (what Hittable::bounding_box() returns).  Output = rl_bvh_build's format (include/rl_render.h):
// nodes in creation order (node, left subtree, right subtree), children of inner nodes as RL_H_BVH node_base + index.
// bvh.rs:49 sorts with sort_unstable_by, whose order among EQUAL keys is implementation-defined; this restatement (like the
// product) uses a stable sort on f64::total_cmp keys — the documented variant, unpinned by any reference fixture.
namespace {
struct BvhBuilder {
  const double *boxes;
  const rl_href *prims;
  uint32_t node_base;
  std::vector<rl_bvh_node> out;
  static bool total_less(double a, double b) {  // f64::total_cmp(a, b) == Less
    int64_t x, y;
    std::memcpy(&x, &a, 8), std::memcpy(&y, &b, 8);
    x ^= (int64_t)((uint64_t)(x >> 63) >> 1), y ^= (int64_t)((uint64_t)(y >> 63) >> 1);
    return x < y;
  }
  uint32_t build(std::vector<uint32_t> hs) {
    const uint32_t idx = (uint32_t)out.size();
    out.push_back(rl_bvh_node{});
    rl_bvh_node nd{};
    auto merge_into = [&](double *b, uint32_t h) {  // AABB::merge (aabb.rs:135-140): Interval::merge per axis, f64::min / max
      for (int ax = 0; ax < 3; ax++) b[2 * ax] = std::fmin(b[2 * ax], boxes[6 * (size_t)h + 2 * ax]), b[2 * ax + 1] = std::fmax(b[2 * ax + 1], boxes[6 * (size_t)h + 2 * ax + 1]);
    };
    if (hs.size() == 1) {  // bvh.rs:27-30
      for (int k = 0; k < 6; k++) nd.bbox[k] = boxes[6 * (size_t)hs[0] + k];
      nd.n_children = 1, nd.child[0] = prims[hs[0]];
    } else if (hs.size() == 2) {  // bvh.rs:31-35: left.bounding_box().merge(&right.bounding_box())
      for (int k = 0; k < 6; k++) nd.bbox[k] = boxes[6 * (size_t)hs[0] + k];
      merge_into(nd.bbox, hs[1]);
      nd.n_children = 2, nd.child[0] = prims[hs[0]], nd.child[1] = prims[hs[1]];
    } else {
      for (int ax = 0; ax < 3; ax++) nd.bbox[2 * ax] = INFINITY, nd.bbox[2 * ax + 1] = -INFINITY;  // AABB::empty()
      for (uint32_t h : hs) merge_into(nd.bbox, h);                                                   // bvh.rs:37-39
      double sx = nd.bbox[1] - nd.bbox[0], sy = nd.bbox[3] - nd.bbox[2], sz = nd.bbox[5] - nd.bbox[4];
      int axis = sx > sy ? (sx > sz ? 0 : 2) : (sy > sz ? 1 : 2);  // bvh.rs:63-77
      std::stable_sort(hs.begin(), hs.end(), [&](uint32_t l, uint32_t r) { return total_less(boxes[6 * (size_t)l + 2 * axis], boxes[6 * (size_t)r + 2 * axis]); });
      size_t mid = hs.size() / 2;  // bvh.rs:51-53
      std::vector<uint32_t> ls(hs.begin(), hs.begin() + (long)mid), rs(hs.begin() + (long)mid, hs.end());
      hs.clear();
      hs.shrink_to_fit();
      uint32_t l = build(std::move(ls));
      uint32_t r = build(std::move(rs));
      nd.n_children = 2;
      nd.child[0] = rl_href{RL_H_BVH, node_base + l}, nd.child[1] = rl_href{RL_H_BVH, node_base + r};
    }
    out[idx] = nd;
    return idx;
  }
};
}  // namespace
int rlo_bvh_build(const double *prim_boxes, const rl_href *prims, uint32_t n, uint32_t node_base, rl_bvh_node *out_nodes, uint32_t cap, uint32_t *out_n_nodes) {
  if (!prim_boxes || !prims || !out_nodes || n == 0) return RL_E_INVALID;  // bvh.rs:23-25 panics on an empty list
  BvhBuilder b{prim_boxes, prims, node_base, {}};
  std::vector<uint32_t> all(n);
  for (uint32_t i = 0; i < n; i++) all[i] = i;
  b.build(std::move(all));
  if (out_n_nodes) *out_n_nodes = (uint32_t)b.out.size();
  if (b.out.size() > cap) return RL_E_INVALID;
  std::memcpy(out_nodes, b.out.data(), b.out.size() * sizeof(rl_bvh_node));
  return RL_OK;
}

int rlo_rtiow_render(const rl_rtiow_scene_desc *desc, const rl_rtiow_camera *cam, uint64_t first_sample, int threads, double *out, rl_stats *stats) {
  return rlo_rtiow_render_rows(desc, cam, first_sample, 0, 1, threads, out, stats);
}

int rlo_rtc_render_rows(const rl_rtc_scene_desc *desc, const rl_rtc_camera *cam, uint32_t aa, uint32_t row_first, uint32_t row_step, int threads,
                        double *out, rl_stats *stats) {
  if (!desc || !cam || !out || row_step == 0 || aa == 0) return RL_E_INVALID;
  uint32_t H = cam->vsize, W = cam->hsize;
  uint32_t nrows = row_first < H ? (H - row_first + row_step - 1) / row_step : 0;
  if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
  std::vector<Counters> per(threads > 0 ? threads : 1);
  parallel_rows(nrows, threads, [&](uint32_t r, int t) {
    RtcCtx cx{desc, Counters{}};
    uint32_t y = row_first + r * row_step;
    for (uint32_t x = 0; x < W; x++) rtc_pixel(cx, *cam, aa, x, y, out + ((size_t)r * W + x) * 3);
    add_counters(per[t], cx.c);
  });
  Counters tot;
  for (auto &c : per) add_counters(tot, c);
  store_stats(stats, tot, 0.0);
  return tot.flagged ? RL_E_DEGENERATE : RL_OK;
}
int rlo_rtc_render(const rl_rtc_scene_desc *desc, const rl_rtc_camera *cam, uint32_t aa, int threads, double *out, rl_stats *stats) {
  return rlo_rtc_render_rows(desc, cam, aa, 0, 1, threads, out, stats);
}

// ---------------------------------------------------------------- unit probes (known-answer tests)
void rlo_chacha_key(uint64_t seed, uint32_t key_out[8]) {
  ChaCha8 r;
  r.seed_from_u64(seed);
  std::memcpy(key_out, r.key, sizeof r.key);
}
void rlo_chacha_block(uint64_t seed, uint64_t ctr, uint64_t stream, uint32_t out16[16]) {
  ChaCha8 r;
  r.seed_from_u64(seed);
  ChaCha8::block(r.key, ctr, stream, out16);
}
// Draw script: ops[i] = 0: gen_f64, 1: uniform(-1,1), 2: set_stream(args[i]) (no output), 3: next_u64 (as double bits)
void rlo_chacha_script(uint64_t seed, const uint32_t *ops, const uint64_t *args, uint32_t n, double *out, uint64_t *final_pos) {
  ChaCha8 r;
  r.seed_from_u64(seed);
  for (uint32_t i = 0; i < n; i++) {
    switch (ops[i]) {
      case 0: out[i] = r.gen_f64(); break;
      case 1: out[i] = r.uniform_m1_1(); break;
      case 2: r.set_stream(args[i]); out[i] = 0.0; break;
      case 3: {
        uint64_t v = r.next_u64();
        std::memcpy(&out[i], &v, 8);
        break;
      }
    }
  }
  if (final_pos) *final_pos = r.pos;
}
// per-pixel sums + final word position of the RTIOW pixel loop
int rlo_rtiow_pixel(const rl_rtiow_scene_desc *desc, const rl_rtiow_camera *cam, uint64_t first_sample, uint32_t x, uint32_t y, double out[3],
                    uint64_t *words) {
  RtiowCtx cx{desc, Counters{}};
  rtiow_pixel(cx, *cam, first_sample, x, y, out);
  if (words) *words = cx.c.rng_words;
  return cx.c.flagged ? RL_E_DEGENERATE : RL_OK;
}
// Hittable::hit on the scene root: returns 1 on hit; out = {t, px,py,pz, nx,ny,nz, front, u, v, mat}
int rlo_rtiow_hit(const rl_rtiow_scene_desc *desc, const double o[3], const double d[3], double time, double tmin, double tmax, double out[11]) {
  RtiowCtx cx{desc, Counters{}};
  HitRecord rec;
  Ray r{v3(o), v3(d), time};
  if (!hit_href(cx, desc->root, r, tmin, tmax, rec)) return 0;
  double v[11] = {rec.t, rec.p.x, rec.p.y, rec.p.z, rec.normal.x, rec.normal.y, rec.normal.z, rec.front ? 1.0 : 0.0, rec.u, rec.v, (double)rec.mat};
  std::memcpy(out, v, sizeof v);
  return 1;
}
void rlo_sphere_uv(const double p[3], double out[2]) { sphere_uv(v3(p), out[0], out[1]); }
double rlo_perlin_noise(const rl_perlin *pn, const double p[3]) { return perlin_noise(*pn, v3(p)); }
double rlo_perlin_turb(const rl_perlin *pn, const double p[3], uint32_t depth) { return perlin_turb(*pn, v3(p), depth); }
int rlo_aabb_hit(const double bbox[6], const double o[3], const double d[3], double tmin, double tmax) {
  Ray r{v3(o), v3(d), 0.0};
  return aabb_hit(bbox, r, tmin, tmax) ? 1 : 0;
}
// World::intersect (sorted): writes up to cap (t, object) pairs; returns the count
int rlo_rtc_intersect(const rl_rtc_scene_desc *desc, const double o[3], const double d[3], double *ts, uint32_t *objs, double *normals, uint32_t cap) {
  RtcCtx cx{desc, Counters{}};
  std::vector<Isect> xs = world_intersect(cx, RRay{P3{o[0], o[1], o[2]}, v3(d)});
  for (uint32_t i = 0; i < xs.size() && i < cap; i++) {
    ts[i] = xs[i].t;
    objs[i] = xs[i].object;
    if (normals) normals[3 * i] = xs[i].normal.x, normals[3 * i + 1] = xs[i].normal.y, normals[3 * i + 2] = xs[i].normal.z;
  }
  return (int)xs.size();
}
void rlo_rtc_color_at(const rl_rtc_scene_desc *desc, const double o[3], const double d[3], double out[3]) {
  RtcCtx cx{desc, Counters{}};
  V3 c = color_at_internal(cx, RRay{P3{o[0], o[1], o[2]}, v3(d)}, desc->max_reflection_depth);
  out[0] = c.x, out[1] = c.y, out[2] = c.z;
}
// material::lighting (scene/material.rs:54-90)
void rlo_rtc_lighting(const rl_rtc_material *m, const double point[3], const double light_pos[3], const double light_int[3], const double eyev[3],
                      const double normalv[3], double shadow_att, double out[3]) {
  rl_rtc_light l{};
  std::memcpy(l.position, light_pos, 24);
  std::memcpy(l.intensity, light_int, 24);
  V3 c = lighting(*m, P3{point[0], point[1], point[2]}, v3(m->color), l, v3(eyev), v3(normalv), shadow_att);
  out[0] = c.x, out[1] = c.y, out[2] = c.z;
}

}  // extern "C"
