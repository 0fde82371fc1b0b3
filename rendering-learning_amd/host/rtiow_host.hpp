// Host-side mirror of ray-tracing-one-weekend's scene-building API (the part of the reference that
// STAYS on the host): CameraParams/Camera::new, Sphere/Quad/Triangle/Translate/Transform/Bvh,
// materials and textures, plus the flattener that turns the object tree into the POD arrays of
// include/rl_render.h.  The reference is Rust; no Rust toolchain exists in this image, so this is
// the C++ restatement of the *host* half.  Names and argument meaning follow the reference.
// All arithmetic that feeds a device-side comparison is written in the reference's operation order.
//
// Written from scratch; reference file:line cited per function (paths under
// /root/reference/ray-tracing-one-weekend/).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rl_render.h"

namespace rtiow {

// ---------------------------------------------------------------- vec3.rs
struct Vec3 {
  double e[3];
  Vec3() : e{0, 0, 0} {}
  Vec3(double x, double y, double z) : e{x, y, z} {}
  double x() const { return e[0]; }
  double y() const { return e[1]; }
  double z() const { return e[2]; }
  double length_squared() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }  // vec3.rs:36
  double length() const { return std::sqrt(length_squared()); }
  double dot(const Vec3 &r) const { return e[0] * r.e[0] + e[1] * r.e[1] + e[2] * r.e[2]; }
  Vec3 cross(const Vec3 &r) const {  // vec3.rs:48
    return Vec3(e[1] * r.e[2] - e[2] * r.e[1], e[2] * r.e[0] - e[0] * r.e[2],
                e[0] * r.e[1] - e[1] * r.e[0]);
  }
};
using Point3 = Vec3;
using Color = Vec3;
inline Vec3 operator+(const Vec3 &a, const Vec3 &b) { return Vec3(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
inline Vec3 operator-(const Vec3 &a, const Vec3 &b) { return Vec3(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
inline Vec3 operator*(const Vec3 &a, const Vec3 &b) { return Vec3(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }
inline Vec3 operator-(const Vec3 &a) { return Vec3(-a.e[0], -a.e[1], -a.e[2]); }
inline Vec3 operator*(const Vec3 &a, double s) { return Vec3(a.e[0] * s, a.e[1] * s, a.e[2] * s); }
inline Vec3 operator*(double s, const Vec3 &a) { return a * s; }            // vec3.rs:168 (rhs * self)
inline Vec3 operator/(const Vec3 &a, double s) { return a * (1.0 / s); }    // vec3.rs:177-179
inline Vec3 normalize(const Vec3 &a) { return a / a.length(); }             // vec3.rs:56

// NormalizedVec3::try_from (vec3.rs:236-247): Err when |v|^2 ~ 0 (== 0 or <= 1e-16)
inline Vec3 try_normalize(const Vec3 &v, const char *what) {
  double m = v.length_squared();
  if (m == 0.0 || std::fabs(m - 0.0) <= 1e-16) throw std::runtime_error(std::string("cannot normalize vector with magnitude 0: ") + what);
  return normalize(v);
}

// ---------------------------------------------------------------- interval.rs / aabb.rs
struct Interval {
  double min, max;
  double size() const { return max - min; }
  Interval expand(double delta) const {  // interval.rs:43
    double padding = delta / 2.0;
    return Interval{min - padding, max + padding};
  }
  Interval merge(const Interval &o) const { return Interval{std::fmin(min, o.min), std::fmax(max, o.max)}; }
};
static const double INF = std::numeric_limits<double>::infinity();

struct AABB {
  Interval x, y, z;
  static AABB make(Interval x, Interval y, Interval z) {  // AABB::new aabb.rs:14-27
    const double DELTA = 1e-4;
    AABB b;
    b.x = x.size() < DELTA ? x.expand(DELTA) : x;
    b.y = y.size() < DELTA ? y.expand(DELTA) : y;
    b.z = z.size() < DELTA ? z.expand(DELTA) : z;
    return b;
  }
  static AABB from_extrema(const Point3 &a, const Point3 &b) {  // aabb.rs:30-65
    Interval x = a.x() <= b.x() ? Interval{a.x(), b.x()} : Interval{b.x(), a.x()};
    Interval y = a.y() <= b.y() ? Interval{a.y(), b.y()} : Interval{b.y(), a.y()};
    Interval z = a.z() <= b.z() ? Interval{a.z(), b.z()} : Interval{b.z(), a.z()};
    return make(x, y, z);
  }
  static AABB from_points(const std::vector<Point3> &pts) {  // aabb.rs:69-91
    double mn[3] = {pts[0].x(), pts[0].y(), pts[0].z()}, mx[3] = {pts[0].x(), pts[0].y(), pts[0].z()};
    for (const auto &p : pts)
      for (int k = 0; k < 3; k++) {
        mn[k] = std::fmin(mn[k], p.e[k]);
        mx[k] = std::fmax(mx[k], p.e[k]);
      }
    return from_extrema(Point3(mn[0], mn[1], mn[2]), Point3(mx[0], mx[1], mx[2]));
  }
  static AABB empty() { return AABB{{INF, -INF}, {INF, -INF}, {INF, -INF}}; }
  static AABB universe() { return AABB{{-INF, INF}, {-INF, INF}, {-INF, INF}}; }
  AABB merge(const AABB &o) const { return AABB{x.merge(o.x), y.merge(o.y), z.merge(o.z)}; }  // aabb.rs:135 (no re-pad)
  AABB offset(const Vec3 &o) const {  // aabb.rs:160-166 (&AABB + &Point3 -> AABB::new, pads again)
    return make(Interval{x.min + o.x(), x.max + o.x()}, Interval{y.min + o.y(), y.max + o.y()},
                Interval{z.min + o.z(), z.max + o.z()});
  }
};

// ---------------------------------------------------------------- texture.rs / material.rs
struct ImageData {
  uint32_t width = 0, height = 0;
  std::vector<float> rgb;  // linear
};
// perlin.rs:9-36 Perlin::new(rand): 256 random unit vectors, then three Fisher-Yates permutations of 0..255, all
// drawn from the caller's Rng in that order.  R needs next_u64() (rand_core RngCore).  The draws restate
// rand_distr 0.4.3 UnitSphere (Uniform::new(-1., 1.) pairs, Marsaglia) and rand 0.8.5 gen_range(0..i) for usize
// (widening-multiply rejection, UniformInt::sample_single).  No reference test pins these tables: "unpinned".
struct Perlin {
  rl_perlin tab;
  template <class R>
  static double uniform_m1_1(R &rng) {
    uint64_t bits = (rng.next_u64() >> 12) | 0x3FF0000000000000ull;
    double v12;
    std::memcpy(&v12, &bits, 8);
    return (v12 - 1.0) * 2.0 + (-1.0);
  }
  template <class R>
  static uint64_t gen_range_usize(R &rng, uint64_t low, uint64_t high) {  // low..high, high > low
    uint64_t range = high - low;
    uint64_t zone = (range << __builtin_clzll(range)) - 1;
    for (;;) {
      unsigned __int128 m = (unsigned __int128)rng.next_u64() * range;
      uint64_t hi = (uint64_t)(m >> 64), lo = (uint64_t)m;
      if (lo <= zone) return low + hi;
    }
  }
  template <class R>
  static void permute(uint32_t *p, R &rng) {  // perlin.rs:83-90
    for (uint64_t i = 255; i >= 1; i--) {
      uint64_t target = gen_range_usize(rng, 0, i);
      std::swap(p[i], p[target]);
    }
  }
  template <class R>
  static std::shared_ptr<Perlin> create(R &rng) {
    auto pn = std::make_shared<Perlin>();
    for (int i = 0; i < 256; i++) {  // Vec3::random_unit_vector (vec3.rs:72-75)
      for (;;) {
        double x1 = uniform_m1_1(rng), x2 = uniform_m1_1(rng);
        double sum = x1 * x1 + x2 * x2;
        if (sum >= 1.0) continue;
        double factor = 2.0 * std::sqrt(1.0 - sum);
        pn->tab.randvec[i][0] = x1 * factor, pn->tab.randvec[i][1] = x2 * factor, pn->tab.randvec[i][2] = 1.0 - 2.0 * sum;
        break;
      }
    }
    for (uint32_t *perm : {pn->tab.perm_x, pn->tab.perm_y, pn->tab.perm_z}) {
      for (uint32_t i = 0; i < 256; i++) perm[i] = i;
      permute(perm, rng);
    }
    return pn;
  }
};

struct Texture {
  uint32_t kind = RL_TEX_SOLID;
  Color color;
  double inv_scale = 0.0;  // Checker: 1/scale; Noise: scale
  std::shared_ptr<Texture> even, odd;
  std::shared_ptr<ImageData> image;
  std::shared_ptr<Perlin> noise;
};
using TexturePtr = std::shared_ptr<Texture>;
inline TexturePtr SolidColor(const Color &albedo) {
  auto t = std::make_shared<Texture>();
  t->kind = RL_TEX_SOLID;
  t->color = albedo;
  return t;
}
inline TexturePtr Checker(double scale, TexturePtr even, TexturePtr odd) {  // texture.rs:31-38
  auto t = std::make_shared<Texture>();
  t->kind = RL_TEX_CHECKER;
  t->inv_scale = 1.0 / scale;
  t->even = even;
  t->odd = odd;
  return t;
}
inline TexturePtr Image(std::shared_ptr<ImageData> img) {
  auto t = std::make_shared<Texture>();
  t->kind = RL_TEX_IMAGE;
  t->image = img;
  return t;
}
inline TexturePtr Noise(std::shared_ptr<Perlin> noise, double scale) {  // texture.rs:84-87
  auto t = std::make_shared<Texture>();
  t->kind = RL_TEX_NOISE;
  t->noise = noise;
  t->inv_scale = scale;
  return t;
}
struct Material {
  uint32_t kind = RL_MAT_FLAT;
  TexturePtr texture;
  Color albedo;
  double fuzz = 0.0, ior = 1.0;
};
using MaterialPtr = std::shared_ptr<Material>;
inline MaterialPtr Flat() { return std::make_shared<Material>(); }
inline MaterialPtr Lambertian(TexturePtr t) {
  auto m = std::make_shared<Material>();
  m->kind = RL_MAT_LAMBERTIAN;
  m->texture = t;
  return m;
}
inline MaterialPtr Metal(const Color &albedo, double fuzz) {
  auto m = std::make_shared<Material>();
  m->kind = RL_MAT_METAL;
  m->albedo = albedo;
  m->fuzz = fuzz;
  return m;
}
inline MaterialPtr Dielectric(double refraction_index) {
  auto m = std::make_shared<Material>();
  m->kind = RL_MAT_DIELECTRIC;
  m->ior = refraction_index;
  return m;
}
inline MaterialPtr DiffuseLight(TexturePtr t) {
  auto m = std::make_shared<Material>();
  m->kind = RL_MAT_DIFFUSE_LIGHT;
  m->texture = t;
  return m;
}
inline MaterialPtr Isotropic(TexturePtr t) {  // material.rs:197-199
  auto m = std::make_shared<Material>();
  m->kind = RL_MAT_ISOTROPIC;
  m->texture = t;
  return m;
}

// ---------------------------------------------------------------- flattener
struct Flattened {
  std::vector<rl_sphere> spheres;
  std::vector<rl_planar> planars;
  std::vector<rl_translate> translates;
  std::vector<rl_transform> transforms;
  std::vector<rl_bvh_node> bvh_nodes;
  std::vector<rl_list> lists;
  std::vector<rl_href> list_items;
  std::vector<rl_material> materials;
  std::vector<rl_texture> textures;
  std::vector<rl_image> images;
  std::vector<std::shared_ptr<ImageData>> image_keep;
  std::vector<rl_perlin> perlins;
  std::vector<rl_medium> media;
  std::map<const Perlin *, uint32_t> perlin_ids;
  std::map<const Material *, uint32_t> mat_ids;
  std::map<const Texture *, uint32_t> tex_ids;
  std::map<const ImageData *, uint32_t> img_ids;
  rl_href root{RL_H_NONE, 0};

  uint32_t texture_id(const TexturePtr &t) {
    if (!t) throw std::runtime_error("material without texture");
    auto it = tex_ids.find(t.get());
    if (it != tex_ids.end()) return it->second;
    rl_texture r{};
    r.kind = t->kind;
    r.color[0] = t->color.x(), r.color[1] = t->color.y(), r.color[2] = t->color.z();
    r.inv_scale = t->inv_scale;
    if (t->kind == RL_TEX_CHECKER) {
      r.even = texture_id(t->even);
      r.odd = texture_id(t->odd);
    } else if (t->kind == RL_TEX_IMAGE) {
      auto ii = img_ids.find(t->image.get());
      if (ii == img_ids.end()) {
        rl_image im{t->image->width, t->image->height, t->image->rgb.data()};
        images.push_back(im);
        image_keep.push_back(t->image);
        ii = img_ids.emplace(t->image.get(), (uint32_t)images.size() - 1).first;
      }
      r.image = ii->second;
    } else if (t->kind == RL_TEX_NOISE) {
      auto pi = perlin_ids.find(t->noise.get());
      if (pi == perlin_ids.end()) {
        perlins.push_back(t->noise->tab);
        pi = perlin_ids.emplace(t->noise.get(), (uint32_t)perlins.size() - 1).first;
      }
      r.image = pi->second;
    }
    textures.push_back(r);
    uint32_t id = (uint32_t)textures.size() - 1;
    tex_ids[t.get()] = id;
    return id;
  }
  uint32_t material_id(const MaterialPtr &m) {
    auto it = mat_ids.find(m.get());
    if (it != mat_ids.end()) return it->second;
    rl_material r{};
    r.kind = m->kind;
    if (m->kind == RL_MAT_LAMBERTIAN || m->kind == RL_MAT_DIFFUSE_LIGHT || m->kind == RL_MAT_ISOTROPIC) r.texture = texture_id(m->texture);
    r.albedo[0] = m->albedo.x(), r.albedo[1] = m->albedo.y(), r.albedo[2] = m->albedo.z();
    r.fuzz = m->fuzz;
    r.ior = m->ior;
    materials.push_back(r);
    uint32_t id = (uint32_t)materials.size() - 1;
    mat_ids[m.get()] = id;
    return id;
  }
  rl_rtiow_scene_desc desc() const {
    rl_rtiow_scene_desc d{};
    d.spheres = spheres.data(), d.n_spheres = (uint32_t)spheres.size();
    d.planars = planars.data(), d.n_planars = (uint32_t)planars.size();
    d.translates = translates.data(), d.n_translates = (uint32_t)translates.size();
    d.transforms = transforms.data(), d.n_transforms = (uint32_t)transforms.size();
    d.bvh_nodes = bvh_nodes.data(), d.n_bvh_nodes = (uint32_t)bvh_nodes.size();
    d.lists = lists.data(), d.n_lists = (uint32_t)lists.size();
    d.list_items = list_items.data(), d.n_list_items = (uint32_t)list_items.size();
    d.materials = materials.data(), d.n_materials = (uint32_t)materials.size();
    d.textures = textures.data(), d.n_textures = (uint32_t)textures.size();
    d.images = images.data(), d.n_images = (uint32_t)images.size();
    d.perlins = perlins.data(), d.n_perlins = (uint32_t)perlins.size();
    d.media = media.data(), d.n_media = (uint32_t)media.size();
    d.root = root;
    return d;
  }
};

// ---------------------------------------------------------------- hittable/mod.rs:40 trait Hittable
struct Hittable {
  virtual ~Hittable() {}
  virtual AABB bounding_box() const = 0;
  virtual rl_href flatten(Flattened &f) const = 0;
};
using HittablePtr = std::shared_ptr<Hittable>;

inline void put3(double *d, const Vec3 &v) { d[0] = v.x(), d[1] = v.y(), d[2] = v.z(); }

struct Center {  // sphere.rs:11-14
  bool moving;
  Point3 p1, p2;
  static Center Stationary(const Point3 &p) { return Center{false, p, p}; }
  static Center Moving(const Point3 &a, const Point3 &b) { return Center{true, a, b}; }
};

struct Sphere : Hittable {  // sphere.rs:16-21
  Center center;
  double radius;
  MaterialPtr material;
  Sphere(Center c, double r, MaterialPtr m) : center(c), radius(r), material(m) {}
  AABB bounding_box() const override {  // sphere.rs:77-88
    Vec3 rvec(radius, radius, radius);
    if (!center.moving) return AABB::from_extrema(center.p1 - rvec, center.p1 + rvec);
    AABB a = AABB::from_extrema(center.p1 - rvec, center.p1 + rvec);
    AABB b = AABB::from_extrema(center.p2 - rvec, center.p2 + rvec);
    return a.merge(b);
  }
  rl_href flatten(Flattened &f) const override {
    rl_sphere s{};
    put3(s.center0, center.p1);
    put3(s.center1, center.p2);
    s.radius = radius;
    s.moving = center.moving ? 1u : 0u;
    s.material = f.material_id(material);
    f.spheres.push_back(s);
    return rl_href{RL_H_SPHERE, (uint32_t)f.spheres.size() - 1};
  }
};

struct PlaneData {  // flat/plane.rs:23-41 Plane::new
  Point3 q;
  Vec3 u, v, w, normal;
  double d;
  PlaneData(const Point3 &q_, const Vec3 &u_, const Vec3 &v_) : q(q_), u(u_), v(v_) {
    Vec3 n = u.cross(v);
    normal = try_normalize(n, "Failed to find normal because u and v were parallel");
    d = normal.dot(q);
    w = n / n.dot(n);
  }
  void fill(rl_planar &p) const {
    put3(p.q, q), put3(p.u, u), put3(p.v, v), put3(p.w, w), put3(p.normal, normal);
    p.d = d;
  }
};

struct Plane : Hittable {
  PlaneData plane;
  MaterialPtr material;
  Plane(const Point3 &q, const Vec3 &u, const Vec3 &v, MaterialPtr m) : plane(q, u, v), material(m) {}
  AABB bounding_box() const override { return AABB::universe(); }  // plane.rs:102
  rl_href flatten(Flattened &f) const override {
    rl_planar p{};
    plane.fill(p);
    p.kind = RL_PLANAR_PLANE;
    p.material = f.material_id(material);
    f.planars.push_back(p);
    return rl_href{RL_H_PLANAR, (uint32_t)f.planars.size() - 1};
  }
};

struct Quad : Hittable {  // flat/quad.rs:22-34
  PlaneData plane;
  AABB bbox;
  MaterialPtr material;
  Quad(const Point3 &q, const Vec3 &u, const Vec3 &v, MaterialPtr m) : plane(q, u, v), material(m) {
    AABB d1 = AABB::from_extrema(q, q + u + v);
    AABB d2 = AABB::from_extrema(q + u, q + v);
    bbox = d1.merge(d2);
  }
  AABB bounding_box() const override { return bbox; }
  rl_href flatten(Flattened &f) const override {
    rl_planar p{};
    plane.fill(p);
    p.kind = RL_PLANAR_QUAD;
    p.material = f.material_id(material);
    f.planars.push_back(p);
    return rl_href{RL_H_PLANAR, (uint32_t)f.planars.size() - 1};
  }
};

struct Triangle : Hittable {  // flat/triangle.rs:20-55
  PlaneData plane;
  AABB bbox;
  bool has_normals = false, has_uvs = false;
  Vec3 normals[3];
  double uvs[6];
  MaterialPtr material;
  // Triangle::from_model(points, texture_coords, normals, material)
  Triangle(const Point3 pts[3], const double *uv6, const Vec3 *n3, MaterialPtr m)
      : plane(pts[0], pts[1] - pts[0], pts[2] - pts[0]), material(m) {
    bbox = AABB::from_points({pts[0], pts[1], pts[2]});
    if (uv6) {
      has_uvs = true;
      std::memcpy(uvs, uv6, sizeof uvs);
    }
    if (n3) {
      has_normals = true;
      for (int i = 0; i < 3; i++) normals[i] = n3[i];
    }
  }
  // Triangle::new(q,u,v,material)  triangle.rs:24-28
  static std::shared_ptr<Triangle> from_quv(const Point3 &q, const Vec3 &u, const Vec3 &v, MaterialPtr m) {
    Point3 pts[3] = {q, q + u, q + v};
    return std::make_shared<Triangle>(pts, nullptr, nullptr, m);
  }
  AABB bounding_box() const override { return bbox; }
  rl_href flatten(Flattened &f) const override {
    rl_planar p{};
    plane.fill(p);
    p.kind = RL_PLANAR_TRIANGLE;
    p.material = f.material_id(material);
    p.has_normals = has_normals, p.has_uvs = has_uvs;
    for (int i = 0; i < 3; i++) put3(p.normals + 3 * i, normals[i]);
    if (has_uvs) std::memcpy(p.uvs, uvs, sizeof uvs);
    f.planars.push_back(p);
    return rl_href{RL_H_PLANAR, (uint32_t)f.planars.size() - 1};
  }
};

struct Translate : Hittable {  // translate.rs
  HittablePtr object;
  Vec3 offset;
  Translate(HittablePtr o, const Vec3 &off) : object(o), offset(off) {}
  AABB bounding_box() const override { return object->bounding_box().offset(offset); }
  rl_href flatten(Flattened &f) const override {
    rl_translate t{};
    put3(t.offset, offset);
    t.child = object->flatten(f);
    f.translates.push_back(t);
    return rl_href{RL_H_TRANSLATE, (uint32_t)f.translates.size() - 1};
  }
};

struct ConstantMedium : Hittable {  // hittable/constant_medium.rs:9-25 (evaluated with the pixel's RNG: include/rl_render.h rl_medium)
  HittablePtr boundary;
  double neg_inv_density;
  MaterialPtr phase_function;
  ConstantMedium(HittablePtr b, double density, MaterialPtr m) : boundary(b), neg_inv_density(-1.0 / density), phase_function(m) {}
  AABB bounding_box() const override { return boundary->bounding_box(); }  // constant_medium.rs:82-84
  rl_href flatten(Flattened &f) const override {
    rl_medium m{};
    m.boundary = boundary->flatten(f);
    m.neg_inv_density = neg_inv_density;
    m.material = f.material_id(phase_function);
    f.media.push_back(m);
    return rl_href{RL_H_MEDIUM, (uint32_t)f.media.size() - 1};
  }
};

struct Matrix3 {  // matrix.rs
  double m[3][3];
  Matrix3 transpose() const {
    Matrix3 o;
    for (int n = 0; n < 3; n++)
      for (int k = 0; k < 3; k++) o.m[n][k] = m[k][n];
    return o;
  }
  Vec3 mul(const Vec3 &v) const {  // matrix.rs:42-60: accumulate from 0.0
    double out[3];
    for (int n = 0; n < 3; n++) {
      double sum = 0.0;
      for (int k = 0; k < 3; k++) sum += m[n][k] * v.e[k];
      out[n] = sum;
    }
    return Vec3(out[0], out[1], out[2]);
  }
};

struct Transform : Hittable {  // transform.rs:13-139
  HittablePtr object;
  AABB bbox;
  Matrix3 transformation, inv_transformation, inv_transpose_transformation;
  Transform(HittablePtr o, const Matrix3 &t, const Matrix3 &inv) : object(o), transformation(t), inv_transformation(inv) {
    AABB b = object->bounding_box();
    double mn[3] = {INF, INF, INF}, mx[3] = {-INF, -INF, -INF};
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++)
        for (int k = 0; k < 2; k++) {
          double i_f = i, j_f = j, k_f = k;
          double x = i_f * b.x.max + (1.0 - i_f) * b.x.min;
          double y = j_f * b.y.max + (1.0 - j_f) * b.y.min;
          double z = k_f * b.z.max + (1.0 - k_f) * b.z.min;
          Vec3 tester = transformation.mul(Point3(x, y, z));
          for (int a = 0; a < 3; a++) {
            mn[a] = std::fmin(mn[a], tester.e[a]);
            mx[a] = std::fmax(mx[a], tester.e[a]);
          }
        }
    bbox = AABB::from_extrema(Point3(mn[0], mn[1], mn[2]), Point3(mx[0], mx[1], mx[2]));
    inv_transpose_transformation = inv_transformation.transpose();
  }
  static double to_radians(double deg) { return deg * (M_PI / 180.0); }  // f64::to_radians: x * (PI/180)
  static std::shared_ptr<Transform> rotate_x(HittablePtr o, double degrees) {
    double r = to_radians(degrees), s = std::sin(r), c = std::cos(r);
    return std::make_shared<Transform>(o, Matrix3{{{1, 0, 0}, {0, c, -s}, {0, s, c}}}, Matrix3{{{1, 0, 0}, {0, c, s}, {0, -s, c}}});
  }
  static std::shared_ptr<Transform> rotate_y(HittablePtr o, double degrees) {
    double r = to_radians(degrees), s = std::sin(r), c = std::cos(r);
    return std::make_shared<Transform>(o, Matrix3{{{c, 0, s}, {0, 1, 0}, {-s, 0, c}}}, Matrix3{{{c, 0, -s}, {0, 1, 0}, {s, 0, c}}});
  }
  static std::shared_ptr<Transform> rotate_z(HittablePtr o, double degrees) {
    double r = to_radians(degrees), s = std::sin(r), c = std::cos(r);
    return std::make_shared<Transform>(o, Matrix3{{{c, -s, 0}, {s, c, 0}, {0, 0, 1}}}, Matrix3{{{c, s, 0}, {-s, c, 0}, {0, 0, 1}}});
  }
  static std::shared_ptr<Transform> scale(HittablePtr o, double sc) {
    double is = 1.0 / sc;
    return std::make_shared<Transform>(o, Matrix3{{{sc, 0, 0}, {0, sc, 0}, {0, 0, sc}}}, Matrix3{{{is, 0, 0}, {0, is, 0}, {0, 0, is}}});
  }
  AABB bounding_box() const override { return bbox; }
  rl_href flatten(Flattened &f) const override {
    rl_transform t{};
    for (int n = 0; n < 3; n++)
      for (int k = 0; k < 3; k++) {
        t.m[3 * n + k] = transformation.m[n][k];
        t.inv[3 * n + k] = inv_transformation.m[n][k];
        t.inv_t[3 * n + k] = inv_transpose_transformation.m[n][k];
      }
    t.child = object->flatten(f);
    f.transforms.push_back(t);
    return rl_href{RL_H_TRANSFORM, (uint32_t)f.transforms.size() - 1};
  }
};

// a slice / Vec of hittables used directly as the world (hittable/mod.rs:88)
struct HittableList : Hittable {
  std::vector<HittablePtr> items;
  HittableList() {}
  explicit HittableList(std::vector<HittablePtr> v) : items(std::move(v)) {}
  AABB bounding_box() const override {
    AABB b = AABB::empty();
    for (auto &h : items) b = b.merge(h->bounding_box());
    return b;
  }
  rl_href flatten(Flattened &f) const override {
    std::vector<rl_href> refs;
    for (auto &h : items) refs.push_back(h->flatten(f));
    rl_list l{(uint32_t)f.list_items.size(), (uint32_t)refs.size()};
    f.list_items.insert(f.list_items.end(), refs.begin(), refs.end());
    f.lists.push_back(l);
    return rl_href{RL_H_LIST, (uint32_t)f.lists.size() - 1};
  }
};

// f64::total_cmp (used by bvh.rs:49)
inline bool total_less(double a, double b) {
  int64_t x, y;
  std::memcpy(&x, &a, 8);
  std::memcpy(&y, &b, 8);
  x ^= (int64_t)((uint64_t)(x >> 63) >> 1);
  y ^= (int64_t)((uint64_t)(y >> 63) >> 1);
  return x < y;
}

struct Bvh : Hittable {  // bvh.rs:11-60
  bool leaf = false;
  std::vector<HittablePtr> children;  // leaf: 1-2 hittables; inner: 2 Bvh
  AABB bbox;
  static int find_longest_axis(const AABB &b) {  // bvh.rs:63-77
    if (b.x.size() > b.y.size()) return b.x.size() > b.z.size() ? 0 : 2;
    return b.y.size() > b.z.size() ? 1 : 2;
  }
  // Bvh::new. The reference sorts with sort_unstable_by(total_cmp): order among EQUAL keys is
  // implementation-defined there; here a stable sort is used (documented in DESIGN.md).
  explicit Bvh(std::vector<HittablePtr> hs) {
    if (hs.empty()) throw std::runtime_error("Cannot make a BVH node without hittables.");
    if (hs.size() == 1) {
      leaf = true;
      bbox = hs[0]->bounding_box();
      children = hs;
    } else if (hs.size() == 2) {
      leaf = true;
      bbox = hs[0]->bounding_box().merge(hs[1]->bounding_box());
      children = hs;
    } else {
      AABB b = AABB::empty();
      for (auto &h : hs) b = b.merge(h->bounding_box());
      bbox = b;
      int axis = find_longest_axis(b);
      std::vector<std::pair<double, HittablePtr>> keyed;
      keyed.reserve(hs.size());
      for (auto &h : hs) {
        AABB hb = h->bounding_box();
        keyed.emplace_back(axis == 0 ? hb.x.min : axis == 1 ? hb.y.min : hb.z.min, h);
      }
      std::stable_sort(keyed.begin(), keyed.end(), [](const auto &l, const auto &r) { return total_less(l.first, r.first); });
      size_t mid = keyed.size() / 2;
      std::vector<HittablePtr> ls, rs;
      for (size_t i = 0; i < keyed.size(); i++) (i < mid ? ls : rs).push_back(keyed[i].second);
      children.push_back(std::make_shared<Bvh>(std::move(ls)));
      children.push_back(std::make_shared<Bvh>(std::move(rs)));
    }
  }
  AABB bounding_box() const override { return bbox; }
  rl_href flatten(Flattened &f) const override {
    uint32_t idx = (uint32_t)f.bvh_nodes.size();
    f.bvh_nodes.push_back(rl_bvh_node{});
    rl_bvh_node n{};
    n.bbox[0] = bbox.x.min, n.bbox[1] = bbox.x.max, n.bbox[2] = bbox.y.min, n.bbox[3] = bbox.y.max, n.bbox[4] = bbox.z.min, n.bbox[5] = bbox.z.max;
    n.n_children = (uint32_t)children.size();
    for (size_t i = 0; i < children.size(); i++) n.child[i] = children[i]->flatten(f);
    f.bvh_nodes[idx] = n;
    return rl_href{RL_H_BVH, idx};
  }
};

// Bvh::new evaluated on the DEVICE (rl_bvh_build): same tree as `Bvh` above — same boxes, same
// split axes, same stable order — for worlds where the host recursion takes seconds (1 M spheres).  The hittables are
// flattened first (in the caller's order), then the node records are appended after them.
struct DeviceBvh : Hittable {
  std::vector<HittablePtr> hittables;
  explicit DeviceBvh(std::vector<HittablePtr> hs) : hittables(std::move(hs)) {
    if (hittables.empty()) throw std::runtime_error("Cannot make a BVH node without hittables.");
  }
  AABB bounding_box() const override {
    AABB b = AABB::empty();
    for (auto &h : hittables) b = b.merge(h->bounding_box());
    return b;
  }
  rl_href flatten(Flattened &f) const override {
    std::vector<rl_href> hrefs;
    std::vector<double> boxes;
    hrefs.reserve(hittables.size()), boxes.reserve(hittables.size() * 6);
    for (auto &h : hittables) {
      hrefs.push_back(h->flatten(f));
      AABB b = h->bounding_box();
      for (double v : {b.x.min, b.x.max, b.y.min, b.y.max, b.z.min, b.z.max}) boxes.push_back(v);
    }
    uint32_t base = (uint32_t)f.bvh_nodes.size(), count = 0;
    f.bvh_nodes.resize((size_t)base + 2 * hittables.size());
    int rc = rl_bvh_build(boxes.data(), hrefs.data(), (uint32_t)hittables.size(), base, f.bvh_nodes.data() + base, (uint32_t)(2 * hittables.size()), &count);
    if (rc != RL_OK) throw std::runtime_error(std::string("rl_bvh_build: ") + rl_last_error());
    f.bvh_nodes.resize((size_t)base + count);
    return rl_href{RL_H_BVH, base};
  }
};

// ---------------------------------------------------------------- camera.rs
struct CameraParams {  // camera.rs:23-59 (defaults as in the reference)
  double aspect_ratio = 1.0;
  size_t image_width = 100;
  size_t samples_per_pixel = 10;
  size_t max_depth = 10;
  double vfov = 90.0;
  Point3 lookfrom = Point3(0, 0, 0);
  Point3 lookat = Point3(0, 0, -1);
  Vec3 vup = Vec3(0, 1, 0);
  double defocus_angle = 0.0;
  double focus_dist = 10.0;
  Color background = Color(0.7, 0.8, 1.0);
  uint64_t seed = 0;
};

inline double degrees_to_radians(double degrees) { return degrees * M_PI / 180.0; }  // utility.rs:1-3

struct Canvas {  // camera.rs:263-296: data holds SUMS over samples
  size_t samples = 0, width = 0, height = 0;
  std::vector<double> data;  // W*H*3
  Canvas merge(const Canvas &o) const {  // camera.rs:273-291
    if (width != o.width || height != o.height || data.size() != o.data.size()) throw std::runtime_error("Canvas::merge: size mismatch");
    Canvas c{samples + o.samples, width, height, std::vector<double>(data.size())};
    for (size_t i = 0; i < data.size(); i++) c.data[i] = data[i] + o.data[i];
    return c;
  }
};

// Checkpoint codec (examples/common/mod.rs:23-71): `bincode::serialize(&canvas)` with bincode 1.3.3 defaults —
// little-endian, fixed-width integers: u64 samples, u64 width, u64 height, u64 data.len(), then len x [f64; 3]
// (Color is a newtype over [f64; 3]: no per-element length).
inline std::vector<uint8_t> canvas_to_bincode(const Canvas &c) {
  std::vector<uint8_t> out(32 + c.data.size() * 8);
  uint64_t hdr[4] = {(uint64_t)c.samples, (uint64_t)c.width, (uint64_t)c.height, (uint64_t)(c.data.size() / 3)};
  std::memcpy(out.data(), hdr, 32);
  if (!c.data.empty()) std::memcpy(out.data() + 32, c.data.data(), c.data.size() * 8);
  return out;
}
inline Canvas canvas_from_bincode(const uint8_t *bytes, size_t len) {
  if (len < 32) throw std::runtime_error("checkpoint too short");
  uint64_t hdr[4];
  std::memcpy(hdr, bytes, 32);
  if (hdr[3] > (len - 32) / 24 || 32 + hdr[3] * 24 != len) throw std::runtime_error("checkpoint length does not match its pixel count");
  Canvas c{(size_t)hdr[0], (size_t)hdr[1], (size_t)hdr[2], std::vector<double>((size_t)hdr[3] * 3)};
  if (hdr[3]) std::memcpy(c.data.data(), bytes + 32, (size_t)hdr[3] * 24);
  return c;
}

struct Camera {
  CameraParams params;
  size_t image_height;
  Point3 pixel_00_location;
  Vec3 pixel_du, pixel_dv, defocus_disk_u, defocus_disk_v;

  explicit Camera(const CameraParams &p) : params(p) {  // Camera::new camera.rs:72-118
    size_t image_width = params.image_width;
    image_height = std::max<size_t>((size_t)((double)image_width / params.aspect_ratio), 1);
    const Point3 &camera_center = params.lookfrom;
    double theta = degrees_to_radians(params.vfov);
    double h = std::tan(theta / 2.0);
    double viewport_height = 2.0 * h * params.focus_dist;
    double viewport_width = viewport_height * ((double)image_width / (double)image_height);
    Vec3 w = try_normalize(camera_center - params.lookat, "camera w");
    Vec3 u = try_normalize(params.vup.cross(w), "camera u");
    Vec3 v = try_normalize(w.cross(u), "camera v");
    Vec3 viewport_u = viewport_width * u;
    Vec3 viewport_v = viewport_height * (-v);
    pixel_du = viewport_u / (double)image_width;
    pixel_dv = viewport_v / (double)image_height;
    Vec3 viewport_upper_left = camera_center - (params.focus_dist * w) - viewport_u / 2.0 - viewport_v / 2.0;
    pixel_00_location = viewport_upper_left + 0.5 * (pixel_du + pixel_dv);
    double defocus_radius = params.focus_dist * std::tan(degrees_to_radians(params.defocus_angle / 2.0));
    defocus_disk_u = u * defocus_radius;
    defocus_disk_v = v * defocus_radius;
  }

  rl_rtiow_camera derived() const {
    rl_rtiow_camera c{};
    c.image_width = (uint32_t)params.image_width;
    c.image_height = (uint32_t)image_height;
    c.samples_per_pixel = (uint32_t)params.samples_per_pixel;
    c.max_depth = (uint32_t)params.max_depth;
    put3(c.lookfrom, params.lookfrom);
    put3(c.pixel_00, pixel_00_location);
    put3(c.pixel_du, pixel_du);
    put3(c.pixel_dv, pixel_dv);
    put3(c.defocus_disk_u, defocus_disk_u);
    put3(c.defocus_disk_v, defocus_disk_v);
    c.defocus_angle = params.defocus_angle;
    put3(c.background, params.background);
    c.seed = params.seed;
    return c;
  }

  // Camera::render / render_from_checkpoint on the GPU through the C ABI (defined in host_render.cpp)
  Canvas render(const Hittable &world) const;
  Canvas render_from_checkpoint(const Hittable &world, const Canvas &checkpoint) const;
  Canvas render_internal(uint64_t samples_already_rendered, const Hittable &world) const;
};

// ---------------------------------------------------------------- color.rs / output.rs
namespace srgb {  // color.rs:114-136
inline double srgb_to_linear(double u) { return u <= 0.04045 ? u / 12.92 : std::pow((u + 0.055) / (1.0 + 0.055), 2.4); }
inline double linear_to_srgb(double v) { return v <= 0.0031308 ? 12.92 * v : (1.0 + 0.055) * std::pow(v, 1.0 / 2.4) - 0.055; }
}  // namespace srgb

inline int channel_to_u8(double val) {  // color.rs:47-50: floor(val*255.999) as i16 (saturating), clamp 0..255
  double f = std::floor(val * 255.999);
  int n;
  if (std::isnan(f)) n = 0;
  else if (f >= 32767.0) n = 32767;
  else if (f <= -32768.0) n = -32768;
  else n = (int)f;
  return n < 0 ? 0 : n > 255 ? 255 : n;
}

// output::output_ppm (output.rs:5-14) over Canvas::pixel_data (camera.rs:293: c / samples)
inline std::string output_ppm(const double *rgb_sum, size_t width, size_t height, size_t samples) {
  std::string out = "P3\n" + std::to_string(width) + " " + std::to_string(height) + "\n255\n";
  out.reserve(out.size() + width * height * 12);
  double inv = 1.0 / (double)samples;
  for (size_t i = 0; i < width * height; i++) {
    int c[3];
    for (int k = 0; k < 3; k++) c[k] = channel_to_u8(srgb::linear_to_srgb(rgb_sum[3 * i + k] * inv));
    out += std::to_string(c[0]) + " " + std::to_string(c[1]) + " " + std::to_string(c[2]) + "\n";
  }
  return out;
}
inline std::string output_ppm(const Canvas &c) { return output_ppm(c.data.data(), c.width, c.height, c.samples); }

}  // namespace rtiow
