// Scene definitions used as INPUTS by tests and bench: the reference's own test scenes and
// example scenes, restated (values, not code) on top of rtiow_host.hpp / rtc_host.hpp.
//   golden_test_scene      <- ray-tracing-one-weekend/tests/ray_tracing_one_weekend.rs:14-75
//   bouncing_spheres       <- ray-tracing-one-weekend/examples/bouncing_spheres.rs:15-134
//   rtc_test_obj_scene     <- ray-tracer-challenge/tests/ray_tracer.rs:242-275
//   rtiow WavefrontObj     <- ray-tracing-one-weekend/src/io/wavefront_obj.rs
//   cow_scene              <- ray-tracing-one-weekend/examples/cow.rs:32-136
//   checkered_spheres, quads_scene, cornell_scene, teapot_scene, final_scene, make_box
//                          <- examples/checkered_spheres.rs, quads.rs, flat_world.rs, cornell_box.rs, cornell_smoke.rs, teapot.rs,
//                             final_scene.rs, common/mod.rs:72-128
#pragma once
#include "rtc_host.hpp"
#include "rtiow_host.hpp"

namespace scenes {

// std::f64::consts (M_PI/3.0 is NOT FRAC_PI_3 in binary64 — differs by one ulp)
static const double FRAC_PI_2 = 1.57079632679489661923132169163975144;
static const double FRAC_PI_3 = 1.04719755119659774615421446109316763;
static const double FRAC_PI_6 = 0.52359877559829887307710723054658381;

// rand_xoshiro 0.6.0 Xoshiro256PlusPlus with SplitMix64 seed_from_u64; rand 0.8.5 Standard f64 and
// gen_range(lo..hi).  UNPINNED by any reference test: published algorithm only.
struct Xoshiro256PlusPlus {
  uint64_t s[4];
  static Xoshiro256PlusPlus seed_from_u64(uint64_t x) {
    Xoshiro256PlusPlus r;
    for (int i = 0; i < 4; i++) {
      x += 0x9e3779b97f4a7c15ull;
      uint64_t z = x;
      z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
      z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
      r.s[i] = z ^ (z >> 31);
    }
    return r;
  }
  static uint64_t rotl(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }
  uint64_t next_u64() {
    uint64_t result = rotl(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return result;
  }
  double gen_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
  double gen_range(double lo, double hi) {
    double scale = hi - lo;
    for (;;) {
      uint64_t bits = (next_u64() >> 12) | 0x3FF0000000000000ull;
      double v12;
      std::memcpy(&v12, &bits, 8);
      double res = (v12 - 1.0) * scale + lo;
      if (res < hi) return res;
    }
  }
};

struct RtiowScene {
  std::shared_ptr<rtiow::Hittable> world;
  rtiow::CameraParams params;
};

inline RtiowScene golden_test_scene() {
  using namespace rtiow;
  std::vector<HittablePtr> w;
  w.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(0.0, -100.5, -1.0)), 100.0, Lambertian(SolidColor(Color(0.8, 0.8, 0.0)))));
  w.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(0.0, 0.0, -1.2)), 0.5, Lambertian(SolidColor(Color(0.1, 0.2, 0.5)))));
  w.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(-1.0, 0.0, -1.0)), 0.5, Dielectric(1.5)));
  w.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(-1.0, 0.0, -1.0)), 0.4, Dielectric(1.0 / 1.5)));
  w.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(1.0, 0.0, -1.0)), 0.5, Metal(Color(0.8, 0.6, 0.2), 1.0)));
  CameraParams p;
  p.aspect_ratio = 16.0 / 9.0;
  p.image_width = 300;
  p.samples_per_pixel = 10;
  p.max_depth = 10;
  p.vfov = 20.0;
  p.lookfrom = Point3(-2.0, 2.0, 1.0);
  p.lookat = Point3(0.0, 0.0, -1.0);
  p.vup = Vec3(0.0, 1.0, 0.0);
  p.defocus_angle = 10.0;
  p.focus_dist = 3.4;
  p.background = Color(0.7, 0.8, 1.0);
  p.seed = 0;
  return RtiowScene{std::make_shared<HittableList>(std::move(w)), p};
}

// grid_half = 11 reproduces the example; the camera is the example's with max_depth left to the caller
inline RtiowScene bouncing_spheres(uint64_t master_seed = 1) {
  using namespace rtiow;
  auto rng = Xoshiro256PlusPlus::seed_from_u64(master_seed);
  std::vector<HittablePtr> world;
  auto checker = Checker(0.32, SolidColor(Color(0.2, 0.23, 0.1)), SolidColor(Color(0.9, 0.9, 0.9)));
  world.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(0.0, -1000.0, 0.0)), 1000.0, Lambertian(checker)));
  for (int a = -11; a < 11; a++)
    for (int b = -11; b < 11; b++) {
      double choose_mat = rng.gen_f64();
      double cx = (double)a + 0.9 * rng.gen_f64();
      double cz = (double)b + 0.9 * rng.gen_f64();
      Point3 center_point(cx, 0.2, cz);
      if ((center_point - Point3(4.0, 0.2, 0.0)).length() > 0.9) {
        if (choose_mat < 0.8) {
          Point3 center2 = center_point + Vec3(0.0, rng.gen_range(0.0, 0.5), 0.0);
          double r1 = rng.gen_f64(), g1 = rng.gen_f64(), b1 = rng.gen_f64();
          double r2 = rng.gen_f64(), g2 = rng.gen_f64(), b2 = rng.gen_f64();
          Color albedo = Color(r1, g1, b1) * Color(r2, g2, b2);
          world.push_back(std::make_shared<Sphere>(Center::Moving(center_point, center2), 0.2, Lambertian(SolidColor(albedo))));
        } else if (choose_mat < 0.95) {
          double r = rng.gen_range(0.5, 1.0), g = rng.gen_range(0.5, 1.0), bb = rng.gen_range(0.5, 1.0);
          double fuzz = rng.gen_f64();
          world.push_back(std::make_shared<Sphere>(Center::Stationary(center_point), 0.2, Metal(Color(r, g, bb), fuzz)));
        } else {
          world.push_back(std::make_shared<Sphere>(Center::Stationary(center_point), 0.2, Dielectric(1.5)));
        }
      }
    }
  world.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(0.0, 1.0, 0.0)), 1.0, Dielectric(1.5)));
  world.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(-4.0, 1.0, 0.0)), 1.0, Lambertian(SolidColor(Color(0.4, 0.2, 0.1)))));
  world.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(4.0, 1.0, 0.0)), 1.0, Metal(Color(0.7, 0.6, 0.5), 0.0)));
  CameraParams p;
  p.aspect_ratio = 16.0 / 9.0;
  p.image_width = 400;
  p.samples_per_pixel = 100;
  p.max_depth = 10;
  p.vfov = 20.0;
  p.lookfrom = Point3(13.0, 2.0, 3.0);
  p.lookat = Point3(0.0, 0.0, 0.0);
  p.vup = Vec3(0.0, 1.0, 0.0);
  p.defocus_angle = 0.6;
  p.focus_dist = 10.0;
  return RtiowScene{std::make_shared<Bvh>(std::move(world)), p};
}

// ------------------------------------------------------------------ RTIOW OBJ loader (io/wavefront_obj.rs)
// v / vt / vn / f / g; fan triangulation; per-triangle Option<uvs>, Option<normals>; -> Bvh<Triangle>
struct RtiowObj {
  struct Tri {
    rtiow::Point3 p[3];
    bool has_uv, has_n;
    double uv[6];
    rtiow::Vec3 n[3];
  };
  std::vector<std::pair<std::string, std::vector<Tri>>> groups;
  std::vector<rtiow::Point3> vertices;
  std::vector<rtiow::Vec3> normals;
  std::vector<std::pair<double, double>> texcoords;
  uint32_t ignored = 0;

  static RtiowObj parse(const std::string &content) {
    RtiowObj obj;
    std::string current_name = "\x01" "default";
    std::vector<Tri> current;
    auto commit = [&](const std::string &name, std::vector<Tri> &&val) {
      for (auto &g : obj.groups)
        if (g.first == name) {
          g.second = std::move(val);
          return;
        }
      obj.groups.emplace_back(name, std::move(val));
    };
    std::istringstream in(content);
    std::string line;
    while (std::getline(in, line)) {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      size_t sp = line.find(' ');
      bool ok = false;
      if (sp != std::string::npos) {
        std::string head = line.substr(0, sp), tail = line.substr(sp + 1);
        size_t a = tail.find_first_not_of(" \t\r\n\f\v"), b = tail.find_last_not_of(" \t\r\n\f\v");
        std::string trimmed = a == std::string::npos ? "" : tail.substr(a, b - a + 1);
        std::vector<double> ns;
        if (head == "v") {
          if (rtc::WavefrontObj::parse_floats(trimmed, ns) && ns.size() == 3) obj.vertices.emplace_back(ns[0], ns[1], ns[2]), ok = true;
        } else if (head == "vn") {
          if (rtc::WavefrontObj::parse_floats(trimmed, ns) && ns.size() == 3) obj.normals.emplace_back(ns[0], ns[1], ns[2]), ok = true;
        } else if (head == "vt") {  // :107-118: 1 value -> (u,0); >=2 -> (u,v)
          if (rtc::WavefrontObj::parse_floats(trimmed, ns) && !ns.empty()) obj.texcoords.emplace_back(ns[0], ns.size() > 1 ? ns[1] : 0.0), ok = true;
        } else if (head == "f") {
          std::vector<Tri> ts;
          if (obj.parse_face(trimmed, ts)) current.insert(current.end(), ts.begin(), ts.end()), ok = true;
        } else if (head == "g") {
          commit(current_name, std::move(current));
          current.clear();
          current_name = trimmed;
          ok = true;
        }
      }
      if (!ok) obj.ignored++;
    }
    commit(current_name, std::move(current));
    return obj;
  }
  bool parse_face(const std::string &tail, std::vector<Tri> &out) const {  // :150-240
    struct VTN {
      size_t v;
      bool ht, hn;
      size_t t, n;
    };
    std::vector<VTN> idx;
    std::istringstream ss(tail);
    std::string token;
    while (ss >> token) {
      std::vector<std::string> parts;
      size_t start = 0;
      for (;;) {
        size_t p = token.find('/', start);
        if (p == std::string::npos) {
          parts.push_back(token.substr(start));
          break;
        }
        parts.push_back(token.substr(start, p - start));
        start = p + 1;
      }
      if (parts.size() < 1 || parts.size() > 3) return false;
      VTN e{0, false, false, 0, 0};
      if (!rtc::WavefrontObj::parse_usize(parts[0], e.v)) return false;
      if (parts.size() >= 2 && !parts[1].empty()) e.ht = rtc::WavefrontObj::parse_usize(parts[1], e.t);  // bad index -> None
      if (parts.size() == 3) e.hn = rtc::WavefrontObj::parse_usize(parts[2], e.n);
      idx.push_back(e);
    }
    for (auto &e : idx) {
      if (e.v < 1 || e.v > vertices.size()) throw std::runtime_error("obj: vertex index out of range");
      if (e.ht && (e.t < 1 || e.t > texcoords.size())) throw std::runtime_error("obj: texcoord index out of range");
      if (e.hn && (e.n < 1 || e.n > normals.size())) throw std::runtime_error("obj: normal index out of range");
    }
    if (idx.size() < 3) return false;
    for (size_t i = 2; i < idx.size(); i++) {
      const VTN *vs[3] = {&idx[0], &idx[i - 1], &idx[i]};
      Tri t{};
      for (int k = 0; k < 3; k++) t.p[k] = vertices[vs[k]->v - 1];
      t.has_uv = vs[0]->ht && vs[1]->ht && vs[2]->ht;
      t.has_n = vs[0]->hn && vs[1]->hn && vs[2]->hn;
      if (t.has_uv)
        for (int k = 0; k < 3; k++) t.uv[2 * k] = texcoords[vs[k]->t - 1].first, t.uv[2 * k + 1] = texcoords[vs[k]->t - 1].second;
      if (t.has_n)
        for (int k = 0; k < 3; k++) t.n[k] = normals[vs[k]->n - 1];
      out.push_back(t);
    }
    return true;
  }
  std::shared_ptr<rtiow::Bvh> to_object(rtiow::MaterialPtr material) const {  // :84-100
    std::vector<rtiow::HittablePtr> tris;
    for (auto &g : groups)
      for (auto &t : g.second) tris.push_back(std::make_shared<rtiow::Triangle>(t.p, t.has_uv ? t.uv : nullptr, t.has_n ? t.n : nullptr, material));
    return std::make_shared<rtiow::Bvh>(std::move(tris));
  }
};

// examples/cow.rs scene. rgb8: the decoded spot_texture.png (w*h*3 bytes, top row first).
// image crate into_rgb32f: u8 -> f32 via (v as f32)/255.0; then srgb_to_linear(u as f64) as f32 (cow.rs:27-29).
inline RtiowScene cow_scene(const std::string &obj_text, const uint8_t *rgb8, uint32_t tw, uint32_t th) {
  using namespace rtiow;
  auto img = std::make_shared<ImageData>();
  img->width = tw, img->height = th;
  img->rgb.resize((size_t)tw * th * 3);
  for (size_t i = 0; i < img->rgb.size(); i++) {
    float u = (float)rgb8[i] / 255.0f;
    img->rgb[i] = (float)srgb::srgb_to_linear((double)u);
  }
  auto cow_surface = Lambertian(Image(img));
  auto cow = RtiowObj::parse(obj_text).to_object(cow_surface);
  HittablePtr transformed_cow =
      std::make_shared<Translate>(Transform::rotate_y(Transform::scale(cow, 200.0), 45.0), Vec3(240.0, 165.0, 240.0));
  auto red = Lambertian(SolidColor(Color(0.65, 0.05, 0.05)));
  auto white = Lambertian(SolidColor(Color(0.73, 0.73, 0.73)));
  auto green = Lambertian(SolidColor(Color(0.12, 0.45, 0.15)));
  auto light = DiffuseLight(SolidColor(Color(5.0, 5.0, 5.0)));
  std::vector<HittablePtr> world;
  world.push_back(std::make_shared<Quad>(Point3(555, 0, 0), Vec3(0, 555, 0), Vec3(0, 0, 555), green));
  world.push_back(std::make_shared<Quad>(Point3(0, 0, 0), Vec3(0, 555, 0), Vec3(0, 0, 555), red));
  world.push_back(std::make_shared<Quad>(Point3(113, 554, 127), Vec3(330, 0, 0), Vec3(0, 0, 305), light));
  world.push_back(std::make_shared<Quad>(Point3(0, 0, 0), Vec3(555, 0, 0), Vec3(0, 0, 555), white));
  world.push_back(std::make_shared<Quad>(Point3(555, 555, 555), Vec3(-555, 0, 0), Vec3(0, 0, -555), white));
  world.push_back(std::make_shared<Quad>(Point3(0, 0, 555), Vec3(555, 0, 0), Vec3(0, 555, 0), white));
  world.push_back(transformed_cow);
  CameraParams p;
  p.aspect_ratio = 1.0;
  p.image_width = 600;
  p.samples_per_pixel = 200;
  p.max_depth = 40;
  p.background = Color(0.0, 0.0, 0.0);
  p.vfov = 40.0;
  p.lookfrom = Point3(278.0, 278.0, -800.0);
  p.lookat = Point3(278.0, 278.0, 0.0);
  p.vup = Vec3(0.0, 1.0, 0.0);
  p.defocus_angle = 0.0;
  return RtiowScene{std::make_shared<Bvh>(std::move(world)), p};
}

// examples/perlin_spheres.rs (light = false) and examples/simple_light.rs (light = true): two spheres with one shared
// Lambertian{Noise{Perlin::new(Xoshiro256PlusPlus::seed_from_u64(1)), scale 4}}, plus a quad and a sphere light.
inline RtiowScene perlin_scene(bool light) {
  using namespace rtiow;
  auto rng = Xoshiro256PlusPlus::seed_from_u64(1);
  auto material = Lambertian(Noise(Perlin::create(rng), 4.0));
  std::vector<HittablePtr> world;
  world.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(0.0, -1000.0, 0.0)), 1000.0, material));
  world.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(0.0, 2.0, 0.0)), 2.0, material));
  CameraParams p;
  p.aspect_ratio = 16.0 / 9.0;
  p.image_width = 400;
  p.samples_per_pixel = 100;
  p.max_depth = 50;
  p.vfov = 20.0;
  p.lookfrom = Point3(13.0, 2.0, 3.0);
  p.lookat = Point3(0.0, 0.0, 0.0);
  p.vup = Vec3(0.0, 1.0, 0.0);
  p.defocus_angle = 0.0;
  if (light) {
    auto light_material = DiffuseLight(SolidColor(Color(4.0, 4.0, 4.0)));
    world.push_back(std::make_shared<Quad>(Point3(3.0, 1.0, -2.0), Vec3(2.0, 0.0, 0.0), Vec3(0.0, 2.0, 0.0), light_material));
    world.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(0.0, 7.0, 0.0)), 2.0, light_material));
    p.background = Color(0.0, 0.0, 0.0);
    p.lookfrom = Point3(26.0, 3.0, 6.0);
    p.lookat = Point3(0.0, 2.0, 0.0);
  }
  return RtiowScene{std::make_shared<HittableList>(std::move(world)), p};
}

// examples/earth.rs: one sphere of radius 2 with Lambertian{Image}; the image is the caller's (sRGB8 -> linear f32 as the
// example converts it; earthmap.jpg itself is not decoded here).
inline RtiowScene earth_scene(const uint8_t *rgb8, uint32_t tw, uint32_t th) {
  using namespace rtiow;
  auto img = std::make_shared<ImageData>();
  img->width = tw, img->height = th;
  img->rgb.resize((size_t)tw * th * 3);
  for (size_t i = 0; i < img->rgb.size(); i++) img->rgb[i] = (float)srgb::srgb_to_linear((double)((float)rgb8[i] / 255.0f));
  HittablePtr globe = std::make_shared<Sphere>(Center::Stationary(Point3(0.0, 0.0, 0.0)), 2.0, Lambertian(Image(img)));
  CameraParams p;
  p.aspect_ratio = 16.0 / 9.0;
  p.image_width = 400;
  p.samples_per_pixel = 100;
  p.max_depth = 50;
  p.vfov = 20.0;
  p.lookfrom = Point3(0.0, 0.0, 12.0);
  p.lookat = Point3(0.0, 0.0, 0.0);
  p.vup = Vec3(0.0, 1.0, 0.0);
  p.defocus_angle = 0.0;
  return RtiowScene{globe, p};
}

// ---- the reference's remaining example scenes (values restated; the example files themselves are drivers, not the hot path).
// examples/common/mod.rs:72-128 make_box: the six sides (front, right, back, left, top, bottom) of the box spanned by two opposite
// vertices, as a `[Quad; 6]` = a list hittable.
inline rtiow::HittablePtr make_box(const rtiow::Point3 &a, const rtiow::Point3 &b, rtiow::MaterialPtr m) {
  using namespace rtiow;
  Point3 lo(std::fmin(a.x(), b.x()), std::fmin(a.y(), b.y()), std::fmin(a.z(), b.z()));
  Point3 hi(std::fmax(a.x(), b.x()), std::fmax(a.y(), b.y()), std::fmax(a.z(), b.z()));
  Vec3 dx(hi.x() - lo.x(), 0.0, 0.0), dy(0.0, hi.y() - lo.y(), 0.0), dz(0.0, 0.0, hi.z() - lo.z());
  std::vector<HittablePtr> q;
  q.push_back(std::make_shared<Quad>(Point3(lo.x(), lo.y(), hi.z()), dx, dy, m));
  q.push_back(std::make_shared<Quad>(Point3(hi.x(), lo.y(), hi.z()), -dz, dy, m));
  q.push_back(std::make_shared<Quad>(Point3(hi.x(), lo.y(), lo.z()), -dx, dy, m));
  q.push_back(std::make_shared<Quad>(Point3(lo.x(), lo.y(), lo.z()), dz, dy, m));
  q.push_back(std::make_shared<Quad>(Point3(lo.x(), hi.y(), hi.z()), dx, -dz, m));
  q.push_back(std::make_shared<Quad>(Point3(lo.x(), lo.y(), lo.z()), dx, dz, m));
  return std::make_shared<HittableList>(std::move(q));
}

inline rtiow::CameraParams cornell_camera() {  // cornell_box.rs:95-107 (= cornell_smoke.rs, cow.rs up to max_depth)
  rtiow::CameraParams p;
  p.aspect_ratio = 1.0, p.image_width = 600, p.samples_per_pixel = 200, p.max_depth = 50;
  p.background = rtiow::Color(0.0, 0.0, 0.0);
  p.vfov = 40.0;
  p.lookfrom = rtiow::Point3(278.0, 278.0, -800.0), p.lookat = rtiow::Point3(278.0, 278.0, 0.0), p.vup = rtiow::Vec3(0.0, 1.0, 0.0);
  p.defocus_angle = 0.0;
  return p;
}

// examples/checkered_spheres.rs: two radius-10 spheres sharing one Checker Lambertian, a plain slice
inline RtiowScene checkered_spheres() {
  using namespace rtiow;
  auto lambertian = Lambertian(Checker(0.32, SolidColor(Color(0.2, 0.3, 0.1)), SolidColor(Color(0.9, 0.9, 0.9))));
  std::vector<HittablePtr> w;
  w.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(0.0, -10.0, 0.0)), 10.0, lambertian));
  w.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(0.0, 10.0, 0.0)), 10.0, lambertian));
  CameraParams p;
  p.aspect_ratio = 16.0 / 9.0, p.image_width = 400, p.samples_per_pixel = 100, p.max_depth = 50, p.vfov = 20.0;
  p.lookfrom = Point3(13.0, 2.0, 3.0), p.lookat = Point3(0.0, 0.0, 0.0), p.vup = Vec3(0.0, 1.0, 0.0), p.defocus_angle = 0.0;
  return RtiowScene{std::make_shared<HittableList>(std::move(w)), p};
}

// examples/quads.rs (flat = false): five quads; examples/flat_world.rs (flat = true): the same room with the back wall a Triangle and
// the floor an unbounded Plane with a Checker
inline RtiowScene quads_scene(bool flat) {
  using namespace rtiow;
  auto left_red = Lambertian(SolidColor(Color(1.0, 0.2, 0.2)));
  auto back_green = Lambertian(SolidColor(Color(0.2, 1.0, 0.2)));
  auto right_blue = Lambertian(SolidColor(Color(0.2, 0.2, 1.0)));
  auto upper_orange = Lambertian(SolidColor(Color(1.0, 0.5, 0.0)));
  std::vector<HittablePtr> w;
  w.push_back(std::make_shared<Quad>(Point3(-3.0, -2.0, 5.0), Vec3(0.0, 0.0, -4.0), Vec3(0.0, 4.0, 0.0), left_red));
  if (flat) w.push_back(Triangle::from_quv(Point3(-2.0, -2.0, 0.0), Vec3(4.0, 0.0, 0.0), Vec3(0.0, 4.0, 0.0), back_green));
  else w.push_back(std::make_shared<Quad>(Point3(-2.0, -2.0, 0.0), Vec3(4.0, 0.0, 0.0), Vec3(0.0, 4.0, 0.0), back_green));
  w.push_back(std::make_shared<Quad>(Point3(3.0, -2.0, 1.0), Vec3(0.0, 0.0, 4.0), Vec3(0.0, 4.0, 0.0), right_blue));
  w.push_back(std::make_shared<Quad>(Point3(-2.0, 3.0, 1.0), Vec3(4.0, 0.0, 0.0), Vec3(0.0, 0.0, 4.0), upper_orange));
  if (flat) {
    auto lower_checker = Lambertian(Checker(1.0, SolidColor(Color(0.2, 0.8, 0.8)), SolidColor(Color(0.8, 0.8, 0.8))));
    w.push_back(std::make_shared<Plane>(Point3(-2.0, -3.0, 5.0), Vec3(4.0, 0.0, 0.0), Vec3(0.0, 0.0, -4.0), lower_checker));
  } else {
    auto lower_teal = Lambertian(SolidColor(Color(0.2, 0.8, 0.8)));
    w.push_back(std::make_shared<Quad>(Point3(-2.0, -3.0, 5.0), Vec3(4.0, 0.0, 0.0), Vec3(0.0, 0.0, -4.0), lower_teal));
  }
  CameraParams p;
  p.aspect_ratio = 1.0, p.image_width = 400, p.samples_per_pixel = 100, p.max_depth = 50, p.vfov = 80.0;
  p.lookfrom = Point3(0.0, 0.0, 9.0), p.lookat = Point3(0.0, 0.0, 0.0), p.vup = Vec3(0.0, 1.0, 0.0), p.defocus_angle = 0.0;
  return RtiowScene{std::make_shared<HittableList>(std::move(w)), p};
}

// examples/cornell_box.rs (smoke = false) / examples/cornell_smoke.rs (smoke = true: the two boxes become the boundaries of constant
// media of density 0.01 — evaluated with the pixel's RNG stream, include/rl_render.h rl_medium); Bvh::new over everything
inline RtiowScene cornell_scene(bool smoke) {
  using namespace rtiow;
  auto red = Lambertian(SolidColor(Color(0.65, 0.05, 0.05)));
  auto white = Lambertian(SolidColor(Color(0.73, 0.73, 0.73)));
  auto green = Lambertian(SolidColor(Color(0.12, 0.45, 0.15)));
  auto light = DiffuseLight(SolidColor(smoke ? Color(7.0, 7.0, 7.0) : Color(15.0, 15.0, 15.0)));
  std::vector<HittablePtr> w;
  w.push_back(std::make_shared<Quad>(Point3(555, 0, 0), Vec3(0, 555, 0), Vec3(0, 0, 555), green));
  w.push_back(std::make_shared<Quad>(Point3(0, 0, 0), Vec3(0, 555, 0), Vec3(0, 0, 555), red));
  if (smoke) {
    w.push_back(std::make_shared<Quad>(Point3(113, 554, 127), Vec3(330, 0, 0), Vec3(0, 0, 305), light));
    w.push_back(std::make_shared<Quad>(Point3(0, 555, 0), Vec3(555, 0, 0), Vec3(0, 0, 555), white));
    w.push_back(std::make_shared<Quad>(Point3(0, 0, 0), Vec3(555, 0, 0), Vec3(0, 0, 555), white));
  } else {
    w.push_back(std::make_shared<Quad>(Point3(343, 554, 332), Vec3(-130, 0, 0), Vec3(0, 0, -105), light));
    w.push_back(std::make_shared<Quad>(Point3(0, 0, 0), Vec3(555, 0, 0), Vec3(0, 0, 555), white));
    w.push_back(std::make_shared<Quad>(Point3(555, 555, 555), Vec3(-555, 0, 0), Vec3(0, 0, -555), white));
  }
  w.push_back(std::make_shared<Quad>(Point3(0, 0, 555), Vec3(555, 0, 0), Vec3(0, 555, 0), white));
  HittablePtr box1 = std::make_shared<Translate>(Transform::rotate_y(make_box(Point3(0, 0, 0), Point3(165, 330, 165), white), 15.0), Vec3(265, 0, 295));
  HittablePtr box2 = std::make_shared<Translate>(Transform::rotate_y(make_box(Point3(0, 0, 0), Point3(165, 165, 165), white), -18.0), Vec3(130, 0, 65));
  if (smoke) {
    w.push_back(std::make_shared<ConstantMedium>(box1, 0.01, Isotropic(SolidColor(Color(0.0, 0.0, 0.0)))));
    w.push_back(std::make_shared<ConstantMedium>(box2, 0.01, Isotropic(SolidColor(Color(1.0, 1.0, 1.0)))));
  } else {
    w.push_back(box1);
    w.push_back(box2);
  }
  return RtiowScene{std::make_shared<Bvh>(std::move(w)), cornell_camera()};
}

// examples/teapot.rs: the OBJ under scale(2) -> rotate_x(-90) next to a sphere light, Bvh::new over the two
inline RtiowScene teapot_scene(const std::string &obj_text) {
  using namespace rtiow;
  auto diffuse = Lambertian(SolidColor(Color(0.53, 0.32, 0.75)));
  auto light = DiffuseLight(SolidColor(Color(5.0, 5.0, 5.0)));
  auto teapot = RtiowObj::parse(obj_text).to_object(diffuse);
  std::vector<HittablePtr> w;
  w.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(40.0, 20.0, -40.0)), 10.0, light));
  w.push_back(Transform::rotate_x(Transform::scale(teapot, 2.0), -90.0));
  CameraParams p;
  p.aspect_ratio = 1.0, p.image_width = 600, p.samples_per_pixel = 200, p.max_depth = 40, p.vfov = 40.0;
  p.lookfrom = Point3(0.0, 0.0, -100.0), p.lookat = Point3(0.0, 0.0, 0.0), p.vup = Vec3(0.0, 1.0, 0.0), p.defocus_angle = 0.0;
  p.background = Color(0.1, 0.1, 0.1);
  return RtiowScene{std::make_shared<Bvh>(std::move(w)), p};
}

// examples/final_scene.rs ("The Next Week"): Xoshiro256++ seed 0 drives, in this order, Perlin::new, the 400 floor-box heights
// (gen_range(1.0..101.0)) and the 1000 sphere centres (Point3::random_in_range(0, 165): x, y, z).  The earth image is the caller's
// (sRGB8; the example embeds earthmap.jpg, JPEG decoding is out of scope).  The world is a plain slice of boxed hittables.
inline RtiowScene final_scene(const uint8_t *rgb8, uint32_t tw, uint32_t th) {
  using namespace rtiow;
  auto rng = Xoshiro256PlusPlus::seed_from_u64(0);
  auto ground_material = Lambertian(SolidColor(Color(0.48, 0.83, 0.53)));
  auto light_material = DiffuseLight(SolidColor(Color(7.0, 7.0, 7.0)));
  auto sphere_material = Lambertian(SolidColor(Color(0.7, 0.3, 0.1)));
  auto glass_material = Dielectric(1.5);
  auto metal_material = Metal(Color(0.8, 0.8, 0.9), 1.0);
  auto subsurface_material = Isotropic(SolidColor(Color(0.2, 0.4, 0.9)));
  auto fog_material = Isotropic(SolidColor(Color(1.0, 1.0, 1.0)));
  auto img = std::make_shared<ImageData>();
  img->width = tw, img->height = th;
  img->rgb.resize((size_t)tw * th * 3);
  for (size_t i = 0; i < img->rgb.size(); i++) img->rgb[i] = (float)srgb::srgb_to_linear((double)((float)rgb8[i] / 255.0f));
  auto earth_material = Lambertian(Image(img));
  auto perlin_material = Lambertian(Noise(Perlin::create(rng), 0.2));
  auto white_material = Lambertian(SolidColor(Color(0.73, 0.73, 0.73)));
  std::vector<HittablePtr> world;
  std::vector<HittablePtr> boxes;
  for (int i = 0; i < 20; i++)
    for (int j = 0; j < 20; j++) {
      double w = 100.0, x0 = -1000.0 + (double)i * w, z0 = -1000.0 + (double)j * w, y0 = 0.0;
      double x1 = x0 + w, y1 = rng.gen_range(1.0, 101.0), z1 = z0 + w;
      boxes.push_back(make_box(Point3(x0, y0, z0), Point3(x1, y1, z1), ground_material));
    }
  world.push_back(std::make_shared<Bvh>(std::move(boxes)));
  world.push_back(std::make_shared<Quad>(Point3(123.0, 554.0, 147.0), Vec3(300.0, 0.0, 0.0), Vec3(0.0, 0.0, 265.0), light_material));
  Point3 center1(400.0, 400.0, 200.0);
  world.push_back(std::make_shared<Sphere>(Center::Moving(center1, center1 + Vec3(30.0, 0.0, 0.0)), 50.0, sphere_material));
  world.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(260.0, 150.0, 45.0)), 50.0, glass_material));
  world.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(0.0, 150.0, 145.0)), 50.0, metal_material));
  auto subsurface_boundary = [&]() { return std::make_shared<Sphere>(Center::Stationary(Point3(360.0, 150.0, 145.0)), 70.0, glass_material); };
  world.push_back(subsurface_boundary());
  world.push_back(std::make_shared<ConstantMedium>(subsurface_boundary(), 0.2, subsurface_material));
  world.push_back(std::make_shared<ConstantMedium>(std::make_shared<Sphere>(Center::Stationary(Point3(0.0, 0.0, 0.0)), 5000.0, glass_material), 0.0001, fog_material));
  world.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(400.0, 200.0, 400.0)), 100.0, earth_material));
  world.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(220.0, 280.0, 300.0)), 80.0, perlin_material));
  std::vector<HittablePtr> boxes2;
  for (int k = 0; k < 1000; k++) {
    double x = rng.gen_range(0.0, 165.0), y = rng.gen_range(0.0, 165.0), z = rng.gen_range(0.0, 165.0);
    boxes2.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(x, y, z)), 10.0, white_material));
  }
  world.push_back(std::make_shared<Translate>(Transform::rotate_y(std::make_shared<Bvh>(std::move(boxes2)), 15.0), Vec3(-100.0, 270.0, 395.0)));
  CameraParams p;
  p.aspect_ratio = 1.0, p.image_width = 400, p.samples_per_pixel = 250, p.max_depth = 4;  // the example's "dev" parameters
  p.background = Color(0.0, 0.0, 0.0);
  p.vfov = 40.0;
  p.lookfrom = Point3(478.0, 278.0, -600.0), p.lookat = Point3(278.0, 278.0, 0.0), p.vup = Vec3(0.0, 1.0, 0.0), p.defocus_angle = 0.0;
  return RtiowScene{std::make_shared<HittableList>(std::move(world)), p};
}

// BASELINE configs[4] ("1M random spheres + 100k-triangle OBJ"): NOT in the reference — defined by
// SURVEY.md §8d item 5 and frozen here.  n_side = 1000, subdiv = 2 gives 1,000,000 small spheres + the
// ground sphere + spot_triangulated.obj midpoint-subdivided twice (5856 * 16 = 93,696 triangles, UVs
// interpolated) under cow.rs's transform chain scaled x0.05 and centred on the origin:
// scale(10) -> rotate_y(45) -> translate(0, 8.25, 0).  Materials are chosen exactly as
// bouncing_spheres.rs:38-83 (0.8 / 0.95 thresholds, moving diffuse spheres) but without the
// (4,0.2,0) exclusion; Xoshiro256++ seed 5.  Camera: bouncing_spheres' (caller sets W / spp / depth).
// Only the CPU oracle pins this config.
inline RtiowScene stress_scene(int n_side, int subdiv, const std::string &obj_text, const uint8_t *rgb8, uint32_t tw, uint32_t th,
                               uint64_t seed = 5, bool device_bvh = false) {
  using namespace rtiow;
  auto rng = Xoshiro256PlusPlus::seed_from_u64(seed);
  std::vector<HittablePtr> world;
  auto checker = Checker(0.32, SolidColor(Color(0.2, 0.23, 0.1)), SolidColor(Color(0.9, 0.9, 0.9)));
  world.push_back(std::make_shared<Sphere>(Center::Stationary(Point3(0.0, -1e6, 0.0)), 1e6, Lambertian(checker)));
  int half = n_side / 2;
  for (int a = -half; a < n_side - half; a++)
    for (int b = -half; b < n_side - half; b++) {
      double choose_mat = rng.gen_f64();
      double cx = (double)a + 0.9 * rng.gen_f64();
      double cz = (double)b + 0.9 * rng.gen_f64();
      Point3 center_point(cx, 0.2, cz);
      if (choose_mat < 0.8) {
        Point3 center2 = center_point + Vec3(0.0, rng.gen_range(0.0, 0.5), 0.0);
        double r1 = rng.gen_f64(), g1 = rng.gen_f64(), b1 = rng.gen_f64();
        double r2 = rng.gen_f64(), g2 = rng.gen_f64(), b2 = rng.gen_f64();
        world.push_back(std::make_shared<Sphere>(Center::Moving(center_point, center2), 0.2, Lambertian(SolidColor(Color(r1, g1, b1) * Color(r2, g2, b2)))));
      } else if (choose_mat < 0.95) {
        double r = rng.gen_range(0.5, 1.0), g = rng.gen_range(0.5, 1.0), bb = rng.gen_range(0.5, 1.0);
        double fuzz = rng.gen_f64();
        world.push_back(std::make_shared<Sphere>(Center::Stationary(center_point), 0.2, Metal(Color(r, g, bb), fuzz)));
      } else {
        world.push_back(std::make_shared<Sphere>(Center::Stationary(center_point), 0.2, Dielectric(1.5)));
      }
    }
  if (!obj_text.empty()) {
    auto img = std::make_shared<ImageData>();
    img->width = tw, img->height = th;
    img->rgb.resize((size_t)tw * th * 3);
    for (size_t i = 0; i < img->rgb.size(); i++) img->rgb[i] = (float)srgb::srgb_to_linear((double)((float)rgb8[i] / 255.0f));
    auto surface = Lambertian(Image(img));
    RtiowObj obj = RtiowObj::parse(obj_text);
    std::vector<RtiowObj::Tri> tris;
    for (auto &g : obj.groups)
      for (auto &t : g.second) tris.push_back(t);
    for (int s = 0; s < subdiv; s++) {  // midpoint subdivision: (p0,m01,m20) (m01,p1,m12) (m20,m12,p2) (m01,m12,m20)
      std::vector<RtiowObj::Tri> next;
      next.reserve(tris.size() * 4);
      for (auto &t : tris) {
        auto midp = [](const Point3 &x, const Point3 &y) { return (x + y) * 0.5; };
        Point3 m01 = midp(t.p[0], t.p[1]), m12 = midp(t.p[1], t.p[2]), m20 = midp(t.p[2], t.p[0]);
        double u01[2], u12[2], u20[2];
        for (int k = 0; k < 2; k++) {
          u01[k] = (t.uv[k] + t.uv[2 + k]) * 0.5;
          u12[k] = (t.uv[2 + k] + t.uv[4 + k]) * 0.5;
          u20[k] = (t.uv[4 + k] + t.uv[k]) * 0.5;
        }
        auto mk = [&](const Point3 &a, const double *ua, const Point3 &b, const double *ub, const Point3 &c, const double *uc) {
          RtiowObj::Tri n{};
          n.p[0] = a, n.p[1] = b, n.p[2] = c;
          n.has_uv = t.has_uv, n.has_n = false;
          n.uv[0] = ua[0], n.uv[1] = ua[1], n.uv[2] = ub[0], n.uv[3] = ub[1], n.uv[4] = uc[0], n.uv[5] = uc[1];
          next.push_back(n);
        };
        mk(t.p[0], t.uv, m01, u01, m20, u20);
        mk(m01, u01, t.p[1], t.uv + 2, m12, u12);
        mk(m20, u20, m12, u12, t.p[2], t.uv + 4);
        mk(m01, u01, m12, u12, m20, u20);
      }
      tris.swap(next);
    }
    std::vector<HittablePtr> ths;
    for (auto &t : tris) ths.push_back(std::make_shared<Triangle>(t.p, t.has_uv ? t.uv : nullptr, nullptr, surface));
    HittablePtr mesh = device_bvh ? HittablePtr(std::make_shared<DeviceBvh>(std::move(ths))) : HittablePtr(std::make_shared<Bvh>(std::move(ths)));
    world.push_back(std::make_shared<Translate>(Transform::rotate_y(Transform::scale(mesh, 10.0), 45.0), Vec3(0.0, 8.25, 0.0)));
  }
  CameraParams p;
  p.aspect_ratio = 16.0 / 9.0;
  p.image_width = 3840;
  p.samples_per_pixel = 4096;
  p.max_depth = 50;
  p.vfov = 20.0;
  p.lookfrom = Point3(13.0, 2.0, 3.0);
  p.lookat = Point3(0.0, 0.0, 0.0);
  p.vup = Vec3(0.0, 1.0, 0.0);
  p.defocus_angle = 0.6;
  p.focus_dist = 10.0;
  if (device_bvh) return RtiowScene{std::make_shared<DeviceBvh>(std::move(world)), p};
  return RtiowScene{std::make_shared<Bvh>(std::move(world)), p};
}

// ------------------------------------------------------------------ RTC
struct RtcScene {
  rtc::World world;
  std::shared_ptr<rtc::Camera> camera;
};

inline RtcScene rtc_test_obj_scene(const std::string &teapot_obj_text, size_t res_x = 300, size_t res_y = 200) {
  using namespace rtc;
  auto obj = std::make_shared<Transformed>(
      WavefrontObj::parse(teapot_obj_text).to_object(),
      InvertibleMatrix4::try_from(transformation::sequence({transformation::rotation_x(-FRAC_PI_2)})));
  RtcScene s;
  s.world.objects.push_back(obj);
  s.world.lights.push_back(PointLight{Point3d{-2.0, 20.0, -30.0}, Color{1.0, 1.0, 1.0}});
  Point3d from{0.0, 15.0, -30.0}, to{0.0, 5.0, 0.0};
  Vec3d up{0.0, 1.0, 0.0};
  s.camera = std::make_shared<Camera>(res_x, res_y, FRAC_PI_3, InvertibleMatrix4::try_from(transformation::view_transform(from, to, up)));
  return s;
}

// tests/ray_tracer.rs:56-240 (mirror_scene) and :277-368 (csg_scene)
inline rtc::InvertibleMatrix4 rtc_xf(const std::vector<rtc::Matrix4> &seq) { return rtc::InvertibleMatrix4::try_from(rtc::transformation::sequence(seq)); }

inline RtcScene rtc_test_mirror_scene(size_t res_x = 300, size_t res_y = 200) {
  using namespace rtc;
  namespace T = transformation;
  const double FRAC_PI_4 = 0.785398163397448309615660845819875721;
  auto xf = [](ObjectPtr o, const std::vector<Matrix4> &seq) { return std::make_shared<Transformed>(o, rtc_xf(seq)); };
  auto gs1 = xf(Shape::sphere(), {T::translation(-0.5, 0.0, 0.0)});
  auto gs2 = xf(Shape::sphere(), {T::translation(0.5, 0.0, 0.0)});
  ObjectPtr sphere_group = std::make_shared<Bounded>(
      xf(std::make_shared<Group>(std::vector<ObjectPtr>{gs1, gs2}), {T::rotation_z(FRAC_PI_2), T::translation(-2.0, 2.0, 0.0)}));
  Material fm;
  fm.pattern.kind = RL_PAT_CHECKER3D, fm.pattern.a = Color{1, 1, 1}, fm.pattern.b = Color{0, 0, 0};
  fm.pattern.transform = rtc_xf({T::translation(0.0, -0.01, 0.0)});
  fm.specular = 0.0, fm.reflectivity = 0.02;
  ObjectPtr floor = Shape::plane(fm);
  Material lw;
  lw.surface = Color{1, 1, 1}, lw.specular = 1.0, lw.reflectivity = 0.9, lw.shininess = 400.0, lw.diffuse = 0.0;
  ObjectPtr left_wall = xf(Shape::plane(lw), {T::rotation_x(FRAC_PI_2), T::rotation_y(-FRAC_PI_3), T::translation(-8.0, 0.0, 0.0)});
  Material rw = lw;
  rw.reflectivity = 1.0;
  ObjectPtr right_wall = xf(Shape::plane(rw), {T::rotation_x(FRAC_PI_2), T::rotation_y(FRAC_PI_4), T::translation(10.0, 0.0, 0.0)});
  Material mw;
  mw.surface = Color{0.945, 0.788, 0.647}, mw.specular = 0.1, mw.shininess = 50.0;
  ObjectPtr middle_wall = xf(Shape::plane(mw), {T::rotation_x(FRAC_PI_2), T::translation(0.0, 0.0, 7.0)});
  Material bm;
  bm.surface = Color{0.059, 0.322, 0.729}, bm.diffuse = 0.3, bm.specular = 1.0, bm.reflectivity = 0.9, bm.transparency = 0.75, bm.refractive_index = 1.52;
  ObjectPtr ball = xf(Shape::sphere(bm), {T::translation(0.0, 2.0, 0.0)});
  Material am;
  am.surface = Color{1, 1, 1}, am.ambient = 0.0, am.diffuse = 0.0, am.specular = 0.0, am.transparency = 1.0, am.refractive_index = 1.0, am.reflectivity = 1.0;
  ObjectPtr inner_air_pocket = xf(Shape::sphere(am), {T::scaling(0.5, 0.5, 0.5), T::translation(0.0, 2.0, 0.0)});
  Material cm;
  cm.pattern.kind = RL_PAT_STRIPE, cm.pattern.a = Color{0.545, 0.0, 0.0}, cm.pattern.b = Color{0.0, 0.392, 0.0};
  cm.pattern.transform = rtc_xf({T::scaling(0.2, 1.0, 1.0)});
  ObjectPtr behind_cube = xf(Shape::cube(cm), {T::translation(3.0, 0.0, -10.0)});
  Material bw;
  bw.surface = Color{0.678, 0.847, 0.902}, bw.specular = 0.1, bw.shininess = 50.0;
  ObjectPtr behind_wall = xf(Shape::plane(bw), {T::rotation_x(FRAC_PI_2), T::translation(0.0, 0.0, -100.0)});
  RtcScene s;
  s.world.objects = {floor, left_wall, right_wall, middle_wall, ball, inner_air_pocket, behind_cube, behind_wall, sphere_group};
  s.world.lights.push_back(PointLight{Point3d{-10.0, 10.0, -10.0}, Color{1.0, 1.0, 1.0}});
  s.camera = std::make_shared<Camera>(res_x, res_y, FRAC_PI_3,
                                      InvertibleMatrix4::try_from(T::view_transform(Point3d{0.0, 2.0, -7.0}, Point3d{0.0, 1.5, 0.0}, Vec3d{0.0, 1.0, 0.0})));
  return s;
}

inline RtcScene rtc_test_csg_scene(size_t res_x = 300, size_t res_y = 200) {
  using namespace rtc;
  namespace T = transformation;
  auto xf = [](ObjectPtr o, const std::vector<Matrix4> &seq) { return std::make_shared<Transformed>(o, rtc_xf(seq)); };
  Material rm;
  rm.pattern.kind = RL_PAT_CHECKER3D, rm.pattern.a = Color{0.6, 0.6, 0.6}, rm.pattern.b = Color{0.7, 0.7, 0.7};
  rm.pattern.transform = rtc_xf({T::translation(0.01, 0.01, 0.01), T::scaling(0.02, 0.02, 0.02)});
  rm.reflectivity = 0.0, rm.ambient = 0.5, rm.shininess = 10.0, rm.diffuse = 0.3, rm.specular = 0.3;
  ObjectPtr room = xf(Shape::cube(rm), {T::scaling(50.0, 50.0, 50.0)});
  Material g, b, r;
  g.surface = Color{0, 1, 0}, b.surface = Color{0, 0, 1}, r.surface = Color{1, 0, 0};
  ObjectPtr hollow = std::make_shared<Csg>(Shape::sphere(g), xf(Shape::sphere(b), {T::scaling(0.7, 0.7, 0.7)}), RL_CSG_DIFFERENCE);
  ObjectPtr object = std::make_shared<Csg>(hollow, xf(Shape::cube(r), {T::translation(1.0, 0.0, 0.0)}), RL_CSG_DIFFERENCE);
  ObjectPtr object_t = xf(object, {T::rotation_y(FRAC_PI_6), T::scaling(7.0, 7.0, 7.0)});
  RtcScene s;
  s.world.objects = {room, object_t};
  s.world.lights.push_back(PointLight{Point3d{-2.0, 20.0, -30.0}, Color{0.5, 0.5, 0.5}});
  s.world.lights.push_back(PointLight{Point3d{10.0, 20.0, -30.0}, Color{0.5, 0.5, 0.5}});
  s.camera = std::make_shared<Camera>(res_x, res_y, FRAC_PI_3,
                                      InvertibleMatrix4::try_from(T::view_transform(Point3d{0.0, 0.0, -30.0}, Point3d{0.0, 0.0, 0.0}, Vec3d{0.0, 1.0, 0.0})));
  return s;
}

}  // namespace scenes
