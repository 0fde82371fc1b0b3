// C entry points over the C++ host mirror (rtiow_host.hpp / rtc_host.hpp / scenes.hpp) so that
// Python (ctypes) tests and bench.py can build the reference's scenes, derive cameras, flatten to
// the rl_render.h descriptors and encode PPMs.  Also defines Camera::render (both crates) on top of
// the C ABI — the call a Rust maintainer would add as `Camera::render_gpu`.
#include <cstdio>
#include <cstdlib>

#include "scenes.hpp"

extern "C" {

struct rlh_camera_params {  // POD mirror of rtiow::CameraParams (camera.rs:23-39)
  double aspect_ratio;
  uint64_t image_width, samples_per_pixel, max_depth;
  double vfov;
  double lookfrom[3], lookat[3], vup[3];
  double defocus_angle, focus_dist;
  double background[3];
  uint64_t seed;
};

struct rlh_rtiow {
  rtiow::Flattened flat;
  rl_rtiow_scene_desc desc;
  rtiow::CameraParams params;
};

static thread_local std::string g_err;
const char *rlh_last_error() { return g_err.c_str(); }

static rlh_rtiow *finish(scenes::RtiowScene &&s) {
  auto *h = new rlh_rtiow();
  h->flat.root = s.world->flatten(h->flat);
  h->desc = h->flat.desc();
  h->params = s.params;
  return h;
}

rlh_rtiow *rlh_rtiow_golden_test_scene() {
  try {
    return finish(scenes::golden_test_scene());
  } catch (std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}
rlh_rtiow *rlh_rtiow_bouncing_spheres(uint64_t master_seed) {
  try {
    return finish(scenes::bouncing_spheres(master_seed));
  } catch (std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}
rlh_rtiow *rlh_rtiow_cow_scene(const char *obj_text, uint64_t obj_len, const uint8_t *rgb8, uint32_t w, uint32_t h) {
  try {
    return finish(scenes::cow_scene(std::string(obj_text, obj_len), rgb8, w, h));
  } catch (std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}

// examples/perlin_spheres.rs (light = 0) / examples/simple_light.rs (light = 1)
rlh_rtiow *rlh_rtiow_perlin_scene(int light) {
  try {
    return finish(scenes::perlin_scene(light != 0));
  } catch (std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}
// examples/earth.rs with the caller's sRGB8 image
rlh_rtiow *rlh_rtiow_earth_scene(const uint8_t *rgb8, uint32_t w, uint32_t h) {
  try {
    return finish(scenes::earth_scene(rgb8, w, h));
  } catch (std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}
// The reference's remaining example scenes by name: "checkered_spheres", "quads", "flat_world", "cornell_box", "cornell_smoke",
// "teapot" (obj_text = the OBJ), "final_scene" (rgb8 = the earth image, sRGB8).  Unused arguments may be NULL / 0.
rlh_rtiow *rlh_rtiow_example_scene(const char *name, const char *obj_text, uint64_t obj_len, const uint8_t *rgb8, uint32_t w, uint32_t h) {
  try {
    std::string n(name ? name : "");
    if (n == "checkered_spheres") return finish(scenes::checkered_spheres());
    if (n == "quads") return finish(scenes::quads_scene(false));
    if (n == "flat_world") return finish(scenes::quads_scene(true));
    if (n == "cornell_box") return finish(scenes::cornell_scene(false));
    if (n == "cornell_smoke") return finish(scenes::cornell_scene(true));
    if (n == "teapot" && obj_text) return finish(scenes::teapot_scene(std::string(obj_text, obj_len)));
    if (n == "final_scene" && rgb8) return finish(scenes::final_scene(rgb8, w, h));
    g_err = "unknown example scene (or a missing OBJ / image): " + n;
    return nullptr;
  } catch (std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}
// BASELINE configs[4] stress scene (scenes.hpp::stress_scene); obj_text may be NULL (spheres only)
rlh_rtiow *rlh_rtiow_stress_scene(int n_side, int subdiv, const char *obj_text, uint64_t obj_len, const uint8_t *rgb8, uint32_t w, uint32_t h,
                                  uint64_t seed, int device_bvh) {
  try {
    return finish(scenes::stress_scene(n_side, subdiv, obj_text ? std::string(obj_text, obj_len) : std::string(), rgb8, w, h, seed, device_bvh != 0));
  } catch (std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}

// Generic: a world of spheres, either a plain slice (use_bvh=0, `[Sphere]`) or Bvh::new(spheres).
// materials/textures are rl_render.h PODs; sphere.material indexes `materials`.
rlh_rtiow *rlh_rtiow_from_spheres(const rl_sphere *spheres, uint32_t n, const rl_material *materials, uint32_t n_mat,
                                  const rl_texture *textures, uint32_t n_tex, int use_bvh) {
  try {
    using namespace rtiow;
    std::vector<TexturePtr> tex(n_tex);
    // textures may reference later ids (checker): two passes
    for (uint32_t i = 0; i < n_tex; i++) tex[i] = std::make_shared<Texture>();
    for (uint32_t i = 0; i < n_tex; i++) {
      const rl_texture &t = textures[i];
      tex[i]->kind = t.kind;
      tex[i]->color = Color(t.color[0], t.color[1], t.color[2]);
      tex[i]->inv_scale = t.inv_scale;
      if (t.kind == RL_TEX_CHECKER) {
        if (t.even >= n_tex || t.odd >= n_tex) throw std::runtime_error("texture index out of range");
        tex[i]->even = tex[t.even];
        tex[i]->odd = tex[t.odd];
      } else if (t.kind != RL_TEX_SOLID)
        throw std::runtime_error("only solid/checker textures here");
    }
    std::vector<MaterialPtr> mats(n_mat);
    for (uint32_t i = 0; i < n_mat; i++) {
      const rl_material &m = materials[i];
      auto p = std::make_shared<Material>();
      p->kind = m.kind;
      if (m.kind == RL_MAT_LAMBERTIAN || m.kind == RL_MAT_DIFFUSE_LIGHT) {
        if (m.texture >= n_tex) throw std::runtime_error("material texture out of range");
        p->texture = tex[m.texture];
      }
      p->albedo = Color(m.albedo[0], m.albedo[1], m.albedo[2]);
      p->fuzz = m.fuzz;
      p->ior = m.ior;
      mats[i] = p;
    }
    std::vector<HittablePtr> hs;
    for (uint32_t i = 0; i < n; i++) {
      const rl_sphere &s = spheres[i];
      if (s.material >= n_mat) throw std::runtime_error("sphere material out of range");
      Center c = s.moving ? Center::Moving(Point3(s.center0[0], s.center0[1], s.center0[2]), Point3(s.center1[0], s.center1[1], s.center1[2]))
                          : Center::Stationary(Point3(s.center0[0], s.center0[1], s.center0[2]));
      hs.push_back(std::make_shared<Sphere>(c, s.radius, mats[s.material]));
    }
    scenes::RtiowScene sc;
    if (use_bvh) sc.world = std::make_shared<Bvh>(std::move(hs));
    else sc.world = std::make_shared<HittableList>(std::move(hs));
    return finish(std::move(sc));
  } catch (std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}

// ---- generic scene builder: compose any world the reference's API can (ids index the builder's tables)
struct rlh_builder {
  std::vector<rtiow::TexturePtr> tex;
  std::vector<rtiow::MaterialPtr> mat;
  std::vector<rtiow::HittablePtr> obj;
};
rlh_builder *rlh_builder_new() { return new rlh_builder(); }
void rlh_builder_free(rlh_builder *b) { delete b; }
#define RLH_TRY(expr)           \
  try {                         \
    expr;                       \
  } catch (std::exception & e) { \
    g_err = e.what();           \
    return -1;                  \
  }
int rlh_b_solid(rlh_builder *b, const double *c) {
  b->tex.push_back(rtiow::SolidColor(rtiow::Color(c[0], c[1], c[2])));
  return (int)b->tex.size() - 1;
}
int rlh_b_checker(rlh_builder *b, double scale, int even, int odd) {
  if (even < 0 || odd < 0 || (size_t)even >= b->tex.size() || (size_t)odd >= b->tex.size()) return -1;
  b->tex.push_back(rtiow::Checker(scale, b->tex[even], b->tex[odd]));
  return (int)b->tex.size() - 1;
}
// linear f32 RGB image, top row first (what texture.rs:58 Image holds)
int rlh_b_image(rlh_builder *b, const float *rgb, uint32_t w, uint32_t h) {
  auto img = std::make_shared<rtiow::ImageData>();
  img->width = w, img->height = h;
  img->rgb.assign(rgb, rgb + (size_t)w * h * 3);
  b->tex.push_back(rtiow::Image(img));
  return (int)b->tex.size() - 1;
}
// Noise{Perlin::new(&mut Xoshiro256PlusPlus::seed_from_u64(seed)), scale} (texture.rs:84-87)
int rlh_b_noise(rlh_builder *b, double scale, uint64_t seed) {
  auto rng = scenes::Xoshiro256PlusPlus::seed_from_u64(seed);
  b->tex.push_back(rtiow::Noise(rtiow::Perlin::create(rng), scale));
  return (int)b->tex.size() - 1;
}
int rlh_b_material(rlh_builder *b, uint32_t kind, int tex, const double *albedo, double fuzz, double ior) {
  auto m = std::make_shared<rtiow::Material>();
  m->kind = kind;
  if (kind == RL_MAT_LAMBERTIAN || kind == RL_MAT_DIFFUSE_LIGHT || kind == RL_MAT_ISOTROPIC) {
    if (tex < 0 || (size_t)tex >= b->tex.size()) return -1;
    m->texture = b->tex[tex];
  }
  if (albedo) m->albedo = rtiow::Color(albedo[0], albedo[1], albedo[2]);
  m->fuzz = fuzz, m->ior = ior;
  b->mat.push_back(m);
  return (int)b->mat.size() - 1;
}
static bool okm(rlh_builder *b, int m) { return m >= 0 && (size_t)m < b->mat.size(); }
static bool oko(rlh_builder *b, int o) { return o >= 0 && (size_t)o < b->obj.size(); }
int rlh_b_sphere(rlh_builder *b, const double *c0, const double *c1_or_null, double radius, int mat) {
  if (!okm(b, mat)) return -1;
  using namespace rtiow;
  Center c = c1_or_null ? Center::Moving(Point3(c0[0], c0[1], c0[2]), Point3(c1_or_null[0], c1_or_null[1], c1_or_null[2]))
                        : Center::Stationary(Point3(c0[0], c0[1], c0[2]));
  b->obj.push_back(std::make_shared<Sphere>(c, radius, b->mat[mat]));
  return (int)b->obj.size() - 1;
}
// kind: RL_PLANAR_PLANE / QUAD / TRIANGLE built from (q, u, v) like Plane::new / Quad::new / Triangle::new
int rlh_b_planar(rlh_builder *b, uint32_t kind, const double *q, const double *u, const double *v, int mat) {
  if (!okm(b, mat)) return -1;
  using namespace rtiow;
  Point3 Q(q[0], q[1], q[2]);
  Vec3 U(u[0], u[1], u[2]), V(v[0], v[1], v[2]);
  RLH_TRY(
      if (kind == RL_PLANAR_PLANE) b->obj.push_back(std::make_shared<Plane>(Q, U, V, b->mat[mat]));
      else if (kind == RL_PLANAR_QUAD) b->obj.push_back(std::make_shared<Quad>(Q, U, V, b->mat[mat]));
      else b->obj.push_back(Triangle::from_quv(Q, U, V, b->mat[mat])))
  return (int)b->obj.size() - 1;
}
// Triangle::from_model(points, texture_coords?, normals?, material)
int rlh_b_triangle(rlh_builder *b, const double *p9, const double *uv6_or_null, const double *n9_or_null, int mat) {
  if (!okm(b, mat)) return -1;
  using namespace rtiow;
  Point3 pts[3] = {Point3(p9[0], p9[1], p9[2]), Point3(p9[3], p9[4], p9[5]), Point3(p9[6], p9[7], p9[8])};
  Vec3 ns[3];
  if (n9_or_null)
    for (int i = 0; i < 3; i++) ns[i] = Vec3(n9_or_null[3 * i], n9_or_null[3 * i + 1], n9_or_null[3 * i + 2]);
  RLH_TRY(b->obj.push_back(std::make_shared<Triangle>(pts, uv6_or_null, n9_or_null ? ns : nullptr, b->mat[mat])))
  return (int)b->obj.size() - 1;
}
int rlh_b_translate(rlh_builder *b, int obj, const double *off) {
  if (!oko(b, obj)) return -1;
  b->obj.push_back(std::make_shared<rtiow::Translate>(b->obj[obj], rtiow::Vec3(off[0], off[1], off[2])));
  return (int)b->obj.size() - 1;
}
int rlh_b_medium(rlh_builder *b, int boundary, double density, int mat) {  // ConstantMedium::new (constant_medium.rs:16-24)
  if (!oko(b, boundary) || !okm(b, mat)) return -1;
  b->obj.push_back(std::make_shared<rtiow::ConstantMedium>(b->obj[boundary], density, b->mat[mat]));
  return (int)b->obj.size() - 1;
}
// op: 0 rotate_x, 1 rotate_y, 2 rotate_z (degrees), 3 uniform scale
int rlh_b_transform(rlh_builder *b, int obj, int op, double value) {
  if (!oko(b, obj)) return -1;
  using rtiow::Transform;
  auto o = b->obj[obj];
  b->obj.push_back(op == 0 ? Transform::rotate_x(o, value) : op == 1 ? Transform::rotate_y(o, value) : op == 2 ? Transform::rotate_z(o, value) : Transform::scale(o, value));
  return (int)b->obj.size() - 1;
}
int rlh_b_group(rlh_builder *b, const int *objs, uint32_t n, int as_bvh) {
  std::vector<rtiow::HittablePtr> hs;
  for (uint32_t i = 0; i < n; i++) {
    if (!oko(b, objs[i])) return -1;
    hs.push_back(b->obj[objs[i]]);
  }
  RLH_TRY(
      if (as_bvh) b->obj.push_back(std::make_shared<rtiow::Bvh>(std::move(hs)));
      else b->obj.push_back(std::make_shared<rtiow::HittableList>(std::move(hs))))
  return (int)b->obj.size() - 1;
}
// OBJ text -> Bvh<Triangle> with one material (io/wavefront_obj.rs to_object)
int rlh_b_obj(rlh_builder *b, const char *text, uint64_t len, int mat) {
  if (!okm(b, mat)) return -1;
  RLH_TRY(b->obj.push_back(scenes::RtiowObj::parse(std::string(text, len)).to_object(b->mat[mat])))
  return (int)b->obj.size() - 1;
}
rlh_rtiow *rlh_b_finish(rlh_builder *b, int root) {
  if (!oko(b, root)) {
    g_err = "bad root";
    return nullptr;
  }
  try {
    scenes::RtiowScene sc;
    sc.world = b->obj[root];
    return finish(std::move(sc));
  } catch (std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}

const rl_rtiow_scene_desc *rlh_rtiow_desc(const rlh_rtiow *h) { return &h->desc; }
void rlh_rtiow_free(rlh_rtiow *h) { delete h; }
// element counts of the flattened scene: spheres, planars, media, translates, transforms, lists, bvh nodes, materials, textures
void rlh_rtiow_counts(const rlh_rtiow *h, uint32_t out[9]) {
  const rl_rtiow_scene_desc &d = h->desc;
  const uint32_t v[9] = {d.n_spheres, d.n_planars, d.n_media, d.n_translates, d.n_transforms, d.n_lists, d.n_bvh_nodes, d.n_materials, d.n_textures};
  for (int i = 0; i < 9; i++) out[i] = v[i];
}

void rlh_rtiow_get_params(const rlh_rtiow *h, rlh_camera_params *o) {
  const auto &p = h->params;
  o->aspect_ratio = p.aspect_ratio;
  o->image_width = p.image_width, o->samples_per_pixel = p.samples_per_pixel, o->max_depth = p.max_depth;
  o->vfov = p.vfov;
  rtiow::put3(o->lookfrom, p.lookfrom), rtiow::put3(o->lookat, p.lookat), rtiow::put3(o->vup, p.vup);
  o->defocus_angle = p.defocus_angle, o->focus_dist = p.focus_dist;
  rtiow::put3(o->background, p.background);
  o->seed = p.seed;
}

static rtiow::CameraParams to_params(const rlh_camera_params *i) {
  rtiow::CameraParams p;
  p.aspect_ratio = i->aspect_ratio;
  p.image_width = i->image_width, p.samples_per_pixel = i->samples_per_pixel, p.max_depth = i->max_depth;
  p.vfov = i->vfov;
  p.lookfrom = rtiow::Point3(i->lookfrom[0], i->lookfrom[1], i->lookfrom[2]);
  p.lookat = rtiow::Point3(i->lookat[0], i->lookat[1], i->lookat[2]);
  p.vup = rtiow::Vec3(i->vup[0], i->vup[1], i->vup[2]);
  p.defocus_angle = i->defocus_angle, p.focus_dist = i->focus_dist;
  p.background = rtiow::Color(i->background[0], i->background[1], i->background[2]);
  p.seed = i->seed;
  return p;
}

// Camera::new (camera.rs:72-118). Returns 0, or -1 where the reference would panic (unwrap on a
// degenerate basis, camera.rs:86-88).
int rlh_rtiow_camera_new(const rlh_camera_params *in, rl_rtiow_camera *out) {
  try {
    rtiow::Camera cam(to_params(in));
    *out = cam.derived();
    return 0;
  } catch (std::exception &e) {
    g_err = e.what();
    return -1;
  }
}

static char *dup_string(const std::string &s, uint64_t *len) {
  char *p = (char *)std::malloc(s.size() + 1);
  std::memcpy(p, s.data(), s.size());
  p[s.size()] = 0;
  if (len) *len = s.size();
  return p;
}
void rlh_free(void *p) { std::free(p); }

char *rlh_rtiow_output_ppm(const double *rgb_sum, uint64_t w, uint64_t h, uint64_t samples, uint64_t *len) {
  return dup_string(rtiow::output_ppm(rgb_sum, w, h, samples), len);
}

// Canvas checkpoint codec (bincode 1.x layout of camera.rs:263-270). encode: returns malloc'd bytes (rlh_free).
void *rlh_canvas_to_bincode(uint64_t samples, uint64_t width, uint64_t height, const double *rgb_sum, uint64_t n_pixels, uint64_t *len) {
  rtiow::Canvas c{(size_t)samples, (size_t)width, (size_t)height, std::vector<double>(rgb_sum, rgb_sum + n_pixels * 3)};
  std::vector<uint8_t> b = rtiow::canvas_to_bincode(c);
  void *p = std::malloc(b.size() ? b.size() : 1);
  std::memcpy(p, b.data(), b.size());
  *len = b.size();
  return p;
}
// decode header: returns 0 and fills samples/width/height/n_pixels, or -1 (malformed); pixels start at byte 32
int rlh_canvas_from_bincode(const uint8_t *bytes, uint64_t len, uint64_t *samples, uint64_t *width, uint64_t *height, uint64_t *n_pixels) {
  try {
    rtiow::Canvas c = rtiow::canvas_from_bincode(bytes, (size_t)len);
    *samples = c.samples, *width = c.width, *height = c.height, *n_pixels = c.data.size() / 3;
    return 0;
  } catch (std::exception &e) {
    g_err = e.what();
    return -1;
  }
}

// Bvh::new on the host (rtiow::Bvh) against rl_bvh_build on the device over the same hittables: n spheres (every 7th moving,
// radii and centres from a small LCG; `ties` > 0 snaps the centres to a grid so that many sort keys are EQUAL and the
// stable order matters) plus n / 4 triangles.  Returns 0 when every node record is identical (boxes bit for bit, same
// children), else the 1-based index of the first differing node, or -1 on error.
int64_t rlh_bvh_build_compare(uint32_t n, uint64_t seed, int ties) {
  try {
    using namespace rtiow;
    uint64_t st = seed * 6364136223846793005ull + 1442695040888963407ull;
    auto rnd = [&]() {
      st = st * 6364136223846793005ull + 1442695040888963407ull;
      return (double)(st >> 11) * (1.0 / 9007199254740992.0);
    };
    auto coord = [&]() {
      double v = rnd() * 40.0 - 20.0;
      return ties > 0 ? std::floor(v / 4.0) * 4.0 : v;
    };
    auto mat = Lambertian(SolidColor(Color(0.5, 0.5, 0.5)));
    std::vector<HittablePtr> hs;
    for (uint32_t i = 0; i < n; i++) {
      Point3 c(coord(), coord(), coord());
      double r = 0.05 + rnd();
      if (i % 7 == 3) hs.push_back(std::make_shared<Sphere>(Center::Moving(c, c + Vec3(rnd(), rnd(), rnd())), r, mat));
      else hs.push_back(std::make_shared<Sphere>(Center::Stationary(c), r, mat));
    }
    for (uint32_t i = 0; i < n / 4; i++) {
      Point3 q(coord(), coord(), coord());
      Point3 pts[3] = {q, q + Vec3(rnd(), 0.0, rnd()), q + Vec3(0.0, rnd(), ties ? 0.0 : rnd())};
      hs.push_back(std::make_shared<Triangle>(pts, nullptr, nullptr, mat));
    }
    Flattened fh, fd;
    fh.root = Bvh(hs).flatten(fh);
    fd.root = DeviceBvh(hs).flatten(fd);
    if (fh.bvh_nodes.size() != fd.bvh_nodes.size()) return -1;
    auto same_prim = [&](rl_href a, rl_href b) {
      if (a.kind != b.kind) return false;
      if (a.kind == RL_H_SPHERE) return std::memcmp(&fh.spheres[a.index], &fd.spheres[b.index], sizeof(rl_sphere)) == 0;
      if (a.kind == RL_H_PLANAR) return std::memcmp(&fh.planars[a.index], &fd.planars[b.index], sizeof(rl_planar)) == 0;
      return a.index == b.index;  // RL_H_BVH: both numberings are the recursion order
    };
    for (size_t k = 0; k < fh.bvh_nodes.size(); k++) {
      const rl_bvh_node &a = fh.bvh_nodes[k], &b = fd.bvh_nodes[k];
      bool ok = std::memcmp(a.bbox, b.bbox, sizeof a.bbox) == 0 && a.n_children == b.n_children;
      for (uint32_t c = 0; ok && c < a.n_children; c++) ok = same_prim(a.child[c], b.child[c]);
      if (!ok) return (int64_t)k + 1;
    }
    return 0;
  } catch (std::exception &e) {
    g_err = e.what();
    return -1;
  }
}

// ------------------------------------------------------------------ the reference's integration tests, end to end in C++
// tests/ray_tracing_one_weekend.rs:77-95: scene -> Camera::new(params).render(&world) -> output_ppm, through the
// C++ mirror (Camera::render in host_render.cpp calls the C ABI).  Returns the PPM text (rlh_free) or NULL.
char *rlh_rtiow_run_golden_test(int from_checkpoint, uint64_t *len) {
  try {
    scenes::RtiowScene s = scenes::golden_test_scene();
    if (!from_checkpoint) {
      rtiow::Canvas c = rtiow::Camera(s.params).render(*s.world);
      return dup_string(rtiow::output_ppm(c), len);
    }
    // tests/ray_tracing_one_weekend.rs:118-162: render half the samples, round-trip the checkpoint through bincode, resume
    rtiow::CameraParams half = s.params;
    half.samples_per_pixel = s.params.samples_per_pixel / 2;
    rtiow::Canvas first = rtiow::Camera(half).render(*s.world);
    std::vector<uint8_t> bytes = rtiow::canvas_to_bincode(first);
    rtiow::Canvas checkpoint = rtiow::canvas_from_bincode(bytes.data(), bytes.size());
    rtiow::CameraParams rest = s.params;
    rest.samples_per_pixel = s.params.samples_per_pixel - half.samples_per_pixel;
    rtiow::Canvas c = rtiow::Camera(rest).render_from_checkpoint(*s.world, checkpoint);
    return dup_string(rtiow::output_ppm(c), len);
  } catch (std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}
// tests/ray_tracer.rs:242-275 (which = 0, needs the OBJ text), :56-240 mirror (1), :277-368 csg (2): Camera::render -> Canvas::ppm
char *rlh_rtc_run_golden_test(int which, const char *obj_text, uint64_t obj_len, uint64_t *len) {
  try {
    scenes::RtcScene s = which == 0 ? scenes::rtc_test_obj_scene(std::string(obj_text, obj_len))
                                    : (which == 1 ? scenes::rtc_test_mirror_scene() : scenes::rtc_test_csg_scene());
    rtc::Canvas c = s.camera->render(s.world, rtc::RenderOpts{});
    return dup_string(rtc::canvas_ppm(c), len);
  } catch (std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}

// ------------------------------------------------------------------ RTC
struct rlh_rtc {
  rtc::Flattened flat;
  rl_rtc_scene_desc desc;
  rl_rtc_camera camera;
};

rlh_rtc *rlh_rtc_test_obj_scene(const char *obj_text, uint64_t obj_len, uint64_t res_x, uint64_t res_y) {
  try {
    auto s = scenes::rtc_test_obj_scene(std::string(obj_text, obj_len), res_x, res_y);
    auto *h = new rlh_rtc();
    s.world.flatten(h->flat);
    h->desc = h->flat.desc();
    h->camera = s.camera->derived();
    return h;
  } catch (std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}
// which: 0 = mirror scene, 1 = CSG scene (ray-tracer-challenge/tests/ray_tracer.rs:56, :277)
rlh_rtc *rlh_rtc_named_scene(int which, uint64_t res_x, uint64_t res_y) {
  try {
    auto s = which == 0 ? scenes::rtc_test_mirror_scene(res_x, res_y) : scenes::rtc_test_csg_scene(res_x, res_y);
    auto *h = new rlh_rtc();
    s.world.flatten(h->flat);
    h->desc = h->flat.desc();
    h->camera = s.camera->derived();
    return h;
  } catch (std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}
const rl_rtc_scene_desc *rlh_rtc_desc(const rlh_rtc *h) { return &h->desc; }
void rlh_rtc_get_camera(const rlh_rtc *h, rl_rtc_camera *out) { *out = h->camera; }
void rlh_rtc_free(rlh_rtc *h) { delete h; }

// Camera::new(hsize, vsize, fov, view_transform(from,to,up))  (scene/camera.rs:35, transformation.rs:74)
int rlh_rtc_camera_new(uint64_t hsize, uint64_t vsize, double fov, const double *from, const double *to, const double *up, rl_rtc_camera *out) {
  try {
    using namespace rtc;
    Camera c(hsize, vsize, fov,
             InvertibleMatrix4::try_from(transformation::view_transform(Point3d{from[0], from[1], from[2]}, Point3d{to[0], to[1], to[2]}, Vec3d{up[0], up[1], up[2]})));
    *out = c.derived();
    return 0;
  } catch (std::exception &e) {
    g_err = e.what();
    return -1;
  }
}
// Camera::new with an explicit 4x4 transform (row-major); returns -1 if not invertible
int rlh_rtc_camera_from_matrix(uint64_t hsize, uint64_t vsize, double fov, const double *m16, rl_rtc_camera *out) {
  try {
    rtc::Matrix4 M;
    std::memcpy(M.m, m16, sizeof M.m);
    rtc::Camera c(hsize, vsize, fov, rtc::InvertibleMatrix4::try_from(M));
    *out = c.derived();
    return 0;
  } catch (std::exception &e) {
    g_err = e.what();
    return -1;
  }
}
// Transformed::new(child, transform) helper for Python-built scenes: fills inverse + inverse_transpose
int rlh_rtc_make_transformed(const double *m16, rl_rtc_transformed *out) {
  rtc::Matrix4 M, inv;
  std::memcpy(M.m, m16, sizeof M.m);
  if (!rtc::invert(M, inv)) return -1;
  rtc::Matrix4 it = inv.transpose();
  std::memcpy(out->inverse, inv.m, sizeof inv.m);
  std::memcpy(out->inverse_transpose, it.m, sizeof it.m);
  return 0;
}
// transformation::{translation,scaling,rotation_*,view_transform}, sequence — exposed for tests
void rlh_rtc_rotation(int axis, double radians, double *out16) {
  rtc::Matrix4 m = axis == 0 ? rtc::transformation::rotation_x(radians) : axis == 1 ? rtc::transformation::rotation_y(radians) : rtc::transformation::rotation_z(radians);
  std::memcpy(out16, m.m, sizeof m.m);
}
void rlh_rtc_matmul(const double *a16, const double *b16, double *out16) {
  rtc::Matrix4 A, B;
  std::memcpy(A.m, a16, sizeof A.m);
  std::memcpy(B.m, b16, sizeof B.m);
  rtc::Matrix4 C = A.mul(B);
  std::memcpy(out16, C.m, sizeof C.m);
}
int rlh_rtc_invert(const double *m16, double *out16) {
  rtc::Matrix4 M, inv;
  std::memcpy(M.m, m16, sizeof M.m);
  if (!rtc::invert(M, inv)) return -1;
  std::memcpy(out16, inv.m, sizeof inv.m);
  return 0;
}

char *rlh_rtc_canvas_ppm(const double *rgb, uint64_t w, uint64_t h, uint64_t *len) { return dup_string(rtc::canvas_ppm(rgb, w, h), len); }

}  // extern "C"

