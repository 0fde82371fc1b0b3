// Host-side mirror of ray-tracer-challenge's scene-building API (what stays on the host):
// 4x4 matrices with cofactor inverse, transformation builders, Camera::new, Material, PointLight,
// Triangle/Group/Bounded/Transformed, World, the Wavefront OBJ loader and Canvas::ppm; plus the
// flattener to include/rl_render.h's rl_rtc_scene_desc.  C++ restatement of the HOST half only
// (no Rust toolchain in this image).  Reference file:line cited per function (paths under
// /root/reference/ray-tracer-challenge/src/).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rl_render.h"

namespace rtc {

// ---------------------------------------------------------------- math/matrix.rs
struct Matrix4 {
  double m[4][4];
  static Matrix4 identity() {
    Matrix4 r{};
    for (int i = 0; i < 4; i++) r.m[i][i] = 1.0;
    return r;
  }
  Matrix4 transpose() const {
    Matrix4 r;
    for (int n = 0; n < 4; n++)
      for (int k = 0; k < 4; k++) r.m[k][n] = m[n][k];
    return r;
  }
  Matrix4 mul(const Matrix4 &rhs) const {  // matrix.rs:192-210: sum from 0.0 over i
    Matrix4 o;
    for (int n = 0; n < 4; n++)
      for (int k = 0; k < 4; k++) {
        double sum = 0.0;
        for (int i = 0; i < 4; i++) sum += m[n][i] * rhs.m[i][k];
        o.m[n][k] = sum;
      }
    return o;
  }
};

// determinant / cofactor / minor by recursive expansion along row 0 (matrix.rs:88-131)
inline double det_n(const double *a, int n) {  // a: row-major n x n
  if (n == 2) return a[0] * a[3] - a[1] * a[2];
  double sum = 0.0;
  for (int i = 0; i < n; i++) {
    double sub[9];
    int p = 0;
    for (int r = 1; r < n; r++)
      for (int c = 0; c < n; c++)
        if (c != i) sub[p++] = a[r * n + c];
    double minor = det_n(sub, n - 1);
    double cof = (i % 2 == 0) ? minor : -minor;
    sum += a[i] * cof;
  }
  return sum;
}
inline double cofactor4(const Matrix4 &M, int n, int m) {
  double sub[9];
  int p = 0;
  for (int r = 0; r < 4; r++)
    if (r != n)
      for (int c = 0; c < 4; c++)
        if (c != m) sub[p++] = M.m[r][c];
  double minor = det_n(sub, 3);
  return ((n + m) % 2 == 0) ? minor : -minor;
}
inline bool invert(const Matrix4 &M, Matrix4 &out) {  // matrix.rs:67-85
  double det = det_n(&M.m[0][0], 4);
  if (det == 0.0) return false;
  for (int n = 0; n < 4; n++)
    for (int m = 0; m < 4; m++) out.m[m][n] = cofactor4(M, n, m) / det;
  return true;
}

struct InvertibleMatrix4 {  // matrix.rs:243-247
  Matrix4 matrix, inverse;
  static InvertibleMatrix4 identity() { return InvertibleMatrix4{Matrix4::identity(), Matrix4::identity()}; }
  static InvertibleMatrix4 try_from(const Matrix4 &m) {
    InvertibleMatrix4 r;
    r.matrix = m;
    if (!invert(m, r.inverse)) throw std::runtime_error("Matrix is not invertible.");
    return r;
  }
};

// ---------------------------------------------------------------- math/vector.rs, point.rs
struct Vec3d {
  double x, y, z;
  double mag() const { return std::sqrt(x * x + y * y + z * z); }  // vector.rs:32
  Vec3d cross(const Vec3d &r) const { return Vec3d{y * r.z - z * r.y, z * r.x - x * r.z, x * r.y - y * r.x}; }
  double dot(const Vec3d &r) const { return x * r.x + y * r.y + z * r.z; }
};
struct Point3d {
  double x, y, z;
};
inline Vec3d operator-(const Point3d &a, const Point3d &b) { return Vec3d{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3d norm(const Vec3d &v) {  // vector.rs:36-43 (true division)
  double m = v.mag();
  if (m == 0.0) throw std::runtime_error("cannot normalize zero vector");
  return Vec3d{v.x / m, v.y / m, v.z / m};
}
// NormalizedVec3d::try_from (vector.rs:142-156): each component /= mag
inline Vec3d normalized(const Vec3d &v) { return norm(v); }

// ---------------------------------------------------------------- scene/transformation.rs
namespace transformation {
inline Matrix4 translation(double x, double y, double z) { return Matrix4{{{1, 0, 0, x}, {0, 1, 0, y}, {0, 0, 1, z}, {0, 0, 0, 1}}}; }
inline Matrix4 scaling(double x, double y, double z) { return Matrix4{{{x, 0, 0, 0}, {0, y, 0, 0}, {0, 0, z, 0}, {0, 0, 0, 1}}}; }
inline Matrix4 rotation_x(double r) {
  double s = std::sin(r), c = std::cos(r);
  return Matrix4{{{1, 0, 0, 0}, {0, c, -s, 0}, {0, s, c, 0}, {0, 0, 0, 1}}};
}
inline Matrix4 rotation_y(double r) {
  double s = std::sin(r), c = std::cos(r);
  return Matrix4{{{c, 0, s, 0}, {0, 1, 0, 0}, {-s, 0, c, 0}, {0, 0, 0, 1}}};
}
inline Matrix4 rotation_z(double r) {
  double s = std::sin(r), c = std::cos(r);
  return Matrix4{{{c, -s, 0, 0}, {s, c, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}};
}
inline Matrix4 shearing(double xy, double xz, double yx, double yz, double zx, double zy) {
  return Matrix4{{{1, xy, xz, 0}, {yx, 1, yz, 0}, {zx, zy, 1, 0}, {0, 0, 0, 1}}};
}
inline Matrix4 sequence(const std::vector<Matrix4> &ts) {  // transformation.rs:68-72: fold(identity, |acc,t| t*acc)
  Matrix4 acc = Matrix4::identity();
  for (const auto &t : ts) acc = t.mul(acc);
  return acc;
}
inline Matrix4 view_transform(const Point3d &from, const Point3d &to, const Vec3d &up) {  // :74-88
  Vec3d forward = norm(to - from);
  Vec3d upn = norm(up);
  Vec3d left = forward.cross(upn);
  Vec3d true_up = left.cross(forward);
  Matrix4 orientation{{{left.x, left.y, left.z, 0}, {true_up.x, true_up.y, true_up.z, 0}, {-forward.x, -forward.y, -forward.z, 0}, {0, 0, 0, 1}}};
  return orientation.mul(translation(-from.x, -from.y, -from.z));
}
}  // namespace transformation

// ---------------------------------------------------------------- draw/color.rs, scene/material.rs, light.rs
struct Color {
  double r, g, b;
};
struct Pattern {  // scene/pattern/{stripe,ring,gradient,checker3d}.rs
  uint32_t kind = 0;  // RL_PAT_*; 0 = none (Surface::Color)
  Color a{1.0, 1.0, 1.0}, b{0.0, 0.0, 0.0};
  InvertibleMatrix4 transform = InvertibleMatrix4::identity();
};
struct Material {  // material.rs:22-52
  Color surface{1.0, 1.0, 1.0};  // Surface::Color(..) when pattern.kind == 0
  Pattern pattern;               // Surface::Pattern(..) otherwise
  double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0;
  double reflectivity = 0.0, transparency = 0.0, refractive_index = 1.0;
};
struct PointLight {
  Point3d position;
  Color intensity;
};

// ---------------------------------------------------------------- flattener
struct Flattened {
  std::vector<rl_rtc_triangle> triangles;
  std::vector<rl_rtc_group> groups;
  std::vector<rl_oref> group_items;
  std::vector<rl_rtc_bounded> boundeds;
  std::vector<rl_rtc_transformed> transformeds;
  std::vector<rl_rtc_material> materials;
  std::vector<rl_oref> objects;
  std::vector<rl_rtc_light> lights;
  std::vector<rl_rtc_shape> shapes;
  std::vector<rl_rtc_csg> csgs;
  std::vector<rl_rtc_pattern> patterns;
  uint32_t max_reflection_depth = 5;
  double void_color[3] = {0, 0, 0};
  uint32_t add_material(const Material &m) {
    rl_rtc_material r{};
    if (m.pattern.kind != 0) {
      rl_rtc_pattern pt{};
      pt.kind = m.pattern.kind;
      pt.a[0] = m.pattern.a.r, pt.a[1] = m.pattern.a.g, pt.a[2] = m.pattern.a.b;
      pt.b[0] = m.pattern.b.r, pt.b[1] = m.pattern.b.g, pt.b[2] = m.pattern.b.b;
      std::memcpy(pt.inverse, m.pattern.transform.inverse.m, sizeof pt.inverse);
      patterns.push_back(pt);
      r.pattern = (uint32_t)patterns.size();  // 1-based
    }
    r.color[0] = m.surface.r, r.color[1] = m.surface.g, r.color[2] = m.surface.b;
    r.ambient = m.ambient, r.diffuse = m.diffuse, r.specular = m.specular, r.shininess = m.shininess;
    r.reflectivity = m.reflectivity, r.transparency = m.transparency, r.refractive_index = m.refractive_index;
    materials.push_back(r);
    return (uint32_t)materials.size() - 1;
  }
  rl_rtc_scene_desc desc() const {
    rl_rtc_scene_desc d{};
    d.triangles = triangles.data(), d.n_triangles = (uint32_t)triangles.size();
    d.groups = groups.data(), d.n_groups = (uint32_t)groups.size();
    d.group_items = group_items.data(), d.n_group_items = (uint32_t)group_items.size();
    d.boundeds = boundeds.data(), d.n_boundeds = (uint32_t)boundeds.size();
    d.transformeds = transformeds.data(), d.n_transformeds = (uint32_t)transformeds.size();
    d.materials = materials.data(), d.n_materials = (uint32_t)materials.size();
    d.objects = objects.data(), d.n_objects = (uint32_t)objects.size();
    d.lights = lights.data(), d.n_lights = (uint32_t)lights.size();
    d.max_reflection_depth = max_reflection_depth;
    d.void_color[0] = void_color[0], d.void_color[1] = void_color[1], d.void_color[2] = void_color[2];
    d.shapes = shapes.data(), d.n_shapes = (uint32_t)shapes.size();
    d.csgs = csgs.data(), d.n_csgs = (uint32_t)csgs.size();
    d.patterns = patterns.data(), d.n_patterns = (uint32_t)patterns.size();
    return d;
  }
};

struct Bounds {  // bounded.rs:11-14
  Point3d minimum, maximum;
  static Bounds from_points(const std::vector<Point3d> &pts) {  // bounded.rs:34-58
    if (pts.empty()) return Bounds{{0, 0, 0}, {0, 0, 0}};
    double mn[3] = {pts[0].x, pts[0].y, pts[0].z}, mx[3] = {pts[0].x, pts[0].y, pts[0].z};
    for (const auto &p : pts) {
      mn[0] = std::fmin(mn[0], p.x), mn[1] = std::fmin(mn[1], p.y), mn[2] = std::fmin(mn[2], p.z);
      mx[0] = std::fmax(mx[0], p.x), mx[1] = std::fmax(mx[1], p.y), mx[2] = std::fmax(mx[2], p.z);
    }
    return Bounds{{mn[0], mn[1], mn[2]}, {mx[0], mx[1], mx[2]}};
  }
};

struct Object {  // scene/object/mod.rs:10-14
  virtual ~Object() {}
  virtual Bounds bounds() const = 0;
  virtual rl_oref flatten(Flattened &f) const = 0;
};
using ObjectPtr = std::shared_ptr<Object>;

struct Triangle : Object {  // object/triangle.rs:22-56
  Point3d points[3];
  Vec3d e1, e2;
  bool is_smooth;
  Vec3d n[3];
  Material material;
  static std::shared_ptr<Triangle> flat(const Point3d p[3], const Material &m) {
    auto t = std::make_shared<Triangle>();
    for (int i = 0; i < 3; i++) t->points[i] = p[i];
    t->e1 = p[1] - p[0];
    t->e2 = p[2] - p[0];
    t->is_smooth = false;
    t->n[0] = normalized(t->e2.cross(t->e1));
    t->n[1] = t->n[2] = Vec3d{0, 0, 0};
    t->material = m;
    return t;
  }
  static std::shared_ptr<Triangle> smooth(const Point3d p[3], const Vec3d nn[3], const Material &m) {
    auto t = std::make_shared<Triangle>();
    for (int i = 0; i < 3; i++) t->points[i] = p[i], t->n[i] = nn[i];
    t->e1 = p[1] - p[0];
    t->e2 = p[2] - p[0];
    t->is_smooth = true;
    t->material = m;
    return t;
  }
  Bounds bounds() const override { return Bounds::from_points({points[0], points[1], points[2]}); }
  rl_oref flatten(Flattened &f) const override {
    rl_rtc_triangle t{};
    t.p1[0] = points[0].x, t.p1[1] = points[0].y, t.p1[2] = points[0].z;
    t.e1[0] = e1.x, t.e1[1] = e1.y, t.e1[2] = e1.z;
    t.e2[0] = e2.x, t.e2[1] = e2.y, t.e2[2] = e2.z;
    t.smooth = is_smooth;
    t.material = f.add_material(material);  // one Material by value per triangle (triangle.rs:26)
    t.n1[0] = n[0].x, t.n1[1] = n[0].y, t.n1[2] = n[0].z;
    t.n2[0] = n[1].x, t.n2[1] = n[1].y, t.n2[2] = n[1].z;
    t.n3[0] = n[2].x, t.n3[1] = n[2].y, t.n3[2] = n[2].z;
    f.triangles.push_back(t);
    return rl_oref{RL_O_TRIANGLE, (uint32_t)f.triangles.size() - 1};
  }
};

// analytic shapes (object/{sphere,plane,cube,cylinder,cone}.rs) in their own object space
struct Shape : Object {
  uint32_t kind;
  Material material;
  bool has_min = false, has_max = false, closed = false;
  double minimum = 0.0, maximum = 0.0;
  Shape(uint32_t k, const Material &m) : kind(k), material(m) {}
  static std::shared_ptr<Shape> sphere(const Material &m = Material{}) { return std::make_shared<Shape>(RL_O_SPHERE, m); }
  static std::shared_ptr<Shape> plane(const Material &m = Material{}) { return std::make_shared<Shape>(RL_O_PLANE, m); }
  static std::shared_ptr<Shape> cube(const Material &m = Material{}) { return std::make_shared<Shape>(RL_O_CUBE, m); }
  static std::shared_ptr<Shape> cylinder(const Material &m, bool hmin, double mn, bool hmax, double mx, bool closed) {
    auto s = std::make_shared<Shape>(RL_O_CYLINDER, m);
    s->has_min = hmin, s->minimum = mn, s->has_max = hmax, s->maximum = mx, s->closed = closed;
    return s;
  }
  static std::shared_ptr<Shape> cone(const Material &m, bool hmin, double mn, bool hmax, double mx, bool closed) {
    auto s = cylinder(m, hmin, mn, hmax, mx, closed);
    s->kind = RL_O_CONE;
    return s;
  }
  Bounds bounds() const override {
    const double I = std::numeric_limits<double>::infinity();
    switch (kind) {
      case RL_O_PLANE: return Bounds{{-I, -1e8, -I}, {I, 1e8, I}};  // plane.rs:44-50
      case RL_O_CYLINDER: return Bounds{{-1.0, has_min ? minimum : -I, -1.0}, {1.0, has_max ? maximum : I, 1.0}};
      case RL_O_CONE: {  // cone.rs:140-150
        double y_min = has_min ? minimum : -I, y_max = has_max ? maximum : I;
        double radius = std::fmax(std::fabs(y_max), std::fabs(y_min));
        return Bounds{{-radius, y_min, -radius}, {radius, y_max, radius}};
      }
      default: return Bounds{{-1.0, -1.0, -1.0}, {1.0, 1.0, 1.0}};
    }
  }
  rl_oref flatten(Flattened &f) const override {
    rl_rtc_shape s{};
    s.kind = kind;
    s.material = f.add_material(material);
    s.has_minimum = has_min, s.has_maximum = has_max, s.closed = closed;
    s.minimum = minimum, s.maximum = maximum;
    f.shapes.push_back(s);
    return rl_oref{kind, (uint32_t)f.shapes.size() - 1};
  }
};

struct Csg : Object {  // object/csg.rs:32-36
  ObjectPtr left, right;
  uint32_t operation;
  Csg(ObjectPtr l, ObjectPtr r, uint32_t op) : left(l), right(r), operation(op) {}
  Bounds bounds() const override {  // Bounds::from_bounds(&[left, right])
    Bounds a = left->bounds(), b = right->bounds();
    return Bounds::from_points({a.minimum, a.maximum, b.minimum, b.maximum});
  }
  rl_oref flatten(Flattened &f) const override {
    rl_rtc_csg c{};
    c.operation = operation;
    c.left = left->flatten(f);
    c.right = right->flatten(f);
    f.csgs.push_back(c);
    return rl_oref{RL_O_CSG, (uint32_t)f.csgs.size() - 1};
  }
};

struct Group : Object {  // object/group.rs
  std::vector<ObjectPtr> children;
  explicit Group(std::vector<ObjectPtr> c) : children(std::move(c)) {}
  Bounds bounds() const override {  // Bounds::from_bounds bounded.rs:60-75
    std::vector<Point3d> pts;
    for (auto &c : children) {
      Bounds b = c->bounds();
      pts.push_back(b.minimum);
      pts.push_back(b.maximum);
    }
    return Bounds::from_points(pts);
  }
  rl_oref flatten(Flattened &f) const override {
    std::vector<rl_oref> refs;
    for (auto &c : children) refs.push_back(c->flatten(f));
    rl_rtc_group g{(uint32_t)f.group_items.size(), (uint32_t)refs.size()};
    f.group_items.insert(f.group_items.end(), refs.begin(), refs.end());
    f.groups.push_back(g);
    return rl_oref{RL_O_GROUP, (uint32_t)f.groups.size() - 1};
  }
};

struct Bounded : Object {  // object/bounded.rs:86-97
  Bounds b;
  ObjectPtr child;
  explicit Bounded(ObjectPtr c) : b(c->bounds()), child(c) {}
  Bounds bounds() const override { return b; }
  rl_oref flatten(Flattened &f) const override {
    rl_rtc_bounded r{};
    r.minimum[0] = b.minimum.x, r.minimum[1] = b.minimum.y, r.minimum[2] = b.minimum.z;
    r.maximum[0] = b.maximum.x, r.maximum[1] = b.maximum.y, r.maximum[2] = b.maximum.z;
    r.child = child->flatten(f);
    f.boundeds.push_back(r);
    return rl_oref{RL_O_BOUNDED, (uint32_t)f.boundeds.size() - 1};
  }
};

inline Point3d mul_point(const Matrix4 &M, const Point3d &p) {  // point.rs:89-96 (w forced to 1)
  double v[4] = {p.x, p.y, p.z, 1.0}, o[3];
  for (int n = 0; n < 3; n++) {
    double sum = 0.0;
    for (int i = 0; i < 4; i++) sum += M.m[n][i] * v[i];
    o[n] = sum;
  }
  return Point3d{o[0], o[1], o[2]};
}

struct Transformed : Object {  // object/transformed.rs:12-27
  ObjectPtr child;
  InvertibleMatrix4 transform;
  Matrix4 inverse_transpose;
  Transformed(ObjectPtr c, const InvertibleMatrix4 &t) : child(c), transform(t), inverse_transpose(t.inverse.transpose()) {}
  Bounds bounds() const override {  // transformed.rs:53-57
    Bounds cb = child->bounds();
    const Point3d &mn = cb.minimum, &mx = cb.maximum;
    std::vector<Point3d> pts = {{mn.x, mn.y, mn.z}, {mn.x, mn.y, mx.z}, {mn.x, mx.y, mn.z}, {mn.x, mx.y, mx.z},
                                {mx.x, mn.y, mn.z}, {mx.x, mn.y, mx.z}, {mx.x, mx.y, mn.z}, {mx.x, mx.y, mx.z}};
    for (auto &p : pts) p = mul_point(transform.matrix, p);
    return Bounds::from_points(pts);
  }
  rl_oref flatten(Flattened &f) const override {
    rl_rtc_transformed t{};
    std::memcpy(t.inverse, transform.inverse.m, sizeof t.inverse);
    std::memcpy(t.inverse_transpose, inverse_transpose.m, sizeof t.inverse_transpose);
    t.child = child->flatten(f);
    f.transformeds.push_back(t);
    return rl_oref{RL_O_TRANSFORMED, (uint32_t)f.transformeds.size() - 1};
  }
};

// ---------------------------------------------------------------- io/wavefront_obj.rs
// v / vn / f / g; vt ignored; fan triangulation; smooth iff all three vertices carry a normal.
struct WavefrontObj {
  uint32_t ignored = 0;
  std::vector<std::pair<std::string, std::vector<std::shared_ptr<Triangle>>>> groups;  // insertion order
  std::vector<Point3d> vertices;
  std::vector<Vec3d> normals;

  static bool parse_floats(const std::string &tail, std::vector<double> &out) {
    std::istringstream ss(tail);
    std::string tok;
    while (ss >> tok) {
      char *end = nullptr;
      double v = std::strtod(tok.c_str(), &end);
      if (end == tok.c_str() || *end != '\0') return false;
      out.push_back(v);
    }
    return true;
  }
  static bool parse_usize(const std::string &s, size_t &out) {
    if (s.empty()) return false;
    size_t i = 0;
    if (s[0] == '+') i = 1;  // Rust usize::from_str accepts a leading '+'
    if (i >= s.size()) return false;
    uint64_t v = 0;
    for (; i < s.size(); i++) {
      if (s[i] < '0' || s[i] > '9') return false;
      v = v * 10 + (uint64_t)(s[i] - '0');
    }
    out = (size_t)v;
    return true;
  }

  static WavefrontObj parse(const std::string &content) {  // wavefront_obj.rs:22-70
    WavefrontObj obj;
    std::string current_name = "\x01" "default";
    std::vector<std::shared_ptr<Triangle>> current;
    auto commit = [&](const std::string &name, std::vector<std::shared_ptr<Triangle>> &&val) {
      for (auto &g : obj.groups)
        if (g.first == name) {  // HashMap::insert replaces
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
        if (head == "v") {
          std::vector<double> ns;
          if (parse_floats(trimmed, ns) && ns.size() == 3) obj.vertices.push_back(Point3d{ns[0], ns[1], ns[2]}), ok = true;
        } else if (head == "vn") {
          std::vector<double> ns;
          if (parse_floats(trimmed, ns) && ns.size() == 3) obj.normals.push_back(Vec3d{ns[0], ns[1], ns[2]}), ok = true;
        } else if (head == "f") {
          std::vector<std::shared_ptr<Triangle>> ts;
          if (obj.parse_face(trimmed, ts)) {
            current.insert(current.end(), ts.begin(), ts.end());
            ok = true;
          }
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

  bool parse_face(const std::string &tail, std::vector<std::shared_ptr<Triangle>> &out) const {  // :103-170
    std::istringstream ss(tail);
    std::string token;
    struct VN {
      size_t v;
      bool has_n;
      size_t n;
    };
    std::vector<VN> idx;
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
      VN e{0, false, 0};
      if (parts.size() == 1 || parts.size() == 2) {
        if (!parse_usize(parts[0], e.v)) return false;
      } else if (parts.size() == 3) {
        if (!parse_usize(parts[2], e.n)) return false;
        if (!parse_usize(parts[0], e.v)) return false;
        e.has_n = true;
      } else
        return false;
      idx.push_back(e);
    }
    for (auto &e : idx) {
      if (e.v < 1 || e.v > vertices.size()) throw std::runtime_error("obj: vertex index out of range");
      if (e.has_n && (e.n < 1 || e.n > normals.size())) throw std::runtime_error("obj: normal index out of range");
    }
    if (idx.size() < 3) return false;
    for (size_t i = 2; i < idx.size(); i++) {  // fan_triangulate :172-183
      const VN *vs[3] = {&idx[0], &idx[i - 1], &idx[i]};
      Point3d p[3] = {vertices[vs[0]->v - 1], vertices[vs[1]->v - 1], vertices[vs[2]->v - 1]};
      if (vs[0]->has_n && vs[1]->has_n && vs[2]->has_n) {
        Vec3d nn[3] = {normals[vs[0]->n - 1], normals[vs[1]->n - 1], normals[vs[2]->n - 1]};
        out.push_back(Triangle::smooth(p, nn, Material{}));
      } else
        out.push_back(Triangle::flat(p, Material{}));
    }
    return true;
  }

  // to_object (:72-75): Bounded(Group(all triangles)); group iteration order is HashMap order in the
  // reference (unspecified); here insertion order.
  ObjectPtr to_object() const {
    std::vector<ObjectPtr> all;
    for (auto &g : groups)
      for (auto &t : g.second) all.push_back(t);
    return std::make_shared<Bounded>(std::make_shared<Group>(std::move(all)));
  }
};

// ---------------------------------------------------------------- scene/world.rs, camera.rs
struct World {
  std::vector<ObjectPtr> objects;
  std::vector<PointLight> lights;
  size_t max_reflection_depth = 5;  // world.rs:167
  Color void_color{0, 0, 0};
  void flatten(Flattened &f) const {
    for (auto &o : objects) f.objects.push_back(o->flatten(f));
    for (auto &l : lights) {
      rl_rtc_light r{};
      r.position[0] = l.position.x, r.position[1] = l.position.y, r.position[2] = l.position.z;
      r.intensity[0] = l.intensity.r, r.intensity[1] = l.intensity.g, r.intensity[2] = l.intensity.b;
      f.lights.push_back(r);
    }
    f.max_reflection_depth = (uint32_t)max_reflection_depth;
    f.void_color[0] = void_color.r, f.void_color[1] = void_color.g, f.void_color[2] = void_color.b;
  }
};

struct RenderOpts {
  size_t anti_aliasing_samples = 1;
};

struct Canvas {  // draw/canvas.rs
  size_t width = 0, height = 0;
  std::vector<double> data;  // row-major W*H*3
};

struct Camera {  // scene/camera.rs:11-57
  size_t hsize, vsize;
  double fov;
  InvertibleMatrix4 transform;
  double pixel_size, half_width, half_height;
  Camera(size_t h, size_t v, double fov_, const InvertibleMatrix4 &t) : hsize(h), vsize(v), fov(fov_), transform(t) {
    double half_view = std::tan(fov / 2.0);
    double aspect = (double)hsize / (double)vsize;
    if (aspect >= 1.0) {
      half_width = half_view;
      half_height = half_view / aspect;
    } else {
      half_width = half_view * aspect;
      half_height = half_view;
    }
    pixel_size = half_width * 2.0 / (double)hsize;
  }
  rl_rtc_camera derived() const {
    rl_rtc_camera c{};
    c.hsize = (uint32_t)hsize, c.vsize = (uint32_t)vsize;
    std::memcpy(c.inverse, transform.inverse.m, sizeof c.inverse);
    c.pixel_size = pixel_size, c.half_width = half_width, c.half_height = half_height;
    return c;
  }
  Canvas render(const World &world, const RenderOpts &opts) const;  // GPU, through the C ABI (host_render.cpp)
};

// Canvas::ppm (draw/canvas.rs:50-97): round(c*255) clamp 0..255, 70-column wrap per row, no gamma
inline std::string canvas_ppm(const double *rgb, size_t width, size_t height) {
  auto translate = [](double c) {
    double v = std::round(c * 255.0);  // f64::round: half away from zero
    int i;
    if (std::isnan(v)) i = 0;
    else if (v >= 2147483647.0) i = 2147483647;
    else if (v <= -2147483648.0) i = (int)-2147483648LL;
    else i = (int)v;
    return i < 0 ? 0 : i > 255 ? 255 : i;
  };
  std::string s = "P3\n" + std::to_string(width) + " " + std::to_string(height) + "\n255\n";
  for (size_t y = 0; y < height; y++) {
    std::string acc;
    size_t line_len = 0;
    for (size_t x = 0; x < width; x++)
      for (int k = 0; k < 3; k++) {
        std::string v = std::to_string(translate(rgb[(y * width + x) * 3 + k]));
        if (x == 0 && k == 0) {
          acc = v;
          line_len = v.size();
        } else if (line_len + v.size() + 1 > 70) {
          acc += "\n" + v;
          line_len = v.size();
        } else {
          acc += " " + v;
          line_len += 1 + v.size();
        }
      }
    s += acc;
    if (y + 1 < height) s += "\n";
  }
  s += "\n";
  return s;
}
inline std::string canvas_ppm(const Canvas &c) { return canvas_ppm(c.data.data(), c.width, c.height); }

}  // namespace rtc
