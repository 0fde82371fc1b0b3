// Internal (not part of the ABI): the device-side scene layout.
//
// rl_*_scene_create "compiles" the caller's object graph (rl_href / rl_oref trees) into a THREADED
// PROGRAM: ops laid out in the reference's depth-first evaluation order, every box op carrying the
// pc to jump to when its test fails.  Traversal is then `pc = hit ? pc + 1 : skip` — no per-lane
// stack at all — and visits nodes and primitives in exactly the order of the reference's recursive
// Hittable::hit / Object::intersect, so closest-hit tie-breaking ("later wins") and every AABB test's
// ray_t.max are identical to the reference (bvh.rs:79-95, hittable/mod.rs:88-105).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rl_render.h"

namespace rl {

// ---------------------------------------------------------------- RTIOW
enum : uint32_t {
  OP_END = 0,
  OP_BOX = 1,        // AABB test; miss -> pc = skip
  OP_BOX_SPH = 2,    // BVH leaf of 1-2 spheres (a, b; b = NONE if single): AABB test then the spheres; next = skip
  OP_SPHERE = 3,     // sphere a
  OP_PLANAR = 4,     // planar a
  OP_BOX_PLANAR = 5, // BVH leaf of 1-2 planars
  OP_PUSH_TRANSLATE = 6,
  OP_POP_TRANSLATE = 7,
  OP_PUSH_TRANSFORM = 8,
  OP_POP_TRANSFORM = 9,
  OP_MEDIUM_BEGIN = 10,  // ConstantMedium a: the boundary's ops follow up to the matching OP_MEDIUM_END at `skip` - 1; next = skip
  OP_MEDIUM_END = 11,    // b = pc of the matching OP_MEDIUM_BEGIN (never stepped by the world traversal)
};
static const uint32_t BOX_FINITE = 0x100u;  // flag OR-ed into a box op's code: all six bounds finite, |b| <= 1e100
static const uint32_t NONE = 0xFFFFFFFFu;
static const uint32_t SPH_MOVING = 0x80000000u;  // flag bit in a sphere index payload
static const uint32_t SPH_UV = 0x40000000u;      // flag bit: the sphere's texture tree holds an Image, so Sphere::hit must produce get_sphere_uv (sphere.rs:70,91-99)
static const uint32_t SPH_INDEX = 0x3FFFFFFFu;

struct alignas(16) DevOp {  // 64 B
  double box[6];            // x.min,x.max,y.min,y.max,z.min,z.max
  uint32_t code, skip, a, b;
};
static_assert(sizeof(DevOp) == 64, "DevOp must be 64 B");

struct alignas(16) CompactOp {  // 32 B: the wave kernel's LDS form of a box / guard op
  float box[6];            // binary32 (round to nearest) box; NaN = never certain
  uint32_t w_hit, w_miss;  // successor words: state << 29 | op index
};
static_assert(sizeof(CompactOp) == 32, "CompactOp must be 32 B");

// Node of the wave kernel's FAST traversal structure (rl_fast_bvh.cpp): a binary tree over the spheres built by the surface-area
// heuristic, one 64-byte record per inner node holding BOTH children's boxes (binary32, rounded outwards) so that one step tests
// the two of them and descends into the nearer one first.  The boxes only ever REJECT (rl_rtiow_wave.h).  child = eA | eB << 16 with
// entry ids e: inner node i -> i, sphere s -> n_inner + s; FAST_NONE = empty stack.
static const uint32_t FAST_NONE = 1023u;
static const uint32_t FAST_MAX_DEPTH = 14;  // the per-lane stack holds 15 ten-bit entries above its sentinel
struct alignas(16) FastNode {
  float box[2][6];  // [child][x.min, x.max, y.min, y.max, z.min, z.max]
  uint32_t child;   // eA | eB << 16
  uint32_t pad[3];
};
static_assert(sizeof(FastNode) == 64, "FastNode must be 64 B");

struct alignas(16) DevSphere {  // 64 B
  double c0[3];
  double dc[3];   // center1 - center0 (sphere.rs:27: p2 - p1), 0 when stationary
  double r2;      // radius * radius   (sphere.rs:40)
  double inv_r;   // 1.0 / radius      (vec3.rs:177-179 via sphere.rs:62)
};
static_assert(sizeof(DevSphere) == 64, "DevSphere must be 64 B");

struct alignas(16) DevPlanar {  // 128 B core (plane.rs:12-20) + optional per-vertex data
  double q[3], u[3], v[3], w[3], normal[3];
  double d;
  uint32_t kind, material, has_normals, has_uvs;
  double normals[9];
  double uvs[6];
  double pad;
};

static const uint32_t MAT_TEX_SOLID = 0x100u;  // flag in a flattened per-sphere material's kind: albedo holds the Solid texture's colour
struct DevMaterial {  // 48 B
  uint32_t kind, texture;
  double albedo[3];
  double fuzz, ior;
};
struct DevTexture {  // 48 B
  uint32_t kind, even, odd, image;
  double color[3];
  double inv_scale;
};
struct DevImage {
  uint32_t width, height;
  uint64_t offset;  // float offset into the image pool
};

struct RtiowProgram {
  std::vector<DevOp> ops;
  std::vector<DevSphere> spheres;
  std::vector<uint32_t> sphere_material;
  std::vector<DevPlanar> planars;
  std::vector<rl_translate> translates;
  std::vector<rl_transform> transforms;
  std::vector<DevMaterial> materials;
  std::vector<DevTexture> textures;
  std::vector<DevImage> images;
  std::vector<float> image_pool;
  std::vector<rl_perlin> perlins;
  std::vector<rl_medium> media;
  std::vector<uint8_t> sphere_uv;  // per sphere: texture tree reaches an Image
  bool has_planars = false, has_instances = false, has_images = false, has_noise = false, has_sphere_uv = false, has_media = false;
  uint32_t max_instance_depth = 0;
};

// Where the reject-only boxes' padding is valid (rl_fast_bvh.cpp).
struct GuardFrame {
  double center[3];  // centre of the scene's bounding box
  double half;       // half of its diagonal
  double reach;      // the camera must be within this distance of `center` (checked per render)
  double L;          // bound on |ray origin - sphere centre| for every ray the kernel traces
  double M;          // bound on every coordinate magnitude of origins and centres
  bool normals_safe; // no Sphere::hit of this scene can trip the from_normalized assert (vec3.rs:219) for rays inside the frame
};
GuardFrame guard_frame(const rl_rtiow_scene_desc &d);
double guard_pad(const GuardFrame &f, double radius);  // how far outside a sphere a reject-only box must stay
// Fast traversal structure for a sphere-only scene; false when the scene does not qualify (too many spheres for the LDS budget,
// degenerate spheres, spheres referenced more than once or not at all...).  root_entry: entry id a new ray starts at.
bool build_fast_bvh(const rl_rtiow_scene_desc &d, const RtiowProgram &rt, const GuardFrame &f, std::vector<FastNode> &nodes, uint32_t &root_entry);

// ---- fast traversal structure for GENERAL scenes (planars, instances, any number of primitives; rl_fast_bvh.cpp, rl_rtiow_fastgen.h):
// a surface-area-heuristic binary tree in WORLD space over the primitive OCCURRENCES of the threaded program (a primitive under a
// Translate / Transform chain = one item per occurrence), boxes reject-only, nodes and items read from HBM / L2 / Infinity Cache.
static const uint32_t FASTG_LEAF = 0x80000000u;  // child / entry: bit 31 set -> item index, else inner node index
static const uint32_t FASTG_MAX_DEPTH = 40;      // the per-lane LDS stack holds 40 entries
struct alignas(16) FastNodeG {  // the binary tree as built (host side only; the device walks the four-wide form below)
  float box[2][6];    // [child][x.min, x.max, y.min, y.max, z.min, z.max], rounded outwards
  uint32_t child[2];
  uint32_t pad[2];
};
static_assert(sizeof(FastNodeG) == 64, "FastNodeG must be 64 B");
// The same tree with every second level folded away — what the device walks: up to four children per node, 128 B = one L2 line (one
// dependent fetch per TWO levels of the binary tree).  Children are the binary tree's, largest box expanded first; an empty slot has
// child == NONE.
struct alignas(16) FastNodeQ {
  float lo[3][4];  // [axis][child], rounded outwards (the binary tree's own numbers)
  float hi[3][4];
  uint32_t child[4];
  uint32_t pad[4];
};
static_assert(sizeof(FastNodeQ) == 128, "FastNodeQ must be 128 B");
// Eight-wide form with QUANTISED child boxes (round 3; EXPERIMENTAL: measured slower than the four-wide nodes, kept for A/B in
// librl_render_exp.so with RL_FASTG_OCTO=1): the binary tree folded until a node holds up to eight children, their boxes stored
// relative to the node's own lower corner on an 8-bit grid of power-of-two pitch per axis,
//     plane(axis, k) = o[axis] + q[axis][k] * 2^(e[axis] - 127)        (qlo rounded down, qhi rounded up: the stored box CONTAINS the child's)
// so that one dependent fetch (six 16-byte loads instead of seven) advances a ray THREE levels of the binary tree: a third fewer dependent
// steps per ray than the four-wide nodes, half their L1 accesses.  Empty slots: child == NONE (and an inverted box).
struct alignas(16) FastNodeO {
  float o[3];          // lower corner of the node's box (the minimum over the children's boxes, exact)
  uint32_t exps;       // e[0] | e[1] << 8 | e[2] << 16: biased binary32 exponents of the grid pitch (as_float(e << 23) = 2^(e - 127))
  uint8_t qlo[3][8];   // [axis][child]
  uint8_t qhi[3][8];
  uint32_t child[8];
  uint32_t pad[8];
};
static_assert(sizeof(FastNodeO) == 128, "FastNodeO must be 128 B");
struct alignas(16) FastItem {
  uint32_t kind;     // 0 sphere, 1 planar
  uint32_t payload;  // sphere payload (index | SPH_MOVING | SPH_UV) or planar index
  uint32_t chain;    // pc of the innermost PUSH op around the primitive (NONE: world space); ops[pc].b = the next one outwards
  uint32_t op_pc;    // the op that holds the primitive in the reference's program
};
// A ConstantMedium of the scene as the fast traversal sees it (round 3): the reference evaluates it where its fold reaches it — after
// every primitive that precedes it in the threaded program — with ray_t.max = the closest hit found so far (constant_medium.rs:27-80).
// The items are therefore cut into SEGMENTS at the media (segment s = the primitive occurrences between medium s - 1 and medium s in
// program order), each segment has its own tree, and a ray walks tree 0, evaluates medium 0, walks tree 1, ... (rl_rtiow_fastgen.h MEDIA).
static const uint32_t FASTG_MEDIUM = 0x40000000u;  // entry / `best`: a medium (low bits: its index in FastGeneral::media)
static const uint32_t FASTG_TOP_NODES = 512;  // the breadth-first prefix of the node array; a launch stages as many of them as its LDS has room for
struct FastMedium {
  uint32_t pc;     // its OP_MEDIUM_BEGIN
  uint32_t chain;  // pc of the innermost PUSH op around it (NONE: world space)
  // the boundary's shape, where it is simple enough for both boundary hits of constant_medium.rs:28-40 to come out of ONE evaluation:
  // 1 = PUSH* PLANAR+ POP* (quads / triangles without vertex normals, e.g. a rotated and translated box of six quads),
  // 2 = PUSH* SPHERE POP*, 0 = anything else (the reference's fold over the boundary's ops, twice)
  uint32_t shape;
  uint32_t first, count;  // shape 1, 2: pc of the first primitive op, number of primitive ops
  uint32_t chain_in;      // shape 1, 2: pc of the innermost PUSH op around the primitives (NONE: world space)
};
struct FastGeneral {
  std::vector<FastNodeG> nodes;
  std::vector<FastNodeQ> qnodes;  // four-wide form of `nodes` (collapse_fast_general)
  std::vector<FastNodeO> onodes;  // eight-wide form with quantised boxes (what the product kernel walks)
  uint32_t oroot = NONE;
  std::vector<FastItem> items;
  std::vector<DevSphere> item_spheres;  // [item]: the sphere record of a sphere item (zero for planars), so that LEAF fetches item and sphere side by side
  std::vector<uint32_t> item_material;  // [item]: material index of the primitive
  uint32_t root = NONE;      // entry a new ray starts at (NONE: nothing to hit)
  uint32_t qroot = NONE;     // ... in the four-wide form (= seg_roots[0])
  std::vector<uint32_t> seg_roots;  // four-wide root entry of every segment (one segment, no media: {qroot})
  std::vector<FastMedium> media;    // seg_roots.size() == media.size() + 1
  // what a ray walks, in order: per segment its unbounded Planes (leaf entries: every ray tests them) and its tree, then the medium behind
  // it (a one-child box node, or the medium's leaf entry itself when its boundary holds a Plane); empty stages are left out
  std::vector<uint32_t> stage_roots;
  uint32_t top_nodes = 0;  // qnodes[0 .. top_nodes) = the top of segment 0's tree in breadth-first order (what the kernels stage in LDS)
  std::vector<uint32_t> media_stage;  // index of medium k's stage in stage_roots (NONE: the boundary has no parts)
  float center[3] = {0, 0, 0};  // rays whose origin is within r_safe (Euclidean) of `center` may use the structure; the rest
  float r_safe = 0;             // walk it with grown boxes and without pruning by the closest hit (rl_rtiow_fastgen.h start_ray)
  float radius = 0, pad_k = 0;  // ... box growth for those rays = pad_k * (distance to centre + radius)^2 (world units)
  bool ok = false;
};
bool build_fast_general(const rl_rtiow_scene_desc &d, const RtiowProgram &rt, FastGeneral &out);
void set_build_octo(bool on);  // experimental library, RL_FASTG_OCTO=1: also build FastGeneral::onodes

// Returns RL_OK or RL_E_INVALID (err filled).
int compile_rtiow(const rl_rtiow_scene_desc &d, RtiowProgram &out, std::string &err);

// ---------------------------------------------------------------- RTC
enum : uint32_t {
  ROP_END = 0,
  ROP_ENTER = 1,   // Transformed: a = transformed index
  ROP_EXIT = 2,    // a = transformed index
  ROP_BOUNDS = 3,  // Bounded: box test (tmin <= tmax); miss -> pc = skip
  ROP_TRIS = 4,    // triangles [a, a+b)
  ROP_SHAPE = 5,   // analytic shape a (sphere / plane / cube / cylinder / cone)
  ROP_CSG_BEGIN = 6,  // a = csg index; children follow: left ops, ROP_CSG_MID, right ops, ROP_CSG_END
  ROP_CSG_MID = 7,
  ROP_CSG_END = 8,    // a = csg index
};

struct alignas(16) DevTri {  // 160 B
  double p1[3], e1[3], e2[3];
  double n1[3], n2[3], n3[3];
  uint32_t smooth, material;
  double pad;
};
static_assert(sizeof(DevTri) == 160, "DevTri must be 160 B");

struct alignas(16) RtcGuard {  // 32 B: node of the reject-only binary32 box tree over one ROP_TRIS range (rl_render.hip build_rtc_guards)
  float box[6];   // x.min,x.max,y.min,y.max,z.min,z.max of the triangles below, padded outwards
  uint32_t skip;  // next node when this box is certainly missed (inner: hit -> next node in the array)
  uint32_t tri;   // leaf: the triangle to test; NONE for inner nodes
};
static_assert(sizeof(RtcGuard) == 32, "RtcGuard must be 32 B");

struct RtcProgram {
  std::vector<DevOp> ops;
  std::vector<DevTri> tris;
  std::vector<rl_rtc_transformed> xforms;
  std::vector<rl_rtc_material> materials;
  std::vector<rl_rtc_light> lights;
  uint32_t max_reflection_depth = 5;
  double void_color[3] = {0, 0, 0};
  uint32_t max_xform_depth = 0;
  bool needs_secondary = false;  // any reflective / transparent material
  bool needs_full = false;       // shapes / CSG / patterns / secondary rays: the full color_at kernel
  std::vector<rl_rtc_shape> shapes;
  std::vector<rl_rtc_csg> csgs;
  std::vector<rl_rtc_pattern> patterns;
  uint32_t max_csg_depth = 0;
};
int compile_rtc(const rl_rtc_scene_desc &d, RtcProgram &out, std::string &err);

}  // namespace rl
