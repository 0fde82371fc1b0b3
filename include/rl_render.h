/*
 * rl_render.h — the C-ABI drop-in boundary for the per-pixel / per-ray hot path of
 * marcantony/rendering-learning, rebuilt as hand-written HIP for MI355X (gfx950).
 *
 * The reference has NO existing FFI; this header *creates* the seam.  Each entry point
 * names the reference interface it replaces (paths relative to /root/reference):
 *
 *   rl_rtiow_render*      <- ray-tracing-one-weekend/src/camera.rs:122  Camera::render
 *                            camera.rs:136  Camera::render_from_checkpoint (first_sample)
 *                            camera.rs:145-199  Camera::_render  (the per-pixel loop)
 *   rl_rtc_render*        <- ray-tracer-challenge/src/scene/camera.rs:93  Camera::render
 *                            scene/mod.rs:24  Scene::render
 *   rl_rtiow_scene_create <- what `world: H where H: Hittable` carries into render()
 *                            (hittable/mod.rs:40-43 trait, bvh.rs:11-20, sphere.rs:16-21 ...)
 *   rl_rtc_scene_create   <- scene/world.rs:26-31  World{objects,lights,max_reflection_depth,void_color}
 *
 * Conventions: plain pointers and sizes only; 0 on success, negative RL_E_* otherwise; the
 * library never unwinds or aborts across the boundary (every reference panic site becomes an
 * error code or a flagged-pixel count); scene_create deep-copies, so the caller may free its
 * arrays as soon as it returns; the caller owns all host pointers.
 * Concurrency: as Camera::render takes `&self` and a `Sync` world (camera.rs:122), every render entry point may be
 * called on ONE rl_scene from several host threads and on several HIP streams at once.  The scene's program is
 * immutable; it owns one set of work buffers (counters, the cost-sorted tile order), so the library serialises the
 * host side per scene and makes a render wait on the device for the scene's previous one when their streams differ:
 * concurrent callers get the frames a lone caller gets, rendered one after the other (each fills the GPU).  Use
 * several scene handles for renders that should share the GPU.  rl_init / rl_init_multi / rl_shutdown / scene
 * creation and destruction are NOT to be raced against renders.
 * Everything is IEEE binary64 unless stated.  There is NO CPU fallback: without a GPU / without
 * the HIP code object every compute entry point fails with RL_E_NO_DEVICE.
 */
#ifndef RL_RENDER_H
#define RL_RENDER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RL_ABI_VERSION 6

/* ------------------------------------------------------------------ errors */
#define RL_OK 0
#define RL_E_INVALID (-1)     /* bad argument / malformed scene graph (out-of-range index, cycle) */
#define RL_E_NO_DEVICE (-2)   /* no HIP device / library not initialised */
#define RL_E_DEVICE (-3)      /* HIP runtime error (text in rl_last_error) */
#define RL_E_UNSUPPORTED (-4) /* scene uses a feature this build has no kernel for */
#define RL_E_DEGENERATE (-5)  /* a reference panic site was reached; output is written, stats.flagged > 0 */
#define RL_E_NOMEM (-6)

/* ------------------------------------------------------------------ lifetime */
/* device < 0: keep the process's current HIP device.  One GPU (use this form for one process per GPU, e.g. under
 * torch.distributed / MPI, with rl_*_render_device + the caller's own gather). */
int rl_init(int device);
/* One process, n_devices GPUs (0 = every visible one): the reference calls ONE Camera::render(&world) from one thread
 * (ray-tracing-one-weekend/src/camera.rs:122, examples/common/mod.rs:16; ray-tracer-challenge/src/scene/camera.rs:93), so a drop-in
 * host reaches all GPUs of the node through rl_*_render_multi.  The library owns one stream per device and the RCCL communicators
 * (ncclCommInitAll, one rank per GPU; librccl is dlopen'ed here, a single-GPU host never loads it).  Scenes created AFTERWARDS are
 * replicated on every device. */
int rl_init_multi(int n_devices);
int rl_device_count(void); /* device contexts the library drives: 1 after rl_init, n after rl_init_multi, 0 before either */
void rl_shutdown(void);
const char *rl_last_error(void); /* thread-local, owned by the library */
int rl_abi_version(void);
/* Fills name (NUL-terminated, <= cap) with the device's gcnArchName; returns CU count or <0. */
int rl_device_info(char *name, int cap);

typedef struct rl_scene rl_scene;
void rl_scene_destroy(rl_scene *);

/* per-render counters; every field is a sum over the pixels rendered by the call.  They count the calls the REFERENCE's
 * algorithm makes on the same input (the oracle's counters are equal, which is how the tests prove that every branch went the
 * same way): where the device proves with a conservative bounding-box test that a Sphere::hit / Triangle::intersect cannot
 * produce an intersection the reference would keep and skips the arithmetic, the call is still counted. */
typedef struct rl_stats {
  uint64_t rays;            /* RTIOW: ray_color calls with depth>0 (camera.rs:232). RTC: color_at + shadow rays */
  uint64_t node_tests;      /* AABB::hit calls (aabb.rs:123) / Bounded::test (bounded.rs:100) */
  uint64_t sphere_tests;    /* Sphere::hit calls (sphere.rs:34) */
  uint64_t planar_tests;    /* Plane::hit_ab (plane.rs:51) / RTC Triangle::intersect (triangle.rs:63) */
  uint64_t instance_enters; /* Transform::hit + Translate::hit / RTC Transformed::intersect */
  uint64_t rng_words;       /* ChaCha8 u32 words consumed (RTIOW only) */
  uint64_t flagged;         /* reference panic sites reached (see RL_E_DEGENERATE) */
  double kernel_ms;         /* device time of the render kernel(s), HIP events on the launch stream */
} rl_stats;

/* =====================================================================
 *  RTIOW  (ray-tracing-one-weekend)
 * ===================================================================== */

/* reference to any Hittable: (kind, index into that kind's array) */
typedef struct rl_href {
  uint32_t kind;
  uint32_t index;
} rl_href;
enum {
  RL_H_NONE = 0,
  RL_H_SPHERE = 1,    /* hittable/sphere.rs:16 Sphere<M> */
  RL_H_PLANAR = 2,    /* hittable/flat/{plane,quad,triangle}.rs */
  RL_H_TRANSLATE = 3, /* hittable/translate.rs:6 */
  RL_H_TRANSFORM = 4, /* hittable/transform.rs:13 */
  RL_H_BVH = 5,       /* bvh.rs:11 Bvh<H> node */
  RL_H_LIST = 6,      /* hittable/mod.rs:88 impl Hittable for [H] */
  RL_H_MEDIUM = 7     /* hittable/constant_medium.rs:9 ConstantMedium (deterministic variant, see rl_medium) */
};

typedef struct rl_sphere { /* sphere.rs:11-21 */
  double center0[3];       /* Center::Stationary(p) or Moving(p1, _) */
  double center1[3];       /* Moving(_, p2); ignored unless moving */
  double radius;
  uint32_t moving;
  uint32_t material;
} rl_sphere;

enum { RL_PLANAR_PLANE = 0, RL_PLANAR_QUAD = 1, RL_PLANAR_TRIANGLE = 2 };
typedef struct rl_planar { /* flat/plane.rs:12-20 + quad.rs:13 + triangle.rs:13-18 */
  double q[3], u[3], v[3];
  double w[3];      /* n/(n.n), n = u x v            (plane.rs:30) */
  double normal[3]; /* unit(n)                       (plane.rs:26) */
  double d;         /* normal . q                    (plane.rs:28) */
  uint32_t kind;    /* RL_PLANAR_* */
  uint32_t material;
  uint32_t has_normals; /* triangle.rs:16 Option<[Vec3;3]> */
  uint32_t has_uvs;     /* triangle.rs:17 Option<[(f64,f64);3]> */
  double normals[9];    /* v1,v2,v3 */
  double uvs[6];        /* (u,v) x3 */
} rl_planar;

typedef struct rl_translate { /* translate.rs:6-9 */
  double offset[3];
  rl_href child;
} rl_translate;

typedef struct rl_transform { /* transform.rs:13-19; row-major 3x3 */
  double m[9], inv[9], inv_t[9];
  rl_href child;
} rl_transform;

typedef struct rl_bvh_node { /* bvh.rs:11-20 */
  double bbox[6];            /* x.min,x.max,y.min,y.max,z.min,z.max (aabb.rs:7-11), already padded */
  uint32_t n_children;       /* 1 or 2, in the reference's stored order */
  uint32_t reserved;
  rl_href child[2];
} rl_bvh_node;

/* hittable/constant_medium.rs:9-80 with ONE deliberate difference: the reference draws the free path from the process-global
 * `rand::random::<f64>()` (constant_medium.rs:55; its own comment: "This breaks deterministic/repeatable renders"), which no seeded
 * reference run can reproduce.  Here the draw comes from the pixel's ChaCha8 stream — the `rng` Camera::_render hands to scatter —
 * at the moment ConstantMedium::hit reaches it, in the reference's evaluation order: hit_distance = neg_inv_density * ln(gen::<f64>()).
 * Everything else is the reference's: boundary.hit(r, universe), boundary.hit(r, [t1 + 1e-4, inf]), clamping to ray_t, the arbitrary
 * normal (1, 0, 0) / uv (0, 0) / Face::Front.  The boundary may be any hittable except another medium. */
typedef struct rl_medium {
  rl_href boundary;
  double neg_inv_density; /* -1.0 / density (constant_medium.rs:19) */
  uint32_t material;      /* phase function: meant to be RL_MAT_ISOTROPIC */
  uint32_t reserved;
} rl_medium;

typedef struct rl_list { /* a slice of hittables; items live in list_items[first .. first+count) */
  uint32_t first, count;
} rl_list;

enum {
  RL_MAT_FLAT = 0,       /* material.rs:52 */
  RL_MAT_LAMBERTIAN = 1, /* material.rs:69 */
  RL_MAT_METAL = 2,      /* material.rs:99 */
  RL_MAT_DIELECTRIC = 3, /* material.rs:134 */
  RL_MAT_DIFFUSE_LIGHT = 4, /* material.rs:178 */
  RL_MAT_ISOTROPIC = 5      /* material.rs:197: scatters into Vec3::random_unit_vector, attenuation = texture (uses `texture`) */
};
typedef struct rl_material {
  uint32_t kind;
  uint32_t texture;  /* Lambertian / DiffuseLight */
  double albedo[3];  /* Metal */
  double fuzz;       /* Metal */
  double ior;        /* Dielectric.refraction_index */
} rl_material;

enum { RL_TEX_SOLID = 0, RL_TEX_CHECKER = 1, RL_TEX_IMAGE = 2, RL_TEX_NOISE = 3 };
typedef struct rl_texture { /* texture.rs:15,25,58,84 */
  uint32_t kind;
  uint32_t even, odd; /* Checker: texture ids */
  uint32_t image;     /* Image: image id;  Noise: perlin id */
  double color[3];    /* SolidColor.albedo */
  double inv_scale;   /* Checker.inv_scale;  Noise.scale */
} rl_texture;

/* perlin.rs:9-14: the tables Perlin::new drew from the caller's Rng (the device only evaluates noise()/turb()) */
typedef struct rl_perlin {
  double randvec[256][3];
  uint32_t perm_x[256], perm_y[256], perm_z[256]; /* each a permutation of 0..255 */
} rl_perlin;

typedef struct rl_image { /* texture.rs:58 Image{Rgb32FImage}: linear RGB f32, row-major, top row first */
  uint32_t width, height;
  const float *rgb;
} rl_image;

typedef struct rl_rtiow_scene_desc {
  const rl_sphere *spheres;       uint32_t n_spheres;
  const rl_planar *planars;       uint32_t n_planars;
  const rl_translate *translates; uint32_t n_translates;
  const rl_transform *transforms; uint32_t n_transforms;
  const rl_bvh_node *bvh_nodes;   uint32_t n_bvh_nodes;
  const rl_list *lists;           uint32_t n_lists;
  const rl_href *list_items;      uint32_t n_list_items;
  const rl_material *materials;   uint32_t n_materials;
  const rl_texture *textures;     uint32_t n_textures;
  const rl_image *images;         uint32_t n_images;
  rl_href root;
  const rl_perlin *perlins;       uint32_t n_perlins; /* ABI v3 */
  const rl_medium *media;         uint32_t n_media;   /* ABI v5 */
} rl_rtiow_scene_desc;

/* The DERIVED camera: outputs of Camera::new (camera.rs:72-118). The host keeps Camera::new
 * (tan() stays on the host); the device receives the vectors. */
typedef struct rl_rtiow_camera {
  uint32_t image_width, image_height;
  uint32_t samples_per_pixel, max_depth;
  double lookfrom[3];
  double pixel_00[3], pixel_du[3], pixel_dv[3];
  double defocus_disk_u[3], defocus_disk_v[3];
  double defocus_angle;
  double background[3];
  uint64_t seed;
} rl_rtiow_camera;

rl_scene *rl_rtiow_scene_create(const rl_rtiow_scene_desc *desc);

/* Bvh::new (bvh.rs:22-60) on the device, for worlds too big to build comfortably on the host (1 M spheres: seconds).
 * Input: n hittables as their bounding boxes (6 doubles each: x.min, x.max, y.min, y.max, z.min, z.max = what
 * Hittable::bounding_box() returns, already padded by AABB::new) and their hrefs.  Output: the nodes in the order the
 * reference's recursion creates them (node, left subtree, right subtree; node 0 is the root): box = merge of the boxes
 * below (aabb.rs:135), split on the longest axis (bvh.rs:63-77) after a STABLE sort by `box.axis.min` under
 * f64::total_cmp (bvh.rs:49 uses sort_unstable_by: the order of equal keys is implementation-defined there), left half =
 * the first len / 2, leaves of 1-2 hittables.  Inner nodes reference their children as RL_H_BVH node_base + index, so the
 * records can be appended to a scene's bvh_nodes at position node_base.  *out_n_nodes receives the node count (also
 * when cap is too small: RL_E_INVALID).  Host pointers. */
int rl_bvh_build(const double *prim_boxes, const rl_href *prims, uint32_t n, uint32_t node_base, rl_bvh_node *out_nodes,
                 uint32_t cap, uint32_t *out_n_nodes);

/* Replaces Camera::render / render_from_checkpoint's _render(first_sample, world).
 * out_rgb_sum: caller-owned host buffer, W*H*3 doubles, row-major, holds SUMS over samples
 * exactly like Canvas.data (camera.rs:269). */
int rl_rtiow_render(const rl_scene *, const rl_rtiow_camera *, uint64_t first_sample,
                    double *out_rgb_sum, rl_stats *opt_stats);

/* Row-sharded form (multi-GPU: rank g renders rows g, g+G, ...). Output holds only those rows,
 * compact: nrows = ceil((H - row_first) / row_step), each W*3 doubles. Host output buffer. */
int rl_rtiow_render_rows(const rl_scene *, const rl_rtiow_camera *, uint64_t first_sample,
                         uint32_t row_first, uint32_t row_step, double *out_rgb_sum,
                         rl_stats *opt_stats);

/* Same, but the output stays in HBM: d_out_rgb_sum is a DEVICE pointer (e.g. a torch tensor's
 * data_ptr), hip_stream is a hipStream_t (NULL = default stream). Asynchronous unless opt_stats
 * is non-NULL (stats need a sync). This is what bench.py times. */
int rl_rtiow_render_device(const rl_scene *, const rl_rtiow_camera *, uint64_t first_sample,
                           uint32_t row_first, uint32_t row_step, void *d_out_rgb_sum,
                           void *hip_stream, rl_stats *opt_stats);

/* Completion + status of the ASYNCHRONOUS renders of this scene since the last call (rl_*_render_device / rl_*_render_multi_device
 * with opt_stats == NULL): waits for all of them, fills opt_stats->rays with the ray count of the most recently enqueued one and
 * ->flagged with the panic sites reached by any of them (the other counters need a counting render) and returns RL_E_DEGENERATE
 * if a reference panic site (camera.rs:86, material.rs:151, vec3.rs:220, ...) was reached, else RL_OK. */
int rl_render_status(const rl_scene *, rl_stats *opt_stats);

/* Progress of the RTIOW render that is executing on this scene — the reference logs "Scanline-equivalents remaining" once per `image_width`
 * finished pixels (camera.rs:176-184); a host that wants that line polls this from another thread.  pixels_claimed: pixel slots the
 * kernel's lanes have taken so far in the launch that is running; pixels_total: slots of that launch (the shard's pixels rounded up to
 * 8 x 8 tiles); phase: a render of >= 64 samples per pixel is two launches (0: samples [0, 8) of every pixel, 1: the rest).  Never waits:
 * the FIRST call switches progress counting on for the renders enqueued after it (their work counters then live in pinned host memory the
 * kernels reach over PCIe, one atomic per wave per 64 pixels) and reports zeros; later calls are two host loads.  Frames of at most ~41 k
 * pixels (cooperative kernel) and counting renders report through the same words.  All three outputs are optional (NULL). */
int rl_rtiow_render_progress(const rl_scene *, uint64_t *pixels_claimed, uint64_t *pixels_total, uint32_t *phase);

/* Camera::render on every GPU of rl_init_multi: image row r is rendered by GPU r mod G with the single-GPU
 * kernels (no collective during the render), then ONE exchange — ncclSend / ncclRecv of ceil(H/G)*W*3 f64 per peer to GPU 0 in one
 * RCCL group, each peer over its own xGMI link — and a de-interleave kernel on GPU 0.  The frame is bit-identical for every G.
 * out_rgb_sum: host buffer, W*H*3 doubles (Canvas.data order, sums).  With G = 1 this is rl_rtiow_render. */
int rl_rtiow_render_multi(const rl_scene *, const rl_rtiow_camera *, uint64_t first_sample, double *out_rgb_sum, rl_stats *opt_stats);
/* Same with the frame left in GPU 0's HBM (d_out_rgb_sum: device pointer on device 0).  Asynchronous on the library's streams
 * unless opt_stats is non-NULL; rl_render_status(scene) waits for the frame. */
int rl_rtiow_render_multi_device(const rl_scene *, const rl_rtiow_camera *, uint64_t first_sample, void *d_out_rgb_sum, rl_stats *opt_stats);

/* Output stage on the device (color.rs:22-57, output.rs:5-14): mean = sum * (1/samples), linear_to_srgb,
 * floor(v * 255.999) clamped to 0..255.  d_rgb_sum / d_rgb8 are DEVICE pointers (n_pixels*3 f64 / u8). */
int rl_rtiow_encode_rgb8_device(const void *d_rgb_sum, uint64_t n_pixels, uint32_t samples, void *d_rgb8, void *hip_stream);
/* Render + encode on the device, copy back only the W*H*3 bytes a P3 PPM prints (24x less PCIe traffic). */
int rl_rtiow_render_rgb8(const rl_scene *, const rl_rtiow_camera *, uint64_t first_sample, uint8_t *out_rgb8, rl_stats *opt_stats);

/* =====================================================================
 *  RTC  (ray-tracer-challenge)
 * ===================================================================== */

typedef struct rl_oref { /* reference to any Object (scene/object/mod.rs:10) */
  uint32_t kind;
  uint32_t index;
} rl_oref;
enum {
  RL_O_NONE = 0,
  RL_O_TRIANGLE = 1,    /* object/triangle.rs:22 */
  RL_O_GROUP = 2,       /* object/group.rs:14 */
  RL_O_BOUNDED = 3,     /* object/bounded.rs:86 */
  RL_O_TRANSFORMED = 4, /* object/transformed.rs:12 */
  RL_O_SPHERE = 5,      /* object/sphere.rs:10   (index into shapes[]) */
  RL_O_PLANE = 6,       /* object/plane.rs:11 */
  RL_O_CUBE = 7,        /* object/cube.rs:10 */
  RL_O_CYLINDER = 8,    /* object/cylinder.rs:13 */
  RL_O_CONE = 9,        /* object/cone.rs:13 */
  RL_O_CSG = 10         /* object/csg.rs:32      (index into csgs[]) */
};

typedef struct rl_rtc_triangle { /* triangle.rs:22-27 */
  double p1[3];
  double e1[3], e2[3];
  uint32_t smooth;  /* TriangleNormal::Smooth vs Flat */
  uint32_t material;
  double n1[3], n2[3], n3[3]; /* smooth: vertex normals; flat: n1 = the flat normal */
} rl_rtc_triangle;

typedef struct rl_rtc_group { /* children live in group_items[first .. first+count) */
  uint32_t first, count;
} rl_rtc_group;

typedef struct rl_rtc_bounded { /* bounded.rs:11-14,86-89 */
  double minimum[3], maximum[3];
  rl_oref child;
} rl_rtc_bounded;

typedef struct rl_rtc_transformed { /* transformed.rs:12-16; row-major 4x4 */
  double inverse[16];
  double inverse_transpose[16];
  rl_oref child;
} rl_rtc_transformed;

typedef struct rl_rtc_material { /* scene/material.rs:22-31 */
  double color[3];   /* Surface::Color(c) when pattern == 0 */
  double ambient, diffuse, specular, shininess;
  double reflectivity, transparency, refractive_index;
  uint32_t pattern;  /* 0: Surface::Color; k > 0: Surface::Pattern(patterns[k-1]) */
  uint32_t reserved;
} rl_rtc_material;

/* analytic shapes in their own object space: unit sphere, xz plane, [-1,1]^3 cube, y-axis cylinder / double cone */
typedef struct rl_rtc_shape { /* object/{sphere,plane,cube,cylinder,cone}.rs */
  uint32_t kind;      /* RL_O_SPHERE .. RL_O_CONE */
  uint32_t material;
  uint32_t has_minimum, has_maximum; /* cylinder / cone: Option<f64> */
  uint32_t closed, reserved;
  double minimum, maximum;
} rl_rtc_shape;

enum { RL_CSG_UNION = 0, RL_CSG_INTERSECTION = 1, RL_CSG_DIFFERENCE = 2 };
typedef struct rl_rtc_csg { /* object/csg.rs:9-36 */
  uint32_t operation;
  uint32_t reserved;
  rl_oref left, right;
} rl_rtc_csg;

enum { RL_PAT_STRIPE = 1, RL_PAT_RING = 2, RL_PAT_GRADIENT = 3, RL_PAT_CHECKER3D = 4 };
typedef struct rl_rtc_pattern { /* scene/pattern/{stripe,ring,gradient,checker3d}.rs: two colours + the pattern's own transform.inverse() (row-major) */
  uint32_t kind;
  uint32_t reserved;
  double a[3], b[3];
  double inverse[16];
} rl_rtc_pattern;

typedef struct rl_rtc_light { /* scene/light.rs:4-7 */
  double position[3];
  double intensity[3];
} rl_rtc_light;

typedef struct rl_rtc_scene_desc { /* scene/world.rs:26-31 */
  const rl_rtc_triangle *triangles;       uint32_t n_triangles;
  const rl_rtc_group *groups;             uint32_t n_groups;
  const rl_oref *group_items;             uint32_t n_group_items;
  const rl_rtc_bounded *boundeds;         uint32_t n_boundeds;
  const rl_rtc_transformed *transformeds; uint32_t n_transformeds;
  const rl_rtc_material *materials;       uint32_t n_materials;
  const rl_oref *objects;                 uint32_t n_objects; /* World.objects, in order */
  const rl_rtc_light *lights;             uint32_t n_lights;
  uint32_t max_reflection_depth;
  uint32_t reserved;
  double void_color[3];
  const rl_rtc_shape *shapes;             uint32_t n_shapes;
  const rl_rtc_csg *csgs;                 uint32_t n_csgs;
  const rl_rtc_pattern *patterns;         uint32_t n_patterns;
} rl_rtc_scene_desc;

typedef struct rl_rtc_camera { /* scene/camera.rs:11-19: the derived fields + transform.inverse() */
  uint32_t hsize, vsize;
  double inverse[16]; /* row-major */
  double pixel_size, half_width, half_height;
} rl_rtc_camera;

rl_scene *rl_rtc_scene_create(const rl_rtc_scene_desc *desc);

/* Replaces Camera::render(&world, &RenderOpts{anti_aliasing_samples}) (scene/camera.rs:93).
 * out_rgb: W*H*3 doubles, row-major (Canvas.data order, draw/canvas.rs:44), pixel MEANS. */
int rl_rtc_render(const rl_scene *, const rl_rtc_camera *, uint32_t aa_samples, double *out_rgb,
                  rl_stats *opt_stats);
int rl_rtc_render_rows(const rl_scene *, const rl_rtc_camera *, uint32_t aa_samples,
                       uint32_t row_first, uint32_t row_step, double *out_rgb, rl_stats *opt_stats);
int rl_rtc_render_device(const rl_scene *, const rl_rtc_camera *, uint32_t aa_samples,
                         uint32_t row_first, uint32_t row_step, void *d_out_rgb, void *hip_stream,
                         rl_stats *opt_stats);

/* Camera::render(&world, opts) over every GPU of rl_init_multi: rows interleaved, one RCCL exchange to GPU 0 (see rl_rtiow_render_multi). */
int rl_rtc_render_multi(const rl_scene *, const rl_rtc_camera *, uint32_t aa_samples, double *out_rgb, rl_stats *opt_stats);
int rl_rtc_render_multi_device(const rl_scene *, const rl_rtc_camera *, uint32_t aa_samples, void *d_out_rgb, rl_stats *opt_stats);

/* Output stage on the device (draw/canvas.rs:53-56): round(c * 255) (half away from zero) clamped to 0..255. */
int rl_rtc_encode_rgb8_device(const void *d_rgb, uint64_t n_pixels, void *d_rgb8, void *hip_stream);
int rl_rtc_render_rgb8(const rl_scene *, const rl_rtc_camera *, uint32_t aa_samples, uint8_t *out_rgb8, rl_stats *opt_stats);

#ifdef __cplusplus
}
#endif
#endif /* RL_RENDER_H */
