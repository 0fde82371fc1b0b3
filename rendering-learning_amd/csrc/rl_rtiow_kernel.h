// RTIOW hot path on gfx950: Camera::_render's per-pixel loop (camera.rs:145-199), get_ray (:203-230),
// ray_color (:232-260), Bvh/slice/Sphere hit (bvh.rs:79-95, hittable/mod.rs:88-105, sphere.rs:32-75,
// aabb.rs:123-152), the materials (material.rs) and textures (texture.rs), ChaCha8 stream RNG
// (rand_chacha 0.3.1 as used at camera.rs:161-170).
//
// Parallel unit = the pixel: samples of one pixel are sequentially dependent (set_stream keeps the
// ChaCha word position), pixels are independent.  One lane owns one pixel at a time and claims the
// next from a global counter when done (persistent lanes).  The scene program (rl_program.h) and the
// sphere table are staged in LDS once per workgroup; traversal is stackless.
#pragma once
#ifndef RL_KERNEL_ALIGN
// Every big kernel starts on a 64 KB boundary of the code object.  Measured (round 3): the stealing instantiation of rtiow_wave_kernel (77 KB of
// code: main loop + cooperative body, more than the 64 KB instruction cache two CUs share) ran the 1/8 shard in 174 or in 189 ms depending on
// nothing but where the linker happened to put it (0x...ad00 against 0x...d600 after unrelated kernels grew); aligned, the placement is fixed.
#define RL_KERNEL_ALIGN __attribute__((aligned(65536)))
#endif
#include "rl_device.h"

namespace rl {

struct RtiowParams {
  const DevOp *ops;
  const DevSphere *spheres;
  const uint32_t *sphere_material;
  const DevPlanar *planars;
  const rl_translate *translates;
  const rl_transform *transforms;
  const DevMaterial *materials;
  const DevTexture *textures;
  const DevImage *images;
  const float *image_pool;
  const rl_perlin *perlins;
  const rl_medium *media;
  uint32_t n_ops, n_spheres;
  const DevMaterial *sphere_flat;  // wave kernel: per-sphere flattened material (see flatten_sphere_materials)
  const DevOp *lops;  // wave kernel: ops with {code, skip} replaced by linked successor words (state << 29 | op index), see link_ops
  uint32_t entry0;    // linked word of op 0: where (and in which state) a new ray starts
  const CompactOp *cops;    // wave kernel LDS_SCENE = 3: guarded compact ops (n_ops originals + one guard per sphere)
  const uint32_t *movbits;  // ... one bit per sphere: Center::Moving
  uint32_t n_cops, centry0;
  const FastNode *fast_nodes;  // wave kernel LDS_SCENE = 4: the fast traversal structure (n_fast_inner nodes; entry ids, rl_program.h)
  uint32_t n_fast_inner, fast_root;
  const FastNodeQ *fg_nodes;  // rl_rtiow_fastgen.h: fast traversal structure of a general scene (rl_fast_bvh.cpp build_fast_general)
  const DevSphere *fg_spheres;  // [item] sphere record of sphere items
  const uint32_t *fg_material;   // [item] material index
  const FastItem *fg_items;
  uint32_t fg_root;
#ifdef RL_EXPERIMENTAL
  const FastNodeO *fg_onodes;  // eight-wide quantised form (FastNodeO) and its root entry (A/B)
  uint32_t fg_oroot;
#endif
  float fg_center[3], fg_rsafe2;  // r_safe squared
  float fg_radius, fg_pad_k;      // far-origin rays: box growth = fg_pad_k * (distance + fg_radius)^2 (rl_rtiow_fastgen.h start_ray)
  rl_rtiow_camera cam;
  uint32_t key[8];
  uint64_t first_sample;
  uint32_t row_first, row_step, nrows;
  uint32_t tiles_x, n_slots;
  uint32_t *work_counter;
  double *out;
  uint32_t sample_begin, sample_end, resume;  // wave kernel: this launch renders samples [begin, end); resume = continue saved pixels
  uint32_t *pos_state;                         // per-pixel ChaCha word position (saved at pixel end when non-null)
  const uint32_t *tile_order;                  // slot>>6 -> tile (LPT order) or null
  uint32_t *tile_cost;                         // per-tile ray count accumulated at pixel end, or null
  double k8u;                 // 8 * 2^-53, passed as a kernel argument so it lives in SGPRs (one v_fma instead of v_mov + v_fmac with a literal)
  uint32_t tune[4];           // wave kernel: [0] max TRAV steps per scheduling round, [1] leave-TRAV population floor in 1/16ths
  uint32_t thin_tiles;        // fast wave kernel, resume launch: the first thin_tiles tiles of tile_order are handed out 64 >> thin_shift pixels per wave
  uint32_t thin_shift;        // 2 (16 pixels per wave) .. 6 (one pixel per wave)
  uint32_t prio_tiles;
  // work stealing on small shards (rl_rtiow_wave.h STEAL instantiation): a wave whose lanes have all run out of pixels takes over pixels
  // other lanes are still rendering, at a sample boundary, and continues them with the cooperative one-wave-per-pixel body.
  // steal_state[pix]: 0 queued / running, 1 take-over requested, 2 released (P.out, pos_state and steal_n hold the state), 3 finished
  uint32_t *steal_state, *steal_n, *steal_counter;
  const float *coop_leaf_boxes;        // fast wave kernel, resume launch (A/B): a wave holding a pixel of the first prio_tiles tiles runs at s_setprio 3
  uint32_t *pix_rays;         // debug (tools/): per-pixel ray counts, accumulated at pixel end by the counting wave kernel, or null
  unsigned long long *stats;  // [0]=rays [1]=node_tests [2]=sphere_tests [3]=planar [4]=instance [5]=rng_words [6]=flagged
  // (appended last: the by-value kernels' kernarg offsets of everything above stay where they were)
  const uint32_t *fg_seg_roots;  // rl_rtiow_fastgen.h MEDIA: root entry of every program segment (FastGeneral::seg_roots)
  const FastMedium *fg_media;    // ... and the media between them
  uint32_t fg_n_seg;
  uint32_t fg_top;  // the first fg_top nodes of fg_nodes (the tree's top, breadth first) are copied into LDS by every workgroup (0: none)
};

// ---------------------------------------------------------------- ChaCha8
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return __builtin_rotateleft32(x, r); }

#define RL_QR(a, b, c, d)   \
  a += b, d ^= a, d = rotl32(d, 16); \
  c += d, b ^= c, b = rotl32(b, 12); \
  a += b, d ^= a, d = rotl32(d, 8);  \
  c += d, b ^= c, b = rotl32(b, 7);

// Generates block (key, ctr, stream) and stores its 16 words as 8 u64 into the lane's LDS column.
// ROLLED: the four double rounds as a loop (a quarter of the code).  A block is ~420 instructions (2 KB) unrolled, and inlined at every
// gen_f64 call site the copies made up half of the wave kernels' code — 37 of the 77 KB of the stealing instantiation, more than the 64 KB
// instruction cache two CUs share.  The wave-scheduled kernels (rl_rtiow_wave.h Ring) use the rolled form everywhere: main kernel 7292 ->
// 4728 instructions and no spills left, stealing instantiation 13943 -> 8986 and 38 -> 6 spilled VGPRs; 6709 -> 6770 Mrays/s, 1/8 shard
// 174 -> 163 ms.  The cooperative one-wave-per-pixel body keeps the unrolled form: there a block is on the pixel's critical path.
// (A real function call for the rare mid-SHADE refill was measured too: its register convention spills 36 more VGPRs in the main loop, 4940 Mrays/s.)
template <int NT, bool ROLLED = false>
__device__ __forceinline__ void chacha8_block_to_lds(const uint32_t *key, uint32_t ctr_lo, uint64_t stream, unsigned long long *s_rng, int tid) {
  const uint32_t c0 = 0x61707865u, c1 = 0x3320646eu, c2 = 0x79622d32u, c3 = 0x6b206574u;
  uint32_t s12 = ctr_lo, s13 = 0u, s14 = (uint32_t)stream, s15 = (uint32_t)(stream >> 32);
  uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = c3, x4 = key[0], x5 = key[1], x6 = key[2], x7 = key[3], x8 = key[4], x9 = key[5], x10 = key[6],
           x11 = key[7], x12 = s12, x13 = s13, x14 = s14, x15 = s15;
#pragma unroll(ROLLED ? 1 : 4)
  for (int r = 0; r < 4; r++) {
    RL_QR(x0, x4, x8, x12) RL_QR(x1, x5, x9, x13) RL_QR(x2, x6, x10, x14) RL_QR(x3, x7, x11, x15)
    RL_QR(x0, x5, x10, x15) RL_QR(x1, x6, x11, x12) RL_QR(x2, x7, x8, x13) RL_QR(x3, x4, x9, x14)
  }
  x0 += c0, x1 += c1, x2 += c2, x3 += c3;
  x4 += key[0], x5 += key[1], x6 += key[2], x7 += key[3], x8 += key[4], x9 += key[5], x10 += key[6], x11 += key[7];
  x12 += s12, x13 += s13, x14 += s14, x15 += s15;
  s_rng[0 * NT + tid] = (unsigned long long)x0 | ((unsigned long long)x1 << 32);
  s_rng[1 * NT + tid] = (unsigned long long)x2 | ((unsigned long long)x3 << 32);
  s_rng[2 * NT + tid] = (unsigned long long)x4 | ((unsigned long long)x5 << 32);
  s_rng[3 * NT + tid] = (unsigned long long)x6 | ((unsigned long long)x7 << 32);
  s_rng[4 * NT + tid] = (unsigned long long)x8 | ((unsigned long long)x9 << 32);
  s_rng[5 * NT + tid] = (unsigned long long)x10 | ((unsigned long long)x11 << 32);
  s_rng[6 * NT + tid] = (unsigned long long)x12 | ((unsigned long long)x13 << 32);
  s_rng[7 * NT + tid] = (unsigned long long)x14 | ((unsigned long long)x15 << 32);
}

// per-lane RNG state: stream + word position; the current block lives in LDS
struct Rng {
  uint64_t stream;
  uint32_t pos;      // u32 word position since the pixel started (set_stream keeps it, camera.rs:170)
  uint32_t buf_ctr;  // block counter of the block held in LDS, 0xFFFFFFFF = none
};

template <int NT>
struct RngCtx {
  const uint32_t *key;
  unsigned long long *s_rng;
  int tid;
  __device__ __forceinline__ uint64_t next_u64(Rng &r) const {
    uint32_t ctr = r.pos >> 4;
    if (ctr != r.buf_ctr) {
      chacha8_block_to_lds<NT>(key, ctr, r.stream, s_rng, tid);
      r.buf_ctr = ctr;
    }
    uint64_t v = s_rng[((r.pos & 15u) >> 1) * NT + tid];
    r.pos += 2;
    return v;
  }
  // rand 0.8.5 Standard for f64: 53 random bits * 2^-53
  __device__ __forceinline__ double gen_f64(Rng &r) const { return (double)(next_u64(r) >> 11) * 0x1.0p-53; }
  // rand 0.8.5 Uniform::new(-1.0, 1.0): (value1_2 - 1.0) * scale + low, scale = 2
  __device__ __forceinline__ double uniform_m1_1(Rng &r) const {
    double v = __longlong_as_double((long long)((next_u64(r) >> 12) | 0x3FF0000000000000ull));
    return (v - 1.0) * 2.0 + (-1.0);
  }
  // rand_distr 0.4.3 UnitSphere (Marsaglia 1972): reject s >= 1
  __device__ __forceinline__ D3 unit_sphere(Rng &r) const {
    for (;;) {
      double x1 = uniform_m1_1(r), x2 = uniform_m1_1(r);
      double s = x1 * x1 + x2 * x2;
      if (s >= 1.0) continue;
      double f = 2.0 * sqrt(1.0 - s);
      return D3{x1 * f, x2 * f, 1.0 - 2.0 * s};
    }
  }
  // rand_distr 0.4.3 UnitDisc: accept s <= 1
  __device__ __forceinline__ void unit_disc(Rng &r, double &a, double &b) const {
    for (;;) {
      a = uniform_m1_1(r);
      b = uniform_m1_1(r);
      if (a * a + b * b <= 1.0) return;
    }
  }
};

// ---------------------------------------------------------------- AABB (aabb.rs:123-152)
__device__ __forceinline__ void intersect_axis(double mn, double mx, double origin, double speed, double &lo, double &hi) {
  double t0 = (mn - origin) / speed;
  double t1 = (mx - origin) / speed;
  bool lt = t0 < t1;
  lo = lt ? t0 : t1;
  hi = lt ? t1 : t0;
}
__device__ __forceinline__ bool aabb_hit(const double *b, D3 o, D3 d, double tmin_, double tmax_) {
  double xl, xh, yl, yh, zl, zh;
  intersect_axis(b[0], b[1], o.x, d.x, xl, xh);
  intersect_axis(b[2], b[3], o.y, d.y, yl, yh);
  intersect_axis(b[4], b[5], o.z, d.z, zl, zh);
  // f64::max / f64::min ignore NaN == fmax / fmin
  double tmin = fmax(fmax(fmax(xl, yl), zl), tmin_);
  double tmax = fmin(fmin(fmin(xh, yh), zh), tmax_);
  return tmin < tmax;
}

struct Hit {  // what the traversal keeps: enough to rebuild the HitRecord afterwards
  double t;
  uint32_t prim;  // sphere payload (index | SPH_MOVING) or NONE
};

// Sphere::hit (sphere.rs:32-75). Updates `h` when the sphere is hit inside [tmin, h.t]; returns true
// if the reference's NormalizedVec3::from_normalized assert (vec3.rs:219-222) would have fired.
__device__ __forceinline__ bool sphere_hit(const DevSphere &s, uint32_t payload, D3 o, D3 d, double time, double tmin, Hit &h) {
  D3 c0 = ld3(s.c0);
  D3 center = (payload & SPH_MOVING) ? c0 + ld3(s.dc) * time : c0;  // sphere.rs:24-29
  D3 oc = o - center;
  double a = len2(d);
  double half_b = dot(oc, d);
  double c = len2(oc) - s.r2;
  double disc = half_b * half_b - a * c;
  if (disc < 0.0) return false;
  double sq = sqrt(disc);
  double r_l = (-half_b - sq) / a;
  double r_u = (-half_b + sq) / a;
  double t;
  if (tmin <= r_l && r_l <= h.t) t = r_l;
  else if (tmin <= r_u && r_u <= h.t) t = r_u;
  else return false;
  h.t = t;
  h.prim = payload;
  D3 p = o + d * t;
  D3 outward = (p - center) * s.inv_r;
  double l2 = len2(outward);
  return !(l2 == 1.0 || fabs(l2 - 1.0) <= 1e-5);
}

// Rust `f64 as i32`: saturating, NaN -> 0
__device__ __forceinline__ int f64_as_i32(double x) {
  return x != x ? 0 : (x >= 2147483647.0 ? 2147483647 : (x <= -2147483648.0 ? (int)(-2147483647 - 1) : (int)x));
}

// perlin.rs:39-66,101-125: Perlin::noise = trilinear Hermite blend of dot(randvec[hash], offset) over the cell's 8 corners
__device__ __forceinline__ double perlin_noise(const rl_perlin &pn, D3 p) {
  double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
  double u = p.x - fx, v = p.y - fy, w = p.z - fz;
  uint32_t i = (uint32_t)f64_as_i32(fx), j = (uint32_t)f64_as_i32(fy), k = (uint32_t)f64_as_i32(fz);
  double uu = u * u * (3.0 - 2.0 * u);
  double vv = v * v * (3.0 - 2.0 * v);
  double ww = w * w * (3.0 - 2.0 * w);
  double accum = 0.0;
  // deliberately NOT unrolled: eight live corner vectors cost the all-primitives kernels ~70 VGPRs (and with them their
  // occupancy or scratch spills); one corner at a time accumulates in the reference's order just the same
#pragma unroll 1
  for (uint32_t di = 0; di < 2; di++)
#pragma unroll 1
    for (uint32_t dj = 0; dj < 2; dj++)
#pragma unroll 1
      for (uint32_t dk = 0; dk < 2; dk++) {
        uint32_t h = pn.perm_x[(i + di) & 255u] ^ pn.perm_y[(j + dj) & 255u] ^ pn.perm_z[(k + dk) & 255u];
        D3 c = ld3(pn.randvec[h & 255u]);
        double i_f = (double)di, j_f = (double)dj, k_f = (double)dk;
        D3 weight_v = d3(u - i_f, v - j_f, w - k_f);
        accum += (i_f * uu + (1.0 - i_f) * (1.0 - uu)) * (j_f * vv + (1.0 - j_f) * (1.0 - vv)) * (k_f * ww + (1.0 - k_f) * (1.0 - ww)) * dot(c, weight_v);
      }
  return accum;
}

// perlin.rs:68-80
__device__ __forceinline__ double perlin_turb(const rl_perlin &pn, D3 p, uint32_t depth) {
  double accum = 0.0, weight = 1.0;
  D3 temp_p = p;
#pragma unroll 1
  for (uint32_t it = 0; it < depth; it++) {
    accum += weight * perlin_noise(pn, temp_p);
    weight *= 0.5;
    temp_p = temp_p * 2.0;
  }
  return fabs(accum);
}

// sphere.rs:91-99 get_sphere_uv (acos / atan2 are the device libm's: colour-only)
__device__ __forceinline__ void sphere_uv(D3 p, double &u, double &v) {
  const double PI = 3.14159265358979323846;
  double theta = acos(-p.y);
  double phi = atan2(-p.z, p.x) + PI;
  u = phi / (2.0 * PI);
  v = theta / PI;
}

// texture.rs: value(u, v, p).  MODE 0 = Solid / Checker only (the sphere-only kernels: scenes with Image / Noise textures
// are routed to the all-primitives kernels by rl_rtiow_render_device), 1 = + Image, 2 = + Noise.  The branches are
// compiled out, not just skipped: sin() and the Perlin code cost the kernels that carry them ~70 VGPRs.
template <int MODE = 0>
__device__ __forceinline__ D3 texture_value(const RtiowParams &P, uint32_t tex, double u, double v, D3 p) {
  for (int guard = 0; guard < 64; guard++) {
    const DevTexture &t = P.textures[tex];
    if (t.kind == RL_TEX_SOLID) return ld3(t.color);
    if (t.kind == RL_TEX_CHECKER) {  // texture.rs:41-55
      // floor() as i64 (saturating; |p*inv_scale| < 2^63 for every sane scene), Rust % keeps the sign
      long long xi = (long long)floor(p.x * t.inv_scale);
      long long yi = (long long)floor(p.y * t.inv_scale);
      long long zi = (long long)floor(p.z * t.inv_scale);
      long long sum = (long long)((unsigned long long)xi + (unsigned long long)yi + (unsigned long long)zi);
      tex = ((sum % 2) == 0) ? t.even : t.odd;
      continue;
    }
    if (MODE == 0) return D3{0.0, 0.0, 0.0};
    if (MODE == 2 && t.kind == RL_TEX_NOISE) {  // texture.rs:84-94: Color(0.5,0.5,0.5) * (1 + sin(scale * p.z + 10 * turb(p, 7)))
      double sv = 1.0 + sin(t.inv_scale * p.z + 10.0 * perlin_turb(P.perlins[t.image], p, 7u));
      return D3{0.5 * sv, 0.5 * sv, 0.5 * sv};
    }
    // RL_TEX_IMAGE (texture.rs:62-82): nearest texel, f32 linear RGB
    const DevImage &im = P.images[t.image];
    double uu = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
    double vc = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    double vv = 1.0 - vc;
    double fi = uu * (double)(im.width - 1), fj = vv * (double)(im.height - 1);
    uint32_t i = !(fi > 0.0) ? 0u : (fi >= 4294967295.0 ? 4294967295u : (uint32_t)fi);
    uint32_t j = !(fj > 0.0) ? 0u : (fj >= 4294967295.0 ? 4294967295u : (uint32_t)fj);
    const float *px = P.image_pool + im.offset + ((size_t)j * im.width + i) * 3;
    return D3{(double)px[0], (double)px[1], (double)px[2]};
  }
  return D3{0.0, 0.0, 0.0};
}

struct Counters {
  unsigned long long rays, nodes, spheres, planars, instances, flagged;
};

// ================================================================= the kernel
// LDS_SCENE: ops + spheres staged in LDS (they fit); otherwise read through L1/L2 from HBM.
template <int NT, bool LDS_SCENE, bool STATS>
__global__ void __launch_bounds__(NT) rtiow_spheres_kernel(RtiowParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  unsigned long long *s_rng = (unsigned long long *)smem;  // [8][NT]
  const DevOp *ops = P.ops;
  const DevSphere *spheres = P.spheres;
  if (LDS_SCENE) {
    DevOp *s_ops = (DevOp *)(smem + (size_t)8 * NT * sizeof(unsigned long long));
    DevSphere *s_sph = (DevSphere *)(s_ops + P.n_ops);
    // 16-byte cooperative copy
    const uint4 *g = (const uint4 *)P.ops;
    uint4 *l = (uint4 *)s_ops;
    for (uint32_t i = tid; i < P.n_ops * 4u; i += NT) l[i] = g[i];
    g = (const uint4 *)P.spheres;
    l = (uint4 *)s_sph;
    for (uint32_t i = tid; i < P.n_spheres * 4u; i += NT) l[i] = g[i];
    __syncthreads();
    ops = s_ops;
    spheres = s_sph;
  }
  RngCtx<NT> rc{P.key, s_rng, tid};
  const rl_rtiow_camera &cam = P.cam;
  const uint32_t W = cam.image_width;
  const uint64_t WH = (uint64_t)cam.image_width * (uint64_t)cam.image_height;
  const D3 p00 = ld3(cam.pixel_00), du = ld3(cam.pixel_du), dv = ld3(cam.pixel_dv);
  const D3 lookfrom = ld3(cam.lookfrom), ddu = ld3(cam.defocus_disk_u), ddv = ld3(cam.defocus_disk_v);
  const D3 background = ld3(cam.background);
  const double INF = __longlong_as_double(0x7FF0000000000000ll);
  Counters cnt{0, 0, 0, 0, 0, 0};
  unsigned long long words = 0;

  for (;;) {
    uint32_t slot = wave_claim(P.work_counter);
    if (slot >= P.n_slots) break;
    // 8x8 tiles over (x, virtual row)
    uint32_t tile = slot >> 6, in = slot & 63u;
    uint32_t x = (tile % P.tiles_x) * 8u + (in & 7u);
    uint32_t r = (tile / P.tiles_x) * 8u + (in >> 3);
    if (x >= W || r >= P.nrows) continue;
    uint32_t y = P.row_first + r * P.row_step;

    Rng rng{0ull, 0u, 0xFFFFFFFFu};
    D3 sum = d3(0.0, 0.0, 0.0);
    for (uint32_t n = 0; n < cam.samples_per_pixel; n++) {
      uint64_t sample_index = (uint64_t)n + P.first_sample;
      rng.stream = sample_index * WH + (uint64_t)x * (uint64_t)W + (uint64_t)y;  // camera.rs:167-169 (x*W, sic)
      rng.buf_ctr = 0xFFFFFFFFu;                                                   // new stream: block must be regenerated
      // get_ray camera.rs:203-216
      D3 pixel_center = (p00 + du * (double)x) + dv * (double)y;
      double px = -0.5 + rc.gen_f64(rng);
      double py = -0.5 + rc.gen_f64(rng);
      D3 pixel_sample = pixel_center + (du * px + dv * py);
      D3 o;
      if (cam.defocus_angle <= 0.0) o = lookfrom;
      else {
        double a, b;
        rc.unit_disc(rng, a, b);
        o = (lookfrom + ddu * a) + ddv * b;
      }
      D3 d = pixel_sample - o;
      double time = rc.gen_f64(rng);

      // ray_color camera.rs:232-260 in throughput form (colour-only reassociation, <= a few ulps)
      D3 thr = d3(1.0, 1.0, 1.0);
      D3 color = d3(0.0, 0.0, 0.0);
      for (uint32_t depth = cam.max_depth; depth > 0; depth--) {
        cnt.rays++;
        // ---- world.hit(r, [1e-10, inf]) : threaded traversal in the reference's DFS order
        Hit h{INF, NONE};
        uint32_t pc = 0;
        for (;;) {
          const DevOp &op = ops[pc];
          uint32_t code = op.code & 0xFFu;
          if (code == OP_END) break;
          if (code == OP_SPHERE) {
            if (STATS) cnt.spheres++;
            uint32_t a = op.a;
            if (sphere_hit(spheres[a & SPH_INDEX], a, o, d, time, 1e-10, h)) cnt.flagged++;
            pc++;
            continue;
          }
          // OP_BOX / OP_BOX_SPH
          if (STATS) cnt.nodes++;
          uint32_t skip = op.skip;
          if (!aabb_hit(op.box, o, d, 1e-10, h.t)) {
            pc = skip;
            continue;
          }
          if (code == OP_BOX) {
            pc++;
            continue;
          }
          uint32_t a = op.a, b = op.b;
          if (STATS) cnt.spheres++;
          if (sphere_hit(spheres[a & SPH_INDEX], a, o, d, time, 1e-10, h)) cnt.flagged++;
          if (b != NONE) {
            if (STATS) cnt.spheres++;
            if (sphere_hit(spheres[b & SPH_INDEX], b, o, d, time, 1e-10, h)) cnt.flagged++;
          }
          pc = skip;
        }
        if (h.prim == NONE) {  // miss -> background
          color = color + thr * background;
          break;
        }
        // rebuild the HitRecord of the winning sphere (same arithmetic as at test time)
        uint32_t si = h.prim & SPH_INDEX;
        const DevSphere &s = spheres[si];
        D3 c0 = ld3(s.c0);
        D3 center = (h.prim & SPH_MOVING) ? c0 + ld3(s.dc) * time : c0;
        D3 p = o + d * h.t;
        D3 outward = (p - center) * s.inv_r;
        bool front = dot(d, outward) <= 0.0;  // hittable/mod.rs:32-38
        D3 normal = front ? outward : -outward;
        const DevMaterial &m = P.materials[P.sphere_material[si]];
        uint32_t kind = m.kind;
        D3 nd;
        if (kind == RL_MAT_LAMBERTIAN) {  // material.rs:74-92
          D3 dir = normal + rc.unit_sphere(rng);
          bool near_zero = approx_eq_eps(dir.x, 0.0, 1e-8) && approx_eq_eps(dir.y, 0.0, 1e-8) && approx_eq_eps(dir.z, 0.0, 1e-8);
          nd = near_zero ? normal : dir;
          thr = thr * texture_value(P, m.texture, 0.0, 0.0, p);
        } else if (kind == RL_MAT_METAL) {  // material.rs:105-122
          D3 reflected = d - normal * (2.0 * dot(d, normal));
          nd = normalize(reflected) + rc.unit_sphere(rng) * m.fuzz;
          if (!(dot(nd, normal) > 0.0)) break;  // absorbed: emitted (0) only
          thr = thr * ld3(m.albedo);
        } else if (kind == RL_MAT_DIELECTRIC) {  // material.rs:139-165
          double ri = front ? 1.0 / m.ior : m.ior;
          double m2 = len2(d);
          D3 ud;
          if (approx_eq_eps(m2, 0.0, 1e-16)) {  // "How did the incident ray have magnitude 0?"
            cnt.flagged++;
            ud = d;
          } else
            ud = normalize(d);
          double cos_theta = fmin(dot(-ud, normal), 1.0);
          double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
          bool reflect = ri * sin_theta > 1.0;
          if (!reflect) {  // Schlick, material.rs:173-176; short-circuit: the draw happens only here
            double q = (1.0 - ri) / (1.0 + ri);
            double r0 = q * q;
            double xx = 1.0 - cos_theta;
            double x2 = xx * xx;
            double refl = r0 + (1.0 - r0) * (xx * (x2 * x2));
            reflect = refl > rc.gen_f64(rng);
          }
          if (reflect) nd = ud - normal * (2.0 * dot(ud, normal));
          else {  // vec3.rs:224-230
            D3 perp = (ud + normal * cos_theta) * ri;
            D3 par = normal * (-sqrt(fabs(1.0 - len2(perp))));
            nd = perp + par;
          }
          // attenuation (1,1,1): thr unchanged (x * 1.0 == x)
        } else if (kind == RL_MAT_DIFFUSE_LIGHT) {  // material.rs:182-195: emitted, no scatter
          color = color + thr * texture_value(P, m.texture, 0.0, 0.0, p);
          break;
        } else {  // Flat
          break;
        }
        o = p;
        d = nd;
      }
      sum = sum + color;  // camera.rs:174: sequential fold in sample order
    }
    words += rng.pos;
    double *outp = P.out + ((size_t)r * W + x) * 3;
    outp[0] = sum.x, outp[1] = sum.y, outp[2] = sum.z;
  }

  // counters: one atomic per wave per counter
  unsigned long long v;
  v = wave_sum(cnt.rays);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[0], v);
  v = wave_sum(cnt.flagged);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[6], v);
  if (STATS) {
    v = wave_sum(cnt.nodes);
    if ((tid & 63) == 0) atomicAdd(&P.stats[1], v);
    v = wave_sum(cnt.spheres);
    if ((tid & 63) == 0) atomicAdd(&P.stats[2], v);
    v = wave_sum(words);
    if ((tid & 63) == 0) atomicAdd(&P.stats[5], v);
  }
}

}  // namespace rl
