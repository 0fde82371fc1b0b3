// RTIOW sphere/BVH kernel, wave-scheduled form ("v2").
//
// Same arithmetic and same per-pixel / per-sample / per-ray order as rl_rtiow_kernel.h (and as the
// reference), different *schedule*: each lane is a small state machine
//     GEN  (claim pixel / start next sample: new ChaCha stream, camera ray)
//     TRAV (ONE box op of the threaded scene program per step)
//     LEAF (the 1-2 Sphere::hit tests of a BVH leaf whose box passed)
//     FILL (top the lane's ChaCha ring up by one block)
//     SHADE(miss -> background; hit -> rebuild HitRecord, scatter, next ray)
// and the 64-wide wave repeatedly runs the state that has the most lanes waiting in it.  A lane
// whose path ends starts its pixel's next sample at once (path regeneration; legal because the
// per-pixel sample order is preserved), so no lane idles while another finishes a long path, and
// ChaCha block generation always runs for many lanes at once instead of lane-by-lane inside
// rejection loops.
//
// AABB test: the reference divides six times per box ((min-o)/d, (max-o)/d per axis, aabb.rs:143-152).
// Here each ray carries inv = 1.0/d (three IEEE divisions per ray) and oi = o*inv, and a box test uses
// t'' = fma(b, inv, -oi), one instruction per bound.  The DECISION tmin < tmax is taken from the
// approximations only when |tmax''-tmin''| exceeds a rigorous error margin (see RayAux); otherwise — and for non-finite boxes or rays with a zero / tiny / huge
// direction component — the lane re-evaluates the reference's exact divisions.  Decisions are
// therefore bit-identical to the reference (tests assert equal AABB/sphere/ray/RNG-word counters).
#pragma once
#include <type_traits>

#include "rl_rtiow_kernel.h"

namespace rl {

// ST_SHADE2 (fast traversal only): the specular half of SHADE — Metal and Dielectric hits — so that the many Lambertian / miss lanes
// do not walk through normalize(), Schlick and refract() code they never need
#ifndef RL_PK_FMA
#define RL_PK_FMA 0  // the slab test's plane pairs as v_pk_fma_f32: measured 6750 -> 6600 Mrays/s — the broadcast operand pairs cost moves, one VGPR spills
#endif
#ifndef RL_SPLIT_LEAF
#define RL_SPLIT_LEAF false
#endif
enum : uint32_t { ST_GEN = 0, ST_TRAV = 1, ST_SHADE = 2, ST_FILL = 3, ST_DONE = 4, ST_LEAF = 5, ST_SHADE2 = 6, ST_PARK = 7, ST_LEAF2 = 8 };
// The latency modes of DESIGN.md §6 (RL_THIN: thin tiles, RL_PRIO: wave priority) — measured, no gain — are compiled into the
// experimental library only (make exp); the product kernel carries none of their code.
#ifdef RL_EXPERIMENTAL
static constexpr bool LATENCY_MODES = true;
#else
static constexpr bool LATENCY_MODES = false;
#endif

// two-block ChaCha ring in LDS: 16 u64 slots per lane, slot-major ([slot][lane]) => conflict-free
// ROLL_HOT: also the block generations on the hot paths (stream reset, FILL) use the rolled block function (rl_rtiow_kernel.h
// chacha8_block_to_lds<NT, true>); the refill in the middle of a SHADE block (rare: long rejection streaks) always does.
template <int NT, bool ROLL_HOT = true>
struct Ring {
  const uint32_t *key;
  unsigned long long *s_rng;  // [16][NT]
  int tid;
  uint64_t stream;
  uint32_t pos;     // u32 word position within the pixel
  uint32_t blk_lo;  // lowest resident block counter
  uint32_t nres;    // resident blocks: blk_lo .. blk_lo+nres-1 (0..2); block c lives in half (c & 1)

  __device__ __forceinline__ void gen_block(uint32_t c) { chacha8_block_to_lds<NT, ROLL_HOT>(key, c, stream, s_rng + (size_t)(c & 1u) * 8 * NT, tid); }
  __device__ __forceinline__ void reset_stream(uint64_t s) {  // set_stream keeps pos; both resident blocks become stale
    stream = s;
    blk_lo = pos >> 4;
#pragma unroll 1
    for (uint32_t k = 0; k < 2u; k++) gen_block(blk_lo + k);  // (rolled: one copy of the block function's 2 KB, not two)
    nres = 2;
  }
  __device__ __forceinline__ bool low() const { return (pos >> 4) >= blk_lo + nres - 1u; }  // reading from the newest block
  __device__ __forceinline__ void top_up() {  // make block (newest+1) resident, dropping the oldest
    uint32_t c = blk_lo + nres;
    gen_block(c);
    if (nres == 2) blk_lo++;
    else nres++;
  }
  __device__ __forceinline__ uint64_t next_u64() {
    uint32_t c = pos >> 4;
    if (__builtin_expect(c - blk_lo >= nres, 0)) {  // rare inline path (long rejection streaks): always the rolled block function, a quarter of the code per call site
      const uint32_t nc = blk_lo + nres;
      chacha8_block_to_lds<NT, true>(key, nc, stream, s_rng + (size_t)(nc & 1u) * 8 * NT, tid);
      if (nres == 2) blk_lo++;
      else nres++;
    }
    uint64_t v = s_rng[((size_t)(c & 1u) * 8 + ((pos & 15u) >> 1)) * NT + tid];
    pos += 2;
    return v;
  }
  __device__ __forceinline__ double gen_f64() { return (double)(next_u64() >> 11) * 0x1.0p-53; }
  __device__ __forceinline__ double uniform_m1_1() {
    double v = __longlong_as_double((long long)((next_u64() >> 12) | 0x3FF0000000000000ull));
    return (v - 1.0) * 2.0 + (-1.0);
  }
  __device__ __forceinline__ D3 unit_sphere() {
    for (;;) {
      double x1 = uniform_m1_1(), x2 = uniform_m1_1();
      double s = x1 * x1 + x2 * x2;
      if (s >= 1.0) continue;
      double f = 2.0 * sqrt(1.0 - s);
      return D3{x1 * f, x2 * f, 1.0 - 2.0 * s};
    }
  }
  __device__ __forceinline__ void unit_disc(double &a, double &b) {
    for (;;) {
      a = uniform_m1_1();
      b = uniform_m1_1();
      if (a * a + b * b <= 1.0) return;
    }
  }
};

// per-ray reciprocal + eligibility for the filtered AABB test
__device__ __forceinline__ bool ray_fast_ok(D3 o, D3 d) {
  auto okd = [](double v) { double a = fabs(v); return a >= 1e-100 && a <= 1e100; };
  auto oko = [](double v) { return fabs(v) <= 1e100; };
  return okd(d.x) && okd(d.y) && okd(d.z) && oko(o.x) && oko(o.y) && oko(o.z);
}

// Per-ray constants of the filtered AABB test.
//   inv = 1/d (IEEE), oi = o*inv, slack = 4u * max|oi|.
// A box test evaluates t'' = fma(b, inv, -oi) ~ (b-o)/d with ONE instruction per bound.  Against the
// reference's q^ = RN(RN(b-o)/d):  |t'' - q^| <= 4.1u|t''| + 1.1u|oi|  (u = 2^-53; three roundings in
// t'', two in q^, plus the cancellation term of b*inv - o*inv).  min/max commute with the relative part
// and are 1-Lipschitz in the absolute part, so |tmin'' - tmin| <= 4.2u|tmin''| + 1.1u max|oi| and the
// same for tmax: the decision is certain whenever |tmax''-tmin''| > 8u(|tmin''|+|tmax''|) + slack.
struct RayAux {
  D3 inv, oi;
  double slack;
  bool fast_ok;
};
__device__ __forceinline__ RayAux ray_aux(D3 o, D3 d) {
  RayAux a;
  a.inv = d3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
  a.oi = d3(o.x * a.inv.x, o.y * a.inv.y, o.z * a.inv.z);
  a.slack = fmax(fmax(fabs(a.oi.x), fabs(a.oi.y)), fabs(a.oi.z)) * 4.4408920985006262e-16;  // 4u
  a.fast_ok = ray_fast_ok(o, d);
  return a;
}

// Filtered AABB::hit. tmin_ is the kernel constant 1e-10 (> 0), tmax_ the closest hit so far.
// `certain` = false: the caller must evaluate the reference's divisions (aabb_hit).
__device__ __forceinline__ bool aabb_fast(const double *b, const RayAux &ra, double tmax_, bool &certain, double k8u = 8.8817841970012523e-16) {
  const double tmin_ = 1e-10;
  double t0x = fma(b[0], ra.inv.x, -ra.oi.x), t1x = fma(b[1], ra.inv.x, -ra.oi.x);
  double t0y = fma(b[2], ra.inv.y, -ra.oi.y), t1y = fma(b[3], ra.inv.y, -ra.oi.y);
  double t0z = fma(b[4], ra.inv.z, -ra.oi.z), t1z = fma(b[5], ra.inv.z, -ra.oi.z);
  double tmin = fmax(fmax(fmax(fmin(t0x, t1x), fmin(t0y, t1y)), fmin(t0z, t1z)), tmin_);
  double tmax3 = fmin(fmin(fmax(t0x, t1x), fmax(t0y, t1y)), fmax(t0z, t1z)), tmax;
  // = fmin(tmax3, tmax_): v_min_f64 is IEEE minNum (quiets a signalling NaN itself); written as asm because the
  // compiler otherwise spends a v_max_f64 x,x canonicalisation on the loop-carried tmax_ in every step
  asm("v_min_f64 %0, %1, %2" : "=v"(tmax) : "v"(tmax3), "v"(tmax_));
  double diff = tmax - tmin;
  double thresh = fma(tmin + fabs(tmax), k8u, ra.slack);  // 8u(|tmin|+|tmax|) + slack
  certain = fabs(diff) > thresh;                                            // false also for NaN / inf arithmetic
  return diff > 0.0;
}

// Single-precision form of the filter, for LDS-resident ops whose box has been re-stored as six floats (round to nearest).
//   inv32 = RN32(1/d), oi32 = RN32(o * (1/d)), t'' = fmaf(b32, inv32, -oi32), everything in binary32, u = 2^-24.
// |b32*inv32 - b/d| <= 2.1u|b/d| <= 2.1u(|t| + |o/d|), |oi32 - o/d| <= 1.1u|o/d|, one more rounding in the fma, and the
// clamp closest32 = RN32(closest) moves tmax by <= u|tmax|: each end of [tmin, tmax] is off by <= 3.2u|t| + 3.3u max|o/d|.
// The decision tmin < tmax is therefore certain when |tmax''-tmin''| > 8u(|tmin''|+|tmax''|) + 16u max|oi32| (twice the
// bound, which also absorbs the roundings of this comparison itself); otherwise, and whenever anything is non-finite, the
// lane evaluates the reference's divisions in binary64.  About 1 test in 10^5 falls back.
struct RayAux32 {
  float invx, invy, invz, oix, oiy, oiz, slack;
  // ray_aux32_direct only: max_k |o_k / d_k|, the coordinate scale of this ray in units of t (slack = 24u * that; +inf stays +inf)
  __device__ __forceinline__ float oimax() const { return slack * 699050.6875f; }
};
__device__ __forceinline__ RayAux32 ray_aux32(const RayAux &a) {
  RayAux32 r;
  r.invx = (float)a.inv.x, r.invy = (float)a.inv.y, r.invz = (float)a.inv.z;
  r.oix = (float)a.oi.x, r.oiy = (float)a.oi.y, r.oiz = (float)a.oi.z;
  float m = fmaxf(fmaxf(fabsf(r.oix), fabsf(r.oiy)), fabsf(r.oiz));
  // the bound above needs every product and sum to stay a NORMAL binary32 number: 1/d within [1e-30, 1e30] and |o/d| <= 1e30
  // (a zero, denormal-scale or huge direction component, or a far-away origin, makes the lane use the exact path always)
  float imin = fminf(fminf(fabsf(r.invx), fabsf(r.invy)), fabsf(r.invz)), imax = fmaxf(fmaxf(fabsf(r.invx), fabsf(r.invy)), fabsf(r.invz));
  bool ok = a.fast_ok && imin >= 1e-30f && imax <= 1e30f && m <= 1e30f;  // NaN compares false
  r.slack = ok ? m * 9.5367431640625e-07f : __int_as_float(0x7F800000);  // 16u, or +inf: never certain
  return r;
}
__device__ __forceinline__ bool aabb_fast32(const float *b, const RayAux32 &ra, float closest32, bool &certain) {
  float t0x = fmaf(b[0], ra.invx, -ra.oix), t1x = fmaf(b[1], ra.invx, -ra.oix);
  float t0y = fmaf(b[2], ra.invy, -ra.oiy), t1y = fmaf(b[3], ra.invy, -ra.oiy);
  float t0z = fmaf(b[4], ra.invz, -ra.oiz), t1z = fmaf(b[5], ra.invz, -ra.oiz);
  float tmin = fmaxf(fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z)), 1e-10f);
  float tmax = fminf(fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z)), closest32);
  float diff = tmax - tmin;
  float thresh = fmaf(tmin + fabsf(tmax), 4.76837158203125e-07f, ra.slack);  // 8u(|tmin|+|tmax|) + slack
  certain = fabsf(diff) > thresh;                                            // false also for NaN / inf arithmetic
  return diff > 0.0f;
}

// Per-ray constants of the FAST traversal's reject-only test, computed in binary32 from the start (no binary64 division):
//   d32 = RN32(d), inv32 = v_rcp_f32(d32) (<= 1 ulp), o32 = RN32(o), oi32 = RN32(o32 * inv32), u = 2^-24.
// inv32 = (1/d)(1 + e), |e| <= 3u;  b32 * inv32 is off by <= 4u|b/d| <= 4u(|t| + |o/d|) (the node boxes are stored exactly, rounded
// OUTWARDS);  oi32 by <= 5u|o/d|;  one more rounding in the fma and the clamp closest32 = RN32(closest): each end of [tmin, tmax] is
// off by <= 6u|t| + 9u max|o/d|.  A box is therefore CERTAINLY missed when tmin'' - tmax'' > 12u(|tmin''| + |tmax''|) + 24u max|oi32|
// (the bound for both ends together, plus room for the roundings of this comparison itself).  Rays with a zero, denormal-scale or
// huge direction component or a far-away origin get slack = +inf: nothing is ever certain for them, and start_ray() sends them
// to the reference-order fold.
__device__ __forceinline__ RayAux32 ray_aux32_direct(D3 o, D3 d) {
  RayAux32 r;
  float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
  r.invx = __builtin_amdgcn_rcpf(dx), r.invy = __builtin_amdgcn_rcpf(dy), r.invz = __builtin_amdgcn_rcpf(dz);
  r.oix = (float)o.x * r.invx, r.oiy = (float)o.y * r.invy, r.oiz = (float)o.z * r.invz;
  float m = fmaxf(fmaxf(fabsf(r.oix), fabsf(r.oiy)), fabsf(r.oiz));
  float imin = fminf(fminf(fabsf(r.invx), fabsf(r.invy)), fabsf(r.invz)), imax = fmaxf(fmaxf(fabsf(r.invx), fabsf(r.invy)), fabsf(r.invz));
  float omax = fmaxf(fmaxf(fabsf((float)o.x), fabsf((float)o.y)), fabsf((float)o.z));
  // every product and sum must stay a NORMAL binary32 number: 1/d within [1e-30, 1e30], |o| and |o/d| <= 1e30 (NaN compares false)
  bool ok = imin >= 1e-30f && imax <= 1e30f && m <= 1e30f && omax <= 1e30f;
  r.slack = ok ? m * 1.430511474609375e-06f : __int_as_float(0x7F800000);  // 24u, or +inf: never certain
  return r;
}
// Two accepted roots closer than this are treated as a tie (the ray is re-traced in the reference's order).  The reference's own
// decisions that involve two different primitives — `tmin < closest` on its boxes, `t <= closest` in Sphere::hit — compare values
// whose rounding errors are a few u (2^-53) times the SCALES involved: the far root of the sphere (cancellation in -half_b -+ sqrt),
// and the coordinates of origin and boxes over the direction (max |o/d|).  4e-12 is ~4e4 u.
__device__ __forceinline__ double fast_tie_band(double scale_t, float oimax) { return 4e-12 * (scale_t + (double)oimax); }

// LDS_SCENE: 0 = scene read from HBM / L2, 1 = linked ops + spheres staged in LDS, 2 = linked ops in LDS, spheres from L2,
// 3 = compact (32-byte) guarded ops in LDS (P.cops: the n_ops originals + one guard op per sphere), spheres from L2,
// 4 = FAST traversal (counter-free renders only): P.fast_nodes in LDS — see below
typedef __attribute__((address_space(3))) CompactOp LdsCompactOp;
typedef float Float4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const Float4 LdsFloat4;
typedef __attribute__((address_space(3))) const uint32_t LdsU32;

// ---- FAST traversal (LDS_SCENE = 4).  The reference's fold over its BVH visits ~40 boxes and ~4.7 spheres per ray because its tree
// is a median split walked in stored order.  For the RESULT of the fold only the smallest accepted root matters (rl_fast_bvh.cpp), so
// the timed kernel walks its own surface-area-heuristic binary tree instead: one TRAV step fetches a 64-byte node with BOTH children's
// boxes, tests the two with the binary32 filter in reject-only form (a box is skipped only when it is CERTAINLY missed inside
// [1e-10, closest]), descends into the nearer child and pushes the farther one on a 16-entry stack held in five VGPRs (ten-bit entry
// ids, v_alignbit shifts) — ~12 steps and ~1.9 Sphere::hit per ray.  Whenever the visiting ORDER could influence the reference's answer
// the ray is flagged and re-traced by fast_slow_trace, the reference's own fold with exact divisions:
//   * two roots within a few 1e4 ulps of the scales involved (fast_tie_band; exact ties go to the LAST sphere in the reference's
//     order; `tmin < closest` pruning on the reference's boxes),
//   * a grazing hit (chord below 1e-6 relative: the reference's unpadded leaf box may or may not be passed), or a hit next to one
//     of the sphere's axis poles, where it touches its own box (fast_near_box_face),
//   * (a hit whose outward normal trips the from_normalized assert, vec3.rs:219, would make the panic-site count order dependent:
//     scenes in which that assert is reachable do not get a fast structure at all, rl_fast_bvh.cpp normals_safe),
//   * a ray outside the filter's range (zero / denormal-scale / huge direction component, far-away origin).
static const uint32_t FAST_SLOW = 0xFFFFFFFFu;  // pc of a lane in ST_LEAF that must re-trace its ray in the reference's order

// A sphere touches its own bounding box (and every ancestor box it is extremal in) at its six axis poles.  The ROUNDED root can sit a
// little past the point where the ray leaves that box — e.g. the self-intersection root ~1e-10 of a ray that starts on top of a huge
// ground sphere, whose cancellation error is as large as the root itself — and then the reference's exact slab test (aabb.rs:123-152)
// prunes a sphere that Sphere::hit would have accepted.  The hit is order-sensitive when its point is closer to a face of
// [centre - r, centre + r] than the root's error bound moves it along the ray.
// Bound on the error of Sphere::hit's rounded roots, in units of t: the discriminant carries ~16u a |oc|^2 of rounding (cancellation in
// |oc|^2 - r^2 and half_b^2 - a c), i.e. 8u |oc|^2 / sqrt(disc) in the root, plus a few u of the roots' magnitudes and of the coordinate
// scale max |o/d|; times 64.  (The tie band above is far wider; this one decides how close to a box face a hit may lie.)
__device__ __forceinline__ double fast_root_error(double len2_oc, double sq, double r_l, double r_u, float oimax) {
  float q = (float)len2_oc * __builtin_amdgcn_rcpf((float)sq);  // a tolerance: binary32 is plenty; sq = 0 -> +inf
  return 64.0 * (1.8e-15 * (double)q + 4.5e-16 * (fabs(r_l) + fabs(r_u) + (double)oimax));
}
__device__ __forceinline__ bool fast_near_box_face(D3 o, D3 d, double t, D3 center, double r, double band) {
  D3 q = (o + d * t) - center;
  return r - fabs(q.x) <= band * fabs(d.x) || r - fabs(q.y) <= band * fabs(d.y) || r - fabs(q.z) <= band * fabs(d.z);
}

// Grazing hits and hits next to an axis pole, for an ACCEPTED root t (rl_rtiow_fastgen.h: for the winning hit).  A binary32 pre-filter
// on what Sphere::hit has already computed lets all but ~3 hits in 10^4 skip the exact test: |q_k| < 0.9999 r for q = oc + t d (not
// within 1e-4 r of a face), sqrt(disc) > 1e-3 |d|_max r (not grazing), and the root's error bound below half of that 1e-4 r.
__device__ __forceinline__ bool fast_hit_is_order_sensitive(D3 oc, D3 d, double t, double r, double half_b, double sq, double r_l, double r_u, float oimax) {
  float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z, t32 = (float)t, r32 = (float)r, sq32 = (float)sq;
  float ox = (float)oc.x, oy = (float)oc.y, oz = (float)oc.z;
  float qm = fmaxf(fmaxf(fabsf(fmaf(dx, t32, ox)), fabsf(fmaf(dy, t32, oy))), fabsf(fmaf(dz, t32, oz)));
  float dmax = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
  float l2 = fmaf(ox, ox, fmaf(oy, oy, oz * oz));
  // eps |d_k| <= 64 * 1.8e-15 |oc|^2 / sqrt(disc) * max |d_k| (+ lower-order terms) must stay below 5e-5 r
  bool quiet = 2.4e-13f * l2 * dmax < 5e-5f * r32 * sq32 && qm < 0.9999f * r32 && sq32 > 1e-3f * dmax * r32;
  if (__builtin_expect(quiet, 1)) return false;
  D3 q = oc + d * t;
  double eps = fast_root_error(len2(oc), sq, r_l, r_u, oimax);
  bool grazing = sq <= 1e-6 * fabs(half_b);
  bool pole = r - fabs(q.x) <= eps * fabs(d.x) || r - fabs(q.y) <= eps * fabs(d.y) || r - fabs(q.z) <= eps * fabs(d.z);
  return grazing || pole || !(eps * (double)dmax < 5e-5 * r);
}

// the first lines of Sphere::hit (sphere.rs:32-47): true when the discriminant is negative (same arithmetic as fast_sphere_hit below)
__device__ __forceinline__ bool fast_sphere_misses(const DevSphere &s, uint32_t payload, D3 o, D3 d, double time) {
  D3 c0 = ld3(s.c0);
  D3 center = (payload & SPH_MOVING) ? c0 + ld3(s.dc) * time : c0;
  D3 oc = o - center;
  double a = len2(d);
  double half_b = dot(oc, d);
  double c = len2(oc) - s.r2;
  return half_b * half_b - a * c < 0.0;
}

// Sphere::hit (sphere.rs:32-75) with the acceptance window widened by the tie band; same arithmetic, same root values.
__device__ __forceinline__ void fast_sphere_hit(const DevSphere &s, uint32_t payload, D3 o, D3 d, double time, float oimax, double &closest, uint32_t &hit_prim,
                                                bool &amb) {
  D3 c0 = ld3(s.c0);
  D3 center = (payload & SPH_MOVING) ? c0 + ld3(s.dc) * time : c0;
  D3 oc = o - center;
  double a = len2(d);
  double half_b = dot(oc, d);
  double c = len2(oc) - s.r2;
  double disc = half_b * half_b - a * c;
  if (disc < 0.0) return;
  double sq = sqrt(disc);
  double r_l = (-half_b - sq) / a;
  double r_u = (-half_b + sq) / a;
  const double band = fast_tie_band(fabs(r_l) + fabs(r_u), oimax);
  const double hi = closest + band;  // +inf stays +inf
  double t;
  if (1e-10 <= r_l && r_l <= hi) t = r_l;
  else if (1e-10 <= r_u && r_u <= hi) t = r_u;
  else return;
  // no from_normalized check here: build_fast_bvh only accepts scenes whose frame makes that assert unreachable (GuardFrame::normals_safe)
  amb = amb || (hit_prim != NONE && fabs(t - closest) <= band) || fast_hit_is_order_sensitive(oc, d, t, s.r2 * s.inv_r, half_b, sq, r_l, r_u, oimax);
  if (t <= closest) closest = t, hit_prim = payload;
}

// The reference's fold, verbatim: threaded program in stored order, exact divisions, Sphere::hit with ray_t.max = closest so far
// (bvh.rs:79-95, hittable/mod.rs:88-111).  Returns the number of from_normalized asserts tripped on the way.
__device__ __forceinline__ uint32_t fast_slow_trace(const DevOp *ops, const DevSphere *spheres, D3 o, D3 d, double time, double &closest, uint32_t &hit_prim) {
  Hit h{__longlong_as_double(0x7FF0000000000000ll), NONE};
  uint32_t flags = 0, pc = 0;
#pragma unroll 1
  for (;;) {
    const DevOp &op = ops[pc];
    uint32_t code = op.code & 0xFFu;
    if (code == OP_END) break;
    uint32_t a = op.a, b = NONE, next = pc + 1;
    if (code != OP_SPHERE) {  // OP_BOX / OP_BOX_SPH
      double bx[6] = {op.box[0], op.box[1], op.box[2], op.box[3], op.box[4], op.box[5]};
      if (!aabb_hit(bx, o, d, 1e-10, h.t)) {
        pc = op.skip;
        continue;
      }
      if (code == OP_BOX) {
        pc = next;
        continue;
      }
      b = op.b, next = op.skip;
    }
#pragma unroll 1
    for (int k = 0; k < 2; k++) {
      uint32_t payload = k == 0 ? a : b;
      if (payload != NONE && sphere_hit(spheres[payload & SPH_INDEX], payload, o, d, time, 1e-10, h)) flags++;
    }
    pc = next;
  }
  closest = h.t, hit_prim = h.prim;
  return flags;
}

#ifdef RL_FASTG_VERIFY  // debug build (make verify, tools/verify_fastg.py): every ray of the fast kernels is ALSO traced in the reference's order
__device__ unsigned int g_vcount;                                 // rays whose two answers differ
__device__ double g_vlog[64][12];                                  // the first 64 of them
__device__ unsigned long long g_vstats[4];                         // general kernel: TRAV steps, LEAF visits, far-origin rays; [3] rays verified by the sphere kernels
__device__ __forceinline__ void fast_verify_ray(const DevOp *ops, const DevSphere *spheres, D3 o, D3 d, double time, double closest, uint32_t hit_prim, double who) {
  double c2;
  uint32_t h2;
  fast_slow_trace(ops, spheres, o, d, time, c2, h2);
  const bool same = (hit_prim == NONE) == (h2 == NONE) && (h2 == NONE || ((h2 & SPH_INDEX) == (hit_prim & SPH_INDEX) && c2 == closest));
  if (!same) {
    unsigned k = atomicAdd(&g_vcount, 1u);
    if (k < 64) {
      double *L = g_vlog[k];
      L[0] = o.x, L[1] = o.y, L[2] = o.z, L[3] = d.x, L[4] = d.y, L[5] = d.z, L[6] = time, L[7] = hit_prim == NONE ? -1.0 : closest;
      L[8] = (double)(hit_prim & SPH_INDEX), L[9] = h2 == NONE ? -1.0 : c2, L[10] = (double)(h2 & SPH_INDEX), L[11] = who;
    }
  }
}
#endif
// rl_rtiow_coop.h (included after this header): what a wave of the STEAL instantiation runs once all its lanes have run out of pixels
template <int NT>
__device__ __forceinline__ void rtiow_steal_loop(const RtiowParams &P, unsigned long long *s_rng);

// STEAL (LDS_SCENE = 4, counter-free, the cost-sorted resume launch of a small shard): lanes hand a pixel over, at a sample boundary, to
// a wave that asks for it (P.steal_state), and a wave without work left asks — see rtiow_steal_loop
// (The parameter block arrives BY VALUE.  Measured in round 3: reading it from a device copy instead — scalar loads where a field is needed —
// removes every spill of the <1024, 4, false> instantiation (5 VGPRs / 75 SGPRs / 20 B of scratch -> 0 / 0 / 0) and is 1.2 % SLOWER
// (6625 against 6707 Mrays/s; the work-stealing instantiation 212 against 178 ms on the 1/8 shard): the v_readlane restores sit in block
// preambles, the scalar loads would sit in the blocks.)
template <int NT, int LDS_SCENE, bool STATS, bool STEAL = false>
__global__ void RL_KERNEL_ALIGN __launch_bounds__(NT) rtiow_wave_kernel(RtiowParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  // LDS layout: [linked ops][spheres][ChaCha rings 16 x NT u64]; with the scene in HBM the rings start at 0
  const size_t bits_words = ((size_t)P.n_spheres + 31) / 32 + 1;
  const size_t scene_lds = LDS_SCENE == 4   ? (((size_t)P.n_fast_inner * sizeof(FastNode) + 2 * bits_words * sizeof(uint32_t) + 15) & ~(size_t)15)
                           : LDS_SCENE == 3 ? (((size_t)P.n_cops * sizeof(CompactOp) + bits_words * sizeof(uint32_t) + 15) & ~(size_t)15)
                           : LDS_SCENE      ? ((size_t)P.n_ops * sizeof(DevOp) + (LDS_SCENE == 1 ? (size_t)P.n_spheres * sizeof(DevSphere) : 0))
                                            : 0;
  unsigned long long *s_rng = (unsigned long long *)(smem + scene_lds);  // [16][NT]
  const unsigned char *opbase = (const unsigned char *)P.lops;  // pc is an index (HBM) or a byte offset (LDS) into this
  const DevSphere *spheres = P.spheres;
  const uint32_t *s_bits = nullptr;  // LDS_SCENE == 3: one bit per sphere (Center::Moving)
  const uint32_t lds_base = (uint32_t)(uintptr_t)smem;  // the dynamic LDS segment's own address (low half of the flat address)
  if (LDS_SCENE == 4) {
    const uint4 *g = (const uint4 *)P.fast_nodes;
    uint4 *l = (uint4 *)smem;
    for (uint32_t i = tid; i < P.n_fast_inner * 4u; i += NT) l[i] = g[i];
    uint32_t *bl = (uint32_t *)(smem + (size_t)P.n_fast_inner * sizeof(FastNode));
    for (uint32_t i = tid; i < 2u * (uint32_t)bits_words; i += NT) bl[i] = P.movbits[i];  // [moving bits][specular-material bits]
    __syncthreads();
    opbase = smem;
    s_bits = bl;
  } else if (LDS_SCENE == 3) {
    const uint4 *g = (const uint4 *)P.cops;
    uint4 *l = (uint4 *)smem;
    for (uint32_t i = tid; i < P.n_cops * 2u; i += NT) {
      uint4 v = g[i];
      if (i & 1u) {  // {box[4], box[5], w_hit, w_miss}: successor indices -> absolute LDS addresses (no add per step)
        v.z = (v.z & 0xE0000000u) | (((v.z & 0x1FFFFFFFu) << 5) + lds_base);
        v.w = (v.w & 0xE0000000u) | (((v.w & 0x1FFFFFFFu) << 5) + lds_base);
      }
      l[i] = v;
    }
    uint32_t *bl = (uint32_t *)(smem + (size_t)P.n_cops * sizeof(CompactOp));
    for (uint32_t i = tid; i < (uint32_t)bits_words; i += NT) bl[i] = P.movbits[i];
    __syncthreads();
    opbase = smem;
    s_bits = bl;
  } else if (LDS_SCENE) {
    DevOp *s_ops = (DevOp *)smem;
    DevSphere *s_sph = (DevSphere *)(s_ops + P.n_ops);
    const uint4 *g = (const uint4 *)P.lops;
    uint4 *l = (uint4 *)s_ops;
    for (uint32_t i = tid; i < P.n_ops * 4u; i += NT) {
      uint4 v = g[i];
      if ((i & 3u) == 3u) {  // {w_hit, w_miss, a, b}: successor indices -> LDS byte offsets
        v.x = (v.x & 0xE0000000u) | ((v.x & 0x1FFFFFFFu) << 6);
        v.y = (v.y & 0xE0000000u) | ((v.y & 0x1FFFFFFFu) << 6);
        l[i] = v;
      }
    }
    // the box, re-stored as six floats in the first 24 bytes of the op (the binary64 box stays in HBM for the exact path)
    for (uint32_t i = tid; i < P.n_ops; i += NT) {
      const double *bx = P.lops[i].box;
      float *f = (float *)(s_ops + i);
#pragma unroll
      for (int k = 0; k < 6; k++) f[k] = (float)bx[k];
    }
    if (LDS_SCENE == 1) {
      g = (const uint4 *)P.spheres;
      l = (uint4 *)s_sph;
      for (uint32_t i = tid; i < P.n_spheres * 4u; i += NT) l[i] = g[i];
      spheres = s_sph;
    }
    __syncthreads();
    opbase = smem;
  }
  const uint32_t entry0 = LDS_SCENE == 4   ? 0u  // fast traversal: rays start through fast_start()
                          : LDS_SCENE == 3 ? ((P.centry0 & 0xE0000000u) | (((P.centry0 & 0x1FFFFFFFu) << 5) + lds_base))
                          : LDS_SCENE  ? ((P.entry0 & 0xE0000000u) | ((P.entry0 & 0x1FFFFFFFu) << 6))
                                       : P.entry0;
  const rl_rtiow_camera &cam = P.cam;
  const uint32_t W = cam.image_width;
  const uint32_t s_begin = P.sample_begin, spp = P.sample_end;  // this launch renders samples [s_begin, spp) of every pixel
  const uint64_t WH = (uint64_t)cam.image_width * (uint64_t)cam.image_height;
  const double INF = __longlong_as_double(0x7FF0000000000000ll);

  // ---- per-lane persistent state
  // (the stealing instantiation keeps the unrolled block function on its hot paths: measured, 1/8 and 1/4 shard 170 / 263 ms against 163 / 283 rolled)
  Ring<NT, !STEAL> rng{P.key, s_rng, tid, 0ull, 0u, 0u, 0u};
  uint32_t state = ST_GEN;
  uint32_t px = 0, pr = 0, n = spp;  // n == spp: no pixel owned yet
  uint32_t ptile = 0, pix_rays = 0;
  bool have_pixel = false, thin_pix = false;
  D3 sum = d3(0.0, 0.0, 0.0);
  D3 o = d3(0.0, 0.0, 0.0), d = d3(0.0, 0.0, 1.0), thr = d3(1.0, 1.0, 1.0);
  RayAux ra = ray_aux(o, d);
  RayAux32 ra32 = ray_aux32(ra);
  double time = 0.0, closest = INF;
  uint32_t pc = 0, hit_prim = NONE, depth = 0;
  uint32_t c_rays = 0, c_flag = 0, c_slow = 0;  // c_slow: rays the fast traversal handed to the reference-order fold
  unsigned long long c_nodes = 0, c_sph = 0, c_words = 0;

  // fast traversal (LDS_SCENE = 4): 16 ten-bit entry ids in five registers, newest in the low bits of stk0; all ones = empty
  uint32_t stk0 = ~0u, stk1 = ~0u, stk2 = ~0u, stk3 = ~0u, stk4 = ~0u;
  bool amb = false;  // this ray must be re-traced in the reference's order
  auto fast_push = [&](uint32_t e) {
    stk4 = __builtin_amdgcn_alignbit(stk4, stk3, 22), stk3 = __builtin_amdgcn_alignbit(stk3, stk2, 22);
    stk2 = __builtin_amdgcn_alignbit(stk2, stk1, 22), stk1 = __builtin_amdgcn_alignbit(stk1, stk0, 22);
    stk0 = (stk0 << 10) | e;
  };
  auto fast_pop = [&]() -> uint32_t {
    uint32_t e = stk0 & 1023u;
    stk0 = __builtin_amdgcn_alignbit(stk1, stk0, 10), stk1 = __builtin_amdgcn_alignbit(stk2, stk1, 10);
    stk2 = __builtin_amdgcn_alignbit(stk3, stk2, 10), stk3 = __builtin_amdgcn_alignbit(stk4, stk3, 10);
    stk4 = (stk4 >> 10) | (FAST_NONE << 22);
    return e;
  };
  // measured and lost (kept switchable): a separate block for Metal / Dielectric hits shortens SHADE (35.7 % -> 23.3 + 5.2 % of the
  // wave time) but a sixth state thins every other block (TRAV population 21.2 -> 18.5, LEAF 30.8 -> 27.0): 6.14 -> 5.52 Grays/s
  constexpr bool SPLIT_SHADE = false;
  // SPLIT_LEAF (fast traversal): LEAF only evaluates the discriminant — a sphere whose box the ray passed but which it misses (about half of
  // the visits) sends the lane straight back to TRAV — and the roots, the tie band and the order checks of an actual hit run in LEAF2.
  // Measured (round 3, -DRL_SPLIT_LEAF=true): 6709 -> 5907 Mrays/s, 1/8 shard 173 -> 205 ms — like SPLIT_SHADE, one more scheduling class
  // costs more in population per block than the shorter blocks give back (tools/sched.py: LEAF is 32 % of the time at 35 lanes per block)
  constexpr bool SPLIT_LEAF = RL_SPLIT_LEAF;
  auto shade_state = [&]() -> uint32_t {  // where a finished traversal is shaded
    if (!SPLIT_SHADE || LDS_SCENE != 4 || hit_prim == NONE) return ST_SHADE;
    const uint32_t si = hit_prim & SPH_INDEX;
    return ((s_bits[bits_words + (si >> 5)] >> (si & 31u)) & 1u) ? ST_SHADE2 : ST_SHADE;
  };
  auto fast_go = [&](uint32_t e) {  // continue with entry e: an inner node (TRAV), a sphere (LEAF), or nothing left
    if (e == FAST_NONE) {
#ifdef RL_FASTG_VERIFY
      if (LDS_SCENE == 4 && !amb) {  // a walk that trusts its own answer: the reference's fold must give the same one
        fast_verify_ray(P.ops, spheres, o, d, time, closest, hit_prim, 2.0);
        atomicAdd(&g_vstats[3], 1ull);
      }
#endif
      if (amb) pc = FAST_SLOW, state = ST_LEAF;
      else if (hit_prim == NONE) {  // a miss needs no SHADE visit: background (camera.rs:257), sample done (+10 %)
        sum = sum + thr * ld3(P.cam.background);
        n++;
        state = ST_GEN;
      } else state = shade_state();
    } else if (e >= P.n_fast_inner) pc = e, state = ST_LEAF;
    else pc = lds_base + (e << 6), state = ST_TRAV;
  };
  auto start_ray = [&]() {  // o, d set: per-ray constants of the box filter, then the first traversal state
    closest = INF, hit_prim = NONE;
    if (LDS_SCENE == 4) ra32 = ray_aux32_direct(o, d);
    else {
      ra = ray_aux(o, d);
      if (!ra.fast_ok) ra.slack = INF;
      ra32 = ray_aux32(ra);
    }
    if (LDS_SCENE == 4) {
      stk0 = stk1 = stk2 = stk3 = stk4 = ~0u;
      amb = !(ra32.slack < __int_as_float(0x7F800000));  // outside the binary32 filter's range: the reference's order from the start
      fast_go(amb ? FAST_NONE : P.fast_root);
    } else {
      pc = entry0 & 0x1FFFFFFFu;
      state = entry0 >> 29;
    }
  };

  // SHADE: miss -> background; hit -> rebuild the HitRecord, scatter, next ray.  MODE 0 = every material (reference-order kernels),
  // 1 = everything but Metal / Dielectric, 2 = Metal / Dielectric only (ST_SHADE2): the fast kernel's two halves
  auto shade = [&](auto mode) {
    constexpr int MODE = decltype(mode)::value;
        bool path_done = false;
        D3 nd = d;
        D3 p = o;
        if (MODE != 2 && hit_prim == NONE) {  // miss -> background (camera.rs:257)
          sum = sum + thr * ld3(cam.background);
          path_done = true;
        } else {
          uint32_t si = hit_prim & SPH_INDEX;
          const DevSphere &s = spheres[si];
          D3 c0 = ld3(s.c0);
          D3 center = (hit_prim & SPH_MOVING) ? c0 + ld3(s.dc) * time : c0;
          p = o + d * closest;
          D3 outward = (p - center) * s.inv_r;
          bool front = dot(d, outward) <= 0.0;
          D3 normal = front ? outward : -outward;
          // one flattened record per sphere (rl_render.hip flatten_sphere_materials): the material with a Solid texture's
          // colour inlined, and for a Dielectric the constants 1/ior and Schlick's r0 of both orientations — one load
          // instead of the sphere -> material -> texture chain, same values bit for bit
          const DevMaterial &m = P.sphere_flat[si];
          const uint32_t kind = m.kind & 0xFFu;
          const bool solid = (m.kind & MAT_TEX_SOLID) != 0u;
          // shared sub-expressions, evaluated once per block instead of once per material branch (same values, same
          // RNG order: the unit-sphere draw is the first draw of both Lambertian and Metal scatter)
          const bool is_lamb = MODE != 2 && kind == RL_MAT_LAMBERTIAN, is_metal = MODE != 1 && kind == RL_MAT_METAL, is_diel = MODE != 1 && kind == RL_MAT_DIELECTRIC;
          D3 us = d3(0.0, 0.0, 0.0);
          if (is_lamb | is_metal) us = rng.unit_sphere();
          D3 reflected = d - normal * (2.0 * dot(d, normal));  // material.rs reflect(): used by Metal
          D3 vin = is_metal ? reflected : d;
          D3 vn = vin;
          double m2 = len2(vin);
          if (is_metal | is_diel) vn = div_s(vin, sqrt(m2));  // normalize(): vec3.rs:56
          if (is_lamb) {
            D3 dir = normal + us;
            bool near_zero = approx_eq_eps(dir.x, 0.0, 1e-8) && approx_eq_eps(dir.y, 0.0, 1e-8) && approx_eq_eps(dir.z, 0.0, 1e-8);
            nd = near_zero ? normal : dir;
            thr = thr * (solid ? ld3(m.albedo) : texture_value(P, m.texture, 0.0, 0.0, p));
          } else if (is_metal) {
            nd = vn + us * m.fuzz;
            if (!(dot(nd, normal) > 0.0)) path_done = true;  // absorbed
            else thr = thr * ld3(m.albedo);
          } else if (is_diel) {
            double ri = front ? m.albedo[0] : m.ior;  // albedo[0] = 1.0 / ior
            D3 ud = vn;
            if (approx_eq_eps(m2, 0.0, 1e-16)) {
              c_flag++;
              ud = d;
            }
            double cos_theta = fmin(dot(-ud, normal), 1.0);
            double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
            bool reflect = ri * sin_theta > 1.0;
            if (!reflect) {
              double r0 = front ? m.albedo[1] : m.albedo[2];  // ((1 - ri) / (1 + ri))^2 for ri = 1/ior and ri = ior
              double xx = 1.0 - cos_theta;
              double x2 = xx * xx;
              double refl = r0 + (1.0 - r0) * (xx * (x2 * x2));
              reflect = refl > rng.gen_f64();
            }
            if (reflect) nd = ud - normal * (2.0 * dot(ud, normal));
            else {
              D3 perp = (ud + normal * cos_theta) * ri;
              D3 par = normal * (-sqrt(fabs(1.0 - len2(perp))));
              nd = perp + par;
            }
          } else if (MODE != 2 && kind == RL_MAT_DIFFUSE_LIGHT) {
            sum = sum + thr * (solid ? ld3(m.albedo) : texture_value(P, m.texture, 0.0, 0.0, p));
            path_done = true;
          } else {
            path_done = true;  // Flat
          }
        }
        if (!path_done) {
          depth--;
          if (depth == 0) path_done = true;  // ray_color(.., 0) = black
        }
        if (path_done) {
          n++;
          state = ST_GEN;
        } else {
          c_rays++;
          pix_rays++;
          o = p;
          d = nd;
          start_ray();
        }
  };

  unsigned long long sc_exec[7] = {0, 0, 0, 0, 0, 0, 0}, sc_pop[7] = {0, 0, 0, 0, 0, 0, 0}, sc_cyc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (;;) {
    // a finished traversal goes to SHADE; lanes reading from their newest ChaCha block top the ring up first
    if ((state == ST_SHADE || (SPLIT_SHADE && LDS_SCENE == 4 && state == ST_SHADE2)) && rng.low()) state = ST_FILL;
    // ---- wave scheduler: run the state with the most lanes in it (ties -> TRAV, SHADE, FILL, GEN)
    int n_trav = __popcll(__ballot(state == ST_TRAV));
    int n_shade = __popcll(__ballot(state == ST_SHADE));
    int n_fill = __popcll(__ballot(state == ST_FILL));
    int n_gen = __popcll(__ballot(state == ST_GEN));
    int n_leaf = __popcll(__ballot(state == ST_LEAF));
    int n_shade2 = SPLIT_SHADE && LDS_SCENE == 4 ? __popcll(__ballot(state == ST_SHADE2)) : 0;
    int n_leaf2 = SPLIT_LEAF && LDS_SCENE == 4 ? __popcll(__ballot(state == ST_LEAF2)) : 0;
    if (LATENCY_MODES && LDS_SCENE == 4 && P.thin_tiles != 0u) {  // parked lanes wake up when no lane of the wave holds a thin pixel any more
      if (__ballot(state == ST_PARK) != 0ull && __ballot(have_pixel && thin_pix) == 0ull) {
        if (state == ST_PARK) state = ST_GEN;
        n_gen = __popcll(__ballot(state == ST_GEN));
      }
    }
    if ((n_trav | n_shade | n_fill | n_gen | n_leaf | n_shade2 | n_leaf2) == 0) break;
    uint32_t pick = ST_TRAV;
    // (A/B, RL_TUNE third field w: TRAV competes with n_trav * w / 4 — a TRAV step costs a tenth of a LEAF or SHADE block, so running it for
    // fewer lanes feeds bigger LEAF / SHADE blocks; w = 4 is the plain most-lanes rule)
    int best = LDS_SCENE == 4 ? (n_trav * (int)P.tune[2]) >> 2 : n_trav;
    if (n_leaf > best) pick = ST_LEAF, best = n_leaf;
    if (n_shade > best) pick = ST_SHADE, best = n_shade;
    if (n_fill > best) pick = ST_FILL, best = n_fill;
    if (n_gen > best) pick = ST_GEN, best = n_gen;
    if (SPLIT_SHADE && LDS_SCENE == 4 && n_shade2 > best) pick = ST_SHADE2, best = n_shade2;
    if (SPLIT_LEAF && LDS_SCENE == 4 && n_leaf2 > best) pick = ST_LEAF2, best = n_leaf2;

    unsigned long long t_begin = 0;
    if (STATS) {  // debug (tools/sched.py): block executions, lanes served and shader cycles per state, per wave
      t_begin = __builtin_readcyclecounter();
      if (pick != ST_TRAV) {
#pragma unroll
        for (int k = 0; k < 7; k++)
          if (pick == (uint32_t)k) sc_exec[k]++, sc_pop[k] += (unsigned)best;
      }
    }
    if (pick == ST_TRAV) {
      // several steps per scheduling decision while the population stays near its starting size
      int floor_n = ((LDS_SCENE == 4 ? n_trav : best) * (int)P.tune[1]) >> 4;
      auto trav_step = [&]() {
        if (STATS) {
          int np = __popcll(__ballot(state == ST_TRAV));
          sc_exec[ST_TRAV]++, sc_pop[ST_TRAV] += (unsigned)np;
        }
        if (state == ST_TRAV) {
          // one LDS round trip: the whole 64-B linked op {box, w_hit, w_miss}; every op stepped here is a box op,
          // the successor words already carry the state the lane enters there (rl_render.hip link_ops)
          uint32_t w_hit, w_miss;
          bool certain, hitb;
          if (LDS_SCENE == 4) {  // one node = both children: reject-only binary32 tests, nearer child first, the other one pushed
            LdsFloat4 *nd = (LdsFloat4 *)(size_t)pc;
            const Float4 q0 = nd[0], q1 = nd[1], q2 = nd[2];
            const uint32_t w = *(LdsU32 *)(size_t)(pc + 48u);
            const float c32 = (float)closest;
            auto missed = [&](float b0, float b1, float b2, float b3, float b4, float b5, float &tmin) {
#if RL_PK_FMA  // both planes of a slab in one v_pk_fma_f32 (same IEEE fma per component)
              typedef float F2 __attribute__((ext_vector_type(2)));
              const F2 tx = __builtin_elementwise_fma(F2{b0, b1}, F2{ra32.invx, ra32.invx}, F2{-ra32.oix, -ra32.oix});
              const F2 ty = __builtin_elementwise_fma(F2{b2, b3}, F2{ra32.invy, ra32.invy}, F2{-ra32.oiy, -ra32.oiy});
              const F2 tz = __builtin_elementwise_fma(F2{b4, b5}, F2{ra32.invz, ra32.invz}, F2{-ra32.oiz, -ra32.oiz});
              const float t0x = tx.x, t1x = tx.y, t0y = ty.x, t1y = ty.y, t0z = tz.x, t1z = tz.y;
#else
              float t0x = fmaf(b0, ra32.invx, -ra32.oix), t1x = fmaf(b1, ra32.invx, -ra32.oix);
              float t0y = fmaf(b2, ra32.invy, -ra32.oiy), t1y = fmaf(b3, ra32.invy, -ra32.oiy);
              float t0z = fmaf(b4, ra32.invz, -ra32.oiz), t1z = fmaf(b5, ra32.invz, -ra32.oiz);
#endif
              tmin = fmaxf(fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z)), 1e-10f);
              float tmax = fminf(fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z)), c32);
              float diff = tmax - tmin;
              float thresh = fmaf(tmin + fabsf(tmax), 7.152557373046875e-07f, ra32.slack);  // 12u(|tmin|+|tmax|) + slack (ray_aux32_direct)
              return diff < -thresh;  // certainly tmin > tmax; false for NaN arithmetic: visit
            };
            if (STATS) c_nodes += 2;  // debug instantiation only (rl_debug_fast_stats): the fast structure's own tests, not the reference's
            float tA, tB;
            const bool hitA = !missed(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tA);
            const bool hitB = !missed(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tB);
            const uint32_t eA = w & 0xFFFFu, eB = w >> 16;
            const bool a_first = hitA && (!hitB || tA <= tB);
            uint32_t first = a_first ? eA : eB;
            if (hitA && hitB) fast_push(a_first ? eB : eA);
            if (!(hitA || hitB)) first = fast_pop();
            fast_go(first);
          } else if (LDS_SCENE == 3) {  // 32-byte op: binary32 box + the two successor words
            const LdsCompactOp &op = *(const LdsCompactOp *)(size_t)pc;
            float bx[6] = {op.box[0], op.box[1], op.box[2], op.box[3], op.box[4], op.box[5]};
            w_hit = op.w_hit, w_miss = op.w_miss;
            hitb = aabb_fast32(bx, ra32, (float)closest, certain);
            const uint32_t op_index = (pc - lds_base) >> 5;
            const bool guard = op_index >= P.n_ops;  // a sphere's own box: only ever REJECTS; not one of the reference's tests
            if (!certain) hitb = guard ? true : aabb_hit(P.ops[op_index].box, o, d, 1e-10, closest);  // rare: exact divisions
            if (STATS) c_nodes += guard ? 0u : 1u, c_sph += guard ? 1u : 0u;  // the guarded Sphere::hit counts, skipped or not
          } else {
            const DevOp &op = *(const DevOp *)(LDS_SCENE ? opbase + pc : opbase + (size_t)pc * sizeof(DevOp));
            w_hit = op.code, w_miss = op.skip;
            if (LDS_SCENE) {
              const float *fb = (const float *)&op;
              float bx[6] = {fb[0], fb[1], fb[2], fb[3], fb[4], fb[5]};
              hitb = aabb_fast32(bx, ra32, (float)closest, certain);  // non-finite boxes are NaN here, !fast_ok rays have slack = inf: never certain
            } else {
              double bx[6] = {op.box[0], op.box[1], op.box[2], op.box[3], op.box[4], op.box[5]};
              hitb = aabb_fast(bx, ra, closest, certain, P.k8u);
            }
            if (!certain) hitb = aabb_hit(P.ops[LDS_SCENE ? (pc >> 6) : pc].box, o, d, 1e-10, closest);  // rare: exact divisions
            if (STATS) c_nodes++;
          }
          if (LDS_SCENE != 4) {
            uint32_t w = hitb ? w_hit : w_miss;
            pc = w & 0x1FFFFFFFu;
            state = w >> 29;
          }
        }
      };
      for (int it = 0; it < (int)P.tune[0]; it += 2) {  // two steps per population check
        trav_step();
        trav_step();
        if (__popcll(__ballot(state == ST_TRAV)) < floor_n) break;
      }
    } else if (pick == ST_LEAF) {
      if (LDS_SCENE == 4) {
        if (state == ST_LEAF) {
          if (__builtin_expect(pc == FAST_SLOW, 0)) {  // rare: the answer may depend on the visiting order -> the reference's own fold
            c_flag += fast_slow_trace(P.ops, spheres, o, d, time, closest, hit_prim);
            c_slow++;
            state = shade_state();
          } else {
            const uint32_t sidx = pc - P.n_fast_inner;
            const uint32_t payload = sidx | (((s_bits[sidx >> 5] >> (sidx & 31u)) & 1u) ? SPH_MOVING : 0u);
            if (STATS) c_sph++;
            if (SPLIT_LEAF) {
              if (fast_sphere_misses(spheres[sidx], payload, o, d, time)) fast_go(fast_pop());
              else state = ST_LEAF2;
            } else {
              fast_sphere_hit(spheres[sidx], payload, o, d, time, ra32.oimax(), closest, hit_prim, amb);
              fast_go(fast_pop());
            }
          }
        }
      } else if (state == ST_LEAF) {  // Sphere::hit for the 1-2 spheres of a BVH leaf / one list item, in stored order
        uint32_t a, b, w;
        if (LDS_SCENE == 3) {  // a guard op: the ONE sphere it stands for (index = op index - n_ops), counted at the guard step
          const LdsCompactOp &op = *(const LdsCompactOp *)(size_t)pc;
          const uint32_t sidx = ((pc - lds_base) >> 5) - P.n_ops;
          a = sidx | (((s_bits[sidx >> 5] >> (sidx & 31u)) & 1u) ? SPH_MOVING : 0u), b = NONE, w = op.w_miss;
        } else {
          const DevOp &op = *(const DevOp *)(LDS_SCENE ? opbase + pc : opbase + (size_t)pc * sizeof(DevOp));
          a = op.a, b = op.b, w = op.skip;
          if (STATS) c_sph++;
        }
        Hit h{closest, hit_prim};
        if (sphere_hit(spheres[a & SPH_INDEX], a, o, d, time, 1e-10, h)) c_flag++;
        if (LDS_SCENE != 3 && b != NONE) {
          if (STATS) c_sph++;
          if (sphere_hit(spheres[b & SPH_INDEX], b, o, d, time, 1e-10, h)) c_flag++;
        }
        closest = h.t, hit_prim = h.prim;
        pc = w & 0x1FFFFFFFu;
        state = w >> 29;
      }
    } else if (SPLIT_LEAF && LDS_SCENE == 4 && pick == ST_LEAF2) {
      if (state == ST_LEAF2) {
        const uint32_t sidx = pc - P.n_fast_inner;
        const uint32_t payload = sidx | (((s_bits[sidx >> 5] >> (sidx & 31u)) & 1u) ? SPH_MOVING : 0u);
        fast_sphere_hit(spheres[sidx], payload, o, d, time, ra32.oimax(), closest, hit_prim, amb);
        fast_go(fast_pop());
      }
    } else if (pick == ST_FILL) {
      if (state == ST_FILL) {
        rng.top_up();
        state = shade_state();
      }
    } else if (pick == ST_GEN) {
      if (state == ST_GEN) {
        bool active = true;
        if (STEAL && P.steal_state && have_pixel && n < spp) {  // a sample boundary: has a wave without work asked for this pixel?
          const size_t pix = (size_t)pr * W + px;
          if (__atomic_load_n(&P.steal_state[pix], __ATOMIC_RELAXED) == 1u) {
            double *outp = P.out + pix * 3;
            outp[0] = sum.x, outp[1] = sum.y, outp[2] = sum.z;
            P.pos_state[pix] = rng.pos;
            P.steal_n[pix] = n;
            __threadfence();  // the state above is visible before the release
            if (atomicCAS(&P.steal_state[pix], 1u, 2u) == 1u) {
              have_pixel = false;
              n = spp;  // -> claim (the queue is empty by now: the lane is done)
            }  // else: the request was withdrawn in the meantime — the pixel stays here
          }
        }
        if (n >= spp) {  // pixel finished (or none yet): write it out, claim the next slot
          if (have_pixel) {
            size_t pix = (size_t)pr * W + px;
            double *outp = P.out + pix * 3;
            outp[0] = sum.x, outp[1] = sum.y, outp[2] = sum.z;
            if (STEAL && P.steal_state) atomicExch(&P.steal_state[pix], 3u);  // finished: a request that arrives now finds nothing to take
            if (P.pos_state) P.pos_state[pix] = rng.pos;               // resumable: the next launch continues this pixel
            if (P.tile_cost) atomicAdd(&P.tile_cost[ptile], pix_rays);  // cost estimate for the LPT order of the next launch
            if (STATS && !P.tile_cost) c_words += rng.pos;
            if (STATS && P.pix_rays) P.pix_rays[pix] += pix_rays;
            have_pixel = false;
          }
          uint32_t slot = wave_claim(P.work_counter);
          // THIN tiles (latency mode, LDS_SCENE = 4, cost-sorted resume launch of a small shard): the P.thin_tiles most expensive tiles
          // hand out their 64 pixels over 1 << thin_shift wave-claims of 64 slots (16 … 1 pixels each); a lane that draws an empty slot
          // parks until its wave's thin pixels are done.  A wave then serialises 16 sample chains instead of 64: the longest chains
          // of the frame (the shard's critical path) see a quarter of the state divergence.
          const uint32_t sh = P.thin_shift;  // a thin tile's 64 pixels go out over 1 << sh wave-claims, 64 >> sh pixels each
          const uint32_t thin_slots = LATENCY_MODES && LDS_SCENE == 4 ? (P.thin_tiles * 64u) << sh : 0u;
          thin_pix = false;
          bool parked = false;
          if (slot < thin_slots) {
            const uint32_t sub = slot & ((64u << sh) - 1u), l = sub & 63u;
            parked = (l & ((1u << sh) - 1u)) != 0u;
            thin_pix = !parked;
            slot = (slot >> (6u + sh)) * 64u + (sub >> 6) * (64u >> sh) + (l >> sh);
          } else if (LATENCY_MODES) {
            slot -= thin_slots - (LDS_SCENE == 4 ? P.thin_tiles * 64u : 0u);
            if (LDS_SCENE == 4 && P.prio_tiles != 0u) thin_pix = (slot >> 6) < P.prio_tiles;
          }
          if (parked) {
            state = ST_PARK;
            active = false;
          } else if (slot >= P.n_slots) {
            state = ST_DONE;
            active = false;
          } else {
            uint32_t tile = slot >> 6, in = slot & 63u;
            if (P.tile_order) tile = P.tile_order[tile];  // expensive tiles first
            ptile = tile;
            px = (tile % P.tiles_x) * 8u + (in & 7u);
            pr = (tile / P.tiles_x) * 8u + (in >> 3);
            if (px >= W || pr >= P.nrows) active = false;  // slot outside the image: stay in GEN, claim again next time
            else {
              have_pixel = true;
              n = s_begin;
              pix_rays = 0;
              if (P.resume) {  // continue where the previous launch stopped: same sums, same ChaCha word position
                size_t pix = (size_t)pr * W + px;
                const double *inp = P.out + pix * 3;
                sum = d3(inp[0], inp[1], inp[2]);
                rng.pos = P.pos_state[pix];
              } else {
                rng.pos = 0;
                sum = d3(0.0, 0.0, 0.0);
              }
              rng.nres = 0;
              if (n >= spp) active = false;
            }
          }
        }
        if (active) {
          uint32_t y = P.row_first + pr * P.row_step;
          uint64_t sample_index = (uint64_t)n + P.first_sample;
          rng.reset_stream(sample_index * WH + (uint64_t)px * (uint64_t)W + (uint64_t)y);  // camera.rs:167-170
          // get_ray camera.rs:203-216
          D3 p00 = ld3(cam.pixel_00), du = ld3(cam.pixel_du), dv = ld3(cam.pixel_dv);
          D3 pixel_center = (p00 + du * (double)px) + dv * (double)y;
          double sx = -0.5 + rng.gen_f64();
          double sy = -0.5 + rng.gen_f64();
          D3 pixel_sample = pixel_center + (du * sx + dv * sy);
          if (cam.defocus_angle <= 0.0) o = ld3(cam.lookfrom);
          else {
            double a, b;
            rng.unit_disc(a, b);
            o = (ld3(cam.lookfrom) + ld3(cam.defocus_disk_u) * a) + ld3(cam.defocus_disk_v) * b;
          }
          d = pixel_sample - o;
          time = rng.gen_f64();
          thr = d3(1.0, 1.0, 1.0);
          depth = cam.max_depth;
          if (depth == 0) {  // ray_color(depth 0) = black: the sample contributes (0,0,0)
            sum = sum + d3(0.0, 0.0, 0.0);
            n++;
          } else {
            c_rays++;
            pix_rays++;
            start_ray();
          }
        }
      }
      if (LATENCY_MODES && LDS_SCENE == 4 && P.prio_tiles != 0u) {  // A/B: issue priority for the waves that hold the frame's longest sample chains
        if (__ballot(have_pixel && thin_pix) != 0ull) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(0);
      }
    } else if (SPLIT_SHADE && pick == ST_SHADE2) {
      if (state == ST_SHADE2) shade(std::integral_constant<int, 2>{});
    } else {  // ST_SHADE
      if (state == ST_SHADE) {
        if (SPLIT_SHADE && LDS_SCENE == 4) shade(std::integral_constant<int, 1>{});
        else shade(std::integral_constant<int, 0>{});
      }
    }
    if (STATS) {
      unsigned long long dt = __builtin_readcyclecounter() - t_begin;
#pragma unroll
      for (int k = 0; k < 7; k++)
        if (pick == (uint32_t)k) sc_cyc[k] += dt;
    }
  }
  if (STATS && (tid & 63) == 0) {
    unsigned long long *sched = P.stats + 8;  // [3*s] executions, [3*s+1] lanes served, [3*s+2] cycles
#pragma unroll
    for (int s = 0; s < 7; s++) {
      atomicAdd(&sched[3 * s], sc_exec[s]);
      atomicAdd(&sched[3 * s + 1], sc_pop[s]);
      atomicAdd(&sched[3 * s + 2], sc_cyc[s]);
    }
  }

  if (STEAL && LDS_SCENE == 4 && P.steal_state) {
    // every lane of this wave is out of work: take over pixels that other lanes are still rendering (most expensive tiles first) and
    // run them one at a time with all 64 lanes (its own counters go to P.stats from there)
    rtiow_steal_loop<NT>(P, rng.s_rng);
  }
  unsigned long long v;
  v = wave_sum((unsigned long long)c_rays);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[0], v);
  v = wave_sum((unsigned long long)c_flag);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[6], v);
  if (LDS_SCENE == 4) {
    v = wave_sum((unsigned long long)c_slow);
    if ((tid & 63) == 0 && v) atomicAdd(&P.stats[7], v);
  }
  if (STATS) {
    v = wave_sum(c_nodes);
    if ((tid & 63) == 0) atomicAdd(&P.stats[1], v);
    v = wave_sum(c_sph);
    if ((tid & 63) == 0) atomicAdd(&P.stats[2], v);
    v = wave_sum(c_words);
    if ((tid & 63) == 0) atomicAdd(&P.stats[5], v);
  }
}

}  // namespace rl
