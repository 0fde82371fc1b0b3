// RTIOW all-primitives kernel with the FAST traversal (counter-free renders only): the wave-scheduled state machine of
// rl_rtiow_wave_general.h walking the world-space surface-area-heuristic tree of rl_fast_bvh.cpp (build_fast_general) instead of the
// reference's threaded program.
//
//   TRAV  one 64-byte node from HBM / L2 / Infinity Cache = both children's binary32 boxes: reject-only tests (rl_rtiow_wave.h
//         ray_aux32_direct), nearer child first, the farther one pushed on a per-lane stack in LDS ([entry][lane]: conflict-free)
//   LEAF  one primitive OCCURRENCE: the world ray is taken through the occurrence's PUSH chain (transform.rs:145-149, translate.rs:15)
//         and the reference's Sphere::hit / Plane::hit_ab arithmetic yields the root only — no HitRecord is kept while traversing
//   SHADE the winning occurrence is evaluated once more with ray_t.max = its root (same arithmetic, same root) for the full
//         HitRecord, which then takes the POP chain (transform.rs:152-161) — or, for an order-sensitive ray, the whole ray is re-traced by
//         general_slow_trace, the reference's own fold — and is shaded as in rl_rtiow_wave_general.h
// No XF state, no HitRecord and no second ray live across states: 2 instead of 6 wave states touch the scene.
// Order-sensitive rays (rl_fast_bvh.cpp): two roots within fast_tie_band of each other, grazing sphere hits, planar hits within 1e-9 of an edge,
// rays outside the binary32 filter's range or whose origin is farther than r_safe from the scene's centre, stack overflow.
#pragma once
#include "rl_rtiow_general.h"
#include "rl_rtiow_wave.h"

namespace rl {

// The reference's fold over its own program (bvh.rs:79-95, hittable/mod.rs:88-111, transform.rs:143-164), exact divisions; returns the
// HitRecord in world space and the number of panic sites reached.
template <bool TRANS>
__device__ __forceinline__ uint32_t general_slow_trace(const RtiowParams &P, const DevOp *ops, D3 wo, D3 wd, double time, Rec &rec) {
  uint32_t flags = 0;
  D3 o = wo, d = wd;
  uint32_t pc = 0;
#pragma unroll 1
  for (;;) {
    const DevOp &op = ops[pc];
    uint32_t code = op.code & 0xFFu;
    if (code == OP_END) break;
    if (code == OP_BOX || code == OP_BOX_SPH || code == OP_BOX_PLANAR) {
      double bx[6] = {op.box[0], op.box[1], op.box[2], op.box[3], op.box[4], op.box[5]};
      if (!aabb_hit(bx, o, d, 1e-10, rec.t)) {
        pc = op.skip;
        continue;
      }
      if (code == OP_BOX) {
        pc++;
        continue;
      }
    }
    if (code == OP_BOX_SPH || code == OP_SPHERE) {
#pragma unroll 1
      for (int k = 0; k < 2; k++) {
        uint32_t pl = k == 0 ? op.a : op.b;
        if (pl == NONE || (k == 1 && code == OP_SPHERE)) continue;
        uint32_t si = pl & SPH_INDEX;
        if (sphere_hit_rec(P.spheres[si], pl, P.sphere_material[si], pc, o, d, time, rec)) flags++;
      }
      pc = code == OP_SPHERE ? pc + 1 : op.skip;
      continue;
    }
    if (code == OP_BOX_PLANAR || code == OP_PLANAR) {
#pragma unroll 1
      for (int k = 0; k < 2; k++) {
        uint32_t pl = k == 0 ? op.a : op.b;
        if (pl == NONE || (k == 1 && code == OP_PLANAR)) continue;
        if (planar_hit_rec(P.planars[pl], pc, o, d, rec)) flags++;
      }
      pc = code == OP_PLANAR ? pc + 1 : op.skip;
      continue;
    }
    if (code == OP_PUSH_TRANSLATE) o = o - ld3(P.translates[op.a].offset);
    else if (code == OP_PUSH_TRANSFORM) {
      const rl_transform &t = P.transforms[op.a];
      D3 no = mat3_mul(t.inv, o), nd = mat3_mul(t.inv, d);
      o = no, d = nd;
    } else {  // POP: op.b = the matching PUSH, whose .b is the parent PUSH
      uint32_t push_pc = op.b;
      if (rec.any && rec.pc > push_pc) {
        if (code == OP_POP_TRANSLATE) rec.p = rec.p + ld3(P.translates[op.a].offset);
        else {
          const rl_transform &t = P.transforms[op.a];
          rec.p = mat3_mul(t.m, rec.p);
          D3 wn = mat3_mul(t.inv_t, rec.normal);
          double m = len2(wn);
          if (approx_eq_eps(m, 0.0, 1e-16)) flags++;
          else rec.normal = normalize(wn);
        }
      }
      replay_chain(P, ops, ops[push_pc].b, wo, wd, o, d);
    }
    pc++;
  }
  return flags;
}

// A fast walk that has taken this many steps (nodes + primitive tests) is given up: the ray is re-traced by the reference's own fold, which
// prunes by ITS boxes (cfg 5: ~140 tests).  The rays that get here are far-origin rays (`unsafe`: every box widened by `grow`, no pruning by
// the closest hit) whose widened boxes overlap half the scene — a handful per frame walk 1e5 ... 1e6 steps, 30 ... 460 ms EACH in a
// traversal-only launch, and the megakernel's last lanes were those too: cfg 5 at 64 spp 2.54 -> 1.87 s.  Any budget is correct (the fold
// is the reference); measured at 64 spp: 64 steps 3.56 s (16.7 M rays re-traced), 128: 2.20 s (5.1 M), 256: 1.89 s, 512: 1.87 s, 2048: 1.88 s
// (4.15 M: the rays that are re-traced for other reasons).
static const uint32_t FASTG_STEP_BUDGET = 512;
#ifdef RL_FASTG_VERIFY  // debug build (tools/verify_fastg.py): every ray is ALSO traced in the reference's order; mismatches are logged
// (g_vcount / g_vlog / g_vstats: rl_rtiow_wave.h — the sphere kernels log into them too)
#endif
// Sphere::hit / Plane::hit_ab for the ROOT only, acceptance window widened by the tie band (see fast_sphere_hit in rl_rtiow_wave.h)
__device__ __forceinline__ void fastg_planar_hit(const DevPlanar &pl, D3 o, D3 d, float oimax, uint32_t item, double &closest, uint32_t &best, bool &amb) {
  D3 normal = ld3(pl.normal);
  double denom = dot(normal, d);
  if (fabs(denom) < 1e-8) return;
  double t = (pl.d - dot(normal, o)) / denom;
  // t's rounding error scales with the cancelling terms of its numerator, not with t
  const double band = fast_tie_band((fabs(pl.d) + fabs(normal.x * o.x) + fabs(normal.y * o.y) + fabs(normal.z * o.z)) / fabs(denom), oimax);
  if (!(1e-10 <= t && t <= closest + band)) return;
  D3 p = o + d * t;
  D3 hp = p - ld3(pl.q);
  D3 w = ld3(pl.w);
  double alpha = dot(w, cross(hp, ld3(pl.v)));
  double beta = dot(w, cross(ld3(pl.u), hp));
  // within the hit point's own uncertainty (root error x |d|, seen through alpha = w . (hp x v), beta = w . (u x hp)) of an edge: the
  // reference's leaf box is the exact bound of the vertices, so such a hit may or may not pass it
  const double w1 = fabs(w.x) + fabs(w.y) + fabs(w.z), d1 = fabs(d.x) + fabs(d.y) + fabs(d.z);
  const double uv1 = fmax(fabs(pl.u[0]) + fabs(pl.u[1]) + fabs(pl.u[2]), fabs(pl.v[0]) + fabs(pl.v[1]) + fabs(pl.v[2]));
  const double e = 1e-9 + band * d1 * w1 * uv1;
  bool inside, edge;
  if (pl.kind == RL_PLANAR_QUAD) {
    inside = 0.0 <= alpha && alpha <= 1.0 && 0.0 <= beta && beta <= 1.0;
    edge = fabs(alpha) <= e || fabs(alpha - 1.0) <= e || fabs(beta) <= e || fabs(beta - 1.0) <= e;
  } else if (pl.kind == RL_PLANAR_TRIANGLE) {
    inside = 0.0 <= alpha && 0.0 <= beta && alpha + beta <= 1.0;
    edge = fabs(alpha) <= e || fabs(beta) <= e || fabs(alpha + beta - 1.0) <= e;
  } else {  // an unbounded Plane (a stage of its own: FastGeneral::stage_roots): plane.rs:51-100 has no interior test, and no box anywhere
    inside = true, edge = false;
  }
  if (!inside) return;
  if (edge) amb = true;  // an edge-grazing hit may or may not pass the reference's own leaf box (the exact bound of the vertices)
  if (best != NONE && fabs(t - closest) <= band) amb = true;
  if (t <= closest) closest = t, best = item;
}

__device__ __forceinline__ void fastg_sphere_hit(const DevSphere &s, uint32_t payload, D3 o, D3 d, double time, float oimax, uint32_t item, double &closest,
                                                 uint32_t &best, bool &amb) {
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
  // a sphere seen from more than ~5e4 radii away: the reference's from_normalized assert (vec3.rs:219) can fire for it, and whether
  // it is accepted along the way depends on the reference's order (rays from inside r_safe never get here: build_fast_general)
  if ((r_l >= 1e-10 || r_u >= 1e-10) && c + s.r2 > 2.5e9 * s.r2) amb = true;
  const double band = fast_tie_band(fabs(r_l) + fabs(r_u), oimax);
  const double hi = closest + band;
  double t;
  if (1e-10 <= r_l && r_l <= hi) t = r_l;
  else if (1e-10 <= r_u && r_u <= hi) t = r_u;
  else return;
  if (best != NONE && fabs(t - closest) <= band) amb = true;  // grazing / pole hits: only the winner matters, checked in SHADE
  if (t <= closest) closest = t, best = item;
}

// SD = entries of the per-lane LDS stack (a deeper pending list re-traces the ray in the reference's order: the tree may be 40 deep,
// the list of pending far children hardly ever is)
// Nodes are four wide (FastNodeQ, 128 B = one L2 line: one dependent fetch per two levels of the surface-area-heuristic binary tree,
// children visited nearest first).  Measured against the two-wide pair nodes they replaced: cfg 4 +1.3 %, cfg 5 (150 MB of nodes, items
// and spheres behind 4 MB of L2 per XCD) +6 ... 8 %.  Also measured, and dropped: a 10-bit lower bound of the entry distance in every stack
// entry, so that pop() can skip entries that have fallen behind the closest hit (cfg 4 -10 %, cfg 5 -6 %: the skipped visits are worth less
// than the dependent LDS round trips of the skipping loop).
// OCTO: the eight-wide nodes with quantised boxes (FastNodeO) instead of the four-wide ones (FastNodeQ)
// MEDIA (round 3): scenes with ConstantMedium objects (constant_medium.rs:27-80, deterministic variant: include/rl_render.h rl_medium).  The
// reference evaluates a medium where its fold reaches it, with ray_t.max = the closest hit found by everything BEFORE it in the program,
// and draws the free path from the RNG only if the ray's stretch inside the boundary, clamped to that closest hit, is non-empty.  So the
// items are cut into segments at the media (FastGeneral::seg_roots): a ray walks segment 0's tree, evaluates medium 0 with the closest hit
// so far (the boundary by the reference's own fold over its ops, twice; the draw from the pixel's ChaCha8 ring), walks segment 1's tree
// with the same running closest hit, ... — exactly the reference's order at the granularity that matters.  (BVH boxes around a medium
// never decide whether it draws: a ray that misses a box inside [t_min, closest] has an empty stretch inside the boundary, too.)  An
// order-sensitive ray rewinds the ring to the word position the ray started at and is re-traced by the reference's fold WITH its media.
template <int NT, int SD, bool TRANS, bool OCTO = false, bool MEDIA = false>
__global__ void RL_KERNEL_ALIGN __launch_bounds__(NT) rtiow_fast_general_kernel(const RtiowParams *__restrict__ Pp) {
  // the parameter block is read from memory where it is needed (uniform addresses: scalar loads through the constant cache) instead of
  // arriving by value: by value every field that is live anywhere is loaded at kernel entry and pins SGPRs for the kernel's life time
  const RtiowParams &P = *Pp;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  unsigned long long *s_rng = (unsigned long long *)smem;                                   // [16][NT]
  uint32_t *s_stack = (uint32_t *)(smem + (size_t)16 * NT * sizeof(unsigned long long));  // [SD][NT]
  uint4 *s_top = (uint4 *)(smem + (size_t)NT * (16 * sizeof(unsigned long long) + SD * sizeof(uint32_t)));  // [fg_top] FastNodeQ: the tree's top
  const DevOp *ops = P.ops;
  const FastNodeQ *nodes = P.fg_nodes;
  const uint32_t top = OCTO ? 0u : P.fg_top;
  if (top) {
    for (uint32_t i = (uint32_t)tid; i < top * 8u; i += (uint32_t)NT) s_top[i] = ((const uint4 *)nodes)[i];
    __syncthreads();
  }
  const FastItem *items = P.fg_items;
  const rl_rtiow_camera &cam = P.cam;
  const uint32_t W = cam.image_width;
  const uint32_t s_begin = P.sample_begin, spp = P.sample_end;
  const uint64_t WH = (uint64_t)cam.image_width * (uint64_t)cam.image_height;
  const double INF = __longlong_as_double(0x7FF0000000000000ll);
  const float FINF = __int_as_float(0x7F800000);

  Ring<NT> rng{P.key, s_rng, tid, 0ull, 0u, 0u, 0u};
  uint32_t state = ST_GEN;
  uint32_t px = 0, pr = 0, n = spp;
  uint32_t ptile = 0, pix_rays = 0;
  bool have_pixel = false;
  D3 sum = d3(0.0, 0.0, 0.0);
  D3 wo = d3(0.0, 0.0, 0.0), wd = d3(0.0, 0.0, 1.0);
  D3 thr = d3(1.0, 1.0, 1.0);
  RayAux32 ra32 = ray_aux32_direct(wo, wd);
  double time = 0.0, closest = INF;
  uint32_t pc = 0, best = NONE, depth = 0, sp = 0, steps = 0;
  const uint32_t step_budget = P.tune[3];
  uint32_t seg = 0, pos0 = 0, ray_flags = 0;  // MEDIA: stage being walked, ChaCha word position at the start of the ray, panic sites of its boundary walks
  bool amb = false;
  // rays that start farther than r_safe from the scene's centre (e.g. inside a huge ground sphere): the boxes' padding was sized for
  // origins inside r_safe, so such a ray widens every box interval by `grow` and does NOT prune by the closest hit — every sphere its
  // line touches is then tested, and fastg_sphere_hit's far-origin check covers everything the reference could accept on the way
  float grow = 0.0f;
  bool unsafe = false;
  uint32_t c_rays = 0, c_flag = 0, c_slow = 0;
#ifdef RL_FASTG_VERIFY
  unsigned long long c_steps = 0, c_leaves = 0, c_unsafe = 0;
#endif

  auto go = [&](uint32_t e) {  // continue with entry e: an inner node (TRAV), an item (LEAF), or nothing left (SHADE)
    if (++steps > step_budget) amb = true, e = NONE;
    // MEDIA: this stage is done: the next one (a segment's tree, or a medium behind its box node: P.fg_seg_roots holds FastGeneral::stage_roots)
    if (MEDIA && !amb)
      while (e == NONE && seg + 1u < P.fg_n_seg) e = P.fg_seg_roots[++seg];
    if (e == NONE) {
#ifdef RL_FASTG_VERIFY
      if (false) {
#else
      if (!amb && best == NONE) {  // a miss needs no SHADE visit: background (camera.rs:257), sample done
#endif
        if (MEDIA) c_flag += ray_flags;
        sum = sum + thr * ld3(cam.background);
        n++;
        state = ST_GEN;
      } else state = ST_SHADE;
    } else {
      pc = e;
      state = (e & FASTG_LEAF) ? ST_LEAF : ST_TRAV;
    }
  };
  auto pop = [&]() -> uint32_t {
    if (sp == 0) return NONE;
    sp--;
    return s_stack[(size_t)sp * NT + tid];
  };
  auto start_ray = [&]() {
    closest = INF, best = NONE, sp = 0, steps = 0;
    if (MEDIA) seg = 0, pos0 = rng.pos, ray_flags = 0;
    ra32 = ray_aux32_direct(wo, wd);
    float fx = (float)wo.x - P.fg_center[0], fy = (float)wo.y - P.fg_center[1], fz = (float)wo.z - P.fg_center[2];
    float far2 = fmaf(fx, fx, fmaf(fy, fy, fz * fz));
    amb = !(ra32.slack < FINF);  // outside the binary32 filter's range: the reference's order
    unsafe = !(far2 <= P.fg_rsafe2);
    grow = 0.0f;
#ifdef RL_FASTG_VERIFY
    c_unsafe += unsafe ? 1u : 0u;
    if (unsafe && P.tune[2] == 77u) {  // RL_TUNE=a,b,77: log the first far-origin rays instead of mismatches (what starts out there?)
      unsigned k = atomicAdd(&g_vcount, 1u);
      if (k < 64) {
        double *L = g_vlog[k];
        L[0] = wo.x, L[1] = wo.y, L[2] = wo.z, L[3] = wd.x, L[4] = wd.y, L[5] = wd.z, L[6] = time, L[7] = (double)far2;
        L[8] = (double)depth, L[9] = (double)P.fg_rsafe2, L[10] = (double)px, L[11] = 9.0;
      }
    }
#endif
    if (unsafe) {  // pad(L) = fg_pad_k * L^2 in world units (rl_fast_bvh.cpp), L = distance to the centre + the scene's radius; in units of t: / min |d_k|
      float L = sqrtf(far2) + P.fg_radius;
      grow = P.fg_pad_k * L * L * fmaxf(fmaxf(fabsf(ra32.invx), fabsf(ra32.invy)), fabsf(ra32.invz));
      if (!(grow < FINF)) amb = true;
    }
#ifdef RL_EXPERIMENTAL
    go(amb ? NONE : (MEDIA ? P.fg_seg_roots[0] : OCTO ? P.fg_oroot : P.fg_root));
#else
    go(amb ? NONE : (MEDIA ? P.fg_seg_roots[0] : P.fg_root));
#endif
  };

  for (;;) {
    if (state == ST_SHADE && rng.low()) state = ST_FILL;
    int n_trav = __popcll(__ballot(state == ST_TRAV));
    int n_shade = __popcll(__ballot(state == ST_SHADE));
    int n_fill = __popcll(__ballot(state == ST_FILL));
    int n_gen = __popcll(__ballot(state == ST_GEN));
    int n_leaf = __popcll(__ballot(state == ST_LEAF));
    if ((n_trav | n_shade | n_fill | n_gen | n_leaf) == 0) break;
    uint32_t pick = ST_TRAV;
    int bestn = (n_trav * (int)P.tune[2]) >> 2;  // (A/B, RL_TUNE third field: TRAV's weight against the other states in quarters; 4 = the plain most-lanes rule)
    if (n_leaf > bestn) pick = ST_LEAF, bestn = n_leaf;
    if (n_shade > bestn) pick = ST_SHADE, bestn = n_shade;
    if (n_fill > bestn) pick = ST_FILL, bestn = n_fill;
    if (n_gen > bestn) pick = ST_GEN, bestn = n_gen;

    if (pick == ST_TRAV) {
      int floor_n = (n_trav * (int)P.tune[1]) >> 4;
      for (int it = 0; it < (int)P.tune[0]; it++) {
        if (state == ST_TRAV) {
#ifdef RL_FASTG_VERIFY
          c_steps++;
#endif
#ifdef RL_EXPERIMENTAL
          if (OCTO) {
            const float c32 = unsafe ? FINF : (float)closest;
            auto push = [&](uint32_t e) {
              if (sp < (uint32_t)SD) s_stack[(size_t)sp * NT + tid] = e, sp++;
              else amb = true;  // more pending children than the stack holds: the reference's order decides
            };
            // ---- one FastNodeO: eight children, boxes on the node's 8-bit grid.  For child k and axis x the stored planes are
            // B = o_x + q * S_x (S_x a power of two), so t = (B - ray.o_x) / d_x = q * a_x + b_x with a_x = S_x / d_x (exact scaling of the
            // ray's binary32 reciprocal) and b_x = (o_x - ray.o_x) / d_x = fma(o_x, inv_x, -oi_x): TWO instructions per plane (byte -> float,
            // fma), and the near / far plane of a slab is known from the sign of d_x — no per-axis min / max.
            // Error against the exact quotient, with ray_aux32_direct's bounds (inv32 = (1/d)(1 + e), |e| <= 3u; oi32 within 5u |o/d|; u = 2^-24):
            //   |q a - q S/d| <= 3u |q S/d|,  |b'' - b| <= 3u |o_x/d| (product) + 5u |o/d| + u |b|,  one rounding in the final fma:
            //   |t'' - t| <= 4u |t''| + 7u |b_x| + 8u max|o/d|   (|q S/d| <= |t| + |b|, |o_x/d| <= |b| + |o/d|).
            // The first and last terms are inside the threshold every reject-only test of this kernel already uses (6u |t| + 9u max|o/d| per end);
            // the middle one is folded into the planes themselves: the near plane is evaluated with b - 10u |b|, the far one with b + 10u |b|
            // (per node and axis, three instructions), so each computed interval CONTAINS the exact one whatever the other axes do — an axis
            // the ray runs nearly parallel to has huge |b| and huge errors, and simply never decides (its near / far are -/+ huge).
            const uint4 *nd = (const uint4 *)(P.fg_onodes + pc);
            const uint4 h = nd[0], qa = nd[1], qb = nd[2], qc = nd[3], ca = nd[4], cb = nd[5];
            float nax, nbn, nbf, nay, nbyn, nbyf, naz, nbzn, nbzf;
            auto axis = [&](uint32_t obits, uint32_t ebits, float inv, float oi, float &a, float &bn, float &bf) {
              a = __uint_as_float((ebits & 0xFFu) << 23) * inv;
              const float b = fmaf(__uint_as_float(obits), inv, -oi);
              const float c = fabsf(b) * 5.9604644775390625e-07f;  // 10u |b|
              bn = b - c, bf = b + c;
              // outside the range in which q a + b is a faithful sum (products near the ends of binary32): the axis does not constrain
              const bool ok = fabsf(a) > 1e-30f && fabsf(a) < 1e27f && fabsf(b) < 1e30f;
              a = ok ? a : 0.0f, bn = ok ? bn : -FINF, bf = ok ? bf : FINF;
            };
            axis(h.x, h.w, ra32.invx, ra32.oix, nax, nbn, nbf);
            axis(h.y, h.w >> 8, ra32.invy, ra32.oiy, nay, nbyn, nbyf);
            axis(h.z, h.w >> 16, ra32.invz, ra32.oiz, naz, nbzn, nbzf);
            // the planes a ray meets first / last on each axis: qlo / qhi swapped for negative directions (eight children = two words per axis)
            const bool ngx = ra32.invx < 0.0f, ngy = ra32.invy < 0.0f, ngz = ra32.invz < 0.0f;
            const uint32_t lx0 = qa.x, lx1 = qa.y, ly0 = qa.z, ly1 = qa.w, lz0 = qb.x, lz1 = qb.y;
            const uint32_t hx0 = qb.z, hx1 = qb.w, hy0 = qc.x, hy1 = qc.y, hz0 = qc.z, hz1 = qc.w;
            const uint32_t nx0 = ngx ? hx0 : lx0, nx1 = ngx ? hx1 : lx1, fx0 = ngx ? lx0 : hx0, fx1 = ngx ? lx1 : hx1;
            const uint32_t ny0 = ngy ? hy0 : ly0, ny1 = ngy ? hy1 : ly1, fy0 = ngy ? ly0 : hy0, fy1 = ngy ? ly1 : hy1;
            const uint32_t nz0 = ngz ? hz0 : lz0, nz1 = ngz ? hz1 : lz1, fz0 = ngz ? lz0 : hz0, fz1 = ngz ? lz1 : hz1;
            uint32_t key[8];
            int nh = 0;
            auto child = [&](int k, float qnx, float qfx, float qny, float qfy, float qnz, float qfz, uint32_t cid) {
              const float tn = fmaxf(fmaxf(fmaf(qnx, nax, nbn), fmaf(qny, nay, nbyn)), fmaf(qnz, naz, nbzn));
              const float tf = fminf(fminf(fmaf(qfx, nax, nbf), fmaf(qfy, nay, nbyf)), fmaf(qfz, naz, nbzf));
              const float tmin = fmaxf(tn - grow, 1e-10f), tmax = fminf(tf + grow, c32);
              const float diff = tmax - tmin;
              const float thresh = fmaf(tmin + fabsf(tmax), 7.152557373046875e-07f, ra32.slack);  // 12u(|tmin|+|tmax|) + slack (ray_aux32_direct)
              const bool hit = !(diff < -thresh) && cid != NONE;  // NaN arithmetic: not certainly missed
              nh += hit ? 1 : 0;
              // sort key: entry distance (non-negative float: integer order = float order), bit 3 = "missed", low three bits = the slot
              key[k] = hit ? ((__float_as_uint(tmin) & ~15u) | (uint32_t)k) : (0x7F800008u | (uint32_t)k);
            };
#define RL_UB(w, i) ((float)(((w) >> (8 * (i))) & 0xFFu))  /* v_cvt_f32_ubyte<i> */
            child(0, RL_UB(nx0, 0), RL_UB(fx0, 0), RL_UB(ny0, 0), RL_UB(fy0, 0), RL_UB(nz0, 0), RL_UB(fz0, 0), ca.x);
            child(1, RL_UB(nx0, 1), RL_UB(fx0, 1), RL_UB(ny0, 1), RL_UB(fy0, 1), RL_UB(nz0, 1), RL_UB(fz0, 1), ca.y);
            child(2, RL_UB(nx0, 2), RL_UB(fx0, 2), RL_UB(ny0, 2), RL_UB(fy0, 2), RL_UB(nz0, 2), RL_UB(fz0, 2), ca.z);
            child(3, RL_UB(nx0, 3), RL_UB(fx0, 3), RL_UB(ny0, 3), RL_UB(fy0, 3), RL_UB(nz0, 3), RL_UB(fz0, 3), ca.w);
            child(4, RL_UB(nx1, 0), RL_UB(fx1, 0), RL_UB(ny1, 0), RL_UB(fy1, 0), RL_UB(nz1, 0), RL_UB(fz1, 0), cb.x);
            child(5, RL_UB(nx1, 1), RL_UB(fx1, 1), RL_UB(ny1, 1), RL_UB(fy1, 1), RL_UB(nz1, 1), RL_UB(fz1, 1), cb.y);
            child(6, RL_UB(nx1, 2), RL_UB(fx1, 2), RL_UB(ny1, 2), RL_UB(fy1, 2), RL_UB(nz1, 2), RL_UB(fz1, 2), cb.z);
            child(7, RL_UB(nx1, 3), RL_UB(fx1, 3), RL_UB(ny1, 3), RL_UB(fy1, 3), RL_UB(nz1, 3), RL_UB(fz1, 3), cb.w);
#undef RL_UB
            // Batcher's odd-even merge sort of the eight keys (19 compare-exchanges, min / max on unsigned words): hits first, nearest first
            auto cex = [&](int i, int j) {
              const uint32_t lo = min(key[i], key[j]), hi = max(key[i], key[j]);
              key[i] = lo, key[j] = hi;
            };
            cex(0, 1), cex(2, 3), cex(4, 5), cex(6, 7), cex(0, 2), cex(1, 3), cex(4, 6), cex(5, 7), cex(1, 2), cex(5, 6);
            cex(0, 4), cex(1, 5), cex(2, 6), cex(3, 7), cex(2, 4), cex(3, 5), cex(1, 2), cex(3, 4), cex(5, 6);
            auto child_of = [&](uint32_t kk) -> uint32_t {  // the child id in slot (kk & 7)
              const uint32_t s = kk & 7u;
              const uint32_t a0 = (s & 1u) ? ca.y : ca.x, a1 = (s & 1u) ? ca.w : ca.z, b0 = (s & 1u) ? cb.y : cb.x, b1 = (s & 1u) ? cb.w : cb.z;
              const uint32_t a = (s & 2u) ? a1 : a0, b = (s & 2u) ? b1 : b0;
              return (s & 4u) ? b : a;
            };
            // the farther hits wait on the stack, farthest first
#pragma unroll
            for (int j = 7; j >= 1; j--)
              if (nh > j) push(child_of(key[j]));
            go(nh ? child_of(key[0]) : pop());
          } else
#endif
          {
            const float c32 = unsafe ? FINF : (float)closest;
            auto missed = [&](float b0, float b1, float b2, float b3, float b4, float b5, float &tmin) {
              float t0x = fmaf(b0, ra32.invx, -ra32.oix), t1x = fmaf(b1, ra32.invx, -ra32.oix);
              float t0y = fmaf(b2, ra32.invy, -ra32.oiy), t1y = fmaf(b3, ra32.invy, -ra32.oiy);
              float t0z = fmaf(b4, ra32.invz, -ra32.oiz), t1z = fmaf(b5, ra32.invz, -ra32.oiz);
              tmin = fmaxf(fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z)) - grow, 1e-10f);
              float tmax = fminf(fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z)) + grow, c32);
              float diff = tmax - tmin;
              float thresh = fmaf(tmin + fabsf(tmax), 7.152557373046875e-07f, ra32.slack);  // 12u(|tmin|+|tmax|) + slack (ray_aux32_direct)
              return diff < -thresh;
            };
            auto push = [&](uint32_t e) {
              if (sp < (uint32_t)SD) s_stack[(size_t)sp * NT + tid] = e, sp++;
              else amb = true;  // more pending children than the stack holds: the reference's order decides
            };
            // the tree's top (breadth first, FastGeneral::top_nodes) sits in LDS, the rest comes through L1 / L2 / Infinity Cache
            const Float4 *nd = pc < top ? (const Float4 *)(s_top + pc * 8u) : (const Float4 *)(nodes + pc);
            // Measured and dropped (round 3, RL_TUNE experiment bits): the node's first 16 bytes alone, then an s_waitcnt, then the other six
            // loads (so that they find the line in L1 instead of pending): cfg 5 -2.5 %, cfg 4 -1 %; the same 112 bytes as fourteen 8-byte
            // loads (twice the L1 accesses): cfg 5 -9.5 %, cfg 4 -3.5 % — the L1 access rate is a second-order cost, not the bound.
            const Float4 lx = nd[0], ly = nd[1], lz = nd[2], hx = nd[3], hy = nd[4], hz = nd[5];
            const uint4 ch = *(const uint4 *)(nd + 6);
            float k0, k1, k2, k3;
            const bool h0 = !missed(lx.x, hx.x, ly.x, hy.x, lz.x, hz.x, k0);  // slot 0 and 1 are never empty
            const bool h1 = !missed(lx.y, hx.y, ly.y, hy.y, lz.y, hz.y, k1) && (!MEDIA || ch.y != NONE);  // (a medium's box node has one child)
            const bool h2 = !missed(lx.z, hx.z, ly.z, hy.z, lz.z, hz.z, k2) && ch.z != NONE;
            const bool h3 = !missed(lx.w, hx.w, ly.w, hy.w, lz.w, hz.w, k3) && ch.w != NONE;
            const int nh = (int)h0 + (int)h1 + (int)h2 + (int)h3;
            k0 = h0 ? k0 : FINF, k1 = h1 ? k1 : FINF, k2 = h2 ? k2 : FINF, k3 = h3 ? k3 : FINF;
            uint32_t c0 = ch.x, c1 = ch.y, c2 = ch.z, c3 = ch.w;
            // entry-distance order (a sorting network: the hits end up first, nearest first); keys carry their hit bit in the
            // lowest mantissa bit so that a hit at +inf (non-finite arithmetic: not certainly missed) still sorts before a miss
            uint32_t u0 = (__float_as_uint(k0) & ~1u) | (h0 ? 0u : 1u), u1 = (__float_as_uint(k1) & ~1u) | (h1 ? 0u : 1u);
            uint32_t u2 = (__float_as_uint(k2) & ~1u) | (h2 ? 0u : 1u), u3 = (__float_as_uint(k3) & ~1u) | (h3 ? 0u : 1u);
            auto cex = [&](uint32_t &ka, uint32_t &kb, uint32_t &ca, uint32_t &cb) {  // keys are non-negative floats: integer order = float order
              const bool sw = kb < ka;
              const uint32_t tk = sw ? kb : ka, tc = sw ? cb : ca;
              kb = sw ? ka : kb, cb = sw ? ca : cb;
              ka = tk, ca = tc;
            };
            cex(u0, u1, c0, c1), cex(u2, u3, c2, c3), cex(u0, u2, c0, c2), cex(u1, u3, c1, c3), cex(u1, u2, c1, c2);
            if (nh >= 4) push(c3);
            if (nh >= 3) push(c2);
            if (nh >= 2) push(c1);
            go(nh ? c0 : pop());
          }
        }
        if (__popcll(__ballot(state == ST_TRAV)) < floor_n) break;
      }
    } else if (pick == ST_LEAF) {
      if (state == ST_LEAF) {
#ifdef RL_FASTG_VERIFY
        c_leaves++;
#endif
        if (MEDIA && (pc & FASTG_MEDIUM)) {  // ConstantMedium::hit (constant_medium.rs:27-80) with ray_t = [1e-10, closest so far]
          const uint32_t k = pc & 0xFFFFu;
          const FastMedium fm = P.fg_media[k];
          const DevOp &mop = ops[fm.pc];
          const rl_medium &m = P.media[mop.a];
          D3 om, dm, o, d;  // the ray in the medium's own scope (its length enters the free path), and in the scope of the boundary's parts
          replay_chain(P, ops, fm.chain, wo, wd, om, dm);
          o = om, d = dm;
          Rec r1, r2;
          r1.t = INF, r1.any = false, r1.pc = 0, r2.t = INF, r2.any = false, r2.pc = 0;
          GenCounters gc{0, 0, 0, 0, 0};
          bool folded = false;  // both boundary hits (constant_medium.rs:28-40) still to be found by the reference's fold over the boundary's ops
          if (fm.shape == 1u) {
            // PUSH* PLANAR+ POP*: a planar's t and its inside test do not depend on the interval, so ONE evaluation of every part gives
            // both hits: boundary.hit(r, universe).t = the smallest valid t, boundary.hit(r, [t1 + 1e-4, inf)).t = the smallest valid t
            // >= t1 + 1e-4 (the fold's `t <= closest` replaces on ties, which changes the part, not t; the POPs touch p and the normal
            // only, and their panic site needs |M^-T n|^2 <= 1e-16, which the builder's norm bound 1e5 on every transform excludes).
            // The three smallest t are kept; a fourth hit with all three within 1e-4 of each other goes the general way.
            if (fm.chain_in != fm.chain) replay_chain(P, ops, fm.chain_in, wo, wd, o, d);
            double ta = INF, tb = INF, tc = INF;
            uint32_t cnt = 0;
#pragma unroll 1
            for (uint32_t i = 0; i < fm.count; i++) {
              const DevPlanar &pl = P.planars[ops[fm.first + i].a];
              const D3 normal = ld3(pl.normal);
              const double denom = dot(normal, d);
              if (fabs(denom) < 1e-8) continue;
              const double t = (pl.d - dot(normal, o)) / denom;
              if (!(-INF <= t && t <= INF)) continue;
              const D3 hp = (o + d * t) - ld3(pl.q);
              const D3 w = ld3(pl.w);
              const double alpha = dot(w, cross(hp, ld3(pl.v))), beta = dot(w, cross(ld3(pl.u), hp));
              const bool in = pl.kind == RL_PLANAR_QUAD ? (0.0 <= alpha && alpha <= 1.0 && 0.0 <= beta && beta <= 1.0) : (0.0 <= alpha && 0.0 <= beta && alpha + beta <= 1.0);
              if (!in) continue;
              cnt++;
              if (t < ta) tc = tb, tb = ta, ta = t;
              else if (t < tb) tc = tb, tb = t;
              else if (t < tc) tc = t;
            }
            const double thr = ta + 1e-4;
            folded = true;
            if (cnt >= 1u) r1.any = true, r1.t = ta;
            if (cnt >= 2u && thr <= tb) r2.any = true, r2.t = tb;
            else if (cnt >= 3u && thr <= tc) r2.any = true, r2.t = tc;
            else if (cnt > 3u) folded = false, r1.any = false, r1.t = INF;
          } else if (fm.shape == 2u) {
            // PUSH* SPHERE POP*: both roots out of one discriminant (sphere.rs:32-75 twice, with its normal check at either hit)
            if (fm.chain_in != fm.chain) replay_chain(P, ops, fm.chain_in, wo, wd, o, d);
            const uint32_t payload = ops[fm.first].a;
            const DevSphere &s = P.spheres[payload & SPH_INDEX];
            const D3 c0 = ld3(s.c0);
            const D3 center = (payload & SPH_MOVING) ? c0 + ld3(s.dc) * time : c0;
            const D3 oc = o - center;
            const double a = len2(d), half_b = dot(oc, d), c = len2(oc) - s.r2;
            const double disc = half_b * half_b - a * c;
            folded = true;
            if (!(disc < 0.0)) {
              const double sq = sqrt(disc);
              const double r_l = (-half_b - sq) / a, r_u = (-half_b + sq) / a;
              auto unit_flag = [&](double t) {
                const D3 outward = ((o + d * t) - center) * s.inv_r;
                const double l2 = len2(outward);
                return !(l2 == 1.0 || fabs(l2 - 1.0) <= 1e-5);
              };
              if (-INF <= r_l && r_l <= INF) r1.any = true, r1.t = r_l;
              else if (-INF <= r_u && r_u <= INF) r1.any = true, r1.t = r_u;
              if (r1.any) {
                gc.flagged += unit_flag(r1.t) ? 1u : 0u;
                const double thr = r1.t + 1e-4;
                if (thr <= r_l && r_l <= INF) r2.any = true, r2.t = r_l;
                else if (thr <= r_u && r_u <= INF) r2.any = true, r2.t = r_u;
                if (r2.any) gc.flagged += unit_flag(r2.t) ? 1u : 0u;
              }
            }
          }
          if (!folded) {
            auto nodraw = []() { return 0.0; };
            general_trace<false, false>(P, ops, fm.pc + 1u, mop.skip - 1u, om, dm, wo, wd, time, -INF, r1, gc, nodraw);  // boundary.hit(r, universe)
            if (r1.any) general_trace<false, false>(P, ops, fm.pc + 1u, mop.skip - 1u, om, dm, wo, wd, time, r1.t + 1e-4, r2, gc, nodraw);
          }
          ray_flags += (uint32_t)gc.flagged;
          if (r1.any && r2.any) {
            double t1 = fmax(r1.t, 1e-10), t2 = fmin(r2.t, closest);
            if (!(t1 >= t2)) {
              t1 = fmax(t1, 0.0);
              const double ray_length = sqrt(len2(dm));
              const double distance_inside_boundary = (t2 - t1) * ray_length;
              const double hit_distance = m.neg_inv_density * log(rng.gen_f64());  // the draw, where the reference's fold makes it
              if (!(hit_distance > distance_inside_boundary)) {
                const double t = t1 + hit_distance / ray_length;
                // (a scatter point within the tie band of the hit it replaces: the reference's comparison chain decides)
                if (best != NONE && fabs(t - closest) <= fast_tie_band(fabs(t) + fabs(closest), ra32.oimax())) amb = true;
                closest = t, best = FASTG_MEDIUM | k;
              }
            }
          }
          go(pop());  // (nothing is pending behind a medium's node: the next stage)
        } else {
        const uint32_t item = pc & ~FASTG_LEAF;
        const FastItem it = items[item];
        const DevSphere isph = P.fg_spheres[item];  // fetched side by side with the item (one round trip, not two)
        D3 o, d;
        replay_chain(P, ops, it.chain, wo, wd, o, d);
        // the tie band's coordinate scale max |o_k / d_k| of the ray the test actually sees (binary32 is plenty for a tolerance)
        float oimax = ra32.oimax();
        if (it.chain != NONE)
          oimax = fmaxf(fmaxf(fabsf((float)o.x * __builtin_amdgcn_rcpf((float)d.x)), fabsf((float)o.y * __builtin_amdgcn_rcpf((float)d.y))),
                        fabsf((float)o.z * __builtin_amdgcn_rcpf((float)d.z)));
        if (!(oimax < FINF)) oimax = FINF;  // NaN (0 * inf) -> every hit of this item counts as a tie
        if (it.kind == 0) fastg_sphere_hit(isph, it.payload, o, d, time, oimax, item, closest, best, amb);
        else fastg_planar_hit(P.planars[it.payload], o, d, oimax, item, closest, best, amb);
        go(pop());
        }
      }
    } else if (pick == ST_FILL) {
      if (state == ST_FILL) {
        rng.top_up();
        state = ST_SHADE;
      }
    } else if (pick == ST_GEN) {
      if (state == ST_GEN) {
        bool active = true;
        if (n >= spp) {
          if (have_pixel) {
            size_t pix = (size_t)pr * W + px;
            double *outp = P.out + pix * 3;
            outp[0] = sum.x, outp[1] = sum.y, outp[2] = sum.z;
            if (P.pos_state) P.pos_state[pix] = rng.pos;
            if (P.tile_cost) atomicAdd(&P.tile_cost[ptile], pix_rays);
            have_pixel = false;
          }
          uint32_t slot = wave_claim(P.work_counter);
          if (slot >= P.n_slots) {
            state = ST_DONE;
            active = false;
          } else {
            uint32_t tile = slot >> 6, in = slot & 63u;
            if (P.tile_order) tile = P.tile_order[tile];
            ptile = tile;
            px = (tile % P.tiles_x) * 8u + (in & 7u);
            pr = (tile / P.tiles_x) * 8u + (in >> 3);
            if (px >= W || pr >= P.nrows) active = false;
            else {
              have_pixel = true;
              n = s_begin;
              pix_rays = 0;
              if (P.resume) {
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
          rng.reset_stream(sample_index * WH + (uint64_t)px * (uint64_t)W + (uint64_t)y);
          D3 p00 = ld3(cam.pixel_00), du = ld3(cam.pixel_du), dv = ld3(cam.pixel_dv);
          D3 pixel_center = (p00 + du * (double)px) + dv * (double)y;
          double sx = -0.5 + rng.gen_f64();
          double sy = -0.5 + rng.gen_f64();
          D3 pixel_sample = pixel_center + (du * sx + dv * sy);
          if (cam.defocus_angle <= 0.0) wo = ld3(cam.lookfrom);
          else {
            double a, b;
            rng.unit_disc(a, b);
            wo = (ld3(cam.lookfrom) + ld3(cam.defocus_disk_u) * a) + ld3(cam.defocus_disk_v) * b;
          }
          wd = pixel_sample - wo;
          time = rng.gen_f64();
          thr = d3(1.0, 1.0, 1.0);
          depth = cam.max_depth;
          if (depth == 0) n++;
          else {
            c_rays++;
            pix_rays++;
            start_ray();
          }
        }
      }
    } else {  // ST_SHADE
      if (state == ST_SHADE) {
        bool path_done = false;
        D3 nd = wd;
        Rec rec;
        rec.t = INF, rec.any = false, rec.pc = 0, rec.mat = 0, rec.u = 0.0, rec.v = 0.0, rec.w = 0.0, rec.uv3 = false, rec.front = true;
        rec.p = d3(0.0, 0.0, 0.0), rec.normal = d3(0.0, 0.0, 0.0);
        auto slow_trace = [&]() {  // the reference's own fold over the whole program
          if (MEDIA) {  // ... media included: the ring goes back to where this ray started, and the fold draws as the reference does
            rng.pos = pos0, rng.blk_lo = pos0 >> 4, rng.nres = 1;
            rng.gen_block(rng.blk_lo);
            GenCounters gc{0, 0, 0, 0, 0};
            auto draw = [&]() { return rng.gen_f64(); };
            general_trace<false, true>(P, ops, 0u, NONE, wo, wd, wo, wd, time, 1e-10, rec, gc, draw);
            c_flag += (uint32_t)gc.flagged;
          } else c_flag += general_slow_trace<TRANS>(P, ops, wo, wd, time, rec);
          c_slow++;
        };
        if (amb) {  // rare: the answer may depend on the visiting order -> the reference's own fold
          slow_trace();
        } else if (MEDIA && best != NONE && (best & FASTG_MEDIUM)) {  // scattered inside a medium: the record of constant_medium.rs:69-78
          c_flag += ray_flags;
          const FastMedium fm = P.fg_media[best & 0xFFFFu];
          D3 o, d;
          replay_chain(P, ops, fm.chain, wo, wd, o, d);
          rec.t = closest, rec.p = o + d * closest, rec.normal = d3(1.0, 0.0, 0.0), rec.u = 0.0, rec.v = 0.0, rec.w = 0.0, rec.uv3 = false;
          rec.front = true, rec.mat = P.media[ops[fm.pc].a].material, rec.pc = fm.pc, rec.any = true;
          uint32_t push_pc = fm.chain;  // the POP chain, innermost first (transform.rs:152-161, translate.rs:18)
#pragma unroll 1
          while (push_pc != NONE) {
            const DevOp &op = ops[push_pc];
            if ((op.code & 0xFFu) == OP_PUSH_TRANSLATE) rec.p = rec.p + ld3(P.translates[op.a].offset);
            else {
              const rl_transform &t = P.transforms[op.a];
              rec.p = mat3_mul(t.m, rec.p);
              D3 wn = mat3_mul(t.inv_t, rec.normal);
              double mm = len2(wn);
              if (approx_eq_eps(mm, 0.0, 1e-16)) c_flag++;
              else rec.normal = normalize(wn);
            }
            push_pc = op.b;
          }
        } else if (best != NONE) {
          if (MEDIA) c_flag += ray_flags;
          bool push_skip = false;
          // the HitRecord of the winner: the same test once more with ray_t.max = its root (same arithmetic -> same root), then the
          // POP chain innermost first, as the reference's recursion unwinds (transform.rs:152-161, translate.rs:18)
          const FastItem it = items[best];
          const DevSphere sp = P.fg_spheres[best];  // item, sphere record and material index side by side: one round trip
          const uint32_t wmat = P.fg_material[best];
          D3 o, d;
          replay_chain(P, ops, it.chain, wo, wd, o, d);
          rec.t = closest;
          bool sensitive = false;
          if (it.kind == 0) {
            sphere_hit_rec(sp, it.payload, wmat, it.op_pc, o, d, time, rec);  // its flag: unreachable (r_safe)
            D3 c0 = ld3(sp.c0);
            D3 center = (it.payload & SPH_MOVING) ? c0 + ld3(sp.dc) * time : c0;
            float oimax = ra32.oimax();
            if (it.chain != NONE)
              oimax = fmaxf(fmaxf(fabsf((float)o.x * __builtin_amdgcn_rcpf((float)d.x)), fabsf((float)o.y * __builtin_amdgcn_rcpf((float)d.y))),
                            fabsf((float)o.z * __builtin_amdgcn_rcpf((float)d.z)));
            if (!(oimax < FINF)) oimax = FINF;
            {  // the winner only: a sphere the reference prunes changes the reference's answer only if it would have won
              D3 oc = o - center;
              double half_b = dot(oc, d), sq = sp.r2 * sp.inv_r * fabs(dot(d, rec.normal)), a = len2(d);
              double other = 2.0 * sq * (double)__builtin_amdgcn_rcpf((float)a);  // the other root is t -+ 2 sqrt(disc) / a
              sensitive = fast_hit_is_order_sensitive(oc, d, closest, sp.r2 * sp.inv_r, half_b, sq, closest, fabs(closest) + other, oimax);
            }
          } else planar_hit_rec(P.planars[it.payload], it.op_pc, o, d, rec);
          if (sensitive) {  // rare: grazing or next to an axis pole -> the reference's own fold decides
            rec.t = INF, rec.any = false;
            if (MEDIA) c_flag -= ray_flags;  // (the fold counts the boundary walks' panic sites itself)
            slow_trace();
            push_skip = true;
          }
          uint32_t push_pc = push_skip ? NONE : it.chain;  // (the slow trace returns a world-space record)
#pragma unroll 1
          while (push_pc != NONE) {
            const DevOp &op = ops[push_pc];
            if ((op.code & 0xFFu) == OP_PUSH_TRANSLATE) rec.p = rec.p + ld3(P.translates[op.a].offset);
            else {
              const rl_transform &t = P.transforms[op.a];
              rec.p = mat3_mul(t.m, rec.p);
              D3 wn = mat3_mul(t.inv_t, rec.normal);
              double m = len2(wn);
              if (approx_eq_eps(m, 0.0, 1e-16)) c_flag++;  // unreachable: build_fast_general bounds the matrices
              else rec.normal = normalize(wn);
            }
            push_pc = op.b;
          }
        }
#ifdef RL_FASTG_VERIFY
        {
          Rec r2;
          r2.t = INF, r2.any = false, r2.pc = 0, r2.mat = 0, r2.u = 0.0, r2.v = 0.0, r2.w = 0.0, r2.uv3 = false, r2.front = true;
          r2.p = d3(0.0, 0.0, 0.0), r2.normal = d3(0.0, 0.0, 0.0);
          general_slow_trace<TRANS>(P, ops, wo, wd, time, r2);
          bool same = rec.any == r2.any && (!rec.any || (rec.t == r2.t && rec.pc == r2.pc && rec.p.x == r2.p.x && rec.normal.y == r2.normal.y));
          if (!same) {
            unsigned k = atomicAdd(&g_vcount, 1u);
            if (k < 64) {
              double *L = g_vlog[k];
              L[0] = wo.x, L[1] = wo.y, L[2] = wo.z, L[3] = wd.x, L[4] = wd.y, L[5] = wd.z, L[6] = time, L[7] = rec.any ? rec.t : -1.0;
              L[8] = (double)rec.pc, L[9] = r2.any ? r2.t : -1.0, L[10] = (double)r2.pc, L[11] = unsafe ? 1.0 : 0.0;
            }
          }
        }
#endif
        D3 p = rec.p;
        if (!rec.any) {
          sum = sum + thr * ld3(cam.background);
          path_done = true;
        } else {
          const DevMaterial &m = P.materials[rec.mat];
          D3 texc = d3(0.0, 0.0, 0.0);
          if (m.kind == RL_MAT_LAMBERTIAN || m.kind == RL_MAT_DIFFUSE_LIGHT || (MEDIA && m.kind == RL_MAT_ISOTROPIC)) {
            double tu, tv;
            rec_uv<TRANS>(rec, tu, tv);
            texc = texture_value<(TRANS ? 2 : 1)>(P, m.texture, tu, tv, rec.p);
          }
          uint32_t kind = m.kind;
          D3 normal = rec.normal;
          if (MEDIA && kind == RL_MAT_ISOTROPIC) {  // material.rs:201-214: Vec3::random_unit_vector, attenuation = texture.value(uv, p)
            nd = rng.unit_sphere();
            thr = thr * texc;
          } else if (kind == RL_MAT_LAMBERTIAN) {
            D3 dir = normal + rng.unit_sphere();
            bool near_zero = approx_eq_eps(dir.x, 0.0, 1e-8) && approx_eq_eps(dir.y, 0.0, 1e-8) && approx_eq_eps(dir.z, 0.0, 1e-8);
            nd = near_zero ? normal : dir;
            thr = thr * texc;
          } else if (kind == RL_MAT_METAL) {
            D3 reflected = wd - normal * (2.0 * dot(wd, normal));
            nd = normalize(reflected) + rng.unit_sphere() * m.fuzz;
            if (!(dot(nd, normal) > 0.0)) path_done = true;
            else thr = thr * ld3(m.albedo);
          } else if (kind == RL_MAT_DIELECTRIC) {
            double ri = rec.front ? 1.0 / m.ior : m.ior;
            double m2 = len2(wd);
            D3 ud;
            if (approx_eq_eps(m2, 0.0, 1e-16)) {
              c_flag++;
              ud = wd;
            } else
              ud = normalize(wd);
            double cos_theta = fmin(dot(-ud, normal), 1.0);
            double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
            bool reflect = ri * sin_theta > 1.0;
            if (!reflect) {
              double q = (1.0 - ri) / (1.0 + ri);
              double r0 = q * q;
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
          } else if (kind == RL_MAT_DIFFUSE_LIGHT) {
            sum = sum + thr * texc;
            path_done = true;
          } else {
            path_done = true;
          }
        }
        if (!path_done) {
          depth--;
          if (depth == 0) path_done = true;
        }
        if (path_done) {
          n++;
          state = ST_GEN;
        } else {
          c_rays++;
          pix_rays++;
          wo = p, wd = nd;
          start_ray();
        }
      }
    }
  }

  unsigned long long v;
  v = wave_sum((unsigned long long)c_rays);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[0], v);
  v = wave_sum((unsigned long long)c_flag);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[6], v);
  v = wave_sum((unsigned long long)c_slow);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[7], v);
#ifdef RL_FASTG_VERIFY
  v = wave_sum(c_steps);
  if ((tid & 63) == 0) atomicAdd(&g_vstats[0], v);
  v = wave_sum(c_leaves);
  if ((tid & 63) == 0) atomicAdd(&g_vstats[1], v);
  v = wave_sum(c_unsafe);
  if ((tid & 63) == 0) atomicAdd(&g_vstats[2], v);
#endif
}

}  // namespace rl
