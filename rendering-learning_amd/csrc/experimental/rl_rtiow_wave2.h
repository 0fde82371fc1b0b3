// RTIOW sphere/BVH kernel, two pixel contexts per lane ("v6").
//
// Same arithmetic, same per-pixel / per-sample / per-ray order and the same wave scheduler as rl_rtiow_wave.h; the
// difference is the supply of work per lane.  With ONE pixel per lane a state block (TRAV, LEAF, SHADE, GEN, FILL)
// runs with ~40 % of the wave's lanes, because the 64 pixels of a wave spread over five states.  Here every lane
// owns TWO independent pixels (contexts A and B, both in registers); the scheduler counts a lane for state X when
// EITHER of its contexts is in X, lanes whose parked context is the one in X exchange A <-> B (v_swap_b32 under the
// exec mask) and the block then runs on A.  The population of a block rises from p to 1 - (1 - p)^2.
//
// Registers pay for it (one workgroup of 512 lanes = 2 waves per SIMD instead of 3), and so does LDS: 1024 contexts
// cannot keep the two-block ChaCha ring of rl_rtiow_wave.h, so each context holds ONE block (8 u64) and every consumer
// of random numbers is RESTARTABLE: a draw that finds the block exhausted sends the context to FILL and the block of
// code that wanted it runs again from its top when the next block is resident (nothing is consumed twice: the word
// position lives in the context; the first member of a Marsaglia pair that straddles two blocks is kept in `x1`).
// GEN is split into sub-steps (stream + block, sx, sy, defocus disc, time) for the same reason.
//
// Pixels, counters and RNG word positions are bit-identical to the other kernels (tests assert it).
//
// MEASURED (BASELINE configs[1], 256 spp): block populations rise as intended (TRAV 25.9 -> 30.8, LEAF 29.8 -> 39.1,
// SHADE 26.8 -> 33.3, GEN 23.3 -> 33.6 lanes of 64) but the kernel is SLOWER than rl_rtiow_wave.h: 3.09 Grays/s at 512
// lanes (two waves per SIMD are latency-bound: a block costs the same cycles as with three), 3.42 Grays/s forced into
// 168 VGPRs at 768 lanes (spills; the exchange before every block costs ~120 v_cndmask), against 3.78 Grays/s.  The
// one-context form of the same restartable RNG (TWO = false) is slower too (3.33): every bail-out is an extra visit of
// GEN / SHADE, which lowers all populations.  Kept selectable (RL_RTIOW_KERNEL=wave2) as the measured alternative.
#pragma once
#include "../rl_rtiow_wave.h"

namespace rl {

struct Ctx2 {
  uint32_t state;      // ST_*
  uint32_t flags;      // bits 0-2 GEN sub-step, bit 3 have_pixel, bit 4 have_x1, bit 5 FILL returns to GEN (else SHADE)
  uint32_t px, pr, n, ptile, pix_rays, depth;
  uint32_t pc, hit_prim;
  uint32_t pos, blk, rid;  // ChaCha: u32 word position, resident block (NONE = none), ring column
  uint64_t stream;
  D3 sum, o, d, thr;
  D3 inv, oi;
  double slack, time, closest, x1;
};
enum : uint32_t { F_GSUB = 7u, F_HAVE_PIXEL = 8u, F_HAVE_X1 = 16u, F_FILL_GEN = 32u };

#define RL_SWAP32(a, b)      \
  {                          \
    uint32_t t_ = a;         \
    a = b;                   \
    b = t_;                  \
  }
#define RL_SWAP64(a, b)      \
  {                          \
    double t_ = a;           \
    a = b;                   \
    b = t_;                  \
  }
#define RL_SWAPD3(a, b) RL_SWAP64(a.x, b.x) RL_SWAP64(a.y, b.y) RL_SWAP64(a.z, b.z)

__device__ __forceinline__ void ctx2_swap(Ctx2 &A, Ctx2 &B) {
  RL_SWAP32(A.state, B.state) RL_SWAP32(A.flags, B.flags) RL_SWAP32(A.px, B.px) RL_SWAP32(A.pr, B.pr) RL_SWAP32(A.n, B.n)
  RL_SWAP32(A.ptile, B.ptile) RL_SWAP32(A.pix_rays, B.pix_rays) RL_SWAP32(A.depth, B.depth) RL_SWAP32(A.pc, B.pc)
  RL_SWAP32(A.hit_prim, B.hit_prim) RL_SWAP32(A.pos, B.pos) RL_SWAP32(A.blk, B.blk) RL_SWAP32(A.rid, B.rid)
  {
    uint64_t t_ = A.stream;
    A.stream = B.stream;
    B.stream = t_;
  }
  RL_SWAPD3(A.sum, B.sum) RL_SWAPD3(A.o, B.o) RL_SWAPD3(A.d, B.d) RL_SWAPD3(A.thr, B.thr) RL_SWAPD3(A.inv, B.inv) RL_SWAPD3(A.oi, B.oi)
  RL_SWAP64(A.slack, B.slack) RL_SWAP64(A.time, B.time) RL_SWAP64(A.closest, B.closest) RL_SWAP64(A.x1, B.x1)
}

template <int NT, bool TWO, bool STATS>
__global__ void __launch_bounds__(NT) rtiow_wave2_kernel(RtiowParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  constexpr int NC = TWO ? 2 * NT : NT;  // contexts per workgroup (TWO = false: one context per lane, same restartable RNG)
  // LDS: [linked ops][spheres][one ChaCha block per context: 8 x NC u64, slot-major]
  const size_t scene_lds = (size_t)P.n_ops * sizeof(DevOp) + (size_t)P.n_spheres * sizeof(DevSphere);
  unsigned long long *s_rng = (unsigned long long *)(smem + scene_lds);
  const unsigned char *opbase = smem;
  const DevSphere *spheres;
  {
    DevOp *s_ops = (DevOp *)smem;
    DevSphere *s_sph = (DevSphere *)(s_ops + P.n_ops);
    const uint4 *g = (const uint4 *)P.lops;
    uint4 *l = (uint4 *)s_ops;
    for (uint32_t i = tid; i < P.n_ops * 4u; i += NT) {
      uint4 v = g[i];
      if ((i & 3u) == 3u) {
        v.x = (v.x & 0xE0000000u) | ((v.x & 0x1FFFFFFFu) << 6);
        v.y = (v.y & 0xE0000000u) | ((v.y & 0x1FFFFFFFu) << 6);
      }
      l[i] = v;
    }
    g = (const uint4 *)P.spheres;
    l = (uint4 *)s_sph;
    for (uint32_t i = tid; i < P.n_spheres * 4u; i += NT) l[i] = g[i];
    __syncthreads();
    spheres = s_sph;
  }
  const uint32_t entry0 = (P.entry0 & 0xE0000000u) | ((P.entry0 & 0x1FFFFFFFu) << 6);
  const rl_rtiow_camera &cam = P.cam;
  const uint32_t W = cam.image_width;
  const uint32_t s_begin = P.sample_begin, spp = P.sample_end;
  const uint64_t WH = (uint64_t)cam.image_width * (uint64_t)cam.image_height;
  const double INF = __longlong_as_double(0x7FF0000000000000ll);

  Ctx2 A, B;
  A.state = ST_GEN, A.flags = 0, A.px = A.pr = 0, A.n = spp, A.ptile = 0, A.pix_rays = 0, A.depth = 0, A.pc = 0, A.hit_prim = NONE;
  A.pos = 0, A.blk = NONE, A.rid = (uint32_t)tid, A.stream = 0;
  A.sum = d3(0.0, 0.0, 0.0), A.o = d3(0.0, 0.0, 0.0), A.d = d3(0.0, 0.0, 1.0), A.thr = d3(1.0, 1.0, 1.0);
  A.inv = d3(1.0, 1.0, 1.0), A.oi = d3(0.0, 0.0, 0.0), A.slack = 0.0, A.time = 0.0, A.closest = INF, A.x1 = 0.0;
  B = A;
  B.rid = (uint32_t)tid + NT;
  if (!TWO) B.state = ST_DONE;

  uint32_t c_rays = 0, c_flag = 0;
  unsigned long long c_nodes = 0, c_sph = 0, c_words = 0;
  unsigned long long sc_exec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sc_pop[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sc_cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};

  // ---- RNG helpers on the active context (one resident block; callers check avail() first)
  auto avail = [&]() -> bool { return (A.pos >> 4) == A.blk; };
  auto next_u64 = [&]() -> uint64_t {
    uint64_t v = s_rng[(size_t)((A.pos & 15u) >> 1) * NC + A.rid];
    A.pos += 2;
    return v;
  };
  auto to_f64 = [](uint64_t v) -> double { return (double)(v >> 11) * 0x1.0p-53; };
  auto to_m1_1 = [](uint64_t v) -> double {
    double x = __longlong_as_double((long long)((v >> 12) | 0x3FF0000000000000ull));
    return (x - 1.0) * 2.0 + (-1.0);
  };
  // draws the next (x1, x2) pair of a Marsaglia / disc rejection loop; false = block exhausted (x1 may have been kept)
  auto draw_pair = [&](double &x1, double &x2) -> bool {
    if (!(A.flags & F_HAVE_X1)) {
      if (!avail()) return false;
      A.x1 = to_m1_1(next_u64());
      A.flags |= F_HAVE_X1;
    }
    if (!avail()) return false;
    x1 = A.x1;
    x2 = to_m1_1(next_u64());
    A.flags &= ~F_HAVE_X1;
    return true;
  };
  auto start_ray = [&]() {
    RayAux ra = ray_aux(A.o, A.d);
    A.inv = ra.inv, A.oi = ra.oi, A.slack = ra.fast_ok ? ra.slack : INF;
    A.pc = entry0 & 0x1FFFFFFFu, A.closest = INF, A.hit_prim = NONE;
    A.state = entry0 >> 29;
  };

  for (;;) {
    // a finished traversal goes to SHADE; a context whose block is used up tops it up first
    if (A.state == ST_SHADE && !avail()) A.state = ST_FILL, A.flags &= ~F_FILL_GEN;
    if (TWO && B.state == ST_SHADE && (B.pos >> 4) != B.blk) B.state = ST_FILL, B.flags &= ~F_FILL_GEN;

    int n_trav = __popcll(__ballot(A.state == ST_TRAV || (TWO && B.state == ST_TRAV)));
    int n_leaf = __popcll(__ballot(A.state == ST_LEAF || (TWO && B.state == ST_LEAF)));
    int n_shade = __popcll(__ballot(A.state == ST_SHADE || (TWO && B.state == ST_SHADE)));
    int n_fill = __popcll(__ballot(A.state == ST_FILL || (TWO && B.state == ST_FILL)));
    int n_gen = __popcll(__ballot(A.state == ST_GEN || (TWO && B.state == ST_GEN)));
    if ((n_trav | n_shade | n_fill | n_gen | n_leaf) == 0) break;
    uint32_t pick = ST_TRAV;
    int best = n_trav;
    if (n_leaf > best) pick = ST_LEAF, best = n_leaf;
    if (n_shade > best) pick = ST_SHADE, best = n_shade;
    if (n_fill > best) pick = ST_FILL, best = n_fill;
    if (n_gen > best) pick = ST_GEN, best = n_gen;

    unsigned long long t_begin = 0;
    if (STATS) {
      t_begin = __builtin_readcyclecounter();
      if (pick != ST_TRAV) {
#pragma unroll
        for (int k = 0; k < 6; k++)
          if (pick == (uint32_t)k) sc_exec[k]++, sc_pop[k] += (unsigned)best;
      }
    }
    // bring the context that is in the picked state to the front
    if (TWO && A.state != pick && B.state == pick) ctx2_swap(A, B);

    if (pick == ST_TRAV) {
      int floor_n = (best * (int)P.tune[1]) >> 4;
      for (int it = 0; it < (int)P.tune[0]; it++) {
        if (STATS) {
          int np = __popcll(__ballot(A.state == ST_TRAV));
          sc_exec[ST_TRAV]++, sc_pop[ST_TRAV] += (unsigned)np;
        }
        if (A.state == ST_TRAV) {
          const DevOp &op = *(const DevOp *)(opbase + A.pc);
          double bx[6] = {op.box[0], op.box[1], op.box[2], op.box[3], op.box[4], op.box[5]};
          uint32_t w_hit = op.code, w_miss = op.skip;
          RayAux ra;
          ra.inv = A.inv, ra.oi = A.oi, ra.slack = A.slack;
          bool certain;
          bool hitb = aabb_fast(bx, ra, A.closest, certain, P.k8u);
          if (!certain) hitb = aabb_hit(P.ops[A.pc >> 6].box, A.o, A.d, 1e-10, A.closest);  // rare: exact divisions
          if (STATS) c_nodes++;
          uint32_t w = hitb ? w_hit : w_miss;
          A.pc = w & 0x1FFFFFFFu;
          A.state = w >> 29;
        }
        if (__popcll(__ballot(A.state == ST_TRAV)) < floor_n) break;
      }
    } else if (pick == ST_LEAF) {
      if (A.state == ST_LEAF) {
        const DevOp &op = *(const DevOp *)(opbase + A.pc);
        uint32_t a = op.a, b = op.b, w = op.skip;
        Hit h{A.closest, A.hit_prim};
        if (STATS) c_sph++;
        if (sphere_hit(spheres[a & SPH_INDEX], a, A.o, A.d, A.time, 1e-10, h)) c_flag++;
        if (b != NONE) {
          if (STATS) c_sph++;
          if (sphere_hit(spheres[b & SPH_INDEX], b, A.o, A.d, A.time, 1e-10, h)) c_flag++;
        }
        A.closest = h.t, A.hit_prim = h.prim;
        A.pc = w & 0x1FFFFFFFu;
        A.state = w >> 29;
      }
    } else if (pick == ST_FILL) {
      if (A.state == ST_FILL) {
        uint32_t c = A.pos >> 4;
        chacha8_block_to_lds<NC>(P.key, c, A.stream, s_rng, (int)A.rid);
        A.blk = c;
        A.state = (A.flags & F_FILL_GEN) ? (uint32_t)ST_GEN : (uint32_t)ST_SHADE;
      }
    } else if (pick == ST_GEN) {
      if (A.state == ST_GEN) {
        uint32_t gsub = A.flags & F_GSUB;
        bool active = true;
        if (gsub == 0) {
          if (A.n >= spp) {  // pixel finished (or none yet): write it out, claim the next slot
            if (A.flags & F_HAVE_PIXEL) {
              size_t pix = (size_t)A.pr * W + A.px;
              double *outp = P.out + pix * 3;
              outp[0] = A.sum.x, outp[1] = A.sum.y, outp[2] = A.sum.z;
              if (P.pos_state) P.pos_state[pix] = A.pos;
              if (P.tile_cost) atomicAdd(&P.tile_cost[A.ptile], A.pix_rays);
              if (STATS && !P.tile_cost) c_words += A.pos;
              A.flags &= ~F_HAVE_PIXEL;
            }
            uint32_t slot = wave_claim(P.work_counter);
            if (slot >= P.n_slots) {
              A.state = ST_DONE;
              active = false;
            } else {
              uint32_t tile = slot >> 6, in = slot & 63u;
              if (P.tile_order) tile = P.tile_order[tile];
              A.ptile = tile;
              A.px = (tile % P.tiles_x) * 8u + (in & 7u);
              A.pr = (tile / P.tiles_x) * 8u + (in >> 3);
              if (A.px >= W || A.pr >= P.nrows) active = false;  // slot outside the image: claim again next time
              else {
                A.flags |= F_HAVE_PIXEL;
                A.n = s_begin;
                A.pix_rays = 0;
                if (P.resume) {
                  size_t pix = (size_t)A.pr * W + A.px;
                  const double *inp = P.out + pix * 3;
                  A.sum = d3(inp[0], inp[1], inp[2]);
                  A.pos = P.pos_state[pix];
                } else {
                  A.pos = 0;
                  A.sum = d3(0.0, 0.0, 0.0);
                }
                if (A.n >= spp) active = false;
              }
            }
          }
          if (active) {  // start sample n: new stream, word position kept (camera.rs:167-170); the block holding pos
            uint32_t y = P.row_first + A.pr * P.row_step;
            uint64_t sample_index = (uint64_t)A.n + P.first_sample;
            A.stream = sample_index * WH + (uint64_t)A.px * (uint64_t)W + (uint64_t)y;
            uint32_t c = A.pos >> 4;
            chacha8_block_to_lds<NC>(P.key, c, A.stream, s_rng, (int)A.rid);
            A.blk = c;
            A.flags &= ~F_HAVE_X1;
            gsub = 1;
          }
        }
        // get_ray camera.rs:203-216, one draw (or pair) per sub-step; a sub-step that finds the block used up parks
        // the context in FILL and is retried afterwards
        bool need_fill = false;
        if (active && gsub == 1) {
          if (!avail()) need_fill = true;
          else {
            A.d.x = -0.5 + to_f64(next_u64());  // sx, kept in d.x until sy is known
            gsub = 2;
          }
        }
        if (active && !need_fill && gsub == 2) {
          if (!avail()) need_fill = true;
          else {
            double sx = A.d.x, sy = -0.5 + to_f64(next_u64());
            uint32_t y = P.row_first + A.pr * P.row_step;
            D3 p00 = ld3(cam.pixel_00), du = ld3(cam.pixel_du), dv = ld3(cam.pixel_dv);
            D3 pixel_center = (p00 + du * (double)A.px) + dv * (double)y;
            A.d = pixel_center + (du * sx + dv * sy);  // pixel_sample, kept in d until the origin is known
            gsub = 3;
          }
        }
        if (active && !need_fill && gsub == 3) {
          if (cam.defocus_angle <= 0.0) {
            A.o = ld3(cam.lookfrom);
            A.d = A.d - A.o;
            gsub = 4;
          } else {
            for (;;) {  // rand_distr UnitDisc: accept a*a + b*b <= 1
              double a, b;
              if (!draw_pair(a, b)) {
                need_fill = true;
                break;
              }
              if (a * a + b * b <= 1.0) {
                A.o = (ld3(cam.lookfrom) + ld3(cam.defocus_disk_u) * a) + ld3(cam.defocus_disk_v) * b;
                A.d = A.d - A.o;
                gsub = 4;
                break;
              }
            }
          }
        }
        if (active && !need_fill && gsub == 4) {
          if (!avail()) need_fill = true;
          else {
            A.time = to_f64(next_u64());
            A.thr = d3(1.0, 1.0, 1.0);
            A.depth = cam.max_depth;
            gsub = 0;
            if (A.depth == 0) {  // ray_color(depth 0) = black: the sample contributes (0,0,0)
              A.sum = A.sum + d3(0.0, 0.0, 0.0);
              A.n++;
            } else {
              c_rays++;
              A.pix_rays++;
              start_ray();
            }
          }
        }
        A.flags = (A.flags & ~F_GSUB) | gsub;
        if (need_fill) A.state = ST_FILL, A.flags |= F_FILL_GEN;
      }
    } else {  // ST_SHADE
      if (A.state == ST_SHADE) {
        bool path_done = false, bail = false;
        D3 nd = A.d;
        D3 p = A.o;
        if (A.hit_prim == NONE) {  // miss -> background (camera.rs:257)
          A.sum = A.sum + A.thr * ld3(cam.background);
          path_done = true;
        } else {
          uint32_t si = A.hit_prim & SPH_INDEX;
          const DevSphere &s = spheres[si];
          D3 c0 = ld3(s.c0);
          D3 center = (A.hit_prim & SPH_MOVING) ? c0 + ld3(s.dc) * A.time : c0;
          p = A.o + A.d * A.closest;
          D3 outward = (p - center) * s.inv_r;
          bool front = dot(A.d, outward) <= 0.0;
          D3 normal = front ? outward : -outward;
          const DevMaterial &m = P.materials[P.sphere_material[si]];
          uint32_t kind = m.kind;
          const bool is_lamb = kind == RL_MAT_LAMBERTIAN, is_metal = kind == RL_MAT_METAL, is_diel = kind == RL_MAT_DIELECTRIC;
          D3 us = d3(0.0, 0.0, 0.0);
          if (is_lamb | is_metal) {  // rand_distr UnitSphere (Marsaglia): the first draw of both scatter functions
            for (;;) {
              double x1, x2;
              if (!draw_pair(x1, x2)) {
                bail = true;
                break;
              }
              double sq = x1 * x1 + x2 * x2;
              if (sq >= 1.0) continue;
              double f = 2.0 * sqrt(1.0 - sq);
              us = D3{x1 * f, x2 * f, 1.0 - 2.0 * sq};
              break;
            }
          }
          if (!bail) {
            D3 reflected = A.d - normal * (2.0 * dot(A.d, normal));
            D3 vin = is_metal ? reflected : A.d;
            D3 vn = vin;
            double m2 = len2(vin);
            if (is_metal | is_diel) vn = div_s(vin, sqrt(m2));
            if (is_lamb) {
              D3 dir = normal + us;
              bool near_zero = approx_eq_eps(dir.x, 0.0, 1e-8) && approx_eq_eps(dir.y, 0.0, 1e-8) && approx_eq_eps(dir.z, 0.0, 1e-8);
              nd = near_zero ? normal : dir;
              A.thr = A.thr * texture_value(P, m.texture, 0.0, 0.0, p);
            } else if (is_metal) {
              nd = vn + us * m.fuzz;
              if (!(dot(nd, normal) > 0.0)) path_done = true;  // absorbed
              else A.thr = A.thr * ld3(m.albedo);
            } else if (is_diel) {
              double ri = front ? 1.0 / m.ior : m.ior;
              D3 ud = vn;
              bool zero_len = approx_eq_eps(m2, 0.0, 1e-16);
              if (zero_len) ud = A.d;
              double cos_theta = fmin(dot(-ud, normal), 1.0);
              double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
              bool reflect = ri * sin_theta > 1.0;
              if (!reflect) {
                if (!avail()) bail = true;  // the Schlick draw needs one u64: top the block up and run SHADE again
                else {
                  double q = (1.0 - ri) / (1.0 + ri);
                  double r0 = q * q;
                  double xx = 1.0 - cos_theta;
                  double x2 = xx * xx;
                  double refl = r0 + (1.0 - r0) * (xx * (x2 * x2));
                  reflect = refl > to_f64(next_u64());
                }
              }
              if (!bail) {
                if (zero_len) c_flag++;
                if (reflect) nd = ud - normal * (2.0 * dot(ud, normal));
                else {
                  D3 perp = (ud + normal * cos_theta) * ri;
                  D3 par = normal * (-sqrt(fabs(1.0 - len2(perp))));
                  nd = perp + par;
                }
              }
            } else if (kind == RL_MAT_DIFFUSE_LIGHT) {
              A.sum = A.sum + A.thr * texture_value(P, m.texture, 0.0, 0.0, p);
              path_done = true;
            } else {
              path_done = true;  // Flat
            }
          }
        }
        if (bail) {
          A.state = ST_FILL;
          A.flags &= ~F_FILL_GEN;
        } else {
          if (!path_done) {
            A.depth--;
            if (A.depth == 0) path_done = true;  // ray_color(.., 0) = black
          }
          if (path_done) {
            A.n++;
            A.state = ST_GEN;
          } else {
            c_rays++;
            A.pix_rays++;
            A.o = p;
            A.d = nd;
            start_ray();
          }
        }
      }
    }
    if (STATS) {
      unsigned long long dt = __builtin_readcyclecounter() - t_begin;
#pragma unroll
      for (int k = 0; k < 6; k++)
        if (pick == (uint32_t)k) sc_cyc[k] += dt;
    }
  }
  if (STATS && (tid & 63) == 0) {
    unsigned long long *sched = P.stats + 8;
#pragma unroll
    for (int s = 0; s < 6; s++) {
      atomicAdd(&sched[3 * s], sc_exec[s]);
      atomicAdd(&sched[3 * s + 1], sc_pop[s]);
      atomicAdd(&sched[3 * s + 2], sc_cyc[s]);
    }
  }

  unsigned long long v;
  v = wave_sum((unsigned long long)c_rays);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[0], v);
  v = wave_sum((unsigned long long)c_flag);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[6], v);
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
