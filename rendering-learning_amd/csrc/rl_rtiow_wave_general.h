// RTIOW all-primitives kernel in wave-scheduled form: the schedule of rl_rtiow_wave.h (per-lane state
// machine, the wave runs the most populated state, path regeneration, two-block ChaCha ring in LDS,
// filtered AABB test) with the primitive coverage of rl_rtiow_general.h (planes / quads / triangles,
// Translate / Transform scopes, Image textures).  The scene program is read from HBM through L1 / L2 /
// Infinity Cache (cfg 4: 2 MB; cfg 5: ~150 MB), LDS holds the RNG rings only.
//
// States: GEN, TRAV (one box op), LEAF (1-2 sphere or planar tests with a full HitRecord), XF (enter /
// leave an instance scope), FILL, SHADE.  Instance scopes stay stackless exactly as in the nested-loop
// kernel: PUSH transforms the ray, POP transforms the hit found inside and restores the parent ray by
// replaying the enclosing PUSH chain from the world ray.
#pragma once
#include "rl_rtiow_general.h"
#include "rl_rtiow_wave.h"

namespace rl {

enum : uint32_t { ST_XF = 6 };

// TRANS: the scene needs the transcendental texture code (Noise, or Image textures on spheres -> get_sphere_uv); without it
// the kernel fits 168 VGPRs (3 waves per SIMD) instead of 256 (2 waves per SIMD)
// MEDIA: the scene holds ConstantMedium hittables (constant_medium.rs:27-80, the deterministic variant of include/rl_render.h rl_medium).
// A medium is a SCOPE of the threaded program, like an instance: OP_MEDIUM_BEGIN parks the closest hit found so far in LDS and walks the
// boundary's ops with ray_t = universe; OP_MEDIUM_END either restarts the walk with ray_t = (t1 + 0.0001, inf) (first pass found a hit)
// or closes the scope: the parked record comes back, and with both boundary hits the free path is drawn from the pixel's stream exactly
// where the reference's traversal evaluates the medium.  Boxes inside the scope are tested with the reference's divisions (the
// filtered test is derived for [1e-10, closest]).  Media do not nest (rl_program.cpp).
static const int MEDIA_SAVE_WORDS = 12;  // parked Rec: t, p, normal, u, v, w, {mat, pc}, flags
template <int NT, bool TRANS, bool STATS, bool MEDIA = false>
__global__ void RL_KERNEL_ALIGN __launch_bounds__(NT) rtiow_wave_general_kernel(RtiowParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  unsigned long long *s_rng = (unsigned long long *)smem;  // [16][NT]
  unsigned long long *s_save = s_rng + (size_t)16 * NT;    // [MEDIA_SAVE_WORDS][NT] (MEDIA only)
  const DevOp *ops = P.ops;
  const rl_rtiow_camera &cam = P.cam;
  const uint32_t W = cam.image_width;
  const uint32_t s_begin = P.sample_begin, spp = P.sample_end;
  const uint64_t WH = (uint64_t)cam.image_width * (uint64_t)cam.image_height;
  const double INF = __longlong_as_double(0x7FF0000000000000ll);

  Ring<NT> rng{P.key, s_rng, tid, 0ull, 0u, 0u, 0u};
  uint32_t state = ST_GEN;
  uint32_t px = 0, pr = 0, n = spp;
  uint32_t ptile = 0, pix_rays = 0;
  bool have_pixel = false;
  D3 sum = d3(0.0, 0.0, 0.0);
  D3 wo = d3(0.0, 0.0, 0.0), wd = d3(0.0, 0.0, 1.0);  // world ray
  D3 o = wo, d = wd;                                    // ray in the current instance scope
  D3 thr = d3(1.0, 1.0, 1.0);
  RayAux ra = ray_aux(o, d);
  double time = 0.0;
  Rec rec;
  rec.t = INF, rec.any = false, rec.pc = 0, rec.mat = 0, rec.u = 0.0, rec.v = 0.0, rec.w = 0.0, rec.uv3 = false, rec.front = true;
  rec.p = d3(0.0, 0.0, 0.0), rec.normal = d3(0.0, 0.0, 0.0);
  uint32_t pc = 0, depth = 0;
  uint32_t c_rays = 0, c_flag = 0;
  unsigned long long c_nodes = 0, c_sph = 0, c_pl = 0, c_inst = 0, c_words = 0;
  // MEDIA: ray_t.min of the current walk (1e-10 outside a medium scope), the scope's pass (0 = none) and the first pass's hit distance
  double tmin = 1e-10, m_t1 = 0.0;
  uint32_t mphase = 0;
  auto fresh_rec = [&]() { rec.t = INF, rec.any = false, rec.pc = 0; };
  auto park_rec = [&]() {
    auto st = [&](int k, unsigned long long v) { s_save[(size_t)k * NT + tid] = v; };
    auto bits = [](double x) { return (unsigned long long)__double_as_longlong(x); };
    st(0, bits(rec.t)), st(1, bits(rec.p.x)), st(2, bits(rec.p.y)), st(3, bits(rec.p.z));
    st(4, bits(rec.normal.x)), st(5, bits(rec.normal.y)), st(6, bits(rec.normal.z));
    st(7, bits(rec.u)), st(8, bits(rec.v)), st(9, bits(rec.w));
    st(10, (unsigned long long)rec.mat | ((unsigned long long)rec.pc << 32));
    st(11, (rec.front ? 1ull : 0ull) | (rec.any ? 2ull : 0ull) | (rec.uv3 ? 4ull : 0ull));
  };
  auto unpark_rec = [&]() {
    auto ld = [&](int k) { return s_save[(size_t)k * NT + tid]; };
    auto dbl = [](unsigned long long x) { return __longlong_as_double((long long)x); };
    rec.t = dbl(ld(0)), rec.p = d3(dbl(ld(1)), dbl(ld(2)), dbl(ld(3)));
    rec.normal = d3(dbl(ld(4)), dbl(ld(5)), dbl(ld(6)));
    rec.u = dbl(ld(7)), rec.v = dbl(ld(8)), rec.w = dbl(ld(9));
    unsigned long long mp = ld(10), fl = ld(11);
    rec.mat = (uint32_t)mp, rec.pc = (uint32_t)(mp >> 32);
    rec.front = (fl & 1ull) != 0ull, rec.any = (fl & 2ull) != 0ull, rec.uv3 = (fl & 4ull) != 0ull;
  };

  for (;;) {
    int n_trav = __popcll(__ballot(state == ST_TRAV));
    int n_shade = __popcll(__ballot(state == ST_SHADE));
    int n_fill = __popcll(__ballot(state == ST_FILL));
    int n_gen = __popcll(__ballot(state == ST_GEN));
    int n_leaf = __popcll(__ballot(state == ST_LEAF));
    int n_xf = __popcll(__ballot(state == ST_XF));
    if ((n_trav | n_shade | n_fill | n_gen | n_leaf | n_xf) == 0) break;
    uint32_t pick = ST_TRAV;
    int best = n_trav;
    if (n_leaf > best) pick = ST_LEAF, best = n_leaf;
    if (n_xf > best) pick = ST_XF, best = n_xf;
    if (n_shade > best) pick = ST_SHADE, best = n_shade;
    if (n_fill > best) pick = ST_FILL, best = n_fill;
    if (n_gen > best) pick = ST_GEN, best = n_gen;

    if (pick == ST_TRAV) {
      int floor_n = (best * (int)P.tune[1]) >> 4;
      for (int it = 0; it < (int)P.tune[0]; it++) {
        if (state == ST_TRAV) {
          const DevOp &op = ops[pc];
          double bx[6] = {op.box[0], op.box[1], op.box[2], op.box[3], op.box[4], op.box[5]};
          uint32_t code = op.code, skip = op.skip;
          uint32_t kind = code & 0xFFu;
          bool is_box = (kind == OP_BOX) | (kind == OP_BOX_SPH) | (kind == OP_BOX_PLANAR);
          bool certain;
          bool hitb = aabb_fast(bx, ra, rec.t, certain);
          if (MEDIA && mphase != 0u) certain = false;  // inside a medium scope ray_t.min is not 1e-10: the reference's own test
          if (is_box && !(certain && ra.fast_ok && (code & BOX_FINITE))) hitb = aabb_hit(bx, o, d, MEDIA ? tmin : 1e-10, rec.t);
          if (STATS) c_nodes += is_box ? 1u : 0u;
          bool leaf_kind = (kind == OP_BOX_SPH) | (kind == OP_BOX_PLANAR);
          bool to_leaf = (kind == OP_SPHERE) | (kind == OP_PLANAR) | (leaf_kind & hitb);
          bool to_xf = kind >= OP_PUSH_TRANSLATE;
          uint32_t npc = (is_box & !hitb) ? skip : ((kind == OP_BOX) ? pc + 1u : pc);
          uint32_t nstate = (kind == OP_END) ? (rng.low() ? ST_FILL : ST_SHADE) : (to_leaf ? ST_LEAF : (to_xf ? ST_XF : ST_TRAV));
          pc = npc;
          state = nstate;
        }
        if (__popcll(__ballot(state == ST_TRAV)) < floor_n) break;
      }
    } else if (pick == ST_LEAF) {
      if (state == ST_LEAF) {
        const DevOp &op = ops[pc];
        uint32_t kind = op.code & 0xFFu;
        uint32_t a = op.a, b = op.b;
        if (kind == OP_BOX_SPH || kind == OP_SPHERE) {
          uint32_t ai = a & SPH_INDEX;
          if (STATS) c_sph++;
          if (sphere_hit_rec(P.spheres[ai], a, P.sphere_material[ai], pc, o, d, time, rec, MEDIA ? tmin : 1e-10)) c_flag++;
          if (b != NONE) {
            uint32_t bi = b & SPH_INDEX;
            if (STATS) c_sph++;
            if (sphere_hit_rec(P.spheres[bi], b, P.sphere_material[bi], pc, o, d, time, rec, MEDIA ? tmin : 1e-10)) c_flag++;
          }
        } else {
          if (STATS) c_pl++;
          if (planar_hit_rec(P.planars[a], pc, o, d, rec, MEDIA ? tmin : 1e-10)) c_flag++;
          if (b != NONE) {
            if (STATS) c_pl++;
            if (planar_hit_rec(P.planars[b], pc, o, d, rec, MEDIA ? tmin : 1e-10)) c_flag++;
          }
        }
        pc = op.skip;
        state = ST_TRAV;
      }
    } else if (pick == ST_XF) {
      if (state == ST_XF) {
        const DevOp &op = ops[pc];
        uint32_t kind = op.code & 0xFFu;
        uint32_t next_pc = pc + 1u;
        if (MEDIA && kind == OP_MEDIUM_BEGIN) {  // boundary.hit(r, universe)
          park_rec();
          fresh_rec();
          tmin = -INF, mphase = 1u;
        } else if (MEDIA && kind == OP_MEDIUM_END) {
          if (mphase == 1u && rec.any) {  // boundary.hit(r, (rec1.t + 0.0001, inf)): the same ops once more
            m_t1 = rec.t;
            fresh_rec();
            tmin = m_t1 + 1e-4, mphase = 2u;
            next_pc = op.b + 1u;
          } else {
            const bool both = mphase == 2u && rec.any;
            const double t_exit = rec.t;
            unpark_rec();
            tmin = 1e-10, mphase = 0u;
            if (both) {  // constant_medium.rs:43-80 with ray_t = [1e-10, closest so far]
              const rl_medium &m = P.media[op.a];
              double t1 = fmax(m_t1, 1e-10), t2 = fmin(t_exit, rec.t);
              if (!(t1 >= t2)) {
                t1 = fmax(t1, 0.0);
                double ray_length = sqrt(len2(d));
                double distance_inside_boundary = (t2 - t1) * ray_length;
                double hit_distance = m.neg_inv_density * log(rng.gen_f64());
                if (!(hit_distance > distance_inside_boundary)) {
                  double t = t1 + hit_distance / ray_length;
                  rec.t = t, rec.p = o + d * t, rec.normal = d3(1.0, 0.0, 0.0), rec.u = 0.0, rec.v = 0.0, rec.w = 0.0, rec.uv3 = false;
                  rec.front = true, rec.mat = m.material, rec.pc = op.b, rec.any = true;
                }
              }
            }
          }
        } else if (kind == OP_PUSH_TRANSLATE) {
          if (STATS) c_inst++;
          o = o - ld3(P.translates[op.a].offset);
        } else if (kind == OP_PUSH_TRANSFORM) {
          if (STATS) c_inst++;
          const rl_transform &t = P.transforms[op.a];
          D3 no = mat3_mul(t.inv, o), nd = mat3_mul(t.inv, d);
          o = no, d = nd;
        } else {  // POP: op.b = the matching PUSH, whose .b is the parent PUSH
          uint32_t push_pc = op.b;
          if (rec.any && rec.pc > push_pc) {
            if (kind == OP_POP_TRANSLATE) rec.p = rec.p + ld3(P.translates[op.a].offset);
            else {
              const rl_transform &t = P.transforms[op.a];
              rec.p = mat3_mul(t.m, rec.p);
              D3 wn = mat3_mul(t.inv_t, rec.normal);
              double m = len2(wn);
              if (approx_eq_eps(m, 0.0, 1e-16)) c_flag++;
              else rec.normal = normalize(wn);
            }
          }
          replay_chain(P, ops, ops[push_pc].b, wo, wd, o, d);
        }
        ra = ray_aux(o, d);
        pc = next_pc;
        state = ST_TRAV;
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
            if (STATS && !P.tile_cost) c_words += rng.pos;
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
            o = wo, d = wd;
            ra = ray_aux(o, d);
            pc = 0, rec.t = INF, rec.any = false;
            state = ST_TRAV;
          }
        }
      }
    } else {  // ST_SHADE
      if (state == ST_SHADE) {
        bool path_done = false;
        D3 nd = wd;
        D3 p = rec.p;
        if (!rec.any) {
          sum = sum + thr * ld3(cam.background);
          path_done = true;
        } else {
          const DevMaterial &m = P.materials[rec.mat];
          // texture first (it draws no random numbers): the transcendental code in here (acos / atan2 for sphere UVs, sin and
          // Perlin for Noise) is register-hungry, so it runs before the scatter temporaries are live
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
          o = wo, d = wd;
          ra = ray_aux(o, d);
          pc = 0, rec.t = INF, rec.any = false;
          state = ST_TRAV;
        }
      }
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
    v = wave_sum(c_pl);
    if ((tid & 63) == 0) atomicAdd(&P.stats[3], v);
    v = wave_sum(c_inst);
    if ((tid & 63) == 0) atomicAdd(&P.stats[4], v);
    v = wave_sum(c_words);
    if ((tid & 63) == 0) atomicAdd(&P.stats[5], v);
  }
}

}  // namespace rl
