// RTIOW general kernel: every Hittable the reference composes on this path — spheres, planes / quads /
// triangles (flat/plane.rs, quad.rs, triangle.rs), Translate / Transform instances (translate.rs,
// transform.rs), BVH nodes and plain slices — with Lambertian / Metal / Dielectric / DiffuseLight / Flat
// materials and Solid / Checker / Image textures.  The scene program is read from HBM through L1/L2
// (a 1M-sphere BVH is 64 MB of ops + 64 MB of spheres: Infinity-Cache resident, not LDS-sized), the
// ChaCha block of each lane lives in LDS.  One lane owns one pixel at a time (persistent lanes).
//
// Instances: PUSH ops transform the ray into object space (transform.rs:145-149, translate.rs:15);
// the matching POP transforms the hit found inside back (p, normal; t and face are NOT recomputed,
// transform.rs:152-161) and restores the parent-space ray by replaying the enclosing PUSH chain from
// the world ray — the same arithmetic as the first time, so bit-identical — which keeps traversal
// stackless.
#pragma once
#include "rl_rtiow_wave.h"  // RayAux / aabb_fast (filtered AABB test with exact fallback)

namespace rl {

__device__ __forceinline__ D3 mat3_mul(const double *m, D3 v) {  // matrix.rs:42-60: each row accumulates from 0.0
  double o[3];
#pragma unroll
  for (int n = 0; n < 3; n++) {
    double sum = 0.0;
    sum += m[3 * n + 0] * v.x;
    sum += m[3 * n + 1] * v.y;
    sum += m[3 * n + 2] * v.z;
    o[n] = sum;
  }
  return D3{o[0], o[1], o[2]};
}

struct Rec {  // hittable/mod.rs:24-30 HitRecord + the material id
  D3 p, normal;
  double t, u, v, w;  // uv3: (u, v, w) hold the object-space outward normal of a sphere whose UVs are still to be derived
  uint32_t mat;
  uint32_t pc;  // op where it was found (instance scope test)
  bool front, any, uv3;
};

__device__ __forceinline__ void face_normal(D3 d, D3 outward, D3 &normal, bool &front) {  // hittable/mod.rs:32-38
  front = dot(d, outward) <= 0.0;
  normal = front ? outward : -outward;
}

// Sphere::hit with the full record (sphere.rs:32-75); returns the from_normalized flag
__device__ __forceinline__ bool sphere_hit_rec(const DevSphere &s, uint32_t payload, uint32_t mat, uint32_t pc, D3 o, D3 d, double time, Rec &r,
                                               double tmin = 1e-10) {
  D3 c0 = ld3(s.c0);
  D3 center = (payload & SPH_MOVING) ? c0 + ld3(s.dc) * time : c0;
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
  if (tmin <= r_l && r_l <= r.t) t = r_l;
  else if (tmin <= r_u && r_u <= r.t) t = r_u;
  else return false;
  D3 p = o + d * t;
  D3 outward = (p - center) * s.inv_r;
  double l2 = len2(outward);
  r.t = t, r.p = p, r.u = 0.0, r.v = 0.0, r.w = 0.0, r.mat = mat, r.pc = pc, r.any = true;
  // get_sphere_uv (sphere.rs:70) is needed only by spheres whose texture tree samples an Image, and only for the hit that
  // wins: keep its argument and evaluate acos / atan2 in SHADE (rec_uv) — in LEAF they cost ~70 VGPRs of pressure
  r.uv3 = (payload & SPH_UV) != 0u;
  if (r.uv3) r.u = outward.x, r.v = outward.y, r.w = outward.z;
  face_normal(d, outward, r.normal, r.front);
  return !(l2 == 1.0 || fabs(l2 - 1.0) <= 1e-5);
}

// Plane::hit_ab + Plane/Quad/Triangle::hit (plane.rs:51-100, quad.rs:37-42, triangle.rs:60-95); returns flag
__device__ __forceinline__ bool planar_hit_rec(const DevPlanar &pl, uint32_t pc, D3 o, D3 d, Rec &r, double tmin = 1e-10) {
  D3 normal = ld3(pl.normal);
  double denom = dot(normal, d);
  if (fabs(denom) < 1e-8) return false;
  double t = (pl.d - dot(normal, o)) / denom;
  if (!(tmin <= t && t <= r.t)) return false;
  D3 p = o + d * t;
  D3 hp = p - ld3(pl.q);
  D3 w = ld3(pl.w);
  double alpha = dot(w, cross(hp, ld3(pl.v)));
  double beta = dot(w, cross(ld3(pl.u), hp));
  bool flag = false;
  double uu = alpha, vv = beta;
  D3 n = normal;
  if (pl.kind == RL_PLANAR_QUAD) {
    if (!(0.0 <= alpha && alpha <= 1.0 && 0.0 <= beta && beta <= 1.0)) return false;
  } else if (pl.kind == RL_PLANAR_TRIANGLE) {
    if (!(0.0 <= alpha && 0.0 <= beta && alpha + beta <= 1.0)) return false;
    double frac2 = alpha, frac3 = beta, frac1 = 1.0 - alpha - beta;
    if (pl.has_normals) {
      D3 nn = (ld3(pl.normals + 3) * frac2 + ld3(pl.normals + 6) * frac3) + ld3(pl.normals) * frac1;
      double m = len2(nn);
      if (approx_eq_eps(m, 0.0, 1e-16)) flag = true;  // NormalizedVec3::try_from(..).unwrap() would panic
      else n = normalize(nn);
    }
    if (pl.has_uvs) {
      uu = pl.uvs[0] * frac1 + pl.uvs[2] * frac2 + pl.uvs[4] * frac3;
      vv = pl.uvs[1] * frac1 + pl.uvs[3] * frac2 + pl.uvs[5] * frac3;
    }
  }
  r.t = t, r.p = p, r.u = uu, r.v = vv, r.uv3 = false, r.mat = pl.material, r.pc = pc, r.any = true;
  face_normal(d, n, r.normal, r.front);
  return flag;
}

// the (u, v) the winning hit's texture lookup sees
template <bool SPH_UVS = true>
__device__ __forceinline__ void rec_uv(const Rec &r, double &u, double &v) {
  u = r.u, v = r.v;
  if (SPH_UVS && r.uv3) sphere_uv(d3(r.u, r.v, r.w), u, v);
}

// transform the world ray through the chain of PUSH ops that ends at `push_pc` (NONE: identity)
__device__ __forceinline__ void replay_chain(const RtiowParams &P, const DevOp *ops, uint32_t push_pc, D3 wo, D3 wd, D3 &o, D3 &d) {
  uint32_t stack[8];
  int n = 0;
  while (push_pc != NONE && n < 8) {
    stack[n++] = push_pc;
    push_pc = ops[push_pc].b;
  }
  o = wo, d = wd;
  for (int i = n - 1; i >= 0; i--) {
    const DevOp &op = ops[stack[i]];
    if ((op.code & 0xFFu) == OP_PUSH_TRANSLATE) o = o - ld3(P.translates[op.a].offset);
    else {
      const rl_transform &t = P.transforms[op.a];
      D3 no = mat3_mul(t.inv, o), nd = mat3_mul(t.inv, d);
      o = no, d = nd;
    }
  }
}

struct GenCounters {
  unsigned long long nodes, spheres, planars, instances, flagged;
};

// The reference's fold over ops [pc, pc_end) (pc_end == NONE: up to OP_END): bvh.rs:79-95, hittable/mod.rs:88-111,
// transform.rs:143-164, translate.rs:14-21 with ray_t = [tmin, rec.t]; (o, d) = the ray in the scope the range starts in,
// (wo, wd) = the world ray (POP ops restore the parent-scope ray by replaying the PUSH chain from it).
// MEDIA: ConstantMedium ops are evaluated (constant_medium.rs:27-80, deterministic variant of include/rl_render.h rl_medium:
// `draw()` supplies gen::<f64>() from the pixel's ChaCha8 stream); their boundaries run through this function with MEDIA = false.
template <bool STATS, bool MEDIA, class Draw>
__device__ __forceinline__ void general_trace(const RtiowParams &P, const DevOp *ops, uint32_t pc, uint32_t pc_end, D3 o, D3 d, D3 wo, D3 wd, double time,
                                              double tmin, Rec &rec, GenCounters &gc, Draw &draw) {
  const double INF = __longlong_as_double(0x7FF0000000000000ll);
  RayAux ra = ray_aux(o, d);
#pragma unroll 1
  for (;;) {
    if (pc == pc_end) break;
    const DevOp &op = ops[pc];
    uint32_t code = op.code & 0xFFu;
    if (code == OP_END) break;
    if (code == OP_BOX || code == OP_BOX_SPH || code == OP_BOX_PLANAR) {
      if (STATS) gc.nodes++;
      double bx[6] = {op.box[0], op.box[1], op.box[2], op.box[3], op.box[4], op.box[5]};
      bool hitb;
      if (tmin == 1e-10) {  // the filtered test is derived for the world interval [1e-10, closest]
        bool certain;
        hitb = aabb_fast(bx, ra, rec.t, certain);
        if (!(certain && ra.fast_ok && (op.code & BOX_FINITE))) hitb = aabb_hit(bx, o, d, 1e-10, rec.t);  // rare: the reference's divisions
      } else hitb = aabb_hit(bx, o, d, tmin, rec.t);
      if (!hitb) {
        pc = op.skip;
        continue;
      }
      if (code == OP_BOX) {
        pc++;
        continue;
      }
      uint32_t a = op.a, b = op.b;
      if (code == OP_BOX_SPH) {
        if (STATS) gc.spheres++;
        uint32_t ai = a & SPH_INDEX;
        if (sphere_hit_rec(P.spheres[ai], a, P.sphere_material[ai], pc, o, d, time, rec, tmin)) gc.flagged++;
        if (b != NONE) {
          if (STATS) gc.spheres++;
          uint32_t bi = b & SPH_INDEX;
          if (sphere_hit_rec(P.spheres[bi], b, P.sphere_material[bi], pc, o, d, time, rec, tmin)) gc.flagged++;
        }
      } else {
        if (STATS) gc.planars++;
        if (planar_hit_rec(P.planars[a], pc, o, d, rec, tmin)) gc.flagged++;
        if (b != NONE) {
          if (STATS) gc.planars++;
          if (planar_hit_rec(P.planars[b], pc, o, d, rec, tmin)) gc.flagged++;
        }
      }
      pc = op.skip;
      continue;
    }
    if (code == OP_SPHERE) {
      if (STATS) gc.spheres++;
      uint32_t a = op.a, ai = a & SPH_INDEX;
      if (sphere_hit_rec(P.spheres[ai], a, P.sphere_material[ai], pc, o, d, time, rec, tmin)) gc.flagged++;
      pc++;
      continue;
    }
    if (code == OP_PLANAR) {
      if (STATS) gc.planars++;
      if (planar_hit_rec(P.planars[op.a], pc, o, d, rec, tmin)) gc.flagged++;
      pc++;
      continue;
    }
    if (code == OP_PUSH_TRANSLATE) {  // translate.rs:15
      if (STATS) gc.instances++;
      o = o - ld3(P.translates[op.a].offset);
      ra = ray_aux(o, d);
      pc++;
      continue;
    }
    if (code == OP_PUSH_TRANSFORM) {  // transform.rs:145-149
      if (STATS) gc.instances++;
      const rl_transform &t = P.transforms[op.a];
      D3 no = mat3_mul(t.inv, o), nd = mat3_mul(t.inv, d);
      o = no, d = nd;
      ra = ray_aux(o, d);
      pc++;
      continue;
    }
    if (code == OP_MEDIUM_BEGIN) {
      if (MEDIA) {  // constant_medium.rs:27-80
        const rl_medium &m = P.media[op.a];
        const uint32_t b0 = pc + 1, b1 = op.skip - 1;  // the boundary's ops; ops[b1] = OP_MEDIUM_END
        Rec r1, r2;
        r1.t = INF, r1.any = false, r1.pc = 0, r2.t = INF, r2.any = false, r2.pc = 0;
        general_trace<STATS, false>(P, ops, b0, b1, o, d, wo, wd, time, -INF, r1, gc, draw);  // boundary.hit(r, universe)
        if (r1.any) general_trace<STATS, false>(P, ops, b0, b1, o, d, wo, wd, time, r1.t + 1e-4, r2, gc, draw);
        if (r1.any && r2.any) {
          double t1 = fmax(r1.t, tmin), t2 = fmin(r2.t, rec.t);
          if (!(t1 >= t2)) {
            t1 = fmax(t1, 0.0);
            double ray_length = sqrt(len2(d));
            double distance_inside_boundary = (t2 - t1) * ray_length;
            double hit_distance = m.neg_inv_density * log(draw());
            if (!(hit_distance > distance_inside_boundary)) {
              double t = t1 + hit_distance / ray_length;
              rec.t = t, rec.p = o + d * t, rec.normal = d3(1.0, 0.0, 0.0), rec.u = 0.0, rec.v = 0.0, rec.w = 0.0, rec.uv3 = false;
              rec.front = true, rec.mat = m.material, rec.pc = pc, rec.any = true;
            }
          }
        }
      }
      pc = op.skip;
      continue;
    }
    // POP: op.b = pc of the matching PUSH, whose .b is the parent PUSH
    uint32_t push_pc = op.b;
    if (rec.any && rec.pc > push_pc) {  // the current closest hit was found inside this instance
      if (code == OP_POP_TRANSLATE) rec.p = rec.p + ld3(P.translates[op.a].offset);  // translate.rs:18
      else {                                                                              // transform.rs:152-161
        const rl_transform &t = P.transforms[op.a];
        rec.p = mat3_mul(t.m, rec.p);
        D3 wn = mat3_mul(t.inv_t, rec.normal);
        double m = len2(wn);
        if (approx_eq_eps(m, 0.0, 1e-16)) gc.flagged++;  // "Instance normal couldn't be normalized"
        else rec.normal = normalize(wn);
      }
    }
    replay_chain(P, ops, ops[push_pc].b, wo, wd, o, d);
    ra = ray_aux(o, d);
    pc++;
  }
}

// REGS_FOR: the workgroup size the register budget is computed for (NT: one wave per SIMD, 2 NT: two, 3 NT: three)
template <int NT, bool STATS, int REGS_FOR = NT>
__global__ void RL_KERNEL_ALIGN __launch_bounds__(REGS_FOR) rtiow_general_kernel(RtiowParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  unsigned long long *s_rng = (unsigned long long *)smem;  // [8][NT]
  const DevOp *ops = P.ops;
  RngCtx<NT> rc{P.key, s_rng, tid};
  const rl_rtiow_camera &cam = P.cam;
  const uint32_t W = cam.image_width;
  const uint64_t WH = (uint64_t)cam.image_width * (uint64_t)cam.image_height;
  const D3 p00 = ld3(cam.pixel_00), du = ld3(cam.pixel_du), dv = ld3(cam.pixel_dv);
  const D3 lookfrom = ld3(cam.lookfrom), ddu = ld3(cam.defocus_disk_u), ddv = ld3(cam.defocus_disk_v);
  const D3 background = ld3(cam.background);
  const double INF = __longlong_as_double(0x7FF0000000000000ll);
  unsigned long long c_rays = 0, c_nodes = 0, c_sph = 0, c_pl = 0, c_inst = 0, c_flag = 0, c_words = 0;

  for (;;) {
    uint32_t slot = wave_claim(P.work_counter);
    if (slot >= P.n_slots) break;
    uint32_t tile = slot >> 6, in = slot & 63u;
    uint32_t x = (tile % P.tiles_x) * 8u + (in & 7u);
    uint32_t r = (tile / P.tiles_x) * 8u + (in >> 3);
    if (x >= W || r >= P.nrows) continue;
    uint32_t y = P.row_first + r * P.row_step;
    Rng rng{0ull, 0u, 0xFFFFFFFFu};
    D3 sum = d3(0.0, 0.0, 0.0);
    for (uint32_t n = 0; n < cam.samples_per_pixel; n++) {
      uint64_t sample_index = (uint64_t)n + P.first_sample;
      rng.stream = sample_index * WH + (uint64_t)x * (uint64_t)W + (uint64_t)y;
      rng.buf_ctr = 0xFFFFFFFFu;
      D3 pixel_center = (p00 + du * (double)x) + dv * (double)y;
      double px = -0.5 + rc.gen_f64(rng);
      double py = -0.5 + rc.gen_f64(rng);
      D3 pixel_sample = pixel_center + (du * px + dv * py);
      D3 wo;
      if (cam.defocus_angle <= 0.0) wo = lookfrom;
      else {
        double a, b;
        rc.unit_disc(rng, a, b);
        wo = (lookfrom + ddu * a) + ddv * b;
      }
      D3 wd = pixel_sample - wo;
      double time = rc.gen_f64(rng);
      D3 thr = d3(1.0, 1.0, 1.0);
      D3 color = d3(0.0, 0.0, 0.0);
      for (uint32_t depth = cam.max_depth; depth > 0; depth--) {
        c_rays++;
        Rec rec;
        rec.t = INF, rec.any = false, rec.pc = 0, rec.mat = 0, rec.u = 0.0, rec.v = 0.0, rec.w = 0.0, rec.uv3 = false, rec.front = true;
        rec.p = d3(0.0, 0.0, 0.0), rec.normal = d3(0.0, 0.0, 0.0);
        {
          GenCounters gc{0, 0, 0, 0, 0};
          auto draw = [&]() { return rc.gen_f64(rng); };
          general_trace<STATS, true>(P, ops, 0u, NONE, wo, wd, wo, wd, time, 1e-10, rec, gc, draw);
          c_nodes += gc.nodes, c_sph += gc.spheres, c_pl += gc.planars, c_inst += gc.instances, c_flag += gc.flagged;
        }
        if (!rec.any) {
          color = color + thr * background;
          break;
        }
        const DevMaterial &m = P.materials[rec.mat];
          // texture first (it draws no random numbers): the transcendental code in here (acos / atan2 for sphere UVs, sin and
          // Perlin for Noise) is register-hungry, so it runs before the scatter temporaries are live
          D3 texc = d3(0.0, 0.0, 0.0);
          if (m.kind == RL_MAT_LAMBERTIAN || m.kind == RL_MAT_DIFFUSE_LIGHT) {
            double tu, tv;
            rec_uv(rec, tu, tv);
            texc = texture_value<2>(P, m.texture, tu, tv, rec.p);
          }
        uint32_t kind = m.kind;
        D3 normal = rec.normal, p = rec.p;
        D3 nd;
        if (kind == RL_MAT_ISOTROPIC) {  // material.rs:201-214: Vec3::random_unit_vector, attenuation = texture.value(uv, p)
          nd = rc.unit_sphere(rng);
          thr = thr * texture_value<2>(P, m.texture, rec.u, rec.v, rec.p);
        } else if (kind == RL_MAT_LAMBERTIAN) {
          D3 dir = normal + rc.unit_sphere(rng);
          bool near_zero = approx_eq_eps(dir.x, 0.0, 1e-8) && approx_eq_eps(dir.y, 0.0, 1e-8) && approx_eq_eps(dir.z, 0.0, 1e-8);
          nd = near_zero ? normal : dir;
          thr = thr * texc;
        } else if (kind == RL_MAT_METAL) {
          D3 reflected = wd - normal * (2.0 * dot(wd, normal));
          nd = normalize(reflected) + rc.unit_sphere(rng) * m.fuzz;
          if (!(dot(nd, normal) > 0.0)) break;
          thr = thr * ld3(m.albedo);
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
            reflect = refl > rc.gen_f64(rng);
          }
          if (reflect) nd = ud - normal * (2.0 * dot(ud, normal));
          else {
            D3 perp = (ud + normal * cos_theta) * ri;
            D3 par = normal * (-sqrt(fabs(1.0 - len2(perp))));
            nd = perp + par;
          }
        } else if (kind == RL_MAT_DIFFUSE_LIGHT) {
          color = color + thr * texc;
          break;
        } else {
          break;
        }
        wo = p;
        wd = nd;
      }
      sum = sum + color;
    }
    c_words += rng.pos;
    double *outp = P.out + ((size_t)r * W + x) * 3;
    outp[0] = sum.x, outp[1] = sum.y, outp[2] = sum.z;
  }
  unsigned long long v;
  v = wave_sum(c_rays);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[0], v);
  v = wave_sum(c_flag);
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
