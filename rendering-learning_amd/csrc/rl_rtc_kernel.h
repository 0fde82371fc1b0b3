// RTC hot path on gfx950: Camera::render / rays_for_pixel (scene/camera.rs:63-124), World::color_at
// (world.rs:89-102), World::intersect + hit (world.rs:46-55, intersect.rs:159-172), Transformed /
// Bounded / Group / Triangle intersect (object/{transformed,bounded,group,triangle}.rs),
// prepare_computations (intersect.rs:48-115), shade_hit + shadow_attenuation (world.rs:57-126),
// lighting (material.rs:54-90).
//
// The reference gathers every intersection into a Vec, stable-sorts by t and picks the lowest t >= 0
// with "later wins" on ties.  That selection equals a single pass in evaluation order that replaces
// the best hit whenever t >= 0 and t <= best.t — no list, no sort (the sorted list itself is only
// needed for refraction, which this kernel does not cover: scenes with reflective or transparent
// materials are rejected with RL_E_UNSUPPORTED at scene creation).
// One lane per pixel; the n x n anti-aliasing grid is summed in the reference's order.
#pragma once
#ifndef RL_KERNEL_ALIGN
// Every big kernel starts on a 64 KB boundary of the code object.  Measured (round 3): the stealing instantiation of rtiow_wave_kernel (77 KB of
// code: main loop + cooperative body, more than the 64 KB instruction cache two CUs share) ran the 1/8 shard in 174 or in 189 ms depending on
// nothing but where the linker happened to put it (0x...ad00 against 0x...d600 after unrelated kernels grew); aligned, the placement is fixed.
#define RL_KERNEL_ALIGN __attribute__((aligned(65536)))
#endif
#include "rl_device.h"

namespace rl {

struct RtcParams {
  const DevOp *ops;
  const DevTri *tris;
  const rl_rtc_transformed *xforms;
  const rl_rtc_material *materials;
  const rl_rtc_light *lights;
  const RtcGuard *guards;  // reject-only box trees of the ROP_TRIS ops (op.skip = root + 1), or null
  uint32_t n_guards;
  uint32_t n_ops, n_tris, n_xforms, n_lights;
  rl_rtc_camera cam;
  uint32_t aa;
  uint32_t row_first, row_step, nrows;
  double void_color[3];
  double *out;
  unsigned long long *stats;
};

// 4x4 (row-major) * point (w = 1) / vector (w = 0): 4-term sums accumulated from 0.0
// (math/matrix.rs:192-210, point.rs:89-96, vector.rs:115-122)
__device__ __forceinline__ D3 mul_point(const double *m, D3 p) {
  double o[3];
#pragma unroll
  for (int n = 0; n < 3; n++) {
    double sum = 0.0;
    sum += m[4 * n + 0] * p.x;
    sum += m[4 * n + 1] * p.y;
    sum += m[4 * n + 2] * p.z;
    sum += m[4 * n + 3] * 1.0;
    o[n] = sum;
  }
  return D3{o[0], o[1], o[2]};
}
__device__ __forceinline__ D3 mul_vec(const double *m, D3 p) {
  double o[3];
#pragma unroll
  for (int n = 0; n < 3; n++) {
    double sum = 0.0;
    sum += m[4 * n + 0] * p.x;
    sum += m[4 * n + 1] * p.y;
    sum += m[4 * n + 2] * p.z;
    sum += m[4 * n + 3] * 0.0;
    o[n] = sum;
  }
  return D3{o[0], o[1], o[2]};
}
__device__ __forceinline__ double mag(D3 v) { return sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }  // vector.rs:32
// vector.rs:36-43 / 142-156: TRUE division by the magnitude; false when the magnitude is 0
__device__ __forceinline__ bool norm(D3 v, D3 &out) {
  double m = mag(v);
  if (m == 0.0) return false;
  out = D3{v.x / m, v.y / m, v.z / m};
  return true;
}
__device__ __forceinline__ D3 reflect(D3 v, D3 n) { return v - (n * 2.0) * dot(v, n); }  // vector.rs:57-59

struct RtcCounters {
  unsigned long long rays, nodes, tris, enters, flagged;
};

struct RtcHit {
  double t;
  uint32_t tri;  // NONE = no hit
  uint32_t pc;   // op index where it was found (scope test at ROP_EXIT)
  D3 normal;
};

// ---- reject-only binary32 box test for the guard trees (same error analysis as aabb_fast32 in rl_rtiow_wave.h, u = 2^-24):
// with inv32 = RN32(1/d), oi32 = RN32(o/d) each slab parameter is off by <= 3.2u|t| + 3.3u max|o/d|.  The Group of a mesh has
// no acceleration structure in the reference (every ray tests every triangle, group.rs); a triangle whose padded box the ray
// certainly misses inside the parameter range that can still matter ([0, best.t], shadow rays (0, distance)) certainly has no
// intersection the reference would keep, so its (binary64) test is skipped — the counter still counts it.
struct RtcAux32 {
  float invx, invy, invz, oix, oiy, oiz, slack;
};
__device__ __forceinline__ RtcAux32 rtc_aux32(D3 o, D3 d) {
  RtcAux32 r;
  double ix = 1.0 / d.x, iy = 1.0 / d.y, iz = 1.0 / d.z;
  r.invx = (float)ix, r.invy = (float)iy, r.invz = (float)iz;
  r.oix = (float)(o.x * ix), r.oiy = (float)(o.y * iy), r.oiz = (float)(o.z * iz);
  float m = fmaxf(fmaxf(fabsf(r.oix), fabsf(r.oiy)), fabsf(r.oiz));
  float imin = fminf(fminf(fabsf(r.invx), fabsf(r.invy)), fabsf(r.invz)), imax = fmaxf(fmaxf(fabsf(r.invx), fabsf(r.invy)), fabsf(r.invz));
  bool ok = imin >= 1e-30f && imax <= 1e30f && m <= 1e30f;                        // NaN compares false
  r.slack = ok ? fmaf(m, 9.5367431640625e-07f, 1e-30f) : __int_as_float(0x7F800000);  // 16u max|oi|, or +inf: never reject
  return r;
}
__device__ __forceinline__ bool guard_reject32(const float *b, const RtcAux32 &ra, float tbound) {
  float t0x = fmaf(b[0], ra.invx, -ra.oix), t1x = fmaf(b[1], ra.invx, -ra.oix);
  float t0y = fmaf(b[2], ra.invy, -ra.oiy), t1y = fmaf(b[3], ra.invy, -ra.oiy);
  float t0z = fmaf(b[4], ra.invz, -ra.oiz), t1z = fmaf(b[5], ra.invz, -ra.oiz);
  float tmin = fmaxf(fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z)), 0.0f);
  float tmax = fminf(fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z)), tbound);
  float thresh = fmaf(tmin + fabsf(tmax), 4.76837158203125e-07f, ra.slack);  // 8u(|tmin|+|tmax|) + slack
  return (tmin - tmax) > thresh;                                            // false for NaN / inf arithmetic
}

// One pass over the program. SHADOW=false: closest hit (lowest t >= 0, later wins).
// SHADOW=true: product of transparency over hits with 0 < t < distance (world.rs:104-126).
template <bool SHADOW>
__device__ __forceinline__ double rtc_traverse(const RtcParams &P, const DevOp *ops, const DevTri *tris, const RtcGuard *guards, D3 wo, D3 wd, double distance,
                                               RtcHit &best, RtcCounters &cnt) {
  D3 o = wo, d = wd;
  double atten = 1.0;
  uint32_t pc = 0;
  for (;;) {
    const DevOp &op = ops[pc];
    uint32_t code = op.code;
    if (code == ROP_END) break;
    if (code == ROP_TRIS) {
      uint32_t first = op.a, count = op.b;
      auto test_triangle = [&](uint32_t ti) {  // triangle.rs:63-101
        const DevTri &t = tris[ti];
        D3 e1 = ld3(t.e1), e2 = ld3(t.e2);
        D3 dir_cross_e2 = cross(d, e2);
        double det = dot(e1, dir_cross_e2);
        if (fabs(det) < 1e-8) return;
        double f = 1.0 / det;
        D3 p1_to_origin = o - ld3(t.p1);
        double u = f * dot(p1_to_origin, dir_cross_e2);
        if (!(0.0 <= u && u <= 1.0)) return;
        D3 origin_cross_e1 = cross(p1_to_origin, e1);
        double v = f * dot(d, origin_cross_e1);
        if (v < 0.0 || (u + v) > 1.0) return;
        double tt = f * dot(e2, origin_cross_e1);
        if (SHADOW) {
          if (tt > 0.0 && tt < distance) atten = atten * P.materials[t.material].transparency;
        } else if (tt >= 0.0 && !(best.t < tt)) {  // intersect.rs:159-168: `lowest.t < i.t ? lowest : i`
          best.t = tt;
          best.tri = ti;
          best.pc = pc;
          if (t.smooth) {
            D3 n = (ld3(t.n2) * u + ld3(t.n3) * v) + ld3(t.n1) * (1.0 - u - v);
            D3 nn;
            if (!norm(n, nn)) {
              cnt.flagged++;
              nn = d3(0.0, 0.0, 0.0);
            }
            best.normal = nn;
          } else
            best.normal = ld3(t.n1);
        }
      };
      if (guards && op.skip != 0u) {  // walk the reject-only box tree; its leaves come in triangle order, so ties resolve as in the loop
        cnt.tris += count;             // the reference tests every triangle of the group
        const RtcAux32 ax = rtc_aux32(o, d);
        uint32_t g = op.skip - 1u;
        const uint32_t gend = guards[g].skip;
        while (g < gend) {
          const RtcGuard &nd = guards[g];
          const float bx[6] = {nd.box[0], nd.box[1], nd.box[2], nd.box[3], nd.box[4], nd.box[5]};
          const uint32_t nskip = nd.skip, ntri = nd.tri;
          const float tbound = SHADOW ? (float)distance : (float)best.t;
          if (guard_reject32(bx, ax, tbound)) g = nskip;
          else if (ntri == NONE) g++;
          else {
            test_triangle(ntri);
            g = nskip;
          }
        }
      } else {
        for (uint32_t k = 0; k < count; k++) {
          cnt.tris++;
          test_triangle(first + k);
        }
      }
      pc++;
      continue;
    }
    if (code == ROP_BOUNDS) {  // bounded.rs:100-139: swap if tmin > tmax; pass iff tmin <= tmax; NaN-ignoring max/min
      cnt.nodes++;
      double lo[3], hi[3];
      const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
#pragma unroll
      for (int a = 0; a < 3; a++) {
        double tmin = (op.box[2 * a] - oo[a]) / dd[a];
        double tmax = (op.box[2 * a + 1] - oo[a]) / dd[a];
        bool sw = tmin > tmax;
        lo[a] = sw ? tmax : tmin;
        hi[a] = sw ? tmin : tmax;
      }
      double tmin = fmax(fmax(lo[0], lo[1]), lo[2]);
      double tmax = fmin(fmin(hi[0], hi[1]), hi[2]);
      pc = (tmin <= tmax) ? pc + 1 : op.skip;
      continue;
    }
    if (code == ROP_ENTER) {  // transformed.rs:39-51: local ray = inverse * ray
      cnt.enters++;
      const rl_rtc_transformed &x = P.xforms[op.a];
      D3 no = mul_point(x.inverse, o), nd = mul_vec(x.inverse, d);
      o = no, d = nd;
      pc++;
      continue;
    }
    // ROP_EXIT: a = xform, skip = pc of the matching ENTER, b = offset into the enclosing-chain table (-> restore parent ray)
    {
      if (!SHADOW && best.tri != NONE && best.pc > op.skip) {  // hit found inside this scope: normal <- unit(inverse_transpose * n)
        const rl_rtc_transformed &x = P.xforms[op.a];
        D3 wn = mul_vec(x.inverse_transpose, best.normal), nn;
        if (!norm(wn, nn)) {
          cnt.flagged++;
          nn = best.normal;
        }
        best.normal = nn;
      }
      // parent-space ray: replay the enclosing ENTERs from the world ray (bit-identical to the first pass)
      o = wo, d = wd;
      uint32_t chain = op.b;  // ops index list encoded as a linked list: each ENTER op stores its parent ENTER pc in .b
      // walk outermost -> innermost: collect up to 8 parents
      uint32_t stack[8];
      int n = 0;
      while (chain != NONE && n < 8) {
        stack[n++] = chain;
        chain = ops[chain].b;
      }
      for (int i = n - 1; i >= 0; i--) {
        const rl_rtc_transformed &x = P.xforms[ops[stack[i]].a];
        D3 no = mul_point(x.inverse, o), nd = mul_vec(x.inverse, d);
        o = no, d = nd;
      }
      pc++;
      continue;
    }
  }
  return atten;
}

// REGS_FOR: the workgroup size the register budget is computed for (NT: one wave per SIMD and up to 512 registers, 2 NT / 3 NT / 4 NT: the
// budget of two / three / four waves per SIMD)
template <int NT, bool LDS_SCENE, int REGS_FOR = NT>
__global__ void RL_KERNEL_ALIGN __launch_bounds__(REGS_FOR) rtc_kernel(RtcParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const DevOp *ops = P.ops;
  const DevTri *tris = P.tris;
  const RtcGuard *guards = P.guards;
  if (LDS_SCENE) {
    DevOp *s_ops = (DevOp *)smem;
    DevTri *s_tris = (DevTri *)(s_ops + P.n_ops);
    const uint4 *g = (const uint4 *)P.ops;
    uint4 *l = (uint4 *)s_ops;
    for (uint32_t i = tid; i < P.n_ops * 4u; i += NT) l[i] = g[i];
    g = (const uint4 *)P.tris;
    l = (uint4 *)s_tris;
    for (uint32_t i = tid; i < P.n_tris * 10u; i += NT) l[i] = g[i];
    if (guards) {
      RtcGuard *s_guards = (RtcGuard *)(s_tris + P.n_tris);
      g = (const uint4 *)P.guards;
      l = (uint4 *)s_guards;
      for (uint32_t i = tid; i < P.n_guards * 2u; i += NT) l[i] = g[i];
      guards = s_guards;
    }
    __syncthreads();
    ops = s_ops;
    tris = s_tris;
  }
  const rl_rtc_camera &cam = P.cam;
  const uint32_t W = cam.hsize;
  RtcCounters cnt{0, 0, 0, 0, 0};
  const double INF = __longlong_as_double(0x7FF0000000000000ll);
  const uint64_t total = (uint64_t)W * P.nrows;
  for (uint64_t idx = (uint64_t)blockIdx.x * NT + tid; idx < total; idx += (uint64_t)gridDim.x * NT) {
    uint32_t r = (uint32_t)(idx / W), px = (uint32_t)(idx % W);
    uint32_t py = P.row_first + r * P.row_step;
    D3 acc = d3(0.0, 0.0, 0.0);
    bool have = false;
    for (uint32_t nx = 0; nx < P.aa; nx++)
      for (uint32_t ny = 0; ny < P.aa; ny++) {  // camera.rs:71-88
        double sample_offset = 1.0 / (double)P.aa;
        double xoffset = ((double)px + sample_offset * ((double)nx + 0.5)) * cam.pixel_size;
        double yoffset = ((double)py + sample_offset * ((double)ny + 0.5)) * cam.pixel_size;
        double world_x = cam.half_width - xoffset;
        double world_y = cam.half_height - yoffset;
        D3 pixel = mul_point(cam.inverse, d3(world_x, world_y, -1.0));
        D3 origin = mul_point(cam.inverse, d3(0.0, 0.0, 0.0));
        D3 dir;
        if (!norm(pixel - origin, dir)) {
          cnt.flagged++;
          dir = d3(0.0, 0.0, 0.0);
        }
        // World::color_at -> color_at_internal (world.rs:89-102)
        cnt.rays++;
        RtcHit best{INF, NONE, 0u, d3(0.0, 0.0, 0.0)};
        rtc_traverse<false>(P, ops, tris, guards, origin, dir, 0.0, best, cnt);
        D3 c = ld3(P.void_color);
        if (best.tri != NONE && P.n_lights > 0) {
          // prepare_computations (intersect.rs:48-71)
          const DevTri &tr = tris[best.tri];
          const rl_rtc_material &m = P.materials[tr.material];
          D3 point = origin + dir * best.t;
          D3 eye_v;
          if (!norm(-dir, eye_v)) {
            cnt.flagged++;
            eye_v = -dir;
          }
          D3 normal_v = best.normal;
          if (dot(normal_v, eye_v) < 0.0) normal_v = -normal_v;
          D3 over_point = point + normal_v * 1e-5;
          D3 object_color = ld3(m.color);
          D3 lsum = d3(0.0, 0.0, 0.0);
          for (uint32_t li = 0; li < P.n_lights; li++) {  // shade_hit (world.rs:57-87)
            const rl_rtc_light &light = P.lights[li];
            D3 lpos = ld3(light.position), intensity = ld3(light.intensity);
            // shadow_attenuation (world.rs:104-126)
            D3 v = lpos - over_point;
            double distance = mag(v);
            D3 sdir;
            double shadow_att = 1.0;
            if (norm(v, sdir)) {
              cnt.rays++;
              RtcHit dummy{INF, NONE, 0u, d3(0.0, 0.0, 0.0)};
              shadow_att = rtc_traverse<true>(P, ops, tris, guards, over_point, sdir, distance, dummy, cnt);
            }
            // lighting (material.rs:54-90)
            D3 effective = object_color * intensity;
            D3 lightv;
            if (!norm(lpos - point, lightv)) lightv = d3(0.0, 0.0, 0.0);
            D3 ambient = effective * m.ambient;
            double ldn = dot(lightv, normal_v);
            D3 diffuse = d3(0.0, 0.0, 0.0), specular = d3(0.0, 0.0, 0.0);
            if (!(ldn < 0.0)) {
              D3 diff = (effective * m.diffuse) * ldn;
              D3 reflectv = -reflect(lightv, normal_v);
              double rde = dot(reflectv, eye_v);
              diffuse = diff * shadow_att;
              if (!(rde <= 0.0)) {
                double factor = pow(rde, m.shininess);
                specular = intensity * (m.specular * factor * shadow_att);
              }
            }
            D3 surface = (ambient + diffuse) + specular;
            // reflectivity == transparency == 0 on this path: surface + (black + black)
            D3 col = surface + (d3(0.0, 0.0, 0.0) + d3(0.0, 0.0, 0.0));
            lsum = (li == 0) ? col : lsum + col;
          }
          c = lsum;
        }
        acc = have ? acc + c : c;
        have = true;
      }
    D3 res = acc * (1.0 / (double)((uint64_t)P.aa * P.aa));
    double *outp = P.out + idx * 3;
    outp[0] = res.x, outp[1] = res.y, outp[2] = res.z;
  }
  unsigned long long v;
  v = wave_sum(cnt.rays);
  if ((tid & 63) == 0) atomicAdd(&P.stats[0], v);
  v = wave_sum(cnt.nodes);
  if ((tid & 63) == 0) atomicAdd(&P.stats[1], v);
  v = wave_sum(cnt.tris);
  if ((tid & 63) == 0) atomicAdd(&P.stats[3], v);
  v = wave_sum(cnt.enters);
  if ((tid & 63) == 0) atomicAdd(&P.stats[4], v);
  v = wave_sum(cnt.flagged);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[6], v);
}

}  // namespace rl
