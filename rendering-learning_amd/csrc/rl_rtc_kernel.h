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
#include "rl_device.h"

namespace rl {

struct RtcParams {
  const DevOp *ops;
  const DevTri *tris;
  const rl_rtc_transformed *xforms;
  const rl_rtc_material *materials;
  const rl_rtc_light *lights;
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

// One pass over the program. SHADOW=false: closest hit (lowest t >= 0, later wins).
// SHADOW=true: product of transparency over hits with 0 < t < distance (world.rs:104-126).
template <bool SHADOW>
__device__ __forceinline__ double rtc_traverse(const RtcParams &P, const DevOp *ops, const DevTri *tris, D3 wo, D3 wd, double distance, RtcHit &best,
                                               RtcCounters &cnt) {
  D3 o = wo, d = wd;
  double atten = 1.0;
  uint32_t pc = 0;
  for (;;) {
    const DevOp &op = ops[pc];
    uint32_t code = op.code;
    if (code == ROP_END) break;
    if (code == ROP_TRIS) {
      uint32_t first = op.a, count = op.b;
      for (uint32_t k = 0; k < count; k++) {  // triangle.rs:63-101
        const DevTri &t = tris[first + k];
        cnt.tris++;
        D3 e1 = ld3(t.e1), e2 = ld3(t.e2);
        D3 dir_cross_e2 = cross(d, e2);
        double det = dot(e1, dir_cross_e2);
        if (fabs(det) < 1e-8) continue;
        double f = 1.0 / det;
        D3 p1_to_origin = o - ld3(t.p1);
        double u = f * dot(p1_to_origin, dir_cross_e2);
        if (!(0.0 <= u && u <= 1.0)) continue;
        D3 origin_cross_e1 = cross(p1_to_origin, e1);
        double v = f * dot(d, origin_cross_e1);
        if (v < 0.0 || (u + v) > 1.0) continue;
        double tt = f * dot(e2, origin_cross_e1);
        if (SHADOW) {
          if (tt > 0.0 && tt < distance) atten = atten * P.materials[t.material].transparency;
        } else if (tt >= 0.0 && !(best.t < tt)) {  // intersect.rs:159-168: `lowest.t < i.t ? lowest : i`
          best.t = tt;
          best.tri = first + k;
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

template <int NT, bool LDS_SCENE>
__global__ void __launch_bounds__(NT) rtc_kernel(RtcParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const DevOp *ops = P.ops;
  const DevTri *tris = P.tris;
  if (LDS_SCENE) {
    DevOp *s_ops = (DevOp *)smem;
    DevTri *s_tris = (DevTri *)(s_ops + P.n_ops);
    const uint4 *g = (const uint4 *)P.ops;
    uint4 *l = (uint4 *)s_ops;
    for (uint32_t i = tid; i < P.n_ops * 4u; i += NT) l[i] = g[i];
    g = (const uint4 *)P.tris;
    l = (uint4 *)s_tris;
    for (uint32_t i = tid; i < P.n_tris * 10u; i += NT) l[i] = g[i];
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
        rtc_traverse<false>(P, ops, tris, origin, dir, 0.0, best, cnt);
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
              shadow_att = rtc_traverse<true>(P, ops, tris, over_point, sdir, distance, dummy, cnt);
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
