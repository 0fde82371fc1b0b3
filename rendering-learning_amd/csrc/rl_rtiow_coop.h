// Cooperative latency kernel for the per-pixel sample chain: ONE WAVE PER PIXEL.
//
// The samples of a pixel are sequential in the reference (camera.rs:161-174: sample n + 1 starts at the ChaCha word position sample n
// stopped at), so when a frame has few pixels its time is the time of its longest chains — and a chain advances at the latency of one
// ray through the state machine of rl_rtiow_wave.h (9.6 us for a lone lane, 22.6 us inside a full wave).  This kernel — what the library
// renders small frames of sphere scenes with — works differently: all 64 lanes of a wave carry the SAME pixel (same RNG
// stream, same arithmetic, redundantly — there is no divergence and no scheduler), and the one thing that can be spread over the lanes
// is: World::hit.  Each lane tests the reject-only binary32 leaf boxes of eight of the scene's spheres (the boxes of rl_fast_bvh.cpp,
// same certain-miss test as the fast traversal), the candidates are compacted through LDS, one lane per candidate evaluates the
// reference's Sphere::hit, and a wave reduction picks the closest root.  Order-sensitive rays (two roots within the tie band, a
// grazing / pole hit of the winner, a ray outside the binary32 range) are re-traced by fast_slow_trace, the reference's own fold,
// exactly as in the fast traversal — so the pixels are the bits the other kernels produce.
#pragma once
#include "rl_rtiow_wave.h"

namespace rl {

struct CoopParams {
  const uint32_t *pixels;   // [n_pixels] virtual pixel index pr * W + px (pr = row within the shard)
  uint32_t n_pixels;
  const float *leaf_boxes;  // [n_spheres][8]: x.min, x.max, y.min, y.max, z.min, z.max, 0, 0 (padded, rounded outwards)
  uint32_t *counter;        // work counter (zeroed before the launch)
  uint32_t max_cand;        // capacity of the per-wave candidate list in LDS
};

__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    double o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;  // NaN never enters (lanes without a hit hold +inf)
  }
  return v;
}

// The body: NT threads = NT / 64 waves, each wave claims pixels of C.pixels until the list is exhausted.
// LDS: [8][NT] u64 RNG blocks (one per lane, all lanes of a wave hold the same) + [NT / 64][max_cand] u32 candidate lists.
// BOXES_IN_REGS: lane l keeps the leaf boxes of spheres l, l + 64, ... l + 448 in 48 registers (scenes of up to 512 spheres) — the
// lowest latency per ray; without it the boxes come from L2 every ray and the kernel fits the register budget of four waves per SIMD
// The pixel's ChaCha8 words for a whole wave that works on ONE pixel: every lane generates a DIFFERENT block of the current stream — lane l the
// block (base + l) — into its own column of the [8][NT] u64 block rows, so one pass of the block function (the same ~420 instructions that used
// to produce 16 words, identically in all 64 lanes) yields 1024 consecutive words: more than a sample ever draws (a fresh stream per sample,
// camera.rs:167-170, the word position runs on).  A draw is one uniform LDS read from the column of the block that holds the word.
template <int NT>
struct WaveRng {
  const uint32_t *key;
  unsigned long long *s_rng;  // [8][NT]
  int tid;
  uint64_t stream;
  uint32_t pos;   // u32 word position since the pixel started
  uint32_t base;  // block counter held by lane 0 of the wave (blocks base .. base + 63 are resident); 0xFFFFFFFF = none
  template <bool ROLLED>
  __device__ __forceinline__ void fill(uint32_t b) {  // make blocks b .. b + 63 of `stream` resident
    __builtin_amdgcn_wave_barrier();  // every lane has finished reading the previous blocks
    chacha8_block_to_lds<NT, ROLLED>(key, b + (uint32_t)(tid & 63), stream, s_rng, tid);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    base = b;
  }
  __device__ __forceinline__ void set_stream(uint64_t s) {  // (the one unrolled copy of the block function: once per sample)
    stream = s;
    fill<false>(pos >> 4);
  }
  __device__ __forceinline__ uint64_t next_u64() {
    const uint32_t b = pos >> 4;
    if (b - base >= 64u) fill<true>(b);  // more than 1024 words in one sample (or none resident yet): the rolled form, a quarter of the code per call site
    const uint64_t v = s_rng[(size_t)((pos & 15u) >> 1) * NT + (size_t)((tid & ~63) + (int)(b - base))];
    pos += 2;
    return v;
  }
  __device__ __forceinline__ double gen_f64() { return (double)(next_u64() >> 11) * 0x1.0p-53; }
  __device__ __forceinline__ double uniform_m1_1() {
    double v = __longlong_as_double((long long)((next_u64() >> 12) | 0x3FF0000000000000ull));
    return (v - 1.0) * 2.0 + (-1.0);
  }
  __device__ __forceinline__ D3 unit_sphere() {  // rand_distr 0.4.3 UnitSphere (Marsaglia 1972): reject s >= 1
    for (;;) {
      double x1 = uniform_m1_1(), x2 = uniform_m1_1();
      double s = x1 * x1 + x2 * x2;
      if (s >= 1.0) continue;
      double f = 2.0 * sqrt(1.0 - s);
      return D3{x1 * f, x2 * f, 1.0 - 2.0 * s};
    }
  }
  __device__ __forceinline__ void unit_disc(double &a, double &b) {  // rand_distr 0.4.3 UnitDisc: accept s <= 1
    for (;;) {
      a = uniform_m1_1();
      b = uniform_m1_1();
      if (a * a + b * b <= 1.0) return;
    }
  }
};

template <int NT, bool BOXES_IN_REGS, class Source, class Cand>
__device__ __forceinline__ void rtiow_coop_body(const RtiowParams &P, const float *leaf_boxes, uint32_t max_cand, unsigned long long *s_rng, Cand cand_at, Source src) {
  const int tid = threadIdx.x, lane = tid & 63;
  const DevOp *ops = P.ops;
  const DevSphere *spheres = P.spheres;
  const rl_rtiow_camera &cam = P.cam;
  const uint32_t W = cam.image_width;
  const uint64_t WH = (uint64_t)cam.image_width * (uint64_t)cam.image_height;
  const D3 p00 = ld3(cam.pixel_00), du = ld3(cam.pixel_du), dv = ld3(cam.pixel_dv);
  const D3 lookfrom = ld3(cam.lookfrom), ddu = ld3(cam.defocus_disk_u), ddv = ld3(cam.defocus_disk_v);
  const D3 background = ld3(cam.background);
  const double INF = __longlong_as_double(0x7FF0000000000000ll);
  const float FINF = __int_as_float(0x7F800000);
  const uint32_t n_sph = P.n_spheres;
  uint32_t c_rays = 0, c_flag = 0, c_slow = 0;
  // up to 512 spheres: lane l tests the leaf boxes of spheres l, l + 64, ... l + 448 for every ray — 48 registers, loaded once
  const bool boxes_in_regs = BOXES_IN_REGS && n_sph <= 512u;
  float rb[BOXES_IN_REGS ? 8 : 1][6];
#pragma unroll
  for (int k = 0; k < (BOXES_IN_REGS ? 8 : 1); k++) {
    const uint32_t i = (uint32_t)(k * 64 + lane);
    Float4 b0 = {0.0f, 0.0f, 0.0f, 0.0f}, b1 = b0;
    if (boxes_in_regs && i < n_sph) b0 = *(const Float4 *)(leaf_boxes + (size_t)i * 8), b1 = *(const Float4 *)(leaf_boxes + (size_t)i * 8 + 4);
    rb[k][0] = b0.x, rb[k][1] = b0.y, rb[k][2] = b0.z, rb[k][3] = b0.w, rb[k][4] = b1.x, rb[k][5] = b1.y;
  }

  for (;;) {
    uint32_t pix = 0, n_begin = 0, pos0 = 0;
    D3 sum = d3(0.0, 0.0, 0.0);
    if (!src.next(P, lane, pix, sum, pos0, n_begin)) break;
    const uint32_t x = pix % W, r = pix / W;
    const uint32_t y = P.row_first + r * P.row_step;
    WaveRng<NT> rng{P.key, s_rng, tid, 0ull, pos0, 0xFFFFFFFFu};
    for (uint32_t n = n_begin; n < P.sample_end; n++) {
      uint64_t sample_index = (uint64_t)n + P.first_sample;
      rng.set_stream(sample_index * WH + (uint64_t)x * (uint64_t)W + (uint64_t)y);  // camera.rs:167-169 (x*W, sic)
      D3 pixel_center = (p00 + du * (double)x) + dv * (double)y;
      double sx = -0.5 + rng.gen_f64();
      double sy = -0.5 + rng.gen_f64();
      D3 pixel_sample = pixel_center + (du * sx + dv * sy);
      D3 o;
      if (cam.defocus_angle <= 0.0) o = lookfrom;
      else {
        double a, b;
        rng.unit_disc(a, b);
        o = (lookfrom + ddu * a) + ddv * b;
      }
      D3 d = pixel_sample - o;
      double time = rng.gen_f64();
      D3 thr = d3(1.0, 1.0, 1.0);
      D3 color = d3(0.0, 0.0, 0.0);
      for (uint32_t depth = cam.max_depth; depth > 0; depth--) {
        c_rays++;
        // ---- world.hit(r, [1e-10, inf]), cooperatively
        double closest = INF;
        uint32_t hit_prim = NONE;
        const RayAux32 ra32 = ray_aux32_direct(o, d);
        bool amb = !(ra32.slack < FINF);  // outside the binary32 filter's range: the reference's order
        if (!amb) {
          // (1) every lane: the leaf boxes of spheres lane, lane + 64, ... (certain-miss test of the fast traversal, closest = +inf)
          uint32_t ncand = 0;
          __builtin_amdgcn_wave_barrier();  // the previous ray's candidate list is no longer read
          auto scan = [&](uint32_t i, float bx0, float bx1, float by0, float by1, float bz0, float bz1) {
            bool cand = false;
            if (i < n_sph) {
              float t0x = fmaf(bx0, ra32.invx, -ra32.oix), t1x = fmaf(bx1, ra32.invx, -ra32.oix);
              float t0y = fmaf(by0, ra32.invy, -ra32.oiy), t1y = fmaf(by1, ra32.invy, -ra32.oiy);
              float t0z = fmaf(bz0, ra32.invz, -ra32.oiz), t1z = fmaf(bz1, ra32.invz, -ra32.oiz);
              float tmin = fmaxf(fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z)), 1e-10f);
              float tmax = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
              float thresh = fmaf(tmin + fabsf(tmax), 7.152557373046875e-07f, ra32.slack);  // 12u(|tmin|+|tmax|) + slack (ray_aux32_direct)
              cand = !((tmax - tmin) < -thresh);                                           // NaN arithmetic: candidate
            }
            const unsigned long long mask = __ballot(cand);
            if (cand) {
              const uint32_t slot = ncand + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
              if (slot < max_cand) cand_at(slot) = i;
            }
            ncand += (uint32_t)__popcll(mask);
          };
          if (BOXES_IN_REGS && boxes_in_regs) {  // the lane's own eight boxes never change: they were fetched once, before the first pixel
#pragma unroll
            for (int k = 0; k < (BOXES_IN_REGS ? 8 : 1); k++) scan((uint32_t)(k * 64 + lane), rb[k][0], rb[k][1], rb[k][2], rb[k][3], rb[k][4], rb[k][5]);
          } else {
            for (uint32_t base = 0; base < n_sph; base += 64u) {
              const uint32_t i = base + (uint32_t)lane;
              Float4 b0 = {0.0f, 0.0f, 0.0f, 0.0f}, b1 = b0;
              if (i < n_sph) b0 = *(const Float4 *)(leaf_boxes + (size_t)i * 8), b1 = *(const Float4 *)(leaf_boxes + (size_t)i * 8 + 4);
              scan(i, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the candidate list is read by other lanes of this wave
          __builtin_amdgcn_wave_barrier();
          if (ncand > max_cand) amb = true;  // more candidates than the list holds: the reference's fold
          else {
            // (2) one lane per candidate: Sphere::hit (sphere.rs:32-75) with ray_t = [1e-10, +inf): same arithmetic, same roots as
            // fast_sphere_hit; (3) wave reduction: closest root, runner-up and the widest tie band seen
            double second = INF, band_all = 0.0;
            bool sens_w = false;
            for (uint32_t c0 = 0; c0 < ncand; c0 += 64u) {
              const uint32_t k = c0 + (uint32_t)lane;
              double t_k = INF, band_k = 0.0;
              uint32_t payload_k = NONE;
              bool sens_k = false;
              if (k < ncand) {
                const uint32_t si = cand_at(k);
                const uint32_t payload = si | (((P.movbits[si >> 5] >> (si & 31u)) & 1u) ? SPH_MOVING : 0u);
                const DevSphere &s = spheres[si];
                D3 c0v = ld3(s.c0);
                D3 center = (payload & SPH_MOVING) ? c0v + ld3(s.dc) * time : c0v;
                D3 oc = o - center;
                double a = len2(d);
                double half_b = dot(oc, d);
                double c = len2(oc) - s.r2;
                double disc = half_b * half_b - a * c;
                if (!(disc < 0.0)) {
                  double sq = sqrt(disc);
                  double r_l = (-half_b - sq) / a;
                  double r_u = (-half_b + sq) / a;
                  double t = INF;
                  bool ok = true;
                  if (1e-10 <= r_l) t = r_l;
                  else if (1e-10 <= r_u) t = r_u;
                  else ok = false;
                  if (ok) {
                    t_k = t, payload_k = payload;
                    band_k = fast_tie_band(fabs(r_l) + fabs(r_u), ra32.oimax());
                    sens_k = fast_hit_is_order_sensitive(oc, d, t, s.r2 * s.inv_r, half_b, sq, r_l, r_u, ra32.oimax());
                  }
                }
              }
              const double t_min = wave_min_f64(t_k);
              const unsigned long long hit_mask = __ballot(payload_k != NONE);
              if (hit_mask == 0ull) continue;
              const unsigned long long win_mask = __ballot(payload_k != NONE && t_k == t_min);
              const int wl = __ffsll((long long)win_mask) - 1;
              const double t_2nd = wave_min_f64(lane == wl ? INF : t_k);
              band_all = fmax(band_all, -wave_min_f64(-band_k));
              if (win_mask != 0ull && t_min < closest) {  // (a NaN root never wins: win_mask is empty then, and the ray is re-traced below)
                second = fmin(fmin(second, closest), t_2nd);
                closest = t_min;
                hit_prim = __shfl(payload_k, wl, 64);
                sens_w = __shfl(sens_k ? 1 : 0, wl, 64) != 0;
              } else second = fmin(second, t_min);
              if (win_mask == 0ull) amb = true;  // NaN arithmetic in a root: the reference's fold decides
            }
            // order-sensitive: the winner grazes / sits next to an axis pole, or the runner-up is within the tie band (exact ties go
            // to the LAST sphere in the reference's order)
            if (hit_prim != NONE && (sens_w || !(second - closest > band_all))) amb = true;
          }
        }
        if (amb) {  // rare: the reference's own fold (uniform: every lane holds the same ray)
          c_flag += fast_slow_trace(ops, spheres, o, d, time, closest, hit_prim);
          c_slow++;
        }
#ifdef RL_FASTG_VERIFY
        else if (lane == 0) {  // the cooperative answer against the reference's fold
          fast_verify_ray(ops, spheres, o, d, time, closest, hit_prim, 3.0);
          atomicAdd(&g_vstats[3], 1ull);
        }
#endif
        if (hit_prim == NONE) {
          color = color + thr * background;
          break;
        }
        // ---- the HitRecord of the winner and Material::scatter, as in rl_rtiow_kernel.h (same arithmetic as at test time)
        uint32_t si = hit_prim & SPH_INDEX;
        const DevSphere &s = spheres[si];
        D3 c0v = ld3(s.c0);
        D3 center = (hit_prim & SPH_MOVING) ? c0v + ld3(s.dc) * time : c0v;
        D3 p = o + d * closest;
        D3 outward = (p - center) * s.inv_r;
        bool front = dot(d, outward) <= 0.0;
        D3 normal = front ? outward : -outward;
        const DevMaterial &m = P.materials[P.sphere_material[si]];
        uint32_t kind = m.kind;
        D3 nd;
        if (kind == RL_MAT_LAMBERTIAN) {
          D3 dir = normal + rng.unit_sphere();
          bool near_zero = approx_eq_eps(dir.x, 0.0, 1e-8) && approx_eq_eps(dir.y, 0.0, 1e-8) && approx_eq_eps(dir.z, 0.0, 1e-8);
          nd = near_zero ? normal : dir;
          thr = thr * texture_value(P, m.texture, 0.0, 0.0, p);
        } else if (kind == RL_MAT_METAL) {
          D3 reflected = d - normal * (2.0 * dot(d, normal));
          nd = normalize(reflected) + rng.unit_sphere() * m.fuzz;
          if (!(dot(nd, normal) > 0.0)) break;
          thr = thr * ld3(m.albedo);
        } else if (kind == RL_MAT_DIELECTRIC) {
          double ri = front ? 1.0 / m.ior : m.ior;
          double m2 = len2(d);
          D3 ud;
          if (approx_eq_eps(m2, 0.0, 1e-16)) {
            c_flag++;
            ud = d;
          } else
            ud = normalize(d);
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
          color = color + thr * texture_value(P, m.texture, 0.0, 0.0, p);
          break;
        } else {
          break;
        }
        o = p;
        d = nd;
      }
      sum = sum + color;
    }
    src.finish(P, lane, pix, sum, rng.pos);
  }
  // every lane of a wave counted the same rays: lane 0 reports
  if (lane == 0) {
    if (c_rays) atomicAdd(&P.stats[0], (unsigned long long)c_rays);
    if (c_flag) atomicAdd(&P.stats[6], (unsigned long long)c_flag);
    if (c_slow) atomicAdd(&P.stats[7], (unsigned long long)c_slow);
  }
}

// Work source of the stand-alone kernel: the pixels of a list, each from the state the previous launch left (P.resume) or from scratch
struct CoopListSource {
  const uint32_t *pixels;
  uint32_t n_pixels;
  uint32_t *counter;
  __device__ __forceinline__ bool next(const RtiowParams &P, int lane, uint32_t &pix, D3 &sum, uint32_t &pos, uint32_t &n_begin) const {
    uint32_t idx = 0;
    if (lane == 0) idx = atomicAdd(counter, 1u);
    idx = __shfl(idx, 0, 64);
    if (idx >= n_pixels) return false;
    pix = pixels[idx];
    n_begin = P.sample_begin, pos = 0u, sum = d3(0.0, 0.0, 0.0);
    if (P.resume) {  // continue where the previous launch stopped: same sums, same ChaCha word position
      const double *inp = P.out + (size_t)pix * 3;
      sum = d3(inp[0], inp[1], inp[2]);
      pos = P.pos_state[pix];
    }
    return true;
  }
  __device__ __forceinline__ void finish(const RtiowParams &P, int lane, uint32_t pix, D3 sum, uint32_t pos) const {
    if (lane == 0) {
      double *outp = P.out + (size_t)pix * 3;
      outp[0] = sum.x, outp[1] = sum.y, outp[2] = sum.z;
      if (P.pos_state) P.pos_state[pix] = pos;
    }
  }
};
struct CoopLinearCand {  // candidate list of a wave: max_cand consecutive words
  uint32_t *base;
  __device__ __forceinline__ uint32_t &operator()(uint32_t k) const { return base[k]; }
};

// stand-alone form (small frames, RL_RTIOW_KERNEL=coop): every pixel of the list through the cooperative body.
// NW waves per workgroup; LDS: [8][NW * 64] u64 RNG blocks (one per lane, all lanes of a wave hold the same) + [NW][max_cand] u32.
// REGS_FOR: the workgroup size the register budget is computed for (1024: 128 VGPRs, four waves per SIMD)
template <int NW, bool BOXES_IN_REGS, int REGS_FOR>
__global__ void RL_KERNEL_ALIGN __launch_bounds__(REGS_FOR) rtiow_coop_kernel(RtiowParams P, CoopParams C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NT = NW * 64;
  unsigned long long *s_rng = (unsigned long long *)smem;
  uint32_t *s_cand = (uint32_t *)(smem + (size_t)8 * NT * sizeof(unsigned long long)) + (size_t)(threadIdx.x >> 6) * C.max_cand;
  rtiow_coop_body<NT, BOXES_IN_REGS>(P, C.leaf_boxes, C.max_cand, s_rng, CoopLinearCand{s_cand}, CoopListSource{C.pixels, C.n_pixels, C.counter});
}

// ---- work stealing (the STEAL instantiation of rtiow_wave_kernel, small shards): see RtiowParams::steal_state
// Candidates are the pixels of the tiles in the cost-sorted order of the resume launch, most expensive first.  A wave asks for one
// (0 -> 1), waits until the lane that renders it reaches its next sample boundary and releases it (2: P.out, pos_state and steal_n hold
// the sums, the ChaCha word position and the next sample), and continues it with all 64 lanes.  Every wait is bounded: a request that
// is not answered in time is withdrawn (1 -> 0) and the pixel stays where it is.
struct CoopStealSource {
  __device__ __forceinline__ bool next(const RtiowParams &P, int lane, uint32_t &pix, D3 &sum, uint32_t &pos, uint32_t &n_begin) const {
    const uint32_t W = P.cam.image_width;
    for (;;) {
      uint32_t k = 0;
      if (lane == 0) k = atomicAdd(P.steal_counter, 1u);
      k = __shfl(k, 0, 64);
      if (k >= P.n_slots) return false;  // every candidate has been offered
      const uint32_t tile = P.tile_order[k >> 6], in = k & 63u;
      const uint32_t x = (tile % P.tiles_x) * 8u + (in & 7u), r = (tile / P.tiles_x) * 8u + (in >> 3);
      if (x >= W || r >= P.nrows) continue;
      pix = r * W + x;
      uint32_t s = 0;
      if (lane == 0) s = atomicCAS(&P.steal_state[pix], 0u, 1u);
      s = __shfl(s, 0, 64);
      if (s != 0u) continue;  // finished already (3)
      s = 1u;
      for (int spins = 0; spins < 16384; spins++) {  // a sample of the busiest pixels takes up to ~2 ms; each round sleeps ~1 us
        if (lane == 0) s = __atomic_load_n(&P.steal_state[pix], __ATOMIC_RELAXED);
        s = __shfl(s, 0, 64);
        if (s != 1u) break;
        __builtin_amdgcn_s_sleep(32);
      }
      if (s == 1u) {  // not answered: withdraw the request, unless the answer arrives right now
        if (lane == 0) s = atomicCAS(&P.steal_state[pix], 1u, 0u);
        s = __shfl(s, 0, 64);
        if (s == 1u) continue;
      }
      if (s != 2u) continue;  // the pixel finished meanwhile
      __threadfence();        // acquire: what the releasing lane wrote
      const volatile double *inp = P.out + (size_t)pix * 3;
      sum = d3(inp[0], inp[1], inp[2]);
      pos = ((const volatile uint32_t *)P.pos_state)[pix];
      n_begin = ((const volatile uint32_t *)P.steal_n)[pix];
      return true;
    }
  }
  __device__ __forceinline__ void finish(const RtiowParams &P, int lane, uint32_t pix, D3 sum, uint32_t pos) const {
    if (lane == 0) {
      double *outp = P.out + (size_t)pix * 3;
      outp[0] = sum.x, outp[1] = sum.y, outp[2] = sum.z;
      P.pos_state[pix] = pos;
      atomicExch(&P.steal_state[pix], 3u);
    }
  }
};
template <int NT>
struct CoopRingCand {  // candidate list of a wave inside its own (now idle) ChaCha ring columns: rows 8 and 9 of [16][NT] u64, low words
  unsigned long long *s_rng;
  int wave;
  __device__ __forceinline__ uint32_t &operator()(uint32_t k) const {
    return *(uint32_t *)(s_rng + ((size_t)(8u + (k >> 6)) * NT + (size_t)wave * 64u + (k & 63u)));
  }
};
template <int NT>
__device__ __forceinline__ void rtiow_steal_loop(const RtiowParams &P, unsigned long long *s_rng) {
  rtiow_coop_body<NT, false>(P, P.coop_leaf_boxes, 128u, s_rng, CoopRingCand<NT>{s_rng, (int)(threadIdx.x >> 6)}, CoopStealSource{});
}

}  // namespace rl
