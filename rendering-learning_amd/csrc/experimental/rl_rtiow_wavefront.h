// RTIOW sphere/BVH path, WAVEFRONT form ("v3"): rays live in HBM, stages are separate kernels.
//
// Every pixel of the shard has exactly one ray in flight.  One PASS advances every live pixel by one
// ray (one ray_color recursion level of the reference, camera.rs:232-260):
//     TRAV   each wave owns a contiguous range of the pass's ray queue; its lanes pull 128-byte ray records
//            from that range as they finish (no atomics), traverse the LDS-resident scene program
//            (stackless, reference order, filtered AABB test) and write a 16-byte hit record
//     SHADE  one lane per queue entry: miss -> background, hit -> rebuild the HitRecord, scatter, write the
//            next ray record (appended to the next pass's queue); a finished path goes to the GEN queue
//     GEN    next sample of a pixel: new ChaCha stream at the pixel's word position, get_ray
// Per-pixel state (sums, throughput, ChaCha word position + cached block, sample / depth counters) is a
// 128-byte record in HBM; samples of one pixel stay strictly sequential (the word position couples
// them), different pixels never interact.  Lanes are not bound to pixels, so no lane idles behind
// another lane's long path and the tail of a frame is short even when a GPU owns few pixels.
// Arithmetic is the wave kernel's (rl_rtiow_wave.h): results are bit-identical.
//
// STATUS: experimental (RL_RTIOW_KERNEL=wavefront).  Measured on MI355X it is ~3x SLOWER than the wave-scheduled
// megakernel on BASELINE configs[1] (1080p, 32 spp: 192 ms vs 67 ms; traversal alone 2.4 Grays/s, the 256 B of
// per-ray state traffic and ~700 passes x 4 launches cost the rest), and the number of passes equals the longest
// per-pixel ray chain, so it does not improve strong scaling either.  Kept as the measured alternative that
// DESIGN.md §3.5 discusses; the default kernel is rl_rtiow_wave.h.
#pragma once
#include "../rl_rtiow_wave.h"

namespace rl {

struct alignas(16) PixState {  // 128 B
  double sum[3];
  double thr[3];
  unsigned long long blk[8];  // cached ChaCha block of the current stream
  uint32_t n;                 // next sample index (relative to first_sample)
  uint32_t pos;               // ChaCha u32 word position since the pixel started
  uint32_t blk_ctr;           // block counter of blk[], 0xFFFFFFFF = none
  uint32_t depth;             // remaining ray_color depth of the ray in flight
};
static_assert(sizeof(PixState) == 128, "PixState must be 128 B");

struct alignas(16) RayRec {  // 128 B
  double o[3], d[3], inv[3], oi[3];
  double time, slack;
  uint32_t fast_ok, pad[3];
};
static_assert(sizeof(RayRec) == 128, "RayRec must be 128 B");

struct alignas(16) HitRec {  // 16 B
  double t;
  uint32_t prim, pad;
};

// control words (u32) in device memory
enum : uint32_t {
  WC_PARITY = 0,      // which half of q_trav is the current pass's queue
  WC_NTRAV = 1,       // entries in the current ray queue
  WC_NNEXT = 2,       // entries appended to the next pass's ray queue
  WC_TRAV_CLAIM = 3,  // claim counter of TRAV
  WC_NGEN = 4,        // entries in the GEN queue
  WC_GEN_CLAIM = 5,
  WC_SPARE = 6,
  WC_FINISHED = 7,     // pixels that completed all samples
  WC_WORDS = 16
};

struct WfParams {
  RtiowParams R;  // scene, camera, key, first_sample, row mapping, stats
  PixState *pix;
  RayRec *ray;
  HitRec *hit;
  uint32_t *q_trav;   // [2][npix]
  uint32_t *q_gen;    // [npix]
  uint32_t *ctl;      // [WC_WORDS]
  uint32_t npix;
};

__device__ __forceinline__ uint32_t wave_append(uint32_t *counter, bool want) {  // returns slot (valid where want)
  unsigned long long mask = __ballot(want);
  uint32_t lane = __lane_id();
  uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
  uint32_t base = 0;
  if (mask) {
    uint32_t leader = __ffsll((long long)mask) - 1;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader, 64);
  }
  return base + rank;
}

// ---------------------------------------------------------------- pass bookkeeping (1 thread)
__global__ void wf_begin_pass(WfParams W) {
  uint32_t *c = W.ctl;
  c[WC_PARITY] ^= 1u;
  c[WC_NTRAV] = c[WC_NNEXT];
  c[WC_NNEXT] = 0;
  c[WC_TRAV_CLAIM] = 0;
  c[WC_NGEN] = 0;
  c[WC_GEN_CLAIM] = 0;
}

__global__ void wf_init(WfParams W) {  // every pixel starts in the GEN queue with zeroed state
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < W.npix) {
    PixState &p = W.pix[i];
    p.sum[0] = p.sum[1] = p.sum[2] = 0.0;
    p.thr[0] = p.thr[1] = p.thr[2] = 1.0;
    p.n = 0, p.pos = 0, p.blk_ctr = 0xFFFFFFFFu, p.depth = 0;
    W.q_gen[i] = i;
  }
  if (i == 0) {
    for (int k = 0; k < (int)WC_WORDS; k++) W.ctl[k] = 0;
    W.ctl[WC_NGEN] = W.npix;
  }
}

__global__ void wf_finish(WfParams W) {  // sums -> Canvas.data layout
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < W.npix) {
    const PixState &p = W.pix[i];
    double *o = W.R.out + (size_t)i * 3;
    o[0] = p.sum[0], o[1] = p.sum[1], o[2] = p.sum[2];
  }
}

// ---------------------------------------------------------------- helpers shared by GEN / SHADE
template <int NT>
__device__ __forceinline__ void rng_load(const PixState &p, unsigned long long *s_rng, int tid, Rng &rng, uint64_t stream) {
  rng.stream = stream, rng.pos = p.pos, rng.buf_ctr = p.blk_ctr;
  if (p.blk_ctr != 0xFFFFFFFFu) {
#pragma unroll
    for (int k = 0; k < 8; k++) s_rng[k * NT + tid] = p.blk[k];
  }
}
template <int NT>
__device__ __forceinline__ void rng_store(PixState &p, const unsigned long long *s_rng, int tid, const Rng &rng) {
  p.pos = rng.pos, p.blk_ctr = rng.buf_ctr;
  if (rng.buf_ctr != 0xFFFFFFFFu) {
#pragma unroll
    for (int k = 0; k < 8; k++) p.blk[k] = s_rng[k * NT + tid];
  }
}
__device__ __forceinline__ void write_ray(RayRec &r, D3 o, D3 d, double time) {
  RayAux a = ray_aux(o, d);
  r.o[0] = o.x, r.o[1] = o.y, r.o[2] = o.z;
  r.d[0] = d.x, r.d[1] = d.y, r.d[2] = d.z;
  r.inv[0] = a.inv.x, r.inv[1] = a.inv.y, r.inv[2] = a.inv.z;
  r.oi[0] = a.oi.x, r.oi[1] = a.oi.y, r.oi[2] = a.oi.z;
  r.time = time, r.slack = a.slack, r.fast_ok = a.fast_ok ? 1u : 0u;
}
__device__ __forceinline__ uint64_t pixel_stream(const RtiowParams &P, uint32_t pix, uint32_t n, uint32_t &x, uint32_t &y) {
  const uint32_t Wd = P.cam.image_width;
  uint32_t r = pix / Wd;
  x = pix % Wd;
  y = P.row_first + r * P.row_step;
  uint64_t WH = (uint64_t)P.cam.image_width * (uint64_t)P.cam.image_height;
  return ((uint64_t)n + P.first_sample) * WH + (uint64_t)x * (uint64_t)Wd + (uint64_t)y;  // camera.rs:167-169 (x*W, sic)
}

// ---------------------------------------------------------------- GEN
template <int NT>
__global__ void __launch_bounds__(NT) wf_gen(WfParams W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long *s_rng = (unsigned long long *)smem;  // [8][NT]
  const int tid = threadIdx.x;
  const RtiowParams &P = W.R;
  const rl_rtiow_camera &cam = P.cam;
  RngCtx<NT> rc{P.key, s_rng, tid};
  const uint32_t n_gen = W.ctl[WC_NGEN];
  if ((uint64_t)blockIdx.x * NT >= n_gen) return;  // grids are sized for a full frame; late passes need few blocks
  uint32_t *q_next = W.q_trav + (size_t)(W.ctl[WC_PARITY] ^ 1u) * W.npix;
  unsigned long long c_rays = 0, c_words = 0;
  uint32_t finished = 0;
  for (;;) {
    uint32_t base = 0;
    if ((tid & 63) == 0) base = atomicAdd(&W.ctl[WC_GEN_CLAIM], 64u);
    base = __shfl(base, 0, 64);
    if (base >= n_gen) break;
    uint32_t idx = base + (tid & 63);
    bool live = idx < n_gen;
    bool emit = false;
    uint32_t pix = 0;
    if (live) {
      pix = W.q_gen[idx];
      PixState &ps = W.pix[pix];
      uint32_t n = ps.n;
      Rng rng;
      rng.pos = ps.pos, rng.buf_ctr = 0xFFFFFFFFu, rng.stream = 0;
      while (n < cam.samples_per_pixel) {
        uint32_t x, y;
        rng.stream = pixel_stream(P, pix, n, x, y);
        rng.buf_ctr = 0xFFFFFFFFu;  // new stream: the cached block is stale
        D3 p00 = ld3(cam.pixel_00), du = ld3(cam.pixel_du), dv = ld3(cam.pixel_dv);
        D3 pixel_center = (p00 + du * (double)x) + dv * (double)y;  // get_ray camera.rs:203-216
        double sx = -0.5 + rc.gen_f64(rng);
        double sy = -0.5 + rc.gen_f64(rng);
        D3 pixel_sample = pixel_center + (du * sx + dv * sy);
        D3 o;
        if (cam.defocus_angle <= 0.0) o = ld3(cam.lookfrom);
        else {
          double a, b;
          rc.unit_disc(rng, a, b);
          o = (ld3(cam.lookfrom) + ld3(cam.defocus_disk_u) * a) + ld3(cam.defocus_disk_v) * b;
        }
        D3 d = pixel_sample - o;
        double time = rc.gen_f64(rng);
        if (cam.max_depth == 0) {  // ray_color(depth 0) = black
          n++;
          continue;
        }
        c_rays++;
        write_ray(W.ray[pix], o, d, time);
        ps.thr[0] = ps.thr[1] = ps.thr[2] = 1.0;
        ps.depth = cam.max_depth;
        emit = true;
        break;
      }
      ps.n = n;
      rng_store<NT>(ps, s_rng, tid, rng);
      if (!emit) {  // all samples done
        finished++;
        c_words += rng.pos;
      }
    }
    uint32_t slot = wave_append(&W.ctl[WC_NNEXT], emit);
    if (emit) q_next[slot] = pix;
  }
  unsigned long long v = wave_sum(c_rays);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[0], v);
  v = wave_sum(c_words);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[5], v);
  v = wave_sum((unsigned long long)finished);
  if ((tid & 63) == 0 && v) atomicAdd(&W.ctl[WC_FINISHED], (uint32_t)v);
}

// ---------------------------------------------------------------- SHADE
template <int NT>
__global__ void __launch_bounds__(NT) wf_shade(WfParams W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long *s_rng = (unsigned long long *)smem;  // [8][NT]
  const int tid = threadIdx.x;
  const RtiowParams &P = W.R;
  const rl_rtiow_camera &cam = P.cam;
  RngCtx<NT> rc{P.key, s_rng, tid};
  const uint32_t n_cur = W.ctl[WC_NTRAV];
  const uint32_t total_chunks = (n_cur + 63u) >> 6;
  if ((uint64_t)blockIdx.x * (NT / 64) >= total_chunks) return;
  const uint32_t *q_cur = W.q_trav + (size_t)W.ctl[WC_PARITY] * W.npix;
  uint32_t *q_next = W.q_trav + (size_t)(W.ctl[WC_PARITY] ^ 1u) * W.npix;
  unsigned long long c_rays = 0, c_flag = 0;
  for (uint32_t chunk = blockIdx.x * (NT / 64) + (tid >> 6); chunk < total_chunks; chunk += gridDim.x * (NT / 64)) {
    uint32_t idx = chunk * 64u + (tid & 63);
    bool live = idx < n_cur;
    bool cont = false, to_gen = false;
    uint32_t pix = 0;
    if (live) {
      pix = q_cur[idx];
      PixState &ps = W.pix[pix];
      const RayRec &rr = W.ray[pix];
      const HitRec hr = W.hit[pix];
      D3 thr = ld3(ps.thr);
      D3 o = ld3(rr.o), d = ld3(rr.d);
      double time = rr.time;
      bool path_done = false;
      D3 nd = d, p = o;
      if (hr.prim == NONE) {  // miss -> background (camera.rs:257)
        D3 s = ld3(ps.sum) + thr * ld3(cam.background);
        ps.sum[0] = s.x, ps.sum[1] = s.y, ps.sum[2] = s.z;
        path_done = true;
      } else {
        uint32_t x, y;
        Rng rng;
        rng_load<NT>(ps, s_rng, tid, rng, pixel_stream(P, pix, ps.n, x, y));
        uint32_t si = hr.prim & SPH_INDEX;
        const DevSphere &s = P.spheres[si];
        D3 c0 = ld3(s.c0);
        D3 center = (hr.prim & SPH_MOVING) ? c0 + ld3(s.dc) * time : c0;
        p = o + d * hr.t;
        D3 outward = (p - center) * s.inv_r;
        bool front = dot(d, outward) <= 0.0;
        D3 normal = front ? outward : -outward;
        const DevMaterial &m = P.materials[P.sphere_material[si]];
        uint32_t kind = m.kind;
        if (kind == RL_MAT_LAMBERTIAN) {
          D3 dir = normal + rc.unit_sphere(rng);
          bool near_zero = approx_eq_eps(dir.x, 0.0, 1e-8) && approx_eq_eps(dir.y, 0.0, 1e-8) && approx_eq_eps(dir.z, 0.0, 1e-8);
          nd = near_zero ? normal : dir;
          thr = thr * texture_value(P, m.texture, 0.0, 0.0, p);
        } else if (kind == RL_MAT_METAL) {
          D3 reflected = d - normal * (2.0 * dot(d, normal));
          nd = normalize(reflected) + rc.unit_sphere(rng) * m.fuzz;
          if (!(dot(nd, normal) > 0.0)) path_done = true;
          else thr = thr * ld3(m.albedo);
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
            reflect = refl > rc.gen_f64(rng);
          }
          if (reflect) nd = ud - normal * (2.0 * dot(ud, normal));
          else {
            D3 perp = (ud + normal * cos_theta) * ri;
            D3 par = normal * (-sqrt(fabs(1.0 - len2(perp))));
            nd = perp + par;
          }
        } else if (kind == RL_MAT_DIFFUSE_LIGHT) {
          D3 sm = ld3(ps.sum) + thr * texture_value(P, m.texture, 0.0, 0.0, p);
          ps.sum[0] = sm.x, ps.sum[1] = sm.y, ps.sum[2] = sm.z;
          path_done = true;
        } else {
          path_done = true;
        }
        rng_store<NT>(ps, s_rng, tid, rng);
      }
      uint32_t depth = ps.depth;
      if (!path_done) {
        depth--;
        if (depth == 0) path_done = true;
      }
      if (path_done) {
        ps.n = ps.n + 1;
        to_gen = true;
      } else {
        c_rays++;
        ps.depth = depth;
        ps.thr[0] = thr.x, ps.thr[1] = thr.y, ps.thr[2] = thr.z;
        write_ray(W.ray[pix], p, nd, time);
        cont = true;
      }
    }
    uint32_t slot = wave_append(&W.ctl[WC_NNEXT], cont);
    if (cont) q_next[slot] = pix;
    slot = wave_append(&W.ctl[WC_NGEN], to_gen);
    if (to_gen) W.q_gen[slot] = pix;
  }
  unsigned long long v = wave_sum(c_rays);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[0], v);
  v = wave_sum(c_flag);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[6], v);
}

// ---------------------------------------------------------------- TRAV
enum : uint32_t { TS_FETCH = 0, TS_TRAV = 1, TS_LEAF = 2, TS_DONE = 3 };

template <int NT, bool LDS_SCENE, bool STATS>
__global__ void __launch_bounds__(NT) wf_trav(WfParams W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const RtiowParams &P = W.R;
  const DevOp *ops = P.ops;
  const DevSphere *spheres = P.spheres;
  const uint32_t n_trav = W.ctl[WC_NTRAV];
  if ((uint64_t)blockIdx.x * NT >= n_trav) return;  // before staging the scene: late passes need few blocks (ranges are >= 64 per wave)
  if (LDS_SCENE) {
    DevOp *s_ops = (DevOp *)smem;
    DevSphere *s_sph = (DevSphere *)(s_ops + P.n_ops);
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
  const double INF = __longlong_as_double(0x7FF0000000000000ll);
  const uint32_t *q_cur = W.q_trav + (size_t)W.ctl[WC_PARITY] * W.npix;
  // each wave owns a contiguous range of the pass's ray queue (multiple of 64 entries)
  const uint32_t n_waves = gridDim.x * (NT / 64);
  uint32_t per_wave = ((n_trav + n_waves - 1) / n_waves + 63u) & ~63u;
  const uint32_t wave_id = blockIdx.x * (NT / 64) + (tid >> 6);
  uint32_t wave_next = wave_id * per_wave;
  uint32_t wave_end = wave_next + per_wave < n_trav ? wave_next + per_wave : n_trav;
  if (wave_next > n_trav) wave_next = n_trav;
  uint32_t state = TS_FETCH;
  bool have = false;
  uint32_t pix = 0, pc = 0, hit_prim = NONE;
  D3 o = d3(0, 0, 0), d = d3(0, 0, 1);
  RayAux ra;
  ra.inv = d3(0, 0, 1), ra.oi = d3(0, 0, 0), ra.slack = 0.0, ra.fast_ok = false;
  double time = 0.0, closest = INF;
  unsigned long long c_nodes = 0, c_sph = 0, c_flag = 0;

  for (;;) {
    int n_trv = __popcll(__ballot(state == TS_TRAV));
    int n_leaf = __popcll(__ballot(state == TS_LEAF));
    int n_fetch = __popcll(__ballot(state == TS_FETCH));
    if ((n_trv | n_leaf | n_fetch) == 0) break;
    uint32_t pick = TS_TRAV;
    int best = n_trv;
    if (n_leaf > best) pick = TS_LEAF, best = n_leaf;
    if (n_fetch > best) pick = TS_FETCH, best = n_fetch;

    if (pick == TS_TRAV) {
      int floor_n = (best * (int)P.tune[1]) >> 4;
      for (int it = 0; it < (int)P.tune[0]; it++) {
        if (state == TS_TRAV) {
          const DevOp &op = ops[pc];
          double bx[6] = {op.box[0], op.box[1], op.box[2], op.box[3], op.box[4], op.box[5]};
          uint32_t code = op.code, skip = op.skip;
          uint32_t kind = code & 0xFFu;
          bool is_box = (kind == OP_BOX) | (kind == OP_BOX_SPH);
          bool certain;
          bool hitb = aabb_fast(bx, ra, closest, certain);
          if (is_box && !(certain && ra.fast_ok && (code & BOX_FINITE))) hitb = aabb_hit(bx, o, d, 1e-10, closest);
          if (STATS) c_nodes += is_box ? 1u : 0u;
          bool to_leaf = (kind == OP_SPHERE) | ((kind == OP_BOX_SPH) & hitb);
          uint32_t npc = (is_box & !hitb) ? skip : ((kind == OP_BOX) ? pc + 1u : pc);
          uint32_t nstate = (kind == OP_END) ? TS_FETCH : (to_leaf ? TS_LEAF : TS_TRAV);
          pc = npc;
          state = nstate;
        }
        if (__popcll(__ballot(state == TS_TRAV)) < floor_n) break;
      }
    } else if (pick == TS_LEAF) {
      if (state == TS_LEAF) {
        const DevOp &op = ops[pc];
        uint32_t a = op.a, b = op.b;
        Hit h{closest, hit_prim};
        if (STATS) c_sph++;
        if (sphere_hit(spheres[a & SPH_INDEX], a, o, d, time, 1e-10, h)) c_flag++;
        if (b != NONE) {
          if (STATS) c_sph++;
          if (sphere_hit(spheres[b & SPH_INDEX], b, o, d, time, 1e-10, h)) c_flag++;
        }
        closest = h.t, hit_prim = h.prim;
        pc = op.skip;
        state = TS_TRAV;
      }
    } else {  // TS_FETCH: retire the finished ray (hit record), take the next entry of this wave's range
      bool mine = state == TS_FETCH;
      if (mine && have) {
        HitRec hr;
        hr.t = closest, hr.prim = hit_prim, hr.pad = 0;
        W.hit[pix] = hr;
      }
      unsigned long long mask = __ballot(mine);
      uint32_t rank = __popcll(mask & ((1ull << (tid & 63)) - 1ull));
      uint32_t idx = wave_next + rank;
      wave_next += (uint32_t)__popcll(mask);  // uniform: no atomics, the range is private to the wave
      if (mine) {
        have = false;
        if (idx >= wave_end) state = TS_DONE;
        else {
          pix = q_cur[idx];
          const RayRec &rr = W.ray[pix];
          o = ld3(rr.o), d = ld3(rr.d);
          ra.inv = ld3(rr.inv), ra.oi = ld3(rr.oi), ra.slack = rr.slack, ra.fast_ok = rr.fast_ok != 0;
          time = rr.time;
          pc = 0, closest = INF, hit_prim = NONE;
          have = true;
          state = TS_TRAV;
        }
      }
    }
  }
  unsigned long long v = wave_sum(c_flag);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[6], v);
  if (STATS) {
    v = wave_sum(c_nodes);
    if ((tid & 63) == 0) atomicAdd(&P.stats[1], v);
    v = wave_sum(c_sph);
    if ((tid & 63) == 0) atomicAdd(&P.stats[2], v);
  }
}

}  // namespace rl
