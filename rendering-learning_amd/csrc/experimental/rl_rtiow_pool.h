// RTIOW sphere/BVH kernel, pooled form ("v5"): the wave-scheduled state machine of rl_rtiow_wave.h with the
// RAY decoupled from the PIXEL that owns it.
//
// Every lane carries two independent things:
//   * a pixel CONTEXT  (GEN / SHADE / FILL / WAIT): sums, throughput, sample counter, ChaCha ring — exactly the
//     per-pixel, per-sample order of the reference (camera.rs:145-199);
//   * a traversal SLOT (IDLE / TRAV / LEAF): one ray being walked through the linked scene program.
// A context that produced a ray posts it in a workgroup-wide LDS pool (origin, direction, time; one record per
// context) and waits; any slot of the same lane column (lane l of any of the workgroup's waves) may claim it,
// traverse it and write back (t, primitive).  Rays therefore drift to whichever waves are traversing while
// other waves shade, so TRAV / LEAF blocks run with most of their 64 lanes busy instead of the ~45 % a
// one-ray-per-pixel-lane schedule reaches.  Which lane traverses a ray cannot change its result: traversal is a
// pure function of (ray, scene) and all per-pixel state stays with the owner, so pixels, counters and RNG word
// positions are bit-identical to rtiow_wave_kernel (tests assert it).
//
// LDS: [linked ops][spheres][ChaCha rings 16 x NT u64][ray records 7 x NT f64][status NT u32][prim NT u32].
#pragma once
#include "../rl_rtiow_wave.h"

namespace rl {

enum : uint32_t { ST_WAIT = 7 };                          // context: ray posted, result not back yet
enum : uint32_t { SL_IDLE = 0, SL_TRAV = 1, SL_FIN = 2, SL_LEAF = 5 };  // slot (TRAV / FIN / LEAF = the linked words' state field)
enum : uint32_t { RS_NONE = 0, RS_READY = 1, RS_TAKEN = 2, RS_DONE = 3 };  // ray record status

__device__ __forceinline__ uint32_t lds_load_acquire(uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store_release(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <int NT, bool STATS>
__global__ void __launch_bounds__(NT) rtiow_pool_kernel(RtiowParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const uint32_t lane = (uint32_t)tid & 63u, wave = (uint32_t)tid >> 6;
  constexpr uint32_t NW = NT / 64;
  const size_t scene_lds = (size_t)P.n_ops * sizeof(DevOp) + (size_t)P.n_spheres * sizeof(DevSphere);
  unsigned long long *s_rng = (unsigned long long *)(smem + scene_lds);  // [16][NT]
  double *s_rec = (double *)(s_rng + (size_t)16 * NT);                   // [7][NT]: o.xyz d.xyz time (-> t when done)
  uint32_t *s_status = (uint32_t *)(s_rec + (size_t)7 * NT);             // [NT]
  uint32_t *s_prim = s_status + NT;                                      // [NT]
  const unsigned char *opbase = smem;
  const DevSphere *spheres;
  {
    DevOp *s_ops = (DevOp *)smem;
    DevSphere *s_sph = (DevSphere *)(s_ops + P.n_ops);
    const uint4 *g = (const uint4 *)P.lops;
    uint4 *l = (uint4 *)s_ops;
    for (uint32_t i = tid; i < P.n_ops * 4u; i += NT) {
      uint4 v = g[i];
      if ((i & 3u) == 3u) {  // {w_hit, w_miss, a, b}: successor indices -> LDS byte offsets
        v.x = (v.x & 0xE0000000u) | ((v.x & 0x1FFFFFFFu) << 6);
        v.y = (v.y & 0xE0000000u) | ((v.y & 0x1FFFFFFFu) << 6);
      }
      l[i] = v;
    }
    g = (const uint4 *)P.spheres;
    l = (uint4 *)s_sph;
    for (uint32_t i = tid; i < P.n_spheres * 4u; i += NT) l[i] = g[i];
    s_status[tid] = RS_NONE;
    __syncthreads();
    spheres = s_sph;
  }
  const uint32_t entry0 = (P.entry0 & 0xE0000000u) | ((P.entry0 & 0x1FFFFFFFu) << 6);
  const rl_rtiow_camera &cam = P.cam;
  const uint32_t W = cam.image_width;
  const uint32_t s_begin = P.sample_begin, spp = P.sample_end;
  const uint64_t WH = (uint64_t)cam.image_width * (uint64_t)cam.image_height;
  const double INF = __longlong_as_double(0x7FF0000000000000ll);
  const int refill_min = (int)P.tune[2];

  // ---- pixel context
  Ring<NT> rng{P.key, s_rng, tid, 0ull, 0u, 0u, 0u};
  uint32_t cstate = ST_GEN;
  uint32_t px = 0, pr = 0, n = spp;
  uint32_t ptile = 0, pix_rays = 0;
  bool have_pixel = false;
  D3 sum = d3(0.0, 0.0, 0.0);
  D3 o = d3(0.0, 0.0, 0.0), d = d3(0.0, 0.0, 1.0), thr = d3(1.0, 1.0, 1.0);
  double time = 0.0, res_t = INF;
  uint32_t res_prim = NONE, depth = 0;
  // ---- traversal slot
  uint32_t sstate = SL_IDLE, owner = 0, pc = 0, hit_prim = NONE;
  D3 so = o, sd = d;
  RayAux ra = ray_aux(so, sd);
  double stime = 0.0, closest = INF;

  uint32_t c_rays = 0, c_flag = 0;
  unsigned long long c_nodes = 0, c_sph = 0, c_words = 0;
  unsigned long long sc_exec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sc_pop[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t backoff = 0, patience = 3;
  unsigned long long sc_cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
  int prev_cat = -1;

  auto post_ray = [&]() {  // context -> pool
    s_rec[0 * NT + tid] = o.x, s_rec[1 * NT + tid] = o.y, s_rec[2 * NT + tid] = o.z;
    s_rec[3 * NT + tid] = d.x, s_rec[4 * NT + tid] = d.y, s_rec[5 * NT + tid] = d.z;
    s_rec[6 * NT + tid] = time;
    lds_store_release(&s_status[tid], RS_READY);
    cstate = ST_WAIT;
  };

  for (;;) {
    if (STATS) {  // debug (tools/sched.py): shader cycles per block category, per wave
      unsigned long long now = __builtin_readcyclecounter();
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (prev_cat == k) sc_cyc[k] += now - t_prev;
      t_prev = now;
    }
    // ---- bookkeeping that costs a few instructions per round
    if (sstate == SL_FIN) {  // traversal finished: hand (t, primitive) back to the owner
      s_rec[6 * NT + owner] = closest;
      s_prim[owner] = hit_prim;
      lds_store_release(&s_status[owner], RS_DONE);
      sstate = SL_IDLE;
    }
    if (cstate == ST_WAIT) {
      if (lds_load_acquire(&s_status[tid]) == RS_DONE) {
        res_t = s_rec[6 * NT + tid];
        res_prim = s_prim[tid];
        cstate = ST_SHADE;
      }
    }
    if (cstate == ST_SHADE && rng.low()) cstate = ST_FILL;

    int n_trav = __popcll(__ballot(sstate == SL_TRAV));
    int n_leaf = __popcll(__ballot(sstate == SL_LEAF));
    int n_shade = __popcll(__ballot(cstate == ST_SHADE));
    int n_fill = __popcll(__ballot(cstate == ST_FILL));
    int n_gen = __popcll(__ballot(cstate == ST_GEN));
    int n_wait = __popcll(__ballot(cstate == ST_WAIT));
    int n_busy = n_trav + n_leaf, n_idle = 64 - n_busy;
    int shade_best = n_shade > n_fill ? n_shade : n_fill;
    if (n_gen > shade_best) shade_best = n_gen;
    if ((n_busy | shade_best | n_wait) == 0) break;  // every context DONE, every slot idle

    uint32_t pick = ST_TRAV;
    int best = n_trav;
    if (n_leaf > best) pick = ST_LEAF, best = n_leaf;
    if (n_shade > best) pick = ST_SHADE, best = n_shade;
    if (n_fill > best) pick = ST_FILL, best = n_fill;
    if (n_gen > best) pick = ST_GEN, best = n_gen;

    // ---- refill idle slots from the pool.  There are never more rays than slots in the workgroup, so taking rays
    // eagerly spreads them thinly over all waves; a wave therefore only takes a BATCH: at least tune[2] rays that
    // bring it to at least tune[3] busy slots.  Waves that cannot get a batch shade their own contexts or sleep,
    // which concentrates the rays in fewer, fuller waves.  A wave with nothing else to do loses patience after a
    // few sleeps and takes whatever its columns offer (its own contexts' rays at the latest): progress is guaranteed.
    if (backoff) backoff--;
    const bool nothing = best == 0;  // every context waits on a ray some slot (here or in another wave) still has to walk
    if (n_idle > 0 && (nothing || (n_busy >= shade_best && backoff == 0))) {
      uint32_t cand = NONE;
      if (sstate == SL_IDLE) {
        // the NW records of this lane column; the own wave's record is looked at last and wins
#pragma unroll
        for (uint32_t k = 0; k < NW; k++) {
          uint32_t c = ((wave + 1u + k) % NW) * 64u + lane;
          uint32_t st = __hip_atomic_load(&s_status[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          cand = st == RS_READY ? c : cand;
        }
      }
      int n_avail = __popcll(__ballot(cand != NONE));
      bool take = n_avail > 0 && ((n_avail >= refill_min && n_busy + n_avail >= (int)P.tune[3]) || (nothing && patience == 0));
      int got = 0;
      if (take && cand != NONE) {
        uint32_t expect = RS_READY;
        if (__hip_atomic_compare_exchange_strong(&s_status[cand], &expect, RS_TAKEN, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
          owner = cand;
          so = d3(s_rec[0 * NT + cand], s_rec[1 * NT + cand], s_rec[2 * NT + cand]);
          sd = d3(s_rec[3 * NT + cand], s_rec[4 * NT + cand], s_rec[5 * NT + cand]);
          stime = s_rec[6 * NT + cand];
          ra = ray_aux(so, sd);
          if (!ra.fast_ok) ra.slack = INF;
          pc = entry0 & 0x1FFFFFFFu, closest = INF, hit_prim = NONE;
          sstate = entry0 >> 29;
          got = 1;
        }
      }
      int n_got = __popcll(__ballot(got != 0));
      if (STATS) sc_exec[6]++, sc_pop[6] += (unsigned)n_got, prev_cat = nothing ? 7 : 6;
      if (n_got == 0) {
        backoff = 3;
        if (nothing) {
          if (patience) patience--;
          __builtin_amdgcn_s_sleep(4);  // only waiting on other waves
        }
      } else
        patience = 3;
      if (nothing || n_got) continue;
    }
    if (STATS) prev_cat = (int)pick;
    if (STATS && pick != ST_TRAV) {
#pragma unroll
      for (int k = 0; k < 6; k++)
        if (pick == (uint32_t)k) sc_exec[k]++, sc_pop[k] += (unsigned)best;
    }

    if (pick == ST_TRAV) {
      int floor_n = (best * (int)P.tune[1]) >> 4;
      for (int it = 0; it < (int)P.tune[0]; it++) {
        if (STATS) {
          int np = __popcll(__ballot(sstate == SL_TRAV));
          sc_exec[ST_TRAV]++, sc_pop[ST_TRAV] += (unsigned)np;
        }
        if (sstate == SL_TRAV) {
          const DevOp &op = *(const DevOp *)(opbase + pc);
          double bx[6] = {op.box[0], op.box[1], op.box[2], op.box[3], op.box[4], op.box[5]};
          uint32_t w_hit = op.code, w_miss = op.skip;
          bool certain;
          bool hitb = aabb_fast(bx, ra, closest, certain);
          if (!certain) hitb = aabb_hit(P.ops[pc >> 6].box, so, sd, 1e-10, closest);  // rare: exact divisions
          if (STATS) c_nodes++;
          uint32_t w = hitb ? w_hit : w_miss;
          pc = w & 0x1FFFFFFFu;
          sstate = w >> 29;
        }
        if (__popcll(__ballot(sstate == SL_TRAV)) < floor_n) break;
      }
    } else if (pick == ST_LEAF) {
      if (sstate == SL_LEAF) {
        const DevOp &op = *(const DevOp *)(opbase + pc);
        uint32_t a = op.a, b = op.b, w = op.skip;
        Hit h{closest, hit_prim};
        if (STATS) c_sph++;
        if (sphere_hit(spheres[a & SPH_INDEX], a, so, sd, stime, 1e-10, h)) c_flag++;
        if (b != NONE) {
          if (STATS) c_sph++;
          if (sphere_hit(spheres[b & SPH_INDEX], b, so, sd, stime, 1e-10, h)) c_flag++;
        }
        closest = h.t, hit_prim = h.prim;
        pc = w & 0x1FFFFFFFu;
        sstate = w >> 29;
      }
    } else if (pick == ST_FILL) {
      if (cstate == ST_FILL) {
        rng.top_up();
        cstate = ST_SHADE;
      }
    } else if (pick == ST_GEN) {
      if (cstate == ST_GEN) {
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
            cstate = ST_DONE;
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
          rng.reset_stream(sample_index * WH + (uint64_t)px * (uint64_t)W + (uint64_t)y);  // camera.rs:167-170
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
          if (depth == 0) {
            sum = sum + d3(0.0, 0.0, 0.0);
            n++;
          } else {
            c_rays++;
            pix_rays++;
            post_ray();
          }
        }
      }
    } else {  // ST_SHADE
      if (cstate == ST_SHADE) {
        bool path_done = false;
        D3 nd = d;
        D3 p = o;
        if (res_prim == NONE) {
          sum = sum + thr * ld3(cam.background);
          path_done = true;
        } else {
          uint32_t si = res_prim & SPH_INDEX;
          const DevSphere &s = spheres[si];
          D3 c0 = ld3(s.c0);
          D3 center = (res_prim & SPH_MOVING) ? c0 + ld3(s.dc) * time : c0;
          p = o + d * res_t;
          D3 outward = (p - center) * s.inv_r;
          bool front = dot(d, outward) <= 0.0;
          D3 normal = front ? outward : -outward;
          const DevMaterial &m = P.materials[P.sphere_material[si]];
          uint32_t kind = m.kind;
          if (kind == RL_MAT_LAMBERTIAN) {
            D3 dir = normal + rng.unit_sphere();
            bool near_zero = approx_eq_eps(dir.x, 0.0, 1e-8) && approx_eq_eps(dir.y, 0.0, 1e-8) && approx_eq_eps(dir.z, 0.0, 1e-8);
            nd = near_zero ? normal : dir;
            thr = thr * texture_value(P, m.texture, 0.0, 0.0, p);
          } else if (kind == RL_MAT_METAL) {
            D3 reflected = d - normal * (2.0 * dot(d, normal));
            nd = normalize(reflected) + rng.unit_sphere() * m.fuzz;
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
              reflect = refl > rng.gen_f64();
            }
            if (reflect) nd = ud - normal * (2.0 * dot(ud, normal));
            else {
              D3 perp = (ud + normal * cos_theta) * ri;
              D3 par = normal * (-sqrt(fabs(1.0 - len2(perp))));
              nd = perp + par;
            }
          } else if (kind == RL_MAT_DIFFUSE_LIGHT) {
            sum = sum + thr * texture_value(P, m.texture, 0.0, 0.0, p);
            path_done = true;
          } else {
            path_done = true;  // Flat
          }
        }
        if (!path_done) {
          depth--;
          if (depth == 0) path_done = true;
        }
        if (path_done) {
          n++;
          cstate = ST_GEN;
        } else {
          c_rays++;
          pix_rays++;
          o = p;
          d = nd;
          post_ray();
        }
      }
    }
  }
  if (STATS && lane == 0) {
    unsigned long long *sched = P.stats + 8;  // [3*s] executions, [3*s+1] lanes served
#pragma unroll
    for (int s = 0; s < 8; s++) {
      atomicAdd(&sched[3 * s], sc_exec[s]);
      atomicAdd(&sched[3 * s + 1], sc_pop[s]);
      atomicAdd(&sched[3 * s + 2], sc_cyc[s]);
    }
  }

  unsigned long long v;
  v = wave_sum((unsigned long long)c_rays);
  if (lane == 0 && v) atomicAdd(&P.stats[0], v);
  v = wave_sum((unsigned long long)c_flag);
  if (lane == 0 && v) atomicAdd(&P.stats[6], v);
  if (STATS) {
    v = wave_sum(c_nodes);
    if (lane == 0) atomicAdd(&P.stats[1], v);
    v = wave_sum(c_sph);
    if (lane == 0) atomicAdd(&P.stats[2], v);
    v = wave_sum(c_words);
    if (lane == 0) atomicAdd(&P.stats[5], v);
  }
}

}  // namespace rl
