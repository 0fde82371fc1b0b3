// RTIOW all-primitives path in WAVEFRONT form — EXPERIMENTAL (librl_render_exp.so only, RL_WAVEFRONT=1 / RL_RTIOW_KERNEL variant 1035): built in
// round 3 as the "traversal-only kernel at twice the waves per SIMD" lever for BASELINE configs[4], bit-equal to the megakernel, and SLOWER:
// cfg 5 at 64 spp 3.0 - 3.4 s against 1.9 - 2.0 s (DESIGN.md section 3.2c has the per-kernel times and what they show: the traversal is not
// short of waves — wfg_trav at 4 / 5 / 6 / 8 waves per SIMD with every lane traversing reaches 1.5 Grays/s, the megakernel's overall rate; what
// bounds both is the L1's lane-access rate (each lane gathers its own 112-byte node in seven 16-byte accesses: doubling the accesses costs
// wfg_trav +37 %) and a pass can only end when its longest ray has).
//
// The megakernel of rl_rtiow_fastgen.h keeps a pixel's whole state in registers (168 VGPRs -> 3 waves per SIMD) and runs one of five
// states per wave at a time; on a scene that lives behind L2 (cfg 5: 150 MB of nodes, items and spheres) its waves spend half their
// cycles waiting for node fetches with a third of the lanes active (profiles/r03_roofline_cfg5.json).  Here the same arithmetic is cut
// into two kernels per PASS, one ray per unfinished pixel per pass:
//   wfg_logic  one lane = one pixel of the pass's queue: SHADE the ray that was just traced (the winner's HitRecord, material, scatter —
//              rl_rtiow_fastgen.h's SHADE block verbatim; order-sensitive rays through general_slow_trace) and / or GEN the pixel's next
//              sample (camera.rs:160-216), write the new ray, append the pixel to the next queue; a pixel whose last sample ended is
//              written out.  Register-hungry (binary64 HitRecord, textures), short, memory-streaming.
//   wfg_trav   persistent lanes pull rays from the queue and walk the four-wide reject-only tree (TRAV / LEAF of rl_rtiow_fastgen.h
//              verbatim): no RNG, no HitRecord, no pixel state -> half the registers, twice the waves per SIMD, every lane traversing.
// Per pixel nothing changes: samples in order, draws in order (the ChaCha word position travels in the pixel record, camera.rs:170),
// sums folded in order -> the frame is the megakernel's, bit for bit (tests/test_gpu_wavefront.py, bench.py's check).
// State per pixel in HBM: WfgPix 64 B + WfgRay 80 B (~300 B of traffic per ray, ~1 TB/s at 3 Grays/s).
#pragma once
#include "../rl_rtiow_fastgen.h"

namespace rl {

struct alignas(16) WfgPix {  // 64 B
  double sum[3];    // camera.rs:174, in sample order
  double thr[3];    // throughput of the path in flight
  uint32_t pos;     // ChaCha word position (kept across set_stream, camera.rs:170)
  uint32_t n;       // sample in flight (flags & 1) or next to start
  uint32_t depth;   // remaining depth of the path in flight
  uint32_t flags;   // bit 0: a ray of this pixel is in the queue
};
struct alignas(16) WfgRay {  // 80 B: [0, 64) written by wfg_logic, [64, 80) by wfg_trav (one 16-byte store)
  double o[3], d[3], time, pad;
  double closest;
  uint32_t best, amb;
};
static_assert(sizeof(WfgPix) == 64 && sizeof(WfgRay) == 80, "record sizes");

// control words, one per 64-byte line: [0] / [1] entries in queue 0 / 1, [2] wfg_logic's claim cursor, [3] wfg_trav's, [4] pixels finished
// [5] entries in the slow queue (pixels whose ray must be re-traced in the reference's order), [6] wfg_logic<SLOW>'s claim cursor
enum : uint32_t { WFG_COUNT0 = 0, WFG_COUNT1 = 16, WFG_CUR_LOGIC = 32, WFG_CUR_TRAV = 48, WFG_DONE = 64, WFG_COUNT_SLOW = 80, WFG_CUR_SLOW = 96, WFG_CTL_WORDS = 112 };

static const uint32_t WFG_CLAIM = 256;
struct WfgParams {
  WfgPix *pix;
  WfgRay *ray;
  uint32_t *queue[2];
  uint32_t *slow_queue;
  uint32_t *ctl;
  uint32_t in;  // the queue wfg_logic reads in this pass (wfg_trav reads the other one)
};

// queue 0 <- every pixel of the shard in 8 x 8 tile order (NONE for slots outside the image), fresh pixel records
__global__ void wfg_init(RtiowParams P, WfgParams Q) {
  uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot == 0) {
    Q.ctl[WFG_COUNT0] = P.n_slots, Q.ctl[WFG_COUNT1] = 0, Q.ctl[WFG_CUR_LOGIC] = 0, Q.ctl[WFG_CUR_TRAV] = 0, Q.ctl[WFG_DONE] = 0;
    Q.ctl[WFG_COUNT_SLOW] = 0, Q.ctl[WFG_CUR_SLOW] = 0;
  }
  if (slot >= P.n_slots) return;
  const uint32_t W = P.cam.image_width;
  uint32_t tile = slot >> 6, in = slot & 63u;
  uint32_t px = (tile % P.tiles_x) * 8u + (in & 7u), pr = (tile / P.tiles_x) * 8u + (in >> 3);
  if (px >= W || pr >= P.nrows) {
    Q.queue[0][slot] = NONE;
    return;
  }
  uint32_t pix = pr * W + px;
  Q.queue[0][slot] = pix;
  WfgPix st;
  st.sum[0] = st.sum[1] = st.sum[2] = 0.0;
  st.thr[0] = st.thr[1] = st.thr[2] = 1.0;
  st.pos = 0, st.n = P.sample_begin, st.depth = 0, st.flags = 0;
  Q.pix[pix] = st;
}

// ---------------------------------------------------------------------------------------------------------------- wfg_logic
// SLOW = false: the pass's queue; a pixel whose ray is order-sensitive (flagged by wfg_trav, or a grazing / pole hit of the winner) is
// only appended to the slow queue.  SLOW = true (launched right after): the slow queue — every lane re-traces its ray with the reference's
// own fold (~150 dependent fetches), so those lanes no longer hold up waves of ordinary pixels.
template <int NT, bool TRANS, bool SLOW>
__global__ void __launch_bounds__(NT, (SLOW || TRANS) ? 2 : 3) wfg_logic(RtiowParams P, WfgParams Q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  unsigned long long *s_rng = (unsigned long long *)smem;  // [16][NT]
  const DevOp *ops = P.ops;
  const FastItem *items = P.fg_items;
  const rl_rtiow_camera &cam = P.cam;
  const uint32_t W = cam.image_width;
  const uint32_t spp = P.sample_end;
  const uint64_t WH = (uint64_t)cam.image_width * (uint64_t)cam.image_height;
  const double INF = __longlong_as_double(0x7FF0000000000000ll);
  const float FINF = __int_as_float(0x7F800000);
  const uint32_t n_in = SLOW ? Q.ctl[WFG_COUNT_SLOW] : Q.ctl[Q.in ? WFG_COUNT1 : WFG_COUNT0];
  const uint32_t *q_in = SLOW ? Q.slow_queue : Q.queue[Q.in];
  uint32_t *q_out = Q.queue[Q.in ^ 1u];
  uint32_t *count_out = &Q.ctl[Q.in ? WFG_COUNT0 : WFG_COUNT1];
  if (!SLOW && blockIdx.x == 0 && tid == 0) Q.ctl[WFG_CUR_TRAV] = 0;  // (wfg_trav is not running now)
  Ring<NT> rng{P.key, s_rng, tid, 0ull, 0u, 0u, 0u};
  uint32_t c_rays = 0, c_flag = 0, c_slow = 0, c_done = 0;

  // (no claim counter: the queue's length is known and a pixel's SHADE + GEN costs about the same everywhere)
  for (uint32_t idx = blockIdx.x * NT + tid; __ballot(idx < n_in) != 0ull; idx += gridDim.x * NT) {
    const uint32_t pix = idx < n_in ? q_in[idx] : NONE;
    bool emit = false, defer = false;
    if (pix != NONE) {
      WfgPix st = Q.pix[pix];
      const uint32_t px = pix % W, pr = pix / W;
      const uint32_t y = P.row_first + pr * P.row_step;
      D3 sum = ld3(st.sum), thr = ld3(st.thr);
      uint32_t n = st.n, depth = st.depth;
      rng.pos = st.pos;
      D3 wo = d3(0.0, 0.0, 0.0), wd = d3(0.0, 0.0, 1.0);
      double time = 0.0;
      bool has_ray = false;
      if (st.flags & 1u) {  // ---- SHADE the ray wfg_trav has just finished (rl_rtiow_fastgen.h SHADE, same arithmetic)
        const WfgRay rr = Q.ray[pix];
        wo = ld3(rr.o), wd = ld3(rr.d), time = rr.time;
        const double closest = rr.closest;
        const uint32_t best = rr.best;
        const bool amb = rr.amb != 0u;
        // the sample's stream (camera.rs:167-170); only the block the word position points into is generated here, next_u64 tops up
        rng.stream = ((uint64_t)n + P.first_sample) * WH + (uint64_t)px * (uint64_t)W + (uint64_t)y;
        rng.blk_lo = rng.pos >> 4;
        rng.gen_block(rng.blk_lo);
        rng.nres = 1;
        bool path_done = false;
        D3 nd = wd;
        Rec rec;
        rec.t = INF, rec.any = false, rec.pc = 0, rec.mat = 0, rec.u = 0.0, rec.v = 0.0, rec.w = 0.0, rec.uv3 = false, rec.front = true;
        rec.p = d3(0.0, 0.0, 0.0), rec.normal = d3(0.0, 0.0, 0.0);
        if (SLOW) {  // the answer may depend on the visiting order -> the reference's own fold
          c_flag += general_slow_trace<TRANS>(P, ops, wo, wd, time, rec);
          c_slow++;
        } else if (amb) defer = true;
        else if (best != NONE) {
          bool push_skip = false;
          const FastItem it = items[best];
          const DevSphere sp = P.fg_spheres[best];
          const uint32_t wmat = P.fg_material[best];
          D3 o, d;
          replay_chain(P, ops, it.chain, wo, wd, o, d);
          rec.t = closest;
          bool sensitive = false;
          if (it.kind == 0) {
            sphere_hit_rec(sp, it.payload, wmat, it.op_pc, o, d, time, rec);
            D3 c0 = ld3(sp.c0);
            D3 center = (it.payload & SPH_MOVING) ? c0 + ld3(sp.dc) * time : c0;
            const RayAux32 ra32 = ray_aux32_direct(wo, wd);
            float oimax = ra32.oimax();
            if (it.chain != NONE)
              oimax = fmaxf(fmaxf(fabsf((float)o.x * __builtin_amdgcn_rcpf((float)d.x)), fabsf((float)o.y * __builtin_amdgcn_rcpf((float)d.y))),
                            fabsf((float)o.z * __builtin_amdgcn_rcpf((float)d.z)));
            if (!(oimax < FINF)) oimax = FINF;
            {
              D3 oc = o - center;
              double half_b = dot(oc, d), sq = sp.r2 * sp.inv_r * fabs(dot(d, rec.normal)), a = len2(d);
              double other = 2.0 * sq * (double)__builtin_amdgcn_rcpf((float)a);
              sensitive = fast_hit_is_order_sensitive(oc, d, closest, sp.r2 * sp.inv_r, half_b, sq, closest, fabs(closest) + other, oimax);
            }
          } else planar_hit_rec(P.planars[it.payload], it.op_pc, o, d, rec);
          if (sensitive) defer = true, push_skip = true;  // grazing / pole hit of the winner: the slow queue
          uint32_t push_pc = push_skip ? NONE : it.chain;
#pragma unroll 1
          while (push_pc != NONE) {
            const DevOp &op = ops[push_pc];
            if ((op.code & 0xFFu) == OP_PUSH_TRANSLATE) rec.p = rec.p + ld3(P.translates[op.a].offset);
            else {
              const rl_transform &t = P.transforms[op.a];
              rec.p = mat3_mul(t.m, rec.p);
              D3 wn = mat3_mul(t.inv_t, rec.normal);
              double m = len2(wn);
              if (approx_eq_eps(m, 0.0, 1e-16)) c_flag++;
              else rec.normal = normalize(wn);
            }
            push_pc = op.b;
          }
        }
        D3 p = rec.p;
        if (defer) {
        } else if (!rec.any) {
          sum = sum + thr * ld3(cam.background);
          path_done = true;
        } else {
          const DevMaterial &m = P.materials[rec.mat];
          D3 texc = d3(0.0, 0.0, 0.0);
          if (m.kind == RL_MAT_LAMBERTIAN || m.kind == RL_MAT_DIFFUSE_LIGHT) {
            double tu, tv;
            rec_uv<TRANS>(rec, tu, tv);
            texc = texture_value<(TRANS ? 2 : 1)>(P, m.texture, tu, tv, rec.p);
          }
          uint32_t kind = m.kind;
          D3 normal = rec.normal;
          if (kind == RL_MAT_LAMBERTIAN) {
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
        if (path_done) n++;
        else wo = p, wd = nd, has_ray = true;
      }
      if (defer) has_ray = false, n = spp;  // nothing of the pixel's record has changed: the SLOW launch of this pass starts over from it
      // ---- GEN: the pixel's next sample (camera.rs:160-216), until one of them has a ray to trace (max_depth 0: none ever has)
#pragma unroll 1
      while (!has_ray && n < spp) {
        rng.reset_stream(((uint64_t)n + P.first_sample) * WH + (uint64_t)px * (uint64_t)W + (uint64_t)y);
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
        else has_ray = true;
      }
      if (defer) {
      } else if (has_ray) {
        c_rays++;
        WfgRay rr;
        rr.o[0] = wo.x, rr.o[1] = wo.y, rr.o[2] = wo.z, rr.d[0] = wd.x, rr.d[1] = wd.y, rr.d[2] = wd.z, rr.time = time;
        rr.pad = 0.0, rr.closest = INF, rr.best = NONE, rr.amb = 0u;
        Q.ray[pix] = rr;
        st.sum[0] = sum.x, st.sum[1] = sum.y, st.sum[2] = sum.z, st.thr[0] = thr.x, st.thr[1] = thr.y, st.thr[2] = thr.z;
        st.pos = rng.pos, st.n = n, st.depth = depth, st.flags = 1u;
        Q.pix[pix] = st;
        emit = true;
      } else {  // every sample done: Canvas.data (camera.rs:193-198)
        double *outp = P.out + (size_t)pix * 3;
        outp[0] = sum.x, outp[1] = sum.y, outp[2] = sum.z;
        if (P.pos_state) P.pos_state[pix] = rng.pos;
        c_done++;
      }
    }
    if (!SLOW) {  // order-sensitive rays: the slow queue
      const unsigned long long dm = __ballot(defer);
      if (dm) {
        const uint32_t lane = __lane_id();
        uint32_t base = 0;
        if (lane == (uint32_t)__ffsll((long long)dm) - 1u) base = atomicAdd(&Q.ctl[WFG_COUNT_SLOW], (uint32_t)__popcll(dm));
        base = __shfl(base, __ffsll((long long)dm) - 1, 64);
        if (defer) Q.slow_queue[base + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull))] = pix;
      }
    }
    // append the pixels that have a ray to the next queue: one atomic per wave
    const unsigned long long em = __ballot(emit);
    if (em) {
      const uint32_t lane = __lane_id();
      uint32_t base = 0;
      if (lane == (uint32_t)__ffsll((long long)em) - 1u) base = atomicAdd(count_out, (uint32_t)__popcll(em));
      base = __shfl(base, __ffsll((long long)em) - 1, 64);
      if (emit) q_out[base + (uint32_t)__popcll(em & ((1ull << lane) - 1ull))] = pix;
    }
  }
  unsigned long long v;
  v = wave_sum((unsigned long long)c_rays);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[0], v);
  v = wave_sum((unsigned long long)c_flag);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[6], v);
  v = wave_sum((unsigned long long)c_slow);
  if ((tid & 63) == 0 && v) atomicAdd(&P.stats[7], v);
  v = wave_sum((unsigned long long)c_done);
  if ((tid & 63) == 0 && v) atomicAdd(&Q.ctl[WFG_DONE], (uint32_t)v);
}

// ---------------------------------------------------------------------------------------------------------------- wfg_trav
// SD = entries of the per-lane LDS stack; WPE = waves per SIMD the register budget is computed for
template <int NT, int SD, int WPE>
__global__ void __launch_bounds__(NT, WPE) wfg_trav(RtiowParams P, WfgParams Q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  uint32_t *s_stack = (uint32_t *)smem;  // [SD][NT]
  const DevOp *ops = P.ops;
  const FastNodeQ *nodes = P.fg_nodes;
  const FastItem *items = P.fg_items;
  const double INF = __longlong_as_double(0x7FF0000000000000ll);
  const float FINF = __int_as_float(0x7F800000);
  const uint32_t n_q = Q.ctl[Q.in ? WFG_COUNT0 : WFG_COUNT1];  // what wfg_logic has just appended
  const uint32_t *queue = Q.queue[Q.in ^ 1u];
  if (blockIdx.x == 0 && tid == 0)  // (wfg_logic is not running now)
    Q.ctl[WFG_CUR_LOGIC] = 0, Q.ctl[Q.in ? WFG_COUNT1 : WFG_COUNT0] = 0, Q.ctl[WFG_COUNT_SLOW] = 0, Q.ctl[WFG_CUR_SLOW] = 0;

  uint32_t state = ST_GEN;  // ST_GEN here = "store the finished ray's result, fetch the next ray"
  uint32_t cur = NONE;
  // queue indices are claimed WFG_CLAIM at a time per wave (one memory-side atomic round trip per 256 rays instead of one per GEN round)
  uint32_t pool_base = 0, pool_left = 0;
  D3 wo = d3(0.0, 0.0, 0.0), wd = d3(0.0, 0.0, 1.0);
  RayAux32 ra32 = ray_aux32_direct(wo, wd);
  double time = 0.0, closest = INF;
  uint32_t pc = 0, best = NONE, sp = 0, steps = 0;
  const uint32_t step_budget = P.tune[3];
  bool amb = false, unsafe = false;
  float grow = 0.0f;

  auto go = [&](uint32_t e) {
    // FASTG_STEP_BUDGET (rl_rtiow_fastgen.h): a walk this long is handed to the reference's own fold, whatever it has found so far
    if (++steps > step_budget) amb = true, e = NONE;
    if (e == NONE) state = ST_GEN;
    else {
      pc = e;
      state = (e & FASTG_LEAF) ? ST_LEAF : ST_TRAV;
    }
  };
  auto pop = [&]() -> uint32_t {
    if (sp == 0) return NONE;
    sp--;
    return s_stack[(size_t)sp * NT + tid];
  };
  auto start_ray = [&]() {  // rl_rtiow_fastgen.h start_ray
    closest = INF, best = NONE, sp = 0, steps = 0;
    ra32 = ray_aux32_direct(wo, wd);
    float fx = (float)wo.x - P.fg_center[0], fy = (float)wo.y - P.fg_center[1], fz = (float)wo.z - P.fg_center[2];
    float far2 = fmaf(fx, fx, fmaf(fy, fy, fz * fz));
    amb = !(ra32.slack < FINF);
    unsafe = !(far2 <= P.fg_rsafe2);
    grow = 0.0f;
    if (unsafe) {
      float L = sqrtf(far2) + P.fg_radius;
      grow = P.fg_pad_k * L * L * fmaxf(fmaxf(fabsf(ra32.invx), fabsf(ra32.invy)), fabsf(ra32.invz));
      if (!(grow < FINF)) amb = true;
    }
    go(amb ? NONE : P.fg_root);
  };

  for (;;) {
    int n_trav = __popcll(__ballot(state == ST_TRAV));
    int n_gen = __popcll(__ballot(state == ST_GEN));
    int n_leaf = __popcll(__ballot(state == ST_LEAF));
    if ((n_trav | n_gen | n_leaf) == 0) break;
    uint32_t pick = ST_TRAV;
    int bestn = n_trav;
    if (n_leaf > bestn) pick = ST_LEAF, bestn = n_leaf;
    if (n_gen > bestn) pick = ST_GEN, bestn = n_gen;

    if (pick == ST_TRAV) {
      int floor_n = (bestn * (int)P.tune[1]) >> 4;
      for (int it = 0; it < (int)P.tune[0]; it++) {
        if (state == ST_TRAV) {
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
          const Float4 *nd = (const Float4 *)(nodes + pc);
          const Float4 lx = nd[0], ly = nd[1], lz = nd[2], hx = nd[3], hy = nd[4], hz = nd[5];
          const uint4 ch = *(const uint4 *)(nd + 6);
          float k0, k1, k2, k3;
          const bool h0 = !missed(lx.x, hx.x, ly.x, hy.x, lz.x, hz.x, k0);
          const bool h1 = !missed(lx.y, hx.y, ly.y, hy.y, lz.y, hz.y, k1);
          const bool h2 = !missed(lx.z, hx.z, ly.z, hy.z, lz.z, hz.z, k2) && ch.z != NONE;
          const bool h3 = !missed(lx.w, hx.w, ly.w, hy.w, lz.w, hz.w, k3) && ch.w != NONE;
          const int nh = (int)h0 + (int)h1 + (int)h2 + (int)h3;
          k0 = h0 ? k0 : FINF, k1 = h1 ? k1 : FINF, k2 = h2 ? k2 : FINF, k3 = h3 ? k3 : FINF;
          uint32_t c0 = ch.x, c1 = ch.y, c2 = ch.z, c3 = ch.w;
          uint32_t u0 = (__float_as_uint(k0) & ~1u) | (h0 ? 0u : 1u), u1 = (__float_as_uint(k1) & ~1u) | (h1 ? 0u : 1u);
          uint32_t u2 = (__float_as_uint(k2) & ~1u) | (h2 ? 0u : 1u), u3 = (__float_as_uint(k3) & ~1u) | (h3 ? 0u : 1u);
          auto cex = [&](uint32_t &ka, uint32_t &kb, uint32_t &ca, uint32_t &cb) {
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
        if (__popcll(__ballot(state == ST_TRAV)) < floor_n) break;
      }
    } else if (pick == ST_LEAF) {
      if (state == ST_LEAF) {
        const uint32_t item = pc & ~FASTG_LEAF;
        const FastItem it = items[item];
        const DevSphere isph = P.fg_spheres[item];
        D3 o, d;
        replay_chain(P, ops, it.chain, wo, wd, o, d);
        float oimax = ra32.oimax();
        if (it.chain != NONE)
          oimax = fmaxf(fmaxf(fabsf((float)o.x * __builtin_amdgcn_rcpf((float)d.x)), fabsf((float)o.y * __builtin_amdgcn_rcpf((float)d.y))),
                        fabsf((float)o.z * __builtin_amdgcn_rcpf((float)d.z)));
        if (!(oimax < FINF)) oimax = FINF;
        if (it.kind == 0) fastg_sphere_hit(isph, it.payload, o, d, time, oimax, item, closest, best, amb);
        else fastg_planar_hit(P.planars[it.payload], o, d, oimax, item, closest, best, amb);
        go(pop());
      }
    } else {  // ST_GEN: hand the finished ray back, take the next one (pool_base / pool_left are wave-uniform: every lane updates them)
      const bool g = state == ST_GEN;
      if (g && cur != NONE) {
        const unsigned long long cbits = (unsigned long long)__double_as_longlong(closest);
        *(uint4 *)&Q.ray[cur].closest = uint4{(uint32_t)cbits, (uint32_t)(cbits >> 32), best, amb ? 1u : 0u};
        cur = NONE;
      }
      if (pool_left == 0u) {
        uint32_t b = 0;
        if (__lane_id() == 0u) b = atomicAdd(&Q.ctl[WFG_CUR_TRAV], WFG_CLAIM);
        pool_base = __builtin_amdgcn_readfirstlane(b), pool_left = WFG_CLAIM;
      }
      const unsigned long long m = __ballot(g);
      const uint32_t rank = (uint32_t)__popcll(m & ((1ull << __lane_id()) - 1ull));
      const uint32_t take = min((uint32_t)__popcll(m), pool_left);
      if (g && rank < take) {  // (the others stay in ST_GEN: the pool is refilled in the next round)
        const uint32_t idx = pool_base + rank;
        if (idx >= n_q) state = ST_DONE;
        else {
          cur = queue[idx];
          const WfgRay *r = Q.ray + cur;
          wo = ld3(r->o), wd = ld3(r->d), time = r->time;
          start_ray();
        }
      }
      pool_base += take, pool_left -= take;
    }
  }
}

}  // namespace rl
