// rl_render.hip — the C ABI of include/rl_render.h over the gfx950 kernels (single-device entry points; rl_multi.hip drives
// several GPUs from one process on top of the launch halves defined here).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see csrc/Makefile).
// No CPU fallback: every compute entry point fails with RL_E_NO_DEVICE unless rl_init succeeded on a GPU.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "rl_scene.h"
#include "rl_rtc_kernel.h"
#include "rl_rtc_full_kernel.h"
#include "rl_rtiow_kernel.h"
#include "rl_rtiow_general.h"
#include "rl_rtiow_wave.h"
#include "rl_rtiow_wave_general.h"
#include "rl_rtiow_fastgen.h"
#include "rl_rtiow_coop.h"
#ifdef RL_EXPERIMENTAL  // the measured-and-lost restructurings: only in librl_render_exp.so, never in the product library
#include "experimental/rl_rtiow_pool.h"
#include "experimental/rl_rtiow_wave2.h"
#include "experimental/rl_rtiow_wavefront.h"
#include "experimental/rl_rtiow_wfg.h"
#endif

using namespace rl;

namespace {

thread_local std::string g_err;
std::mutex g_mu;
bool g_ready = false;
std::vector<DevCtx> g_ctx;  // [0] is the device rl_init chose; rl_init_multi appends the others
int g_cus = 0;
size_t g_lds_max = 65536;
// Every RL_* environment switch (A/B and debugging only: DESIGN.md section 3.6) is read ONCE, by rl_init; the render path reads this struct.
struct Switches {
  int rtiow_variant = 0;         // RL_RTIOW_KERNEL (0 = automatic)
  bool lpt = true;               // RL_LPT=0: single launch instead of the cost-sorted two-phase render
  bool coop_small = true;        // RL_COOP=0: small frames through the wave-scheduled kernel instead of the cooperative one
  double steal_max_fill = 3.0;   // RL_STEAL=<pixels per lane> (0 = off): work stealing on small shards
  bool fast_traversal = true;    // RL_FAST=0: counter-free renders use the reference-order kernels too
  unsigned fastg_top_max = 512;  // RL_FASTG_TOP=<n > 1>: at most n nodes
  int rtc_blocks_per_cu = 0;     // RL_RTC_BLOCKS=<n>: workgroups per CU the RTC kernels are launched with (0 = as many as are resident at once: occupancy API)
  int fastg_nt256 = -1;          // RL_FASTG_NT256=0|1: never / always the one-wave-per-SIMD form of the fast general kernel (default: by frame size)
  bool fastg_top = true;         // RL_FASTG_TOP=0: the fast general kernel reads every node through L1 (A/B of the LDS tree top)
  bool tune_set = false;         // RL_TUNE="steps,floor16[,batch,fill]"
  unsigned tune[4] = {24, 6, 24, 40};
  unsigned blocks_cap = 0;       // RL_BLOCKS
  int coop_mode = -1;            // RL_COOP_MODE
  int general_regs = 512;        // RL_GENERAL_REGS (256 / 768: experimental library only)
  bool fastg512 = false;         // RL_FASTG512 (experimental library only)
  int fastg_octo = 0;            // RL_FASTG_OCTO=1: the eight-wide quantised nodes instead of the four-wide ones (experimental library only)
  int general_nt = 512;          // RL_GENERAL_NT (768: experimental library only)
  double thin_permille = 0.0, prio_permille = 0.0;  // RL_THIN / RL_PRIO (experimental library only)
  int thin_shift = 2;            // RL_THIN_SHIFT
  int wavefront = -1;            // RL_WAVEFRONT=1: general fast-traversal scenes in wavefront form (experimental library only)
  bool rtc_force_full = false;   // RL_RTC_FORCE_FULL
  int rtc_regs = 256;            // RL_RTC_REGS=768|1024: rtc_kernel at the register budget of three / four waves per SIMD (experimental library only)
  int rtc_full_regs = 768;       // RL_RTC_FULL_REGS (256 / 512: experimental library only)
} g_sw;
unsigned long long g_last_slow_traces = 0;
bool g_fast_debug_stats = false;  // tools only (experimental library): counting renders run the fast kernel too (counters are then NOT the reference's)

void read_switches() {
  Switches w;
  if (const char *v = std::getenv("RL_RTIOW_KERNEL")) {
    std::string sv(v);
    w.rtiow_variant = sv == "v1" ? 1 : sv == "general" ? 2 : sv == "wavefront" ? 3 : sv == "wavegeneral" ? 4 : sv == "pool" ? 5 : sv == "pool256" ? 6 : sv == "wave2" ? 7 : sv == "wave256" ? 256 : sv == "wave512" ? 512 : sv == "wave768" ? 768 : sv == "wave1024" ? 1024 : sv == "wave1024ops" ? 1025 : sv == "wave1024guard" ? 1027 : sv == "wave1024fast" ? 1029 : sv == "coop" ? 1033 : 0;
  }
  if (const char *v = std::getenv("RL_LPT")) w.lpt = std::string(v) != "0";
  if (const char *v = std::getenv("RL_COOP")) w.coop_small = std::string(v) != "0";
  if (const char *v = std::getenv("RL_STEAL")) w.steal_max_fill = std::atof(v);
  if (const char *v = std::getenv("RL_FAST")) w.fast_traversal = std::string(v) != "0";
  if (const char *t = std::getenv("RL_TUNE")) {
    unsigned a = 16, b = 12, c = 24, d = 40;
    int nf = std::sscanf(t, "%u,%u,%u,%u", &a, &b, &c, &d);
    if (nf >= 2) w.tune[0] = a, w.tune[1] = b, w.tune_set = true;
    if (nf >= 3) w.tune[2] = c;
    if (nf >= 4) w.tune[3] = d;
  }
  if (const char *v = std::getenv("RL_BLOCKS")) w.blocks_cap = (unsigned)std::atoi(v);
  if (const char *v = std::getenv("RL_COOP_MODE")) w.coop_mode = std::atoi(v);
  if (const char *v = std::getenv("RL_GENERAL_REGS")) w.general_regs = std::atoi(v);
  w.fastg512 = std::getenv("RL_FASTG512") != nullptr;
  if (const char *v = std::getenv("RL_FASTG_OCTO")) w.fastg_octo = std::atoi(v);
  if (const char *v = std::getenv("RL_RTC_BLOCKS")) w.rtc_blocks_per_cu = std::max(0, std::atoi(v));
  if (const char *v = std::getenv("RL_FASTG_NT256")) w.fastg_nt256 = std::atoi(v) != 0;
  if (const char *v = std::getenv("RL_FASTG_TOP")) w.fastg_top = std::atoi(v) != 0, w.fastg_top_max = (unsigned)std::atoi(v) > 1u ? (unsigned)std::atoi(v) : 512u;
  if (const char *v = std::getenv("RL_GENERAL_NT")) w.general_nt = std::atoi(v);
  if (const char *v = std::getenv("RL_THIN")) w.thin_permille = std::atof(v);
  if (const char *v = std::getenv("RL_PRIO")) w.prio_permille = std::atof(v);
  if (const char *v = std::getenv("RL_THIN_SHIFT")) w.thin_shift = std::min(6, std::max(1, std::atoi(v)));
  if (const char *v = std::getenv("RL_WAVEFRONT")) w.wavefront = std::atoi(v);
  w.rtc_force_full = std::getenv("RL_RTC_FORCE_FULL") != nullptr;
  if (const char *v = std::getenv("RL_RTC_REGS")) w.rtc_regs = std::atoi(v);
  if (const char *v = std::getenv("RL_RTC_FULL_REGS")) w.rtc_full_regs = std::atoi(v);
  g_sw = w;
#ifdef RL_EXPERIMENTAL
  rl::set_build_octo(w.fastg_octo != 0);
#endif
}

// hipFuncAttributeMaxDynamicSharedMemorySize is sticky per (kernel, device): set it when a launch needs more than any before it did
std::map<std::pair<const void *, int>, size_t> g_lds_attr;
int ensure_lds_attr(const void *kern, size_t lds) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> lk(g_mu);
  size_t &have = g_lds_attr[{kern, dev}];
  if (lds <= have) return 0;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
  have = lds;
  return 0;
}

int set_err(int code, const std::string &m) {
  g_err = m;
  return code;
}
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return set_err(RL_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

template <class T>
int upload(const std::vector<T> &v, T **out) {
  *out = nullptr;
  size_t bytes = (v.size() ? v.size() : 1) * sizeof(T);
  HIP_TRY(hipMalloc((void **)out, bytes));
  if (!v.empty()) HIP_TRY(hipMemcpy(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return RL_OK;
}

__global__ void iota_u32(uint32_t *out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = i;
}

// Output stages
__global__ void encode_rtiow_rgb8(const double *sum, unsigned long long n_vals, double inv_samples, unsigned char *out) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_vals) return;
  double v = sum[i] * inv_samples;                                                 // camera.rs:293
  double s = v <= 0.0031308 ? 12.92 * v : (1.0 + 0.055) * pow(v, 1.0 / 2.4) - 0.055;  // color.rs:130-136
  double f = floor(s * 255.999);                                                   // color.rs:47-50 (as i16 saturates; NaN -> 0)
  int q = isnan(f) ? 0 : (f >= 32767.0 ? 32767 : (f <= -32768.0 ? -32768 : (int)f));
  out[i] = (unsigned char)(q < 0 ? 0 : (q > 255 ? 255 : q));
}
__global__ void encode_rtc_rgb8(const double *rgb, unsigned long long n_vals, unsigned char *out) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_vals) return;
  double v = round(rgb[i] * 255.0);  // canvas.rs:53-56: f64::round = half away from zero; as i32 saturates
  int q = isnan(v) ? 0 : (v >= 2147483647.0 ? 2147483647 : (v <= -2147483648.0 ? (int)-2147483648LL : (int)v));
  out[i] = (unsigned char)(q < 0 ? 0 : (q > 255 ? 255 : q));
}

int init_context(DevCtx &c, int device) {
  c.device = device;
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return set_err(RL_E_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
  g_cus = prop.multiProcessorCount;
  g_lds_max = prop.sharedMemPerBlock > 65536 ? prop.sharedMemPerBlock : 65536;
  if (prop.maxSharedMemoryPerMultiProcessor > g_lds_max) g_lds_max = prop.maxSharedMemoryPerMultiProcessor;
  if (g_lds_max > 163840) g_lds_max = 163840;
  if (!c.stream) HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  if (!c.ev) HIP_TRY(hipEventCreateWithFlags(&c.ev, hipEventDisableTiming));
  return RL_OK;
}

}  // namespace

namespace rl {
int set_err_public(int code, const std::string &m) { return set_err(code, m); }  // for rl_bvh_build.hip / rl_multi.hip
bool lib_ready() { return g_ready; }
int n_contexts() { return (int)g_ctx.size(); }
DevCtx &context(int i) { return g_ctx[(size_t)i]; }
int use_context(int i) {
  HIP_TRY(hipSetDevice(g_ctx[(size_t)i].device));
  return RL_OK;
}
// rl_multi.hip: (re)build the context list — devices[g] for context g; context 0 keeps its stream when the device is unchanged
int set_contexts(const std::vector<int> &devices) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (size_t i = devices.size(); i < g_ctx.size(); i++) {
    if (g_ctx[i].stream && hipSetDevice(g_ctx[i].device) == hipSuccess) hipStreamDestroy(g_ctx[i].stream), hipEventDestroy(g_ctx[i].ev);
  }
  g_ctx.resize(devices.size());
  for (size_t i = 0; i < devices.size(); i++) {
    if (g_ctx[i].stream && g_ctx[i].device != devices[i]) {
      hipSetDevice(g_ctx[i].device);
      hipStreamDestroy(g_ctx[i].stream), hipEventDestroy(g_ctx[i].ev);
      g_ctx[i] = DevCtx{};
    }
    int rc = init_context(g_ctx[i], devices[i]);
    if (rc != RL_OK) return rc;
  }
  if (!g_ctx.empty()) hipSetDevice(g_ctx[0].device);
  return RL_OK;
}
int sort_tiles_by_cost_desc(const uint32_t *d_cost, uint32_t *d_keys_tmp, uint32_t *d_order_in, uint32_t *d_order_out, uint32_t n, void **temp, size_t *temp_bytes,
                            hipStream_t stream);  // rl_bvh_build.hip (hipCUB)
void add_stats(rl_stats *acc, const rl_stats &s) {
  acc->rays += s.rays, acc->node_tests += s.node_tests, acc->sphere_tests += s.sphere_tests, acc->planar_tests += s.planar_tests;
  acc->instance_enters += s.instance_enters, acc->rng_words += s.rng_words, acc->flagged += s.flagged;
  acc->kernel_ms = std::max(acc->kernel_ms, s.kernel_ms);
}
}  // namespace rl

#ifdef RL_EXPERIMENTAL
struct ExpBuffers {  // wavefront (v3) buffers: per-pixel state, ray and hit records, queues, control words
  PixState *wf_pix = nullptr;
  RayRec *wf_ray = nullptr;
  HitRec *wf_hit = nullptr;
  uint32_t *wf_qtrav = nullptr, *wf_qshade = nullptr, *wf_qgen = nullptr, *wf_ctl = nullptr;
  size_t wf_npix = 0;
};
#endif

extern "C" {

int rl_abi_version(void) { return RL_ABI_VERSION; }
const char *rl_last_error(void) { return g_err.c_str(); }

int rl_init(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return set_err(RL_E_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (device >= n) return set_err(RL_E_INVALID, "device index out of range");
  if (device < 0) HIP_TRY(hipGetDevice(&device));
  rl::drop_multi_state();
  int rc = rl::set_contexts({device});
  if (rc != RL_OK) return rc;
  read_switches();
  g_ready = true;
  return RL_OK;
}

void rl_shutdown(void) {
  rl::drop_multi_state();
  std::lock_guard<std::mutex> lk(g_mu);
  for (DevCtx &c : g_ctx)
    if (c.stream && hipSetDevice(c.device) == hipSuccess) hipStreamDestroy(c.stream), hipEventDestroy(c.ev);
  g_ctx.clear();
  g_ready = false;
}

int rl_device_info(char *name, int cap) {
  if (!g_ready) return set_err(RL_E_NO_DEVICE, "rl_init has not succeeded");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, g_ctx[0].device));
  if (name && cap > 0) {
    std::strncpy(name, prop.gcnArchName, (size_t)cap - 1);
    name[cap - 1] = 0;
  }
  return prop.multiProcessorCount;
}

static void destroy_one(rl_scene *s) {
  if ((size_t)s->ctx < g_ctx.size()) hipSetDevice(g_ctx[(size_t)s->ctx].device);
  hipFree(s->d_ops), hipFree(s->d_lops), hipFree(s->d_sphere_flat), hipFree(s->d_cops), hipFree(s->d_movbits), hipFree(s->d_spheres), hipFree(s->d_sphere_material), hipFree(s->d_planars), hipFree(s->d_translates);
  hipFree(s->d_transforms), hipFree(s->d_materials), hipFree(s->d_textures), hipFree(s->d_images), hipFree(s->d_image_pool), hipFree(s->d_perlins), hipFree(s->d_media);
  hipFree(s->d_tris), hipFree(s->d_xforms), hipFree(s->d_rmaterials), hipFree(s->d_lights), hipFree(s->d_scratch);
  hipFree(s->d_pos), hipFree(s->d_tile_cost), hipFree(s->d_tile_order), hipFree(s->d_tile_keys), hipFree(s->d_tile_iota), hipFree(s->d_sort_temp);
  hipFree(s->d_shapes), hipFree(s->d_csgs), hipFree(s->d_patterns), hipFree(s->d_guards), hipFree(s->d_shard), hipFree(s->d_pix_rays), hipFree(s->d_fast_nodes), hipFree(s->d_fast_leaf_boxes), hipFree(s->d_coop_pixels), hipFree(s->d_steal_state), hipFree(s->d_steal_n), hipFree(s->d_fg_nodes), hipFree(s->d_fg_onodes), hipFree(s->d_fg_seg_roots), hipFree(s->d_fg_media), hipFree(s->d_fg_items), hipFree(s->d_fg_spheres), hipFree(s->d_fg_material);
#ifdef RL_EXPERIMENTAL
  if (ExpBuffers *E = (ExpBuffers *)s->exp) {
    hipFree(E->wf_pix), hipFree(E->wf_ray), hipFree(E->wf_hit), hipFree(E->wf_qtrav), hipFree(E->wf_qshade), hipFree(E->wf_qgen), hipFree(E->wf_ctl);
    delete E;
  }
#endif
  if (s->h_status) hipHostFree(s->h_status);
  for (hipEvent_t e : s->ev_status)
    if (e) hipEventDestroy(e);
  if (s->ev_last) hipEventDestroy(s->ev_last);
  if (s->h_progress) hipHostFree(s->h_progress);
  if (s->ev0) hipEventDestroy(s->ev0);
  if (s->ev1) hipEventDestroy(s->ev1);
  if (s->ev_gather_read) hipEventDestroy(s->ev_gather_read);
  hipFree(s->d_params);
  hipFree(s->d_wfg_pix), hipFree(s->d_wfg_ray), hipFree(s->d_wfg_q0), hipFree(s->d_wfg_q1), hipFree(s->d_wfg_qs), hipFree(s->d_wfg_ctl);
  if (s->h_wfg) hipHostFree(s->h_wfg);
  delete s;
}

void rl_scene_destroy(rl_scene *s) {
  if (!s) return;
  for (size_t g = 1; g < s->replicas.size(); g++) destroy_one(s->replicas[g]);
  destroy_one(s);
  if (!g_ctx.empty()) hipSetDevice(g_ctx[0].device);
}

static int scene_common(rl_scene *s) {
  HIP_TRY(hipMalloc((void **)&s->d_scratch, 512));
  HIP_TRY(hipMemset(s->d_scratch, 0, 512));
  HIP_TRY(hipEventCreate(&s->ev0));
  HIP_TRY(hipEventCreate(&s->ev1));
  for (hipEvent_t &e : s->ev_status) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&s->ev_last, hipEventDisableTiming));
  HIP_TRY(hipHostMalloc((void **)&s->h_status, (size_t)rl_scene::N_STATUS * 64, hipHostMallocDefault));
  std::memset(s->h_status, 0, (size_t)rl_scene::N_STATUS * 64);
  return RL_OK;
}

// Linked form of a sphere-only program for the wave kernel: every op keeps its box and a, b, but {code, skip}
// become {w_hit, w_miss} = (state the lane enters there) << 29 | (op index): BOX hit -> op i+1, BOX_SPH hit ->
// the same op in LEAF, miss / leaf done -> op `skip`; a target that is OP_END means "SHADE", a bare OP_SPHERE
// "LEAF".  Boxes that are not BOX_FINITE are stored as NaN so the filtered test can never call them certain.
//
// guards != nullptr: every sphere additionally gets a GUARD op at index n_ops + sphere — the sphere's own bounding box
// (sphere.rs:77-88), tested by the same filter in a TRAV step but only ever used to REJECT (not one of the reference's
// tests, not counted) — and a guard leads to a LEAF visit of that ONE sphere:
//     leaf box hit -> guard(a) -> [LEAF a] -> guard(b) -> [LEAF b] -> the leaf's skip target.
// A ray that certainly misses a sphere's box certainly misses the sphere, so the expensive binary64 Sphere::hit is skipped
// for it (65 % of the leaf visits of BASELINE configs[1] end without a new closest hit).  Needs every sphere to be
// referenced exactly once (*guards_ok = false otherwise).
static uint32_t link_ops(const std::vector<DevOp> &ops, std::vector<DevOp> &out, const rl_rtiow_scene_desc *guards = nullptr, bool *guards_ok = nullptr,
                         const GuardFrame *frame = nullptr) {
  const uint32_t n0 = (uint32_t)ops.size();
  if (guards) {
    std::vector<uint8_t> seen(guards->n_spheres, 0);
    bool ok = true;
    for (uint32_t i = 0; i < n0 && ok; i++) {
      uint32_t kind = ops[i].code & 0xFFu;
      if (kind != OP_BOX_SPH && kind != OP_SPHERE) continue;
      for (uint32_t payload : {ops[i].a, ops[i].b}) {
        if (payload == NONE) continue;
        uint32_t sidx = payload & SPH_INDEX;
        if (seen[sidx]) ok = false;
        seen[sidx] = 1;
      }
    }
    if (guards_ok) *guards_ok = ok;
    if (!ok) guards = nullptr;
  }
  auto guard_of = [&](uint32_t payload) { return n0 + (payload & SPH_INDEX); };
  auto entry = [&](uint32_t t) -> uint32_t {
    uint32_t kind = ops[t].code & 0xFFu;
    if (guards && kind == OP_SPHERE) return (ST_TRAV << 29) | guard_of(ops[t].a);
    uint32_t st = kind == OP_END ? ST_SHADE : (kind == OP_SPHERE ? ST_LEAF : ST_TRAV);
    return (st << 29) | t;
  };
  out = ops;
  const double qnan = std::numeric_limits<double>::quiet_NaN();
  for (size_t i = 0; i < ops.size(); i++) {
    uint32_t kind = ops[i].code & 0xFFu;
    if (kind == OP_END) continue;
    DevOp &L = out[i];
    L.skip = entry(ops[i].skip);
    if (kind == OP_BOX) L.code = entry((uint32_t)i + 1u);
    else if (kind == OP_BOX_SPH) L.code = guards ? ((ST_TRAV << 29) | guard_of(ops[i].a)) : ((ST_LEAF << 29) | (uint32_t)i);
    else L.code = L.skip;  // OP_SPHERE: never stepped in TRAV
    if ((kind == OP_BOX || kind == OP_BOX_SPH) && !(ops[i].code & BOX_FINITE))
      for (double &b : L.box) b = qnan;
  }
  if (guards) {
    DevOp none{};
    for (double &b : none.box) b = qnan;
    none.code = none.skip = ST_SHADE << 29;
    none.a = none.b = NONE;
    out.resize((size_t)n0 + guards->n_spheres, none);  // spheres outside the tree keep a harmless record
    for (uint32_t i = 0; i < n0; i++) {
      uint32_t kind = ops[i].code & 0xFFu;
      if (kind != OP_BOX_SPH && kind != OP_SPHERE) continue;
      const uint32_t after = entry(ops[i].skip);
      for (int k = 0; k < 2; k++) {
        uint32_t payload = k == 0 ? ops[i].a : ops[i].b;
        if (payload == NONE) continue;
        const rl_sphere &sp = guards->spheres[payload & SPH_INDEX];
        DevOp G{};
        double r = std::fabs(sp.radius);
        for (int ax = 0; ax < 3; ax++) {
          double c0 = sp.center0[ax], c1 = sp.moving ? sp.center1[ax] : sp.center0[ax];
          double lo = std::fmin(c0, c1) - r, hi = std::fmax(c0, c1) + r;
          // this box only ever rejects: keep it outside the sphere by more than Sphere::hit's rounding can bridge (rl_fast_bvh.cpp guard_pad)
          double pad = std::fmax(1e-9 * (std::fabs(lo) + std::fabs(hi) + r), frame ? guard_pad(*frame, r) : 0.0);  // r = 0: inf -> NaN box
          G.box[2 * ax] = lo - pad, G.box[2 * ax + 1] = hi + pad;
        }
        for (int ax = 0; ax < 6; ax++)
          if (!(std::fabs(G.box[ax]) <= 1e30)) {  // NaN / inf / huge: never certain, i.e. always test the sphere
            for (double &b : G.box) b = qnan;
            break;
          }
        const uint32_t self = guard_of(payload);
        G.code = (ST_LEAF << 29) | self;  // box "hit" (or not certain): test the sphere
        const bool more = k == 0 && ops[i].b != NONE;
        G.skip = more ? ((ST_TRAV << 29) | guard_of(ops[i].b)) : after;  // then the leaf's other sphere, then on
        G.a = payload, G.b = NONE;
        out[self] = G;
      }
    }
  }
  return ops.empty() ? (uint32_t)(ST_SHADE << 29) : entry(0);
}

// One DevMaterial per sphere for the wave kernel's SHADE: the sphere's material with (a) a Solid texture's colour copied
// into albedo (flag MAT_TEX_SOLID) and (b) for a Dielectric albedo = {1/ior, r0(ri = 1/ior), r0(ri = ior)} with
// r0 = ((1 - ri) / (1 + ri))^2 — the very expressions material.rs:140-166 evaluates per scatter (IEEE, no contraction).
static void flatten_sphere_materials(const RtiowProgram &rt, std::vector<DevMaterial> &out) {
  out.resize(rt.spheres.size());
  for (size_t i = 0; i < rt.spheres.size(); i++) {
    DevMaterial m = rt.materials[rt.sphere_material[i]];
    if ((m.kind == RL_MAT_LAMBERTIAN || m.kind == RL_MAT_DIFFUSE_LIGHT) && rt.textures[m.texture].kind == RL_TEX_SOLID) {
      const DevTexture &t = rt.textures[m.texture];
      m.albedo[0] = t.color[0], m.albedo[1] = t.color[1], m.albedo[2] = t.color[2];
      m.kind |= MAT_TEX_SOLID;
    } else if (m.kind == RL_MAT_DIELECTRIC) {
      auto r0 = [](double ri) {
        double q = (1.0 - ri) / (1.0 + ri);
        return q * q;
      };
      double inv = 1.0 / m.ior;
      m.albedo[0] = inv, m.albedo[1] = r0(inv), m.albedo[2] = r0(m.ior);
    }
    out[i] = m;
  }
}

// Host half of rl_rtiow_scene_create: validate + lower the graph, link the ops, flatten the materials.  Device-independent.
static int build_host_rtiow(const rl_rtiow_scene_desc *desc, std::shared_ptr<const HostRtiow> &out) {
  auto H = std::make_shared<HostRtiow>();
  std::string err;
  if (compile_rtiow(*desc, H->rt, err) != RL_OK) return set_err(RL_E_INVALID, err);
  const RtiowProgram &rt = H->rt;
  if (!(rt.has_planars || rt.has_instances || rt.has_images || rt.has_noise || rt.has_media) && rt.ops.size() < (1u << 29)) {
    H->entry0 = link_ops(rt.ops, H->lops);
    flatten_sphere_materials(rt, H->sphere_flat);
    // compact guarded form (32-byte ops: binary32 box + the two successor words) for the 4-waves-per-SIMD layout
    std::vector<DevOp> gops;
    bool gok = false;
    GuardFrame frame = guard_frame(*desc);
    uint32_t gentry = link_ops(rt.ops, gops, desc, &gok, &frame);
    if (gok && gops.size() < (1u << 24)) {
      H->cops.resize(gops.size());
      for (size_t i = 0; i < gops.size(); i++) {
        for (int k = 0; k < 6; k++) H->cops[i].box[k] = (float)gops[i].box[k];
        H->cops[i].w_hit = gops[i].code, H->cops[i].w_miss = gops[i].skip;
      }
      // [one bit per sphere: Center::Moving][one bit per sphere: Metal / Dielectric material (the fast kernel's ST_SHADE2)]
      const size_t bw = (rt.spheres.size() + 31) / 32 + 1;
      H->movbits.assign(2 * bw, 0u);
      for (size_t i = 0; i < rt.spheres.size(); i++) {
        if (desc->spheres[i].moving) H->movbits[i >> 5] |= 1u << (i & 31);
        uint32_t kind = rt.materials[rt.sphere_material[i]].kind;
        if (kind == RL_MAT_METAL || kind == RL_MAT_DIELECTRIC) H->movbits[bw + (i >> 5)] |= 1u << (i & 31);
      }
      H->centry0 = gentry;
      std::memcpy(H->guard_center, frame.center, sizeof frame.center);
      H->guard_reach = frame.reach;
    }
    // the fast traversal structure of the timed (counter-free) kernel: ordered binary tree, reject-only boxes (rl_fast_bvh.cpp)
    if (!H->cops.empty() && !build_fast_bvh(*desc, rt, frame, H->fast_nodes, H->fast_root)) H->fast_nodes.clear(), H->fast_root = FAST_NONE;
    if (H->fast_root != FAST_NONE) {  // the spheres' own (padded) leaf boxes, by sphere index: what the cooperative kernel scans
      const uint32_t n_inner = (uint32_t)H->fast_nodes.size(), ns = (uint32_t)rt.spheres.size();
      H->fast_leaf_boxes.assign((size_t)ns * 8, 0.0f);
      for (uint32_t s = 0; s < ns; s++)  // a sphere the tree does not hold (n == 1: no node at all) is always a candidate
        for (int k = 0; k < 3; k++) H->fast_leaf_boxes[(size_t)s * 8 + 2 * k] = -3.0e38f, H->fast_leaf_boxes[(size_t)s * 8 + 2 * k + 1] = 3.0e38f;
      for (const FastNode &nd : H->fast_nodes)
        for (int k = 0; k < 2; k++) {
          const uint32_t e = k == 0 ? (nd.child & 0xFFFFu) : (nd.child >> 16);
          if (e >= n_inner && e - n_inner < ns) std::memcpy(&H->fast_leaf_boxes[(size_t)(e - n_inner) * 8], nd.box[k], 6 * sizeof(float));
        }
    }
  } else if (rt.ops.size() < (1u << 31)) {
    // general scenes (planars, instances, image / noise textures): world-space tree over the primitive occurrences
    if (!build_fast_general(*desc, rt, H->fg)) H->fg = FastGeneral{};
  }
  out = H;
  return RL_OK;
}

// Device half: one replica on device context `ctx` (the current device must already be that context's)
static rl_scene *upload_rtiow(const std::shared_ptr<const HostRtiow> &H, int ctx) {
  rl_scene *s = new rl_scene();
  s->kind = 1, s->ctx = ctx, s->device = g_ctx[(size_t)ctx].device, s->hrt = H;
  const RtiowProgram &rt = H->rt;
  int rc = RL_OK;
  if ((rc = upload(rt.ops, &s->d_ops)) || (rc = upload(rt.spheres, &s->d_spheres)) || (rc = upload(rt.sphere_material, &s->d_sphere_material)) ||
      (rc = upload(rt.planars, &s->d_planars)) || (rc = upload(rt.translates, &s->d_translates)) || (rc = upload(rt.transforms, &s->d_transforms)) ||
      (rc = upload(rt.materials, &s->d_materials)) || (rc = upload(rt.textures, &s->d_textures)) || (rc = upload(rt.images, &s->d_images)) ||
      (rc = upload(rt.image_pool, &s->d_image_pool)) || (rc = upload(rt.perlins, &s->d_perlins)) || (rc = upload(rt.media, &s->d_media)) ||
      (rc = scene_common(s)) ||
      (!H->lops.empty() && ((rc = upload(H->lops, &s->d_lops)) || (rc = upload(H->sphere_flat, &s->d_sphere_flat)))) ||
      (!H->cops.empty() && ((rc = upload(H->cops, &s->d_cops)) || (rc = upload(H->movbits, &s->d_movbits)))) ||
      (H->fast_root != FAST_NONE && ((rc = upload(H->fast_nodes, &s->d_fast_nodes)) || (rc = upload(H->fast_leaf_boxes, &s->d_fast_leaf_boxes)))) ||
      (H->fg.ok && ((rc = upload(H->fg.qnodes, &s->d_fg_nodes)) || (rc = upload(H->fg.onodes, &s->d_fg_onodes)) || (rc = upload(H->fg.stage_roots, &s->d_fg_seg_roots)) ||
                    (rc = upload(H->fg.media, &s->d_fg_media)) || (rc = upload(H->fg.items, &s->d_fg_items)) || (rc = upload(H->fg.item_spheres, &s->d_fg_spheres)) || (rc = upload(H->fg.item_material, &s->d_fg_material))))) {
    destroy_one(s);
    return nullptr;
  }
  return s;
}

rl_scene *rl_rtiow_scene_create(const rl_rtiow_scene_desc *desc) {
  if (!g_ready) {
    set_err(RL_E_NO_DEVICE, "rl_init has not succeeded (no GPU, or not called)");
    return nullptr;
  }
  if (!desc) {
    set_err(RL_E_INVALID, "null scene descriptor");
    return nullptr;
  }
  std::shared_ptr<const HostRtiow> H;
  if (build_host_rtiow(desc, H) != RL_OK) return nullptr;
  // one replica per device context (rl_init: one; rl_init_multi: one per GPU — the scene is replicated)
  std::vector<rl_scene *> reps;
  for (int g = 0; g < (int)g_ctx.size(); g++) {
    rl_scene *r = rl::use_context(g) == RL_OK ? upload_rtiow(H, g) : nullptr;
    if (!r) {
      for (rl_scene *q : reps) destroy_one(q);
      hipSetDevice(g_ctx[0].device);
      return nullptr;
    }
    reps.push_back(r);
  }
  hipSetDevice(g_ctx[0].device);
  if (reps.size() > 1) reps[0]->replicas = reps;
  return reps[0];
}

// ChaCha8Rng::seed_from_u64 (rand_core 0.6.4): PCG32 expands the u64 into the 256-bit key
static void chacha_key_from_seed(uint64_t state, uint32_t key[8]) {
  const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
  for (int k = 0; k < 8; k++) {
    state = state * MUL + INC;
    uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
    uint32_t rot = (uint32_t)(state >> 59);
    key[k] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
  }
}

static void read_stats(const unsigned long long *h, float ms, rl_stats *st) {
  st->rays = h[0], st->node_tests = h[1], st->sphere_tests = h[2], st->planar_tests = h[3];
  st->instance_enters = h[4], st->rng_words = h[5], st->flagged = h[6];
  st->kernel_ms = ms;
}

}  // extern "C"

namespace rl {
// counting renders: ev0 / ev1 bracket the kernels on `stream`; the stats words are read back synchronously
int collect_stats(const rl_scene *scene, hipStream_t stream, rl_stats *st) {
  unsigned long long h[8];
  HIP_TRY(hipMemcpyAsync(h, scene->d_scratch + 64, sizeof h, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, scene->ev0, scene->ev1));
  read_stats(h, ms, st);
  if (st->flagged) return set_err(RL_E_DEGENERATE, "a reference panic site was reached (see stats.flagged)");
  return RL_OK;
}
}  // namespace rl

namespace rl {
// one finished slot of the status ring -> the scene's folded figures (rays: of the most recently enqueued render)
static void fold_status(rl_scene *ms, int slot) {
  const unsigned long long *h = ms->h_status + (size_t)slot * 8;
  ms->folded_flagged += h[6], ms->folded_slow += h[7];
  if (ms->status_seq[slot] > ms->folded_seq) ms->folded_seq = ms->status_seq[slot], ms->folded_rays = h[0];
  ms->status_pending[slot] = false;
}
// asynchronous renders: leave the stats words in pinned host memory behind an event (rl_render_status reads them).  Caller holds scene->mu.
int post_status(const rl_scene *scene, hipStream_t stream) {
  rl_scene *ms = const_cast<rl_scene *>(scene);
  const int slot = ms->status_next;
  if (ms->status_pending[slot]) {  // the ring has come round: N_STATUS renders are in flight, wait for the oldest
    HIP_TRY(hipEventSynchronize(ms->ev_status[slot]));
    fold_status(ms, slot);
  }
  HIP_TRY(hipMemcpyAsync(ms->h_status + (size_t)slot * 8, scene->d_scratch + 64, 64, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipEventRecord(ms->ev_status[slot], stream));
  ms->status_pending[slot] = true, ms->status_seq[slot] = ++ms->next_seq;
  ms->status_next = (slot + 1) % rl_scene::N_STATUS;
  return RL_OK;
}
int order_after_previous(const rl_scene *scene, hipStream_t stream) {
  if (scene->has_last && scene->last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, scene->ev_last, 0));
  return RL_OK;
}
int mark_render_end(const rl_scene *scene, hipStream_t stream) {
  rl_scene *ms = const_cast<rl_scene *>(scene);
  HIP_TRY(hipEventRecord(ms->ev_last, stream));
  ms->last_stream = stream, ms->has_last = true;
  return RL_OK;
}
}  // namespace rl
using rl::post_status;

#ifdef RL_EXPERIMENTAL
// Wavefront (v3) driver: one PASS = begin_pass, TRAV, SHADE, GEN (rl_rtiow_wavefront.h); passes are enqueued in
// chunks and the finished-pixel counter is polled once per chunk.
static int render_wavefront(const rl_scene *scene, RtiowParams &P, uint32_t nrows, hipStream_t stream, bool want_stats) {
  rl_scene *sc = const_cast<rl_scene *>(scene);  // work buffers only; the scene program is immutable
  if (!sc->exp) sc->exp = new ExpBuffers();
  ExpBuffers *ms = (ExpBuffers *)sc->exp;
  const uint32_t Wd = P.cam.image_width;
  size_t npix = (size_t)nrows * Wd;
  if (npix >= 0xFFFF0000ull) return set_err(RL_E_INVALID, "image too large");
  if (ms->wf_npix < npix) {
    hipFree(ms->wf_pix), hipFree(ms->wf_ray), hipFree(ms->wf_hit), hipFree(ms->wf_qtrav), hipFree(ms->wf_qshade), hipFree(ms->wf_qgen);
    ms->wf_pix = nullptr, ms->wf_ray = nullptr, ms->wf_hit = nullptr, ms->wf_qtrav = ms->wf_qshade = ms->wf_qgen = nullptr, ms->wf_npix = 0;
    HIP_TRY(hipMalloc((void **)&ms->wf_pix, npix * sizeof(PixState)));
    HIP_TRY(hipMalloc((void **)&ms->wf_ray, npix * sizeof(RayRec)));
    HIP_TRY(hipMalloc((void **)&ms->wf_hit, npix * sizeof(HitRec)));
    HIP_TRY(hipMalloc((void **)&ms->wf_qtrav, 2 * npix * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&ms->wf_qgen, npix * sizeof(uint32_t)));
    ms->wf_npix = npix;
  }
  if (!ms->wf_ctl) HIP_TRY(hipMalloc((void **)&ms->wf_ctl, WC_WORDS * sizeof(uint32_t)));
  WfParams Wp{};
  Wp.R = P;
  Wp.pix = ms->wf_pix, Wp.ray = ms->wf_ray, Wp.hit = ms->wf_hit;
  Wp.q_trav = ms->wf_qtrav, Wp.q_gen = ms->wf_qgen, Wp.ctl = ms->wf_ctl;
  Wp.npix = (uint32_t)npix;

  constexpr int NTS = 256;  // GEN / SHADE workgroups
  size_t lds_s = (size_t)8 * NTS * sizeof(unsigned long long);
  uint32_t flat_blocks = (uint32_t)((npix + NTS - 1) / NTS);
  uint32_t stage_blocks = flat_blocks < (uint32_t)g_cus * 8 ? flat_blocks : (uint32_t)g_cus * 8;
  constexpr int NTT = 768;  // TRAV workgroups (12 waves; two per CU = 6 waves/SIMD at <= 80 VGPRs), scene staged in LDS when it fits
  size_t scene_bytes = (size_t)P.n_ops * sizeof(DevOp) + (size_t)P.n_spheres * sizeof(DevSphere);
  bool in_lds = scene_bytes <= g_lds_max;
  size_t lds_t = in_lds ? scene_bytes : 0;
  uint32_t per_cu = in_lds ? (uint32_t)(g_lds_max / (scene_bytes ? scene_bytes : 1)) : 2;
  if (per_cu > 2) per_cu = 2;
  if (per_cu < 1) per_cu = 1;
  uint32_t trav_blocks = (uint32_t)g_cus * per_cu;
  {
    uint32_t need = (uint32_t)((npix + NTT - 1) / NTT);
    if (trav_blocks > need) trav_blocks = need;
  }
  if (in_lds) {
    HIP_TRY(hipFuncSetAttribute((const void *)wf_trav<NTT, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
    HIP_TRY(hipFuncSetAttribute((const void *)wf_trav<NTT, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
  }
  auto trav = [&]() -> int {
    if (in_lds) {
      if (want_stats) hipLaunchKernelGGL((wf_trav<NTT, true, true>), dim3(trav_blocks), dim3(NTT), lds_t, stream, Wp);
      else hipLaunchKernelGGL((wf_trav<NTT, true, false>), dim3(trav_blocks), dim3(NTT), lds_t, stream, Wp);
    } else {
      if (want_stats) hipLaunchKernelGGL((wf_trav<NTT, false, true>), dim3(trav_blocks), dim3(NTT), 0, stream, Wp);
      else hipLaunchKernelGGL((wf_trav<NTT, false, false>), dim3(trav_blocks), dim3(NTT), 0, stream, Wp);
    }
    return RL_OK;
  };
  hipLaunchKernelGGL(wf_init, dim3(flat_blocks), dim3(NTS), 0, stream, Wp);
  hipLaunchKernelGGL((wf_gen<NTS>), dim3(stage_blocks), dim3(NTS), lds_s, stream, Wp);
  HIP_TRY(hipGetLastError());
  const int CHUNK = 64;
  uint32_t finished = 0;
  for (long pass = 0; pass < (1l << 40); pass += CHUNK) {
    for (int k = 0; k < CHUNK; k++) {
      hipLaunchKernelGGL(wf_begin_pass, dim3(1), dim3(1), 0, stream, Wp);
      int rc = trav();
      if (rc != RL_OK) return rc;
      hipLaunchKernelGGL((wf_shade<NTS>), dim3(stage_blocks), dim3(NTS), lds_s, stream, Wp);
      hipLaunchKernelGGL((wf_gen<NTS>), dim3(stage_blocks), dim3(NTS), lds_s, stream, Wp);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&finished, ms->wf_ctl + WC_FINISHED, sizeof finished, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (std::getenv("RL_WF_DEBUG")) std::fprintf(stderr, "[wf] passes %ld finished %u / %zu\n", pass + CHUNK, finished, npix);
    if (finished >= npix) break;
  }
  hipLaunchKernelGGL(wf_finish, dim3(flat_blocks), dim3(NTS), 0, stream, Wp);
  HIP_TRY(hipGetLastError());
  return RL_OK;
}

#endif

#ifdef RL_EXPERIMENTAL
// Wavefront form of the general fast traversal (rl_rtiow_wfg.h): init, then PASSES of wfg_logic + wfg_trav until every pixel has finished.
// The pass chain is enqueued in chunks and the finished-pixel counter is polled once per chunk (one chunk ahead of the one being waited
// for), so this render returns when the frame is complete: the call is synchronous on `stream`.
static int render_wfg(const rl_scene *scene, RtiowParams &P, uint32_t nrows, hipStream_t stream, bool trans) {
  rl_scene *ms = const_cast<rl_scene *>(scene);  // work buffers only; the scene program is immutable
  const size_t npix = (size_t)nrows * P.cam.image_width, nslots = P.n_slots;
  if (ms->wfg_pix_cap < npix || ms->wfg_slot_cap < nslots) {
    hipFree(ms->d_wfg_pix), hipFree(ms->d_wfg_ray), hipFree(ms->d_wfg_q0), hipFree(ms->d_wfg_q1), hipFree(ms->d_wfg_qs);
    ms->d_wfg_pix = nullptr, ms->d_wfg_ray = nullptr, ms->d_wfg_q0 = ms->d_wfg_q1 = ms->d_wfg_qs = nullptr, ms->wfg_pix_cap = ms->wfg_slot_cap = 0;
    HIP_TRY(hipMalloc((void **)&ms->d_wfg_pix, npix * sizeof(WfgPix)));
    HIP_TRY(hipMalloc((void **)&ms->d_wfg_ray, npix * sizeof(WfgRay)));
    HIP_TRY(hipMalloc((void **)&ms->d_wfg_q0, nslots * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&ms->d_wfg_q1, nslots * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&ms->d_wfg_qs, nslots * sizeof(uint32_t)));
    ms->wfg_pix_cap = npix, ms->wfg_slot_cap = nslots;
  }
  if (!ms->d_wfg_ctl) HIP_TRY(hipMalloc((void **)&ms->d_wfg_ctl, WFG_CTL_WORDS * sizeof(uint32_t)));
  if (!ms->h_wfg) HIP_TRY(hipHostMalloc((void **)&ms->h_wfg, 64, hipHostMallocDefault));
  WfgParams Q{};
  Q.pix = (WfgPix *)ms->d_wfg_pix, Q.ray = (WfgRay *)ms->d_wfg_ray, Q.queue[0] = ms->d_wfg_q0, Q.queue[1] = ms->d_wfg_q1, Q.slow_queue = ms->d_wfg_qs, Q.ctl = ms->d_wfg_ctl;
  P.sample_begin = 0, P.sample_end = P.cam.samples_per_pixel, P.resume = 0, P.pos_state = nullptr, P.tile_order = nullptr, P.tile_cost = nullptr;
  if (!g_sw.tune_set) P.tune[0] = 4, P.tune[3] = FASTG_STEP_BUDGET;
  hipLaunchKernelGGL(wfg_init, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, stream, P, Q);
  HIP_TRY(hipGetLastError());
  constexpr int LNT = 256;                         // wfg_logic: blocks of one wave per SIMD, 3 (2 with transcendental textures) per CU, rings in LDS
  const size_t llds = (size_t)16 * LNT * sizeof(unsigned long long);
  constexpr int TNT = 256, TSD = 20;               // wfg_trav: 20-entry stacks in LDS, register budget of `twpe` waves per SIMD
  const size_t tlds = (size_t)TSD * TNT * sizeof(uint32_t);
  const int twpe = (g_sw.tune_set && (g_sw.tune[2] == 4 || g_sw.tune[2] == 5 || g_sw.tune[2] == 8)) ? (int)g_sw.tune[2] : 6;  // A/B: RL_TUNE third field
  const uint32_t lblocks = (uint32_t)g_cus * (trans ? 2u : 3u), tblocks = (uint32_t)g_cus * (uint32_t)twpe;
  const uint32_t sblocks = (uint32_t)g_cus;  // the slow queue holds a fraction of a percent of a pass's rays
  const void *tk = twpe == 4 ? (const void *)wfg_trav<TNT, TSD, 4> : twpe == 5 ? (const void *)wfg_trav<TNT, TSD, 5> : twpe == 8 ? (const void *)wfg_trav<TNT, TSD, 8> : (const void *)wfg_trav<TNT, TSD, 6>;
  if (ensure_lds_attr(trans ? (const void *)wfg_logic<LNT, true, false> : (const void *)wfg_logic<LNT, false, false>, llds) != 0 ||
      ensure_lds_attr(trans ? (const void *)wfg_logic<LNT, true, true> : (const void *)wfg_logic<LNT, false, true>, llds) != 0 || ensure_lds_attr(tk, tlds) != 0)
    return set_err(RL_E_DEVICE, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
  const int CHUNK = 16;
  volatile uint32_t *h_done = (volatile uint32_t *)ms->h_wfg;
  uint32_t pass = 0;
  for (long chunk = 0; chunk < (1l << 36); chunk++) {
    for (int k = 0; k < CHUNK; k++, pass++) {
      Q.in = pass & 1u;
      if (trans) {
        hipLaunchKernelGGL((wfg_logic<LNT, true, false>), dim3(lblocks), dim3(LNT), llds, stream, P, Q);
        hipLaunchKernelGGL((wfg_logic<LNT, true, true>), dim3(sblocks), dim3(LNT), llds, stream, P, Q);
      } else {
        hipLaunchKernelGGL((wfg_logic<LNT, false, false>), dim3(lblocks), dim3(LNT), llds, stream, P, Q);
        hipLaunchKernelGGL((wfg_logic<LNT, false, true>), dim3(sblocks), dim3(LNT), llds, stream, P, Q);
      }
      if (twpe == 4) hipLaunchKernelGGL((wfg_trav<TNT, TSD, 4>), dim3(tblocks), dim3(TNT), tlds, stream, P, Q);
      else if (twpe == 5) hipLaunchKernelGGL((wfg_trav<TNT, TSD, 5>), dim3(tblocks), dim3(TNT), tlds, stream, P, Q);
      else if (twpe == 8) hipLaunchKernelGGL((wfg_trav<TNT, TSD, 8>), dim3(tblocks), dim3(TNT), tlds, stream, P, Q);
      else hipLaunchKernelGGL((wfg_trav<TNT, TSD, 6>), dim3(tblocks), dim3(TNT), tlds, stream, P, Q);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync((void *)h_done, ms->d_wfg_ctl + WFG_DONE, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (*h_done >= npix) break;
  }
  return RL_OK;
}

#endif

// The parameter block of one render (what every RTIOW kernel receives): scene pointers, derived camera, ChaCha key, shard geometry.
static int fill_rtiow_params(const rl_scene *scene, const rl_rtiow_camera *cam, uint64_t first_sample, uint32_t row_first, uint32_t row_step, uint32_t nrows, void *d_out,
                             bool want_stats, RtiowParams &P, uint64_t &slots) {
  const RtiowProgram &rt = scene->rt();
  const HostRtiow &H = *scene->hrt;
  const uint32_t W = cam->image_width;
  P = RtiowParams{};
  P.ops = scene->d_ops, P.spheres = scene->d_spheres, P.sphere_material = scene->d_sphere_material;
  P.planars = scene->d_planars, P.translates = scene->d_translates, P.transforms = scene->d_transforms;
  P.materials = scene->d_materials, P.textures = scene->d_textures, P.images = scene->d_images, P.image_pool = scene->d_image_pool, P.perlins = scene->d_perlins, P.media = scene->d_media;
  P.n_ops = (uint32_t)rt.ops.size(), P.n_spheres = (uint32_t)rt.spheres.size();
  P.lops = scene->d_lops, P.entry0 = H.entry0, P.sphere_flat = scene->d_sphere_flat;
  P.cops = scene->d_cops, P.n_cops = (uint32_t)H.cops.size(), P.centry0 = H.centry0, P.movbits = scene->d_movbits;
  P.fast_nodes = scene->d_fast_nodes, P.n_fast_inner = (uint32_t)H.fast_nodes.size(), P.fast_root = H.fast_root;
  P.fg_nodes = scene->d_fg_nodes, P.fg_items = scene->d_fg_items, P.fg_spheres = scene->d_fg_spheres, P.fg_material = scene->d_fg_material, P.fg_root = H.fg.qroot, P.fg_rsafe2 = H.fg.r_safe * H.fg.r_safe * 0.9999f;  // binary32 evaluation on the device: keep a margin
#ifdef RL_EXPERIMENTAL
  P.fg_onodes = scene->d_fg_onodes, P.fg_oroot = H.fg.oroot;
#endif
  P.fg_seg_roots = scene->d_fg_seg_roots, P.fg_media = scene->d_fg_media, P.fg_n_seg = (uint32_t)H.fg.stage_roots.size();
  P.fg_top = 0;  // (set by the launch that stages it: fastg_lds)
  P.fg_center[0] = H.fg.center[0], P.fg_center[1] = H.fg.center[1], P.fg_center[2] = H.fg.center[2];
  P.fg_radius = H.fg.radius, P.fg_pad_k = H.fg.pad_k;
  P.cam = *cam;
  chacha_key_from_seed(cam->seed, P.key);
  P.first_sample = first_sample;
  P.row_first = row_first, P.row_step = row_step, P.nrows = nrows;
  P.tiles_x = (W + 7) / 8;
  slots = (uint64_t)P.tiles_x * ((nrows + 7) / 8) * 64ull;
  if (slots >= 0xFFFF0000ull) return set_err(RL_E_INVALID, "image too large");
  P.n_slots = (uint32_t)slots;
  P.work_counter = (uint32_t *)scene->d_scratch;
  P.stats = (unsigned long long *)(scene->d_scratch + 64);
  P.out = (double *)d_out;
  P.k8u = 8.8817841970012523e-16;
  P.tune[0] = g_sw.tune[0], P.tune[1] = g_sw.tune[1], P.tune[2] = g_sw.tune[2], P.tune[3] = g_sw.tune[3];
  P.pix_rays = want_stats ? scene->d_pix_rays : nullptr;
  return RL_OK;
}

// Which kernel renders this frame (RL_RTIOW_KERNEL / rl_debug_set_rtiow_variant force one where the scene qualifies), and the LDS bytes of its scene part.
struct RtiowChoice {
  int variant = 0;
  bool general = false;
  size_t compact_bytes = 0, fast_bytes = 0, scene_bytes = 0;
  bool fits_fast = false;
};
static int choose_rtiow_variant(const rl_scene *scene, const rl_rtiow_camera *cam, const RtiowParams &P, uint32_t nrows, bool want_stats, RtiowChoice &out) {
  const RtiowProgram &rt = scene->rt();
  const HostRtiow &H = *scene->hrt;
  const uint32_t W = cam->image_width, n_cops = P.n_cops;
  const size_t scene_bytes = (size_t)P.n_ops * sizeof(DevOp) + (size_t)P.n_spheres * sizeof(DevSphere);
  // ---- kernel variant.  The PRODUCT library carries, for sphere-only worlds, the four layouts the automatic choice below can reach
  // (1029 fast traversal / 1027 guarded compact ops / 1025 linked ops in LDS / 1024 scene through L2), the cooperative kernel (1033) and
  // the nested-loop all-primitives kernel (2: the A/B reference the tests compare against); for general worlds 1031 (fast traversal) and
  // 4 (reference order).  Every other instantiation (wave256 / 512 / 768, whole-scene-in-LDS layouts, v1, other register budgets,
  // pool / wave2 / wavefront) is A/B material and lives in librl_render_exp.so (make exp).
  int variant = g_sw.rtiow_variant;
  const bool general = rt.has_planars || rt.has_instances || rt.has_images || rt.has_noise || rt.has_media;
  // a ConstantMedium (RL_H_MEDIUM) draws from the pixel's RNG where the reference's fold reaches it: the reference-order kernels evaluate
  // it as a scope of the threaded program (wave-scheduled) or by recursion (RL_RTIOW_KERNEL=general); the fast traversal walks one tree per
  // program segment between two media (rl_rtiow_fastgen.h MEDIA) — counter-free renders only, as for every fast traversal
  if (rt.has_media && variant != 2 && variant != 0 && variant != 1031) variant = 4;
#ifndef RL_EXPERIMENTAL
  if (variant == 1 || variant == 3 || variant == 5 || variant == 6 || variant == 7 || variant == 256 || variant == 512 || variant == 768 || variant == 1035)
    return set_err(RL_E_UNSUPPORTED, "this kernel variant is A/B material and lives in librl_render_exp.so only (make -C rendering-learning_amd/csrc exp)");
#endif
  // 1031 = the FAST traversal for general scenes (rl_rtiow_fastgen.h): counter-free renders only, like 1029
  const bool fits_fastg = general && H.fg.ok && (!want_stats) && g_sw.fast_traversal;
  if (variant == 2) variant = 2;               // the nested-loop all-primitives kernel (A/B reference)
  else if ((variant == 0 || variant == 1031 || variant == 1035) && fits_fastg) {
    // 1035 = the same traversal in WAVEFRONT form (experimental/rl_rtiow_wfg.h): measured, slower, experimental library only
    bool wf = false;
#ifdef RL_EXPERIMENTAL
    wf = (variant == 1035 || (variant == 0 && g_sw.wavefront == 1)) && (uint64_t)nrows * W < 0xFFFF0000ull && !rt.has_media;
#endif
    variant = wf ? 1035 : 1031;
  }
  else if (general || variant == 4 || variant == 1031) variant = 4;  // wave-scheduled all-primitives kernel (scene read from HBM/L2)
  const size_t compact_bytes = ((size_t)n_cops * sizeof(CompactOp) + (((size_t)P.n_spheres + 31) / 32 + 1) * sizeof(uint32_t) + 15) & ~(size_t)15;
  bool fits_compact = n_cops != 0 && (size_t)16 * 1024 * sizeof(unsigned long long) + compact_bytes <= g_lds_max;
  if (fits_compact) {
    // the guard boxes' padding covers the rounding of Sphere::hit only for ray origins within guard_reach of the scene
    // (link_ops); every later origin is a hit point inside the scene, so only the camera has to be checked
    double far = 0.0;
    for (int k = 0; k < 3; k++) {
      double c = cam->lookfrom[k] - H.guard_center[k];
      far += c * c;
    }
    double disk = 0.0;
    for (int k = 0; k < 3; k++) disk += std::fabs(cam->defocus_disk_u[k]) + std::fabs(cam->defocus_disk_v[k]);
    if (!(std::sqrt(far) + disk <= H.guard_reach)) fits_compact = false;  // also for NaN
  }
  // 1029 = the FAST traversal (ordered binary tree, reject-only boxes, exact re-trace of ambiguous rays): counter-free renders only —
  // its box / sphere test counts are not the reference's, so a render that asks for rl_stats runs the counting kernel (1027)
  const size_t fast_bytes = ((size_t)P.n_fast_inner * sizeof(FastNode) + 2 * (((size_t)P.n_spheres + 31) / 32 + 1) * sizeof(uint32_t) + 15) & ~(size_t)15;
  const bool fits_fast = fits_compact && H.fast_root != FAST_NONE && (!want_stats || g_fast_debug_stats) && (size_t)16 * 1024 * sizeof(unsigned long long) + fast_bytes <= g_lds_max &&
                         g_sw.fast_traversal;
  const bool fits_ops = (size_t)16 * 1024 * sizeof(unsigned long long) + (size_t)P.n_ops * sizeof(DevOp) <= g_lds_max;
  if (variant == 1029 && (general || !fits_fast)) variant = 0;
  if (variant == 1033 && (general || !fits_fast || want_stats)) variant = 0;  // cooperative kernel: the fast structure's scenes, counter-free renders
  if (variant == 1027 && (general || !fits_compact)) variant = 0;
  if (variant == 1025 && (general || !fits_ops)) variant = 0;
  if (variant == 7 && (general || (size_t)8 * 1024 * sizeof(unsigned long long) + scene_bytes > g_lds_max)) variant = 0;  // two-context kernel needs the scene in LDS
  if ((variant == 5 || variant == 6) && (general || (size_t)(variant == 5 ? 512 : 256) * 192 + scene_bytes > g_lds_max)) variant = 0;  // pool kernel needs the scene in LDS
  if (variant == 0 && !general) {
    // automatic: 4 waves per SIMD (1024 lanes per CU, 128 KB of ChaCha rings) with, in LDS next to the rings, the fast tree / the
    // guarded compact ops / the linked ops — whichever fits first — and otherwise the whole scene read through L2
    variant = fits_fast ? 1029 : fits_compact ? 1027 : fits_ops ? 1025 : 1024;
    // Small frames (at most 10 pixels per wave the GPU can hold) are pure latency: every pixel's sample chain runs alone, and the
    // cooperative one-wave-per-pixel kernel advances a chain in 3.9 us per ray instead of ~22 (rl_rtiow_coop.h).  Measured at 1024 spp:
    // 2.2 k pixels 184 -> 30 ms, 9 k 248 -> 68, 20 k 267 -> 109, 37 k 264 -> 167, 90 k 291 -> 380 (tools/coop_check.py); from ~40 k
    // pixels on the wave-scheduled kernel with work stealing is as fast or faster (37 k: 173 ms, 90 k: 170 ms; tools/steal_ab.py).
    if (variant == 1029 && !want_stats && g_sw.coop_small && (uint64_t)nrows * W <= (uint64_t)g_cus * 16u * 10u) variant = 1033;
  }
  out.variant = variant, out.general = general, out.compact_bytes = compact_bytes, out.fast_bytes = fast_bytes, out.scene_bytes = scene_bytes, out.fits_fast = fits_fast;
  return RL_OK;
}

namespace rl {
int rtiow_render_launch(const rl_scene *scene, const rl_rtiow_camera *cam, uint64_t first_sample, uint32_t row_first, uint32_t row_step, void *d_out,
                        hipStream_t stream, bool want_stats) {
  const RtiowProgram &rt = scene->rt();
  const HostRtiow &H = *scene->hrt;
  (void)H;  // (experimental builds read more of it)
  uint32_t H_ = cam->image_height, W = cam->image_width;
  uint32_t nrows = row_first < H_ ? (H_ - row_first + row_step - 1) / row_step : 0;
  RtiowParams P;
  uint64_t slots = 0;
  {
    int rcp = fill_rtiow_params(scene, cam, first_sample, row_first, row_step, nrows, d_out, want_stats, P, slots);
    if (rcp != RL_OK) return rcp;
  }

  {
    int rco = order_after_previous(scene, stream);
    if (rco != RL_OK) return rco;
  }
  HIP_TRY(hipMemsetAsync(scene->d_scratch, 0, 512, stream));  // [0] work counter, [64..] stats, [256] cooperative kernel's / stealing counter
  if (scene->progress_on) {  // the work counter in host-visible memory (rl_rtiow_render_progress)
    const_cast<rl_scene *>(scene)->progress_total = slots;
    P.work_counter = scene->d_progress;
    HIP_TRY(hipMemsetAsync(scene->d_progress, 0, 8, stream));
  }
  size_t scene_bytes = (size_t)P.n_ops * sizeof(DevOp) + (size_t)P.n_spheres * sizeof(DevSphere);
  auto launch = [&](auto kern, int nt, size_t rng_bytes, bool lds_scene) -> int {
    size_t lds = rng_bytes + (lds_scene ? scene_bytes : 0);
    uint32_t blocks = (uint32_t)((slots + nt - 1) / nt);
    uint32_t per_cu = (uint32_t)(g_lds_max / (lds ? lds : 1));  // persistent lanes: as many workgroups as stay resident
    if (per_cu < 1) per_cu = 1;
    if (per_cu * nt > 2048) per_cu = 2048 / nt;
    if (blocks > (uint32_t)g_cus * per_cu) blocks = (uint32_t)g_cus * per_cu;
    if (g_sw.blocks_cap >= 1 && g_sw.blocks_cap < blocks) blocks = g_sw.blocks_cap;  // A/B only
    if (ensure_lds_attr((const void *)kern, lds) != 0) return set_err(RL_E_DEVICE, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(nt), lds, stream, P);
    HIP_TRY(hipGetLastError());
    return RL_OK;
  };
  auto launch_ptr = [&](auto kern, int nt, size_t rng_bytes) -> int {  // kernels that take the parameter block by pointer (device copy)
    rl_scene *ms = const_cast<rl_scene *>(scene);
    if (!ms->d_params) HIP_TRY(hipMalloc((void **)&ms->d_params, 2 * sizeof(RtiowParams)));
    // two slots: the cost-sorted render launches twice with different parameters, both enqueued before the first one runs
    RtiowParams *slot = (RtiowParams *)ms->d_params + (ms->params_slot++ & 1);
    HIP_TRY(hipMemcpyAsync(slot, &P, sizeof(RtiowParams), hipMemcpyHostToDevice, stream));
    size_t lds = rng_bytes;
    uint32_t blocks = (uint32_t)((slots + nt - 1) / nt);
    uint32_t per_cu = (uint32_t)(g_lds_max / (lds ? lds : 1));
    if (per_cu < 1) per_cu = 1;
    if (per_cu * nt > 2048) per_cu = 2048 / nt;
    if (blocks > (uint32_t)g_cus * per_cu) blocks = (uint32_t)g_cus * per_cu;
    if (g_sw.blocks_cap >= 1 && g_sw.blocks_cap < blocks) blocks = g_sw.blocks_cap;
    if (ensure_lds_attr((const void *)kern, lds) != 0) return set_err(RL_E_DEVICE, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(nt), lds, stream, (const RtiowParams *)slot);
    HIP_TRY(hipGetLastError());
    return RL_OK;
  };
  RtiowChoice choice;
  {
    int rcv = choose_rtiow_variant(scene, cam, P, nrows, want_stats, choice);
    if (rcv != RL_OK) return rcv;
  }
  const int variant = choice.variant;
  const size_t compact_bytes = choice.compact_bytes, fast_bytes = choice.fast_bytes;
  bool steal = false;  // set for the resume launch of a small shard (variant 1029)
  auto launch_coop = [&](const uint32_t *d_pixels, uint32_t n_pixels) -> int {
    constexpr int NW = 4;
    CoopParams C{};
    C.pixels = d_pixels, C.n_pixels = n_pixels, C.leaf_boxes = scene->d_fast_leaf_boxes;
    C.counter = (uint32_t *)(scene->d_scratch + 256);
    C.max_cand = 128;
    HIP_TRY(hipMemsetAsync(C.counter, 0, 4, stream));
    size_t lds = (size_t)8 * NW * 64 * sizeof(unsigned long long) + (size_t)NW * C.max_cand * sizeof(uint32_t);
    uint32_t blocks = (n_pixels + NW - 1) / NW;
    uint32_t cap = (uint32_t)g_cus * (16 / NW);  // 4 waves per SIMD at 128 VGPRs
    if (blocks > cap) blocks = cap;
    if (blocks == 0) return RL_OK;
    // the lowest latency (boxes in registers: 3.3 us per ray, two waves per SIMD) while every pixel gets a wave at once; the register
    // budget of four waves per SIMD (boxes from L2: 4.0 us per ray, +25 % throughput) for larger frames.  RL_COOP_MODE forces one (A/B).
    int mode = g_sw.coop_mode >= 0 ? g_sw.coop_mode : (n_pixels <= (uint32_t)g_cus * 8u ? 2 : 0);
    if (mode == 2) {
      if (ensure_lds_attr((const void *)rtiow_coop_kernel<NW, true, NW * 64>, lds) != 0) return set_err(RL_E_DEVICE, "hipFuncSetAttribute failed");
      hipLaunchKernelGGL((rtiow_coop_kernel<NW, true, NW * 64>), dim3(blocks), dim3(NW * 64), lds, stream, P, C);
#ifdef RL_EXPERIMENTAL
    } else if (mode == 1) {  // L2 boxes at the register budget of two waves per SIMD
      if (ensure_lds_attr((const void *)rtiow_coop_kernel<NW, false, NW * 64>, lds) != 0) return set_err(RL_E_DEVICE, "hipFuncSetAttribute failed");
      hipLaunchKernelGGL((rtiow_coop_kernel<NW, false, NW * 64>), dim3(blocks), dim3(NW * 64), lds, stream, P, C);
#endif
    } else {
      if (ensure_lds_attr((const void *)rtiow_coop_kernel<NW, false, 1024>, lds) != 0) return set_err(RL_E_DEVICE, "hipFuncSetAttribute failed");
      hipLaunchKernelGGL((rtiow_coop_kernel<NW, false, 1024>), dim3(blocks), dim3(NW * 64), lds, stream, P, C);
    }
    HIP_TRY(hipGetLastError());
    return RL_OK;
  };
  // LDS of a fast general launch: rings + stacks, and the tree's top in what is left (P.fg_top nodes, breadth first: FastGeneral::top_nodes)
  auto fastg_lds = [&](int NT, int SD) -> size_t {
    const size_t base = (size_t)NT * (16 * sizeof(unsigned long long) + (size_t)SD * sizeof(uint32_t));
    const size_t room = g_lds_max > base ? (g_lds_max - base) / sizeof(FastNodeQ) : 0;
    P.fg_top = g_sw.fastg_top ? (uint32_t)std::min<size_t>(H.fg.top_nodes, std::min<size_t>(room, g_sw.fastg_top_max)) : 0u;
    return base + (size_t)P.fg_top * sizeof(FastNodeQ);
  };
  auto launch_variant = [&]() -> int {
    int rc;
    if (variant == 2) {
      constexpr int NT = 256;
      size_t rb = (size_t)8 * NT * sizeof(unsigned long long);
      // register budget of two waves per SIMD (256 registers instead of 256 + 151): cornell_smoke 309 -> 493 Mrays/s, final_scene 170 -> 219
      // (three waves: 390 / 182).  RL_GENERAL_REGS=256|768 selects the others (experimental library).
#ifdef RL_EXPERIMENTAL
      if (g_sw.general_regs == 768) rc = want_stats ? launch(rtiow_general_kernel<NT, true, 768>, NT, rb, false) : launch(rtiow_general_kernel<NT, false, 768>, NT, rb, false);
      else if (g_sw.general_regs == 256) rc = want_stats ? launch(rtiow_general_kernel<NT, true>, NT, rb, false) : launch(rtiow_general_kernel<NT, false>, NT, rb, false);
      else
#endif
        rc = want_stats ? launch(rtiow_general_kernel<NT, true, 512>, NT, rb, false) : launch(rtiow_general_kernel<NT, false, 512>, NT, rb, false);
#ifdef RL_EXPERIMENTAL
    } else if (variant == 1035) {
      rc = render_wfg(scene, P, nrows, stream, rt.has_noise || rt.has_sphere_uv);
#endif
    } else if (variant == 1031) {  // rings + the traversal stacks in LDS
      // four steps per scheduling round (a step is an Infinity Cache / L2 round trip here, not an LDS one: lanes that fall out of TRAV
      // should not wait 24 of them): cfg 5 +6.6 %, cfg 4 +0.7 % against the sphere kernel's 24
      if (!g_sw.tune_set) P.tune[0] = 4, P.tune[2] = 4, P.tune[3] = FASTG_STEP_BUDGET;
      bool trans = rt.has_noise || rt.has_sphere_uv;
      // The flavour with both the media code and the Perlin / acos / atan2 code spills 126 VGPRs at 256 registers.  For small frames (at most
      // three pixels per lane of the 512-lane form: the frame's time is per-ray latency, and scratch round trips are part of it) it
      // runs as ONE wave per SIMD with the 512-register budget instead (what would spill lives in AGPRs): final_scene.rs at 400x400 442 ->
      // 522 Mrays/s, at 560x560 664 -> 782, at 640x640 786 -> 811; from 720x720 on the two-wave form wins (975 against 818; 1100 against 810
      // at 1000x1000: the one-wave form's throughput ends at ~810 Mrays/s).  Measured and not taken: the same for the other flavours (quads.rs -17 %, flat_world.rs -24 %, cornell_smoke.rs at
      // 600x600 -30 %: they spill little, and lose the second wave's latency hiding).  RL_FASTG_NT256=1 / 0 forces it on / off (A/B).
      const bool media = H.fg.stage_roots.size() > 1;
      const bool one_wave = trans && media && (g_sw.fastg_nt256 >= 0 ? g_sw.fastg_nt256 != 0 : (uint64_t)nrows * W <= (uint64_t)g_cus * 1536u);
      if (one_wave) {
        constexpr int NT = 256, SD = 40;
        size_t rb = fastg_lds(NT, SD);
        if (rb < 90000) rb = 90000;  // one workgroup per CU
        rc = launch_ptr(rtiow_fast_general_kernel<NT, SD, true, false, true>, NT, rb);
      } else if (media) {  // several stages (media, unbounded Planes): the boundary walks and the Isotropic phase function need the 256-register budget
        constexpr int NT = 512, SD = 40;
        size_t rb = fastg_lds(NT, SD);
        rc = trans ? launch_ptr(rtiow_fast_general_kernel<NT, SD, true, false, true>, NT, rb) : launch_ptr(rtiow_fast_general_kernel<NT, SD, false, false, true>, NT, rb);
      } else if (trans) {  // 512 lanes per CU (the transcendental texture code needs 256 VGPRs), 40-entry stacks
        constexpr int NT = 512, SD = 40;
        size_t rb = fastg_lds(NT, SD);
        rc = launch_ptr(rtiow_fast_general_kernel<NT, SD, true>, NT, rb);
#ifdef RL_EXPERIMENTAL
      } else if (g_sw.fastg512) {
        constexpr int NT = 512, SD = 40;
        size_t rb = fastg_lds(NT, SD);
        rc = launch_ptr(rtiow_fast_general_kernel<NT, SD, false>, NT, rb);
#endif
      } else {  // 768 lanes per CU (3 waves per SIMD hide more of the node-fetch latency), 20-entry stacks: 208 B of LDS per lane, and the
        // 4 KB that leaves of 160 KB hold the top 32 nodes of the tree (cfg 4 +1.1 %, cfg 5 +0.4 %; 128 nodes with 16-entry stacks: the same)
        constexpr int NT = 768, SD = 20;
        size_t rb = fastg_lds(NT, SD);
#ifdef RL_EXPERIMENTAL  // eight-wide quantised nodes (FastNodeO): cow scene 6.1 -> 3.8 steps per ray but -3.5 %, cfg 5 16.5 -> 13.9 steps, +10 % LEAF visits, -15 %
        if (!H.fg.onodes.empty() && g_sw.fastg_octo != 0) rc = launch_ptr(rtiow_fast_general_kernel<NT, SD, false, true>, NT, rb);
        else
#endif
          rc = launch_ptr(rtiow_fast_general_kernel<NT, SD, false>, NT, rb);
      }
    } else if (variant == 4) {
      // 512 lanes per CU (2 waves per SIMD): the kernel needs ~200 VGPRs (~260 with the sin / Perlin / acos / atan2 code of
      // scenes that have Noise textures or Image textures on spheres).  At 768 lanes (168 VGPRs) the spills land in the TRAV
      // loop and cost 2.3x (measured, cfg 4: 1101 vs 465 Mrays/s; RL_GENERAL_NT=768 in the experimental library).
      bool trans = rt.has_noise || rt.has_sphere_uv;
      size_t rb = (size_t)16 * 512 * sizeof(unsigned long long);
      if (rt.has_media) {  // + the parked HitRecord of a medium scope: 96 B of LDS per lane
        size_t mb = (size_t)512 * (16 + MEDIA_SAVE_WORDS) * sizeof(unsigned long long);
        if (trans) rc = want_stats ? launch(rtiow_wave_general_kernel<512, true, true, true>, 512, mb, false) : launch(rtiow_wave_general_kernel<512, true, false, true>, 512, mb, false);
        else rc = want_stats ? launch(rtiow_wave_general_kernel<512, false, true, true>, 512, mb, false) : launch(rtiow_wave_general_kernel<512, false, false, true>, 512, mb, false);
#ifdef RL_EXPERIMENTAL
      } else if (g_sw.general_nt == 768) {
        size_t rb7 = (size_t)16 * 768 * sizeof(unsigned long long);
        if (trans) rc = want_stats ? launch(rtiow_wave_general_kernel<768, true, true>, 768, rb7, false) : launch(rtiow_wave_general_kernel<768, true, false>, 768, rb7, false);
        else rc = want_stats ? launch(rtiow_wave_general_kernel<768, false, true>, 768, rb7, false) : launch(rtiow_wave_general_kernel<768, false, false>, 768, rb7, false);
#endif
      } else if (trans) rc = want_stats ? launch(rtiow_wave_general_kernel<512, true, true>, 512, rb, false) : launch(rtiow_wave_general_kernel<512, true, false>, 512, rb, false);
      else rc = want_stats ? launch(rtiow_wave_general_kernel<512, false, true>, 512, rb, false) : launch(rtiow_wave_general_kernel<512, false, false>, 512, rb, false);
#ifdef RL_EXPERIMENTAL
    } else if (variant == 5) {
      constexpr int NT = 512;
      rc = want_stats ? launch(rtiow_pool_kernel<NT, true>, NT, (size_t)NT * 192, true) : launch(rtiow_pool_kernel<NT, false>, NT, (size_t)NT * 192, true);
    } else if (variant == 7) {
      constexpr int NT = 512;
      size_t rb = (size_t)8 * 2 * NT * sizeof(unsigned long long);
      rc = want_stats ? launch(rtiow_wave2_kernel<NT, true, true>, NT, rb, true) : launch(rtiow_wave2_kernel<NT, true, false>, NT, rb, true);
    } else if (variant == 6) {
      constexpr int NT = 256;
      rc = want_stats ? launch(rtiow_pool_kernel<NT, true>, NT, (size_t)NT * 192, true) : launch(rtiow_pool_kernel<NT, false>, NT, (size_t)NT * 192, true);
    } else if (variant == 1) {  // the first correct kernel: nested loops, exact divisions
      constexpr int NT = 1024;
      size_t rb = (size_t)8 * NT * sizeof(unsigned long long);
      bool in_lds = rb + scene_bytes <= g_lds_max;
      if (in_lds) rc = want_stats ? launch(rtiow_spheres_kernel<NT, true, true>, NT, rb, true) : launch(rtiow_spheres_kernel<NT, true, false>, NT, rb, true);
      else rc = want_stats ? launch(rtiow_spheres_kernel<NT, false, true>, NT, rb, false) : launch(rtiow_spheres_kernel<NT, false, false>, NT, rb, false);
#define RL_LAUNCH_WAVE(NT)                                                                                              \
  {                                                                                                                     \
    size_t rb = (size_t)16 * NT * sizeof(unsigned long long);                                                           \
    bool in_lds = rb + scene_bytes <= g_lds_max;                                                                        \
    if (in_lds) rc = want_stats ? launch(rtiow_wave_kernel<NT, 1, true>, NT, rb, true) : launch(rtiow_wave_kernel<NT, 1, false>, NT, rb, true); \
    else rc = want_stats ? launch(rtiow_wave_kernel<NT, 0, true>, NT, rb, false) : launch(rtiow_wave_kernel<NT, 0, false>, NT, rb, false);   \
  }
    } else if (variant == 768) RL_LAUNCH_WAVE(768)
    else if (variant == 256) RL_LAUNCH_WAVE(256)
    else if (variant == 512) RL_LAUNCH_WAVE(512)
#undef RL_LAUNCH_WAVE
#else
    }
#endif
    else if (variant == 1025) {  // 4 waves per SIMD: rings + linked ops in LDS, spheres read from L2
      constexpr int NT = 1024;
      size_t rb = (size_t)16 * NT * sizeof(unsigned long long) + (size_t)P.n_ops * sizeof(DevOp);
      rc = want_stats ? launch(rtiow_wave_kernel<NT, 2, true>, NT, rb, false) : launch(rtiow_wave_kernel<NT, 2, false>, NT, rb, false);
    } else if (variant == 1033) {  // A/B: EVERY pixel through the cooperative one-wave-per-pixel kernel (rl_rtiow_coop.h)
      rl_scene *ms = const_cast<rl_scene *>(scene);
      const size_t npix = (size_t)nrows * W;
      if (ms->coop_pixels_cap < npix) {
        hipFree(ms->d_coop_pixels);
        ms->d_coop_pixels = nullptr, ms->coop_pixels_cap = 0;
        HIP_TRY(hipMalloc((void **)&ms->d_coop_pixels, npix * sizeof(uint32_t)));
        ms->coop_pixels_cap = npix;
      }
      hipLaunchKernelGGL(iota_u32, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, stream, ms->d_coop_pixels, (uint32_t)npix);
      rc = launch_coop(ms->d_coop_pixels, (uint32_t)npix);
    } else if (variant == 1029) {  // 4 waves per SIMD: rings + fast traversal nodes in LDS, spheres read from L2; never a counting render
      constexpr int NT = 1024;
      size_t rb = (size_t)16 * NT * sizeof(unsigned long long) + fast_bytes;
      // leave a TRAV round when fewer than a quarter of the lanes it started with are still walking (the counting kernels: 3/8): with
      // pair nodes and inline misses the walks are short and uneven — +2.1 % (6596 -> 6734 Mrays/s; 24,2: 6740; 24,1: 6586; tools: RL_TUNE)
      if (!g_sw.tune_set) P.tune[1] = 4, P.tune[2] = 4;
      if (steal) rc = launch(rtiow_wave_kernel<NT, 4, false, true>, NT, rb, false);
#ifdef RL_EXPERIMENTAL  // <.., 4, true> only under rl_debug_fast_stats (tools/sched.py): scheduler occupancy of the fast kernel; its box / sphere counts are its own
      else if (want_stats) rc = launch(rtiow_wave_kernel<NT, 4, true>, NT, rb, false);
#endif
      else rc = launch(rtiow_wave_kernel<NT, 4, false>, NT, rb, false);
    } else if (variant == 1027) {  // 4 waves per SIMD: rings + compact guarded ops in LDS, spheres read from L2
      constexpr int NT = 1024;
      size_t rb = (size_t)16 * NT * sizeof(unsigned long long) + compact_bytes;
      rc = want_stats ? launch(rtiow_wave_kernel<NT, 3, true>, NT, rb, false) : launch(rtiow_wave_kernel<NT, 3, false>, NT, rb, false);
    } else {  // 1024: everything through L2 (sphere-only worlds too large for LDS)
      constexpr int NT = 1024;
      size_t rb = (size_t)16 * NT * sizeof(unsigned long long);
      rc = want_stats ? launch(rtiow_wave_kernel<NT, 0, true>, NT, rb, false) : launch(rtiow_wave_kernel<NT, 0, false>, NT, rb, false);
    }
    return rc;
  };
  P.sample_begin = 0, P.sample_end = cam->samples_per_pixel, P.resume = 0;
  P.pos_state = nullptr, P.tile_order = nullptr, P.tile_cost = nullptr;
  // Cost-sorted two-phase render (wave kernels, spp >= 64): a short first launch renders samples [0, 8) of every
  // pixel and records each 8x8 tile's ray count; the tiles are then sorted by cost and the remaining samples are
  // rendered expensive-tiles-first (LPT), so the tail of the launch holds cheap pixels only.  Pixels are resumed
  // with their exact sums and ChaCha word positions: results are bit-identical to a single launch.
  const bool lpt_enabled = g_sw.lpt;
  const uint32_t lpt_first = 8;  // (round 3: 2 / 4 / 16 probe samples measure 6727 / 6752 / 6710 Mrays/s against 6711 — flat)
  bool lpt = lpt_enabled && variant != 1035 && (variant >= 256 || variant == 4 || variant == 1031 || variant == 5 || variant == 6 || variant == 7) && cam->samples_per_pixel >= 64;
  if (want_stats) HIP_TRY(hipEventRecord(scene->ev0, stream));
  int rc = RL_OK;
  if (variant == 3) {
#ifdef RL_EXPERIMENTAL
    rc = render_wavefront(scene, P, nrows, stream, want_stats);
#endif
  } else if (!lpt) rc = launch_variant();
  else {
    rl_scene *ms = const_cast<rl_scene *>(scene);  // scratch buffers only; the scene program itself is immutable
    size_t npix = (size_t)nrows * W, ntiles = (size_t)(slots >> 6);
    if (ms->lpt_pix < npix) {
      hipFree(ms->d_pos);
      ms->d_pos = nullptr, ms->lpt_pix = 0;
      HIP_TRY(hipMalloc((void **)&ms->d_pos, npix * sizeof(uint32_t)));
      ms->lpt_pix = npix;
    }
    if (ms->lpt_tiles < ntiles) {
      hipFree(ms->d_tile_cost), hipFree(ms->d_tile_order), hipFree(ms->d_tile_keys), hipFree(ms->d_tile_iota);
      ms->d_tile_cost = ms->d_tile_order = ms->d_tile_keys = ms->d_tile_iota = nullptr, ms->lpt_tiles = 0;
      HIP_TRY(hipMalloc((void **)&ms->d_tile_cost, ntiles * sizeof(uint32_t)));
      HIP_TRY(hipMalloc((void **)&ms->d_tile_order, ntiles * sizeof(uint32_t)));
      HIP_TRY(hipMalloc((void **)&ms->d_tile_keys, ntiles * sizeof(uint32_t)));
      HIP_TRY(hipMalloc((void **)&ms->d_tile_iota, ntiles * sizeof(uint32_t)));
      ms->lpt_tiles = ntiles;
    }
    HIP_TRY(hipMemsetAsync(ms->d_tile_cost, 0, ntiles * sizeof(uint32_t), stream));
    P.sample_end = lpt_first, P.pos_state = ms->d_pos, P.tile_cost = ms->d_tile_cost;
    rc = launch_variant();
    if (rc != RL_OK) return rc;
    // tiles by cost, most expensive first, on the device (stable radix sort): the whole render stays asynchronous on `stream`
    rc = rl::sort_tiles_by_cost_desc(ms->d_tile_cost, ms->d_tile_keys, ms->d_tile_iota, ms->d_tile_order, (uint32_t)ntiles, &ms->d_sort_temp, &ms->sort_temp_bytes,
                                     stream);
    if (rc != RL_OK) return rc;
    HIP_TRY(hipMemsetAsync(scene->d_scratch, 0, 4, stream));  // work counter only; stats keep accumulating
    if (scene->progress_on) P.work_counter = scene->d_progress + 1;  // the resume launch counts in the second host-visible word
    P.sample_begin = lpt_first, P.sample_end = cam->samples_per_pixel, P.resume = 1;
    P.tile_order = ms->d_tile_order, P.tile_cost = nullptr;
    if (variant == 1029 && !want_stats && g_sw.steal_max_fill > 0.0 && (double)npix <= g_sw.steal_max_fill * (double)g_cus * 1024.0) {
      // small shard: waves that run out of pixels take over pixels other lanes are still rendering (rl_rtiow_coop.h rtiow_steal_loop)
      if (ms->steal_pix < npix) {
        hipFree(ms->d_steal_state), hipFree(ms->d_steal_n);
        ms->d_steal_state = ms->d_steal_n = nullptr, ms->steal_pix = 0;
        HIP_TRY(hipMalloc((void **)&ms->d_steal_state, npix * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void **)&ms->d_steal_n, npix * sizeof(uint32_t)));
        ms->steal_pix = npix;
      }
      HIP_TRY(hipMemsetAsync(ms->d_steal_state, 0, npix * sizeof(uint32_t), stream));
      HIP_TRY(hipMemsetAsync(scene->d_scratch + 256, 0, 4, stream));
      P.steal_state = ms->d_steal_state, P.steal_n = ms->d_steal_n, P.steal_counter = (uint32_t *)(scene->d_scratch + 256);
      P.coop_leaf_boxes = scene->d_fast_leaf_boxes;
      steal = true;
    }
    if (variant == 1029) {
      // latency modes (A/B only, default off): RL_THIN=<permille of the tiles> renders the most expensive tiles
      // 64 >> RL_THIN_SHIFT pixels per wave, RL_PRIO=<permille> raises the issue priority of the waves that hold them.  Measured on the
      // emulated 1/8 shard: 271 - 283 ms against 279 at best, slower when more than ~1 % of the tiles are thinned: the longest sample
      // chain's time is per-ray LATENCY (9.6 us for a pixel alone on the GPU, tools/lone_ray.py), which neither shortens
#ifdef RL_EXPERIMENTAL
      P.thin_tiles = (uint32_t)((double)ntiles * g_sw.thin_permille / 1000.0);
      P.thin_shift = (uint32_t)g_sw.thin_shift;
      P.prio_tiles = (uint32_t)((double)ntiles * g_sw.prio_permille / 1000.0);  // A/B
      uint64_t total = slots + (((uint64_t)P.thin_tiles * 64u) << P.thin_shift);
      if (total >= 0xFFFF0000ull) P.thin_tiles = 0;
#endif
    }
    rc = launch_variant();
  }
  if (want_stats) HIP_TRY(hipEventRecord(scene->ev1, stream));
  if (rc == RL_OK) rc = mark_render_end(scene, stream);
  return rc;
}
}  // namespace rl

extern "C" {

int rl_rtiow_render_device(const rl_scene *scene, const rl_rtiow_camera *cam, uint64_t first_sample, uint32_t row_first, uint32_t row_step,
                           void *d_out, void *hip_stream, rl_stats *st) {
  if (!g_ready) return set_err(RL_E_NO_DEVICE, "rl_init has not succeeded");
  if (!scene || scene->kind != 1 || !cam || !d_out || row_step == 0) return set_err(RL_E_INVALID, "bad argument");
  if (cam->image_width == 0 || cam->image_height == 0) return set_err(RL_E_INVALID, "empty image");
  hipStream_t stream = (hipStream_t)hip_stream;
  if (row_first >= cam->image_height) {
    if (st) std::memset(st, 0, sizeof *st);
    return RL_OK;
  }
  std::lock_guard<std::mutex> lk(scene->mu);  // concurrent renders of one scene: see rl_scene::mu
  int rc = rl::rtiow_render_launch(scene, cam, first_sample, row_first, row_step, d_out, stream, st != nullptr);
  if (rc != RL_OK) return rc;
  return st ? rl::collect_stats(scene, stream, st) : post_status(scene, stream);
}

// Completion + status of the last ASYNCHRONOUS render of this scene (rl_*_render_device / rl_*_render_multi_device with
// opt_stats == NULL): waits for it, fills rays and flagged (the other counters need a counting render) and returns
// RL_E_DEGENERATE when a reference panic site was reached.  RL_OK with zero counters when nothing is pending.
int rl_render_status(const rl_scene *scene, rl_stats *st) {
  if (!g_ready) return set_err(RL_E_NO_DEVICE, "rl_init has not succeeded");
  if (!scene) return set_err(RL_E_INVALID, "bad argument");
  rl_stats acc;
  std::memset(&acc, 0, sizeof acc);
  g_last_slow_traces = 0;
  size_t n = scene->replicas.empty() ? 1 : scene->replicas.size();
  for (size_t g = 0; g < n; g++) {
    rl_scene *r = const_cast<rl_scene *>(scene->replicas.empty() ? scene : scene->replicas[g]);
    std::lock_guard<std::mutex> lk(r->mu);
    bool any = false;
    for (int k = 0; k < rl_scene::N_STATUS; k++) any |= r->status_pending[k];
    if (any) {
      if (r->ctx < 0 || (size_t)r->ctx >= g_ctx.size() || g_ctx[(size_t)r->ctx].device != r->device)
        return set_err(RL_E_INVALID, "scene belongs to a device context that no longer exists (created under another rl_init / rl_init_multi)");
      int rc = rl::use_context(r->ctx);
      if (rc != RL_OK) return rc;
      for (int k = 0; k < rl_scene::N_STATUS; k++)
        if (r->status_pending[k]) {
          HIP_TRY(hipEventSynchronize(r->ev_status[k]));
          rl::fold_status(r, k);
        }
    }
    // rays: of the most recently enqueued render of this replica (summed over the replicas of a multi-GPU scene); flagged: every render since the last call
    acc.rays += r->folded_rays, acc.flagged += r->folded_flagged;
    g_last_slow_traces += r->folded_slow;
    r->folded_rays = r->folded_flagged = r->folded_slow = 0, r->folded_seq = r->next_seq;
  }
  if (n > 1) rl::use_context(0);
  if (st) *st = acc;
  if (acc.flagged) return set_err(RL_E_DEGENERATE, "a reference panic site was reached (see stats.flagged)");
  return RL_OK;
}

int rl_rtiow_render_progress(const rl_scene *scene, uint64_t *pixels_claimed, uint64_t *pixels_total, uint32_t *phase) {
  if (!g_ready) return set_err(RL_E_NO_DEVICE, "rl_init has not succeeded");
  if (!scene || scene->kind != 1) return set_err(RL_E_INVALID, "bad argument");
  rl_scene *ms = const_cast<rl_scene *>(scene);
  unsigned long long total = 0, c0 = 0, c1 = 0;
  {
    std::lock_guard<std::mutex> lk(ms->mu);
    if (!ms->progress_on) {  // first call: renders enqueued from now on count in host-visible memory
      if (ms->ctx < 0 || (size_t)ms->ctx >= g_ctx.size()) return set_err(RL_E_INVALID, "scene belongs to a device context that no longer exists");
      int rc = rl::use_context(ms->ctx);
      if (rc != RL_OK) return rc;
      HIP_TRY(hipHostMalloc((void **)&ms->h_progress, 64, hipHostMallocMapped | hipHostMallocCoherent));
      std::memset(ms->h_progress, 0, 64);
      HIP_TRY(hipHostGetDevicePointer((void **)&ms->d_progress, ms->h_progress, 0));
      ms->progress_on = true;
    }
    total = ms->progress_total;
    c0 = __atomic_load_n(&ms->h_progress[0], __ATOMIC_RELAXED), c1 = __atomic_load_n(&ms->h_progress[1], __ATOMIC_RELAXED);
  }
  // lanes that find the queue empty still bump the counter: clamp; the resume launch has started once its counter moves
  const unsigned ph = c1 != 0 ? 1u : 0u;
  unsigned long long claimed = ph ? c1 : c0;
  if (claimed > total) claimed = total;
  if (pixels_claimed) *pixels_claimed = claimed;
  if (pixels_total) *pixels_total = total;
  if (phase) *phase = ph;
  return RL_OK;
}

// Not part of the ABI (tests / tools only): force an RTIOW kernel variant (0 auto, 1 nested-loop, 2 general, 512/768/1024 wave).
// Host-only self-check of the traversal structures a scene compiles to (no device needed: tests/test_host_structures.py runs it in the
// CPU tier).  out[16]: [0] bit 0: sphere fast tree built, bit 1: general fast structure built; general structure: [1] items,
// [2] binary nodes, [3] four-wide nodes, [4] leaf entries reached from the root, [5] items reached more than once, [6] items never
// reached, [7] boxes that fail to contain what lies below them (child boxes of inner children; the vertices / swept spheres of
// world-space items), [8] depth of the four-wide tree; sphere tree: [9] inner nodes, [10] leaves reached, [11] spheres reached more
// than once or never, [12] boxes that fail to contain the swept sphere below, [13] depth.  Returns RL_OK or the compile error.
int rl_debug_host_structures(const rl_rtiow_scene_desc *desc, unsigned long long *out16) {
  if (!desc || !out16) return set_err(RL_E_INVALID, "bad argument");
  std::shared_ptr<const HostRtiow> H;
  int rc = build_host_rtiow(desc, H);
  if (rc != RL_OK) return rc;
  for (int i = 0; i < 16; i++) out16[i] = 0;
  const FastGeneral &fg = H->fg;
  if (H->fast_root != FAST_NONE) out16[0] |= 1ull;
  if (fg.ok) out16[0] |= 2ull;
  if (fg.ok) {
    out16[1] = fg.items.size(), out16[2] = fg.nodes.size(), out16[3] = fg.qnodes.size();
    std::vector<uint32_t> seen(fg.items.size(), 0u);
    struct Frame {
      uint32_t e;
      float box[6];
      bool has_box;
      unsigned depth;
    };
    std::vector<Frame> st;
    for (uint32_t sr : fg.stage_roots)  // every stage: plane leaves, segment trees, media (their entries carry FASTG_MEDIUM and are no items)
      if (sr != NONE) st.push_back(Frame{sr, {0, 0, 0, 0, 0, 0}, false, 1u});
    auto inside = [](const float *outer, const double *lo, const double *hi) {
      for (int ax = 0; ax < 3; ax++)
        if (!((double)outer[2 * ax] <= lo[ax] && hi[ax] <= (double)outer[2 * ax + 1])) return false;
      return true;
    };
    while (!st.empty()) {
      Frame f = st.back();
      st.pop_back();
      out16[8] = std::max<unsigned long long>(out16[8], f.depth);
      if ((f.e & FASTG_LEAF) && (f.e & FASTG_MEDIUM)) continue;
      if (f.e & FASTG_LEAF) {
        const uint32_t item = f.e & ~FASTG_LEAF;
        out16[4]++;
        if (item >= seen.size()) {
          out16[7]++;
          continue;
        }
        seen[item]++;
        const FastItem &it = fg.items[item];
        if (f.has_box && it.chain == NONE) {  // world-space item: its geometry must lie inside the leaf's box
          double lo[3], hi[3];
          if (it.kind == 0) {
            const rl_sphere &sp = desc->spheres[it.payload & SPH_INDEX];
            for (int ax = 0; ax < 3; ax++) {
              double c1 = sp.moving ? sp.center1[ax] : sp.center0[ax];
              lo[ax] = std::fmin(sp.center0[ax], c1) - std::fabs(sp.radius), hi[ax] = std::fmax(sp.center0[ax], c1) + std::fabs(sp.radius);
            }
          } else if (desc->planars[it.payload].kind == RL_PLANAR_PLANE) {
            out16[7]++;  // an unbounded Plane must not sit below a box
            continue;
          } else {
            const rl_planar &pl = desc->planars[it.payload];
            for (int ax = 0; ax < 3; ax++) {
              double v[4] = {pl.q[ax], pl.q[ax] + pl.u[ax], pl.q[ax] + pl.v[ax], pl.kind == RL_PLANAR_QUAD ? pl.q[ax] + pl.u[ax] + pl.v[ax] : pl.q[ax]};
              lo[ax] = std::fmin(std::fmin(v[0], v[1]), std::fmin(v[2], v[3])), hi[ax] = std::fmax(std::fmax(v[0], v[1]), std::fmax(v[2], v[3]));
            }
          }
          if (!inside(f.box, lo, hi)) out16[7]++;
        }
        continue;
      }
      if (f.e >= fg.qnodes.size()) {
        out16[7]++;
        continue;
      }
      const FastNodeQ &q = fg.qnodes[f.e];
      double ulo[3] = {INFINITY, INFINITY, INFINITY}, uhi[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (int k = 0; k < 4; k++) {
        if (q.child[k] == NONE) continue;
        Frame c{q.child[k], {q.lo[0][k], q.hi[0][k], q.lo[1][k], q.hi[1][k], q.lo[2][k], q.hi[2][k]}, true, f.depth + 1u};
        for (int ax = 0; ax < 3; ax++) ulo[ax] = std::fmin(ulo[ax], (double)c.box[2 * ax]), uhi[ax] = std::fmax(uhi[ax], (double)c.box[2 * ax + 1]);
        st.push_back(c);
      }
      if (f.has_box && !inside(f.box, ulo, uhi)) out16[7]++;  // a node's box (held by its parent) contains its children's boxes
    }
    for (uint32_t c : seen) out16[5] += c > 1u ? 1u : 0u, out16[6] += c == 0u ? 1u : 0u;
    // media: [14] = how many, [15] = one byte per medium (the first eight): FastMedium::shape, + 0x10 when the medium has a box node
    out16[14] = fg.media.size();
    for (size_t k = 0; k < fg.media.size() && k < 8; k++)
      out16[15] |= (unsigned long long)(fg.media[k].shape | (fg.media_stage[k] != NONE && !(fg.stage_roots[fg.media_stage[k]] & FASTG_LEAF) ? 0x10u : 0u)) << (8 * k);
  }
  if (H->fast_root != FAST_NONE) {
    const std::vector<FastNode> &nodes = H->fast_nodes;
    const uint32_t n_inner = (uint32_t)nodes.size(), n_sph = desc->n_spheres;
    out16[9] = n_inner;
    std::vector<uint32_t> seen(n_sph, 0u);
    struct Frame {
      uint32_t e;
      float box[6];
      bool has_box;
      unsigned depth;
    };
    std::vector<Frame> st;
    st.push_back(Frame{H->fast_root, {0, 0, 0, 0, 0, 0}, false, 1u});
    while (!st.empty()) {
      Frame f = st.back();
      st.pop_back();
      out16[13] = std::max<unsigned long long>(out16[13], f.depth);
      if (f.e >= n_inner) {
        const uint32_t si = f.e - n_inner;
        out16[10]++;
        if (si >= n_sph) {
          out16[12]++;
          continue;
        }
        seen[si]++;
        if (f.has_box) {
          const rl_sphere &sp = desc->spheres[si];
          for (int ax = 0; ax < 3; ax++) {
            double c1 = sp.moving ? sp.center1[ax] : sp.center0[ax];
            double lo = std::fmin(sp.center0[ax], c1) - std::fabs(sp.radius), hi = std::fmax(sp.center0[ax], c1) + std::fabs(sp.radius);
            if (!((double)f.box[2 * ax] <= lo && hi <= (double)f.box[2 * ax + 1])) {
              out16[12]++;
              break;
            }
          }
        }
        continue;
      }
      const FastNode &nd = nodes[f.e];
      for (int k = 0; k < 2; k++) {
        Frame c{k == 0 ? (nd.child & 0xFFFFu) : (nd.child >> 16), {nd.box[k][0], nd.box[k][1], nd.box[k][2], nd.box[k][3], nd.box[k][4], nd.box[k][5]}, true, f.depth + 1u};
        if (f.has_box)
          for (int ax = 0; ax < 3; ax++)
            if (!(f.box[2 * ax] <= c.box[2 * ax] && c.box[2 * ax + 1] <= f.box[2 * ax + 1])) {
              out16[12]++;
              break;
            }
        st.push_back(c);
      }
    }
    for (uint32_t c : seen) out16[11] += c != 1u ? 1u : 0u;
  }
  return RL_OK;
}
void rl_debug_set_rtiow_variant(int v) { g_sw.rtiow_variant = v; }
void rl_debug_set_lpt(int on) { g_sw.lpt = on != 0; }
void rl_debug_set_coop(int on) { g_sw.coop_small = on != 0; }
void rl_debug_set_steal(double max_fill) { g_sw.steal_max_fill = max_fill; }
void rl_debug_set_fast_traversal(int on) { g_sw.fast_traversal = on != 0; }
void rl_debug_set_rtc_blocks(int per_cu) { g_sw.rtc_blocks_per_cu = per_cu < 0 ? 0 : per_cu; }  // 0: as many as are resident (default); n: n per CU (tests)
void rl_debug_set_fastg_one_wave(int mode) { g_sw.fastg_nt256 = mode; }  // -1: by frame size (default), 0 / 1: never / always the one-wave-per-SIMD form (tests)
void rl_debug_fast_stats(int on) {  // the instrumented fast kernel <1024, 4, true> exists in the experimental library only
#ifdef RL_EXPERIMENTAL
  g_fast_debug_stats = on != 0;
#else
  (void)on;
#endif
}
#ifdef RL_FASTG_VERIFY
int rl_debug_fastg_verify(unsigned int *count, double *log768) {
  HIP_TRY(hipMemcpyFromSymbol(count, HIP_SYMBOL(rl::g_vcount), 4));
  HIP_TRY(hipMemcpyFromSymbol(log768, HIP_SYMBOL(rl::g_vlog), 64 * 12 * 8));
  return RL_OK;
}
int rl_debug_fastg_counts(unsigned long long *out4) {
  HIP_TRY(hipMemcpyFromSymbol(out4, HIP_SYMBOL(rl::g_vstats), 32));
  return RL_OK;
}
#endif
// rays of the render rl_render_status last waited for that the fast traversal re-traced in the reference's order
unsigned long long rl_debug_slow_traces(void) { return g_last_slow_traces; }
int rl_debug_has_experimental(void) {
#ifdef RL_EXPERIMENTAL
  return 1;
#else
  return 0;
#endif
}

// Not part of the ABI (tools only): scheduler occupancy counters of the last STATS launch, 32 x u64.
int rl_debug_sched(const rl_scene *scene, unsigned long long *out32) {
  if (!scene || !out32) return RL_E_INVALID;
  HIP_TRY(hipMemcpy(out32, scene->d_scratch + 128, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return RL_OK;
}

// Not part of the ABI (tools only): per-pixel ray counts of the NEXT counting renders of this scene, n_pixels u32 (rows x W of
// the shard rendered); rl_debug_pixel_rays_read copies them back.  Pass 0 to switch the recording off.
int rl_debug_pixel_rays(const rl_scene *scene, uint64_t n_pixels) {
  if (!scene) return RL_E_INVALID;
  rl_scene *ms = const_cast<rl_scene *>(scene);
  hipFree(ms->d_pix_rays);
  ms->d_pix_rays = nullptr;
  if (n_pixels) {
    HIP_TRY(hipMalloc((void **)&ms->d_pix_rays, n_pixels * sizeof(uint32_t)));
    HIP_TRY(hipMemset(ms->d_pix_rays, 0, n_pixels * sizeof(uint32_t)));
  }
  return RL_OK;
}
int rl_debug_pixel_rays_read(const rl_scene *scene, uint32_t *out, uint64_t n_pixels) {
  if (!scene || !scene->d_pix_rays || !out) return RL_E_INVALID;
  HIP_TRY(hipMemcpy(out, scene->d_pix_rays, n_pixels * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return RL_OK;
}

int rl_rtiow_render_rows(const rl_scene *scene, const rl_rtiow_camera *cam, uint64_t first_sample, uint32_t row_first, uint32_t row_step,
                         double *out, rl_stats *st) {
  if (!g_ready) return set_err(RL_E_NO_DEVICE, "rl_init has not succeeded");
  if (!scene || !cam || !out || row_step == 0) return set_err(RL_E_INVALID, "bad argument");
  uint32_t H = cam->image_height, W = cam->image_width;
  uint32_t nrows = row_first < H ? (H - row_first + row_step - 1) / row_step : 0;
  size_t bytes = (size_t)nrows * W * 3 * sizeof(double);
  if (bytes == 0) {
    if (st) std::memset(st, 0, sizeof *st);
    return RL_OK;
  }
  int rc0 = rl::use_context(scene->ctx);
  if (rc0 != RL_OK) return rc0;
  double *d_out = nullptr;
  HIP_TRY(hipMalloc((void **)&d_out, bytes));
  rl_stats local;
  int rc = rl_rtiow_render_device(scene, cam, first_sample, row_first, row_step, d_out, g_ctx[(size_t)scene->ctx].stream, &local);
  if (rc == RL_OK || rc == RL_E_DEGENERATE) {
    hipError_t e = hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = set_err(RL_E_DEVICE, std::string("hipMemcpy D2H: ") + hipGetErrorString(e));
  }
  hipFree(d_out);
  if (st) *st = local;
  return rc;
}

int rl_rtiow_encode_rgb8_device(const void *d_rgb_sum, uint64_t n_pixels, uint32_t samples, void *d_rgb8, void *hip_stream) {
  if (!g_ready) return set_err(RL_E_NO_DEVICE, "rl_init has not succeeded");
  if (!d_rgb_sum || !d_rgb8 || samples == 0) return set_err(RL_E_INVALID, "bad argument");
  unsigned long long n = n_pixels * 3ull;
  if (n == 0) return RL_OK;
  hipLaunchKernelGGL(encode_rtiow_rgb8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, (const double *)d_rgb_sum, n,
                     1.0 / (double)samples, (unsigned char *)d_rgb8);
  HIP_TRY(hipGetLastError());
  return RL_OK;
}

int rl_rtiow_render_rgb8(const rl_scene *scene, const rl_rtiow_camera *cam, uint64_t first_sample, uint8_t *out, rl_stats *st) {
  if (!g_ready) return set_err(RL_E_NO_DEVICE, "rl_init has not succeeded");
  if (!scene || !cam || !out) return set_err(RL_E_INVALID, "bad argument");
  size_t npix = (size_t)cam->image_width * cam->image_height;
  if (npix == 0 || cam->samples_per_pixel == 0) return set_err(RL_E_INVALID, "empty image / zero samples");
  int rc0 = rl::use_context(scene->ctx);
  if (rc0 != RL_OK) return rc0;
  hipStream_t stream = g_ctx[(size_t)scene->ctx].stream;
  double *d_sum = nullptr;
  unsigned char *d_u8 = nullptr;
  HIP_TRY(hipMalloc((void **)&d_sum, npix * 3 * sizeof(double)));
  hipError_t e = hipMalloc((void **)&d_u8, npix * 3);
  if (e != hipSuccess) {
    hipFree(d_sum);
    return set_err(RL_E_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
  }
  rl_stats local;
  int rc = rl_rtiow_render_device(scene, cam, first_sample, 0, 1, d_sum, stream, &local);
  if (rc == RL_OK || rc == RL_E_DEGENERATE) {
    int rc2 = rl_rtiow_encode_rgb8_device(d_sum, npix, cam->samples_per_pixel, d_u8, stream);
    if (rc2 != RL_OK) rc = rc2;
    else {
      e = hipMemcpyAsync(out, d_u8, npix * 3, hipMemcpyDeviceToHost, stream);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      if (e != hipSuccess) rc = set_err(RL_E_DEVICE, std::string("D2H: ") + hipGetErrorString(e));
    }
  }
  hipFree(d_sum), hipFree(d_u8);
  if (st) *st = local;
  return rc;
}

int rl_rtiow_render(const rl_scene *scene, const rl_rtiow_camera *cam, uint64_t first_sample, double *out, rl_stats *st) {
  return rl_rtiow_render_rows(scene, cam, first_sample, 0, 1, out, st);
}

// ------------------------------------------------------------------ RTC
// Reject-only acceleration for the triangle Groups of the RTC path.  The reference's Group (group.rs) intersects every child
// for every ray; for each ROP_TRIS range of >= 8 triangles a binary tree over the triangles IN INDEX ORDER is built here (node =
// index range, box = padded union of the triangles' bounding boxes, stored as floats; leaves = single triangles; nodes in
// depth-first order so that a left-to-right walk meets the triangles in the reference's order) and the op's `skip` field gets
// root + 1.  The kernel skips a triangle only when its box is CERTAINLY missed (guard_reject32), so every intersection the
// reference would keep is still found, in the same order.
static void build_rtc_guards(RtcProgram &rc, std::vector<RtcGuard> &guards) {
  struct Builder {
    const std::vector<DevTri> &tris;
    std::vector<RtcGuard> &out;
    void box_of(uint32_t l, uint32_t r, float *b) const {
      double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (uint32_t i = l; i < r; i++) {
        const DevTri &t = tris[i];
        for (int ax = 0; ax < 3; ax++) {
          const double v[3] = {t.p1[ax], t.p1[ax] + t.e1[ax], t.p1[ax] + t.e2[ax]};
          for (double x : v) mn[ax] = std::fmin(mn[ax], x), mx[ax] = std::fmax(mx[ax], x);
        }
      }
      bool bad = false;
      for (int ax = 0; ax < 3; ax++) {
        double pad = 1e-6 * (std::fabs(mn[ax]) + std::fabs(mx[ax]) + (mx[ax] - mn[ax])) + 1e-30;  // the box only ever rejects: keep it outside the triangles
        b[2 * ax] = (float)(mn[ax] - pad), b[2 * ax + 1] = (float)(mx[ax] + pad);
        bad |= !(std::fabs(mn[ax]) <= 1e30 && std::fabs(mx[ax]) <= 1e30);
      }
      if (bad)
        for (int k = 0; k < 6; k++) b[k] = NAN;  // never certainly missed
    }
    void build(uint32_t l, uint32_t r) {
      const uint32_t idx = (uint32_t)out.size();
      out.push_back(RtcGuard{});
      RtcGuard nd{};
      box_of(l, r, nd.box);
      if (r - l == 1) nd.tri = l, nd.skip = idx + 1;
      else {
        const uint32_t m = l + (r - l) / 2;
        build(l, m);
        build(m, r);
        nd.tri = NONE, nd.skip = (uint32_t)out.size();
      }
      out[idx] = nd;
    }
  };
  Builder b{rc.tris, guards};
  for (DevOp &op : rc.ops)
    if (op.code == ROP_TRIS && op.b >= 8) {
      op.skip = (uint32_t)guards.size() + 1u;
      b.build(op.a, op.a + op.b);
    }
}

static rl_scene *upload_rtc(const std::shared_ptr<const HostRtc> &H, int ctx) {
  rl_scene *s = new rl_scene();
  s->kind = 2, s->ctx = ctx, s->device = g_ctx[(size_t)ctx].device, s->hrc = H;
  const RtcProgram &rc_ = H->rc;
  int rc = RL_OK;
  if ((!H->guards.empty() && (rc = upload(H->guards, &s->d_guards))) || (rc = upload(rc_.ops, &s->d_ops)) || (rc = upload(rc_.tris, &s->d_tris)) ||
      (rc = upload(rc_.xforms, &s->d_xforms)) || (rc = upload(rc_.materials, &s->d_rmaterials)) || (rc = upload(rc_.lights, &s->d_lights)) ||
      (rc = upload(rc_.shapes, &s->d_shapes)) || (rc = upload(rc_.csgs, &s->d_csgs)) || (rc = upload(rc_.patterns, &s->d_patterns)) || (rc = scene_common(s))) {
    destroy_one(s);
    return nullptr;
  }
  return s;
}

rl_scene *rl_rtc_scene_create(const rl_rtc_scene_desc *desc) {
  if (!g_ready) {
    set_err(RL_E_NO_DEVICE, "rl_init has not succeeded (no GPU, or not called)");
    return nullptr;
  }
  if (!desc) {
    set_err(RL_E_INVALID, "null scene descriptor");
    return nullptr;
  }
  auto H = std::make_shared<HostRtc>();
  std::string err;
  if (compile_rtc(*desc, H->rc, err) != RL_OK) {
    set_err(RL_E_INVALID, err);
    return nullptr;
  }
  if (H->rc.lights.size() > 0xFFFFu) {
    set_err(RL_E_UNSUPPORTED, "too many lights");
    return nullptr;
  }
  // rtc_full_kernel walks the reflection / refraction tree with a fixed stack of pending rays (RTC_MAX_PENDING): a material that is
  // both reflective and transparent keeps up to max_reflection_depth + 1 of them alive (world.rs:128-159 recurses without a cap)
  if (H->rc.needs_full && H->rc.max_reflection_depth + 1u > RTC_MAX_PENDING) {
    set_err(RL_E_UNSUPPORTED, "max_reflection_depth above " + std::to_string(RTC_MAX_PENDING - 1) + " is not supported by the device kernel");
    return nullptr;
  }
  build_rtc_guards(H->rc, H->guards);  // both kernels walk them (the full kernel with the unbounded line test)
  std::shared_ptr<const HostRtc> Hc = H;
  std::vector<rl_scene *> reps;
  for (int g = 0; g < (int)g_ctx.size(); g++) {
    rl_scene *r = rl::use_context(g) == RL_OK ? upload_rtc(Hc, g) : nullptr;
    if (!r) {
      for (rl_scene *q : reps) destroy_one(q);
      hipSetDevice(g_ctx[0].device);
      return nullptr;
    }
    reps.push_back(r);
  }
  hipSetDevice(g_ctx[0].device);
  if (reps.size() > 1) reps[0]->replicas = reps;
  return reps[0];
}

}  // extern "C"

namespace rl {
int rtc_render_launch(const rl_scene *scene, const rl_rtc_camera *cam, uint32_t aa, uint32_t row_first, uint32_t row_step, void *d_out, hipStream_t stream,
                      bool want_stats) {
  const RtcProgram &rc_ = scene->rc();
  uint32_t H = cam->vsize, W = cam->hsize;
  uint32_t nrows = row_first < H ? (H - row_first + row_step - 1) / row_step : 0;
  const uint32_t n_guards = (uint32_t)scene->hrc->guards.size();
  RtcParams P{};
  P.ops = scene->d_ops, P.tris = scene->d_tris, P.xforms = scene->d_xforms, P.materials = scene->d_rmaterials, P.lights = scene->d_lights;
  P.n_ops = (uint32_t)rc_.ops.size(), P.n_tris = (uint32_t)rc_.tris.size();
  P.guards = n_guards ? scene->d_guards : nullptr, P.n_guards = n_guards;
  P.n_xforms = (uint32_t)rc_.xforms.size(), P.n_lights = (uint32_t)rc_.lights.size();
  P.cam = *cam;
  P.aa = aa;
  P.row_first = row_first, P.row_step = row_step, P.nrows = nrows;
  std::memcpy(P.void_color, rc_.void_color, 24);
  P.out = (double *)d_out;
  P.stats = (unsigned long long *)(scene->d_scratch + 64);
  {
    int rco = order_after_previous(scene, stream);
    if (rco != RL_OK) return rco;
  }
  HIP_TRY(hipMemsetAsync(scene->d_scratch, 0, 512, stream));
  constexpr int NT = 256;
  size_t scene_bytes = (size_t)P.n_ops * sizeof(DevOp) + (size_t)P.n_tris * sizeof(DevTri) + (size_t)P.n_guards * sizeof(RtcGuard);
  bool lds_scene = scene_bytes <= 65536;
  size_t lds = lds_scene ? scene_bytes : 0;
  uint64_t total = (uint64_t)W * nrows;
  uint64_t want = (total + NT - 1) / NT;
  // The grid is exactly what is RESIDENT at once (occupancy API: 2 workgroups per CU for rtc_kernel's 241 VGPRs, 3 for rtc_full_kernel's budget):
  // every workgroup stages the scene in LDS once and strides over the pixels.  Measured against the 8 per CU of rounds 1 - 2 (workgroups
  // queueing behind the resident ones, each staging the scene again, the last round of them half empty): teapot AA 1 0.548 -> 0.317 ms per
  // frame, AA 8 19.5 -> 17.0 ms; mirror scene 13.5 -> 10.8 ms, CSG scene 2.11 -> 1.76 ms.  RL_RTC_BLOCKS=<n> forces n per CU (A/B).
  auto grid_for = [&](const void *kern, size_t dyn_lds) -> uint32_t {
    uint64_t per_cu = (uint64_t)g_sw.rtc_blocks_per_cu;
    if (per_cu == 0) {
      static std::mutex mu;
      static std::map<std::pair<const void *, size_t>, int> cache;
      std::lock_guard<std::mutex> lk(mu);
      auto key = std::make_pair(kern, dyn_lds ^ ((size_t)scene->device << 48));
      auto it = cache.find(key);
      if (it == cache.end()) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NT, dyn_lds) != hipSuccess || nb < 1) nb = 2;
        it = cache.emplace(key, nb).first;
      }
      per_cu = (uint64_t)it->second;
    }
    return (uint32_t)(want < (uint64_t)g_cus * per_cu ? want : (uint64_t)g_cus * per_cu);
  };
  uint32_t blocks = 0;
  if (want_stats) HIP_TRY(hipEventRecord(scene->ev0, stream));
  if (rc_.needs_full || g_sw.rtc_force_full) {  // (RL_RTC_FORCE_FULL: A/B, triangle-only worlds through the full kernel)  // shapes / CSG / patterns / reflection / refraction: the full World::color_at kernel
    RtcFullParams F{};
    F.R = P;
    F.shapes = scene->d_shapes, F.csgs = scene->d_csgs, F.patterns = scene->d_patterns;
    F.n_tris = P.n_tris, F.max_reflection_depth = rc_.max_reflection_depth;
    // register budget of three waves per SIMD (168 VGPRs; ~165 of the kernel's binary64 temporaries then live in scratch, at points that
    // run once per ray): measured 13.3 / 14.0 / 19.2 ms for 3 / 2 / 1 waves on the mirror scene at 1080p, 2.7 / 2.9 / 4.2 ms on the teapot
    // forced through this kernel.  RL_RTC_FULL_REGS=256|512 selects the other budgets (A/B).
#ifdef RL_EXPERIMENTAL
    if (g_sw.rtc_full_regs == 512) { blocks = grid_for((const void *)rtc_full_kernel<NT, 512>, 0); hipLaunchKernelGGL((rtc_full_kernel<NT, 512>), dim3(blocks), dim3(NT), 0, stream, F); }
    else if (g_sw.rtc_full_regs == 256) { blocks = grid_for((const void *)rtc_full_kernel<NT, 256>, 0); hipLaunchKernelGGL((rtc_full_kernel<NT, 256>), dim3(blocks), dim3(NT), 0, stream, F); }
    else
#endif
    {
      blocks = grid_for((const void *)rtc_full_kernel<NT, 768>, 0);
      hipLaunchKernelGGL((rtc_full_kernel<NT, 768>), dim3(blocks), dim3(NT), 0, stream, F);
    }
  // (241 VGPRs -> two waves per SIMD.  Measured in round 3 with the register budgets of three / four waves (68 / 128 spilled VGPRs): AA 8 19.5 -> 21.1 /
  // 22.5 ms, AA 1 0.547 -> 0.518 / 0.517 ms: the binary64 Moeller-Trumbore + Phong temporaries in scratch cost more than the extra waves hide.)
#ifdef RL_EXPERIMENTAL
  } else if (lds_scene && g_sw.rtc_regs == 768) {
    blocks = grid_for((const void *)rtc_kernel<NT, true, 768>, lds);
    hipLaunchKernelGGL((rtc_kernel<NT, true, 768>), dim3(blocks), dim3(NT), lds, stream, P);
  } else if (lds_scene && g_sw.rtc_regs == 1024) {
    blocks = grid_for((const void *)rtc_kernel<NT, true, 1024>, lds);
    hipLaunchKernelGGL((rtc_kernel<NT, true, 1024>), dim3(blocks), dim3(NT), lds, stream, P);
#endif
  } else if (lds_scene) {
    blocks = grid_for((const void *)rtc_kernel<NT, true>, lds);
    hipLaunchKernelGGL((rtc_kernel<NT, true>), dim3(blocks), dim3(NT), lds, stream, P);
  } else {
    blocks = grid_for((const void *)rtc_kernel<NT, false>, 0);
    hipLaunchKernelGGL((rtc_kernel<NT, false>), dim3(blocks), dim3(NT), 0, stream, P);
  }
  HIP_TRY(hipGetLastError());
  if (want_stats) HIP_TRY(hipEventRecord(scene->ev1, stream));
  return mark_render_end(scene, stream);
}
}  // namespace rl

extern "C" {

int rl_rtc_render_device(const rl_scene *scene, const rl_rtc_camera *cam, uint32_t aa, uint32_t row_first, uint32_t row_step, void *d_out,
                         void *hip_stream, rl_stats *st) {
  if (!g_ready) return set_err(RL_E_NO_DEVICE, "rl_init has not succeeded");
  if (!scene || scene->kind != 2 || !cam || !d_out || row_step == 0 || aa == 0) return set_err(RL_E_INVALID, "bad argument");
  if (cam->hsize == 0 || cam->vsize == 0) return set_err(RL_E_INVALID, "empty image");
  hipStream_t stream = (hipStream_t)hip_stream;
  if (row_first >= cam->vsize) {
    if (st) std::memset(st, 0, sizeof *st);
    return RL_OK;
  }
  std::lock_guard<std::mutex> lk(scene->mu);  // concurrent renders of one scene: see rl_scene::mu
  int rc = rl::rtc_render_launch(scene, cam, aa, row_first, row_step, d_out, stream, st != nullptr);
  if (rc != RL_OK) return rc;
  return st ? rl::collect_stats(scene, stream, st) : post_status(scene, stream);
}

int rl_rtc_render_rows(const rl_scene *scene, const rl_rtc_camera *cam, uint32_t aa, uint32_t row_first, uint32_t row_step, double *out,
                       rl_stats *st) {
  if (!g_ready) return set_err(RL_E_NO_DEVICE, "rl_init has not succeeded");
  if (!scene || !cam || !out || row_step == 0) return set_err(RL_E_INVALID, "bad argument");
  uint32_t H = cam->vsize, W = cam->hsize;
  uint32_t nrows = row_first < H ? (H - row_first + row_step - 1) / row_step : 0;
  size_t bytes = (size_t)nrows * W * 3 * sizeof(double);
  if (bytes == 0) {
    if (st) std::memset(st, 0, sizeof *st);
    return RL_OK;
  }
  int rc0 = rl::use_context(scene->ctx);
  if (rc0 != RL_OK) return rc0;
  double *d_out = nullptr;
  HIP_TRY(hipMalloc((void **)&d_out, bytes));
  rl_stats local;
  int rc = rl_rtc_render_device(scene, cam, aa, row_first, row_step, d_out, g_ctx[(size_t)scene->ctx].stream, &local);
  if (rc == RL_OK || rc == RL_E_DEGENERATE) {
    hipError_t e = hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = set_err(RL_E_DEVICE, std::string("hipMemcpy D2H: ") + hipGetErrorString(e));
  }
  hipFree(d_out);
  if (st) *st = local;
  return rc;
}

int rl_rtc_encode_rgb8_device(const void *d_rgb, uint64_t n_pixels, void *d_rgb8, void *hip_stream) {
  if (!g_ready) return set_err(RL_E_NO_DEVICE, "rl_init has not succeeded");
  if (!d_rgb || !d_rgb8) return set_err(RL_E_INVALID, "bad argument");
  unsigned long long n = n_pixels * 3ull;
  if (n == 0) return RL_OK;
  hipLaunchKernelGGL(encode_rtc_rgb8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, (const double *)d_rgb, n, (unsigned char *)d_rgb8);
  HIP_TRY(hipGetLastError());
  return RL_OK;
}

int rl_rtc_render_rgb8(const rl_scene *scene, const rl_rtc_camera *cam, uint32_t aa, uint8_t *out, rl_stats *st) {
  if (!g_ready) return set_err(RL_E_NO_DEVICE, "rl_init has not succeeded");
  if (!scene || !cam || !out) return set_err(RL_E_INVALID, "bad argument");
  size_t npix = (size_t)cam->hsize * cam->vsize;
  if (npix == 0) return set_err(RL_E_INVALID, "empty image");
  int rc0 = rl::use_context(scene->ctx);
  if (rc0 != RL_OK) return rc0;
  hipStream_t stream = g_ctx[(size_t)scene->ctx].stream;
  double *d_rgb = nullptr;
  unsigned char *d_u8 = nullptr;
  HIP_TRY(hipMalloc((void **)&d_rgb, npix * 3 * sizeof(double)));
  hipError_t e = hipMalloc((void **)&d_u8, npix * 3);
  if (e != hipSuccess) {
    hipFree(d_rgb);
    return set_err(RL_E_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
  }
  rl_stats local;
  int rc = rl_rtc_render_device(scene, cam, aa, 0, 1, d_rgb, stream, &local);
  if (rc == RL_OK || rc == RL_E_DEGENERATE) {
    int rc2 = rl_rtc_encode_rgb8_device(d_rgb, npix, d_u8, stream);
    if (rc2 != RL_OK) rc = rc2;
    else {
      e = hipMemcpyAsync(out, d_u8, npix * 3, hipMemcpyDeviceToHost, stream);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      if (e != hipSuccess) rc = set_err(RL_E_DEVICE, std::string("D2H: ") + hipGetErrorString(e));
    }
  }
  hipFree(d_rgb), hipFree(d_u8);
  if (st) *st = local;
  return rc;
}

int rl_rtc_render(const rl_scene *scene, const rl_rtc_camera *cam, uint32_t aa, double *out, rl_stats *st) {
  return rl_rtc_render_rows(scene, cam, aa, 0, 1, out, st);
}

}  // extern "C"
