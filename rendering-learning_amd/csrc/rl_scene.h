// Internal (not part of the ABI): the library's per-device context and the rl_scene object, shared by rl_render.hip (the
// single-device entry points) and rl_multi.hip (one process driving several GPUs).
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "rl_program.h"

namespace rl {

// One per GPU the library drives.  rl_init(device) creates exactly one; rl_init_multi(n) one per device 0 .. n-1.
struct DevCtx {
  int device = -1;        // HIP device ordinal
  hipStream_t stream = nullptr;  // library-owned stream (host-buffer entry points, multi-GPU renders)
  hipEvent_t ev = nullptr;       // "this device's shard has arrived" (peer-copy gather)
};

// What the host side of a scene compiles to; immutable, shared by the per-device replicas of one scene.
struct HostRtiow {
  RtiowProgram rt;
  std::vector<DevOp> lops;  // linked ops (sphere-only scenes)
  std::vector<DevMaterial> sphere_flat;
  std::vector<CompactOp> cops;  // guarded compact ops
  std::vector<uint32_t> movbits;
  uint32_t entry0 = 0, centry0 = 0;
  std::vector<FastNode> fast_nodes;  // fast traversal structure (rl_fast_bvh.cpp); fast_root == FAST_NONE: the scene does not qualify
  uint32_t fast_root = FAST_NONE;
  std::vector<float> fast_leaf_boxes;  // [n_spheres][8]: each sphere's padded leaf box of the fast tree (rl_rtiow_coop.h)
  FastGeneral fg;  // fast traversal structure of a general scene (fg.ok == false: the scene does not qualify)
  // the guard boxes' padding is rigorous for ray origins within guard_reach of guard_center (rl_render.hip link_ops)
  double guard_center[3] = {0, 0, 0}, guard_reach = 0;
};
struct HostRtc {
  RtcProgram rc;
  std::vector<RtcGuard> guards;
};

}  // namespace rl

struct rl_scene {
  int kind = 0;  // 1 = RTIOW, 2 = RTC
  int ctx = 0;   // index of the device context this replica lives on
  int device = -1;  // ... and that context's HIP device ordinal at creation (a later rl_init may point context 0 elsewhere)
  std::shared_ptr<const rl::HostRtiow> hrt;
  std::shared_ptr<const rl::HostRtc> hrc;
  const rl::RtiowProgram &rt() const { return hrt->rt; }
  const rl::RtcProgram &rc() const { return hrc->rc; }
  std::vector<rl_scene *> replicas;  // multi-GPU: replicas[g] lives on device context g; replicas[0] == this (empty: single device)
  // RTIOW
  rl::DevOp *d_ops = nullptr;
  rl::DevOp *d_lops = nullptr;
  rl::DevMaterial *d_sphere_flat = nullptr;
  rl::CompactOp *d_cops = nullptr;
  uint32_t *d_movbits = nullptr;
  rl::FastNode *d_fast_nodes = nullptr;
  float *d_fast_leaf_boxes = nullptr;
  uint32_t *d_coop_pixels = nullptr;  // cooperative kernel: pixel list (scratch, grown on demand)
  size_t coop_pixels_cap = 0;
  uint32_t *d_steal_state = nullptr, *d_steal_n = nullptr;  // work stealing on small shards (RtiowParams::steal_state)
  size_t steal_pix = 0;
  rl::FastNodeQ *d_fg_nodes = nullptr;
  rl::FastNodeO *d_fg_onodes = nullptr;
  uint32_t *d_fg_seg_roots = nullptr;
  rl::FastMedium *d_fg_media = nullptr;
  rl::FastItem *d_fg_items = nullptr;
  rl::DevSphere *d_fg_spheres = nullptr;
  uint32_t *d_fg_material = nullptr;
  rl::DevSphere *d_spheres = nullptr;
  uint32_t *d_sphere_material = nullptr;
  rl::DevPlanar *d_planars = nullptr;
  rl_translate *d_translates = nullptr;
  rl_transform *d_transforms = nullptr;
  rl::DevMaterial *d_materials = nullptr;
  rl::DevTexture *d_textures = nullptr;
  rl::DevImage *d_images = nullptr;
  float *d_image_pool = nullptr;
  rl_perlin *d_perlins = nullptr;
  rl_medium *d_media = nullptr;
  // RTC
  rl::DevTri *d_tris = nullptr;
  rl_rtc_transformed *d_xforms = nullptr;
  rl_rtc_material *d_rmaterials = nullptr;
  rl_rtc_light *d_lights = nullptr;
  rl_rtc_shape *d_shapes = nullptr;
  rl_rtc_csg *d_csgs = nullptr;
  rl_rtc_pattern *d_patterns = nullptr;
  rl::RtcGuard *d_guards = nullptr;
  // per-scene scratch: [0] work counter (u32), [64..] 8 x u64 stats, [128..] scheduler debug counters
  unsigned char *d_scratch = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // Renders of one scene may be issued from several host threads and on several streams at once (the reference's Camera::render takes
  // &self, camera.rs:122).  The scene owns ONE set of work buffers, so: `mu` serialises the host side (enqueueing a render, handing its
  // status over), and a render enqueued on another stream than its predecessor first waits, on the device, for the predecessor's last
  // kernel (`ev_last`) — concurrent callers are safe, their frames are rendered one after the other (each one fills the GPU anyway).
  mutable std::mutex mu;
  hipStream_t last_stream = nullptr;
  hipEvent_t ev_last = nullptr;
  bool has_last = false;
  // status of the asynchronous renders (opt_stats == NULL) not yet collected by rl_render_status: a ring of N_STATUS slots of 8 stats
  // words in pinned host memory, each behind the event that follows its copy; a slot that comes round again while still pending is
  // waited for and folded into `folded_*`, so no panic-site count is ever lost
  static constexpr int N_STATUS = 8;
  unsigned long long *h_status = nullptr;  // [N_STATUS][8]
  hipEvent_t ev_status[N_STATUS] = {};
  bool status_pending[N_STATUS] = {};
  unsigned long long status_seq[N_STATUS] = {};
  int status_next = 0;
  unsigned long long next_seq = 0, folded_seq = 0, folded_rays = 0, folded_flagged = 0, folded_slow = 0;
  // cost-sorted (LPT) two-phase render: per-pixel ChaCha word positions, per-tile cost and order
  uint32_t *d_pos = nullptr, *d_tile_cost = nullptr, *d_tile_order = nullptr, *d_tile_keys = nullptr, *d_tile_iota = nullptr;
  void *d_sort_temp = nullptr;
  size_t sort_temp_bytes = 0;
  size_t lpt_pix = 0, lpt_tiles = 0;
  // multi-GPU: this replica's row shard / on replica 0 the gather buffer [G][max_rows][W][3]
  double *d_shard = nullptr;
  size_t shard_bytes = 0;
  hipEvent_t ev_gather_read = nullptr;  // replica 0: recorded behind the de-interleave kernel that reads the gather slots
  bool ev_gather_read_valid = false;
  // wavefront form (rl_rtiow_wfg.h): per-pixel records, ray records, the two queues, control words, the polled word in pinned memory
  void *d_wfg_pix = nullptr, *d_wfg_ray = nullptr;
  uint32_t *d_wfg_q0 = nullptr, *d_wfg_q1 = nullptr, *d_wfg_qs = nullptr, *d_wfg_ctl = nullptr;
  size_t wfg_pix_cap = 0, wfg_slot_cap = 0;
  void *h_wfg = nullptr;
  // rl_rtiow_render_progress (opt-in: its first call switches it on for the renders that follow): the kernels' work counters then live
  // in two words of pinned HOST memory the device reaches over PCIe — [0] the first (or only) launch of a render, [1] the cost-sorted
  // resume launch — so that the host reads them with plain loads while the kernels run; `progress_total` = slots of the render enqueued last
  bool progress_on = false;
  unsigned long long progress_total = 0;
  uint32_t *h_progress = nullptr;   // host address
  uint32_t *d_progress = nullptr;   // the same words as the device sees them
  void *d_params = nullptr;  // device copies (two slots) of the parameter block for the kernels that take it by pointer
  unsigned params_slot = 0;
  uint32_t *d_pix_rays = nullptr;  // debug (tools/): per-pixel ray counts of the last counting render
  void *exp = nullptr;             // experimental kernels' work buffers (rl_render.hip, RL_EXPERIMENTAL builds only)
};

namespace rl {

int set_err_public(int code, const std::string &m);
bool lib_ready();
int n_contexts();
DevCtx &context(int i);
int use_context(int i);  // hipSetDevice(context(i).device)
void drop_multi_state();  // rl_multi.hip: RCCL communicators + the emulation flag, dropped whenever the context list is rebuilt

// Launch-only halves of the render entry points (rl_render.hip): enqueue everything on `stream`, never synchronise.  With
// want_stats the caller finishes with collect_stats (which synchronises the stream).
int rtiow_render_launch(const rl_scene *scene, const rl_rtiow_camera *cam, uint64_t first_sample, uint32_t row_first, uint32_t row_step, void *d_out,
                        hipStream_t stream, bool want_stats);
int rtc_render_launch(const rl_scene *scene, const rl_rtc_camera *cam, uint32_t aa, uint32_t row_first, uint32_t row_step, void *d_out, hipStream_t stream,
                      bool want_stats);
int collect_stats(const rl_scene *scene, hipStream_t stream, rl_stats *st);  // RL_OK / RL_E_DEGENERATE / RL_E_DEVICE
int post_status(const rl_scene *scene, hipStream_t stream);                  // asynchronous renders: next slot of the status ring
int order_after_previous(const rl_scene *scene, hipStream_t stream);         // start of a render: device-side wait for the scene's previous render
int mark_render_end(const rl_scene *scene, hipStream_t stream);              // end of a render's launch chain
void add_stats(rl_stats *acc, const rl_stats &s);                             // sums counters, max of kernel_ms

}  // namespace rl
