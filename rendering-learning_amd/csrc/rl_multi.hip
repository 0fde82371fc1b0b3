// rl_multi.hip — one process, several GPUs (SURVEY.md §8b "the library owns streams, device buffers and RCCL communicators", §8e).
//
// The reference calls ONE Camera::render(&world) from one thread (ray-tracing-one-weekend/src/camera.rs:122,
// examples/common/mod.rs:16; ray-tracer-challenge/src/scene/camera.rs:93), so a drop-in host reaches G GPUs through one call:
//   rl_init_multi(G)                one device context per GPU, RCCL communicators (ncclCommInitAll, one rank per GPU)
//   rl_*_scene_create               replicates the scene on every context (<= 215 MB even for BASELINE configs[4])
//   rl_*_render_multi[_device]      image row r -> GPU r mod G (interleaving balances sky against ground); every GPU renders its
//                                   rows with the single-device kernels on its own stream — no collective during the render —
//                                   then ONE exchange: ncclSend / ncclRecv of ceil(H/G)*W*3 f64 per peer to GPU 0 in one group
//                                   (each peer on its own xGMI link), and a de-interleave kernel on GPU 0.
// Pixels depend on (seed, x, y, sample) only, so the frame is bit-identical for every G (tests/test_gpu_multi.py).
// RCCL is dlopen'ed (librccl.so.1) when the first multi-GPU context is built: a single-GPU host needs no RCCL at all.  Without it
// (or with RL_MULTI_GATHER=peer) the exchange is hipMemcpyPeerAsync over the same links.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "rl_scene.h"

using namespace rl;

namespace rl {
int set_contexts(const std::vector<int> &devices);  // rl_render.hip
}

namespace {

#define HIP_TRY(expr)                                                                                     \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) return set_err_public(RL_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::vector<ncclComm_t> comms;  // one per device context
} g_rccl;
// the pointer types above are the installed header's own: a signature drift in <rccl/rccl.h> fails the build, not the first multi-GPU run
static_assert(std::is_same<decltype(Rccl::CommInitAll), decltype(&ncclCommInitAll)>::value && std::is_same<decltype(Rccl::CommDestroy), decltype(&ncclCommDestroy)>::value &&
                  std::is_same<decltype(Rccl::GroupStart), decltype(&ncclGroupStart)>::value && std::is_same<decltype(Rccl::GroupEnd), decltype(&ncclGroupEnd)>::value &&
                  std::is_same<decltype(Rccl::Send), decltype(&ncclSend)>::value && std::is_same<decltype(Rccl::Recv), decltype(&ncclRecv)>::value &&
                  std::is_same<decltype(Rccl::GetErrorString), decltype(&ncclGetErrorString)>::value,
              "rl_multi.hip's RCCL entry points no longer match <rccl/rccl.h>");
bool g_emulated = false;  // every context on ONE physical GPU (tests on a one-GPU box): peer-copy exchange, no communicators

bool load_rccl() {
  if (g_rccl.lib) return true;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) return false;
  Rccl r;
  r.lib = h;
  r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
  r.Send = (decltype(r.Send))dlsym(h, "ncclSend");
  r.Recv = (decltype(r.Recv))dlsym(h, "ncclRecv");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv || !r.GetErrorString) {
    dlclose(h);
    return false;
  }
  g_rccl = r;
  return true;
}

void drop_comms() {
  for (ncclComm_t c : g_rccl.comms)
    if (c && g_rccl.CommDestroy) g_rccl.CommDestroy(c);
  g_rccl.comms.clear();
}
}  // namespace
namespace rl {
void drop_multi_state() {  // rl_init / rl_shutdown: communicators never outlive the device contexts they were built on
  drop_comms();
  g_emulated = false;
}
}  // namespace rl
namespace {

// out[r][x][c] = gathered[r % G][r / G][x][c]: the rows of rank g sit compact in slot g of the gather buffer
__global__ void deinterleave_rows(const double *gathered, double *out, uint32_t H, uint32_t row_vals, uint32_t G, uint64_t slot_vals) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t total = (uint64_t)H * row_vals;
  if (i >= total) return;
  uint32_t r = (uint32_t)(i / row_vals), x = (uint32_t)(i % row_vals);
  out[i] = gathered[(uint64_t)(r % G) * slot_vals + (uint64_t)(r / G) * row_vals + x];
}

// One frame over all device contexts.  render_shard(replica, g, G, d_rows, stream, want_stats) enqueues rank g's rows.
template <class F>
int render_multi(const rl_scene *scene, uint32_t W, uint32_t H, void *d_out0, rl_stats *st, F render_shard) {
  const int G = n_contexts();
  std::lock_guard<std::mutex> lk(scene->mu);  // one multi-GPU frame of a scene at a time on the host side (rl_scene::mu)
  // a scene created under another rl_init / rl_init_multi holds buffers on device contexts that are gone (or never were):
  // refuse it rather than render on one GPU silently or touch a device context 0 no longer points at
  if (G > 1 && (int)scene->replicas.size() != G) return set_err_public(RL_E_INVALID, "scene was created before rl_init_multi: recreate it so that every device holds a replica");
  if (G == 1 && (!scene->replicas.empty() || scene->ctx != 0 || scene->device != context(0).device))
    return set_err_public(RL_E_INVALID, "scene was created under another rl_init / rl_init_multi: recreate it");
  if (G == 1) {  // one GPU: the single-device path, straight into the caller's buffer
    int rc = use_context(0);
    if (rc != RL_OK) return rc;
    rc = render_shard(scene, 0, 1, d_out0, context(0).stream, st != nullptr);
    if (rc != RL_OK) return rc;
    if (st) return collect_stats(scene, context(0).stream, st);
    return RL_OK;
  }
  const uint32_t max_rows = (H + (uint32_t)G - 1) / (uint32_t)G;
  const uint64_t row_vals = (uint64_t)W * 3, slot_vals = (uint64_t)max_rows * row_vals;
  if (row_vals >= 0xFFFFFFFFull) return set_err_public(RL_E_INVALID, "image too wide");
  rl_scene *root = scene->replicas[0];
  int rc;
  // buffers: every replica owns its shard; replica 0's holds the whole gather [G][max_rows][W][3] (slot 0 = its own rows)
  for (int g = 0; g < G; g++) {
    rl_scene *r = scene->replicas[(size_t)g];
    size_t need = (size_t)(g == 0 ? (uint64_t)G * slot_vals : slot_vals) * sizeof(double);
    if ((rc = use_context(g)) != RL_OK) return rc;
    if (r->shard_bytes < need) {
      if (r->d_shard) {
        HIP_TRY(hipStreamSynchronize(context(g).stream));
        hipFree(r->d_shard);
      }
      r->d_shard = nullptr, r->shard_bytes = 0;
      HIP_TRY(hipMalloc((void **)&r->d_shard, need));
      r->shard_bytes = need;
    }
  }
  // back-to-back asynchronous frames: the previous frame's de-interleave (stream 0) still reads the gather slots that this frame's
  // peer copies (stream g) will overwrite, and nothing else orders stream g behind it
  if (root->ev_gather_read_valid)
    for (int g = 1; g < G; g++) {
      if ((rc = use_context(g)) != RL_OK) return rc;
      HIP_TRY(hipStreamWaitEvent(context(g).stream, root->ev_gather_read, 0));
    }
  // render: rank g takes rows g, g+G, ... on its own stream; nothing below waits on the host
  for (int g = 0; g < G && (uint32_t)g < H; g++) {  // more GPUs than image rows: the surplus ones have nothing to render
    if ((rc = use_context(g)) != RL_OK) return rc;
    rc = render_shard(scene->replicas[(size_t)g], (uint32_t)g, (uint32_t)G, scene->replicas[(size_t)g]->d_shard, context(g).stream, st != nullptr);
    if (rc != RL_OK) return rc;
  }
  // the one exchange step: shards -> GPU 0
  const bool use_rccl = !g_emulated && (int)g_rccl.comms.size() == G;
  if (use_rccl) {
    ncclResult_t nr = g_rccl.GroupStart();
    for (int g = 1; g < G && nr == ncclSuccess; g++) {
      uint32_t rows = (H > (uint32_t)g) ? (H - (uint32_t)g + (uint32_t)G - 1) / (uint32_t)G : 0;
      size_t count = (size_t)rows * row_vals;
      if (count == 0) continue;
      nr = g_rccl.Send(scene->replicas[(size_t)g]->d_shard, count, ncclDouble, 0, g_rccl.comms[(size_t)g], context(g).stream);
      if (nr == ncclSuccess) nr = g_rccl.Recv(root->d_shard + (uint64_t)g * slot_vals, count, ncclDouble, g, g_rccl.comms[0], context(0).stream);
    }
    ncclResult_t ne = g_rccl.GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return set_err_public(RL_E_DEVICE, std::string("RCCL gather: ") + g_rccl.GetErrorString(nr));
  } else {
    for (int g = 1; g < G; g++) {
      uint32_t rows = (H > (uint32_t)g) ? (H - (uint32_t)g + (uint32_t)G - 1) / (uint32_t)G : 0;
      size_t bytes = (size_t)rows * row_vals * sizeof(double);
      if ((rc = use_context(g)) != RL_OK) return rc;
      if (bytes) HIP_TRY(hipMemcpyPeerAsync(root->d_shard + (uint64_t)g * slot_vals, context(0).device, scene->replicas[(size_t)g]->d_shard, context(g).device, bytes,
                                            context(g).stream));
      HIP_TRY(hipEventRecord(context(g).ev, context(g).stream));
    }
    if ((rc = use_context(0)) != RL_OK) return rc;
    for (int g = 1; g < G; g++) HIP_TRY(hipStreamWaitEvent(context(0).stream, context(g).ev, 0));
  }
  if ((rc = use_context(0)) != RL_OK) return rc;
  uint64_t total = (uint64_t)H * row_vals;
  hipLaunchKernelGGL(deinterleave_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, context(0).stream, root->d_shard, (double *)d_out0, H, (uint32_t)row_vals,
                     (uint32_t)G, slot_vals);
  HIP_TRY(hipGetLastError());
  if (!root->ev_gather_read) HIP_TRY(hipEventCreateWithFlags(&root->ev_gather_read, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(root->ev_gather_read, context(0).stream));
  root->ev_gather_read_valid = true;
  if (!st) return RL_OK;
  std::memset(st, 0, sizeof *st);
  int worst = RL_OK;
  for (int g = 0; g < G && (uint32_t)g < H; g++) {
    rl_stats one;
    if ((rc = use_context(g)) != RL_OK) return rc;
    rc = collect_stats(scene->replicas[(size_t)g], context(g).stream, &one);
    if (rc != RL_OK && rc != RL_E_DEGENERATE) return rc;
    if (rc == RL_E_DEGENERATE) worst = rc;
    add_stats(st, one);
  }
  if ((rc = use_context(0)) != RL_OK) return rc;
  HIP_TRY(hipStreamSynchronize(context(0).stream));  // the de-interleave
  return worst;
}

int post_status_multi(const rl_scene *scene, uint32_t H) {
  const int G = n_contexts();
  for (int g = 0; g < G && (uint32_t)g < H; g++) {
    rl_scene *r = scene->replicas.empty() ? const_cast<rl_scene *>(scene) : scene->replicas[(size_t)g];
    int rc = use_context(g);
    if (rc != RL_OK) return rc;
    {
      std::lock_guard<std::mutex> lk(r->mu);
      rc = post_status(r, context(g).stream);
    }
    if (rc != RL_OK) return rc;
    if (scene->replicas.empty()) break;
  }
  return use_context(0);
}

int init_contexts(const std::vector<int> &devices, bool emulated) {
  drop_comms();
  g_emulated = emulated;
  int rc = set_contexts(devices);
  if (rc != RL_OK) return rc;
  const int G = (int)devices.size();
  if (G > 1 && !emulated) {
    const char *mode = std::getenv("RL_MULTI_GATHER");
    bool want_rccl = !(mode && std::string(mode) == "peer");
    if (want_rccl && load_rccl()) {
      g_rccl.comms.assign((size_t)G, nullptr);
      ncclResult_t nr = g_rccl.CommInitAll(g_rccl.comms.data(), G, devices.data());
      if (nr != ncclSuccess) {
        g_rccl.comms.clear();
        return set_err_public(RL_E_DEVICE, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(nr));
      }
    }
    if (g_rccl.comms.empty()) {  // peer copies: let every device write into GPU 0
      for (int g = 1; g < G; g++) {
        int can = 0;
        HIP_TRY(hipDeviceCanAccessPeer(&can, devices[(size_t)g], devices[0]));
        if (can) {
          HIP_TRY(hipSetDevice(devices[(size_t)g]));
          hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return set_err_public(RL_E_DEVICE, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
          (void)hipGetLastError();
        }
      }
      HIP_TRY(hipSetDevice(devices[0]));
    }
  }
  return RL_OK;
}

}  // namespace

extern "C" {

int rl_init(int device);

int rl_init_multi(int n_devices) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return set_err_public(RL_E_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (n_devices < 0 || n_devices > n) return set_err_public(RL_E_INVALID, "rl_init_multi: more devices requested than are visible");
  if (n_devices == 0) n_devices = n;
  int rc = rl_init(0);
  if (rc != RL_OK) return rc;
  std::vector<int> devs((size_t)n_devices);
  for (int g = 0; g < n_devices; g++) devs[(size_t)g] = g;
  return init_contexts(devs, false);
}

int rl_device_count(void) { return lib_ready() ? n_contexts() : 0; }

// Not part of the ABI (tests on a one-GPU box): G device contexts that all live on the current GPU, each with its own stream,
// scene replica and shard buffer; the exchange is a device-to-device copy.  Exercises everything but the RCCL calls.
int rl_debug_init_multi_emulated(int G) {
  if (G < 1 || G > 64) return set_err_public(RL_E_INVALID, "bad emulated device count");
  int rc = rl_init(-1);
  if (rc != RL_OK) return rc;
  return init_contexts(std::vector<int>((size_t)G, context(0).device), true);
}

// Not part of the ABI (CPU-tier test, bench.py): 1 when librccl can be loaded and every entry point the gather uses resolves — no device needed.
int rl_debug_rccl_loadable(void) { return load_rccl() ? 1 : 0; }

// Not part of the ABI: 1 when the exchange goes through RCCL communicators, 0 for peer copies.
int rl_debug_multi_uses_rccl(void) { return !g_emulated && (int)g_rccl.comms.size() == n_contexts() && n_contexts() > 1; }

int rl_rtiow_render_multi_device(const rl_scene *scene, const rl_rtiow_camera *cam, uint64_t first_sample, void *d_out_rgb_sum, rl_stats *st) {
  if (!lib_ready()) return set_err_public(RL_E_NO_DEVICE, "rl_init has not succeeded");
  if (!scene || scene->kind != 1 || !cam || !d_out_rgb_sum) return set_err_public(RL_E_INVALID, "bad argument");
  if (cam->image_width == 0 || cam->image_height == 0) return set_err_public(RL_E_INVALID, "empty image");
  int rc = render_multi(scene, cam->image_width, cam->image_height, d_out_rgb_sum, st,
                        [&](const rl_scene *rep, uint32_t g, uint32_t G, void *d_rows, hipStream_t stream, bool want_stats) {
                          return rtiow_render_launch(rep, cam, first_sample, g, G, d_rows, stream, want_stats);
                        });
  if (rc != RL_OK || st) return rc;
  return post_status_multi(scene, cam->image_height);
}

int rl_rtiow_render_multi(const rl_scene *scene, const rl_rtiow_camera *cam, uint64_t first_sample, double *out_rgb_sum, rl_stats *st) {
  if (!lib_ready()) return set_err_public(RL_E_NO_DEVICE, "rl_init has not succeeded");
  if (!scene || !cam || !out_rgb_sum) return set_err_public(RL_E_INVALID, "bad argument");
  size_t bytes = (size_t)cam->image_width * cam->image_height * 3 * sizeof(double);
  if (bytes == 0) return set_err_public(RL_E_INVALID, "empty image");
  int rc = use_context(0);
  if (rc != RL_OK) return rc;
  double *d_out = nullptr;
  HIP_TRY(hipMalloc((void **)&d_out, bytes));
  rl_stats local;
  rc = rl_rtiow_render_multi_device(scene, cam, first_sample, d_out, &local);
  if (rc == RL_OK || rc == RL_E_DEGENERATE) {
    hipError_t e = hipMemcpy(out_rgb_sum, d_out, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = set_err_public(RL_E_DEVICE, std::string("hipMemcpy D2H: ") + hipGetErrorString(e));
  }
  hipFree(d_out);
  if (st) *st = local;
  return rc;
}

int rl_rtc_render_multi_device(const rl_scene *scene, const rl_rtc_camera *cam, uint32_t aa, void *d_out_rgb, rl_stats *st) {
  if (!lib_ready()) return set_err_public(RL_E_NO_DEVICE, "rl_init has not succeeded");
  if (!scene || scene->kind != 2 || !cam || !d_out_rgb || aa == 0) return set_err_public(RL_E_INVALID, "bad argument");
  if (cam->hsize == 0 || cam->vsize == 0) return set_err_public(RL_E_INVALID, "empty image");
  int rc = render_multi(scene, cam->hsize, cam->vsize, d_out_rgb, st, [&](const rl_scene *rep, uint32_t g, uint32_t G, void *d_rows, hipStream_t stream, bool want_stats) {
    return rtc_render_launch(rep, cam, aa, g, G, d_rows, stream, want_stats);
  });
  if (rc != RL_OK || st) return rc;
  return post_status_multi(scene, cam->vsize);
}

int rl_rtc_render_multi(const rl_scene *scene, const rl_rtc_camera *cam, uint32_t aa, double *out_rgb, rl_stats *st) {
  if (!lib_ready()) return set_err_public(RL_E_NO_DEVICE, "rl_init has not succeeded");
  if (!scene || !cam || !out_rgb) return set_err_public(RL_E_INVALID, "bad argument");
  size_t bytes = (size_t)cam->hsize * cam->vsize * 3 * sizeof(double);
  if (bytes == 0) return set_err_public(RL_E_INVALID, "empty image");
  int rc = use_context(0);
  if (rc != RL_OK) return rc;
  double *d_out = nullptr;
  HIP_TRY(hipMalloc((void **)&d_out, bytes));
  rl_stats local;
  rc = rl_rtc_render_multi_device(scene, cam, aa, d_out, &local);
  if (rc == RL_OK || rc == RL_E_DEGENERATE) {
    hipError_t e = hipMemcpy(out_rgb, d_out, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = set_err_public(RL_E_DEVICE, std::string("hipMemcpy D2H: ") + hipGetErrorString(e));
  }
  hipFree(d_out);
  if (st) *st = local;
  return rc;
}

}  // extern "C"
