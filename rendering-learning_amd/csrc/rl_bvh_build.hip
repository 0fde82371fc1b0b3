// Bvh::new (ray-tracing-one-weekend/src/hittable/bvh.rs:22-60) on the device — SURVEY.md §8f row 4.
//
// The reference builds top-down: node box = merge of the boxes of its hittables (aabb.rs:135, plain min / max), split
// axis = the longest one (bvh.rs:63-77), sort by `bounding_box().axis.min` with f64::total_cmp, split at len / 2,
// leaves hold 1-2 hittables.  Every level of that recursion is independent across nodes, so it runs level-synchronous
// here: one segmented reduction (box + axis per open node), one key pass, one STABLE segmented radix sort
// (hipCUB / rocPRIM, keys = the total_cmp order of the doubles as u64) per level, ~log2(n) levels.  Stable matches the
// host mirror (rtiow_host.hpp Bvh, std::stable_sort); the reference's sort_unstable_by leaves the order of EQUAL keys
// implementation-defined.  Output: rl_bvh_node records in the order the recursion (and Bvh::flatten) visits them.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rl_render.h"

namespace rl {
int set_err_public(int code, const std::string &m);  // rl_render.hip
}

namespace {

#define BVH_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) {                                                                             \
      rc = rl::set_err_public(RL_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));          \
      goto done;                                                                                        \
    }                                                                                                   \
  } while (0)

__device__ __forceinline__ unsigned long long total_order_key(double v) {  // ascending u64 == f64::total_cmp
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

struct Seg {
  uint32_t start, len;
};

// one workgroup per open node: box of its primitives (min / max per axis), the split axis, and the node's box record
template <int NT>
__global__ void __launch_bounds__(NT) seg_box_axis(const double *boxes, const uint32_t *idx, const Seg *segs, double *seg_box /*[nseg][6]*/, uint32_t *seg_axis) {
  const Seg s = segs[blockIdx.x];
  const double INF = __longlong_as_double(0x7FF0000000000000ll);
  double mn[3] = {INF, INF, INF}, mx[3] = {-INF, -INF, -INF};
  for (uint32_t i = threadIdx.x; i < s.len; i += NT) {
    const double *b = boxes + (size_t)idx[s.start + i] * 6;
#pragma unroll
    for (int k = 0; k < 3; k++) mn[k] = fmin(mn[k], b[2 * k]), mx[k] = fmax(mx[k], b[2 * k + 1]);
  }
  __shared__ double sh[6][NT];
#pragma unroll
  for (int k = 0; k < 3; k++) sh[2 * k][threadIdx.x] = mn[k], sh[2 * k + 1][threadIdx.x] = mx[k];
  __syncthreads();
  for (int off = NT / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        sh[2 * k][threadIdx.x] = fmin(sh[2 * k][threadIdx.x], sh[2 * k][threadIdx.x + off]);
        sh[2 * k + 1][threadIdx.x] = fmax(sh[2 * k + 1][threadIdx.x], sh[2 * k + 1][threadIdx.x + off]);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double sx = sh[1][0] - sh[0][0], sy = sh[3][0] - sh[2][0], sz = sh[5][0] - sh[4][0];
    uint32_t axis = sx > sy ? (sx > sz ? 0u : 2u) : (sy > sz ? 1u : 2u);  // bvh.rs:63-77 find_longest_axis
    seg_axis[blockIdx.x] = axis;
    for (int k = 0; k < 6; k++) seg_box[(size_t)blockIdx.x * 6 + k] = sh[k][0];
  }
}

// sort key of every primitive inside an open node: its box minimum on the node's axis, in total_cmp order
template <int NT>
__global__ void __launch_bounds__(NT) seg_keys(const double *boxes, const uint32_t *idx, const Seg *segs, const uint32_t *seg_axis, unsigned long long *keys) {
  const Seg s = segs[blockIdx.x];
  const uint32_t axis = seg_axis[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < s.len; i += NT) keys[s.start + i] = total_order_key(boxes[(size_t)idx[s.start + i] * 6 + 2 * axis]);
}

uint32_t count_nodes(uint32_t n, std::unordered_map<uint32_t, uint32_t> &memo) {  // nodes of the subtree over n hittables
  if (n <= 2) return 1;
  auto it = memo.find(n);
  if (it != memo.end()) return it->second;
  uint32_t r = 1 + count_nodes(n / 2, memo) + count_nodes(n - n / 2, memo);
  memo[n] = r;
  return r;
}

}  // namespace

namespace rl {
// Cost-sorted launch order of rl_rtiow_render_device, on the device: order = tile ids sorted by cost, most expensive first,
// ties in ascending tile id (what std::stable_sort(order, cost[a] > cost[b]) gives).  keys_tmp / order_in: n u32 each;
// *temp / *temp_bytes: scratch owned by the caller, grown on demand.  Asynchronous on `stream`.
__global__ void iota_u32(uint32_t *p, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}
int sort_tiles_by_cost_desc(const uint32_t *d_cost, uint32_t *d_keys_tmp, uint32_t *d_order_in, uint32_t *d_order_out, uint32_t n, void **temp, size_t *temp_bytes,
                            hipStream_t stream) {
  hipLaunchKernelGGL(iota_u32, dim3((n + 255) / 256), dim3(256), 0, stream, d_order_in, n);
  size_t need = 0;
  hipError_t e = hipcub::DeviceRadixSort::SortPairsDescending(nullptr, need, d_cost, d_keys_tmp, d_order_in, d_order_out, (int)n, 0, 32, stream);
  if (e != hipSuccess) return set_err_public(RL_E_DEVICE, std::string("tile sort (size query): ") + hipGetErrorString(e));
  if (need > *temp_bytes) {
    if (*temp) {
      hipStreamSynchronize(stream);  // an earlier sort on this stream may still use the old buffer
      hipFree(*temp);
    }
    *temp = nullptr, *temp_bytes = 0;
    e = hipMalloc(temp, need);
    if (e != hipSuccess) return set_err_public(RL_E_DEVICE, std::string("tile sort (temp storage): ") + hipGetErrorString(e));
    *temp_bytes = need;
  }
  need = *temp_bytes;
  e = hipcub::DeviceRadixSort::SortPairsDescending(*temp, need, d_cost, d_keys_tmp, d_order_in, d_order_out, (int)n, 0, 32, stream);
  if (e != hipSuccess) return set_err_public(RL_E_DEVICE, std::string("tile sort: ") + hipGetErrorString(e));
  return RL_OK;
}
}  // namespace rl

extern "C" int rl_bvh_build(const double *prim_boxes, const rl_href *prims, uint32_t n, uint32_t node_base, rl_bvh_node *out_nodes, uint32_t cap,
                            uint32_t *out_n_nodes) {
  if (!prim_boxes || !prims || !out_nodes || n == 0) return rl::set_err_public(RL_E_INVALID, "rl_bvh_build: bad argument (Bvh::new needs at least one hittable)");
  for (size_t i = 0; i < (size_t)n * 6; i++)
    if (prim_boxes[i] != prim_boxes[i]) return rl::set_err_public(RL_E_INVALID, "rl_bvh_build: NaN in a bounding box");
  std::unordered_map<uint32_t, uint32_t> memo;
  const uint32_t total_nodes = count_nodes(n, memo);
  if (out_n_nodes) *out_n_nodes = total_nodes;
  if (cap < total_nodes) return rl::set_err_public(RL_E_INVALID, "rl_bvh_build: output capacity too small");

  int rc = RL_OK;
  double *d_boxes = nullptr, *d_seg_box = nullptr;
  uint32_t *d_idx[2] = {nullptr, nullptr}, *d_seg_axis = nullptr;
  unsigned long long *d_keys[2] = {nullptr, nullptr};
  Seg *d_segs = nullptr;
  int *d_beg = nullptr, *d_end = nullptr;
  void *d_temp = nullptr;
  size_t temp_bytes = 0;
  struct Open {
    uint32_t start, len, node;  // node = index in the output (recursion order)
  };
  std::vector<Open> open, next;
  std::vector<uint32_t> order(n);
  std::vector<Seg> segs;
  std::vector<int> beg, end;
  std::vector<double> seg_box;
  std::vector<std::pair<uint32_t, uint32_t>> box_of_level;  // (node, slot) pairs of the current level
  const size_t max_segs = (size_t)n / 3 + 1;                // open nodes hold >= 3 hittables each
  int cur = 0;

  auto leaf = [&](const Open &o) {  // bvh.rs:26-43: 1 or 2 hittables, box = (merge of) their boxes
    rl_bvh_node nd{};
    const double INF = std::numeric_limits<double>::infinity();
    double b[6] = {INF, -INF, INF, -INF, INF, -INF};
    for (uint32_t i = 0; i < o.len; i++) {
      const double *pb = prim_boxes + (size_t)order[o.start + i] * 6;
      for (int k = 0; k < 3; k++) b[2 * k] = std::fmin(b[2 * k], pb[2 * k]), b[2 * k + 1] = std::fmax(b[2 * k + 1], pb[2 * k + 1]);
      nd.child[i] = prims[order[o.start + i]];
    }
    std::memcpy(nd.bbox, b, sizeof b);
    nd.n_children = o.len;
    out_nodes[o.node] = nd;
  };

  BVH_TRY(hipMalloc((void **)&d_boxes, (size_t)n * 6 * sizeof(double)));
  BVH_TRY(hipMemcpy(d_boxes, prim_boxes, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice));
  for (int k = 0; k < 2; k++) {
    BVH_TRY(hipMalloc((void **)&d_idx[k], (size_t)n * sizeof(uint32_t)));
    BVH_TRY(hipMalloc((void **)&d_keys[k], (size_t)n * sizeof(unsigned long long)));
  }
  BVH_TRY(hipMalloc((void **)&d_segs, max_segs * sizeof(Seg)));
  BVH_TRY(hipMalloc((void **)&d_beg, max_segs * sizeof(int)));
  BVH_TRY(hipMalloc((void **)&d_end, max_segs * sizeof(int)));
  BVH_TRY(hipMalloc((void **)&d_seg_axis, max_segs * sizeof(uint32_t)));
  BVH_TRY(hipMalloc((void **)&d_seg_box, max_segs * 6 * sizeof(double)));
  for (uint32_t i = 0; i < n; i++) order[i] = i;
  BVH_TRY(hipMemcpy(d_idx[0], order.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));

  if (n <= 2) {
    leaf(Open{0, n, 0});
    goto done;
  }
  open.push_back(Open{0, n, 0});
  while (!open.empty()) {
    const size_t ns = open.size();
    segs.resize(ns), beg.resize(ns), end.resize(ns);
    for (size_t i = 0; i < ns; i++) segs[i] = Seg{open[i].start, open[i].len}, beg[i] = (int)open[i].start, end[i] = (int)(open[i].start + open[i].len);
    BVH_TRY(hipMemcpy(d_segs, segs.data(), ns * sizeof(Seg), hipMemcpyHostToDevice));
    BVH_TRY(hipMemcpy(d_beg, beg.data(), ns * sizeof(int), hipMemcpyHostToDevice));
    BVH_TRY(hipMemcpy(d_end, end.data(), ns * sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL((seg_box_axis<256>), dim3((uint32_t)ns), dim3(256), 0, 0, d_boxes, d_idx[cur], d_segs, d_seg_box, d_seg_axis);
    hipLaunchKernelGGL((seg_keys<256>), dim3((uint32_t)ns), dim3(256), 0, 0, d_boxes, d_idx[cur], d_segs, d_seg_axis, d_keys[0]);
    BVH_TRY(hipGetLastError());
    // hittables of closed nodes keep their place
    BVH_TRY(hipMemcpyAsync(d_idx[cur ^ 1], d_idx[cur], (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, 0));
    {
      size_t need = 0;
      BVH_TRY(hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, need, d_keys[0], d_keys[1], d_idx[cur], d_idx[cur ^ 1], (int)n, (int)ns, d_beg, d_end, 0, 64, 0));
      if (need > temp_bytes) {
        hipFree(d_temp);
        d_temp = nullptr;
        BVH_TRY(hipMalloc(&d_temp, need));
        temp_bytes = need;
      }
      BVH_TRY(hipcub::DeviceSegmentedRadixSort::SortPairs(d_temp, need, d_keys[0], d_keys[1], d_idx[cur], d_idx[cur ^ 1], (int)n, (int)ns, d_beg, d_end, 0, 64, 0));
    }
    cur ^= 1;
    seg_box.resize(ns * 6);
    BVH_TRY(hipMemcpy(seg_box.data(), d_seg_box, ns * 6 * sizeof(double), hipMemcpyDeviceToHost));
    next.clear();
    for (size_t i = 0; i < ns; i++) {
      const Open &o = open[i];
      const uint32_t mid = o.len / 2, left_nodes = count_nodes(mid, memo);
      rl_bvh_node nd{};
      std::memcpy(nd.bbox, &seg_box[i * 6], 6 * sizeof(double));
      nd.n_children = 2;
      nd.child[0] = rl_href{RL_H_BVH, node_base + o.node + 1};
      nd.child[1] = rl_href{RL_H_BVH, node_base + o.node + 1 + left_nodes};
      out_nodes[o.node] = nd;
      next.push_back(Open{o.start, mid, o.node + 1});
      next.push_back(Open{o.start + mid, o.len - mid, o.node + 1 + left_nodes});
    }
    // children with <= 2 hittables are leaves; their order is final once this level's sort is back on the host
    bool any_leaf = false;
    for (const Open &o : next) any_leaf |= o.len <= 2;
    if (any_leaf) BVH_TRY(hipMemcpy(order.data(), d_idx[cur], (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    open.clear();
    for (const Open &o : next) {
      if (o.len <= 2) leaf(o);
      else open.push_back(o);
    }
  }
done:
  hipFree(d_boxes), hipFree(d_seg_box), hipFree(d_idx[0]), hipFree(d_idx[1]), hipFree(d_seg_axis), hipFree(d_keys[0]), hipFree(d_keys[1]);
  hipFree(d_segs), hipFree(d_beg), hipFree(d_end), hipFree(d_temp);
  return rc;
}
