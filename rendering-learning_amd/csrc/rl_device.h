// Device-side math shared by the RTIOW and RTC kernels.  Everything that feeds a comparison is
// written in the reference's operation order and this file MUST be compiled with
// -ffp-contract=off (no FMA contraction), IEEE division and sqrt (hipcc defaults for f64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rl_program.h"

namespace rl {

struct D3 {
  double x, y, z;
};
__device__ __forceinline__ D3 d3(double x, double y, double z) { return D3{x, y, z}; }
__device__ __forceinline__ D3 ld3(const double *p) { return D3{p[0], p[1], p[2]}; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator*(D3 a, D3 b) { return D3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ D3 operator*(D3 a, double s) { return D3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ D3 operator-(D3 a) { return D3{-a.x, -a.y, -a.z}; }
// vec3.rs:36,44: (x*x + y*y) + z*z
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double len2(D3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ D3 cross(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// vec3.rs:177-179: v / s == v * (1.0 / s)
__device__ __forceinline__ D3 div_s(D3 a, double s) { return a * (1.0 / s); }
__device__ __forceinline__ D3 normalize(D3 a) { return div_s(a, sqrt(len2(a))); }  // vec3.rs:56
// float-cmp approx_eq with F64Margin::zero().epsilon(e): a == b || |a - b| <= e
__device__ __forceinline__ bool approx_eq_eps(double a, double b, double e) { return a == b || fabs(a - b) <= e; }

// wave-level sum of a u64 (64-wide wavefront), result valid in lane 0
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// One claim of `n`-per-active-lane work items from a global counter with ONE atomic per wave.
__device__ __forceinline__ uint32_t wave_claim(uint32_t *counter) {
  unsigned long long mask = __ballot(1);
  uint32_t lane = __lane_id();
  uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
  uint32_t base = 0;
  if (rank == 0) base = atomicAdd(counter, (uint32_t)__popcll(mask));
  base = __builtin_amdgcn_readfirstlane(base);
  return base + rank;
}

}  // namespace rl
