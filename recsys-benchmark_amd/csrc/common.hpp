// common.hpp — shared helpers for the gfx950 kernels of libmi355x_recsys.so.
// Wave = 64 lanes everywhere (CDNA4); nothing here is portable to 32-wide warps.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/mi355x_recsys.h"

namespace mi {

constexpr int kWave = 64;
constexpr int kBlock = 256;            // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxGrid = 256 * 8;      // 256 CUs x 8 resident 256-thread workgroups

// ---- profiling ring (mi_prof_*) --------------------------------------------
// prof_acquire returns false when the ring is disarmed or full; otherwise the
// event pair that hipExtLaunchKernelGGL stamps with the dispatch's own begin /
// end timestamps (what rocprofv3 --kernel-trace reports), not stream markers.
bool prof_acquire(const char *name, hipEvent_t *a, hipEvent_t *b);

inline int launch_status() {
  return hipGetLastError() == hipSuccess ? MI_OK : MI_ERR_LAUNCH;
}

inline int grid_for_waves(int64_t n_wave_items) {
  int64_t g = (n_wave_items + kWavesPerBlock - 1) / kWavesPerBlock;
  if (g < 1) g = 1;
  if (g > kMaxGrid) g = kMaxGrid;
  return (int)g;
}

__host__ __device__ inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device helpers ----------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// sum over the lanes that differ only in bits >= log2(lo) (lo a power of two):
// i.e. over the "row slot" index r = lane / lo, keeping q = lane % lo apart.
template <int LO>
__device__ __forceinline__ float4 slot_sum(float4 v) {
#pragma unroll
  for (int m = LO; m < kWave; m <<= 1) {
    v.x += __shfl_xor(v.x, m);
    v.y += __shfl_xor(v.y, m);
    v.z += __shfl_xor(v.z, m);
    v.w += __shfl_xor(v.w, m);
  }
  return v;
}

__device__ __forceinline__ float dot4(float4 a, float4 b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
// non-temporal 16-B store (global_store_dwordx4 ... nt): for streams nobody in the kernel reads back
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_nt(float *p, float4 v) {
  f32x4_t x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<f32x4_t *>(p));
}

// "last workgroup sums the partials" without agent-scope fences (on gfx950 a __threadfence() is an L2 write-back +
// invalidate per workgroup: the loss kernels took 13 us with it).  ONE lane per workgroup publishes: the partial travels
// as a device-scope (sc1, write-through) store, an explicit `s_waitcnt vmcnt(0)` holds the lane until that store has
// been acknowledged by the coherence point behind the per-XCD L2s, and only then is the ticket taken (a workgroup-scope
// fence emits NO wait between the store and the atomic — checked in the .s — so the two, at different addresses, could
// be observed out of order).  The workgroup whose ticket came back last reads the partials with sc1 loads after its
// atomic has returned, behind a workgroup barrier (MI355X_MICROARCH.md, hand-offs measured with sc1 loads, row 1).
__device__ __forceinline__ void publish_partial(float *part, unsigned *ticket, float s, bool &last) {
  __hip_atomic_store(part + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
}
__device__ __forceinline__ float read_partial(const float *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace mi

// Launch `kernel` on `stream`; when the profiling ring is armed the dispatch is
// timed by its own start/stop events.
#define MI_LAUNCH(name, kernel, grid, block, stream, ...)                               \
  do {                                                                                  \
    hipEvent_t _ea, _eb;                                                                \
    if (mi::prof_acquire(name, &_ea, &_eb))                                             \
      hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (hipStream_t)(stream),  \
                            _ea, _eb, 0, __VA_ARGS__);                                  \
    else                                                                                \
      hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (hipStream_t)(stream),     \
                         __VA_ARGS__);                                                  \
  } while (0)
