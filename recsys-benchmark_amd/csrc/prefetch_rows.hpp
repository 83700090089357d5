// prefetch_rows.hpp — touch the table rows the NEXT batch's lookup will gather (one dword per row: the 128-byte line of a
// packed row comes into the Infinity Cache), as work any launch can carry in extra workgroups: its own kernel
// (gather_fm.hip, mi_prefetch_rows) or — the form a training step uses — the tail's weight-gradient launch (gemm.hip,
// mi_gemm_f32_multi_ride), which is MFMA-bound and leaves the HBM idle for ~45 us.
#pragma once
#include "common.hpp"

namespace mi {

struct PrefetchJob {
  const int64_t *idx;        // [n] raw ids of the next batch ([B, F] row-major)
  const int64_t *offsets;    // [F], nullable
  const float *W, *w1;       // the tables (w1 nullable: its word shares the row's line)
  int64_t n, N, ldw, ldw1;
  int F;
};

// workgroup `blk` of `nblk` that share the job; `sink` is never non-null in practice, it keeps the loads alive
__device__ __forceinline__ void prefetch_rows_blocks(const PrefetchJob &j, int blk, int nblk, float *sink) {
  float s = 0.f;
  for (int64_t i = (int64_t)blk * kBlock + threadIdx.x; i < j.n; i += (int64_t)nblk * kBlock) {
    const int64_t row = j.idx[i] + (j.offsets ? j.offsets[i % j.F] : 0);
    if ((uint64_t)row < (uint64_t)j.N) {
      s += j.W[row * j.ldw];
      if (j.w1) s += j.w1[row * j.ldw1];
    }
  }
  // the sum is "used" by an empty asm: with a sink pointer the compiler can prove null (a rider's call) the loads would be
  // dead code and vanish — measured: riders that cost nothing and bought nothing
  asm volatile("" ::"v"(s));
  if (sink && s == 1.2345e-30f) sink[0] = s;
}

// host: fills the job from mi_prefetch_rows' arguments; false when there is nothing to do / the arguments are invalid
inline bool prefetch_job(const int64_t *idx, const int64_t *offsets, const float *W, int64_t ldw, const float *w1, int64_t ldw1,
                         int64_t B, int32_t F, int64_t N, PrefetchJob &j) {
  if (B <= 0 || F <= 0 || N <= 0 || !idx || !W || ldw <= 0) return false;
  // a first-order word inside the row's own 128-byte line needs no touch of its own
  const bool same_line = w1 && ldw1 == ldw && w1 > W && (w1 - W) * 4 < 128 && ldw * 4 <= 128;
  j.idx = idx; j.offsets = offsets; j.W = W; j.w1 = same_line ? nullptr : w1;
  j.n = B * F; j.N = N; j.ldw = ldw; j.ldw1 = ldw1 > 0 ? ldw1 : 1; j.F = F;
  return true;
}

}  // namespace mi
