// sort.hip — order a multi-field batch's row ids for the sparse optimizer step.
//
// A generic sort of the B*F int64 ids costs 65 us at the headline shape (torch.sort: block sort + 7 merge passes;
// rocPRIM's onesweep radix sort restricted to bit_width(N) bits: 95 us — both launch-bound at 1e5 keys).  The ids are
// not generic: rows[b, f] = x[b, f] + offsets[f] (src/models/deepfm.py:88) lies in field f's own range
// [offsets[f], offsets[f+1]), and the ranges ascend with f.  So the sorted order of the whole batch is the F columns
// sorted one by one and laid end to end: F independent sorts of B keys (bitonic networks over
// (id - offsets[f]) << 32 | b: unique keys, so the result equals a stable sort).
#include "common.hpp"

namespace {
using namespace mi;

constexpr int kSortThreads = 1024;
constexpr uint64_t kBadRel = 0xFFFFFFFEull;     // an id outside its field's range: sorts behind the field's valid ids

// One workgroup's VALU is the bound of an in-LDS sort (16 waves share 4 SIMDs: a 4096-key column took 36 us on ONE
// CU while 230 CUs idled), so a column is cut into runs of 1024 keys, each sorted by its own workgroup — one key per
// thread: partner distance j < 64 is a wave shuffle, j >= 64 a trip through LDS (10 of the 55 stages) — and a second
// kernel merges: a key's final position is its index in its run plus, for every other run of the column, the number of
// keys below it (binary search; keys are unique).
constexpr int kRun = kSortThreads;

__device__ __forceinline__ void emit(uint64_t key, int64_t lo, int64_t N, int F, int f, int64_t o,
                                     int64_t *__restrict__ rows_sorted, int64_t *__restrict__ perm) {
  const uint64_t rel = key >> 32;
  rows_sorted[o] = rel == kBadRel ? N : lo + (int64_t)rel;    // N: the id every row-wise kernel skips
  perm[o] = (int64_t)(key & 0xFFFFFFFFull) * F + f;
}

__global__ __launch_bounds__(kSortThreads) void k_sort_runs(const int64_t *__restrict__ rows,
                                                            const int64_t *__restrict__ offsets, int64_t N, int B, int F,
                                                            uint64_t *__restrict__ runs, int64_t *__restrict__ rows_sorted,
                                                            int64_t *__restrict__ perm, int *err) {
  __shared__ uint64_t lds[kRun];
  const int f = blockIdx.x, r = blockIdx.y, R = gridDim.y, t = threadIdx.x;
  const int64_t lo = offsets[f], hi = f + 1 < F ? offsets[f + 1] : N;
  const int b = r * kRun + t;
  uint64_t key = ~0ull;                                        // padding of the last run: sorts last
  if (b < B) {
    const int64_t id = rows[(int64_t)b * F + f];
    const uint64_t rel = (id >= lo && id < hi) ? (uint64_t)(id - lo) : kBadRel;
    key = rel << 32 | (uint64_t)b;
    // an id in [0, N) but outside its own field was ACCEPTED by the lookup (the reference reads another field's row there,
    // src/models/deepfm.py:88); its gradient cannot be placed by a per-field sort, so it is dropped LOUDLY: sticky bit
    if (rel == kBadRel && id >= 0 && id < N && err) atomicOr(err, MI_IDX_OUT_OF_FIELD);
  }
  for (int k = 2; k <= kRun; k <<= 1) {
    const bool up = (t & k) == 0;
    for (int j = k >> 1; j > 0; j >>= 1) {
      uint64_t o;
      if (j < kWave) {
        o = (uint64_t)__shfl_xor((long long)key, j);
      } else {
        lds[t] = key;
        __syncthreads();
        o = lds[t ^ j];
        __syncthreads();
      }
      const bool lower = (t & j) == 0;
      key = (lower == up) ? (key < o ? key : o) : (key < o ? o : key);
    }
  }
  if (R == 1) {
    if (t < B) emit(key, lo, N, F, f, (int64_t)f * B + t, rows_sorted, perm);
  } else {
    runs[((int64_t)f * R + r) * kRun + t] = key;
  }
}

__global__ __launch_bounds__(kBlock) void k_merge_runs(const uint64_t *__restrict__ runs,
                                                       const int64_t *__restrict__ offsets, int64_t N, int B, int F, int R,
                                                       int64_t *__restrict__ rows_sorted, int64_t *__restrict__ perm) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;      // (f, r, i) flattened
  const int i = (int)(e % kRun), r = (int)((e / kRun) % R), f = (int)(e / ((int64_t)kRun * R));
  if (f >= F || r * kRun + i >= B) return;
  const uint64_t *col = runs + (int64_t)f * R * kRun;
  const uint64_t key = col[r * kRun + i];
  int pos = i;
  for (int q = 0; q < R; ++q) {
    if (q == r) continue;
    const uint64_t *run = col + q * kRun;
    int lo_i = 0, hi_i = min(kRun, B - q * kRun);                    // keys of the run that are real
    while (lo_i < hi_i) {
      const int mid = (lo_i + hi_i) >> 1;
      if (run[mid] < key) lo_i = mid + 1; else hi_i = mid;
    }
    pos += lo_i;
  }
  emit(key, offsets[f], N, F, f, (int64_t)f * B + pos, rows_sorted, perm);
}
}  // namespace

extern "C" {

int64_t mi_sort_field_rows_workspace_bytes(int64_t B, int32_t F) {
  if (B <= kRun || F <= 0) return 0;
  return (int64_t)F * ((B + kRun - 1) / kRun) * kRun * (int64_t)sizeof(uint64_t);
}

int mi_sort_field_rows(const int64_t *rows, const int64_t *offsets, int64_t N, int64_t B, int32_t F, int64_t *rows_sorted,
                       int64_t *perm, void *workspace, int32_t *err, void *stream) {
  if (B < 0 || F <= 0 || N < 0) return MI_ERR_INVALID_ARG;
  if (B == 0) return MI_OK;
  if (!rows || !offsets || !rows_sorted || !perm) return MI_ERR_INVALID_ARG;
  if (B > 64 * kRun || N >= (int64_t)kBadRel) return MI_ERR_UNSUPPORTED;   // <= 64 runs per column; 32-bit in-field ids
  const int R = (int)((B + kRun - 1) / kRun);
  if (R > 1 && !workspace) return MI_ERR_INVALID_ARG;
  uint64_t *runs = static_cast<uint64_t *>(workspace);
  MI_LAUNCH("sort_runs", k_sort_runs, dim3(F, R), kSortThreads, stream, rows, offsets, N, (int)B, F, runs, rows_sorted, perm,
            err);
  if (R > 1) {
    const int64_t total = (int64_t)F * R * kRun;
    MI_LAUNCH("merge_runs", k_merge_runs, (int)((total + kBlock - 1) / kBlock), kBlock, stream, runs, offsets, N, (int)B, F,
              R, rows_sorted, perm);
  }
  return launch_status();
}

}  // extern "C"
