// sort.hip — order a multi-field batch's row ids for the sparse optimizer step.
//
// A generic sort of the B*F int64 ids costs 65 us at the headline shape (torch.sort: block sort + 7 merge passes;
// rocPRIM's onesweep radix sort restricted to bit_width(N) bits: 95 us — both launch-bound at 1e5 keys).  The ids are
// not generic: rows[b, f] = x[b, f] + offsets[f] (src/models/deepfm.py:88) lies in field f's own range
// [offsets[f], offsets[f+1]), and the ranges ascend with f.  So the sorted order of the whole batch is the F columns
// sorted one by one and laid end to end: F independent sorts of B keys, each by ONE workgroup entirely in LDS
// (bitonic network over (id - offsets[f]) << 32 | b: unique keys, so the result equals a stable sort), one launch.
#include "common.hpp"

namespace {
using namespace mi;

constexpr int kSortThreads = 1024;
constexpr uint64_t kBadRel = 0xFFFFFFFEull;     // an id outside its field's range: sorts behind the field's valid ids

__global__ __launch_bounds__(kSortThreads) void k_sort_fields(const int64_t *__restrict__ rows,
                                                              const int64_t *__restrict__ offsets, int64_t N, int B, int F,
                                                              int P, int64_t *__restrict__ rows_sorted,
                                                              int64_t *__restrict__ perm) {
  extern __shared__ uint64_t keys[];
  const int f = blockIdx.x;
  const int64_t lo = offsets[f], hi = f + 1 < F ? offsets[f + 1] : N;
  for (int b = threadIdx.x; b < P; b += kSortThreads) {
    uint64_t key = ~0ull;                                   // padding up to the power of two: sorts last
    if (b < B) {
      const int64_t r = rows[(int64_t)b * F + f];
      const uint64_t rel = (r >= lo && r < hi) ? (uint64_t)(r - lo) : kBadRel;
      key = rel << 32 | (uint64_t)b;
    }
    keys[b] = key;
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (P >> 1); t += kSortThreads) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // t with a zero inserted at bit log2(j)
        const int l = i | j;
        const uint64_t a = keys[i], c = keys[l];
        if ((a > c) == ((i & k) == 0)) { keys[i] = c; keys[l] = a; }
      }
      __syncthreads();
    }
  }
  for (int b = threadIdx.x; b < B; b += kSortThreads) {
    const uint64_t key = keys[b], rel = key >> 32;
    const int64_t o = (int64_t)f * B + b;
    rows_sorted[o] = rel == kBadRel ? N : lo + (int64_t)rel;  // N: the id every row-wise kernel skips
    perm[o] = (int64_t)(key & 0xFFFFFFFFull) * F + f;
  }
}
}  // namespace

extern "C" {

int mi_sort_field_rows(const int64_t *rows, const int64_t *offsets, int64_t N, int64_t B, int32_t F, int64_t *rows_sorted,
                       int64_t *perm, void *stream) {
  if (B < 0 || F <= 0 || N < 0) return MI_ERR_INVALID_ARG;
  if (B == 0) return MI_OK;
  if (!rows || !offsets || !rows_sorted || !perm) return MI_ERR_INVALID_ARG;
  if (B > 8192 || N >= (int64_t)kBadRel) return MI_ERR_UNSUPPORTED;   // 64 KiB of LDS keys; 32-bit in-field ids
  int P = 2;
  while (P < B) P <<= 1;
  hipLaunchKernelGGL(k_sort_fields, dim3(F), dim3(kSortThreads), (size_t)P * sizeof(uint64_t), (hipStream_t)stream, rows,
                     offsets, N, (int)B, F, P, rows_sorted, perm);
  return launch_status();
}

}  // extern "C"
