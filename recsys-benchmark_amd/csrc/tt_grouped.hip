// tt_grouped.hip — data movement of the grouped TT-Rec lookup (TTRecTorch semantics,
// src/models/embeddings/tensortrain_embeddings.py:100-150).
//
// The per-lookup chain of tt.hip re-reads a whole middle-core slice (r_c*q_c*r_{c+1} floats, 98 KB
// at ranks [128, 96]) for every lookup.  The grouped form sorts the lookups of a level by that
// level's digit so that all lookups sharing a slice form consecutive rows of ONE matrix, and the
// level becomes a GEMM per slice on the MFMA units (gemm.hip: mi_gemm_f32_row_groups for the level
// and its input gradient, mi_gemm_f32_k_groups for the slice gradients):
//
//   level c (1 <= c < ncores), lookups ordered by digit i_c, H = q_0*...*q_{c-1} rows per lookup:
//     res_c[(l,h), (qq,r')] = sum_j res_{c-1}[(l,h), j] * core_c[i_c(l)][j, (qq,r')]
//
// This file holds what is not a GEMM: the mixed-radix digit split (k_tt_digits) and the chunk mover
// that re-orders a level's rows into the next level's grouped layout, builds the first level from
// core 0's slices, restores lookup order at the end, and scatter-adds core 0's gradient
// (k_move_chunks).  Integer work is exact; the moves are copies (or float atomics when accumulating).
#include "common.hpp"

namespace {
using namespace mi;

constexpr int kMaxCores = 4;

struct DigitArgs {
  int ncores;
  int p[kMaxCores];
};

// digits[c*n + i] = c-th mixed-radix digit of idx[i] over p[0..ncores); out of range -> all 0 + error word
__global__ __launch_bounds__(kBlock) void k_tt_digits(const int64_t *__restrict__ idx, int64_t n,
                                                      int64_t N, DigitArgs t,
                                                      int32_t *__restrict__ digits,
                                                      uint8_t *__restrict__ valid, int *err) {
  int bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t id = idx[i];
    const bool ok = (uint64_t)id < (uint64_t)N;
    bad |= !ok;
    if (!ok) id = 0;
    if (valid) valid[i] = ok ? 1 : 0;
    int64_t big = 1;
    for (int c = 0; c < t.ncores; ++c) big *= t.p[c];
    for (int c = 0; c < t.ncores; ++c) {
      big /= t.p[c];
      digits[(int64_t)c * n + i] = (int32_t)(id / big);
      id = id % big;
    }
  }
  if (bad && err) atomicOr(err, MI_IDX_OUT_OF_RANGE);
}

// dst[dst_row(i)*dst_stride + e] (=|+=) src[src_row(i)*src_stride + e] * (mask ? mask[i] : 1),  e < width
// one float4 per thread; null row arrays mean the identity
template <bool ACC>
__global__ __launch_bounds__(kBlock) void k_move_chunks(
    const float *__restrict__ src, const int64_t *__restrict__ src_row, int64_t src_stride,
    float *__restrict__ dst, const int64_t *__restrict__ dst_row, int64_t dst_stride, int width4,
    int64_t n, const uint8_t *__restrict__ mask) {
  const int64_t total = n * width4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / width4;
    const int w = (int)(e - i * width4) * 4;
    const int64_t sr = src_row ? src_row[i] : i;
    const int64_t dr = dst_row ? dst_row[i] : i;
    float4 v = ld4(src + sr * src_stride + w);
    if (mask && !mask[i]) v = make_float4(0.f, 0.f, 0.f, 0.f);
    float *d = dst + dr * dst_stride + w;
    if constexpr (ACC) {
      if (!mask || mask[i]) {
        atomicAdd(d + 0, v.x);
        atomicAdd(d + 1, v.y);
        atomicAdd(d + 2, v.z);
        atomicAdd(d + 3, v.w);
      }
    } else {
      st4(d, v);
    }
  }
}

// out[seg.slice*ldo + col] += sum_{r < seg.rows} X[(seg.first + r)*ldx + col]; blockIdx.x = segment,
// blockIdx.y = block of 256 columns; rows are read as coalesced runs, 4 independent partial sums
__global__ __launch_bounds__(kBlock) void k_segment_sum(const float *__restrict__ X, int64_t ldx,
                                                        int width, const long long *__restrict__ kseg,
                                                        float *__restrict__ out, int64_t ldo) {
  const long long k0 = kseg[blockIdx.x * 3], K = kseg[blockIdx.x * 3 + 1], g = kseg[blockIdx.x * 3 + 2];
  const int col = blockIdx.y * kBlock + threadIdx.x;
  if (K <= 0 || col >= width) return;
  const float *p = X + k0 * ldx + col;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  long long r = 0;
  for (; r + 4 <= K; r += 4) {
    a0 += p[(r + 0) * ldx];
    a1 += p[(r + 1) * ldx];
    a2 += p[(r + 2) * ldx];
    a3 += p[(r + 3) * ldx];
  }
  for (; r < K; ++r) a0 += p[r * ldx];
  atomicAdd(out + g * ldo + col, (a0 + a1) + (a2 + a3));
}

}  // namespace

extern "C" {

int mi_segment_sum(const float *X, int64_t ldx, int32_t width, const int64_t *kseg, int32_t nseg,
                   float *out, int64_t ldo, void *stream) {
  if (width < 0 || nseg < 0 || ldx < 0 || ldo < 0) return MI_ERR_INVALID_ARG;
  if (width == 0 || nseg == 0) return MI_OK;
  if (!X || !kseg || !out) return MI_ERR_INVALID_ARG;
  dim3 grid(nseg, (width + kBlock - 1) / kBlock);
  if (grid.y > 65535) return MI_ERR_UNSUPPORTED;
  MI_LAUNCH("segment_sum", k_segment_sum, grid, kBlock, stream, X, ldx, width,
            reinterpret_cast<const long long *>(kseg), out, ldo);
  return launch_status();
}

int mi_tt_digits(const int64_t *idx, int64_t n, int64_t N, const int32_t *p_shapes, int32_t ncores,
                 int32_t *digits, uint8_t *valid, int32_t *err, void *stream) {
  if (n < 0 || N < 0 || ncores < 1 || ncores > kMaxCores || !p_shapes) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!idx || !digits) return MI_ERR_INVALID_ARG;
  DigitArgs t;
  t.ncores = ncores;
  for (int c = 0; c < kMaxCores; ++c) t.p[c] = c < ncores ? p_shapes[c] : 1;
  for (int c = 0; c < ncores; ++c)
    if (t.p[c] < 1) return MI_ERR_INVALID_ARG;
  int64_t g = (n + kBlock - 1) / kBlock;
  if (g > kMaxGrid) g = kMaxGrid;
  MI_LAUNCH("tt_digits", k_tt_digits, (int)g, kBlock, stream, idx, n, N, t, digits, valid, err);
  return launch_status();
}

int mi_move_chunks(const float *src, const int64_t *src_row, int64_t src_stride, float *dst,
                   const int64_t *dst_row, int64_t dst_stride, int32_t width, int64_t n,
                   const uint8_t *mask, int32_t accumulate, void *stream) {
  if (n < 0 || width < 0 || src_stride < 0 || dst_stride < 0) return MI_ERR_INVALID_ARG;
  if (n == 0 || width == 0) return MI_OK;
  if (!src || !dst) return MI_ERR_INVALID_ARG;
  if ((width & 3) || (src_stride & 3) || (dst_stride & 3) || !aligned16(src) || !aligned16(dst))
    return MI_ERR_UNSUPPORTED;
  const int width4 = width / 4;
  int64_t g = (n * width4 + kBlock - 1) / kBlock;
  if (g > kMaxGrid * 4) g = kMaxGrid * 4;
  if (accumulate)
    MI_LAUNCH("move_chunks_acc", k_move_chunks<true>, (int)g, kBlock, stream, src, src_row, src_stride,
              dst, dst_row, dst_stride, width4, n, mask);
  else
    MI_LAUNCH("move_chunks", k_move_chunks<false>, (int)g, kBlock, stream, src, src_row, src_stride, dst,
              dst_row, dst_stride, width4, n, mask);
  return launch_status();
}

}  // extern "C"
