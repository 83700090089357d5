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

// ---- the LAST level (r_{n} = 1): res[(l,h), j] = sum_k chunk_l[h, k] * core[i(l)][k, j], j < q ------------
// Only q (2..4) output columns: as a GEMM its 64-wide tiles are 15/16 padding.  Here one wave owns a lookup:
// its chunk (H x K floats, contiguous in the previous level's layout) and its core slice (K x q floats, the
// whole last core is ~200 KB and stays in L2) are staged in a wave-private LDS slab, the H*q = D outputs are
// D lanes x (64/D) partial sums over k joined by shuffles.  No ordering of the lookups is needed forward; the
// slice gradient walks the lookups in digit order (segments of <= 64) and keeps K*q partial sums in registers.
struct LastArgs {
  const float *src;          // chunks: src + src_row[l] * src_stride, H*K floats each
  const int64_t *src_row;
  int64_t src_stride;
  const float *core;         // [p, K, q]
  const int32_t *digit;      // [n]
  const uint8_t *valid;      // nullable
  int H, K, q;
  int64_t n;
};

constexpr int kLastMaxU = 8;   // K*q <= 512 partial sums per wave in the slice gradient

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kLastRegs = 8;   // floats per lane of a staged chunk / slice: H*K, K*q <= 512

// stage `count` floats of a contiguous run into registers (lane-strided), clamped to the run
__device__ __forceinline__ void fetch_run(const float *__restrict__ p, int count, int lane, bool live, float (&r)[kLastRegs]) {
#pragma unroll
  for (int u = 0; u < kLastRegs; ++u) {
    const int e = lane + u * kWave;
    r[u] = (live && e < count) ? p[e] : 0.f;
  }
}
__device__ __forceinline__ void spill_run(float *lds, int count, int lane, const float (&r)[kLastRegs]) {
#pragma unroll
  for (int u = 0; u < kLastRegs; ++u) {
    const int e = lane + u * kWave;
    if (e < count) lds[e] = r[u];
  }
}

// LDS layout of the forward: chunk rows padded to K+4 floats, the slice TRANSPOSED to [q][K+4], so that a lane's
// k-range is contiguous in both and is read as float4 (K % 4 == 0): 4x fewer LDS instructions than scalar reads.
__global__ __launch_bounds__(kBlock) void k_tt_last_fwd(LastArgs a, float *__restrict__ out) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int HK = a.H * a.K, Kq = a.K * a.q, D = a.H * a.q, KS = a.K + 4;
  float *c = lds + wv * ((a.H + a.q) * KS), *sT = c + a.H * KS;
  const int o = lane % D, kp = lane / D, KP = kWave / D;
  const int h = o / a.q, j = o - h * a.q;
  // this lane's contiguous k-range: [k_lo, k_hi), a multiple of 4 long
  const int kchunk = ((a.K / 4 + KP - 1) / KP) * 4;
  const int k_lo = kp * kchunk < a.K ? kp * kchunk : a.K;
  const int k_hi = k_lo + kchunk < a.K ? k_lo + kchunk : a.K;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + wv, nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  float rc[kLastRegs], rs[kLastRegs];
  if (wave0 < a.n) {
    const bool ok = !a.valid || a.valid[wave0];
    fetch_run(a.src + a.src_row[wave0] * a.src_stride, HK, lane, ok, rc);
    fetch_run(a.core + (int64_t)a.digit[wave0] * Kq, Kq, lane, true, rs);
  }
  for (int64_t l = wave0; l < a.n; l += nwaves) {
#pragma unroll
    for (int u = 0; u < kLastRegs; ++u) {
      const int e = lane + u * kWave;
      if (e < HK) { const int hh = e / a.K; c[hh * KS + (e - hh * a.K)] = rc[u]; }
      if (e < Kq) { const int kk = e / a.q; sT[(e - kk * a.q) * KS + kk] = rs[u]; }
    }
    wave_lds_sync();
    const int64_t ln = l + nwaves < a.n ? l + nwaves : l;      // (the last iteration re-fetches itself: no branch)
    const bool okn = !a.valid || a.valid[ln];
    fetch_run(a.src + a.src_row[ln] * a.src_stride, HK, lane, okn, rc);
    fetch_run(a.core + (int64_t)a.digit[ln] * Kq, Kq, lane, true, rs);
    float acc = 0.f;
    const float *cr = c + h * KS, *sr = sT + j * KS;
    for (int k = k_lo; k < k_hi; k += 4) {
      const float4 x = *reinterpret_cast<const float4 *>(cr + k), y = *reinterpret_cast<const float4 *>(sr + k);
      acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    for (int m = D; m < kWave; m <<= 1) acc += __shfl_xor(acc, m);
    if (kp == 0) out[l * D + o] = acc;
    wave_lds_sync();
  }
}

// dchunk_l[h, k] = sum_j g[l, h*q + j] * core[i(l)][k, j]  -> dst + dst_row[l] * dst_stride
__global__ __launch_bounds__(kBlock) void k_tt_last_bwd_in(LastArgs a, const float *__restrict__ g,
                                                           float *__restrict__ dst,
                                                           const int64_t *__restrict__ dst_row,
                                                           int64_t dst_stride) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int HK = a.H * a.K, Kq = a.K * a.q, D = a.H * a.q;
  float *sl = lds + wv * (Kq + D), *gl = sl + Kq;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + wv, nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  // the (h, k) of this lane's outputs do not depend on the lookup: no integer division inside the loop
  int goff[kLastRegs], soff[kLastRegs];
#pragma unroll
  for (int u = 0; u < kLastRegs; ++u) {
    const int e = lane + u * kWave, h = e / a.K, k = e - h * a.K;
    goff[u] = h * a.q;
    soff[u] = k * a.q;
  }
  float rs[kLastRegs], rg;
  if (wave0 < a.n) {
    fetch_run(a.core + (int64_t)a.digit[wave0] * Kq, Kq, lane, true, rs);
    rg = (lane < D && (!a.valid || a.valid[wave0])) ? g[wave0 * D + lane] : 0.f;
  }
  for (int64_t l = wave0; l < a.n; l += nwaves) {
    spill_run(sl, Kq, lane, rs);
    if (lane < D) gl[lane] = rg;
    wave_lds_sync();
    const int64_t ln = l + nwaves < a.n ? l + nwaves : l;
    fetch_run(a.core + (int64_t)a.digit[ln] * Kq, Kq, lane, true, rs);
    rg = (lane < D && (!a.valid || a.valid[ln])) ? g[ln * D + lane] : 0.f;
    float *d = dst + dst_row[l] * dst_stride;
#pragma unroll
    for (int u = 0; u < kLastRegs; ++u) {
      const int e = lane + u * kWave;
      if (e < HK) {
        float v = 0.f;
        for (int j = 0; j < a.q; ++j) v += gl[goff[u] + j] * sl[soff[u] + j];
        d[e] = v;
      }
    }
    wave_lds_sync();
  }
}

// gcore[group][k, j] += sum over the segment's lookups l, sum_h chunk_l[h, k] * g[l, h*q + j]
// one wave per segment (first sorted position, count, group); order[] maps sorted positions to lookups
template <int U>
__global__ __launch_bounds__(kBlock) void k_tt_last_bwd_core(LastArgs a, const float *__restrict__ g,
                                                             const int64_t *__restrict__ order,
                                                             const long long *__restrict__ kseg, int nseg,
                                                             float *__restrict__ gcore) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int HK = a.H * a.K, Kq = a.K * a.q, D = a.H * a.q;
  float *c = lds + wv * (HK + D), *gl = c + HK;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + wv, nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t sgi = wave0; sgi < nseg; sgi += nwaves) {
    const long long first = kseg[sgi * 3], cnt = kseg[sgi * 3 + 1], grp = kseg[sgi * 3 + 2];
    if (cnt <= 0) continue;
    float acc[U];
    int ck[U], cj[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = lane + u * kWave;
      acc[u] = 0.f;
      ck[u] = e / a.q;
      cj[u] = e - ck[u] * a.q;
    }
    float rc[kLastRegs], rg;
    {
      const int64_t l = order[first];
      fetch_run(a.src + a.src_row[l] * a.src_stride, HK, lane, !a.valid || a.valid[l], rc);
      rg = lane < D ? g[l * D + lane] : 0.f;
    }
    for (long long r = 0; r < cnt; ++r) {
      spill_run(c, HK, lane, rc);
      if (lane < D) gl[lane] = rg;
      wave_lds_sync();
      const int64_t ln = order[first + (r + 1 < cnt ? r + 1 : r)];
      fetch_run(a.src + a.src_row[ln] * a.src_stride, HK, lane, !a.valid || a.valid[ln], rc);
      rg = lane < D ? g[ln * D + lane] : 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (lane + u * kWave < Kq) {
          float v = 0.f;
          for (int h = 0; h < a.H; ++h) v += c[h * a.K + ck[u]] * gl[h * a.q + cj[u]];
          acc[u] += v;
        }
      }
      wave_lds_sync();
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = lane + u * kWave;
      if (e < Kq) atomicAdd(gcore + grp * Kq + e, acc[u]);
    }
  }
}

inline int last_check(LastArgs &a, const float *src, const int64_t *src_row, int64_t src_stride, const float *core,
                      const int32_t *digit, const uint8_t *valid, int H, int K, int q, int64_t n) {
  if (n < 0 || H < 1 || K < 1 || q < 1 || src_stride < 0) return MI_ERR_INVALID_ARG;
  if (n > 0 && (!src || !src_row || !core || !digit)) return MI_ERR_INVALID_ARG;
  const int D = H * q;
  if (D > kWave || (kWave % D) != 0 || K * q > kLastMaxU * kWave || H * K > kLastRegs * kWave) return MI_ERR_UNSUPPORTED;
  a.src = src; a.src_row = src_row; a.src_stride = src_stride; a.core = core; a.digit = digit; a.valid = valid;
  a.H = H; a.K = K; a.q = q; a.n = n;
  return MI_OK;
}

// ---- level planner: a counting sort by digit, all on the device --------------------------------------
// digit[n] in [0, p) -> for every lookup its first row in the level's layout (groups in digit order, each
// padded to whole 64-row tiles, H rows per lookup), the B slice of every row tile, and the reduction
// segments of every group.  Order inside a group is arrival order of the atomics (any order is valid: rows
// of a group are independent in the level GEMM and summed in the slice gradient).
constexpr int kPlanMaxP = 4096;
constexpr int kPlanTile = 64;

__global__ __launch_bounds__(kBlock) void k_plan_count(const int32_t *__restrict__ digit, int64_t n, int p,
                                                       int32_t *__restrict__ counts) {
  extern __shared__ int hist[];
  for (int g = threadIdx.x; g < p; g += blockDim.x) hist[g] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    atomicAdd(&hist[digit[i]], 1);
  __syncthreads();
  for (int g = threadIdx.x; g < p; g += blockDim.x)
    if (hist[g]) atomicAdd(counts + g, hist[g]);
}

// one workgroup of 1024 threads: scans over the p groups, tile map, segment table; zeroes the cursors
__global__ __launch_bounds__(1024) void k_plan_scan(const int32_t *__restrict__ counts, int p, int H, int seg,
                                                    int64_t *__restrict__ pbeg, int32_t *__restrict__ cursor,
                                                    int32_t *__restrict__ mtile_b, int ntiles,
                                                    long long *__restrict__ kseg, int nseg) {
  __shared__ long long scan_a[kPlanMaxP], scan_b[kPlanMaxP];   // padded rows / segment counts (inclusive)
  const int t = threadIdx.x;
  for (int g = t; g < p; g += 1024) {
    const long long rows = (long long)counts[g] * H;
    scan_a[g] = (rows + kPlanTile - 1) / kPlanTile * kPlanTile;
    scan_b[g] = (rows + seg - 1) / seg;
    cursor[g] = 0;
  }
  for (int j = t; j < ntiles; j += 1024) mtile_b[j] = -1;
  for (int j = t; j < nseg; j += 1024) { kseg[j * 3] = 0; kseg[j * 3 + 1] = 0; kseg[j * 3 + 2] = 0; }
  __syncthreads();
  // inclusive scans (Hillis-Steele over <= 4096 entries, 4 per thread)
  for (int d = 1; d < p; d <<= 1) {
    long long va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int g = t + u * 1024;
      va[u] = (g < p && g >= d) ? scan_a[g - d] : 0;
      vb[u] = (g < p && g >= d) ? scan_b[g - d] : 0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int g = t + u * 1024;
      if (g < p) { scan_a[g] += va[u]; scan_b[g] += vb[u]; }
    }
    __syncthreads();
  }
  for (int g = t; g < p; g += 1024) {
    const long long rows = (long long)counts[g] * H;
    const long long rpad = (rows + kPlanTile - 1) / kPlanTile * kPlanTile;
    const long long beg = scan_a[g] - rpad;
    pbeg[g] = beg;
    for (long long j = 0; j < rpad / kPlanTile; ++j) mtile_b[beg / kPlanTile + j] = g;
    const long long chunks = (rows + seg - 1) / seg, c0 = scan_b[g] - chunks;
    for (long long j = 0; j < chunks; ++j) {
      const long long k0 = j * seg;
      kseg[(c0 + j) * 3] = beg + k0;
      kseg[(c0 + j) * 3 + 1] = (rows - k0 < seg) ? rows - k0 : seg;
      kseg[(c0 + j) * 3 + 2] = g;
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_plan_assign(const int32_t *__restrict__ digit, int64_t n, int p, int H,
                                                        const int64_t *__restrict__ pbeg,
                                                        int32_t *__restrict__ cursor,
                                                        int64_t *__restrict__ pos) {
  extern __shared__ int sh[];          // [p] local counts, then [p] bases
  int *cnt = sh, *base = sh + p;
  for (int g = threadIdx.x; g < p; g += blockDim.x) cnt[g] = 0;
  __syncthreads();
  // each workgroup owns one contiguous chunk of lookups: rank inside the chunk, then one global reservation
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = (lo + per < n) ? lo + per : n;
  for (int64_t i0 = lo; i0 < hi; i0 += blockDim.x) {
    const int64_t i = i0 + threadIdx.x;
    if (i < hi) pos[i] = atomicAdd(&cnt[digit[i]], 1);          // local rank, finalised below
  }
  __syncthreads();
  for (int g = threadIdx.x; g < p; g += blockDim.x) base[g] = cnt[g] ? atomicAdd(cursor + g, cnt[g]) : 0;
  __syncthreads();
  for (int64_t i0 = lo; i0 < hi; i0 += blockDim.x) {
    const int64_t i = i0 + threadIdx.x;
    if (i < hi) {
      const int g = digit[i];
      pos[i] = pbeg[g] + (int64_t)(base[g] + (int)pos[i]) * H;
    }
  }
}

}  // namespace

extern "C" {

int mi_tt_last_fwd(const float *src, const int64_t *src_row, int64_t src_stride, const float *core,
                   const int32_t *digit, const uint8_t *valid, int32_t H, int32_t K, int32_t q, float *out,
                   int64_t n, void *stream) {
  LastArgs a;
  const int rc = last_check(a, src, src_row, src_stride, core, digit, valid, H, K, q, n);
  if (rc != MI_OK) return rc;
  if (n == 0) return MI_OK;
  if (!out) return MI_ERR_INVALID_ARG;
  if (K & 3) return MI_ERR_UNSUPPORTED;
  const size_t lds = sizeof(float) * kWavesPerBlock * (size_t)((H + q) * (K + 4));
  hipEvent_t ea, eb;
  if (mi::prof_acquire("tt_last_fwd", &ea, &eb))
    hipExtLaunchKernelGGL(k_tt_last_fwd, dim3(grid_for_waves(n)), dim3(kBlock), lds, (hipStream_t)stream, ea, eb, 0, a, out);
  else
    hipLaunchKernelGGL(k_tt_last_fwd, dim3(grid_for_waves(n)), dim3(kBlock), lds, (hipStream_t)stream, a, out);
  return launch_status();
}

int mi_tt_last_bwd(const float *src, const int64_t *src_row, int64_t src_stride, const float *core,
                   const int32_t *digit, const uint8_t *valid, int32_t H, int32_t K, int32_t q, const float *g,
                   float *dst, const int64_t *dst_row, int64_t dst_stride, const int64_t *order,
                   const int64_t *kseg, int32_t nseg, float *gcore, int64_t n, void *stream) {
  LastArgs a;
  const int rc = last_check(a, src, src_row, src_stride, core, digit, valid, H, K, q, n);
  if (rc != MI_OK) return rc;
  if (n == 0) return MI_OK;
  if (!g) return MI_ERR_INVALID_ARG;
  const hipStream_t st = (hipStream_t)stream;
  hipEvent_t ea, eb;
  if (dst) {   // input gradient, written in the previous level's layout
    if (!dst_row) return MI_ERR_INVALID_ARG;
    const size_t lds = sizeof(float) * kWavesPerBlock * (size_t)(K * q + H * q);
    if (mi::prof_acquire("tt_last_bwd_in", &ea, &eb))
      hipExtLaunchKernelGGL(k_tt_last_bwd_in, dim3(grid_for_waves(n)), dim3(kBlock), lds, st, ea, eb, 0, a, g, dst, dst_row, dst_stride);
    else
      hipLaunchKernelGGL(k_tt_last_bwd_in, dim3(grid_for_waves(n)), dim3(kBlock), lds, st, a, g, dst, dst_row, dst_stride);
  }
  if (gcore) {  // slice gradients, lookups walked in digit order
    if (!order || !kseg || nseg < 0) return MI_ERR_INVALID_ARG;
    const size_t lds = sizeof(float) * kWavesPerBlock * (size_t)(H * K + H * q);
    const int U = (K * q + kWave - 1) / kWave;
    const dim3 grid(grid_for_waves(nseg));
    const long long *ks = reinterpret_cast<const long long *>(kseg);
    const bool prof = mi::prof_acquire("tt_last_bwd_core", &ea, &eb);
#define GO(UU)                                                                                              \
  do {                                                                                                      \
    if (prof) hipExtLaunchKernelGGL((k_tt_last_bwd_core<UU>), grid, dim3(kBlock), lds, st, ea, eb, 0, a, g, order, ks, nseg, gcore); \
    else hipLaunchKernelGGL((k_tt_last_bwd_core<UU>), grid, dim3(kBlock), lds, st, a, g, order, ks, nseg, gcore);  \
  } while (0)
    switch (U) {
      case 1: GO(1); break;
      case 2: GO(2); break;
      case 3: GO(3); break;
      case 4: GO(4); break;
      case 5: GO(5); break;
      case 6: GO(6); break;
      case 7: GO(7); break;
      default: GO(8); break;
    }
#undef GO
  }
  return launch_status();
}

int mi_tt_plan_level(const int32_t *digit, int64_t n, int32_t p, int32_t H, int32_t seg, int32_t *workspace,
                     int64_t *pbeg, int64_t *pos, int32_t *mtile_b, int32_t ntiles, int64_t *kseg,
                     int32_t nseg, void *stream) {
  if (n < 0 || p < 1 || H < 1 || seg < 1 || ntiles < 0 || nseg < 0) return MI_ERR_INVALID_ARG;
  if (p > kPlanMaxP) return MI_ERR_UNSUPPORTED;
  if (!workspace || !pbeg || !mtile_b || !kseg || (n > 0 && (!digit || !pos))) return MI_ERR_INVALID_ARG;
  // the caller sized the tables for the worst case: every group wastes less than one tile / segment
  if ((int64_t)ntiles < (n * H + kPlanTile - 1) / kPlanTile + p || (int64_t)nseg < (n * H + seg - 1) / seg + p)
    return MI_ERR_INVALID_ARG;
  int32_t *counts = workspace, *cursor = workspace + p;
  if (hipMemsetAsync(counts, 0, sizeof(int32_t) * p, (hipStream_t)stream) != hipSuccess) return MI_ERR_LAUNCH;
  int64_t g = (n + 1023) / 1024;
  if (g < 1) g = 1;
  if (g > 1024) g = 1024;
  const hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_plan_count, dim3((int)g), dim3(kBlock), sizeof(int) * p, st, digit, n, p, counts);
  hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(1024), 0, st, counts, p, H, seg, pbeg, cursor, mtile_b, ntiles,
                     reinterpret_cast<long long *>(kseg), nseg);
  hipLaunchKernelGGL(k_plan_assign, dim3((int)g), dim3(kBlock), sizeof(int) * 2 * p, st, digit, n, p, H, pbeg, cursor, pos);
  return launch_status();
}

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
