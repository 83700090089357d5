// route.hip — the table-sharded DeepFM lookup's routing on the device.
//
// The reference has one table on one device (src/models/embeddings/base.py:52-57), so there is no
// reference code for this file; the arithmetic it feeds (gather, FM, first-order term) is the
// reference's (src/models/deepfm.py:88-98).  Tables are row-sharded `owner = row % world`,
// `local = row / world`.  Every step a rank
//   1. buckets its n = B*F global rows by owner into a [world, cap] send buffer of owner-local
//      row ids (k_route_count / k_route_scan / k_route_assign: STABLE, i.e. bucket order = lookup
//      order, so results do not depend on scheduling), remembering each lookup's slot;
//   2. after the id all-to-all, packs W[local] and w1[local] into rows of ldo = D + 4 floats
//      (k_gather_pack: one buffer, one all-to-all for both tables);
//   3. after the row all-to-all, the fused gather + FM kernel of gather_fm.hip reads the received
//      buffer through the slot array (mi_slot_fm_fwd), and its backward scatters the gradient
//      rows straight into the outgoing buffer (mi_slot_fm_bwd).
// Static shapes throughout (no host sync, hipGraph-friendly): a bucket holds exactly `cap` slots;
// unused slots carry the owner's SINK row (one extra all-zero-gradient row at the end of every
// shard), lookups that do not fit or are out of range go to the DUMP slot world*cap (a zero row
// of the received buffer) and raise the overflow / index-error words.
#include "common.hpp"

namespace {
using namespace mi;

constexpr int kIt = 4;                    // lookups per thread
constexpr int kChunk = kBlock * kIt;      // lookups per workgroup
constexpr int kMaxWorld = 64;
constexpr int kSelfScanGroups = 512;    // up to this many count workgroups the assign launch scans the counts itself

// floor-mod owner / floor-div local row; own = -1 for an out-of-range or inactive lookup
__device__ __forceinline__ void classify(const int64_t *__restrict__ idx,
                                         const int64_t *__restrict__ offsets, int F, int64_t i,
                                         int64_t n, int world, int64_t N, int &own, int64_t &loc,
                                         bool &valid) {
  valid = i < n;
  own = -1;
  loc = 0;
  if (valid) {
    const int64_t row = idx[i] + (offsets ? offsets[i % F] : 0);
    if ((uint64_t)row < (uint64_t)N) {
      loc = row / world;
      own = (int)(row - loc * world);
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_route_count(
    const int64_t *__restrict__ idx, const int64_t *__restrict__ offsets, int F, int64_t n,
    int world, int64_t N, int32_t *__restrict__ counts) {
  __shared__ int c[kMaxWorld];
  if (threadIdx.x < kMaxWorld) c[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < kIt; ++k) {
    const int64_t i = (int64_t)blockIdx.x * kChunk + k * kBlock + threadIdx.x;
    int own;
    int64_t loc;
    bool valid;
    classify(idx, offsets, F, i, n, world, N, own, loc, valid);
    for (int w = 0; w < world; ++w) {
      const unsigned long long m = __ballot(own == w);
      if (lane == 0 && m) atomicAdd(&c[w], __popcll(m));
    }
  }
  __syncthreads();
  if (threadIdx.x < world) counts[(int64_t)blockIdx.x * world + threadIdx.x] = c[threadIdx.x];
}

// exclusive prefix of counts over workgroups, per owner; one workgroup
__global__ __launch_bounds__(kBlock) void k_route_scan(const int32_t *__restrict__ counts, int G,
                                                        int world, int32_t *__restrict__ base,
                                                        int32_t *__restrict__ total) {
  __shared__ int wsum[kWavesPerBlock];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int w = 0; w < world; ++w) {
    int carry = 0;
    for (int g0 = 0; g0 < G; g0 += kBlock) {
      const int g = g0 + threadIdx.x;
      const int v = g < G ? counts[(int64_t)g * world + w] : 0;
      int incl = v;
#pragma unroll
      for (int d = 1; d < kWave; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
      }
      if (lane == 63) wsum[wv] = incl;
      __syncthreads();
      int before = 0, all = 0;
#pragma unroll
      for (int j = 0; j < kWavesPerBlock; ++j) {
        before += j < wv ? wsum[j] : 0;
        all += wsum[j];
      }
      if (g < G) base[(int64_t)g * world + w] = carry + before + incl - v;
      carry += all;
      __syncthreads();
    }
    if (threadIdx.x == 0) total[w] = carry;
  }
}

// SELF_SCAN: no k_route_scan launch ran — every workgroup adds up the counts of the workgroups before it (its base) and of
// all of them (the totals) itself: G x world integers, a few KB while G is a few hundred (B F = 106 K lookups: G = 104).
// One launch (~4 us of boundary + a one-workgroup kernel) less on the critical path of every sharded step.
template <bool SELF_SCAN>
__global__ __launch_bounds__(kBlock) void k_route_assign(
    const int64_t *__restrict__ idx, const int64_t *__restrict__ offsets, int F, int64_t n,
    int world, int64_t N, int64_t cap, const int32_t *__restrict__ base,
    const int32_t *__restrict__ total, int64_t *__restrict__ send_rows,
    int64_t *__restrict__ slot, int32_t *overflow, int32_t *err, const int32_t *__restrict__ counts) {
  __shared__ int cnt[kIt][kWavesPerBlock][kMaxWorld];
  __shared__ int sbase[kMaxWorld], stotal[kMaxWorld];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if constexpr (SELF_SCAN) {
    const int G = gridDim.x;
    for (int w = wv; w < world; w += kWavesPerBlock) {
      int b = 0, t = 0;
      for (int g = lane; g < G; g += kWave) {
        const int v = counts[(int64_t)g * world + w];
        t += v;
        b += g < (int)blockIdx.x ? v : 0;
      }
#pragma unroll
      for (int off = 32; off; off >>= 1) {
        b += __shfl_xor(b, off);
        t += __shfl_xor(t, off);
      }
      if (lane == 0) { sbase[w] = b; stotal[w] = t; }
    }
  }
  const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
  int own[kIt], rank[kIt];
  int64_t loc[kIt];
  bool valid[kIt];
#pragma unroll
  for (int k = 0; k < kIt; ++k) {
    const int64_t i = (int64_t)blockIdx.x * kChunk + k * kBlock + threadIdx.x;
    classify(idx, offsets, F, i, n, world, N, own[k], loc[k], valid[k]);
    rank[k] = 0;
    for (int w = 0; w < world; ++w) {
      const unsigned long long m = __ballot(own[k] == w);
      if (own[k] == w) rank[k] = __popcll(m & below);
      if (lane == 0) cnt[k][wv][w] = __popcll(m);
    }
  }
  __syncthreads();
  const int64_t dump = (int64_t)world * cap;
  int over = 0, bad = 0;
#pragma unroll
  for (int k = 0; k < kIt; ++k) {
    if (!valid[k]) continue;
    const int64_t i = (int64_t)blockIdx.x * kChunk + k * kBlock + threadIdx.x;
    int64_t s = dump;
    if (own[k] >= 0) {
      int64_t pos = (SELF_SCAN ? sbase[own[k]] : base[(int64_t)blockIdx.x * world + own[k]]) + rank[k];
      // lookups before this one in index order: earlier k, or same k and an earlier wave
      for (int kk = 0; kk <= k; ++kk) {
        const int wend = kk < k ? kWavesPerBlock : wv;
        for (int j = 0; j < wend; ++j) pos += cnt[kk][j][own[k]];
      }
      if (pos < cap) {
        s = own[k] * cap + pos;
        send_rows[s] = loc[k];
      } else {
        over = 1;
      }
    } else {
      bad = 1;
    }
    slot[i] = s;
  }
  // unused slots of every bucket point at the owner's sink row
  for (int w = 0; w < world; ++w) {
    const int64_t tw = SELF_SCAN ? stotal[w] : total[w];
    const int64_t used = tw < cap ? tw : cap;
    const int64_t sink = (N - w + world - 1) / world;   // rows the owner really has
    for (int64_t j = used + (int64_t)blockIdx.x * kBlock + threadIdx.x; j < cap;
         j += (int64_t)gridDim.x * kBlock)
      send_rows[w * cap + j] = sink;
  }
  if (over && overflow) atomicOr(overflow, 1);
  if (bad && err) atomicOr(err, MI_IDX_OUT_OF_RANGE);
}

// out[i, 0:D] = W[rows[i], :], out[i, D:D+4] = (w1[rows[i]], 0, 0, 0); rows of ldo = D + 4 floats
template <int LPR>
__global__ __launch_bounds__(kBlock) void k_gather_pack(
    const int64_t *__restrict__ rows, const float *__restrict__ W, const float *__restrict__ w1,
    float *__restrict__ out, int64_t m, int64_t Nl, int *err) {
  constexpr int RS = kWave / LPR;
  constexpr int D = LPR * 4;
  constexpr int LDO = D + 4;
  constexpr int U = 4;
  const int lane = threadIdx.x & 63;
  const int q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ntiles = (m + (int64_t)RS * U - 1) / ((int64_t)RS * U);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  int bad = 0;
  for (int64_t t = wave0; t < ntiles; t += nwaves) {
    int64_t row[U];
    bool act[U], ok[U];
    float4 v[U];
    float l[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t i = (t * U + k) * RS + r;
      act[k] = i < m;
      row[k] = act[k] ? rows[i] : 0;
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      ok[k] = act[k] && (uint64_t)row[k] < (uint64_t)Nl;
      bad |= (act[k] && !ok[k]);
      v[k] = ok[k] ? ld4(W + row[k] * D + q * 4) : z;
      l[k] = (ok[k] && q == 0) ? w1[row[k]] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t i = (t * U + k) * RS + r;
      if (act[k]) {
        st4(out + i * LDO + q * 4, v[k]);
        if (q == 0) st4(out + i * LDO + D, make_float4(l[k], 0.f, 0.f, 0.f));
      }
    }
  }
  if (bad && err) atomicOr(err, MI_IDX_OUT_OF_RANGE);
}

// vals[i, 0:D] = packed[i, 0:D], lin[i] = packed[i, D]: the received gradient rows as the two contiguous value arrays of
// the shards' row-form gradients, one launch (two strided torch copies before: 2 x ~5 us, launch-bound)
template <int LPR>
__global__ __launch_bounds__(kBlock) void k_unpack_rows(const float *__restrict__ packed, float *__restrict__ vals,
                                                        float *__restrict__ lin, int64_t m) {
  constexpr int RS = kWave / LPR;
  constexpr int D = LPR * 4;
  constexpr int LDO = D + 4;
  const int lane = threadIdx.x & 63;
  const int q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t t = wave0; t * RS < m; t += nwaves) {
    const int64_t i = t * RS + r;
    if (i < m) {
      st4(vals + i * D + q * 4, ld4(packed + i * LDO + q * 4));
      if (q == 0) lin[i] = packed[i * LDO + D];
    }
  }
}

inline bool vec_ok(int D) {
  return D >= 4 && D <= 256 && (D & 3) == 0 && ((D >> 2) & ((D >> 2) - 1)) == 0;
}
inline int64_t route_groups(int64_t n) { return (n + kChunk - 1) / kChunk; }

}  // namespace

extern "C" {

int64_t mi_route_workspace_elems(int64_t n, int32_t world) {
  if (n < 0 || world < 1) return 0;
  const int64_t G = route_groups(n) > 0 ? route_groups(n) : 1;
  return 2 * G * world + world;
}

int mi_route_buckets(const int64_t *idx, const int64_t *offsets, int64_t n, int32_t F,
                     int32_t world, int64_t N, int64_t cap, int32_t *workspace,
                     int64_t *send_rows, int64_t *slot, int32_t *overflow, int32_t *err,
                     void *stream) {
  if (n < 0 || F < 1 || world < 1 || N < 0 || cap < 0) return MI_ERR_INVALID_ARG;
  if (world > kMaxWorld || n >= (1ll << 31) || world * cap >= (1ll << 40)) return MI_ERR_UNSUPPORTED;
  if (!workspace || (!send_rows && world * cap > 0) || (n > 0 && (!idx || !slot)))
    return MI_ERR_INVALID_ARG;
  const int64_t G64 = route_groups(n) > 0 ? route_groups(n) : 1;
  const int G = (int)G64;
  int32_t *counts = workspace, *base = workspace + G64 * world, *total = base + G64 * world;
  MI_LAUNCH("route_count", k_route_count, G, kBlock, stream, idx, offsets, F, n, world, N, counts);
  if (G <= kSelfScanGroups) {      // two launches: the assigning workgroups scan the counts themselves
    MI_LAUNCH("route_assign", k_route_assign<true>, G, kBlock, stream, idx, offsets, F, n, world, N, cap, base, total,
              send_rows, slot, overflow, err, counts);
    return launch_status();
  }
  MI_LAUNCH("route_scan", k_route_scan, 1, kBlock, stream, counts, G, world, base, total);
  MI_LAUNCH("route_assign", k_route_assign<false>, G, kBlock, stream, idx, offsets, F, n, world, N, cap,
            base, total, send_rows, slot, overflow, err, counts);
  return launch_status();
}

int mi_gather_pack_rows(const int64_t *rows, const float *W, const float *w1, float *out,
                        int64_t m, int32_t D, int64_t Nl, int32_t *err, void *stream) {
  if (m < 0 || D <= 0 || Nl < 0) return MI_ERR_INVALID_ARG;
  if (m == 0) return MI_OK;
  if (!rows || !W || !w1 || !out) return MI_ERR_INVALID_ARG;
  if (!vec_ok(D) || !aligned16(W) || !aligned16(out)) return MI_ERR_UNSUPPORTED;
  const int64_t rows_per_tile = (int64_t)(kWave / (D / 4)) * 4;
  const int grid = grid_for_waves((m + rows_per_tile - 1) / rows_per_tile);
  switch (D / 4) {
#define CASE(LPR)                                                                            \
  case LPR:                                                                                  \
    MI_LAUNCH("gather_pack", (k_gather_pack<LPR>), grid, kBlock, stream, rows, W, w1, out, m, \
              Nl, err);                                                                      \
    break;
    CASE(1) CASE(2) CASE(4) CASE(8) CASE(16) CASE(32) CASE(64)
#undef CASE
    default: return MI_ERR_UNSUPPORTED;
  }
  return launch_status();
}

int mi_unpack_rows(const float *packed, float *vals, float *lin, int64_t m, int32_t D, void *stream) {
  if (m < 0 || D <= 0) return MI_ERR_INVALID_ARG;
  if (m == 0) return MI_OK;
  if (!packed || !vals || !lin) return MI_ERR_INVALID_ARG;
  if (!vec_ok(D) || !aligned16(packed) || !aligned16(vals)) return MI_ERR_UNSUPPORTED;
  const int64_t rows_per_wave = kWave / (D / 4);
  const int grid = grid_for_waves((m + rows_per_wave - 1) / rows_per_wave);
  switch (D / 4) {
#define CASE(LPR)                                                                                 \
  case LPR:                                                                                       \
    MI_LAUNCH("unpack_rows", (k_unpack_rows<LPR>), grid, kBlock, stream, packed, vals, lin, m);  \
    break;
    CASE(1) CASE(2) CASE(4) CASE(8) CASE(16) CASE(32) CASE(64)
#undef CASE
    default: return MI_ERR_UNSUPPORTED;
  }
  return launch_status();
}

}  // extern "C"
