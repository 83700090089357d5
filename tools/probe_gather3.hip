// probe_gather3.hip — round 3: table LAYOUTS for the gather+FM forward at B=4096, and what the clocks read.
// (tools/, not product code)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_gather3.hip -o tools/bin/probe_gather3
//   tools/bin/probe_gather3                 # dispatch-clock table + graph-wall table
//   rocprofv3 --kernel-trace --stats ... -- tools/bin/probe_gather3 --graph-only   # the profiler's clock on in-graph kernels
//
// Layouts of the two DeepFM tables (src/models/deepfm.py:47-51 of the reference: embedding [N,16] + fc [N,1]):
//   split   W fp32[N,16] (64-B rows) + w1 fp32[N]            — the reference's tensors; TWO random sectors per lookup
//   p80     one table fp32[N,20]: 16 embedding floats, w1, 3 pad (80-B rows, unaligned to 64-B sectors)
//   p128    one table fp32[N,32]: 16 embedding floats, w1, 15 pad (128-B rows = one cache line per lookup)
// All three hold the same logical values, so every forward variant must produce the same emb / y_fm (checked).
//
// Regimes: b2b (launches back to back, fresh ids each), step (~100 MB of unrelated streaming between launches),
// cold (512 MB fill between launches).  Clocks: the dispatch's own begin/end events (eager launches), and WALL time per
// kernel inside a replayed hipGraph (b2b: 64 copies; step: (thrash + kernel) pairs minus the thrash alone).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kWave = 64, kBlock = 256, kWPB = 4;
static const int64_t DIMS[26] = {1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194,
                                 27, 14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572};
constexpr int F = 26, D = 16;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void st4nt(float *p, float4 v) {
  __builtin_nontemporal_store(v.x, p); __builtin_nontemporal_store(v.y, p + 1);
  __builtin_nontemporal_store(v.z, p + 2); __builtin_nontemporal_store(v.w, p + 3);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
template <int LO>
__device__ __forceinline__ float4 slot_sum(float4 v) {
#pragma unroll
  for (int m = LO; m < kWave; m <<= 1) {
    v.x += __shfl_xor(v.x, m); v.y += __shfl_xor(v.y, m); v.z += __shfl_xor(v.z, m); v.w += __shfl_xor(v.w, m);
  }
  return v;
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

__device__ __forceinline__ float hval(int64_t row, int c) {
  uint64_t h = (uint64_t)(row * 17 + c) * 0x9E3779B97F4A7C15ull;
  h ^= h >> 31; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 29;
  return (float)(h & 0xFFFF) * (1.0f / 65536.0f) - 0.5f;
}
// table[row*LD + c] = value(row, c) for c < 17 (c = 16: the first-order weight), 0 in the padding
__global__ void k_fill_table(float *t, int64_t N, int LD, int ncol) {
  const int64_t total = N * LD;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / LD;
    const int c = (int)(i % LD);
    t[i] = c < ncol ? hval(row, c) : 0.f;
  }
}
__global__ void k_fill_w1(float *t, int64_t N) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) t[i] = hval(i, 16);
}
__global__ void k_fill_hash(float *p, int64_t n, uint32_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (float)(h & 0xFFFF) * (1.0f / 65536.0f) - 0.5f;
  }
}
__global__ void k_fill_ids(int64_t *x, const int64_t *dims, int64_t B, uint32_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B * F; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
    h ^= h >> 31; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 29;
    x[i] = (int64_t)(h % (uint64_t)dims[f]);
  }
}
__global__ void k_stream(const float4 *__restrict__ a, float4 *__restrict__ b, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = a[i]; v.x += 1.f; b[i] = v;
  }
}
__global__ __launch_bounds__(kBlock) void k_empty(int *sink) {
  if (threadIdx.x == 1023) sink[0] = 1;
}
// checksum of a float buffer (order-independent enough for an equality check between variants: integer sum of the bits)
__global__ void k_checksum(const uint32_t *p, int64_t n, unsigned long long *out) {
  unsigned long long s = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += p[i];
  atomicAdd(out, s);
}

// ------------------------------------------------------------------------------------------------ forward variants
// A: 4 lanes per row (float4 each) + lane q==0 fetches the first-order weight with a second (4-B) load.
//    split: W + row*16, w1 + row.   packed: T + row*LD, T + row*LD + 16   (the product kernel's ldw / ldw1 form)
template <int LD, bool NT>
__global__ __launch_bounds__(kBlock) void k_fwd_A(const int64_t *__restrict__ idx, const int64_t *__restrict__ offsets,
                                                  const float *__restrict__ W, const float *__restrict__ w1, int64_t ld1,
                                                  float bias, float *__restrict__ emb, float *__restrict__ yfm,
                                                  int64_t *__restrict__ rows_out, int64_t B, int64_t N) {
  constexpr int LPR = 4, RS = 16, NIT = 2;
  const int lane = threadIdx.x & 63, q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWPB + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * kWPB;
  for (int64_t b = wave0; b < B; b += nw) {
    int64_t row[NIT];
    float4 v[NIT];
    float l[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = r + k * RS;
      row[k] = f < F ? idx[b * F + f] + offsets[f] : -1;
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const bool ok = (uint64_t)row[k] < (uint64_t)N;
      v[k] = ok ? ld4(W + row[k] * LD + q * 4) : make_float4(0, 0, 0, 0);
      l[k] = (ok && q == 0) ? w1[row[k] * ld1] : 0.f;
    }
    float4 S = make_float4(0, 0, 0, 0);
    float ss = 0.f, lin = 0.f;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = r + k * RS;
      if (f < F) {
        if (NT) st4nt(emb + (b * F + f) * D + q * 4, v[k]); else st4(emb + (b * F + f) * D + q * 4, v[k]);
        if (q == 0) rows_out[b * F + f] = row[k];
      }
      S.x += v[k].x; S.y += v[k].y; S.z += v[k].z; S.w += v[k].w;
      ss += dot4(v[k], v[k]);
      lin += l[k];
    }
    S = slot_sum<LPR>(S);
    float t = (r == 0 ? dot4(S, S) : 0.f) - ss;
    t = wave_sum(0.5f * t + lin);
    if (lane == 0) yfm[b] = t + bias;
  }
}

// B: 128-B rows, 8 lanes per row slot, lanes q<4 take the embedding floats, lane q==4 the float4 that starts with w1,
//    q>4 idle: ONE load instruction per 8 rows brings both tables' data (4 instructions per sample).
template <bool NT>
__global__ __launch_bounds__(kBlock) void k_fwd_B(const int64_t *__restrict__ idx, const int64_t *__restrict__ offsets,
                                                  const float *__restrict__ T, float bias, float *__restrict__ emb,
                                                  float *__restrict__ yfm, int64_t *__restrict__ rows_out, int64_t B, int64_t N) {
  constexpr int LPR = 8, RS = 8, NIT = 4, LD = 32;
  const int lane = threadIdx.x & 63, q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWPB + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * kWPB;
  for (int64_t b = wave0; b < B; b += nw) {
    int64_t row[NIT];
    float4 v[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = r + k * RS;
      row[k] = f < F ? idx[b * F + f] + offsets[f] : -1;
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const bool ok = (uint64_t)row[k] < (uint64_t)N && q < 5;
      v[k] = ok ? ld4(T + row[k] * LD + q * 4) : make_float4(0, 0, 0, 0);
    }
    float4 S = make_float4(0, 0, 0, 0);
    float ss = 0.f, lin = 0.f;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = r + k * RS;
      if (f < F) {
        if (q < 4) { if (NT) st4nt(emb + (b * F + f) * D + q * 4, v[k]); else st4(emb + (b * F + f) * D + q * 4, v[k]); }
        if (q == 0) rows_out[b * F + f] = row[k];
      }
      if (q < 4) {
        S.x += v[k].x; S.y += v[k].y; S.z += v[k].z; S.w += v[k].w;
        ss += dot4(v[k], v[k]);
      } else {
        lin += v[k].x;      // q == 4: first float of the chunk is w1; q > 4 loaded zeros
      }
    }
    S = slot_sum<LPR>(S);     // lanes q >= 4 carry zeros in S
    float t = (r == 0 ? dot4(S, S) : 0.f) - ss;
    t = wave_sum(0.5f * t + lin);
    if (lane == 0) yfm[b] = t + bias;
  }
}

// C: 80-B rows, 5 lanes per row (float4 each: 4 embedding chunks + the chunk that starts with w1), 12 rows per
//    instruction (lanes 60..63 idle), 3 instructions per sample.
template <bool NT>
__global__ __launch_bounds__(kBlock) void k_fwd_C(const int64_t *__restrict__ idx, const int64_t *__restrict__ offsets,
                                                  const float *__restrict__ T, float bias, float *__restrict__ emb,
                                                  float *__restrict__ yfm, int64_t *__restrict__ rows_out, int64_t B, int64_t N) {
  constexpr int LPR = 5, RS = 12, NIT = 3, LD = 20;
  const int lane = threadIdx.x & 63, q = lane % LPR, r = lane / LPR;    // r = 12 for lanes 60..63 (idle)
  const int64_t wave0 = (int64_t)blockIdx.x * kWPB + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * kWPB;
  for (int64_t b = wave0; b < B; b += nw) {
    int64_t row[NIT];
    float4 v[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = r + k * RS;
      row[k] = (r < RS && f < F) ? idx[b * F + f] + offsets[f] : -1;
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const bool ok = (uint64_t)row[k] < (uint64_t)N;
      v[k] = ok ? ld4(T + row[k] * LD + q * 4) : make_float4(0, 0, 0, 0);
    }
    float4 S = make_float4(0, 0, 0, 0);
    float ss = 0.f, lin = 0.f;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = r + k * RS;
      if (r < RS && f < F) {
        if (q < 4) { if (NT) st4nt(emb + (b * F + f) * D + q * 4, v[k]); else st4(emb + (b * F + f) * D + q * 4, v[k]); }
        if (q == 0) rows_out[b * F + f] = row[k];
      }
      if (q < 4) {
        S.x += v[k].x; S.y += v[k].y; S.z += v[k].z; S.w += v[k].w;
        ss += dot4(v[k], v[k]);
      } else {
        lin += v[k].x;
      }
    }
    if (q == 4) S = make_float4(0, 0, 0, 0);
    // sum over the 12 row slots (lanes r*5 + q): 8 -> 4 -> 2 -> 1, result in the lanes of r == 0
#pragma unroll
    for (int step = 8; step >= 1; step >>= 1) {
      const float4 o = make_float4(__shfl_down(S.x, step * LPR), __shfl_down(S.y, step * LPR),
                                   __shfl_down(S.z, step * LPR), __shfl_down(S.w, step * LPR));
      const bool take = r < step && r + step < RS;
      if (take) { S.x += o.x; S.y += o.y; S.z += o.z; S.w += o.w; }
    }
    float t = ((r == 0 && q < 4) ? dot4(S, S) : 0.f) - ss;
    t = wave_sum(0.5f * t + lin);
    if (lane == 0) yfm[b] = t + bias;
  }
}

// D: as A on a packed table, but the ids of the sample arrive by ONE coalesced load (lane l < F loads idx[b,l]) and are
//    handed to the row slots by shuffles (one dependent vector-memory instruction for the ids instead of two).
template <int LD, bool NT>
__global__ __launch_bounds__(kBlock) void k_fwd_D(const int64_t *__restrict__ idx, const int64_t *__restrict__ offsets,
                                                  const float *__restrict__ W, const float *__restrict__ w1, int64_t ld1,
                                                  float bias, float *__restrict__ emb, float *__restrict__ yfm,
                                                  int64_t *__restrict__ rows_out, int64_t B, int64_t N) {
  constexpr int LPR = 4, RS = 16, NIT = 2;
  const int lane = threadIdx.x & 63, q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWPB + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * kWPB;
  const int64_t myoff = lane < F ? offsets[lane] : 0;
  for (int64_t b = wave0; b < B; b += nw) {
    const int64_t mine = lane < F ? idx[b * F + lane] + myoff : -1;
    if (lane < F) rows_out[b * F + lane] = mine;
    int64_t row[NIT];
    float4 v[NIT];
    float l[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = r + k * RS;
      const int lo = __shfl((int)(uint32_t)mine, f & 63), hi = __shfl((int)(mine >> 32), f & 63);
      row[k] = f < F ? (((int64_t)hi << 32) | (uint32_t)lo) : -1;
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const bool ok = (uint64_t)row[k] < (uint64_t)N;
      v[k] = ok ? ld4(W + row[k] * LD + q * 4) : make_float4(0, 0, 0, 0);
      l[k] = (ok && q == 0) ? w1[row[k] * ld1] : 0.f;
    }
    float4 S = make_float4(0, 0, 0, 0);
    float ss = 0.f, lin = 0.f;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = r + k * RS;
      if (f < F) { if (NT) st4nt(emb + (b * F + f) * D + q * 4, v[k]); else st4(emb + (b * F + f) * D + q * 4, v[k]); }
      S.x += v[k].x; S.y += v[k].y; S.z += v[k].z; S.w += v[k].w;
      ss += dot4(v[k], v[k]);
      lin += l[k];
    }
    S = slot_sum<LPR>(S);
    float t = (r == 0 ? dot4(S, S) : 0.f) - ss;
    t = wave_sum(0.5f * t + lin);
    if (lane == 0) yfm[b] = t + bias;
  }
}

// backward (row form), as the product kernel
__global__ __launch_bounds__(kBlock) void k_bwd(const float *__restrict__ emb, const float *__restrict__ g_y,
                                                const float *__restrict__ g_emb, float *__restrict__ gvals,
                                                float *__restrict__ g1vals, int64_t B) {
  constexpr int LPR = 4, RS = 16, NIT = 2;
  const int lane = threadIdx.x & 63, q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWPB + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * kWPB;
  const float4 z = make_float4(0, 0, 0, 0);
  for (int64_t b = wave0; b < B; b += nw) {
    float4 e[NIT], ge[NIT];
    const float gy = g_y[b];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = r + k * RS;
      const int64_t o = (b * F + f) * D + q * 4;
      e[k] = f < F ? ld4(emb + o) : z;
      ge[k] = f < F ? ld4(g_emb + o) : z;
    }
    float4 S = z;
#pragma unroll
    for (int k = 0; k < NIT; ++k) { S.x += e[k].x; S.y += e[k].y; S.z += e[k].z; S.w += e[k].w; }
    S = slot_sum<LPR>(S);
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = r + k * RS;
      if (f < F) {
        float4 o4;
        o4.x = ge[k].x + gy * (S.x - e[k].x); o4.y = ge[k].y + gy * (S.y - e[k].y);
        o4.z = ge[k].z + gy * (S.z - e[k].z); o4.w = ge[k].w + gy * (S.w - e[k].w);
        st4(gvals + (b * F + f) * D + q * 4, o4);
        if (q == 0) g1vals[b * F + f] = gy;
      }
    }
  }
}

// backward + the bias gradient (sum of g_y) in the same launch.  MODE 0: workgroup 0 sums it, then does its rows (r02);
// 1: one EXTRA workgroup, the last one; 2: one extra workgroup, the FIRST one; 3: extra first workgroup, float4 loads
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_bwd_bias(const float *__restrict__ emb, const float *__restrict__ g_y,
                                                     const float *__restrict__ g_emb, float *__restrict__ gvals,
                                                     float *__restrict__ g1vals, float *__restrict__ gbias, int64_t B) {
  constexpr int LPR = 4, RS = 16, NIT = 2;
  __shared__ float part[kWPB];
  const int nblk = MODE == 0 ? gridDim.x : gridDim.x - 1;
  const bool bias_blk = MODE == 0 ? blockIdx.x == 0 : (MODE == 1 ? (int)blockIdx.x == nblk : blockIdx.x == 0);
  if (bias_blk) {
    float s = 0.f;
    if (MODE == 3) {
      for (int64_t b = threadIdx.x * 4; b + 3 < B; b += kBlock * 4) { const float4 v = ld4(g_y + b); s += (v.x + v.y) + (v.z + v.w); }
      for (int64_t b = (B & ~3ll) + threadIdx.x; b < B; b += kBlock) s += g_y[b];
    } else {
      for (int64_t b = threadIdx.x; b < B; b += kBlock) s += g_y[b];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) gbias[0] = (part[0] + part[1]) + (part[2] + part[3]);
    if (MODE != 0) return;
  }
  const int blk = (MODE >= 2) ? blockIdx.x - 1 : blockIdx.x;
  const int lane = threadIdx.x & 63, q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blk * kWPB + (threadIdx.x >> 6), nw = (int64_t)nblk * kWPB;
  const float4 z = make_float4(0, 0, 0, 0);
  for (int64_t b = wave0; b < B; b += nw) {
    float4 e[NIT], ge[NIT];
    const float gy = g_y[b];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = r + k * RS;
      const int64_t o = (b * F + f) * D + q * 4;
      e[k] = f < F ? ld4(emb + o) : z;
      ge[k] = f < F ? ld4(g_emb + o) : z;
    }
    float4 S = z;
#pragma unroll
    for (int k = 0; k < NIT; ++k) { S.x += e[k].x; S.y += e[k].y; S.z += e[k].z; S.w += e[k].w; }
    S = slot_sum<LPR>(S);
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = r + k * RS;
      if (f < F) {
        float4 o4;
        o4.x = ge[k].x + gy * (S.x - e[k].x); o4.y = ge[k].y + gy * (S.y - e[k].y);
        o4.z = ge[k].z + gy * (S.z - e[k].z); o4.w = ge[k].w + gy * (S.w - e[k].w);
        st4(gvals + (b * F + f) * D + q * 4, o4);
        if (q == 0) g1vals[b * F + f] = gy;
      }
    }
  }
}

// backward, first-order values written by lanes < F as ONE coalesced store per sample (not 16 + 10 sparse-lane stores)
__global__ __launch_bounds__(kBlock) void k_bwd_g1c(const float *__restrict__ emb, const float *__restrict__ g_y,
                                                    const float *__restrict__ g_emb, float *__restrict__ gvals,
                                                    float *__restrict__ g1vals, int64_t B) {
  constexpr int LPR = 4, RS = 16, NIT = 2;
  const int lane = threadIdx.x & 63, q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWPB + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * kWPB;
  const float4 z = make_float4(0, 0, 0, 0);
  for (int64_t b = wave0; b < B; b += nw) {
    float4 e[NIT], ge[NIT];
    const float gy = g_y[b];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = r + k * RS;
      const int64_t o = (b * F + f) * D + q * 4;
      e[k] = f < F ? ld4(emb + o) : z;
      ge[k] = f < F ? ld4(g_emb + o) : z;
    }
    if (lane < F) g1vals[b * F + lane] = gy;
    float4 S = z;
#pragma unroll
    for (int k = 0; k < NIT; ++k) { S.x += e[k].x; S.y += e[k].y; S.z += e[k].z; S.w += e[k].w; }
    S = slot_sum<LPR>(S);
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int f = r + k * RS;
      if (f < F) {
        float4 o4;
        o4.x = ge[k].x + gy * (S.x - e[k].x); o4.y = ge[k].y + gy * (S.y - e[k].y);
        o4.z = ge[k].z + gy * (S.z - e[k].z); o4.w = ge[k].w + gy * (S.w - e[k].w);
        st4(gvals + (b * F + f) * D + q * 4, o4);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ harness
struct Stat { double avg, med, mn; };
static Stat stats(std::vector<float> v) {
  std::sort(v.begin(), v.end());
  double s = 0; for (float x : v) s += x;
  return {s / v.size() * 1e3, v[v.size() / 2] * 1e3, v[0] * 1e3};
}

int main(int argc, char **argv) {
  bool graph_only = false, no_graph = false;
  int64_t B = 4096;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--graph-only")) graph_only = true;
    else if (!strcmp(argv[i], "--no-graph")) no_graph = true;
    else B = atoll(argv[i]);
  }
  const int reps = 40;
  int64_t N = 0, offs_h[F];
  for (int f = 0; f < F; ++f) { offs_h[f] = N; N += DIMS[f]; }
  float *W, *w1, *T80, *T128, *emb, *emb_ref, *yfm, *yfm_ref, *gemb, *gy, *gvals, *g1, *junkA, *junkB;
  int64_t *ids[16], *rows_out, *offs, *dims;
  int *sink;
  unsigned long long *cks;
  const int64_t junk_floats = 128ll << 20;     // 512 MB each
  CK(hipMalloc(&W, N * D * 4)); CK(hipMalloc(&w1, N * 4));
  CK(hipMalloc(&T80, N * 20 * 4)); CK(hipMalloc(&T128, N * 32 * 4));
  CK(hipMalloc(&emb, B * F * D * 4)); CK(hipMalloc(&emb_ref, B * F * D * 4));
  CK(hipMalloc(&yfm, B * 4)); CK(hipMalloc(&yfm_ref, B * 4)); CK(hipMalloc(&gemb, B * F * D * 4));
  CK(hipMalloc(&gy, B * 4)); CK(hipMalloc(&gvals, B * F * D * 4)); CK(hipMalloc(&g1, B * F * 4));
  CK(hipMalloc(&junkA, junk_floats * 4)); CK(hipMalloc(&junkB, junk_floats * 4));
  CK(hipMalloc(&rows_out, B * F * 8)); CK(hipMalloc(&offs, F * 8)); CK(hipMalloc(&dims, F * 8)); CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&cks, 64));
  CK(hipMemcpy(offs, offs_h, F * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dims, DIMS, F * 8, hipMemcpyHostToDevice));
  k_fill_table<<<4096, 256>>>(W, N, 16, 16); k_fill_w1<<<2048, 256>>>(w1, N);
  k_fill_table<<<4096, 256>>>(T80, N, 20, 17); k_fill_table<<<4096, 256>>>(T128, N, 32, 17);
  k_fill_hash<<<2048, 256>>>(gemb, B * F * D, 3u); k_fill_hash<<<64, 256>>>(gy, B, 4u);
  k_fill_hash<<<2048, 256>>>(junkA, junk_floats, 5u);
  for (int i = 0; i < 16; ++i) { CK(hipMalloc(&ids[i], B * F * 8)); k_fill_ids<<<512, 256>>>(ids[i], dims, B, 100u + i); }
  CK(hipDeviceSynchronize());
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t ea[reps], eb[reps];
  for (int i = 0; i < reps; ++i) { CK(hipEventCreate(&ea[i])); CK(hipEventCreate(&eb[i])); }
  const int g1k = (int)((B + 3) / 4);
  const double fb = (12.0 * F + 8.0 * F * D + 4) * B, bb = (12.0 * F + 12.0 * F * D + 4) * B;
  printf("B=%lld F=%d D=%d N=%lld  fwd alg bytes %.0f  bwd alg bytes %.0f\n", (long long)B, F, D, (long long)N, fb, bb);

  using Enq = std::function<void(int)>;                 // enqueue variant for id batch i on `st`
  struct Variant { const char *name; Enq enq; std::function<void(int, hipEvent_t, hipEvent_t)> timed; };
  std::vector<Variant> V;
#define ADD(NAME, KERN, ...)                                                                                      \
  V.push_back({NAME,                                                                                               \
               [&](int i) { (void)i; hipLaunchKernelGGL(KERN, dim3(g1k), dim3(kBlock), 0, st, __VA_ARGS__); },      \
               [&](int i, hipEvent_t a, hipEvent_t b) { (void)i;                                                    \
                 hipExtLaunchKernelGGL(KERN, dim3(g1k), dim3(kBlock), 0, st, a, b, 0, __VA_ARGS__); }})
#define IDS (const int64_t *)ids[i & 15], (const int64_t *)offs
  ADD("A split    (product today)", (k_fwd_A<16, false>), IDS, (const float *)W, (const float *)w1, (int64_t)1, 0.1f, emb, yfm, rows_out, B, N);
  ADD("A p80      4 lanes + w1 load", (k_fwd_A<20, false>), IDS, (const float *)T80, (const float *)(T80 + 16), (int64_t)20, 0.1f, emb, yfm, rows_out, B, N);
  ADD("A p128     4 lanes + w1 load", (k_fwd_A<32, false>), IDS, (const float *)T128, (const float *)(T128 + 16), (int64_t)32, 0.1f, emb, yfm, rows_out, B, N);
  ADD("B p128     5 of 8 lanes, 1 load", (k_fwd_B<false>), IDS, (const float *)T128, 0.1f, emb, yfm, rows_out, B, N);
  ADD("C p80      5 lanes, 1 load", (k_fwd_C<false>), IDS, (const float *)T80, 0.1f, emb, yfm, rows_out, B, N);
  ADD("D p128     A + ids by shuffle", (k_fwd_D<32, false>), IDS, (const float *)T128, (const float *)(T128 + 16), (int64_t)32, 0.1f, emb, yfm, rows_out, B, N);
  ADD("A p128 nt  nt emb stores", (k_fwd_A<32, true>), IDS, (const float *)T128, (const float *)(T128 + 16), (int64_t)32, 0.1f, emb, yfm, rows_out, B, N);
  ADD("B p128 nt  nt emb stores", (k_fwd_B<true>), IDS, (const float *)T128, 0.1f, emb, yfm, rows_out, B, N);
  ADD("C p80 nt   nt emb stores", (k_fwd_C<true>), IDS, (const float *)T80, 0.1f, emb, yfm, rows_out, B, N);
  ADD("E p128     D + nt (product r03)", (k_fwd_D<32, true>), IDS, (const float *)T128, (const float *)(T128 + 16), (int64_t)32, 0.1f, emb, yfm, rows_out, B, N);
  ADD("E split    D + nt (product r03)", (k_fwd_D<16, true>), IDS, (const float *)W, (const float *)w1, (int64_t)1, 0.1f, emb, yfm, rows_out, B, N);
  const size_t n_fwd = V.size();
  const int gcap = std::min(g1k, 2048);
  ADD("bwd rows   no bias grad", k_bwd, (const float *)emb, (const float *)gy, (const float *)gemb, gvals, g1, B);
  V.push_back({"bwd g1 coalesced, no bias",
               [&](int) { hipLaunchKernelGGL(k_bwd_g1c, dim3(gcap), dim3(kBlock), 0, st, (const float *)emb, (const float *)gy, (const float *)gemb, gvals, g1, B); },
               [&](int, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_bwd_g1c, dim3(gcap), dim3(kBlock), 0, st, a, b, 0, (const float *)emb, (const float *)gy, (const float *)gemb, gvals, g1, B); }});
  V.push_back({"bwd rows grid<=2048, no bias",
               [&](int) { hipLaunchKernelGGL(k_bwd, dim3(gcap), dim3(kBlock), 0, st, (const float *)emb, (const float *)gy, (const float *)gemb, gvals, g1, B); },
               [&](int, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_bwd, dim3(gcap), dim3(kBlock), 0, st, a, b, 0, (const float *)emb, (const float *)gy, (const float *)gemb, gvals, g1, B); }});
#define ADDB(NAME, MODE, EXTRA)                                                                                                   \
  V.push_back({NAME,                                                                                                               \
               [&](int) { hipLaunchKernelGGL((k_bwd_bias<MODE>), dim3(gcap + EXTRA), dim3(kBlock), 0, st, (const float *)emb, (const float *)gy, (const float *)gemb, gvals, g1, (float *)sink + 8, B); },   \
               [&](int, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((k_bwd_bias<MODE>), dim3(gcap + EXTRA), dim3(kBlock), 0, st, a, b, 0, (const float *)emb, (const float *)gy, (const float *)gemb, gvals, g1, (float *)sink + 8, B); }})
  ADDB("bwd bias in wg 0 + rows (r02)", 0, 0);
  ADDB("bwd bias extra LAST wg", 1, 1);
  ADDB("bwd bias extra FIRST wg", 2, 1);
  ADDB("bwd bias extra FIRST wg float4", 3, 1);
  ADD("empty      grid=B/4", k_empty, sink);

  // ---- equality of the forward variants (same ids batch 0) ----
  {
    auto sums = [&](unsigned long long *out3) {
      CK(hipMemsetAsync(cks, 0, 24, st));
      k_checksum<<<256, 256, 0, st>>>((const uint32_t *)emb, B * F * D, cks);
      k_checksum<<<64, 256, 0, st>>>((const uint32_t *)yfm, B, cks + 1);
      k_checksum<<<64, 256, 0, st>>>((const uint32_t *)rows_out, B * F * 2, cks + 2);
      CK(hipMemcpyAsync(out3, cks, 24, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
    };
    unsigned long long ref[3], got[3];
    std::vector<float> yr(B), yg(B);
    for (size_t v = 0; v < n_fwd; ++v) {
      CK(hipMemsetAsync(emb, 0xFF, B * F * D * 4, st)); CK(hipMemsetAsync(yfm, 0xFF, B * 4, st));
      V[v].enq(0);
      sums(v == 0 ? ref : got);
      CK(hipMemcpy((v == 0 ? yr : yg).data(), yfm, B * 4, hipMemcpyDeviceToHost));
      if (v > 0) {
        double md = 0;
        for (int64_t i = 0; i < B; ++i) md = std::max(md, (double)fabsf(yr[i] - yg[i]));
        printf("check %-32s emb %s rows %s  max|dy_fm| %.3g\n", V[v].name, got[0] == ref[0] ? "same" : "DIFFERENT",
               got[2] == ref[2] ? "same" : "DIFFERENT", md);
      }
    }
  }

  enum Regime { B2B, STEP, COLD };
  const char *rname[] = {"b2b", "step", "cold"};
  auto between = [&](int regime) {
    if (regime == STEP) k_stream<<<2048, 256, 0, st>>>((const float4 *)junkA, (float4 *)junkB, (48ll << 20) / 16);
    if (regime == COLD) k_stream<<<2048, 256, 0, st>>>((const float4 *)junkA, (float4 *)junkB, junk_floats / 4);
  };
  if (!graph_only) {
    printf("---- dispatch clock (begin/end events of eager launches), us\n");
    for (auto &v : V) {
      const bool is_fwd = &v - &V[0] < (long)n_fwd;
      const double bytes = is_fwd ? fb : (strstr(v.name, "bwd") ? bb : 0);
      for (int regime : {B2B, STEP, COLD}) {
        for (int i = 0; i < 5; ++i) { between(regime); v.enq(i); }
        CK(hipStreamSynchronize(st));
        for (int i = 0; i < reps; ++i) { between(regime); v.timed(i, ea[i], eb[i]); }
        CK(hipStreamSynchronize(st));
        std::vector<float> t(reps);
        for (int i = 0; i < reps; ++i) CK(hipEventElapsedTime(&t[i], ea[i], eb[i]));
        Stat s = stats(t);
        printf("%-34s %-5s avg %6.2f  med %6.2f  min %6.2f", v.name, rname[regime], s.avg, s.med, s.mn);
        if (bytes > 0) printf("   frac(med) %.3f", bytes / (s.med * 1e-6) / 8e12);
        printf("\n");
        fflush(stdout);
      }
    }
  }
  if (!no_graph) {
    printf("---- WALL per kernel inside a replayed hipGraph, us (b2b: 64 copies; step/cold: (thrash + kernel) pairs minus the thrash alone)\n");
    auto graph_time = [&](int nk, const Enq &enqueue) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int i = 0; i < nk; ++i) enqueue(i);
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      const int R = 20;
      CK(hipEventRecord(a, st));
      for (int i = 0; i < R; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(b, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
      CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
      return (double)ms * 1e3 / (R * nk);
    };
    const double thr_step = graph_time(16, [&](int) { between(STEP); });
    const double thr_cold = graph_time(8, [&](int) { between(COLD); });
    printf("thrash alone: step %.2f us, cold %.2f us\n", thr_step, thr_cold);
    for (auto &v : V) {
      const bool is_fwd = &v - &V[0] < (long)n_fwd;
      const double bytes = is_fwd ? fb : (strstr(v.name, "bwd") ? bb : 0);
      const double t0 = graph_time(64, v.enq);
      const double t1 = graph_time(32, [&](int i) { between(STEP); v.enq(i); }) - thr_step;
      const double t2 = graph_time(8, [&](int i) { between(COLD); v.enq(i); }) - thr_cold;
      printf("%-34s b2b %6.2f  step %6.2f  cold %6.2f", v.name, t0, t1, t2);
      if (bytes > 0) printf("   frac b2b %.3f step %.3f", bytes / (t0 * 1e-6) / 8e12, bytes / (t1 * 1e-6) / 8e12);
      printf("\n");
      fflush(stdout);
    }
    // the pair as a step sees it: thrash, fwd, thrash, bwd
    for (size_t v : {size_t(0), size_t(2), n_fwd - 2, n_fwd - 1}) {
      const double t = graph_time(16, [&](int i) { between(STEP); V[v].enq(i); between(STEP); V[n_fwd].enq(i); }) - 2 * thr_step;
      printf("pair: %-28s + bwd, step regime: %6.2f us  frac %.3f\n", V[v].name, t, (fb + bb) / (t * 1e-6) / 8e12);
    }
  }
  CK(hipDeviceSynchronize());
  return 0;
}
