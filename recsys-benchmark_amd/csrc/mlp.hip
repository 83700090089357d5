// mlp.hip — the memory-bound part of the MLP tail: BatchNorm1d (training or
// eval) + ReLU + Dropout after each Linear of DeepFM._deep_branch / DCN._dnn
// (src/models/deepfm.py:53-66, src/models/dcn.py:56-66), forward and backward.  The Linear
// contractions stay on hipBLASLt/rocBLAS through PyTorch; these kernels replace the ~10 separate
// elementwise / reduction launches per layer that dominated the step (rocprof r01: PyTorch's
// batch_norm_collect_statistics 33 us and batch_norm_backward_reduce 31 us per layer at
// [4096, 400]) with two passes over the activation each way.
//
//   forward   stats:  s1[n] = sum_m (z - c_n), s2[n] = sum_m (z - c_n)^2, c_n = z[0,n] (shifted sums:
//                     single pass without the catastrophic cancellation of E[z^2] - E[z]^2)
//             apply:  mean = c + s1/M, var = s2/M - (s1/M)^2 (biased, as F.batch_norm normalises);
//                     y = relu(gamma*(z-mean)*rstd + beta) * keep/(1-p); running stats updated with
//                     `momentum` and the unbiased variance; keep-mask (1 byte) saved for backward
//   backward  reduce: dbeta[n] = sum_m dyh, dgamma[n] = sum_m dyh*zh,  dyh = dy*keep/(1-p)*[pre>0]
//             apply:  dz = gamma*rstd*(dyh - dbeta/M - zh*dgamma/M)   (training);  gamma*rstd*dyh (eval)
// Dropout uses a counter-based generator: splitmix64(seed, element index), seed read from a device
// word so the launch is graph-capture safe.
#include "common.hpp"

namespace {
using namespace mi;

__device__ __forceinline__ uint32_t rng32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((z ^ (z >> 31)) >> 16);
}

// Column reductions, vectorised: a thread owns 4 adjacent columns (one float4 per row), a wave covers 256
// columns = 1 KiB per row and instruction, the 4 waves of a workgroup take 4 different rows; rows are
// strided over blockIdx.y with 4 rows in flight per thread.  Partial sums meet in LDS, one float atomic per
// column per workgroup.  (N % 4 == 0 path; otherwise the scalar form below.)
constexpr int kColTile = 256;

__global__ __launch_bounds__(kBlock) void k_col_stats_v4(const float *__restrict__ Z, int ld, float *__restrict__ s1,
                                                         float *__restrict__ s2, int M, int N, int64_t *bump) {
  __shared__ float4 p1[4][64], p2[4][64];
  if (bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) bump[0] += 1;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n0 = blockIdx.x * kColTile + lane * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (n0 < N) {
    const float4 sh = ld4(Z + n0);
    const int stride = gridDim.y * 4;
    int m = blockIdx.y * 4 + w;
    for (; m + 3 * stride < M; m += 4 * stride) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = ld4(Z + (int64_t)(m + u * stride) * ld + n0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float dx = v[u].x - sh.x, dy = v[u].y - sh.y, dz = v[u].z - sh.z, dw = v[u].w - sh.w;
        a.x += dx; a.y += dy; a.z += dz; a.w += dw;
        b.x += dx * dx; b.y += dy * dy; b.z += dz * dz; b.w += dw * dw;
      }
    }
    for (; m < M; m += stride) {
      const float4 v = ld4(Z + (int64_t)m * ld + n0);
      const float dx = v.x - sh.x, dy = v.y - sh.y, dz = v.z - sh.z, dw = v.w - sh.w;
      a.x += dx; a.y += dy; a.z += dz; a.w += dw;
      b.x += dx * dx; b.y += dy * dy; b.z += dz * dz; b.w += dw * dw;
    }
  }
  p1[w][lane] = a;
  p2[w][lane] = b;
  __syncthreads();
  // flush: thread t owns column t of the tile, so every atomic wave-instruction is 64 CONTIGUOUS floats
  {
    const int t = threadIdx.x, n = blockIdx.x * kColTile + t;
    if (n < N) {
      const float *q1 = reinterpret_cast<const float *>(&p1[0][0]);
      const float *q2 = reinterpret_cast<const float *>(&p2[0][0]);
      atomicAdd(s1 + n, q1[t] + q1[256 + t] + q1[512 + t] + q1[768 + t]);
      atomicAdd(s2 + n, q2[t] + q2[256 + t] + q2[512 + t] + q2[768 + t]);
    }
  }
}

// grid (ceil(N/64), RB); block 256 = 64 columns x 4 row lanes   (any N)
__global__ __launch_bounds__(kBlock) void k_col_stats(const float *__restrict__ Z, int ld, float *__restrict__ s1,
                                                      float *__restrict__ s2, int M, int N, int64_t *bump) {
  __shared__ float p1[4][64], p2[4][64];
  // stream order puts this launch before every apply kernel of the pass that reads the word
  if (bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) bump[0] += 1;
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  float a = 0.f, b = 0.f;
  if (n < N) {
    const float shift = Z[n];
    const int stride = gridDim.y * 4;
    int m = blockIdx.y * 4 + rl;
    for (; m + 7 * stride < M; m += 8 * stride) {   // 8 independent row loads in flight
      float d[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) d[u] = Z[(int64_t)(m + u * stride) * ld + n] - shift;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a += d[u];
        b += d[u] * d[u];
      }
    }
    for (; m < M; m += stride) {
      const float d = Z[(int64_t)m * ld + n] - shift;
      a += d;
      b += d * d;
    }
  }
  p1[rl][c] = a;
  p2[rl][c] = b;
  __syncthreads();
  if (rl == 0 && n < N) {
    atomicAdd(s1 + n, p1[0][c] + p1[1][c] + p1[2][c] + p1[3][c]);
    atomicAdd(s2 + n, p2[0][c] + p2[1][c] + p2[2][c] + p2[3][c]);
  }
}

struct BnArgs {
  const float *Z;
  int ld;
  int M, N;
  int has_bn, training;
  const float *s1, *s2;          // training stats (shifted sums)
  const float *mean_offset;      // [N] added to the batch mean in the running_mean update only (nullable)
  const float *gamma, *beta;     // nullable = 1 / 0
  float *running_mean, *running_var;
  float momentum, eps;
  float p;                       // dropout probability (0 = none)
  const int64_t *seed;
  int64_t salt;
  float *Y;
  uint8_t *keep;                 // [M,N] when p > 0 and training
  float *save_mean, *save_rstd;  // [N]
  int64_t *num_batches_tracked;  // nullable: += 1 (training BatchNorm bookkeeping)
};

// VEC consecutive columns per thread; every stream (activation, mask, per-column parameters) is
// read/written as one VEC-wide access so a wave-instruction stays a contiguous run.
template <int VEC>
struct Vec {
  float v[VEC];
};
template <int VEC>
__device__ __forceinline__ Vec<VEC> ldv(const float *p) {
  Vec<VEC> r;
  if constexpr (VEC == 4) {
    const float4 t = ld4(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else {
    r.v[0] = p[0];
  }
  return r;
}
template <int VEC>
__device__ __forceinline__ Vec<VEC> ldv_or(const float *p, int64_t off, float dflt) {
  if (p) return ldv<VEC>(p + off);
  Vec<VEC> r;
#pragma unroll
  for (int j = 0; j < VEC; ++j) r.v[j] = dflt;
  return r;
}
template <int VEC>
__device__ __forceinline__ void stv(float *p, const Vec<VEC> &r) {
  if constexpr (VEC == 4) st4(p, make_float4(r.v[0], r.v[1], r.v[2], r.v[3]));
  else p[0] = r.v[0];
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void k_bn_relu_drop_fwd(BnArgs a) {
  const int nv = a.N / VEC;
  const int64_t total = (int64_t)a.M * nv;
  const bool drop = a.training && a.p > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - a.p) : 1.f;
  const uint32_t thresh = drop ? (uint32_t)((double)a.p * 4294967296.0) : 0u;  // keep iff rng >= p * 2^32
  const uint64_t seed = drop ? (uint64_t)(a.seed[0] + a.salt) : 0ull;
  const float invM = 1.f / (float)a.M;
  for (int64_t ev = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ev < total;
       ev += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(ev / nv), n0 = (int)(ev % nv) * VEC;
    const Vec<VEC> z = ldv<VEC>(a.Z + (int64_t)m * a.ld + n0);
    Vec<VEC> mean, rstd, g, b, y;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { mean.v[j] = 0.f; rstd.v[j] = 1.f; g.v[j] = 1.f; b.v[j] = 0.f; }
    if (a.has_bn) {
      g = ldv_or<VEC>(a.gamma, n0, 1.f);
      b = ldv_or<VEC>(a.beta, n0, 0.f);
      if (a.training) {
        const Vec<VEC> s1 = ldv<VEC>(a.s1 + n0), s2 = ldv<VEC>(a.s2 + n0), c = ldv<VEC>(a.Z + n0);
        Vec<VEC> var;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float d1 = s1.v[j] * invM;
          mean.v[j] = c.v[j] + d1;
          var.v[j] = fmaxf(s2.v[j] * invM - d1 * d1, 0.f);
          rstd.v[j] = rsqrtf(var.v[j] + a.eps);
        }
        if (ev == 0 && a.num_batches_tracked) a.num_batches_tracked[0] += 1;
        if (m == 0) {  // one thread per column group owns the bookkeeping
          stv<VEC>(a.save_mean + n0, mean);
          stv<VEC>(a.save_rstd + n0, rstd);
          if (a.running_mean) {
            const float ub = a.M > 1 ? (float)a.M / (float)(a.M - 1) : 1.f;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              const float mo = a.mean_offset ? a.mean_offset[n0 + j] : 0.f;
              a.running_mean[n0 + j] = (1.f - a.momentum) * a.running_mean[n0 + j] + a.momentum * (mean.v[j] + mo);
              a.running_var[n0 + j] = (1.f - a.momentum) * a.running_var[n0 + j] + a.momentum * var.v[j] * ub;
            }
          }
        }
      } else {
        mean = ldv<VEC>(a.running_mean + n0);
        const Vec<VEC> rv = ldv<VEC>(a.running_var + n0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) rstd.v[j] = rsqrtf(rv.v[j] + a.eps);
        if (m == 0) {
          stv<VEC>(a.save_mean + n0, mean);
          stv<VEC>(a.save_rstd + n0, rstd);
        }
      }
    }
    const int64_t e0 = (int64_t)m * a.N + n0;
    uint8_t kv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float t = g.v[j] * (z.v[j] - mean.v[j]) * rstd.v[j] + b.v[j];
      t = t > 0.f ? t : 0.f;
      kv[j] = 1;
      if (drop) {
        const bool k = rng32(seed, (uint64_t)(e0 + j)) >= thresh;
        kv[j] = k ? 1 : 0;
        t = k ? t * keep_scale : 0.f;
      }
      y.v[j] = t;
    }
    stv<VEC>(a.Y + e0, y);
    if (drop) {
      if constexpr (VEC == 4) *reinterpret_cast<uchar4 *>(a.keep + e0) = make_uchar4(kv[0], kv[1], kv[2], kv[3]);
      else a.keep[e0] = kv[0];
    }
  }
}

struct BnBwdArgs {
  const float *dY, *Z;           // dY NULL: the upstream gradient is rank-1, dY[m,n] = gvec[m] * wvec[n]
  const float *gvec, *wvec;      // (the backward of a following 1-output Linear, never materialised)
  int ld;
  int M, N;
  int has_bn, training;
  const uint8_t *keep;          // nullable
  float p;
  const float *gamma, *beta, *save_mean, *save_rstd;
  float *dbeta, *dgamma;        // [N] caller-zeroed (reduce) / read (apply)
  float *dZ;
};

__device__ __forceinline__ float bwd_dyh(const BnBwdArgs &a, int64_t e, int n, float z, float &zh) {
  float mean = 0.f, rstd = 1.f, g = 1.f, b = 0.f;
  if (a.has_bn) {
    mean = a.save_mean[n];
    rstd = a.save_rstd[n];
    g = a.gamma ? a.gamma[n] : 1.f;
    b = a.beta ? a.beta[n] : 0.f;
  }
  zh = (z - mean) * rstd;
  const float pre = g * zh + b;
  float d = a.dY ? a.dY[e] : a.gvec[e / a.N] * a.wvec[n];
  if (a.keep) d = a.keep[e] ? d * (1.f / (1.f - a.p)) : 0.f;
  return pre > 0.f ? d : 0.f;
}

// RANK1: the upstream gradient is gvec[m] * wvec[n] (compile-time, so that no branch sits between the loads)
template <bool RANK1>
__global__ __launch_bounds__(kBlock) void k_bn_bwd_reduce_v4(BnBwdArgs a) {
  __shared__ float4 p1[4][64], p2[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n0 = blockIdx.x * kColTile + lane * 4;
  float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), sg = sb;
  if (n0 < a.N) {
    float mean[4] = {0.f, 0.f, 0.f, 0.f}, rstd[4] = {1.f, 1.f, 1.f, 1.f}, g[4] = {1.f, 1.f, 1.f, 1.f}, bt[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.has_bn) {
      const float4 m4 = ld4(a.save_mean + n0), r4 = ld4(a.save_rstd + n0);
      mean[0] = m4.x; mean[1] = m4.y; mean[2] = m4.z; mean[3] = m4.w;
      rstd[0] = r4.x; rstd[1] = r4.y; rstd[2] = r4.z; rstd[3] = r4.w;
      if (a.gamma) { const float4 t = ld4(a.gamma + n0); g[0] = t.x; g[1] = t.y; g[2] = t.z; g[3] = t.w; }
      if (a.beta) { const float4 t = ld4(a.beta + n0); bt[0] = t.x; bt[1] = t.y; bt[2] = t.z; bt[3] = t.w; }
    }
    const float ks = a.keep ? 1.f / (1.f - a.p) : 1.f;
    float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (RANK1) w4 = ld4(a.wvec + n0);
    const int stride = gridDim.y * 4;
    float accb[4] = {0.f, 0.f, 0.f, 0.f}, accg[4] = {0.f, 0.f, 0.f, 0.f};
    for (int m = blockIdx.y * 4 + w; m < a.M; m += 2 * stride) {
      float4 z[2], dy[2];
      uchar4 k4[2];
      bool ok[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int mm = m + u * stride;
        ok[u] = mm < a.M;
        const int mr = ok[u] ? mm : m;
        z[u] = ld4(a.Z + (int64_t)mr * a.ld + n0);
        if constexpr (!RANK1) {
          dy[u] = ld4(a.dY + (int64_t)mr * a.N + n0);
        } else {
          const float gm = a.gvec[mr];
          dy[u] = make_float4(gm * w4.x, gm * w4.y, gm * w4.z, gm * w4.w);
        }
        k4[u] = a.keep ? *reinterpret_cast<const uchar4 *>(a.keep + (int64_t)mr * a.N + n0) : make_uchar4(1, 1, 1, 1);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (!ok[u]) continue;
        const float zz[4] = {z[u].x, z[u].y, z[u].z, z[u].w}, dd[4] = {dy[u].x, dy[u].y, dy[u].z, dy[u].w};
        const unsigned char kk[4] = {k4[u].x, k4[u].y, k4[u].z, k4[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float zh = (zz[j] - mean[j]) * rstd[j];
          float d = kk[j] ? dd[j] * ks : 0.f;
          d = (g[j] * zh + bt[j]) > 0.f ? d : 0.f;
          accb[j] += d;
          accg[j] += d * zh;
        }
      }
    }
    sb = make_float4(accb[0], accb[1], accb[2], accb[3]);
    sg = make_float4(accg[0], accg[1], accg[2], accg[3]);
  }
  p1[w][lane] = sb;
  p2[w][lane] = sg;
  __syncthreads();
  {
    const int t = threadIdx.x, n = blockIdx.x * kColTile + t;
    if (n < a.N) {
      const float *q1 = reinterpret_cast<const float *>(&p1[0][0]);
      const float *q2 = reinterpret_cast<const float *>(&p2[0][0]);
      atomicAdd(a.dbeta + n, q1[t] + q1[256 + t] + q1[512 + t] + q1[768 + t]);
      atomicAdd(a.dgamma + n, q2[t] + q2[256 + t] + q2[512 + t] + q2[768 + t]);
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_bn_bwd_reduce(BnBwdArgs a) {
  __shared__ float p1[4][64], p2[4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  float sb = 0.f, sg = 0.f;
  if (n < a.N) {
    const int stride = gridDim.y * 4;
    int m = blockIdx.y * 4 + rl;
    for (; m + 3 * stride < a.M; m += 4 * stride) {  // 4 x 3 independent loads in flight
      float zz[4], zh[4], d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) zz[u] = a.Z[(int64_t)(m + u * stride) * a.ld + n];
#pragma unroll
      for (int u = 0; u < 4; ++u) d[u] = bwd_dyh(a, (int64_t)(m + u * stride) * a.N + n, n, zz[u], zh[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        sb += d[u];
        sg += d[u] * zh[u];
      }
    }
    for (; m < a.M; m += stride) {
      float zh;
      const float d = bwd_dyh(a, (int64_t)m * a.N + n, n, a.Z[(int64_t)m * a.ld + n], zh);
      sb += d;
      sg += d * zh;
    }
  }
  p1[rl][c] = sb;
  p2[rl][c] = sg;
  __syncthreads();
  if (rl == 0 && n < a.N) {
    atomicAdd(a.dbeta + n, p1[0][c] + p1[1][c] + p1[2][c] + p1[3][c]);
    atomicAdd(a.dgamma + n, p2[0][c] + p2[1][c] + p2[2][c] + p2[3][c]);
  }
}

template <int VEC, bool RANK1>
__global__ __launch_bounds__(kBlock) void k_bn_bwd_apply(BnBwdArgs a) {
  const int nv = a.N / VEC;
  const int64_t total = (int64_t)a.M * nv;
  const float invM = 1.f / (float)a.M;
  const float keep_scale = a.keep ? 1.f / (1.f - a.p) : 1.f;
  for (int64_t ev = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ev < total;
       ev += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(ev / nv), n0 = (int)(ev % nv) * VEC;
    const int64_t e0 = (int64_t)m * a.N + n0;
    const Vec<VEC> z = ldv<VEC>(a.Z + (int64_t)m * a.ld + n0);
    Vec<VEC> dy;
    if constexpr (!RANK1) {
      dy = ldv<VEC>(a.dY + e0);
    } else {
      const float gm = a.gvec[m];
      const Vec<VEC> wv = ldv<VEC>(a.wvec + n0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) dy.v[j] = gm * wv.v[j];
    }
    uint8_t kv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) kv[j] = 1;
    if (a.keep) {
      if constexpr (VEC == 4) {
        const uchar4 k4 = *reinterpret_cast<const uchar4 *>(a.keep + e0);
        kv[0] = k4.x; kv[1] = k4.y; kv[2] = k4.z; kv[3] = k4.w;
      } else {
        kv[0] = a.keep[e0];
      }
    }
    Vec<VEC> mean, rstd, g, b, db, dg, out;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { mean.v[j] = 0.f; rstd.v[j] = 1.f; g.v[j] = 1.f; b.v[j] = 0.f; db.v[j] = 0.f; dg.v[j] = 0.f; }
    if (a.has_bn) {
      mean = ldv<VEC>(a.save_mean + n0);
      rstd = ldv<VEC>(a.save_rstd + n0);
      g = ldv_or<VEC>(a.gamma, n0, 1.f);
      b = ldv_or<VEC>(a.beta, n0, 0.f);
      if (a.training) {
        db = ldv<VEC>(a.dbeta + n0);
        dg = ldv<VEC>(a.dgamma + n0);
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float zh = (z.v[j] - mean.v[j]) * rstd.v[j];
      const float pre = g.v[j] * zh + b.v[j];
      float d = kv[j] ? dy.v[j] * keep_scale : 0.f;
      d = pre > 0.f ? d : 0.f;
      float dz = d;
      if (a.has_bn)
        dz = a.training ? g.v[j] * rstd.v[j] * (d - db.v[j] * invM - zh * dg.v[j] * invM) : g.v[j] * rstd.v[j] * d;
      out.v[j] = dz;
    }
    stv<VEC>(a.dZ + e0, out);
  }
}

inline int grid_for_elems(int64_t total) {
  int64_t g = (total + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  if (g > kMaxGrid) g = kMaxGrid;
  return (int)g;
}
inline dim3 col_grid_v4(int M, int N) {
  const int cb = (N + kColTile - 1) / kColTile;
  int rb = (M + 31) / 32;                  // >= 8 rows per wave
  const int want = (384 + cb - 1) / cb;    // ~1.5 workgroups per CU: few adders per column
  if (rb > want) rb = want;
  if (rb < 1) rb = 1;
  return dim3(cb, rb);
}
inline dim3 col_grid(int M, int N) {
  // enough row-blocks that (column blocks x row blocks) is a few workgroups per CU
  const int cb = (N + 63) / 64;
  int rb = (M + 31) / 32;
  const int want = (1024 + cb - 1) / cb;
  if (rb > want) rb = want;
  if (rb < 1) rb = 1;
  return dim3(cb, rb);
}

}  // namespace

extern "C" {

int mi_bn_relu_dropout_fwd(const float *Z, int32_t ldz, int32_t M, int32_t N, int32_t has_bn, int32_t training,
                           const float *gamma, const float *beta, float *running_mean, float *running_var,
                           float momentum, float eps, float p, int64_t *seed, int64_t salt, int32_t bump_seed,
                           int64_t *num_batches_tracked,
                           float *stats /*[2,N] caller-zeroed; training BN only*/, const float *mean_offset,
                           float *Y, uint8_t *keep, float *save_mean, float *save_rstd, void *stream) {
  if (M < 0 || N < 0 || p < 0.f || p >= 1.f) return MI_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return MI_OK;
  if (!Z || !Y) return MI_ERR_INVALID_ARG;
  const bool drop = training && p > 0.f;
  if (drop && (!seed || !keep)) return MI_ERR_INVALID_ARG;
  if (has_bn && (!save_mean || !save_rstd)) return MI_ERR_INVALID_ARG;
  if (has_bn && training && !stats) return MI_ERR_INVALID_ARG;
  if (has_bn && !training && (!running_mean || !running_var)) return MI_ERR_INVALID_ARG;
  if (bump_seed && !(has_bn && training)) return MI_ERR_INVALID_ARG;  // only the statistics launch can bump
  if (has_bn && training) {
    int64_t *bump = bump_seed ? seed : nullptr;
    if (N % 4 == 0 && ldz % 4 == 0 && aligned16(Z) && aligned16(stats))
      MI_LAUNCH("bn_col_stats", k_col_stats_v4, col_grid_v4(M, N), kBlock, stream, Z, ldz, stats, stats + N, M, N, bump);
    else
      MI_LAUNCH("bn_col_stats", k_col_stats, col_grid(M, N), kBlock, stream, Z, ldz, stats, stats + N, M, N, bump);
  }
  BnArgs a;
  a.Z = Z; a.ld = ldz; a.M = M; a.N = N; a.has_bn = has_bn; a.training = training;
  a.s1 = stats; a.s2 = stats ? stats + N : nullptr;
  a.mean_offset = (has_bn && training) ? mean_offset : nullptr;
  a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var;
  a.momentum = momentum; a.eps = eps; a.p = p; a.seed = seed; a.salt = salt;
  a.Y = Y; a.keep = drop ? keep : nullptr; a.save_mean = save_mean; a.save_rstd = save_rstd;
  a.num_batches_tracked = (has_bn && training) ? num_batches_tracked : nullptr;
  const bool v4 = (N % 4 == 0) && (ldz % 4 == 0) && aligned16(Z) && aligned16(Y) && (!a.keep || ((uintptr_t)a.keep & 3) == 0) &&
                  (!gamma || aligned16(gamma)) && (!beta || aligned16(beta)) && (!stats || aligned16(stats)) &&
                  (!has_bn || (aligned16(save_mean) && aligned16(save_rstd))) &&
                  (!running_mean || (aligned16(running_mean) && aligned16(running_var)));
  if (v4) MI_LAUNCH("bn_relu_dropout_fwd", k_bn_relu_drop_fwd<4>, grid_for_elems((int64_t)M * N / 4), kBlock, stream, a);
  else MI_LAUNCH("bn_relu_dropout_fwd", k_bn_relu_drop_fwd<1>, grid_for_elems((int64_t)M * N), kBlock, stream, a);
  return launch_status();
}

int mi_bn_relu_dropout_bwd(const float *dY, const float *Z, int32_t ldz, int32_t M, int32_t N, int32_t has_bn,
                           int32_t training, const uint8_t *keep, float p, const float *gamma, const float *beta,
                           const float *save_mean, const float *save_rstd, float *dgamma_dbeta /*[2,N] zeroed*/,
                           float *dZ, const float *gvec, const float *wvec, void *stream) {
  if (M < 0 || N < 0 || p < 0.f || p >= 1.f) return MI_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return MI_OK;
  if (!Z || !dZ || (!dY && (!gvec || !wvec))) return MI_ERR_INVALID_ARG;
  if (has_bn && (!save_mean || !save_rstd || !dgamma_dbeta)) return MI_ERR_INVALID_ARG;
  BnBwdArgs a;
  a.dY = dY; a.gvec = dY ? nullptr : gvec; a.wvec = dY ? nullptr : wvec; a.Z = Z; a.ld = ldz; a.M = M; a.N = N; a.has_bn = has_bn; a.training = training;
  a.keep = keep; a.p = p; a.gamma = gamma; a.beta = beta; a.save_mean = save_mean; a.save_rstd = save_rstd;
  a.dgamma = dgamma_dbeta; a.dbeta = dgamma_dbeta ? dgamma_dbeta + N : nullptr;
  a.dZ = dZ;
  if (has_bn) {
    const bool v4r = (N % 4 == 0) && (ldz % 4 == 0) && aligned16(Z) && (dY ? aligned16(dY) : aligned16(wvec)) && (!keep || ((uintptr_t)keep & 3) == 0) &&
                     aligned16(save_mean) && aligned16(save_rstd) && aligned16(dgamma_dbeta) && (!gamma || aligned16(gamma)) &&
                     (!beta || aligned16(beta));
    if (v4r && dY) MI_LAUNCH("bn_bwd_reduce", k_bn_bwd_reduce_v4<false>, col_grid_v4(M, N), kBlock, stream, a);
    else if (v4r) MI_LAUNCH("bn_bwd_reduce", k_bn_bwd_reduce_v4<true>, col_grid_v4(M, N), kBlock, stream, a);
    else MI_LAUNCH("bn_bwd_reduce", k_bn_bwd_reduce, col_grid(M, N), kBlock, stream, a);
  }
  const bool v4 = (N % 4 == 0) && (ldz % 4 == 0) && aligned16(Z) && aligned16(dZ) && (dY ? aligned16(dY) : aligned16(wvec)) &&
                  (!keep || ((uintptr_t)keep & 3) == 0) && (!gamma || aligned16(gamma)) && (!beta || aligned16(beta)) &&
                  (!has_bn || (aligned16(save_mean) && aligned16(save_rstd) && aligned16(dgamma_dbeta)));
  if (v4 && dY) MI_LAUNCH("bn_relu_dropout_bwd", (k_bn_bwd_apply<4, false>), grid_for_elems((int64_t)M * N / 4), kBlock, stream, a);
  else if (v4) MI_LAUNCH("bn_relu_dropout_bwd", (k_bn_bwd_apply<4, true>), grid_for_elems((int64_t)M * N / 4), kBlock, stream, a);
  else if (dY) MI_LAUNCH("bn_relu_dropout_bwd", (k_bn_bwd_apply<1, false>), grid_for_elems((int64_t)M * N), kBlock, stream, a);
  else MI_LAUNCH("bn_relu_dropout_bwd", (k_bn_bwd_apply<1, true>), grid_for_elems((int64_t)M * N), kBlock, stream, a);
  return launch_status();
}

}  // extern "C"
