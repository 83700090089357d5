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

// grid (ceil(N/64), RB); block 256 = 64 columns x 4 row lanes
__global__ __launch_bounds__(kBlock) void k_col_stats(const float *__restrict__ Z, int ld, float *__restrict__ s1,
                                                      float *__restrict__ s2, int M, int N) {
  __shared__ float p1[4][64], p2[4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  float a = 0.f, b = 0.f;
  if (n < N) {
    const float shift = Z[n];
    for (int m = blockIdx.y * 4 + rl; m < M; m += gridDim.y * 4) {
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
  const float *gamma, *beta;     // nullable = 1 / 0
  float *running_mean, *running_var;
  float momentum, eps;
  float p;                       // dropout probability (0 = none)
  const int64_t *seed;
  int64_t salt;
  float *Y;
  uint8_t *keep;                 // [M,N] when p > 0 and training
  float *save_mean, *save_rstd;  // [N]
};

__global__ __launch_bounds__(kBlock) void k_bn_relu_drop_fwd(BnArgs a) {
  const int64_t total = (int64_t)a.M * a.N;
  const bool drop = a.training && a.p > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - a.p) : 1.f;
  const uint32_t thresh = drop ? (uint32_t)((double)a.p * 4294967296.0) : 0u;  // keep iff rng >= p * 2^32
  const uint64_t seed = drop ? (uint64_t)(a.seed[0] + a.salt) : 0ull;
  const float invM = 1.f / (float)a.M;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(e / a.N), n = (int)(e % a.N);
    const float z = a.Z[(int64_t)m * a.ld + n];
    float mean = 0.f, rstd = 1.f, g = 1.f, b = 0.f;
    if (a.has_bn) {
      if (a.training) {
        const float d1 = a.s1[n] * invM;
        mean = a.Z[n] + d1;
        const float var = fmaxf(a.s2[n] * invM - d1 * d1, 0.f);
        rstd = rsqrtf(var + a.eps);
        if (m == 0) {  // one thread per column owns the bookkeeping
          a.save_mean[n] = mean;
          a.save_rstd[n] = rstd;
          if (a.running_mean) {
            const float unbiased = a.M > 1 ? var * ((float)a.M / (float)(a.M - 1)) : var;
            a.running_mean[n] = (1.f - a.momentum) * a.running_mean[n] + a.momentum * mean;
            a.running_var[n] = (1.f - a.momentum) * a.running_var[n] + a.momentum * unbiased;
          }
        }
      } else {
        mean = a.running_mean[n];
        rstd = rsqrtf(a.running_var[n] + a.eps);
        if (m == 0) {
          a.save_mean[n] = mean;
          a.save_rstd[n] = rstd;
        }
      }
      g = a.gamma ? a.gamma[n] : 1.f;
      b = a.beta ? a.beta[n] : 0.f;
    }
    float y = g * (z - mean) * rstd + b;
    y = y > 0.f ? y : 0.f;
    if (drop) {
      const bool k = rng32(seed, (uint64_t)e) >= thresh;
      a.keep[e] = k ? 1 : 0;
      y = k ? y * keep_scale : 0.f;
    }
    a.Y[e] = y;
  }
}

struct BnBwdArgs {
  const float *dY, *Z;
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
  float d = a.dY[e];
  if (a.keep) d = a.keep[e] ? d * (1.f / (1.f - a.p)) : 0.f;
  return pre > 0.f ? d : 0.f;
}

__global__ __launch_bounds__(kBlock) void k_bn_bwd_reduce(BnBwdArgs a) {
  __shared__ float p1[4][64], p2[4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  float sb = 0.f, sg = 0.f;
  if (n < a.N) {
    for (int m = blockIdx.y * 4 + rl; m < a.M; m += gridDim.y * 4) {
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

__global__ __launch_bounds__(kBlock) void k_bn_bwd_apply(BnBwdArgs a) {
  const int64_t total = (int64_t)a.M * a.N;
  const float invM = 1.f / (float)a.M;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(e / a.N), n = (int)(e % a.N);
    float zh;
    const float d = bwd_dyh(a, e, n, a.Z[(int64_t)m * a.ld + n], zh);
    float dz = d;
    if (a.has_bn) {
      const float g = a.gamma ? a.gamma[n] : 1.f;
      const float rstd = a.save_rstd[n];
      dz = a.training ? g * rstd * (d - a.dbeta[n] * invM - zh * a.dgamma[n] * invM) : g * rstd * d;
    }
    a.dZ[e] = dz;
  }
}

inline int grid_for_elems(int64_t total) {
  int64_t g = (total + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  if (g > kMaxGrid) g = kMaxGrid;
  return (int)g;
}
inline dim3 col_grid(int M, int N) {
  int rb = (M + 127) / 128;
  if (rb > 64) rb = 64;
  if (rb < 1) rb = 1;
  return dim3((N + 63) / 64, rb);
}

}  // namespace

extern "C" {

int mi_bn_relu_dropout_fwd(const float *Z, int32_t ldz, int32_t M, int32_t N, int32_t has_bn, int32_t training,
                           const float *gamma, const float *beta, float *running_mean, float *running_var,
                           float momentum, float eps, float p, const int64_t *seed, int64_t salt,
                           float *stats /*[2,N] caller-zeroed; training BN only*/, float *Y, uint8_t *keep,
                           float *save_mean, float *save_rstd, void *stream) {
  if (M < 0 || N < 0 || p < 0.f || p >= 1.f) return MI_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return MI_OK;
  if (!Z || !Y) return MI_ERR_INVALID_ARG;
  const bool drop = training && p > 0.f;
  if (drop && (!seed || !keep)) return MI_ERR_INVALID_ARG;
  if (has_bn && (!save_mean || !save_rstd)) return MI_ERR_INVALID_ARG;
  if (has_bn && training && !stats) return MI_ERR_INVALID_ARG;
  if (has_bn && !training && (!running_mean || !running_var)) return MI_ERR_INVALID_ARG;
  if (has_bn && training) {
    hipEvent_t ea, eb;
    if (mi::prof_acquire("bn_col_stats", &ea, &eb))
      hipExtLaunchKernelGGL(k_col_stats, col_grid(M, N), dim3(kBlock), 0, (hipStream_t)stream, ea, eb, 0, Z, ldz, stats,
                            stats + N, M, N);
    else
      hipLaunchKernelGGL(k_col_stats, col_grid(M, N), dim3(kBlock), 0, (hipStream_t)stream, Z, ldz, stats, stats + N,
                         M, N);
  }
  BnArgs a;
  a.Z = Z; a.ld = ldz; a.M = M; a.N = N; a.has_bn = has_bn; a.training = training;
  a.s1 = stats; a.s2 = stats ? stats + N : nullptr;
  a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var;
  a.momentum = momentum; a.eps = eps; a.p = p; a.seed = seed; a.salt = salt;
  a.Y = Y; a.keep = drop ? keep : nullptr; a.save_mean = save_mean; a.save_rstd = save_rstd;
  MI_LAUNCH("bn_relu_dropout_fwd", k_bn_relu_drop_fwd, grid_for_elems((int64_t)M * N), kBlock, stream, a);
  return launch_status();
}

int mi_bn_relu_dropout_bwd(const float *dY, const float *Z, int32_t ldz, int32_t M, int32_t N, int32_t has_bn,
                           int32_t training, const uint8_t *keep, float p, const float *gamma, const float *beta,
                           const float *save_mean, const float *save_rstd, float *dgamma_dbeta /*[2,N] zeroed*/,
                           float *dZ, void *stream) {
  if (M < 0 || N < 0 || p < 0.f || p >= 1.f) return MI_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return MI_OK;
  if (!dY || !Z || !dZ) return MI_ERR_INVALID_ARG;
  if (has_bn && (!save_mean || !save_rstd || !dgamma_dbeta)) return MI_ERR_INVALID_ARG;
  BnBwdArgs a;
  a.dY = dY; a.Z = Z; a.ld = ldz; a.M = M; a.N = N; a.has_bn = has_bn; a.training = training;
  a.keep = keep; a.p = p; a.gamma = gamma; a.beta = beta; a.save_mean = save_mean; a.save_rstd = save_rstd;
  a.dgamma = dgamma_dbeta; a.dbeta = dgamma_dbeta ? dgamma_dbeta + N : nullptr;
  a.dZ = dZ;
  if (has_bn) {
    hipEvent_t ea, eb;
    if (mi::prof_acquire("bn_bwd_reduce", &ea, &eb))
      hipExtLaunchKernelGGL(k_bn_bwd_reduce, col_grid(M, N), dim3(kBlock), 0, (hipStream_t)stream, ea, eb, 0, a);
    else
      hipLaunchKernelGGL(k_bn_bwd_reduce, col_grid(M, N), dim3(kBlock), 0, (hipStream_t)stream, a);
  }
  MI_LAUNCH("bn_relu_dropout_bwd", k_bn_bwd_apply, grid_for_elems((int64_t)M * N), kBlock, stream, a);
  return launch_status();
}

}  // extern "C"
