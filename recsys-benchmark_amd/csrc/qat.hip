// qat.hip — quantisation-aware-training embedding lookup:
// QAT_EmbInt.forward = VanillaEmbedding.forward followed by StotasticRounding
// (src/models/embeddings/qat_emb.py:16-45,117-119), one kernel each way.
//
//   q  = clamp(w / scale, q_min, q_max);  fl = floor(q);  p_floor = fl + 1 - q
//   out = (fl + [u > p_floor]) * scale,   u ~ U[0,1)   (torch.rand_like in the reference)
//   backward (:49-84): dW = g (straight-through, scattered to the looked-up rows);
//   dscale = sum g * m,  m = q_max if w/scale >= q_max, q_min if w/scale <= q_min, else (fl + [u > p_floor]) - w/scale
//
// u comes from the library's counter generator (splitmix64 of a device seed word + element index), so the
// backward re-derives the forward's rounding instead of storing it; tests may pass an explicit `prob`
// array (the reference's torch.rand_like draw) to pin the arithmetic bit for bit.
#include "common.hpp"

namespace {
using namespace mi;

__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  const uint32_t r = (uint32_t)((z ^ (z >> 31)) >> 40);   // 24 random bits: exactly representable, < 1
  return (float)r * (1.0f / 16777216.0f);
}

struct QatArgs {
  const int64_t *idx;   // nullable: identity (row i)
  const float *W;
  const float *scale;   // device scalar
  float qmin, qmax;
  const float *prob;    // nullable
  const int64_t *seed;  // device word (used when prob == NULL)
  int64_t salt;
  int64_t n;
  int D;
  int64_t N;
};

__device__ __forceinline__ float sr_round(const QatArgs &a, float w, float s, int64_t e, float &qf) {
  qf = w / s;
  const float q = fminf(fmaxf(qf, a.qmin), a.qmax);
  const float fl = floorf(q);
  const float pf = fl + 1.f - q;
  const float u = a.prob ? a.prob[e] : uniform01((uint64_t)(a.seed[0] + a.salt), (uint64_t)e);
  return fl + (u > pf ? 1.f : 0.f);
}

__global__ __launch_bounds__(kBlock) void k_qat_fwd(QatArgs a, float *__restrict__ out, int *err) {
  const int64_t total = a.n * a.D;
  const float s = a.scale[0];
  int bad = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / a.D;
    const int d = (int)(e - i * a.D);
    const int64_t row = a.idx ? a.idx[i] : i;
    const bool ok = (uint64_t)row < (uint64_t)a.N;
    bad |= !ok;
    float qf;
    out[e] = ok ? sr_round(a, a.W[row * a.D + d], s, e, qf) * s : 0.f;
  }
  if (bad && err) atomicOr(err, MI_IDX_OUT_OF_RANGE);
}

__global__ __launch_bounds__(kBlock) void k_qat_bwd(QatArgs a, const float *__restrict__ g,
                                                    float *__restrict__ dW, float *__restrict__ dscale) {
  __shared__ float red[kWavesPerBlock];
  const int64_t total = a.n * a.D;
  const float s = a.scale[0];
  float acc = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / a.D;
    const int d = (int)(e - i * a.D);
    const int64_t row = a.idx ? a.idx[i] : i;
    if ((uint64_t)row >= (uint64_t)a.N) continue;
    const float ge = g[e];
    if (dW) {
      if (a.idx) atomicAdd(dW + row * a.D + d, ge);   // rows repeat
      else dW[e] = ge;
    }
    if (dscale) {
      float qf;
      const float res = sr_round(a, a.W[row * a.D + d], s, e, qf);
      const float m = qf >= a.qmax ? a.qmax : (qf <= a.qmin ? a.qmin : res - qf);
      acc += ge * m;
    }
  }
  if (dscale) {
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int j = 0; j < kWavesPerBlock; ++j) t += red[j];
      atomicAdd(dscale, t);
    }
  }
}

inline int grid_for(int64_t total) {
  int64_t g = (total + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  if (g > kMaxGrid) g = kMaxGrid;
  return (int)g;
}

inline int fill(QatArgs &a, const int64_t *idx, const float *W, const float *scale, int32_t n_bits,
                const float *prob, const int64_t *seed, int64_t salt, int64_t n, int32_t D, int64_t N) {
  if (n < 0 || D <= 0 || N < 0 || n_bits < 2 || n_bits > 24) return MI_ERR_INVALID_ARG;
  if (!W || !scale || (!prob && !seed)) return MI_ERR_INVALID_ARG;
  a.idx = idx; a.W = W; a.scale = scale;
  a.qmin = -(float)(1 << (n_bits - 1));
  a.qmax = (float)((1 << (n_bits - 1)) - 1);
  a.prob = prob; a.seed = seed; a.salt = salt; a.n = n; a.D = D; a.N = N;
  return MI_OK;
}

}  // namespace

extern "C" {

int mi_qat_gather_fwd(const int64_t *idx, const float *W, const float *scale, int32_t n_bits,
                      const float *prob, const int64_t *seed, int64_t salt, float *out, int64_t n,
                      int32_t D, int64_t N, int32_t *err, void *stream) {
  QatArgs a;
  const int rc = fill(a, idx, W, scale, n_bits, prob, seed, salt, n, D, N);
  if (rc != MI_OK) return rc;
  if (n == 0) return MI_OK;
  if (!out) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("qat_gather_fwd", k_qat_fwd, grid_for(n * D), kBlock, stream, a, out, err);
  return launch_status();
}

int mi_qat_gather_bwd(const int64_t *idx, const float *W, const float *scale, int32_t n_bits,
                      const float *prob, const int64_t *seed, int64_t salt, const float *g, float *dW,
                      float *dscale, int64_t n, int32_t D, int64_t N, void *stream) {
  QatArgs a;
  const int rc = fill(a, idx, W, scale, n_bits, prob, seed, salt, n, D, N);
  if (rc != MI_OK) return rc;
  if (n == 0) return MI_OK;
  if (!g) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("qat_gather_bwd", k_qat_bwd, grid_for(n * D), kBlock, stream, a, g, dW, dscale);
  return launch_status();
}

}  // extern "C"
