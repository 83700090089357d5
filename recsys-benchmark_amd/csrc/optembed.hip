// optembed.hip — OptEmbed supernet lookup (last flavour):
// OptEmbed.forward of src/models/embeddings/deepfm_opt_embed.py:203-243 with _MaskEmbeddingModule and
// BinaryStep of optembed_utils.py:10-112 — row gather, row mask e = step(||w||_p - t), dimension mask
// d = [dim <= dmax] in one kernel each way.
//
//   y[i, :] = W[row_i, :] * s_i * [dim <= dmax_i],   s_i = [ ||W[row_i]||_p - t[tix_i] > 0 ],  p = 1 or 2
//   backward: dW[row] += g * md * s  +  (sum_d g_d md_d w_d) * a(u) * d||w||/dw,   dt[tix] -= (sum ...) * a(u)
//   with BinaryStep's surrogate a(u) = 2 - 4|u| for |u| <= 0.4, 0.4 for 0.4 < |u| <= 1, 0 beyond
//   (optembed_utils.py:34-43), d||w||_1/dw = sign(w), d||w||_2/dw = w / ||w||_2.
// tix_i = i % F (training forward: one threshold per field, the lookup is [B, F]) or row_i / a per-row map
// given by the caller (get_weight over the whole table with per-feature or per-field thresholds).
// One wave per lookup, lanes stride over the row (coalesced); norms by a wave reduction.
#include "common.hpp"

namespace {
using namespace mi;

struct OptArgs {
  const int64_t *idx;     // [n] rows
  const float *W;         // [N, D]
  const float *t;         // thresholds (nullable: no row mask)
  const int64_t *tix;     // [n] threshold index per lookup (nullable: i % F)
  int F;
  const int64_t *dmax;    // [n] last kept dimension per lookup (nullable: keep all)
  int norm;               // 1 or 2
  int64_t n;
  int D;
  int64_t N;
};

__device__ __forceinline__ float surrogate(float u) {
  const float a = fabsf(u);
  return a > 1.f ? 0.f : (a > 0.4f ? 0.4f : 2.f - 4.f * a);
}

__global__ __launch_bounds__(kBlock) void k_optembed_fwd(OptArgs a, float *__restrict__ out, int *err) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  int bad = 0;
  for (int64_t i = wave0; i < a.n; i += nwaves) {
    const int64_t row = a.idx[i];
    const bool ok = (uint64_t)row < (uint64_t)a.N;
    bad |= !ok;
    float s = 1.f;
    if (a.t && ok) {
      float acc = 0.f;
      for (int d = lane; d < a.D; d += kWave) {
        const float w = a.W[row * a.D + d];
        acc += a.norm == 1 ? fabsf(w) : w * w;
      }
      acc = wave_sum(acc);
      const float nrm = a.norm == 1 ? acc : sqrtf(acc);
      const int64_t ti = a.tix ? a.tix[i] : (i % a.F);
      s = (nrm - a.t[ti]) > 0.f ? 1.f : 0.f;
    }
    const int64_t dm = a.dmax ? a.dmax[i] : (int64_t)a.D;
    for (int d = lane; d < a.D; d += kWave)
      out[i * a.D + d] = (ok && d <= dm) ? a.W[row * a.D + d] * s : 0.f;
  }
  if (bad && err) atomicOr(err, MI_IDX_OUT_OF_RANGE);
}

__global__ __launch_bounds__(kBlock) void k_optembed_bwd(OptArgs a, const float *__restrict__ g,
                                                         float *__restrict__ dW, float *__restrict__ dt) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t i = wave0; i < a.n; i += nwaves) {
    const int64_t row = a.idx[i];
    if ((uint64_t)row >= (uint64_t)a.N) continue;
    const int64_t dm = a.dmax ? a.dmax[i] : (int64_t)a.D;
    float nacc = 0.f, gw = 0.f;
    for (int d = lane; d < a.D; d += kWave) {
      const float w = a.W[row * a.D + d];
      nacc += a.norm == 1 ? fabsf(w) : w * w;
      if (d <= dm) gw += g[i * a.D + d] * w;            // dL/ds
    }
    float s = 1.f, coef = 0.f, nrm = 1.f;
    int64_t ti = 0;
    if (a.t) {
      nacc = wave_sum(nacc);
      gw = wave_sum(gw);
      nrm = a.norm == 1 ? nacc : sqrtf(nacc);
      ti = a.tix ? a.tix[i] : (i % a.F);
      const float u = nrm - a.t[ti];
      s = u > 0.f ? 1.f : 0.f;
      coef = gw * surrogate(u);
      if (lane == 0 && dt && coef != 0.f) atomicAdd(dt + ti, -coef);
    }
    if (dW) {
      for (int d = lane; d < a.D; d += kWave) {
        const float w = a.W[row * a.D + d];
        float v = d <= dm ? g[i * a.D + d] * s : 0.f;
        if (a.t) {
          const float dn = a.norm == 1 ? (w > 0.f ? 1.f : (w < 0.f ? -1.f : 0.f)) : (nrm > 0.f ? w / nrm : 0.f);
          v += coef * dn;
        }
        atomicAdd(dW + row * a.D + d, v);
      }
    }
  }
}

inline int fill(OptArgs &a, const int64_t *idx, const float *W, const float *t, const int64_t *tix, int32_t F,
                const int64_t *dmax, int32_t norm, int64_t n, int32_t D, int64_t N) {
  if (n < 0 || D <= 0 || N < 0 || (norm != 1 && norm != 2)) return MI_ERR_INVALID_ARG;
  if (n > 0 && (!idx || !W)) return MI_ERR_INVALID_ARG;
  if (t && !tix && F < 1) return MI_ERR_INVALID_ARG;
  a.idx = idx; a.W = W; a.t = t; a.tix = tix; a.F = F > 0 ? F : 1; a.dmax = dmax; a.norm = norm;
  a.n = n; a.D = D; a.N = N;
  return MI_OK;
}

}  // namespace

extern "C" {

int mi_optembed_fwd(const int64_t *idx, const float *W, const float *t, const int64_t *tix, int32_t F,
                    const int64_t *dmax, int32_t norm, float *out, int64_t n, int32_t D, int64_t N,
                    int32_t *err, void *stream) {
  OptArgs a;
  const int rc = fill(a, idx, W, t, tix, F, dmax, norm, n, D, N);
  if (rc != MI_OK) return rc;
  if (n == 0) return MI_OK;
  if (!out) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("optembed_fwd", k_optembed_fwd, grid_for_waves(n), kBlock, stream, a, out, err);
  return launch_status();
}

int mi_optembed_bwd(const int64_t *idx, const float *W, const float *t, const int64_t *tix, int32_t F,
                    const int64_t *dmax, int32_t norm, const float *g, float *dW, float *dt, int64_t n,
                    int32_t D, int64_t N, void *stream) {
  OptArgs a;
  const int rc = fill(a, idx, W, t, tix, F, dmax, norm, n, D, N);
  if (rc != MI_OK) return rc;
  if (n == 0) return MI_OK;
  if (!g) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("optembed_bwd", k_optembed_bwd, grid_for_waves(n), kBlock, stream, a, g, dW, dt);
  return launch_status();
}

}  // extern "C"
