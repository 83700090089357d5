// tail_masks.hpp — the dropout keep bits of the MLP tail as a job any launch can carry in extra workgroups (tail.hip's
// own mask kernel, its first finalize launch, or — DeepFM's fused step — the gather + FM forward in front of the tail).
#pragma once
#include "common.hpp"
#include "tail_gemm.hpp"

namespace mi {
using tg::mix64;

__device__ __forceinline__ uint64_t layer_seed(const int64_t *seed, int64_t salt) {
  return (seed ? (uint64_t)seed[0] : 0ull) + 0xD1B54A32D192ED03ull * (uint64_t)salt;
}
// keep bits of up to 8 layers in one launch: byte b of layer l covers elements 8b .. 8b+7 of its [M, ld] activation
struct MaskJob {
  uint8_t *bits[8];
  int64_t salt[8];
  int64_t nbytes[8];       // M * ld / 8
  uint32_t thr[8];
  int n;
};
// workgroup `blk` of `nblk` that share the job
__device__ __forceinline__ void mask_blocks(const MaskJob &j, const int64_t *seed, float4 *__restrict__ zero4, int64_t nzero4,
                                            int blk, int nblk) {
  // rides along: the zero fill of the backward pass's accumulation buffer (split-K weight gradients), one launch less
  for (int64_t i = (int64_t)blk * kBlock + threadIdx.x; i < nzero4; i += (int64_t)nblk * kBlock)
    zero4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int l = 0; l < j.n; ++l) {
    const uint64_t sd = layer_seed(seed, j.salt[l]);
    const uint32_t thr = j.thr[l];
    for (int64_t b = (int64_t)blk * kBlock + threadIdx.x; b < j.nbytes[l]; b += (int64_t)nblk * kBlock) {
      uint32_t byte = 0;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const uint64_t h = mix64(sd, (uint64_t)(2 * b + half));
#pragma unroll
        for (int e = 0; e < 4; ++e) byte |= (((uint32_t)(h >> (16 * e)) & 0xFFFFu) >= thr ? 1u : 0u) << (4 * half + e);
      }
      j.bits[l][b] = (uint8_t)byte;
    }
  }
}
// Per-column constants (mu, sc, be, rstd) of layers that normalise with FIXED statistics or not at all (eval-mode BatchNorm,
// use_batchnorm=False: see k_tail_affine_consts, tail.hip) — up to 8 layers, a job like the keep bits: its own launch
// (mi_tail_affine_consts) or extra workgroups of the lookup in front of the tail (an inference forward is six launches of a
// few microseconds each: one fewer is 8 % of its latency at batch 64).
struct AffineJob {
  const float *gamma[8], *beta[8], *rmean[8], *rvar[8], *bias[8];
  float *mu[8], *sc[8], *be[8], *rstd[8];
  float eps[8];
  int n[8];
  int nl;
};
__device__ __forceinline__ void affine_consts_blocks(const AffineJob &j, int blk, int nblk) {
  for (int l = 0; l < j.nl; ++l) {
    for (int c = blk * kBlock + (int)threadIdx.x; c < j.n[l]; c += nblk * kBlock) {
      const float b = j.bias[l] ? j.bias[l][c] : 0.f;
      if (j.rvar[l]) {
        const float r = rsqrtf(j.rvar[l][c] + j.eps[l]);
        j.mu[l][c] = j.rmean[l][c] - b;
        j.sc[l][c] = (j.gamma[l] ? j.gamma[l][c] : 1.f) * r;
        j.be[l][c] = j.beta[l] ? j.beta[l][c] : 0.f;
        j.rstd[l][c] = r;
      } else {
        j.mu[l][c] = 0.f;
        j.sc[l][c] = 1.f;
        j.be[l][c] = b;
        j.rstd[l][c] = 1.f;
      }
    }
  }
}
// host: builds the job from mi_tail_affine_consts' arguments; *blocks = workgroups that suit it (0: nothing to do)
inline int affine_job(int32_t nlayers, const int32_t *widths, const float *const *gamma, const float *const *beta,
                      const float *const *running_mean, const float *const *running_var, const float *const *bias,
                      const float *eps, float *const *mu, float *const *sc, float *const *be, float *const *rstd, AffineJob &j,
                      int *blocks) {
  *blocks = 0;
  j.nl = 0;
  if (nlayers < 0 || nlayers > 8) return MI_ERR_INVALID_ARG;
  if (nlayers == 0) return MI_OK;
  if (!widths || !gamma || !beta || !running_mean || !running_var || !bias || !eps || !mu || !sc || !be || !rstd)
    return MI_ERR_INVALID_ARG;
  j.nl = nlayers;
  int widest = 0;
  for (int l = 0; l < nlayers; ++l) {
    if (widths[l] <= 0 || !mu[l] || !sc[l] || !be[l] || !rstd[l]) return MI_ERR_INVALID_ARG;
    if ((running_mean[l] == nullptr) != (running_var[l] == nullptr)) return MI_ERR_INVALID_ARG;
    j.gamma[l] = gamma[l]; j.beta[l] = beta[l]; j.rmean[l] = running_mean[l]; j.rvar[l] = running_var[l]; j.bias[l] = bias[l];
    j.mu[l] = mu[l]; j.sc[l] = sc[l]; j.be[l] = be[l]; j.rstd[l] = rstd[l];
    j.eps[l] = eps[l]; j.n[l] = widths[l];
    widest = widths[l] > widest ? widths[l] : widest;
  }
  *blocks = (widest + kBlock - 1) / kBlock;
  return MI_OK;
}

// the jobs as a kernel argument
struct MaskRide {
  MaskJob j;
  const int64_t *seed;
  float4 *zero4;
  int64_t nzero4;
  int mask_blocks;      // workgroups of the keep bits + zero fill (0: none); the affine job's follow them
  AffineJob aff;        // aff.nl == 0: none
};
// builds the device job; *grid = workgroups that suit it (0: nothing to do)
inline int mask_job(const int64_t *seed, int32_t nlayers, const int64_t *salts, const float *ps, const int32_t *lds,
                    uint8_t *const *bits, int32_t M, float *zero_buf, int64_t zero_floats, MaskJob &j, int64_t *grid_out) {
  *grid_out = 0;
  j.n = 0;
  if (nlayers < 0 || nlayers > 8 || M < 0 || zero_floats < 0 || (zero_floats & 3)) return MI_ERR_INVALID_ARG;
  if (zero_floats && (!zero_buf || !aligned16(zero_buf))) return MI_ERR_INVALID_ARG;
  if ((nlayers == 0 || M == 0) && zero_floats == 0) return MI_OK;
  if (nlayers && M && (!seed || !salts || !ps || !lds || !bits)) return MI_ERR_INVALID_ARG;
  int64_t most = zero_floats / 4;
  for (int l = 0; l < nlayers && M > 0; ++l) {
    if (ps[l] <= 0.f) continue;
    if (!bits[l] || lds[l] <= 0 || lds[l] % 8 || ps[l] >= 1.f) return MI_ERR_INVALID_ARG;
    j.bits[j.n] = bits[l];
    j.salt[j.n] = salts[l];
    j.nbytes[j.n] = (int64_t)M * lds[l] / 8;
    j.thr[j.n] = (uint32_t)(ps[l] * 65536.f + 0.5f);
    most = j.nbytes[j.n] > most ? j.nbytes[j.n] : most;
    ++j.n;
  }
  if (j.n == 0 && zero_floats == 0) return MI_OK;
  int64_t grid = (most + kBlock - 1) / kBlock;
  if (grid > kMaxGrid) grid = kMaxGrid;
  if (grid < 1) grid = 1;
  *grid_out = grid;
  return MI_OK;
}

}  // namespace mi
