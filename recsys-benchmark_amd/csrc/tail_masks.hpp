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
// the job as a kernel argument
struct MaskRide {
  MaskJob j;
  const int64_t *seed;
  float4 *zero4;
  int64_t nzero4;
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
