// gather_fm.hip — DeepFM's embedding gather fused with the FM second-order term
// and the first-order (EmbeddingBag(N,1,sum)) term, forward and backward.
//
// Reference arithmetic: src/models/deepfm.py:88-98 and
// src/models/embeddings/base.py:74-75 (see include/mi355x_recsys.h).
//
// Mapping (CDNA4, wave = 64): ONE WAVE PER SAMPLE.  A row of D = 4*LPR floats is
// read by LPR adjacent lanes as one float4 each (16 B per lane, the widest
// coalesced access), so a wave-instruction moves RS = 64/LPR rows = 1 KiB.
// lane = r*LPR + q  (r = row slot, q = which float4 of the row).  The F rows of
// the sample are covered in NIT = ceil(F/RS) unrolled steps whose loads are all
// issued before the first use, so each wave has NIT row gathers in flight.
// sum_f e (per d) is a shuffle-xor reduction over the r bits of the lane id;
// the per-sample scalar is one 64-lane reduction.  emb[b] is F*D contiguous
// floats, so the stores of a step are one contiguous <=1 KiB run.
//
// HBM-bound integer/copy work: no LDS staging is needed in the forward (each
// byte is used once); the dense backward stages a 1 KiB tile per wave in LDS to
// turn 16-B-strided float4 fragments into 256-B contiguous atomic instructions.
#include "common.hpp"
#include "prefetch_rows.hpp"
#include "tail_masks.hpp"

namespace {
using namespace mi;

// ---------------------------------------------------------------- forward ----
// SHFL (F <= 64): the sample's F ids arrive by ONE coalesced load (lane l < F takes idx[b, l] + offsets[l]) and reach
// the row slots by shuffles — one dependent vector-memory instruction in front of the row gathers instead of NIT id
// loads plus NIT offset loads (measured -0.45 us of 5.8 at the headline shape, tools/probe_gather3.hip) — and
// rows_out is one coalesced store.  emb is written with non-temporal stores: nobody in this kernel reads it back, and
// what is not left dirty in L2 is not written back at the kernel's end (-0.4 us).
// ldw / ldw1: floats between consecutive rows of W / w1.  (D, 1) for the reference's two tensors; (32, 32) when both
// are views of ONE packed table fp32[N, 32] = {16 embedding floats, w1, padding} — a lookup then touches one 128-B
// line instead of two unrelated 64-B sectors (DeepFM.pack_tables()); (D + 4, D + 4) for the packed rows a sharded
// lookup received (route.hip).
// (SHFL is ignored by the generic NIT = 0 form.)
// (blk of nblk: the workgroups of the launch that gather — a launch may carry others, k_gather_fm_fwd_ride)
template <int LPR, int NIT, bool SHFL>
__device__ __forceinline__ void gather_fm_fwd_blocks(
    const int64_t *__restrict__ idx, const int64_t *__restrict__ offsets,
    const float *__restrict__ W, const float *__restrict__ w1, const float *__restrict__ bias,
    float *__restrict__ emb, float *__restrict__ yfm, int64_t *__restrict__ rows_out,
    int64_t B, int F, int64_t N, int64_t ldw, int64_t ldw1, int *err, float *__restrict__ sum_out, int blk, int nblk) {
  constexpr int RS = kWave / LPR;
  constexpr int D = LPR * 4;
  const int lane = threadIdx.x & 63;
  const int q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blk * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)nblk * kWavesPerBlock;
  const float bv = bias ? bias[0] : 0.f;
  int bad = 0;
  int64_t myoff = 0;
  if constexpr (SHFL) myoff = (offsets && lane < F) ? offsets[lane] : 0;

  for (int64_t b = wave0; b < B; b += nwaves) {
    float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
    float ss = 0.f, lin = 0.f;
    const int64_t base = b * F;
    if constexpr (NIT > 0) {
      int64_t row[NIT];
      bool act[NIT], ok[NIT];
      float4 v[NIT];
      float l[NIT];
      if constexpr (SHFL) {
        const int64_t mine = lane < F ? idx[base + lane] + myoff : 0;
        if (rows_out && lane < F) rows_out[base + lane] = mine;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          const int f = r + k * RS;
          act[k] = f < F;
          row[k] = __shfl(mine, f & 63);
        }
      } else {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          const int f = r + k * RS;
          act[k] = f < F;
          row[k] = act[k] ? idx[base + f] + (offsets ? offsets[f] : 0) : 0;
        }
      }
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        ok[k] = act[k] && (uint64_t)row[k] < (uint64_t)N;
        bad |= (act[k] && !ok[k]);
        v[k] = ok[k] ? ld4(W + row[k] * ldw + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        l[k] = (ok[k] && q == 0) ? w1[row[k] * ldw1] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int f = r + k * RS;
        if (act[k]) {
          st4_nt(emb + (base + f) * D + q * 4, v[k]);
          if (!SHFL && rows_out && q == 0) rows_out[base + f] = row[k];
        }
        S.x += v[k].x; S.y += v[k].y; S.z += v[k].z; S.w += v[k].w;
        ss += dot4(v[k], v[k]);
        lin += l[k];
      }
    } else {
      for (int f = r; f < F; f += RS) {
        const int64_t row = idx[base + f] + (offsets ? offsets[f] : 0);
        const bool ok = (uint64_t)row < (uint64_t)N;
        bad |= !ok;
        const float4 v = ok ? ld4(W + row * ldw + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && q == 0) lin += w1[row * ldw1];
        st4(emb + (base + f) * D + q * 4, v);
        if (rows_out && q == 0) rows_out[base + f] = row;
        S.x += v.x; S.y += v.y; S.z += v.z; S.w += v.w;
        ss += dot4(v, v);
      }
    }
    S = slot_sum<LPR>(S);
    // sum_f e[b, f, :] — what the FM backward needs besides the rows themselves (dE_bf = g_b (S_b - e_bf)): kept when the
    // backward runs in the epilogue of the MLP's first input-gradient product (tail.hip), which sees a tile of 6-7 fields
    // of a sample and cannot re-derive the sum over all F
    if (sum_out && r == 0) st4(sum_out + b * D + q * 4, S);
    float t = (r == 0 ? dot4(S, S) : 0.f) - ss;
    t = wave_sum(0.5f * t + lin);
    if (lane == 0) yfm[b] = t + bv;
  }
  if (bad && err) atomicOr(err, MI_IDX_OUT_OF_RANGE);
}
template <int LPR, int NIT, bool SHFL>
__global__ __launch_bounds__(kBlock) void k_gather_fm_fwd(
    const int64_t *__restrict__ idx, const int64_t *__restrict__ offsets,
    const float *__restrict__ W, const float *__restrict__ w1, const float *__restrict__ bias,
    float *__restrict__ emb, float *__restrict__ yfm, int64_t *__restrict__ rows_out,
    int64_t B, int F, int64_t N, int64_t ldw, int64_t ldw1, int *err, float *__restrict__ sum_out) {
  gather_fm_fwd_blocks<LPR, NIT, SHFL>(idx, offsets, W, w1, bias, emb, yfm, rows_out, B, F, N, ldw, ldw1, err, sum_out,
                                       (int)blockIdx.x, (int)gridDim.x);
}
// The same launch carrying the MLP tail's dropout keep bits and the zero fill of its accumulation buffer in workgroups
// past the first `ngather` (tail_masks.hpp): in DeepFM's fused step this kernel is the first of the step, every reader of
// the bits and every adder into the buffer comes later, and the mask work (~1 us spread over the chip, all ALU) runs
// beside a gather that waits on memory — one launch less per step.  The extra workgroups sit at the END of the grid: the
// gather's are dispatched first.
template <int LPR, int NIT>
__global__ __launch_bounds__(kBlock) void k_gather_fm_fwd_ride(
    const int64_t *__restrict__ idx, const int64_t *__restrict__ offsets,
    const float *__restrict__ W, const float *__restrict__ w1, const float *__restrict__ bias,
    float *__restrict__ emb, float *__restrict__ yfm, int64_t *__restrict__ rows_out,
    int64_t B, int F, int64_t N, int64_t ldw, int64_t ldw1, int *err, float *__restrict__ sum_out, int ngather, MaskRide ride) {
  if ((int)blockIdx.x >= ngather) {
    const int rb = (int)blockIdx.x - ngather;
    if (rb < ride.mask_blocks) mask_blocks(ride.j, ride.seed, ride.zero4, ride.nzero4, rb, ride.mask_blocks);
    else affine_consts_blocks(ride.aff, rb - ride.mask_blocks, (int)gridDim.x - ngather - ride.mask_blocks);
    return;
  }
  // every workgroup of the launch is resident at once (1 024 + ~200 of the chip's 2 048 slots): the gather's waves wait on
  // memory most of the time and should win the issue slot against the mask work's 64-bit multiplies whenever they are ready
  __builtin_amdgcn_s_setprio(2);
  gather_fm_fwd_blocks<LPR, NIT, true>(idx, offsets, W, w1, bias, emb, yfm, rows_out, B, F, N, ldw, ldw1, err, sum_out,
                                       (int)blockIdx.x, ngather);
}

__global__ __launch_bounds__(kBlock) void k_mask_job(MaskRide ride) {
  const int rb = (int)blockIdx.x;
  if (rb < ride.mask_blocks) mask_blocks(ride.j, ride.seed, ride.zero4, ride.nzero4, rb, ride.mask_blocks);
  else affine_consts_blocks(ride.aff, rb - ride.mask_blocks, (int)gridDim.x - ride.mask_blocks);
}

// Any D (scalar accesses): wave per sample, lanes stride over d.
__global__ __launch_bounds__(kBlock) void k_gather_fm_fwd_anyD(
    const int64_t *__restrict__ idx, const int64_t *__restrict__ offsets,
    const float *__restrict__ W, const float *__restrict__ w1, const float *__restrict__ bias,
    float *__restrict__ emb, float *__restrict__ yfm, int64_t *__restrict__ rows_out,
    int64_t B, int F, int D, int64_t N, int *err, float *__restrict__ sum_out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const float bv = bias ? bias[0] : 0.f;
  int bad = 0;
  for (int64_t b = wave0; b < B; b += nwaves) {
    const int64_t base = b * F;
    float t = 0.f;
    for (int d0 = 0; d0 < D || d0 == 0; d0 += kWave) {
      const int d = d0 + lane;
      float S = 0.f, ss = 0.f;
      for (int f = 0; f < F; ++f) {
        const int64_t row = idx[base + f] + offsets[f];
        const bool ok = (uint64_t)row < (uint64_t)N;
        bad |= !ok;
        if (d0 == 0 && lane == 0) {
          if (ok) t += w1[row];
          if (rows_out) rows_out[base + f] = row;
        }
        if (d < D) {
          const float v = ok ? W[row * D + d] : 0.f;
          emb[(base + f) * D + d] = v;
          S += v;
          ss += v * v;
        }
      }
      if (sum_out && d < D) sum_out[b * D + d] = S;
      t += 0.5f * (S * S - ss);
    }
    t = wave_sum(t);
    if (lane == 0) yfm[b] = t + bv;
  }
  if (bad && err) atomicOr(err, MI_IDX_OUT_OF_RANGE);
}

// bias gradient = sum_b g_y[b] (the bias is added to every sample's y_fm): ONE EXTRA workgroup of a backward launch —
// workgroup 0, the first one dispatched; the launcher adds it when gbias is given — adds it up in a fixed order
// (deterministic, no atomics, no zero-fill, no separate launch) and does nothing else.  Round 2 had workgroup 0 sum it
// with one scalar load per thread and trip IN FRONT of its share of the rows: a 3.4 us kernel then ended 1.2 us late on
// that one workgroup (and at B = 65 536 the 256 dependent trips doubled the kernel: 55 -> 105 us,
// tools/probe_gather3.hip).  Now: float4 loads, four independent partial sums per thread, the other workgroups' ids
// shifted down by one.  Returns true in that workgroup; blk / nblk = this workgroup's index among, and the number of,
// the workgroups that share the rows.
__device__ __forceinline__ bool bias_grad_block(const float *__restrict__ g_y, int64_t B, float *__restrict__ gbias,
                                                int &blk, int &nblk) {
  blk = blockIdx.x;
  nblk = gridDim.x;
  if (!gbias) return false;
  nblk = gridDim.x - 1;
  blk = (int)blockIdx.x - 1;
  if (blockIdx.x != 0) return false;
  __shared__ float part[kWavesPerBlock];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int64_t B4 = aligned16(g_y) ? (B & ~(int64_t)3) : 0;
  int64_t b = (int64_t)threadIdx.x * 4;
  for (; b + 3 * kBlock * 4 < B4; b += 4 * kBlock * 4) {         // four float4 loads in flight per thread
    const float4 a0 = ld4(g_y + b), a1 = ld4(g_y + b + kBlock * 4), a2 = ld4(g_y + b + 2 * kBlock * 4),
                 a3 = ld4(g_y + b + 3 * kBlock * 4);
    s0 += (a0.x + a0.y) + (a0.z + a0.w);
    s1 += (a1.x + a1.y) + (a1.z + a1.w);
    s2 += (a2.x + a2.y) + (a2.z + a2.w);
    s3 += (a3.x + a3.y) + (a3.z + a3.w);
  }
  for (; b < B4; b += kBlock * 4) {
    const float4 a0 = ld4(g_y + b);
    s0 += (a0.x + a0.y) + (a0.z + a0.w);
  }
  for (int64_t t = B4 + threadIdx.x; t < B; t += kBlock) s1 += g_y[t];
  float s = wave_sum((s0 + s1) + (s2 + s3));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int j = 0; j < kWavesPerBlock; ++j) t += part[j];
    gbias[0] = t;
  }
  return true;
}

// ----------------------------------------------------- backward, row form ----
// SLOT = false: gvals[b,f,:] / g1vals[b,f] in lookup order (the reference's COO values).
// SLOT = true : the row goes to gvals + slot[b,f]*(D+4), its first-order gradient into column D of
//               the same packed row; slots >= nslot (the dump slot of route.hip) are skipped.
template <int LPR, int NIT, bool SLOT>
__global__ __launch_bounds__(kBlock) void k_gather_fm_bwd_rows(
    const float *__restrict__ emb, const float *__restrict__ g_y,
    const float *__restrict__ g_emb, float *__restrict__ gvals, float *__restrict__ g1vals,
    int64_t B, int F, const int64_t *__restrict__ slot, int64_t nslot, float *__restrict__ gbias) {
  constexpr int RS = kWave / LPR;
  constexpr int D = LPR * 4;
  int blk, nblk;
  if (bias_grad_block(g_y, B, gbias, blk, nblk)) return;
  const int lane = threadIdx.x & 63;
  const int q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blk * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)nblk * kWavesPerBlock;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int64_t b = wave0; b < B; b += nwaves) {
    const int64_t base = b * F;
    const float gy = g_y[b];
    float4 S = z;
    if constexpr (NIT > 0) {
      float4 e[NIT], ge[NIT];
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int f = r + k * RS;
        const bool act = f < F;
        const int64_t o = (base + f) * D + q * 4;
        e[k] = act ? ld4(emb + o) : z;
        ge[k] = (act && g_emb) ? ld4(g_emb + o) : z;
      }
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        S.x += e[k].x; S.y += e[k].y; S.z += e[k].z; S.w += e[k].w;
      }
      S = slot_sum<LPR>(S);
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int f = r + k * RS;
        if (f < F) {
          float4 o4;
          o4.x = ge[k].x + gy * (S.x - e[k].x);
          o4.y = ge[k].y + gy * (S.y - e[k].y);
          o4.z = ge[k].z + gy * (S.z - e[k].z);
          o4.w = ge[k].w + gy * (S.w - e[k].w);
          if constexpr (SLOT) {
            const int64_t s = slot[base + f];
            if ((uint64_t)s < (uint64_t)nslot) {
              st4(gvals + s * (D + 4) + q * 4, o4);
              if (q == 0) st4(gvals + s * (D + 4) + D, make_float4(gy, 0.f, 0.f, 0.f));
            }
          } else {
            st4(gvals + (base + f) * D + q * 4, o4);
            if (q == 0) g1vals[base + f] = gy;
          }
        }
      }
    } else {
      for (int f = r; f < F; f += RS) {
        const float4 e = ld4(emb + (base + f) * D + q * 4);
        S.x += e.x; S.y += e.y; S.z += e.z; S.w += e.w;
      }
      S = slot_sum<LPR>(S);
      for (int f = r; f < F; f += RS) {
        const int64_t o = (base + f) * D + q * 4;
        const float4 e = ld4(emb + o);
        const float4 ge = g_emb ? ld4(g_emb + o) : z;
        float4 o4;
        o4.x = ge.x + gy * (S.x - e.x);
        o4.y = ge.y + gy * (S.y - e.y);
        o4.z = ge.z + gy * (S.z - e.z);
        o4.w = ge.w + gy * (S.w - e.w);
        if constexpr (SLOT) {
          const int64_t s = slot[base + f];
          if ((uint64_t)s < (uint64_t)nslot) {
            st4(gvals + s * (D + 4) + q * 4, o4);
            if (q == 0) st4(gvals + s * (D + 4) + D, make_float4(gy, 0.f, 0.f, 0.f));
          }
        } else {
          st4(gvals + o, o4);
          if (q == 0) g1vals[base + f] = gy;
        }
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_gather_fm_bwd_rows_anyD(
    const float *__restrict__ emb, const float *__restrict__ g_y,
    const float *__restrict__ g_emb, float *__restrict__ gvals, float *__restrict__ g1vals,
    int64_t B, int F, int D, float *__restrict__ gbias) {
  int blk, nblk;
  if (bias_grad_block(g_y, B, gbias, blk, nblk)) return;
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blk * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)nblk * kWavesPerBlock;
  for (int64_t b = wave0; b < B; b += nwaves) {
    const int64_t base = b * F;
    const float gy = g_y[b];
    for (int f = lane; f < F; f += kWave) g1vals[base + f] = gy;
    for (int d = lane; d < D; d += kWave) {
      float S = 0.f;
      for (int f = 0; f < F; ++f) S += emb[(base + f) * D + d];
      for (int f = 0; f < F; ++f) {
        const int64_t o = (base + f) * D + d;
        gvals[o] = (g_emb ? g_emb[o] : 0.f) + gy * (S - emb[o]);
      }
    }
  }
}

// --------------------------------------------------- backward, dense form ----
// Same gradient rows, scatter-added into gW / gw1.  The wave's RS x D tile of a
// step (256 floats = the lanes' float4 fragments back to back) goes through a
// wave-private 1 KiB LDS slab so that each of the 4 atomic wave-instructions
// covers 64 CONSECUTIVE floats of the tile: whole 256-B runs per row (D >= 64)
// or 64/D whole rows (D < 64) — the shape global float atomics run fastest at
// (MI355X_MICROARCH.md, "Global float atomics").
template <int LPR, int NIT>
__global__ __launch_bounds__(kBlock) void k_gather_fm_bwd_dense(
    const int64_t *__restrict__ rows, const float *__restrict__ emb,
    const float *__restrict__ g_y, const float *__restrict__ g_emb,
    float *__restrict__ gW, float *__restrict__ gw1, int64_t B, int F, int64_t N, float *__restrict__ gbias) {
  constexpr int RS = kWave / LPR;
  constexpr int D = LPR * 4;
  __shared__ float slab[kWavesPerBlock][kWave * 4];
  int blk, nblk;
  if (bias_grad_block(g_y, B, gbias, blk, nblk)) return;
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blk * kWavesPerBlock + wib;
  const int64_t nwaves = (int64_t)nblk * kWavesPerBlock;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float *my = slab[wib];
  constexpr int NSTEP = NIT > 0 ? NIT : 1;

  for (int64_t b = wave0; b < B; b += nwaves) {
    const int64_t base = b * F;
    const float gy = g_y[b];
    float4 S = z;
    for (int f = r; f < F; f += RS) {
      const float4 e = ld4(emb + (base + f) * D + q * 4);
      S.x += e.x; S.y += e.y; S.z += e.z; S.w += e.w;
    }
    S = slot_sum<LPR>(S);
    const int nsteps = NIT > 0 ? NSTEP : (F + RS - 1) / RS;
    for (int k = 0; k < nsteps; ++k) {
      const int f = r + k * RS;
      const bool act = f < F;
      int64_t row = -1;
      float4 o4 = z;
      if (act) {
        const int64_t o = (base + f) * D + q * 4;
        const float4 e = ld4(emb + o);
        const float4 ge = g_emb ? ld4(g_emb + o) : z;
        row = rows[base + f];
        if ((uint64_t)row >= (uint64_t)N) row = -1;
        o4.x = ge.x + gy * (S.x - e.x);
        o4.y = ge.y + gy * (S.y - e.y);
        o4.z = ge.z + gy * (S.z - e.z);
        o4.w = ge.w + gy * (S.w - e.w);
        if (q == 0 && row >= 0) atomicAdd(gw1 + row, gy);
      }
      st4(my + lane * 4, o4);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int el = j * kWave + lane;  // element of the RS x D tile
        const int tr = el / D, tc = el % D;
        const int64_t trow = __shfl(row, tr * LPR);
        const float val = my[el];
        if (trow >= 0) atomicAdd(gW + trow * D + tc, val);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_gather_fm_bwd_dense_anyD(
    const int64_t *__restrict__ rows, const float *__restrict__ emb,
    const float *__restrict__ g_y, const float *__restrict__ g_emb,
    float *__restrict__ gW, float *__restrict__ gw1, int64_t B, int F, int D, int64_t N, float *__restrict__ gbias) {
  int blk, nblk;
  if (bias_grad_block(g_y, B, gbias, blk, nblk)) return;
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blk * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)nblk * kWavesPerBlock;
  for (int64_t b = wave0; b < B; b += nwaves) {
    const int64_t base = b * F;
    const float gy = g_y[b];
    for (int f = lane; f < F; f += kWave) {
      const int64_t row = rows[base + f];
      if ((uint64_t)row < (uint64_t)N) atomicAdd(gw1 + row, gy);
    }
    for (int d = lane; d < D; d += kWave) {
      float S = 0.f;
      for (int f = 0; f < F; ++f) S += emb[(base + f) * D + d];
      for (int f = 0; f < F; ++f) {
        const int64_t o = (base + f) * D + d;
        const int64_t row = rows[base + f];
        if ((uint64_t)row < (uint64_t)N)
          atomicAdd(gW + row * D + d, (g_emb ? g_emb[o] : 0.f) + gy * (S - emb[o]));
      }
    }
  }
}

// ------------------------------------------------- next batch's table lines ----
// Touches the table rows the NEXT batch's lookup will gather (one dword per row is enough: the 128-byte line — a whole
// packed row — comes into the Infinity Cache), so that the next step's forward finds them on-die instead of in HBM.  Meant
// to run on a side stream under an MFMA-bound kernel of the CURRENT step (the tail's weight gradients leave the HBM idle for
// ~45 us).  `sink` is never non-null in practice; it keeps the loads alive.
__global__ __launch_bounds__(kBlock) void k_prefetch_rows(PrefetchJob j, float *__restrict__ sink) {
  prefetch_rows_blocks(j, (int)blockIdx.x, (int)gridDim.x, sink);
}

// ------------------------------------------------------- plain row gather ----
template <int LPR>
__global__ __launch_bounds__(kBlock) void k_gather_rows(
    const int64_t *__restrict__ idx, const float *__restrict__ W, float *__restrict__ out,
    int64_t n, int64_t N, int *err) {
  constexpr int RS = kWave / LPR;
  constexpr int D = LPR * 4;
  constexpr int U = 4;  // row gathers in flight per lane
  const int lane = threadIdx.x & 63;
  const int q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ntiles = (n + (int64_t)RS * U - 1) / ((int64_t)RS * U);
  int bad = 0;
  for (int64_t t = wave0; t < ntiles; t += nwaves) {
    int64_t row[U];
    bool act[U], ok[U];
    float4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t i = (t * U + k) * RS + r;
      act[k] = i < n;
      row[k] = act[k] ? idx[i] : 0;
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      ok[k] = act[k] && (uint64_t)row[k] < (uint64_t)N;
      bad |= (act[k] && !ok[k]);
      v[k] = ok[k] ? ld4(W + row[k] * D + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t i = (t * U + k) * RS + r;
      if (act[k]) st4(out + i * D + q * 4, v[k]);
    }
  }
  if (bad && err) atomicOr(err, MI_IDX_OUT_OF_RANGE);
}

__global__ __launch_bounds__(kBlock) void k_gather_rows_anyD(
    const int64_t *__restrict__ idx, const float *__restrict__ W, float *__restrict__ out,
    int64_t n, int D, int64_t N, int *err) {
  // one thread per output element: coalesced for any D (D = 1 included)
  const int64_t total = n * D;
  int bad = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx[e / D];
    const bool ok = (uint64_t)row < (uint64_t)N;
    bad |= !ok;
    out[e] = ok ? W[row * D + e % D] : 0.f;
  }
  if (bad && err) atomicOr(err, MI_IDX_OUT_OF_RANGE);
}

// gW[idx[i],:] += g[i,:]; same LDS re-tiling as the dense FM backward.
template <int LPR>
__global__ __launch_bounds__(kBlock) void k_scatter_add_rows(
    const int64_t *__restrict__ idx, const float *__restrict__ g, float *__restrict__ gW,
    int64_t n, int64_t N) {
  constexpr int RS = kWave / LPR;
  constexpr int D = LPR * 4;
  __shared__ float slab[kWavesPerBlock][kWave * 4];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int r = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + wib;
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ntiles = (n + RS - 1) / RS;
  float *my = slab[wib];
  for (int64_t t = wave0; t < ntiles; t += nwaves) {
    const int64_t i = t * RS + r;
    int64_t row = -1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
      row = idx[i];
      if ((uint64_t)row >= (uint64_t)N) row = -1;
      // the tile's rows are consecutive in g: lane*4 floats from the tile start
      v = ld4(g + t * RS * D + lane * 4);
    }
    st4(my + lane * 4, v);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int el = j * kWave + lane;
      const int tr = el / D, tc = el % D;
      const int64_t trow = __shfl(row, tr * LPR);
      const float val = my[el];
      if (trow >= 0) atomicAdd(gW + trow * D + tc, val);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(kBlock) void k_scatter_add_rows_anyD(
    const int64_t *__restrict__ idx, const float *__restrict__ g, float *__restrict__ gW,
    int64_t n, int D, int64_t N) {
  const int64_t total = n * D;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx[e / D];
    if ((uint64_t)row < (uint64_t)N) atomicAdd(gW + row * D + e % D, g[e]);
  }
}

// ------------------------------------------------------------- dispatch ------
inline bool vec_ok(int D) { return D >= 4 && D <= 256 && (D & 3) == 0 && ((D >> 2) & ((D >> 2) - 1)) == 0; }
inline int nit_for(int F, int LPR) {
  const int RS = kWave / LPR;
  const int n = (F + RS - 1) / RS;
  return n <= 4 ? n : 0;
}

// Expands `CALL(LPR, NIT)` for the run-time (lpr, nit) pair.
#define MI_DISPATCH_LPR_NIT(lpr, nit, CALL)                               \
  switch (lpr) {                                                          \
    case 1: MI_DISPATCH_NIT(1, nit, CALL); break;                         \
    case 2: MI_DISPATCH_NIT(2, nit, CALL); break;                         \
    case 4: MI_DISPATCH_NIT(4, nit, CALL); break;                         \
    case 8: MI_DISPATCH_NIT(8, nit, CALL); break;                         \
    case 16: MI_DISPATCH_NIT(16, nit, CALL); break;                       \
    case 32: MI_DISPATCH_NIT(32, nit, CALL); break;                       \
    case 64: MI_DISPATCH_NIT(64, nit, CALL); break;                       \
    default: return MI_ERR_UNSUPPORTED;                                   \
  }
#define MI_DISPATCH_NIT(LPR, nit, CALL) \
  switch (nit) {                        \
    case 1: CALL(LPR, 1); break;        \
    case 2: CALL(LPR, 2); break;        \
    case 3: CALL(LPR, 3); break;        \
    case 4: CALL(LPR, 4); break;        \
    default: CALL(LPR, 0); break;       \
  }
#define MI_DISPATCH_LPR(lpr, CALL)      \
  switch (lpr) {                        \
    case 1: CALL(1); break;             \
    case 2: CALL(2); break;             \
    case 4: CALL(4); break;             \
    case 8: CALL(8); break;             \
    case 16: CALL(16); break;           \
    case 32: CALL(32); break;           \
    case 64: CALL(64); break;           \
    default: return MI_ERR_UNSUPPORTED; \
  }

}  // namespace

extern "C" {

int mi_gather_fm_fwd_sum(const int64_t *idx, const int64_t *offsets, const float *W, int64_t ldw, const float *w1,
                         int64_t ldw1, const float *bias, float *emb_out, float *yfm_out, int64_t *rows_out, float *sum_out,
                         int64_t B, int32_t F, int32_t D, int64_t N, int32_t *err, void *stream) {
  if (B < 0 || F < 0 || D <= 0 || N < 0 || ldw < D || ldw1 < 1) return MI_ERR_INVALID_ARG;
  if (B == 0) return MI_OK;
  if (!idx || !offsets || !W || !w1 || !emb_out || !yfm_out) return MI_ERR_INVALID_ARG;
  const int grid = grid_for_waves(B);
  if (vec_ok(D) && aligned16(W) && (ldw & 3) == 0 && aligned16(emb_out)) {
    const int lpr = D / 4, nit = nit_for(F, lpr);
    if (nit > 0 && F <= kWave) {
#define CALL(LPR, NIT)                                                                                \
  MI_LAUNCH("gather_fm_fwd", (k_gather_fm_fwd<LPR, NIT, true>), grid, kBlock, stream, idx, offsets,   \
            W, w1, bias, emb_out, yfm_out, rows_out, B, F, N, ldw, ldw1, err, sum_out)
      MI_DISPATCH_LPR_NIT(lpr, nit, CALL)
#undef CALL
    } else {
#define CALL(LPR, NIT)                                                                               \
  MI_LAUNCH("gather_fm_fwd", (k_gather_fm_fwd<LPR, NIT, false>), grid, kBlock, stream, idx, offsets, \
            W, w1, bias, emb_out, yfm_out, rows_out, B, F, N, ldw, ldw1, err, sum_out)
      MI_DISPATCH_LPR_NIT(lpr, nit, CALL)
#undef CALL
    }
  } else {
    if (ldw != D || ldw1 != 1) return MI_ERR_UNSUPPORTED;      // the scalar fallback reads the reference's two tensors only
    MI_LAUNCH("gather_fm_fwd", k_gather_fm_fwd_anyD, grid, kBlock, stream, idx, offsets, W, w1,
              bias, emb_out, yfm_out, rows_out, B, F, D, N, err, sum_out);
  }
  return launch_status();
}

int mi_gather_fm_fwd_ride(const int64_t *idx, const int64_t *offsets, const float *W, int64_t ldw, const float *w1,
                          int64_t ldw1, const float *bias, float *emb_out, float *yfm_out, int64_t *rows_out, float *sum_out,
                          int64_t B, int32_t F, int32_t D, int64_t N, int32_t *err, const mi_tail_mask_ride *ride,
                          void *stream) {
  if (!ride) return mi_gather_fm_fwd_sum(idx, offsets, W, ldw, w1, ldw1, bias, emb_out, yfm_out, rows_out, sum_out, B, F, D, N, err, stream);
  if (B < 0 || F < 0 || D <= 0 || N < 0 || ldw < D || ldw1 < 1) return MI_ERR_INVALID_ARG;
  MaskRide r;
  int64_t extra = 0;
  const int rc = mask_job(ride->seed, ride->nlayers, ride->salts, ride->ps, ride->lds, ride->bits, ride->M, ride->zero_buf,
                          ride->zero_floats, r.j, &extra);
  if (rc != MI_OK) return rc;
  r.seed = ride->seed;
  r.zero4 = reinterpret_cast<float4 *>(ride->zero_buf);
  r.nzero4 = ride->zero_floats / 4;
  r.aff.nl = 0;
  int aff_blocks = 0;
  if (ride->affine) {
    const mi_tail_affine_job *q = ride->affine;
    const int rc2 = affine_job(q->nlayers, q->widths, q->gamma, q->beta, q->running_mean, q->running_var, q->bias, q->eps, q->mu,
                               q->sc, q->be, q->rstd, r.aff, &aff_blocks);
    if (rc2 != MI_OK) return rc2;
  }
  const int lpr = D / 4, nit = vec_ok(D) ? nit_for(F, lpr) : 0;
  const bool fits = B > 0 && idx && W && w1 && emb_out && yfm_out && vec_ok(D) && aligned16(W) && (ldw & 3) == 0 &&
                    aligned16(emb_out) && nit > 0 && F <= kWave;      // (offsets may be NULL: slot lookups, mi_slot_fm_fwd's operands)
  // (mask workgroups: a quarter of what the job would take alone — each walks four strides — so the launch stays one wave
  //  of workgroups over the chip's slots)
  int nmask = (int)((extra + 3) / 4);
  if (nmask > 1024) nmask = 1024;
  r.mask_blocks = nmask;
  if (nmask + aff_blocks == 0 || !fits) {      // nothing to carry, or a gather form without the extra workgroups: two launches
    if (nmask + aff_blocks > 0) {
      MI_LAUNCH("tail_dropout_masks", k_mask_job, nmask + aff_blocks, kBlock, stream, r);
      const int st = launch_status();
      if (st != MI_OK) return st;
    }
    return mi_gather_fm_fwd_sum(idx, offsets, W, ldw, w1, ldw1, bias, emb_out, yfm_out, rows_out, sum_out, B, F, D, N, err, stream);
  }
  const int ngather = grid_for_waves(B);
#define CALL(LPR, NIT)                                                                                         \
  MI_LAUNCH("gather_fm_fwd_ride", (k_gather_fm_fwd_ride<LPR, NIT>), ngather + nmask + aff_blocks, kBlock, stream, idx, offsets, \
            W, w1, bias, emb_out, yfm_out, rows_out, B, F, N, ldw, ldw1, err, sum_out, ngather, r)
  MI_DISPATCH_LPR_NIT(lpr, nit, CALL)
#undef CALL
  return launch_status();
}

int mi_gather_fm_fwd_ld(const int64_t *idx, const int64_t *offsets, const float *W, int64_t ldw, const float *w1,
                        int64_t ldw1, const float *bias, float *emb_out, float *yfm_out, int64_t *rows_out,
                        int64_t B, int32_t F, int32_t D, int64_t N, int32_t *err, void *stream) {
  return mi_gather_fm_fwd_sum(idx, offsets, W, ldw, w1, ldw1, bias, emb_out, yfm_out, rows_out, nullptr, B, F, D, N, err, stream);
}

int mi_gather_fm_fwd(const int64_t *idx, const int64_t *offsets, const float *W, const float *w1,
                     const float *bias, float *emb_out, float *yfm_out, int64_t *rows_out,
                     int64_t B, int32_t F, int32_t D, int64_t N, int32_t *err, void *stream) {
  return mi_gather_fm_fwd_ld(idx, offsets, W, D, w1, 1, bias, emb_out, yfm_out, rows_out, B, F, D, N, err, stream);
}

int mi_gather_fm_bwd_rows(const float *emb, const float *g_y, const float *g_emb, float *gvals,
                          float *g1vals, float *gbias, int64_t B, int32_t F, int32_t D, void *stream) {
  if (B < 0 || F < 0 || D <= 0) return MI_ERR_INVALID_ARG;
  if (B == 0) return MI_OK;
  if (!emb || !g_y || !gvals || !g1vals) return MI_ERR_INVALID_ARG;
  const int grid = grid_for_waves(B) + (gbias ? 1 : 0);      // + the workgroup that only sums the bias gradient
  if (vec_ok(D) && aligned16(emb) && aligned16(gvals) && (!g_emb || aligned16(g_emb))) {
    const int lpr = D / 4, nit = nit_for(F, lpr);
#define CALL(LPR, NIT)                                                                      \
  MI_LAUNCH("gather_fm_bwd_rows", (k_gather_fm_bwd_rows<LPR, NIT, false>), grid, kBlock, stream, \
            emb, g_y, g_emb, gvals, g1vals, B, F, (const int64_t *)nullptr, (int64_t)0, gbias)
    MI_DISPATCH_LPR_NIT(lpr, nit, CALL)
#undef CALL
  } else {
    MI_LAUNCH("gather_fm_bwd_rows", k_gather_fm_bwd_rows_anyD, grid, kBlock, stream, emb, g_y,
              g_emb, gvals, g1vals, B, F, D, gbias);
  }
  return launch_status();
}

int mi_gather_fm_bwd_dense(const int64_t *rows, const float *emb, const float *g_y,
                           const float *g_emb, float *gW, float *gw1, float *gbias, int64_t B,
                           int32_t F, int32_t D, int64_t N, void *stream) {
  if (B < 0 || F < 0 || D <= 0 || N < 0) return MI_ERR_INVALID_ARG;
  if (B == 0) return MI_OK;
  if (!rows || !emb || !g_y || !gW || !gw1) return MI_ERR_INVALID_ARG;
  const int grid = grid_for_waves(B) + (gbias ? 1 : 0);      // + the workgroup that only sums the bias gradient
  if (vec_ok(D) && aligned16(emb) && (!g_emb || aligned16(g_emb))) {
    const int lpr = D / 4, nit = nit_for(F, lpr);
#define CALL(LPR, NIT)                                                                         \
  MI_LAUNCH("gather_fm_bwd_dense", (k_gather_fm_bwd_dense<LPR, NIT>), grid, kBlock, stream, rows, \
            emb, g_y, g_emb, gW, gw1, B, F, N, gbias)
    MI_DISPATCH_LPR_NIT(lpr, nit, CALL)
#undef CALL
  } else {
    MI_LAUNCH("gather_fm_bwd_dense", k_gather_fm_bwd_dense_anyD, grid, kBlock, stream, rows, emb,
              g_y, g_emb, gW, gw1, B, F, D, N, gbias);
  }
  return launch_status();
}

int mi_prefetch_rows(const int64_t *idx, const int64_t *offsets, const float *W, int64_t ldw, const float *w1, int64_t ldw1,
                     int64_t B, int32_t F, int64_t N, void *stream) {
  if (B < 0 || F <= 0 || N < 0 || ldw <= 0) return MI_ERR_INVALID_ARG;
  if (B == 0) return MI_OK;
  if (!idx || !W) return MI_ERR_INVALID_ARG;
  PrefetchJob j;
  if (!prefetch_job(idx, offsets, W, ldw, w1, ldw1, B, F, N, j)) return MI_OK;
  int64_t grid = (j.n + kBlock - 1) / kBlock;
  if (grid > kMaxGrid) grid = kMaxGrid;
  MI_LAUNCH("prefetch_rows", k_prefetch_rows, (int)grid, kBlock, stream, j, (float *)nullptr);
  return launch_status();
}

int mi_gather_rows_fwd(const int64_t *idx, const float *W, float *out, int64_t n, int32_t D,
                       int64_t N, int32_t *err, void *stream) {
  if (n < 0 || D <= 0 || N < 0) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!idx || !W || !out) return MI_ERR_INVALID_ARG;
  if (vec_ok(D) && aligned16(W) && aligned16(out)) {
    const int lpr = D / 4;
    const int64_t tiles = (n + (int64_t)(kWave / lpr) * 4 - 1) / ((int64_t)(kWave / lpr) * 4);
    const int grid = grid_for_waves(tiles);
#define CALL(LPR) \
  MI_LAUNCH("gather_rows", (k_gather_rows<LPR>), grid, kBlock, stream, idx, W, out, n, N, err)
    MI_DISPATCH_LPR(lpr, CALL)
#undef CALL
  } else {
    MI_LAUNCH("gather_rows", k_gather_rows_anyD, grid_for_waves(((int64_t)n * D + kWave - 1) / kWave),
              kBlock, stream, idx, W, out, n, D, N, err);
  }
  return launch_status();
}

int mi_scatter_add_rows(const int64_t *idx, const float *g, float *gW, int64_t n, int32_t D,
                        int64_t N, void *stream) {
  if (n < 0 || D <= 0 || N < 0) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!idx || !g || !gW) return MI_ERR_INVALID_ARG;
  if (vec_ok(D) && aligned16(g)) {
    const int lpr = D / 4;
    const int64_t tiles = (n + (kWave / lpr) - 1) / (kWave / lpr);
    const int grid = grid_for_waves(tiles);
#define CALL(LPR) \
  MI_LAUNCH("scatter_add_rows", (k_scatter_add_rows<LPR>), grid, kBlock, stream, idx, g, gW, n, N)
    MI_DISPATCH_LPR(lpr, CALL)
#undef CALL
  } else {
    MI_LAUNCH("scatter_add_rows", k_scatter_add_rows_anyD,
              grid_for_waves(((int64_t)n * D + kWave - 1) / kWave), kBlock, stream, idx, g, gW, n, D, N);
  }
  return launch_status();
}

// ---- the same two kernels over the packed rows a sharded lookup received (route.hip) ----
int mi_slot_fm_fwd(const int64_t *slot, const float *buf, int64_t nrows, const float *bias,
                   float *emb_out, float *yfm_out, int64_t B, int32_t F, int32_t D, int32_t *err,
                   void *stream) {
  if (B < 0 || F < 0 || D <= 0 || nrows < 0) return MI_ERR_INVALID_ARG;
  if (B == 0) return MI_OK;
  if (!slot || !buf || !emb_out || !yfm_out) return MI_ERR_INVALID_ARG;
  if (!vec_ok(D) || !aligned16(buf) || !aligned16(emb_out)) return MI_ERR_UNSUPPORTED;
  const int grid = grid_for_waves(B);
  const int lpr = D / 4, nit = nit_for(F, lpr);
  const int64_t ld = D + 4;
  if (nit > 0 && F <= kWave) {
#define CALL(LPR, NIT)                                                                                 \
  MI_LAUNCH("slot_fm_fwd", (k_gather_fm_fwd<LPR, NIT, true>), grid, kBlock, stream, slot,              \
            (const int64_t *)nullptr, buf, buf + D, bias, emb_out, yfm_out, (int64_t *)nullptr,        \
            B, F, nrows, ld, ld, err, (float *)nullptr)
    MI_DISPATCH_LPR_NIT(lpr, nit, CALL)
#undef CALL
  } else {
#define CALL(LPR, NIT)                                                                               \
  MI_LAUNCH("slot_fm_fwd", (k_gather_fm_fwd<LPR, NIT, false>), grid, kBlock, stream, slot,           \
            (const int64_t *)nullptr, buf, buf + D, bias, emb_out, yfm_out, (int64_t *)nullptr,      \
            B, F, nrows, ld, ld, err, (float *)nullptr)
    MI_DISPATCH_LPR_NIT(lpr, nit, CALL)
#undef CALL
  }
  return launch_status();
}

int mi_slot_fm_bwd(const int64_t *slot, const float *emb, const float *g_y, const float *g_emb,
                   float *gbuf, float *gbias, int64_t nslot, int64_t B, int32_t F, int32_t D, void *stream) {
  if (B < 0 || F < 0 || D <= 0 || nslot < 0) return MI_ERR_INVALID_ARG;
  if (nslot > 0 && !gbuf) return MI_ERR_INVALID_ARG;
  if (!vec_ok(D) || !aligned16(gbuf) || (emb && !aligned16(emb)) || (g_emb && !aligned16(g_emb)))
    return MI_ERR_UNSUPPORTED;
  // padding slots must read as zero gradients at the owner
  if (nslot > 0 &&
      hipMemsetAsync(gbuf, 0, (size_t)nslot * (D + 4) * sizeof(float), (hipStream_t)stream) != hipSuccess)
    return MI_ERR_LAUNCH;
  if (B == 0) return MI_OK;
  if (!slot || !emb || !g_y) return MI_ERR_INVALID_ARG;
  const int grid = grid_for_waves(B) + (gbias ? 1 : 0);      // + the workgroup that only sums the bias gradient
  const int lpr = D / 4, nit = nit_for(F, lpr);
#define CALL(LPR, NIT)                                                                         \
  MI_LAUNCH("slot_fm_bwd", (k_gather_fm_bwd_rows<LPR, NIT, true>), grid, kBlock, stream, emb,  \
            g_y, g_emb, gbuf, (float *)nullptr, B, F, slot, nslot, gbias)
  MI_DISPATCH_LPR_NIT(lpr, nit, CALL)
#undef CALL
  return launch_status();
}

}  // extern "C"
