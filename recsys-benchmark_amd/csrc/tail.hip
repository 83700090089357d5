// tail.hip — the MLP tail of DeepFM / DCN as fused gfx950 kernels:
//   (Linear, BatchNorm1d, ReLU, Dropout) x k + Linear(., 1)          src/models/deepfm.py:53-66,100-102
// Round 1 ran 9 library GEMMs + 12 BatchNorm-family passes per step.  Here every contraction is tail_gemm.hpp's
// 256-workgroup MFMA kernel and the element work rides inside it:
//   forward  layer l : z_l = a_{l-1} W_l^T with a_{l-1} = dropout(relu(bn(z_{l-1}))) recomputed in the operand load;
//                      the epilogue stores z_l and per-workgroup column statistics (mean, M2 over its 64 rows);
//                      k_bn_finalize_fwd merges them (Chan, fixed order: deterministic) into mu / gamma*rstd / beta,
//                      updates the running statistics and bumps the dropout seed.
//   head             : out = a_k . w + b (+ y_fm) — one pass over z_k; backward of the head is one pass too
//                      (dy_k, its column sums, dw, db).
//   backward layer l : dz_l = al*dy + bz*(z - mu) + de in the operand loads of BOTH products;
//                      dgrad epilogue turns da_{l-1} into dy_{l-1} (ReLU / dropout mask recomputed) and emits the column
//                      sums for dgamma / dbeta; wgrad reduces over the batch in K-slices whose slabs a second kernel adds
//                      in slice order (no atomics anywhere: the step is bit-reproducible).
// The dropout decisions are ONE BIT per element, written once per step (k_tail_dropmask) and read by every kernel that
// needs them (a byte per float4 of features).
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "tail_gemm.hpp"
#include "tail_masks.hpp"

namespace {
using namespace mi;
using namespace tg;

__device__ __forceinline__ Drop make_drop(const uint8_t *bits, float p, int ld) {
  Drop d;
  d.bits = p > 0.f ? bits : nullptr;
  d.inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  d.ld = ld;
  return d;
}

__global__ __launch_bounds__(kBlock) void k_tail_dropmask(MaskJob j, const int64_t *seed, float4 *__restrict__ zero4, int64_t nzero4) {
  mask_blocks(j, seed, zero4, nzero4, blockIdx.x, gridDim.x);
}

// The consumers' accumulators as a [64][kTilePitch] tile in LDS: lane (r, g) of wave w holds rows 16 w + r, columns
// 16 s + 4 g .. + 3 of sub-tile s.
constexpr int kTilePitch = BNT + 4;
__device__ __forceinline__ void acc_to_lds(const floatx4 (&acc)[NSUB], float *T, int wave, int lane) {
  if (wave < 4) {
    const int r = lane & 15, g = lane >> 4;
    float *row = T + (wave * 16 + r) * kTilePitch + 4 * g;
#pragma unroll
    for (int s = 0; s < NSUB; ++s) st4(row + 16 * s, make_float4(acc[s][0], acc[s][1], acc[s][2], acc[s][3]));
  }
}

// 16-lane (one MFMA row group) sum: lanes that share lane / 16
__device__ __forceinline__ float sum16(float v) {
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
  return v;
}

struct ActDesc {      // how to recompute an activation from its saved pre-activation (all null/0: the matrix is used as is)
  const float *Z;     // [M, ld]
  int ld;
  const float *mu, *sc, *be;
  float p;
  const uint8_t *keep;   // the layer's keep bits (k_tail_dropmask), used when p > 0
};
__device__ __forceinline__ LoadAct make_act(const ActDesc &d) {
  LoadAct a;
  a.Z = d.Z; a.ld = d.ld; a.mu = d.mu; a.sc = d.sc; a.be = d.be;
  a.drop = make_drop(d.keep, d.p, d.ld);
  return a;
}
struct DzDesc {       // dz = al*dy + bz*(z - mu) + de; PLAIN (al == null): the matrix DY itself
  const float *DY, *Z;
  int ld;
  const float *mu, *al, *bz, *de;
};
__device__ __forceinline__ LoadDz make_dz(const DzDesc &d) {
  LoadDz l;
  l.DY = d.DY; l.Z = d.Z; l.ld = d.ld; l.mu = d.mu; l.al = d.al; l.bz = d.bz; l.de = d.de;
  return l;
}


// ============================================================================ BatchNorm joins in the CONSUMER's prologue ====
// Round 2 joined the per-tile statistics of a product in a launch of their own (k_bn_finalize_fwd / _bwd: six near-empty
// kernels per step, ~5.5 us each, almost all of it the kernel boundary).  Here every workgroup of the NEXT kernel joins
// them itself, in a fixed order, into LDS while its first operand loads are in flight: all workgroups compute the same
// bits (same order, same arithmetic), workgroup 0 alone writes the constants the backward pass reads later, the running
// statistics and the counters.  No atomics, no tickets, deterministic.  Constants live in LDS as [row][kCstPitch].
constexpr int kCstPitch = 1024;                 // features <= 1024 (tail.py's plan and the entry points check it)
constexpr int kCstFloats = 4 * kCstPitch;

struct BnFwd {           // device view of mi_tail_bn_fwd
  const float *part, *gamma, *beta, *mean_offset;
  float *running_mean, *running_var;
  int64_t *nbt, *seed_bump;
  float *mu, *sc, *be, *rstd;
  float momentum, eps;
  int nrep;              // > 0: `part` holds SHIFTED SUMS double[nrep][2][N] accumulated by f64 atomics (below), not tile statistics
  const float *shift;    // [N] the shift they were taken around
};
// ---- statistics as shifted sums (round 4) ----------------------------------------------------------------------------
// A product's epilogue adds, per column c, t1 = sum_m (z - s_c) and t2 = sum_m (z - s_c)^2 of its 64-row tile (formed from
// the tile's exact (n, mean, M2)) into sums[r][0][c] / sums[r][1][c] — DOUBLES — with f64 atomics (r = row tile % nrep: replicas spread the same-address traffic; each
// atomic wave-instruction covers 64 consecutive floats), s_c = the layer's running mean minus the Linear bias the product
// left out — a value near the batch mean after a few steps, and any value is algebraically exact:
//   mean = s + t1 / M,   M2 = t2 - t1^2 / M
// (the cancellation in M2 is of relative size (mean - s)^2 / var, which is O(1) for a BatchNorm pre-activation).  The
// CONSUMING kernel derives mu / gamma rstd / beta from (4 nrep + 3) N floats in its prologue — 14 to 30 KB, one round trip,
// against the 205 KB of tile statistics round 3's joined form pulled through every CU — and no finalize launch exists.
// Float-atomic order makes the statistics vary in the last bits between runs: deterministic mode keeps the tile
// statistics and the finalize launches.
template <int NT>
__device__ __forceinline__ void bn_derive_fwd(const BnFwd &b, int M, int N, float *cst, bool writer, int tid) {
  const double inv_m = 1.0 / (double)M;
  const double *P = reinterpret_cast<const double *>(b.part);
  for (int n = tid * 2; n < N; n += NT * 2) {
    double t1x = 0.0, t1y = 0.0, t2x = 0.0, t2y = 0.0;
    for (int r = 0; r < b.nrep; ++r) {
      const double2 a = *reinterpret_cast<const double2 *>(P + (int64_t)(2 * r) * N + n);
      const double2 q = *reinterpret_cast<const double2 *>(P + (int64_t)(2 * r + 1) * N + n);
      t1x += a.x; t1y += a.y; t2x += q.x; t2y += q.y;
    }
    const float2 sh = *reinterpret_cast<const float2 *>(b.shift + n);
    const float2 gm2 = b.gamma ? *reinterpret_cast<const float2 *>(b.gamma + n) : make_float2(1.f, 1.f);
    const float2 bt2 = b.beta ? *reinterpret_cast<const float2 *>(b.beta + n) : make_float2(0.f, 0.f);
    float2 mo2 = make_float2(0.f, 0.f), rm2 = mo2, rv2 = mo2;
    if (writer && b.running_mean) {
      rm2 = *reinterpret_cast<const float2 *>(b.running_mean + n);
      rv2 = *reinterpret_cast<const float2 *>(b.running_var + n);
      if (b.mean_offset) mo2 = *reinterpret_cast<const float2 *>(b.mean_offset + n);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n + j;
      const double s1 = j ? t1y : t1x, s2 = j ? t2y : t2x;
      const double dm = s1 * inv_m;
      const float mean = (float)((double)(j ? sh.y : sh.x) + dm);
      const float m2 = fmaxf((float)(s2 - s1 * dm), 0.f);
      const float var = m2 * (float)inv_m;
      const float rstd = rsqrtf(var + b.eps);
      const float gm = j ? gm2.y : gm2.x, bt = j ? bt2.y : bt2.x;
      cst[col] = mean;
      cst[kCstPitch + col] = gm * rstd;
      cst[2 * kCstPitch + col] = bt;
      if (writer) {
        b.mu[col] = mean;
        b.sc[col] = gm * rstd;
        b.be[col] = bt;
        b.rstd[col] = rstd;
        if (b.running_mean) {
          b.running_mean[col] = (1.f - b.momentum) * (j ? rm2.y : rm2.x) + b.momentum * (mean + (j ? mo2.y : mo2.x));
          const float unb = M > 1 ? m2 / (float)(M - 1) : var;
          b.running_var[col] = (1.f - b.momentum) * (j ? rv2.y : rv2.x) + b.momentum * unb;
        }
      }
    }
  }
  if (writer && tid == 0) {
    if (b.nbt) b.nbt[0] += 1;
    if (b.seed_bump) b.seed_bump[0] += 1;
  }
}
// (mean, M2) of the 64-row tiles -> batch mean and variance in ONE pass with every load of a chunk in flight (a thread
// that walks 64 tiles 8 at a time pays 8 dependent L2 round trips: measured +10 us per kernel).  The tile means are
// shifted by tile 0's mean c before they are summed and squared (tile means differ from the grand mean by ~sigma / 8,
// so the subtraction at the end cancels nothing that matters):
//   d_t = mean_t - c;  mean = c + sum n_t d_t / M;  M2 = sum M2_t + sum n_t d_t^2 - (sum n_t d_t)^2 / M
// A thread joins TWO adjacent columns (one float4 = (mean, M2) x 2 per tile), kMergeChunk tiles per round trip.
// `tid` / NT: this thread's index among, and the number of, the threads that take part.  cst rows: 0 mu, 1 sc, 2 be.
constexpr int kMergeChunk = 32;
template <int NT>
__device__ __forceinline__ void bn_merge_fwd(const BnFwd &b, int M, int N, float *cst, bool writer, int tid) {
  if (b.nrep > 0) {
    bn_derive_fwd<NT>(b, M, N, cst, writer, tid);
    return;
  }
  const int MT = (M + BM - 1) / BM;
  const float inv_m = 1.f / (float)M;
  for (int n = tid * 2; n < N; n += NT * 2) {
    const float *p = b.part + (int64_t)n * 2;
    // per-column inputs of the last step: issued with the first chunk, not after it
    const float2 gm2 = b.gamma ? *reinterpret_cast<const float2 *>(b.gamma + n) : make_float2(1.f, 1.f);
    const float2 bt2 = b.beta ? *reinterpret_cast<const float2 *>(b.beta + n) : make_float2(0.f, 0.f);
    float2 mo2 = make_float2(0.f, 0.f), rm2 = mo2, rv2 = mo2;
    if (writer && b.running_mean) {
      rm2 = *reinterpret_cast<const float2 *>(b.running_mean + n);
      rv2 = *reinterpret_cast<const float2 *>(b.running_var + n);
      if (b.mean_offset) mo2 = *reinterpret_cast<const float2 *>(b.mean_offset + n);
    }
    float c0 = 0.f, c1 = 0.f, sn0 = 0.f, sn1 = 0.f, sq0 = 0.f, sq1 = 0.f, sm0 = 0.f, sm1 = 0.f;
    for (int base = 0; base < MT; base += kMergeChunk) {
      float4 v[kMergeChunk];
#pragma unroll
      for (int u = 0; u < kMergeChunk; ++u) v[u] = ld4(p + (int64_t)min(base + u, MT - 1) * N * 2);
      if (base == 0) { c0 = v[0].x; c1 = v[0].z; }
#pragma unroll
      for (int u = 0; u < kMergeChunk; ++u) {
        const int t = base + u;
        if (t < MT) {
          const float nb = (float)min(BM, M - t * BM);
          const float d0 = v[u].x - c0, d1 = v[u].z - c1;
          sn0 = fmaf(nb, d0, sn0); sq0 = fmaf(nb * d0, d0, sq0); sm0 += v[u].y;
          sn1 = fmaf(nb, d1, sn1); sq1 = fmaf(nb * d1, d1, sq1); sm1 += v[u].w;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n + j;
      const float c = j ? c1 : c0, sn = j ? sn1 : sn0, sq = j ? sq1 : sq0, sm = j ? sm1 : sm0;
      const float mean = c + sn * inv_m;
      const float m2 = fmaxf(sm + (sq - sn * sn * inv_m), 0.f);
      const float var = m2 * inv_m;
      const float rstd = rsqrtf(var + b.eps);
      const float gm = j ? gm2.y : gm2.x, bt = j ? bt2.y : bt2.x;
      cst[col] = mean;
      cst[kCstPitch + col] = gm * rstd;
      cst[2 * kCstPitch + col] = bt;
      if (writer) {
        b.mu[col] = mean;
        b.sc[col] = gm * rstd;
        b.be[col] = bt;
        b.rstd[col] = rstd;
        if (b.running_mean) {
          b.running_mean[col] = (1.f - b.momentum) * (j ? rm2.y : rm2.x) + b.momentum * (mean + (j ? mo2.y : mo2.x));
          const float unb = M > 1 ? m2 / (float)(M - 1) : var;
          b.running_var[col] = (1.f - b.momentum) * (j ? rv2.y : rv2.x) + b.momentum * unb;
        }
      }
    }
  }
  if (writer && tid == 0) {
    if (b.nbt) b.nbt[0] += 1;
    if (b.seed_bump) b.seed_bump[0] += 1;
  }
}

struct BnBwd {           // device view of mi_tail_bn_bwd
  const float *part;     // [nblk, N, 2] (sum dy, sum dy (z - mu))
  int nblk;
  const float *gamma, *rstd, *mu;
  float *dgamma, *dbeta, *al, *bz, *de;
  const float *wpart;    // [nwblk, N + 4] head pieces, nullable
  int nwblk;
  float *dw, *db;
  float *db2;            // nullable: a second destination of db (DeepFM's scalar bias gets the head bias's gradient: no copy launch)
  // affine != 0: the layer normalised with FIXED statistics (eval-mode BatchNorm: rstd of the running variance) or not at
  // all (no BatchNorm: gamma null, rstd = 1): dz = al dy with al = gamma rstd, no batch terms (bz = de = 0); dbias
  // (nullable) receives the Linear bias's gradient sum_m dz = al * sum_m dy (under a training-mode BatchNorm it is zero)
  int affine;
  float *dbias;
};
// cst rows: 0 mu, 1 al, 2 bz, 3 de (the constants of LoadDz).  Two adjacent columns per thread, kMergeChunk partial rows in
// flight.  The head's dw / db pieces are joined by ANOTHER workgroup (wwriter) so that no workgroup carries both joins.
template <int NT, int CH>
__device__ __forceinline__ void bn_merge_bwd_impl(const BnBwd &b, int M, int N, float *cst, bool writer, bool wwriter, int tid);
// few partial rows (the replicas of an atomically accumulated sum, mi_tail_dgrad_gemm_s): 8 loads per trip instead of 32
template <int NT>
__device__ __forceinline__ void bn_merge_bwd(const BnBwd &b, int M, int N, float *cst, bool writer, bool wwriter, int tid) {
  if (b.nblk <= 8 && (!b.wpart || b.nwblk <= 8)) bn_merge_bwd_impl<NT, 8>(b, M, N, cst, writer, wwriter, tid);
  else if (b.nblk <= 16 && (!b.wpart || b.nwblk <= 16)) bn_merge_bwd_impl<NT, 16>(b, M, N, cst, writer, wwriter, tid);
  else bn_merge_bwd_impl<NT, kMergeChunk>(b, M, N, cst, writer, wwriter, tid);
}
template <int NT, int CH>
__device__ __forceinline__ void bn_merge_bwd_impl(const BnBwd &b, int M, int N, float *cst, bool writer, bool wwriter, int tid) {
  const float inv_m = 1.f / (float)M;
  for (int n = tid * 2; n < N; n += NT * 2) {
    const float *p = b.part + (int64_t)n * 2;
    const float2 r2 = *reinterpret_cast<const float2 *>(b.rstd + n), mu2 = *reinterpret_cast<const float2 *>(b.mu + n);
    const float2 gm2 = b.gamma ? *reinterpret_cast<const float2 *>(b.gamma + n) : make_float2(1.f, 1.f);
    float s10 = 0.f, s20 = 0.f, s11 = 0.f, s21 = 0.f;
    for (int base = 0; base < b.nblk; base += CH) {
      float4 v[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) v[u] = ld4(p + (int64_t)min(base + u, b.nblk - 1) * N * 2);
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (base + u < b.nblk) { s10 += v[u].x; s20 += v[u].y; s11 += v[u].z; s21 += v[u].w; }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n + j;
      const float s1 = j ? s11 : s10, s2 = j ? s21 : s20;
      const float r = j ? r2.y : r2.x, gm = j ? gm2.y : gm2.x;
      const float dg = s2 * r;
      const float al = gm * r, bz = b.affine ? 0.f : -gm * r * r * dg * inv_m, de = b.affine ? 0.f : -gm * r * s1 * inv_m;
      cst[col] = j ? mu2.y : mu2.x;
      cst[kCstPitch + col] = al;
      cst[2 * kCstPitch + col] = bz;
      cst[3 * kCstPitch + col] = de;
      if (writer) {
        if (b.dgamma) b.dgamma[col] = dg;
        if (b.dbeta) b.dbeta[col] = s1;
        if (b.affine && b.dbias) b.dbias[col] = al * s1;
        b.al[col] = al; b.bz[col] = bz; b.de[col] = de;
      }
    }
  }
  if (wwriter && b.wpart) {
    for (int n = tid; n <= N; n += NT) {
      float sw = 0.f;
      for (int base = 0; base < b.nwblk; base += CH) {
        float v[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) v[u] = b.wpart[(int64_t)min(base + u, b.nwblk - 1) * (N + 4) + n];
#pragma unroll
        for (int u = 0; u < CH; ++u)
          if (base + u < b.nwblk) sw += v[u];
      }
      if (n < N) b.dw[n] = sw;
      else {
        if (b.db) b.db[0] = sw;
        if (b.db2) b.db2[0] = sw;
      }
    }
  }
}
// generic -> LDS address space (p must point into a __shared__ array)
__device__ __forceinline__ lds_cfp as_lds(const float *p) { return (lds_cfp)p; }

// ============================================================================================== forward GEMM ====
struct FwdArgs {
  ActDesc x;          // R operand [M, K]
  const float *W;     // [N, K]
  int ldw;
  float *Z;           // [M, N] out
  int ldz;
  float *part;        // [MT, N, 2] (mean, M2) of every 64-row tile, nullable
  float *a_out;       // [M, K] (pitch x.ld), nullable: the activation a(X) as the operand load computed it
  int M, N, K;
  int ncols;          // columns per workgroup (multiple of 4, <= 112)
  int ntn;            // column tiles
  BnFwd bn;           // MERGE: the statistics behind x's constants, joined here (x.mu / sc / be are then not read)
  // stat_rep > 0: `part` is the shifted-sum accumulator [stat_rep][2][N] (zeroed by the caller); the epilogue adds this
  // tile's sums with float atomics around shift = stat_rm - stat_off (either nullable: 0) and row tile 0 stores the shift
  int stat_rep;
  float *stat_shift;
  const float *stat_rm, *stat_off;
  // head_w != null (an INFERENCE forward's last hidden layer: fixed statistics, no dropout, nothing kept for a backward): the
  // head Linear(., 1) runs in this epilogue — a tile's 64 x ncols values go through relu((z - mu) sc + be), are dotted with
  // its slice of the head's weight and ADDED into head_out[m] (zeroed by an earlier launch; column tile 0 also adds the
  // bias and head_add[m]); Z is not stored.  One launch less per forward, 4 float atomics per logit.
  const float *head_w, *head_b, *head_add, *h_mu, *h_sc, *h_be;
  float *head_out;
};

// NS: the 16-column sub-tiles a workgroup computes (7 = up to 112 columns: the training shapes, one workgroup per CU at
// M = 4096).  NS = 2 (32 columns) is the SMALL-BATCH form: an inference forward at the reference's default batch 64
// (scripts/deepfm/infer_deepfm.py:36) is ONE row tile — 4 workgroups of 112 columns leave 252 CUs idle and each still walks
// 56 MFMAs per slice; 13 workgroups of 32 columns walk 16 (plain forms only: no statistics joins at that size).
template <bool ACT, bool MERGE, bool DMA = false, int NS = NSUB>
__global__ __launch_bounds__(kThreads) void k_tail_fwd(FwdArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[DMA ? kDmaLdsFloats : kLdsFloats + (MERGE ? kCstFloats : 0)];
  const int mt_total = (a.M + BM - 1) / BM;
  const int tile = xcd_logical(blockIdx.x, mt_total * a.ntn);
  if (tile < 0) return;

  const int mt = tile / a.ntn, nt = tile % a.ntn;
  const int m0 = mt * BM, n0 = nt * a.ncols;
  const int rows_valid = min(BM, a.M - m0), cols_valid = min(a.ncols, a.N - n0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  floatx4 acc[NSUB];
#pragma unroll
  for (int s = 0; s < NSUB; ++s) acc[s] = floatx4{0.f, 0.f, 0.f, 0.f};

  const LoadAct act = make_act(a.x);
  const LoadPlain xp{a.x.Z, a.x.ld};
  const LoadPlain wp{a.W, a.ldw};
  const KcOperand<NS * 16, LoadPlain> opC{wp, n0, cols_valid, a.K};
  if constexpr (MERGE) {
    const lds_cfp c = as_lds(lds + kLdsFloats);
    const LoadActL actl{a.x.Z, a.x.ld, c, c + kCstPitch, c + 2 * kCstPitch, act.drop};
    float *cst = lds + kLdsFloats;
    const bool writer = tile == 0;
    main_loop<true, true>(acc, lds, 0, a.K, KcOperand<64, Tee<LoadActL>>{Tee<LoadActL>{actl, nt == 0 ? a.a_out : nullptr, a.x.ld}, m0, rows_valid, a.K}, opC,
                          make_pre([&]() { bn_merge_fwd<kThreads - kProd>(a.bn, a.M, a.K, cst, writer, (int)threadIdx.x); }));
  } else if constexpr (DMA && ACT)      // the weights by LDS-DMA (two loader waves), the activation through two producer waves
    main_loop_dma(acc, lds, 0, a.K, KcOperand<64, Tee<LoadAct>, 128>{Tee<LoadAct>{act, nt == 0 ? a.a_out : nullptr, a.x.ld}, m0, rows_valid, a.K},
                  a.W, a.ldw, n0, cols_valid);
  else if constexpr (DMA)
    main_loop_dma(acc, lds, 0, a.K, KcOperand<64, LoadPlain, 128>{xp, m0, rows_valid, a.K}, a.W, a.ldw, n0, cols_valid);
  else if constexpr (ACT)
    main_loop<true, true, NS>(acc, lds, 0, a.K, KcOperand<64, Tee<LoadAct>>{Tee<LoadAct>{act, nt == 0 ? a.a_out : nullptr, a.x.ld}, m0, rows_valid, a.K}, opC);
  else main_loop<true, true, NS>(acc, lds, 0, a.K, KcOperand<64, LoadPlain>{xp, m0, rows_valid, a.K}, opC);

  // ---- epilogue.  A lane holds z[m0 + 16 wave + r][n0 + 16 s + 4 g + v] (waves 0-3); the tile goes through LDS once so
  // that ALL 8 waves store whole 448-byte row segments and the column statistics are plain column walks.
  float *T = lds;                                        // [64][kTilePitch]
  acc_to_lds(acc, T, wave, lane);
  __syncthreads();
  const int t = threadIdx.x;
  if (a.head_w) {      // thread (row = t / 8, part = t % 8): float4 chunks part, part + 8, part + 16, part + 24 of the row's tile
    const int row = t >> 3, part = t & 7;
    float sum = 0.f;
    if (row < rows_valid) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = (part + 8 * j) * 4;
        if (c < cols_valid) {
          const float4 z = ld4(T + row * kTilePitch + c);
          const float4 u = ld4(a.h_mu + n0 + c), sc = ld4(a.h_sc + n0 + c), be = ld4(a.h_be + n0 + c), w = ld4(a.head_w + n0 + c);
          sum += fmaxf(fmaf(z.x - u.x, sc.x, be.x), 0.f) * w.x + fmaxf(fmaf(z.y - u.y, sc.y, be.y), 0.f) * w.y +
                 fmaxf(fmaf(z.z - u.z, sc.z, be.z), 0.f) * w.z + fmaxf(fmaf(z.w - u.w, sc.w, be.w), 0.f) * w.w;
        }
      }
    }
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    sum += __shfl_xor(sum, 4);
    if (part == 0 && row < rows_valid) {
      if (nt == 0) sum += (a.head_b ? a.head_b[0] : 0.f) + (a.head_add ? a.head_add[m0 + row] : 0.f);
      atomicAdd(a.head_out + m0 + row, sum);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = t + k * kThreads, row = i / 28, c = (i % 28) * 4;
    if (i < 64 * 28 && row < rows_valid && c < cols_valid)
      st4(a.Z + (int64_t)(m0 + row) * a.ldz + n0 + c, ld4(T + row * kTilePitch + c));
  }
  if (!a.part) return;
  float stat_s = 0.f;
  if (a.stat_rep > 0 && t < cols_valid) {
    if (a.stat_rm) stat_s = a.stat_rm[n0 + t];
    if (a.stat_off) stat_s -= a.stat_off[n0 + t];
  }
  // column statistics: thread (rg = t / 128, col = t % 128) walks rows 16 rg .. 16 rg + 15 of its column with shifted
  // sums around the group's first row (a value of the column itself: no cancellation); one thread per column then
  // merges the 4 groups (Chan) — the same grouping and order for every launch: deterministic.
  float *ws = lds + 64 * kTilePitch;                     // [4][BNT][3]
  {
    const int col = t & 127, rg = t >> 7;
    const int cnt = max(0, min(16, a.M - (m0 + rg * 16)));
    if (col < cols_valid && cnt > 0) {
      const float *p = T + rg * 16 * kTilePitch + col;
      const float shift = p[0];
      float s1 = 0.f, s2 = 0.f;
      for (int rr = 1; rr < cnt; ++rr) {
        const float d = p[rr * kTilePitch] - shift;
        s1 += d;
        s2 += d * d;
      }
      float *o = ws + (rg * BNT + col) * 3;
      o[0] = shift; o[1] = s1; o[2] = s2;
    }
  }
  __syncthreads();
  if (t < cols_valid) {
    float n_a = 0.f, mean_a = 0.f, m2_a = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int cw = max(0, min(16, a.M - (m0 + w * 16)));
      if (cw == 0) continue;
      const float *o = ws + (w * BNT + t) * 3;
      const float n_b = (float)cw, mean_b = o[0] + o[1] / n_b, m2_b = o[2] - o[1] * o[1] / n_b;
      const float n = n_a + n_b, delta = mean_b - mean_a;
      mean_a += delta * (n_b / n);
      m2_a += m2_b + delta * delta * (n_a * n_b / n);
      n_a = n;
    }
    if (a.stat_rep > 0) {
      // the tile's (n, mean, M2) are exact to fp32 around the tile's own values; what meets the other tiles' is kept in
      // DOUBLE (global_atomic_add_f64): sum n d and sum (n d^2 + M2) with d = mean - shift cancel against each other by
      // (mean - shift)^2 / var when the batch variance is derived, and a first layer fed by freshly initialised (tiny)
      // embeddings has that ratio in the hundreds (fp32 sums: logits off by 9e-5 against the oracle at step 1)
      const double d = (double)mean_a - (double)stat_s;
      double *o = reinterpret_cast<double *>(a.part) + (int64_t)((mt % a.stat_rep) * 2) * a.N + n0 + t;
      atomicAdd(o, (double)n_a * d);
      atomicAdd(o + a.N, (double)n_a * d * d + (double)m2_a);
      if (mt == 0) a.stat_shift[n0 + t] = stat_s;
    } else {
      float *o = a.part + ((int64_t)mt * a.N + n0 + t) * 2;
      o[0] = mean_a;
      o[1] = m2_a;
    }
  }
}

// Merge the per-tile statistics (fixed order), produce the constants the next loads need, update the running statistics
// like F.batch_norm(training=True) does (momentum; UNBIASED variance), count the batch, bump the dropout seed once per
// step.  mean_offset: the Linear's bias, which the contraction left out because it cancels in the normalisation — it
// only shifts the running mean.
constexpr int kFinCols = 16, kFinGroups = kBlock / kFinCols;      // a workgroup: 16 columns x 16 tile groups
// RIDE: the launch carries extra workgroups (those past the ceil(N / 16) that join statistics) which write the step's
// dropout keep bits and zero-fill the backward pass's accumulation buffer — the work of k_tail_dropmask without a launch
// of its own (a kernel boundary inside a step costs ~2.5 us, the mask work itself under 1 us spread over the chip; this
// kernel is the first of the step whose successors need the bits).  The seed the bits are drawn from must not be bumped
// by this launch (seed_bump rides with a later one).
template <bool RIDE>
__global__ __launch_bounds__(kBlock) void k_bn_finalize_fwd(const float *__restrict__ part, int M, int N,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            const float *__restrict__ mean_offset, float *running_mean,
                                                            float *running_var, float momentum, float eps,
                                                            int64_t *nbt, int64_t *seed_bump, float *__restrict__ mu,
                                                            float *__restrict__ sc, float *__restrict__ be,
                                                            float *__restrict__ rstd_out, MaskRide ride) {
  if constexpr (RIDE) {
    const int nfin = (N + kFinCols - 1) / kFinCols;
    if ((int)blockIdx.x >= nfin) {
      mask_blocks(ride.j, ride.seed, ride.zero4, ride.nzero4, blockIdx.x - nfin, gridDim.x - nfin);
      return;
    }
  }
  // Latency is all this kernel is: 16 columns x 16 tile groups per workgroup (25 workgroups at N = 400), so that at
  // B = 4096 (64 tiles) a thread has 4 tiles and ONE round trip of loads (round 2: 64 columns x 4 groups, 16 tiles per
  // thread in two trips of 8).  A group joins its run of tiles in one pass around its first tile's mean (see
  // bn_merge_fwd), the 16 groups are merged in group order (Chan) by the group-0 thread.  Same tree every launch.
  __shared__ float sh[kFinGroups][kFinCols][3];
  const int cl = threadIdx.x % kFinCols, grp = threadIdx.x / kFinCols;
  const int n = blockIdx.x * kFinCols + cl;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (nbt) nbt[0] += 1;
    if (seed_bump) seed_bump[0] += 1;
  }
  const int MT = (M + BM - 1) / BM;
  const int per = (MT + kFinGroups - 1) / kFinGroups, t0 = grp * per, t1 = min(MT, t0 + per);
  // the per-column inputs of the last step travel with the first trip
  float gm = 1.f, bt = 0.f, mo = 0.f, rm = 0.f, rv = 0.f;
  if (grp == 0 && n < N) {
    if (gamma) gm = gamma[n];
    if (beta) bt = beta[n];
    if (running_mean) { rm = running_mean[n]; rv = running_var[n]; if (mean_offset) mo = mean_offset[n]; }
  }
  float n_g = 0.f, c = 0.f, sn = 0.f, sq = 0.f, sm = 0.f;
  if (n < N && t0 < t1) {
    for (int base = t0; base < t1; base += 8) {
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float2 *>(part + ((int64_t)min(base + u, t1 - 1) * N + n) * 2);
      if (base == t0) c = v[0].x;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = base + u;
        if (t < t1) {
          const float nb = (float)min(BM, M - t * BM), d = v[u].x - c;
          n_g += nb;
          sn = fmaf(nb, d, sn);
          sq = fmaf(nb * d, d, sq);
          sm += v[u].y;
        }
      }
    }
  }
  const float mean_g = n_g > 0.f ? c + sn / n_g : 0.f;
  const float m2_g = n_g > 0.f ? fmaxf(sm + (sq - sn * sn / n_g), 0.f) : 0.f;
  sh[grp][cl][0] = n_g; sh[grp][cl][1] = mean_g; sh[grp][cl][2] = m2_g;
  __syncthreads();
  if (grp != 0 || n >= N) return;
  float n_a = 0.f, mean_a = 0.f, m2_a = 0.f;
#pragma unroll
  for (int q = 0; q < kFinGroups; ++q) {
    const float n_b = sh[q][cl][0];
    if (n_b > 0.f) {
      const float tot = n_a + n_b, delta = sh[q][cl][1] - mean_a;
      mean_a += delta * (n_b / tot);
      m2_a += sh[q][cl][2] + delta * delta * (n_a * n_b / tot);
      n_a = tot;
    }
  }
  const float var = m2_a / (float)M;
  const float rstd = rsqrtf(var + eps);
  mu[n] = mean_a;
  sc[n] = gm * rstd;
  be[n] = bt;
  rstd_out[n] = rstd;
  if (running_mean) {
    running_mean[n] = (1.f - momentum) * rm + momentum * (mean_a + mo);
    const float unb = M > 1 ? m2_a / (float)(M - 1) : var;
    running_var[n] = (1.f - momentum) * rv + momentum * unb;
  }
}

// ============================================================================================== head (Linear(., 1)) ====
// out[m] = sum_n a(m, n) w[n] + b + add[m]: a wave per row group, float4 per lane over the features.
template <bool MERGE>
__global__ __launch_bounds__(kBlock) void k_tail_head_fwd(ActDesc x, const float *__restrict__ w, const float *__restrict__ b,
                                                          const float *__restrict__ add, float *__restrict__ out, int M,
                                                          int N, BnFwd bn, int64_t *bump) {
  __shared__ __attribute__((aligned(16))) float cst[MERGE ? 3 * kCstPitch : 4];
  // (a step whose layers have no statistics to join — no BatchNorm, or eval mode — advances the dropout seed here: every
  //  kernel that reads the step's keep bits was launched before this one)
  if (bump && blockIdx.x == 0 && threadIdx.x == 0) bump[0] += 1;
  const int lane = threadIdx.x & 63;
  const int wave0 = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), nw = gridDim.x * kWavesPerBlock;
  const float bv = b ? b[0] : 0.f;
  // `between`: work placed after the first trip's loads have been issued and before their values are used — the joined
  // form derives the constants there (they are not needed to ISSUE a row load, only to turn z into the activation), so
  // the join's own round trip runs under the rows' instead of in front of it (it cost +4 us in front)
  auto rows = [&](const auto &act, const auto &between) {
    if (N <= 512) {                              // up to 4 rows of a wave in flight (a workgroup per CU walks 16 rows)
      const int c0 = lane * 4, c1 = lane * 4 + 256;
      const bool v0 = c0 < N, v1 = c1 < N;
      const float4 w0 = v0 ? ld4(w + c0) : make_float4(0.f, 0.f, 0.f, 0.f), w1 = v1 ? ld4(w + c1) : make_float4(0.f, 0.f, 0.f, 0.f);
      bool first = true;
      for (int m0 = wave0; m0 < M || first; m0 += 4 * nw) {
        typename std::remove_reference<decltype(act)>::type::Raw r0[4], r1[4];
        if (m0 < M) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int m = min(m0 + u * nw, M - 1);
            if (v0) r0[u] = act.fetch(m, c0);
            if (v1) r1[u] = act.fetch(m, c1);
          }
        }
        if (first) between();
        first = false;
        if (m0 >= M) break;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int m = m0 + u * nw;
          if (m >= M) break;
          float s = 0.f;
          if (v0) { const float4 a = act.finish(r0[u], act.consts(c0), m, c0); s += a.x * w0.x + a.y * w0.y + a.z * w0.z + a.w * w0.w; }
          if (v1) { const float4 a = act.finish(r1[u], act.consts(c1), m, c1); s += a.x * w1.x + a.y * w1.y + a.z * w1.z + a.w * w1.w; }
          s = wave_sum(s);
          if (lane == 0) out[m] = s + bv + (add ? add[m] : 0.f);
        }
      }
      return;
    }
    between();
    for (int m = wave0; m < M; m += nw) {
      float s = 0.f;
      for (int c = lane * 4; c < N; c += 256) {
        const float4 a = act.finish(act.fetch(m, c), act.consts(c), m, c), ww = ld4(w + c);
        s += a.x * ww.x + a.y * ww.y + a.z * ww.z + a.w * ww.w;
      }
      s = wave_sum(s);
      if (lane == 0) out[m] = s + bv + (add ? add[m] : 0.f);
    }
  };
  const LoadAct act = make_act(x);
  if constexpr (MERGE) {
    const lds_cfp c = as_lds(cst);
    rows(LoadActL{x.Z, x.ld, c, c + kCstPitch, c + 2 * kCstPitch, act.drop}, [&]() {
      bn_merge_fwd<kBlock>(bn, M, N, cst, blockIdx.x == 0, (int)threadIdx.x);
      __syncthreads();
    });
  } else {
    rows(act, []() {});
  }
}

// Backward of the head into the last hidden layer: da(m, n) = g[m] w[n];
//   dy(m, n) = da * keepscale * [pre > 0]                       -> DY
//   part[blk][n] = (sum_m dy, sum_m dy * (z - mu))               (dbeta / dgamma pieces)
//   wpart[blk][n] = sum_m g[m] * a(m, n),  wpart[blk][N] = sum_m g[m]    (dw / db pieces)
// One workgroup = 16 rows x all features per trip; per-workgroup partials are joined by k_bn_finalize_bwd in block
// order (deterministic).
constexpr int kHeadRows = 8;
// reps > 0: part / wpart have `reps` rows (zeroed by the caller) and workgroup b ADDS its sums into row b % reps with float
// atomics — the next product's prologue then joins `reps` rows (bn_merge_bwd) and no finalize launch is needed.  Same-address
// float atomics serialise at ~26 ns apiece (measured: 64 adders per address cost a kernel ~1.7 us), so reps is chosen to
// leave <= 16 adders per address.
__global__ __launch_bounds__(kBlock) void k_tail_head_bwd(ActDesc x, const float *__restrict__ g, const float *__restrict__ w,
                                                          float *__restrict__ DY, float *__restrict__ part,
                                                          float *__restrict__ wpart, int M, int N, int reps) {
  // thread t owns columns 4t .. 4t+3 (N <= 1024)
  const Drop drop = make_drop(x.keep, x.p, x.ld);
  const int c = threadIdx.x * 4;
  const bool cv = c < N;
  float4 s_dy = zero4(), s_dyz = zero4(), s_ga = zero4();
  float s_g = 0.f;
  float4 u_ = zero4(), sc = zero4(), be = zero4(), ww = zero4();
  if (cv) { u_ = ld4(x.mu + c); sc = ld4(x.sc + c); be = ld4(x.be + c); ww = ld4(w + c); }
  const int rows_per = (M + gridDim.x - 1) / gridDim.x;
  const int mb = blockIdx.x * rows_per, me = min(M, mb + rows_per);
  for (int m4 = mb; m4 < me; m4 += kHeadRows) {   // kHeadRows rows in flight per thread: 16 rows per workgroup = 2 trips
    float4 z[kHeadRows];
    uint32_t kb[kHeadRows];
    float gv[kHeadRows];
#pragma unroll
    for (int u = 0; u < kHeadRows; ++u) {
      const int m = min(m4 + u, me - 1);
      gv[u] = g[m];
      if (cv) {
        z[u] = ld4(x.Z + (int64_t)m * x.ld + c);
        kb[u] = drop.fetch(m, c);
      }
    }
#pragma unroll
    for (int u = 0; u < kHeadRows; ++u) {
      const int m = m4 + u;
      if (m >= me) break;
      const float gm = gv[u];
      s_g += gm;
      if (!cv) continue;
      const float4 k = drop.scale4(kb[u], c);
      const float4 zc = make_float4(z[u].x - u_.x, z[u].y - u_.y, z[u].z - u_.z, z[u].w - u_.w);
      const float4 pre = make_float4(fmaf(zc.x, sc.x, be.x), fmaf(zc.y, sc.y, be.y), fmaf(zc.z, sc.z, be.z), fmaf(zc.w, sc.w, be.w));
      float4 dy;
      dy.x = pre.x > 0.f ? gm * ww.x * k.x : 0.f;
      dy.y = pre.y > 0.f ? gm * ww.y * k.y : 0.f;
      dy.z = pre.z > 0.f ? gm * ww.z * k.z : 0.f;
      dy.w = pre.w > 0.f ? gm * ww.w * k.w : 0.f;
      st4(DY + (int64_t)m * x.ld + c, dy);
      s_dy.x += dy.x; s_dy.y += dy.y; s_dy.z += dy.z; s_dy.w += dy.w;
      s_dyz.x += dy.x * zc.x; s_dyz.y += dy.y * zc.y; s_dyz.z += dy.z * zc.z; s_dyz.w += dy.w * zc.w;
      s_ga.x += gm * fmaxf(pre.x, 0.f) * k.x; s_ga.y += gm * fmaxf(pre.y, 0.f) * k.y;
      s_ga.z += gm * fmaxf(pre.z, 0.f) * k.z; s_ga.w += gm * fmaxf(pre.w, 0.f) * k.w;
    }
  }
  if (reps > 0) {
    // through LDS first: a thread's sums are 8 + 4 consecutive floats, so 64 lanes adding them directly touch sixteen 128-B
    // lines per wave-instruction (measured: the kernel took 18.9 us instead of 11.0); re-read linearly, every atomic
    // wave-instruction covers 64 consecutive floats — the shape the memory-side atomic units run fastest at
    __shared__ float sp[2 * 4 * kBlock + 4 * kBlock + 4];
    float *sw = sp + 2 * 4 * kBlock;
    if (cv) {
      st4(sp + c * 2, make_float4(s_dy.x, s_dyz.x, s_dy.y, s_dyz.y));
      st4(sp + c * 2 + 4, make_float4(s_dy.z, s_dyz.z, s_dy.w, s_dyz.w));
      st4(sw + c, s_ga);
    }
    if (threadIdx.x == 0) sw[N] = s_g;
    __syncthreads();
    const int rr = blockIdx.x % reps;
    float *o = part + (int64_t)rr * N * 2;
    for (int i = threadIdx.x; i < 2 * N; i += kBlock) atomicAdd(o + i, sp[i]);
    float *q = wpart + (int64_t)rr * (N + 4);
    for (int i = threadIdx.x; i <= N; i += kBlock) atomicAdd(q + i, sw[i]);
    return;
  }
  if (cv) {
    float *o = part + ((int64_t)blockIdx.x * N + c) * 2;
    st4(o, make_float4(s_dy.x, s_dyz.x, s_dy.y, s_dyz.y));
    st4(o + 4, make_float4(s_dy.z, s_dyz.z, s_dy.w, s_dyz.w));
    st4(wpart + (int64_t)blockIdx.x * (N + 4) + c, s_ga);
  }
  if (threadIdx.x == 0) wpart[(int64_t)blockIdx.x * (N + 4) + N] = s_g;
}

// Head + criterion + the head's backward in ONE launch (round 4): when the labels are known at forward time and the
// criterion is BCE-with-logits (mean) — the reference trainer's (src/trainer/deepfm.py:32,51) — a row's logit, its loss
// term, g[m] = (sigmoid(logit) - y) / M and its contributions to every column sum of k_tail_head_bwd depend on that row
// alone (given the layer's constants), so one pass over Z does what k_tail_head_fwd, k_bce_logits_fwd and
// k_tail_head_bwd do in three launches with two passes.  Wave per row (up to 4 rows in flight), lane owns columns
// 4 lane .. + 3 and 256 + 4 lane .. + 3 (N <= 512); column sums are kept per lane across the wave's rows, met in LDS,
// and added into row (block % reps) of part / wpart with linear float atomics exactly like k_tail_head_bwd's summed form.
//   loss_ws: [0] the loss (mean), [4 .. 4 + reps) per-replica partial sums, [4 + reps .. 4 + 2 reps) arrival counts
//   (uint32) — all zero at launch.  The workgroups of a replica add their partial, the last of them (ticket) moves the
//   replica's sum into [0]: <= gridDim / reps adders per address at each level, no single word that every workgroup hits.
// The sums are those of the upstream gradient upstream[0] (a device scalar the caller will seed the backward with; NULL: 1 —
// the criterion is the last op of a step); tail.py falls back to k_tail_head_bwd when the backward arrives with anything else.
constexpr int kHeadLossCols = 512;
constexpr int kHeadLossRed = 3 * kHeadLossCols + 4;
// (MERGE: 0 constants read from memory, 1 joined from tile statistics, 2 derived from shifted sums — apart so that the
//  sums form does not carry the tile join's 32-deep load chunks in its register budget: 359 VGPRs + scratch with both)
template <int MERGE>
__global__ __launch_bounds__(kBlock) void k_tail_head_bce(ActDesc x, const float *__restrict__ w, const float *__restrict__ b,
                                                          const float *__restrict__ add, const float *__restrict__ y,
                                                          float *__restrict__ out, float *__restrict__ gout,
                                                          float *__restrict__ DY, float *__restrict__ part,
                                                          float *__restrict__ wpart, float *__restrict__ loss_ws, int M,
                                                          int N, BnFwd bn, int reps, int64_t *bump,
                                                          const float *__restrict__ upstream) {
  __shared__ __attribute__((aligned(16))) float cst[MERGE != 0 ? 3 * kCstPitch : 4];
  __shared__ __attribute__((aligned(16))) float red[kWavesPerBlock][kHeadLossRed];
  if (bump && blockIdx.x == 0 && threadIdx.x == 0) bump[0] += 1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave0 = blockIdx.x * kWavesPerBlock + wv, nw = gridDim.x * kWavesPerBlock;
  const Drop drop = make_drop(x.keep, x.p, x.ld);
  const float bv = b ? b[0] : 0.f;
  const float inv_m = 1.f / (float)M;
  const float gsc = upstream ? upstream[0] * inv_m : inv_m;      // d objective / d loss (a device scalar), over the mean's M
  const int c0 = lane * 4, c1 = lane * 4 + 256;
  const bool v0 = c0 < N, v1 = c1 < N;
  const float4 w0 = v0 ? ld4(w + c0) : zero4(), w1 = v1 ? ld4(w + c1) : zero4();
  float4 mu0 = zero4(), sc0 = zero4(), be0 = zero4(), mu1 = zero4(), sc1 = zero4(), be1 = zero4();
  float4 dy0 = zero4(), dz0 = zero4(), ga0 = zero4(), dy1 = zero4(), dz1 = zero4(), ga1 = zero4();
  float s_g = 0.f, s_loss = 0.f;
  bool first = true;
  for (int m0 = wave0; m0 < M || first; m0 += 4 * nw) {
    float4 z0[4], z1[4];
    uint32_t k0[4], k1[4];
    float av[4], yv[4];
    if (m0 < M) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int m = min(m0 + u * nw, M - 1);
        if (v0) { z0[u] = ld4(x.Z + (int64_t)m * x.ld + c0); k0[u] = drop.fetch(m, c0); }
        if (v1) { z1[u] = ld4(x.Z + (int64_t)m * x.ld + c1); k1[u] = drop.fetch(m, c1); }
        av[u] = add ? add[m] : 0.f;
        yv[u] = y[m];
      }
    }
    if (first) {      // the constants are not needed to ISSUE the row loads: their join runs under the rows' round trip
      if constexpr (MERGE != 0) {
        if constexpr (MERGE == 2) bn_derive_fwd<kBlock>(bn, M, N, cst, blockIdx.x == 0, (int)threadIdx.x);
        else bn_merge_fwd<kBlock>(bn, M, N, cst, blockIdx.x == 0, (int)threadIdx.x);
        __syncthreads();
        const lds_cfp c = as_lds(cst);
        if (v0) { mu0 = vld4(c + c0); sc0 = vld4(c + kCstPitch + c0); be0 = vld4(c + 2 * kCstPitch + c0); }
        if (v1) { mu1 = vld4(c + c1); sc1 = vld4(c + kCstPitch + c1); be1 = vld4(c + 2 * kCstPitch + c1); }
      } else {
        if (v0) { mu0 = ld4(x.mu + c0); sc0 = ld4(x.sc + c0); be0 = ld4(x.be + c0); }
        if (v1) { mu1 = ld4(x.mu + c1); sc1 = ld4(x.sc + c1); be1 = ld4(x.be + c1); }
      }
    }
    first = false;
    if (m0 >= M) break;
    // the four rows' dots first, then ONE butterfly over the four sums side by side (a row at a time is four dependent chains
    // of six lane exchanges, ~1 us of a kernel that is all latency), then the rows' remaining arithmetic
    float sa[4];
    auto row0 = [&](int u, float4 &zc, float4 &pre, float4 &kk) {
      kk = drop.scale4(k0[u], c0);
      zc = make_float4(z0[u].x - mu0.x, z0[u].y - mu0.y, z0[u].z - mu0.z, z0[u].w - mu0.w);
      pre = make_float4(fmaf(zc.x, sc0.x, be0.x), fmaf(zc.y, sc0.y, be0.y), fmaf(zc.z, sc0.z, be0.z), fmaf(zc.w, sc0.w, be0.w));
    };
    auto row1 = [&](int u, float4 &zc, float4 &pre, float4 &kk) {
      kk = drop.scale4(k1[u], c1);
      zc = make_float4(z1[u].x - mu1.x, z1[u].y - mu1.y, z1[u].z - mu1.z, z1[u].w - mu1.w);
      pre = make_float4(fmaf(zc.x, sc1.x, be1.x), fmaf(zc.y, sc1.y, be1.y), fmaf(zc.z, sc1.z, be1.z), fmaf(zc.w, sc1.w, be1.w));
    };
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float4 zc, pre, kk;
      float s = 0.f;
      if (v0) {
        row0(u, zc, pre, kk);
        s += fmaxf(pre.x, 0.f) * kk.x * w0.x + fmaxf(pre.y, 0.f) * kk.y * w0.y + fmaxf(pre.z, 0.f) * kk.z * w0.z +
             fmaxf(pre.w, 0.f) * kk.w * w0.w;
      }
      if (v1) {
        row1(u, zc, pre, kk);
        s += fmaxf(pre.x, 0.f) * kk.x * w1.x + fmaxf(pre.y, 0.f) * kk.y * w1.y + fmaxf(pre.z, 0.f) * kk.z * w1.z +
             fmaxf(pre.w, 0.f) * kk.w * w1.w;
      }
      sa[u] = s;
    }
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) sa[u] += __shfl_xor(sa[u], msk);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = m0 + u * nw;
      if (m >= M) break;
      // (the row's pre-activations again from its loaded values: a few instructions, against 24 registers per row kept alive)
      float4 zc0 = zero4(), zc1 = zero4(), pre0 = zero4(), pre1 = zero4(), kk0 = zero4(), kk1 = zero4();
      if (v0) row0(u, zc0, pre0, kk0);
      if (v1) row1(u, zc1, pre1, kk1);
      const float s = sa[u];
      const float xl = s + bv + av[u];
      // the criterion's own arithmetic (k_bce_logits_fwd / _bwd, cross.hip)
      const float gm = gsc * (1.f / (1.f + expf(-xl)) - yv[u]);
      s_loss += fmaxf(xl, 0.f) - xl * yv[u] + log1pf(expf(-fabsf(xl)));
      s_g += gm;
      if (lane == 0) { out[m] = xl; gout[m] = gm; }
      if (v0) {
        float4 d;
        d.x = pre0.x > 0.f ? gm * w0.x * kk0.x : 0.f;
        d.y = pre0.y > 0.f ? gm * w0.y * kk0.y : 0.f;
        d.z = pre0.z > 0.f ? gm * w0.z * kk0.z : 0.f;
        d.w = pre0.w > 0.f ? gm * w0.w * kk0.w : 0.f;
        st4(DY + (int64_t)m * x.ld + c0, d);
        dy0.x += d.x; dy0.y += d.y; dy0.z += d.z; dy0.w += d.w;
        dz0.x += d.x * zc0.x; dz0.y += d.y * zc0.y; dz0.z += d.z * zc0.z; dz0.w += d.w * zc0.w;
        ga0.x += gm * fmaxf(pre0.x, 0.f) * kk0.x; ga0.y += gm * fmaxf(pre0.y, 0.f) * kk0.y;
        ga0.z += gm * fmaxf(pre0.z, 0.f) * kk0.z; ga0.w += gm * fmaxf(pre0.w, 0.f) * kk0.w;
      }
      if (v1) {
        float4 d;
        d.x = pre1.x > 0.f ? gm * w1.x * kk1.x : 0.f;
        d.y = pre1.y > 0.f ? gm * w1.y * kk1.y : 0.f;
        d.z = pre1.z > 0.f ? gm * w1.z * kk1.z : 0.f;
        d.w = pre1.w > 0.f ? gm * w1.w * kk1.w : 0.f;
        st4(DY + (int64_t)m * x.ld + c1, d);
        dy1.x += d.x; dy1.y += d.y; dy1.z += d.z; dy1.w += d.w;
        dz1.x += d.x * zc1.x; dz1.y += d.y * zc1.y; dz1.z += d.z * zc1.z; dz1.w += d.w * zc1.w;
        ga1.x += gm * fmaxf(pre1.x, 0.f) * kk1.x; ga1.y += gm * fmaxf(pre1.y, 0.f) * kk1.y;
        ga1.z += gm * fmaxf(pre1.z, 0.f) * kk1.z; ga1.w += gm * fmaxf(pre1.w, 0.f) * kk1.w;
      }
    }
  }
  // the four waves' sums meet in LDS: [0, 2 * 512) (sum dy, sum dy (z - mu)) interleaved per column, [1024, 1536) the
  // head's dw pieces, [1536] sum g, [1537] the loss terms
  float *rp = red[wv];
  if (v0) {
    st4(rp + 2 * c0, make_float4(dy0.x, dz0.x, dy0.y, dz0.y));
    st4(rp + 2 * c0 + 4, make_float4(dy0.z, dz0.z, dy0.w, dz0.w));
    st4(rp + 2 * kHeadLossCols + c0, ga0);
  }
  if (v1) {
    st4(rp + 2 * c1, make_float4(dy1.x, dz1.x, dy1.y, dz1.y));
    st4(rp + 2 * c1 + 4, make_float4(dy1.z, dz1.z, dy1.w, dz1.w));
    st4(rp + 2 * kHeadLossCols + c1, ga1);
  }
  if (lane == 0) { rp[3 * kHeadLossCols] = s_g; rp[3 * kHeadLossCols + 1] = s_loss; }
  __syncthreads();
  const int rr = blockIdx.x % reps;
  // the last wave takes the loss (a chain of dependent round trips: add, fence, ticket), the other three the column sums
  // (fire and forget): the two run side by side instead of one after the other
  constexpr int kSumThreads = kBlock - kWave;
  if (wv < kWavesPerBlock - 1) {
    float *o = part + (int64_t)rr * N * 2;
    for (int i = threadIdx.x; i < 2 * N; i += kSumThreads) atomicAdd(o + i, (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]));
    float *q = wpart + (int64_t)rr * (N + 4);
    for (int i = threadIdx.x; i <= N; i += kSumThreads) {
      const int j = i < N ? 2 * kHeadLossCols + i : 3 * kHeadLossCols;
      atomicAdd(q + i, (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]));
    }
  } else if (lane == 0) {
    const int j = 3 * kHeadLossCols + 1;
    const float lp = ((red[0][j] + red[1][j]) + (red[2][j] + red[3][j])) * inv_m;
    atomicAdd(loss_ws + 4 + rr, lp);
    __threadfence();
    const uint32_t mine = ((uint32_t)gridDim.x - (uint32_t)rr + (uint32_t)reps - 1u) / (uint32_t)reps;   // workgroups of this replica
    const uint32_t old = atomicAdd(reinterpret_cast<uint32_t *>(loss_ws + 4 + reps) + rr, 1u);
    if (old == mine - 1u) {
      __threadfence();
      const float v = __hip_atomic_load(loss_ws + 4 + rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicAdd(loss_ws, v);
    }
  }
}

// Join the backward column sums (block order), write dgamma / dbeta and the three constants of LoadDz.
//   dbeta = sum dy, dgamma = rstd * sum dy (z - mu)
//   dz = gamma rstd (dy - dbeta / M - zhat dgamma / M) = al dy + bz (z - mu) + de
// wpart (nullable): the head's dw / db pieces [nblk][N + 4] -> dw[N], db[1].
__global__ __launch_bounds__(kBlock) void k_bn_finalize_bwd(const float *__restrict__ part, int nblk, int M, int N,
                                                            const float *__restrict__ gamma, const float *__restrict__ rstd,
                                                            float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                            float *__restrict__ al, float *__restrict__ bz,
                                                            float *__restrict__ de, const float *__restrict__ wpart,
                                                            int nwblk, float *__restrict__ dw, float *__restrict__ db,
                                                            int affine, float *__restrict__ dbias, float *__restrict__ db2) {
  // 16 columns x 16 row groups per workgroup (see k_bn_finalize_fwd): a group adds its run of partial rows in row order
  // with up to 16 loads in flight (64 partial rows: 4 per thread, one round trip; the head's 256: 16 per thread, one
  // trip), the 16 groups are added in group order through LDS.  Column N of the head's pieces is db.
  __shared__ float sh[kFinGroups][kFinCols][3];
  const int cl = threadIdx.x % kFinCols, grp = threadIdx.x / kFinCols;
  const int n = blockIdx.x * kFinCols + cl;
  float r = 0.f, gm = 1.f;
  if (grp == 0 && n < N) { r = rstd[n]; if (gamma) gm = gamma[n]; }
  float s1 = 0.f, s2 = 0.f, sw = 0.f;
  if (n < N) {
    const int per = (nblk + kFinGroups - 1) / kFinGroups, t0 = grp * per, t1 = min(nblk, t0 + per);
    for (int base = t0; base < t1; base += 16) {
      float2 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float2 *>(part + ((int64_t)min(base + u, t1 - 1) * N + n) * 2);
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (base + u < t1) { s1 += v[u].x; s2 += v[u].y; }
    }
  }
  if (wpart && n <= N) {
    const int per = (nwblk + kFinGroups - 1) / kFinGroups, t0 = grp * per, t1 = min(nwblk, t0 + per);
    for (int base = t0; base < t1; base += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = wpart[(int64_t)min(base + u, t1 - 1) * (N + 4) + n];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (base + u < t1) sw += v[u];
    }
  }
  sh[grp][cl][0] = s1; sh[grp][cl][1] = s2; sh[grp][cl][2] = sw;
  __syncthreads();
  if (grp != 0 || n > N) return;
  s1 = 0.f; s2 = 0.f; sw = 0.f;
#pragma unroll
  for (int q = 0; q < kFinGroups; ++q) { s1 += sh[q][cl][0]; s2 += sh[q][cl][1]; sw += sh[q][cl][2]; }
  if (wpart) {
    if (n < N) dw[n] = sw;
    else {
      if (db) db[0] = sw;
      if (db2) db2[0] = sw;
    }
  }
  if (n >= N) return;
  const float dg = s2 * r;
  if (dgamma) dgamma[n] = dg;
  if (dbeta) dbeta[n] = s1;
  const float inv_m = 1.f / (float)M;
  al[n] = gm * r;
  bz[n] = affine ? 0.f : -gm * r * r * dg * inv_m;
  de[n] = affine ? 0.f : -gm * r * s1 * inv_m;
  if (affine && dbias) dbias[n] = gm * r * s1;
}

// Constants of layers that normalise with FIXED statistics or not at all (eval mode, use_batchnorm=False), all layers of a
// tail in one launch:  a = relu((z - mu) * sc + be) with z = x W^T (the Linear bias b stays out of the product)
//   eval-mode BatchNorm (src/models/deepfm.py:57-58 under model.eval()): rstd = 1/sqrt(running_var + eps),
//       mu = running_mean - b,  sc = gamma rstd,  be = beta
//   no BatchNorm (DeepFM's default use_batchnorm=False):  mu = 0, sc = 1, be = b, rstd = 1
__global__ __launch_bounds__(kBlock) void k_tail_affine_consts(AffineJob j) {
  affine_consts_blocks(j, (int)blockIdx.x, (int)gridDim.x);
}

// ============================================================================================== dgrad GEMM ====
// da_prev[m][k] = sum_n dz(m, n) W[n][k]; epilogue: dy_prev = da_prev * keepscale_prev * [pre_prev > 0] (+ column sums)
// or the plain da_prev when the previous "layer" is the input (prev.mu == null).
struct DgradArgs {
  DzDesc dz;          // R operand [M, N], reduction over N
  const float *W;     // [N, K]: OC operand (reduction rows n, outputs k)
  int ldw;
  ActDesc prev;       // the layer below (its Z / constants / dropout) — mu == null: none, store da as is
  float *OUT;         // [M, K] dy_prev or da_prev
  int ldo;
  float *part;        // [MT, K, 2] (sum dy, sum dy (z - mu)), nullable
  int part_rep;       // > 0: part is [part_rep, K, 2], zeroed by the caller; row tile mt ADDS into row mt % part_rep (atomics)
  float *dz_out;      // [M, N] (pitch dz.ld), nullable: dz as the operand load computed it
  int M, N, K;
  int ncols, ntn;     // column tiles over K
  BnBwd bn;           // MERGE: the column sums behind dz's constants, joined here (dz.al / bz / de are then not read)
  // FM: the tail's input is DeepFM's embedding block emb[M, F*D] and OUT is the row-form gradient of the lookup table
  // (src/models/deepfm.py:88-98 through autograd): OUT[m, f*D + d] = da[m, f*D + d] + gy[m] * (S[m, d] - emb[m, f*D + d]),
  // g1[m, f] = gy[m] — the whole of mi_gather_fm_bwd_rows in this epilogue, da never stored
  const float *fm_emb;   // [M, K] saved rows (pitch K)
  const float *fm_sum;   // [M, fm_D] sum over the fields (mi_gather_fm_fwd_sum)
  const float *fm_gy;    // [M] dL/d y_fm
  float *fm_g1;          // [M, K / fm_D] first-order gradient values (nullable)
  int fm_D;
  // ... and for the table-sharded lookup (sharded.py): the rows came out of a receive buffer of packed rows
  // {fm_D embedding floats, first-order weight, pad} at row slot[m, f]; the gradient goes back the same way — OUT is that
  // buffer's gradient (pitch fm_pitch), row slot[m, f] gets the embedding part at [0, fm_D) and gy[m] at [fm_D]
  // (mi_slot_fm_bwd in this epilogue; slots no sample points at are left as the caller zeroed them)
  const int64_t *fm_slot;   // [M, K / fm_D], nullable
  int fm_pitch;
};

template <bool DZ, bool MID, bool MERGE, bool FM = false>
__global__ __launch_bounds__(kThreads) void k_tail_dgrad(DgradArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats + (MERGE ? kCstFloats : 0)];
  const int mt_total = (a.M + BM - 1) / BM;
  const int tile = xcd_logical(blockIdx.x, mt_total * a.ntn);
  if (tile < 0) return;

  const int mt = tile / a.ntn, nt = tile % a.ntn;
  const int m0 = mt * BM, k0 = nt * a.ncols;
  const int rows_valid = min(BM, a.M - m0), cols_valid = min(a.ncols, a.K - k0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  floatx4 acc[NSUB];
#pragma unroll
  for (int s = 0; s < NSUB; ++s) acc[s] = floatx4{0.f, 0.f, 0.f, 0.f};

  const LoadDz dzl = make_dz(a.dz);
  const LoadPlain dyp{a.dz.DY, a.dz.ld};
  const LoadPlain wp{a.W, a.ldw};     // [red = n][out = k]
  const OtOperand<BNT, LoadPlain> opC{wp, k0, cols_valid, a.N};     // transposed into a KC tile on the way in
  if constexpr (MERGE) {
    const lds_cfp c = as_lds(lds + kLdsFloats);
    const LoadDzL dzm{a.dz.DY, a.dz.Z, a.dz.ld, c, c + kCstPitch, c + 2 * kCstPitch, c + 3 * kCstPitch};
    float *cst = lds + kLdsFloats;
    const bool writer = tile == 0, wwriter = tile == min(1, mt_total * a.ntn - 1);
    main_loop<true, true>(acc, lds, 0, a.N, KcOperand<64, Tee<LoadDzL>>{Tee<LoadDzL>{dzm, nt == 0 ? a.dz_out : nullptr, a.dz.ld}, m0, rows_valid, a.N}, opC,
                          make_pre([&]() { bn_merge_bwd<kThreads - kProd>(a.bn, a.M, a.N, cst, writer, wwriter, (int)threadIdx.x); }));
  } else if constexpr (DZ)
    main_loop<true, true>(acc, lds, 0, a.N, KcOperand<64, Tee<LoadDz>>{Tee<LoadDz>{dzl, nt == 0 ? a.dz_out : nullptr, a.dz.ld}, m0, rows_valid, a.N}, opC);
  else main_loop<true, true>(acc, lds, 0, a.N, KcOperand<64, LoadPlain>{dyp, m0, rows_valid, a.N}, opC);

  // ---- epilogue through LDS (see k_tail_fwd): all 8 waves, whole row segments
  float *T = lds;
  acc_to_lds(acc, T, wave, lane);
  __syncthreads();
  const int t = threadIdx.x;
  if constexpr (FM) {
    static_assert(!MID, "the FM epilogue belongs to the product whose output is the tail's input gradient");
    // all loads of the thread's four chunks first (saved rows from HBM / Infinity Cache, sums and gy from L2), then the
    // arithmetic: one round trip for the epilogue
    float4 e[4], sm[4], da[4];
    float gy[4];
    int64_t sl[4] = {0, 0, 0, 0};
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = t + k * kThreads, row = i / 28, c = (i % 28) * 4;
      ok[k] = i < 64 * 28 && row < rows_valid && c < cols_valid;
      const int m = m0 + (ok[k] ? row : 0), kc = k0 + (ok[k] ? c : 0);
      e[k] = ld4(a.fm_emb + (int64_t)m * a.K + kc);
      sm[k] = ld4(a.fm_sum + (int64_t)m * a.fm_D + kc % a.fm_D);
      gy[k] = a.fm_gy[m];
      da[k] = ld4(T + (ok[k] ? row : 0) * kTilePitch + (ok[k] ? c : 0));
      if (a.fm_slot) sl[k] = a.fm_slot[(int64_t)m * (a.K / a.fm_D) + kc / a.fm_D];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = t + k * kThreads, row = i / 28, c = (i % 28) * 4;
      if (ok[k]) {
        float4 o;
        o.x = da[k].x + gy[k] * (sm[k].x - e[k].x);
        o.y = da[k].y + gy[k] * (sm[k].y - e[k].y);
        o.z = da[k].z + gy[k] * (sm[k].z - e[k].z);
        o.w = da[k].w + gy[k] * (sm[k].w - e[k].w);
        if (a.fm_slot) {
          float *dst = a.OUT + sl[k] * a.fm_pitch;
          const int d = (k0 + c) % a.fm_D;
          st4(dst + d, o);
          if (d == 0) dst[a.fm_D] = gy[k];          // (the chunk that opens a field also writes its first-order value)
        } else {
          st4(a.OUT + (int64_t)(m0 + row) * a.ldo + k0 + c, o);
        }
      }
    }
    if (a.fm_g1 && nt == 0) {      // the first-order table's gradient values of this row tile: gy[m] for every field
      const int F = a.K / a.fm_D;
      for (int i = t; i < rows_valid * F; i += kThreads) a.fm_g1[(int64_t)m0 * F + i] = a.fm_gy[m0 + i / F];
    }
  } else if constexpr (!MID) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = t + k * kThreads, row = i / 28, c = (i % 28) * 4;
      if (i < 64 * 28 && row < rows_valid && c < cols_valid)
        st4(a.OUT + (int64_t)(m0 + row) * a.ldo + k0 + c, ld4(T + row * kTilePitch + c));
    }
  } else {
    // thread (cg = t % 28, rl = t / 28 < 18) owns columns 4 cg .. 4 cg + 3 (their constants are loaded once) and rows
    // rl, rl + 18, rl + 36, rl + 54: dy = da * keep/(1-p) * [pre > 0], its column sums ride along in registers
    const Drop drop = make_drop(a.prev.keep, a.prev.p, a.prev.ld);
    const int cg = t % 28, rl = t / 28, c = cg * 4, kc = k0 + c;
    const bool cv = rl < 18 && c < cols_valid;
    float4 s_dy = zero4(), s_dyz = zero4();
    if (cv) {
      const float4 u = ld4(a.prev.mu + kc), sc = ld4(a.prev.sc + kc), be = ld4(a.prev.be + kc);
      float4 z[4], da[4];
      uint32_t kb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = rl + 18 * k;
        const int rc = row < rows_valid ? row : rows_valid - 1;
        z[k] = ld4(a.prev.Z + (int64_t)(m0 + rc) * a.prev.ld + kc);
        kb[k] = drop.fetch(m0 + rc, kc);
        da[k] = ld4(T + (row < 64 ? row : 63) * kTilePitch + c);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = rl + 18 * k;
        if (row < rows_valid) {
          const float4 kk = drop.scale4(kb[k], kc);
          const float4 zc = make_float4(z[k].x - u.x, z[k].y - u.y, z[k].z - u.z, z[k].w - u.w);
          float4 dy;
          dy.x = fmaf(zc.x, sc.x, be.x) > 0.f ? da[k].x * kk.x : 0.f;
          dy.y = fmaf(zc.y, sc.y, be.y) > 0.f ? da[k].y * kk.y : 0.f;
          dy.z = fmaf(zc.z, sc.z, be.z) > 0.f ? da[k].z * kk.z : 0.f;
          dy.w = fmaf(zc.w, sc.w, be.w) > 0.f ? da[k].w * kk.w : 0.f;
          st4(a.OUT + (int64_t)(m0 + row) * a.ldo + kc, dy);
          s_dy.x += dy.x; s_dy.y += dy.y; s_dy.z += dy.z; s_dy.w += dy.w;
          s_dyz.x += dy.x * zc.x; s_dyz.y += dy.y * zc.y; s_dyz.z += dy.z * zc.z; s_dyz.w += dy.w * zc.w;
        }
      }
    }
    if (!a.part) return;
    float *ws = lds + 64 * kTilePitch;                   // [18][BNT][2]
    if (rl < 18 && c < BNT) {
      float *o = ws + (rl * BNT + c) * 2;
      st4(o, make_float4(s_dy.x, s_dyz.x, s_dy.y, s_dyz.y));
      st4(o + 4, make_float4(s_dy.z, s_dyz.z, s_dy.w, s_dyz.w));
    }
    __syncthreads();
    if (t < cols_valid) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int q = 0; q < 18; ++q) { s1 += ws[(q * BNT + t) * 2]; s2 += ws[(q * BNT + t) * 2 + 1]; }
      if (a.part_rep > 0) {
        float *o = a.part + ((int64_t)(mt % a.part_rep) * a.K + k0 + t) * 2;
        atomicAdd(o, s1);
        atomicAdd(o + 1, s2);
      } else {
        float *o = a.part + ((int64_t)mt * a.K + k0 + t) * 2;
        o[0] = s1; o[1] = s2;
      }
    }
  }
}

// ============================================================================================== wgrad GEMM ====
// dW[n][k] = sum_m dz(m, n) a_prev(m, k): both operands OC (reduction rows m).  Output tiles 64 (n) x ncols (k); the batch
// is cut into `splits` slices, slice s writes slab[s][N][K]; k_slab_sum adds the slabs in slice order.
struct WgradArgs {
  DzDesc dz;          // [M, N]
  ActDesc prev;       // [M, K] activation below (mu == null: plain matrix prev.Z)
  float *slab;        // [splits, N, K]
  int M, N, K;
  int ncols, ntk;     // column tiles over K
  int ntn;            // row tiles over N (64 each)
  int splits;         // slice s covers the 32-row chunks [s * chunks / splits, (s + 1) * chunks / splits)
};

template <bool DZ, bool ACT>
__global__ __launch_bounds__(kThreads) void k_tail_wgrad(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
  const int per_split = a.ntn * a.ntk;
  const int tile = xcd_logical(blockIdx.x, per_split * a.splits);
  if (tile < 0) return;
  const int sp = tile / per_split, rem = tile % per_split;
  const int tn = rem / a.ntk, tk = rem % a.ntk;
  const int n0 = tn * 64, k0 = tk * a.ncols;
  const int nrows_valid = min(64, a.N - n0), cols_valid = min(a.ncols, a.K - k0);
  const int chunks = (a.M + BK - 1) / BK;
  const int mb = (int)((int64_t)sp * chunks / a.splits) * BK, me = min(a.M, (int)((int64_t)(sp + 1) * chunks / a.splits) * BK);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  floatx4 acc[NSUB];
#pragma unroll
  for (int s = 0; s < NSUB; ++s) acc[s] = floatx4{0.f, 0.f, 0.f, 0.f};

  const LoadDz dzl = make_dz(a.dz);
  const LoadPlain dyp{a.dz.DY, a.dz.ld};
  const LoadAct act = make_act(a.prev);
  const LoadPlain xp{a.prev.Z, a.prev.ld};
  // (general addressing: two transformed, transposed operands with three register stages each leave no room for the
  // lean form's per-thread offsets — it spilled; this kernel only runs in deterministic mode)
  auto run = [&](const auto &opR) {       // both operands have the batch as the slow index: transposed into KC tiles
    if constexpr (ACT) main_loop<true, true>(acc, lds, mb, me, opR, OtOperandG<BNT, LoadAct>{act, k0, cols_valid, me});
    else main_loop<true, true>(acc, lds, mb, me, opR, OtOperandG<BNT, LoadPlain>{xp, k0, cols_valid, me});
  };
  if constexpr (DZ) run(OtOperandG<64, LoadDz>{dzl, n0, nrows_valid, me});
  else run(OtOperandG<64, LoadPlain>{dyp, n0, nrows_valid, me});

  const int r = lane & 15, g = lane >> 4;
  const int n = wave < 4 ? n0 + wave * 16 + r : a.N;      // waves 4-7 were the producers
  float *S = a.slab + (int64_t)sp * a.N * a.K;
#pragma unroll
  for (int s = 0; s < NSUB; ++s) {
    const int c = 16 * s + 4 * g;
    if (n < a.N && c < cols_valid) st4(S + (int64_t)n * a.K + k0 + c, make_float4(acc[s][0], acc[s][1], acc[s][2], acc[s][3]));
  }
}

__global__ __launch_bounds__(kBlock) void k_slab_sum(const float *__restrict__ slab, int splits, int64_t n4, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n4) return;
  float4 s = ld4(slab + i * 4);
  for (int k = 1; k < splits; ++k) {
    const float4 v = ld4(slab + ((int64_t)k * n4 + i) * 4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  st4(out + i * 4, s);
}

int cols_per_tile(int n, int *ntiles) {
  // the fewest column tiles of <= 112 columns (multiples of 4) that cover n: 400 -> 4 x 100, 416 -> 4 x 104
  int nt = (n + BNT - 1) / BNT;
  int nc = ((n + nt - 1) / nt + 3) / 4 * 4;
  *ntiles = (n + nc - 1) / nc;
  return nc;
}
inline int grid8(int tiles) { return (tiles + 7) / 8 * 8; }
bool vec_ok(const void *p, int ld) { return aligned16(p) && ld % 4 == 0; }

}  // namespace

extern "C" {

static bool bn_fwd_view(const mi_tail_bn_fwd *m, BnFwd *out) {
  if (!m->part || !aligned16(m->part) || !m->mu || !m->sc || !m->be || !m->rstd) return false;
  if ((m->running_mean == nullptr) != (m->running_var == nullptr)) return false;
  if (m->nrep < 0 || m->nrep > 64 || (m->nrep > 0 && !m->shift)) return false;
  *out = BnFwd{m->part, m->gamma, m->beta, m->mean_offset, m->running_mean, m->running_var, m->num_batches_tracked,
               m->seed_bump, m->mu, m->sc, m->be, m->rstd, m->momentum, m->eps, m->nrep, m->shift};
  return true;
}

int mi_tail_fwd_gemm_m(const float *X, int32_t ldx, const float *x_mu, const float *x_sc, const float *x_be, float x_p,
                       const uint8_t *x_keep, const float *W, int32_t ldw, float *Z, int32_t ldz, float *part, float *a_out,
                       int32_t M, int32_t N, int32_t K, const mi_tail_bn_fwd *x_stats, void *stream) {
  return mi_tail_fwd_gemm_s(X, ldx, x_mu, x_sc, x_be, x_p, x_keep, W, ldw, Z, ldz, part, a_out, M, N, K, x_stats, 0, nullptr,
                            nullptr, nullptr, stream);
}

static int fwd_launch(const float *X, int32_t ldx, const float *x_mu, const float *x_sc, const float *x_be, float x_p,
                      const uint8_t *x_keep, const float *W, int32_t ldw, float *Z, int32_t ldz, float *part, float *a_out,
                      int32_t M, int32_t N, int32_t K, const mi_tail_bn_fwd *x_stats, int32_t sum_reps, float *shift_out,
                      const float *shift_running_mean, const float *shift_mean_offset, const mi_tail_head_in_epilogue *head,
                      void *stream);

int mi_tail_fwd_gemm_s(const float *X, int32_t ldx, const float *x_mu, const float *x_sc, const float *x_be, float x_p,
                       const uint8_t *x_keep, const float *W, int32_t ldw, float *Z, int32_t ldz, float *part, float *a_out,
                       int32_t M, int32_t N, int32_t K, const mi_tail_bn_fwd *x_stats, int32_t sum_reps, float *shift_out,
                       const float *shift_running_mean, const float *shift_mean_offset, void *stream) {
  return fwd_launch(X, ldx, x_mu, x_sc, x_be, x_p, x_keep, W, ldw, Z, ldz, part, a_out, M, N, K, x_stats, sum_reps, shift_out,
                    shift_running_mean, shift_mean_offset, nullptr, stream);
}

int mi_tail_fwd_gemm_head(const float *X, int32_t ldx, const float *x_mu, const float *x_sc, const float *x_be, float x_p,
                          const uint8_t *x_keep, const float *W, int32_t ldw, int32_t M, int32_t N, int32_t K,
                          const mi_tail_bn_fwd *x_stats, const mi_tail_head_in_epilogue *head, void *stream) {
  if (!head) return MI_ERR_INVALID_ARG;
  return fwd_launch(X, ldx, x_mu, x_sc, x_be, x_p, x_keep, W, ldw, nullptr, N, nullptr, nullptr, M, N, K, x_stats, 0, nullptr,
                    nullptr, nullptr, head, stream);
}

static int fwd_launch(const float *X, int32_t ldx, const float *x_mu, const float *x_sc, const float *x_be, float x_p,
                      const uint8_t *x_keep, const float *W, int32_t ldw, float *Z, int32_t ldz, float *part, float *a_out,
                      int32_t M, int32_t N, int32_t K, const mi_tail_bn_fwd *x_stats, int32_t sum_reps, float *shift_out,
                      const float *shift_running_mean, const float *shift_mean_offset, const mi_tail_head_in_epilogue *head,
                      void *stream) {
  if (M < 0 || N <= 0 || K <= 0) return MI_ERR_INVALID_ARG;
  if (sum_reps < 0 || sum_reps > 64 || (sum_reps > 0 && (!part || !shift_out || ((uintptr_t)part & 7)))) return MI_ERR_INVALID_ARG;
  if (M == 0) return MI_OK;
  if (!X || !W || (!Z && !head)) return MI_ERR_INVALID_ARG;
  if (!vec_ok(X, ldx) || !vec_ok(W, ldw) || (Z && !vec_ok(Z, ldz)) || N % 4 || K % 4) return MI_ERR_UNSUPPORTED;
  FwdArgs a;
  a.bn = BnFwd{};
  if (x_stats) {            // the constants of X are joined in the kernel's prologue and written to x_stats->mu / sc / be
    if (K > kCstPitch) return MI_ERR_UNSUPPORTED;
    if (!bn_fwd_view(x_stats, &a.bn)) return MI_ERR_INVALID_ARG;
    x_mu = x_stats->mu; x_sc = x_stats->sc; x_be = x_stats->be;
  }
  if (x_mu && (!x_sc || !x_be || !aligned16(x_mu) || !aligned16(x_sc) || !aligned16(x_be))) return MI_ERR_INVALID_ARG;
  if (x_mu && x_p > 0.f && (!x_keep || ldx % 8)) return MI_ERR_INVALID_ARG;
  a.x = ActDesc{X, ldx, x_mu, x_sc, x_be, x_p, x_keep};
  a.W = W; a.ldw = ldw; a.Z = Z; a.ldz = ldz; a.part = part;
  a.stat_rep = sum_reps; a.stat_shift = shift_out; a.stat_rm = shift_running_mean; a.stat_off = shift_mean_offset;
  a.head_w = head ? head->w : nullptr;
  if (head) {
    if (!head->w || !head->out || !head->mu || !head->sc || !head->be || part || a_out || sum_reps) return MI_ERR_INVALID_ARG;
    if (!aligned16(head->w) || !aligned16(head->mu) || !aligned16(head->sc) || !aligned16(head->be)) return MI_ERR_UNSUPPORTED;
    a.head_b = head->b; a.head_add = head->add; a.h_mu = head->mu; a.h_sc = head->sc; a.h_be = head->be; a.head_out = head->out;
  }
  if (a_out && (!x_mu || !aligned16(a_out))) return MI_ERR_INVALID_ARG;      // only a transformed operand has anything to keep
  a.a_out = a_out;
  a.M = M; a.N = N; a.K = K;
  a.ncols = cols_per_tile(N, &a.ntn);
  int tiles = ((M + BM - 1) / BM) * a.ntn;
  static const bool dma_env = [] { const char *e = getenv("MI_TAIL_DMA"); return e && e[0] == '1'; }();
  const bool dma = dma_env && K >= 2 * BK;
  // small batches: 32-column tiles put 3.5x as many workgroups on the chip, each with 2/7 of the MFMAs per slice
  const int mt = (M + BM - 1) / BM;
  if (!x_stats && !dma && !part && tiles < 128 && N > 32) {
    a.ncols = 32;
    a.ntn = (N + 31) / 32;
    tiles = mt * a.ntn;
    if (x_mu) MI_LAUNCH("tail_fwd_gemm_small", (k_tail_fwd<true, false, false, 2>), grid8(tiles), kThreads, stream, a);
    else MI_LAUNCH("tail_fwd_gemm_small", (k_tail_fwd<false, false, false, 2>), grid8(tiles), kThreads, stream, a);
    return launch_status();
  }

  if (x_stats) MI_LAUNCH("tail_fwd_gemm", (k_tail_fwd<true, true>), grid8(tiles), kThreads, stream, a);
  else if (x_mu && dma) MI_LAUNCH("tail_fwd_gemm", (k_tail_fwd<true, false, true>), grid8(tiles), kThreads, stream, a);
  else if (x_mu) MI_LAUNCH("tail_fwd_gemm", (k_tail_fwd<true, false>), grid8(tiles), kThreads, stream, a);
  else if (dma) MI_LAUNCH("tail_fwd_gemm", (k_tail_fwd<false, false, true>), grid8(tiles), kThreads, stream, a);
  else MI_LAUNCH("tail_fwd_gemm", (k_tail_fwd<false, false>), grid8(tiles), kThreads, stream, a);
  return launch_status();
}

int mi_tail_fwd_gemm(const float *X, int32_t ldx, const float *x_mu, const float *x_sc, const float *x_be, float x_p,
                     const uint8_t *x_keep, const float *W, int32_t ldw, float *Z, int32_t ldz, float *part, float *a_out,
                     int32_t M, int32_t N, int32_t K, void *stream) {
  return mi_tail_fwd_gemm_m(X, ldx, x_mu, x_sc, x_be, x_p, x_keep, W, ldw, Z, ldz, part, a_out, M, N, K, nullptr, stream);
}

int mi_tail_dropout_masks_z(const int64_t *seed, int32_t nlayers, const int64_t *salts, const float *ps, const int32_t *lds,
                            uint8_t *const *bits, int32_t M, float *zero_buf, int64_t zero_floats, void *stream) {
  MaskJob j;
  int64_t grid;
  const int rc = mask_job(seed, nlayers, salts, ps, lds, bits, M, zero_buf, zero_floats, j, &grid);
  if (rc != MI_OK || grid == 0) return rc;
  MI_LAUNCH("tail_dropout_masks", k_tail_dropmask, (int)grid, kBlock, stream, j, seed, reinterpret_cast<float4 *>(zero_buf),
            zero_floats / 4);
  return launch_status();
}

int mi_tail_dropout_masks(const int64_t *seed, int32_t nlayers, const int64_t *salts, const float *ps, const int32_t *lds,
                          uint8_t *const *bits, int32_t M, void *stream) {
  return mi_tail_dropout_masks_z(seed, nlayers, salts, ps, lds, bits, M, nullptr, 0, stream);
}

int64_t mi_tail_part_elems(int32_t M, int32_t N) { return (int64_t)((M + BM - 1) / BM) * N * 2; }

int mi_tail_bn_finalize_fwd(const float *part, int32_t M, int32_t N, const float *gamma, const float *beta,
                            const float *mean_offset, float *running_mean, float *running_var, float momentum, float eps,
                            int64_t *num_batches_tracked, int64_t *seed_bump, float *mu, float *sc, float *be, float *rstd,
                            void *stream) {
  return mi_tail_bn_finalize_fwd_r(part, M, N, gamma, beta, mean_offset, running_mean, running_var, momentum, eps,
                                   num_batches_tracked, seed_bump, mu, sc, be, rstd, nullptr, stream);
}

int mi_tail_bn_finalize_fwd_r(const float *part, int32_t M, int32_t N, const float *gamma, const float *beta,
                              const float *mean_offset, float *running_mean, float *running_var, float momentum, float eps,
                              int64_t *num_batches_tracked, int64_t *seed_bump, float *mu, float *sc, float *be, float *rstd,
                              const mi_tail_mask_ride *ride, void *stream) {
  if (M <= 0 || N <= 0 || !part || !mu || !sc || !be || !rstd) return MI_ERR_INVALID_ARG;
  const int nfin = (N + kFinCols - 1) / kFinCols;
  MaskRide r;
  r.j.n = 0; r.seed = nullptr; r.zero4 = nullptr; r.nzero4 = 0;
  r.mask_blocks = 0; r.aff.nl = 0;      // (this launch carries the keep bits / zero fill only: ride->affine is not read here)
  int64_t extra = 0;
  if (ride) {
    if (seed_bump && seed_bump == ride->seed) return MI_ERR_INVALID_ARG;      // the bits are drawn from it in this launch
    const int rc = mask_job(ride->seed, ride->nlayers, ride->salts, ride->ps, ride->lds, ride->bits, ride->M, ride->zero_buf,
                            ride->zero_floats, r.j, &extra);
    if (rc != MI_OK) return rc;
    r.seed = ride->seed;
    r.zero4 = reinterpret_cast<float4 *>(ride->zero_buf);
    r.nzero4 = ride->zero_floats / 4;
  }
  if (extra > 0) {
    MI_LAUNCH("tail_bn_finalize_fwd_ride", k_bn_finalize_fwd<true>, nfin + (int)extra, kBlock, stream, part, M, N, gamma, beta,
              mean_offset, running_mean, running_var, momentum, eps, num_batches_tracked, seed_bump, mu, sc, be, rstd, r);
  } else {
    MI_LAUNCH("tail_bn_finalize_fwd", k_bn_finalize_fwd<false>, nfin, kBlock, stream, part, M, N, gamma, beta,
              mean_offset, running_mean, running_var, momentum, eps, num_batches_tracked, seed_bump, mu, sc, be, rstd, r);
  }
  return launch_status();
}

int mi_tail_head_fwd_m(const float *Z, int32_t ldz, const float *mu, const float *sc, const float *be, float p,
                       const uint8_t *keep, const float *w, const float *b, const float *add, float *out, int32_t M,
                       int32_t N, const mi_tail_bn_fwd *stats, void *stream) {
  if (M < 0 || N <= 0) return MI_ERR_INVALID_ARG;
  if (M == 0) return MI_OK;
  BnFwd bn{};
  int64_t *bump = nullptr;
  if (stats && !stats->part) {        // nothing to join: only the seed (if given) is advanced, by the plain kernel
    bump = stats->seed_bump;
    stats = nullptr;
  }
  if (stats) {
    if (N > kCstPitch) return MI_ERR_UNSUPPORTED;
    if (!bn_fwd_view(stats, &bn)) return MI_ERR_INVALID_ARG;
    mu = stats->mu; sc = stats->sc; be = stats->be;
  }
  if (!Z || !mu || !sc || !be || !w || !out) return MI_ERR_INVALID_ARG;
  if (!vec_ok(Z, ldz) || N % 4 || !aligned16(w)) return MI_ERR_UNSUPPORTED;
  if (p > 0.f && (!keep || ldz % 8)) return MI_ERR_INVALID_ARG;
  const ActDesc x{Z, ldz, mu, sc, be, p, keep};
  if (stats) {
    // every workgroup joins the statistics itself: one workgroup per CU, its waves walking the rows
    const int per_cu = (M + kWavesPerBlock - 1) / kWavesPerBlock < 256 ? (M + kWavesPerBlock - 1) / kWavesPerBlock : 256;
    const int grid = per_cu;      // (a wave per row — 1024 workgroups each deriving all constants — measured no better: 0.2428 vs 0.2419 ms)
    MI_LAUNCH("tail_head_fwd", (k_tail_head_fwd<true>), grid, kBlock, stream, x, w, b, add, out, M, N, bn, (int64_t *)nullptr);
  } else {
    MI_LAUNCH("tail_head_fwd", (k_tail_head_fwd<false>), grid_for_waves(M), kBlock, stream, x, w, b, add, out, M, N, bn, bump);
  }
  return launch_status();
}

int mi_tail_head_fwd(const float *Z, int32_t ldz, const float *mu, const float *sc, const float *be, float p,
                     const uint8_t *keep, const float *w, const float *b, const float *add, float *out, int32_t M,
                     int32_t N, void *stream) {
  return mi_tail_head_fwd_m(Z, ldz, mu, sc, be, p, keep, w, b, add, out, M, N, nullptr, stream);
}

int32_t mi_tail_head_blocks(int32_t M) { return M >= 64 * 256 ? 256 : (M + 15) / 16 > 0 ? (M + 15) / 16 : 1; }

int mi_tail_head_bwd(const float *Z, int32_t ldz, const float *mu, const float *sc, const float *be, float p,
                     const uint8_t *keep, const float *g, const float *w, float *DY, float *part, float *wpart,
                     int32_t M, int32_t N, void *stream) {
  return mi_tail_head_bwd_s(Z, ldz, mu, sc, be, p, keep, g, w, DY, part, wpart, 0, M, N, stream);
}

int mi_tail_head_bwd_s(const float *Z, int32_t ldz, const float *mu, const float *sc, const float *be, float p,
                       const uint8_t *keep, const float *g, const float *w, float *DY, float *part, float *wpart,
                       int32_t sum_reps, int32_t M, int32_t N, void *stream) {
  if (M <= 0 || N <= 0 || sum_reps < 0 || sum_reps > 64) return MI_ERR_INVALID_ARG;
  if (!Z || !mu || !sc || !be || !g || !w || !DY || !part || !wpart) return MI_ERR_INVALID_ARG;
  if (!vec_ok(Z, ldz) || N % 4 || N > 4 * kBlock || !aligned16(w) || !aligned16(DY)) return MI_ERR_UNSUPPORTED;
  if (p > 0.f && (!keep || ldz % 8)) return MI_ERR_INVALID_ARG;
  const ActDesc x{Z, ldz, mu, sc, be, p, keep};
  MI_LAUNCH("tail_head_bwd", k_tail_head_bwd, mi_tail_head_blocks(M), kBlock, stream, x, g, w, DY, part, wpart, M, N, sum_reps);
  return launch_status();
}

int32_t mi_tail_head_bce_ws_elems(int32_t sum_reps) { return sum_reps > 0 ? (4 + 2 * sum_reps + 3) / 4 * 4 : 0; }

int mi_tail_head_bce(const float *Z, int32_t ldz, const float *mu, const float *sc, const float *be, float p,
                     const uint8_t *keep, const float *w, const float *b, const float *add, const float *y, float *out,
                     float *g, float *DY, float *part, float *wpart, int32_t sum_reps, float *loss_ws, int32_t M, int32_t N,
                     const mi_tail_bn_fwd *stats, const float *upstream, void *stream) {
  if (M <= 0 || N <= 0 || sum_reps <= 0 || sum_reps > 64) return MI_ERR_INVALID_ARG;
  BnFwd bn{};
  int64_t *bump = nullptr;
  if (stats && !stats->part) {        // nothing to join: only the seed (if given) is advanced
    bump = stats->seed_bump;
    stats = nullptr;
  }
  if (stats) {
    if (!bn_fwd_view(stats, &bn)) return MI_ERR_INVALID_ARG;
    mu = stats->mu; sc = stats->sc; be = stats->be;
  }
  if (!Z || !mu || !sc || !be || !w || !y || !out || !g || !DY || !part || !wpart || !loss_ws) return MI_ERR_INVALID_ARG;
  if (!vec_ok(Z, ldz) || N % 4 || N > kHeadLossCols || !aligned16(w) || !aligned16(DY)) return MI_ERR_UNSUPPORTED;
  if (p > 0.f && (!keep || ldz % 8)) return MI_ERR_INVALID_ARG;
  const ActDesc x{Z, ldz, mu, sc, be, p, keep};
  // one workgroup per CU, 16 rows each at M = 4096 (a single trip of 4 rows per wave)
  const int waves = (M + 3) / 4;
  const int grid = (waves + kWavesPerBlock - 1) / kWavesPerBlock < 256 ? (waves + kWavesPerBlock - 1) / kWavesPerBlock : 256;
  if (stats && bn.nrep > 0)
    MI_LAUNCH("tail_head_bce", (k_tail_head_bce<2>), grid, kBlock, stream, x, w, b, add, y, out, g, DY, part, wpart, loss_ws, M,
              N, bn, sum_reps, (int64_t *)nullptr, upstream);
  else if (stats)
    MI_LAUNCH("tail_head_bce", (k_tail_head_bce<1>), grid, kBlock, stream, x, w, b, add, y, out, g, DY, part, wpart, loss_ws, M,
              N, bn, sum_reps, (int64_t *)nullptr, upstream);
  else
    MI_LAUNCH("tail_head_bce", (k_tail_head_bce<0>), grid, kBlock, stream, x, w, b, add, y, out, g, DY, part, wpart, loss_ws,
              M, N, bn, sum_reps, bump, upstream);
  return launch_status();
}

int mi_tail_bn_finalize_bwd(const float *part, int32_t nblk, int32_t M, int32_t N, const float *gamma, const float *rstd,
                            float *dgamma, float *dbeta, float *al, float *bz, float *de, const float *wpart,
                            int32_t nwblk, float *dw, float *db, void *stream) {
  return mi_tail_bn_finalize_bwd_a(part, nblk, M, N, gamma, rstd, dgamma, dbeta, al, bz, de, wpart, nwblk, dw, db, 0, nullptr,
                                   stream);
}

int mi_tail_bn_finalize_bwd_a(const float *part, int32_t nblk, int32_t M, int32_t N, const float *gamma, const float *rstd,
                              float *dgamma, float *dbeta, float *al, float *bz, float *de, const float *wpart,
                              int32_t nwblk, float *dw, float *db, int32_t affine, float *dbias, void *stream) {
  return mi_tail_bn_finalize_bwd_b(part, nblk, M, N, gamma, rstd, dgamma, dbeta, al, bz, de, wpart, nwblk, dw, db, affine, dbias,
                                   nullptr, stream);
}

int mi_tail_bn_finalize_bwd_b(const float *part, int32_t nblk, int32_t M, int32_t N, const float *gamma, const float *rstd,
                              float *dgamma, float *dbeta, float *al, float *bz, float *de, const float *wpart,
                              int32_t nwblk, float *dw, float *db, int32_t affine, float *dbias, float *db2, void *stream) {
  if (M <= 0 || N <= 0 || nblk <= 0 || !part || !rstd || !al || !bz || !de) return MI_ERR_INVALID_ARG;
  if (wpart && !dw) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("tail_bn_finalize_bwd", k_bn_finalize_bwd, (N + 1 + kFinCols - 1) / kFinCols, kBlock, stream, part, nblk, M, N,
            gamma, rstd, dgamma, dbeta, al, bz, de, wpart, nwblk, dw, db, affine, dbias, db2);
  return launch_status();
}

int mi_tail_affine_consts(int32_t nlayers, const int32_t *widths, const float *const *gamma, const float *const *beta,
                          const float *const *running_mean, const float *const *running_var, const float *const *bias,
                          const float *eps, float *const *mu, float *const *sc, float *const *be, float *const *rstd,
                          void *stream) {
  AffineJob j;
  int blocks = 0;
  const int rc = affine_job(nlayers, widths, gamma, beta, running_mean, running_var, bias, eps, mu, sc, be, rstd, j, &blocks);
  if (rc != MI_OK || blocks == 0) return rc;
  MI_LAUNCH("tail_affine_consts", k_tail_affine_consts, blocks, kBlock, stream, j);
  return launch_status();
}

static int dgrad_launch(const float *DY, const float *Zl, int32_t ld, const float *mu, const float *al, const float *bz,
                        const float *de, const float *W, int32_t ldw, const float *pZ, int32_t pld, const float *p_mu,
                        const float *p_sc, const float *p_be, float p_p, const uint8_t *p_keep, float *OUT, int32_t ldo,
                        float *part, float *dz_out, int32_t M, int32_t N, int32_t K, const mi_tail_bn_bwd *sums,
                        const float *fm_emb, const float *fm_sum, const float *fm_gy, float *fm_g1, int32_t fm_D,
                        int32_t part_reps, void *stream, const int64_t *fm_slot = nullptr);

int mi_tail_dgrad_gemm_m(const float *DY, const float *Zl, int32_t ld, const float *mu, const float *al, const float *bz,
                         const float *de, const float *W, int32_t ldw, const float *pZ, int32_t pld, const float *p_mu,
                         const float *p_sc, const float *p_be, float p_p, const uint8_t *p_keep, float *OUT, int32_t ldo,
                         float *part, float *dz_out, int32_t M, int32_t N, int32_t K, const mi_tail_bn_bwd *sums,
                         void *stream) {
  return dgrad_launch(DY, Zl, ld, mu, al, bz, de, W, ldw, pZ, pld, p_mu, p_sc, p_be, p_p, p_keep, OUT, ldo, part, dz_out, M, N,
                      K, sums, nullptr, nullptr, nullptr, nullptr, 0, 0, stream);
}

int mi_tail_dgrad_gemm_s(const float *DY, const float *Zl, int32_t ld, const float *mu, const float *al, const float *bz,
                         const float *de, const float *W, int32_t ldw, const float *pZ, int32_t pld, const float *p_mu,
                         const float *p_sc, const float *p_be, float p_p, const uint8_t *p_keep, float *OUT, int32_t ldo,
                         float *part, int32_t part_reps, float *dz_out, int32_t M, int32_t N, int32_t K,
                         const mi_tail_bn_bwd *sums, void *stream) {
  if (part_reps < 0 || part_reps > 64 || (part_reps > 0 && !part)) return MI_ERR_INVALID_ARG;
  return dgrad_launch(DY, Zl, ld, mu, al, bz, de, W, ldw, pZ, pld, p_mu, p_sc, p_be, p_p, p_keep, OUT, ldo, part, dz_out, M, N,
                      K, sums, nullptr, nullptr, nullptr, nullptr, 0, part_reps, stream);
}

int mi_tail_dgrad_gemm_fm(const float *DY, const float *Zl, int32_t ld, const float *mu, const float *al, const float *bz,
                          const float *de, const float *W, int32_t ldw, float *gvals, float *dz_out, int32_t M, int32_t N,
                          int32_t K, const mi_tail_bn_bwd *sums, const float *emb, const float *emb_sum, const float *g_y,
                          float *g1vals, int32_t D, void *stream) {
  if (!emb || !emb_sum || !g_y || D <= 0 || D % 4 || K % D) return MI_ERR_INVALID_ARG;
  if (!aligned16(emb) || !aligned16(emb_sum)) return MI_ERR_UNSUPPORTED;
  return dgrad_launch(DY, Zl, ld, mu, al, bz, de, W, ldw, nullptr, K, nullptr, nullptr, nullptr, 0.f, nullptr, gvals, K,
                      nullptr, dz_out, M, N, K, sums, emb, emb_sum, g_y, g1vals, D, 0, stream);
}

int mi_tail_dgrad_gemm_fm_slot(const float *DY, const float *Zl, int32_t ld, const float *mu, const float *al, const float *bz,
                               const float *de, const float *W, int32_t ldw, float *gbuf, float *dz_out, int32_t M, int32_t N,
                               int32_t K, const mi_tail_bn_bwd *sums, const float *emb, const float *emb_sum, const float *g_y,
                               const int64_t *slot, int32_t D, void *stream) {
  if (!emb || !emb_sum || !g_y || !slot || !gbuf || D <= 0 || D % 4 || K % D) return MI_ERR_INVALID_ARG;
  if (!aligned16(emb) || !aligned16(emb_sum) || !aligned16(gbuf)) return MI_ERR_UNSUPPORTED;
  return dgrad_launch(DY, Zl, ld, mu, al, bz, de, W, ldw, nullptr, K, nullptr, nullptr, nullptr, 0.f, nullptr, gbuf, D + 4,
                      nullptr, dz_out, M, N, K, sums, emb, emb_sum, g_y, nullptr, D, 0, stream, slot);
}

static int dgrad_launch(const float *DY, const float *Zl, int32_t ld, const float *mu, const float *al, const float *bz,
                        const float *de, const float *W, int32_t ldw, const float *pZ, int32_t pld, const float *p_mu,
                        const float *p_sc, const float *p_be, float p_p, const uint8_t *p_keep, float *OUT, int32_t ldo,
                        float *part, float *dz_out, int32_t M, int32_t N, int32_t K, const mi_tail_bn_bwd *sums,
                        const float *fm_emb, const float *fm_sum, const float *fm_gy, float *fm_g1, int32_t fm_D,
                        int32_t part_reps, void *stream, const int64_t *fm_slot) {
  if (M < 0 || N <= 0 || K <= 0) return MI_ERR_INVALID_ARG;
  if (M == 0) return MI_OK;
  if (!DY || !W || !OUT) return MI_ERR_INVALID_ARG;
  DgradArgs a;
  a.part_rep = part_reps;
  a.fm_emb = fm_emb; a.fm_sum = fm_sum; a.fm_gy = fm_gy; a.fm_g1 = fm_g1; a.fm_D = fm_D;
  a.fm_slot = fm_slot; a.fm_pitch = fm_D + 4;
  a.bn = BnBwd{};
  if (sums) {               // al / bz / de are joined from the column sums in the kernel's prologue and written to sums->al ...
    if (N > kCstPitch) return MI_ERR_UNSUPPORTED;
    if (!sums->part || !aligned16(sums->part) || sums->nblk <= 0 || !sums->rstd || !sums->al || !sums->bz || !sums->de || !mu || !Zl) return MI_ERR_INVALID_ARG;
    if (sums->wpart && (!sums->dw || sums->nwblk <= 0)) return MI_ERR_INVALID_ARG;
    a.bn = BnBwd{sums->part, sums->nblk, sums->gamma, sums->rstd, mu, sums->dgamma, sums->dbeta, sums->al, sums->bz, sums->de,
                 sums->wpart, sums->nwblk, sums->dw, sums->db, sums->db2, sums->affine, sums->dbias};
    al = sums->al; bz = sums->bz; de = sums->de;
  }
  if (al && (!Zl || !mu || !bz || !de)) return MI_ERR_INVALID_ARG;
  if (p_mu && (!pZ || !p_sc || !p_be)) return MI_ERR_INVALID_ARG;
  if (!vec_ok(DY, ld) || !vec_ok(W, ldw) || !vec_ok(OUT, ldo) || N % 4 || K % 4 || (pZ && !vec_ok(pZ, pld)))
    return MI_ERR_UNSUPPORTED;
  a.dz = DzDesc{DY, Zl, ld, mu, al, bz, de};
  a.W = W; a.ldw = ldw;
  if (p_mu && p_p > 0.f && (!p_keep || pld % 8)) return MI_ERR_INVALID_ARG;
  a.prev = ActDesc{pZ, pld, p_mu, p_sc, p_be, p_p, p_keep};
  a.OUT = OUT; a.ldo = ldo; a.part = part;
  if (dz_out && (!al || !aligned16(dz_out))) return MI_ERR_INVALID_ARG;
  a.dz_out = dz_out;
  a.M = M; a.N = N; a.K = K;
  a.ncols = cols_per_tile(K, &a.ntn);
  const int tiles = ((M + BM - 1) / BM) * a.ntn;
  const bool dz = al != nullptr, mid = p_mu != nullptr;
  if (fm_emb) {
    if (mid) return MI_ERR_INVALID_ARG;
    if (sums) MI_LAUNCH("tail_dgrad_gemm_fm", (k_tail_dgrad<true, false, true, true>), grid8(tiles), kThreads, stream, a);
    else if (dz) MI_LAUNCH("tail_dgrad_gemm_fm", (k_tail_dgrad<true, false, false, true>), grid8(tiles), kThreads, stream, a);
    else MI_LAUNCH("tail_dgrad_gemm_fm", (k_tail_dgrad<false, false, false, true>), grid8(tiles), kThreads, stream, a);
    return launch_status();
  }
  if (sums && mid) MI_LAUNCH("tail_dgrad_gemm", (k_tail_dgrad<true, true, true>), grid8(tiles), kThreads, stream, a);
  else if (sums) MI_LAUNCH("tail_dgrad_gemm", (k_tail_dgrad<true, false, true>), grid8(tiles), kThreads, stream, a);
  else if (dz && mid) MI_LAUNCH("tail_dgrad_gemm", (k_tail_dgrad<true, true, false>), grid8(tiles), kThreads, stream, a);
  else if (dz) MI_LAUNCH("tail_dgrad_gemm", (k_tail_dgrad<true, false, false>), grid8(tiles), kThreads, stream, a);
  else if (mid) MI_LAUNCH("tail_dgrad_gemm", (k_tail_dgrad<false, true, false>), grid8(tiles), kThreads, stream, a);
  else MI_LAUNCH("tail_dgrad_gemm", (k_tail_dgrad<false, false, false>), grid8(tiles), kThreads, stream, a);
  return launch_status();
}

int mi_tail_dgrad_gemm(const float *DY, const float *Zl, int32_t ld, const float *mu, const float *al, const float *bz,
                       const float *de, const float *W, int32_t ldw, const float *pZ, int32_t pld, const float *p_mu,
                       const float *p_sc, const float *p_be, float p_p, const uint8_t *p_keep, float *OUT, int32_t ldo,
                       float *part, float *dz_out, int32_t M, int32_t N, int32_t K, void *stream) {
  return mi_tail_dgrad_gemm_m(DY, Zl, ld, mu, al, bz, de, W, ldw, pZ, pld, p_mu, p_sc, p_be, p_p, p_keep, OUT, ldo, part,
                              dz_out, M, N, K, nullptr, stream);
}

// slices of the batch for the weight-gradient product: as many as keep <= 256 workgroups busy, each a multiple of 32 rows
int32_t mi_tail_wgrad_splits(int32_t M, int32_t N, int32_t K) {
  int ntk;
  cols_per_tile(K, &ntk);
  const int tiles = ((N + 63) / 64) * ntk;
  int s = 256 / (tiles > 0 ? tiles : 1);
  const int max_s = (M + BK - 1) / BK;
  if (s > max_s) s = max_s;
  return s < 1 ? 1 : s;
}

int mi_tail_wgrad_gemm(const float *DY, const float *Zl, int32_t ld, const float *mu, const float *al, const float *bz,
                       const float *de, const float *pZ, int32_t pld, const float *p_mu, const float *p_sc,
                       const float *p_be, float p_p, const uint8_t *p_keep, float *slab, float *dW, int32_t M, int32_t N,
                       int32_t K, void *stream) {
  if (M <= 0 || N <= 0 || K <= 0) return MI_ERR_INVALID_ARG;
  if (!DY || !pZ || !slab || !dW) return MI_ERR_INVALID_ARG;
  if (al && (!Zl || !mu || !bz || !de)) return MI_ERR_INVALID_ARG;
  if (p_mu && (!p_sc || !p_be)) return MI_ERR_INVALID_ARG;
  if (!vec_ok(DY, ld) || !vec_ok(pZ, pld) || N % 4 || K % 4 || !aligned16(slab) || !aligned16(dW)) return MI_ERR_UNSUPPORTED;
  WgradArgs a;
  a.dz = DzDesc{DY, Zl, ld, mu, al, bz, de};
  if (p_mu && p_p > 0.f && (!p_keep || pld % 8)) return MI_ERR_INVALID_ARG;
  a.prev = ActDesc{pZ, pld, p_mu, p_sc, p_be, p_p, p_keep};
  a.slab = slab;
  a.M = M; a.N = N; a.K = K;
  a.ncols = cols_per_tile(K, &a.ntk);
  a.ntn = (N + 63) / 64;
  a.splits = mi_tail_wgrad_splits(M, N, K);
  const int tiles = a.ntn * a.ntk * a.splits;
  const bool dz = al != nullptr, act = p_mu != nullptr;
  if (dz && act) MI_LAUNCH("tail_wgrad_gemm", (k_tail_wgrad<true, true>), grid8(tiles), kThreads, stream, a);
  else if (dz) MI_LAUNCH("tail_wgrad_gemm", (k_tail_wgrad<true, false>), grid8(tiles), kThreads, stream, a);
  else if (act) MI_LAUNCH("tail_wgrad_gemm", (k_tail_wgrad<false, true>), grid8(tiles), kThreads, stream, a);
  else MI_LAUNCH("tail_wgrad_gemm", (k_tail_wgrad<false, false>), grid8(tiles), kThreads, stream, a);
  const int64_t n4 = (int64_t)N * K / 4;
  MI_LAUNCH("tail_slab_sum", k_slab_sum, (int)((n4 + kBlock - 1) / kBlock), kBlock, stream, (const float *)slab, a.splits, n4, dW);
  return launch_status();
}

}  // extern "C"
