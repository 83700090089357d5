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
#include "common.hpp"
#include "tail_gemm.hpp"

namespace {
using namespace mi;
using namespace tg;

__device__ __forceinline__ uint64_t layer_seed(const int64_t *seed, int64_t salt) {
  return (seed ? (uint64_t)seed[0] : 0ull) + 0xD1B54A32D192ED03ull * (uint64_t)salt;
}
__device__ __forceinline__ Drop make_drop(const uint8_t *bits, float p, int ld) {
  Drop d;
  d.bits = p > 0.f ? bits : nullptr;
  d.inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  d.ld = ld;
  return d;
}

// keep bits of up to 8 layers in one launch: byte b of layer l covers elements 8b .. 8b+7 of its [M, ld] activation
struct MaskJob {
  uint8_t *bits[8];
  int64_t salt[8];
  int64_t nbytes[8];       // M * ld / 8
  uint32_t thr[8];
  int n;
};
__global__ __launch_bounds__(kBlock) void k_tail_dropmask(MaskJob j, const int64_t *seed) {
  for (int l = 0; l < j.n; ++l) {
    const uint64_t sd = layer_seed(seed, j.salt[l]);
    const uint32_t thr = j.thr[l];
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < j.nbytes[l]; b += (int64_t)gridDim.x * kBlock) {
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
};

template <bool ACT>
__global__ __launch_bounds__(kThreads) void k_tail_fwd(FwdArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
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
  const KcOperand<BNT, LoadPlain> opC{wp, n0, cols_valid, a.K};
  if constexpr (ACT)
    main_loop<true, true>(acc, lds, 0, a.K, KcOperand<64, Tee<LoadAct>>{Tee<LoadAct>{act, nt == 0 ? a.a_out : nullptr, a.x.ld}, m0, rows_valid, a.K}, opC);
  else main_loop<true, true>(acc, lds, 0, a.K, KcOperand<64, LoadPlain>{xp, m0, rows_valid, a.K}, opC);

  // ---- epilogue.  A lane holds z[m0 + 16 wave + r][n0 + 16 s + 4 g + v] (waves 0-3); the tile goes through LDS once so
  // that ALL 8 waves store whole 448-byte row segments and the column statistics are plain column walks.
  float *T = lds;                                        // [64][kTilePitch]
  acc_to_lds(acc, T, wave, lane);
  __syncthreads();
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = t + k * kThreads, row = i / 28, c = (i % 28) * 4;
    if (i < 64 * 28 && row < rows_valid && c < cols_valid)
      st4(a.Z + (int64_t)(m0 + row) * a.ldz + n0 + c, ld4(T + row * kTilePitch + c));
  }
  if (!a.part) return;
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
    float *o = a.part + ((int64_t)mt * a.N + n0 + t) * 2;
    o[0] = mean_a;
    o[1] = m2_a;
  }
}

// Merge the per-tile statistics (fixed order), produce the constants the next loads need, update the running statistics
// like F.batch_norm(training=True) does (momentum; UNBIASED variance), count the batch, bump the dropout seed once per
// step.  mean_offset: the Linear's bias, which the contraction left out because it cancels in the normalisation — it
// only shifts the running mean.
__global__ __launch_bounds__(kBlock) void k_bn_finalize_fwd(const float *__restrict__ part, int M, int N,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            const float *__restrict__ mean_offset, float *running_mean,
                                                            float *running_var, float momentum, float eps,
                                                            int64_t *nbt, int64_t *seed_bump, float *__restrict__ mu,
                                                            float *__restrict__ sc, float *__restrict__ be,
                                                            float *__restrict__ rstd_out) {
  // 64 columns x 4 tile groups per workgroup: a group merges its run of 64-row tiles in tile order (8 loads in flight —
  // a single thread walking all tiles is one exposed memory latency per tile: 19 us at 64 tiles), then the 4 groups are
  // merged in group order through LDS.  Same tree for every launch: deterministic.
  __shared__ float sh[4][64][3];
  const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + cl;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (nbt) nbt[0] += 1;
    if (seed_bump) seed_bump[0] += 1;
  }
  const int MT = (M + BM - 1) / BM;
  const int per = (MT + 3) / 4, t0 = grp * per, t1 = min(MT, t0 + per);
  float n_a = 0.f, mean_a = 0.f, m2_a = 0.f;
  if (n < N) {
    for (int base = t0; base < t1; base += 8) {
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = min(base + u, t1 - 1);
        v[u] = *reinterpret_cast<const float2 *>(part + ((int64_t)t * N + n) * 2);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = base + u;
        if (t < t1) {
          const float n_b = (float)min(BM, M - t * BM);
          const float tot = n_a + n_b, delta = v[u].x - mean_a;
          mean_a += delta * (n_b / tot);
          m2_a += v[u].y + delta * delta * (n_a * n_b / tot);
          n_a = tot;
        }
      }
    }
  }
  sh[grp][cl][0] = n_a; sh[grp][cl][1] = mean_a; sh[grp][cl][2] = m2_a;
  __syncthreads();
  if (grp != 0 || n >= N) return;
  n_a = 0.f; mean_a = 0.f; m2_a = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
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
  const float gm = gamma ? gamma[n] : 1.f;
  mu[n] = mean_a;
  sc[n] = gm * rstd;
  be[n] = beta ? beta[n] : 0.f;
  rstd_out[n] = rstd;
  if (running_mean) {
    const float mo = mean_offset ? mean_offset[n] : 0.f;
    running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * (mean_a + mo);
    const float unb = M > 1 ? m2_a / (float)(M - 1) : var;
    running_var[n] = (1.f - momentum) * running_var[n] + momentum * unb;
  }
}

// ============================================================================================== head (Linear(., 1)) ====
// out[m] = sum_n a(m, n) w[n] + b + add[m]: a wave per row group, float4 per lane over the features.
__global__ __launch_bounds__(kBlock) void k_tail_head_fwd(ActDesc x, const float *__restrict__ w, const float *__restrict__ b,
                                                          const float *__restrict__ add, float *__restrict__ out, int M,
                                                          int N) {
  const LoadAct act = make_act(x);
  const int lane = threadIdx.x & 63;
  const int wave0 = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), nw = gridDim.x * kWavesPerBlock;
  const float bv = b ? b[0] : 0.f;
  for (int m = wave0; m < M; m += nw) {
    float s = 0.f;
    for (int c = lane * 4; c < N; c += 256) {
      const float4 a = act.finish(act.fetch(m, c), act.consts(c), m, c), ww = ld4(w + c);
      s += a.x * ww.x + a.y * ww.y + a.z * ww.z + a.w * ww.w;
    }
    s = wave_sum(s);
    if (lane == 0) out[m] = s + bv + (add ? add[m] : 0.f);
  }
}

// Backward of the head into the last hidden layer: da(m, n) = g[m] w[n];
//   dy(m, n) = da * keepscale * [pre > 0]                       -> DY
//   part[blk][n] = (sum_m dy, sum_m dy * (z - mu))               (dbeta / dgamma pieces)
//   wpart[blk][n] = sum_m g[m] * a(m, n),  wpart[blk][N] = sum_m g[m]    (dw / db pieces)
// One workgroup = 16 rows x all features per trip; per-workgroup partials are joined by k_bn_finalize_bwd in block
// order (deterministic).
__global__ __launch_bounds__(kBlock) void k_tail_head_bwd(ActDesc x, const float *__restrict__ g, const float *__restrict__ w,
                                                          float *__restrict__ DY, float *__restrict__ part,
                                                          float *__restrict__ wpart, int M, int N) {
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
  for (int m4 = mb; m4 < me; m4 += 4) {          // 4 rows in flight per thread
    float4 z[4];
    uint32_t kb[4];
    float gv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = min(m4 + u, me - 1);
      gv[u] = g[m];
      if (cv) {
        z[u] = ld4(x.Z + (int64_t)m * x.ld + c);
        kb[u] = drop.fetch(m, c);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
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
  if (cv) {
    float *o = part + ((int64_t)blockIdx.x * N + c) * 2;
    st4(o, make_float4(s_dy.x, s_dyz.x, s_dy.y, s_dyz.y));
    st4(o + 4, make_float4(s_dy.z, s_dyz.z, s_dy.w, s_dyz.w));
    st4(wpart + (int64_t)blockIdx.x * (N + 4) + c, s_ga);
  }
  if (threadIdx.x == 0) wpart[(int64_t)blockIdx.x * (N + 4) + N] = s_g;
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
                                                            int nwblk, float *__restrict__ dw, float *__restrict__ db) {
  // 32 columns x 8 row groups per workgroup (see k_bn_finalize_fwd): a group adds its run of partial rows in row order
  // with 8 loads in flight, the 8 groups are added in group order through LDS.  Column N of the head's pieces is db.
  __shared__ float sh[8][32][3];
  const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + cl;
  float s1 = 0.f, s2 = 0.f, sw = 0.f;
  if (n < N) {
    const int per = (nblk + 7) / 8, t0 = grp * per, t1 = min(nblk, t0 + per);
    for (int base = t0; base < t1; base += 8) {
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float2 *>(part + ((int64_t)min(base + u, t1 - 1) * N + n) * 2);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (base + u < t1) { s1 += v[u].x; s2 += v[u].y; }
    }
  }
  if (wpart && n <= N) {
    const int per = (nwblk + 7) / 8, t0 = grp * per, t1 = min(nwblk, t0 + per);
    for (int base = t0; base < t1; base += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = wpart[(int64_t)min(base + u, t1 - 1) * (N + 4) + n];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (base + u < t1) sw += v[u];
    }
  }
  sh[grp][cl][0] = s1; sh[grp][cl][1] = s2; sh[grp][cl][2] = sw;
  __syncthreads();
  if (grp != 0 || n > N) return;
  s1 = 0.f; s2 = 0.f; sw = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) { s1 += sh[q][cl][0]; s2 += sh[q][cl][1]; sw += sh[q][cl][2]; }
  if (wpart) {
    if (n < N) dw[n] = sw; else if (db) db[0] = sw;
  }
  if (n >= N) return;
  const float r = rstd[n], gm = gamma ? gamma[n] : 1.f;
  const float dg = s2 * r;
  if (dgamma) dgamma[n] = dg;
  if (dbeta) dbeta[n] = s1;
  const float inv_m = 1.f / (float)M;
  al[n] = gm * r;
  bz[n] = -gm * r * r * dg * inv_m;
  de[n] = -gm * r * s1 * inv_m;
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
  float *dz_out;      // [M, N] (pitch dz.ld), nullable: dz as the operand load computed it
  int M, N, K;
  int ncols, ntn;     // column tiles over K
};

template <bool DZ, bool MID>
__global__ __launch_bounds__(kThreads) void k_tail_dgrad(DgradArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
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
  if constexpr (DZ)
    main_loop<true, true>(acc, lds, 0, a.N, KcOperand<64, Tee<LoadDz>>{Tee<LoadDz>{dzl, nt == 0 ? a.dz_out : nullptr, a.dz.ld}, m0, rows_valid, a.N}, opC);
  else main_loop<true, true>(acc, lds, 0, a.N, KcOperand<64, LoadPlain>{dyp, m0, rows_valid, a.N}, opC);

  // ---- epilogue through LDS (see k_tail_fwd): all 8 waves, whole row segments
  float *T = lds;
  acc_to_lds(acc, T, wave, lane);
  __syncthreads();
  const int t = threadIdx.x;
  if constexpr (!MID) {
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
      float *o = a.part + ((int64_t)mt * a.K + k0 + t) * 2;
      o[0] = s1; o[1] = s2;
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
  auto run = [&](const auto &opR) {       // both operands have the batch as the slow index: transposed into KC tiles
    if constexpr (ACT) main_loop<true, true>(acc, lds, mb, me, opR, OtOperand<BNT, LoadAct>{act, k0, cols_valid, me});
    else main_loop<true, true>(acc, lds, mb, me, opR, OtOperand<BNT, LoadPlain>{xp, k0, cols_valid, me});
  };
  if constexpr (DZ) run(OtOperand<64, LoadDz>{dzl, n0, nrows_valid, me});
  else run(OtOperand<64, LoadPlain>{dyp, n0, nrows_valid, me});

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

int mi_tail_fwd_gemm(const float *X, int32_t ldx, const float *x_mu, const float *x_sc, const float *x_be, float x_p,
                     const uint8_t *x_keep, const float *W, int32_t ldw, float *Z, int32_t ldz, float *part, float *a_out,
                     int32_t M, int32_t N, int32_t K, void *stream) {
  if (M < 0 || N <= 0 || K <= 0) return MI_ERR_INVALID_ARG;
  if (M == 0) return MI_OK;
  if (!X || !W || !Z) return MI_ERR_INVALID_ARG;
  if (!vec_ok(X, ldx) || !vec_ok(W, ldw) || !vec_ok(Z, ldz) || N % 4 || K % 4) return MI_ERR_UNSUPPORTED;
  if (x_mu && (!x_sc || !x_be || !aligned16(x_mu) || !aligned16(x_sc) || !aligned16(x_be))) return MI_ERR_INVALID_ARG;
  FwdArgs a;
  if (x_mu && x_p > 0.f && (!x_keep || ldx % 8)) return MI_ERR_INVALID_ARG;
  a.x = ActDesc{X, ldx, x_mu, x_sc, x_be, x_p, x_keep};
  a.W = W; a.ldw = ldw; a.Z = Z; a.ldz = ldz; a.part = part;
  if (a_out && (!x_mu || !aligned16(a_out))) return MI_ERR_INVALID_ARG;      // only a transformed operand has anything to keep
  a.a_out = a_out;
  a.M = M; a.N = N; a.K = K;
  a.ncols = cols_per_tile(N, &a.ntn);
  const int tiles = ((M + BM - 1) / BM) * a.ntn;
  if (x_mu) MI_LAUNCH("tail_fwd_gemm", (k_tail_fwd<true>), grid8(tiles), kThreads, stream, a);
  else MI_LAUNCH("tail_fwd_gemm", (k_tail_fwd<false>), grid8(tiles), kThreads, stream, a);
  return launch_status();
}

int mi_tail_dropout_masks(const int64_t *seed, int32_t nlayers, const int64_t *salts, const float *ps, const int32_t *lds,
                          uint8_t *const *bits, int32_t M, void *stream) {
  if (nlayers < 0 || nlayers > 8 || M < 0) return MI_ERR_INVALID_ARG;
  if (nlayers == 0 || M == 0) return MI_OK;
  if (!seed || !salts || !ps || !lds || !bits) return MI_ERR_INVALID_ARG;
  MaskJob j;
  j.n = 0;
  int64_t most = 0;
  for (int l = 0; l < nlayers; ++l) {
    if (ps[l] <= 0.f) continue;
    if (!bits[l] || lds[l] <= 0 || lds[l] % 8 || ps[l] >= 1.f) return MI_ERR_INVALID_ARG;
    j.bits[j.n] = bits[l];
    j.salt[j.n] = salts[l];
    j.nbytes[j.n] = (int64_t)M * lds[l] / 8;
    j.thr[j.n] = (uint32_t)(ps[l] * 65536.f + 0.5f);
    most = j.nbytes[j.n] > most ? j.nbytes[j.n] : most;
    ++j.n;
  }
  if (j.n == 0) return MI_OK;
  int64_t grid = (most + kBlock - 1) / kBlock;
  if (grid > kMaxGrid) grid = kMaxGrid;
  MI_LAUNCH("tail_dropout_masks", k_tail_dropmask, (int)grid, kBlock, stream, j, seed);
  return launch_status();
}

int64_t mi_tail_part_elems(int32_t M, int32_t N) { return (int64_t)((M + BM - 1) / BM) * N * 2; }

int mi_tail_bn_finalize_fwd(const float *part, int32_t M, int32_t N, const float *gamma, const float *beta,
                            const float *mean_offset, float *running_mean, float *running_var, float momentum, float eps,
                            int64_t *num_batches_tracked, int64_t *seed_bump, float *mu, float *sc, float *be, float *rstd,
                            void *stream) {
  if (M <= 0 || N <= 0 || !part || !mu || !sc || !be || !rstd) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("tail_bn_finalize_fwd", k_bn_finalize_fwd, (N + 63) / 64, kBlock, stream, part, M, N, gamma, beta,
            mean_offset, running_mean, running_var, momentum, eps, num_batches_tracked, seed_bump, mu, sc, be, rstd);
  return launch_status();
}

int mi_tail_head_fwd(const float *Z, int32_t ldz, const float *mu, const float *sc, const float *be, float p,
                     const uint8_t *keep, const float *w, const float *b, const float *add, float *out, int32_t M,
                     int32_t N, void *stream) {
  if (M < 0 || N <= 0) return MI_ERR_INVALID_ARG;
  if (M == 0) return MI_OK;
  if (!Z || !mu || !sc || !be || !w || !out) return MI_ERR_INVALID_ARG;
  if (!vec_ok(Z, ldz) || N % 4 || !aligned16(w)) return MI_ERR_UNSUPPORTED;
  if (p > 0.f && (!keep || ldz % 8)) return MI_ERR_INVALID_ARG;
  const ActDesc x{Z, ldz, mu, sc, be, p, keep};
  MI_LAUNCH("tail_head_fwd", k_tail_head_fwd, grid_for_waves(M), kBlock, stream, x, w, b, add, out, M, N);
  return launch_status();
}

int32_t mi_tail_head_blocks(int32_t M) { return M >= 64 * 256 ? 256 : (M + 15) / 16 > 0 ? (M + 15) / 16 : 1; }

int mi_tail_head_bwd(const float *Z, int32_t ldz, const float *mu, const float *sc, const float *be, float p,
                     const uint8_t *keep, const float *g, const float *w, float *DY, float *part, float *wpart,
                     int32_t M, int32_t N, void *stream) {
  if (M <= 0 || N <= 0) return MI_ERR_INVALID_ARG;
  if (!Z || !mu || !sc || !be || !g || !w || !DY || !part || !wpart) return MI_ERR_INVALID_ARG;
  if (!vec_ok(Z, ldz) || N % 4 || N > 4 * kBlock || !aligned16(w) || !aligned16(DY)) return MI_ERR_UNSUPPORTED;
  if (p > 0.f && (!keep || ldz % 8)) return MI_ERR_INVALID_ARG;
  const ActDesc x{Z, ldz, mu, sc, be, p, keep};
  MI_LAUNCH("tail_head_bwd", k_tail_head_bwd, mi_tail_head_blocks(M), kBlock, stream, x, g, w, DY, part, wpart, M, N);
  return launch_status();
}

int mi_tail_bn_finalize_bwd(const float *part, int32_t nblk, int32_t M, int32_t N, const float *gamma, const float *rstd,
                            float *dgamma, float *dbeta, float *al, float *bz, float *de, const float *wpart,
                            int32_t nwblk, float *dw, float *db, void *stream) {
  if (M <= 0 || N <= 0 || nblk <= 0 || !part || !rstd || !al || !bz || !de) return MI_ERR_INVALID_ARG;
  if (wpart && !dw) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("tail_bn_finalize_bwd", k_bn_finalize_bwd, (N + 1 + 31) / 32, kBlock, stream, part, nblk, M, N,
            gamma, rstd, dgamma, dbeta, al, bz, de, wpart, nwblk, dw, db);
  return launch_status();
}

int mi_tail_dgrad_gemm(const float *DY, const float *Zl, int32_t ld, const float *mu, const float *al, const float *bz,
                       const float *de, const float *W, int32_t ldw, const float *pZ, int32_t pld, const float *p_mu,
                       const float *p_sc, const float *p_be, float p_p, const uint8_t *p_keep, float *OUT, int32_t ldo,
                       float *part, float *dz_out, int32_t M, int32_t N, int32_t K, void *stream) {
  if (M < 0 || N <= 0 || K <= 0) return MI_ERR_INVALID_ARG;
  if (M == 0) return MI_OK;
  if (!DY || !W || !OUT) return MI_ERR_INVALID_ARG;
  if (al && (!Zl || !mu || !bz || !de)) return MI_ERR_INVALID_ARG;
  if (p_mu && (!pZ || !p_sc || !p_be)) return MI_ERR_INVALID_ARG;
  if (!vec_ok(DY, ld) || !vec_ok(W, ldw) || !vec_ok(OUT, ldo) || N % 4 || K % 4 || (pZ && !vec_ok(pZ, pld)))
    return MI_ERR_UNSUPPORTED;
  DgradArgs a;
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
  if (dz && mid) MI_LAUNCH("tail_dgrad_gemm", (k_tail_dgrad<true, true>), grid8(tiles), kThreads, stream, a);
  else if (dz) MI_LAUNCH("tail_dgrad_gemm", (k_tail_dgrad<true, false>), grid8(tiles), kThreads, stream, a);
  else if (mid) MI_LAUNCH("tail_dgrad_gemm", (k_tail_dgrad<false, true>), grid8(tiles), kThreads, stream, a);
  else MI_LAUNCH("tail_dgrad_gemm", (k_tail_dgrad<false, false>), grid8(tiles), kThreads, stream, a);
  return launch_status();
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
