// mish_mlp.hip — the element work of DHEmbedding's MLP (reference
// src/models/embeddings/dh_embedding.py:100-117,345-356): per layer Linear -> {Mish | BatchNorm1d -> Mish | Mish ->
// BatchNorm1d} (use_bn = 0 / 2 / 1).  The contractions run on tail_gemm.hpp's MFMA products (mi_tail_fwd_gemm_s /
// mi_tail_dgrad_gemm_s with plain operands, mi_gemm_f32_multi for the weight gradients); what is here is the column-wise
// affine + activation between them, forward and backward, with the BatchNorm column statistics / gradient column sums
// emitted as per-tile partials in the layouts the tail's finalize kernels (tail.hip) already join:
//   forward   out = ACT((in - mu[c]) * sc[c] + be[c])                 ACT = mish or identity
//             part[tile][c] = (mean, M2) of `out` over the tile's 64 rows      -> mi_tail_bn_finalize_fwd
//   backward  dy = g * ACT'(pre),  part[tile][c] = (sum dy, sum dy (in - mu))   -> mi_tail_bn_finalize_bwd[_a]
//   use_bn=1  dz = (al g + bz (m - mu) + de) * mish'(z + b)  (BatchNorm backward, then Mish backward), column sums of dz
// HBM-bound passes: one read of each input, one write of each output, float4 per lane, a 64-row x 256-column tile per
// 256-thread workgroup (thread = 4 consecutive columns x 16 of the rows... see below).
//
// Mish as torch computes it on the CPU (aten/src/ATen/native/cpu/Activation.cpp): x * tanh(log1p(exp(x))), and its
// derivative tanh(sp) + x * sigmoid(x) * (1 - tanh(sp)^2).
#include "common.hpp"

namespace {
using namespace mi;

__device__ __forceinline__ float mish_f(float x) { return x * tanhf(log1pf(expf(x))); }
__device__ __forceinline__ float mish_d(float x) {
  const float t = tanhf(log1pf(expf(x)));
  const float sg = 1.f / (1.f + expf(-x));
  return t + x * sg * (1.f - t * t);
}

constexpr int kRows = 64;        // rows per tile (the tail's BM: the finalize kernels count 64-row tiles)
constexpr int kColsPerBlk = 256; // columns per workgroup: 64 threads x float4, 4 row groups of 16 rows

// thread (cq = t % 64, rg = t / 64): columns c0 + 4 cq .. + 3, rows 16 rg .. 16 rg + 15 of the tile
template <bool MISH>
__global__ __launch_bounds__(kBlock) void k_col_act_fwd(const float *__restrict__ in, int ld, const float *__restrict__ mu,
                                                        const float *__restrict__ sc, const float *__restrict__ be,
                                                        float *__restrict__ out, float *__restrict__ part, int M, int N) {
  __shared__ float ws[4][kColsPerBlk][3];
  const int t = threadIdx.x, cq = t & 63, rg = t >> 6;
  const int c = blockIdx.y * kColsPerBlk + cq * 4;
  const int m0 = blockIdx.x * kRows + rg * 16;
  const bool cv = c < N;
  float4 u = make_float4(0.f, 0.f, 0.f, 0.f), s = make_float4(1.f, 1.f, 1.f, 1.f), b = u;
  if (cv) {
    if (mu) u = ld4(mu + c);
    if (sc) s = ld4(sc + c);
    if (be) b = ld4(be + c);
  }
  float4 shift = make_float4(0.f, 0.f, 0.f, 0.f), s1 = shift, s2 = shift;
  int cnt = 0;
  if (cv) {
    float4 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = min(m0 + r, M - 1);
      v[r] = ld4(in + (int64_t)m * ld + c);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + r;
      if (m >= M) break;
      float4 o;
      o.x = fmaf(v[r].x - u.x, s.x, b.x); o.y = fmaf(v[r].y - u.y, s.y, b.y);
      o.z = fmaf(v[r].z - u.z, s.z, b.z); o.w = fmaf(v[r].w - u.w, s.w, b.w);
      if (MISH) { o.x = mish_f(o.x); o.y = mish_f(o.y); o.z = mish_f(o.z); o.w = mish_f(o.w); }
      st4(out + (int64_t)m * N + c, o);
      if (cnt == 0) shift = o;
      else {
        const float4 d = make_float4(o.x - shift.x, o.y - shift.y, o.z - shift.z, o.w - shift.w);
        s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
        s2.x += d.x * d.x; s2.y += d.y * d.y; s2.z += d.z * d.z; s2.w += d.w * d.w;
      }
      ++cnt;
    }
  }
  if (!part) return;
  // (mean, M2) of the tile's rows per column: the 4 row groups merged in group order (Chan), like tail.hip's epilogue
  const float sh[4] = {shift.x, shift.y, shift.z, shift.w}, a1[4] = {s1.x, s1.y, s1.z, s1.w}, a2[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ws[rg][cq * 4 + j][0] = (float)cnt;
    ws[rg][cq * 4 + j][1] = cnt ? sh[j] + a1[j] / (float)cnt : 0.f;
    ws[rg][cq * 4 + j][2] = cnt ? a2[j] - a1[j] * a1[j] / (float)cnt : 0.f;
  }
  __syncthreads();
  const int col = blockIdx.y * kColsPerBlk + t;
  if (col < N) {
    float n_a = 0.f, mean_a = 0.f, m2_a = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float n_b = ws[w][t][0];
      if (n_b > 0.f) {
        const float n = n_a + n_b, delta = ws[w][t][1] - mean_a;
        mean_a += delta * (n_b / n);
        m2_a += ws[w][t][2] + delta * delta * (n_a * n_b / n);
        n_a = n;
      }
    }
    float *o = part + ((int64_t)blockIdx.x * N + col) * 2;
    o[0] = mean_a;
    o[1] = m2_a;
  }
}

// dy = g * ACT'((in - mu) sc + be)   (dy nullable: identity needs no copy), part[tile][c] = (sum dy, sum dy (in - mu))
template <bool MISH>
__global__ __launch_bounds__(kBlock) void k_col_act_bwd(const float *__restrict__ g, const float *__restrict__ in, int ld,
                                                        const float *__restrict__ mu, const float *__restrict__ sc,
                                                        const float *__restrict__ be, float *__restrict__ dy,
                                                        float *__restrict__ part, int M, int N) {
  __shared__ float ws[4][kColsPerBlk][2];
  const int t = threadIdx.x, cq = t & 63, rg = t >> 6;
  const int c = blockIdx.y * kColsPerBlk + cq * 4;
  const int m0 = blockIdx.x * kRows + rg * 16;
  const bool cv = c < N;
  float4 u = make_float4(0.f, 0.f, 0.f, 0.f), s = make_float4(1.f, 1.f, 1.f, 1.f), b = u;
  if (cv) {
    if (mu) u = ld4(mu + c);
    if (sc) s = ld4(sc + c);
    if (be) b = ld4(be + c);
  }
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (cv) {
    float4 v[8], gg[8];
    for (int r0 = 0; r0 < 16; r0 += 8) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int m = min(m0 + r0 + r, M - 1);
        v[r] = ld4(in + (int64_t)m * ld + c);
        gg[r] = ld4(g + (int64_t)m * N + c);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int m = m0 + r0 + r;
        if (m >= M) break;
        const float4 zc = make_float4(v[r].x - u.x, v[r].y - u.y, v[r].z - u.z, v[r].w - u.w);
        float4 d = gg[r];
        if (MISH) {
          d.x *= mish_d(fmaf(zc.x, s.x, b.x)); d.y *= mish_d(fmaf(zc.y, s.y, b.y));
          d.z *= mish_d(fmaf(zc.z, s.z, b.z)); d.w *= mish_d(fmaf(zc.w, s.w, b.w));
        }
        if (dy) st4(dy + (int64_t)m * N + c, d);
        s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
        s2.x += d.x * zc.x; s2.y += d.y * zc.y; s2.z += d.z * zc.z; s2.w += d.w * zc.w;
      }
    }
  }
  const float a1[4] = {s1.x, s1.y, s1.z, s1.w}, a2[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) { ws[rg][cq * 4 + j][0] = a1[j]; ws[rg][cq * 4 + j][1] = a2[j]; }
  __syncthreads();
  const int col = blockIdx.y * kColsPerBlk + t;
  if (col < N) {
    float *o = part + ((int64_t)blockIdx.x * N + col) * 2;
    o[0] = (ws[0][t][0] + ws[1][t][0]) + (ws[2][t][0] + ws[3][t][0]);
    o[1] = (ws[0][t][1] + ws[1][t][1]) + (ws[2][t][1] + ws[3][t][1]);
  }
}

// use_bn = 1 (Linear -> Mish -> BatchNorm): a = (m - mu) sc + be with m = mish(z + b).
//   dm = al g + bz (m - mu) + de   (al = gamma rstd, bz / de the batch terms, 0 in eval mode: LoadDz's constants)
//   dz = dm * mish'(z + b);  part[tile][c] = (sum dz, 0)   (dbias = sum dz)
__global__ __launch_bounds__(kBlock) void k_bn_mish_bwd(const float *__restrict__ g, const float *__restrict__ m_act,
                                                        const float *__restrict__ z, int ldz, const float *__restrict__ bias,
                                                        const float *__restrict__ mu, const float *__restrict__ al,
                                                        const float *__restrict__ bz, const float *__restrict__ de,
                                                        float *__restrict__ dz, float *__restrict__ part, int M, int N) {
  __shared__ float ws[4][kColsPerBlk];
  const int t = threadIdx.x, cq = t & 63, rg = t >> 6;
  const int c = blockIdx.y * kColsPerBlk + cq * 4;
  const int m0 = blockIdx.x * kRows + rg * 16;
  const bool cv = c < N;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 u = zero, a = make_float4(1.f, 1.f, 1.f, 1.f), bb = zero, d0 = zero, bi = zero;
  if (cv) {
    u = ld4(mu + c); a = ld4(al + c); bb = ld4(bz + c); d0 = ld4(de + c);
    if (bias) bi = ld4(bias + c);
  }
  float4 s1 = zero;
  if (cv) {
    for (int r0 = 0; r0 < 16; r0 += 4) {
      float4 gg[4], mm[4], zz[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = min(m0 + r0 + r, M - 1);
        gg[r] = ld4(g + (int64_t)m * N + c);
        mm[r] = ld4(m_act + (int64_t)m * N + c);
        zz[r] = ld4(z + (int64_t)m * ldz + c);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + r0 + r;
        if (m >= M) break;
        float4 d;
        d.x = fmaf(a.x, gg[r].x, fmaf(bb.x, mm[r].x - u.x, d0.x)) * mish_d(zz[r].x + bi.x);
        d.y = fmaf(a.y, gg[r].y, fmaf(bb.y, mm[r].y - u.y, d0.y)) * mish_d(zz[r].y + bi.y);
        d.z = fmaf(a.z, gg[r].z, fmaf(bb.z, mm[r].z - u.z, d0.z)) * mish_d(zz[r].z + bi.z);
        d.w = fmaf(a.w, gg[r].w, fmaf(bb.w, mm[r].w - u.w, d0.w)) * mish_d(zz[r].w + bi.w);
        st4(dz + (int64_t)m * N + c, d);
        s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
      }
    }
  }
  const float a1[4] = {s1.x, s1.y, s1.z, s1.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) ws[rg][cq * 4 + j] = a1[j];
  __syncthreads();
  const int col = blockIdx.y * kColsPerBlk + t;
  if (col < N) {
    float *o = part + ((int64_t)blockIdx.x * N + col) * 2;
    o[0] = (ws[0][t] + ws[1][t]) + (ws[2][t] + ws[3][t]);
    o[1] = 0.f;
  }
}

// dz = al dy + bz (z - mu) + de (the BatchNorm backward of tail_gemm.hpp's LoadDz, materialised): for a layer whose
// input needs no gradient — the first: its input is the hash features — no input-gradient product exists whose operand load
// would compute dz on the fly, and the weight-gradient product wants it as a plain matrix
__global__ __launch_bounds__(kBlock) void k_bn_dz(const float *__restrict__ dy, const float *__restrict__ z, int ldz,
                                                  const float *__restrict__ mu, const float *__restrict__ al,
                                                  const float *__restrict__ bz, const float *__restrict__ de,
                                                  float *__restrict__ dz, int64_t n4, int N) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const int64_t e = i * 4, m = e / N;
    const int c = (int)(e - m * N);
    const float4 u = ld4(mu + c), a = ld4(al + c), b = ld4(bz + c), d = ld4(de + c);
    const float4 g = ld4(dy + e), zz = ld4(z + m * ldz + c);
    st4(dz + e, make_float4(fmaf(a.x, g.x, fmaf(b.x, zz.x - u.x, d.x)), fmaf(a.y, g.y, fmaf(b.y, zz.y - u.y, d.y)),
                            fmaf(a.z, g.z, fmaf(b.z, zz.z - u.z, d.z)), fmaf(a.w, g.w, fmaf(b.w, zz.w - u.w, d.w))));
  }
}

inline dim3 tile_grid(int M, int N) { return dim3((M + kRows - 1) / kRows, (N + kColsPerBlk - 1) / kColsPerBlk); }
inline bool ok4(const void *p) { return p == nullptr || aligned16(p); }

}  // namespace

extern "C" {

int mi_col_act_fwd(const float *in, int32_t ld, const float *mu, const float *sc, const float *be, int32_t act, float *out,
                   float *part, int32_t M, int32_t N, void *stream) {
  if (M < 0 || N <= 0 || (act != 0 && act != 1)) return MI_ERR_INVALID_ARG;
  if (M == 0) return MI_OK;
  if (!in || !out) return MI_ERR_INVALID_ARG;
  if (N % 4 || ld % 4 || ld < N || !aligned16(in) || !aligned16(out) || !ok4(mu) || !ok4(sc) || !ok4(be)) return MI_ERR_UNSUPPORTED;
  if (act == 1) MI_LAUNCH("col_act_fwd", (k_col_act_fwd<true>), tile_grid(M, N), kBlock, stream, in, ld, mu, sc, be, out, part, M, N);
  else MI_LAUNCH("col_act_fwd", (k_col_act_fwd<false>), tile_grid(M, N), kBlock, stream, in, ld, mu, sc, be, out, part, M, N);
  return launch_status();
}

int mi_col_act_bwd(const float *g, const float *in, int32_t ld, const float *mu, const float *sc, const float *be, int32_t act,
                   float *dy, float *part, int32_t M, int32_t N, void *stream) {
  if (M <= 0 || N <= 0 || (act != 0 && act != 1)) return MI_ERR_INVALID_ARG;
  if (!g || !in || !part) return MI_ERR_INVALID_ARG;
  if (N % 4 || ld % 4 || ld < N || !aligned16(in) || !aligned16(g) || !ok4(dy) || !ok4(mu) || !ok4(sc) || !ok4(be)) return MI_ERR_UNSUPPORTED;
  if (act == 1) MI_LAUNCH("col_act_bwd", (k_col_act_bwd<true>), tile_grid(M, N), kBlock, stream, g, in, ld, mu, sc, be, dy, part, M, N);
  else MI_LAUNCH("col_act_bwd", (k_col_act_bwd<false>), tile_grid(M, N), kBlock, stream, g, in, ld, mu, sc, be, dy, part, M, N);
  return launch_status();
}

int mi_bn_mish_bwd(const float *g, const float *m_act, const float *z, int32_t ldz, const float *bias, const float *mu,
                   const float *al, const float *bz, const float *de, float *dz, float *part, int32_t M, int32_t N,
                   void *stream) {
  if (M <= 0 || N <= 0) return MI_ERR_INVALID_ARG;
  if (!g || !m_act || !z || !mu || !al || !bz || !de || !dz || !part) return MI_ERR_INVALID_ARG;
  if (N % 4 || ldz % 4 || ldz < N || !aligned16(g) || !aligned16(m_act) || !aligned16(z) || !aligned16(dz) || !ok4(bias) ||
      !aligned16(mu) || !aligned16(al) || !aligned16(bz) || !aligned16(de))
    return MI_ERR_UNSUPPORTED;
  MI_LAUNCH("bn_mish_bwd", k_bn_mish_bwd, tile_grid(M, N), kBlock, stream, g, m_act, z, ldz, bias, mu, al, bz, de, dz, part, M, N);
  return launch_status();
}

int mi_bn_dz(const float *dy, const float *z, int32_t ldz, const float *mu, const float *al, const float *bz, const float *de,
             float *dz, int32_t M, int32_t N, void *stream) {
  if (M <= 0 || N <= 0) return MI_ERR_INVALID_ARG;
  if (!dy || !z || !mu || !al || !bz || !de || !dz) return MI_ERR_INVALID_ARG;
  if (N % 4 || ldz % 4 || ldz < N || !aligned16(dy) || !aligned16(z) || !aligned16(dz) || !aligned16(mu) || !aligned16(al) ||
      !aligned16(bz) || !aligned16(de))
    return MI_ERR_UNSUPPORTED;
  const int64_t n4 = (int64_t)M * N / 4;
  int64_t grid = (n4 + kBlock - 1) / kBlock;
  if (grid > kMaxGrid) grid = kMaxGrid;
  MI_LAUNCH("bn_dz", k_bn_dz, (int)grid, kBlock, stream, dy, z, ldz, mu, al, bz, de, dz, n4, N);
  return launch_status();
}

}  // extern "C"
