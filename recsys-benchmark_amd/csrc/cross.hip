// cross.hip — the memory-bound elementwise / reduction pieces of the CrossNet backward
// (src/models/layer_dcn.py:90-140 differentiated by hand); the contractions are in gemm.hip.
#include <cstdlib>

#include "common.hpp"

namespace {
using namespace mi;

// dlin = g * x0 ;  dx0 (+)= g * lin          (DCNHead / DCN_MixHead:  x_{l+1} = x_l + x_0 * lin)
__global__ __launch_bounds__(kBlock) void k_cross_bwd_pre(const float *__restrict__ g, const float *__restrict__ x0,
                                                          const float *__restrict__ lin, float *__restrict__ dlin,
                                                          float *__restrict__ dx0, int64_t n, int accumulate) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float gv = g[e];
    dlin[e] = gv * x0[e];
    const float t = gv * lin[e];
    dx0[e] = accumulate ? dx0[e] + t : t;
  }
}

// out[n] += sum_m X[m,n] * rs(m),  rs(m) = sum_{e<nrs} rowscale[m*nrs+e] (1 if null).  out caller-zeroed.
// rs_sum (nullable, caller-zeroed): += sum_m rs(m) — the bias gradient of a 1-output Linear, whose weight
// gradient is this very column sum (dW = g^T x, db = sum g): one launch for both.
__global__ __launch_bounds__(kBlock) void k_colsum(const float *__restrict__ X, int ldx, const float *__restrict__ rowscale,
                                                   int nrs, float *__restrict__ out, int M, int N,
                                                   float *__restrict__ rs_sum) {
  __shared__ float part[4][64];
  __shared__ float part_rs[4];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  const bool tally = rs_sum && blockIdx.x == 0 && c == 0;   // column 0's lanes see every row of their row block
  float s = 0.f, t = 0.f;
  if (n < N) {
    const int stride = gridDim.y * 4;
    int m = blockIdx.y * 4 + rl;
    if (!rowscale || nrs == 1) {
      for (; m + 7 * stride < M; m += 8 * stride) {   // 8 independent row loads in flight
        float v[8], rs[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v[u] = X[(int64_t)(m + u * stride) * ldx + n];
          rs[u] = rowscale ? rowscale[m + u * stride] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          s += v[u] * rs[u];
          t += rs[u];
        }
      }
    }
    for (; m < M; m += stride) {
      float rs = 1.f;
      if (rowscale) {
        rs = 0.f;
        for (int e = 0; e < nrs; ++e) rs += rowscale[(int64_t)m * nrs + e];
      }
      s += X[(int64_t)m * ldx + n] * rs;
      t += rs;
    }
  }
  part[rl][c] = s;
  if (tally) part_rs[rl] = t;
  __syncthreads();
  if (rl == 0 && n < N) atomicAdd(out + n, part[0][c] + part[1][c] + part[2][c] + part[3][c]);
  if (tally && rl == 0) atomicAdd(rs_sum, part_rs[0] + part_rs[1] + part_rs[2] + part_rs[3]);
}

// out[m] = sum_n X[m,n] * v[n] (+ bias[0]) (+ addend[m])
__global__ __launch_bounds__(kBlock) void k_rowdot(const float *__restrict__ X, int ldx, const float *__restrict__ v,
                                                   const float *__restrict__ bias, const float *__restrict__ addend,
                                                   float *__restrict__ out, int M, int N) {
  const int lane = threadIdx.x & 63;
  const int wave0 = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * kWavesPerBlock;
  for (int m = wave0; m < M; m += nwaves) {
    float s = 0.f;
    for (int n = lane; n < N; n += kWave) s += X[(int64_t)m * ldx + n] * v[n];
    s = wave_sum(s);
    if (lane == 0) out[m] = s + (bias ? bias[0] : 0.f) + (addend ? addend[m] : 0.f);
  }
}

// out[m,e] = sum_n X[m,n] * W[e,n], e < E <= 8: the DCN_MixHead gate (g_e = x_l . G_e) — E dot products per row in one pass
// over the row instead of a GEMM launch with a 4-column output.  One wave per row, float4 columns.
template <int E>
__global__ __launch_bounds__(kBlock) void k_rowdot_multi(const float *__restrict__ X, int ldx, const float *__restrict__ W,
                                                         float *__restrict__ out, int M, int N) {
  const int lane = threadIdx.x & 63;
  const int wave0 = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * kWavesPerBlock;
  for (int m = wave0; m < M; m += nwaves) {
    float s[E];
#pragma unroll
    for (int e = 0; e < E; ++e) s[e] = 0.f;
    for (int n = lane * 4; n < N; n += kWave * 4) {
      const float4 x = ld4(X + (int64_t)m * ldx + n);
#pragma unroll
      for (int e = 0; e < E; ++e) s[e] += dot4(x, ld4(W + (int64_t)e * N + n));
    }
#pragma unroll
    for (int e = 0; e < E; ++e) s[e] = wave_sum(s[e]);
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < E; ++e) out[(int64_t)m * E + e] = s[e];
    }
  }
}

// The head of a CrossNet layer's backward in ONE pass over the row (x_{l+1} = x_l + x_0 * lin, lin = ... + b * rs):
//   dlin[m,:] = g[m,:] * x0[m,:]            dx0[m,:] (+)= g[m,:] * lin[m,:]
//   db[n]    += sum_m dlin[m,n] * rs(m)     rs(m) = sum_e gate[m,e]  (1 without a gate: DCNHead)
//   dgs[m]    = sum_n dlin[m,n] * b[n]      (DCN_MixHead only: the bias' share of the gate gradient)
// One wave per row, NJ float4 column groups per lane (N <= 1024 * NJ / 4 ... see the launcher); the column sums stay in
// registers over the wave's rows, meet in LDS and leave as one float atomic per column and workgroup.
template <int NJ>
__global__ __launch_bounds__(kBlock) void k_cross_bwd_head(const float *__restrict__ g, const float *__restrict__ x0,
                                                           const float *__restrict__ lin, const float *__restrict__ gate,
                                                           int E, const float *__restrict__ b, float *__restrict__ dlin,
                                                           float *__restrict__ dx0, int accumulate, float *__restrict__ db,
                                                           float *__restrict__ dgs, int M, int N) {
  __shared__ float4 part[kWavesPerBlock][NJ * kWave];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wave0 = blockIdx.x * kWavesPerBlock + w;
  const int nwaves = gridDim.x * kWavesPerBlock;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 col[NJ], bv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    col[j] = zero;
    const int n = (j * kWave + lane) * 4;
    bv[j] = (b && n < N) ? ld4(b + n) : zero;
  }
  constexpr int R = 2;                       // rows in flight per wave: 2 x NJ x (3..4) float4 loads before the first use
  for (int m0 = wave0; m0 < M; m0 += R * nwaves) {
    float4 gv[R][NJ], xv[R][NJ], lv[R][NJ], dv[R][NJ];
    float gt[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int m = m0 + u * nwaves;
      const int64_t row = (int64_t)(m < M ? m : m0) * N;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n = (j * kWave + lane) * 4;
        const int64_t o = row + (n < N ? n : 0);
        gv[u][j] = ld4(g + o); xv[u][j] = ld4(x0 + o); lv[u][j] = ld4(lin + o);
        dv[u][j] = accumulate ? ld4(dx0 + o) : zero;
      }
      gt[u] = (gate && lane < E) ? gate[(int64_t)(m < M ? m : m0) * E + lane] : 0.f;   // lane e holds gate[m,e]
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int m = m0 + u * nwaves;
      if (m >= M) continue;                  // wave-uniform
      const float rs = gate ? wave_sum(gt[u]) : 1.f;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n = (j * kWave + lane) * 4;
        if (n >= N) continue;
        const int64_t o = (int64_t)m * N + n;
        const float4 gg = gv[u][j], xx = xv[u][j], ll = lv[u][j], dd = dv[u][j];
        const float4 d = make_float4(gg.x * xx.x, gg.y * xx.y, gg.z * xx.z, gg.w * xx.w);
        st4(dlin + o, d);
        st4(dx0 + o, make_float4(dd.x + gg.x * ll.x, dd.y + gg.y * ll.y, dd.z + gg.z * ll.z, dd.w + gg.w * ll.w));
        col[j].x += d.x * rs; col[j].y += d.y * rs; col[j].z += d.z * rs; col[j].w += d.w * rs;
        s += dot4(d, bv[j]);
      }
      if (dgs) {
        s = wave_sum(s);
        if (lane == 0) dgs[m] = s;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) part[w][j * kWave + lane] = col[j];
  __syncthreads();
  const float *q = reinterpret_cast<const float *>(&part[0][0]);
  for (int n = threadIdx.x; n < N; n += kBlock) {
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < kWavesPerBlock; ++u) t += q[u * NJ * kWave * 4 + n];
    atomicAdd(db + n, t);
  }
}

// out[m,n] = g[m] * w[n]   (backward of a 1-output Linear w.r.t. its input)
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_outer(const float *__restrict__ g, const float *__restrict__ w,
                                                  float *__restrict__ out, int M, int N) {
  const int nv = N / VEC;
  const int64_t total = (int64_t)M * nv;
  for (int64_t ev = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ev < total;
       ev += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(ev / nv), n0 = (int)(ev % nv) * VEC;
    const float gm = g[m];
    if constexpr (VEC == 4) {
      const float4 wv = ld4(w + n0);
      st4(out + (int64_t)m * N + n0, make_float4(gm * wv.x, gm * wv.y, gm * wv.z, gm * wv.w));
    } else {
      out[(int64_t)m * N + n0] = gm * w[n0];
    }
  }
}

// DCN_MixHead gate/tanh backward for one layer, one wave per row m:
//   dgate[m,e] = sum_k dH2g[m,e*r+k] * H2[m,e*r+k] + dgsum[m]
//   dZ2[m,e*r+k] = dH2g[m,e*r+k] * gate[m,e] * (1 - H2^2)
__global__ __launch_bounds__(kBlock) void k_mix_gate_bwd(const float *__restrict__ dH2g, const float *__restrict__ H2,
                                                         const float *__restrict__ gate, const float *__restrict__ dgsum,
                                                         float *__restrict__ dgate, float *__restrict__ dZ2, int M, int E,
                                                         int r) {
  const int lane = threadIdx.x & 63;
  const int wave0 = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * kWavesPerBlock;
  const int ld = E * r;
  for (int m = wave0; m < M; m += nwaves) {
    const float dgs = dgsum[m];
    for (int e = 0; e < E; ++e) {
      const float gt = gate[(int64_t)m * E + e];
      float s = 0.f;
      for (int k = lane; k < r; k += kWave) {
        const int64_t o = (int64_t)m * ld + e * r + k;
        const float dh = dH2g[o], h = H2[o];
        s += dh * h;
        dZ2[o] = dh * gt * (1.f - h * h);
      }
      s = wave_sum(s);
      if (lane == 0) dgate[(int64_t)m * E + e] = s + dgs;
    }
  }
}

// mean_i [ max(x,0) - x*y + log1p(exp(-|x|)) ]  (torch's numerically stable binary_cross_entropy_with_logits,
// reduction="mean"); ONE workgroup so the reduction needs no atomics and no zeroed target.
// dx_unit (optional): the gradient for an upstream gradient of exactly 1, (sigmoid(x) - y)/n in the backward kernel's own
// arithmetic — when the criterion is the last op of the step the backward pass then has nothing left to launch
__global__ __launch_bounds__(1024) void k_bce_logits_fwd(const float *__restrict__ x, const float *__restrict__ y,
                                                         float *__restrict__ loss, float *__restrict__ dx_unit,
                                                         int64_t n) {
  __shared__ float part[16];
  float s = 0.f;
  const float sc = 1.f / (float)n;
  // four elements per thread and trip, all eight loads issued before the first use: hipcc does not pipeline the
  // one-element loop, which at B = 4096 was four dependent round trips (7.2 us for a kernel that reads 32 KB)
  constexpr int U = 4;
  for (int64_t i0 = threadIdx.x; i0 < n; i0 += (int64_t)U * blockDim.x) {
    float xv[U], yv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * blockDim.x;
      xv[u] = i < n ? x[i] : 0.f;
      yv[u] = i < n ? y[i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * blockDim.x;
      if (i < n) {
        s += fmaxf(xv[u], 0.f) - xv[u] * yv[u] + log1pf(expf(-fabsf(xv[u])));
        if (dx_unit) dx_unit[i] = sc * (1.f / (1.f + expf(-xv[u])) - yv[u]);
      }
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w];
    loss[0] = t / (float)n;
  }
}

// dx = g[0] * (sigmoid(x) - y) / n
__global__ __launch_bounds__(kBlock) void k_bce_logits_bwd(const float *__restrict__ x, const float *__restrict__ y,
                                                           const float *__restrict__ g, float *__restrict__ dx,
                                                           int64_t n) {
  const float sc = g[0] / (float)n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dx[i] = sc * (1.f / (1.f + expf(-x[i])) - y[i]);
}

inline int grid_for_elems(int64_t total) {
  int64_t g = (total + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  if (g > kMaxGrid) g = kMaxGrid;
  return (int)g;
}
}  // namespace

extern "C" {

int mi_cross_bwd_pre(const float *g, const float *x0, const float *lin, float *dlin, float *dx0, int64_t n,
                     int32_t accumulate, void *stream) {
  if (n < 0) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!g || !x0 || !lin || !dlin || !dx0) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("cross_bwd_pre", k_cross_bwd_pre, grid_for_elems(n), kBlock, stream, g, x0, lin, dlin, dx0, n, accumulate);
  return launch_status();
}

int mi_colsum(const float *X, int32_t ldx, const float *rowscale, int32_t nrs, float *out, float *rs_sum, int32_t M,
              int32_t N, void *stream) {
  if (M < 0 || N < 0) return MI_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return MI_OK;
  if (!X || !out) return MI_ERR_INVALID_ARG;
  const int cb = (N + 63) / 64;
  int rb = (M + 31) / 32;
  const int want = (1024 + cb - 1) / cb;
  if (rb > want) rb = want;
  if (rb < 1) rb = 1;
  dim3 grid(cb, rb);
  MI_LAUNCH("colsum", k_colsum, grid, kBlock, stream, X, ldx, rowscale, nrs, out, M, N, rs_sum);
  return launch_status();
}

int mi_rowdot(const float *X, int32_t ldx, const float *v, const float *bias, const float *addend, float *out,
              int32_t M, int32_t N, void *stream) {
  if (M < 0 || N < 0) return MI_ERR_INVALID_ARG;
  if (M == 0) return MI_OK;
  if (!X || !v || !out) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("rowdot", k_rowdot, grid_for_waves(M), kBlock, stream, X, ldx, v, bias, addend, out, M, N);
  return launch_status();
}

int mi_rowdot_multi(const float *X, int32_t ldx, const float *W, float *out, int32_t M, int32_t N, int32_t E, void *stream) {
  if (M < 0 || N < 0 || E < 1 || E > 8) return MI_ERR_INVALID_ARG;
  if (M == 0) return MI_OK;
  if (!X || !W || !out) return MI_ERR_INVALID_ARG;
  if (N % 4 != 0 || ldx % 4 != 0 || !aligned16(X) || !aligned16(W)) return MI_ERR_UNSUPPORTED;
#define MI_RDM(EE) case EE: MI_LAUNCH("rowdot_multi", k_rowdot_multi<EE>, grid_for_waves(M), kBlock, stream, X, ldx, W, out, M, N); break
  switch (E) { MI_RDM(1); MI_RDM(2); MI_RDM(3); MI_RDM(4); MI_RDM(5); MI_RDM(6); MI_RDM(7); MI_RDM(8); }
#undef MI_RDM
  return launch_status();
}

int mi_cross_bwd_head(const float *g, const float *x0, const float *lin, const float *gate, int32_t E, const float *b,
                      float *dlin, float *dx0, int32_t accumulate, float *db, float *dgs, int32_t M, int32_t N,
                      void *stream) {
  if (M < 0 || N < 0 || (gate && (E < 1 || E > kWave))) return MI_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return MI_OK;
  if (!g || !x0 || !lin || !dlin || !dx0 || !db || (dgs && !b)) return MI_ERR_INVALID_ARG;
  if (N % 4 != 0 || N > 1024 || !aligned16(g) || !aligned16(x0) || !aligned16(lin) || !aligned16(dlin) || !aligned16(dx0) ||
      (b && !aligned16(b)))
    return MI_ERR_UNSUPPORTED;
  // ~4 rows per wave, two in flight: the column partials amortise over them.  Every workgroup ends with one float atomic per
  // column on the SAME N words (same-address atomics serialise at ~26 ns apiece: 256 workgroups = 6.7 us behind the last
  // row), so the grid is capped (MI_CROSS_BWD_GRID, default 192: profiles/r04_ab_runs.txt)
  static const int cap = [] { const char *e = getenv("MI_CROSS_BWD_GRID"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 2048 ? v : 192; }();
  int grid = (M + 15) / 16;
  if (grid > cap) grid = cap;
  if (grid < 1) grid = 1;
  const int nj = (N / 4 + kWave - 1) / kWave;
#define MI_CBH(J) MI_LAUNCH("cross_bwd_head", k_cross_bwd_head<J>, grid, kBlock, stream, g, x0, lin, gate, E, b, dlin, dx0, accumulate, db, dgs, M, N)
  if (nj <= 1) MI_CBH(1);
  else if (nj == 2) MI_CBH(2);
  else if (nj == 3) MI_CBH(3);
  else MI_CBH(4);
#undef MI_CBH
  return launch_status();
}

int mi_bce_logits_fwd(const float *x, const float *y, float *loss, float *dx_unit, int64_t n, void *stream) {
  if (n <= 0 || !x || !y || !loss) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("bce_logits_fwd", k_bce_logits_fwd, 1, 1024, stream, x, y, loss, dx_unit, n);
  return launch_status();
}

int mi_bce_logits_bwd(const float *x, const float *y, const float *g, float *dx, int64_t n, void *stream) {
  if (n <= 0 || !x || !y || !g || !dx) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("bce_logits_bwd", k_bce_logits_bwd, grid_for_elems(n), kBlock, stream, x, y, g, dx, n);
  return launch_status();
}

int mi_outer(const float *g, const float *w, float *out, int32_t M, int32_t N, void *stream) {
  if (M < 0 || N < 0) return MI_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return MI_OK;
  if (!g || !w || !out) return MI_ERR_INVALID_ARG;
  if (N % 4 == 0 && aligned16(w) && aligned16(out))
    MI_LAUNCH("outer", k_outer<4>, grid_for_elems((int64_t)M * N / 4), kBlock, stream, g, w, out, M, N);
  else
    MI_LAUNCH("outer", k_outer<1>, grid_for_elems((int64_t)M * N), kBlock, stream, g, w, out, M, N);
  return launch_status();
}

int mi_mix_gate_bwd(const float *dH2g, const float *H2, const float *gate, const float *dgsum, float *dgate,
                    float *dZ2, int32_t M, int32_t E, int32_t r, void *stream) {
  if (M < 0 || E <= 0 || r <= 0) return MI_ERR_INVALID_ARG;
  if (M == 0) return MI_OK;
  if (!dH2g || !H2 || !gate || !dgsum || !dgate || !dZ2) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("mix_gate_bwd", k_mix_gate_bwd, grid_for_waves(M), kBlock, stream, dH2g, H2, gate, dgsum, dgate, dZ2, M, E, r);
  return launch_status();
}

}  // extern "C"
