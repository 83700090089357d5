// optim.hip — fused row-sparse optimizer steps on row-form (COO, possibly uncoalesced) gradients.
// SURVEY.md §8f rank 1; reference: get_optimizers' sparse branch (src/models/deepfm.py:163-184:
// torch.optim.SparseAdam on model.embedding.parameters(), no weight decay), and the SGD branch.
//
// torch.optim.SparseAdam semantics (torch/optim/sparse_adam.py, _single_tensor_sparse_adam):
// coalesce the gradient (duplicates SUMMED), then for the touched rows only
//     m += (1-b1)(g - m);  v += (1-b2)(g*g - v);  p -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
// Here the caller sorts the row ids once (torch.sort: no host sync, no dynamic shape); every
// sorted position whose predecessor holds a different row is a segment head: it sums the
// gradient rows of its segment and applies the update — one pass, traffic ~ B*F rows, not N.
#include "common.hpp"

namespace {
using namespace mi;

struct AdamArgs {
  const int64_t *rows;   // sorted
  const int64_t *perm;   // sorted position -> original entry
  const float *vals;     // [n, D] original order
  float *W, *M, *V;      // [N, D]
  int64_t n, N;
  float step_size, beta1, beta2, eps;
};

template <int LPR>
__global__ __launch_bounds__(kBlock) void k_sparse_adam(AdamArgs a) {
  constexpr int RS = kWave / LPR;
  constexpr int D = LPR * 4;
  const int lane = threadIdx.x & 63;
  const int q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ntiles = (a.n + RS - 1) / RS;
  for (int64_t t = wave0; t < ntiles; t += nwaves) {
    const int64_t i = t * RS + r;
    if (i >= a.n) continue;
    const int64_t row = a.rows[i];
    if ((uint64_t)row >= (uint64_t)a.N) continue;
    if (i > 0 && a.rows[i - 1] == row) continue;   // not a segment head
    float4 g = ld4(a.vals + a.perm[i] * D + q * 4);
    for (int64_t j = i + 1; j < a.n && a.rows[j] == row; ++j) {
      const float4 t4 = ld4(a.vals + a.perm[j] * D + q * 4);
      g.x += t4.x; g.y += t4.y; g.z += t4.z; g.w += t4.w;
    }
    const int64_t o = row * D + q * 4;
    float4 m = ld4(a.M + o), v = ld4(a.V + o), w = ld4(a.W + o);
    const float c1 = 1.f - a.beta1, c2 = 1.f - a.beta2;
    m.x += c1 * (g.x - m.x); m.y += c1 * (g.y - m.y); m.z += c1 * (g.z - m.z); m.w += c1 * (g.w - m.w);
    v.x += c2 * (g.x * g.x - v.x); v.y += c2 * (g.y * g.y - v.y);
    v.z += c2 * (g.z * g.z - v.z); v.w += c2 * (g.w * g.w - v.w);
    w.x -= a.step_size * (m.x / (sqrtf(v.x) + a.eps));
    w.y -= a.step_size * (m.y / (sqrtf(v.y) + a.eps));
    w.z -= a.step_size * (m.z / (sqrtf(v.z) + a.eps));
    w.w -= a.step_size * (m.w / (sqrtf(v.w) + a.eps));
    st4(a.M + o, m);
    st4(a.V + o, v);
    st4(a.W + o, w);
  }
}

// any D: one wave per sorted position (heads only do work), lanes stride the row
__global__ __launch_bounds__(kBlock) void k_sparse_adam_anyD(AdamArgs a, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t i = wave0; i < a.n; i += nwaves) {
    const int64_t row = a.rows[i];
    if ((uint64_t)row >= (uint64_t)a.N) continue;
    if (i > 0 && a.rows[i - 1] == row) continue;
    for (int d = lane; d < D; d += kWave) {
      float g = a.vals[a.perm[i] * D + d];
      for (int64_t j = i + 1; j < a.n && a.rows[j] == row; ++j) g += a.vals[a.perm[j] * D + d];
      const int64_t o = row * D + d;
      float m = a.M[o], v = a.V[o];
      m += (1.f - a.beta1) * (g - m);
      v += (1.f - a.beta2) * (g * g - v);
      a.M[o] = m;
      a.V[o] = v;
      a.W[o] -= a.step_size * (m / (sqrtf(v) + a.eps));
    }
  }
}

// W[idx[i], :] += alpha * g[i, :]   (row-sparse SGD: linear, so duplicates need no coalescing)
__global__ __launch_bounds__(kBlock) void k_scatter_axpy(const int64_t *__restrict__ idx, const float *__restrict__ g,
                                                         float alpha, float *__restrict__ W, int64_t n, int D,
                                                         int64_t N) {
  const int64_t total = n * D;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx[e / D];
    if ((uint64_t)row < (uint64_t)N) atomicAdd(W + row * D + e % D, alpha * g[e]);
  }
}

inline bool vec_ok(int D) { return D >= 4 && D <= 256 && (D & 3) == 0 && ((D >> 2) & ((D >> 2) - 1)) == 0; }
}  // namespace

extern "C" {

int mi_sparse_adam_sorted(const int64_t *rows_sorted, const int64_t *perm, const float *vals, float *W,
                          float *exp_avg, float *exp_avg_sq, int64_t n, int32_t D, int64_t N, float step_size,
                          float beta1, float beta2, float eps, void *stream) {
  if (n < 0 || D <= 0 || N < 0) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!rows_sorted || !perm || !vals || !W || !exp_avg || !exp_avg_sq) return MI_ERR_INVALID_ARG;
  AdamArgs a{rows_sorted, perm, vals, W, exp_avg, exp_avg_sq, n, N, step_size, beta1, beta2, eps};
  if (vec_ok(D) && aligned16(vals) && aligned16(W) && aligned16(exp_avg) && aligned16(exp_avg_sq)) {
    const int lpr = D / 4;
    const int grid = grid_for_waves((n + (kWave / lpr) - 1) / (kWave / lpr));
#define CALL(LPR) MI_LAUNCH("sparse_adam", (k_sparse_adam<LPR>), grid, kBlock, stream, a)
    switch (lpr) {
      case 1: CALL(1); break;
      case 2: CALL(2); break;
      case 4: CALL(4); break;
      case 8: CALL(8); break;
      case 16: CALL(16); break;
      case 32: CALL(32); break;
      case 64: CALL(64); break;
      default: return MI_ERR_UNSUPPORTED;
    }
#undef CALL
  } else {
    MI_LAUNCH("sparse_adam", k_sparse_adam_anyD, grid_for_waves(n), kBlock, stream, a, D);
  }
  return launch_status();
}

int mi_scatter_axpy_rows(const int64_t *idx, const float *g, float alpha, float *W, int64_t n, int32_t D, int64_t N,
                         void *stream) {
  if (n < 0 || D <= 0 || N < 0) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!idx || !g || !W) return MI_ERR_INVALID_ARG;
  int64_t gsz = (n * D + kBlock - 1) / kBlock;
  if (gsz > kMaxGrid) gsz = kMaxGrid;
  MI_LAUNCH("scatter_axpy_rows", k_scatter_axpy, (int)gsz, kBlock, stream, idx, g, alpha, W, n, D, N);
  return launch_status();
}

}  // extern "C"
