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
  float *acc;            // [n, D] zero on entry and on exit: sums of segments longer than one pass
  int64_t n, N;
  float step_size, beta1, beta2, eps;
};

__device__ __forceinline__ float adam1(const AdamArgs &a, float g, float &m, float &v) {
  m += (1.f - a.beta1) * (g - m);
  v += (1.f - a.beta2) * (g * g - v);
  return a.step_size * (m / (sqrtf(v) + a.eps));
}

__device__ __forceinline__ void adam_row4(const AdamArgs &a, int64_t o, const float4 &g) {
  float4 m = ld4(a.M + o), v = ld4(a.V + o), w = ld4(a.W + o);
  w.x -= adam1(a, g.x, m.x, v.x);
  w.y -= adam1(a, g.y, m.y, v.y);
  w.z -= adam1(a, g.z, m.z, v.z);
  w.w -= adam1(a, g.w, m.w, v.w);
  st4(a.M + o, m);
  st4(a.V + o, v);
  st4(a.W + o, w);
}

// first sorted position holding `row` (it is known to occur before `hi`)
__device__ __forceinline__ int64_t first_of(const int64_t *rows, int64_t hi, int64_t row) {
  int64_t lo = 0;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (rows[mid] < row) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// A pass = RS consecutive sorted positions, LPR lanes (one float4 each) per position.  Every position loads its
// gradient row (no serial walk over duplicates); a segmented suffix sum over the pass (log2 RS shuffle steps: the
// keys are sorted, so "position i+s has my row" means everything between has it too) leaves each segment's in-pass
// total with its first position.  A segment that lies inside one pass is applied at once; the pieces of one that
// spans passes are added into acc[first position of the row] and applied by k_sparse_adam_long.
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
    const bool live = i < a.n;
    const int64_t row = live ? a.rows[i] : (int64_t)-1 - r;          // dead positions: keys nobody shares
    float4 g = live ? ld4(a.vals + a.perm[i] * D + q * 4) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 1; s < RS; s <<= 1) {
      const int64_t nrow = __shfl_down(row, s * LPR);
      const float nx = __shfl_down(g.x, s * LPR), ny = __shfl_down(g.y, s * LPR);
      const float nz = __shfl_down(g.z, s * LPR), nw = __shfl_down(g.w, s * LPR);
      if (r + s < RS && nrow == row) { g.x += nx; g.y += ny; g.z += nz; g.w += nw; }
    }
    if (!live || (uint64_t)row >= (uint64_t)a.N) continue;
    const bool before = i > 0 && a.rows[i - 1] == row;
    if (r > 0 && before) continue;                                   // not the first of its segment in this pass
    const int64_t pe = t * RS + RS;                                  // first position after the pass
    const bool after = pe < a.n && a.rows[pe] == row;
    if (!before && !after) {
      adam_row4(a, row * D + q * 4, g);
    } else {
      const int64_t h = before ? first_of(a.rows, i, row) : i;
      float *dst = a.acc + h * D + q * 4;
      atomicAdd(dst, g.x); atomicAdd(dst + 1, g.y); atomicAdd(dst + 2, g.z); atomicAdd(dst + 3, g.w);
    }
  }
}

// second kernel of the pair: the first position of every segment that runs past its pass owns acc[position]
template <int LPR>
__global__ __launch_bounds__(kBlock) void k_sparse_adam_long(AdamArgs a) {
  constexpr int RS = kWave / LPR;
  constexpr int D = LPR * 4;
  const int lane = threadIdx.x & 63;
  const int q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ntiles = (a.n + RS - 1) / RS;
  for (int64_t t = wave0; t < ntiles; t += nwaves) {
    const int64_t i = t * RS + r, pe = t * RS + RS;
    if (i >= a.n || pe >= a.n) continue;
    const int64_t row = a.rows[i];
    if ((uint64_t)row >= (uint64_t)a.N || a.rows[pe] != row) continue;
    if (i > 0 && a.rows[i - 1] == row) continue;
    float *src = a.acc + i * D + q * 4;
    const float4 g = ld4(src);
    st4(src, float4{0.f, 0.f, 0.f, 0.f});
    adam_row4(a, row * D + q * 4, g);
  }
}

// any D: one wave per sorted position, lanes stride the row.  A row that occurs once is applied at once; the
// entries of a repeated row are added into acc[first position of the row] and applied by the second kernel.
template <bool LONG>
__global__ __launch_bounds__(kBlock) void k_sparse_adam_anyD(AdamArgs a, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t i = wave0; i < a.n; i += nwaves) {
    const int64_t row = a.rows[i];
    if ((uint64_t)row >= (uint64_t)a.N) continue;
    const bool before = i > 0 && a.rows[i - 1] == row;
    const bool after = i + 1 < a.n && a.rows[i + 1] == row;
    if (LONG) {
      if (before || !after) continue;
      for (int d = lane; d < D; d += kWave) {
        const float g = a.acc[i * D + d];
        a.acc[i * D + d] = 0.f;
        const int64_t o = row * D + d;
        float m = a.M[o], v = a.V[o];
        a.W[o] -= adam1(a, g, m, v);
        a.M[o] = m;
        a.V[o] = v;
      }
    } else if (!before && !after) {
      for (int d = lane; d < D; d += kWave) {
        const int64_t o = row * D + d;
        float m = a.M[o], v = a.V[o];
        a.W[o] -= adam1(a, a.vals[a.perm[i] * D + d], m, v);
        a.M[o] = m;
        a.V[o] = v;
      }
    } else {
      const int64_t h = before ? first_of(a.rows, i, row) : i;
      for (int d = lane; d < D; d += kWave) atomicAdd(a.acc + h * D + d, a.vals[a.perm[i] * D + d]);
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
                          float *exp_avg, float *exp_avg_sq, float *acc, int64_t n, int32_t D, int64_t N,
                          float step_size, float beta1, float beta2, float eps, void *stream) {
  if (n < 0 || D <= 0 || N < 0) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!rows_sorted || !perm || !vals || !W || !exp_avg || !exp_avg_sq || !acc) return MI_ERR_INVALID_ARG;
  AdamArgs a{rows_sorted, perm, vals, W, exp_avg, exp_avg_sq, acc, n, N, step_size, beta1, beta2, eps};
  if (vec_ok(D) && aligned16(vals) && aligned16(W) && aligned16(exp_avg) && aligned16(exp_avg_sq) && aligned16(acc)) {
    const int lpr = D / 4;
    const int grid = grid_for_waves((n + (kWave / lpr) - 1) / (kWave / lpr));
#define CALL(LPR)                                                                     \
  do {                                                                                \
    MI_LAUNCH("sparse_adam", (k_sparse_adam<LPR>), grid, kBlock, stream, a);          \
    MI_LAUNCH("sparse_adam_long", (k_sparse_adam_long<LPR>), grid, kBlock, stream, a); \
  } while (0)
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
    MI_LAUNCH("sparse_adam", (k_sparse_adam_anyD<false>), grid_for_waves(n), kBlock, stream, a, D);
    MI_LAUNCH("sparse_adam_long", (k_sparse_adam_anyD<true>), grid_for_waves(n), kBlock, stream, a, D);
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
