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
  float *acc;            // [n, D] scratch: per-pass pieces of the segments longer than one pass
  int64_t n, N;
  float step_size, beta1, beta2, eps;
  const float *step_size_dev;   // when given: the step size lives on the device (mi_adam_tick), hipGraph replays
};

// t += 1; step_size = lr * sqrt(1 - b2^t) / (1 - b1^t) in double, like the host computes it in the eager path
__global__ void k_adam_tick(float *step, float *step_size, double lr, double beta1, double beta2) {
  const double t = (double)step[0] + 1.0;
  step[0] = (float)t;
  step_size[0] = (float)(lr * sqrt(1.0 - pow(beta2, t)) / (1.0 - pow(beta1, t)));
}

__device__ __forceinline__ float adam1(const AdamArgs &a, float g, float &m, float &v) {
  m += (1.f - a.beta1) * (g - m);
  v += (1.f - a.beta2) * (g * g - v);
  return a.step_size * (m / (sqrtf(v) + a.eps));
}
__device__ __forceinline__ void resolve_step_size(AdamArgs &a) {
  if (a.step_size_dev) a.step_size = a.step_size_dev[0];
}

// VW consecutive floats of a row held by one lane: float4 (VW = 4) or a scalar (VW = 1: the [N,1] first-order table)
template <int VW> struct Vec;
template <> struct Vec<4> {
  float4 v;
  static __device__ __forceinline__ Vec load(const float *p) { return {ld4(p)}; }
  __device__ __forceinline__ void store(float *p) const { st4(p, v); }
  __device__ __forceinline__ void add(const Vec &o) { v.x += o.v.x; v.y += o.v.y; v.z += o.v.z; v.w += o.v.w; }
  __device__ __forceinline__ Vec down(int d) const {
    return {float4{__shfl_down(v.x, d), __shfl_down(v.y, d), __shfl_down(v.z, d), __shfl_down(v.w, d)}};
  }
  __device__ __forceinline__ Vec across(int m) const {
    return {float4{__shfl_xor(v.x, m), __shfl_xor(v.y, m), __shfl_xor(v.z, m), __shfl_xor(v.w, m)}};
  }
  static __device__ __forceinline__ Vec zero() { return {float4{0.f, 0.f, 0.f, 0.f}}; }
};
template <> struct Vec<1> {
  float v;
  static __device__ __forceinline__ Vec load(const float *p) { return {*p}; }
  __device__ __forceinline__ void store(float *p) const { *p = v; }
  __device__ __forceinline__ void add(const Vec &o) { v += o.v; }
  __device__ __forceinline__ Vec down(int d) const { return {__shfl_down(v, d)}; }
  __device__ __forceinline__ Vec across(int m) const { return {__shfl_xor(v, m)}; }
  static __device__ __forceinline__ Vec zero() { return {0.f}; }
};

__device__ __forceinline__ void adam_row(const AdamArgs &a, int64_t o, const Vec<4> &g) {
  float4 m = ld4(a.M + o), v = ld4(a.V + o), w = ld4(a.W + o);
  w.x -= adam1(a, g.v.x, m.x, v.x);
  w.y -= adam1(a, g.v.y, m.y, v.y);
  w.z -= adam1(a, g.v.z, m.z, v.z);
  w.w -= adam1(a, g.v.w, m.w, v.w);
  st4(a.M + o, m);
  st4(a.V + o, v);
  st4(a.W + o, w);
}
__device__ __forceinline__ void adam_row(const AdamArgs &a, int64_t o, const Vec<1> &g) {
  float m = a.M[o], v = a.V[o];
  a.W[o] -= adam1(a, g.v, m, v);
  a.M[o] = m;
  a.V[o] = v;
}

// A pass = RS consecutive sorted positions, LPR lanes (VW floats each, D = LPR*VW) per position.  Every position loads
// its gradient row (no serial walk over duplicates); a segmented suffix sum over the pass (log2 RS shuffle steps: the
// keys are sorted, so "position i+s has my row" means everything between has it too) leaves each segment's in-pass
// total with its first position in the pass.  A segment that lies inside one pass is applied at once; one that spans
// passes leaves a piece per pass in acc[that first position] (plain stores: every slot has one writer) and
// k_sparse_adam_long adds the pieces in a fixed order — no atomics, so the step is deterministic.
template <int LPR, int VW>
__global__ __launch_bounds__(kBlock) void k_sparse_adam(AdamArgs a) {
  resolve_step_size(a);
  constexpr int RS = kWave / LPR;
  constexpr int D = LPR * VW;
  const int lane = threadIdx.x & 63;
  const int q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ntiles = (a.n + RS - 1) / RS;
  for (int64_t t = wave0; t < ntiles; t += nwaves) {
    const int64_t i = t * RS + r;
    const bool live = i < a.n;
    const int64_t row = live ? a.rows[i] : (int64_t)-1 - r;          // dead positions: keys nobody shares
    Vec<VW> g = live ? Vec<VW>::load(a.vals + a.perm[i] * D + q * VW) : Vec<VW>::zero();
#pragma unroll
    for (int s = 1; s < RS; s <<= 1) {
      const int64_t nrow = __shfl_down(row, s * LPR);
      const Vec<VW> ng = g.down(s * LPR);
      if (r + s < RS && nrow == row) g.add(ng);
    }
    if (!live || (uint64_t)row >= (uint64_t)a.N) continue;
    const bool before = i > 0 && a.rows[i - 1] == row;
    if (r > 0 && before) continue;                                   // not the first of its segment in this pass
    const int64_t pe = t * RS + RS;                                  // first position after the pass
    const bool after = pe < a.n && a.rows[pe] == row;
    if (!before && !after) adam_row(a, row * D + q * VW, g);
    else g.store(a.acc + i * D + q * VW);
  }
}

// second kernel of the pair: the wave that holds the first position of a segment running past its pass gathers the
// pieces — acc[first position], then acc[start of each following pass] while that pass still begins with the row —
// its RS lane groups striding over them, joined by a fixed shuffle tree.
template <int LPR, int VW>
__global__ __launch_bounds__(kBlock) void k_sparse_adam_long(AdamArgs a) {
  resolve_step_size(a);
  constexpr int RS = kWave / LPR;
  constexpr int D = LPR * VW;
  const int lane = threadIdx.x & 63;
  const int q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ntiles = (a.n + RS - 1) / RS;
  for (int64_t t = wave0; t < ntiles; t += nwaves) {
    const int64_t i = t * RS + r, pe = t * RS + RS;
    if (pe >= a.n) break;                                            // the last pass has nothing after it (wave-uniform)
    const int64_t row = a.rows[i];
    const bool head = (uint64_t)row < (uint64_t)a.N && a.rows[pe] == row && !(i > 0 && a.rows[i - 1] == row);
    uint64_t todo = __ballot(head && q == 0);
    while (todo) {
      const int src = __ffsll((unsigned long long)todo) - 1;
      todo &= todo - 1;
      const int64_t hrow = __shfl(row, src), hi = __shfl(i, src);
      Vec<VW> sum = Vec<VW>::zero();
      for (int64_t k = r;; k += RS) {
        const int64_t pos = k == 0 ? hi : pe + (k - 1) * RS;
        if (pos >= a.n || a.rows[pos] != hrow) break;
        sum.add(Vec<VW>::load(a.acc + pos * D + q * VW));
      }
#pragma unroll
      for (int m = LPR; m < kWave; m <<= 1) sum.add(sum.across(m));
      if (r == 0) adam_row(a, hrow * D + q * VW, sum);
    }
  }
}

// any D: one wave per sorted position, lanes stride the row; the first position of a row walks its duplicates
__global__ __launch_bounds__(kBlock) void k_sparse_adam_anyD(AdamArgs a, int D) {
  resolve_step_size(a);
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
      a.W[o] -= adam1(a, g, m, v);
      a.M[o] = m;
      a.V[o] = v;
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
                          float step_size, const float *step_size_dev, float beta1, float beta2, float eps,
                          void *stream) {
  if (n < 0 || D <= 0 || N < 0) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!rows_sorted || !perm || !vals || !W || !exp_avg || !exp_avg_sq || !acc) return MI_ERR_INVALID_ARG;
  AdamArgs a{rows_sorted, perm, vals, W, exp_avg, exp_avg_sq, acc, n, N, step_size, beta1, beta2, eps, step_size_dev};
#define CALL(LPR, VW)                                                                        \
  do {                                                                                       \
    const int grid = grid_for_waves((n + (kWave / LPR) - 1) / (kWave / LPR));                \
    MI_LAUNCH("sparse_adam", (k_sparse_adam<LPR, VW>), grid, kBlock, stream, a);             \
    MI_LAUNCH("sparse_adam_long", (k_sparse_adam_long<LPR, VW>), grid, kBlock, stream, a);   \
  } while (0)
  if (D == 1) {
    CALL(1, 1);
  } else if (D == 2) {
    CALL(2, 1);
  } else if (vec_ok(D) && aligned16(vals) && aligned16(W) && aligned16(exp_avg) && aligned16(exp_avg_sq) && aligned16(acc)) {
    switch (D / 4) {
      case 1: CALL(1, 4); break;
      case 2: CALL(2, 4); break;
      case 4: CALL(4, 4); break;
      case 8: CALL(8, 4); break;
      case 16: CALL(16, 4); break;
      case 32: CALL(32, 4); break;
      case 64: CALL(64, 4); break;
      default: return MI_ERR_UNSUPPORTED;
    }
#undef CALL
  } else {
    MI_LAUNCH("sparse_adam", k_sparse_adam_anyD, grid_for_waves(n), kBlock, stream, a, D);
  }
  return launch_status();
}

int mi_adam_tick(float *step, float *step_size, double lr, double beta1, double beta2, void *stream) {
  if (!step || !step_size) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("adam_tick", k_adam_tick, 1, 1, stream, step, step_size, lr, beta1, beta2);
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
