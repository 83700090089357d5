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
  float step_size, omb1, omb2, eps;   // omb = 1 - beta, formed in double by the host (1.f - 0.999f is off by 1.3e-5)
  const float *step_size_dev;   // when given: the step size lives on the device (mi_adam_tick), hipGraph replays
  float *G;                     // when given: no update — the coalesced gradient row is STORED to G[row] (mi_coalesce_rows_sorted)
  int64_t ldw;                  // floats between consecutive rows of W (D, or the row stride of a packed table); M, V, G: D
};

// t += 1; step_size = lr * sqrt(1 - b2^t) / (1 - b1^t) in double, like the host computes it in the eager path
constexpr int kTickSlots = 8;
struct TickTable {
  float *step[kTickSlots];
  float *step_size[kTickSlots];
  int count;
};
__global__ void k_adam_tick(TickTable tt, double lr, double beta1, double beta2) {
  const int i = threadIdx.x;
  if (i >= tt.count) return;
  const double t = (double)tt.step[i][0] + 1.0;
  tt.step[i][0] = (float)t;
  tt.step_size[i][0] = (float)(lr * sqrt(1.0 - pow(beta2, t)) / (1.0 - pow(beta1, t)));
}

__device__ __forceinline__ float adam1(const AdamArgs &a, float g, float &m, float &v) {
  m += a.omb1 * (g - m);
  v += a.omb2 * (g * g - v);
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

__device__ __forceinline__ void adam_row(const AdamArgs &a, int64_t o, int64_t ow, const Vec<4> &g) {
  if (a.G) { st4(a.G + o, g.v); return; }
  float4 m = ld4(a.M + o), v = ld4(a.V + o), w = ld4(a.W + ow);
  w.x -= adam1(a, g.v.x, m.x, v.x);
  w.y -= adam1(a, g.v.y, m.y, v.y);
  w.z -= adam1(a, g.v.z, m.z, v.z);
  w.w -= adam1(a, g.v.w, m.w, v.w);
  st4(a.M + o, m);
  st4(a.V + o, v);
  st4(a.W + ow, w);
}
__device__ __forceinline__ void adam_row(const AdamArgs &a, int64_t o, int64_t ow, const Vec<1> &g) {
  if (a.G) { a.G[o] = g.v; return; }
  float m = a.M[o], v = a.V[o];
  a.W[ow] -= adam1(a, g.v, m, v);
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
    if (!before && !after) adam_row(a, row * D + q * VW, row * a.ldw + q * VW, g);
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
      if (r == 0) adam_row(a, hrow * D + q * VW, hrow * a.ldw + q * VW, sum);
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
      if (a.G) { a.G[o] = g; continue; }
      float m = a.M[o], v = a.V[o];
      a.W[row * a.ldw + d] -= adam1(a, g, m, v);
      a.M[o] = m;
      a.V[o] = v;
    }
  }
}

// ---- dense Adam over a list of tensors (torch.optim.Adam, L2 weight decay) --------------------------------------
// One launch updates up to kAdamTensors tensors: workgroup c owns one 4096-element chunk of one tensor (chunk counts are
// prefix-summed in the table), every thread four float4 groups with all loads issued before the arithmetic — 7 streams
// (p, g, m, v in; p, m, v out), HBM-bound for the big tables and one launch instead of a 40 us multi_tensor_apply for the
// MLP's 0.5 M parameters.  The step count is a device float per tensor (torch's capturable layout): this kernel reads
// t = step + 1, k_adam_steps_inc advances the counters afterwards.
constexpr int kAdamTensors = 24;
constexpr int kAdamChunk = kBlock * 16;
struct AdamDenseTable {
  float *p[kAdamTensors];
  const float *g[kAdamTensors];
  float *m[kAdamTensors];
  float *v[kAdamTensors];
  float *step[kAdamTensors];
  int64_t numel[kAdamTensors];
  int64_t chunk_end[kAdamTensors];   // prefix sum of ceil(numel / kAdamChunk)
  int32_t count;
  uint32_t vec_ok;                   // bit k: all four pointers of tensor k are 16-byte aligned
  unsigned *ticket;                  // when given: the last workgroup to finish advances the step counts (no second launch)
};

__device__ __forceinline__ void adam_dense1(float &p, float g, float &m, float &v, float omb1, float b2, float omb2, float wd,
                                            float step_size, float inv_bc2s, float eps) {
  g += wd * p;
  m += omb1 * (g - m);                             // m.lerp_(g, 1 - b1)
  v = b2 * v + omb2 * g * g;
  p -= step_size * (m / (sqrtf(v) * inv_bc2s + eps));
}

__global__ __launch_bounds__(kBlock) void k_adam_dense(AdamDenseTable t, float lr, double beta1, double beta2, float eps,
                                                       float wd) {
  __shared__ float s_step_size, s_inv_bc2s;
  int k = 0;
  while (k + 1 < t.count && (int64_t)blockIdx.x >= t.chunk_end[k]) ++k;
  const int64_t chunk = (int64_t)blockIdx.x - (k ? t.chunk_end[k - 1] : 0);
  if (threadIdx.x == 0) {
    const double tt = (double)t.step[k][0] + 1.0;
    s_step_size = (float)((double)lr / (1.0 - pow(beta1, tt)));
    s_inv_bc2s = (float)(1.0 / sqrt(1.0 - pow(beta2, tt)));
  }
  __syncthreads();
  const float step_size = s_step_size, inv_bc2s = s_inv_bc2s;
  const float omb1 = (float)(1.0 - beta1), b2 = (float)beta2, omb2 = (float)(1.0 - beta2);
  float *P = t.p[k], *M = t.m[k], *V = t.v[k];
  const float *G = t.g[k];
  const int64_t n = t.numel[k], base = chunk * kAdamChunk;
  if (((t.vec_ok >> k) & 1u) && base + kAdamChunk <= n) {
    float4 p[4], g[4], m[4], v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t e = base + (int64_t)(u * kBlock + threadIdx.x) * 4;
      p[u] = ld4(P + e); g[u] = ld4(G + e); m[u] = ld4(M + e); v[u] = ld4(V + e);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      adam_dense1(p[u].x, g[u].x, m[u].x, v[u].x, omb1, b2, omb2, wd, step_size, inv_bc2s, eps);
      adam_dense1(p[u].y, g[u].y, m[u].y, v[u].y, omb1, b2, omb2, wd, step_size, inv_bc2s, eps);
      adam_dense1(p[u].z, g[u].z, m[u].z, v[u].z, omb1, b2, omb2, wd, step_size, inv_bc2s, eps);
      adam_dense1(p[u].w, g[u].w, m[u].w, v[u].w, omb1, b2, omb2, wd, step_size, inv_bc2s, eps);
      const int64_t e = base + (int64_t)(u * kBlock + threadIdx.x) * 4;
      st4(P + e, p[u]); st4(M + e, m[u]); st4(V + e, v[u]);
    }
  } else {                                         // a tensor's last chunk, or unaligned views
    for (int64_t e = base + threadIdx.x; e < n && e < base + kAdamChunk; e += kBlock) {
      float p = P[e], m = M[e], v = V[e];
      adam_dense1(p, G[e], m, v, omb1, b2, omb2, wd, step_size, inv_bc2s, eps);
      P[e] = p; M[e] = m; V[e] = v;
    }
  }
  if (t.ticket) {
    // every workgroup read its tensor's step count before doing anything else, so once all of them have finished the
    // counts may advance: the last one to take a ticket does it (and re-arms the ticket for the next launch)
    __syncthreads();
    if (threadIdx.x == 0 &&
        __hip_atomic_fetch_add(t.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
      for (int i = 0; i < t.count; ++i) t.step[i][0] += 1.f;
      __hip_atomic_store(t.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ void k_adam_steps_inc(AdamDenseTable t) {
  if ((int)threadIdx.x < t.count) t.step[threadIdx.x][0] += 1.f;
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

static int launch_sorted_rows(AdamArgs a, int32_t D, const float *vals, const void *p0, const void *p1, const void *p2,
                              float *acc, void *stream);

int mi_sparse_adam_sorted_ld(const int64_t *rows_sorted, const int64_t *perm, const float *vals, float *W, int64_t ldw,
                             float *exp_avg, float *exp_avg_sq, float *acc, int64_t n, int32_t D, int64_t N,
                             float step_size, const float *step_size_dev, double beta1, double beta2, float eps,
                             void *stream) {
  if (n < 0 || D <= 0 || N < 0 || ldw < D) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!rows_sorted || !perm || !vals || !W || !exp_avg || !exp_avg_sq || !acc) return MI_ERR_INVALID_ARG;
  if (vec_ok(D) && (ldw & 3) != 0) return MI_ERR_UNSUPPORTED;   // float4 row accesses need 16-byte aligned rows
  AdamArgs a{rows_sorted, perm, vals, W, exp_avg, exp_avg_sq, acc, n, N, step_size, (float)(1.0 - beta1), (float)(1.0 - beta2),
             eps, step_size_dev, nullptr, ldw};
  return launch_sorted_rows(a, D, vals, W, exp_avg, exp_avg_sq, acc, stream);
}

int mi_sparse_adam_sorted(const int64_t *rows_sorted, const int64_t *perm, const float *vals, float *W,
                          float *exp_avg, float *exp_avg_sq, float *acc, int64_t n, int32_t D, int64_t N,
                          float step_size, const float *step_size_dev, double beta1, double beta2, float eps,
                          void *stream) {
  return mi_sparse_adam_sorted_ld(rows_sorted, perm, vals, W, D, exp_avg, exp_avg_sq, acc, n, D, N, step_size, step_size_dev,
                                  beta1, beta2, eps, stream);
}

// The same segmented sums, stored instead of applied: G[row, :] = sum of vals[perm[i], :] over the sorted positions i with
// rows_sorted[i] == row, added in a FIXED order (no atomics) — a bit-reproducible dense gradient out of row-form values.
// Rows that do not occur are not written: the caller zero-fills G.
int mi_coalesce_rows_sorted(const int64_t *rows_sorted, const int64_t *perm, const float *vals, float *G, float *acc,
                            int64_t n, int32_t D, int64_t N, void *stream) {
  if (n < 0 || D <= 0 || N < 0) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!rows_sorted || !perm || !vals || !G || !acc) return MI_ERR_INVALID_ARG;
  AdamArgs a{rows_sorted, perm, vals, nullptr, nullptr, nullptr, acc, n, N, 0.f, 0.f, 0.f, 0.f, nullptr, G, D};
  return launch_sorted_rows(a, D, vals, G, G, G, acc, stream);
}

static int launch_sorted_rows(AdamArgs a, int32_t D, const float *vals, const void *W, const void *exp_avg,
                              const void *exp_avg_sq, float *acc, void *stream) {
  const int64_t n = a.n;
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
  TickTable tt{};
  tt.step[0] = step; tt.step_size[0] = step_size; tt.count = 1;
  MI_LAUNCH("adam_tick", k_adam_tick, 1, kWave, stream, tt, lr, beta1, beta2);
  return launch_status();
}

int mi_adam_tick_multi(float *const *steps, float *const *step_sizes, int32_t count, double lr, double beta1, double beta2,
                       void *stream) {
  if (count < 0 || (count > 0 && (!steps || !step_sizes))) return MI_ERR_INVALID_ARG;
  for (int32_t first = 0; first < count; first += kTickSlots) {
    TickTable tt{};
    for (int32_t i = first; i < count && tt.count < kTickSlots; ++i) {
      if (!steps[i] || !step_sizes[i]) return MI_ERR_INVALID_ARG;
      tt.step[tt.count] = steps[i];
      tt.step_size[tt.count++] = step_sizes[i];
    }
    MI_LAUNCH("adam_tick", k_adam_tick, 1, kWave, stream, tt, lr, beta1, beta2);
  }
  return launch_status();
}

int mi_adam_dense_multi(float *const *params, const float *const *grads, float *const *exp_avgs,
                        float *const *exp_avg_sqs, float *const *steps, const int64_t *numels, int32_t count, float lr,
                        double beta1, double beta2, float eps, float weight_decay, uint32_t *tickets, void *stream) {
  if (count < 0) return MI_ERR_INVALID_ARG;
  if (count == 0) return MI_OK;
  if (!params || !grads || !exp_avgs || !exp_avg_sqs || !steps || !numels) return MI_ERR_INVALID_ARG;
  int launch = 0;
  for (int32_t first = 0; first < count; first += kAdamTensors) {
    AdamDenseTable t{};
    int64_t chunks = 0;
    for (int32_t i = first; i < count && t.count < kAdamTensors; ++i) {
      if (numels[i] < 0) return MI_ERR_INVALID_ARG;
      if (numels[i] == 0) continue;
      if (!params[i] || !grads[i] || !exp_avgs[i] || !exp_avg_sqs[i] || !steps[i]) return MI_ERR_INVALID_ARG;
      const int k = t.count++;
      t.p[k] = params[i]; t.g[k] = grads[i]; t.m[k] = exp_avgs[i]; t.v[k] = exp_avg_sqs[i]; t.step[k] = steps[i];
      t.numel[k] = numels[i];
      chunks += (numels[i] + kAdamChunk - 1) / kAdamChunk;
      t.chunk_end[k] = chunks;
      if (aligned16(params[i]) && aligned16(grads[i]) && aligned16(exp_avgs[i]) && aligned16(exp_avg_sqs[i]))
        t.vec_ok |= 1u << k;
    }
    if (t.count == 0) continue;
    if (chunks > 0x7fffffff) return MI_ERR_UNSUPPORTED;
    t.ticket = tickets ? tickets + launch++ : nullptr;
    MI_LAUNCH("adam_dense", k_adam_dense, (int)chunks, kBlock, stream, t, lr, beta1, beta2, eps, weight_decay);
    if (!tickets) MI_LAUNCH("adam_steps_inc", k_adam_steps_inc, 1, kWave, stream, t);
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
