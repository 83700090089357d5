// infonce.hip — the contrastive loss of the LightGCN step.
//
// Reference: src/losses.py:25-47 (info_nce), called with both views equal to the batch's distinct user and
// positive-item rows (src/trainer/lightgcn.py:215-229):
//     v = F.normalize(v, dim=1)  (b_cos)      -> k_rownorm_fwd / k_rownorm_bwd, one wave per row
//     S = v1 v2^T / T                          -> mi_gemm_f32 (the host launches it; a real GEMM)
//     loss = -mean_i log_softmax(S)[i, i]      -> k_lse_diag_fwd: one workgroup per row, online max/sum-exp,
//                                                 the last workgroup sums the per-row terms in index order
// backward: dS = g/n/T * (softmax(S) - I) written over S by k_lse_diag_bwd; dv1 = dS v2, dv2 = dS^T v1 are GEMMs.
#include "common.hpp"

namespace {
using namespace mi;

__global__ __launch_bounds__(kBlock) void k_rownorm_fwd(const float *__restrict__ X, int64_t n, int D, float eps,
                                                        float *__restrict__ Y, float *__restrict__ inv) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t r = wave0; r < n; r += nwaves) {
    const float *x = X + r * D;
    float ss = 0.f;
    for (int j = lane; j < D; j += kWave) ss += x[j] * x[j];
    const float iv = 1.f / fmaxf(sqrtf(wave_sum(ss)), eps);
    for (int j = lane; j < D; j += kWave) Y[r * D + j] = x[j] * iv;
    if (lane == 0) inv[r] = iv;
  }
}

// y = x * iv, iv = 1/|x|  =>  dx = iv * (dy - y <y, dy>);  a row clamped at eps (iv == 1/eps) is y = x/eps: dx = dy/eps
__global__ __launch_bounds__(kBlock) void k_rownorm_bwd(const float *__restrict__ Y, const float *__restrict__ inv,
                                                        const float *__restrict__ dY, int64_t n, int D, float inv_eps,
                                                        float *__restrict__ dX) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t r = wave0; r < n; r += nwaves) {
    const float *y = Y + r * D, *dy = dY + r * D;
    float dot = 0.f;
    for (int j = lane; j < D; j += kWave) dot += y[j] * dy[j];
    dot = wave_sum(dot);
    const float iv = inv[r];
    if (iv >= inv_eps) dot = 0.f;
    for (int j = lane; j < D; j += kWave) dX[r * D + j] = iv * (dy[j] - y[j] * dot);
  }
}

// running (max, sum of exp(v - max)) pair and its merge
struct MaxSum { float mx, sum; };
__device__ __forceinline__ void ms_push(MaxSum &a, float v) {
  if (v == -INFINITY) return;                      // a masked column
  if (v > a.mx) { a.sum = a.sum * expf(a.mx - v) + 1.f; a.mx = v; }
  else a.sum += expf(v - a.mx);
}
__device__ __forceinline__ MaxSum ms_merge(MaxSum a, MaxSum b) {
  if (b.mx > a.mx) { const MaxSum t = a; a = b; b = t; }
  if (b.mx > -INFINITY) a.sum += b.sum * expf(b.mx - a.mx);
  return a;
}

// valid (nullable): rows and columns with valid[i] == 0 do not exist — the fixed-shape stand-in for torch.unique: the
// batch's rows as they come, duplicates masked out, count[0] = number of valid rows
__global__ __launch_bounds__(kBlock) void k_lse_diag_fwd(const float *__restrict__ S, int64_t ld, int n, float inv_t,
                                                         const uint8_t *__restrict__ valid, const float *__restrict__ count,
                                                         float *__restrict__ lse, float *__restrict__ part,
                                                         unsigned *ticket, float *__restrict__ loss) {
  __shared__ MaxSum red[kWavesPerBlock];
  __shared__ float redf[kWavesPerBlock];
  __shared__ bool last;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float term = 0.f;                            // thread 0: this workgroup's sum of (lse_i - s_ii)
  // a few hundred workgroups walk the rows: one ticket per workgroup, not per row (same-address atomics serialise)
  for (int64_t row = blockIdx.x; row < n; row += gridDim.x) {
    if (valid && !valid[row]) {                  // workgroup-uniform
      if (threadIdx.x == 0) lse[row] = 0.f;
      continue;
    }
    const float *s = S + row * ld;
    MaxSum a{-INFINITY, 0.f};
    if ((ld & 3) == 0 && aligned16(S)) {
      // four independent float4 loads in flight per thread, then one rescale for the 16 values
      const int n4 = n >> 2;
      for (int j0 = threadIdx.x; j0 < n4; j0 += 4 * kBlock) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = j0 + u * kBlock;
          v[u] = j < n4 ? ld4(s + 4 * j) : float4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
          if (valid && j < n4) {
            const uchar4 ok = *reinterpret_cast<const uchar4 *>(valid + 4 * j);
            if (!ok.x) v[u].x = -INFINITY;
            if (!ok.y) v[u].y = -INFINITY;
            if (!ok.z) v[u].z = -INFINITY;
            if (!ok.w) v[u].w = -INFINITY;
          }
        }
        float mx = a.mx;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v[u].x *= inv_t; v[u].y *= inv_t; v[u].z *= inv_t; v[u].w *= inv_t;
          mx = fmaxf(fmaxf(fmaxf(mx, v[u].x), fmaxf(v[u].y, v[u].z)), v[u].w);
        }
        if (mx == -INFINITY) continue;              // nothing valid so far
        float sum = a.mx > -INFINITY ? a.sum * expf(a.mx - mx) : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          sum += expf(v[u].x - mx) + expf(v[u].y - mx) + expf(v[u].z - mx) + expf(v[u].w - mx);
        a = MaxSum{mx, sum};
      }
      for (int j = (n4 << 2) + threadIdx.x; j < n; j += kBlock) ms_push(a, (!valid || valid[j]) ? s[j] * inv_t : -INFINITY);
    } else {
      for (int j = threadIdx.x; j < n; j += kBlock) ms_push(a, (!valid || valid[j]) ? s[j] * inv_t : -INFINITY);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      MaxSum b{__shfl_xor(a.mx, m), __shfl_xor(a.sum, m)};
      a = ms_merge(a, b);
    }
    if (lane == 0) red[wv] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
      MaxSum t = red[0];
      for (int j = 1; j < kWavesPerBlock; ++j) t = ms_merge(t, red[j]);
      const float l = t.mx + logf(t.sum);
      lse[row] = l;
      term += l - s[row] * inv_t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) publish_partial(part, ticket, term, last);
  __syncthreads();
  if (last) {                                  // fixed summation order over the rows: deterministic
    float t = 0.f;
    for (unsigned j = threadIdx.x; j < gridDim.x; j += kBlock) t += read_partial(part + j);
    t = wave_sum(t);
    if (lane == 0) redf[wv] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
      float u = 0.f;
      for (int j = 0; j < kWavesPerBlock; ++j) u += redf[j];
      loss[0] = u / (count ? count[0] : (float)n);
      *ticket = 0;
    }
  }
}

// S[i, j] <- g/n/T * (exp(S[i, j]/T - lse_i) - [i == j])
__global__ __launch_bounds__(kBlock) void k_lse_diag_bwd(float *__restrict__ S, int64_t ld, int n, float inv_t,
                                                         const uint8_t *__restrict__ valid, const float *__restrict__ count,
                                                         const float *__restrict__ lse, const float *__restrict__ g,
                                                         int symmetric) {
  const float c = g[0] * inv_t / (count ? count[0] : (float)n);
  const int64_t total = (int64_t)n * n;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
    const int i = (int)(e / n), j = (int)(e - (int64_t)i * n);
    float *p = S + i * ld + j;
    const bool live = !valid || (valid[i] && valid[j]);
    // symmetric (both views are one matrix, S = S^T): dS + dS^T in one pass, so that dv = (dS + dS^T) v is ONE product
    const float s = *p * inv_t, d = i == j ? 1.f : 0.f;
    float val = expf(s - lse[i]) - d;
    if (symmetric) val += expf(s - lse[j]) - d;
    *p = live ? c * val : 0.f;
  }
}
}  // namespace

extern "C" {

int mi_rownorm_fwd(const float *X, int64_t n, int32_t D, float eps, float *Y, float *inv, void *stream) {
  if (n < 0 || D <= 0 || !(eps > 0.f)) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!X || !Y || !inv) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("rownorm_fwd", k_rownorm_fwd, grid_for_waves(n), kBlock, stream, X, n, D, eps, Y, inv);
  return launch_status();
}

int mi_rownorm_bwd(const float *Y, const float *inv, const float *dY, int64_t n, int32_t D, float eps, float *dX,
                   void *stream) {
  if (n < 0 || D <= 0 || !(eps > 0.f)) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!Y || !inv || !dY || !dX) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("rownorm_bwd", k_rownorm_bwd, grid_for_waves(n), kBlock, stream, Y, inv, dY, n, D, 1.f / eps, dX);
  return launch_status();
}

constexpr int kLseGrid = 1024;
int64_t mi_lse_diag_workspace_elems(int32_t) { return kLseGrid + 1; }

int mi_lse_diag_fwd(const float *S, int64_t ld, int32_t n, float inv_t, const uint8_t *valid, const float *count, float *lse,
                    float *workspace, float *loss, void *stream) {
  if (n <= 0 || ld < n) return MI_ERR_INVALID_ARG;
  if (!S || !lse || !workspace || !loss || (valid != nullptr) != (count != nullptr)) return MI_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(valid) & 3u) return MI_ERR_INVALID_ARG;       // read four flags at a time
  const int grid = n < kLseGrid ? n : kLseGrid;
  if (hipMemsetAsync(workspace + kLseGrid, 0, sizeof(unsigned), (hipStream_t)stream) != hipSuccess) return MI_ERR_LAUNCH;
  MI_LAUNCH("lse_diag_fwd", k_lse_diag_fwd, grid, kBlock, stream, S, ld, n, inv_t, valid, count, lse, workspace,
            reinterpret_cast<unsigned *>(workspace + kLseGrid), loss);
  return launch_status();
}

int mi_lse_diag_bwd(float *S, int64_t ld, int32_t n, float inv_t, const uint8_t *valid, const float *count, const float *lse,
                    const float *g, int32_t symmetric, void *stream) {
  if (n <= 0 || ld < n) return MI_ERR_INVALID_ARG;
  if (!S || !lse || !g || (valid != nullptr) != (count != nullptr)) return MI_ERR_INVALID_ARG;
  int64_t grid = ((int64_t)n * n + kBlock - 1) / kBlock;
  if (grid > kMaxGrid) grid = kMaxGrid;
  MI_LAUNCH("lse_diag_bwd", k_lse_diag_bwd, (int)grid, kBlock, stream, S, ld, n, inv_t, valid, count, lse, g, symmetric);
  return launch_status();
}

}  // extern "C"
