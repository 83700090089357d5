// tt.hip — TT-Rec lookup (TTRecTorch semantics): mixed-radix split of the id over tt_p_shapes,
// gather of one slice per core, chained small contractions.
// Reference: src/models/embeddings/tensortrain_embeddings.py:100-150 (reshape_cores with permute
// [1,0,2,3], _core_dot_prod, tt_rec_torch_forward).  Core c is stored [1, p_c, r_c*q_c*r_{c+1}]
// and viewed (p_c, r_c, q_c, r_{c+1}), so the slice of index i_c is r_c*q_c*r_{c+1} contiguous floats.
//
//   res_0[q_0, r_1]            = core_0[i_0]
//   res_c[(h, qq), r']         = sum_j res_{c-1}[h, j] * core_c[i_c][j, qq, r']        c = 1..n-1
//   out[(h_0 .. h_{n-1})]      = res_{n-1}[:, 0]
//
// One workgroup per lookup; every level lives in LDS (a few hundred floats at the reference's
// ranks [128, 96]); the core slices stream from L2 (the middle core is ~4 MB in total).  The
// backward recomputes the levels, then walks the chain in reverse: slice gradients are
// float-atomic adds into the dense core gradients, level gradients stay in LDS.
// First correct version: a grouped (sort-by-digit) MFMA formulation is the known next step.
#include "common.hpp"

namespace {
using namespace mi;

constexpr int kMaxCores = 4;

struct TTDesc {
  int ncores;
  int p[kMaxCores], q[kMaxCores], r[kMaxCores + 1];
  const float *core[kMaxCores];
  float *gcore[kMaxCores];
  int lvl_off[kMaxCores];  // offset of level c in LDS (floats)
  int lvl_total;           // sum of level sizes
  int lvl_max;             // largest level
};

__device__ __forceinline__ bool tt_digits(int64_t id, const TTDesc &t, int64_t N, int *dig) {
  if (id < 0 || id >= N) return false;
  int64_t big = 1;
  for (int c = 0; c < t.ncores; ++c) big *= t.p[c];
  for (int c = 0; c < t.ncores; ++c) {
    big /= t.p[c];
    dig[c] = (int)(id / big);
    id = id % big;
  }
  return true;
}

// levels[c] for c in [0, ncores) into L (LDS); returns nothing (caller syncs)
__device__ __forceinline__ void tt_forward_levels(const TTDesc &t, const int *dig, float *L) {
  const int tid = threadIdx.x, nt = blockDim.x;
  {
    const int sz = t.q[0] * t.r[1];
    const float *S = t.core[0] + (int64_t)dig[0] * sz;
    for (int e = tid; e < sz; e += nt) L[t.lvl_off[0] + e] = S[e];
  }
  __syncthreads();
  int H = t.q[0];
  for (int c = 1; c < t.ncores; ++c) {
    const int rc = t.r[c], QR = t.q[c] * t.r[c + 1];
    const float *S = t.core[c] + (int64_t)dig[c] * rc * QR;
    const float *prev = L + t.lvl_off[c - 1];
    float *cur = L + t.lvl_off[c];
    const int total = H * QR;
    for (int e = tid; e < total; e += nt) {
      const int h = e / QR, rem = e % QR;
      float a = 0.f;
      for (int j = 0; j < rc; ++j) a += prev[h * rc + j] * S[(int64_t)j * QR + rem];
      cur[e] = a;
    }
    __syncthreads();
    H *= t.q[c];
  }
}

__global__ __launch_bounds__(kBlock) void k_tt_fwd(const int64_t *__restrict__ idx, TTDesc t, float *__restrict__ out,
                                                   int64_t n, int D, int64_t N, int *err) {
  extern __shared__ float smem[];
  int dig[kMaxCores];
  int bad = 0;
  for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
    const bool ok = tt_digits(idx[i], t, N, dig);
    if (!ok) {
      bad = 1;
      for (int e = threadIdx.x; e < D; e += blockDim.x) out[i * D + e] = 0.f;
      continue;
    }
    tt_forward_levels(t, dig, smem);
    const float *last = smem + t.lvl_off[t.ncores - 1];
    for (int e = threadIdx.x; e < D; e += blockDim.x) out[i * D + e] = last[e];
    __syncthreads();
  }
  if (bad && err && threadIdx.x == 0) atomicOr(err, MI_IDX_OUT_OF_RANGE);
}

__global__ __launch_bounds__(kBlock) void k_tt_bwd(const int64_t *__restrict__ idx, TTDesc t,
                                                   const float *__restrict__ g, int64_t n, int D, int64_t N) {
  extern __shared__ float smem[];
  float *dA = smem + t.lvl_total;       // gradient of the current level
  float *dB = dA + t.lvl_max;           // gradient of the previous level
  int dig[kMaxCores];
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
    if (!tt_digits(idx[i], t, N, dig)) continue;
    tt_forward_levels(t, dig, smem);
    for (int e = tid; e < D; e += nt) dA[e] = g[i * D + e];
    __syncthreads();
    int H = 1;
    for (int c = 0; c < t.ncores - 1; ++c) H *= t.q[c];   // H_{c-1} for c = ncores-1
    for (int c = t.ncores - 1; c >= 1; --c) {
      const int rc = t.r[c], QR = t.q[c] * t.r[c + 1];
      const float *S = t.core[c] + (int64_t)dig[c] * rc * QR;
      float *gS = t.gcore[c] + (int64_t)dig[c] * rc * QR;
      const float *prev = smem + t.lvl_off[c - 1];
      // d core_c[i_c][j, rem] += sum_h prev[h, j] * dA[h*QR + rem]
      for (int e = tid; e < rc * QR; e += nt) {
        const int j = e / QR, rem = e % QR;
        float a = 0.f;
        for (int h = 0; h < H; ++h) a += prev[h * rc + j] * dA[h * QR + rem];
        atomicAdd(gS + e, a);
      }
      // d prev[h, j] = sum_rem dA[h*QR + rem] * S[j, rem]
      for (int e = tid; e < H * rc; e += nt) {
        const int h = e / rc, j = e % rc;
        float a = 0.f;
        for (int rem = 0; rem < QR; ++rem) a += dA[h * QR + rem] * S[(int64_t)j * QR + rem];
        dB[e] = a;
      }
      __syncthreads();
      float *tmp = dA; dA = dB; dB = tmp;
      H /= t.q[c - 1];
    }
    {
      const int sz = t.q[0] * t.r[1];
      float *gS = t.gcore[0] + (int64_t)dig[0] * sz;
      for (int e = tid; e < sz; e += nt) atomicAdd(gS + e, dA[e]);
    }
    __syncthreads();
    dA = smem + t.lvl_total;
    dB = dA + t.lvl_max;
  }
}

int fill_desc(TTDesc &t, int32_t ncores, const int32_t *p, const int32_t *q, const int32_t *r,
              const float *const *cores, float *const *gcores, int32_t D) {
  if (ncores < 2 || ncores > kMaxCores || !p || !q || !r || !cores) return MI_ERR_INVALID_ARG;
  t.ncores = ncores;
  long long H = 1;
  t.lvl_total = 0;
  t.lvl_max = 0;
  for (int c = 0; c < ncores; ++c) {
    if (p[c] <= 0 || q[c] <= 0 || r[c] <= 0 || r[c + 1] <= 0 || !cores[c]) return MI_ERR_INVALID_ARG;
    t.p[c] = p[c]; t.q[c] = q[c]; t.r[c] = r[c];
    t.core[c] = cores[c];
    t.gcore[c] = gcores ? gcores[c] : nullptr;
    H *= q[c];
    const long long sz = H * r[c + 1];
    if (sz > 16384) return MI_ERR_UNSUPPORTED;
    t.lvl_off[c] = t.lvl_total;
    t.lvl_total += (int)sz;
    if (sz > t.lvl_max) t.lvl_max = (int)sz;
  }
  t.r[ncores] = r[ncores];
  if (r[0] != 1 || r[ncores] != 1 || H != D) return MI_ERR_INVALID_ARG;
  return MI_OK;
}

}  // namespace

extern "C" {

int mi_tt_fwd(const int64_t *idx, const float *const *cores /*host array of device pointers*/, int32_t ncores,
              const int32_t *p_shapes, const int32_t *q_shapes, const int32_t *ranks /*host, ncores+1*/, float *out,
              int64_t n, int32_t D, int64_t N, int32_t *err, void *stream) {
  if (n < 0 || D <= 0) return MI_ERR_INVALID_ARG;
  TTDesc t;
  int rc = fill_desc(t, ncores, p_shapes, q_shapes, ranks, cores, nullptr, D);
  if (rc != MI_OK) return rc;
  if (n == 0) return MI_OK;
  if (!idx || !out) return MI_ERR_INVALID_ARG;
  const size_t shmem = (size_t)t.lvl_total * sizeof(float);
  if (shmem > 160 * 1024) return MI_ERR_UNSUPPORTED;
  const int grid = (int)(n < 4096 ? n : 4096);
  hipEvent_t ea, eb;
  if (mi::prof_acquire("tt_fwd", &ea, &eb))
    hipExtLaunchKernelGGL(k_tt_fwd, dim3(grid), dim3(kBlock), shmem, (hipStream_t)stream, ea, eb, 0, idx, t, out, n, D, N, err);
  else
    hipLaunchKernelGGL(k_tt_fwd, dim3(grid), dim3(kBlock), shmem, (hipStream_t)stream, idx, t, out, n, D, N, err);
  return launch_status();
}

int mi_tt_bwd(const int64_t *idx, const float *g_out, const float *const *cores, float *const *gcores,
              int32_t ncores, const int32_t *p_shapes, const int32_t *q_shapes, const int32_t *ranks, int64_t n,
              int32_t D, int64_t N, void *stream) {
  if (n < 0 || D <= 0 || !gcores) return MI_ERR_INVALID_ARG;
  TTDesc t;
  int rc = fill_desc(t, ncores, p_shapes, q_shapes, ranks, cores, gcores, D);
  if (rc != MI_OK) return rc;
  for (int c = 0; c < ncores; ++c)
    if (!gcores[c]) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!idx || !g_out) return MI_ERR_INVALID_ARG;
  const size_t shmem = (size_t)(t.lvl_total + 2 * t.lvl_max) * sizeof(float);
  if (shmem > 160 * 1024) return MI_ERR_UNSUPPORTED;
  const int grid = (int)(n < 4096 ? n : 4096);
  hipEvent_t ea, eb;
  if (mi::prof_acquire("tt_bwd", &ea, &eb))
    hipExtLaunchKernelGGL(k_tt_bwd, dim3(grid), dim3(kBlock), shmem, (hipStream_t)stream, ea, eb, 0, idx, t, g_out, n, D, N);
  else
    hipLaunchKernelGGL(k_tt_bwd, dim3(grid), dim3(kBlock), shmem, (hipStream_t)stream, idx, t, g_out, n, D, N);
  return launch_status();
}

}  // extern "C"
