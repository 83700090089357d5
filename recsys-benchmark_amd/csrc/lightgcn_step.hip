// lightgcn_step.hip — the LightGCN step either side of the propagation.
//
//  * BPR loss over rows picked from the propagated tables — src/losses.py:6-22 applied to
//    torch.index_select(all_user_emb, 0, users) etc. (src/trainer/lightgcn.py:395-399): gathers,
//    two row dots, -logsigmoid, mean in ONE launch; the backward scatters the three gradient rows
//    straight into the dense table gradients (what index_select's backward builds with index_add).
//  * validation scoring tail — src/trainer/lightgcn.py:122-138: scores[ind0, ind1] = -inf for the
//    items a user already has in train (a Python double loop in the reference; here a CSR row per
//    user on the device) and torch.topk(scores, k) indices, one workgroup per user row.
//
// Ordering of the top-k: score descending, ties by ascending item index (a strict total order, so
// the result does not depend on scheduling).
#include "common.hpp"

namespace {
using namespace mi;

// row counts of the three tables the index arrays address, and the sticky error word (common convention of the lookups:
// an out-of-range id never touches memory, it is skipped and MI_IDX_OUT_OF_RANGE is OR-ed into *err)
struct RowBounds {
  int64_t nU, nP, nN;
  int *err;
  __device__ __forceinline__ bool ok(int64_t u, int64_t p, int64_t n) const {
    return (uint64_t)u < (uint64_t)nU && (uint64_t)p < (uint64_t)nP && (uint64_t)n < (uint64_t)nN;
  }
};

__device__ __forceinline__ float softplus(float x) {   // log(1 + e^x), stable
  return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
}

// 16 lanes (one float4 each) per sample when D == 64; generic D: lanes stride over d.
// part[blockIdx.x] = sum over the block's samples of softplus(-(u.p - u.n)); the last block to finish
// adds the partials in index order (deterministic) and writes the mean.
__global__ __launch_bounds__(kBlock) void k_bpr_fwd(
    const float *__restrict__ U, const int64_t *__restrict__ ui, const float *__restrict__ P,
    const int64_t *__restrict__ pi, const float *__restrict__ Nn, const int64_t *__restrict__ ni,
    int64_t B, int D, RowBounds nb, float *__restrict__ sig, float *__restrict__ part, unsigned *ticket,
    float *__restrict__ loss, const float *__restrict__ plus, float plus_w) {
  __shared__ float red[kWavesPerBlock];
  __shared__ bool last;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float acc = 0.f;
  bool bad = false;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + wv;
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t b = wave0; b < B; b += nwaves) {
    const int64_t ur = ui ? ui[b] : b, pr = pi ? pi[b] : b, nr = ni ? ni[b] : b;
    float d = 0.f;
    if (nb.ok(ur, pr, nr)) {               // an out-of-range triple reads nothing and counts as u = p = n = 0
      const float *u = U + ur * D, *p = P + pr * D, *q = Nn + nr * D;
      for (int j = lane; j < D; j += kWave) d += u[j] * (p[j] - q[j]);
    } else {
      bad = true;
    }
    d = wave_sum(d);                       // y_pos - y_neg
    if (lane == 0) {
      sig[b] = 1.f / (1.f + expf(d));      // sigmoid(-d) = -dL_b/dd
      acc += softplus(-d);                 // -logsigmoid(d)
    }
  }
  if (bad && lane == 0 && nb.err) atomicOr(nb.err, MI_IDX_OUT_OF_RANGE);
  if (lane == 0) red[wv] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int j = 0; j < kWavesPerBlock; ++j) s += red[j];
    publish_partial(part, ticket, s, last);
  }
  __syncthreads();
  if (last) {                              // fixed summation tree over the partials: deterministic
    float s = 0.f;
    for (unsigned j = threadIdx.x; j < gridDim.x; j += kBlock) s += read_partial(part + j);
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int j = 0; j < kWavesPerBlock; ++j) t += red[j];
      // plus (nullable): another term of the step's objective (the trainer's reg_weight * get_reg_loss, a scalar an earlier
      // launch wrote) joins here instead of through a scale launch and an add launch
      loss[0] = t / (float)B + (plus ? plus_w * plus[0] : 0.f);
      if (plus) loss[1] = t / (float)B;      // (the bare BPR term beside the sum: what a trainer logs)
      *ticket = 0;                         // re-armed for the next launch
    }
  }
}

// The same for rows of D = 4 LPR floats (LPR a power of two <= 64, 16-byte aligned tables): LPR lanes x float4 per sample,
// 64 / LPR samples per wave side by side.  A quarter of the workgroups of the form above at D = 64 — the launch is a chain
// of round trips (ids, rows, partial, ticket, partials) and the ticket is ONE word every workgroup adds to: same-address
// atomics serialise at ~26 ns apiece, 512 of them were most of the kernel's 10 us.
template <int LPR>
__global__ __launch_bounds__(kBlock) void k_bpr_fwd_v(
    const float *__restrict__ U, const int64_t *__restrict__ ui, const float *__restrict__ P,
    const int64_t *__restrict__ pi, const float *__restrict__ Nn, const int64_t *__restrict__ ni,
    int64_t B, int D, RowBounds nb, float *__restrict__ sig, float *__restrict__ part, unsigned *ticket,
    float *__restrict__ loss, const float *__restrict__ plus, float plus_w) {
  constexpr int SPW = kWave / LPR;
  __shared__ float red[kWavesPerBlock];
  __shared__ bool last;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int q = lane % LPR, k = lane / LPR;
  float acc = 0.f;
  bool bad = false;
  const int64_t step = (int64_t)gridDim.x * kWavesPerBlock * SPW;
  for (int64_t b0 = ((int64_t)blockIdx.x * kWavesPerBlock + wv) * SPW; b0 < B; b0 += step) {
    const int64_t b = b0 + k;
    const bool valid = b < B;
    const int64_t ur = !valid ? 0 : (ui ? ui[b] : b), pr = !valid ? 0 : (pi ? pi[b] : b), nr = !valid ? 0 : (ni ? ni[b] : b);
    float d = 0.f;
    if (valid && nb.ok(ur, pr, nr)) {
      const float4 u = ld4(U + ur * D + q * 4), p = ld4(P + pr * D + q * 4), n = ld4(Nn + nr * D + q * 4);
      d = u.x * (p.x - n.x) + u.y * (p.y - n.y) + u.z * (p.z - n.z) + u.w * (p.w - n.w);
    } else if (valid) {
      bad = true;                          // (an out-of-range triple reads nothing and counts as u = p = n = 0)
    }
#pragma unroll
    for (int m = 1; m < LPR; m <<= 1) d += __shfl_xor(d, m);
    if (q == 0 && valid) {
      sig[b] = 1.f / (1.f + expf(d));
      acc += softplus(-d);
    }
  }
  if (__any(bad) && lane == 0 && nb.err) atomicOr(nb.err, MI_IDX_OUT_OF_RANGE);
  acc = wave_sum(acc);
  if (lane == 0) red[wv] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int j = 0; j < kWavesPerBlock; ++j) s += red[j];
    publish_partial(part, ticket, s, last);
  }
  __syncthreads();
  if (last) {
    float s = 0.f;
    for (unsigned j = threadIdx.x; j < gridDim.x; j += kBlock) s += read_partial(part + j);
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int j = 0; j < kWavesPerBlock; ++j) t += red[j];
      loss[0] = t / (float)B + (plus ? plus_w * plus[0] : 0.f);
      if (plus) loss[1] = t / (float)B;
      *ticket = 0;
    }
  }
}

// dU[ui[b]] += c (p - n), dP[pi[b]] += c u, dN[ni[b]] -= c u with c = -g * sig[b] / B
// (float atomics when an index array is given — rows repeat —, plain stores otherwise)
__global__ __launch_bounds__(kBlock) void k_bpr_bwd(
    const float *__restrict__ U, const int64_t *__restrict__ ui, const float *__restrict__ P,
    const int64_t *__restrict__ pi, const float *__restrict__ Nn, const int64_t *__restrict__ ni,
    int64_t B, int D, RowBounds nb, const float *__restrict__ sig, const float *__restrict__ g,
    float *__restrict__ dU, float *__restrict__ dP, float *__restrict__ dN) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const float scale = -g[0] / (float)B;
  for (int64_t b = wave0; b < B; b += nwaves) {
    const int64_t ur = ui ? ui[b] : b, pr = pi ? pi[b] : b, nr = ni ? ni[b] : b;
    if (!nb.ok(ur, pr, nr)) continue;      // flagged by the forward; nothing is read or added out of bounds
    const float c = scale * sig[b];
    for (int j = lane; j < D; j += kWave) {
      const float u = U[ur * D + j], p = P[pr * D + j], q = Nn[nr * D + j];
      if (dU) { if (ui) atomicAdd(dU + ur * D + j, c * (p - q)); else dU[ur * D + j] = c * (p - q); }
      if (dP) { if (pi) atomicAdd(dP + pr * D + j, c * u); else dP[pr * D + j] = c * u; }
      if (dN) { if (ni) atomicAdd(dN + nr * D + j, -c * u); else dN[nr * D + j] = -c * u; }
    }
  }
}

// ---- L2 regulariser over the batch rows: LightGCN.get_reg_loss (src/models/lightgcn.py:90-100) ----------------
//   reg = ( ||U[ui]||_F^2 + ||P[pi]||_F^2 + ||Nn[ni]||_F^2 ) / (2 B)
// (x.norm(2).pow(2) of the three gathered [B, D] blocks): gathers, squares and the deterministic mean-style join in
// one launch; backward dU[ui[b]] += g * U[ui[b]] / B etc. (float atomics into caller-zeroed dense gradients).
__global__ __launch_bounds__(kBlock) void k_rowsq_fwd(
    const float *__restrict__ U, const int64_t *__restrict__ ui, const float *__restrict__ P,
    const int64_t *__restrict__ pi, const float *__restrict__ Nn, const int64_t *__restrict__ ni,
    int64_t B, int D, RowBounds nb, float *__restrict__ part, unsigned *ticket, float *__restrict__ out) {
  __shared__ float red[kWavesPerBlock];
  __shared__ bool last;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float acc = 0.f;
  bool bad = false;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + wv;
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t b = wave0; b < B; b += nwaves) {
    const int64_t ur = ui[b], pr = pi[b], nr = ni[b];
    if (!nb.ok(ur, pr, nr)) { bad = true; continue; }
    const float *u = U + ur * D, *p = P + pr * D, *q = Nn + nr * D;
    for (int j = lane; j < D; j += kWave) acc += u[j] * u[j] + p[j] * p[j] + q[j] * q[j];
  }
  if (bad && lane == 0 && nb.err) atomicOr(nb.err, MI_IDX_OUT_OF_RANGE);
  acc = wave_sum(acc);
  if (lane == 0) red[wv] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int j = 0; j < kWavesPerBlock; ++j) s += red[j];
    publish_partial(part, ticket, s, last);
  }
  __syncthreads();
  if (last) {
    float s = 0.f;
    for (unsigned j = threadIdx.x; j < gridDim.x; j += kBlock) s += read_partial(part + j);
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int j = 0; j < kWavesPerBlock; ++j) t += red[j];
      out[0] = t / (2.f * (float)B);
      *ticket = 0;
    }
  }
}

template <int LPR>      // (the float4 form, as k_bpr_fwd_v)
__global__ __launch_bounds__(kBlock) void k_rowsq_fwd_v(
    const float *__restrict__ U, const int64_t *__restrict__ ui, const float *__restrict__ P,
    const int64_t *__restrict__ pi, const float *__restrict__ Nn, const int64_t *__restrict__ ni,
    int64_t B, int D, RowBounds nb, float *__restrict__ part, unsigned *ticket, float *__restrict__ out) {
  constexpr int SPW = kWave / LPR;
  __shared__ float red[kWavesPerBlock];
  __shared__ bool last;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int q = lane % LPR, k = lane / LPR;
  float acc = 0.f;
  bool bad = false;
  const int64_t step = (int64_t)gridDim.x * kWavesPerBlock * SPW;
  for (int64_t b0 = ((int64_t)blockIdx.x * kWavesPerBlock + wv) * SPW; b0 < B; b0 += step) {
    const int64_t b = b0 + k;
    if (b >= B) continue;
    const int64_t ur = ui[b], pr = pi[b], nr = ni[b];
    if (!nb.ok(ur, pr, nr)) { bad = true; continue; }
    const float4 u = ld4(U + ur * D + q * 4), p = ld4(P + pr * D + q * 4), n = ld4(Nn + nr * D + q * 4);
    acc += dot4(u, u) + dot4(p, p) + dot4(n, n);
  }
  if (__any(bad) && lane == 0 && nb.err) atomicOr(nb.err, MI_IDX_OUT_OF_RANGE);
  acc = wave_sum(acc);
  if (lane == 0) red[wv] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int j = 0; j < kWavesPerBlock; ++j) s += red[j];
    publish_partial(part, ticket, s, last);
  }
  __syncthreads();
  if (last) {
    float s = 0.f;
    for (unsigned j = threadIdx.x; j < gridDim.x; j += kBlock) s += read_partial(part + j);
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int j = 0; j < kWavesPerBlock; ++j) t += red[j];
      out[0] = t / (2.f * (float)B);
      *ticket = 0;
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_rowsq_bwd(
    const float *__restrict__ U, const int64_t *__restrict__ ui, const float *__restrict__ P,
    const int64_t *__restrict__ pi, const float *__restrict__ Nn, const int64_t *__restrict__ ni,
    int64_t B, int D, RowBounds nb, const float *__restrict__ g, float *__restrict__ dU, float *__restrict__ dP,
    float *__restrict__ dN) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const float c = g[0] / (float)B;          // d/dw of w^2 / (2B) = w / B
  for (int64_t b = wave0; b < B; b += nwaves) {
    const int64_t ur = ui[b], pr = pi[b], nr = ni[b];
    if (!nb.ok(ur, pr, nr)) continue;
    for (int j = lane; j < D; j += kWave) {
      if (dU) atomicAdd(dU + ur * D + j, c * U[ur * D + j]);
      if (dP) atomicAdd(dP + pr * D + j, c * P[pr * D + j]);
      if (dN) atomicAdd(dN + nr * D + j, c * Nn[nr * D + j]);
    }
  }
}

// ------------------------------------------------------------------ mask + top-k per row ----
constexpr int kCand = 2048;   // candidate slots in LDS
constexpr int kSample = 4096; // row prefix that sets the candidate bound

struct Cand {
  float v;
  int i;
};

// f(value, column) over one row, 256 threads; float4 loads (4 independent elements in flight per
// iteration, two iterations unrolled) when the row is 16-byte aligned
template <class Fn>
__device__ __forceinline__ void scan_row(const float *__restrict__ row, int64_t ncol, bool vec, Fn f) {
  const int tid = threadIdx.x;
  int64_t done = 0;
  if (vec) {
    const int64_t n4 = ncol >> 2;
#pragma unroll 2
    for (int64_t c = tid; c < n4; c += kBlock) {
      const float4 x = ld4(row + c * 4);
      const int j = (int)(c * 4);
      f(x.x, j);
      f(x.y, j + 1);
      f(x.z, j + 2);
      f(x.w, j + 3);
    }
    done = n4 << 2;
  }
  for (int64_t j = done + tid; j < ncol; j += kBlock) f(row[j], (int)j);
}
__device__ __forceinline__ bool aligned16_dev(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
__device__ __forceinline__ bool before(float av, int ai, float bv, int bi) {
  return av > bv || (av == bv && ai < bi);
}

__global__ __launch_bounds__(kBlock) void k_mask_topk(
    float *__restrict__ scores, int64_t ld, int64_t ncol, const int64_t *__restrict__ users,
    const int64_t *__restrict__ crow, const int64_t *__restrict__ col, int k,
    int64_t *__restrict__ out_idx, float *__restrict__ out_val) {
  __shared__ Cand cand[kCand];
  __shared__ float lmax[kBlock];
  __shared__ int lidx[kBlock];
  __shared__ float thr_v;
  __shared__ int thr_i;
  __shared__ int count;
  const int tid = threadIdx.x;
  float *row = scores + (int64_t)blockIdx.x * ld;
  const float ninf = -__builtin_huge_valf();

  if (crow) {   // items the user already interacted with in train never rank
    const int64_t u = users ? users[blockIdx.x] : blockIdx.x;
    for (int64_t e = crow[u] + tid; e < crow[u + 1]; e += kBlock) {
      const int64_t c = col[e];
      if ((uint64_t)c < (uint64_t)ncol) row[c] = ninf;
    }
    __syncthreads();
  }

  // pass 1 over a PREFIX of the row (a full second read of a 150 KB row would come from HBM again: 2048
  // rows are in flight): every thread's best element of the sample; the k-th best of those 256 bounds the
  // k-th best of the whole row from below
  const bool vec = ((ld & 3) == 0) && aligned16_dev(scores);
  float bv = ninf;
  int bi = 0x7fffffff;
  scan_row(row, ncol < kSample ? ncol : (int64_t)kSample, vec, [&](float v, int j) {
    if (before(v, j, bv, bi)) { bv = v; bi = j; }
  });
  lmax[tid] = bv;
  lidx[tid] = bi;
  if (tid == 0) {
    count = 0;
    thr_v = ninf;          // default bound admits everything (fewer than k threads own an element)
    thr_i = 0x7fffffff;
  }
  __syncthreads();
  if (bi != 0x7fffffff) {
    int rank = 0;
    for (int t = 0; t < kBlock; ++t) rank += before(lmax[t], lidx[t], bv, bi);
    if (rank == k - 1) { thr_v = bv; thr_i = bi; }
  }
  __syncthreads();
  const float tv = thr_v;
  const int ti = thr_i;
  // pass 2, the one full read: everything not after the bound is a candidate (~k * ncol / kSample of them)
  scan_row(row, ncol, vec, [&](float v, int j) {
    if (!before(tv, ti, v, j)) {
      const int s = atomicAdd(&count, 1);
      if (s < kCand) { cand[s].v = v; cand[s].i = j; }
    }
  });
  __syncthreads();
  const int n = count;
  if (n <= kCand) {
    // bitonic sort of the candidates in LDS under the strict total order (padded to a power of two with
    // entries that sort last); the first k are the answer
    int P = 64;
    while (P < n) P <<= 1;
    for (int a = n + tid; a < P; a += kBlock) { cand[a].v = ninf; cand[a].i = 0x7fffffff; }
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = tid; t < (P >> 1); t += kBlock) {
          const int lo = 2 * t - (t & (stride - 1));
          const int hi = lo + stride;
          const Cand a = cand[lo], b = cand[hi];
          const bool desc = (lo & size) == 0;
          const bool swap = desc ? before(b.v, b.i, a.v, a.i) : before(a.v, a.i, b.v, b.i);
          if (swap) { cand[lo] = b; cand[hi] = a; }
        }
        __syncthreads();
      }
    }
    if (tid < k) {
      out_idx[(int64_t)blockIdx.x * k + tid] = cand[tid].i;
      if (out_val) out_val[(int64_t)blockIdx.x * k + tid] = cand[tid].v;
    }
    return;
  }
  // too many candidates (massive ties): k rounds of "best element after the previous pick"
  float pv = __builtin_huge_valf();
  int pi = -1;
  for (int r = 0; r < k; ++r) {
    float cv = ninf;
    int ci = 0x7fffffff;
    scan_row(row, ncol, vec, [&](float v, int j) {
      const bool after_prev = (pi < 0) || before(pv, pi, v, j);
      if (after_prev && before(v, j, cv, ci)) { cv = v; ci = j; }
    });
    __syncthreads();
    lmax[tid] = cv;
    lidx[tid] = ci;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
      if (tid < s && before(lmax[tid + s], lidx[tid + s], lmax[tid], lidx[tid])) {
        lmax[tid] = lmax[tid + s];
        lidx[tid] = lidx[tid + s];
      }
      __syncthreads();
    }
    pv = lmax[0];
    pi = lidx[0];
    if (tid == 0) {
      out_idx[(int64_t)blockIdx.x * k + r] = pi;
      if (out_val) out_val[(int64_t)blockIdx.x * k + r] = pv;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" {

int64_t mi_bpr_workspace_elems(int64_t B) {
  if (B < 0) return 0;
  return grid_for_waves(B) + 1;   // per-block partial sums + the ticket word
}

static int bpr_fwd_impl(bool zero_ticket, const float *U, const int64_t *ui, const float *P, const int64_t *pi, const float *Nn,
               const int64_t *ni, int64_t B, int32_t D, int64_t nU, int64_t nP, int64_t nN, int32_t *err,
               float *sig, float *workspace, float *loss, void *stream, const float *plus = nullptr, float plus_w = 0.f) {
  if (B <= 0 || D <= 0) return MI_ERR_INVALID_ARG;
  if ((!ui && nU < B) || (!pi && nP < B) || (!ni && nN < B)) return MI_ERR_INVALID_ARG;
  const RowBounds nb{nU, nP, nN, err};
  if (!U || !P || !Nn || !sig || !workspace || !loss) return MI_ERR_INVALID_ARG;
  const int grid = grid_for_waves(B);
  // workspace[grid] is the ticket: zeroed here once per call (captured as a memset node in a graph)
  if (zero_ticket && hipMemsetAsync(workspace + grid, 0, sizeof(unsigned), (hipStream_t)stream) != hipSuccess)
    return MI_ERR_LAUNCH;
  unsigned *ticket = reinterpret_cast<unsigned *>(workspace + grid);      // workspace[mi_bpr_workspace_elems(B) - 1], both forms
  const int lpr = D / 4;
  if (D % 4 == 0 && lpr <= 64 && (lpr & (lpr - 1)) == 0 && aligned16(U) && aligned16(P) && aligned16(Nn)) {
    const int gv = grid_for_waves((B + 64 / lpr - 1) / (64 / lpr));
#define BPRV(L) MI_LAUNCH("bpr_fwd", (k_bpr_fwd_v<L>), gv, kBlock, stream, U, ui, P, pi, Nn, ni, B, D, nb, sig, workspace, ticket, loss, plus, plus_w)
    switch (lpr) {
      case 1: BPRV(1); break;
      case 2: BPRV(2); break;
      case 4: BPRV(4); break;
      case 8: BPRV(8); break;
      case 16: BPRV(16); break;
      case 32: BPRV(32); break;
      default: BPRV(64); break;
    }
#undef BPRV
    return launch_status();
  }
  MI_LAUNCH("bpr_fwd", k_bpr_fwd, grid, kBlock, stream, U, ui, P, pi, Nn, ni, B, D, nb, sig, workspace, ticket, loss, plus, plus_w);
  return launch_status();
}

int mi_bpr_fwd(const float *U, const int64_t *ui, const float *P, const int64_t *pi, const float *Nn,
               const int64_t *ni, int64_t B, int32_t D, int64_t nU, int64_t nP, int64_t nN, int32_t *err,
               float *sig, float *workspace, float *loss, void *stream) {
  return bpr_fwd_impl(true, U, ui, P, pi, Nn, ni, B, D, nU, nP, nN, err, sig, workspace, loss, stream);
}

// the same without the memset node: the caller promises that the ticket word (workspace[mi_bpr_workspace_elems(B) - 1])
// is zero on entry — the kernel leaves it zero, so a workspace zeroed ONCE and kept serves every later call
int mi_bpr_fwd_armed(const float *U, const int64_t *ui, const float *P, const int64_t *pi, const float *Nn,
               const int64_t *ni, int64_t B, int32_t D, int64_t nU, int64_t nP, int64_t nN, int32_t *err,
               float *sig, float *workspace, float *loss, void *stream) {
  return bpr_fwd_impl(false, U, ui, P, pi, Nn, ni, B, D, nU, nP, nN, err, sig, workspace, loss, stream);
}

int mi_bpr_fwd_plus(const float *U, const int64_t *ui, const float *P, const int64_t *pi, const float *Nn,
                    const int64_t *ni, int64_t B, int32_t D, int64_t nU, int64_t nP, int64_t nN, int32_t *err,
                    float *sig, float *workspace, int32_t armed, const float *plus, float plus_weight, float *loss,
                    void *stream) {
  return bpr_fwd_impl(!armed, U, ui, P, pi, Nn, ni, B, D, nU, nP, nN, err, sig, workspace, loss, stream, plus, plus_weight);
}

int mi_bpr_bwd(const float *U, const int64_t *ui, const float *P, const int64_t *pi, const float *Nn,
               const int64_t *ni, int64_t B, int32_t D, int64_t nU, int64_t nP, int64_t nN, const float *sig,
               const float *g, float *dU, float *dP, float *dN, void *stream) {
  if (B <= 0 || D <= 0) return MI_ERR_INVALID_ARG;
  if ((!ui && nU < B) || (!pi && nP < B) || (!ni && nN < B)) return MI_ERR_INVALID_ARG;
  const RowBounds nb{nU, nP, nN, nullptr};
  if (!U || !P || !Nn || !sig || !g) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("bpr_bwd", k_bpr_bwd, grid_for_waves(B), kBlock, stream, U, ui, P, pi, Nn, ni, B, D, nb, sig, g,
            dU, dP, dN);
  return launch_status();
}

static int rowsq_fwd_impl(bool zero_ticket, const float *U, const int64_t *ui, const float *P, const int64_t *pi, const float *Nn,
                 const int64_t *ni, int64_t B, int32_t D, int64_t nU, int64_t nP, int64_t nN, int32_t *err,
                 float *workspace, float *out, void *stream) {
  if (B <= 0 || D <= 0) return MI_ERR_INVALID_ARG;
  const RowBounds nb{nU, nP, nN, err};
  if (!U || !P || !Nn || !ui || !pi || !ni || !workspace || !out) return MI_ERR_INVALID_ARG;
  const int grid = grid_for_waves(B);
  if (zero_ticket && hipMemsetAsync(workspace + grid, 0, sizeof(unsigned), (hipStream_t)stream) != hipSuccess)
    return MI_ERR_LAUNCH;
  unsigned *ticket = reinterpret_cast<unsigned *>(workspace + grid);
  const int lpr = D / 4;
  if (D % 4 == 0 && lpr <= 64 && (lpr & (lpr - 1)) == 0 && aligned16(U) && aligned16(P) && aligned16(Nn)) {
    const int gv = grid_for_waves((B + 64 / lpr - 1) / (64 / lpr));
#define RSQV(L) MI_LAUNCH("rowsq_fwd", (k_rowsq_fwd_v<L>), gv, kBlock, stream, U, ui, P, pi, Nn, ni, B, D, nb, workspace, ticket, out)
    switch (lpr) {
      case 1: RSQV(1); break;
      case 2: RSQV(2); break;
      case 4: RSQV(4); break;
      case 8: RSQV(8); break;
      case 16: RSQV(16); break;
      case 32: RSQV(32); break;
      default: RSQV(64); break;
    }
#undef RSQV
    return launch_status();
  }
  MI_LAUNCH("rowsq_fwd", k_rowsq_fwd, grid, kBlock, stream, U, ui, P, pi, Nn, ni, B, D, nb, workspace, ticket, out);
  return launch_status();
}

int mi_rowsq_fwd(const float *U, const int64_t *ui, const float *P, const int64_t *pi, const float *Nn,
                 const int64_t *ni, int64_t B, int32_t D, int64_t nU, int64_t nP, int64_t nN, int32_t *err,
                 float *workspace, float *out, void *stream) {
  return rowsq_fwd_impl(true, U, ui, P, pi, Nn, ni, B, D, nU, nP, nN, err, workspace, out, stream);
}

int mi_rowsq_fwd_armed(const float *U, const int64_t *ui, const float *P, const int64_t *pi, const float *Nn,
                 const int64_t *ni, int64_t B, int32_t D, int64_t nU, int64_t nP, int64_t nN, int32_t *err,
                 float *workspace, float *out, void *stream) {      // see mi_bpr_fwd_armed
  return rowsq_fwd_impl(false, U, ui, P, pi, Nn, ni, B, D, nU, nP, nN, err, workspace, out, stream);
}

int mi_rowsq_bwd(const float *U, const int64_t *ui, const float *P, const int64_t *pi, const float *Nn,
                 const int64_t *ni, int64_t B, int32_t D, int64_t nU, int64_t nP, int64_t nN, const float *g,
                 float *dU, float *dP, float *dN, void *stream) {
  if (B <= 0 || D <= 0) return MI_ERR_INVALID_ARG;
  const RowBounds nb{nU, nP, nN, nullptr};
  if (!U || !P || !Nn || !ui || !pi || !ni || !g) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("rowsq_bwd", k_rowsq_bwd, grid_for_waves(B), kBlock, stream, U, ui, P, pi, Nn, ni, B, D, nb, g, dU, dP, dN);
  return launch_status();
}

int mi_mask_topk_rows(float *scores, int64_t ld, int64_t nrows, int64_t ncol, const int64_t *users,
                      const int64_t *crow, const int64_t *col, int32_t k, int64_t *out_idx,
                      float *out_val, void *stream) {
  if (nrows < 0 || ncol < 0 || ld < ncol || k < 1) return MI_ERR_INVALID_ARG;
  if (k > ncol || ncol >= (1ll << 31)) return MI_ERR_INVALID_ARG;   // torch.topk raises for k > size
  if (k > kBlock) return MI_ERR_UNSUPPORTED;
  if (nrows == 0) return MI_OK;
  if (!scores || !out_idx || (crow && !col)) return MI_ERR_INVALID_ARG;
  if (nrows > 0x7fffffff) return MI_ERR_UNSUPPORTED;
  MI_LAUNCH("mask_topk", k_mask_topk, (int)nrows, kBlock, stream, scores, ld, ncol, users, crow, col, k,
            out_idx, out_val);
  return launch_status();
}

}  // extern "C"
