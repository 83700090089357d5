// spmm.hip — CSR SpMM for LightGCN propagation  E(k+1) = A_hat * E(k)
// (src/models/lightgcn.py:82-85,166-170; operand built by src/graph_utils.py:47-98),
// with the layer sum `res = res + step` and the final `res / (L+1)` fused as an epilogue.
//
// Mapping (wave = 64): a dense row of D = 4*LPR floats is LPR lanes x float4, so one
// wave-instruction gathers NPW = 64/LPR neighbour rows (1 KiB); the loop is unrolled so 4 such
// gathers are in flight per wave.  Partial sums live in registers and are combined across the
// NPW neighbour slots with shuffle-xor.  Rows are split by degree on the host side once per
// matrix: "short" rows get one wave each; "long" rows (power-law hubs) get a whole 16-wave
// workgroup whose waves stride the row and combine through LDS — a hub of degree 5,000 otherwise
// serialises ~300 dependent gathers on one wave while the rest of the chip has finished.
// X (and the running sum) may be given as two row segments (user table | item table) so the
// reference's torch.cat of the two weight tables is never materialised.
#include "common.hpp"

namespace {
using namespace mi;

struct Seg2 {            // rows [0,split) live at a, rows [split, ..) at b (b may alias a + split*D)
  const float *a, *b;
  int split;
};
__device__ __forceinline__ const float *seg_row(const Seg2 &s, int row, int D) {
  return row < s.split ? s.a + (int64_t)row * D : s.b + (int64_t)(row - s.split) * D;
}

// xmask (nullable): one bit per row of X, 0 = the row is all zeros and is not fetched.  The first layer of a backward
// propagation multiplies A^T with a gradient that is non-zero only on the batch's rows (BPR: <= 3 B of 69 716 rows at
// Yelp2018 size), and the gather traffic — what bounds this kernel — shrinks with the fraction of rows that are not.
template <int LPR>
__device__ __forceinline__ float4 row_dot(const int *__restrict__ col, const float *__restrict__ val,
                                          const Seg2 &X, int lo, int hi, int slot, int nslots, int q,
                                          const uint32_t *__restrict__ xmask = nullptr) {
  constexpr int D = LPR * 4;
  constexpr int U = 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = lo + slot; j0 < hi; j0 += nslots * U) {
    int c[U];
    float v[U];
    float4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * nslots;
      const bool ok = j < hi;
      c[u] = ok ? col[j] : 0;
      v[u] = ok ? val[j] : 0.f;
    }
    if (xmask) {
      uint32_t mw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) mw[u] = xmask[c[u] >> 5];
#pragma unroll
      for (int u = 0; u < U; ++u)
        x[u] = ((mw[u] >> (c[u] & 31)) & 1u) ? ld4(seg_row(X, c[u], D) + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = ld4(seg_row(X, c[u], D) + q * 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a.x += v[u] * x[u].x; a.y += v[u] * x[u].y; a.z += v[u] * x[u].z; a.w += v[u] * x[u].w;
    }
  }
  return a;
}

// mask[w] bit b = row 32 w + b of X has a non-zero element; a wave builds one word, RPP = min(32, 64 / LPR) rows per pass
// (D = 4: a wave could cover 64 rows, i.e. two words — its upper 32 lanes sit the single pass out instead)
template <int LPR>
__global__ __launch_bounds__(kBlock) void k_row_mask(Seg2 X, int n_rows, uint32_t *__restrict__ mask, int n_words) {
  constexpr int NPW = kWave / LPR;
  constexpr int RPP = NPW > 32 ? 32 : NPW;
  constexpr int D = LPR * 4;
  const int lane = threadIdx.x & 63, q = lane % LPR, k = lane / LPR;
  const int word = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (word >= n_words) return;
  uint32_t bits = 0;
#pragma unroll
  for (int p = 0; p < 32 / RPP; ++p) {
    const int row = word * 32 + p * RPP + k;
    bool nz = false;
    if (k < RPP && row < n_rows) {
      const float4 x = ld4(seg_row(X, row, D) + q * 4);
      nz = x.x != 0.f || x.y != 0.f || x.z != 0.f || x.w != 0.f;
    }
    const unsigned long long b = __ballot(nz);
    // lanes of row slot k are k * LPR .. k * LPR + LPR - 1
#pragma unroll
    for (int kk = 0; kk < RPP; ++kk) {
      const unsigned long long grp = LPR == 64 ? ~0ull : (((1ull << LPR) - 1ull) << (kk * LPR));
      if (b & grp) bits |= 1u << (p * RPP + kk);
    }
  }
  if (lane == 0) mask[word] = bits;
}

__device__ __forceinline__ void epilogue4(float4 y, int row, int q, int D, float *Y, const Seg2 &acc_in,
                                          bool has_acc_in, float *acc_out, float scale) {
  if (Y) st4(Y + (int64_t)row * D + q * 4, y);
  if (acc_out) {
    float4 r = y;
    if (has_acc_in) {
      const float4 p = ld4(seg_row(acc_in, row, D) + q * 4);
      r.x += p.x; r.y += p.y; r.z += p.z; r.w += p.w;
    }
    r.x *= scale; r.y *= scale; r.z *= scale; r.w *= scale;
    st4(acc_out + (int64_t)row * D + q * 4, r);
  }
}

// one wave per (short) row
template <int LPR>
__global__ __launch_bounds__(kBlock) void k_spmm_wave_rows(
    const int *__restrict__ crow, const int *__restrict__ col, const float *__restrict__ val, Seg2 X,
    float *__restrict__ Y, Seg2 acc_in, int has_acc_in, float *__restrict__ acc_out, float scale,
    const int *__restrict__ rows, int n_items) {
  constexpr int NPW = kWave / LPR;
  constexpr int D = LPR * 4;
  const int lane = threadIdx.x & 63;
  const int q = lane % LPR, k = lane / LPR;
  const int wave0 = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * kWavesPerBlock;
  for (int it = wave0; it < n_items; it += nwaves) {
    const int row = rows ? rows[it] : it;
    const int lo = crow[row], hi = crow[row + 1];
    float4 a = row_dot<LPR>(col, val, X, lo, hi, k, NPW, q);
    a = slot_sum<LPR>(a);
    if (k == 0) epilogue4(a, row, q, D, Y, acc_in, has_acc_in != 0, acc_out, scale);
  }
}

// ONE launch of 1024-thread workgroups (16 waves): workgroups [0, n_long) each take a hub row
// (16 x NPW neighbour slots stride the row, combined through LDS); the remaining workgroups give
// every wave its own short row.  Hubs start first and overlap with the short rows instead of
// serialising behind them as a second kernel.
constexpr int kHubWaves = 16;
template <int LPR>
__global__ __launch_bounds__(kHubWaves * kWave) void k_spmm_planned(
    const int *__restrict__ crow, const int *__restrict__ col, const float *__restrict__ val, Seg2 X,
    float *__restrict__ Y, Seg2 acc_in, int has_acc_in, float *__restrict__ acc_out, float scale,
    const int *__restrict__ short_rows, int n_short, const int *__restrict__ long_rows, int n_long,
    const uint32_t *__restrict__ xmask, uint8_t *__restrict__ hub_need) {
  constexpr int NPW = kWave / LPR;
  constexpr int D = LPR * 4;
  __shared__ float4 part[kHubWaves][LPR];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int q = lane % LPR, k = lane / LPR;
  if ((int)blockIdx.x < n_long) {
    const int row = long_rows[blockIdx.x];
    // hub_need (nullable, one byte per row): a hub row is computed only when its byte is set (mi_batch_row_list sets it for
    // the hubs of the batch) and the byte is cleared again here: the array is all zeros between launches
    if (hub_need && !hub_need[row]) return;
    const int lo = crow[row], hi = crow[row + 1];
    float4 a = row_dot<LPR>(col, val, X, lo, hi, w * NPW + k, NPW * kHubWaves, q, xmask);
    a = slot_sum<LPR>(a);
    if (k == 0) part[w][q] = a;
    __syncthreads();
    if (w == 0 && k == 0) {
      float4 s = part[0][q];
#pragma unroll
      for (int i = 1; i < kHubWaves; ++i) {
        const float4 p = part[i][q];
        s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
      }
      epilogue4(s, row, q, D, Y, acc_in, has_acc_in != 0, acc_out, scale);
    }
    if (hub_need && threadIdx.x == 0) hub_need[row] = 0;
    return;
  }
  const int nblk = gridDim.x - n_long;
  for (int it = (blockIdx.x - n_long) * kHubWaves + w; it < n_short; it += nblk * kHubWaves) {
    const int row = short_rows[it];
    const int lo = crow[row], hi = crow[row + 1];
    float4 a = row_dot<LPR>(col, val, X, lo, hi, k, NPW, q, xmask);
    a = slot_sum<LPR>(a);
    if (k == 0) epilogue4(a, row, q, D, Y, acc_in, has_acc_in != 0, acc_out, scale);
  }
}


// ------------------------------------------------------------------------------- task-balanced, slice-phased (round 4) ----
// What bounds the row-per-wave kernels above is not their arithmetic but WHERE their gathers land: every XCD's waves
// gather rows of X from all over the operand (17.8 MB at Yelp2018 size against 4 MiB of L2 per XCD: hit rate 0.38, 451 MB
// of fabric traffic per layer against 54 MB compulsory, profiles/r03_pmc_summary.csv).  This form makes all waves of the
// chip walk the COLUMNS of A in the same order at about the same pace:
//   * the columns (= rows of X) are cut into slices of <= ~2 MiB of X;
//   * the rows of A are packed, once per sparsity pattern, into TASKS of about equal work whose edges are stored slice by
//     slice; a wave owns one task at a time and keeps its running sums in registers (no LDS, no atomics):
//       narrow task: kGroupRows = 4 rows per lane group (a lane group = the LPR lanes that hold one row of X): every group
//                    walks the edges of ITS rows — no combining across groups, four predicated adds per edge;
//       wide task:   1-4 rows of 48..256 nonzeros each, all lane groups stride the same edge range (combined by shuffles
//                    at the end), so that a heavy row does not leave the other groups idle;
//       hubs (> 256 nonzeros) keep a whole 8-wave workgroup (their columns ascend: slice order too);
//     tasks hold ~512 nonzeros so that ALL of them (5.3 K at Yelp2018 size + the hubs' waves) are resident at once — with
//     2-row groups and 16-wave workgroups the 11.3 K tasks ran in three rounds of 4096 waves, the later rounds out of phase
//     with the first: L2 hit rate 0.51 instead of 0.38, 70 us instead of 75 (profiles/r04_spmm_counters.txt);
//   * while the chip is "in" slice p, each XCD pulls that slice of X into its L2 ONCE and serves the ~30 re-uses per row from
//     there.  No synchronisation: the alignment is only as good as the balance, and only speed depends on it;
//   * 4 edges in flight per lane group, the next trip's edge words and values loaded before this trip's rows of X are used.
// An edge = column | (which of the owner's four rows) << 28.  Sum order is fixed by the plan: bit-reproducible.
// (First version, measured 100 -> 87 us per Yelp2018 layer against 75 for the row-per-wave kernel: 8 rows per wave with ALL
// groups striding one flat edge list and an 8-way predicated add per edge — ~3 200 vector instructions per task, the
// vector ALU became the bound.)
constexpr int kTaskColBits = 28;
constexpr int kGroupRows = 4;
constexpr int kTaskWaves = 8;          // waves per workgroup of the sliced kernel (a hub row gets all of them)

template <int LPR>
__global__ __launch_bounds__(kTaskWaves * kWave, 6) void k_spmm_sliced(
    const int *__restrict__ crow, const int *__restrict__ col, const float *__restrict__ val,      // hubs: the CSR itself
    const int *__restrict__ tptr, const int *__restrict__ trows, const uint8_t *__restrict__ twide,
    const int *__restrict__ ecol, const float *__restrict__ eval, int n_tasks, Seg2 X, float *__restrict__ Y, Seg2 acc_in,
    int has_acc_in, float *__restrict__ acc_out, float scale, const int *__restrict__ long_rows, int n_long,
    const uint32_t *__restrict__ xmask) {
  constexpr int NPW = kWave / LPR;
  constexpr int D = LPR * 4;
  constexpr int U = 4;
  __shared__ float4 part[kTaskWaves][LPR];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int q = lane % LPR, k = lane / LPR;
  if ((int)blockIdx.x < n_long) {          // a hub row: the whole workgroup strides it
    const int row = long_rows[blockIdx.x];
    const int lo = crow[row], hi = crow[row + 1];
    float4 a = row_dot<LPR>(col, val, X, lo, hi, w * NPW + k, NPW * kTaskWaves, q, xmask);
    a = slot_sum<LPR>(a);
    if (k == 0) part[w][q] = a;
    __syncthreads();
    if (w == 0 && k == 0) {
      float4 s = part[0][q];
#pragma unroll
      for (int i = 1; i < kTaskWaves; ++i) {
        const float4 p = part[i][q];
        s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
      }
      epilogue4(s, row, q, D, Y, acc_in, has_acc_in != 0, acc_out, scale);
    }
    return;
  }
  const int nblk = gridDim.x - n_long;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t = (blockIdx.x - n_long) * kTaskWaves + w; t < n_tasks; t += nblk * kTaskWaves) {
    const bool wide = twide[t] != 0;
    const int pb = t * (NPW + 1), rb = t * kGroupRows * NPW;
    const int lo = tptr[pb + (wide ? 0 : k)], hi = tptr[pb + (wide ? 1 : k + 1)];
    const int step = wide ? NPW : 1;
    int cur = lo + (wide ? k : 0);
    float4 acc[kGroupRows];
#pragma unroll
    for (int j = 0; j < kGroupRows; ++j) acc[j] = z4;
    int cw[U];
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = cur + u * step;
      const bool ok = e < hi;
      cw[u] = ok ? ecol[e] : 0;
      v[u] = ok ? eval[e] : 0.f;
    }
    while (__any(cur < hi)) {
      float4 x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = cw[u] & ((1 << kTaskColBits) - 1);
        bool on = v[u] != 0.f;                     // (an edge past the end, or a zero value: nothing to fetch)
        if (xmask) on = on && ((xmask[c >> 5] >> (c & 31)) & 1u);
        x[u] = on ? ld4(seg_row(X, c, D) + q * 4) : z4;
      }
      cur += U * step;
      int which[U];
#pragma unroll
      for (int u = 0; u < U; ++u) which[u] = (unsigned)cw[u] >> kTaskColBits;
      float pv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) pv[u] = v[u];
#pragma unroll
      for (int u = 0; u < U; ++u) {          // the next trip's edge words and values: in flight under this trip's rows of X
        const int e = cur + u * step;
        const bool ok = e < hi;
        cw[u] = ok ? ecol[e] : 0;
        v[u] = ok ? eval[e] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int j = 0; j < kGroupRows; ++j) {
          const float vj = which[u] == j ? pv[u] : 0.f;
          acc[j].x = fmaf(vj, x[u].x, acc[j].x); acc[j].y = fmaf(vj, x[u].y, acc[j].y);
          acc[j].z = fmaf(vj, x[u].z, acc[j].z); acc[j].w = fmaf(vj, x[u].w, acc[j].w);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < kGroupRows; ++j) {
      const int row = trows[rb + j * NPW + (wide ? 0 : k)];
      float4 a = acc[j];
      if (wide) a = slot_sum<LPR>(a);
      if (row >= 0 && (!wide || k == 0)) epilogue4(a, row, q, D, Y, acc_in, has_acc_in != 0, acc_out, scale);
    }
  }
}

// ---------------------------------------------------------------------------------------------- tiled, edge-parallel ----
// Round 3.  The row-per-wave kernels above keep a row's partial sum in registers, so a row's gathers form a dependent
// chain and every XCD gathers rows of X from all over the 17.8 MB operand: L2 hit rate 0.38, 411 MB of fabric traffic
// per launch against 54 MB compulsory (profiles/r02_spmm_counters.md).  This form:
//   * a workgroup owns a TILE of consecutive output rows (at most kTileRows; tiles hold about the same number of edges,
//     so a few hub rows make a small tile instead of a long tail) and keeps their sums in LDS for the whole launch;
//   * the tile's edges were bucketed ONCE per sparsity pattern by COLUMN BLOCK (blocks of X of about 2 MiB, half an
//     XCD's L2): every workgroup walks the column blocks in the same order, so the block being gathered from stays in
//     each XCD's L2 while its workgroups use it (temporal locality across workgroups, no synchronisation: speed only);
//   * one edge per LPR-lane group: a wave takes 64 packed edges with one coalesced load, hands them to its 64/LPR groups
//     by shuffles, and every group gathers one row of X (LPR x float4) and adds val * x into the tile with ds_add_f32 —
//     no per-row chain, nothing to combine across workgroups;
//   * one coalesced pass writes the tile with the fused epilogue  Y = y;  acc_out = (acc_in + y) * scale.
// The LDS adds land in an order that depends on timing, so the sums are not bit-reproducible from run to run (within
// float rounding of each other); use_deterministic_algorithms(True) keeps the row-per-wave form.
// Edge word: column | (row - tile's first row) << 23  (columns < 2^23, kTileRows <= 512).
constexpr int kTileThreads = 512;
constexpr int kTileColBits = 23;

template <int LPR>
__global__ __launch_bounds__(kTileThreads) void k_spmm_tiled(
    const int *__restrict__ tile_edge0, const int *__restrict__ tile_row0, const int *__restrict__ ecr,
    const float *__restrict__ eval, Seg2 X, float *__restrict__ Y, Seg2 acc_in, int has_acc_in,
    float *__restrict__ acc_out, float scale) {
  constexpr int D = LPR * 4;
  constexpr int G = kWave / LPR;             // edges per wave-instruction
  constexpr int U = 4;                       // gathers in flight per lane
  extern __shared__ __attribute__((aligned(16))) float tile[];       // [rows][D]
  const int t = blockIdx.x;
  const int r0 = tile_row0[t], nr = tile_row0[t + 1] - r0;
  const int e0 = tile_edge0[t], e1 = tile_edge0[t + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane % LPR, g = lane / LPR;
  for (int i = threadIdx.x; i < nr * LPR; i += kTileThreads) st4(tile + i * 4, make_float4(0.f, 0.f, 0.f, 0.f));
  __syncthreads();
  constexpr int kWaves = kTileThreads / kWave;
  for (int base = e0 + wave * kWave; base < e1; base += kWaves * kWave) {
    // 64 edges of this wave: one coalesced load each of the packed word and the value
    const int me = base + lane;
    const int cr = me < e1 ? ecr[me] : 0;
    const float vv = me < e1 ? eval[me] : 0.f;
    const int cnt = min(kWave, e1 - base);
    for (int s0 = 0; s0 < cnt; s0 += G * U) {
      int c[U];
      float v[U];
      float4 x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int src = s0 + u * G + g;              // < 64 always; edges past cnt carry val = 0 and column 0
        c[u] = __shfl(cr, src & 63);
        v[u] = __shfl(vv, src & 63);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = ld4(seg_row(X, c[u] & ((1 << kTileColBits) - 1), D) + q * 4);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (s0 + u * G + g < cnt) {
          float *dst = tile + ((unsigned)c[u] >> kTileColBits) * D + q * 4;
          atomicAdd(dst + 0, v[u] * x[u].x);
          atomicAdd(dst + 1, v[u] * x[u].y);
          atomicAdd(dst + 2, v[u] * x[u].z);
          atomicAdd(dst + 3, v[u] * x[u].w);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nr * LPR; i += kTileThreads) {
    const int row = r0 + i / LPR, qq = i % LPR;
    epilogue4(ld4(tile + i * 4), row, qq, D, Y, acc_in, has_acc_in != 0, acc_out, scale);
  }
}

// any D: one wave per row, lanes stride the columns of the dense operand
__global__ __launch_bounds__(kBlock) void k_spmm_anyD(
    const int *__restrict__ crow, const int *__restrict__ col, const float *__restrict__ val, Seg2 X,
    float *__restrict__ Y, Seg2 acc_in, int has_acc_in, float *__restrict__ acc_out, float scale,
    int n_rows, int D) {
  const int lane = threadIdx.x & 63;
  const int wave0 = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * kWavesPerBlock;
  for (int row = wave0; row < n_rows; row += nwaves) {
    const int lo = crow[row], hi = crow[row + 1];
    for (int d = lane; d < D; d += kWave) {
      float a = 0.f;
      for (int j = lo; j < hi; ++j) a += val[j] * seg_row(X, col[j], D)[d];
      if (Y) Y[(int64_t)row * D + d] = a;
      if (acc_out) {
        float r = a + (has_acc_in ? seg_row(acc_in, row, D)[d] : 0.f);
        acc_out[(int64_t)row * D + d] = r * scale;
      }
    }
  }
}

inline bool vec_ok(int D) { return D >= 4 && D <= 256 && (D & 3) == 0 && ((D >> 2) & ((D >> 2) - 1)) == 0; }

}  // namespace

extern "C" {

int mi_spmm_csr(const int32_t *crow, const int32_t *col, const float *val, const float *Xa,
                const float *Xb, int32_t x_split, float *Y, const float *acc_in_a, const float *acc_in_b,
                int32_t acc_split, float *acc_out, float scale, int32_t n_rows, int32_t D,
                const int32_t *short_rows, int32_t n_short, const int32_t *long_rows, int32_t n_long,
                void *stream) {
  return mi_spmm_csr_masked(crow, col, val, Xa, Xb, x_split, Y, acc_in_a, acc_in_b, acc_split, acc_out, scale, n_rows, D,
                            short_rows, n_short, long_rows, n_long, nullptr, stream);
}

int mi_row_mask(const float *Xa, const float *Xb, int32_t x_split, int32_t n_rows, int32_t D, uint32_t *mask, void *stream) {
  if (n_rows < 0 || D <= 0 || x_split < 0) return MI_ERR_INVALID_ARG;
  if (n_rows == 0) return MI_OK;
  if (!Xa || !mask) return MI_ERR_INVALID_ARG;
  if (!vec_ok(D) || D > 256 || (D / 4 & (D / 4 - 1)) || !aligned16(Xa) || (Xb && !aligned16(Xb))) return MI_ERR_UNSUPPORTED;
  Seg2 X{Xa, Xb ? Xb : Xa + (int64_t)x_split * D, Xb ? x_split : 0x7fffffff};
  const int n_words = (n_rows + 31) / 32, grid = (n_words + kWavesPerBlock - 1) / kWavesPerBlock;
  switch (D / 4) {
    case 1: MI_LAUNCH("row_mask", k_row_mask<1>, grid, kBlock, stream, X, n_rows, mask, n_words); break;
    case 2: MI_LAUNCH("row_mask", k_row_mask<2>, grid, kBlock, stream, X, n_rows, mask, n_words); break;
    case 4: MI_LAUNCH("row_mask", k_row_mask<4>, grid, kBlock, stream, X, n_rows, mask, n_words); break;
    case 8: MI_LAUNCH("row_mask", k_row_mask<8>, grid, kBlock, stream, X, n_rows, mask, n_words); break;
    case 16: MI_LAUNCH("row_mask", k_row_mask<16>, grid, kBlock, stream, X, n_rows, mask, n_words); break;
    case 32: MI_LAUNCH("row_mask", k_row_mask<32>, grid, kBlock, stream, X, n_rows, mask, n_words); break;
    default: MI_LAUNCH("row_mask", k_row_mask<64>, grid, kBlock, stream, X, n_rows, mask, n_words); break;
  }
  return launch_status();
}

int mi_spmm_csr_masked(const int32_t *crow, const int32_t *col, const float *val, const float *Xa,
                       const float *Xb, int32_t x_split, float *Y, const float *acc_in_a, const float *acc_in_b,
                       int32_t acc_split, float *acc_out, float scale, int32_t n_rows, int32_t D,
                       const int32_t *short_rows, int32_t n_short, const int32_t *long_rows, int32_t n_long,
                       const uint32_t *xmask, void *stream) {
  return mi_spmm_csr_sel(crow, col, val, Xa, Xb, x_split, Y, acc_in_a, acc_in_b, acc_split, acc_out, scale, n_rows, D,
                         short_rows, n_short, long_rows, n_long, xmask, nullptr, stream);
}

// k_batch_row_list: the one-wave-per-row list of a training batch for mi_spmm_csr_sel: out[j] = row j of
// (users | U + pos | U + neg) unless that row is a hub (is_hub byte set), whose place takes `filler` (any non-hub row:
// computing it is harmless) and whose hub_need byte is set instead.  Rows outside [0, n_rows) take the filler too.
__global__ __launch_bounds__(kBlock) void k_batch_row_list(const int64_t *__restrict__ users, const int64_t *__restrict__ pos,
                                                           const int64_t *__restrict__ neg, int64_t B, int64_t U,
                                                           int64_t n_rows, const uint8_t *__restrict__ is_hub, int filler,
                                                           int32_t *__restrict__ out, uint8_t *__restrict__ hub_need) {
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= 3 * B) return;
  const int64_t r = j < B ? users[j] : (j < 2 * B ? U + pos[j - B] : U + neg[j - 2 * B]);
  int o = filler;
  if ((uint64_t)r < (uint64_t)n_rows) {
    if (is_hub[r]) hub_need[r] = 1;
    else o = (int)r;
  }
  out[j] = o;
}

int mi_batch_row_list(const int64_t *users, const int64_t *pos, const int64_t *neg, int64_t B, int64_t U, int64_t n_rows,
                      const uint8_t *is_hub, int32_t filler, int32_t *out, uint8_t *hub_need, void *stream) {
  if (B < 0 || U < 0 || n_rows <= 0 || filler < 0 || filler >= n_rows) return MI_ERR_INVALID_ARG;
  if (B == 0) return MI_OK;
  if (!users || !pos || !neg || !is_hub || !out || !hub_need) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("batch_row_list", k_batch_row_list, (int)((3 * B + kBlock - 1) / kBlock), kBlock, stream, users, pos, neg, B, U, n_rows,
            is_hub, filler, out, hub_need);
  return launch_status();
}

int mi_spmm_csr_sel(const int32_t *crow, const int32_t *col, const float *val, const float *Xa,
                    const float *Xb, int32_t x_split, float *Y, const float *acc_in_a, const float *acc_in_b,
                    int32_t acc_split, float *acc_out, float scale, int32_t n_rows, int32_t D,
                    const int32_t *short_rows, int32_t n_short, const int32_t *long_rows, int32_t n_long,
                    const uint32_t *xmask, uint8_t *hub_need, void *stream) {
  if (n_rows < 0 || D <= 0 || n_short < 0 || n_long < 0 || x_split < 0 || acc_split < 0) return MI_ERR_INVALID_ARG;
  if (n_rows == 0) return MI_OK;
  if (!crow || !Xa || (!Y && !acc_out)) return MI_ERR_INVALID_ARG;
  if ((n_short > 0 && !short_rows) || (n_long > 0 && !long_rows)) return MI_ERR_INVALID_ARG;
  Seg2 X{Xa, Xb ? Xb : Xa + (int64_t)x_split * D, Xb ? x_split : 0x7fffffff};
  const int has_acc = acc_in_a != nullptr;
  Seg2 A{acc_in_a, acc_in_b ? acc_in_b : acc_in_a, acc_in_b ? acc_split : 0x7fffffff};
  const bool planned = short_rows || long_rows;
  // (xmask only lets the planned float4 kernel skip fetches of all-zero rows; the other paths ignore it — same result)
  const bool al = aligned16(Xa) && (!Xb || aligned16(Xb)) && (!Y || aligned16(Y)) &&
                  (!acc_in_a || aligned16(acc_in_a)) && (!acc_in_b || aligned16(acc_in_b)) &&
                  (!acc_out || aligned16(acc_out));
  if (planned && n_long > 60000) return MI_ERR_UNSUPPORTED;
  if (vec_ok(D) && al) {
#define CALL(LPR)                                                                                       \
  do {                                                                                                  \
    if (!planned) {                                                                                     \
      MI_LAUNCH("spmm_csr_rows", (k_spmm_wave_rows<LPR>), grid_for_waves(n_rows), kBlock, stream, crow, \
                col, val, X, Y, A, has_acc, acc_out, scale, nullptr, n_rows);                           \
    } else {                                                                                            \
      int sb = (n_short + kHubWaves - 1) / kHubWaves;                                                   \
      if (sb > 1024) sb = 1024;                                                                         \
      MI_LAUNCH("spmm_csr", (k_spmm_planned<LPR>), n_long + sb, kHubWaves * kWave, stream, crow, col,   \
                val, X, Y, A, has_acc, acc_out, scale, short_rows, n_short, long_rows, n_long, xmask,   \
                hub_need);                                                                              \
    }                                                                                                   \
  } while (0)
    switch (D / 4) {
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
    MI_LAUNCH("spmm_csr_rows", k_spmm_anyD, grid_for_waves(n_rows), kBlock, stream, crow, col, val, X, Y, A,
              has_acc, acc_out, scale, n_rows, D);
  }
  return launch_status();
}

// The tiled form (k_spmm_tiled above).  tile_edge0 / tile_row0 int32[ntiles + 1]: first edge / first row of every tile
// (rows of tile t: [tile_row0[t], tile_row0[t+1]), at most max_tile_rows); ecr int32[nnz] = column | local row << 23 and
// eval fp32[nnz] in tile order, each tile's edges grouped by column block.  The row ranges of the tiles must cover
// [0, n_rows) exactly once (every output row is written by its tile).
int mi_spmm_tiled(const int32_t *tile_edge0, const int32_t *tile_row0, int32_t ntiles, int32_t max_tile_rows,
                  const int32_t *ecr, const float *eval, const float *Xa, const float *Xb, int32_t x_split, float *Y,
                  const float *acc_in_a, const float *acc_in_b, int32_t acc_split, float *acc_out, float scale,
                  int32_t D, void *stream) {
  if (ntiles < 0 || D <= 0 || max_tile_rows <= 0 || x_split < 0 || acc_split < 0) return MI_ERR_INVALID_ARG;
  if (ntiles == 0) return MI_OK;
  if (!tile_edge0 || !tile_row0 || !ecr || !eval || !Xa || (!Y && !acc_out)) return MI_ERR_INVALID_ARG;
  const size_t lds = (size_t)max_tile_rows * D * sizeof(float);
  if (!vec_ok(D) || max_tile_rows > 512 || lds > 64 * 1024) return MI_ERR_UNSUPPORTED;      // 64 KiB: no opt-in needed
  if (!aligned16(Xa) || (Xb && !aligned16(Xb)) || (Y && !aligned16(Y)) || (acc_in_a && !aligned16(acc_in_a)) ||
      (acc_in_b && !aligned16(acc_in_b)) || (acc_out && !aligned16(acc_out)))
    return MI_ERR_UNSUPPORTED;
  Seg2 X{Xa, Xb ? Xb : Xa + (int64_t)x_split * D, Xb ? x_split : 0x7fffffff};
  const int has_acc = acc_in_a != nullptr;
  Seg2 A{acc_in_a, acc_in_b ? acc_in_b : acc_in_a, acc_in_b ? acc_split : 0x7fffffff};
  hipEvent_t ea, eb;
  const bool timed = mi::prof_acquire("spmm_tiled", &ea, &eb);
#define CALL(LPR)                                                                                                       \
  do {                                                                                                                  \
    if (timed)                                                                                                          \
      hipExtLaunchKernelGGL((k_spmm_tiled<LPR>), dim3(ntiles), dim3(kTileThreads), lds, (hipStream_t)stream, ea, eb, 0, \
                            tile_edge0, tile_row0, ecr, eval, X, Y, A, has_acc, acc_out, scale);                        \
    else                                                                                                                \
      hipLaunchKernelGGL((k_spmm_tiled<LPR>), dim3(ntiles), dim3(kTileThreads), lds, (hipStream_t)stream, tile_edge0,  \
                         tile_row0, ecr, eval, X, Y, A, has_acc, acc_out, scale);                                       \
  } while (0)
  switch (D / 4) {
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
  return launch_status();
}

// Task-balanced, slice-phased SpMM (k_spmm_sliced above), NPW = 256 / D lane groups per wave.
//   tptr  int32[n_tasks, NPW + 1]: narrow task: edge range of lane group k = [tptr[k], tptr[k + 1]); wide task: [tptr[0], tptr[1])
//   trows int32[n_tasks, 4 * NPW]: narrow: group k owns rows trows[j * NPW + k], j = 0..3; wide: rows trows[j * NPW]; -1 = none
//   twide uint8[n_tasks]: 1 = wide;  ecol int32[n_edges] = column | (which of the owner's rows) << 28, eval the values
// The tasks' rows and long_rows together must cover every output row exactly once; long_rows (hubs) are computed from the
// CSR (crow, col, val) itself.  Columns < 2^28.
int mi_spmm_sliced(const int32_t *crow, const int32_t *col, const float *val, const int32_t *tptr, const int32_t *trows,
                   const uint8_t *twide, const int32_t *ecol, const float *eval, int32_t n_tasks, const float *Xa,
                   const float *Xb, int32_t x_split, float *Y, const float *acc_in_a, const float *acc_in_b, int32_t acc_split,
                   float *acc_out, float scale, int32_t D, const int32_t *long_rows, int32_t n_long, const uint32_t *xmask,
                   void *stream) {
  if (n_tasks < 0 || n_long < 0 || D <= 0 || x_split < 0 || acc_split < 0) return MI_ERR_INVALID_ARG;
  if (n_tasks == 0 && n_long == 0) return MI_OK;
  if (!Xa || (!Y && !acc_out) || (n_tasks > 0 && (!tptr || !trows || !twide || !ecol || !eval))) return MI_ERR_INVALID_ARG;
  if (n_long > 0 && (!long_rows || !crow || !col || !val)) return MI_ERR_INVALID_ARG;
  if (!vec_ok(D) || n_long > 60000) return MI_ERR_UNSUPPORTED;
  if (!aligned16(Xa) || (Xb && !aligned16(Xb)) || (Y && !aligned16(Y)) || (acc_in_a && !aligned16(acc_in_a)) ||
      (acc_in_b && !aligned16(acc_in_b)) || (acc_out && !aligned16(acc_out)))
    return MI_ERR_UNSUPPORTED;
  Seg2 X{Xa, Xb ? Xb : Xa + (int64_t)x_split * D, Xb ? x_split : 0x7fffffff};
  const int has_acc = acc_in_a != nullptr;
  Seg2 A{acc_in_a, acc_in_b ? acc_in_b : acc_in_a, acc_in_b ? acc_split : 0x7fffffff};
  int tb = (n_tasks + kTaskWaves - 1) / kTaskWaves;
  if (tb > 4096) tb = 4096;
#define CALL(LPR)                                                                                                            \
  MI_LAUNCH("spmm_sliced", (k_spmm_sliced<LPR>), n_long + tb, kTaskWaves * kWave, stream, crow, col, val, tptr, trows, twide, \
            ecol, eval, n_tasks, X, Y, A, has_acc, acc_out, scale, long_rows, n_long, xmask)
  switch (D / 4) {
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
  return launch_status();
}

}  // extern "C"
