// gemm.hip — fp32 GEMM on the gfx950 matrix cores with fused epilogues, the building block of
// the CrossNet heads (src/models/layer_dcn.py:8-140).  These ARE real contractions, so they go
// on MFMA: v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, exact fp32 — no xf32/TF32 on gfx950).
//
//   C[M,N] = epilogue( sum_{g<kgroups} opA(A + g*gA)[M,K] * opB(B + g*gB)[K,N] ),  batched over blockIdx.z
//
// Tiling: 64x64 output tile per 256-thread workgroup, 4 waves as 2x2, one 32x32 MFMA tile per
// wave (M = 4096 with N = 352..416 gives 384..448 workgroups for 256 CUs; bigger tiles would
// leave CUs idle at these shapes).  BK = 32.  Both operands are staged in LDS as [row][k] with k
// contiguous (row stride 36 floats: conflict-free ds_read_b128), whatever their global layout —
// a k-strided source is transposed on the way in.  A lane (i = lane&31, kh = lane>>5) reads one
// float4 = k-slots {8c + 4kh + j}, j=0..3, per operand and issues 4 MFMAs: the k order inside a
// chunk is permuted identically for A and B, which a sum over k does not see.  The next tile's
// global loads are issued before the current tile's MFMAs (register-staged double buffering).
#include <cstdlib>

#include "common.hpp"
#include "prefetch_rows.hpp"

namespace {
using namespace mi;

typedef float floatx16 __attribute__((ext_vector_type(16)));

enum {
  EPI_NONE = 0,       // C = acc
  EPI_BIAS = 1,       // C = acc + bias[n]
  EPI_TANH = 2,       // C = tanh(acc)
  EPI_CROSS = 3,      // lin = acc + bias[n]*rs(m); C = R1 + R2*lin; C2 (opt) = lin     rs(m) = sum_e rowscale[m*nrs+e] or 1
  EPI_ADD = 4,        // C = R1 + acc (+ R2 when given) (+ sum_e rowscale[m*nrs+e] * bias[e*N+n] when both given)
  EPI_MUL_DTANH = 5,  // C = acc * (1 - R1^2)
  EPI_TANH_GATE = 6,  // h = tanh(acc); C = h; C2 = h * rowscale[m*nrs + z]
  EPI_ACCUM = 7,      // C = C + acc
};

struct GemmArgs {
  const float *A, *B;
  float *C;
  int M, N, K;
  int lda, ldb, ldc;
  int transA, transB;
  long long sA, sB, sC;   // batch strides
  int kgroups;
  long long gA, gB;       // K-group strides
  int epi;
  const float *bias;
  const float *R1, *R2;
  int ldr1, ldr2;
  long long sR1, sR2;
  const float *rowscale;
  int nrs;
  float *C2;
  int ldc2;
  long long sC2;
  int alignedA, alignedB;
  int splitk;             // >1: blockIdx.z = batch*splitk + slice; slices add into C atomically (EPI_NONE/ACCUM only)
  // grouped forms (TT-Rec levels, tt_grouped.hip): both null for a plain GEMM
  const int *mtile_b;     // [gridDim.y] B slice of each 64-row tile of A/C (B + idx*sB), -1: tile unused
  const long long *kseg;  // [batch][3] = (first reduction row, rows, C slice): per-z K range, C + slice*sC, atomic adds
  int xcd_swizzle;        // remap workgroup ids so that the tiles of one XCD are neighbours (plain GEMMs)
};

constexpr int BN = 64, BK = 32, LDSS = BK + 4;
constexpr int KQ = BK / 4;               // float4 per tile row (k contiguous)
// The A/C tile is 64 or 128 rows (template BMT): 128 x 64 feeds two MFMA blocks per B fragment read and is used for
// large problems only (see mi_gemm_f32).

// Stage a [64 rows][32 k] tile of an operand into LDS (row = output index, k contiguous).
// trans == 0: global is [row][k] (k contiguous);  trans == 1: global is [k][row].
// LDS tiles are [row][k] with rows of LDSS = 36 floats and the eight 4-float chunks of a row XOR-swizzled by
// (row >> 3) & 7.  The pad makes the fragment reads (8 consecutive rows, same chunk) conflict-free; the swizzle is
// constant inside every aligned group of 8 rows, so it keeps that property, and it spreads the SCALAR stores of a
// transposed operand (a thread owns 4 consecutive rows at one k) over all 32 banks: without it they fall on 8 banks
// (8-way conflicts; SQ_LDS_BANK_CONFLICT was 78 % of the LDS cycles of a weight-gradient GEMM), with it 2-way — the
// minimum for 64 lanes.
__device__ __forceinline__ int swz(int row, int chunk) { return (chunk ^ ((row >> 3) & 7)) << 2; }

template <int ROWS>
struct Staged {
  float4 v[ROWS * BK / 4 / 256];   // float4 loads per thread for a [ROWS][BK] operand tile
};

__device__ __forceinline__ float4 guarded4(const float *p, long long off, int nvalid, bool aligned) {
  if (nvalid >= 4 && aligned) return *reinterpret_cast<const float4 *>(p + off);
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nvalid > 0) r.x = p[off];
  if (nvalid > 1) r.y = p[off + 1];
  if (nvalid > 2) r.z = p[off + 2];
  if (nvalid > 3) r.w = p[off + 3];
  return r;
}

// Tile staging.  FAST (chosen ONCE per workgroup: 16-B aligned operands, the 64 rows of both
// operands in bounds) loads full k-tiles with two unguarded float4 per thread — no branch may sit
// between a load and its use inside the pipelined loop, or the compiler drains vmcnt(0) at the merge
// and the prefetch is lost (seen in the ISA of the first version).  The guarded form handles edge
// workgroups and the K tail.
// Rows past the operand's end (edge workgroups) are CLAMPED to the last valid row / row group: the
// duplicated data only feeds output rows or columns the epilogue never stores, and every address
// stays inside the buffer — so edge workgroups run the same branch-free pipeline.
template <int TRANS, int ROWS>
__device__ __forceinline__ Staged<ROWS> stage_load_fast(const float *P, int ld, int row0, int nrows, int k0) {
  Staged<ROWS> s;
  constexpr int NLD = ROWS * BK / 4 / 256, RQ = ROWS / 4;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int f = t + i * 256;
    if constexpr (!TRANS) {
      int r = row0 + (f / KQ);
      r = r < nrows ? r : nrows - 1;
      s.v[i] = *reinterpret_cast<const float4 *>(P + (long long)r * ld + k0 + (f % KQ) * 4);
    } else {
      int r = row0 + (f % RQ) * 4;          // nrows % 4 == 0 on this path: a group is all-in or all-out
      r = r < nrows ? r : nrows - 4;
      s.v[i] = *reinterpret_cast<const float4 *>(P + (long long)(k0 + (f / RQ)) * ld + r);
    }
  }
  return s;
}

// The two float4 addresses a thread fetches of a [64 rows][32 k] tile at k offset k0 (same mapping as
// stage_load_fast); consecutive k-tiles are a constant stride apart, so the pipelined loop keeps these as
// loop-carried registers and never recomputes an address next to an in-flight load.
template <int ROWS>
struct TilePtr {
  const float *p[ROWS * BK / 4 / 256];
};
template <int TRANS, int ROWS>
__device__ __forceinline__ TilePtr<ROWS> tile_ptrs(const float *P, int ld, int row0, int nrows, int k0) {
  TilePtr<ROWS> tp;
  constexpr int NLD = ROWS * BK / 4 / 256, RQ = ROWS / 4;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int f = t + i * 256;
    if constexpr (!TRANS) {
      int r = row0 + (f / KQ);
      r = r < nrows ? r : nrows - 1;
      tp.p[i] = P + (long long)r * ld + k0 + (f % KQ) * 4;
    } else {
      int r = row0 + (f % RQ) * 4;
      r = r < nrows ? r : nrows - 4;
      tp.p[i] = P + (long long)(k0 + (f / RQ)) * ld + r;
    }
  }
  return tp;
}
template <int ROWS>
__device__ __forceinline__ Staged<ROWS> load_tile(const TilePtr<ROWS> &tp) {
  Staged<ROWS> s;
#pragma unroll
  for (int i = 0; i < ROWS * BK / 4 / 256; ++i) s.v[i] = *reinterpret_cast<const float4 *>(tp.p[i]);
  return s;
}

template <int TRANS, int ROWS>
__device__ __forceinline__ Staged<ROWS> stage_load_guarded(const float *P, int ld, int row0, int nrows, int k0, int K,
                                                           bool aligned) {
  Staged<ROWS> s;
  constexpr int NLD = ROWS * BK / 4 / 256, RQ = ROWS / 4;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int f = t + i * 256;
    if constexpr (!TRANS) {
      const int row = f / KQ, kq = (f % KQ) * 4;
      const int r = row0 + row, k = k0 + kq;
      const int nv = (r < nrows) ? (K - k) : 0;
      s.v[i] = guarded4(P, (long long)r * ld + k, nv, aligned);
    } else {
      const int kk = f / RQ, mq = (f % RQ) * 4;
      const int k = k0 + kk, r = row0 + mq;
      const int nv = (k < K) ? (nrows - r) : 0;
      s.v[i] = guarded4(P, (long long)k * ld + r, nv, aligned);
    }
  }
  return s;
}

template <int TRANS, int ROWS>
__device__ __forceinline__ void stage_store(float (*T)[LDSS], const Staged<ROWS> &s) {
  constexpr int NLD = ROWS * BK / 4 / 256, RQ = ROWS / 4;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int f = t + i * 256;
    if constexpr (!TRANS) {
      const int row = f / KQ;
      *reinterpret_cast<float4 *>(&T[row][swz(row, f % KQ)]) = s.v[i];
    } else {
      const int kk = f / RQ, mq = (f % RQ) * 4;
      // (mq .. mq+3 share row >> 3: one swizzled column for the four rows)
      const int col = swz(mq, kk >> 2) + (kk & 3);
      T[mq + 0][col] = s.v[i].x;
      T[mq + 1][col] = s.v[i].y;
      T[mq + 2][col] = s.v[i].z;
      T[mq + 3][col] = s.v[i].w;
    }
  }
}

// RB 32-row blocks of A per wave against one 32-column block of B: the B fragment is read once per RB MFMA groups
template <int RB>
__device__ __forceinline__ void mma_tile(floatx16 (&acc)[RB], float (*As)[LDSS], float (*Bs)[LDSS], int ar, int br, int kh) {
#pragma unroll
  for (int c = 0; c < BK / 8; ++c) {
    const float4 bv = *reinterpret_cast<const float4 *>(&Bs[br][swz(br, c * 2 + kh)]);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const float4 av = *reinterpret_cast<const float4 *>(&As[ar + rb * 32][swz(ar + rb * 32, c * 2 + kh)]);
      acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[rb], 0, 0, 0);
      acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[rb], 0, 0, 0);
      acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[rb], 0, 0, 0);
      acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[rb], 0, 0, 0);
    }
  }
}

// TA: A is [k][m] in memory; BT: B is staged transposed, i.e. B is [k][n] in memory (transB == 0)
// One workgroup's tile of one problem.  (wx, wy, wz) is the workgroup's place in that problem's (gx, gy, .) grid — the
// launch's own blockIdx for a single problem, a slice of the linear id for a multi-problem launch.
template <int TA, int BT, int BMT>
__device__ __forceinline__ void gemm_workgroup(GemmArgs a, const int wx, const int wy, const int wz, const int gx, const int gy) {
  constexpr int BM = BMT, RB = BMT / 64, WROWS = BMT / 2;   // rows per workgroup tile / 32-row blocks and rows per wave
  __shared__ __attribute__((aligned(16))) float As[2][BM][LDSS];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN][LDSS];
  const int z = wz / a.splitk;
  const int slice = wz % a.splitk;
  // Workgroups are dealt to the 8 XCDs round-robin in launch order, and each XCD has its own L2: with the natural
  // order the 7 column tiles of one row tile land on 7 different XCDs and every L2 fetches that A tile again.  The
  // remap gives XCD x the contiguous range of tiles [x * total/8, ...): neighbours in (row tile, column tile) order
  // share an L2.
  int bx = wx, by = wy;
  if (a.xcd_swizzle) {
    const int total = gx * gy, id = by * gx + bx;
    const int x = id & 7, slot = id >> 3, chunk = total >> 3, rem = total & 7;
    const int logical = x * chunk + (x < rem ? x : rem) + slot;
    bx = logical % gx;
    by = logical / gx;
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int i = lane & 31, kh = lane >> 5;
  const int ar = wr * WROWS + i, br = wc * 32 + i;

  const float *A = a.A + z * a.sA;
  const float *B = a.B + z * a.sB;
  long long cslice = z;
  if (a.mtile_b) {                       // row-grouped: this tile's rows all multiply the same B slice
    const int g = a.mtile_b[wy];
    if (g < 0) return;
    B = a.B + (long long)g * a.sB;
  }
  if (a.kseg) {                          // reduction-grouped: z owns rows [k0, k0+K) of both operands
    const long long k0 = a.kseg[z * 3];
    a.K = (int)a.kseg[z * 3 + 1];
    cslice = a.kseg[z * 3 + 2];
    A = a.A + k0 * (TA ? a.lda : 1);
    B = a.B + k0 * (BT ? a.ldb : 1);
  }

  floatx16 acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;

  const int ktiles = (a.K + BK - 1) / BK;
  const int all = ktiles * a.kgroups;
  // this slice's range of (group, k-tile) steps
  const int per = (all + a.splitk - 1) / a.splitk;
  const int first = slice * per;
  const int total = (first + per <= all) ? first + per : all;
  if (first >= total) return;   // uniform per workgroup: an empty slice adds nothing

  const bool fast = a.alignedA && a.alignedB && (!TA || (a.M % 4 == 0 && a.M >= 4)) && (!BT || (a.N % 4 == 0 && a.N >= 4));
  if (fast) {
    // Pipelined over FULL k-tiles only; a partial last tile of a group is a guarded step.
    const int kfull = a.K / BK;                       // full tiles per group
    auto is_full = [&](int step) { return (step % ktiles) < kfull; };
    int it = first;
    while (it < total) {
      if (!is_full(it)) {   // K tail of a group: one unpipelined guarded tile
        const Staged<BM> ta = stage_load_guarded<TA, BM>(A + (it / ktiles) * a.gA, a.lda, m0, a.M, (it % ktiles) * BK, a.K, true);
        const Staged<BN> tb = stage_load_guarded<BT, BN>(B + (it / ktiles) * a.gB, a.ldb, n0, a.N, (it % ktiles) * BK, a.K, true);
        __syncthreads();
        stage_store<TA, BM>(As[0], ta);
        stage_store<BT, BN>(Bs[0], tb);
        __syncthreads();
        mma_tile<RB>(acc, As[0], Bs[0], ar, br, kh);
        ++it;
        continue;
      }
      // run of consecutive full tiles [it, run_end) of ONE k-group
      const int grp = it / ktiles;
      int run_end = (grp * ktiles + kfull < total) ? grp * ktiles + kfull : total;
      // Two register sets with FIXED roles, the loop unrolled by two k-tiles: the loads of tile s+2 are issued
      // before the MFMAs of tile s and written to LDS after the MFMAs of tile s+1, i.e. every tile's loads have
      // two MFMA phases to land.  (A rotating "next = next2" hand-over costs a register move of an in-flight
      // load's result, which forces vmcnt(0) at the top of every iteration and halves the prefetch distance:
      // seen in the ISA of the previous version.)  Loads past the run are clamped re-loads, never stored.
      const int last = run_end - 1;
      TilePtr<BM> pa = tile_ptrs<TA, BM>(A + grp * a.gA, a.lda, m0, a.M, (it % ktiles) * BK);
      TilePtr<BN> pb = tile_ptrs<BT, BN>(B + grp * a.gB, a.ldb, n0, a.N, (it % ktiles) * BK);
      const long long stepA = TA ? (long long)BK * a.lda : BK, stepB = BT ? (long long)BK * a.ldb : BK;
      int tl = it;                                   // the tile the running pointers address
      auto advance = [&]() {
        const long long da = tl < last ? stepA : 0, db = tl < last ? stepB : 0;
        tl += tl < last ? 1 : 0;
#pragma unroll
        for (int q = 0; q < BM * BK / 4 / 256; ++q) pa.p[q] += da;
#pragma unroll
        for (int q = 0; q < BN * BK / 4 / 256; ++q) pb.p[q] += db;
      };
      __syncthreads();
      Staged<BM> r0a = load_tile(pa);
      Staged<BN> r0b = load_tile(pb);
      advance();
      stage_store<TA, BM>(As[0], r0a);
      stage_store<BT, BN>(Bs[0], r0b);
      r0a = load_tile(pa);
      r0b = load_tile(pb);
      advance();
      Staged<BM> r1a;
      Staged<BN> r1b;
      __syncthreads();
      int s2 = it;
      while (true) {
        // LDS[0] = tile s2, R0 = tile s2+1
        r1a = load_tile(pa);
        r1b = load_tile(pb);
        advance();
        __builtin_amdgcn_sched_barrier(0);
        mma_tile<RB>(acc, As[0], Bs[0], ar, br, kh);
        __builtin_amdgcn_sched_barrier(0);
        if (s2 + 1 < run_end) {
          stage_store<TA, BM>(As[1], r0a);
          stage_store<BT, BN>(Bs[1], r0b);
        }
        __syncthreads();
        if (++s2 >= run_end) break;
        // LDS[1] = tile s2, R1 = tile s2+1
        r0a = load_tile(pa);
        r0b = load_tile(pb);
        advance();
        __builtin_amdgcn_sched_barrier(0);
        mma_tile<RB>(acc, As[1], Bs[1], ar, br, kh);
        __builtin_amdgcn_sched_barrier(0);
        if (s2 + 1 < run_end) {
          stage_store<TA, BM>(As[0], r1a);
          stage_store<BT, BN>(Bs[0], r1b);
        }
        __syncthreads();
        if (++s2 >= run_end) break;
      }
      it = run_end;
    }
  } else {
    for (int it = first; it < total; ++it) {
      const Staged<BM> ta = stage_load_guarded<TA, BM>(A + (it / ktiles) * a.gA, a.lda, m0, a.M, (it % ktiles) * BK, a.K, a.alignedA);
      const Staged<BN> tb = stage_load_guarded<BT, BN>(B + (it / ktiles) * a.gB, a.ldb, n0, a.N, (it % ktiles) * BK, a.K, a.alignedB);
      __syncthreads();
      stage_store<TA, BM>(As[0], ta);
      stage_store<BT, BN>(Bs[0], tb);
      __syncthreads();
      mma_tile<RB>(acc, As[0], Bs[0], ar, br, kh);
    }
  }

  // ---- epilogue: lane holds column n, rows (reg&3) + 8*(reg>>2) + 4*(lane>>5) of its 32x32 tile
  const int n = n0 + wc * 32 + i;
  if (n >= a.N) return;
  float *C = a.C + cslice * a.sC;
  const float *R1 = a.R1 ? a.R1 + z * a.sR1 : nullptr;
  const float *R2 = a.R2 ? a.R2 + z * a.sR2 : nullptr;
  float *C2 = a.C2 ? a.C2 + z * a.sC2 : nullptr;
  const float bn = (a.bias && (a.epi == EPI_BIAS || a.epi == EPI_CROSS)) ? a.bias[n] : 0.f;
#pragma unroll
  for (int rr = 0; rr < 16 * RB; ++rr) {
    const int rb = rr >> 4, r = rr & 15;
    const int m = m0 + wr * WROWS + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
    if (m >= a.M) continue;
    const float v = acc[rb][r];
    const long long co = (long long)m * a.ldc + n;
    if (a.splitk > 1 || a.kseg) {
      atomicAdd(C + co, v);
      continue;
    }
    switch (a.epi) {
      case EPI_NONE: C[co] = v; break;
      case EPI_BIAS: C[co] = v + bn; break;
      case EPI_TANH: C[co] = tanhf(v); break;
      case EPI_CROSS: {
        float rs = 1.f;
        if (a.rowscale) {
          rs = 0.f;
          for (int e = 0; e < a.nrs; ++e) rs += a.rowscale[(long long)m * a.nrs + e];
        }
        const float lin = v + bn * rs;
        C[co] = R1[(long long)m * a.ldr1 + n] + R2[(long long)m * a.ldr2 + n] * lin;
        if (C2) C2[(long long)m * a.ldc2 + n] = lin;
        break;
      }
      case EPI_ADD: {
        float t = R1[(long long)m * a.ldr1 + n] + v + (R2 ? R2[(long long)m * a.ldr2 + n] : 0.f);
        if (a.rowscale && a.bias)   // + a rank-nrs term: sum_e rowscale[m,e] * bias[e,n]  (the gate's share of dx_l)
          for (int e = 0; e < a.nrs; ++e) t += a.rowscale[(long long)m * a.nrs + e] * a.bias[(long long)e * a.N + n];
        C[co] = t;
        break;
      }
      case EPI_MUL_DTANH: {
        const float h = R1[(long long)m * a.ldr1 + n];
        C[co] = v * (1.f - h * h);
        break;
      }
      case EPI_TANH_GATE: {
        const float h = tanhf(v);
        C[co] = h;
        C2[(long long)m * a.ldc2 + n] = h * a.rowscale[(long long)m * a.nrs + z];
        break;
      }
      case EPI_ACCUM: C[co] += v; break;
    }
  }
}

template <int TA, int BT, int BMT>
__global__ __launch_bounds__(256) void k_gemm_f32(GemmArgs a) {
  gemm_workgroup<TA, BT, BMT>(a, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

// Several INDEPENDENT problems of one operand layout in one launch (the weight gradients of a CrossNet backward: each
// is a handful of 64x64 tiles with a 4096-long reduction — alone, none fills the chip even with split-K).  The launch is
// one-dimensional; problem j owns the workgroup ids [end[j-1], end[j]).
constexpr int kMaxProblems = 16;
struct SlimProblem {      // what varies between the problems of one launch; the rest of GemmArgs is a plain product
  const float *A, *B;
  float *C;
  int M, N, K, lda, ldb, ldc;
  long long sA, sB, sC;
  int splitk, epi, alignedA, alignedB, gx, gy, end, pad_;
};
struct MultiArgs {
  int n, total;
  SlimProblem p[kMaxProblems];
  // riders (k_gemm_tn_multi only): workgroups past the product's grid that touch the table rows of the NEXT batch's lookup
  // (prefetch_rows.hpp) — the weight-gradient launch is MFMA-bound and its workgroups leave slots on every CU
  int ride_blocks;
  mi::PrefetchJob pf;
};
template <int TA, int BT>
__global__ __launch_bounds__(256) void k_gemm_f32_multi(MultiArgs m) {
  // Workgroups are dealt to the 8 XCDs round-robin in launch order and each XCD has its own L2.  Logical ids run
  // (slice, row tile, column tile) with the column tile fastest: neighbours share the A slab of their K-slice — so every XCD
  // gets a contiguous run of logical ids (the grid is rounded up to a multiple of 8; the padding ids leave at once).
  const int per_xcd = (m.total + 7) >> 3;
  const int id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || id >= m.total) return;
  int j = 0;
  while (j + 1 < m.n && id >= m.p[j].end) ++j;
  const SlimProblem q = m.p[j];
  const int local = id - (j ? m.p[j - 1].end : 0);
  const int per = q.gx * q.gy;
  const int wz = local / per, rem = local - wz * per;
  GemmArgs a;
  a.A = q.A; a.B = q.B; a.C = q.C;
  a.M = q.M; a.N = q.N; a.K = q.K;
  a.lda = q.lda; a.ldb = q.ldb; a.ldc = q.ldc;
  a.transA = TA; a.transB = !BT;
  a.sA = q.sA; a.sB = q.sB; a.sC = q.sC;
  a.kgroups = 1; a.gA = 0; a.gB = 0;
  a.epi = q.epi; a.bias = nullptr;
  a.R1 = a.R2 = nullptr; a.ldr1 = a.ldr2 = 0; a.sR1 = a.sR2 = 0;
  a.rowscale = nullptr; a.nrs = 0;
  a.C2 = nullptr; a.ldc2 = 0; a.sC2 = 0;
  a.alignedA = q.alignedA; a.alignedB = q.alignedB;
  a.splitk = q.splitk;
  a.mtile_b = nullptr; a.kseg = nullptr;
  a.xcd_swizzle = 0;
  gemm_workgroup<TA, BT, 64>(a, rem % q.gx, rem / q.gx, wz, q.gx, q.gy);
}

// ---- A^T . B with BOTH operands as they lie in memory, moved by LDS-DMA ---------------------------------------------------
// The weight gradients (dW = dz^T . a: reduction over the batch, both operands [k][row] with the row contiguous) need no
// transposition at all when the LDS tile keeps the memory layout: lane (i, kh) of a 32x32x2 MFMA wants A[k = kh][m = i],
// which in a [k][64 rows] tile is 32 consecutive floats of one k-row per half-wave — a conflict-free ds_read_b32.  So the
// tiles go global -> LDS with global_load_lds_dwordx4 (16 B per lane, 1 KiB = four k-rows per wave-instruction, lane-linear
// on the LDS side, which IS this layout): no VGPR staging, no ds_write, no VALU in the loop except the pointer bumps.
// (k_gemm_f32_multi<1,1> pays 16 scalar transposing ds_write_b32 + their address math per thread and k-tile.)
// Two slots of (A tile, B tile) = 32 KB per workgroup: five workgroups fit a CU, which matters more than a deeper ring —
// with 4 slots (two workgroups per CU) the same loop took 78 us instead of 58 for the C2 tail's gradients, with or
// without the DMA: a lone wave per SIMD cannot keep the MFMA pipe fed.  Every wave issues its quarter of a slice (2 + 2
// instructions) and counts its own DMA by hand (s_waitcnt vmcnt), one barrier per slice.  Waves whose 32x32 block lies
// outside the matrix skip the MFMAs (M = 400 is 12.5 blocks: the 64-row tiles' second half is empty there).
// Knock-outs at the C2 tail's shape (4 slices per tile, 588 workgroups): full 58 us; without the epilogue's atomics 58;
// LDS reads replaced by registers 58; two accumulators 58; DMA + barriers + atomics without MFMAs 24; the MFMAs alone (no
// DMA, barriers, LDS) 45 — the product is bound by how evenly the MFMA work lands on the 1024 SIMDs, which is the
// K-slice count's business (auto_splitk below: 5 slices, 49 us).  A 16-wave workgroup on 128 x 128 tiles with one
// workgroup per CU was built and measured too: 76-95 us (16-wave barriers and a DMA round trip per k-tile on the critical
// path of sparse edge tiles); removed.
// Needs: K % 32 == 0, M % 4 == 0, N % 4 == 0, 16-byte aligned operands (else the general kernel).
constexpr int TN_S = 2, TN_TILE = BK * 64;

__device__ __forceinline__ void tn_dma(const float *src, float *lds_dst) {
  // (inline asm, not __builtin_amdgcn_global_load_lds: after the builtin hipcc treats every later LDS read as possibly
  //  out of order with it and waits lgkmcnt(0) in front of each MFMA.  The asm writes M0, which hipcc does not let an asm
  //  declare as clobbered ("reserved register"), so the block SAVES and RESTORES it itself: M0 is read when the DMA
  //  issues, the restore right behind the issue is safe, and code the compiler may one day place around this — LDS-direct
  //  loads, indexed registers — finds M0 as it left it.  Two scalar moves per DMA; the loop is not SALU-bound.)
  const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds_dst);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(off) : "memory");
}

__global__ __launch_bounds__(256) void k_gemm_tn_multi(MultiArgs m) {
  __shared__ __attribute__((aligned(16))) float tn_lds[TN_S * 2 * TN_TILE];
  const int per_xcd = (m.total + 7) >> 3;
  if ((int)blockIdx.x >= 8 * per_xcd) {      // (riders sit at the END of the grid: the product's workgroups are dispatched first)
    mi::prefetch_rows_blocks(m.pf, (int)blockIdx.x - 8 * per_xcd, m.ride_blocks, nullptr);
    return;
  }
  const int id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || id >= m.total) return;
  int j = 0;
  while (j + 1 < m.n && id >= m.p[j].end) ++j;
  const SlimProblem &q = m.p[j];
  const int local = id - (j ? m.p[j - 1].end : 0);
  const int pertile = q.gx * q.gy;
  const int wz = local / pertile, rem = local - wz * pertile;
  const int bx = rem % q.gx, by = rem / q.gx;
  const int z = wz / q.splitk, slice = wz % q.splitk;
  const int ktiles = q.K / BK;
  const int per = (ktiles + q.splitk - 1) / q.splitk;
  const int first = slice * per;
  const int nst = min(ktiles, first + per) - first;
  if (nst <= 0) return;

  const int m0 = by * 64, n0 = bx * BN;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // (block of wave w rotated by the workgroup's id: the empty blocks of edge tiles do not always idle the same SIMDs)
  const int wb = (w + id) & 3;
  const int wr = wb >> 1, wc = wb & 1, i = lane & 31, kh = lane >> 5;
  const bool active = (m0 + wr * 32 < q.M) && (n0 + wc * 32 < q.N);

  // this lane's share of a slice: k-rows 8w + (lane >> 4) and + 4 of both tiles, float4 column lane & 15 (column groups
  // past the matrix re-read its last group: they only feed outputs that are never stored)
  const int dr = 8 * w + (lane >> 4), dc = (lane & 15) * 4;
  const int ca = min(m0 + dc, q.M - 4), cb = min(n0 + dc, q.N - 4);
  const float *pa = q.A + z * q.sA + (long long)(first * BK + dr) * q.lda + ca;
  const float *pb = q.B + z * q.sB + (long long)(first * BK + dr) * q.ldb + cb;
  const long long a4 = 4ll * q.lda, b4 = 4ll * q.ldb, astep = (long long)BK * q.lda, bstep = (long long)BK * q.ldb;
  auto issue = [&](int s) {
    float *sa = tn_lds + (s % TN_S) * (2 * TN_TILE) + w * 512;
    tn_dma(pa, sa);
    tn_dma(pa + a4, sa + 256);
    tn_dma(pb, sa + TN_TILE);
    tn_dma(pb + b4, sa + TN_TILE + 256);
    pa += astep;
    pb += bstep;
  };

  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  issue(0);
  const int fa = kh * 64 + wr * 32 + i, fb = TN_TILE + kh * 64 + wc * 32 + i;
  for (int s = 0; s < nst; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of slice s has landed
    __builtin_amdgcn_s_barrier();                          // everybody's has; everybody is done with the other slot
    if (s + 1 < nst) issue(s + 1);
    if (active) {
      const float *t = tn_lds + (s % TN_S) * (2 * TN_TILE);
#pragma unroll
      for (int c = 0; c < BK / 2; ++c)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t[fa + c * 128], t[fb + c * 128], acc, 0, 0, 0);
      // order: the fragment reads of MFMA pair p + 1 go out before the MFMAs of pair p (the scheduler's own order is
      // read, wait, MFMA, MFMA on one register set)
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int p = 0; p < BK / 4 - 2; ++p) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
  }
  if (!active) return;

  // ---- epilogue: lane holds column n, rows (reg & 3) + 8 (reg >> 2) + 4 kh of its 32x32 block
  const int n = n0 + wc * 32 + i;
  if (n >= q.N) return;
  float *C = q.C + z * q.sC;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mm = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
    if (mm >= q.M) continue;
    float *c = C + (long long)mm * q.ldc + n;
    if (q.splitk > 1) atomicAdd(c, acc[r]);
    else if (q.epi == EPI_ACCUM) *c += acc[r];
    else *c = acc[r];
  }
}

}  // namespace

extern "C" {

int mi_gemm_f32(const float *A, const float *B, float *C, int32_t M, int32_t N, int32_t K, int32_t lda,
                int32_t ldb, int32_t ldc, int32_t transA, int32_t transB, int32_t batch, int64_t sA, int64_t sB,
                int64_t sC, int32_t kgroups, int64_t gA, int64_t gB, int32_t epi, const float *bias,
                const float *R1, int32_t ldr1, int64_t sR1, const float *R2, int32_t ldr2, int64_t sR2,
                const float *rowscale, int32_t nrs, float *C2, int32_t ldc2, int64_t sC2, int32_t splitk,
                void *stream) {
  if (M < 0 || N < 0 || K < 0 || batch < 0 || kgroups < 1) return MI_ERR_INVALID_ARG;
  if (epi < EPI_NONE || epi > EPI_ACCUM) return MI_ERR_INVALID_ARG;
  if (M == 0 || N == 0 || batch == 0) return MI_OK;
  if (!A || !B || !C) return MI_ERR_INVALID_ARG;
  if ((epi == EPI_CROSS && (!R1 || !R2)) || ((epi == EPI_ADD || epi == EPI_MUL_DTANH) && !R1) ||
      (epi == EPI_TANH_GATE && (!rowscale || !C2 || nrs < 1)))
    return MI_ERR_INVALID_ARG;
  if (splitk < 1) return MI_ERR_INVALID_ARG;
  if (splitk > 1 && epi != EPI_NONE && epi != EPI_ACCUM) return MI_ERR_INVALID_ARG;  // C must be pre-zeroed / accumulated
  if ((long long)batch * splitk > 65535) return MI_ERR_UNSUPPORTED;
  GemmArgs a;
  a.splitk = splitk;
  a.A = A; a.B = B; a.C = C;
  a.M = M; a.N = N; a.K = K;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.transA = transA ? 1 : 0; a.transB = transB ? 1 : 0;
  a.sA = sA; a.sB = sB; a.sC = sC;
  a.kgroups = kgroups; a.gA = gA; a.gB = gB;
  a.epi = epi; a.bias = bias;
  a.R1 = R1; a.R2 = R2; a.ldr1 = ldr1; a.ldr2 = ldr2; a.sR1 = sR1; a.sR2 = sR2;
  a.rowscale = rowscale; a.nrs = nrs;
  a.C2 = C2; a.ldc2 = ldc2; a.sC2 = sC2;
  a.mtile_b = nullptr;
  a.kseg = nullptr;
  a.xcd_swizzle = 1;
  a.alignedA = aligned16(A) && (lda % 4 == 0) && (sA % 4 == 0) && (gA % 4 == 0);
  a.alignedB = aligned16(B) && (ldb % 4 == 0) && (sB % 4 == 0) && (gB % 4 == 0);
  // 128-row tiles only where they still leave >= 4 workgroups per CU (8192^3: 119 -> 125 TFLOP/s).  At the tail /
  // CrossNet shapes (M = 4096, N ~ 400) they are SLOWER (26 vs 23 us): 224 workgroups = one per CU leaves nothing to
  // overlap a workgroup's barrier and LDS phases with, while 448 64-row workgroups run 1.75 per CU.
  const long long tiles128 = (long long)((N + BN - 1) / BN) * ((M + 127) / 128) * batch;
  const bool tall = splitk == 1 && tiles128 >= 1024;
  const int bm = tall ? 128 : 64;
  dim3 grid((N + BN - 1) / BN, (M + bm - 1) / bm, batch * splitk);
  if (grid.y > 65535) return MI_ERR_UNSUPPORTED;
  hipEvent_t ea, eb;
  const bool prof = mi::prof_acquire("gemm_f32", &ea, &eb);
#define GO(TA, BT)                                                                                      \
  do {                                                                                                  \
    if (tall) {                                                                                         \
      if (prof) hipExtLaunchKernelGGL((k_gemm_f32<TA, BT, 128>), grid, dim3(256), 0, (hipStream_t)stream, ea, eb, 0, a); \
      else hipLaunchKernelGGL((k_gemm_f32<TA, BT, 128>), grid, dim3(256), 0, (hipStream_t)stream, a);   \
    } else {                                                                                            \
      if (prof) hipExtLaunchKernelGGL((k_gemm_f32<TA, BT, 64>), grid, dim3(256), 0, (hipStream_t)stream, ea, eb, 0, a); \
      else hipLaunchKernelGGL((k_gemm_f32<TA, BT, 64>), grid, dim3(256), 0, (hipStream_t)stream, a);    \
    }                                                                                                   \
  } while (0)
  if (a.transA) { if (a.transB) GO(1, 0); else GO(1, 1); }
  else { if (a.transB) GO(0, 0); else GO(0, 1); }
#undef GO
  return launch_status();
}

// ---- grouped forms used by the TT-Rec levels (tt_grouped.hip) ------------------------------------
static int launch_grouped(GemmArgs &a, dim3 grid, const char *name, void *stream) {
  hipEvent_t ea, eb;
  const bool prof = mi::prof_acquire(name, &ea, &eb);
#define GO(TA, BT)                                                                                      \
  do {                                                                                                  \
    if (prof) hipExtLaunchKernelGGL((k_gemm_f32<TA, BT, 64>), grid, dim3(256), 0, (hipStream_t)stream, ea, eb, 0, a); \
    else hipLaunchKernelGGL((k_gemm_f32<TA, BT, 64>), grid, dim3(256), 0, (hipStream_t)stream, a);      \
  } while (0)
  if (a.transA) { if (a.transB) GO(1, 0); else GO(1, 1); }
  else { if (a.transB) GO(0, 0); else GO(0, 1); }
#undef GO
  return launch_status();
}

static void plain_args(GemmArgs &a, const float *A, const float *B, float *C, int M, int N, int K, int lda,
                       int ldb, int ldc, int transA, int transB) {
  a.splitk = 1;
  a.A = A; a.B = B; a.C = C;
  a.M = M; a.N = N; a.K = K;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.transA = transA ? 1 : 0; a.transB = transB ? 1 : 0;
  a.sA = 0; a.sB = 0; a.sC = 0;
  a.kgroups = 1; a.gA = 0; a.gB = 0;
  a.epi = EPI_NONE; a.bias = nullptr;
  a.R1 = a.R2 = nullptr; a.ldr1 = a.ldr2 = 0; a.sR1 = a.sR2 = 0;
  a.rowscale = nullptr; a.nrs = 0;
  a.C2 = nullptr; a.ldc2 = 0; a.sC2 = 0;
  a.mtile_b = nullptr; a.kseg = nullptr;
  a.xcd_swizzle = 0;
}

static int cu_count() {
  static const int n = []() {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  return n;
}

// K-slices the library picks for the 64 x 64-tile kernels when a problem leaves splitk = 0.  All workgroups of the launch
// are resident together (a few per CU) and a CU is busy for (workgroups it got) x (k-tiles per slice): the cut minimises
//   ceil(tiles * s / CUs) * ceil(ktiles / s) * (1 + 0.035 s)
// — the last factor for the atomics every extra slice adds (a slice adds a whole 64 x 64 tile with float atomics).  A
// slice keeps >= 4 k-tiles of 32.  Round 2 used round(640 / tiles): for the C2 tail (147 tiles, 128 k-tiles) that is 4
// slices = 588 workgroups = 2.3 per CU, i.e. 3 on most CUs (cost 96); 5 slices put 2.9 on every CU (cost 78).
static int auto_splitk(long long launch_tiles, int K, int batch) {
  const int ktiles = (K + BK - 1) / BK, ncu = cu_count();
  long long top = ktiles / 4;
  if (top > 16) top = 16;
  if (top > 65535 / (batch > 0 ? batch : 1)) top = 65535 / (batch > 0 ? batch : 1);
  int best = 1;
  double best_cost = 0;
  for (int sk = 1; sk <= (top < 1 ? 1 : top); ++sk) {
    const double cost = (double)((launch_tiles * sk + ncu - 1) / ncu) * ((ktiles + sk - 1) / sk) * (1.0 + 0.035 * sk);
    if (sk == 1 || cost < best_cost) { best = sk; best_cost = cost; }
  }
  return best;
}

static bool tn_enabled() {      // MI_GEMM_TN_DMA=0: the reduction-major form stays on the general kernel
  static const bool v = []() { const char *e = getenv("MI_GEMM_TN_DMA"); return !(e && e[0] == '0'); }();
  return v;
}

// Validates and cuts a launch: fills m (splitk chosen where a problem leaves it 0) and says whether the LDS-DMA kernel takes it.
static int plan_multi(const mi_gemm_problem *probs, int n, int transA, int transB, MultiArgs &m, bool *tn) {
  m.n = 0;
  m.total = 0;
  *tn = false;
  if (n < 0 || n > kMaxProblems) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!probs) return MI_ERR_INVALID_ARG;
  long long launch_tiles = 0;
  for (int j = 0; j < n; ++j) {
    const mi_gemm_problem &q = probs[j];
    if (q.M < 0 || q.N < 0 || q.K < 0 || q.batch < 0 || q.splitk < 0) return MI_ERR_INVALID_ARG;
    if (q.M == 0 || q.N == 0 || q.batch == 0) continue;
    if (!q.A || !q.B || !q.C) return MI_ERR_INVALID_ARG;
    launch_tiles += (long long)((q.N + BN - 1) / BN) * ((q.M + 63) / 64) * q.batch;
  }
  long long wgs = 0;
  bool fits = tn_enabled() && transA && !transB;
  for (int j = 0; j < n; ++j) {
    const mi_gemm_problem &q = probs[j];
    if (q.M == 0 || q.N == 0 || q.batch == 0) continue;
    SlimProblem &a = m.p[m.n];
    a.A = q.A; a.B = q.B; a.C = q.C;
    a.M = q.M; a.N = q.N; a.K = q.K;
    a.lda = q.lda; a.ldb = q.ldb; a.ldc = q.ldc;
    a.sA = q.sA; a.sB = q.sB; a.sC = q.sC;
    a.splitk = q.splitk > 0 ? q.splitk : auto_splitk(launch_tiles, q.K, q.batch);
    if ((long long)q.batch * a.splitk > 65535) return MI_ERR_UNSUPPORTED;
    a.epi = q.accumulate ? EPI_ACCUM : EPI_NONE;
    a.alignedA = aligned16(q.A) && (q.lda % 4 == 0) && (q.sA % 4 == 0);
    a.alignedB = aligned16(q.B) && (q.ldb % 4 == 0) && (q.sB % 4 == 0);
    a.gx = (q.N + BN - 1) / BN;
    a.gy = (q.M + 63) / 64;
    a.pad_ = 0;
    fits = fits && a.alignedA && a.alignedB && a.K >= BK && a.K % BK == 0 && a.M >= 4 && a.M % 4 == 0 && a.N >= 4 && a.N % 4 == 0;
    wgs += (long long)a.gx * a.gy * q.batch * a.splitk;
    if (wgs > 0x7fffffffLL) return MI_ERR_UNSUPPORTED;
    a.end = (int)wgs;
    ++m.n;
  }
  if (m.n == 0) return MI_OK;
  for (int j = m.n; j < kMaxProblems; ++j) m.p[j] = m.p[m.n - 1];
  m.total = (int)wgs;
  *tn = fits;
  return MI_OK;
}

// How mi_gemm_f32_multi would cut this launch (host arithmetic only, nothing is launched): *kind = 1 the LDS-DMA kernel of
// the reduction-major form, 0 the general kernel; splitk[j] = K-slices of problem j (0 for an empty one).
int mi_gemm_f32_multi_plan(const mi_gemm_problem *probs, int32_t n, int32_t transA, int32_t transB, int32_t *kind,
                           int64_t *workgroups, int32_t *splitk) {
  if (!kind || !workgroups || (n > 0 && !splitk)) return MI_ERR_INVALID_ARG;
  MultiArgs m;
  bool tn;
  const int rc = plan_multi(probs, n, transA, transB, m, &tn);
  if (rc != MI_OK) return rc;
  *kind = tn ? 1 : 0;
  *workgroups = m.total;
  for (int j = 0, k = 0; j < n; ++j) {
    const bool empty = probs[j].M == 0 || probs[j].N == 0 || probs[j].batch == 0;
    splitk[j] = empty ? 0 : m.p[k++].splitk;
  }
  return MI_OK;
}

// n <= 16 independent problems C_j = opA(A_j) opB(B_j) (+ C_j when accumulate), all with the same transposes, in ONE launch.
// split-K slices meet in float atomics: a problem with splitk != 1 needs its C zeroed by the caller (or accumulate);
// splitk = 0 leaves the cut to the library.
int mi_gemm_f32_multi(const mi_gemm_problem *probs, int32_t n, int32_t transA, int32_t transB, void *stream) {
  return mi_gemm_f32_multi_ride(probs, n, transA, transB, nullptr, stream);
}

int mi_gemm_f32_multi_ride(const mi_gemm_problem *probs, int32_t n, int32_t transA, int32_t transB,
                           const mi_prefetch_rows_job *ride, void *stream) {
  MultiArgs m;
  bool tn;
  const int rc = plan_multi(probs, n, transA, transB, m, &tn);
  if (rc != MI_OK) return rc;
  m.ride_blocks = 0;
  bool riding = false;
  if (ride && ride->B > 0) {
    riding = mi::prefetch_job(ride->idx, ride->offsets, ride->W, ride->ldw, ride->w1, ride->ldw1, ride->B, ride->F, ride->N, m.pf);
    if (!riding) return MI_ERR_INVALID_ARG;
  }
  if (riding && (!tn || m.n == 0)) {      // no launch that can carry it: the job as a launch of its own
    const int st = mi_prefetch_rows(ride->idx, ride->offsets, ride->W, ride->ldw, ride->w1, ride->ldw1, ride->B, ride->F, ride->N, stream);
    if (st != MI_OK) return st;
    riding = false;
  }
  if (m.n == 0) return MI_OK;
  hipEvent_t ea, eb;
  if (riding) {      // four rows per rider thread
    int64_t rb = (m.pf.n + 4 * 256 - 1) / (4 * 256);
    m.ride_blocks = (int)(rb > 1024 ? 1024 : rb);
  }
  const dim3 grid((unsigned)((m.total + 7) / 8 * 8 + m.ride_blocks));
  if (tn) {      // A^T B, both operands reduction-major, every problem of the launch fits: the LDS-DMA kernel
    const bool prof = mi::prof_acquire("gemm_tn_multi", &ea, &eb);
    if (prof) hipExtLaunchKernelGGL(k_gemm_tn_multi, grid, dim3(256), 0, (hipStream_t)stream, ea, eb, 0, m);
    else hipLaunchKernelGGL(k_gemm_tn_multi, grid, dim3(256), 0, (hipStream_t)stream, m);
    return launch_status();
  }
  const bool prof = mi::prof_acquire("gemm_f32_multi", &ea, &eb);
#define GO(TA, BT)                                                                                            \
  do {                                                                                                        \
    if (prof) hipExtLaunchKernelGGL((k_gemm_f32_multi<TA, BT>), grid, dim3(256), 0, (hipStream_t)stream, ea, eb, 0, m); \
    else hipLaunchKernelGGL((k_gemm_f32_multi<TA, BT>), grid, dim3(256), 0, (hipStream_t)stream, m);            \
  } while (0)
  if (transA) { if (transB) GO(1, 0); else GO(1, 1); }
  else { if (transB) GO(0, 0); else GO(0, 1); }
#undef GO
  return launch_status();
}

int mi_gemm_f32_row_groups(const float *A, const float *B, float *C, int32_t M, int32_t N, int32_t K,
                           int32_t lda, int32_t ldb, int32_t ldc, int32_t transB, int64_t sB,
                           const int32_t *mtile_b, void *stream) {
  if (M < 0 || N < 0 || K < 0) return MI_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return MI_OK;
  if (!A || !B || !C || !mtile_b) return MI_ERR_INVALID_ARG;
  GemmArgs a;
  plain_args(a, A, B, C, M, N, K, lda, ldb, ldc, 0, transB);
  a.sB = sB;
  a.mtile_b = mtile_b;
  a.alignedA = aligned16(A) && (lda % 4 == 0);
  a.alignedB = aligned16(B) && (ldb % 4 == 0) && (sB % 4 == 0);
  dim3 grid((N + BN - 1) / BN, (M + 63) / 64, 1);
  if (grid.y > 65535) return MI_ERR_UNSUPPORTED;
  return launch_grouped(a, grid, "gemm_row_groups", stream);
}

int mi_gemm_f32_k_groups(const float *A, const float *B, float *C, int32_t M, int32_t N, int32_t lda,
                         int32_t ldb, int32_t ldc, int64_t sC, const int64_t *kseg, int32_t nseg,
                         void *stream) {
  if (M < 0 || N < 0 || nseg < 0) return MI_ERR_INVALID_ARG;
  if (M == 0 || N == 0 || nseg == 0) return MI_OK;
  if (!A || !B || !C || !kseg) return MI_ERR_INVALID_ARG;
  if (nseg > 65535) return MI_ERR_UNSUPPORTED;
  GemmArgs a;
  plain_args(a, A, B, C, M, N, 0, lda, ldb, ldc, 1, 0);   // C[slice] += A[rows,:M]^T . B[rows,:N]
  a.sC = sC;
  a.kseg = reinterpret_cast<const long long *>(kseg);
  a.alignedA = aligned16(A) && (lda % 4 == 0);
  a.alignedB = aligned16(B) && (ldb % 4 == 0);
  dim3 grid((N + BN - 1) / BN, (M + 63) / 64, nseg);
  if (grid.y > 65535) return MI_ERR_UNSUPPORTED;
  return launch_grouped(a, grid, "gemm_k_groups", stream);
}

}  // extern "C"
