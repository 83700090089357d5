// tail_gemm.hpp — the fp32 contractions of the MLP tail (reference src/models/deepfm.py:53-66,
// src/models/dcn.py:56-66) on the gfx950 matrix cores, with the BatchNorm / ReLU / Dropout element work folded into the
// operand loads and the epilogues so that no activation makes a second trip through HBM.
//
//   C[r][c] = sum_red  R(r, red) * Cc(c, red)          v_mfma_f32_16x16x4_f32 (exact fp32)
//
// Shape-driven tiling.  The tail's products are M = 4096 by N,K in {400, 416}: 64 x 64 tiles give 448 workgroups = 1.75
// waves over the 256 CUs (what the library GEMM and round 1's generic kernel pay for).  Here a workgroup owns 64 rows of
// R (4 row blocks of 16) by ONE QUARTER of the columns (100 or 104, computed as 7 sub-tiles of 16 with the surplus
// masked): exactly 64 x 4 = 256 workgroups, one per CU; 8 waves, two per SIMD, 28 accumulator registers per lane.
//
// MFMA roles are swapped (A := Cc fragment, B := R fragment) so that a lane ends up with 4 CONSECUTIVE columns of one
// output row: the epilogue stores float4 and the column reductions (BatchNorm statistics, dgamma / dbeta) are 16-lane
// shuffles.
//
// Operand sources, both staged through LDS (register-staged, transform applied between the global load and the LDS
// write; the loads of tile i+1 are in flight during the MFMAs of tile i):
//   KC  "reduction contiguous": global [row][red]; LDS [row][32] with the eight 16-B chunks of a row XOR-swizzled by
//       (row >> 1) & 7 — conflict-free ds_read_b128 fragments for the 16x16x4 lane pattern, no padding.
//   OC  "output contiguous":   global [red][out]; LDS [32][out] (row pitch = 4 mod 8 floats), ds_read_b32 fragments.
// One b128 (or four b32) per operand feeds 4 MFMA k-steps; lane group g = lane / 16 supplies reduction index
// 16h + 4g + j at step (h, j) for BOTH operands, which a sum over the reduction index does not see.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tg {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 512;     // 8 waves: 4 consumers (MFMA, one per SIMD) + 4 producers (load / transform / LDS write)
constexpr int kProd = 256;        // producer threads; their index is threadIdx.x - 256
constexpr int kWaveLanes = 64;
constexpr int BM = 64;            // rows of R per workgroup (4 waves x 16)
constexpr int BK = 32;            // reduction indices per stage
constexpr int NSUB = 7;           // 16-column sub-tiles per wave
constexpr int BNT = NSUB * 16;    // 112 >= columns per workgroup
constexpr int SR_OC = 68;         // LDS row pitch (floats) of an OC tile of R (64 wide)
constexpr int SC_OC = 116;        // ... of Cc (112 wide); both = 4 mod 8: lane groups 4 rows apart fall 16 banks apart

__device__ __forceinline__ float4 vld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
// per-column constants may live in LDS (merged there from the producing layer's tile statistics in the kernel's prologue,
// tail.hip): an address-space-3 pointer keeps their loads ds_read_b128 — a generic pointer would make them flat loads,
// which wait on BOTH counters and would drain the producers' global loads at every slice
typedef const __attribute__((address_space(3))) float *lds_cfp;
__device__ __forceinline__ float4 vld4(lds_cfp p) {
  const floatx4 v = *reinterpret_cast<const __attribute__((address_space(3))) floatx4 *>(p);
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void vst4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ---- dropout --------------------------------------------------------------------------------------------------------
// The keep decisions of a layer are one BIT per element, written once per step by k_tail_dropmask (tail.hip) from a
// counter-based hash: 16 bits of splitmix64(seed + salt', (m*ld + c) / 4) per feature, keep when >= round(p * 65536).
// An integer hash inside the operand loads would be paid by the matrix pipe: the f32 MFMA runs at the vector-ALU rate
// and shares its issue with the producers' VALU work, and the 64-bit multiplies of the hash are quarter-rate (measured:
// +4 us on a 17 us product, and the same tile is loaded by 4 to 7 workgroups).  Element (m, c) of an activation with
// row pitch ld is bit (c & 7) of byte (m*ld + c) >> 3; a float4 of features reads one byte and uses one nibble.
__device__ __forceinline__ uint64_t mix64(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
struct Drop {
  const uint8_t *bits;   // nullptr = no dropout
  float inv_keep;        // 1 / (1 - p)
  int ld;                // features per row of the activation the mask belongs to (multiple of 8)
  __device__ __forceinline__ uint32_t fetch(int m, int c) const {
    return bits ? (uint32_t)bits[((int64_t)m * ld + c) >> 3] : 0xFFu;
  }
  // multipliers (0 or inv_keep) of the 4 consecutive features starting at column c (c % 4 == 0), from the fetched byte
  __device__ __forceinline__ float4 scale4(uint32_t byte, int c) const {
    const uint32_t nib = byte >> (c & 4);
    float4 s;
    s.x = (nib & 1u) ? inv_keep : 0.f;
    s.y = (nib & 2u) ? inv_keep : 0.f;
    s.z = (nib & 4u) ? inv_keep : 0.f;
    s.w = (nib & 8u) ? inv_keep : 0.f;
    return s;
  }
};

// ---- operand loaders ------------------------------------------------------------------------------------------------
// Every loader is split in two so that nothing computed from a loaded value sits between a global load and the MFMA
// phase it is meant to hide under:  fetch(m, c) only ISSUES loads (raw vectors + the per-column constants),
// finish(raw, consts, m, c) turns them into the operand value right before the LDS write, one or two MFMA phases later.
// (m, c) = row and first of 4 consecutive features of a [M, ld] matrix.
struct NoConsts {};
// LEAN addressing (round 3).  The f32 MFMA runs on the vector ALU's rate and every VALU instruction of a producer wave is
// paid by the matrix pipe of its SIMD (measured: +64 VALU per producer phase = +12 us per product; SQ_INSTS_VALU 3.06 M
// against 0.75 M MFMAs in the fused forward product, profiles/r03_tail_sq_counters.csv).  A producer phase of round 2
// spent ~94 vector instructions on addresses, clamps and selects per slice.  Lean form: an operand element's address is
//   UNIFORM base (pointers shifted to the tile's first row and the slice's first reduction index: scalar registers, SALU)
//   + a per-thread 32-bit byte offset that never changes (row inside the tile x pitch + column inside the slice),
// so a load is `global_load_dwordx4 v, v_off, s[base]` with no vector arithmetic at all; clamps of rows beyond the tile
// happen once, in the offsets; the zeroing of reduction indices beyond the end exists only in the LAST slice, which takes
// the general path (uniform branch).  Each loader L provides:
//   UBase ubase(row, col)            the uniform part for a tile whose element (0, 0) is (row, col); with dropout
//                                    (row * ld + col) % 8 == 0 is required (callers pass col = 0 or a multiple of 32)
//   TC    tconst(r, c)               the thread-constant part of element (r, c) relative to it
//   Raw   fetch_u(ub, tc)            issue the loads
//   Consts consts_u(ub, c)           per-column constants of column c relative to ub (KC operands: c = the thread's column
//                                    inside the slice)
//   float4 finish_u(raw, k, ub, tc)  form the operand value
__device__ __forceinline__ float4 vld4u(const float *base, uint32_t boff) {
  return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + boff);
}
// Plain: the matrix itself (weights, the embedding block, a finished gradient).
struct LoadPlain {
  static constexpr bool kLateConsts = false;
  static constexpr bool kLean = true;
  const float *P;
  int ld;
  struct Raw { float4 a; };
  typedef NoConsts Consts;
  struct UBase { const float *p; };
  struct TC { uint32_t off; };
  __device__ __forceinline__ UBase ubase(int row, int col) const { return UBase{P + (int64_t)row * ld + col}; }
  __device__ __forceinline__ TC tconst(int r, int c) const { return TC{(uint32_t)(r * ld + c) * 4u}; }
  __device__ __forceinline__ Raw fetch_u(const UBase &u, const TC &t) const { return Raw{vld4u(u.p, t.off)}; }
  __device__ __forceinline__ Consts consts_u(const UBase &, int) const { return Consts{}; }
  __device__ __forceinline__ float4 finish_u(const Raw &r, const Consts &, const UBase &, const TC &) const { return r.a; }
  __device__ __forceinline__ Raw fetch(int m, int c) const { return Raw{vld4(P + (int64_t)m * ld + c)}; }
  __device__ __forceinline__ Consts consts(int) const { return Consts{}; }
  __device__ __forceinline__ float4 finish(const Raw &r, const Consts &, int, int) const { return r.a; }
};
// Act: the activation a = dropout(relu(bn(z))) recomputed from the saved pre-activation z:
//   pre = (z - mu[c]) * sc[c] + be[c]   (sc = gamma * rstd; BatchNorm off: mu = 0, sc = 1, be = bias or 0)
template <class CP> struct ConstsTrait { static constexpr bool kLate = false; };
template <> struct ConstsTrait<lds_cfp> { static constexpr bool kLate = true; };   // LDS constants: read when the value is formed
template <class CP>
struct LoadActT {
  static constexpr bool kLateConsts = ConstsTrait<CP>::kLate;
  static constexpr bool kLean = true;
  const float *Z;
  int ld;
  CP mu, sc, be;
  Drop drop;
  struct Raw { float4 z; uint32_t keep; };
  struct Consts { float4 u, s, b; };
  struct UBase { const float *z; const uint8_t *bits; CP mu, sc, be; };
  struct TC { uint32_t off, koff, nib; };     // byte offset of z; byte offset and nibble shift (0 / 4) of the keep bits
  __device__ __forceinline__ UBase ubase(int row, int col) const {
    const int64_t e = (int64_t)row * ld + col;
    return UBase{Z + e, drop.bits ? drop.bits + (e >> 3) : nullptr, mu + col, sc + col, be + col};
  }
  __device__ __forceinline__ TC tconst(int r, int c) const {
    const uint32_t e = (uint32_t)(r * ld + c);
    return TC{e * 4u, e >> 3, (uint32_t)(c & 4)};
  }
  __device__ __forceinline__ Raw fetch_u(const UBase &u, const TC &t) const {
    return Raw{vld4u(u.z, t.off), u.bits ? (uint32_t)u.bits[t.koff] : 0xFFu};
  }
  __device__ __forceinline__ Consts consts_u(const UBase &u, int c) const { return Consts{vld4(u.mu + c), vld4(u.sc + c), vld4(u.be + c)}; }
  __device__ __forceinline__ float4 finish_u(const Raw &r, const Consts &k, const UBase &, const TC &t) const {
    const uint32_t nib = r.keep >> t.nib;
    float4 a;
    a.x = (nib & 1u) ? fmaxf(fmaf(r.z.x - k.u.x, k.s.x, k.b.x), 0.f) * drop.inv_keep : 0.f;
    a.y = (nib & 2u) ? fmaxf(fmaf(r.z.y - k.u.y, k.s.y, k.b.y), 0.f) * drop.inv_keep : 0.f;
    a.z = (nib & 4u) ? fmaxf(fmaf(r.z.z - k.u.z, k.s.z, k.b.z), 0.f) * drop.inv_keep : 0.f;
    a.w = (nib & 8u) ? fmaxf(fmaf(r.z.w - k.u.w, k.s.w, k.b.w), 0.f) * drop.inv_keep : 0.f;
    return a;
  }
  __device__ __forceinline__ Raw fetch(int m, int c) const { return Raw{vld4(Z + (int64_t)m * ld + c), drop.fetch(m, c)}; }
  __device__ __forceinline__ Consts consts(int c) const { return Consts{vld4(mu + c), vld4(sc + c), vld4(be + c)}; }
  __device__ __forceinline__ float4 finish(const Raw &r, const Consts &k, int m, int c) const {
    const float4 d = drop.scale4(r.keep, c);
    float4 a;
    a.x = fmaxf(fmaf(r.z.x - k.u.x, k.s.x, k.b.x), 0.f) * d.x;
    a.y = fmaxf(fmaf(r.z.y - k.u.y, k.s.y, k.b.y), 0.f) * d.y;
    a.z = fmaxf(fmaf(r.z.z - k.u.z, k.s.z, k.b.z), 0.f) * d.z;
    a.w = fmaxf(fmaf(r.z.w - k.u.w, k.s.w, k.b.w), 0.f) * d.w;
    return a;
  }
};
typedef LoadActT<const float *> LoadAct;
typedef LoadActT<lds_cfp> LoadActL;
// Dz: the gradient w.r.t. the pre-activation of a training-mode BatchNorm layer, from dy (= dL/d(bn output), already
// masked by ReLU and dropout) and z:  dz = al[c] * dy + bz[c] * (z - mu[c]) + de[c]
//   al = gamma*rstd, bz = -gamma*rstd^2 * dgamma/M, de = -gamma*rstd * dbeta/M   (bn_finalize_bwd writes them;
//   eval-mode / no BatchNorm: bz = de = 0)
template <class CP>
struct LoadDzT {
  static constexpr bool kLateConsts = ConstsTrait<CP>::kLate;
  static constexpr bool kLean = true;
  const float *DY, *Z;
  int ld;
  CP mu, al, bz, de;
  struct Raw { float4 dy, z; };
  struct Consts { float4 u, a, b, d; };
  struct UBase { const float *dy, *z; CP mu, al, bz, de; };
  struct TC { uint32_t off; };
  __device__ __forceinline__ UBase ubase(int row, int col) const {
    const int64_t e = (int64_t)row * ld + col;
    return UBase{DY + e, Z + e, mu + col, al + col, bz + col, de + col};
  }
  __device__ __forceinline__ TC tconst(int r, int c) const { return TC{(uint32_t)(r * ld + c) * 4u}; }
  __device__ __forceinline__ Raw fetch_u(const UBase &u, const TC &t) const { return Raw{vld4u(u.dy, t.off), vld4u(u.z, t.off)}; }
  __device__ __forceinline__ Consts consts_u(const UBase &u, int c) const {
    return Consts{vld4(u.mu + c), vld4(u.al + c), vld4(u.bz + c), vld4(u.de + c)};
  }
  __device__ __forceinline__ float4 finish_u(const Raw &r, const Consts &k, const UBase &, const TC &) const { return finish(r, k, 0, 0); }
  __device__ __forceinline__ Raw fetch(int m, int c) const {
    return Raw{vld4(DY + (int64_t)m * ld + c), vld4(Z + (int64_t)m * ld + c)};
  }
  __device__ __forceinline__ Consts consts(int c) const { return Consts{vld4(mu + c), vld4(al + c), vld4(bz + c), vld4(de + c)}; }
  __device__ __forceinline__ float4 finish(const Raw &r, const Consts &k, int, int) const {
    float4 o;
    o.x = fmaf(k.a.x, r.dy.x, fmaf(k.b.x, r.z.x - k.u.x, k.d.x));
    o.y = fmaf(k.a.y, r.dy.y, fmaf(k.b.y, r.z.y - k.u.y, k.d.y));
    o.z = fmaf(k.a.z, r.dy.z, fmaf(k.b.z, r.z.z - k.u.z, k.d.z));
    o.w = fmaf(k.a.w, r.dy.w, fmaf(k.b.w, r.z.w - k.u.w, k.d.w));
    return o;
  }
};
typedef LoadDzT<const float *> LoadDz;
typedef LoadDzT<lds_cfp> LoadDzL;

// Tee: the operand value also goes to memory as it passes (out[m][c], row pitch the loader's own), so that a LATER product
// on other kernels can read the transformed matrix (the tail's weight gradients take a(z) and dz from here).  One column
// tile of the grid is given a non-null `out`; rows / reduction indices that were clamped store a duplicate of the same
// value at its own (valid) address.
template <class L>
struct Tee {
  static constexpr bool kLateConsts = L::kLateConsts;
  static constexpr bool kLean = L::kLean;
  L in;
  float *out;
  int ld;
  typedef typename L::Raw Raw;
  typedef typename L::Consts Consts;
  struct UBase { typename L::UBase in; float *out; };
  struct TC { typename L::TC in; uint32_t off; };
  __device__ __forceinline__ UBase ubase(int row, int col) const {
    return UBase{in.ubase(row, col), out ? out + (int64_t)row * ld + col : nullptr};
  }
  __device__ __forceinline__ TC tconst(int r, int c) const { return TC{in.tconst(r, c), (uint32_t)(r * ld + c) * 4u}; }
  __device__ __forceinline__ Raw fetch_u(const UBase &u, const TC &t) const { return in.fetch_u(u.in, t.in); }
  __device__ __forceinline__ Consts consts_u(const UBase &u, int c) const { return in.consts_u(u.in, c); }
  __device__ __forceinline__ float4 finish_u(const Raw &r, const Consts &k, const UBase &u, const TC &t) const {
    const float4 v = in.finish_u(r, k, u.in, t.in);
    if (u.out) *reinterpret_cast<float4 *>(reinterpret_cast<char *>(u.out) + t.off) = v;
    return v;
  }
  __device__ __forceinline__ Raw fetch(int m, int c) const { return in.fetch(m, c); }
  __device__ __forceinline__ Consts consts(int c) const { return in.consts(c); }
  __device__ __forceinline__ float4 finish(const Raw &r, const Consts &k, int m, int c) const {
    const float4 v = in.finish(r, k, m, c);
    if (out) vst4(out + (int64_t)m * ld + c, v);
    return v;
  }
};

template <class L, bool LEAN = L::kLean> struct LeanTypes { struct TC {}; struct UBase {}; };
template <class L> struct LeanTypes<L, true> { typedef typename L::TC TC; typedef typename L::UBase UBase; };

__device__ __forceinline__ int ptid() { return (int)threadIdx.x - (kThreads - kProd); }

// ---- LDS tiles -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int kc_off(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 2); }

// One BK slice of one operand on its way from global memory to LDS: raw vectors + constants in registers.
// KC tile: ROWS x 32, chunk id q = t + 256 i -> (row q / 8, reduction chunk q % 8); all chunks of a thread share the
// reduction columns, so ONE set of constants serves them (the transformed matrices of a KC operand are indexed
// [row = m][red = feature]).  Rows beyond rows_valid re-read the last valid row (finite values that only feed masked
// outputs); reduction indices >= red_end are fetched from a clamped (valid) address and zeroed at the LDS write.
template <int ROWS, class L, int NP = kProd>
struct KcStage {
  static constexpr int NV = (ROWS * 8 + NP - 1) / NP;
  typename L::Raw raw[NV];
  typename L::Consts k;
  int red;       // this thread's first reduction index of the slice (unclamped)
};
template <int ROWS, class L, int NP = kProd>
__device__ __forceinline__ KcStage<ROWS, L, NP> kc_fetch(const L &ld, int row0, int rows_valid, int red0, int red_end) {
  KcStage<ROWS, L, NP> st;
  st.red = red0 + ((ptid() & 7) << 2);
  const int redc = st.red < red_end ? st.red : red_end - 4;
  if constexpr (!L::kLateConsts) st.k = ld.consts(redc);      // global constants travel with the slice's loads
#pragma unroll
  for (int i = 0; i < KcStage<ROWS, L, NP>::NV; ++i) {
    int row = (ptid() >> 3) + i * (NP / 8);
    row = row < rows_valid ? row : rows_valid - 1;
    st.raw[i] = ld.fetch(row0 + row, redc);
  }
  return st;
}
template <int ROWS, class L, int NP = kProd>
__device__ __forceinline__ void kc_finish(float *T, const L &ld, const KcStage<ROWS, L, NP> &st, int row0, int rows_valid, int red_end) {
  const bool live = st.red < red_end;
  const int redc = live ? st.red : red_end - 4;
  typename L::Consts k = st.k;
  if constexpr (L::kLateConsts) k = ld.consts(redc);           // constants in LDS (joined by this workgroup): a ds_read here
#pragma unroll
  for (int i = 0; i < KcStage<ROWS, L, NP>::NV; ++i) {
    const int trow = (ptid() >> 3) + i * (NP / 8);
    if (trow < ROWS) {
      const int row = trow < rows_valid ? trow : rows_valid - 1;
      float4 v = ld.finish(st.raw[i], k, row0 + row, redc);
      if (!live) v = zero4();
      vst4(T + kc_off(trow, ptid() & 7), v);
    }
  }
}
// OC tile: 32 x WIDTH (pitch S): chunk id q -> (reduction row q / (WIDTH/4), outputs (q % (WIDTH/4)) * 4 ..).  The
// transformed matrices of an OC operand are indexed [red = m][out = feature], so a thread's chunks keep their COLUMNS for
// the whole kernel: the per-column constants and the chunk coordinates are prepared ONCE (OcMap) — re-loading 12-16
// constant vectors per stage at the LDS write put their whole latency into every producer phase.  Outputs beyond
// out_valid re-read the last valid group; reduction rows >= red_end are zeroed at the LDS write.
template <int WIDTH, class L>
struct OcMap {
  static constexpr int NV = (WIDTH * 8 + kProd - 1) / kProd;
  typename L::Consts k[NV];
  int rrow[NV];     // reduction row of the chunk inside a slice
  int col[NV];      // first output column (global), clamped to the valid range
  int lofs[NV];     // LDS offset, or -1 for a chunk that does not exist
};
template <int WIDTH, class L>
struct OcStage {
  static constexpr int NV = (WIDTH * 8 + kProd - 1) / kProd;
  typename L::Raw raw[NV];
};
template <int WIDTH, int S, class L>
__device__ __forceinline__ OcMap<WIDTH, L> oc_prepare(const L &ld, int out0, int out_valid) {
  constexpr int CPR = WIDTH / 4;
  OcMap<WIDTH, L> m;
#pragma unroll
  for (int i = 0; i < OcMap<WIDTH, L>::NV; ++i) {
    const int q = ptid() + i * kProd;
    const int qc = q < WIDTH * 8 ? q : WIDTH * 8 - 1;
    const int o = (qc % CPR) << 2;
    m.rrow[i] = qc / CPR;
    m.col[i] = out0 + (o < out_valid ? o : out_valid - 4);
    m.lofs[i] = q < WIDTH * 8 ? (qc / CPR) * S + o : -1;
    m.k[i] = ld.consts(m.col[i]);
  }
  return m;
}
template <int WIDTH, class L>
__device__ __forceinline__ OcStage<WIDTH, L> oc_fetch(const L &ld, const OcMap<WIDTH, L> &m, int red0, int red_end) {
  OcStage<WIDTH, L> st;
#pragma unroll
  for (int i = 0; i < OcStage<WIDTH, L>::NV; ++i) {
    int red = red0 + m.rrow[i];
    red = red < red_end ? red : red_end - 1;
    st.raw[i] = ld.fetch(red, m.col[i]);
  }
  return st;
}
template <int WIDTH, class L>
__device__ __forceinline__ void oc_finish(float *T, const L &ld, const OcStage<WIDTH, L> &st, const OcMap<WIDTH, L> &m, int red0,
                                          int red_end) {
#pragma unroll
  for (int i = 0; i < OcStage<WIDTH, L>::NV; ++i) {
    if (m.lofs[i] >= 0) {
      const int red = red0 + m.rrow[i];
      const bool live = red < red_end;
      float4 v = ld.finish(st.raw[i], m.k[i], live ? red : red_end - 1, m.col[i]);
      if (!live) v = zero4();
      vst4(T + m.lofs[i], v);
    }
  }
}

// OC source, KC tile ("transpose on the way in"): a thread owns a 4 (reduction) x 4 (output) block — four float4 loads along
// the outputs, one per reduction row — and, after the element transform, writes it as four float4 along the REDUCTION
// index into the KC tile (row = output, chunk = reduction / 4).  The transposition is pure register naming, the LDS gets
// the same number of b128 writes as an OC tile would, and the consumers read conflict-free b128 fragments instead of
// four b32 per operand and step (measured: no change in time for the weight-gradient and input-gradient products — the
// consumers were not their limit — so this form is kept for having ONE fragment path).  Block id b = thread: output block b % (ROWS/4), reduction block b / (ROWS/4): threads
// run along the outputs, so a wave-load covers whole 256-B / 448-B row segments.
template <int ROWS, class L>
struct OtMap {
  typename L::Consts k;
  typename LeanTypes<L>::TC tc[4];   // thread-constant address parts of the block's 4 reduction rows (lean loaders)
  int col;          // first output column (global), clamped
  int ob, rb;       // output block, reduction block; ob < 0: this thread has no block
};
template <int ROWS, class L>
struct OtStage {
  typename L::Raw raw[4];
};
// General addressing (loaders without the lean interface: panel_gemm.hip's grouped operand).
template <int ROWS, class L>
struct OtOperandG {
  L ld;
  int out0, out_valid, red_end;
  __device__ __forceinline__ OtMap<ROWS, L> prep() const {
    constexpr int OB = ROWS / 4;
    OtMap<ROWS, L> m;
    const int b = ptid();
    const bool has = b < OB * 8;
    const int bc = has ? b : OB * 8 - 1;
    m.ob = has ? bc % OB : -1;
    m.rb = bc / OB;
    const int o = (bc % OB) << 2;
    m.col = out0 + (o < out_valid ? o : out_valid - 4);
    m.k = ld.consts(m.col);
    return m;
  }
  __device__ __forceinline__ OtStage<ROWS, L> fetch(const OtMap<ROWS, L> &m, int red0) const {
    OtStage<ROWS, L> st;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int red = red0 + 4 * m.rb + j;
      red = red < red_end ? red : red_end - 1;
      st.raw[j] = ld.fetch(red, m.col);
    }
    return st;
  }
  __device__ __forceinline__ void finish(float *T, const OtStage<ROWS, L> &st, const OtMap<ROWS, L> &m, int red0) const {
    if (m.ob < 0) return;
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int red = red0 + 4 * m.rb + j;
      const bool live = red < red_end;
      v[j] = ld.finish(st.raw[j], m.k, live ? red : red_end - 1, m.col);
      if (!live) v[j] = zero4();
    }
    const int row = 4 * m.ob;
    vst4(T + kc_off(row + 0, m.rb), make_float4(v[0].x, v[1].x, v[2].x, v[3].x));
    vst4(T + kc_off(row + 1, m.rb), make_float4(v[0].y, v[1].y, v[2].y, v[3].y));
    vst4(T + kc_off(row + 2, m.rb), make_float4(v[0].z, v[1].z, v[2].z, v[3].z));
    vst4(T + kc_off(row + 3, m.rb), make_float4(v[0].w, v[1].w, v[2].w, v[3].w));
  }
  __device__ __forceinline__ OtStage<ROWS, L> fetch_any(const OtMap<ROWS, L> &m, int red0) const { return fetch(m, red0); }
  __device__ __forceinline__ void finish_any(float *T, const OtStage<ROWS, L> &st, const OtMap<ROWS, L> &m, int red0) const { finish(T, st, m, red0); }
};
// Lean addressing: ONE code path; the last slice, when it is partial (rem = red_end - red0 < BK, a uniform test), re-aims
// the rows beyond the end at the last valid row and zeroes their values at the LDS write.
template <int ROWS, class L>
struct OtOperand {
  static_assert(L::kLean, "OtOperand needs a loader with the lean interface (else OtOperandG)");
  L ld;
  int out0, out_valid, red_end;
  __device__ __forceinline__ OtMap<ROWS, L> prep() const {
    constexpr int OB = ROWS / 4;
    OtMap<ROWS, L> m;
    const int b = ptid();
    const bool has = b < OB * 8;
    const int bc = has ? b : OB * 8 - 1;
    m.ob = has ? bc % OB : -1;
    m.rb = bc / OB;
    const int o = (bc % OB) << 2;
    m.col = out0 + (o < out_valid ? o : out_valid - 4);
    m.k = ld.consts(m.col);
#pragma unroll
    for (int j = 0; j < 4; ++j) m.tc[j] = ld.tconst(4 * m.rb + j, m.col);      // relative to ubase(red0, 0)
    return m;
  }
  __device__ __forceinline__ void store_block(float *T, const OtMap<ROWS, L> &m, const float4 (&v)[4]) const {
    const int row = 4 * m.ob;
    vst4(T + kc_off(row + 0, m.rb), make_float4(v[0].x, v[1].x, v[2].x, v[3].x));
    vst4(T + kc_off(row + 1, m.rb), make_float4(v[0].y, v[1].y, v[2].y, v[3].y));
    vst4(T + kc_off(row + 2, m.rb), make_float4(v[0].z, v[1].z, v[2].z, v[3].z));
    vst4(T + kc_off(row + 3, m.rb), make_float4(v[0].w, v[1].w, v[2].w, v[3].w));
  }
  // full slices (see KcOperand)
  __device__ __forceinline__ OtStage<ROWS, L> fetch(const OtMap<ROWS, L> &m, int red0) const {
    OtStage<ROWS, L> st;
    const auto ub = ld.ubase(red0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) st.raw[j] = ld.fetch_u(ub, m.tc[j]);
    return st;
  }
  __device__ __forceinline__ void finish(float *T, const OtStage<ROWS, L> &st, const OtMap<ROWS, L> &m, int red0) const {
    if (m.ob < 0) return;
    const auto ub = ld.ubase(red0, 0);
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = ld.finish_u(st.raw[j], m.k, ub, m.tc[j]);
    store_block(T, m, v);
  }
  // a slice that may be partial: rows beyond the end re-aimed at the last valid one, their values zeroed
  __device__ __forceinline__ OtStage<ROWS, L> fetch_any(const OtMap<ROWS, L> &m, int red0) const {
    OtStage<ROWS, L> st;
    const auto ub = ld.ubase(red0, 0);
    const int rem = red_end - red0;
    const typename L::TC last = ld.tconst(rem - 1, m.col);
#pragma unroll
    for (int j = 0; j < 4; ++j) st.raw[j] = ld.fetch_u(ub, 4 * m.rb + j >= rem ? last : m.tc[j]);
    return st;
  }
  __device__ __forceinline__ void finish_any(float *T, const OtStage<ROWS, L> &st, const OtMap<ROWS, L> &m, int red0) const {
    if (m.ob < 0) return;
    const auto ub = ld.ubase(red0, 0);
    const int rem = red_end - red0;
    const typename L::TC last = ld.tconst(rem - 1, m.col);
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool dead = 4 * m.rb + j >= rem;
      v[j] = ld.finish_u(st.raw[j], m.k, ub, dead ? last : m.tc[j]);
      if (dead) v[j] = zero4();
    }
    store_block(T, m, v);
  }
};

// Operand adaptors for the main loop: prep() once per producer thread, fetch(prep, red0) issues the loads of a slice,
// finish(T, stage, prep, red0) transforms and writes it.
struct NoPrep {};
// General addressing (loaders without the lean interface).
template <int ROWS, class L, int NP = kProd>
struct KcOperandG {
  L ld;
  int row0, rows_valid, red_end;
  __device__ __forceinline__ NoPrep prep() const { return NoPrep{}; }
  __device__ __forceinline__ KcStage<ROWS, L, NP> fetch(const NoPrep &, int red0) const { return kc_fetch<ROWS, L, NP>(ld, row0, rows_valid, red0, red_end); }
  __device__ __forceinline__ void finish(float *T, const KcStage<ROWS, L, NP> &st, const NoPrep &, int) const {
    kc_finish<ROWS, L, NP>(T, ld, st, row0, rows_valid, red_end);
  }
  __device__ __forceinline__ KcStage<ROWS, L, NP> fetch_any(const NoPrep &p, int red0) const { return fetch(p, red0); }
  __device__ __forceinline__ void finish_any(float *T, const KcStage<ROWS, L, NP> &st, const NoPrep &p, int red0) const { finish(T, st, p, red0); }
};
// Lean addressing (see LoadPlain): per-thread offsets and LDS positions prepared once; a slice costs the producers no
// address arithmetic.  The last slice, when partial (rem < BK: uniform), re-aims the chunks beyond the end at the last
// valid chunk of their row and zeroes their values at the LDS write.
template <int ROWS, class L, int NP = kProd>
struct KcPrep {
  static constexpr int NV = (ROWS * 8 + NP - 1) / NP;
  typename L::TC tc[NV];
  int lofs[NV];                         // LDS float offset of the chunk, -1: the chunk does not exist
  int cc;                               // first column of the thread's chunk inside a slice
};
template <int ROWS, class L, int NP = kProd>
struct KcOperand {
  static_assert(L::kLean, "KcOperand needs a loader with the lean interface (else KcOperandG)");
  static constexpr int NV = KcPrep<ROWS, L, NP>::NV;
  L ld;
  int row0, rows_valid, red_end;
  __device__ __forceinline__ int row_of(int i) const {
    const int trow = (ptid() >> 3) + i * (NP / 8);
    return trow < rows_valid ? trow : rows_valid - 1;
  }
  __device__ __forceinline__ KcPrep<ROWS, L, NP> prep() const {
    KcPrep<ROWS, L, NP> p;
    p.cc = (ptid() & 7) << 2;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int trow = (ptid() >> 3) + i * (NP / 8);
      p.lofs[i] = trow < ROWS ? kc_off(trow, ptid() & 7) : -1;
      p.tc[i] = ld.tconst(row_of(i), p.cc);
    }
    return p;
  }
  // fetch / finish: a FULL slice (red0 + BK <= red_end) — no clamps, no selects, no address arithmetic.
  // fetch_any / finish_any: a slice that may be the partial one (chunks beyond the end re-aimed at the row's last valid
  // chunk, their values zeroed at the LDS write; selects).  main_loop walks the partial slice FIRST, in its prologue, so
  // that the steady state only ever sees full slices: with the selects inside the loop the producers' vector
  // instruction count did not drop at all (SQ_INSTS_VALU, profiles/r03_tail_sq_counters.csv), and a uniform branch
  // around them made hipcc join the register stages with s_waitcnt vmcnt(0).
  __device__ __forceinline__ KcStage<ROWS, L, NP> fetch(const KcPrep<ROWS, L, NP> &p, int red0) const {
    KcStage<ROWS, L, NP> st;
    const auto ub = ld.ubase(row0, red0);
    if constexpr (!L::kLateConsts) st.k = ld.consts_u(ub, p.cc);
#pragma unroll
    for (int i = 0; i < NV; ++i) st.raw[i] = ld.fetch_u(ub, p.tc[i]);
    return st;
  }
  __device__ __forceinline__ void finish(float *T, const KcStage<ROWS, L, NP> &st, const KcPrep<ROWS, L, NP> &p, int red0) const {
    const auto ub = ld.ubase(row0, red0);
    typename L::Consts k = st.k;
    if constexpr (L::kLateConsts) k = ld.consts_u(ub, p.cc);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if ((ROWS * 8) % NP == 0 || i + 1 < NV || p.lofs[i] >= 0) vst4(T + p.lofs[i], ld.finish_u(st.raw[i], k, ub, p.tc[i]));
  }
  __device__ __forceinline__ KcStage<ROWS, L, NP> fetch_any(const KcPrep<ROWS, L, NP> &p, int red0) const {
    KcStage<ROWS, L, NP> st;
    const auto ub = ld.ubase(row0, red0);
    const int rem = red_end - red0;
    const bool dead = p.cc >= rem;
    if constexpr (!L::kLateConsts) st.k = ld.consts_u(ub, dead ? rem - 4 : p.cc);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const typename L::TC re = ld.tconst(row_of(i), rem - 4);
      st.raw[i] = ld.fetch_u(ub, dead ? re : p.tc[i]);
    }
    return st;
  }
  __device__ __forceinline__ void finish_any(float *T, const KcStage<ROWS, L, NP> &st, const KcPrep<ROWS, L, NP> &p, int red0) const {
    const auto ub = ld.ubase(row0, red0);
    const int rem = red_end - red0;
    const bool dead = p.cc >= rem;
    typename L::Consts k = st.k;
    if constexpr (L::kLateConsts) k = ld.consts_u(ub, dead ? rem - 4 : p.cc);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if ((ROWS * 8) % NP == 0 || i + 1 < NV || p.lofs[i] >= 0) {
        const typename L::TC re = ld.tconst(row_of(i), rem - 4);
        float4 v = ld.finish_u(st.raw[i], k, ub, dead ? re : p.tc[i]);      // (a Tee stores the re-aimed value at its own address)
        if (dead) v = zero4();
        vst4(T + p.lofs[i], v);
      }
    }
  }
};
template <int WIDTH, int S, class L>
struct OcOperand {
  L ld;
  int out0, out_valid, red_end;
  __device__ __forceinline__ OcMap<WIDTH, L> prep() const { return oc_prepare<WIDTH, S>(ld, out0, out_valid); }
  __device__ __forceinline__ OcStage<WIDTH, L> fetch(const OcMap<WIDTH, L> &m, int red0) const { return oc_fetch<WIDTH>(ld, m, red0, red_end); }
  __device__ __forceinline__ void finish(float *T, const OcStage<WIDTH, L> &st, const OcMap<WIDTH, L> &m, int red0) const {
    oc_finish<WIDTH>(T, ld, st, m, red0, red_end);
  }
  __device__ __forceinline__ OcStage<WIDTH, L> fetch_any(const OcMap<WIDTH, L> &m, int red0) const { return fetch(m, red0); }
  __device__ __forceinline__ void finish_any(float *T, const OcStage<WIDTH, L> &st, const OcMap<WIDTH, L> &m, int red0) const { finish(T, st, m, red0); }
};

// fragments of step half h (16 reduction indices): 4 values per lane = k-steps j = 0..3
__device__ __forceinline__ float4 kc_frag(const float *T, int row, int h, int g) { return vld4(T + kc_off(row, 4 * h + g)); }
template <int S>
__device__ __forceinline__ float4 oc_frag(const float *T, int col, int h, int g) {
  const float *p = T + (16 * h + 4 * g) * S + col;
  return make_float4(p[0], p[S], p[2 * S], p[3 * S]);
}

// Fragment registers of one half-slice (16 reduction indices) for a consumer wave: its 16 rows of R and the 7 sub-tiles
// of Cc.
// NS <= NSUB: the sub-tiles a kernel actually needs (a 64-column range needs 4 of the 7: the others would be MFMAs on
// columns that are masked anyway).
template <int NS>
struct Frags {
  float4 b, a[NS];
};
template <bool R_KC, bool C_KC, int NS>
__device__ __forceinline__ void read_frags(Frags<NS> &f, const float *Rt, const float *Ct, int wave, int lane, int h) {
  const int r = lane & 15, g = lane >> 4;
  f.b = R_KC ? kc_frag(Rt, wave * 16 + r, h, g) : oc_frag<SR_OC>(Rt, wave * 16 + r, h, g);
#pragma unroll
  for (int s = 0; s < NS; ++s) f.a[s] = C_KC ? kc_frag(Ct, s * 16 + r, h, g) : oc_frag<SC_OC>(Ct, s * 16 + r, h, g);
}
// 4 NS MFMAs, j-major: consecutive ones hit different accumulators (a dependent v_mfma_f32_16x16x4_f32 needs 40 cycles, an
// independent one issues every 32).
template <int NS>
__device__ __forceinline__ void mma_half(floatx4 (&acc)[NSUB], const Frags<NS> &f) {
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[s].x, f.b.x, acc[s], 0, 0, 0);
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[s].y, f.b.y, acc[s], 0, 0, 0);
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[s].z, f.b.z, acc[s], 0, 0, 0);
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[s].w, f.b.w, acc[s], 0, 0, 0);
}

constexpr int kStageFloats = 64 * BK + BNT * BK > 32 * SR_OC + 32 * SC_OC ? 64 * BK + BNT * BK : 32 * SR_OC + 32 * SC_OC;
constexpr int kRing = 4;                            // LDS slots
constexpr int kLdsFloats = kRing * kStageFloats;    // 94 KB
constexpr int kROffKC = 0, kCOffKC = 64 * BK;      // offsets of the two tiles inside a slot
constexpr int kROffOC = 0, kCOffOC = 32 * SR_OC;

// Barriers of the main loop.  __syncthreads() drains vmcnt(0) in front of s_barrier, which would make a producer wait at
// every phase for the global loads it has JUST issued (their latency, not the MFMA phase, then sets the pace: measured
// 21 us against 17 for the forward product).  The ring only needs LDS ordering:
//   producer: its LDS writes have landed (lgkmcnt(0)) before it signals; its global loads stay in flight;
//   consumer: the fragments it read out of the slot that is overwritten next were consumed by MFMAs it has already
//             issued (the waits the compiler placed in front of them), so it arrives without waiting for anything.
__device__ __forceinline__ void producer_barrier() {
  __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0), vmcnt / expcnt untouched
  __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ void consumer_barrier() { __builtin_amdgcn_s_barrier(); }

// The main loop over the BK slices [red_begin, red_end), one barrier per slice, roles split by wave:
//   waves 4-7 (producers): in phase i they transform slice i+3 (fetched during phase i-3) and write it to LDS slot
//     (i+3) % 4, then issue the global loads of slice i+6 into the register stage that just became free (three register
//     stages with fixed roles: the loop is unrolled by three) — every load has three phases (~2 us) to land;
//   waves 0-3 (consumers, one per SIMD): in phase i they run the 56 MFMAs of slice i out of slot i % 4.  The second
//     half's fragments are read at the top of the phase and the first half of slice i+1 (written in phase i-2) in the
//     middle of it, so the matrix pipe never waits for an LDS read across the barrier.
// The element work of the operand loads (BatchNorm / ReLU / dropout bit) runs on the producers' VALU slots beside the
// consumers' MFMAs (not for free: the f32 MFMA shares the vector issue, so it is kept to a few instructions per float4).  fetchR/fetchC(red0) issue the loads of one slice, finishR/finishC(T, stage, red0)
// transform and write it.  Fetches past the last slice re-read the last slice (never written).
// `pre` (optional): work the CONSUMER waves do while the producers' first loads are in flight — joining the statistics
// behind the R operand's constants into LDS (tail.hip) — with one extra barrier between it and the producers' first
// LDS write (which is where those constants are first read: kLateConsts loaders).
struct NoPre {
  static constexpr bool kActive = false;
  __device__ __forceinline__ void operator()() const {}
};
template <class F>
struct Pre {
  static constexpr bool kActive = true;
  F f;
  __device__ __forceinline__ void operator()() const { f(); }
};
template <class F>
__device__ __forceinline__ Pre<F> make_pre(F f) { return Pre<F>{f}; }

template <bool R_KC, bool C_KC, int NS = NSUB, class OR, class OC, class PRE = NoPre>
__device__ __forceinline__ void main_loop(floatx4 (&acc)[NSUB], float *lds, int red_begin, int red_end, const OR &opR,
                                          const OC &opC, const PRE &pre = PRE{}) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int rOff = R_KC ? kROffKC : kROffOC, cOff = C_KC ? kCOffKC : kCOffOC;
  const int nst = (red_end - red_begin + BK - 1) / BK;
  if (nst <= 0) return;
  // Slice ORDER: a sum does not care, so the partial slice (if the range is not a multiple of BK) is walked FIRST —
  // index 0, handled by the prologue's *_any calls — and every slice the steady state touches is a full one.  Indices
  // past the end repeat the last index (never written); with a single, partial slice that is the partial one again,
  // which only the *_any forms may read: `lone` routes every access through them (tiny reductions only).
  const bool partial = (red_end - red_begin) % BK != 0;
  const bool lone = partial && nst == 1;
  auto at = [&](int i) {
    const int ii = i < nst ? i : nst - 1;
    return red_begin + (partial ? (ii == 0 ? nst - 1 : ii - 1) : ii) * BK;
  };
  auto slot = [&](int i) { return lds + (i % kRing) * kStageFloats; };
  if (lone) {
    // a reduction shorter than one slice: one unpipelined step through the *_any forms (the pipeline below would
    // prefetch "the last slice" again with the full-slice loads)
    if (wave >= 4) {
      const auto pR = opR.prep();
      const auto pC = opC.prep();
      const auto r0 = opR.fetch_any(pR, red_begin);
      const auto c0 = opC.fetch_any(pC, red_begin);
      if constexpr (PRE::kActive) producer_barrier();
      opR.finish_any(slot(0) + rOff, r0, pR, red_begin);
      opC.finish_any(slot(0) + cOff, c0, pC, red_begin);
      producer_barrier();
    } else {
      if constexpr (PRE::kActive) {
        pre();
        producer_barrier();
      }
      consumer_barrier();
      Frags<NS> f0, f1;
      read_frags<R_KC, C_KC, NS>(f0, slot(0) + rOff, slot(0) + cOff, wave, lane, 0);
      read_frags<R_KC, C_KC, NS>(f1, slot(0) + rOff, slot(0) + cOff, wave, lane, 1);
      mma_half(acc, f0);
      mma_half(acc, f1);
    }
    __syncthreads();
    return;
  }
  if (wave >= 4) {
    // ---------------------------------------------------------------------------------------------- producers
    const auto pR = opR.prep();
    const auto pC = opC.prep();
    auto fetchR = [&](int red0) { return opR.fetch(pR, red0); };
    auto fetchC = [&](int red0) { return opC.fetch(pC, red0); };
    auto finishR = [&](float *T, const auto &st, int red0) { opR.finish(T, st, pR, red0); };
    auto finishC = [&](float *T, const auto &st, int red0) { opC.finish(T, st, pC, red0); };
    // slices 0..2 straight into their slots, slices 3..5 left in flight in the three register stages
    auto r0 = opR.fetch_any(pR, at(0));
    auto c0 = opC.fetch_any(pC, at(0));
    auto r1 = fetchR(at(1));
    auto c1 = fetchC(at(1));
    auto r2 = fetchR(at(2));
    auto c2 = fetchC(at(2));
    if constexpr (PRE::kActive) producer_barrier();      // the consumers have joined the constants
    opR.finish_any(slot(0) + rOff, r0, pR, at(0));
    opC.finish_any(slot(0) + cOff, c0, pC, at(0));
    r0 = fetchR(at(3));
    c0 = fetchC(at(3));
    if (nst > 1) {
      finishR(slot(1) + rOff, r1, at(1));
      finishC(slot(1) + cOff, c1, at(1));
    }
    r1 = fetchR(at(4));
    c1 = fetchC(at(4));
    if (nst > 2) {
      finishR(slot(2) + rOff, r2, at(2));
      finishC(slot(2) + cOff, c2, at(2));
    }
    r2 = fetchR(at(5));
    c2 = fetchC(at(5));
    producer_barrier();
    for (int i = 0; i < nst; i += 3) {
      // phase i: (r0, c0) = slice i+3, (r1, c1) = slice i+4, (r2, c2) = slice i+5
      if (i + 3 < nst) {
        finishR(slot(i + 3) + rOff, r0, at(i + 3));
        finishC(slot(i + 3) + cOff, c0, at(i + 3));
      }
      r0 = fetchR(at(i + 6));
      c0 = fetchC(at(i + 6));
      producer_barrier();
      if (i + 1 >= nst) break;
      if (i + 4 < nst) {
        finishR(slot(i + 4) + rOff, r1, at(i + 4));
        finishC(slot(i + 4) + cOff, c1, at(i + 4));
      }
      r1 = fetchR(at(i + 7));
      c1 = fetchC(at(i + 7));
      producer_barrier();
      if (i + 2 >= nst) break;
      if (i + 5 < nst) {
        finishR(slot(i + 5) + rOff, r2, at(i + 5));
        finishC(slot(i + 5) + cOff, c2, at(i + 5));
      }
      r2 = fetchR(at(i + 8));
      c2 = fetchC(at(i + 8));
      producer_barrier();
    }
  } else {
    // ---------------------------------------------------------------------------------------------- consumers
    Frags<NS> f0, f1;
    if constexpr (PRE::kActive) {
      pre();
      producer_barrier();              // (lgkmcnt(0): the constants are in LDS) pairs with the producers' extra barrier
    }
    __builtin_amdgcn_s_setprio(3);     // the matrix pipe's wave wins the issue arbitration against the producer beside it
    consumer_barrier();
    read_frags<R_KC, C_KC, NS>(f0, slot(0) + rOff, slot(0) + cOff, wave, lane, 0);
    for (int i = 0; i < nst; ++i) {
      const float *T = slot(i), *Tn = slot(i + 1);
      read_frags<R_KC, C_KC, NS>(f1, T + rOff, T + cOff, wave, lane, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(acc, f0);
      __builtin_amdgcn_sched_barrier(0);
      if (i + 1 < nst) read_frags<R_KC, C_KC, NS>(f0, Tn + rOff, Tn + cOff, wave, lane, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(acc, f1);
      __builtin_amdgcn_sched_barrier(0);
      consumer_barrier();
    }
  }
  __syncthreads();     // (full fence) every wave has passed 1 + nst ring barriers; the ring is free from here on
}


// ---- main loop with the UNTRANSFORMED operand on LDS-DMA ------------------------------------------------------------
// The C operand of a forward product is the weight matrix as it lies in memory: nothing to transform, so it need not pass
// through registers at all.  Here waves 6-7 move it global -> LDS with global_load_lds_dwordx4 (16 B per lane, 1 KiB per
// wave-instruction, no VGPR destination, no ds_write, no VALU): the KC tile's XOR swizzle goes on the per-lane SOURCE
// address, the LDS side is lane-linear (cdna_hip_programming.md, rule 21).  Waves 4-5 stage the transformed R operand
// through registers as before (128 threads: 4 chunks each), waves 0-3 are the consumers.  Roles are split BY WAVE because
// a wave that mixes LDS-DMA with ordinary loads makes hipcc drain vmcnt(0) at every use of an ordinary load's result;
// a loader wave issues nothing but DMA and counts it by hand.
//   R ring: 4 slots of 64 x 32 floats, written in phase i for slice i + 3 (as in main_loop).
//   C ring: 6 slots of 112 x 32 floats; the DMA of slice s is issued in phase s - 5 and must have landed before the
//           barrier that ends phase s - 2 (the consumers read the first half of slice i + 1 in the middle of phase i):
//           three newer slices (21 instructions per loader wave) may still be in flight -> s_waitcnt vmcnt(21).
// Rows of C beyond cols_valid re-read the last valid row; reduction indices >= red_end read a clamped (valid, finite)
// address: the R operand is zero there, so their products vanish.
constexpr int kRSlots = 4, kCSlots = 6;
constexpr int kRSlotFloats = 64 * BK, kCSlotFloats = BNT * BK;
constexpr int kDmaLdsFloats = kRSlots * kRSlotFloats + kCSlots * kCSlotFloats;      // 116 KB
constexpr int kDmaPerSlice = BNT * 8 / kWaveLanes;                                   // 14 wave-instructions per C slice
static_assert(kDmaPerSlice == 14, "two loader waves take 7 each");

__device__ __forceinline__ void dma_c_slice(const float *W, int ldw, int n0, int cols_valid, int red0, int red_end, float *slot,
                                            int lw /* 0 or 1: which loader wave */, int lane) {
  const int cpos = lane & 7;
#pragma unroll
  for (int jj = 0; jj < kDmaPerSlice / 2; ++jj) {
    const int j = 2 * jj + lw;
    const int row = 8 * j + (lane >> 3);
    const int chunk = cpos ^ ((row >> 1) & 7);
    const int rc = row < cols_valid ? row : cols_valid - 1;
    int red = red0 + 4 * chunk;
    red = red < red_end ? red : red_end - 4;
    const float *src = W + (int64_t)(n0 + rc) * ldw + red;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)(slot + j * 256), 16, 0, 0);
  }
}

template <int NS = NSUB, class OR>
__device__ __forceinline__ void main_loop_dma(floatx4 (&acc)[NSUB], float *lds, int red_begin, int red_end, const OR &opR,
                                              const float *W, int ldw, int n0, int cols_valid) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nst = (red_end - red_begin + BK - 1) / BK;
  if (nst <= 0) return;
  const bool partial = (red_end - red_begin) % BK != 0;      // the partial slice goes first (see main_loop); callers keep
  auto at = [&](int i) {                                      // reductions shorter than two slices off this loop
    const int ii = i < nst ? i : nst - 1;
    return red_begin + (partial ? (ii == 0 ? nst - 1 : ii - 1) : ii) * BK;
  };
  auto rslot = [&](int i) { return lds + (i % kRSlots) * kRSlotFloats; };
  auto cslot = [&](int i) { return lds + kRSlots * kRSlotFloats + (i % kCSlots) * kCSlotFloats; };
  if (wave >= 6) {
    // ------------------------------------------------------------------------------------------ C loaders (DMA only)
    const int lw = wave - 6;
    for (int s = 0; s < 5; ++s)
      if (s < nst) dma_c_slice(W, ldw, n0, cols_valid, at(s), red_end, cslot(s), lw, lane);
    // slices 0 and 1 landed (up to three newer ones in flight); fewer slices than that: everything
    if (nst > 4) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = 0; i < nst; ++i) {
      if (i + 5 < nst) dma_c_slice(W, ldw, n0, cols_valid, at(i + 5), red_end, cslot(i + 5), lw, lane);
      // before the barrier that ends phase i, slice i + 2 must be in LDS; issued after it: slices i+3, i+4, i+5 (those
      // that exist)
      const int newer = min(nst - 1, i + 5) - min(nst - 1, i + 2);
      if (newer >= 3) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
      else if (newer == 2) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
      else if (newer == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else if (wave >= 4) {
    // ------------------------------------------------------------------------------------------ R producers
    const auto pR = opR.prep();
    auto fetchR = [&](int red0) { return opR.fetch(pR, red0); };
    auto finishR = [&](float *T, const auto &st, int red0) { opR.finish(T, st, pR, red0); };
    auto r0 = opR.fetch_any(pR, at(0));
    auto r1 = fetchR(at(1));
    auto r2 = fetchR(at(2));
    opR.finish_any(rslot(0), r0, pR, at(0));
    r0 = fetchR(at(3));
    if (nst > 1) finishR(rslot(1), r1, at(1));
    r1 = fetchR(at(4));
    if (nst > 2) finishR(rslot(2), r2, at(2));
    r2 = fetchR(at(5));
    producer_barrier();
    for (int i = 0; i < nst; i += 3) {
      if (i + 3 < nst) finishR(rslot(i + 3), r0, at(i + 3));
      r0 = fetchR(at(i + 6));
      producer_barrier();
      if (i + 1 >= nst) break;
      if (i + 4 < nst) finishR(rslot(i + 4), r1, at(i + 4));
      r1 = fetchR(at(i + 7));
      producer_barrier();
      if (i + 2 >= nst) break;
      if (i + 5 < nst) finishR(rslot(i + 5), r2, at(i + 5));
      r2 = fetchR(at(i + 8));
      producer_barrier();
    }
  } else {
    // ------------------------------------------------------------------------------------------ consumers
    Frags<NS> f0, f1;
    __builtin_amdgcn_s_setprio(3);
    consumer_barrier();
    read_frags<true, true, NS>(f0, rslot(0), cslot(0), wave, lane, 0);
    for (int i = 0; i < nst; ++i) {
      read_frags<true, true, NS>(f1, rslot(i), cslot(i), wave, lane, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(acc, f0);
      __builtin_amdgcn_sched_barrier(0);
      if (i + 1 < nst) read_frags<true, true, NS>(f0, rslot(i + 1), cslot(i + 1), wave, lane, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(acc, f1);
      __builtin_amdgcn_sched_barrier(0);
      consumer_barrier();
    }
  }
  __syncthreads();
}

// XCD-aware tile order: workgroups are dealt to the 8 XCDs round-robin in launch order and each XCD has its own L2, so
// linear id -> (id % 8) * ceil(total / 8) + id / 8 gives every XCD a contiguous run of logical tiles (the tiles that
// share operand rows).  Returns -1 for the padding ids of a grid rounded up to a multiple of 8.
__device__ __forceinline__ int xcd_logical(int id, int total) {
  const int per = (total + 7) >> 3;
  const int l = (id & 7) * per + (id >> 3);
  return ((id >> 3) < per && l < total) ? l : -1;
}

}  // namespace tg
