// embed.hip — compressed-embedding lookups with the index math fused into the gather:
//   * dual-table compositional lookups: QR (quotient-remainder) hashing and CERP
//   * CSR-pruned table rows (the reference's numba kernels K1/K2)
//   * DHE universal-hash feature generator
//   * FM second-order + first-order term over an already gathered emb tensor
//
// Reference arithmetic:
//   QR    src/models/embeddings/qr_embedding.py:95-109   (r = idx % divider, q = idx // divider,
//         out = emb1[r] (*|+|cat) emb2[q]; cat is along dim=1 => [B,2F,D/2] for 2-D input)
//   CERP  src/models/embeddings/cerp_embedding.py:142-175 (q = trunc(idx / q_entity_per_row),
//         p = idx % bucket, out = S(Q)[q] + S(P)[p], S(w) = sign(w)*relu(|w| - sigmoid(s)));
//         retrain variant :329-367 (w * mask)
//   CSR   src/models/embeddings/pruned_embedding.py:136-204
//   DHE   src/models/embeddings/dh_embedding.py:213-236
//   FM    src/models/deepfm.py:91-98
//
// Lane mapping as in gather_fm.hip: a row of De = 4*LPR floats is LPR adjacent lanes x
// float4; a wave-instruction covers RS = 64/LPR lookups.
#include <hip/hip_fp16.h>

#include <cstdlib>

#include "common.hpp"

namespace {
using namespace mi;

enum { OP_MULT = 0, OP_ADD = 1, OP_CAT = 2 };
enum { XF_NONE = 0, XF_SOFT = 1, XF_MASK = 2 };

struct DualTables {
  const float *T1, *T2;      // [n1,De], [n2,De]
  const float *S1, *S2;      // XF_SOFT: threshold logits, same shapes
  const uint8_t *M1, *M2;    // XF_MASK: bool masks, same shapes
  int64_t n1, n2;
  int64_t mod1;              // i1 = idx % mod1
  int64_t div2;              // i2 = idx / div2
};

// i1 = id % mod1, i2 = id / div2 for a NON-NEGATIVE id (callers test id >= 0 first; a negative id takes the 64-bit path and
// is rejected by the range check like before).  A 64-bit division by a run-time value is a ~150-instruction routine on this
// ISA and every one of a lookup's De threads runs it twice: 11 of the dual backward's 24 us at the C3 shape.  Same integers,
// cheaper: power-of-two divisors (QR `divider: 2`, CERP bucket sizes) are a mask and a shift; ids and divisors below 2^32
// divide in 32 bits.
__device__ __forceinline__ void split_id(int64_t id, const DualTables &t, int64_t &i1, int64_t &i2) {
  const uint64_t u = (uint64_t)id;
  const bool small = (u >> 32) == 0 && ((uint64_t)t.mod1 >> 32) == 0 && ((uint64_t)t.div2 >> 32) == 0;
  if (id >= 0 && (t.mod1 & (t.mod1 - 1)) == 0 && (t.div2 & (t.div2 - 1)) == 0) {
    i1 = (int64_t)(u & (uint64_t)(t.mod1 - 1));
    i2 = (int64_t)(u >> (63 - __clzll((long long)t.div2)));
  } else if (small) {
    i1 = (int64_t)((uint32_t)u % (uint32_t)t.mod1);
    i2 = (int64_t)((uint32_t)u / (uint32_t)t.div2);
  } else {
    i1 = id % t.mod1;
    i2 = id / t.div2;
  }
}

__device__ __forceinline__ float sigmoidf_(float s) { return 1.f / (1.f + expf(-s)); }
__device__ __forceinline__ float signf_(float w) { return (w > 0.f) ? 1.f : ((w < 0.f) ? -1.f : 0.f); }
__device__ __forceinline__ float soft_(float w, float s) {
  const float u = fabsf(w) - sigmoidf_(s);
  return signf_(w) * (u > 0.f ? u : 0.f);
}

template <int XF>
__device__ __forceinline__ float4 load_row4(const float *T, const float *S, const uint8_t *M, int64_t o) {
  float4 w = ld4(T + o);
  if constexpr (XF == XF_SOFT) {
    const float4 s = ld4(S + o);
    w.x = soft_(w.x, s.x); w.y = soft_(w.y, s.y); w.z = soft_(w.z, s.z); w.w = soft_(w.w, s.w);
  } else if constexpr (XF == XF_MASK) {
    const uchar4 m = *reinterpret_cast<const uchar4 *>(M + o);
    w.x = m.x ? w.x : 0.f; w.y = m.y ? w.y : 0.f; w.z = m.z ? w.z : 0.f; w.w = m.w ? w.w : 0.f;
  }
  return w;
}

// element offset of lookup i's output row(s); CAT puts table-1 rows at field f and table-2 rows at
// field F+f of a [B,2F,De] tensor (torch.cat(dim=1) quirk).
__device__ __forceinline__ void out_offsets(int op, int64_t i, int F, int De, int64_t &o1, int64_t &o2) {
  if (op == OP_CAT) {
    const int64_t b = i / F, f = i % F;
    o1 = (b * 2 * F + f) * De;
    o2 = o1 + (int64_t)F * De;
  } else {
    o1 = o2 = i * De;
  }
}

// offsets (nullable, [F]): per-field row offsets added to the ids here (idx is then the model's raw [B, F] input), the sums
// written to rows_out (nullable, [n]) for the backward and the optimizer — the model's `x + offsets` without a launch
template <int LPR, int XF>
__global__ __launch_bounds__(kBlock) void k_dual_fwd(const int64_t *__restrict__ idx, DualTables t,
                                                     float *__restrict__ out, int64_t n, int F, int op,
                                                     int *err, const int64_t *__restrict__ offsets,
                                                     int64_t *__restrict__ rows_out) {
  constexpr int RS = kWave / LPR;
  constexpr int De = LPR * 4;
  const int lane = threadIdx.x & 63;
  const int q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ntiles = (n + RS - 1) / RS;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  int bad = 0;
  for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
    const int64_t i = tile * RS + r;
    if (i >= n) continue;
    int64_t id = idx[i];
    if (offsets) {
      id += offsets[(uint64_t)i % (uint32_t)F];
      if (rows_out && q == 0) rows_out[i] = id;
    }
    int64_t i1, i2;
    split_id(id, t, i1, i2);
    const bool ok = id >= 0 && i1 < t.n1 && i2 < t.n2;
    bad |= !ok;
    float4 a = z, b = z;
    if (ok) {
      a = load_row4<XF>(t.T1, t.S1, t.M1, i1 * De + q * 4);
      b = load_row4<XF>(t.T2, t.S2, t.M2, i2 * De + q * 4);
    }
    int64_t o1, o2;
    out_offsets(op, i, F, De, o1, o2);
    if (op == OP_CAT) {
      st4(out + o1 + q * 4, a);
      st4(out + o2 + q * 4, b);
    } else if (op == OP_MULT) {
      st4(out + o1 + q * 4, make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w));
    } else {
      st4(out + o1 + q * 4, make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w));
    }
  }
  if (bad && err) atomicOr(err, MI_IDX_OUT_OF_RANGE);
}

__device__ __forceinline__ float load_el(int xf, const float *T, const float *S, const uint8_t *M, int64_t o) {
  const float w = T[o];
  if (xf == XF_SOFT) return soft_(w, S[o]);
  if (xf == XF_MASK) return M[o] ? w : 0.f;
  return w;
}

// any De: one thread per output element of the lookup.
__global__ __launch_bounds__(kBlock) void k_dual_fwd_anyD(const int64_t *__restrict__ idx, DualTables t,
                                                          float *__restrict__ out, int64_t n, int F, int De,
                                                          int op, int xf, int *err) {
  const int64_t total = n * De;
  int bad = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / De;
    const int d = (int)(e % De);
    const int64_t id = idx[i];
    int64_t i1, i2;
    split_id(id, t, i1, i2);
    const bool ok = id >= 0 && i1 < t.n1 && i2 < t.n2;
    bad |= !ok;
    const float a = ok ? load_el(xf, t.T1, t.S1, t.M1, i1 * De + d) : 0.f;
    const float b = ok ? load_el(xf, t.T2, t.S2, t.M2, i2 * De + d) : 0.f;
    int64_t o1, o2;
    out_offsets(op, i, F, De, o1, o2);
    if (op == OP_CAT) {
      out[o1 + d] = a;
      out[o2 + d] = b;
    } else {
      out[o1 + d] = (op == OP_MULT) ? a * b : a + b;
    }
  }
  if (bad && err) atomicOr(err, MI_IDX_OUT_OF_RANGE);
}

// Backward: one thread per element; dense float-atomic scatter into the table gradients.
// Tables whose gradient fits `lds_floats` are first summed per workgroup in LDS (ds_add_f32) and
// flushed once: QR's remainder table can have as few as 2 rows, and every lookup adding into
// one row of global memory runs an order of magnitude under the atomic rate
// (MI355X_MICROARCH.md, "Global float atomics", contention row).
struct DualGrads {
  float *gT1, *gT2;  // [n1,De], [n2,De]
  float *gS1, *gS2;  // XF_SOFT only
  // row form of table 2's gradient (xf == XF_NONE): g2vals[i, :] = the contribution of lookup i, rows2[i] = its row of
  // table 2 (-1 for an id out of range) — an uncoalesced COO gradient, plain coalesced stores instead of float atomics into
  // scattered rows and no [n2, De] zero-fill.  When set, gT2 is not touched.
  float *g2vals;
  int64_t *rows2;
};

constexpr int kLdsAccFloats = 8192;  // 32 KiB

// Small fields (optional; xf == XF_NONE, idx is [B, F]).  A field with a handful of values sends all B lookups of its column
// to a handful of rows of table 2: same-address float atomics run one after the other (~13 ns each) — Avazu has nine such
// fields of 22, 12.5 us of the 34.5 us this kernel took at the C3 shape (tools/probe_dual_bwd.py).  Their table-2
// contributions are summed per field instead: kSmallParts extra workgroups per small field walk that field's column,
// every thread keeps one register per row of the field's span (<= kSmallRows rows from field_row0[f]), the workgroup
// joins its lookup slots through LDS and adds each (row, d) once — kSmallParts same-address atomics instead of B.
// The main workgroups skip table 2 for those fields (is_small[f]); an id outside its field's span still goes the
// atomic way, so any id is handled.
constexpr int kSmallRows = 16, kSmallParts = 32;
struct SmallFields {
  const int32_t *fields;     // [n] field indices
  const int64_t *row0;       // [F] first table-2 row of every field
  const uint8_t *is_small;   // [F]
  int n, nmain;              // small fields; workgroups of the main part (the extra ones follow)
  int64_t B;
};

// Row form of table 2's gradient on float4 rows (plain tables, De = 4 LPR, table 1 of <= 4 rows: QR with a small divider).
// A thread owns (lookup, float4 column q) for TWO lookups per trip — both lookups' id, gradient and table-2 row loads are
// issued before any is used — writes table 2's value rows with plain float4 stores and keeps table 1's sums in registers
// (one float4 per row of table 1: a thread's q never changes), joined per workgroup through LDS and added with ONE
// coalesced atomic instruction per row.  (The element-per-thread kernel below ran this case as a chain of dependent
// round trips, 5-6 per thread: 24 us at the C3 shape for 20 MB of traffic.)
// ws (nullable): [1 + gridDim.x * n1 * De] floats, word 0 an arrival ticket that is zero between launches.  With it the
// workgroups' table-1 sums are joined by the LAST workgroup to arrive instead of by float atomics: 512 workgroups adding to
// the same 32 addresses serialise at ~26 ns apiece (13 us of this kernel).  Hand-off as MI355X_MICROARCH.md prescribes:
// device-scope (sc1) stores, every storing wave's s_waitcnt vmcnt(0), workgroup barrier, one agent-scope ticket add; the
// workgroup whose add came back last reads with device-scope loads behind a barrier that lane joins.  Fixed order of
// additions: deterministic.
template <int LPR>
__global__ __launch_bounds__(kBlock) void k_dual_bwd_rows4(const int64_t *__restrict__ idx, DualTables t,
                                                           const float *__restrict__ g, float *__restrict__ gT1,
                                                           float *__restrict__ g2vals, int64_t *__restrict__ rows2,
                                                           int64_t n, int F, int op, float *__restrict__ ws) {
  constexpr int De = LPR * 4;
  constexpr int U = 2;
  __shared__ float4 red[4][kBlock];
  const int q = threadIdx.x % LPR;
  const int64_t item0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LPR;
  const int64_t stride = (int64_t)gridDim.x * kBlock / LPR;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 r1[4] = {z, z, z, z};
  for (int64_t i0 = item0; i0 < n; i0 += U * stride) {
    int64_t id[U], i1[U], i2[U], o1[U], o2[U];
    bool ok[U];
    float4 go[U], ga[U], e1[U], e2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      id[u] = i < n ? idx[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      i1[u] = i2[u] = 0;
      if (id[u] >= 0) split_id(id[u], t, i1[u], i2[u]);
      ok[u] = i < n && id[u] >= 0 && i1[u] < t.n1 && i2[u] < t.n2;
      out_offsets(op, i < n ? i : 0, F, De, o1[u], o2[u]);
      go[u] = ok[u] ? ld4(g + o1[u] + q * 4) : z;
      ga[u] = (ok[u] && op == OP_CAT) ? ld4(g + o2[u] + q * 4) : z;
      e1[u] = (ok[u] && op == OP_MULT) ? ld4(t.T1 + i1[u] * De + q * 4) : z;
      e2[u] = (ok[u] && op == OP_MULT) ? ld4(t.T2 + i2[u] * De + q * 4) : z;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i >= n) continue;
      float4 g1, g2;
      if (op == OP_CAT) { g1 = go[u]; g2 = ga[u]; }
      else if (op == OP_MULT) {
        g1 = make_float4(go[u].x * e2[u].x, go[u].y * e2[u].y, go[u].z * e2[u].z, go[u].w * e2[u].w);
        g2 = make_float4(go[u].x * e1[u].x, go[u].y * e1[u].y, go[u].z * e1[u].z, go[u].w * e1[u].w);
      } else { g1 = g2 = go[u]; }
      st4(g2vals + i * De + q * 4, ok[u] ? g2 : z);
      if (q == 0) rows2[i] = ok[u] ? i2[u] : 0;      // (an id out of range: row 0 with a zero value row)
      if (ok[u]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float m = i1[u] == k ? 1.f : 0.f;
          r1[k].x = fmaf(m, g1.x, r1[k].x); r1[k].y = fmaf(m, g1.y, r1[k].y);
          r1[k].z = fmaf(m, g1.z, r1[k].z); r1[k].w = fmaf(m, g1.w, r1[k].w);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = r1[k];
  __syncthreads();
  // thread (k, q) for k < n1, q < LPR adds up the workgroup's kBlock / LPR lookup slots of its float4 column
  if (threadIdx.x < (int)t.n1 * LPR) {
    const int k = threadIdx.x / LPR, qq = threadIdx.x % LPR;
    float4 s = z;
    for (int l = qq; l < kBlock; l += LPR) {
      const float4 p = red[k][l];
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    if (ws) {
      float *o = ws + 1 + ((int64_t)blockIdx.x * t.n1 + k) * De + qq * 4;
      __hip_atomic_store(o + 0, s.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(o + 1, s.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(o + 2, s.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(o + 3, s.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      float *o = gT1 + (int64_t)k * De + qq * 4;
      atomicAdd(o + 0, s.x); atomicAdd(o + 1, s.y); atomicAdd(o + 2, s.z); atomicAdd(o + 3, s.w);
    }
  }
  if (!ws) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int last;
  if (threadIdx.x == 0)
    last = __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(ws), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  // (several workgroups share a CU here, which none of the guide's measured sc1-only hand-offs covers: the last workgroup also
  //  takes the agent-scope acquire — one L1 invalidate in one workgroup, ~2 us)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // the last workgroup: value j of the n1 * De sums by threads j, j + nv, ... (each a share of the workgroups), joined in LDS
  const int nv = (int)t.n1 * De;
  float *redf = reinterpret_cast<float *>(red);
  const int j = threadIdx.x % nv, part = threadIdx.x / nv, parts = kBlock / nv;
  float a = 0.f;
  if (part < parts) {
    // 16 loads in flight per thread (a loop of single device-scope loads ran them one round trip after the other: 64 trips,
    // ~25 us in the last workgroup alone)
    constexpr int CH = 16;
    for (int b0 = part; b0 < (int)gridDim.x; b0 += parts * CH) {
      float v[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int b = min(b0 + u * parts, (int)gridDim.x - 1);
        v[u] = __hip_atomic_load(ws + 1 + (int64_t)b * nv + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (b0 + u * parts < (int)gridDim.x) a += v[u];
    }
  }
  __syncthreads();                       // (red is re-used)
  redf[threadIdx.x] = a;
  __syncthreads();
  if ((int)threadIdx.x < nv) {
    float v = 0.f;
    for (int p = 0; p < parts; ++p) v += redf[p * nv + threadIdx.x];
    gT1[threadIdx.x] = v;            // WRITTEN, not added to: the caller's gT1 need not be zeroed in this form
  }
  if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<unsigned *>(ws), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(kBlock) void k_dual_bwd(const int64_t *__restrict__ idx, DualTables t,
                                                     const float *__restrict__ g, DualGrads gr, int64_t n,
                                                     int F, int De, int op, int xf, int lds1, int lds2, SmallFields sm) {
  __shared__ float acc[kLdsAccFloats];
  // lds1 == 2: table 1 has <= 4 rows (QR `divider: 2`: two) and a thread keeps its column d for the whole launch (De divides
  // the workgroup): the table-1 gradient is summed in REGISTERS, one per row, and joined once per workgroup — every element
  // used to do an LDS float atomic on one of n1 * De addresses, and ds_add_f32 executes a wave's lanes one after the other
  // (profiles/r03_lds_atomic_probe.txt): that, not table 2's scattered atomics, was most of this kernel's 26 us
  const bool reg1 = lds1 == 2;
  float r1[4] = {0.f, 0.f, 0.f, 0.f};
  if (reg1) lds1 = 0;
  if (sm.n > 0 && (int)blockIdx.x >= sm.nmain) {
    // ---- a small field's column: thread (lookup slot lk, column d), registers r[row of the field's span]
    const int sb = blockIdx.x - sm.nmain, f = sm.fields[sb / kSmallParts], part = sb % kSmallParts;
    const int LK = kBlock / De, lk = threadIdx.x / De, d = threadIdx.x % De;
    const int64_t base_row = sm.row0[f];
    float r[kSmallRows];
#pragma unroll
    for (int k = 0; k < kSmallRows; ++k) r[k] = 0.f;
    for (int64_t b = (int64_t)part * LK + lk; b < sm.B; b += (int64_t)kSmallParts * LK) {
      const int64_t i = b * F + f;
      const int64_t id = idx[i];
      int64_t i1, i2;
    split_id(id, t, i1, i2);
      if (!(id >= 0 && i1 < t.n1 && i2 < t.n2)) continue;
      int64_t o1, o2;
      out_offsets(op, i, F, De, o1, o2);
      float g2;
      if (op == OP_CAT) g2 = g[o2 + d];
      else if (op == OP_MULT) g2 = g[o1 + d] * t.T1[i1 * De + d];
      else g2 = g[o1 + d];
      const int64_t rel = i2 - base_row;
      if ((uint64_t)rel < (uint64_t)kSmallRows) {
#pragma unroll
        for (int k = 0; k < kSmallRows; ++k) r[k] += (rel == k) ? g2 : 0.f;
      } else {
        atomicAdd(gr.gT2 + i2 * De + d, g2);          // an id outside its field's span
      }
    }
#pragma unroll
    for (int k = 0; k < kSmallRows; ++k) acc[(lk * kSmallRows + k) * De + d] = r[k];
    __syncthreads();
    for (int o = threadIdx.x; o < kSmallRows * De; o += kBlock) {
      const int k = o / De, dd = o % De;
      float v = 0.f;
      for (int l = 0; l < LK; ++l) v += acc[(l * kSmallRows + k) * De + dd];
      if (v != 0.f && base_row + k < t.n2) atomicAdd(gr.gT2 + (base_row + k) * De + dd, v);
    }
    return;
  }
  // acc layout: [table-1 grads | table-2 grads] for whichever table is LDS-accumulated
  const int n1e = lds1 ? (int)(t.n1 * De) : 0;
  const int n2e = lds2 ? (int)(t.n2 * De) : 0;
  for (int k = threadIdx.x; k < n1e + n2e; k += blockDim.x) acc[k] = 0.f;
  if (n1e + n2e) __syncthreads();

  const int64_t total = n * De;
  const int64_t nblk = sm.n > 0 ? sm.nmain : gridDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += nblk * blockDim.x) {
    const int64_t i = e / De;
    const int d = (int)(e % De);
    const int64_t id = idx[i];
    int64_t i1, i2;
    split_id(id, t, i1, i2);
    if (!(id >= 0 && i1 < t.n1 && i2 < t.n2)) {
      if (gr.g2vals) {
        gr.g2vals[e] = 0.f;
        if (d == 0) gr.rows2[i] = 0;
      }
      continue;
    }
    int64_t o1, o2;
    out_offsets(op, i, F, De, o1, o2);
    const int64_t a1 = i1 * De + d, a2 = i2 * De + d;
    float g1, g2, s1 = 0.f, s2 = 0.f;
    if (op == OP_CAT) {
      g1 = g[o1 + d];
      g2 = g[o2 + d];
    } else if (op == OP_MULT) {
      const float go = g[o1 + d];
      g1 = go * load_el(xf, t.T2, t.S2, t.M2, a2);
      g2 = go * load_el(xf, t.T1, t.S1, t.M1, a1);
    } else {
      g1 = g2 = g[o1 + d];
    }
    if (xf == XF_SOFT) {
      // y = sign(w) relu(|w| - sig(s)):  dy/dw = [|w| > sig(s)],  dy/ds = -sign(w) [..] sig (1 - sig)
      const float w1 = t.T1[a1], w2 = t.T2[a2];
      const float t1 = sigmoidf_(t.S1[a1]), t2 = sigmoidf_(t.S2[a2]);
      const float k1 = (fabsf(w1) - t1 > 0.f) ? 1.f : 0.f, k2 = (fabsf(w2) - t2 > 0.f) ? 1.f : 0.f;
      s1 = -g1 * signf_(w1) * k1 * t1 * (1.f - t1);
      s2 = -g2 * signf_(w2) * k2 * t2 * (1.f - t2);
      g1 *= k1;
      g2 *= k2;
    } else if (xf == XF_MASK) {
      g1 = t.M1[a1] ? g1 : 0.f;
      g2 = t.M2[a2] ? g2 : 0.f;
    }
    if (reg1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) r1[k] += (i1 == k) ? g1 : 0.f;
    } else if (lds1) atomicAdd(&acc[a1], g1);
    else atomicAdd(gr.gT1 + a1, g1);
    if (gr.g2vals) {                                // row form: one coalesced store per element, the row id once per lookup
      gr.g2vals[e] = g2;
      if (d == 0) gr.rows2[i] = i2;
    } else if (sm.n > 0 && sm.is_small[i % F]) {}   // table 2 of a small field: the extra workgroups' job
    else if (lds2) atomicAdd(&acc[n1e + a2], g2);
    else atomicAdd(gr.gT2 + a2, g2);
    if (xf == XF_SOFT) {
      atomicAdd(gr.gS1 + a1, s1);
      atomicAdd(gr.gS2 + a2, s2);
    }
  }
  if (reg1) {
    __syncthreads();                       // (acc may hold table-2 sums of the loop above only when lds2; reg1 launches pass lds2 = 0)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k * kBlock + threadIdx.x] = r1[k];
    __syncthreads();
    const int per = kBlock / De;
    for (int o = threadIdx.x; o < (int)t.n1 * De; o += kBlock) {
      const int k = o / De, d = o % De;
      float v = 0.f;
      for (int l = 0; l < per; ++l) v += acc[k * kBlock + l * De + d];
      if (v != 0.f) atomicAdd(gr.gT1 + o, v);
    }
  }
  if (n1e + n2e) {
    __syncthreads();
    for (int k = threadIdx.x; k < n1e; k += blockDim.x)
      if (acc[k] != 0.f) atomicAdd(gr.gT1 + k, acc[k]);
    for (int k = threadIdx.x; k < n2e; k += blockDim.x)
      if (acc[n1e + k] != 0.f) atomicAdd(gr.gT2 + k, acc[n1e + k]);
  }
}

// ------------------------------------------------------------- CSR-pruned rows ------
// K1/K2 of the reference: out[i,:] = 0; out[i, col[j]] = values[j] for j in the CSR row ids[i].
// One thread per output element: lane d scans the (short) row for its own column, so the row is
// written once, coalesced, with no zero-fill pass and no write race.  Later duplicates win, as in
// the reference's serial loop.
__global__ __launch_bounds__(kBlock) void k_csr_rows(const float *__restrict__ values,
                                                     const int64_t *__restrict__ crow,
                                                     const int64_t *__restrict__ col,
                                                     const int64_t *__restrict__ ids, float *__restrict__ out,
                                                     int64_t n, int D, int64_t N, int *err) {
  const int64_t total = n * D;
  int bad = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / D;
    const int d = (int)(e % D);
    const int64_t row = ids[i];
    float v = 0.f;
    if ((uint64_t)row < (uint64_t)N) {
      const int64_t lo = crow[row], hi = crow[row + 1];
      for (int64_t j = lo; j < hi; ++j)
        if (col[j] == d) v = values[j];
    } else {
      bad = 1;
    }
    out[e] = v;
  }
  if (bad && err) atomicOr(err, MI_IDX_OUT_OF_RANGE);
}

// ------------------------------------------------------------- DHE hash -------------
// out[i,k] = 2 * (((a_k*(id_i+prefix+1) + b_k) mod p_k) mod m) / (m-1) - 1, int64 FLOOR mod as
// torch's % (a_k, b_k may be negative), then the reference's fp32 op order: int -> float,
// true-divide by float(m-1), *2, -1 (dh_embedding.py:229-234).
__device__ __forceinline__ int64_t floormod(int64_t a, int64_t p) {
  int64_t r = a % p;
  return (r != 0 && ((r < 0) != (p < 0))) ? r + p : r;
}

__global__ __launch_bounds__(kBlock) void k_dhe_hash(const int64_t *__restrict__ ids,
                                                     const int64_t *__restrict__ slopes,
                                                     const int64_t *__restrict__ bias,
                                                     const int64_t *__restrict__ primes, float *__restrict__ out,
                                                     int64_t n, int K, int64_t prefix, int64_t m) {
  const int64_t total = n * K;
  const float denom = (float)(m - 1);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / K;
    const int k = (int)(e % K);
    const int64_t v = slopes[k] * (ids[i] + prefix + 1) + bias[k];
    const int64_t h = floormod(floormod(v, primes[k]), m);
    float f = (float)h / denom;
    f = f * 2.f;
    out[e] = f - 1.f;
  }
}

// ------------------------------------------------------------- FM over a given emb --
template <int LPR>
__global__ __launch_bounds__(kBlock) void k_fm_fwd(const float *__restrict__ emb,
                                                   const int64_t *__restrict__ rows,
                                                   const float *__restrict__ w1, const float *__restrict__ bias,
                                                   float *__restrict__ yfm, int64_t B, int F, int64_t N, int *err) {
  constexpr int RS = kWave / LPR;
  constexpr int D = LPR * 4;
  const int lane = threadIdx.x & 63;
  const int q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const float bv = bias ? bias[0] : 0.f;
  int bad = 0;
  for (int64_t b = wave0; b < B; b += nwaves) {
    const int64_t base = b * F;
    float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
    float ss = 0.f, lin = 0.f;
    for (int f = r; f < F; f += RS) {
      const float4 v = ld4(emb + (base + f) * D + q * 4);
      S.x += v.x; S.y += v.y; S.z += v.z; S.w += v.w;
      ss += dot4(v, v);
      if (q == 0) {
        const int64_t row = rows[base + f];
        if ((uint64_t)row < (uint64_t)N) lin += w1[row]; else bad = 1;
      }
    }
    S = slot_sum<LPR>(S);
    float t = (r == 0 ? dot4(S, S) : 0.f) - ss;
    t = wave_sum(0.5f * t + lin);
    if (lane == 0) yfm[b] = t + bv;
  }
  if (bad && err) atomicOr(err, MI_IDX_OUT_OF_RANGE);
}

__global__ __launch_bounds__(kBlock) void k_fm_fwd_anyD(const float *__restrict__ emb,
                                                        const int64_t *__restrict__ rows,
                                                        const float *__restrict__ w1,
                                                        const float *__restrict__ bias, float *__restrict__ yfm,
                                                        int64_t B, int F, int D, int64_t N, int *err) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const float bv = bias ? bias[0] : 0.f;
  int bad = 0;
  for (int64_t b = wave0; b < B; b += nwaves) {
    const int64_t base = b * F;
    float t = 0.f;
    for (int f = lane; f < F; f += kWave) {
      const int64_t row = rows[base + f];
      if ((uint64_t)row < (uint64_t)N) t += w1[row]; else bad = 1;
    }
    for (int d = lane; d < D; d += kWave) {
      float S = 0.f, ss = 0.f;
      for (int f = 0; f < F; ++f) {
        const float v = emb[(base + f) * D + d];
        S += v;
        ss += v * v;
      }
      t += 0.5f * (S * S - ss);
    }
    t = wave_sum(t);
    if (lane == 0) yfm[b] = t + bv;
  }
  if (bad && err) atomicOr(err, MI_IDX_OUT_OF_RANGE);
}

// ------------------------------------------------------------- single-table transformed gather
// PEP (src/models/embeddings/pep_embedding.py:82-92): out = soft(W[idx], s) with the threshold s
// broadcast over rows and/or columns (global [1], dimension [D], feature [N,1], feature_dim [N,D]):
// S element of (row, d) = S[row*srs + d*scs].  RetrainPep (:211-221): out = W[idx] * mask.
// One thread per output element (coalesced for any D); the lookup never materialises soft(W).
struct XformTable {
  const float *W;
  const float *S;
  const uint8_t *M;
  int64_t srs, scs;
  int64_t N;
};

__global__ __launch_bounds__(kBlock) void k_xform_gather_fwd(const int64_t *__restrict__ idx, XformTable t,
                                                             float *__restrict__ out, int64_t n, int D, int xf,
                                                             int *err) {
  const int64_t total = n * D;
  int bad = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx[e / D];
    const int d = (int)(e % D);
    float v = 0.f;
    if ((uint64_t)row < (uint64_t)t.N) {
      const float w = t.W[row * D + d];
      if (xf == XF_SOFT) v = soft_(w, t.S[row * t.srs + d * t.scs]);
      else if (xf == XF_MASK) v = t.M[row * D + d] ? w : 0.f;
      else v = w;
    } else {
      bad = 1;
    }
    out[e] = v;
  }
  if (bad && err) atomicOr(err, MI_IDX_OUT_OF_RANGE);
}

// gW[row,d] += g * [|w| > sig(s)] ; gS[s_off] += -g sign(w) [..] sig (1 - sig).  A threshold tensor of
// <= 8192 elements (global / dimension forms: every lookup adds into the same few words) is summed
// per workgroup in LDS first.
__global__ __launch_bounds__(kBlock) void k_xform_gather_bwd(const int64_t *__restrict__ idx, XformTable t,
                                                             const float *__restrict__ g, float *__restrict__ gW,
                                                             float *__restrict__ gS, int64_t n, int D, int xf,
                                                             int s_numel, int lds_s) {
  __shared__ float acc[kLdsAccFloats];
  if (lds_s) {
    for (int k = threadIdx.x; k < s_numel; k += blockDim.x) acc[k] = 0.f;
    __syncthreads();
  }
  const int64_t total = n * D;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx[e / D];
    const int d = (int)(e % D);
    if ((uint64_t)row >= (uint64_t)t.N) continue;
    const int64_t o = row * D + d;
    float gv = g[e];
    if (xf == XF_SOFT) {
      const int64_t so = row * t.srs + d * t.scs;
      const float w = t.W[o], th = sigmoidf_(t.S[so]);
      const float k = (fabsf(w) - th > 0.f) ? 1.f : 0.f;
      const float gs = -gv * signf_(w) * k * th * (1.f - th);
      if (lds_s) atomicAdd(&acc[so], gs); else atomicAdd(gS + so, gs);
      gv *= k;
    } else if (xf == XF_MASK) {
      gv = t.M[o] ? gv : 0.f;
    }
    atomicAdd(gW + o, gv);
  }
  if (lds_s) {
    __syncthreads();
    for (int k = threadIdx.x; k < s_numel; k += blockDim.x)
      if (acc[k] != 0.f) atomicAdd(gS + k, acc[k]);
  }
}

// ------------------------------------------------------------- quantised tables (PTQ, inference)
// src/models/embeddings/ptq_emb.py:24-25 (fp16 -> fp32) and :85-91 ((code - bias) * scale, int8/int16).
// torch computes (res - bias) in the integer type promoted with the int bias tensor, then * scale
// (fp32): code and bias are small integers, the subtraction is exact in int32.
enum { Q_FP16 = 1, Q_INT8 = 2, Q_INT16 = 3 };

__global__ __launch_bounds__(kBlock) void k_gather_rows_q(const int64_t *__restrict__ idx, const void *__restrict__ W,
                                                          int qtype, const float *__restrict__ scale,
                                                          const void *__restrict__ bias, float *__restrict__ out,
                                                          int64_t n, int D, int64_t N, int *err) {
  const int64_t total = n * D;
  int bad = 0;
  float sc = 1.f;
  int bi = 0;
  if (qtype != Q_FP16) {
    sc = scale[0];
    bi = (qtype == Q_INT8) ? (int)reinterpret_cast<const int8_t *>(bias)[0]
                           : (int)reinterpret_cast<const int16_t *>(bias)[0];
  }
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx[e / D];
    float v = 0.f;
    if ((uint64_t)row < (uint64_t)N) {
      const int64_t o = row * D + e % D;
      if (qtype == Q_FP16) v = __half2float(reinterpret_cast<const __half *>(W)[o]);
      else if (qtype == Q_INT8) v = (float)((int)reinterpret_cast<const int8_t *>(W)[o] - bi) * sc;
      else v = (float)((int)reinterpret_cast<const int16_t *>(W)[o] - bi) * sc;
    } else {
      bad = 1;
    }
    out[e] = v;
  }
  if (bad && err) atomicOr(err, MI_IDX_OUT_OF_RANGE);
}

inline bool vec_ok(int D) { return D >= 4 && D <= 256 && (D & 3) == 0 && ((D >> 2) & ((D >> 2) - 1)) == 0; }
inline int grid_for_elems(int64_t total) {
  int64_t g = (total + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  if (g > kMaxGrid) g = kMaxGrid;
  return (int)g;
}

}  // namespace

extern "C" {

int mi_dual_gather_fwd(const int64_t *idx, const float *T1, const float *T2, const float *S1,
                       const float *S2, const uint8_t *M1, const uint8_t *M2, float *out, int64_t n,
                       int32_t F, int32_t De, int64_t n1, int64_t n2, int64_t mod1, int64_t div2, int32_t op,
                       int32_t xform, int32_t *err, void *stream) {
  return mi_dual_gather_fwd_off(idx, nullptr, nullptr, T1, T2, S1, S2, M1, M2, out, n, F, De, n1, n2, mod1, div2, op, xform, err,
                                stream);
}

int mi_dual_gather_fwd_off(const int64_t *idx, const int64_t *offsets, int64_t *rows_out, const float *T1, const float *T2,
                           const float *S1, const float *S2, const uint8_t *M1, const uint8_t *M2, float *out, int64_t n,
                           int32_t F, int32_t De, int64_t n1, int64_t n2, int64_t mod1, int64_t div2, int32_t op,
                           int32_t xform, int32_t *err, void *stream) {
  if (n < 0 || F <= 0 || De <= 0 || n1 <= 0 || n2 <= 0 || mod1 <= 0 || div2 <= 0) return MI_ERR_INVALID_ARG;
  if (op < OP_MULT || op > OP_CAT || xform < XF_NONE || xform > XF_MASK) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!idx || !T1 || !T2 || !out) return MI_ERR_INVALID_ARG;
  if (xform == XF_SOFT && (!S1 || !S2)) return MI_ERR_INVALID_ARG;
  if (xform == XF_MASK && (!M1 || !M2)) return MI_ERR_INVALID_ARG;
  if (op == OP_CAT && n % F != 0) return MI_ERR_INVALID_ARG;
  if (offsets && (n % F != 0 || !rows_out)) return MI_ERR_INVALID_ARG;
  DualTables t{T1, T2, S1, S2, M1, M2, n1, n2, mod1, div2};
  const bool al = aligned16(T1) && aligned16(T2) && aligned16(out) && (xform != XF_SOFT || (aligned16(S1) && aligned16(S2))) &&
                  (xform != XF_MASK || (((uintptr_t)M1 & 3) == 0 && ((uintptr_t)M2 & 3) == 0));
  if (vec_ok(De) && al) {
    const int lpr = De / 4;
    const int64_t tiles = (n + (kWave / lpr) - 1) / (kWave / lpr);
    const int grid = grid_for_waves(tiles);
#define CALL(LPR)                                                                                            \
  do {                                                                                                       \
    if (xform == XF_NONE) MI_LAUNCH("dual_gather_fwd", (k_dual_fwd<LPR, XF_NONE>), grid, kBlock, stream, idx, t, out, n, F, op, err, offsets, rows_out); \
    else if (xform == XF_SOFT) MI_LAUNCH("dual_gather_fwd", (k_dual_fwd<LPR, XF_SOFT>), grid, kBlock, stream, idx, t, out, n, F, op, err, offsets, rows_out); \
    else MI_LAUNCH("dual_gather_fwd", (k_dual_fwd<LPR, XF_MASK>), grid, kBlock, stream, idx, t, out, n, F, op, err, offsets, rows_out); \
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
    if (offsets) return MI_ERR_UNSUPPORTED;      // (the element-per-thread form takes finished row ids)
    MI_LAUNCH("dual_gather_fwd", k_dual_fwd_anyD, grid_for_elems(n * De), kBlock, stream, idx, t, out, n, F,
              De, op, xform, err);
  }
  return launch_status();
}

int mi_dual_gather_bwd(const int64_t *idx, const float *g_out, const float *T1, const float *T2,
                       const float *S1, const float *S2, const uint8_t *M1, const uint8_t *M2, float *gT1,
                       float *gT2, float *gS1, float *gS2, int64_t n, int32_t F, int32_t De, int64_t n1,
                       int64_t n2, int64_t mod1, int64_t div2, int32_t op, int32_t xform, void *stream) {
  return mi_dual_gather_bwd_fields(idx, g_out, T1, T2, S1, S2, M1, M2, gT1, gT2, gS1, gS2, n, F, De, n1, n2, mod1, div2, op, xform,
                                   nullptr, 0, nullptr, nullptr, stream);
}

int32_t mi_dual_gather_bwd_rows_overwrites(int32_t De, int64_t n1) {
  return n1 >= 1 && n1 <= 4 && De > 0 && De % 4 == 0 && De <= 256 && ((De / 4) & (De / 4 - 1)) == 0 && n1 * De <= kBlock &&
         kBlock % (n1 * De) == 0;
}

int64_t mi_dual_gather_bwd_rows_workspace_elems(int32_t De, int64_t n1) {
  return (n1 >= 1 && n1 <= 4 && De > 0 && n1 * De <= kBlock) ? 1 + (int64_t)512 * n1 * De : 0;
}

int mi_dual_gather_bwd_rows(const int64_t *idx, const float *g_out, const float *T1, const float *T2, float *gT1,
                            float *g2vals, int64_t *rows2, int64_t n, int32_t F, int32_t De, int64_t n1, int64_t n2,
                            int64_t mod1, int64_t div2, int32_t op, float *workspace, void *stream) {
  if (n < 0 || F <= 0 || De <= 0 || n1 <= 0 || n2 <= 0 || mod1 <= 0 || div2 <= 0) return MI_ERR_INVALID_ARG;
  if (op < OP_MULT || op > OP_CAT) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!idx || !g_out || !T1 || !T2 || !gT1 || !g2vals || !rows2) return MI_ERR_INVALID_ARG;
  DualTables t{T1, T2, nullptr, nullptr, nullptr, nullptr, n1, n2, mod1, div2};
  if (n1 <= 4 && De % 4 == 0 && De <= 256 && ((De / 4) & (De / 4 - 1)) == 0 && aligned16(g_out) && aligned16(T1) && aligned16(T2) &&
      aligned16(g2vals)) {
    const int lpr = De / 4;
    int64_t fg = (n * lpr / 2 + kBlock - 1) / kBlock;          // two lookups per thread and trip
    // workgroups: one per CU.  Each adds a partial to the last workgroup's join (profiles/r04_dual_rows_grid_sweep.txt: 512
    // workgroups 15.7 us, 384 14.6, 256 12.6, 192 13.0, 128 13.1 at the C3 shape); MI_DUAL_ROWS_GRID overrides for a sweep
    static const int cap = [] { const char *e = getenv("MI_DUAL_ROWS_GRID"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 512 ? v : 256; }();
    if (fg > cap) fg = cap;
    if (fg < 1) fg = 1;
    float *ws = (workspace && n1 * De <= kBlock && kBlock % (n1 * De) == 0) ? workspace : nullptr;
#define CALL(LPR) MI_LAUNCH("dual_gather_bwd_rows", (k_dual_bwd_rows4<LPR>), (int)fg, kBlock, stream, idx, t, g_out, gT1, g2vals, rows2, n, F, op, ws)
    switch (lpr) {
      case 1: CALL(1); break;
      case 2: CALL(2); break;
      case 4: CALL(4); break;
      case 8: CALL(8); break;
      case 16: CALL(16); break;
      case 32: CALL(32); break;
      default: CALL(64); break;
    }
#undef CALL
    return launch_status();
  }
  DualGrads gr{gT1, nullptr, nullptr, nullptr, g2vals, rows2};
  int lds1 = (n1 * De <= kLdsAccFloats / 2);
  int grid = grid_for_elems(n * De);
  if (n1 <= 4 && kBlock % De == 0) {                         // table 1 summed in registers (see k_dual_bwd)
    lds1 = 2;
    if (grid > 1024) grid = 1024;       // the loop is a chain of table-2 row gathers per thread: many threads in flight; the
  }                                     // join is one coalesced atomic instruction per workgroup
  else if (lds1 && grid > 512) grid = 512;
  SmallFields sm{nullptr, nullptr, nullptr, 0, grid, 0};
  MI_LAUNCH("dual_gather_bwd_rows", k_dual_bwd, grid, kBlock, stream, idx, t, g_out, gr, n, F, De, op, (int)XF_NONE, lds1, 0, sm);
  return launch_status();
}

int mi_dual_gather_bwd_fields(const int64_t *idx, const float *g_out, const float *T1, const float *T2,
                              const float *S1, const float *S2, const uint8_t *M1, const uint8_t *M2, float *gT1,
                              float *gT2, float *gS1, float *gS2, int64_t n, int32_t F, int32_t De, int64_t n1,
                              int64_t n2, int64_t mod1, int64_t div2, int32_t op, int32_t xform,
                              const int32_t *small_fields, int32_t n_small, const int64_t *field_row0,
                              const uint8_t *is_small, void *stream) {
  if (n < 0 || F <= 0 || De <= 0 || n1 <= 0 || n2 <= 0 || mod1 <= 0 || div2 <= 0) return MI_ERR_INVALID_ARG;
  if (op < OP_MULT || op > OP_CAT || xform < XF_NONE || xform > XF_MASK) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!idx || !g_out || !T1 || !T2 || !gT1 || !gT2) return MI_ERR_INVALID_ARG;
  if (xform == XF_SOFT && (!S1 || !S2 || !gS1 || !gS2)) return MI_ERR_INVALID_ARG;
  if (xform == XF_MASK && (!M1 || !M2)) return MI_ERR_INVALID_ARG;
  DualTables t{T1, T2, S1, S2, M1, M2, n1, n2, mod1, div2};
  DualGrads gr{gT1, gT2, gS1, gS2, nullptr, nullptr};
  // LDS pre-aggregation for small tables; fewer, fatter workgroups then bound the flush traffic
  int lds1 = (n1 * De <= kLdsAccFloats / 2), lds2 = (n2 * De <= kLdsAccFloats / 2);
  int grid = grid_for_elems(n * De);
  // (the register form of table 1's sums — lds1 = 2, mi_dual_gather_bwd_rows — measured SLOWER here, 33 vs 27 us at the C3
  //  shape: with twice the workgroups in flight the scattered table-2 atomics contend more)
  if ((lds1 || lds2) && grid > 512) grid = 512;
  SmallFields sm{nullptr, nullptr, nullptr, 0, grid, 0};
  if (n_small < 0 || (n_small > 0 && (!small_fields || !field_row0 || !is_small))) return MI_ERR_INVALID_ARG;
  // the per-field sums need the [B, F] shape, plain tables, a thread layout of whole rows, table 2 on the atomic path
  if (n_small > 0 && xform == XF_NONE && n % F == 0 && kBlock % De == 0 && kSmallRows * De * (kBlock / De) <= kLdsAccFloats && !lds2) {
    sm = SmallFields{small_fields, field_row0, is_small, n_small, grid, n / F};
    grid += n_small * kSmallParts;
  }
  MI_LAUNCH("dual_gather_bwd", k_dual_bwd, grid, kBlock, stream, idx, t, g_out, gr, n, F, De, op, xform, lds1,
            lds2, sm);
  return launch_status();
}

int mi_xform_gather_fwd(const int64_t *idx, const float *W, const float *S, const uint8_t *M, int64_t srs,
                        int64_t scs, float *out, int64_t n, int32_t D, int64_t N, int32_t xform, int32_t *err,
                        void *stream) {
  if (n < 0 || D <= 0 || N < 0 || xform < XF_NONE || xform > XF_MASK) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!idx || !W || !out || (xform == XF_SOFT && !S) || (xform == XF_MASK && !M)) return MI_ERR_INVALID_ARG;
  XformTable t{W, S, M, srs, scs, N};
  MI_LAUNCH("xform_gather_fwd", k_xform_gather_fwd, grid_for_elems(n * D), kBlock, stream, idx, t, out, n, D, xform,
            err);
  return launch_status();
}

int mi_xform_gather_bwd(const int64_t *idx, const float *g_out, const float *W, const float *S, const uint8_t *M,
                        int64_t srs, int64_t scs, float *gW, float *gS, int64_t s_numel, int64_t n, int32_t D,
                        int64_t N, int32_t xform, void *stream) {
  if (n < 0 || D <= 0 || N < 0 || xform < XF_NONE || xform > XF_MASK) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!idx || !g_out || !W || !gW) return MI_ERR_INVALID_ARG;
  if (xform == XF_SOFT && (!S || !gS || s_numel <= 0)) return MI_ERR_INVALID_ARG;
  if (xform == XF_MASK && !M) return MI_ERR_INVALID_ARG;
  XformTable t{W, S, M, srs, scs, N};
  const int lds_s = (xform == XF_SOFT && s_numel <= kLdsAccFloats) ? 1 : 0;
  int grid = grid_for_elems(n * D);
  if (lds_s && grid > 512) grid = 512;
  MI_LAUNCH("xform_gather_bwd", k_xform_gather_bwd, grid, kBlock, stream, idx, t, g_out, gW, gS, n, D, xform,
            (int)(lds_s ? s_numel : 0), lds_s);
  return launch_status();
}

int mi_gather_rows_quant(const int64_t *idx, const void *W, int32_t qtype, const float *scale, const void *bias,
                         float *out, int64_t n, int32_t D, int64_t N, int32_t *err, void *stream) {
  if (n < 0 || D <= 0 || N < 0 || qtype < Q_FP16 || qtype > Q_INT16) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!idx || !W || !out || (qtype != Q_FP16 && (!scale || !bias))) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("gather_rows_quant", k_gather_rows_q, grid_for_elems(n * D), kBlock, stream, idx, W, qtype, scale, bias,
            out, n, D, N, err);
  return launch_status();
}

int mi_csr_rows_fwd(const float *values, const int64_t *crow, const int64_t *col, const int64_t *ids,
                    float *out, int64_t n, int32_t D, int64_t N, int32_t *err, void *stream) {
  if (n < 0 || D <= 0 || N < 0) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!crow || !ids || !out) return MI_ERR_INVALID_ARG;  // values/col may be null for an all-zero table
  MI_LAUNCH("csr_rows", k_csr_rows, grid_for_elems(n * D), kBlock, stream, values, crow, col, ids, out, n, D, N,
            err);
  return launch_status();
}

int mi_dhe_hash(const int64_t *ids, const int64_t *slopes, const int64_t *bias, const int64_t *primes,
                float *out, int64_t n, int32_t K, int64_t prefix, int64_t m, void *stream) {
  if (n < 0 || K <= 0 || m <= 1) return MI_ERR_INVALID_ARG;
  if (n == 0) return MI_OK;
  if (!ids || !slopes || !bias || !primes || !out) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("dhe_hash", k_dhe_hash, grid_for_elems(n * K), kBlock, stream, ids, slopes, bias, primes, out, n, K,
            prefix, m);
  return launch_status();
}

int mi_fm_fwd(const float *emb, const int64_t *rows, const float *w1, const float *bias, float *yfm,
              int64_t B, int32_t F, int32_t D, int64_t N, int32_t *err, void *stream) {
  if (B < 0 || F < 0 || D <= 0 || N < 0) return MI_ERR_INVALID_ARG;
  if (B == 0) return MI_OK;
  if (!emb || !rows || !w1 || !yfm) return MI_ERR_INVALID_ARG;
  const int grid = grid_for_waves(B);
  if (vec_ok(D) && aligned16(emb)) {
#define CALL(LPR) MI_LAUNCH("fm_fwd", (k_fm_fwd<LPR>), grid, kBlock, stream, emb, rows, w1, bias, yfm, B, F, N, err)
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
    MI_LAUNCH("fm_fwd", k_fm_fwd_anyD, grid, kBlock, stream, emb, rows, w1, bias, yfm, B, F, D, N, err);
  }
  return launch_status();
}

}  // extern "C"
