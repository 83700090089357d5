// mix_expert.hip — the per-expert middle of a DCN_MixHead layer (src/models/layer_dcn.py:96-108) with the small r x r
// product riding in the epilogue of the large one.  A workgroup owns a 64-row panel and ONE expert e (tail_gemm.hpp's
// main loop, one workgroup per CU at M = 4096, E = 4):
//
//   forward   H1_e = tanh(x_l V_e)            [64, r]  main loop, K = d
//             H2_e = tanh(H1_e C_e)            second product out of LDS: K = r
//             H2g_e = H2_e * g_e(m)
//   backward  dH2g_e = dT U_e^T                [64, r]  main loop, K = d
//             dgate[m,e] = sum_k dH2g*H2 + dgs[m];  dZ2_e = dH2g_e * g_e(m) * (1 - H2_e^2)
//             dZ1_e = (dZ2_e C_e^T) * (1 - H1_e^2)      second product out of LDS
//
// instead of a batched r x r GEMM launch (10.8 / 8.0 us for 0.13 GFLOP at r = 64: all launch, prologue and epilogue) and
// a separate gate pass each way.  The second product reads its operands straight from LDS — the panel's tile as the row
// fragment, C_e (16 KB) staged once as [column][k] — with the main loop's fragment mapping, 64 MFMAs per consumer wave.
// r in {16, 32, 64}: the row sums of the gate gradient are shuffles over the r/4 lanes that hold a row.
#include "common.hpp"
#include "tail_gemm.hpp"

namespace {
using namespace mi;
using namespace tg;

constexpr int kTilePitch = BNT + 4;      // the panel tile [64][116] — same as the other epilogues
constexpr int kSPitch = 68;              // C_e as [c][k], k <= 64: 68 = 4 mod 8 floats between rows

struct ExpertArgs {
  const float *A;          // x_l / dT [M, d]
  const float *W;          // V [E, d, r] (forward) / U [E, r, d] (backward)
  const float *Cm;         // C [E, r, r]
  const float *gate;       // [M, E]
  int M, d, E;
  // forward outputs / backward inputs
  float *H1, *H2, *H2g;    // [M, E*r]
  // backward
  const float *dgs;        // [M]
  float *dgate;            // [M, E]
  float *dZ2, *dZ1;        // [M, E*r]
};

__device__ __forceinline__ void acc_to_tile(const floatx4 (&acc)[NSUB], float *T, int wave, int lane, int nsub) {
  if (wave < 4) {
    const int r = lane & 15, g = lane >> 4;
    float *row = T + (wave * 16 + r) * kTilePitch + 4 * g;
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
      if (s < nsub) vst4(row + 16 * s, make_float4(acc[s][0], acc[s][1], acc[s][2], acc[s][3]));
  }
}

// acc2[s] = sum_k T[16 wave + r][k] * S[16 s + r][k], k < 16 NS  (consumer waves; same lane mapping as tg::mma_half)
template <int NS>
__device__ __forceinline__ void small_product(floatx4 (&acc2)[NSUB], const float *T, const float *S, int wave, int lane) {
  const int r = lane & 15, g = lane >> 4;
#pragma unroll
  for (int h = 0; h < NS; ++h) {
    const float4 b = vld4(T + (wave * 16 + r) * kTilePitch + 16 * h + 4 * g);
    float4 a[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s] = vld4(S + (16 * s + r) * kSPitch + 16 * h + 4 * g);
#pragma unroll
    for (int s = 0; s < NS; ++s) acc2[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].x, b.x, acc2[s], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) acc2[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].y, b.y, acc2[s], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) acc2[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].z, b.z, acc2[s], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) acc2[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].w, b.w, acc2[s], 0, 0, 0);
  }
}

template <int R, bool BWD>
__global__ __launch_bounds__(kThreads) void k_mix_expert(ExpertArgs a) {
  constexpr int NS = R / 16, CPR = R / 4;          // 16-column sub-tiles; float4 chunks per tile row
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
  const int mt_total = (a.M + BM - 1) / BM;
  const int tile = xcd_logical(blockIdx.x, mt_total * a.E);
  if (tile < 0) return;
  const int mt = tile / a.E, e = tile % a.E;
  const int m0 = mt * BM, rows_valid = min(BM, a.M - m0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t = threadIdx.x;
  const int Er = a.E * R;
  floatx4 acc[NSUB];
#pragma unroll
  for (int s = 0; s < NSUB; ++s) acc[s] = floatx4{0.f, 0.f, 0.f, 0.f};

  const KcOperand<64, LoadPlain> opR{LoadPlain{a.A, a.d}, m0, rows_valid, a.d};
  if constexpr (!BWD) {      // Cc(c, k) = V[e][k][c]: output contiguous
    main_loop<true, true, NS>(acc, lds, 0, a.d, opR, OtOperand<R, LoadPlain>{LoadPlain{a.W + (int64_t)e * a.d * R, R}, 0, R, a.d});
  } else {                   // Cc(c, n) = U[e][c][n]: reduction contiguous
    main_loop<true, true, NS>(acc, lds, 0, a.d, opR, KcOperand<R, LoadPlain>{LoadPlain{a.W + (int64_t)e * R * a.d, a.d}, 0, R, a.d});
  }

  float *T = lds, *S = lds + 64 * kTilePitch;
  acc_to_tile(acc, T, wave, lane, NS);
  __syncthreads();
  const float *Ce = a.Cm + (int64_t)e * R * R;
  // ---- element pass 1 (all 8 waves): the tile becomes the second product's row operand; C_e goes to S as [c][k]
  for (int i = t; i < 64 * CPR; i += kThreads) {
    const int row = i / CPR, c = (i % CPR) * 4;
    const bool live = row < rows_valid;
    const int64_t o = (int64_t)(m0 + (live ? row : 0)) * Er + e * R + c;
    float4 v = vld4(T + row * kTilePitch + c);
    if constexpr (!BWD) {
      v = make_float4(tanhf(v.x), tanhf(v.y), tanhf(v.z), tanhf(v.w));
      if (live) vst4(a.H1 + o, v);
    } else {
      const float4 h2 = live ? vld4(a.H2 + o) : zero4();
      const float gt = live ? a.gate[(int64_t)(m0 + row) * a.E + e] : 0.f;
      float s = v.x * h2.x + v.y * h2.y + v.z * h2.z + v.w * h2.w;
#pragma unroll
      for (int msk = 1; msk < CPR; msk <<= 1) s += __shfl_xor(s, msk);      // the CPR lanes of a row are neighbours
      if (live && c == 0) a.dgate[(int64_t)(m0 + row) * a.E + e] = s + a.dgs[m0 + row];
      v = make_float4(v.x * gt * (1.f - h2.x * h2.x), v.y * gt * (1.f - h2.y * h2.y), v.z * gt * (1.f - h2.z * h2.z),
                      v.w * gt * (1.f - h2.w * h2.w));
      if (live) vst4(a.dZ2 + o, v);
    }
    vst4(T + row * kTilePitch + c, v);
  }
  for (int i = t; i < R * CPR; i += kThreads) {
    const int rr = i / CPR, q = (i % CPR) * 4;
    const float4 w = vld4(Ce + rr * R + q);
    if constexpr (!BWD) {    // H2 = H1 C_e: Cc(c, k) = C_e[k][c] — transpose on the way in
      S[(q + 0) * kSPitch + rr] = w.x; S[(q + 1) * kSPitch + rr] = w.y; S[(q + 2) * kSPitch + rr] = w.z; S[(q + 3) * kSPitch + rr] = w.w;
    } else {                 // dZ1 = dZ2 C_e^T: Cc(c, k) = C_e[c][k] — as stored
      vst4(S + rr * kSPitch + q, w);
    }
  }
  __syncthreads();
  floatx4 acc2[NSUB];
#pragma unroll
  for (int s = 0; s < NSUB; ++s) acc2[s] = floatx4{0.f, 0.f, 0.f, 0.f};
  if (wave < 4) small_product<NS>(acc2, T, S, wave, lane);
  __syncthreads();           // every fragment read of T is done before it is overwritten
  acc_to_tile(acc2, T, wave, lane, NS);
  __syncthreads();
  // ---- element pass 2
  for (int i = t; i < 64 * CPR; i += kThreads) {
    const int row = i / CPR, c = (i % CPR) * 4;
    if (row >= rows_valid) continue;
    const int64_t o = (int64_t)(m0 + row) * Er + e * R + c;
    float4 v = vld4(T + row * kTilePitch + c);
    if constexpr (!BWD) {
      const float gt = a.gate[(int64_t)(m0 + row) * a.E + e];
      v = make_float4(tanhf(v.x), tanhf(v.y), tanhf(v.z), tanhf(v.w));
      vst4(a.H2 + o, v);
      vst4(a.H2g + o, make_float4(v.x * gt, v.y * gt, v.z * gt, v.w * gt));
    } else {
      const float4 h1 = vld4(a.H1 + o);
      vst4(a.dZ1 + o, make_float4(v.x * (1.f - h1.x * h1.x), v.y * (1.f - h1.y * h1.y), v.z * (1.f - h1.z * h1.z),
                                  v.w * (1.f - h1.w * h1.w)));
    }
  }
}

int check_common(const void *A, const void *W, const void *Cm, const void *gate, int M, int d, int E, int r) {
  if (M < 0 || d <= 0 || E <= 0 || r <= 0) return MI_ERR_INVALID_ARG;
  if (!A || !W || !Cm || !gate) return MI_ERR_INVALID_ARG;
  if ((r != 16 && r != 32 && r != 64) || d % 4 != 0 || !aligned16(A) || !aligned16(W) || !aligned16(Cm)) return MI_ERR_UNSUPPORTED;
  return MI_OK;
}

template <bool BWD>
int launch_expert(const ExpertArgs &a, int r, void *stream) {
  const int tiles = ((a.M + BM - 1) / BM) * a.E;
  const int grid = (tiles + 7) / 8 * 8;
  const char *name = BWD ? "mix_expert_bwd" : "mix_expert_fwd";
  if (r == 16) MI_LAUNCH(name, (k_mix_expert<16, BWD>), grid, kThreads, stream, a);
  else if (r == 32) MI_LAUNCH(name, (k_mix_expert<32, BWD>), grid, kThreads, stream, a);
  else MI_LAUNCH(name, (k_mix_expert<64, BWD>), grid, kThreads, stream, a);
  return launch_status();
}

}  // namespace

extern "C" {

int mi_mix_expert_fwd(const float *x, const float *V, const float *Cm, const float *gate, float *H1, float *H2, float *H2g,
                      int32_t M, int32_t d, int32_t E, int32_t r, void *stream) {
  const int rc = check_common(x, V, Cm, gate, M, d, E, r);
  if (rc != MI_OK) return rc;
  if (M == 0) return MI_OK;
  if (!H1 || !H2 || !H2g) return MI_ERR_INVALID_ARG;
  if (!aligned16(H1) || !aligned16(H2) || !aligned16(H2g)) return MI_ERR_UNSUPPORTED;
  ExpertArgs a{};
  a.A = x; a.W = V; a.Cm = Cm; a.gate = gate; a.M = M; a.d = d; a.E = E;
  a.H1 = H1; a.H2 = H2; a.H2g = H2g;
  return launch_expert<false>(a, r, stream);
}

int mi_mix_expert_bwd(const float *dT, const float *U, const float *Cm, const float *gate, const float *H1, const float *H2,
                      const float *dgs, float *dgate, float *dZ2, float *dZ1, int32_t M, int32_t d, int32_t E, int32_t r,
                      void *stream) {
  const int rc = check_common(dT, U, Cm, gate, M, d, E, r);
  if (rc != MI_OK) return rc;
  if (M == 0) return MI_OK;
  if (!H1 || !H2 || !dgs || !dgate || !dZ2 || !dZ1) return MI_ERR_INVALID_ARG;
  if (!aligned16(H1) || !aligned16(H2) || !aligned16(dZ2) || !aligned16(dZ1)) return MI_ERR_UNSUPPORTED;
  ExpertArgs a{};
  a.A = dT; a.W = U; a.Cm = Cm; a.gate = gate; a.M = M; a.d = d; a.E = E;
  a.H1 = const_cast<float *>(H1); a.H2 = const_cast<float *>(H2);
  a.dgs = dgs; a.dgate = dgate; a.dZ2 = dZ2; a.dZ1 = dZ1;
  return launch_expert<true>(a, r, stream);
}

}  // extern "C"
