// probe_mfma.hip — issue rate of v_mfma_f32_16x16x4_f32 at one wave per SIMD, with and without the LDS fragment reads of
// tail_gemm.hpp's stage (tools/, not product code).  Cycles by s_memtime inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int NSUB = 7, BK = 32;
__device__ __forceinline__ int kc_off(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 2); }

// MODE 0: operands from registers; 1: 16 ds_read_b128 per stage (swizzled, as the product); 2: same, un-swizzled layout
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_mfma(float *out, uint64_t *cyc, int stages) {
  __shared__ __attribute__((aligned(16))) float lds[2 * (64 + 112) * BK];
  for (int i = threadIdx.x; i < 2 * (64 + 112) * BK; i += WAVES * 64) lds[i] = (float)(i % 13) * 0.01f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, r = lane & 15, g = lane >> 4;
  floatx4 acc[NSUB];
#pragma unroll
  for (int s = 0; s < NSUB; ++s) acc[s] = floatx4{0.f, 0.f, 0.f, 0.f};
  float4 b[2], a[2][NSUB];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    b[h] = make_float4(lane * 0.001f, 0.5f, 0.25f, 0.125f);
#pragma unroll
    for (int s = 0; s < NSUB; ++s) a[h][s] = make_float4(s * 0.01f + lane * 0.002f, 0.3f, 0.2f, 0.1f);
  }
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int st = 0; st < stages; ++st) {
    const float *Rt = lds + (st & 1) * (64 + 112) * BK, *Ct = Rt + 64 * BK;
    if (MODE != 0) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int rr = wave * 16 + r;
        b[h] = *reinterpret_cast<const float4 *>(Rt + (MODE == 1 ? kc_off(rr, 4 * h + g) : rr * BK + (4 * h + g) * 4));
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
          const int cr = s * 16 + r;
          a[h][s] = *reinterpret_cast<const float4 *>(Ct + (MODE == 1 ? kc_off(cr, 4 * h + g) : cr * BK + (4 * h + g) * 4));
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int s = 0; s < NSUB; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[h][s].x, b[h].x, acc[s], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < NSUB; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[h][s].y, b[h].y, acc[s], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < NSUB; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[h][s].z, b[h].z, acc[s], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < NSUB; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[h][s].w, b[h].w, acc[s], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 3) __syncthreads();
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NSUB; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE, int WAVES>
void run(const char *name, int grid, int stages) {
  float *o; uint64_t *c;
  CK(hipMalloc(&o, grid * WAVES * 64 * 4)); CK(hipMalloc(&c, grid * WAVES * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_mfma<MODE, WAVES><<<grid, WAVES * 64>>>(o, c, stages);
  CK(hipEventRecord(e0));
  k_mfma<MODE, WAVES><<<grid, WAVES * 64>>>(o, c, stages);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<uint64_t> h(grid * WAVES);
  CK(hipMemcpy(h.data(), c, grid * WAVES * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double per = (double)h[h.size() / 2] / (stages * 56.0);
  printf("%-44s grid %3d waves %d: median %6.1f s_memtime ticks per MFMA (56 per stage), kernel %.1f us -> %.1f TFLOP/s\n", name, grid, WAVES, per, ms * 1e3,
         (double)grid * WAVES * stages * 56 * 2048.0 / (ms * 1e-3) / 1e12);
  CK(hipFree(o)); CK(hipFree(c));
}
int main() {
  run<0, 4>("registers only", 256, 2000);
  run<0, 4>("registers only", 1, 2000);
  run<1, 4>("16 ds_read_b128/stage, swizzled", 256, 2000);
  run<2, 4>("16 ds_read_b128/stage, plain rows", 256, 2000);
  run<0, 8>("registers only, 2 waves/SIMD", 256, 2000);
  run<1, 8>("16 ds_read_b128/stage swizzled, 2 waves/SIMD", 256, 2000);
  return 0;
}
