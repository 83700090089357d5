// probe_lds_atomic.hip — what does ds_add_f32 cost?  (tools/, not product code)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_lds_atomic.hip -o tools/bin/probe_lds_atomic
// One workgroup of 512 threads per CU; every lane issues N LDS operations on a 64 KiB tile; patterns:
//   0 distinct addresses per lane, stride 1 float (conflict-free)          1 stride 4 floats (float4-like, 2 passes)
//   2 all 4 lane groups of a wave on the SAME 16 x 4 floats (same-address)  3 random rows (row of 64 floats per 16 lanes)
// Ops: ds_add_f32 (no return), plain read-add-write (ds_read_b32 + ds_write_b32), ds_add_u32 (integer atomic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int OP, int PAT>
__global__ __launch_bounds__(512) void k(float *out, int iters, const int *rnd) {
  extern __shared__ float t[];
  for (int i = threadIdx.x; i < 16384; i += 512) t[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane & 15, g = lane >> 4;
  float v = 1.0f + lane;
  for (int it = 0; it < iters; ++it) {
    int a;
    if (PAT == 0) a = (wave * 64 + lane + it * 512) & 16383;
    else if (PAT == 1) a = ((wave * 64 + lane) * 4 + it * 2048) & 16383;
    else if (PAT == 2) a = (q * 4 + (it & 255) * 64) & 16383;
    else a = ((rnd[(it * 32 + wave * 4 + g) & 65535] & 255) * 64 + q * 4) & 16383;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int aj = PAT == 0 ? (a + j * 4096) & 16383 : a + j;
      if (OP == 0) atomicAdd(&t[aj], v);
      else if (OP == 1) t[aj] += v;
      else atomicAdd(reinterpret_cast<unsigned *>(&t[aj]), 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = t[5];
}

int main() {
  float *out; int *rnd;
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&rnd, 65536 * 4));
  int *h = (int *)malloc(65536 * 4);
  for (int i = 0; i < 65536; ++i) h[i] = rand();
  CK(hipMemcpy(rnd, h, 65536 * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 2000;
  const char *opn[] = {"ds_add_f32", "read+add+write", "ds_add_u32"};
  const char *pn[] = {"stride1 distinct", "stride4 distinct", "4 groups same row", "random rows"};
#define RUN(OP, PAT)                                                                                      \
  {                                                                                                       \
    hipLaunchKernelGGL((k<OP, PAT>), dim3(256), dim3(512), 65536, 0, out, 10, rnd);                       \
    CK(hipEventRecord(a));                                                                                \
    hipLaunchKernelGGL((k<OP, PAT>), dim3(256), dim3(512), 65536, 0, out, iters, rnd);                    \
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));                                                    \
    float ms; CK(hipEventElapsedTime(&ms, a, b));                                                         \
    const double wave_instr = (double)iters * 4 * 8;       /* per CU */                                    \
    printf("%-16s %-20s %8.1f us  %6.1f ns per wave-instruction per CU (8 waves)  = %5.1f cycles@2.1GHz\n", opn[OP], pn[PAT], ms * 1e3, \
           ms * 1e6 / wave_instr, ms * 1e6 / wave_instr * 2.1);                                           \
  }
  RUN(0, 0) RUN(0, 1) RUN(0, 2) RUN(0, 3)
  RUN(1, 0) RUN(1, 1) RUN(1, 3)
  RUN(2, 0) RUN(2, 1) RUN(2, 2) RUN(2, 3)
  return 0;
}
