// probe_fork.hip — does a hipGraph run two independent branches side by side on MI355X?  (tools/, not product code)
// Two spin kernels A and B of `grid` workgroups x 512 threads with `LDSB` bytes of LDS each; timed inside a replayed graph as
// A;B on one stream and as A || B captured on two streams (fork / join by events), and the same pair on two real streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
template <int LDSB>
__global__ __launch_bounds__(512) void k_spin(float *sink, int spin) {
  __shared__ float lds[LDSB / 4];
  float acc = threadIdx.x;
  for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;
  lds[threadIdx.x] = acc;
  __syncthreads();
  if (acc == 123.f) sink[blockIdx.x] = lds[5];
}
template <int LDSB>
float graph_us(int grid, int spin, bool fork, float *sink, hipStream_t s1, hipStream_t s2) {
  const int n = 10, reps = 30;
  hipGraph_t g; hipGraphExec_t ge; hipEvent_t ef, ej;
  CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n; ++i) {
    if (fork) {
      CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0));
      k_spin<LDSB><<<grid, 512, 0, s2>>>(sink + 4096, spin);
      k_spin<LDSB><<<grid, 512, 0, s1>>>(sink, spin);
      CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0));
    } else {
      k_spin<LDSB><<<grid, 512, 0, s1>>>(sink + 4096, spin);
      k_spin<LDSB><<<grid, 512, 0, s1>>>(sink, spin);
    }
  }
  CK(hipStreamEndCapture(s1, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s1));
  CK(hipStreamSynchronize(s1));
  CK(hipEventRecord(e0, s1));
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s1));
  CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return ms * 1e3f / (n * reps);
}
template <int LDSB>
float streams_us(int grid, int spin, float *sink, hipStream_t s1, hipStream_t s2) {
  const int reps = 200;
  hipEvent_t e0, e1, ef, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, s1));
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0));
    k_spin<LDSB><<<grid, 512, 0, s2>>>(sink + 4096, spin);
    k_spin<LDSB><<<grid, 512, 0, s1>>>(sink, spin);
    CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0));
  }
  CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}
template <int LDSB>
void run(const char *name, int grid, int spin) {
  float *sink; CK(hipMalloc(&sink, 8192 * 4));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const float a = graph_us<LDSB>(grid, spin, false, sink, s1, s2), b = graph_us<LDSB>(grid, spin, true, sink, s1, s2);
  const float c = streams_us<LDSB>(grid, spin, sink, s1, s2);
  printf("%-34s grid %4d: pair in a graph, one stream %7.2f us | forked %7.2f us | two real streams %7.2f us\n", name, grid, a, b, c);
  CK(hipFree(sink)); CK(hipStreamDestroy(s1)); CK(hipStreamDestroy(s2));
}
int main() {
  run<2048>("quarter chip, 2 KB LDS", 64, 40000);
  run<2048>("whole chip once, 2 KB LDS", 256, 40000);
  run<71680>("whole chip once, 70 KB LDS", 256, 40000);
  run<2048>("quarter chip, 2 KB LDS, long", 64, 400000);
  run<71680>("whole chip once, 70 KB LDS, long", 256, 400000);
  return 0;
}
