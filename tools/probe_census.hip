// probe_census.hip — where do the workgroups of a 256-workgroup launch land?  (tools/, not product code)
// Each workgroup records XCC id, SE / CU id (HW_REG_HW_ID) and its start / end s_memrealtime; prints workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define GETREG(id, off, size) __builtin_amdgcn_s_getreg(((size - 1) << 11) | ((off) << 6) | (id))
template <int LDSB>
__global__ __launch_bounds__(256) void k_census(uint32_t *out, uint64_t *tm, int spin, float *sink) {
  __shared__ float lds[LDSB / 4];
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  float acc = threadIdx.x;
  for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;
  lds[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t hw = GETREG(4, 0, 32), xcc = GETREG(20, 0, 4);
    out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc;
    tm[blockIdx.x * 2] = t0; tm[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    if (acc == 123.f) sink[0] = lds[5];
  }
}
template <int LDSB>
void run(const char *name, int grid, int spin) {
  uint32_t *d; uint64_t *t; float *s;
  CK(hipMalloc(&d, grid * 8)); CK(hipMalloc(&t, grid * 16)); CK(hipMalloc(&s, 64));
  for (int rep = 0; rep < 3; ++rep) {
    k_census<LDSB><<<grid, 256>>>(d, t, spin, s);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h(grid * 2); std::vector<uint64_t> ht(grid * 2);
    CK(hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ht.data(), t, grid * 16, hipMemcpyDeviceToHost));
    std::map<uint32_t, int> per_cu; std::map<uint32_t, int> per_xcc;
    uint64_t tmin = ~0ull, tmax = 0, smax = 0;
    for (int b = 0; b < grid; ++b) {
      const uint32_t hw = h[b * 2], xcc = h[b * 2 + 1];
      const uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++; per_xcc[xcc]++;
      tmin = ht[b * 2] < tmin ? ht[b * 2] : tmin; tmax = ht[b * 2 + 1] > tmax ? ht[b * 2 + 1] : tmax; smax = ht[b * 2] > smax ? ht[b * 2] : smax;
    }
    int hist[8] = {0};
    for (auto &kv : per_cu) hist[kv.second < 7 ? kv.second : 7]++;
    printf("%-28s grid %4d: distinct CUs %3zu  CUs with 1/2/3/4+ WGs: %d/%d/%d/%d   xcc counts:", name, grid, per_cu.size(), hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7]);
    for (auto &kv : per_xcc) printf(" %d", kv.second);
    printf("   first->last start %.2f us, span %.2f us\n", (smax - tmin) / 100.0, (tmax - tmin) / 100.0);
  }
  CK(hipFree(d)); CK(hipFree(t)); CK(hipFree(s));
}
int main() {
  run<1024>("lds 1 KB", 256, 20000);
  run<47104>("lds 46 KB", 256, 20000);
  run<47104>("lds 46 KB", 264, 20000);
  run<65536>("lds 64 KB", 256, 20000);
  run<1024>("lds 1 KB", 512, 20000);
  return 0;
}
