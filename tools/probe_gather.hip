// probe_gather.hip — where do the microseconds of the gather+FM pair go at B=4096?  (tools/, not product code)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_gather.hip -o tools/bin/probe_gather && tools/bin/probe_gather
//
// Times kernel variants on the headline shape (F=26, D=16, N=33 762 577 rows = 2.16 GB table, 16 rotating id batches)
// with the dispatch's own begin/end events (what rocprofv3 --kernel-trace reports), in three cache regimes:
//   b2b     launches back to back (fresh ids every launch)
//   step    ~100 MB of unrelated streaming between launches (what the MLP tail does between fwd and bwd in a step)
//   cold    512 MB fill between launches
// Floors: an empty kernel of the same grid, an ids-only kernel (one memory round trip), a pure streaming kernel of the
// backward's byte count.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kWave = 64, kBlock = 256, kWPB = 4;
static const int64_t DIMS[26] = {1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194,
                                 27, 14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572};
constexpr int F = 26, D = 16, LPR = 4, RS = 16;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float4 ld4nt(const float *p) {
  float4 v;
  v.x = __builtin_nontemporal_load(p); v.y = __builtin_nontemporal_load(p + 1);
  v.z = __builtin_nontemporal_load(p + 2); v.w = __builtin_nontemporal_load(p + 3);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float4 slot_sum(float4 v) {
#pragma unroll
  for (int m = LPR; m < kWave; m <<= 1) {
    v.x += __shfl_xor(v.x, m); v.y += __shfl_xor(v.y, m); v.z += __shfl_xor(v.z, m); v.w += __shfl_xor(v.w, m);
  }
  return v;
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// ------------------------------------------------------------------------------------------------ init / thrash
__global__ void k_fill_hash(float *p, int64_t n, uint32_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (float)(h & 0xFFFF) * (1.0f / 65536.0f) - 0.5f;
  }
}
__global__ void k_fill_ids(int64_t *x, const int64_t *dims, int64_t B, uint32_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B * F; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
    h ^= h >> 31; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 29;
    x[i] = (int64_t)(h % (uint64_t)dims[f]);
  }
}
__global__ void k_stream(const float4 *__restrict__ a, float4 *__restrict__ b, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = a[i]; v.x += 1.f; b[i] = v;
  }
}
// one word of every 2 MiB page of [p, p+bytes), read by every workgroup slot s.t. each XCD's translation cache sees all
__global__ void k_touch_pages(const char *p, int64_t bytes, int *sink) {
  const int64_t npages = (bytes + (2 << 20) - 1) >> 21;
  int acc = 0;
  for (int64_t i = threadIdx.x; i < npages; i += blockDim.x) acc += *reinterpret_cast<const int *>(p + (i << 21));
  if (acc == 0x7fffffff) sink[0] = acc;
}

// ------------------------------------------------------------------------------------------------ floors
__global__ __launch_bounds__(kBlock) void k_empty(int *sink) {
  if (threadIdx.x == 1023) sink[0] = 1;
}
__global__ __launch_bounds__(64) void k_empty64(int *sink) {
  if (threadIdx.x == 1023) sink[0] = 1;
}
__global__ __launch_bounds__(1024) void k_empty1024(int *sink) {
  if (threadIdx.x == 2047) sink[0] = 1;
}
__global__ __launch_bounds__(kBlock) void k_ids_only(const int64_t *__restrict__ idx, const int64_t *__restrict__ offsets,
                                                     int64_t *__restrict__ rows_out, int64_t B) {
  const int lane = threadIdx.x & 63, r = lane / LPR, q = lane % LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWPB + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * kWPB;
  for (int64_t b = wave0; b < B; b += nw) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int f = r + k * RS;
      if (f < F && q == 0) rows_out[b * F + f] = idx[b * F + f] + offsets[f];
    }
  }
}
// pure streaming kernel with the backward's traffic: read 2 x [B,F,D], write 1 x [B,F,D]
__global__ __launch_bounds__(kBlock) void k_stream3(const float4 *__restrict__ a, const float4 *__restrict__ b,
                                                    float4 *__restrict__ c, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 x = a[i], y = b[i];
    c[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

// ------------------------------------------------------------------------------------------------ forward variants
// MODE bit0: gather W rows + store emb; bit1: first-order gather; bit2: FM reduction + yfm; bit3: rows_out
template <int SPW, int MODE>
__global__ __launch_bounds__(kBlock) void k_fwd(const int64_t *__restrict__ idx, const int64_t *__restrict__ offsets,
                                                const float *__restrict__ W, const float *__restrict__ w1, float bias,
                                                float *__restrict__ emb, float *__restrict__ yfm,
                                                int64_t *__restrict__ rows_out, int64_t B, int64_t N) {
  constexpr int NIT = 2;
  const int lane = threadIdx.x & 63, q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWPB + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * kWPB;
  const int64_t off0 = offsets[r], off1 = (r + RS < F) ? offsets[r + RS] : 0;
  for (int64_t g = wave0; g * SPW < B; g += nw) {
    int64_t row[SPW][NIT];
    float4 v[SPW][NIT];
    float l[SPW][NIT];
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
      const int64_t b = g * SPW + s;
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int f = r + k * RS;
        row[s][k] = (f < F && b < B) ? idx[b * F + f] + (k ? off1 : off0) : -1;
      }
    }
#pragma unroll
    for (int s = 0; s < SPW; ++s)
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const bool ok = (uint64_t)row[s][k] < (uint64_t)N;
        v[s][k] = (ok && (MODE & 1)) ? ld4(W + row[s][k] * D + q * 4) : make_float4(0, 0, 0, 0);
        l[s][k] = (ok && q == 0 && (MODE & 2)) ? w1[row[s][k]] : 0.f;
      }
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
      const int64_t b = g * SPW + s;
      if (b >= B) break;
      float4 S = make_float4(0, 0, 0, 0);
      float ss = 0.f, lin = 0.f;
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int f = r + k * RS;
        if (f < F) {
          if (MODE & 1) st4(emb + (b * F + f) * D + q * 4, v[s][k]);
          if ((MODE & 8) && q == 0) rows_out[b * F + f] = row[s][k];
        }
        S.x += v[s][k].x; S.y += v[s][k].y; S.z += v[s][k].z; S.w += v[s][k].w;
        ss += dot4(v[s][k], v[s][k]);
        lin += l[s][k];
      }
      if (MODE & 4) {
        S = slot_sum(S);
        float t = (r == 0 ? dot4(S, S) : 0.f) - ss;
        t = wave_sum(0.5f * t + lin);
        if (lane == 0) yfm[b] = t + bias;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward variants
template <int SPW, bool NT>
__global__ __launch_bounds__(kBlock) void k_bwd(const float *__restrict__ emb, const float *__restrict__ g_y,
                                                const float *__restrict__ g_emb, float *__restrict__ gvals,
                                                float *__restrict__ g1vals, int64_t B) {
  constexpr int NIT = 2;
  const int lane = threadIdx.x & 63, q = lane % LPR, r = lane / LPR;
  const int64_t wave0 = (int64_t)blockIdx.x * kWPB + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * kWPB;
  const float4 z = make_float4(0, 0, 0, 0);
  for (int64_t g = wave0; g * SPW < B; g += nw) {
    float4 e[SPW][NIT], ge[SPW][NIT];
    float gy[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
      const int64_t b = g * SPW + s;
      gy[s] = b < B ? g_y[b] : 0.f;
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int f = r + k * RS;
        const bool act = f < F && b < B;
        const int64_t o = (b * F + f) * D + q * 4;
        e[s][k] = act ? (NT ? ld4nt(emb + o) : ld4(emb + o)) : z;
        ge[s][k] = act ? (NT ? ld4nt(g_emb + o) : ld4(g_emb + o)) : z;
      }
    }
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
      const int64_t b = g * SPW + s;
      if (b >= B) break;
      float4 S = z;
#pragma unroll
      for (int k = 0; k < NIT; ++k) { S.x += e[s][k].x; S.y += e[s][k].y; S.z += e[s][k].z; S.w += e[s][k].w; }
      S = slot_sum(S);
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int f = r + k * RS;
        if (f < F) {
          float4 o4;
          o4.x = ge[s][k].x + gy[s] * (S.x - e[s][k].x);
          o4.y = ge[s][k].y + gy[s] * (S.y - e[s][k].y);
          o4.z = ge[s][k].z + gy[s] * (S.z - e[s][k].z);
          o4.w = ge[s][k].w + gy[s] * (S.w - e[s][k].w);
          float *dst = gvals + (b * F + f) * D + q * 4;
          if (NT) {
            __builtin_nontemporal_store(o4.x, dst); __builtin_nontemporal_store(o4.y, dst + 1);
            __builtin_nontemporal_store(o4.z, dst + 2); __builtin_nontemporal_store(o4.w, dst + 3);
          } else {
            st4(dst, o4);
          }
          if (q == 0) g1vals[b * F + f] = gy[s];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ harness
struct Stat { double avg, med, mn; };
static Stat stats(std::vector<float> v) {
  std::sort(v.begin(), v.end());
  double s = 0; for (float x : v) s += x;
  return {s / v.size() * 1e3, v[v.size() / 2] * 1e3, v[0] * 1e3};
}

int main(int argc, char **argv) {
  const int64_t B = argc > 1 ? atoll(argv[1]) : 4096;
  const int reps = 40;
  int64_t N = 0, offs_h[F];
  for (int f = 0; f < F; ++f) { offs_h[f] = N; N += DIMS[f]; }
  float *W, *w1, *emb, *yfm, *gemb, *gy, *gvals, *g1, *junkA, *junkB;
  int64_t *ids[16], *rows_out, *offs, *dims;
  int *sink;
  const int64_t junk_floats = 128ll << 20;     // 512 MB each
  CK(hipMalloc(&W, N * D * 4)); CK(hipMalloc(&w1, N * 4));
  CK(hipMalloc(&emb, B * F * D * 4)); CK(hipMalloc(&yfm, B * 4)); CK(hipMalloc(&gemb, B * F * D * 4));
  CK(hipMalloc(&gy, B * 4)); CK(hipMalloc(&gvals, B * F * D * 4)); CK(hipMalloc(&g1, B * F * 4));
  CK(hipMalloc(&junkA, junk_floats * 4)); CK(hipMalloc(&junkB, junk_floats * 4));
  CK(hipMalloc(&rows_out, B * F * 8)); CK(hipMalloc(&offs, F * 8)); CK(hipMalloc(&dims, F * 8)); CK(hipMalloc(&sink, 64));
  CK(hipMemcpy(offs, offs_h, F * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dims, DIMS, F * 8, hipMemcpyHostToDevice));
  k_fill_hash<<<2048, 256>>>(W, N * D, 1u); k_fill_hash<<<2048, 256>>>(w1, N, 2u);
  k_fill_hash<<<2048, 256>>>(gemb, B * F * D, 3u); k_fill_hash<<<64, 256>>>(gy, B, 4u);
  k_fill_hash<<<2048, 256>>>(junkA, junk_floats, 5u);
  for (int i = 0; i < 16; ++i) { CK(hipMalloc(&ids[i], B * F * 8)); k_fill_ids<<<512, 256>>>(ids[i], dims, B, 100u + i); }
  CK(hipDeviceSynchronize());
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t ea[reps], eb[reps];
  for (int i = 0; i < reps; ++i) { CK(hipEventCreate(&ea[i])); CK(hipEventCreate(&eb[i])); }

  enum Regime { B2B, STEP, COLD, WARMTLB };
  const char *rname[] = {"b2b", "step", "cold", "step+tlbwarm"};
  auto between = [&](int regime) {
    if (regime == STEP || regime == WARMTLB) k_stream<<<2048, 256, 0, st>>>((const float4 *)junkA, (float4 *)junkB, (48ll << 20) / 16);
    if (regime == COLD) k_stream<<<2048, 256, 0, st>>>((const float4 *)junkA, (float4 *)junkB, junk_floats / 4);
    if (regime == WARMTLB) {
      k_touch_pages<<<64, 256, 0, st>>>((const char *)W, N * D * 4, sink);
      k_touch_pages<<<64, 256, 0, st>>>((const char *)w1, N * 4, sink);
    }
  };
  // launch(i, ea, eb): enqueue variant on `st` with timing events
  auto run = [&](const char *name, double bytes, std::vector<int> regimes,
                 const std::function<void(int, hipEvent_t, hipEvent_t)> &launch) {
    for (int regime : regimes) {
      for (int i = 0; i < 5; ++i) { between(regime); launch(i, nullptr, nullptr); }
      CK(hipStreamSynchronize(st));
      for (int i = 0; i < reps; ++i) { between(regime); launch(i, ea[i], eb[i]); }
      CK(hipStreamSynchronize(st));
      std::vector<float> t(reps);
      for (int i = 0; i < reps; ++i) CK(hipEventElapsedTime(&t[i], ea[i], eb[i]));
      Stat s = stats(t);
      printf("%-34s %-13s avg %6.2f  med %6.2f  min %6.2f us", name, rname[regime], s.avg, s.med, s.mn);
      if (bytes > 0) printf("   %6.0f GB/s (med)  frac %.3f", bytes / (s.med * 1e-6) / 1e9, bytes / (s.med * 1e-6) / 8e12);
      printf("\n");
      fflush(stdout);
    }
  };
#define LAUNCH(kern, grid, block, ...)                                                          \
  [&](int i, hipEvent_t a, hipEvent_t b) {                                                      \
    (void)i;                                                                                    \
    if (a) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, st, a, b, 0, __VA_ARGS__);   \
    else hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, st, __VA_ARGS__);                 \
  }
  const double fb = (12.0 * F + 8.0 * F * D + 4) * B, bb = (12.0 * F + 12.0 * F * D + 4) * B;
  const int g1k = (int)((B + 3) / 4);
  std::vector<int> all = {B2B, STEP, COLD}, two = {B2B, STEP};
  printf("B=%lld F=%d D=%d N=%lld  fwd alg bytes %.0f  bwd alg bytes %.0f\n", (long long)B, F, D, (long long)N, fb, bb);

  for (int g : {1, 8, 64, 256, 1024, 4096}) {
    char nm[64];
    snprintf(nm, sizeof nm, "empty grid=%d block=256", g);
    run(nm, 0, {B2B}, LAUNCH(k_empty, g, kBlock, sink));
  }
  run("empty grid=1024 block=64", 0, {B2B}, LAUNCH(k_empty64, 1024, 64, sink));
  run("empty grid=4096 block=64", 0, {B2B}, LAUNCH(k_empty64, 4096, 64, sink));
  run("empty grid=256 block=1024", 0, {B2B}, LAUNCH(k_empty1024, 256, 1024, sink));
  {  // wall-clock per kernel inside a replayed hipGraph (no per-dispatch events): what a step really pays per launch
    auto graph_time = [&](const char *name, int nk, const std::function<void(int)> &enqueue) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int i = 0; i < nk; ++i) enqueue(i);
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      const int R = 20;
      CK(hipEventRecord(a, st));
      for (int i = 0; i < R; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(b, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      printf("%-34s graph-wall     %6.2f us per kernel (%d kernels per graph, %d replays)\n", name, ms * 1e3 / (R * nk), nk, R);
      fflush(stdout);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    };
    graph_time("empty grid=1024", 64, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(kBlock), 0, st, sink); });
    graph_time("empty grid=1", 64, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(kBlock), 0, st, sink); });
    graph_time("fwd full spw1", 64, [&](int i) {
      hipLaunchKernelGGL((k_fwd<1, 15>), dim3(g1k), dim3(kBlock), 0, st, (const int64_t *)ids[i & 15], (const int64_t *)offs,
                         (const float *)W, (const float *)w1, 0.1f, emb, yfm, rows_out, B, N);
    });
    graph_time("fwd rows-only", 64, [&](int i) {
      hipLaunchKernelGGL((k_fwd<1, 9>), dim3(g1k), dim3(kBlock), 0, st, (const int64_t *)ids[i & 15], (const int64_t *)offs,
                         (const float *)W, (const float *)w1, 0.1f, emb, yfm, rows_out, B, N);
    });
    graph_time("bwd spw1", 64, [&](int) {
      hipLaunchKernelGGL((k_bwd<1, false>), dim3(g1k), dim3(kBlock), 0, st, (const float *)emb, (const float *)gy,
                         (const float *)gemb, gvals, g1, B);
    });
    graph_time("stream 96MB (thrash) alone", 16, [&](int) {
      k_stream<<<2048, 256, 0, st>>>((const float4 *)junkA, (float4 *)junkB, (48ll << 20) / 16);
    });
    graph_time("thrash + fwd full (pair)", 32, [&](int i) {
      k_stream<<<2048, 256, 0, st>>>((const float4 *)junkA, (float4 *)junkB, (48ll << 20) / 16);
      hipLaunchKernelGGL((k_fwd<1, 15>), dim3(g1k), dim3(kBlock), 0, st, (const int64_t *)ids[i & 15], (const int64_t *)offs,
                         (const float *)W, (const float *)w1, 0.1f, emb, yfm, rows_out, B, N);
    });
    graph_time("thrash + bwd (pair)", 32, [&](int) {
      k_stream<<<2048, 256, 0, st>>>((const float4 *)junkA, (float4 *)junkB, (48ll << 20) / 16);
      hipLaunchKernelGGL((k_bwd<1, false>), dim3(g1k), dim3(kBlock), 0, st, (const float *)emb, (const float *)gy,
                         (const float *)gemb, gvals, g1, B);
    });
    graph_time("thrash + empty1024 (pair)", 32, [&](int) {
      k_stream<<<2048, 256, 0, st>>>((const float4 *)junkA, (float4 *)junkB, (48ll << 20) / 16);
      hipLaunchKernelGGL(k_empty, dim3(1024), dim3(kBlock), 0, st, sink);
    });
  }
  run("ids_only", 0, two, LAUNCH(k_ids_only, g1k, kBlock, (const int64_t *)ids[i & 15], (const int64_t *)offs, rows_out, B));
  run("stream3 (bwd bytes) grid=2048", bb, two, LAUNCH(k_stream3, 2048, kBlock, (const float4 *)emb, (const float4 *)gemb, (float4 *)gvals, B * F * D / 4));
  run("stream3 (bwd bytes) grid=1024", bb, two, LAUNCH(k_stream3, 1024, kBlock, (const float4 *)emb, (const float4 *)gemb, (float4 *)gvals, B * F * D / 4));
  run("stream3 (bwd bytes) grid=512", bb, two, LAUNCH(k_stream3, 512, kBlock, (const float4 *)emb, (const float4 *)gemb, (float4 *)gvals, B * F * D / 4));

#define FWD(SPW, MODE, grid) LAUNCH((k_fwd<SPW, MODE>), grid, kBlock, (const int64_t *)ids[i & 15], (const int64_t *)offs, (const float *)W, (const float *)w1, 0.1f, emb, yfm, rows_out, B, N)
  run("fwd full spw1 (current)", fb, {B2B, STEP, COLD, WARMTLB}, FWD(1, 15, g1k));
  run("fwd rows-only (no w1, no FM)", fb, two, FWD(1, 9, g1k));
  run("fwd no-w1", fb, two, FWD(1, 13, g1k));
  run("fwd no-FM", fb, two, FWD(1, 11, g1k));
  run("fwd full spw2", fb, all, FWD(2, 15, (int)((B / 2 + 3) / 4)));
  run("fwd full spw4", fb, all, FWD(4, 15, (int)((B / 4 + 3) / 4)));
  run("fwd full spw1 grid=512 (loop)", fb, two, FWD(1, 15, 512));
  run("fwd full spw1 grid=256 (loop)", fb, two, FWD(1, 15, 256));

#define BWD(SPW, NT, grid) LAUNCH((k_bwd<SPW, NT>), grid, kBlock, (const float *)emb, (const float *)gy, (const float *)gemb, gvals, g1, B)
  run("bwd spw1 (current)", bb, all, BWD(1, false, g1k));
  run("bwd spw1 nt", bb, two, BWD(1, true, g1k));
  run("bwd spw2", bb, two, BWD(2, false, (int)((B / 2 + 3) / 4)));
  run("bwd spw4", bb, two, BWD(4, false, (int)((B / 4 + 3) / 4)));
  run("bwd spw1 grid=512 (loop)", bb, two, BWD(1, false, 512));
  run("bwd spw2 nt", bb, two, BWD(2, true, (int)((B / 2 + 3) / 4)));
  CK(hipDeviceSynchronize());
  return 0;
}
