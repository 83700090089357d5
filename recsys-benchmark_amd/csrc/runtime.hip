// runtime.hip — library info + the per-kernel HIP-event profiling ring.
#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace {
struct Rec {
  const char *name;
  hipEvent_t a, b;
};
std::mutex g_mu;
std::vector<Rec> g_ring;      // events are created once in mi_prof_enable
std::atomic<int> g_count{0};  // records used
std::atomic<int> g_cap{0};    // 0 = disabled
}  // namespace

namespace mi {
bool prof_acquire(const char *name, hipEvent_t *a, hipEvent_t *b) {
  int cap = g_cap.load(std::memory_order_relaxed);
  if (cap <= 0) return false;
  int i = g_count.fetch_add(1, std::memory_order_relaxed);
  if (i >= cap) {
    g_count.store(cap, std::memory_order_relaxed);
    return false;
  }
  g_ring[i].name = name;
  *a = g_ring[i].a;
  *b = g_ring[i].b;
  return true;
}
}  // namespace mi

namespace {
__global__ void k_empty(int) {}
}  // namespace

extern "C" {

// An empty kernel through the same launcher and profiling ring as every other entry point: what the per-dispatch clock
// (dispatch begin/end events, i.e. what rocprofv3 --kernel-trace reports) reads for a launch that does NOTHING.  bench.py
// reports it next to the kernel durations as `roofline.floor_us`.
int mi_prof_empty_launch(int32_t grid, int32_t block, void *stream) {
  if (grid < 1 || block < 1 || block > 1024) return MI_ERR_INVALID_ARG;
  MI_LAUNCH("empty", k_empty, grid, block, stream, 0);
  return mi::launch_status();
}

int mi_abi_version(void) { return MI_ABI_VERSION; }

const char *mi_strerror(int code) {
  switch (code) {
    case MI_OK: return "ok";
    case MI_ERR_INVALID_ARG: return "invalid argument (null pointer, negative size or bad enum)";
    case MI_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case MI_ERR_LAUNCH: return "HIP kernel launch failed";
    case MI_ERR_STATE: return "profiling ring not armed or index out of range";
    default: return "unknown mi355x_recsys error";
  }
}

int mi_prof_enable(int32_t capacity) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (capacity < 0) return MI_ERR_INVALID_ARG;
  g_cap.store(0);
  g_count.store(0);
  if (capacity == 0) return MI_OK;
  while ((int)g_ring.size() < capacity) {
    Rec r{nullptr, nullptr, nullptr};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess)
      return MI_ERR_LAUNCH;
    g_ring.push_back(r);
  }
  g_cap.store(capacity);
  return MI_OK;
}

int mi_prof_count(void) {
  int c = g_count.load(), cap = g_cap.load();
  return c < cap ? c : cap;
}

int mi_prof_read(int32_t i, char *name_out, float *ms_out) {
  if (!name_out || !ms_out) return MI_ERR_INVALID_ARG;
  if (i < 0 || i >= mi_prof_count()) return MI_ERR_STATE;
  const Rec &r = g_ring[i];
  if (hipEventSynchronize(r.b) != hipSuccess) return MI_ERR_LAUNCH;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return MI_ERR_LAUNCH;
  std::strncpy(name_out, r.name ? r.name : "?", 63);
  name_out[63] = 0;
  *ms_out = ms;
  return MI_OK;
}

}  // extern "C"
