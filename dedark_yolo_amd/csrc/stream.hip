// Cross-stream ordering for the backward pass: dy_stream_fork(from, to) makes `to` wait for the work issued so far on `from`.
#include <hip/hip_runtime.h>

#include <atomic>

#include "dy_common.h"

namespace {
constexpr int RING = 256;                 // a wait captures the event's state at call time, so the ring may wrap freely
hipEvent_t g_ring[RING];
std::atomic<int> g_made{0};
std::atomic<unsigned> g_next{0};
}  // namespace

extern "C" int dy_stream_fork(void* from, void* to) {
  if (from == to) return 0;
  if (!g_made.load(std::memory_order_acquire)) {
    static std::atomic_flag busy = ATOMIC_FLAG_INIT;
    while (busy.test_and_set(std::memory_order_acquire)) {}
    if (!g_made.load(std::memory_order_relaxed)) {
      for (int i = 0; i < RING; ++i) {
        hipError_t e = hipEventCreateWithFlags(&g_ring[i], hipEventDisableTiming);
        if (e != hipSuccess) {
          busy.clear(std::memory_order_release);
          dy_set_error("dy_stream_fork: hipEventCreate failed: %s", hipGetErrorString(e));
          return 3;
        }
      }
      g_made.store(1, std::memory_order_release);
    }
    busy.clear(std::memory_order_release);
  }
  hipEvent_t ev = g_ring[g_next.fetch_add(1, std::memory_order_relaxed) % RING];
  hipError_t e = hipEventRecord(ev, (hipStream_t)from);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)to, ev, 0);
  if (e != hipSuccess) {
    dy_set_error("dy_stream_fork: %s", hipGetErrorString(e));
    return 3;
  }
  return 0;
}
