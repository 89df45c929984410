// Cross-stream ordering for the backward pass: dy_stream_fork(from, to) makes `to` wait for the work issued so far on `from`.
#include <hip/hip_runtime.h>

#include <atomic>

#include "dy_common.h"

namespace {
constexpr int RING = 256;                 // a wait captures the event's state at call time, so the ring may wrap freely
constexpr int MAXDEV = 16;                // one ring per device ordinal: an event can only be recorded on a stream of ITS device
hipEvent_t g_ring[MAXDEV][RING];
std::atomic<int> g_made[MAXDEV];
std::atomic<unsigned> g_next{0};
}  // namespace

extern "C" int dy_stream_fork(void* from, void* to) {
  if (from == to) return 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) {
    dy_set_error("dy_stream_fork: no current device or ordinal >= %d", MAXDEV);
    return 3;
  }
  std::atomic<int>& made = g_made[dev];
  if (!made.load(std::memory_order_acquire)) {
    static std::atomic_flag busy = ATOMIC_FLAG_INIT;
    while (busy.test_and_set(std::memory_order_acquire)) {}
    if (!made.load(std::memory_order_relaxed)) {
      for (int i = 0; i < RING; ++i) {
        hipError_t e = hipEventCreateWithFlags(&g_ring[dev][i], hipEventDisableTiming);
        if (e != hipSuccess) {
          busy.clear(std::memory_order_release);
          dy_set_error("dy_stream_fork: hipEventCreate failed: %s", hipGetErrorString(e));
          return 3;
        }
      }
      made.store(1, std::memory_order_release);
    }
    busy.clear(std::memory_order_release);
  }
  hipEvent_t ev = g_ring[dev][g_next.fetch_add(1, std::memory_order_relaxed) % RING];
  hipError_t e = hipEventRecord(ev, (hipStream_t)from);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)to, ev, 0);
  if (e != hipSuccess) {
    dy_set_error("dy_stream_fork: %s", hipGetErrorString(e));
    return 3;
  }
  return 0;
}
