// Shared host-side helpers for the gtx runtime (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace gtx {

// Flags for the events host threads wait on (detector / stabilizer / GMC results). Blocking waits let a waiting
// thread sleep instead of spinning: one process per GPU with three host stages each would otherwise burn 3 cores
// per rank just waiting (the GPU boxes give a job a CPU quota). GTX_SPIN_WAIT=1 keeps the default spinning waits.
inline unsigned wait_event_flags(bool timing) {
  static const bool spin = [] { const char* e = getenv("GTX_SPIN_WAIT"); return e && e[0] == '1'; }();
  return (spin ? 0u : (unsigned)hipEventBlockingSync) | (timing ? 0u : (unsigned)hipEventDisableTiming);
}


// Error transport: C++ exceptions never cross the C ABI; gtx_api.cpp catches
// them, stores the text in a thread-local buffer and returns a negative code.
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw Error(code, buf);
}

#define GTX_HIP(expr)                                                         \
  do {                                                                        \
    hipError_t e__ = (expr);                                                  \
    if (e__ != hipSuccess)                                                    \
      ::gtx::fail(-2, "HIP error %d (%s) at %s:%d: %s", (int)e__,             \
                  hipGetErrorString(e__), __FILE__, __LINE__, #expr);         \
  } while (0)

#define GTX_CHECK(cond, ...)                                                  \
  do {                                                                        \
    if (!(cond)) ::gtx::fail(-3, __VA_ARGS__);                                \
  } while (0)

// Device allocation owned by a context. 256-B aligned by hipMalloc.
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  explicit DevBuf(size_t n) { alloc(n); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void alloc(size_t n) {
    release();
    if (n == 0) n = 256;
    GTX_HIP(hipMalloc(&p, n));
    bytes = n;
  }
  void release() {
    if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
  }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// fn(i) for every i in [0, n) on up to 8 host threads. For set-up work whose iterations write disjoint ranges and do not throw
// (packing a model's weights: 175 ms on one thread, the largest part of building the first detector of a process).
template <class F>
inline void parallel_for(int n, F fn) {
  const unsigned hw = std::thread::hardware_concurrency();
  const int nt = std::min(std::min<int>(hw ? (int)hw : 1, 8), n);
  if (nt <= 1) {
    for (int i = 0; i < n; ++i) fn(i);
    return;
  }
  std::atomic<int> next{0};
  auto work = [&] { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; ++t) th.emplace_back(work);
  work();
  for (auto& t : th) t.join();
}

}  // namespace gtx

#if defined(__HIPCC__)
// Wave-aggregated list append: the lanes of a wave that want a slot (pred) reserve them with ONE atomic on
// the counter instead of one each -- appends from thousands of waves to a single counter otherwise serialise
// in L2 (measured: fast_detect 75 -> 33 us). Returns the lane's slot, or -1 for lanes with pred == false.
// Every lane of the wave must call it (the ballot needs the whole wave), converged.
__device__ __forceinline__ int gtx_wave_append(int* counter, bool pred) {
  const unsigned long long m = __ballot(pred);
  if (m == 0) return -1;
  const int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, __popcll(m));
  base = __shfl(base, leader);
  return pred ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}
#endif

