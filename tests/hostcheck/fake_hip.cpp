// Test double of the HIP runtime (see fakehip/hip/hip_runtime.h): heap "device" memory, streams = in-order worker threads.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

struct FakeStream {
  std::mutex mu;
  std::condition_variable cv, idle;
  std::deque<std::function<void()>> q;
  bool stop = false, running = false;
  std::thread worker;
  FakeStream() : worker([this] { loop(); }) {}
  ~FakeStream() {
    { std::lock_guard<std::mutex> l(mu); stop = true; }
    cv.notify_all();
    worker.join();
  }
  void loop() {
    std::unique_lock<std::mutex> l(mu);
    for (;;) {
      cv.wait(l, [this] { return stop || !q.empty(); });
      if (q.empty()) return;
      auto f = std::move(q.front());
      q.pop_front();
      running = true;
      l.unlock();
      f();
      l.lock();
      running = false;
      if (q.empty()) idle.notify_all();
    }
  }
  void push(std::function<void()> f) {
    { std::lock_guard<std::mutex> l(mu); q.push_back(std::move(f)); }
    cv.notify_all();
  }
  void sync() {
    std::unique_lock<std::mutex> l(mu);
    idle.wait(l, [this] { return q.empty() && !running; });
  }
};

struct FakeEvent {
  std::mutex mu;
  std::condition_variable cv;
  unsigned long recorded = 0, done = 0;       // generation counters: a wait covers the record that preceded it
};

namespace {
std::mutex g_mu;
std::map<const void *, size_t> g_pinned;      // hipHostMalloc / hipHostRegister ranges
FakeStream *null_stream() { static FakeStream s; return &s; }
FakeStream *S(hipStream_t s) { return s ? s : null_stream(); }
}  // namespace

const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "fake HIP error"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = getenv("FAKE_HIP_NO_DEVICE") ? 0 : 4; return *n ? hipSuccess : hipErrorNoDevice; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 8; strcpy(p->name, "fake"); return hipSuccess; }
hipError_t hipMalloc(void **p, size_t bytes) { *p = malloc(bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) {                 // like the real one: waits for the device
  null_stream()->sync();
  free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) {
  *p = malloc(bytes ? bytes : 1);
  if (!*p) return hipErrorOutOfMemory;
  std::lock_guard<std::mutex> l(g_mu);
  g_pinned[*p] = bytes;
  return hipSuccess;
}
hipError_t hipHostFree(void *p) {
  { std::lock_guard<std::mutex> l(g_mu); g_pinned.erase(p); }
  free(p);
  return hipSuccess;
}
hipError_t hipHostRegister(void *p, size_t bytes, unsigned) { std::lock_guard<std::mutex> l(g_mu); g_pinned[p] = bytes; return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *at, const void *p) {
  std::lock_guard<std::mutex> l(g_mu);
  auto it = g_pinned.upper_bound(p);
  if (it != g_pinned.begin()) {
    --it;
    if ((const char *)p < (const char *)it->first + it->second) { at->type = hipMemoryTypeHost; return hipSuccess; }
  }
  at->type = hipMemoryTypeUnregistered;
  return hipErrorInvalidValue;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t s) {
  S(s)->push([=] { memcpy(dst, src, bytes); });
  return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t s) {
  S(s)->push([=] { memset(dst, value, bytes); });
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new FakeStream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { S(s)->sync(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new FakeEvent(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  unsigned long gen;
  { std::lock_guard<std::mutex> l(e->mu); gen = ++e->recorded; }
  S(s)->push([=] { { std::lock_guard<std::mutex> l(e->mu); if (e->done < gen) e->done = gen; } e->cv.notify_all(); });
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  std::unique_lock<std::mutex> l(e->mu);
  const unsigned long gen = e->recorded;
  e->cv.wait(l, [=] { return e->done >= gen; });
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  unsigned long gen;
  { std::lock_guard<std::mutex> l(e->mu); gen = e->recorded; }
  S(s)->push([=] { std::unique_lock<std::mutex> l(e->mu); e->cv.wait(l, [=] { return e->done >= gen; }); });
  return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
// like the real query: LDS-bound residency rounded in pieces FINER than the 1280 bytes a CU hands out (it over-reports; bench_micro/wg_residency)
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *, int, size_t lds) {
  *n = lds ? (int)((size_t)160 * 1024 / ((lds + 511) / 512 * 512)) : 2;
  if (*n > 32) *n = 32;
  return hipSuccess;
}
void fake_enqueue(hipStream_t s, std::function<void()> work) { S(s)->push(std::move(work)); }
