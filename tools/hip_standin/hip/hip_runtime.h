// hip/hip_runtime.h -- TEST-ONLY stand-in for the slice of the HIP runtime that safe-grid-agents_amd/csrc/sgk_host_core.h uses, so
// that the product's host-side logic (step-server protocol, hipGraph LRU, stream pool, trajectory-ring allocator, the C-ABI's error
// and allocation plumbing) can be compiled with g++ and run under ThreadSanitizer / AddressSanitizer on a CPU box
// (tools/fuzz_host_core.cpp, tools/sanitize_cpu.sh). GPU sanitizers do not exist on this pool. NOT part of the product: nothing
// under safe-grid-agents_amd/ or include/ refers to this directory; it is found only through the harness's -I flag.
//
//   stream        a host thread draining a FIFO of closures; a "kernel" is a closure; hipStreamSynchronize waits for an empty FIFO
//                 and an idle worker; a closure may hand back a LATE WRITE -- something that lands some time after the stream
//                 already reads idle, the way a server's exit word was seen to land on MI355X (EXPERIMENTS R4.10)
//   graph exec    a heap record; destroying one twice, or leaking one, is reported
//   capture       hipStreamBeginCapture .. EndCapture with ROCm 7's rule as MI355X showed it (EXPERIMENTS R5.12), not CUDA's: a
//                 synchronous legacy-stream call (hipDeviceSynchronize, hipMemcpy) from ANY thread while ANY stream is capturing fails
//                 with hipErrorStreamCaptureImplicit and invalidates every capture in progress, whatever the capture mode
//   VMM           address ranges, physical handles and mappings as bookkeeping only (no memory behind them), every misuse an
//                 error code as from the driver, optional fault injection (standin::vmm().fail_one_in)
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <random>
#include <set>
#include <thread>
#include <vector>

typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorNotReady = 600,
              hipErrorStreamCaptureInvalidated = 901, hipErrorStreamCaptureImplicit = 906, hipErrorUnknown = 999 };
inline const char *hipGetErrorString(hipError_t e) {
  switch (e) {
    case hipSuccess: return "no error";
    case hipErrorInvalidValue: return "invalid argument";
    case hipErrorOutOfMemory: return "out of memory";
    case hipErrorInvalidDevice: return "invalid device ordinal";
    case hipErrorNotReady: return "device not ready";
    case hipErrorStreamCaptureInvalidated: return "operation failed due to a previous error during capture";
    case hipErrorStreamCaptureImplicit: return "operation would make the legacy stream depend on a capturing blocking stream";
    default: return "unknown error";
  }
}
inline hipError_t hipGetLastError() { return hipSuccess; }

namespace standin {
constexpr int N_DEVICES = 2;
inline int &current_device() {
  static thread_local int d = 0;
  return d;
}
inline void spin(uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
}
}  // namespace standin
inline hipError_t hipGetDevice(int *d) { *d = standin::current_device(); return hipSuccess; }
inline hipError_t hipSetDevice(int d) {
  if (d < 0 || d >= standin::N_DEVICES) return hipErrorInvalidDevice;
  standin::current_device() = d;
  return hipSuccess;
}

// ---- streams ----------------------------------------------------------------------------------------------------------------------
namespace standin {
// what a "kernel" may leave behind: a write that lands `delay_spins` after the stream has gone idle (0 spins = before it does)
struct LateWrite {
  std::function<void()> write;
  uint32_t delay_spins = 0;
};
struct Stream {
  std::mutex m;
  std::condition_variable cv;
  std::deque<std::function<LateWrite()>> fifo;
  bool busy = false, quit = false;
  std::vector<std::thread> late;  // writers still in flight (joined by drain_late / the destructor)
  std::thread worker;
  Stream() : worker([this] { run(); }) {}
  ~Stream() {
    {
      std::lock_guard<std::mutex> lock(m);
      quit = true;
    }
    cv.notify_all();
    worker.join();
    drain_late();
  }
  void run() {
    for (;;) {
      std::function<LateWrite()> k;
      {
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [this] { return quit || !fifo.empty(); });
        if (fifo.empty()) return;
        k = std::move(fifo.front());
        fifo.pop_front();
        busy = true;
      }
      LateWrite lw = k();
      if (lw.write && lw.delay_spins == 0) {
        lw.write();
        lw.write = nullptr;
      }
      {
        std::lock_guard<std::mutex> lock(m);
        busy = false;
        if (lw.write) {
          const uint32_t d = lw.delay_spins;
          auto w = std::move(lw.write);
          late.emplace_back([w, d] {
            spin(d);
            w();
          });
        }
      }
      cv.notify_all();
    }
  }
  void enqueue(std::function<LateWrite()> k) {
    {
      std::lock_guard<std::mutex> lock(m);
      fifo.push_back(std::move(k));
    }
    cv.notify_all();
  }
  void synchronize() {
    std::unique_lock<std::mutex> lock(m);
    cv.wait(lock, [this] { return fifo.empty() && !busy; });
  }
  bool idle() {
    std::lock_guard<std::mutex> lock(m);
    return fifo.empty() && !busy;
  }
  // wait for every late write of this stream to have landed (the harness calls it before it frees what they write into)
  void drain_late() {
    std::vector<std::thread> mine;
    {
      std::lock_guard<std::mutex> lock(m);
      mine.swap(late);
    }
    for (std::thread &t : mine) t.join();
  }
};
inline std::atomic<long> &live_streams() {
  static std::atomic<long> n{0};
  return n;
}
}  // namespace standin
typedef standin::Stream *hipStream_t;
enum : unsigned { hipStreamNonBlocking = 1 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
  *s = new standin::Stream();
  standin::live_streams()++;
  return hipSuccess;
}
inline hipError_t hipStreamDestroy(hipStream_t s) {
  delete s;
  standin::live_streams()--;
  return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t s) {
  if (s) s->synchronize();
  return hipSuccess;
}
inline hipError_t hipStreamQuery(hipStream_t s) { return (!s || s->idle()) ? hipSuccess : hipErrorNotReady; }

// ---- stream capture -----------------------------------------------------------------------------------------------------------------
namespace standin {
struct Graph { int nodes; };
struct CaptureBook {
  std::mutex m;
  std::map<Stream *, std::pair<int, bool>> capturing;  // stream -> (nodes recorded so far, invalidated)
  std::set<Graph *> live_graphs;
  long sync_calls_during_a_capture = 0;  // legacy-stream synchronous calls that found a capture in progress ...
  long of_them_by_the_product = 0;       // ... and were not marked as somebody else's (foreign_caller): must stay 0
  long invalidated = 0, misuse = 0;
};
inline CaptureBook &captures() {
  static CaptureBook b;
  return b;
}
inline bool &foreign_caller() {  // the harness sets it in threads that play the caller's own code / PyTorch
  static thread_local bool f = false;
  return f;
}
// what every synchronous legacy-stream call does first
inline hipError_t legacy_sync_check() {
  CaptureBook &b = captures();
  std::lock_guard<std::mutex> lock(b.m);
  if (b.capturing.empty()) return hipSuccess;
  b.sync_calls_during_a_capture++;
  if (!foreign_caller()) b.of_them_by_the_product++;
  for (auto &c : b.capturing)
    if (!c.second.second) { c.second.second = true; b.invalidated++; }
  return hipErrorStreamCaptureImplicit;
}
// a kernel launch into a capturing stream (the harness's `record` callbacks call it once per node)
inline hipError_t capture_node(Stream *s) {
  CaptureBook &b = captures();
  std::lock_guard<std::mutex> lock(b.m);
  auto it = b.capturing.find(s);
  if (it == b.capturing.end()) { b.misuse++; return hipErrorInvalidValue; }
  if (it->second.second) return hipErrorStreamCaptureInvalidated;
  it->second.first++;
  return hipSuccess;
}
}  // namespace standin
typedef standin::Graph *hipGraph_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
inline hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) {
  standin::CaptureBook &b = standin::captures();
  std::lock_guard<std::mutex> lock(b.m);
  if (!s || b.capturing.count(s)) { b.misuse++; return hipErrorInvalidValue; }
  b.capturing[s] = {0, false};
  return hipSuccess;
}
inline hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t *graph) {
  standin::CaptureBook &b = standin::captures();
  std::lock_guard<std::mutex> lock(b.m);
  *graph = nullptr;
  auto it = b.capturing.find(s);
  if (it == b.capturing.end()) { b.misuse++; return hipErrorInvalidValue; }
  const std::pair<int, bool> c = it->second;
  b.capturing.erase(it);
  if (c.second) return hipErrorStreamCaptureInvalidated;
  *graph = new standin::Graph{c.first};
  b.live_graphs.insert(*graph);
  return hipSuccess;
}
inline hipError_t hipGraphDestroy(hipGraph_t g) {
  standin::CaptureBook &b = standin::captures();
  std::lock_guard<std::mutex> lock(b.m);
  if (!b.live_graphs.erase(g)) { b.misuse++; return hipErrorInvalidValue; }
  delete g;
  return hipSuccess;
}
inline hipError_t hipDeviceSynchronize() { return standin::legacy_sync_check(); }  // (the harness synchronises its streams itself)
enum hipMemcpyKind { hipMemcpyHostToDevice = 1 };
inline hipError_t hipMemcpy(void *, const void *, size_t, hipMemcpyKind) { return standin::legacy_sync_check(); }

// ---- graph execs ------------------------------------------------------------------------------------------------------------------
namespace standin {
struct GraphExec { int id; };
struct GraphBook {
  std::mutex m;
  std::set<GraphExec *> live;
  long double_destroys = 0;
};
inline GraphBook &graphs() {
  static GraphBook b;
  return b;
}
inline GraphExec *make_graph_exec(int id) {
  GraphExec *g = new GraphExec{id};
  std::lock_guard<std::mutex> lock(graphs().m);
  graphs().live.insert(g);
  return g;
}
}  // namespace standin
typedef standin::GraphExec *hipGraphExec_t;
inline hipError_t hipGraphInstantiate(hipGraphExec_t *exec, hipGraph_t g, void *, void *, size_t) {
  {
    std::lock_guard<std::mutex> lock(standin::captures().m);
    if (!standin::captures().live_graphs.count(g)) { standin::captures().misuse++; return hipErrorInvalidValue; }
  }
  *exec = standin::make_graph_exec(g->nodes);  // (id = the node count: the harness checks that no truncated graph was instantiated)
  return hipSuccess;
}
inline hipError_t hipGraphExecDestroy(hipGraphExec_t g) {
  {
    std::lock_guard<std::mutex> lock(standin::graphs().m);
    if (!standin::graphs().live.erase(g)) {
      standin::graphs().double_destroys++;
      return hipErrorInvalidValue;
    }
  }
  delete g;
  return hipSuccess;
}

// ---- virtual-memory management ------------------------------------------------------------------------------------------------------
struct hipMemLocation { int type; int id; };
struct hipMemAllocationProp { int type; hipMemLocation location; };
struct hipMemAccessDesc { hipMemLocation location; int flags; };
enum : int { hipMemAllocationTypePinned = 1, hipMemLocationTypeDevice = 1, hipMemAllocationGranularityRecommended = 1, hipMemAccessFlagsProtReadWrite = 3 };
namespace standin {
struct MemHandle { size_t bytes; int maps; };
struct Vmm {
  std::mutex m;
  uintptr_t next_va = (uintptr_t)1 << 40;
  std::map<uintptr_t, size_t> ranges;                         // reserved: base -> bytes
  std::map<uintptr_t, std::pair<size_t, MemHandle *>> maps;   // mapped: address -> (bytes, handle)
  std::set<MemHandle *> handles;
  long misuse = 0;            // calls the real driver would have refused AND the product should never make
  uint32_t fail_one_in = 0;   // fault injection: every call fails with probability 1 / fail_one_in (0 = never)
  std::mt19937 rng{12345};
  bool inject() { return fail_one_in && (rng() % fail_one_in) == 0; }
  bool inside_a_range(uintptr_t a, size_t n) {
    auto it = ranges.upper_bound(a);
    if (it == ranges.begin()) return false;
    --it;
    return a >= it->first && a + n <= it->first + it->second;
  }
};
inline Vmm &vmm() {
  static Vmm v;
  return v;
}
}  // namespace standin
typedef standin::MemHandle *hipMemGenericAllocationHandle_t;
inline hipError_t hipMemGetAllocationGranularity(size_t *g, const hipMemAllocationProp *, int) { *g = (size_t)2 << 20; return hipSuccess; }
inline hipError_t hipMemAddressReserve(void **va, size_t bytes, size_t align, void *, unsigned long long) {
  standin::Vmm &v = standin::vmm();
  std::lock_guard<std::mutex> lock(v.m);
  if (v.inject()) return hipErrorOutOfMemory;
  if (!align) align = (size_t)2 << 20;
  v.next_va = (v.next_va + align - 1) / align * align;
  *va = (void *)v.next_va;
  v.ranges[v.next_va] = bytes;
  v.next_va += bytes;
  return hipSuccess;
}
inline hipError_t hipMemAddressFree(void *va, size_t bytes) {
  standin::Vmm &v = standin::vmm();
  std::lock_guard<std::mutex> lock(v.m);
  auto it = v.ranges.find((uintptr_t)va);
  if (it == v.ranges.end() || it->second != bytes) { v.misuse++; return hipErrorInvalidValue; }
  auto m = v.maps.lower_bound((uintptr_t)va);
  if (m != v.maps.end() && m->first < (uintptr_t)va + bytes) { v.misuse++; return hipErrorInvalidValue; }  // still mapped inside
  v.ranges.erase(it);
  return hipSuccess;
}
inline hipError_t hipMemCreate(hipMemGenericAllocationHandle_t *h, size_t bytes, const hipMemAllocationProp *, unsigned long long) {
  standin::Vmm &v = standin::vmm();
  std::lock_guard<std::mutex> lock(v.m);
  if (v.inject()) return hipErrorOutOfMemory;
  *h = new standin::MemHandle{bytes, 0};
  v.handles.insert(*h);
  return hipSuccess;
}
inline hipError_t hipMemRelease(hipMemGenericAllocationHandle_t h) {
  standin::Vmm &v = standin::vmm();
  std::lock_guard<std::mutex> lock(v.m);
  if (!v.handles.count(h) || h->maps != 0) { v.misuse++; return hipErrorInvalidValue; }  // (the product unmaps before it releases)
  v.handles.erase(h);
  delete h;
  return hipSuccess;
}
inline hipError_t hipMemMap(void *p, size_t bytes, size_t, hipMemGenericAllocationHandle_t h, unsigned long long) {
  standin::Vmm &v = standin::vmm();
  std::lock_guard<std::mutex> lock(v.m);
  if (v.inject()) return hipErrorInvalidValue;  // what one box of the pool did now and then (sgk_host_core.h, ring_alloc)
  if (!v.handles.count(h) || h->bytes != bytes || !v.inside_a_range((uintptr_t)p, bytes) || v.maps.count((uintptr_t)p)) { v.misuse++; return hipErrorInvalidValue; }
  v.maps[(uintptr_t)p] = {bytes, h};
  h->maps++;
  return hipSuccess;
}
inline hipError_t hipMemUnmap(void *p, size_t bytes) {
  standin::Vmm &v = standin::vmm();
  std::lock_guard<std::mutex> lock(v.m);
  auto it = v.maps.find((uintptr_t)p);
  if (it == v.maps.end() || it->second.first != bytes) { v.misuse++; return hipErrorInvalidValue; }
  it->second.second->maps--;
  v.maps.erase(it);
  return hipSuccess;
}
inline hipError_t hipMemSetAccess(void *va, size_t bytes, const hipMemAccessDesc *, size_t) {
  standin::Vmm &v = standin::vmm();
  std::lock_guard<std::mutex> lock(v.m);
  if (v.inject()) return hipErrorInvalidValue;
  size_t covered = 0;
  for (auto it = v.maps.lower_bound((uintptr_t)va); it != v.maps.end() && it->first < (uintptr_t)va + bytes; ++it) covered += it->second.first;
  if (covered != bytes) { v.misuse++; return hipErrorInvalidValue; }
  return hipSuccess;
}
