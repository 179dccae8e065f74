// sgk_host_core.h -- the host-side logic of libsgk.so that does not depend on the kernels: the error buffer and the exception
// barrier of the C-ABI, the host-allocation gate, the hipGraph LRU, the per-device stream pool, the trajectory-ring allocator
// (HIP virtual-memory management) and the step server's mailbox protocol.
//
// It is written against <hip/hip_runtime.h> and nothing else of the product, so that the SAME source builds twice:
//   * into libsgk.so (sgk_api.hip includes it; hipcc, the real HIP runtime), and
//   * into the CPU sanitizer harness tools/fuzz_host_core.cpp (g++ -fsanitize=thread / address) against the test-only stand-in
//     tools/hip_standin/hip/hip_runtime.h, where a stream is a host thread, a "kernel" is a host function and the step server is
//     a thread that polls the mailbox with randomised delays and late-landing words (tools/sanitize_cpu.sh runs 10^5 schedules).
// GPU sanitizers do not exist on this pool; round 4's step-taken-twice bug (EXPERIMENTS R4.10) was found by luck on the GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

#include "../../include/sgk.h"
#include "sgk_mailbox.h"

namespace sgk {
namespace host {

// ------------------------------------------------------------------------------------------------
// Errors: a status code + a message in a fixed thread-local buffer. Nothing here allocates: fail() is what reports an
// allocation failure.
// ------------------------------------------------------------------------------------------------
constexpr size_t ERROR_BYTES = 640;
inline char *error_buffer() {
  static thread_local char buf[ERROR_BYTES] = "";
  return buf;
}
__attribute__((format(printf, 2, 3))) inline int fail(int code, const char *fmt, ...) {
  char tmp[ERROR_BYTES];  // (the arguments may point into the buffer itself: "keep the message, add to it")
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(tmp, sizeof(tmp), fmt, ap);
  va_end(ap);
  memcpy(error_buffer(), tmp, sizeof(tmp));
  return code;
}
inline int hip_fail(hipError_t e, const char *what) { return fail(SGK_ERR_HIP, "%s: %s", what, hipGetErrorString(e)); }
// keep / restore the message across a clean-up that may itself fail (and overwrite it)
struct KeepError {
  char saved[ERROR_BYTES];
  KeepError() { memcpy(saved, error_buffer(), ERROR_BYTES); }
  void restore() const { memcpy(error_buffer(), saved, ERROR_BYTES); }
};

#define SGK_HIP(call)                                               \
  do {                                                              \
    hipError_t e__ = (call);                                        \
    if (e__ != hipSuccess) return ::sgk::host::hip_fail(e__, #call); \
  } while (0)

// The exception barrier of the C-ABI (include/sgk.h promises status codes; SURVEY 8(b): no exceptions across the boundary).
// Every extern "C" entry point is a function-try-block:   int sgk_x(...) try { ... } SGK_CATCH_STATUS
// Nothing in a handler allocates.
#define SGK_CATCH_STATUS                                                                                        \
  catch (const std::bad_alloc &) { return ::sgk::host::fail(SGK_ERR_NOMEM, "out of host memory"); }           \
  catch (const std::exception &e__) { return ::sgk::host::fail(SGK_ERR_INTERNAL, "internal error: %s", e__.what()); } \
  catch (...) { return ::sgk::host::fail(SGK_ERR_INTERNAL, "internal error (unknown exception)"); }
// for the few entry points that do not return a status
#define SGK_CATCH_VALUE(v) \
  catch (...) { return (v); }

// ------------------------------------------------------------------------------------------------
// The host-allocation gate: every host allocation the library makes goes through it (HostAlloc for the containers, host_new for
// handles). sgk_debug_fail_host_alloc(k) arms it: the k-th allocation from then on throws std::bad_alloc -- how the tests reach
// the out-of-memory paths (tests/test_abi.py; the sanitizer harness sweeps k over every allocation of a call).
// ------------------------------------------------------------------------------------------------
inline std::atomic<int> &alloc_countdown() {
  static std::atomic<int> c{0};
  return c;
}
inline void alloc_gate() {
  std::atomic<int> &c = alloc_countdown();
  if (c.load(std::memory_order_relaxed) > 0 && c.fetch_sub(1, std::memory_order_relaxed) == 1) throw std::bad_alloc();
}
template <class T>
struct HostAlloc {
  using value_type = T;
  HostAlloc() = default;
  template <class U>
  HostAlloc(const HostAlloc<U> &) {}
  T *allocate(size_t n) {
    alloc_gate();
    return std::allocator<T>().allocate(n);
  }
  void deallocate(T *p, size_t n) { std::allocator<T>().deallocate(p, n); }
  template <class U>
  bool operator==(const HostAlloc<U> &) const { return true; }
  template <class U>
  bool operator!=(const HostAlloc<U> &) const { return false; }
};
template <class T>
using Vec = std::vector<T, HostAlloc<T>>;
template <class K, class V>
using Map = std::map<K, V, std::less<K>, HostAlloc<std::pair<const K, V>>>;
template <class T>
T *host_new() {
  alloc_gate();
  return new T();  // throws std::bad_alloc: the entry point's barrier turns it into SGK_ERR_NOMEM
}

// ------------------------------------------------------------------------------------------------
// Instantiated hipGraphs of a handle, keyed by (n_steps, flags): a small LRU. A caller that varies n_steps call by call would
// otherwise pile up instantiated graphs (each holds its kernel nodes' argument blocks) until the handle is destroyed. The
// least recently used one is destroyed when the cap is reached -- after its stream has drained, because a replay of it may
// still be in flight. The slots are reserved when the cache is made: insert() cannot fail, so an instantiated graph is never
// dropped on the floor.
// ------------------------------------------------------------------------------------------------
struct GraphCache {
  static constexpr size_t CAP = 16;
  struct Item {
    std::pair<int32_t, uint32_t> key;
    hipGraphExec_t exec;
    uint64_t used;
  };
  Vec<Item> items;
  uint64_t tick = 0;
  GraphCache() { items.reserve(CAP); }
  hipGraphExec_t find(const std::pair<int32_t, uint32_t> &key) {
    for (Item &it : items)
      if (it.key == key) {
        it.used = ++tick;
        return it.exec;
      }
    return nullptr;
  }
  void insert(const std::pair<int32_t, uint32_t> &key, hipGraphExec_t exec, hipStream_t replays_on) noexcept {
    if (items.size() >= CAP) {
      size_t lru = 0;
      for (size_t i = 1; i < items.size(); ++i)
        if (items[i].used < items[lru].used) lru = i;
      (void)hipStreamSynchronize(replays_on);
      (void)hipGraphExecDestroy(items[lru].exec);
      items.erase(items.begin() + (long)lru);
    }
    items.push_back(Item{key, exec, ++tick});  // within the reserved capacity
  }
  void clear() noexcept {
    for (Item &it : items) (void)hipGraphExecDestroy(it.exec);
    items.clear();
  }
  size_t size() const { return items.size(); }
};

// ------------------------------------------------------------------------------------------------
// A handle's own stream outlives the handle: sgk_get_stream() hands it to the caller (the Python wrapper wraps it as a
// torch.cuda.ExternalStream), and a caller-side object that remembers it -- torch's pinned-memory allocator records an event on
// every stream a block was used on when the block is freed -- must never find a destroyed stream there. Streams of destroyed handles
// wait in a per-device pool for the next sgk_create on that device; a process holds as many as it ever had handles alive at once.
// ------------------------------------------------------------------------------------------------
struct StreamPool {
  std::mutex mutex;
  Map<int, Vec<hipStream_t>> free_streams;
  hipError_t take(int device, hipStream_t *out) {
    {
      std::lock_guard<std::mutex> lock(mutex);
      auto it = free_streams.find(device);
      if (it != free_streams.end() && !it->second.empty()) {
        *out = it->second.back();
        it->second.pop_back();
        return hipSuccess;
      }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
  }
  // never fails towards the caller (sgk_destroy): a stream the pool has no room to remember stays alive and unused
  void give_back(int device, hipStream_t st) noexcept {
    try {
      std::lock_guard<std::mutex> lock(mutex);
      free_streams[device].push_back(st);
    } catch (...) {
    }
  }
  size_t pooled(int device) {
    std::lock_guard<std::mutex> lock(mutex);
    auto it = free_streams.find(device);
    return it == free_streams.end() ? 0 : it->second.size();
  }
};
inline StreamPool &stream_pool() {
  static StreamPool p;
  return p;
}

// Graph captures (hipStreamBeginCapture .. hipStreamEndCapture) and the library's device-wide synchronous calls exclude each other.
// On ROCm 7 a synchronous legacy-stream call (hipMemcpy, hipDeviceSynchronize) made by ANY thread while ANY stream of the process is
// capturing fails with hipErrorStreamCaptureImplicit and invalidates that capture -- thread-local capture mode and non-blocking
// streams notwithstanding (found by the GPU soak of round 5: one thread's sgk_create uploading its rule table while another
// thread's handle recorded its step graph, 2 failures in 17 runs of the suite). The library itself no longer makes such calls on a
// handle's path (uploads go through the handle's stream); sgk_ring_free's device synchronisation takes this mutex, and so does
// every capture. A capture invalidated by somebody else's synchronous call (the caller's own, PyTorch's) is retried.
inline std::mutex &capture_mutex() {
  static std::mutex m;
  return m;
}

// Wait for a handle's stream. A handle bound to the device's NULL stream (sgk_use_default_stream: where PyTorch queues its work
// unless told otherwise) waits with hipStreamSynchronize(NULL) -- one of the synchronous legacy-stream calls described above --,
// so that wait excludes the library's captures like the other device-wide calls do.
inline hipError_t wait_stream(hipStream_t st) {
  if (st) return hipStreamSynchronize(st);
  std::lock_guard<std::mutex> no_capture_meanwhile(capture_mutex());
  return hipStreamSynchronize(nullptr);
}

// `record(cap)` between hipStreamBeginCapture and hipStreamEndCapture on `cap`, instantiated into *exec. Captures are serialised
// against each other and against the library's device-wide synchronous calls (capture_mutex); one that a foreign synchronous
// call invalidated all the same (hipErrorStreamCaptureInvalidated) is recorded again, a few times.
template <class Record>
int capture_graph(hipStream_t cap, const char *what, Record record, hipGraphExec_t *exec) {
  for (int attempt = 0;; ++attempt) {
    hipGraph_t graph = nullptr;
    hipError_t be, le = hipSuccess, ce = hipSuccess;
    {
      std::lock_guard<std::mutex> one_capture_at_a_time(capture_mutex());
      be = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
      if (be == hipSuccess) {
        le = record(cap);
        ce = hipStreamEndCapture(cap, &graph);
      }
    }
    if (be == hipSuccess && le == hipSuccess && ce == hipSuccess) {
      hipError_t ie = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (ie != hipSuccess) return hip_fail(ie, "hipGraphInstantiate");
      return SGK_OK;
    }
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    const hipError_t first = be != hipSuccess ? be : (le != hipSuccess ? le : ce);
    const bool disturbed = be == hipSuccess && (le == hipErrorStreamCaptureInvalidated || ce == hipErrorStreamCaptureInvalidated ||
                                                le == hipErrorStreamCaptureImplicit || ce == hipErrorStreamCaptureImplicit);
    if (!disturbed || attempt >= 4) return hip_fail(first, be != hipSuccess ? "hipStreamBeginCapture" : what);
  }
}

// The entry points that take no handle run at arbitrary points of the caller's program (sgk_ring_free: from destructors;
// sgk_issue_peak: in the middle of a bench); whatever device they work on, the thread's current device is what it was when they
// return.
struct DeviceGuard {
  int before = -1;
  DeviceGuard() { (void)hipGetDevice(&before); }
  ~DeviceGuard() {
    if (before >= 0) (void)hipSetDevice(before);
    (void)hipGetLastError();
  }
};

// ------------------------------------------------------------------------------------------------
// Trajectory-ring memory: device memory for rings a persistent kernel streams into, through HIP's virtual-memory management:
// one contiguous virtual range backed by physical chunks of 256 MiB. Why: the rate at which the streamed rollout writes a
// multi-GB ring depends on how the ring's PHYSICAL memory is made up -- hipMalloc blocks of one process measure 4.6-4.9 us per
// step at 1 M BoatRace envs or 5.6-6.1, for the block's lifetime, and a ring mapped from chunks of 2 MiB / 32 MiB / 256-512 MiB /
// 1 GiB measures 5.25 / 5.05 / 4.52-4.78 / 5.34-5.53 (64 KiB: 23; profiles/r03/ring_alloc_vmm*.log): chunks of 256 MiB are on the
// fast level every time.
//
// Exception safety: everything that can throw (the block's record, its two vectors) is allocated BEFORE the first VMM call; the
// registry is an intrusive list, so registering a mapped ring cannot fail; the diagnostic goes into the fixed error buffer.
// ------------------------------------------------------------------------------------------------
struct RingBlock {
  int device = 0;
  void *va = nullptr;
  size_t va_bytes = 0;
  Vec<hipMemGenericAllocationHandle_t> chunks;
  Vec<size_t> chunk_bytes;
  RingBlock *next = nullptr;
};
struct RingRegistry {
  std::mutex mutex;
  RingBlock *head = nullptr;
  void add(RingBlock *b) noexcept {
    std::lock_guard<std::mutex> lock(mutex);
    b->next = head;
    head = b;
  }
  RingBlock *take(void *va) noexcept {
    std::lock_guard<std::mutex> lock(mutex);
    for (RingBlock **p = &head; *p; p = &(*p)->next)
      if ((*p)->va == va) {
        RingBlock *b = *p;
        *p = b->next;
        b->next = nullptr;
        return b;
      }
    return nullptr;
  }
  size_t count() {
    std::lock_guard<std::mutex> lock(mutex);
    size_t n = 0;
    for (RingBlock *b = head; b; b = b->next) ++n;
    return n;
  }
};
inline RingRegistry &ring_registry() {
  static RingRegistry r;
  return r;
}

inline void release_ring(void *va, RingBlock &b, size_t mapped_chunks, bool free_range = true) noexcept {
  size_t off = 0;
  for (size_t i = 0; i < b.chunks.size(); ++i) {
    if (i < mapped_chunks) (void)hipMemUnmap((char *)va + off, b.chunk_bytes[i]);
    (void)hipMemRelease(b.chunks[i]);
    off += b.chunk_bytes[i];
  }
  b.chunks.clear();
  if (va && free_range) (void)hipMemAddressFree(va, b.va_bytes);
  (void)hipGetLastError();
}

inline int ring_alloc(int32_t device, size_t bytes, void **dev_ptr) {
  if (!dev_ptr) return fail(SGK_ERR_INVALID, "dev_ptr is NULL");
  *dev_ptr = nullptr;
  if (bytes == 0) return fail(SGK_ERR_INVALID, "bytes == 0");
  DeviceGuard keep_current_device;
  SGK_HIP(hipSetDevice(device));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  // 256 MiB physical chunks, every size a multiple of what the driver maps in (its recommended granularity)
  size_t gran = 0;
  SGK_HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  if (gran < ((size_t)2 << 20) || (gran & (gran - 1))) gran = (size_t)2 << 20;  // never below 2 MiB (a power of two: a multiple of the driver's)
  const size_t chunk = (((size_t)256 << 20) + gran - 1) / gran * gran;
  std::unique_ptr<RingBlock> b(host_new<RingBlock>());
  b->device = device;
  // whole chunks, and a last one rounded up to the granularity (a ring smaller than a chunk is one allocation of its own size)
  const size_t n_chunks = (bytes + chunk - 1) / chunk;
  b->chunk_bytes.reserve(n_chunks);
  b->chunks.reserve(n_chunks);  // the last host allocation of this call
  for (size_t left = bytes; left > 0;) {
    const size_t take = left >= chunk ? chunk : (left + gran - 1) / gran * gran;
    b->chunk_bytes.push_back(take);
    b->va_bytes += take;
    left -= left >= chunk ? chunk : left;
  }
  // Reserve, create, map, open. On one box of this pool the driver refused to map into (or open) a range it had just handed out
  // -- "invalid argument", one call in a few hundred, the same sizes fine a moment later --: such a range is set aside (it stays
  // reserved until the call returns, so that the next reservation is a different one) and another is tried, three times in all.
  constexpr int ATTEMPTS = 3;
  void *va = nullptr;
  void *set_aside[ATTEMPTS] = {nullptr, nullptr, nullptr};
  int n_aside = 0;
  hipError_t err = hipSuccess;
  for (int attempt = 0; attempt < ATTEMPTS; ++attempt) {
    va = nullptr;
    b->chunks.clear();
    err = hipMemAddressReserve(&va, b->va_bytes, (chunk & (chunk - 1)) ? gran : chunk, nullptr, 0);  // aligned to the chunk size
    if (err != hipSuccess) {
      (void)hip_fail(err, "hipMemAddressReserve (trajectory ring)");
      break;
    }
    size_t mapped = 0, off = 0;
    const char *what = "";
    for (size_t i = 0; i < b->chunk_bytes.size() && err == hipSuccess; ++i) {
      hipMemGenericAllocationHandle_t h;
      what = "hipMemCreate";
      err = hipMemCreate(&h, b->chunk_bytes[i], &prop, 0);
      if (err != hipSuccess) break;
      b->chunks.push_back(h);  // within the reserved capacity
      what = "hipMemMap";
      err = hipMemMap((char *)va + off, b->chunk_bytes[i], 0, h, 0);
      if (err == hipSuccess) ++mapped;
      off += b->chunk_bytes[i];
    }
    if (err == hipSuccess) {
      hipMemAccessDesc acc = {};
      acc.location = prop.location;
      acc.flags = hipMemAccessFlagsProtReadWrite;
      what = "hipMemSetAccess";
      err = hipMemSetAccess(va, b->va_bytes, &acc, 1);
    }
    if (err == hipSuccess) break;
    (void)fail(SGK_ERR_HIP, "sgk_ring_alloc: %s: %s (%zu bytes as %zu chunk(s), granularity %zu, chunk %zu, va %p, attempt %d of %d)", what,
               hipGetErrorString(err), bytes, b->chunk_bytes.size(), gran, b->chunks.size(), va, attempt + 1, ATTEMPTS);
    release_ring(va, *b, mapped, false);  // the chunks go (unmapped where they were mapped); the range itself is kept aside
    set_aside[n_aside++] = va;
    va = nullptr;
    if (err == hipErrorOutOfMemory) break;  // not a range's fault
  }
  for (int i = 0; i < n_aside; ++i) (void)hipMemAddressFree(set_aside[i], b->va_bytes);
  (void)hipGetLastError();
  if (err != hipSuccess) return SGK_ERR_HIP;  // (the message is in the buffer)
  b->va = va;
  ring_registry().add(b.release());
  *dev_ptr = va;
  return SGK_OK;
}

inline int ring_free(void *dev_ptr) {
  if (!dev_ptr) return SGK_OK;
  std::unique_ptr<RingBlock> b(ring_registry().take(dev_ptr));
  if (!b) return fail(SGK_ERR_INVALID, "not a pointer sgk_ring_alloc returned");
  DeviceGuard keep_current_device;
  hipError_t e = hipSetDevice(b->device);
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> no_capture_meanwhile(capture_mutex());
    (void)hipDeviceSynchronize();  // nothing may still be writing into it
  }
  release_ring(dev_ptr, *b, b->chunks.size());  // (on a device that cannot be selected any more the driver calls fail by themselves)
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice (sgk_ring_free)");
  return SGK_OK;
}

// ------------------------------------------------------------------------------------------------
// The host side of the single-env step server's protocol over the mailbox of sgk_mailbox.h (pinned, device-mapped host memory).
// The device side is env_server_kernel (sgk_step.hip); the sanitizer harness runs a host-thread model of that kernel's loop
// against this very code.
// ------------------------------------------------------------------------------------------------
// The host's accesses to the mailbox words: release stores (everything written before -- the other envs' actions -- is visible to
// whoever sees the word) and acquire loads (the server's outputs, released before the number, are read after it). Plain moves on
// x86; data-race-free under the C++ model, which is what lets ThreadSanitizer check the rest.
template <class T>
inline void mb_store(volatile T *p, T v) { __atomic_store_n(const_cast<T *>(p), v, __ATOMIC_RELEASE); }
template <class T>
inline T mb_load(const volatile T *p) { return __atomic_load_n(const_cast<const T *>(p), __ATOMIC_ACQUIRE); }
inline void cpu_pause() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}

// One handle's link to its server: the mailbox, the stream the server kernel is launched on, what the host believes, and how to
// launch a server that has served up to `served` (the product: sgk::launch_env_server on the handle's shard).
struct ServerLink {
  SgkMailbox *mb = nullptr;
  hipStream_t stream = nullptr;
  bool running = false;
  uint32_t seq = 0;  // number of the last step requested
  void *launch_ctx = nullptr;
  hipError_t (*launch)(void *ctx, SgkMailbox *mb, uint32_t served, hipStream_t stream) = nullptr;
  // Every server that is launched writes exactly one exit word, and that word can become visible AFTER the server's stream reads
  // idle (seen on MI355X: EXPERIMENTS R4.10). The link keeps count: words still owed may land at any later time -- into a mailbox
  // that has been reused for the next server (handled in server_round_trip) or, if it were freed, into freed memory. Hence
  // words_owed(): sgk_destroy frees the mailbox only when every word has been seen. (Found by the sanitizer harness, round 5: a
  // server that ends while the host is consuming an EARLIER server's late word was never waited for.)
  uint64_t launched = 0, exit_words = 0;
  bool words_owed() const { return exit_words < launched; }
  // Words the bounded wait has already given up on once (two words that landed on top of each other count as one: the missing one
  // never comes). They stay owed -- the mailbox is never freed -- but are not waited for again: a link in that state would otherwise
  // pay the full bounded wait at every stop.
  uint64_t given_up = 0;
  // how long the waits below spin before they give up (the sanitizer harness shortens them)
  uint32_t exit_word_spins = 1u << 22;
  uint64_t stream_check_mask = (1ull << 20) - 1;  // the stream is asked every (mask + 1) spins (a power of two)
  uint64_t answer_spins = 1ull << 33;
};

inline hipError_t launch_server(ServerLink &L, uint32_t served) {
  const hipError_t e = L.launch(L.launch_ctx, L.mb, served, L.stream);
  if (e == hipSuccess) ++L.launched;
  return e;
}
// consume the exit word that is in the mailbox, if any (the exchange: a word landing between a look and a clear is not lost)
inline bool take_exit_word(ServerLink &L) {
  if (mb_load(&L.mb->exited) == 0) return false;
  if (__atomic_exchange_n(const_cast<uint32_t *>(&L.mb->exited), 0u, __ATOMIC_ACQ_REL) == 0) return false;
  ++L.exit_words;
  if (L.given_up > L.launched - L.exit_words) L.given_up = L.launched > L.exit_words ? L.launched - L.exit_words : 0;  // one came after all
  return true;
}
// The stream is idle: every server launched so far has ended and written its word. See them all, bounded (a stream in error has
// no server to wait for; two words that landed on top of each other count as one: the link then stays "owed" for good, which
// costs one mailbox that is never freed and ONE expired wait, never a write into freed memory).
inline void collect_exit_words(ServerLink &L) {
  for (uint32_t spin = 0; L.exit_words + L.given_up < L.launched && spin < L.exit_word_spins; ++spin)
    if (!take_exit_word(L)) cpu_pause();
  if (L.exit_words + L.given_up < L.launched) L.given_up = L.launched - L.exit_words;
}

// Ask the step server to leave and wait until it has: afterwards the handle's arrays in memory are current (state words, metric
// partials) and its stream is free for the next kernel.
inline int stop_server(ServerLink &L) {
  if (!L.running) return SGK_OK;
  mb_store(&L.mb->request, (uint64_t)SGK_SERVER_STOP);
  hipError_t e = hipStreamSynchronize(L.stream);
  if (e == hipSuccess) collect_exit_words(L);
  L.running = false;
  mb_store(&L.mb->request, (uint64_t)L.seq);
  if (e != hipSuccess) return hip_fail(e, "stopping the step server");
  return SGK_OK;
}

// One request to the handle's step server and its answer: `flags8` = the SGK_F_* flags of a step, or SGK_SRV_RESET; `action0`
// rides in the request word. Starts the server when none is running. On failure the request is taken back (the host counters
// have not moved) and the server is marked gone.
inline int server_round_trip(ServerLink &L, uint32_t flags8, uint32_t action0) {
  SgkMailbox *mb = L.mb;
  if (!L.running) {
    (void)take_exit_word(L);  // (a word that was still owed may have landed since)
    mb_store(&mb->request, (uint64_t)L.seq);
    mb_store(&mb->done, L.seq);
    SGK_HIP(launch_server(L, L.seq));
    L.running = true;
  }
  const uint32_t prev = L.seq;
  uint32_t seq = prev + 1u;
  if (seq == SGK_SERVER_STOP) seq = 0u;
  // (release: the other envs' actions, written by the caller, before the request word)
  mb_store(&mb->request, (uint64_t)seq | ((uint64_t)(flags8 & 0xffu) << 32) | ((uint64_t)(action0 & 3u) << 40));
  L.seq = seq;
  // Wait for the answer. The loop watches the mailbox, and every stream_check_mask + 1 spins (~a millisecond) it also asks the
  // STREAM: a server kernel that died, or never started, leaves the stream idle (or in error) with no answer -- the caller then
  // gets an error instead of spinning for minutes. On every failure the request is taken back (the host counters have not moved)
  // and the server is marked gone, so the next call starts from the arrays in memory.
  uint64_t spins = 0;
  int failed = 0, relaunches = 0;
  while (mb_load(&mb->done) != seq) {
    bool server_left = take_exit_word(L);
    hipError_t e = hipSuccess;
    if (server_left) {
      // A server left (idle) without having seen this request -- or the word is an EARLIER server's, landing late while a live
      // server is about to answer. Either way: wait for the stream (whatever runs there ends: it answers the request first if it
      // is alive), see the words of everything that has ended, and look at the answer again before starting anything. (A server
      // started needlessly would find the request already answered in the mailbox: env_server_kernel.)
      e = hipStreamSynchronize(L.stream);
      if (e != hipSuccess) {
        failed = hip_fail(e, "the step server's stream");
        break;
      }
    } else if ((++spins & L.stream_check_mask) == 0) {
      const hipError_t q = hipStreamQuery(L.stream);
      if (q != hipErrorNotReady && q != hipSuccess) {
        failed = hip_fail(q, "the step server's stream");
        break;
      }
      // An idle stream, no answer, no exit word -- as far as the words that have LANDED say: a server that idled out just before
      // the request, whose exit word (even its last answer) is still in flight although its stream already reads idle. Nothing
      // runs on the stream any more, so after the words' bounded wait another server may start whether they showed up or not.
      server_left = q == hipSuccess && mb_load(&mb->done) != seq;
      if (!server_left && spins > L.answer_spins) {
        failed = fail(SGK_ERR_HIP, "the step server did not answer");
        break;
      }
    }
    if (server_left) {
      collect_exit_words(L);
      if (mb_load(&mb->done) == seq) {
        L.running = false;  // (the stream is idle: the next request starts a server afresh)
        break;
      }
      // a server that vanishes again and again without an answer is a dead kernel, not a late word
      e = ++relaunches <= 8 ? launch_server(L, prev) : hipErrorUnknown;
      if (e != hipSuccess) {
        failed = relaunches > 8 ? fail(SGK_ERR_HIP, "the step server is gone (its stream is idle) without an answer, %d times in a row", relaunches - 1)
                                : hip_fail(e, "restarting the step server");
        break;
      }
      continue;
    }
    cpu_pause();
  }
  if (failed) {
    const KeepError keep;
    L.seq = prev;          // the request was not served: the host counters are untouched, so is the request number
    (void)stop_server(L);  // asks a server that may still be there to leave, waits for the stream, sees the exit words
    keep.restore();
    return failed;
  }
  return SGK_OK;  // (the acquire load of `done` orders the reads of the server's outputs after it)
}

}  // namespace host
}  // namespace sgk
