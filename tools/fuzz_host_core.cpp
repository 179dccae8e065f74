// fuzz_host_core.cpp -- CPU sanitizer harness for the host-side logic of libsgk.so (safe-grid-agents_amd/csrc/sgk_host_core.h): the SAME
// source the library is built from, compiled here with g++ against the test-only HIP stand-in (tools/hip_standin) and run under
// ThreadSanitizer / AddressSanitizer + UBSan by tools/sanitize_cpu.sh. TEST INFRASTRUCTURE: nothing in the product refers to it.
//
//   server   the step server's mailbox protocol (stop_server / server_round_trip) against a host-thread model of env_server_kernel's
//            loop (sgk_step.hip:156-245): random polling delays, idle-outs, exit words that land BEFORE or AFTER the stream reads idle
//            (within the host's wait, or beyond it), stale exit words, stops, handle destruction with the mailbox freed unless a word is
//            still owed. Checked after every call: each request taken exactly once, no error, and -- by the sanitizers -- no data
//            race on the mailbox words and no write into a freed mailbox.
//            --protocol prefix runs the protocol AS OF ded8f2b^ (round 4, before the step-taken-twice fix: kept below as the
//            fuzzer's known-bad control): it must FAIL within seconds.
//   graphs   the hipGraph LRU: random keys, eviction at the cap, nothing destroyed twice or leaked, an allocation failure at
//            construction only.
//   streams  the per-device stream pool from several threads: a stream is never handed to two owners.
//   rings    the trajectory-ring allocator over the VMM stand-in with injected driver faults and injected host-allocation failures:
//            whatever fails, no range / handle / mapping is leaked and no VMM call is misused.
//
//   g++ -std=c++17 -O1 -g -fsanitize=thread -I tools/hip_standin -I safe-grid-agents_amd/csrc tools/fuzz_host_core.cpp -o /tmp/fuzz -lpthread
//   /tmp/fuzz server --schedules 100000 [--seed S] [--protocol head|prefix]      /tmp/fuzz graphs|streams|rings --rounds N
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <random>
#include <string>

#include "sgk_host_core.h"

using namespace sgk::host;
using sgk::SgkMailbox;

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// =====================================================================================================================================
// server: the device side's model
// =====================================================================================================================================
struct ServerWorld {
  std::atomic<uint64_t> steps_taken{0};   // what the "envs" did: one per request served (exactly-once is checked against the host's count)
  std::atomic<uint32_t> last_served{0};
  std::atomic<uint64_t> launches{0};
  uint64_t seed = 1;
  bool reads_done_at_start = true;  // ded8f2b: a relaunched server takes what has been served from the mailbox
  uint32_t late_max = 0;            // exit words land up to this many spins after the stream reads idle (0 = always before)
  uint32_t idle_polls = 64;         // the server leaves after this many polls without a request (SGK_SERVER_IDLE_US)
};

// env_server_kernel's loop (sgk_step.hip), protocol part only: poll the request word; serve; publish `done`; leave on STOP or
// after idling; the exit word last. Device-side accesses are system-scope atomics there, __atomic here.
static standin::LateWrite server_model(ServerWorld *w, SgkMailbox *mb, uint32_t last, uint64_t launch_no) {
  std::mt19937_64 rng(w->seed * 0x9E3779B97F4A7C15ull + launch_no);
  if (w->reads_done_at_start) last = __atomic_load_n(const_cast<uint32_t *>(&mb->done), __ATOMIC_ACQUIRE);
  uint32_t idle = 0;
  const uint32_t idle_limit = 1 + (uint32_t)(rng() % w->idle_polls);
  for (;;) {
    standin::spin((uint32_t)(rng() % 24));  // one poll = one PCIe read, of varying length
    const uint64_t word = __atomic_load_n(const_cast<uint64_t *>(&mb->request), __ATOMIC_ACQUIRE);
    const uint32_t req = (uint32_t)word;
    if (req == last) {
      if (++idle > idle_limit) break;
      continue;
    }
    if (req == SGK_SERVER_STOP) break;
    idle = 0;
    standin::spin((uint32_t)(rng() % 40));  // the step itself
    w->steps_taken.fetch_add(1, std::memory_order_relaxed);
    w->last_served.store(req, std::memory_order_relaxed);
    __atomic_store_n(const_cast<uint32_t *>(&mb->done), req, __ATOMIC_RELEASE);
    last = req;
  }
  // "served up to `last`": the exit word, which may become visible after the stream already reads idle
  standin::LateWrite lw;
  const uint32_t word = last + 1u;
  lw.write = [mb, word] { __atomic_store_n(const_cast<uint32_t *>(&mb->exited), word, __ATOMIC_RELEASE); };
  lw.delay_spins = (w->late_max && (rng() & 3) == 0) ? 1 + (uint32_t)(rng() % w->late_max) : 0;
  return lw;
}

static hipError_t launch_model(void *ctx, SgkMailbox *mb, uint32_t served, hipStream_t stream) {
  ServerWorld *w = static_cast<ServerWorld *>(ctx);
  const uint64_t n = w->launches.fetch_add(1) + 1;
  stream->enqueue([w, mb, served, n] { return server_model(w, mb, served, n); });
  return hipSuccess;
}

// ---- the host protocol as of ded8f2b^ (before "a relaunched server takes what has been served from the mailbox; the host looks at the
// answer again after waiting for the stream" and before f930226 "stop_server sees the server's exit word"): the KNOWN-BAD control.
namespace prefix {
static int stop_server(ServerLink &L) {
  if (!L.running) return SGK_OK;
  mb_store(&L.mb->request, (uint64_t)SGK_SERVER_STOP);
  hipError_t e = hipStreamSynchronize(L.stream);
  L.running = false;
  mb_store(&L.mb->request, (uint64_t)L.seq);
  mb_store(&L.mb->exited, 0u);
  return e == hipSuccess ? SGK_OK : hip_fail(e, "stopping the step server");
}
static int server_round_trip(ServerLink &L, uint32_t flags8, uint32_t action0) {
  SgkMailbox *mb = L.mb;
  if (!L.running) {
    mb_store(&mb->request, (uint64_t)L.seq);
    mb_store(&mb->done, L.seq);
    mb_store(&mb->exited, 0u);
    SGK_HIP(L.launch(L.launch_ctx, mb, L.seq, L.stream));
    L.running = true;
  }
  const uint32_t prev = L.seq;
  uint32_t seq = prev + 1u;
  if (seq == SGK_SERVER_STOP) seq = 0u;
  mb_store(&mb->request, (uint64_t)seq | ((uint64_t)(flags8 & 0xffu) << 32) | ((uint64_t)(action0 & 3u) << 40));
  L.seq = seq;
  uint64_t spins = 0;  // (this control keeps no count of the words it is owed either)
  while (mb_load(&mb->done) != seq) {
    if (mb_load(&mb->exited) != 0) {
      hipError_t e = hipStreamSynchronize(L.stream);
      mb_store(&mb->exited, 0u);
      if (e == hipSuccess) e = L.launch(L.launch_ctx, mb, prev, L.stream);  // (no second look at the answer)
      if (e != hipSuccess) return hip_fail(e, "restarting the step server");
    }
    if (++spins > L.answer_spins) return fail(SGK_ERR_HIP, "the step server did not answer");
    cpu_pause();
  }
  return SGK_OK;
}
}  // namespace prefix

struct ServerStats {
  uint64_t schedules = 0, round_trips = 0, stops = 0, destroys = 0, stale_words = 0, restarts = 0, parked = 0, missing = 0;
};

// one schedule = one handle's life: a random sequence of steps (with gaps that let the server idle out), stops, stale exit words
// and a destruction at the end. Returns an empty string, or what went wrong.
static std::string run_schedule(uint64_t seed, bool head, ServerStats &st, std::vector<SgkMailbox *> &parked, hipStream_t stream) {
  std::mt19937_64 rng(seed);
  ServerWorld world;
  world.seed = seed;
  world.reads_done_at_start = head;
  const int lateness = (int)(rng() % 4);  // 0: words always on time; 1: a little late; 2: around the host's wait; 3: well beyond it
  world.late_max = lateness == 0 ? 0 : lateness == 1 ? 200 : lateness == 2 ? 3000 : 40000;
  world.idle_polls = 8 + (uint32_t)(rng() % 120);
  SgkMailbox *mb = new SgkMailbox();
  memset((void *)mb, 0, sizeof(*mb));
  ServerLink L;
  L.mb = mb;
  L.stream = stream;
  L.launch_ctx = &world;
  L.launch = launch_model;
  L.exit_word_spins = 4000;       // (the product: 2^22 pauses; here the late words are scaled down with it)
  L.stream_check_mask = 63;
  L.answer_spins = 1ull << 27;
  uint64_t asked = 0;
  std::string bad;
  const int n_ops = 8 + (int)(rng() % 40);
  for (int op = 0; op < n_ops && bad.empty(); ++op) {
    const uint32_t r = (uint32_t)(rng() % 100);
    if (r < 78) {
      // the caller's own work between two env.step calls: nothing, a little, or long enough for the server to idle out
      const uint32_t gap = (uint32_t)(rng() % 3);
      standin::spin(gap == 0 ? 0 : gap == 1 ? (uint32_t)(rng() % 200) : 2000 + (uint32_t)(rng() % 8000));
      const int rc = head ? server_round_trip(L, (uint32_t)(rng() & 3), (uint32_t)(rng() & 3)) : prefix::server_round_trip(L, 0, 0);
      ++st.round_trips;
      ++asked;
      if (rc != SGK_OK) bad = std::string("round trip failed: ") + error_buffer();
      else if (world.steps_taken.load() != asked)
        bad = "request " + std::to_string(asked) + ": the envs have taken " + std::to_string(world.steps_taken.load()) + " steps";
      else if (world.last_served.load() != L.seq) bad = "the server served request " + std::to_string(world.last_served.load()) + ", asked " + std::to_string(L.seq);
    } else if (r < 90) {
      const int rc = head ? stop_server(L) : prefix::stop_server(L);  // what every other entry point does first (SGK_CHECK_HANDLE)
      ++st.stops;
      if (rc != SGK_OK) bad = std::string("stop failed: ") + error_buffer();
    } else if (r < 94 && L.running) {
      mb_store(&mb->exited, L.seq + 1u);  // sgk_debug_server_stale_exit_word
      L.launched += 1;
      ++st.stale_words;
    } else {
      standin::spin((uint32_t)(rng() % 3000));
    }
  }
  // sgk_destroy: stop the server, wait for the stream, free the mailbox -- unless an exit word is still owed to it
  if (bad.empty()) {
    const int rc = head ? stop_server(L) : prefix::stop_server(L);
    if (rc != SGK_OK) bad = std::string("final stop failed: ") + error_buffer();
    (void)hipStreamSynchronize(stream);
    if (bad.empty() && world.steps_taken.load() != asked) bad = "at destruction: " + std::to_string(world.steps_taken.load()) + " steps taken, " + std::to_string(asked) + " asked";
  }
  st.restarts += world.launches.load();
  ++st.destroys;
  if (!bad.empty() || L.words_owed() || !head) {
    // (the control run and failed runs: never free under a writer -- this harness must die of the PROTOCOL's errors, not its own)
    if (L.words_owed()) ++st.missing;
    stream->synchronize();
    stream->drain_late();
    if (head && bad.empty()) {
      parked.push_back(mb);  // the product leaves such a mailbox alone (sgk_destroy): freed here at the very end
      ++st.parked;
    } else {
      delete mb;
    }
  } else {
    delete mb;  // HEAD's claim: every exit word has been seen, nothing can land in this memory any more (ASan holds it to that)
  }
  ++st.schedules;
  return bad;
}

static int fuzz_server(uint64_t schedules, uint64_t seed0, bool head, double max_seconds) {
  hipStream_t stream = nullptr;
  (void)hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
  ServerStats st;
  std::vector<SgkMailbox *> parked;
  const double t0 = now_s();
  std::string bad;
  uint64_t s = 0;
  for (; s < schedules && bad.empty(); ++s) {
    bad = run_schedule(seed0 + s, head, st, parked, stream);
    if (max_seconds > 0 && now_s() - t0 > max_seconds) { ++s; break; }
  }
  stream->synchronize();
  stream->drain_late();
  for (SgkMailbox *mb : parked) delete mb;
  (void)hipStreamDestroy(stream);
  printf("server protocol %s: %llu schedules, %llu round trips, %llu stops, %llu stale words, %llu server launches, %llu exit words never seen "
         "(mailbox parked), %.1f s\n", head ? "HEAD" : "as of ded8f2b^", (unsigned long long)st.schedules, (unsigned long long)st.round_trips,
         (unsigned long long)st.stops, (unsigned long long)st.stale_words, (unsigned long long)st.restarts, (unsigned long long)st.missing, now_s() - t0);
  if (!bad.empty()) {
    printf("FAILED at schedule seed %llu: %s\n", (unsigned long long)(seed0 + s - 1), bad.c_str());
    return 1;
  }
  printf("ok\n");
  return 0;
}

// =====================================================================================================================================
// graphs, streams, rings
// =====================================================================================================================================
static int fuzz_graphs(uint64_t rounds, uint64_t seed) {
  std::mt19937_64 rng(seed);
  hipStream_t stream = nullptr;
  (void)hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
  uint64_t made = 0, nomem = 0;
  for (uint64_t r = 0; r < rounds; ++r) {
    // an allocation failure can only happen where the cache is made (its slots are reserved there)
    if (rng() % 5 == 0) alloc_countdown().store(1);
    try {
      GraphCache cache;
      alloc_countdown().store(0);
      const int n = (int)(rng() % 60);
      for (int i = 0; i < n; ++i) {
        const std::pair<int32_t, uint32_t> key((int32_t)(rng() % 24), (uint32_t)(rng() % 2));
        alloc_countdown().store((int)(rng() % 3));  // armed or not: insert / find must not allocate
        if (!cache.find(key)) {
          cache.insert(key, standin::make_graph_exec((int)made++), stream);
          if (!cache.find(key)) { printf("FAILED: a graph just inserted is not found\n"); return 1; }
        }
        alloc_countdown().store(0);
        if (cache.size() > GraphCache::CAP) { printf("FAILED: %zu graphs cached (cap %zu)\n", cache.size(), GraphCache::CAP); return 1; }
      }
      cache.clear();
    } catch (const std::bad_alloc &) {
      alloc_countdown().store(0);
      ++nomem;
    }
    std::lock_guard<std::mutex> lock(standin::graphs().m);
    if (!standin::graphs().live.empty() || standin::graphs().double_destroys) {
      printf("FAILED: %zu graph execs leaked, %ld destroyed twice\n", standin::graphs().live.size(), standin::graphs().double_destroys);
      return 1;
    }
  }
  (void)hipStreamDestroy(stream);
  printf("graph cache: %llu rounds, %llu graphs made and destroyed exactly once, %llu allocation failures at construction\nok\n",
         (unsigned long long)rounds, (unsigned long long)made, (unsigned long long)nomem);
  return 0;
}

static int fuzz_streams(uint64_t rounds, uint64_t seed) {
  StreamPool pool;
  std::atomic<int> bad{0};
  std::mutex owned_m;
  std::set<hipStream_t> owned;
  auto work = [&](int tid) {
    std::mt19937_64 rng(seed + (uint64_t)tid);
    std::vector<std::pair<int, hipStream_t>> mine;
    for (uint64_t r = 0; r < rounds; ++r) {
      if (mine.empty() || rng() % 2) {
        const int dev = (int)(rng() % standin::N_DEVICES);
        hipStream_t s = nullptr;
        try {
          if (pool.take(dev, &s) != hipSuccess) { bad = 1; return; }
        } catch (const std::bad_alloc &) { continue; }
        std::lock_guard<std::mutex> lock(owned_m);
        if (!owned.insert(s).second) { bad = 2; return; }  // handed to two owners
        mine.emplace_back(dev, s);
      } else {
        const size_t k = rng() % mine.size();
        {
          std::lock_guard<std::mutex> lock(owned_m);
          owned.erase(mine[k].second);
        }
        if (rng() % 7 == 0) alloc_countdown().store(1);  // give_back must swallow an allocation failure (sgk_destroy cannot fail on it)
        pool.give_back(mine[k].first, mine[k].second);
        mine.erase(mine.begin() + (long)k);
      }
    }
    for (auto &p : mine) {
      {
        std::lock_guard<std::mutex> lock(owned_m);
        owned.erase(p.second);
      }
      pool.give_back(p.first, p.second);
    }
  };
  std::vector<std::thread> ts;
  for (int t = 0; t < 4; ++t) ts.emplace_back(work, t);
  for (auto &t : ts) t.join();
  alloc_countdown().store(0);
  size_t pooled = 0;
  for (int d = 0; d < standin::N_DEVICES; ++d) pooled += pool.pooled(d);
  // every stream ever created is either pooled or was dropped by a give_back that could not remember it: destroy the pooled ones
  for (int d = 0; d < standin::N_DEVICES; ++d) {
    hipStream_t s;
    while (pool.pooled(d) && pool.take(d, &s) == hipSuccess) (void)hipStreamDestroy(s);
  }
  if (bad) { printf("FAILED: stream pool misbehaved (%d)\n", bad.load()); return 1; }
  printf("stream pool: 4 threads x %llu operations, %zu streams pooled at the end, %ld never returned to the pool (allocation failures)\nok\n",
         (unsigned long long)rounds, pooled, standin::live_streams().load());
  // (streams a failed give_back dropped stay alive by design; the stand-in's are leaked here on purpose and LeakSanitizer is told so)
  return 0;
}

static int fuzz_rings(uint64_t rounds, uint64_t seed) {
  std::mt19937_64 rng(seed);
  standin::Vmm &v = standin::vmm();
  std::vector<void *> rings;
  uint64_t ok = 0, refused = 0, nomem = 0;
  for (uint64_t r = 0; r < rounds; ++r) {
    const uint32_t what = (uint32_t)(rng() % 10);
    if (what < 6 || rings.empty()) {
      v.fail_one_in = (rng() % 3 == 0) ? 2 + (uint32_t)(rng() % 6) : 0;
      const int k = (rng() % 4 == 0) ? 1 + (int)(rng() % 4) : 0;
      alloc_countdown().store(k);
      const size_t bytes = ((size_t)1 + (size_t)(rng() % 2000)) << 20;  // 1 MiB .. 2 GiB: up to eight 256 MiB chunks
      void *p = nullptr;
      int rc;
      try {
        rc = ring_alloc((int)(rng() % standin::N_DEVICES), bytes, &p);
      } catch (const std::bad_alloc &) {  // (the C-ABI's barrier in the product)
        rc = SGK_ERR_NOMEM;
        ++nomem;
      }
      alloc_countdown().store(0);
      v.fail_one_in = 0;
      if (rc == SGK_OK) {
        if (!p) { printf("FAILED: SGK_OK without a pointer\n"); return 1; }
        rings.push_back(p);
        ++ok;
      } else {
        if (p) { printf("FAILED: a pointer handed out with status %d\n", rc); return 1; }
        ++refused;
      }
    } else {
      const size_t k = rng() % rings.size();
      if (what == 9 && ring_free((char *)rings[k] + 4096) == SGK_OK) { printf("FAILED: freed an interior pointer\n"); return 1; }
      if (ring_free(rings[k]) != SGK_OK) { printf("FAILED: ring_free: %s\n", error_buffer()); return 1; }
      if (ring_free(rings[k]) == SGK_OK) { printf("FAILED: freed twice\n"); return 1; }
      rings.erase(rings.begin() + (long)k);
    }
    // bookkeeping invariant after every operation: exactly the live rings' ranges, handles and mappings exist
    std::lock_guard<std::mutex> lock(v.m);
    if (v.misuse) { printf("FAILED: %ld VMM calls the driver would have refused\n", v.misuse); return 1; }
    if (v.ranges.size() != rings.size() || ring_registry().count() != rings.size()) {
      printf("FAILED: %zu address ranges / %zu registered for %zu live rings\n", v.ranges.size(), ring_registry().count(), rings.size());
      return 1;
    }
    if (v.handles.size() != v.maps.size()) { printf("FAILED: %zu handles, %zu mappings\n", v.handles.size(), v.maps.size()); return 1; }
  }
  for (void *p : rings) (void)ring_free(p);
  std::lock_guard<std::mutex> lock(v.m);
  if (!v.ranges.empty() || !v.handles.empty() || !v.maps.empty()) { printf("FAILED: leaked %zu ranges, %zu handles, %zu mappings\n", v.ranges.size(), v.handles.size(), v.maps.size()); return 1; }
  printf("ring allocator: %llu operations, %llu rings mapped, %llu refused (injected driver faults), %llu host-allocation failures; nothing leaked, no VMM misuse\nok\n",
         (unsigned long long)rounds, (unsigned long long)ok, (unsigned long long)refused, (unsigned long long)nomem);
  return 0;
}

// Graph captures against the synchronous calls that invalidate them (ROCm's rule, modelled in the stand-in; EXPERIMENTS R5.12).
// `protected_`: the product's capture_graph (serialised on capture_mutex, retried) beside the product's sgk_ring_free, whose device
// synchronisation takes the same mutex, and a "foreign" thread making synchronous legacy-stream calls of its own at random times (the
// caller's code, PyTorch). Checked: no synchronous call of the PRODUCT ever meets a capture; every graph that capture_graph hands
// back holds all its nodes (none instantiated from a truncated capture); a capture gives up -- with an error, not a graph -- only
// after five disturbed attempts; nothing leaks. !protected_ is the control: the capture as the library made it until round 5 (no
// mutex, no retry) beside the same ring thread: it must fail.
static int fuzz_captures(uint64_t rounds, uint64_t seed, bool protected_) {
  std::atomic<bool> stop{false};
  std::atomic<uint64_t> ok{0}, gave_up{0}, truncated{0}, old_style_failures{0}, rings{0}, foreign_calls{0};
  std::atomic<int> failed{0};
  std::atomic<uint64_t> attempts_hist[8] = {};
  std::atomic<int> storm{0};  // capturers in a storm round: the foreign thread calls without pause, so every attempt is disturbed
  auto capturer = [&](uint64_t s) {
    std::mt19937_64 rng(s);
    hipStream_t stream = nullptr;
    (void)hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    for (uint64_t r = 0; r < rounds && !failed.load(); ++r) {
      const bool in_storm = protected_ && rng() % 60 == 0;
      const int nodes = in_storm ? 40 : 1 + (int)(rng() % 40);
      const uint32_t pause = in_storm ? 20000 : (uint32_t)(rng() % 200);
      if (in_storm) storm++;
      int attempts = 0;
      auto record = [&](hipStream_t cap) {
        ++attempts;
        hipError_t e = hipSuccess;
        for (int k = 0; k < nodes && e == hipSuccess; ++k) {
          e = standin::capture_node(cap);
          standin::spin(pause);
        }
        return e;
      };
      hipGraphExec_t exec = nullptr;
      if (protected_) {
        const int rc = capture_graph(stream, "capture (harness)", record, &exec);
        attempts_hist[attempts < 7 ? attempts : 7]++;
        if (in_storm) storm--;
        if (rc == SGK_OK) {
          if (!exec || exec->id != nodes) { truncated++; failed.store(1); }
          ok++;
        } else {
          if (exec) failed.store(1);
          if (!strstr(error_buffer(), "capture")) failed.store(1);  // the message names what failed
          gave_up++;
        }
      } else {  // the library's capture until round 5
        hipGraph_t graph = nullptr;
        (void)hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal);
        const hipError_t le = record(stream);
        const hipError_t ce = hipStreamEndCapture(stream, &graph);
        if (le != hipSuccess || ce != hipSuccess) old_style_failures++;
        if (graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) failed.store(1);
        if (graph) (void)hipGraphDestroy(graph);
      }
      if (exec) (void)hipGraphExecDestroy(exec);
    }
    (void)hipStreamDestroy(stream);
  };
  auto ring_thread = [&] {  // the product's own device-wide synchronous call: sgk_ring_free
    while (!stop.load()) {
      void *p = nullptr;
      if (ring_alloc(0, (size_t)3 << 20, &p) == SGK_OK) {
        if (ring_free(p) != SGK_OK) failed.store(1);
        rings++;
      }
      standin::spin(500);
    }
  };
  auto foreign_thread = [&](uint64_t s) {  // somebody else's hipMemcpy
    standin::foreign_caller() = true;
    std::mt19937_64 rng(s);
    while (!stop.load()) {
      (void)hipMemcpy(nullptr, nullptr, 0, hipMemcpyHostToDevice);
      foreign_calls++;
      if (storm.load() > 0) {  // every attempt of the storm round's capture is to be disturbed: the give-up path
        std::this_thread::yield();
        continue;
      }
      std::this_thread::sleep_for(std::chrono::microseconds(20 + rng() % 400));
    }
  };
  std::vector<std::thread> aux;
  aux.emplace_back(ring_thread);
  if (protected_) aux.emplace_back(foreign_thread, seed * 77 + 1);
  std::vector<std::thread> caps;
  for (int i = 0; i < 3; ++i) caps.emplace_back(capturer, seed * 1000 + (uint64_t)i);
  for (std::thread &t : caps) t.join();
  stop.store(true);
  for (std::thread &t : aux) t.join();
  standin::CaptureBook &b = standin::captures();
  std::lock_guard<std::mutex> lock(b.m);
  printf("captures: %llu recorded in full, %llu given up after five disturbed attempts, %llu rings freed meanwhile, %llu foreign synchronous "
         "calls (%ld met a capture, %ld captures invalidated); synchronous calls of the product that met a capture: %ld\n",
         (unsigned long long)ok.load(), (unsigned long long)gave_up.load(), (unsigned long long)rings.load(),
         (unsigned long long)foreign_calls.load(), b.sync_calls_during_a_capture, b.invalidated, b.of_them_by_the_product);
  if (protected_) {
    printf("attempts per capture:");
    for (int i = 1; i < 8; ++i) printf(" %d: %llu", i, (unsigned long long)attempts_hist[i].load());
    printf("\n");
  }
  if (!protected_) {
    printf("control (no mutex, no retry): %llu captures died of the product's own sgk_ring_free\n", (unsigned long long)old_style_failures.load());
    if (old_style_failures.load() == 0 || b.of_them_by_the_product == 0) { printf("FAILED: the control did not fail\n"); return 1; }
    printf("FAILED (as it must): captures invalidated by the library's own synchronous call\n");
    return 1;
  }
  std::lock_guard<std::mutex> glock(standin::graphs().m);
  if (failed.load() || truncated.load() || b.of_them_by_the_product || b.misuse || !b.live_graphs.empty() || !b.capturing.empty() ||
      !standin::graphs().live.empty() || standin::graphs().double_destroys) {
    printf("FAILED: failed=%d truncated=%llu product syncs in a capture=%ld misuse=%ld graphs leaked=%zu execs leaked=%zu\n", failed.load(),
           (unsigned long long)truncated.load(), b.of_them_by_the_product, b.misuse, b.live_graphs.size(), standin::graphs().live.size());
    return 1;
  }
  if (b.invalidated == 0) { printf("FAILED: no capture was ever disturbed: the run tests nothing\n"); return 1; }
  printf("ok\n");
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s server|graphs|streams|rings|captures [--schedules N] [--rounds N] [--seed S] [--protocol head|prefix] [--seconds T]\n", argv[0]);
    return 2;
  }
  uint64_t schedules = 2000, rounds = 2000, seed = 1;
  bool head = true;
  double seconds = 0;
  for (int i = 2; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "--schedules")) schedules = strtoull(argv[i + 1], nullptr, 10);
    else if (!strcmp(argv[i], "--rounds")) rounds = strtoull(argv[i + 1], nullptr, 10);
    else if (!strcmp(argv[i], "--seed")) seed = strtoull(argv[i + 1], nullptr, 10);
    else if (!strcmp(argv[i], "--seconds")) seconds = atof(argv[i + 1]);
    else if (!strcmp(argv[i], "--protocol")) head = strcmp(argv[i + 1], "prefix") != 0;
  }
  if (!strcmp(argv[1], "server")) return fuzz_server(schedules, seed, head, seconds);
  if (!strcmp(argv[1], "graphs")) return fuzz_graphs(rounds, seed);
  if (!strcmp(argv[1], "streams")) return fuzz_streams(rounds, seed);
  if (!strcmp(argv[1], "rings")) return fuzz_rings(rounds, seed);
  if (!strcmp(argv[1], "captures")) return fuzz_captures(rounds, seed, head);
  return 2;
}
