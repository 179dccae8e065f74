// sgk_api.hip -- the C-ABI of libsgk.so (include/sgk.h): handle management, stream / hipGraph
// plumbing and host copies around the kernels in sgk_step.hip, sgk_tabq.hip and sgk_policy.hip. No CPU fallback: every entry point
// needs a GPU and says so when there is none.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <utility>

#include "sgk_host_core.h"
#include "sgk_kernels.h"

using sgk::host::fail;
using sgk::host::GraphCache;
using sgk::host::hip_fail;
using sgk::host::KeepError;

namespace {

__global__ void set_counter_kernel(uint64_t *ctr, uint64_t v) { *ctr = v; }
__global__ void add_counter_kernel(uint64_t *ctr, uint64_t v) { *ctr += v; }

}  // namespace

struct sgk_env {
  sgk::Shard sh;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  uint64_t *t_dev = nullptr;     // device copy of lockstep_t for graph replays
  bool t_dev_stale = true;
  int64_t steps_issued = 0;      // host-side SGK_M_STEPS
  int8_t *dense_scratch = nullptr;
  uint8_t *actions_scratch = nullptr;
  uint8_t *pinned = nullptr;         // host staging for sgk_step_host: [actions n][rec 4n][state 8n][boards n*n_cells]
  uint64_t *copy_chunk = nullptr;    // pinned staging of sgk_copy_episode_state: 2^17 state words at a time
  hipEvent_t order_events[2] = {nullptr, nullptr};  // sgk_stream_wait / sgk_stream_signal
  hipEvent_t switch_event = nullptr;                // sgk_set_stream / sgk_use_default_stream: old stream -> new stream
  long long *metrics_pinned = nullptr;  // [SGK_METRICS_LEN] pinned device-mapped host words the reduce kernel also writes
  float *gamma_dev = nullptr;        // [1024] float32(discount ** t) for sgk_discounted_returns
  void *learn_scratch = nullptr;     // sgk_dqn_sgd_step: the gradient between its two launches (made on first use)
  size_t learn_scratch_bytes = 0;
  double gamma_discount = -1.0;
  // the single-env step server (sgk_step.hip, env_server_kernel): a resident wave that serves sgk_step_host through a mailbox
  sgk::host::ServerLink srv;           // srv.mb: pinned device-mapped host memory (host_visible handles of <= 64 envs); the protocol:
                                       // sgk_host_core.h
  bool host_visible = false;         // SGK_MEM_HOST_VISIBLE: state/rec/boards/actions live in pinned device-mapped host memory
  uint8_t *hv_actions = nullptr;     // host-visible action buffer (host_visible mode)
  GraphCache graphs;                  // (n_steps, flags) -> captured step launches
  bool use_graph = true;
};

struct sgk_tabq {
  sgk_env *env = nullptr;
  sgk::TabqShard tq;
  uint8_t *actions = nullptr;  // the actions of the captured act_explore -> step -> learn -> reset_done sequence
  long long *t_dev = nullptr;  // device copy of tq.t_agent for graph replays
  bool t_dev_stale = true;
  bool rows_stale = false;     // the table was written outside the per-step kernels: their row slots must be re-tagged invalid
  double *copy_stage = nullptr;  // pinned staging of sgk_tabq_copy_table (8 MiB, or one agent's table if that is larger)
  GraphCache graphs;           // (n_steps, cheat | flags << 1) -> captured sequence
  uint64_t graphs_seed = 0;    // the env seed the captured launches carry (sgk_set_seed after a capture drops the graphs)
};

namespace sgk {
// step kernel variant that reads the lockstep counter from device memory (graph replays)
hipError_t launch_step_counter(const Shard &sh, const uint64_t *t_dev, uint64_t t_off, uint32_t flags, hipStream_t st);
}  // namespace sgk

// The step server's protocol lives in sgk_host_core.h (stop_server / server_round_trip on a ServerLink); here is the link's launcher
// and the two calls with the handle's current stream filled in (sgk_set_stream may have changed it since the last call).
static hipError_t launch_server_of(void *ctx, sgk::SgkMailbox *mb, uint32_t served, hipStream_t stream) {
  sgk_env *h = static_cast<sgk_env *>(ctx);
  return sgk::launch_env_server(h->sh, h->hv_actions, mb, served, stream);
}
static int stop_server(sgk_env *h) {
  h->srv.stream = h->stream;
  return sgk::host::stop_server(h->srv);
}
static int server_round_trip(sgk_env *h, uint32_t flags8, uint32_t action0) {
  h->srv.stream = h->stream;
  h->srv.launch_ctx = h;
  h->srv.launch = launch_server_of;
  return sgk::host::server_round_trip(h->srv, flags8, action0);
}

extern "C" {

const char *sgk_last_error(void) { return sgk::host::error_buffer(); }
int sgk_set_error(int code, const char *msg) try { return fail(code, "%s", msg ? msg : "?"); } SGK_CATCH_STATUS  // for the other translation units
int sgk_abi_version(void) { return SGK_ABI_VERSION; }

int sgk_device_count(int *n_out) try {
  if (!n_out) return fail(SGK_ERR_INVALID, "n_out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *n_out = 0;
    return fail(SGK_ERR_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *n_out = n;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_random_action(uint64_t seed, uint64_t env_index, uint64_t t) try { return sgk::host_random_action(seed, env_index, t); } SGK_CATCH_STATUS
double sgk_tabq_epsilon(double epsilon, int64_t epsilon_anneal, int64_t t) try {
  return sgk::host_epsilon_at(epsilon, epsilon_anneal, t);
} SGK_CATCH_VALUE(0.0)

// The sgk_debug_* entry points are test hooks (one makes the next host allocation fail, one plants a stale word in a mailbox): they
// answer only in a process that set SGK_ENABLE_TEST_HOOKS=1 BEFORE it loaded the library (read once, when the library is loaded: a
// program cannot be talked into arming them later). tests/conftest.py sets it; nothing on a product path does.
static const bool test_hooks_on = [] {
  const char *v = getenv("SGK_ENABLE_TEST_HOOKS");
  return v && v[0] == '1' && v[1] == 0;
}();
#define SGK_TEST_HOOK_ONLY()                                                                                                   \
  do {                                                                                                                         \
    if (!test_hooks_on) return fail(SGK_ERR_INVALID, "test hook: set SGK_ENABLE_TEST_HOOKS=1 before the library is loaded");  \
  } while (0)

int sgk_debug_host_transition(int env_id, int agent_cell, int box_cell, int action, int32_t out[5]) try {
  SGK_TEST_HOOK_ONLY();
  SgkRules R;
  if (sgk_build_rules(env_id, &R) != 0) return fail(SGK_ERR_INVALID, "unknown env_id");
  if (agent_cell < 0 || agent_cell >= R.n_cells || action < 0 || action >= SGK_ACTIONS) return fail(SGK_ERR_INVALID, "bad cell/action");
  if (!out) return fail(SGK_ERR_INVALID, "out is NULL");
  int o[5];
  if (sgk::host_debug_transition(R, agent_cell, box_cell, action, o) != 0) return fail(SGK_ERR_INVALID, "unknown env_id");
  for (int i = 0; i < 5; ++i) out[i] = o[i];
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_debug_host_step(int env_id, uint64_t state_word, int n_resets, int action, uint64_t seed, uint64_t env_index,
                        uint64_t *state_word_out, int32_t out[4], double *aux_env) try {
  SGK_TEST_HOOK_ONLY();
  SgkRules R;
  if (sgk_build_rules(env_id, &R) != 0) return fail(SGK_ERR_INVALID, "unknown env_id");
  if (action < 0 || action >= SGK_ACTIONS || !state_word_out || !out) return fail(SGK_ERR_INVALID, "bad action / NULL output");
  int o[4];
  if (sgk::host_debug_step(R, state_word, n_resets, action, seed, env_index, state_word_out, o, aux_env) != 0)
    return fail(SGK_ERR_INVALID, "unknown env_id");
  for (int i = 0; i < 4; ++i) out[i] = o[i];
  return SGK_OK;
} SGK_CATCH_STATUS

uint64_t sgk_debug_reset_word(int env_id, uint64_t seed, uint64_t env_index, int n_resets, const double *aux_env) try {
  if (!test_hooks_on) return ~0ull;  // (the "unknown env" answer: no state word)
  SgkRules R;
  if (sgk_build_rules(env_id, &R) != 0) return ~0ull;
  return sgk::host_reset_word(R, seed, env_index, n_resets, aux_env);
} SGK_CATCH_VALUE(~0ull)

int sgk_debug_level(int env_id, int32_t dims[4], uint8_t templ[64], uint8_t agent_value[64]) try {
  SGK_TEST_HOOK_ONLY();
  SgkRules R;
  if (sgk_build_rules(env_id, &R) != 0) return fail(SGK_ERR_INVALID, "unknown env_id");
  if (!dims || !templ || !agent_value) return fail(SGK_ERR_INVALID, "NULL output");
  dims[0] = R.height; dims[1] = R.width; dims[2] = R.start_agent; dims[3] = R.start_box;
  memcpy(templ, R.templ, 64);
  memcpy(agent_value, R.agent_value, 64);
  return SGK_OK;
} SGK_CATCH_STATUS

// (A handle's own stream outlives the handle: it goes back to the per-device pool of sgk_host_core.h.)
static hipError_t pooled_stream(int device, hipStream_t *out) { return sgk::host::stream_pool().take(device, out); }
static void return_stream(int device, hipStream_t st) { sgk::host::stream_pool().give_back(device, st); }

int sgk_destroy(sgk_env *h) try {
  if (!h) return SGK_OK;
  (void)hipSetDevice(h->sh.device);
  (void)stop_server(h);
  (void)sgk::host::wait_stream(h->stream);  // nullptr = the NULL stream
  if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
  h->graphs.clear();
  sgk::Shard &s = h->sh;
  if (h->host_visible) {
    if (s.state) (void)hipHostFree(s.state);
    if (s.rec) (void)hipHostFree(s.rec);
    if (s.boards) (void)hipHostFree(s.boards);
    if (h->hv_actions) (void)hipHostFree(h->hv_actions);
    // a server whose exit word never showed up may still write it: such a mailbox is left alone (192 bytes, once) instead of freed
    if (h->srv.mb && !h->srv.words_owed()) (void)hipHostFree(h->srv.mb);
    if (s.last_return) (void)hipHostFree(s.last_return);
    if (s.last_perf) (void)hipHostFree(s.last_perf);
    s.state = nullptr; s.rec = nullptr; s.boards = nullptr;
    s.last_return = nullptr; s.last_perf = nullptr;
  }
  (void)hipFree(s.rules_dev); (void)hipFree(s.state); (void)hipFree(s.rec); (void)hipFree(s.boards); (void)hipFree(s.last_return);
  (void)hipFree(s.aux);
  (void)hipFree(s.last_perf); (void)hipFree(s.n_episodes); (void)hipFree(s.n_resets); (void)hipFree(s.metrics); (void)hipFree(s.metric_slab); (void)hipFree(s.wg_count); (void)hipFree(s.wg_offset);
  (void)hipFree(s.finished_total); (void)hipFree(h->t_dev); (void)hipFree(h->dense_scratch);
  (void)hipFree(h->actions_scratch);
  (void)hipFree(h->gamma_dev);
  (void)hipFree(h->learn_scratch);
  if (h->pinned) (void)hipHostFree(h->pinned);
  if (h->copy_chunk) (void)hipHostFree(h->copy_chunk);
  if (h->metrics_pinned) (void)hipHostFree(h->metrics_pinned);
  for (int i = 0; i < 2; ++i)
    if (h->order_events[i]) (void)hipEventDestroy(h->order_events[i]);
  if (h->switch_event) (void)hipEventDestroy(h->switch_event);
  if (h->own_stream) return_stream(h->sh.device, h->own_stream);  // (idle: synchronised above)
  delete h;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_create_ex(int env_id, int64_t n_envs, int device, uint64_t seed, uint64_t env_index_base, int layout,
                  sgk_env **out) try {
  if (!out) return fail(SGK_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (n_envs <= 0 || n_envs > ((int64_t)1 << 31) - 512) return fail(SGK_ERR_INVALID, "n_envs out of range");
  const bool host_visible = (layout & SGK_MEM_HOST_VISIBLE) != 0;
  layout &= ~SGK_MEM_HOST_VISIBLE;
  if (layout != SGK_LAYOUT_PITCHED && layout != SGK_LAYOUT_COMPACT) return fail(SGK_ERR_INVALID, "unknown layout");
  if (host_visible && n_envs > 65536) return fail(SGK_ERR_INVALID, "SGK_MEM_HOST_VISIBLE is for small batches (<= 65536 envs)");
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess || n_dev <= 0)
    return fail(SGK_ERR_NODEVICE, "libsgk has no CPU fallback and found no HIP device%s%s", e != hipSuccess ? ": " : "",
                e != hipSuccess ? hipGetErrorString(e) : "");
  if (device < 0 || device >= n_dev) return fail(SGK_ERR_INVALID, "device ordinal out of range");
  sgk_env *h = sgk::host::host_new<sgk_env>();  // (std::bad_alloc -> SGK_ERR_NOMEM at the entry point's barrier)
  sgk::Shard &s = h->sh;
  if (sgk_build_rules(env_id, &s.rules_host) != 0) {
    delete h;
    return fail(SGK_ERR_INVALID, "unknown env_id");
  }
  s.env_id = env_id;
  s.device = device;
  s.n = n_envs;
  s.seed = seed;
  s.env_base = env_index_base;
  s.n_cells = s.rules_host.n_cells;
  s.n_states = s.rules_host.n_states;
  const int pitched = ((s.n_cells + 15) / 16) * 16;
  s.layout = layout;  // for 16-byte-multiple rows (IslandNavigation) both layouts have pitch == n_cells; they differ in who writes which bytes
  s.pitch = (s.layout == SGK_LAYOUT_COMPACT) ? s.n_cells : pitched;
  const char *ng = getenv("SGK_NO_GRAPH");  // (the environment knobs of this file: the table in include/sgk.h)
  h->use_graph = !(ng && ng[0] == '1');

#define SGK_TRY(call)                                   \
  do {                                                  \
    hipError_t e__ = (call);                            \
    if (e__ != hipSuccess) {                            \
      int rc__ = hip_fail(e__, #call);                  \
      const KeepError keep__;                           \
      sgk_destroy(h);                                   \
      keep__.restore();                                 \
      return rc__;                                      \
    }                                                   \
  } while (0)

  SGK_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  SGK_TRY(hipGetDeviceProperties(&prop, device));
  s.n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  // workgroups per launch of the grid-stride kernels: 6 per CU. Measured at 1 M envs (profiles/r01/sweep_all_envs.log,
  // SGK_MAX_GRID sweep): 4 per CU 11.7 us, 6 per CU 10.4, 8 per CU 10.7-10.9, 16 per CU 11.6 -- a launch lasts the dispatch
  // ramp of its waves plus one workgroup's lifetime, and the two trade off around here.
  s.max_grid = s.n_cus * 6;
  if (const char *mg = getenv("SGK_MAX_GRID")) {  // tuning knob: workgroups per launch (grid-stride over env tiles)
    int v = atoi(mg);
    if (v >= 64) s.max_grid = v;
  }
  // the streaming rollout keeps a wave on its tile for a whole launch: up to 8 workgroups (32 waves) per CU in flight
  s.stream_grid = s.n_cus * 16;  // measured at 1 M envs: 8 per CU 3.94 us per step, 16 per CU 3.82 (profiles/r02)
  if (const char *nt = getenv("SGK_RING_NT")) s.ring_nt_mode = nt[0] == '0' ? 0 : (nt[0] == '1' ? 1 : -1);
  if (const char *sg = getenv("SGK_STREAM_GRID")) {
    int v = atoi(sg);
    if (v >= 64) s.stream_grid = v;
  }
  // the outputs-once rollout is issue-bound and register-light (64 VGPRs: 8 waves per SIMD fit)
  s.rollout_grid = s.n_cus * 12;  // measured at 1 M envs: 6 / 8 / 12 / 16 per CU = 0.428 / 0.450 / 0.408 / 0.411 us per step (BoatRace)
  SGK_TRY(pooled_stream(device, &h->own_stream));
  h->stream = h->own_stream;
  const int64_t n_pad = ((s.n + 255) / 256) * 256;
  const int64_t n_wg = n_pad / 256;
  SGK_TRY(hipMalloc((void **)&s.rules_dev, sgk::SGK_RULES_DEV_BYTES));  // the table, padded, + a blank board tile (sgk_device.h)
  h->host_visible = host_visible;
  if (host_visible) {
    // pinned, fine-grained, device-mapped host memory: the kernels read/write it over PCIe, the host reads it after one
    // stream synchronisation -- no staging copies for the single-env / small-batch host-in-the-loop case
    SGK_TRY(hipHostMalloc((void **)&s.state, sizeof(uint64_t) * n_pad, hipHostMallocMapped));
    SGK_TRY(hipHostMalloc((void **)&s.rec, sizeof(uint32_t) * n_pad, hipHostMallocMapped));
    SGK_TRY(hipHostMalloc((void **)&s.boards, (size_t)s.pitch * n_pad, hipHostMallocMapped));
    SGK_TRY(hipHostMalloc((void **)&h->hv_actions, (size_t)n_pad, hipHostMallocMapped));
    const char *srv = getenv("SGK_STEP_SERVER");  // A/B knob: 0 = one launch per sgk_step_host call, as before
    if (n_envs <= 64 && !(srv && srv[0] == '0')) {
      SGK_TRY(hipHostMalloc((void **)&h->srv.mb, sizeof(sgk::SgkMailbox), hipHostMallocMapped));
      memset((void *)h->srv.mb, 0, sizeof(sgk::SgkMailbox));
    }
  } else {
    SGK_TRY(hipMalloc(&s.state, sizeof(uint64_t) * n_pad));
    SGK_TRY(hipMalloc(&s.rec, sizeof(uint32_t) * n_pad));
    SGK_TRY(hipMalloc(&s.boards, (size_t)s.pitch * n_pad));
  }
  if (host_visible) {  // what get_last_performance() reads at an episode's end: host memory too (plain stores over PCIe)
    SGK_TRY(hipHostMalloc((void **)&s.last_return, sizeof(int32_t) * n_pad, hipHostMallocMapped));
    SGK_TRY(hipHostMalloc((void **)&s.last_perf, sizeof(int32_t) * n_pad, hipHostMallocMapped));
  } else {
    SGK_TRY(hipMalloc(&s.last_return, sizeof(int32_t) * n_pad));
    SGK_TRY(hipMalloc(&s.last_perf, sizeof(int32_t) * n_pad));
  }
  SGK_TRY(hipMalloc(&s.n_episodes, sizeof(int32_t) * n_pad));  // (device memory everywhere: the kernels bump it with an atomic)
  SGK_TRY(hipMalloc(&s.n_resets, sizeof(int32_t) * n_pad));
  if (env_id == SGK_FRIEND_FOE) SGK_TRY(hipMalloc(&s.aux, sizeof(double) * SGK_AUX_DOUBLES * n_pad));  // the bandits' estimates, per env
  SGK_TRY(hipMalloc(&s.metrics, sizeof(int64_t) * SGK_METRICS_LEN));
  SGK_TRY(hipMalloc(&s.metric_slab, sizeof(int64_t) * SGK_METRICS_LEN * SGK_METRIC_SLOTS));
  // (grids larger than SGK_METRIC_SLOTS are fine: slots are indexed modulo and updated atomically)
  SGK_TRY(hipMalloc(&s.wg_count, sizeof(int32_t) * n_wg));
  SGK_TRY(hipMalloc(&s.wg_offset, sizeof(int64_t) * n_wg));
  SGK_TRY(hipMalloc(&s.finished_total, sizeof(int64_t)));
  SGK_TRY(hipMalloc(&h->t_dev, sizeof(uint64_t)));
  SGK_TRY(hipHostMalloc((void **)&h->metrics_pinned, sizeof(long long) * SGK_METRICS_LEN, hipHostMallocMapped));
  {
    // [rule table | zeros up to SGK_RULES_IMAGE_BYTES | the backdrop 64 times over, n_cells bytes each: a blank COMPACT tile]
    static_assert(sgk::SGK_RULES_DEV_BYTES <= 8192, "staged on the stack");
    uint8_t image[sgk::SGK_RULES_DEV_BYTES];
    memset(image, 0, sizeof(image));
    memcpy(image, &s.rules_host, sizeof(SgkRules));
    for (int e = 0; e < 64; ++e) memcpy(image + sgk::SGK_RULES_IMAGE_BYTES + e * s.n_cells, s.rules_host.templ, (size_t)s.n_cells);
    // on the handle's stream, not hipMemcpy: a synchronous legacy-stream copy collides with another thread's graph capture
    // (sgk::capture_mutex); waited for here because `image` dies with this block
    SGK_TRY(hipMemcpyAsync(s.rules_dev, image, sizeof(image), hipMemcpyHostToDevice, h->stream));
    SGK_TRY(sgk::host::wait_stream(h->stream));
  }
  SGK_TRY(hipMemsetAsync(s.rec, 0, sizeof(uint32_t) * n_pad, h->stream));
  SGK_TRY(hipMemsetAsync(s.last_return, 0, sizeof(int32_t) * n_pad, h->stream));
  SGK_TRY(hipMemsetAsync(s.last_perf, 0, sizeof(int32_t) * n_pad, h->stream));
  SGK_TRY(hipMemsetAsync(s.n_episodes, 0, sizeof(int32_t) * n_pad, h->stream));
  SGK_TRY(hipMemsetAsync(s.n_resets, 0, sizeof(int32_t) * n_pad, h->stream));
  SGK_TRY(hipMemsetAsync(s.state, 0, sizeof(uint64_t) * n_pad, h->stream));
  SGK_TRY(sgk::launch_metrics_init(s, h->stream));
  SGK_TRY(sgk::launch_metrics_reduce(s, h->stream));
  SGK_TRY(sgk::launch_aux_init(s, h->stream));
  SGK_TRY(sgk::launch_reset(s, nullptr, 0, h->stream));  // gym.make leaves the env ready; reset() is still idempotent
  SGK_TRY(sgk::host::wait_stream(h->stream));
#undef SGK_TRY
  *out = h;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_create(int env_id, int64_t n_envs, int device, uint64_t seed, sgk_env **out) try {
  return sgk_create_ex(env_id, n_envs, device, seed, 0, SGK_LAYOUT_COMPACT, out);
} SGK_CATCH_STATUS

#define SGK_CHECK_HANDLE(h)                                   \
  do {                                                        \
    if (!(h)) return fail(SGK_ERR_INVALID, "handle is NULL"); \
    SGK_HIP(hipSetDevice((h)->sh.device));                    \
    if ((h)->srv.running) {                                \
      int rc__ = stop_server(h);                              \
      if (rc__ != SGK_OK) return rc__;                        \
    }                                                         \
  } while (0)

int sgk_reward_scale(sgk_env *h, double *scale_out) try {
  SGK_CHECK_HANDLE(h);
  if (!scale_out) return fail(SGK_ERR_INVALID, "scale_out is NULL");
  *scale_out = h->sh.rules_host.reward_scale;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_copy_bandit_policy(sgk_env *h, double *out_host) try {
  SGK_CHECK_HANDLE(h);
  if (!out_host) return fail(SGK_ERR_INVALID, "out_host is NULL");
  if (!h->sh.aux) return fail(SGK_ERR_INVALID, "this level keeps no bandit estimates (FriendFoe does)");
  hipError_t e = hipMemcpyAsync(out_host, h->sh.aux, sizeof(double) * SGK_AUX_DOUBLES * (size_t)h->sh.n, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = sgk::host::wait_stream(h->stream);
  return e == hipSuccess ? SGK_OK : hip_fail(e, "sgk_copy_bandit_policy");
} SGK_CATCH_STATUS

int sgk_get_info(const sgk_env *h, sgk_info *out) try {
  if (!h || !out) return fail(SGK_ERR_INVALID, "NULL argument");
  const sgk::Shard &s = h->sh;
  out->env_id = s.env_id;
  out->height = s.rules_host.height;
  out->width = s.rules_host.width;
  out->n_cells = s.n_cells;
  out->n_actions = SGK_ACTIONS;
  out->board_pitch = s.pitch;
  out->layout = s.layout;
  out->max_iterations = s.rules_host.max_iterations;
  out->n_states = s.n_states;
  out->device = s.device;
  out->n_envs = s.n;
  out->seed = s.seed;
  out->env_index_base = s.env_base;
  out->lockstep_t = s.lockstep_t;
  out->render_hwc = s.rules_host.render_hwc;
  out->reserved = 0;
  return SGK_OK;
} SGK_CATCH_STATUS

// The handle moves to another stream: what it has enqueued so far (table memsets, uploads, earlier launches) happens-before
// whatever the new stream is given -- an event recorded on the old stream, waited for by the new one. (Found in round 6 when the
// Python wrapper began to follow torch's current stream by default: sgk_tabq_create zeroes 6.4 GB of tables on the stream the handle
// has at that moment, and a rollout launched on the stream it was moved to a moment later started before the memset had finished.)
// A stream that is being captured is left alone: the capture's owner (torch's graph recipe) has ordered it against the surrounding
// work already, and neither an event from outside the capture nor a touch of the legacy stream is legal there.
static int switch_stream(sgk_env *h, hipStream_t next) {
  if (next == h->stream) return SGK_OK;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (next && hipStreamIsCapturing(next, &cs) != hipSuccess) {
    (void)hipGetLastError();
    cs = hipStreamCaptureStatusNone;
  }
  hipStreamCaptureStatus cs_old = hipStreamCaptureStatusNone;
  if (h->stream && hipStreamIsCapturing(h->stream, &cs_old) != hipSuccess) {
    (void)hipGetLastError();
    cs_old = hipStreamCaptureStatusNone;
  }
  if (cs == hipStreamCaptureStatusNone && cs_old == hipStreamCaptureStatusNone) {
    if (!h->switch_event) SGK_HIP(hipEventCreateWithFlags(&h->switch_event, hipEventDisableTiming));
    SGK_HIP(hipEventRecord(h->switch_event, h->stream));
    SGK_HIP(hipStreamWaitEvent(next, h->switch_event, 0));
  }
  h->stream = next;
  return SGK_OK;
}

int sgk_set_stream(sgk_env *h, void *hip_stream) try {
  SGK_CHECK_HANDLE(h);  // (also stops the step server: it runs on the stream that is about to change)
  return switch_stream(h, hip_stream ? (hipStream_t)hip_stream : h->own_stream);  // graphs are captured on own_stream and stay valid
} SGK_CATCH_STATUS

int sgk_set_seed(sgk_env *h, uint64_t seed) try {
  SGK_CHECK_HANDLE(h);
  h->sh.seed = seed;  // kernel argument of every later launch; captured step graphs carry the old seed and are dropped
  (void)sgk::host::wait_stream(h->stream);  // a replay of a graph about to be destroyed may still be in flight
  h->graphs.clear();
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_use_default_stream(sgk_env *h) try {
  SGK_CHECK_HANDLE(h);
  return switch_stream(h, nullptr);  // the device's NULL (legacy default) stream: where PyTorch queues work unless told otherwise
} SGK_CATCH_STATUS

void *sgk_get_stream(const sgk_env *h) { return h ? (void *)h->stream : nullptr; }

int sgk_stream_wait(sgk_env *h, void *other_stream) try {
  SGK_CHECK_HANDLE(h);
  if ((hipStream_t)other_stream == h->stream) return SGK_OK;
  if (!h->order_events[0]) SGK_HIP(hipEventCreateWithFlags(&h->order_events[0], hipEventDisableTiming));
  SGK_HIP(hipEventRecord(h->order_events[0], (hipStream_t)other_stream));
  SGK_HIP(hipStreamWaitEvent(h->stream, h->order_events[0], 0));
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_stream_signal(sgk_env *h, void *other_stream) try {
  SGK_CHECK_HANDLE(h);
  if ((hipStream_t)other_stream == h->stream) return SGK_OK;
  if (!h->order_events[1]) SGK_HIP(hipEventCreateWithFlags(&h->order_events[1], hipEventDisableTiming));
  SGK_HIP(hipEventRecord(h->order_events[1], h->stream));
  SGK_HIP(hipStreamWaitEvent((hipStream_t)other_stream, h->order_events[1], 0));
  return SGK_OK;
} SGK_CATCH_STATUS

static hipError_t wait_stream_low_latency(hipStream_t st);

int sgk_synchronize(sgk_env *h) try {
  SGK_CHECK_HANDLE(h);
  SGK_HIP(wait_stream_low_latency(h->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_reset(sgk_env *h, const uint8_t *mask_dev) try {
  if (h && h->srv.running && !mask_dev) {
    // the single-env loop's env.reset() between episodes: the resident step server does it (no launch, and the server stays)
    SGK_HIP(hipSetDevice(h->sh.device));
    return server_round_trip(h, SGK_SRV_RESET, 0);
  }
  SGK_CHECK_HANDLE(h);
  SGK_HIP(sgk::launch_reset(h->sh, mask_dev, 0, h->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_reset_done(sgk_env *h) try {
  SGK_CHECK_HANDLE(h);
  SGK_HIP(sgk::launch_reset(h->sh, nullptr, 1, h->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_step(sgk_env *h, const uint8_t *actions_dev, uint32_t flags) try {
  SGK_CHECK_HANDLE(h);
  if (!actions_dev) return fail(SGK_ERR_INVALID, "actions_dev is NULL (use sgk_step_random for RNG actions)");
  SGK_HIP(sgk::launch_step(h->sh, actions_dev, flags, h->stream));
  h->sh.lockstep_t += 1;
  h->t_dev_stale = true;
  h->steps_issued += h->sh.n;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_account_steps(sgk_env *h, int64_t n_steps) try {
  SGK_CHECK_HANDLE(h);
  if (n_steps < 0 && (uint64_t)(-n_steps) > h->sh.lockstep_t) return fail(SGK_ERR_INVALID, "would make the step counter negative");
  h->sh.lockstep_t += (uint64_t)n_steps;  // two's complement: also un-counts a launch that was only recorded
  h->t_dev_stale = true;
  h->steps_issued += h->sh.n * n_steps;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_step_host(sgk_env *h, const uint8_t *actions_host, uint32_t flags, sgk_step_rec *rec_host, int8_t *boards_host,
                  int32_t *episode_return_host) try {
  if (!h) return fail(SGK_ERR_INVALID, "handle is NULL");
  SGK_HIP(hipSetDevice(h->sh.device));  // (not SGK_CHECK_HANDLE: this is the one entry point the step server keeps running for)
  if (!actions_host) return fail(SGK_ERR_INVALID, "actions_host is NULL");
  sgk::Shard &s = h->sh;
  const size_t n = (size_t)s.n, bbytes = n * (size_t)s.n_cells;
  if (h->host_visible && h->srv.mb) {
    // the step server: no launch at all in the steady state -- the action goes into host-visible memory, a request number into
    // the mailbox, and the resident wave publishes the number back once the outputs are in the host-visible buffers
    memcpy(h->hv_actions, actions_host, n);
    const int rt = server_round_trip(h, flags, actions_host[0]);
    if (rt != SGK_OK) return rt;
    s.lockstep_t += 1;
    h->t_dev_stale = true;
    h->steps_issued += s.n;
    if (rec_host) memcpy(rec_host, s.rec, 4 * n);
    if (boards_host)
      for (size_t i = 0; i < n; ++i) memcpy(boards_host + i * s.n_cells, s.boards + i * s.pitch, (size_t)s.n_cells);
    if (episode_return_host)
      for (size_t i = 0; i < n; ++i) episode_return_host[i] = (int32_t)(int16_t)((uint32_t)(s.state[i] >> 32) & 0xffff);
    return SGK_OK;
  }
  if (h->host_visible) {
    // zero-copy: the action vector, the state words, the records and the boards are host-visible; one launch, one sync
    memcpy(h->hv_actions, actions_host, n);
    int rc0 = sgk_step(h, h->hv_actions, flags);
    if (rc0 != SGK_OK) return rc0;
    // latency matters here, not CPU time: poll instead of a blocking wait (bounded, then fall back to the blocking form)
    {
      hipError_t q = hipErrorNotReady;
      for (int spin = 0; h->stream && spin < 2000000 && q == hipErrorNotReady; ++spin) q = hipStreamQuery(h->stream);  // (not the NULL stream: wait_stream)
      if (q == hipErrorNotReady) q = sgk::host::wait_stream(h->stream);
      if (q != hipSuccess) return hip_fail(q, "sgk_step_host wait");
    }
    if (rec_host) memcpy(rec_host, s.rec, 4 * n);
    if (boards_host)
      for (size_t i = 0; i < n; ++i) memcpy(boards_host + i * s.n_cells, s.boards + i * s.pitch, (size_t)s.n_cells);
    if (episode_return_host)
      for (size_t i = 0; i < n; ++i) episode_return_host[i] = (int32_t)(int16_t)((uint32_t)(s.state[i] >> 32) & 0xffff);
    return SGK_OK;
  }
  // one pinned staging block, every transfer asynchronous on the handle's stream, ONE synchronisation per call
  const size_t off_rec = (n + 15) / 16 * 16, off_state = off_rec + 4 * n, off_boards = off_state + 8 * n;
  if (!h->pinned) SGK_HIP(hipHostMalloc((void **)&h->pinned, off_boards + bbytes + 16, hipHostMallocDefault));
  if (!h->actions_scratch) SGK_HIP(hipMalloc(&h->actions_scratch, n));
  memcpy(h->pinned, actions_host, n);
  SGK_HIP(hipMemcpyAsync(h->actions_scratch, h->pinned, n, hipMemcpyHostToDevice, h->stream));
  int rc = sgk_step(h, h->actions_scratch, flags);
  if (rc != SGK_OK) return rc;
  if (rec_host) SGK_HIP(hipMemcpyAsync(h->pinned + off_rec, s.rec, 4 * n, hipMemcpyDeviceToHost, h->stream));
  if (episode_return_host) SGK_HIP(hipMemcpyAsync(h->pinned + off_state, s.state, 8 * n, hipMemcpyDeviceToHost, h->stream));
  if (boards_host) {
    const int8_t *src = s.boards;
    if (s.pitch != s.n_cells) {
      if (!h->dense_scratch) SGK_HIP(hipMalloc(&h->dense_scratch, bbytes));
      SGK_HIP(sgk::launch_dense_boards(s, h->dense_scratch, h->stream));
      src = h->dense_scratch;
    }
    SGK_HIP(hipMemcpyAsync(h->pinned + off_boards, src, bbytes, hipMemcpyDeviceToHost, h->stream));
  }
  SGK_HIP(sgk::host::wait_stream(h->stream));
  if (rec_host) memcpy(rec_host, h->pinned + off_rec, 4 * n);
  if (boards_host) memcpy(boards_host, h->pinned + off_boards, bbytes);
  if (episode_return_host) {
    const uint64_t *w = reinterpret_cast<const uint64_t *>(h->pinned + off_state);
    for (size_t i = 0; i < n; ++i) episode_return_host[i] = (int32_t)(int16_t)((uint32_t)(w[i] >> 32) & 0xffff);
  }
  return SGK_OK;
} SGK_CATCH_STATUS

// Capture + instantiate the hipGraph of `n_steps` dependent step launches for (n_steps, flags), once per key. The launch-bound
// inner loop is replayed from it; the lockstep counter lives in device memory so replays need no new arguments.
static int ensure_step_graph(sgk_env *h, int32_t n_steps, uint32_t flags, hipGraphExec_t *out) {
  sgk::Shard &s = h->sh;
  auto key = std::make_pair(n_steps, flags);
  hipGraphExec_t exec = h->graphs.find(key);
  if (!exec) {
    // (never the caller's stream: it may be the NULL stream, which cannot be captured)
    int rc = sgk::host::capture_graph(h->own_stream, "capture step kernels", [&](hipStream_t cap) {
      hipError_t le = hipSuccess;
      for (int32_t k = 0; k < n_steps && le == hipSuccess; ++k) le = sgk::launch_step_counter(s, h->t_dev, (uint64_t)k, flags, cap);
      if (le == hipSuccess) {
        (void)hipGetLastError();
        hipLaunchKernelGGL(add_counter_kernel, dim3(1), dim3(1), 0, cap, h->t_dev, (uint64_t)n_steps);
        le = hipGetLastError();
      }
      return le;
    }, &exec);
    if (rc != SGK_OK) return rc;
    h->graphs.insert(key, exec, h->stream);
  }
  *out = exec;
  return SGK_OK;
}

int sgk_step_random_prepare(sgk_env *h, int32_t n_steps, uint32_t flags) try {
  SGK_CHECK_HANDLE(h);
  if (n_steps < 0) return fail(SGK_ERR_INVALID, "n_steps < 0");
  if (!h->use_graph || n_steps < 4) return SGK_OK;  // these run as eager launches: nothing to prepare
  hipGraphExec_t exec = nullptr;
  int rc = ensure_step_graph(h, n_steps, flags, &exec);
  if (rc != SGK_OK) return rc;
  if (h->t_dev_stale) {  // the counter upload a first replay would otherwise do
    (void)hipGetLastError();
    hipLaunchKernelGGL(set_counter_kernel, dim3(1), dim3(1), 0, h->stream, h->t_dev, h->sh.lockstep_t);
    SGK_HIP(hipGetLastError());
    h->t_dev_stale = false;
  }
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_step_random(sgk_env *h, int32_t n_steps, uint32_t flags) try {
  SGK_CHECK_HANDLE(h);
  if (n_steps < 0) return fail(SGK_ERR_INVALID, "n_steps < 0");
  if (n_steps == 0) return SGK_OK;
  sgk::Shard &s = h->sh;
  if (!h->use_graph || n_steps < 4) {
    for (int32_t k = 0; k < n_steps; ++k) {
      SGK_HIP(sgk::launch_step(s, nullptr, flags, h->stream));
      s.lockstep_t += 1;
    }
    h->t_dev_stale = true;
    h->steps_issued += s.n * (int64_t)n_steps;
    return SGK_OK;
  }
  hipGraphExec_t exec = nullptr;
  int rc = ensure_step_graph(h, n_steps, flags, &exec);
  if (rc != SGK_OK) return rc;
  if (h->t_dev_stale) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(set_counter_kernel, dim3(1), dim3(1), 0, h->stream, h->t_dev, s.lockstep_t);
    SGK_HIP(hipGetLastError());
    h->t_dev_stale = false;
  }
  SGK_HIP(hipGraphLaunch(exec, h->stream));
  s.lockstep_t += (uint64_t)n_steps;
  h->steps_issued += s.n * (int64_t)n_steps;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_step_repeat(sgk_env *h, const uint8_t *actions_dev, int32_t n_steps, uint32_t flags) try {
  SGK_CHECK_HANDLE(h);
  if (!actions_dev) return fail(SGK_ERR_INVALID, "actions_dev is NULL");
  if (n_steps < 0) return fail(SGK_ERR_INVALID, "n_steps < 0");
  for (int32_t k = 0; k < n_steps; ++k) {
    SGK_HIP(sgk::launch_step(h->sh, actions_dev, flags, h->stream));
    h->sh.lockstep_t += 1;
  }
  h->t_dev_stale = true;
  h->steps_issued += h->sh.n * (int64_t)n_steps;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_rollout_random(sgk_env *h, int32_t n_steps, uint32_t flags) try {
  SGK_CHECK_HANDLE(h);
  if (n_steps < 0) return fail(SGK_ERR_INVALID, "n_steps < 0");
  if (n_steps == 0) return SGK_OK;
  SGK_HIP(sgk::launch_rollout_random(h->sh, n_steps, flags, h->stream));
  h->sh.lockstep_t += (uint64_t)n_steps;
  h->t_dev_stale = true;
  h->steps_issued += h->sh.n * (int64_t)n_steps;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_rollout_random_stream(sgk_env *h, int32_t n_steps, uint32_t flags, int8_t *boards_ring_dev, sgk_step_rec *recs_ring_dev,
                              int32_t ring_slices, int32_t first_slice) try {
  SGK_CHECK_HANDLE(h);
  if (n_steps < 0) return fail(SGK_ERR_INVALID, "n_steps < 0");
  if ((boards_ring_dev || recs_ring_dev) && (ring_slices < 1 || first_slice < 0 || first_slice >= ring_slices))
    return fail(SGK_ERR_INVALID, "a trajectory ring needs ring_slices >= 1 and 0 <= first_slice < ring_slices");
  if ((flags & SGK_F_RING_TILE_MAJOR) && boards_ring_dev && ((uintptr_t)boards_ring_dev % 16) != 0)
    return fail(SGK_ERR_INVALID, "a tile-major boards ring must be 16-byte aligned");
  if (n_steps == 0) return SGK_OK;
  const bool rings = boards_ring_dev || recs_ring_dev;
  SGK_HIP(sgk::launch_rollout_stream(h->sh, n_steps, flags, boards_ring_dev, reinterpret_cast<uint32_t *>(recs_ring_dev),
                                     rings ? ring_slices : 1, rings ? first_slice : 0, h->stream));
  h->sh.lockstep_t += (uint64_t)n_steps;
  h->t_dev_stale = true;
  h->steps_issued += h->sh.n * (int64_t)n_steps;
  return SGK_OK;
} SGK_CATCH_STATUS

// ---- trajectory-ring memory: sgk_host_core.h (ring_alloc / ring_free over HIP's virtual-memory management) ----------------------------
int sgk_ring_alloc(int32_t device, size_t bytes, void **dev_ptr) try { return sgk::host::ring_alloc(device, bytes, dev_ptr); } SGK_CATCH_STATUS

int sgk_ring_free(void *dev_ptr) try { return sgk::host::ring_free(dev_ptr); } SGK_CATCH_STATUS

int sgk_ring_probe(sgk_env *h, int8_t *boards_ring_dev, sgk_step_rec *recs_ring_dev, int32_t ring_slices, uint32_t flags,
                   double *us_per_slice) try {
  SGK_CHECK_HANDLE(h);
  if (!us_per_slice) return fail(SGK_ERR_INVALID, "us_per_slice is NULL");
  if (!boards_ring_dev && !recs_ring_dev) return fail(SGK_ERR_INVALID, "no ring to probe");
  if (ring_slices < 1) return fail(SGK_ERR_INVALID, "ring_slices < 1");
  if (flags & ~(uint32_t)SGK_F_RING_TILE_MAJOR) return fail(SGK_ERR_INVALID, "only SGK_F_RING_TILE_MAJOR is meaningful here");
  sgk::Shard &s = h->sh;
  if (boards_ring_dev && (((uintptr_t)boards_ring_dev % 16) != 0 || (!(flags & SGK_F_RING_TILE_MAJOR) && (s.n * s.n_cells) % 16 != 0)))
    return fail(SGK_ERR_INVALID, "the probe writes whole 16-byte-aligned tiles: boards ring and slices must be 16-byte aligned");
  if (s.n < 64) return fail(SGK_ERR_INVALID, "the probe needs at least one whole 64-env tile");
  // one launch to touch every page, then three timed ones: the median, per slice
  hipEvent_t e0 = nullptr, e1 = nullptr;
  SGK_HIP(hipEventCreate(&e0));
  hipError_t err = hipEventCreate(&e1);
  float ms[3] = {0, 0, 0};
  if (err == hipSuccess) err = sgk::launch_ring_probe(s, boards_ring_dev, reinterpret_cast<uint32_t *>(recs_ring_dev), ring_slices, flags, h->stream);
  for (int r = 0; r < 3 && err == hipSuccess; ++r) {
    err = hipEventRecord(e0, h->stream);
    if (err == hipSuccess) err = sgk::launch_ring_probe(s, boards_ring_dev, reinterpret_cast<uint32_t *>(recs_ring_dev), ring_slices, flags, h->stream);
    if (err == hipSuccess) err = hipEventRecord(e1, h->stream);
    if (err == hipSuccess) err = hipEventSynchronize(e1);
    if (err == hipSuccess) err = hipEventElapsedTime(&ms[r], e0, e1);
  }
  (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (err != hipSuccess) return hip_fail(err, "sgk_ring_probe");
  std::sort(ms, ms + 3);
  *us_per_slice = (double)ms[1] * 1e3 / ring_slices;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_boards_dev(sgk_env *h, int8_t **boards_dev, int64_t *pitch) try {
  SGK_CHECK_HANDLE(h);
  if (boards_dev) *boards_dev = h->sh.boards;
  if (pitch) *pitch = h->sh.pitch;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_step_records_dev(sgk_env *h, sgk_step_rec **rec_dev) try {
  if (!h || !rec_dev) return fail(SGK_ERR_INVALID, "NULL argument");
  SGK_CHECK_HANDLE(h);
  *rec_dev = reinterpret_cast<sgk_step_rec *>(h->sh.rec);
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_metrics_dev(sgk_env *h, int64_t **metrics_dev) try {
  if (!h || !metrics_dev) return fail(SGK_ERR_INVALID, "NULL argument");
  SGK_CHECK_HANDLE(h);
  SGK_HIP(sgk::launch_metrics_reduce(h->sh, h->stream));  // folds the per-workgroup partials; stream-ordered
  *metrics_dev = h->sh.metrics;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_episode_arrays_dev(sgk_env *h, int32_t **last_return_dev, int32_t **last_performance_dev, int32_t **n_episodes_dev) try {
  SGK_CHECK_HANDLE(h);
  if (last_return_dev) *last_return_dev = h->sh.last_return;
  if (last_performance_dev) *last_performance_dev = h->sh.last_perf;
  if (n_episodes_dev) *n_episodes_dev = h->sh.n_episodes;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_obs_f32(sgk_env *h, float *dst_dev) try {
  SGK_CHECK_HANDLE(h);
  if (!dst_dev) return fail(SGK_ERR_INVALID, "dst_dev is NULL");
  SGK_HIP(sgk::launch_obs_f32(h->sh, dst_dev, h->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_epsilon_greedy_ex(sgk_env *h, const float *scores_dev, double epsilon, uint64_t draw_index, const double *epsilon_dev,
                          const uint64_t *draw_index_dev, uint8_t *actions_out_dev) try {
  SGK_CHECK_HANDLE(h);
  if (!scores_dev || !actions_out_dev) return fail(SGK_ERR_INVALID, "NULL argument");
  if (((uintptr_t)scores_dev & 15u) != 0) return fail(SGK_ERR_INVALID, "scores_dev must be 16-byte aligned");
  SGK_HIP(sgk::launch_eps_greedy(h->sh, 0, scores_dev, actions_out_dev, epsilon, draw_index, epsilon_dev, draw_index_dev, h->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_categorical_sample(sgk_env *h, const float *logits_dev, uint64_t draw_index, const uint64_t *draw_index_dev,
                           uint8_t *actions_out_dev) try {
  SGK_CHECK_HANDLE(h);
  if (!logits_dev || !actions_out_dev) return fail(SGK_ERR_INVALID, "NULL argument");
  if (((uintptr_t)logits_dev & 15u) != 0) return fail(SGK_ERR_INVALID, "logits_dev must be 16-byte aligned");
  SGK_HIP(sgk::launch_eps_greedy(h->sh, 1, logits_dev, actions_out_dev, 0.0, draw_index, nullptr, draw_index_dev, h->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_epsilon_greedy(sgk_env *h, const float *scores_dev, double epsilon, uint64_t draw_index, uint8_t *actions_out_dev) try {
  return sgk_epsilon_greedy_ex(h, scores_dev, epsilon, draw_index, nullptr, nullptr, actions_out_dev);
} SGK_CATCH_STATUS

int sgk_policy_act(sgk_env *h, const sgk_mlp_weights *w, double epsilon, uint64_t draw_index, const double *epsilon_dev,
                   const uint64_t *draw_index_dev, uint8_t *actions_out_dev, float *scores_out_dev) try {
  SGK_CHECK_HANDLE(h);
  if (!w || !actions_out_dev || !w->w1t || !w->b1 || !w->w2 || !w->b2 || !w->w3t || !w->b3)
    return fail(SGK_ERR_INVALID, "NULL argument");
  if (w->n_hidden != 64 && w->n_hidden != 100 && w->n_hidden != 128)
    return fail(SGK_ERR_INVALID, "sgk_policy_act is built for n_hidden in {64, 100 (the reference default), 128}");
  if (scores_out_dev && ((uintptr_t)scores_out_dev & 15u)) return fail(SGK_ERR_INVALID, "scores_out_dev must be 16-byte aligned");
  sgk::PolicyWeights pw{w->w1t, w->b1, w->w2, w->b2, w->w3t, w->b3, w->n_hidden};
  SGK_HIP(sgk::launch_policy_act(h->sh, 0, pw, actions_out_dev, scores_out_dev, epsilon, draw_index, epsilon_dev, draw_index_dev,
                                 h->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_policy_sample(sgk_env *h, const sgk_mlp_weights *w, uint64_t draw_index, const uint64_t *draw_index_dev,
                      uint8_t *actions_out_dev, float *logits_out_dev) try {
  SGK_CHECK_HANDLE(h);
  if (!w || !actions_out_dev || !w->w1t || !w->b1 || !w->w2 || !w->b2 || !w->w3t || !w->b3)
    return fail(SGK_ERR_INVALID, "NULL argument");
  if (w->n_hidden != 64 && w->n_hidden != 100 && w->n_hidden != 128)
    return fail(SGK_ERR_INVALID, "sgk_policy_sample is built for n_hidden in {64, 100 (the reference default), 128}");
  if (logits_out_dev && ((uintptr_t)logits_out_dev & 15u)) return fail(SGK_ERR_INVALID, "logits_out_dev must be 16-byte aligned");
  sgk::PolicyWeights pw{w->w1t, w->b1, w->w2, w->b2, w->w3t, w->b3, w->n_hidden};
  SGK_HIP(sgk::launch_policy_act(h->sh, 1, pw, actions_out_dev, logits_out_dev, 0.0, draw_index, nullptr, draw_index_dev,
                                 h->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_policy_rollout(sgk_env *h, const sgk_mlp_weights *w, int32_t mode, double epsilon, uint64_t draw_index0, int32_t n_steps,
                       uint32_t flags, int8_t *states_out_dev, uint8_t *actions_out_dev, sgk_step_rec *recs_out_dev) try {
  SGK_CHECK_HANDLE(h);
  if (!w || !w->w1t || !w->b1 || !w->w2 || !w->b2 || !w->w3t || !w->b3) return fail(SGK_ERR_INVALID, "NULL argument");
  if (w->n_hidden != 64 && w->n_hidden != 100 && w->n_hidden != 128)
    return fail(SGK_ERR_INVALID, "sgk_policy_rollout is built for n_hidden in {64, 100 (the reference default), 128}");
  if (mode != 0 && mode != 1) return fail(SGK_ERR_INVALID, "mode must be 0 (epsilon-greedy) or 1 (categorical)");
  if (n_steps < 0) return fail(SGK_ERR_INVALID, "n_steps < 0");
  if (flags & ~(uint32_t)(SGK_F_AUTO_RESET | SGK_F_MASK_FINISHED))
    return fail(SGK_ERR_INVALID, "only SGK_F_AUTO_RESET and SGK_F_MASK_FINISHED are meaningful here");
  if (n_steps == 0) return SGK_OK;
  sgk::Shard &s = h->sh;
  sgk::PolicyWeights pw{w->w1t, w->b1, w->w2, w->b2, w->w3t, w->b3, w->n_hidden};
  SGK_HIP(sgk::launch_policy_rollout(s, mode, pw, epsilon, draw_index0, n_steps, flags, states_out_dev, actions_out_dev,
                                     reinterpret_cast<uint32_t *>(recs_out_dev), h->stream));
  SGK_HIP(sgk::launch_reset(s, nullptr, 2, h->stream));  // materialise the boards of the final states
  s.lockstep_t += (uint64_t)n_steps;
  h->t_dev_stale = true;
  h->steps_issued += s.n * n_steps;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_replay_store(sgk_env *h, int32_t phase, const uint8_t *actions_dev, int32_t cheat, int64_t slice, const int64_t *slice_dev,
                     int8_t *states_ring, int8_t *successors_ring, uint8_t *actions_ring, int8_t *rewards_ring,
                     uint8_t *terminals_ring) try {
  SGK_CHECK_HANDLE(h);
  if (phase != 0 && phase != 1) return fail(SGK_ERR_INVALID, "phase must be 0 (before env.step) or 1 (after)");
  if (!states_ring || !successors_ring) return fail(SGK_ERR_INVALID, "NULL ring pointer");
  if (phase == 1 && (!actions_ring || !rewards_ring || !terminals_ring || (!cheat && !actions_dev)))
    return fail(SGK_ERR_INVALID, "phase 1 needs the action / reward / terminal rings and, unless cheat, the actions");
  if (slice < 0) return fail(SGK_ERR_INVALID, "slice < 0");
  SGK_HIP(sgk::launch_replay_store(h->sh, phase, actions_dev, cheat, slice, reinterpret_cast<const long long *>(slice_dev),
                                   states_ring, successors_ring, actions_ring, rewards_ring, terminals_ring, h->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

static int convq_weights_ok(const sgk_convq_weights *w) {
  if (!w || !w->w1 || !w->b1 || !w->w2 || !w->b2 || !w->wb || !w->bb || !w->wh || !w->bh || !w->wl || !w->bl)
    return fail(SGK_ERR_INVALID, "NULL argument");
  if (w->n_layers != 2) return fail(SGK_ERR_INVALID, "the fused conv body is built for n_layers == 2 (two 3 x 3 convolutions in the trunk)");
  if (w->n_channels != 4 && w->n_channels != 5 && w->n_channels != 8)
    return fail(SGK_ERR_INVALID, "the fused conv body is built for n_channels in {4, 5 (policy_cnn.py's default), 8}");
  return SGK_OK;
}

static int convq_launch(sgk_env *h, const sgk_convq_weights *w, int mode, double epsilon, uint64_t draw_index, const double *epsilon_dev,
                        const uint64_t *draw_index_dev, uint8_t *actions_out_dev, float *scores_out_dev) {
  SGK_CHECK_HANDLE(h);
  if (int rc = convq_weights_ok(w)) return rc;
  if (!actions_out_dev) return fail(SGK_ERR_INVALID, "NULL argument");
  if (scores_out_dev && ((uintptr_t)scores_out_dev & 15u)) return fail(SGK_ERR_INVALID, "the scores / logits output must be 16-byte aligned");
  sgk::ConvQWeights cw{w->w1, w->b1, w->w2, w->b2, w->wb, w->bb, w->wh, w->bh, w->wl, w->bl};
  SGK_HIP(sgk::launch_convq_act(h->sh, cw, w->n_channels, mode, actions_out_dev, scores_out_dev, epsilon, draw_index, epsilon_dev,
                                draw_index_dev, h->stream));
  return SGK_OK;
}

int sgk_convq_rollout(sgk_env *h, const sgk_convq_weights *w, int32_t mode, double epsilon, uint64_t draw_index0, int32_t n_steps,
                      uint32_t flags, int8_t *states_out_dev, uint8_t *actions_out_dev, sgk_step_rec *recs_out_dev) try {
  SGK_CHECK_HANDLE(h);
  if (int rc = convq_weights_ok(w)) return rc;
  if (mode != 0 && mode != 1) return fail(SGK_ERR_INVALID, "mode must be 0 (epsilon-greedy) or 1 (categorical)");
  if (n_steps < 0) return fail(SGK_ERR_INVALID, "n_steps < 0");
  if (flags & ~(uint32_t)(SGK_F_AUTO_RESET | SGK_F_MASK_FINISHED))
    return fail(SGK_ERR_INVALID, "only SGK_F_AUTO_RESET and SGK_F_MASK_FINISHED are meaningful here");
  if (n_steps == 0) return SGK_OK;
  sgk::Shard &s = h->sh;
  sgk::ConvQWeights cw{w->w1, w->b1, w->w2, w->b2, w->wb, w->bb, w->wh, w->bh, w->wl, w->bl};
  SGK_HIP(sgk::launch_convq_rollout(s, cw, w->n_channels, mode, epsilon, draw_index0, n_steps, flags, states_out_dev, actions_out_dev,
                                    reinterpret_cast<uint32_t *>(recs_out_dev), h->stream));
  SGK_HIP(sgk::launch_reset(s, nullptr, 2, h->stream));  // materialise the boards of the final states
  s.lockstep_t += (uint64_t)n_steps;
  h->t_dev_stale = true;
  h->steps_issued += s.n * n_steps;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_convq_act(sgk_env *h, const sgk_convq_weights *w, double epsilon, uint64_t draw_index, const double *epsilon_dev,
                  const uint64_t *draw_index_dev, uint8_t *actions_out_dev, float *scores_out_dev) try {
  return convq_launch(h, w, 0, epsilon, draw_index, epsilon_dev, draw_index_dev, actions_out_dev, scores_out_dev);
} SGK_CATCH_STATUS

int sgk_convq_sample(sgk_env *h, const sgk_convq_weights *w, uint64_t draw_index, const uint64_t *draw_index_dev, uint8_t *actions_out_dev,
                     float *logits_out_dev) try {
  return convq_launch(h, w, 1, 0.0, draw_index, nullptr, draw_index_dev, actions_out_dev, logits_out_dev);
} SGK_CATCH_STATUS

int sgk_step_store(sgk_env *h, const uint8_t *actions_dev, uint32_t flags, int32_t cheat, int64_t slice, const int64_t *slice_dev,
                   int32_t ring_slices, int8_t *successors_ring, uint8_t *actions_ring, int8_t *rewards_ring, uint8_t *terminals_ring) try {
  SGK_CHECK_HANDLE(h);
  if (!actions_dev) return fail(SGK_ERR_INVALID, "actions_dev is NULL");
  if (flags & ~(uint32_t)SGK_F_NO_BOARDS) return fail(SGK_ERR_INVALID, "only SGK_F_NO_BOARDS is meaningful here (the finished envs are reset by sgk_reset_done[_store])");
  if (!successors_ring || !actions_ring || !rewards_ring || !terminals_ring) return fail(SGK_ERR_INVALID, "NULL ring pointer");
  if (slice < 0 || ring_slices < 0 || (slice_dev && ring_slices < 1)) return fail(SGK_ERR_INVALID, "bad slice / ring_slices");
  SGK_HIP(sgk::launch_step_store(h->sh, actions_dev, flags, cheat ? 1 : 0, slice, reinterpret_cast<const long long *>(slice_dev), ring_slices,
                                 successors_ring, actions_ring, rewards_ring, terminals_ring, h->stream));
  h->sh.lockstep_t += 1;
  h->t_dev_stale = true;
  h->steps_issued += h->sh.n;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_reset_done_store(sgk_env *h, uint32_t flags, int64_t slice, const int64_t *slice_dev, int32_t ring_slices, int8_t *states_ring) try {
  SGK_CHECK_HANDLE(h);
  if (flags & ~(uint32_t)SGK_F_NO_BOARDS) return fail(SGK_ERR_INVALID, "only SGK_F_NO_BOARDS is meaningful here");
  if (!states_ring) return fail(SGK_ERR_INVALID, "NULL ring pointer");
  if (slice < 0 || ring_slices < 0 || (slice_dev && ring_slices < 1)) return fail(SGK_ERR_INVALID, "bad slice / ring_slices");
  SGK_HIP(sgk::launch_reset_done_store(h->sh, flags, slice, reinterpret_cast<const long long *>(slice_dev), ring_slices, states_ring, h->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

struct DqnResetStore {  // sgk_dqn_sgd_step_reset_store's second half
  uint32_t flags;
  int64_t slice;
  const int64_t *slice_dev;
  int32_t ring_slices;
  int8_t *states_ring;
};

static int dqn_sgd_step_impl(sgk_env *h, const sgk_dqn_learner *L, const DqnResetStore *rs) {
  SGK_CHECK_HANDLE(h);
  if (!L) return fail(SGK_ERR_INVALID, "learner is NULL");
  const void *need[] = {L->states, L->successors, L->actions, L->rewards, L->terminals, L->w1, L->b1, L->w2, L->b2, L->w3, L->b3,
                        L->w1t, L->w2t, L->w3t, L->tw1t, L->tb1, L->tw2t, L->tb2, L->tw3, L->tb3, L->step};
  for (const void *p : need)
    if (!p) return fail(SGK_ERR_INVALID, "NULL pointer in sgk_dqn_learner");
  for (int i = 0; i < 6; ++i)
    if (!L->m[i] || !L->v[i] || !L->vmax[i]) return fail(SGK_ERR_INVALID, "NULL Adam state in sgk_dqn_learner");
  if ((L->n_hidden != 64 && L->n_hidden != 100) || L->batch < 1 || L->batch > 64 || L->slices_filled < 1)
    return fail(SGK_ERR_INVALID, "sgk_dqn_sgd_step needs n_hidden 64 or 100 (the reference default), 1 <= batch <= 64, slices_filled >= 1");
  if (L->loss_mode != SGK_DQN_LOSS_REFERENCE && L->loss_mode != SGK_DQN_LOSS_PER_SAMPLE)
    return fail(SGK_ERR_INVALID, "loss_mode must be SGK_DQN_LOSS_REFERENCE (0) or SGK_DQN_LOSS_PER_SAMPLE (1)");
  if ((int64_t)L->slices_filled * h->sh.n > (int64_t)INT32_MAX)
    return fail(SGK_ERR_INVALID, "the minibatch is drawn with 32-bit transition indices: slices_filled * n_envs must stay below 2^31");
  if (sgk::dqn_sgd_lds_bytes(h->sh.n_cells, L->n_hidden) > 160u * 1024u)
    return fail(SGK_ERR_INVALID, "this n_cells / n_hidden does not fit the 160 KB of LDS the kernel works in");
  sgk::DqnLearner d;
  d.states = L->states; d.successors = L->successors; d.actions = L->actions; d.rewards = L->rewards; d.terminals = L->terminals;
  d.slices_filled = L->slices_filled;
  d.w1 = L->w1; d.b1 = L->b1; d.w2 = L->w2; d.b2 = L->b2; d.w3 = L->w3; d.b3 = L->b3;
  d.w1t = L->w1t; d.w2t = L->w2t; d.w3t = L->w3t;
  for (int i = 0; i < 6; ++i) { d.m[i] = L->m[i]; d.v[i] = L->v[i]; d.vmax[i] = L->vmax[i]; }
  d.tw1t = L->tw1t; d.tb1 = L->tb1; d.tw2t = L->tw2t; d.tb2 = L->tb2; d.tw3 = L->tw3; d.tb3 = L->tb3;
  d.step = reinterpret_cast<long long *>(L->step);
  d.loss_out = L->loss_out;
  d.n_hidden = L->n_hidden; d.batch = L->batch; d.loss_mode = L->loss_mode;
  d.rows = reinterpret_cast<const long long *>(L->rows); d.rows_out = reinterpret_cast<long long *>(L->rows_out);
  // Adam runs as a second launch over the whole chip (sgk_learn.hip: AdamHeader): it needs a scratch block of the handle's for the
  // gradient, made on the first call -- NOT inside a stream capture: warm the learner up before recording it, as
  // BatchedDeepQAgent.enable_graphs does. SGK_DQN_ONE_LAUNCH=1 in the environment when the library is loaded keeps the update inside
  // the one kernel (the A/B knob of profiles/r06).
  static const bool one_launch = [] { const char *v = getenv("SGK_DQN_ONE_LAUNCH"); return v && v[0] == '1' && v[1] == 0; }();
  d.scratch = nullptr;
  d.multi_wg = 0;
#ifdef SGK_DQN_MULTI_WG
  static const bool four_workgroups = [] { const char *v = getenv("SGK_DQN_WORKGROUPS"); return v && v[0] == '4' && v[1] == 0; }();
  d.multi_wg = four_workgroups ? 1 : 0;
#endif
  if (rs) {
    d.reset_store = 1;
    d.rs_flags = rs->flags; d.rs_slice = rs->slice; d.rs_slice_dev = reinterpret_cast<const long long *>(rs->slice_dev);
    d.rs_ring = rs->ring_slices; d.rs_states_ring = rs->states_ring;
  }
  if (!one_launch || d.multi_wg || rs) {
    const size_t need = sgk::dqn_sgd_scratch_bytes(h->sh.n_cells, L->n_hidden);
    if (h->learn_scratch_bytes < need) {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (h->stream) (void)hipStreamIsCapturing(h->stream, &cs);
      if (cs != hipStreamCaptureStatusNone)
        return fail(SGK_ERR_INVALID, "the first sgk_dqn_sgd_step of a handle allocates its scratch block: call it once before capturing it");
      SGK_HIP(sgk::host::wait_stream(h->stream));
      (void)hipFree(h->learn_scratch);
      h->learn_scratch = nullptr;
      h->learn_scratch_bytes = 0;
      SGK_HIP(hipMalloc(&h->learn_scratch, need));
      SGK_HIP(hipMemsetAsync(h->learn_scratch, 0, need, h->stream));
      h->learn_scratch_bytes = need;
    }
    d.scratch = h->learn_scratch;
  }
  d.lr = L->lr; d.beta1 = L->beta1; d.beta2 = L->beta2; d.eps = L->eps; d.discount = L->discount; d.max_grad_norm = L->max_grad_norm;
  if (rs && d.multi_wg) return fail(SGK_ERR_INVALID, "the four-workgroup experiment has no fused reset");
  SGK_HIP(sgk::launch_dqn_sgd(h->sh, d, h->stream));
  return SGK_OK;
}

int sgk_dqn_sgd_step(sgk_env *h, const sgk_dqn_learner *L) try {
  return dqn_sgd_step_impl(h, L, nullptr);
} SGK_CATCH_STATUS

int sgk_dqn_sgd_step_reset_store(sgk_env *h, const sgk_dqn_learner *L, uint32_t flags, int64_t slice, const int64_t *slice_dev,
                                 int32_t ring_slices, int8_t *states_ring) try {
  SGK_CHECK_HANDLE(h);
  if (flags & ~(uint32_t)SGK_F_NO_BOARDS) return fail(SGK_ERR_INVALID, "only SGK_F_NO_BOARDS is meaningful here");
  if (!states_ring) return fail(SGK_ERR_INVALID, "NULL ring pointer");
  if (slice < 0 || ring_slices < 0 || (slice_dev && ring_slices < 1)) return fail(SGK_ERR_INVALID, "bad slice / ring_slices");
  const DqnResetStore rs{flags, slice, slice_dev, ring_slices, states_ring};
  return dqn_sgd_step_impl(h, L, &rs);
} SGK_CATCH_STATUS

int sgk_ppo_epochs(sgk_env *h, const sgk_ppo_learner *L) try {
  SGK_CHECK_HANDLE(h);
  if (!L) return fail(SGK_ERR_INVALID, "learner is NULL");
  const void *need[] = {L->states, L->actions, L->returns, L->lengths, L->w1, L->b1, L->w2, L->b2, L->wa, L->ba, L->wc, L->bc,
                        L->w1t, L->w2t, L->ow1t, L->ob1, L->ow2t, L->ob2, L->owa, L->oba, L->step};
  for (const void *p : need)
    if (!p) return fail(SGK_ERR_INVALID, "NULL pointer in sgk_ppo_learner");
  for (int i = 0; i < 8; ++i)
    if (!L->m[i] || !L->v[i]) return fail(SGK_ERR_INVALID, "NULL Adam state in sgk_ppo_learner");
  if ((L->n_hidden != 64 && L->n_hidden != 100) || L->batch < 2 || L->batch > 64 || L->n_epochs < 1 || L->horizon < 1 ||
      L->n_trajectories < 1 || L->n_trajectories >= (1ll << 31))
    return fail(SGK_ERR_INVALID, "sgk_ppo_epochs needs n_hidden 64 or 100 (the reference default), 2 <= batch <= 64, n_epochs >= 1, "
                                 "horizon >= 1, 1 <= n_trajectories < 2^31");
  if (sgk::ppo_epochs_lds_bytes(h->sh.n_cells, L->n_hidden) > 160u * 1024u)
    return fail(SGK_ERR_INVALID, "this n_cells / n_hidden does not fit the 160 KB of LDS the kernel works in");
  sgk::PpoLearner d;
  d.states = L->states; d.actions = L->actions; d.returns = L->returns; d.lengths = L->lengths;
  d.horizon = L->horizon; d.n_hidden = L->n_hidden; d.batch = L->batch; d.n_epochs = L->n_epochs;
  d.n_trajectories = L->n_trajectories;
  d.w1 = L->w1; d.b1 = L->b1; d.w2 = L->w2; d.b2 = L->b2; d.wa = L->wa; d.ba = L->ba; d.wc = L->wc; d.bc = L->bc;
  d.w1t = L->w1t; d.w2t = L->w2t;
  for (int i = 0; i < 8; ++i) { d.m[i] = L->m[i]; d.v[i] = L->v[i]; }
  d.ow1t = L->ow1t; d.ob1 = L->ob1; d.ow2t = L->ow2t; d.ob2 = L->ob2; d.owa = L->owa; d.oba = L->oba;
  d.step = reinterpret_cast<long long *>(L->step);
  d.stats_out = L->stats_out;
  d.rows = reinterpret_cast<const long long *>(L->rows);
  d.rows_out = reinterpret_cast<long long *>(L->rows_out);
  d.lr = L->lr; d.beta1 = L->beta1; d.beta2 = L->beta2; d.eps = L->eps;
  d.clipping = L->clipping; d.critic_coeff = L->critic_coeff; d.entropy_bonus = L->entropy_bonus;
  SGK_HIP(sgk::launch_ppo_epochs(h->sh, d, h->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_discounted_returns(sgk_env *h, const float *rewards_dev, const int32_t *lengths_dev, float *returns_dev,
                           int64_t n_trajectories, int32_t t_max, double discount) try {
  SGK_CHECK_HANDLE(h);
  if (!rewards_dev || !returns_dev) return fail(SGK_ERR_INVALID, "NULL argument");
  if (n_trajectories < 0 || t_max < 1 || t_max > 1024) return fail(SGK_ERR_INVALID, "t_max must be in 1..1024");
  if (n_trajectories == 0) return SGK_OK;
  if (!h->gamma_dev) SGK_HIP(hipMalloc(&h->gamma_dev, sizeof(float) * 1024));
  if (h->gamma_discount != discount) {
    float tab[1024];
    for (int t = 0; t < 1024; ++t) tab[t] = (float)std::pow(discount, (double)t);  // Python: float ** int, then float32
    SGK_HIP(sgk::host::wait_stream(h->stream));  // a previous launch may still read the old table
    SGK_HIP(hipMemcpyAsync(h->gamma_dev, tab, sizeof(tab), hipMemcpyHostToDevice, h->stream));  // (not hipMemcpy: sgk::capture_mutex)
    SGK_HIP(sgk::host::wait_stream(h->stream));  // `tab` is on the stack
    h->gamma_discount = discount;
  }
  SGK_HIP(sgk::launch_discounted_returns(h->sh, rewards_dev, lengths_dev, h->gamma_dev, returns_dev, n_trajectories, t_max,
                                         h->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_render_rgb(sgk_env *h, uint8_t *rgb_dev) try {
  SGK_CHECK_HANDLE(h);
  if (!rgb_dev) return fail(SGK_ERR_INVALID, "rgb_dev is NULL");
  SGK_HIP(sgk::launch_render_rgb(h->sh, rgb_dev, h->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

// Host-visible handles keep what the host reads in pinned host memory. While the step server is resident that memory is current
// whenever the host is not inside a request (the server writes, fences and only then publishes the answer): the copies below then
// read it as it is -- stopping the server for a look would cost a launch at the next step. Otherwise: wait for the stream first.
static int host_visible_ready(sgk_env *h) {
  if (h->srv.running) {
    __sync_synchronize();
    return SGK_OK;
  }
  SGK_HIP(sgk::host::wait_stream(h->stream));
  return SGK_OK;
}

int sgk_copy_boards(sgk_env *h, int8_t *boards_host) try {
  if (h && h->host_visible && h->srv.running && boards_host) {
    const sgk::Shard &s = h->sh;
    __sync_synchronize();
    for (int64_t i = 0; i < s.n; ++i) memcpy(boards_host + i * s.n_cells, s.boards + i * s.pitch, (size_t)s.n_cells);
    return SGK_OK;
  }
  SGK_CHECK_HANDLE(h);
  if (!boards_host) return fail(SGK_ERR_INVALID, "boards_host is NULL");
  sgk::Shard &s = h->sh;
  const size_t bytes = (size_t)s.n * s.n_cells;
  if (h->host_visible) {
    // The boards live in pinned HOST memory: wait for the stream, then read them with the CPU. (A hipMemcpyAsync between two host
    // pointers is carried out by the runtime on the calling thread at once -- it is not ordered behind the kernels of the stream --
    // so env.reset() now and then returned the board of the step BEFORE the reset kernel had written: one stale observation in
    // ~5 000 resets, caught by the reference's train() goldens on the single env.)
    SGK_HIP(sgk::host::wait_stream(h->stream));
    for (int64_t i = 0; i < s.n; ++i) memcpy(boards_host + i * s.n_cells, s.boards + i * s.pitch, (size_t)s.n_cells);
    return SGK_OK;
  }
  if (s.pitch == s.n_cells) {
    SGK_HIP(hipMemcpyAsync(boards_host, s.boards, bytes, hipMemcpyDeviceToHost, h->stream));
  } else {
    if (!h->dense_scratch) SGK_HIP(hipMalloc(&h->dense_scratch, bytes));
    SGK_HIP(sgk::launch_dense_boards(s, h->dense_scratch, h->stream));
    SGK_HIP(hipMemcpyAsync(boards_host, h->dense_scratch, bytes, hipMemcpyDeviceToHost, h->stream));
  }
  SGK_HIP(sgk::host::wait_stream(h->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_copy_step_records(sgk_env *h, sgk_step_rec *rec_host) try {
  SGK_CHECK_HANDLE(h);
  if (!rec_host) return fail(SGK_ERR_INVALID, "rec_host is NULL");
  if (h->host_visible) {  // (host memory: synchronise, then a CPU copy -- see sgk_copy_boards)
    SGK_HIP(sgk::host::wait_stream(h->stream));
    memcpy(rec_host, h->sh.rec, sizeof(uint32_t) * (size_t)h->sh.n);
    return SGK_OK;
  }
  SGK_HIP(hipMemcpyAsync(rec_host, h->sh.rec, sizeof(uint32_t) * h->sh.n, hipMemcpyDeviceToHost, h->stream));
  SGK_HIP(sgk::host::wait_stream(h->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_copy_episode_state(sgk_env *h, int32_t *episode_return_host, int32_t *hidden_return_host, int32_t *frame_host,
                           uint8_t *over_host, uint8_t *agent_cell_host, uint8_t *box_cell_host) try {
  SGK_CHECK_HANDLE(h);
  const size_t n = (size_t)h->sh.n;
  auto unpack = [&](const uint64_t *w, size_t first, size_t count) {
    for (size_t k = 0; k < count; ++k) {
      const size_t i = first + k;
      const uint32_t lo = (uint32_t)w[k], hi = (uint32_t)(w[k] >> 32);
      if (agent_cell_host) agent_cell_host[i] = (uint8_t)(lo & 0xff);
      if (box_cell_host) box_cell_host[i] = (uint8_t)((lo >> 8) & 0xff);
      if (frame_host) frame_host[i] = (int32_t)((lo >> 16) & 0xff);
      if (over_host) over_host[i] = (uint8_t)((lo >> 24) & 1);
      if (episode_return_host) episode_return_host[i] = (int32_t)(int16_t)(hi & 0xffff);
      if (hidden_return_host) hidden_return_host[i] = (int32_t)(int16_t)(hi >> 16);
    }
  };
  if (h->host_visible) {  // (host memory: synchronise, then read the words where they are -- see sgk_copy_boards)
    SGK_HIP(sgk::host::wait_stream(h->stream));
    unpack(h->sh.state, 0, n);
    return SGK_OK;
  }
  // device memory: the state words come over in chunks through one small pinned block of the handle's (a staging array of n words
  // would be 16 GB of host memory at the largest batch the library takes)
  constexpr size_t CHUNK = (size_t)1 << 17;  // 1 MiB of state words
  if (!h->copy_chunk) SGK_HIP(hipHostMalloc((void **)&h->copy_chunk, sizeof(uint64_t) * CHUNK, hipHostMallocDefault));
  for (size_t first = 0; first < n; first += CHUNK) {
    const size_t count = n - first < CHUNK ? n - first : CHUNK;
    SGK_HIP(hipMemcpyAsync(h->copy_chunk, h->sh.state + first, sizeof(uint64_t) * count, hipMemcpyDeviceToHost, h->stream));
    SGK_HIP(sgk::host::wait_stream(h->stream));
    unpack(h->copy_chunk, first, count);
  }
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_copy_last_episode(sgk_env *h, int32_t *last_return_host, int32_t *last_performance_host, int32_t *n_episodes_host) try {
  if (h && h->host_visible && !n_episodes_host) {  // (the two arrays are pinned host memory there: see host_visible_ready)
    SGK_HIP(hipSetDevice(h->sh.device));
    const int rc = host_visible_ready(h);
    if (rc != SGK_OK) return rc;
    const size_t nb = sizeof(int32_t) * (size_t)h->sh.n;
    if (last_return_host) memcpy(last_return_host, h->sh.last_return, nb);
    if (last_performance_host) memcpy(last_performance_host, h->sh.last_perf, nb);
    return SGK_OK;
  }
  SGK_CHECK_HANDLE(h);
  const size_t bytes = sizeof(int32_t) * (size_t)h->sh.n;
  if (n_episodes_host) SGK_HIP(hipMemcpyAsync(n_episodes_host, h->sh.n_episodes, bytes, hipMemcpyDeviceToHost, h->stream));
  if (!h->host_visible) {
    if (last_return_host) SGK_HIP(hipMemcpyAsync(last_return_host, h->sh.last_return, bytes, hipMemcpyDeviceToHost, h->stream));
    if (last_performance_host)
      SGK_HIP(hipMemcpyAsync(last_performance_host, h->sh.last_perf, bytes, hipMemcpyDeviceToHost, h->stream));
  }
  SGK_HIP(sgk::host::wait_stream(h->stream));
  if (h->host_visible) {  // host memory: read with the CPU once the stream is idle (a host-to-host hipMemcpyAsync is not stream-ordered)
    if (last_return_host) memcpy(last_return_host, h->sh.last_return, bytes);
    if (last_performance_host) memcpy(last_performance_host, h->sh.last_perf, bytes);
  }
  return SGK_OK;
} SGK_CATCH_STATUS

// Wait for the handle's stream with latency in mind: poll for a bounded time (a blocking wait sleeps on an interrupt and
// wakes tens of microseconds late), then fall back to the blocking form.
static hipError_t wait_stream_low_latency(hipStream_t st) {
  hipError_t q = hipErrorNotReady;
  for (int spin = 0; st && spin < 200000 && q == hipErrorNotReady; ++spin) q = hipStreamQuery(st);  // ~0.1 s at most; the NULL stream goes to wait_stream
  if (q == hipErrorNotReady) q = sgk::host::wait_stream(st);
  return q;
}

int sgk_metrics(sgk_env *h, int64_t out_host[SGK_METRICS_LEN]) try {
  SGK_CHECK_HANDLE(h);
  if (!out_host) return fail(SGK_ERR_INVALID, "out_host is NULL");
  // the fold's 16 words land in pinned device-mapped host memory as well: one launch, one wait, no copy command
  SGK_HIP(sgk::launch_metrics_reduce(h->sh, h->stream, h->metrics_pinned));
  SGK_HIP(wait_stream_low_latency(h->stream));
  for (int i = 0; i < SGK_METRICS_LEN; ++i) out_host[i] = (int64_t)h->metrics_pinned[i];
  out_host[SGK_M_STEPS] = h->steps_issued;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_metrics_allreduced(sgk_env *h, sgk_comm *comm, int64_t out_host[SGK_METRICS_LEN]) try {
  SGK_CHECK_HANDLE(h);
  if (!comm || !out_host) return fail(SGK_ERR_INVALID, "NULL argument");
  SGK_HIP(sgk::launch_metrics_reduce(h->sh, h->stream));
  // SGK_M_STEPS is counted on the host (steps issued): put it into the device vector so that it is summed with the rest
  (void)hipGetLastError();
  hipLaunchKernelGGL(set_counter_kernel, dim3(1), dim3(1), 0, h->stream, reinterpret_cast<uint64_t *>(h->sh.metrics) + SGK_M_STEPS,
                     (uint64_t)h->steps_issued);
  SGK_HIP(hipGetLastError());
  int rc = sgk_allreduce_metrics(comm, h->sh.metrics, (void *)h->stream);
  if (rc != SGK_OK) return rc;
  SGK_HIP(hipMemcpyAsync(out_host, h->sh.metrics, sizeof(int64_t) * SGK_METRICS_LEN, hipMemcpyDeviceToHost, h->stream));
  SGK_HIP(sgk::host::wait_stream(h->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_metrics_reset(sgk_env *h) try {
  SGK_CHECK_HANDLE(h);
  SGK_HIP(sgk::launch_metrics_init(h->sh, h->stream));
  h->steps_issued = 0;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_finished(sgk_env *h, int32_t *ids_dev, int32_t *return_dev, int32_t *performance_dev, int64_t *n_host) try {
  SGK_CHECK_HANDLE(h);
  if (!ids_dev || !return_dev || !performance_dev || !n_host) return fail(SGK_ERR_INVALID, "NULL argument");
  SGK_HIP(sgk::launch_finished(h->sh, ids_dev, return_dev, performance_dev, h->stream));
  SGK_HIP(hipMemcpyAsync(n_host, h->sh.finished_total, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  SGK_HIP(sgk::host::wait_stream(h->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

// ---- tabular Q ------------------------------------------------------------------------------------

int sgk_tabq_destroy(sgk_tabq *q) try {
  if (!q) return SGK_OK;
  if (q->env) {
    (void)hipSetDevice(q->env->sh.device);
    (void)sgk::host::wait_stream(q->env->stream);
  }
  q->graphs.clear();
  (void)hipFree(q->tq.table);
  (void)hipFree(q->tq.tags);
  (void)hipFree(q->tq.keys);
  (void)hipFree(q->tq.hash_overflow);
  (void)hipFree(q->tq.row_cache);
  if (q->copy_stage) (void)hipHostFree(q->copy_stage);
  (void)hipFree(q->actions);
  (void)hipFree(q->t_dev);
  delete q;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_tabq_create_ex(sgk_env *env, double lr, double discount, double epsilon, int64_t epsilon_anneal, int32_t hash_capacity,
                       sgk_tabq **out) try {
  if (!out) return fail(SGK_ERR_INVALID, "out is NULL");
  *out = nullptr;
  SGK_CHECK_HANDLE(env);
  if (epsilon_anneal < 1) return fail(SGK_ERR_INVALID, "epsilon_anneal < 1");
  // (also keeps NaN out: the LDS-resident kernel turns epsilon into an integer threshold, undefined for a negative double)
  if (!(epsilon >= 0.0 && epsilon <= 1.0)) return fail(SGK_ERR_INVALID, "epsilon must be a probability (0 <= epsilon <= 1)");
  const bool hashed = env->sh.env_id == SGK_TOMATO_WATERING;  // 63 x 2^13 distinct boards: no perfect hash
  if (hashed) {
    if (hash_capacity == 0) hash_capacity = 4096;
    if (hash_capacity < 64 || hash_capacity > (1 << 24) || (hash_capacity & (hash_capacity - 1)))
      return fail(SGK_ERR_INVALID, "hash_capacity must be a power of two in 64 .. 2^24 (slots per agent; 0 = 4096)");
  } else if (hash_capacity != 0) {
    return fail(SGK_ERR_INVALID, "this level's boards have a perfect hash: hash_capacity must be 0");
  }
  sgk_tabq *q = sgk::host::host_new<sgk_tabq>();  // (std::bad_alloc -> SGK_ERR_NOMEM at the entry point's barrier)
  q->env = env;
  q->tq.lr = lr;
  q->tq.discount = discount;
  q->tq.eps0 = epsilon;
  q->tq.anneal = epsilon_anneal;
  q->tq.t_agent = 0;
  q->tq.hash_cap = hashed ? hash_capacity : 0;
  q->tq.n_states = hashed ? hash_capacity : env->sh.n_states;
  const size_t n = (size_t)env->sh.n;
  const size_t tbytes = sizeof(double) * n * (size_t)q->tq.n_states * SGK_ACTIONS;
  hipError_t e = hipMalloc(&q->tq.table, tbytes);
  if (e == hipSuccess) e = hipMalloc(&q->tq.tags, sizeof(uint64_t) * n);
  if (e == hipSuccess) e = hipMalloc(&q->actions, n);
  if (e == hipSuccess) e = hipMalloc(&q->t_dev, sizeof(long long));
  if (e == hipSuccess) e = hipMalloc(&q->tq.row_cache, sizeof(double) * 4 * n);
  if (e == hipSuccess) e = hipMalloc(&q->tq.hash_overflow, 2 * sizeof(int32_t));
  if (e == hipSuccess && hashed) e = hipMalloc(&q->tq.keys, sizeof(uint32_t) * n * (size_t)hash_capacity);

  if (e == hipSuccess) e = hipMemsetAsync(q->tq.table, 0, tbytes, env->stream);  // defaultdict(zeros) (value.py:31)
  if (e == hipSuccess) e = hipMemsetAsync(q->tq.tags, 0xff, sizeof(uint64_t) * n, env->stream);
  if (e == hipSuccess) e = hipMemsetAsync(q->tq.hash_overflow, 0, 2 * sizeof(int32_t), env->stream);
  if (e == hipSuccess && hashed) e = hipMemsetAsync(q->tq.keys, 0xff, sizeof(uint32_t) * n * (size_t)hash_capacity, env->stream);
  if (e != hipSuccess) {
    int rc = hip_fail(e, "tabular-Q allocation");
    const KeepError keep;
    sgk_tabq_destroy(q);
    keep.restore();
    return rc;
  }
  *out = q;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_tabq_create(sgk_env *env, double lr, double discount, double epsilon, int64_t epsilon_anneal, sgk_tabq **out) try {
  return sgk_tabq_create_ex(env, lr, discount, epsilon, epsilon_anneal, 0, out);
} SGK_CATCH_STATUS

int sgk_tabq_hash_info(sgk_tabq *q, int32_t *capacity_out, int32_t *max_used_out, int32_t *overflowed_out) try {
  if (!q) return fail(SGK_ERR_INVALID, "handle is NULL");
  SGK_CHECK_HANDLE(q->env);
  if (capacity_out) *capacity_out = q->tq.hash_cap;
  int32_t ov = 0;
  SGK_HIP(hipMemcpyAsync(&ov, q->tq.hash_overflow, sizeof(ov), hipMemcpyDeviceToHost, q->env->stream));
  SGK_HIP(sgk::host::wait_stream(q->env->stream));
  if (overflowed_out) *overflowed_out = ov;
  if (max_used_out) {
    *max_used_out = 0;
    if (q->tq.hash_cap) {  // the fullest agent's slot count, counted on the device (a diagnostic, not a hot path)
      int32_t *scratch = q->tq.hash_overflow + 1;
      SGK_HIP(hipMemsetAsync(scratch, 0, sizeof(int32_t), q->env->stream));
      SGK_HIP(sgk::launch_tabq_hash_used(q->env->sh, q->tq, scratch, q->env->stream));
      SGK_HIP(hipMemcpyAsync(max_used_out, scratch, sizeof(int32_t), hipMemcpyDeviceToHost, q->env->stream));
      SGK_HIP(sgk::host::wait_stream(q->env->stream));
    }
  }
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_tabq_copy_keys(sgk_tabq *q, int64_t env_begin, int64_t env_count, uint32_t *keys_host) try {
  if (!q) return fail(SGK_ERR_INVALID, "handle is NULL");
  SGK_CHECK_HANDLE(q->env);
  if (!q->tq.hash_cap) return fail(SGK_ERR_INVALID, "this level's tables are indexed by a perfect hash: there are no keys");
  if (!keys_host || env_begin < 0 || env_count < 0 || env_begin + env_count > q->env->sh.n) return fail(SGK_ERR_INVALID, "bad range");
  const size_t cap = (size_t)q->tq.hash_cap;
  SGK_HIP(hipMemcpyAsync(keys_host, q->tq.keys + (size_t)env_begin * cap, sizeof(uint32_t) * cap * (size_t)env_count,
                         hipMemcpyDeviceToHost, q->env->stream));
  SGK_HIP(sgk::host::wait_stream(q->env->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

// the per-step kernels' row slots (sgk_tabq.hip) mirror table rows: after anything else wrote the table they are re-tagged invalid
static hipError_t refresh_row_tags(sgk_tabq *q) {
  if (!q->rows_stale) return hipSuccess;
  q->rows_stale = false;
  return sgk::launch_tabq_forget_rows(q->env->sh, q->tq, q->env->stream);
}

int sgk_tabq_act(sgk_tabq *q, int explore, uint8_t *actions_out_dev) try {
  if (!q) return fail(SGK_ERR_INVALID, "handle is NULL");
  SGK_CHECK_HANDLE(q->env);
  if (!actions_out_dev) return fail(SGK_ERR_INVALID, "actions_out_dev is NULL");
  SGK_HIP(refresh_row_tags(q));
  SGK_HIP(sgk::launch_tabq_act(q->env->sh, q->tq, explore, actions_out_dev, q->env->stream));
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_tabq_learn(sgk_tabq *q, const uint8_t *actions_dev, int cheat) try {
  if (!q) return fail(SGK_ERR_INVALID, "handle is NULL");
  SGK_CHECK_HANDLE(q->env);
  if (!actions_dev) return fail(SGK_ERR_INVALID, "actions_dev is NULL");
  SGK_HIP(refresh_row_tags(q));
  SGK_HIP(sgk::launch_tabq_learn(q->env->sh, q->tq, actions_dev, cheat, q->env->stream));
  q->tq.t_agent += 1;  // update_epsilon(), learn.py:82
  q->t_dev_stale = true;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_tabq_step(sgk_tabq *q, int cheat, uint32_t flags, uint8_t *actions_out_dev) try {
  if (!q) return fail(SGK_ERR_INVALID, "handle is NULL");
  sgk_env *h = q->env;
  SGK_CHECK_HANDLE(h);
  if (flags & ~(uint32_t)SGK_F_NO_BOARDS) return fail(SGK_ERR_INVALID, "only SGK_F_NO_BOARDS is meaningful here");
  SGK_HIP(refresh_row_tags(q));
  SGK_HIP(sgk::launch_tabq_step(h->sh, q->tq, cheat, flags, actions_out_dev, h->stream));
  q->tq.t_agent += 1;  // update_epsilon(), learn.py:82
  q->t_dev_stale = true;
  h->sh.lockstep_t += 1;
  h->t_dev_stale = true;
  h->steps_issued += h->sh.n;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_tabq_learn_steps(sgk_tabq *q, int32_t n_steps, int cheat, uint32_t flags) try {
  if (!q) return fail(SGK_ERR_INVALID, "handle is NULL");
  sgk_env *h = q->env;
  SGK_CHECK_HANDLE(h);
  if (n_steps < 0) return fail(SGK_ERR_INVALID, "n_steps < 0");
  if (flags & ~(uint32_t)(SGK_F_NO_BOARDS | SGK_F_SEPARATE_LAUNCHES))
    return fail(SGK_ERR_INVALID, "only SGK_F_NO_BOARDS and SGK_F_SEPARATE_LAUNCHES are meaningful here");
  if (n_steps == 0) return SGK_OK;
  sgk::Shard &s = h->sh;
  const bool separate = (flags & SGK_F_SEPARATE_LAUNCHES) != 0;
  const uint32_t kflags = flags & SGK_F_NO_BOARDS;
  // tabq_learn's loop body (reference learn.py:61-85 inside train.py:62-70) -- act_explore, env.step, learn (+ update_epsilon),
  // reset of the finished envs -- as ONE launch per lockstep step (tabq_step_kernel), or as the four launches of the drop-in
  // call sequence (SGK_F_SEPARATE_LAUNCHES), captured ONCE per (n_steps, cheat, flags) and replayed: the agent step counter lives
  // in device memory, so a replay needs no new arguments.
  if (q->graphs_seed != s.seed) {  // env.seed() re-keyed the exploration draws: the recorded kernel arguments are stale
    (void)sgk::host::wait_stream(h->stream);
    q->graphs.clear();
    q->graphs_seed = s.seed;
  }
  auto key = std::make_pair(n_steps, (uint32_t)(cheat ? 1u : 0u) | (flags << 1));
  hipGraphExec_t exec = q->graphs.find(key);
  if (!exec) {
    sgk::TabqShard tq = q->tq;
    tq.t_ptr = q->t_dev;
    int rc = sgk::host::capture_graph(h->own_stream, "capture the tabular-Q step sequence", [&](hipStream_t cap) {
      hipError_t le = hipSuccess;
      for (int32_t k = 0; k < n_steps && le == hipSuccess; ++k) {
        tq.t_agent = k;  // offset from *t_dev
        if (!separate) {
          le = sgk::launch_tabq_step(s, tq, cheat, kflags, q->actions, cap);
          continue;
        }
        le = sgk::launch_tabq_act(s, tq, 1, q->actions, cap);
        if (le == hipSuccess) le = sgk::launch_step(s, q->actions, kflags, cap);
        if (le == hipSuccess) le = sgk::launch_tabq_learn(s, tq, q->actions, cheat, cap);
        if (le == hipSuccess) le = sgk::launch_reset(s, nullptr, 1 | (kflags ? 4 : 0), cap);
      }
      if (le == hipSuccess) {
        (void)hipGetLastError();
        hipLaunchKernelGGL(add_counter_kernel, dim3(1), dim3(1), 0, cap, reinterpret_cast<uint64_t *>(q->t_dev), (uint64_t)n_steps);
        le = hipGetLastError();
      }
      return le;
    }, &exec);
    if (rc != SGK_OK) return rc;
    q->graphs.insert(key, exec, h->stream);
  }
  SGK_HIP(refresh_row_tags(q));
  if (q->t_dev_stale) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(set_counter_kernel, dim3(1), dim3(1), 0, h->stream, reinterpret_cast<uint64_t *>(q->t_dev), (uint64_t)q->tq.t_agent);
    SGK_HIP(hipGetLastError());
    q->t_dev_stale = false;
  }
  SGK_HIP(hipGraphLaunch(exec, h->stream));
  q->tq.t_agent += n_steps;
  s.lockstep_t += (uint64_t)n_steps;
  h->t_dev_stale = true;
  h->steps_issued += s.n * (int64_t)n_steps;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_tabq_rollout_ex(sgk_tabq *q, int64_t n_steps, int cheat, int kernel) try {
  if (!q) return fail(SGK_ERR_INVALID, "handle is NULL");
  sgk_env *h = q->env;
  SGK_CHECK_HANDLE(h);
  if (n_steps < 0) return fail(SGK_ERR_INVALID, "n_steps < 0");
  if (kernel < SGK_TABQ_KERNEL_AUTO || kernel > SGK_TABQ_KERNEL_HBM) return fail(SGK_ERR_INVALID, "unknown kernel choice");
  if (n_steps == 0) return SGK_OK;
  sgk::Shard &s = h->sh;
  const size_t lds_need = sgk::tabq_rollout_lds_bytes(s);
  const bool lds_possible = lds_need != 0 && lds_need <= 160u * 1024u;
  if (kernel == SGK_TABQ_KERNEL_LDS && !lds_possible)
    return fail(SGK_ERR_INVALID, "this env's tables do not fit LDS: the LDS-resident kernel cannot run");
  // AUTO = the LDS-resident kernel wherever the tables fit: since round 4 (Q image only in LDS, transition table in registers) it is
  // ahead of the HBM-resident one at every size measured -- IslandNavigation 0.40 / 0.44 / 1.70 / 6.6 us per step at 16 K / 64 K /
  // 256 K / 1 M agents against 1.24 / 1.38 / 9.65 / 40.6, DistributionalShift 0.40 / 0.91 / 3.54 / 13.8 against 1.22 / 1.42 / 9.44 /
  // 39.6 (profiles/r04/bench_tabq_sizes_{lds,hbm}.log); the mid-size exception of rounds 1-3 is gone.
  bool use_lds = lds_possible;
  if (kernel == SGK_TABQ_KERNEL_HBM) use_lds = false;
  if (use_lds) {
    SGK_HIP(sgk::launch_tabq_rollout(s, q->tq, n_steps, cheat, h->stream));
  } else {
    // table too large for LDS residency (Sokoban: n_cells^2 states), or the mid-size case above: the same loop with the
    // rows read and written in HBM
    SGK_HIP(sgk::launch_tabq_rollout_hbm(s, q->tq, n_steps, cheat, h->stream));
  }
  SGK_HIP(sgk::launch_reset(s, nullptr, 2, h->stream));  // materialise the boards of the final states
  q->tq.t_agent += n_steps;
  q->t_dev_stale = true;
  q->rows_stale = true;  // the rollout kernels write the table directly
  s.lockstep_t += (uint64_t)n_steps;
  h->t_dev_stale = true;
  h->steps_issued += s.n * n_steps;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_tabq_rollout(sgk_tabq *q, int64_t n_steps, int cheat) try { return sgk_tabq_rollout_ex(q, n_steps, cheat, SGK_TABQ_KERNEL_AUTO); } SGK_CATCH_STATUS

int sgk_tabq_table_dev(sgk_tabq *q, double **table_dev, int64_t *n_states, int64_t *n_actions) try {
  if (!q) return fail(SGK_ERR_INVALID, "handle is NULL");
  q->rows_stale = true;  // the caller may write through the pointer: the per-step kernels re-read the table afterwards
  if (table_dev) *table_dev = q->tq.table;
  if (n_states) *n_states = q->tq.n_states;
  if (n_actions) *n_actions = SGK_ACTIONS;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_tabq_invalidate_rows(sgk_tabq *q) try {
  if (!q) return fail(SGK_ERR_INVALID, "handle is NULL");
  q->rows_stale = true;  // the next sgk_tabq_act / _learn / _learn_steps re-tags every row slot invalid first
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_debug_server_stale_exit_word(sgk_env *h) try {
  SGK_TEST_HOOK_ONLY();
  if (!h) return fail(SGK_ERR_INVALID, "handle is NULL");
  if (!h->srv.mb || !h->srv.running) return fail(SGK_ERR_INVALID, "no resident step server on this handle");
  sgk::host::mb_store(&h->srv.mb->exited, h->srv.seq + 1u);  // what a server that served up to srv.seq writes when it leaves
  h->srv.launched += 1;                                      // (the link counts the words it is owed: this one has a server of its own)
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_debug_fail_host_alloc(int k) {
  if (!test_hooks_on) return SGK_ERR_INVALID;  // (never armed in a process that did not ask for the hooks)
  return sgk::host::alloc_countdown().exchange(k < 0 ? 0 : k);
}

int sgk_debug_graph_count(const sgk_env *h, const sgk_tabq *q, int32_t *env_graphs_out, int32_t *tabq_graphs_out) try {
  SGK_TEST_HOOK_ONLY();
  if (env_graphs_out) *env_graphs_out = h ? (int32_t)h->graphs.size() : 0;
  if (tabq_graphs_out) *tabq_graphs_out = q ? (int32_t)q->graphs.size() : 0;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_tabq_copy_table(sgk_tabq *q, int64_t env_begin, int64_t env_count, double *table_host) try {
  if (!q) return fail(SGK_ERR_INVALID, "handle is NULL");
  SGK_CHECK_HANDLE(q->env);
  if (!table_host || env_begin < 0 || env_count < 0 || env_begin + env_count > q->env->sh.n)
    return fail(SGK_ERR_INVALID, "bad range");
  // The table in HBM is state-major ([n_states][n][4], sgk_tabq.hip: row_of); the caller gets it agent by agent
  // ([env_count][n_states][4]). A block of agents at a time: one strided copy of their columns of every state's plane into a
  // pinned staging block of the handle's (8 MiB at most), transposed on the host.
  const size_t ns = (size_t)q->tq.n_states, n = (size_t)q->env->sh.n, row_bytes = sizeof(double) * SGK_ACTIONS;
  constexpr size_t STAGE_BYTES = (size_t)8 << 20;
  const size_t e_max = std::max<size_t>(1, STAGE_BYTES / (ns * row_bytes));
  if (!q->copy_stage) SGK_HIP(hipHostMalloc((void **)&q->copy_stage, std::max(STAGE_BYTES, ns * row_bytes), hipHostMallocDefault));
  // (a plane is n x 32 bytes: from 2^26 agents on that is a source pitch of 2 GiB and more, beyond what a 2-D copy may take
  // (hipDeviceAttributeMaxPitch) -- the planes' pieces are then copied one by one)
  int max_pitch = 0;
  if (hipDeviceGetAttribute(&max_pitch, hipDeviceAttributeMaxPitch, q->env->sh.device) != hipSuccess) {
    (void)hipGetLastError();
    max_pitch = 0;
  }
  const bool two_d = max_pitch > 0 && n * row_bytes <= (size_t)max_pitch;
  for (size_t e0 = 0; e0 < (size_t)env_count; e0 += e_max) {
    const size_t cnt = std::min(e_max, (size_t)env_count - e0);
    const double *src = q->tq.table + ((size_t)env_begin + e0) * SGK_ACTIONS;
    if (two_d) {
      SGK_HIP(hipMemcpy2DAsync(q->copy_stage, cnt * row_bytes, src, n * row_bytes, cnt * row_bytes, ns, hipMemcpyDeviceToHost, q->env->stream));
    } else {
      for (size_t st = 0; st < ns; ++st)
        SGK_HIP(hipMemcpyAsync(q->copy_stage + st * cnt * SGK_ACTIONS, src + st * n * SGK_ACTIONS, cnt * row_bytes, hipMemcpyDeviceToHost,
                               q->env->stream));
    }
    SGK_HIP(sgk::host::wait_stream(q->env->stream));
    for (size_t s = 0; s < ns; ++s)
      for (size_t e = 0; e < cnt; ++e)
        memcpy(table_host + ((e0 + e) * ns + s) * SGK_ACTIONS, q->copy_stage + (s * cnt + e) * SGK_ACTIONS, row_bytes);
  }
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_tabq_global_step(const sgk_tabq *q, int64_t *t_out) try {
  if (!q || !t_out) return fail(SGK_ERR_INVALID, "NULL argument");
  *t_out = q->tq.t_agent;
  return SGK_OK;
} SGK_CATCH_STATUS

}  // extern "C"
