// sgk_step.hip -- gfx950 (MI355X / CDNA4) kernels of the env side of the path: env.step / env.reset for N grid instances at
// once (reference learn.py:38,69; train.py:64; warmup.py:17-20), RandomAgent.act fused into the step (dummy.py:15-16), the
// K-step fused random rollout, the episode bookkeeping track_metrics reads (meters.py:66-84), the float32 observation
// cast, render("rgb_array") and the done-mask compaction. HBM-bound integer work: no MFMA here.
#include <type_traits>

#include "sgk_device.h"
#include "sgk_step_store.h"

namespace sgk {

// The wave-private tile writer of this file's kernels: the tile image in LDS (WaveTileLds, sgk_device.h)
constexpr int STREAM_MIN_WAVES = 4;  // __launch_bounds__ of the streamed rollout (resident waves per SIMD the register budget must
                                     // allow): 6 and 8 spill and lose (profiles/r03/stream_grid_x_occupancy.log)
#define SGK_TILE_DECLARE(ENV, NC, ON)                                                 \
  __shared__ __attribute__((aligned(16))) uint8_t tile_images[WG / 64][64 * (NC)];    \
  WaveTileLds<ENV, NC> W;                                                             \
  W.bind(tile_images[wave_index()]);                                                  \
  if (ON) stage_rotations(C, R)
#define SGK_TILE_WRITE(INFO, DST) W.write(C, R, (INFO), (DST))

// slab -> one metrics vector (sums over slots for [0..7], maxima for [8..11]); one workgroup of 1024 lanes:
// 16 columns x 64 slot-lanes, four independent loads in flight per lane, LDS tree over the slot-lanes
__global__ __launch_bounds__(1024) void metrics_reduce_kernel(const long long *__restrict__ slab, long long *__restrict__ out,
                                                              long long *__restrict__ out_host) {
  __shared__ long long part[1024];
  const int col = threadIdx.x & 15, lane_slot = threadIdx.x >> 4;  // 64 slot-lanes
  const bool is_max = col >= SGK_M_MAX_RETURN && col <= SGK_M_MAX_MARGIN_POS;
  long long acc = is_max ? LLONG_MIN : 0;
  for (int sl = lane_slot; sl < SGK_METRIC_SLOTS; sl += 256) {
    long long v0 = slab[(size_t)sl * SGK_METRICS_LEN + col];
    long long v1 = slab[(size_t)(sl + 64) * SGK_METRICS_LEN + col];
    long long v2 = slab[(size_t)(sl + 128) * SGK_METRICS_LEN + col];
    long long v3 = slab[(size_t)(sl + 192) * SGK_METRICS_LEN + col];
    acc = is_max ? max(max(acc, v0), max(max(v1, v2), v3)) : acc + ((v0 + v1) + (v2 + v3));
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int half = 32; half >= 1; half >>= 1) {
    if (lane_slot < half) {
      long long o = part[(lane_slot + half) * 16 + col];
      part[threadIdx.x] = is_max ? max(part[threadIdx.x], o) : part[threadIdx.x] + o;
    }
    __syncthreads();
  }
  if (threadIdx.x < 16) {
    out[threadIdx.x] = part[threadIdx.x];
    if (out_host) out_host[threadIdx.x] = part[threadIdx.x];  // pinned, device-mapped host memory: no copy command afterwards
  }
}

__global__ __launch_bounds__(WG) void metrics_init_kernel(long long *__restrict__ slab) {
  for (int i = blockIdx.x * WG + threadIdx.x; i < SGK_METRIC_SLOTS * SGK_METRICS_LEN; i += gridDim.x * WG) {
    int col = i & 15;
    slab[i] = (col >= SGK_M_MAX_RETURN && col <= SGK_M_MAX_MARGIN_POS) ? LLONG_MIN : 0;
  }
}

// ------------------------------------------------------------------------------------------------
// the lockstep step kernel: env.step(action) for every env of the shard. One lane = one env, one wave = one 64-env tile
// (grid-stride over tiles); the COMPACT board tile is assembled and stored by the wave itself (WaveTileLds: no workgroup
// barrier after the rule tables are staged).
// ------------------------------------------------------------------------------------------------
// SMALL (batches of up to STEP_SMALL_MAX_ENVS envs: the launch's whole working set stays in the L2s): one wave per workgroup, four
// times the workgroups, and plain stores for the step record and the board tile -- write-through (sc1) pays at a million envs,
// where board bytes would evict the state words (sgk_device.h), and costs here. Measured on this kernel's body, BoatRace, us per
// launch (tools/exp_step_variants.hip, profiles/r05/step_variants_*.log): 1 024 envs 2.46 -> 2.10, 65 536 envs 2.65 -> 2.54,
// 262 144 envs 3.86 -> 4.50 (hence the size cut).
constexpr int64_t STEP_SMALL_MAX_ENVS = 65536;

template <int ENV, int LAYOUT, bool RANDOM, bool SMALL, bool STORE = false>
__global__ __launch_bounds__(SMALL ? 64 : WG) void step_kernel(StepArgs a, StepStore st) {
  constexpr int WGT = SMALL ? 64 : WG;
  constexpr int NC = Geom<ENV>::NC;
  constexpr bool COMPACT = (LAYOUT == SGK_LAYOUT_COMPACT);
  // wave-private images: the rule table and the board tile. No workgroup barrier in this kernel (WaveRulesLoad, sgk_device.h).
  __shared__ WaveRulesImage rules_images[WGT / 64];
  __shared__ __attribute__((aligned(16))) uint8_t tile_images[COMPACT ? WGT / 64 : 1][COMPACT ? 64 * NC : 16];
  __shared__ int episode_words[WGT / 64][16];
  const int lane = threadIdx.x & 63, wave = wave_index();
  const int64_t n_wt = (a.n + 63) / 64;
  const int64_t wt0 = (int64_t)blockIdx.x * (WGT / 64) + wave, wstride = (int64_t)gridDim.x * (WGT / 64);
  const bool boards_on = !(a.flags & SGK_F_NO_BOARDS);
  // Every load of the wave's first tile is requested before the first wait -- ONE memory round trip ahead of the arithmetic:
  // all kernel arguments (the compiler otherwise fetches some of them where they are first used, behind the vector loads: a
  // second round trip), the lockstep counter of a graph replay (branch-free: a null t_ptr reads a word that exists and drops
  // it), the state words (and actions), the wave's pieces of the rule table and of the blank board tile.
  keep_in_sgprs(a.rec, a.boards, a.last_return, a.last_perf);
  keep_in_sgprs(a.n_episodes, a.n_resets, a.metrics, a.aux);
  keep_in_sgprs(a.seed, a.env_base, a.t, a.flags);
  const uint64_t *t_word = a.t_ptr ? a.t_ptr : reinterpret_cast<const uint64_t *>(a.rules);
  const uint64_t t_base = *t_word;  // (wave-uniform address: a scalar load)
  const long long st_slice = STORE ? store_slice(st) : 0;  // (its scalar load goes out with the others)
  uint64_t w_cur = 0;
  uint8_t a_cur = 0;
  {
    // unconditional, index clamped into the batch: no branch between the entry block's loads (a wave past the last tile drops the word)
    const int64_t e0 = wt0 * 64 + lane;
    const int64_t e0c = e0 < a.n ? e0 : a.n - 1;
    w_cur = a.state[e0c];
    if (!RANDOM) a_cur = a.actions[e0c];
  }
  WaveRulesLoad rules_load;
  rules_load.request(a.rules);
  WaveTileLds<ENV, NC> W;
  W.bind(tile_images[COMPACT ? wave : 0]);
  typename WaveTileLds<ENV, NC>::Blank blank;
  if (COMPACT) W.request_blank(blank, a.rules);  // (also when the caller wants no boards: a branch here would split the batch of loads)
  rules_load.commit(rules_images[wave]);
  if (COMPACT) W.blank_arrived(blank);  // (requested with the table's pieces, in with them: no load is left for a wait behind a store)
  const SgkRules &R = rules_images[wave].r;
  const uint64_t t_now = a.t + (a.t_ptr ? t_base : 0ull);
  WaveEpisodeLds episodes;  // the wave's episode metrics of this launch (sgk_device.h)
  episodes.bind(episode_words[wave]);
  // A wave that has worked its last tile must LEAVE without waiting for anything: on gfx950 loads and stores share one counter, so a
  // wait for the next tile's prefetched words is also a wait for this tile's stores to be acknowledged -- and the compiler, left to
  // itself, put that wait in front of the loop's exit test (a launch 0.1-0.45 us longer at every size: the wave's end and the
  // stores' trip to memory in series instead of side by side). Hence: no loop at all for SMALL (grid == tiles: one tile per wave),
  // and for the rest the prefetched words are taken over behind the exit and behind an opaque use.
  for (int64_t wt = wt0; wt < n_wt;) {
    const int64_t env = wt * 64 + lane;
    const bool valid = env < a.n;
    // this tile's state word was requested before the rule tables were staged / while the previous tile ran
    EnvState s = unpack_state(w_cur);
    const uint8_t act_cur = a_cur;
    const int64_t wt_next = wt + wstride;
    const bool more = !SMALL && wt_next < n_wt;  // wave-uniform
    uint64_t w_next = 0;
    uint32_t a_next = 0;
    if (more) {
      const int64_t ne = wt_next * 64 + lane;
      if (ne < a.n) {
        w_next = a.state[ne];
        if (!RANDOM) a_next = a.actions[ne];
      }
    }
    if (!valid) s = initial_state(R);
    load_episode_index<ENV>(s, a.n_resets, env, valid);
    int action = 0;
    if (RANDOM) {
      uint64_t ge = a.env_base + (uint64_t)env;
      uint32_t x[4];
      philox4x32_10((uint32_t)ge, (uint32_t)(ge >> 32), (uint32_t)(t_now >> 6), 0u, (uint32_t)a.seed,
                    (uint32_t)(a.seed >> 32), x);
      action = action_from_block(x, t_now);
    } else {
      action = act_cur & 3;
    }
    uint32_t rec;
    EpisodeAcc acc;  // this tile's: a lane finishes at most one episode per step
    acc_init(acc);
    step_one<ENV>(R, a, env, valid, action, s, rec, acc);
    episodes.add(acc.n_eps != 0, acc.s_ret, acc.s_perf);
    if (!more) episodes.flush(a.metrics);  // the wave's last tile: the metrics go out ahead of this tile's stores
    if (valid) {
      a.state[env] = pack_state(s);
      if (SMALL) a.rec[env] = rec;
      else __hip_atomic_store(&a.rec[env], rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dword sc1
      if (STORE) {  // contain.py:15-17's action / reward / terminal of this transition
        const int64_t at = (int64_t)st_slice * a.n + env;
        // (& 3: an executed "stay", action 4 under a non-default SGK_INTERRUPT_FORCED_ACTION, has no Q column: stored as UP)
        st.actions[at] = st.cheat ? (uint8_t)((rec >> 24) & 3u) : (uint8_t)(act_cur & 3);
        st.rewards[at] = st.cheat ? (int8_t)(rec >> 8) : (int8_t)rec;
        st.terminals[at] = (uint8_t)((rec >> 16) & 1u);
      }
    }
    if (boards_on || STORE) {
      if (COMPACT) {
        W.draw_from_blank(blank, R, sprite_info<ENV>(R, s));
        if (boards_on) {
          if (SMALL) W.template flush<0>(a.boards + wt * 64 * NC);  // the buffer is padded to whole tiles
          else W.flush(a.boards + wt * 64 * NC);
        }
        if (STORE) store_tile<WaveTileLds<ENV, NC>, NC>(W, st, st_slice, a.n, wt, lane);
      } else if (valid) {
        if (boards_on) write_board_pitched<ENV, Geom<ENV>::PITCH>(R, a.boards, env, s);
        if (STORE) write_row_bytes<ENV, NC>(R, st.boards + ((int64_t)st_slice * a.n + env) * NC, s);
      }
    }
    if (!more) break;
    asm volatile("" : "+v"(w_next), "+v"(a_next));  // (the wait for the prefetch belongs HERE, on the way to the next tile)
    w_cur = w_next;
    a_cur = (uint8_t)a_next;
    wt = wt_next;
  }
}

// ------------------------------------------------------------------------------------------------
// The single-env drop-in's step server (host-in-the-loop: reference learn.py:38,69 calls env.step once per agent decision).
// A launch per step costs the host ~22 us (launch + completion); here ONE wave stays resident and serves steps through a
// mailbox in pinned, device-mapped HOST memory: the host writes the action(s) and a request number, the wave polls the number
// over PCIe, steps its envs (state words in registers between steps, rule tables staged once), writes state word / step record /
// board tile to the host-visible buffers, fences to system scope and publishes the number. The server leaves by itself after
// SGK_SERVER_IDLE_US microseconds without work (a host that went away cannot strand it) and whenever the host asks (request ==
// SGK_SERVER_STOP): every other entry point stops it first, so the arrays in memory are always current outside sgk_step_host.
// ------------------------------------------------------------------------------------------------
template <int ENV, int LAYOUT>
__global__ __launch_bounds__(64) void env_server_kernel(StepArgs a, SgkMailbox *mb, uint32_t last) {
  constexpr int NC = Geom<ENV>::NC;
  constexpr bool COMPACT = (LAYOUT == SGK_LAYOUT_COMPACT);
  __shared__ SgkRules R;
  __shared__ CompactLds<NC> C;
  stage_rules(R, a.rules);
  __shared__ __attribute__((aligned(16))) uint8_t tile_image[64 * NC];
  WaveTileLds<ENV, NC> W;
  W.bind(tile_image);
  if (COMPACT) stage_rotations(C, R);
  const int lane = threadIdx.x;
  const int64_t env = lane;
  const bool valid = env < a.n;
  EnvState s = initial_state(R);
  if (valid) s = unpack_state(a.state[env]);
  load_episode_index<ENV>(s, a.n_resets, env, valid);
  EpisodeAcc acc;
  acc_init(acc);
  // What has been served is what the mailbox says, not what the launcher believes: `done` is written by the servers alone (after
  // every request, behind a system-scope release), and a server starts only after its predecessor has ended (stream order). A
  // server that was started for a request another one has already answered -- the host took a late-landing exit word for a fresh
  // one -- then finds nothing to do instead of taking the step a second time.
  {
    uint32_t served = last;
    if (lane == 0) served = __hip_atomic_load(&mb->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    last = (uint32_t)__builtin_amdgcn_readfirstlane((int)served);
  }
  unsigned long long idle_since = 0;  // wall_clock64() of the first poll without a request, 0 = busy
  for (;;) {
    // lane 0's system-scope load is the poll (one PCIe read); the word is made wave-uniform before it steers control flow
    uint64_t word = 0;
    if (lane == 0) word = __hip_atomic_load(&mb->request, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t req = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)word);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(word >> 32));
    if (req == last) {
      // Nothing asked yet. The server leaves after SGK_SERVER_IDLE_US of that, by the constant 100 MHz clock: a host loop that
      // comes back within microseconds (tabular-Q: ~13 us between an answer and the next request) keeps finding it; one that is
      // away for longer (a DeepQ agent's torch kernels, ~2 ms) finds the device free -- a device-wide synchronisation in the
      // caller's process waits for every resident kernel, this wave included, so the budget bounds what such a call can cost.
      if (idle_since == 0) idle_since = wall_clock64();
      else if (wall_clock64() - idle_since > (unsigned long long)SGK_SERVER_IDLE_US * 100ull) break;
      continue;
    }
    if (req == SGK_SERVER_STOP) break;
    idle_since = 0;
    const uint32_t flags = hi & 0xffu;
    int action = (int)((hi >> 8) & 3u);  // env 0's action rides in the request word
    if (a.n > 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: the other actions were written before the request word
      if (valid) action = (int)(__hip_atomic_load(&a.actions[env], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) & 3u);
    }
    if (flags & SGK_SRV_RESET) {
      // env.reset() for every env of the handle (reset_kernel's mode 0 without a mask): a new episode index for the envs that draw,
      // the initial state; the step record and the episode arrays stay as they are
      if (valid) {
        const int epi = s.epi + 1;
        bump_reset_count<ENV>(a.n_resets, env);
        s = initial_state(R);
        s.epi = epi;
        begin_episode<ENV>(R, s, a.seed, a.env_base + (uint64_t)env, aux_of<ENV>(a.aux, env));
        a.state[env] = pack_state(s);
      }
    } else {
      StepArgs b = a;
      b.flags = flags;
      uint32_t rec;
      step_one<ENV>(R, b, env, valid, action, s, rec, acc);
      if (valid) {
        a.state[env] = pack_state(s);
        a.rec[env] = rec;
      }
    }
    if (!(flags & SGK_F_NO_BOARDS)) {
      if (COMPACT) {  // only the rows of the envs there are: the destination is host memory, every byte crosses PCIe
        W.draw_all(C, R, sprite_info<ENV>(R, s));
        W.flush_prefix(a.boards, (int)a.n * NC);
      } else if (valid) {
        write_board_pitched<ENV, Geom<ENV>::PITCH>(R, a.boards, env, s);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the outputs are visible to the host before the number is
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(&mb->done, req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    last = req;
  }
  acc_flush(acc, a.metrics);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_store(&mb->exited, last + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // "served up to `last`"
}

hipError_t launch_env_server(const Shard &sh, const uint8_t *actions, SgkMailbox *mb, uint32_t last, hipStream_t st) {
  (void)hipGetLastError();
  StepArgs a = make_step_args(sh, actions, 0);
  SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout, env_server_kernel<E, L><<<dim3(1), dim3(64), 0, st>>>(a, mb, last));
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// fused random rollout: n_steps lockstep steps in one launch, the env state in registers for all of them.
//   STREAM = false  boards and the step record are materialised ONCE, after the last step (no per-step output exists);
//   STREAM = true   every step's outputs are MATERIALISED in HBM like a per-step launch's -- the board tile (wave-private
//                   COMPACT writer: streaming 16-byte stores, no barrier) and the step record --, either into the env's own
//                   buffers (each step overwrites the previous one's) or into a caller's trajectory ring
//                   boards [ring][n][NC] / recs [ring][n], step k going to slice (slice0 + k) % ring: the batched
//                   dqn_warmup (reference warmup.py:14-21 stores every random-action transition). What a per-step launch
//                   pays and this does not: the launch boundary, the rule staging, and the state word's round trip
//                   through HBM (8 B read + 8 B written per env-step).
// ------------------------------------------------------------------------------------------------
struct StreamOut {
  int8_t *boards;   // trajectory ring [ring][n][NC], or nullptr = the env's own board buffer
  uint32_t *recs;   // trajectory ring [ring][n], or nullptr = the env's own record buffer
  int32_t ring, slice0;
  int32_t tiles_ok;  // ring slices are 16-byte aligned (n * NC % 16 == 0): whole tiles go through the tile writer
  int32_t tile_major;  // rings laid out [n_tiles][ring][64][...] (SGK_F_RING_TILE_MAJOR): padded tiles, always whole
  int32_t ring_nt;     // board tiles into the ring as non-temporal write-through stores (rings far larger than the caches)
};

template <int ENV, int LAYOUT, bool STREAM>
__global__ __launch_bounds__(WG, STREAM ? STREAM_MIN_WAVES : 1) void rollout_random_kernel(StepArgs a, int32_t n_steps, StreamOut o) {
  constexpr int NC = Geom<ENV>::NC;
  constexpr bool COMPACT = (LAYOUT == SGK_LAYOUT_COMPACT);
  __shared__ SgkRules R;
  __shared__ CompactLds<NC> C;
  // the first tile's state words are asked for BEFORE the rules are staged: behind a saturated write path each of these is a
  // multi-microsecond round trip, and a workgroup's start-up (11 us median at 1 M envs, profiles/r03/stream_timeline.log) was a
  // chain of them
  uint64_t first_word = 0;
  {
    const int64_t env0 = ((int64_t)blockIdx.x * (WG / 64) + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 63);
    if (STREAM && env0 < a.n) first_word = a.state[env0];
  }
  stage_rules(R, a.rules);
  SGK_TILE_DECLARE(ENV, NC, (COMPACT || STREAM));
  // (the outputs-once form stores nothing per step: a scalar tile index buys it nothing and costs it registers -- 64 -> 79 VGPRs)
  const int lane = threadIdx.x & 63, wave = STREAM ? wave_index() : (int)(threadIdx.x >> 6);
  const bool boards_on = !(a.flags & SGK_F_NO_BOARDS);
  const int64_t n_wt = (a.n + 63) / 64;
  EpisodeAcc acc;
  acc_init(acc);
  // (Handing the tile groups out by ticket to a resident-sized grid of persistent workgroups -- the XCDs do not write at the same
  // rate: at 1 M envs the workgroups of the even XCDs live 190-215 us per 100 steps, those of the odd ones 150-165, and the odd
  // XCDs sit idle for the last 15 % of a launch, profiles/r03/stream_timeline.log -- was built and measured: BoatRace 6.25 vs 6.18
  // us per step, IslandNavigation 9.03 vs 9.5, Sokoban 8.1-8.25 vs 8.3, at 97 instead of 80 VGPRs; stream_tickets_ab.log. Not kept.)
  for (int64_t wt = (int64_t)blockIdx.x * (WG / 64) + wave; wt < n_wt; wt += (int64_t)gridDim.x * (WG / 64)) {
    const int64_t env = wt * 64 + lane;
    const bool valid = env < a.n;
    EnvState s = initial_state(R);
    if (valid) s = unpack_state((STREAM && wt < (int64_t)gridDim.x * (WG / 64)) ? first_word : a.state[env]);
    load_episode_index<ENV>(s, a.n_resets, env, valid);
    const uint64_t ge = a.env_base + (uint64_t)env;
    AuxRegs ax;  // the env's float64 side state for the whole launch (friend or foe; dead code elsewhere)
    ax.init();
    if (HasAux<ENV>::value && valid) ax.load(a.aux + env * SGK_AUX_DOUBLES);
    uint32_t x[4] = {0, 0, 0, 0};
    uint32_t rec = 0;
    // The loop is instruction-issue bound when nothing is streamed, so it is kept lean: the 2-bit actions are shifted out of
    // one 32-bit word of the Philox block (a new word every 16 steps, a new block every 64), the step record is packed once
    // after the loop (per step when streamed), and episode ends -- rare -- take a branch instead of predicated bookkeeping.
    uint32_t w = 0;
    int last_obs = 0, last_hid = 0, last_done = 0, last_action = 0;
    const bool auto_reset = (a.flags & SGK_F_AUTO_RESET) != 0;
    const bool whole_tile = wt * 64 + 64 <= a.n;  // wave-uniform
    int32_t slice = o.slice0;
    // the tile image lives in LDS for all n_steps: drawn once here, then only the cells a step changes are re-drawn
    uint32_t drawn = sprite_info<ENV>(R, s);
    if (STREAM && boards_on) W.draw_all(C, R, drawn);
    if (STREAM) {
    // Streamed: at 1 M envs the loop hides behind its stores, but a small batch (65 536 envs = ONE wave per SIMD, the per-GPU
    // share of an 8-GPU run two) is bound by the wave's own instruction latency, so the step carries as little as it can: every
    // destination is a pointer set up here and advanced by a wave-uniform stride (the slice / layout / ring arithmetic and its
    // branches used to run on every step), the action word is counted down instead of compared against the 64-bit step index.
    const bool tm = o.boards && o.tile_major;
    // the env's own buffer is padded; so are tile-major rings; slice-major rings take whole tiles where slices are 16-byte aligned
    const bool tiles = o.boards ? (tm || (o.tiles_ok && whole_tile)) : COMPACT;
    const bool flush_nt = o.boards && o.ring_nt;
    // this lane's record of the current slice (tile-major: the record of env e sits at row e % 64 = its lane) ...
    uint32_t *recs_p = !o.recs ? a.rec + env
                               : (o.tile_major ? o.recs + (wt * (int64_t)o.ring + slice) * 64 + lane : o.recs + (int64_t)slice * a.n + env);
    const int64_t rec_stride = !o.recs ? 0 : (o.tile_major ? 64 : a.n);  // elements per slice
    // ... and the board base of the current slice, wave-uniform, rows of NC bytes (tile-major: `dense + wt * 64 * NC` is this
    // tile's slot of the slice)
    int8_t *dense = !o.boards ? a.boards
                              : (tm ? o.boards + ((wt * (int64_t)o.ring + slice) - wt) * 64 * NC : o.boards + (int64_t)slice * a.n * NC);
    const int64_t brd_stride = !o.boards ? 0 : (tm ? (int64_t)64 * NC : a.n * (int64_t)NC);  // bytes per slice
    const int64_t rec_wrap = rec_stride * (o.ring - 1), brd_wrap = brd_stride * (o.ring - 1);
    uint32_t left = 0;  // steps left in the action word in hand
#pragma nounroll
    for (int32_t k = 0; k < n_steps; ++k) {
      if (left == 0) {  // a new 16-step word; a new Philox block every 64 steps (and on the launch's first step)
        const uint64_t t = a.t + (uint64_t)k;
        const uint32_t tl = (uint32_t)t;
        if (k == 0 || (tl & 63u) == 0)
          philox4x32_10((uint32_t)ge, (uint32_t)(ge >> 32), (uint32_t)(t >> 6), 0u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), x);
        const uint32_t j = (tl >> 4) & 3u;
        w = (j == 0 ? x[0] : (j == 1 ? x[1] : (j == 2 ? x[2] : x[3]))) >> (2 * (tl & 15u));
        left = 16 - (tl & 15u);
      }
      --left;
      int action = (int)(w & 3u);
      w >>= 2;
      const bool live = valid && !s.over;
      if (live) action = env_actual_action<ENV>(R, s, a.seed, ge, action);
      last_action = action;
      last_obs = 0;
      last_hid = 0;
      last_done = valid ? 1 : 0;
      if (live) {
        int r_obs, r_hid, term;
        if (HasAux<ENV>::value) transition_with<ENV>(R, s, action, r_obs, r_hid, term, ax);  // side state in registers
        else transition<ENV>(R, s, action, r_obs, r_hid, term);
        s.frame += 1;
        s.ret += r_obs;
        s.hid += r_hid;
        last_obs = r_obs;
        last_hid = r_hid;
        last_done = 0;
        if (term || s.frame >= R.max_iterations) {
          last_done = 1;
          acc_add(acc, true, s.ret, s.hid);
          // (the index goes through an opaque copy: otherwise the three element addresses are hoisted out of the step loop and
          // live in six VGPRs for all of it -- episode ends are rare, the registers decide the occupancy)
          int64_t env_here = env;
          asm volatile("" : "+v"(env_here));
          a.last_return[env_here] = s.ret;
          a.last_perf[env_here] = s.hid;
          bump_episode_count(a.n_episodes, env_here);
          if (auto_reset) {
            const int epi = s.epi + 1;  // this reset's index; n_resets[env] is brought up to date once, after the loop
            s = initial_state(R);
            s.epi = epi;
            if (HasAux<ENV>::value) begin_episode_with<ENV>(R, s, a.seed, ge, ax);
            else begin_episode<ENV>(R, s, a.seed, ge);
          } else {
            s.over = 1;
          }
        }
      }
      // this step's outputs, as env.step returns them: the record (reward, hidden reward, done, executed action) ...
      const uint32_t rk = pack_rec(last_obs, last_hid, last_done, last_action);
      if (valid) *recs_p = rk;  // (plain stores: write-through dwords cost a fabric write each; non-temporal ones measure the same as
                                // plain at 131 072 and at 1 M envs, round 5)
      // ... and the successor board: one tile store per wave wherever the destination takes whole tiles
      if (boards_on) {
        const uint32_t now = sprite_info<ENV>(R, s);
        W.update(R, drawn, now);
        drawn = now;
        if (tiles) {
          if (flush_nt) W.template flush<RING_STORE_AUX>(dense + wt * 64 * NC);
          else W.flush(dense + wt * 64 * NC);
        }
        else if (o.boards) { if (valid) write_row_bytes<ENV, NC>(R, dense + env * NC, s); }
        else if (valid) write_board_pitched<ENV, Geom<ENV>::PITCH>(R, a.boards, env, s);
      }
      if (++slice == o.ring) {
        slice = 0;
        recs_p -= rec_wrap;
        dense -= brd_wrap;
      } else {
        recs_p += rec_stride;
        dense += brd_stride;
      }
    }
    // the last step's board also into the env's own buffer: it shows the final state without a re-render launch
    if (boards_on && o.boards && tiles && COMPACT && n_steps > 0) W.flush(a.boards + wt * 64 * NC);
    } else {
      // Outputs once: nothing is stored per step, so the loop is bound by instruction ISSUE -- and it was the SCALAR port that
      // was full (profiles/r03/pmc_sq_rollout_boatrace_fused.json: 15.7 SALU + 5.3 branch instructions per wave-step against
      // 20.5 VALU, the scalar issue slot of a SIMD busy ~90 % of the launch): 64-bit step-index arithmetic and word / block
      // comparisons on every step, and three exec-mask instructions per `if`. Here the loop nest follows the action stream's own
      // structure -- a Philox block of 64 steps, its four words of 16 steps, the steps of a word -- so that the step loop carries
      // one 32-bit counter, and a wave in which every lane steps on every step (whole tile, nobody's episode over, auto-reset)
      // runs the LEAN body: no liveness test per step.
      auto one_step = [&](int action, auto check) {
        constexpr bool CHECK = decltype(check)::value;
        const bool live = CHECK ? (valid && !s.over) : true;
        if (live) action = env_actual_action<ENV>(R, s, a.seed, ge, action);
        last_action = action;
        if (live) {
          int r_obs, r_hid, term;
          if (HasAux<ENV>::value) transition_with<ENV>(R, s, action, r_obs, r_hid, term, ax);  // side state in registers
          else transition<ENV>(R, s, action, r_obs, r_hid, term);
          s.frame += 1;
          s.ret += r_obs;
          s.hid += r_hid;
          last_obs = r_obs;
          last_hid = r_hid;
          last_done = 0;
          if (term || s.frame >= R.max_iterations) {
            last_done = 1;
            acc_add(acc, true, s.ret, s.hid);
            a.last_return[env] = s.ret;
            a.last_perf[env] = s.hid;
            bump_episode_count(a.n_episodes, env);
            if (auto_reset) {
              const int epi = s.epi + 1;  // this reset's index; n_resets[env] is brought up to date once, after the loop
              s = initial_state(R);
              s.epi = epi;
              if (HasAux<ENV>::value) begin_episode_with<ENV>(R, s, a.seed, ge, ax);
              else begin_episode<ENV>(R, s, a.seed, ge);
            } else {
              s.over = 1;
            }
          }
        } else {
          last_obs = 0;
          last_hid = 0;
          last_done = valid ? 1 : 0;
        }
      };
      const bool lean = auto_reset && whole_tile && (__ballot(s.over != 0) == 0ull);  // wave-uniform
      uint64_t t = a.t;
      int32_t k = 0;
      bool have_block = false;
#pragma nounroll
      while (k < n_steps) {
        const uint32_t tl = (uint32_t)t;
        if (!have_block || (tl & 63u) == 0u) {
          philox4x32_10((uint32_t)ge, (uint32_t)(ge >> 32), (uint32_t)(t >> 6), 0u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), x);
          have_block = true;
        }
        const uint32_t j = (tl >> 4) & 3u;
        w = (j == 0 ? x[0] : (j == 1 ? x[1] : (j == 2 ? x[2] : x[3]))) >> (2 * (tl & 15u));
        const int32_t m = min((int32_t)(16u - (tl & 15u)), n_steps - k);  // steps this word serves
        if (lean) {
#pragma nounroll
          for (int32_t i = 0; i < m; ++i) {
            one_step((int)(w & 3u), std::false_type{});
            w >>= 2;
          }
        } else {
#pragma nounroll
          for (int32_t i = 0; i < m; ++i) {
            one_step((int)(w & 3u), std::true_type{});
            w >>= 2;
          }
        }
        k += m;
        t += (uint64_t)m;
      }
    }
    rec = pack_rec(last_obs, last_hid, last_done, last_action);
    if (valid) {
      a.state[env] = pack_state(s);
      if (!STREAM || o.recs) a.rec[env] = rec;
      if (HasEnvDraws<ENV>::value) a.n_resets[env] = s.epi;  // this lane is the env's only writer
      if (HasAux<ENV>::value && ax.dirty) ax.store(a.aux + env * SGK_AUX_DOUBLES);
    }
    if (!STREAM && boards_on) {  // the env's own boards show the final state (streamed into a caller's ring: the launcher
                                 // re-materialises them afterwards, reset_kernel mode 2)
      if (COMPACT) SGK_TILE_WRITE(sprite_info<ENV>(R, s), a.boards + wt * 64 * NC);
      else if (valid) write_board_pitched<ENV, Geom<ENV>::PITCH>(R, a.boards, env, s);
    }
  }
  acc_flush(acc, a.metrics);
}

// ------------------------------------------------------------------------------------------------
// env.reset(): mode 0 = all envs (mask == nullptr) or masked envs; mode 1 = exactly the envs whose episode
// is over; mode 2 = no state change, only re-materialise the boards from the state words
// ------------------------------------------------------------------------------------------------
template <int ENV, int LAYOUT, bool STORE = false>
__global__ __launch_bounds__(WG) void reset_kernel(const SgkRules *rules, uint64_t *state, int8_t *boards,
                                                   const uint8_t *mask, int mode_flags, int64_t n, uint64_t seed, uint64_t env_base,
                                                   int32_t *__restrict__ n_resets, const double *__restrict__ aux, StepStore st) {
  reset_body<ENV, LAYOUT, STORE>(rules, state, boards, mask, mode_flags, n, seed, env_base, n_resets, aux, st, (int)blockIdx.x, (int)gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// float32 observation for the Q-network: int8 cells [N][pitch] -> float32 [N][NC] dense
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void obs_f32_kernel(const int8_t *__restrict__ boards, float *__restrict__ dst, int64_t n,
                                                     int nc, int pitch) {
  // one thread per 4 consecutive cells of one env (nc % 4 == 0 for 6x6 and 6x8; generic tail otherwise)
  const int q_per_env = (nc + 3) / 4;
  const int64_t total = n * q_per_env;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t env = i / q_per_env;
    int q = (int)(i - env * q_per_env);
    const int8_t *src = boards + env * pitch + q * 4;
    float *out = dst + env * nc + q * 4;
    if (q * 4 + 4 <= nc && ((nc & 3) == 0) && ((pitch & 3) == 0)) {
      uint32_t w = *reinterpret_cast<const uint32_t *>(src);
      float4 f = make_float4((float)(int8_t)(w & 0xff), (float)(int8_t)((w >> 8) & 0xff),
                             (float)(int8_t)((w >> 16) & 0xff), (float)(int8_t)(w >> 24));
      *reinterpret_cast<float4 *>(out) = f;
    } else {
      for (int k = 0; k < 4 && q * 4 + k < nc; ++k) out[k] = (float)src[k];
    }
  }
}

// render("rgb_array") for every env: int8 cells -> uint8 [N][3][H*W] through the level's value palette
// (reference eval.py:16,30,42 copies these frames; layout (3, H, W) per env)
__global__ __launch_bounds__(WG) void render_rgb_kernel(const SgkRules *__restrict__ rules, const int8_t *__restrict__ boards,
                                                        uint8_t *__restrict__ dst, int64_t n, int nc, int pitch) {
  __shared__ uint8_t pal[8][4];
  if (threadIdx.x < 32) (&pal[0][0])[threadIdx.x] = (&rules->palette[0][0])[threadIdx.x];
  __syncthreads();
  const bool hwc = rules->render_hwc != 0;  // frame layout (H, W, 3) instead of (3, H, W): an include/sgk_levels.h switch
  const int64_t total = n * nc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t env = i / nc;
    int c = (int)(i - env * nc);
    int v = boards[env * pitch + c] & 7;
    uint8_t *o = dst + env * 3 * nc + (hwc ? 3 * c : c);
    const int plane = hwc ? 1 : nc;
    o[0] = pal[v][0];
    o[plane] = pal[v][1];
    o[2 * plane] = pal[v][2];
  }
}

// gather dense [N][NC] int8 boards out of the pitched/compact buffer (for host copies)
__global__ __launch_bounds__(WG) void dense_boards_kernel(const int8_t *__restrict__ boards, int8_t *__restrict__ dst,
                                                          int64_t n, int nc, int pitch) {
  const int64_t total = n * nc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t env = i / nc;
    int c = (int)(i - env * nc);
    dst[i] = boards[env * pitch + c];
  }
}

// ------------------------------------------------------------------------------------------------
// done-mask compaction (deterministic, ascending env id):
//   pass 1: per-workgroup count of done lanes          (ballot + popcount)
//   pass 2: exclusive scan of the workgroup counts     (one workgroup, wave prefix sums)
//   pass 3: scatter with ballot/mbcnt ranks inside the wave and LDS wave offsets inside the workgroup
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_rank(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__global__ __launch_bounds__(WG) void finished_count_kernel(const uint32_t *__restrict__ rec, int32_t *__restrict__ wg_count,
                                                            int64_t n) {
  __shared__ int wave_n[WG / 64];
  const int64_t env = (int64_t)blockIdx.x * WG + threadIdx.x;
  bool done = env < n && ((rec[env] >> 16) & 1u);
  unsigned long long m = __ballot(done);
  if ((threadIdx.x & 63) == 0) wave_n[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) wg_count[blockIdx.x] = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3];
}

__global__ __launch_bounds__(1024) void finished_scan_kernel(const int32_t *__restrict__ wg_count, int64_t *__restrict__ wg_offset,
                                                             int64_t n_wg, int64_t *__restrict__ total) {
  // sequential over 1024-wide slabs; inside a slab: wave inclusive scan via shuffles, then wave totals via LDS
  __shared__ long long wave_tot[16];
  __shared__ long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n_wg; base += 1024) {
    int64_t i = base + threadIdx.x;
    long long v = (i < n_wg) ? (long long)wg_count[i] : 0;
    long long incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      long long o = __shfl_up(incl, off, 64);
      if ((int)(threadIdx.x & 63) >= off) incl += o;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    long long before = carry;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += wave_tot[w];
    if (i < n_wg) wg_offset[i] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(WG) void finished_scatter_kernel(const uint32_t *__restrict__ rec,
                                                              const int64_t *__restrict__ wg_offset,
                                                              const int32_t *__restrict__ last_return,
                                                              const int32_t *__restrict__ last_perf, int32_t *__restrict__ ids,
                                                              int32_t *__restrict__ ret, int32_t *__restrict__ perf, int64_t n) {
  __shared__ int wave_n[WG / 64];
  const int64_t env = (int64_t)blockIdx.x * WG + threadIdx.x;
  bool done = env < n && ((rec[env] >> 16) & 1u);
  unsigned long long m = __ballot(done);
  int rank = lane_rank(m);
  if ((threadIdx.x & 63) == 0) wave_n[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  int wave_off = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wave_off += wave_n[w];
  if (done) {
    int64_t o = wg_offset[blockIdx.x] + wave_off + rank;
    ids[o] = (int32_t)env;
    ret[o] = last_return[env];
    perf[o] = last_perf[env];
  }
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
// the step kernel reading the lockstep counter from device memory (hipGraph replays need no new arguments)
hipError_t launch_step_counter(const Shard &sh, const uint64_t *t_dev, uint64_t t_off, uint32_t flags, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  StepArgs a = make_step_args(sh, nullptr, flags);
  a.t = t_off;
  a.t_ptr = t_dev;
  if (sh.n <= STEP_SMALL_MAX_ENVS) {
    SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout, step_kernel<E, L, true, true><<<dim3((unsigned)((sh.n + 63) / 64)), dim3(64), 0, st>>>(a, StepStore{}));
  } else {
    int grid = grid_for((sh.n + WG - 1) / WG, sh.max_grid);
    SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout, step_kernel<E, L, true, false><<<dim3(grid), dim3(WG), 0, st>>>(a, StepStore{}));
  }
  return hipGetLastError();
}

hipError_t launch_step(const Shard &sh, const uint8_t *actions, uint32_t flags, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  StepArgs a = make_step_args(sh, actions, flags);
  const bool small = sh.n <= STEP_SMALL_MAX_ENVS;
  const dim3 grid(small ? (unsigned)((sh.n + 63) / 64) : (unsigned)grid_for((sh.n + WG - 1) / WG, sh.max_grid)), block(small ? 64 : WG);
  if (actions) {
    if (small) SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout, step_kernel<E, L, false, true><<<grid, block, 0, st>>>(a, StepStore{}));
    else SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout, step_kernel<E, L, false, false><<<grid, block, 0, st>>>(a, StepStore{}));
  } else {
    if (small) SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout, step_kernel<E, L, true, true><<<grid, block, 0, st>>>(a, StepStore{}));
    else SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout, step_kernel<E, L, true, false><<<grid, block, 0, st>>>(a, StepStore{}));
  }
  return hipGetLastError();
}


// env.step + ReplayBuffer.add's second half in one launch (sgk_step_store)
hipError_t launch_step_store(const Shard &sh, const uint8_t *actions, uint32_t flags, int cheat, int64_t slice, const long long *slice_dev,
                             int32_t ring, int8_t *successors, uint8_t *r_actions, int8_t *r_rewards, uint8_t *r_terminals, hipStream_t st) {
  (void)hipGetLastError();
  StepArgs a = make_step_args(sh, actions, flags);
  const StepStore so = make_store(sh, successors, r_actions, r_rewards, r_terminals, slice, slice_dev, ring, cheat);
  const bool small = sh.n <= STEP_SMALL_MAX_ENVS;
  const dim3 grid(small ? (unsigned)((sh.n + 63) / 64) : (unsigned)grid_for((sh.n + WG - 1) / WG, sh.max_grid)), block(small ? 64 : WG);
  if (small) SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout, step_kernel<E, L, false, true, true><<<grid, block, 0, st>>>(a, so));
  else SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout, step_kernel<E, L, false, false, true><<<grid, block, 0, st>>>(a, so));
  return hipGetLastError();
}

hipError_t launch_rollout_random(const Shard &sh, int32_t n_steps, uint32_t flags, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  StepArgs a = make_step_args(sh, nullptr, flags);
  int grid = grid_for((sh.n + WG - 1) / WG, sh.rollout_grid);
  StreamOut o{nullptr, nullptr, 1, 0, 0, 0, 0};
  SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout,
                          rollout_random_kernel<E, L, false><<<dim3(grid), dim3(WG), 0, st>>>(a, n_steps, o));
  return hipGetLastError();
}

// Board tiles into a trajectory ring as non-temporal write-through stores? Where the ring is larger than the Infinity Cache AND
// has enough slices that a launch does not come back to a slice soon. Measured (round 5, BoatRace, us per lockstep step, plain /
// non-temporal; profiles/r05/ring_nt_ab.log), by envs x slices (ring MB):
//   1 M x 8 (243) 4.09 / 4.79    1 M x 32 (973) 4.49 / 5.02    1 M x 48 (1460) 5.78 / 5.11    1 M x 64 (1946) 6.15 / 5.51    1 M x 100 (3041) 5.80 / 5.56
//   512 K x 16 (243) 2.26 / 2.48   512 K x 32 (487) 2.34 / 2.57   512 K x 64 (973) 3.06 / 2.58   512 K x 100 (1520) 3.15 / 2.56
//   262 K x 32 (243) 1.18 / 1.33   262 K x 64 (487) 1.48 / 1.37   262 K x 100 (760) 1.61 / 1.38
//   131 K x 64 (243) 0.61 / 0.64   131 K x 100 (380) 0.79 / 0.65   65 K x 100 (190) 0.48 / 0.49
// Rounds 3-4 switched at 1.5 GB (from the 1 M-env rows alone): the per-GPU shares of a 2- / 4- / 8-GPU run (512 K / 262 K / 131 K
// envs x 100 slices) were on the slow side of that rule by 18-22 %. SGK_RING_NT=0 / 1 (read at sgk_create) overrides the rule: the A/B knob.
static int32_t ring_stores_nt(const Shard &sh, int64_t ring_bytes, int32_t ring_slices) {
  if (sh.ring_nt_mode >= 0) return sh.ring_nt_mode;
  return ring_slices >= 48 && ring_bytes > (256ll << 20);
}

hipError_t launch_rollout_stream(const Shard &sh, int32_t n_steps, uint32_t flags, int8_t *boards_ring, uint32_t *recs_ring,
                                 int32_t ring, int32_t slice0, hipStream_t st) {
  (void)hipGetLastError();
  StepArgs a = make_step_args(sh, nullptr, flags);
  // every wave keeps its tile for all n_steps: more workgroups than the per-step kernel's cap (they stay resident longer,
  // and the stream of stores needs every CU busy), up to one 256-env workgroup per tile
  int grid = grid_for((sh.n + WG - 1) / WG, sh.stream_grid);
  StreamOut o{boards_ring, recs_ring, ring < 1 ? 1 : ring, slice0, (int32_t)(((sh.n * sh.n_cells) % 16) == 0 && ((uintptr_t)boards_ring % 16) == 0),
              (int32_t)((flags & SGK_F_RING_TILE_MAJOR) != 0),
              // non-temporal only where the ring dwarfs the caches. Measured at 1 M BoatRace envs, same box, three repeats: a
              // 100-slice ring (3 GB) 6.32 vs 6.52 us per step with / without, a 32-slice ring (1 GB) 5.34 vs 4.90
              ring_stores_nt(sh, (int64_t)(ring < 1 ? 1 : ring) * sh.n * (sh.n_cells + 4), ring)};
  // (Lowering the residency to whole rounds -- 16 workgroups per CU at 4 resident instead of 3 rounds of 5 and a last one of 1 --
  // by padding the dynamic LDS was measured on a fast ring: 5.17-5.31 us per step at 2, 3, 4 and 5 per CU alike,
  // profiles/r03/stream_residency_ab.log. Not kept.)
  SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout,
                          rollout_random_kernel<E, L, true><<<dim3(grid), dim3(WG), 0, st>>>(a, n_steps, o));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // the env's own boards show the final state: whole compact tiles were written by the kernel's last step; the other cases
  // (pitched rows, a partial last tile, unaligned slices) are re-rendered from the state words
  const bool in_kernel = sh.layout == SGK_LAYOUT_COMPACT && (o.tile_major || (o.tiles_ok && sh.n % 64 == 0));
  if (boards_ring && !(flags & SGK_F_NO_BOARDS) && !in_kernel) return launch_reset(sh, nullptr, 2, st);
  return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// How fast can THIS ring be written? The streamed rollout's stores and nothing else -- a wave per 64-env tile, every slice of
// the ring once per launch, 16-byte write-through buffer stores for the board tile and a dword per env for the record (zeros:
// the ring's contents are undefined afterwards). The rate a persistent kernel writes a multi-GB ring at is a property of the
// ALLOCATION: 4.6-4.9 us per step at 1 M BoatRace envs for some 3 GB hipMalloc blocks and 5.6-6.1 for others on one GPU in one
// process, stable for the block's lifetime, the same with a 16-fold spread of its slices, blind to every L2-side counter
// (profiles/r03/ring_alloc_*.log). A caller that keeps a trajectory ring for a whole run can therefore afford to allocate a few
// candidates, time each with this probe and keep the best (BatchedGridworldEnv.alloc_trajectory_ring).
// ------------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(WG) void ring_probe_kernel(int8_t *boards, uint32_t *recs, int64_t n, int32_t ring, int32_t tile_major,
                                                       int32_t nt) {
  constexpr int BYTES = 64 * NC, CHUNKS = 4 * NC, ITS = (CHUNKS + 63) / 64;
  const int lane = threadIdx.x & 63, wave = wave_index();
  const int64_t n_wt = n / 64;  // whole tiles: a partial last tile goes row by row in the real kernel, a rounding error here
  const sgk_u32x4 zero = {0u, 0u, 0u, 0u};
  for (int64_t wt = (int64_t)blockIdx.x * (WG / 64) + wave; wt < n_wt; wt += (int64_t)gridDim.x * (WG / 64)) {
    for (int32_t s = 0; s < ring; ++s) {
      if (boards) {
        int8_t *dst = tile_major ? boards + (wt * (int64_t)ring + s) * BYTES : boards + ((int64_t)s * n + wt * 64) * NC;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, BYTES, 0x00020000);
#pragma unroll
        for (int it = 0; it < ITS; ++it) {  // (a lane past the last chunk: dropped by the descriptor's range check)
          if (nt) __builtin_amdgcn_raw_buffer_store_b128(zero, rsrc, (lane + 64 * it) * 16, 0, RING_STORE_AUX);
          else __builtin_amdgcn_raw_buffer_store_b128(zero, rsrc, (lane + 64 * it) * 16, 0, BOARD_STORE_AUX);
        }
      }
      if (recs) {
        uint32_t *r = tile_major ? recs + (wt * (int64_t)ring + s) * 64 + lane : recs + (int64_t)s * n + wt * 64 + lane;
        *r = 0u;
      }
    }
  }
}

hipError_t launch_ring_probe(const Shard &sh, int8_t *boards_ring, uint32_t *recs_ring, int32_t ring, uint32_t flags, hipStream_t st) {
  (void)hipGetLastError();
  int grid = grid_for((sh.n + WG - 1) / WG, sh.stream_grid);
  const int32_t tm = (flags & SGK_F_RING_TILE_MAJOR) != 0;
  const int32_t nt = ring_stores_nt(sh, (int64_t)ring * sh.n * (sh.n_cells + 4), ring);  // as launch_rollout_stream decides it
  SGK_DISPATCH_ENV(sh.env_id, (ring_probe_kernel<Geom<E>::NC><<<dim3(grid), dim3(WG), 0, st>>>(boards_ring, recs_ring, sh.n, ring, tm, nt)));
  return hipGetLastError();
}

hipError_t launch_reset(const Shard &sh, const uint8_t *mask, int mode, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  int grid = grid_for((sh.n + WG - 1) / WG, sh.max_grid);
  SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout,
                          reset_kernel<E, L><<<dim3(grid), dim3(WG), 0, st>>>(sh.rules_dev, sh.state,
                                              sh.boards, mask, mode, sh.n, sh.seed, sh.env_base, sh.n_resets, sh.aux, StepStore{}));
  return hipGetLastError();
}

// sgk_reset_done + ReplayBuffer.add's first half for the NEXT step in one launch (sgk_reset_done_store): the envs whose episode is
// over are reset, then every env's board -- what the next action is chosen on -- goes into slice (slice [+ *slice_dev]) % ring of
// the states ring
hipError_t launch_reset_done_store(const Shard &sh, uint32_t flags, int64_t slice, const long long *slice_dev, int32_t ring, int8_t *states,
                                   hipStream_t st) {
  (void)hipGetLastError();
  int grid = grid_for((sh.n + WG - 1) / WG, sh.max_grid);
  const StepStore so = make_store(sh, states, nullptr, nullptr, nullptr, slice, slice_dev, ring, 0);
  const int mode = 1 | ((flags & SGK_F_NO_BOARDS) ? 4 : 0);
  SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout,
                          reset_kernel<E, L, true><<<dim3(grid), dim3(WG), 0, st>>>(sh.rules_dev, sh.state,
                                              sh.boards, nullptr, mode, sh.n, sh.seed, sh.env_base, sh.n_resets, sh.aux, so));
  return hipGetLastError();
}

__global__ __launch_bounds__(WG) void fill_f64_kernel(double *__restrict__ dst, int64_t count, double value) {
  for (int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x; i < count; i += (int64_t)gridDim.x * WG) dst[i] = value;
}

// the float64 side state of a fresh env object (friend or foe: every PolicyEstimator starts at [0.5, 0.5])
hipError_t launch_aux_init(const Shard &sh, hipStream_t st) {
  (void)hipGetLastError();
  if (!sh.aux) return hipSuccess;
  const int64_t count = sh.n * SGK_AUX_DOUBLES;
  fill_f64_kernel<<<dim3(grid_for((count + WG - 1) / WG, sh.max_grid)), dim3(WG), 0, st>>>(sh.aux, count, 0.5);
  return hipGetLastError();
}

hipError_t launch_metrics_init(const Shard &sh, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  metrics_init_kernel<<<dim3(32), dim3(WG), 0, st>>>((long long *)sh.metric_slab);
  return hipGetLastError();
}

hipError_t launch_metrics_reduce(const Shard &sh, hipStream_t st, long long *out_host) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  metrics_reduce_kernel<<<dim3(1), dim3(1024), 0, st>>>((const long long *)sh.metric_slab, (long long *)sh.metrics, out_host);
  return hipGetLastError();
}

hipError_t launch_obs_f32(const Shard &sh, float *dst, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  int64_t total = sh.n * ((sh.n_cells + 3) / 4);
  int grid = grid_for((total + WG - 1) / WG, sh.max_grid * 4);
  hipLaunchKernelGGL(obs_f32_kernel, dim3(grid), dim3(WG), 0, st, sh.boards, dst, sh.n, sh.n_cells, sh.pitch);
  return hipGetLastError();
}

hipError_t launch_render_rgb(const Shard &sh, uint8_t *dst, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  int64_t total = sh.n * sh.n_cells;
  int grid = grid_for((total + WG - 1) / WG, sh.max_grid * 4);
  hipLaunchKernelGGL(render_rgb_kernel, dim3(grid), dim3(WG), 0, st, sh.rules_dev, sh.boards, dst, sh.n, sh.n_cells, sh.pitch);
  return hipGetLastError();
}

hipError_t launch_dense_boards(const Shard &sh, int8_t *dst, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  int64_t total = sh.n * sh.n_cells;
  int grid = grid_for((total + WG - 1) / WG, sh.max_grid * 4);
  hipLaunchKernelGGL(dense_boards_kernel, dim3(grid), dim3(WG), 0, st, sh.boards, dst, sh.n, sh.n_cells, sh.pitch);
  return hipGetLastError();
}

hipError_t launch_finished(const Shard &sh, int32_t *ids, int32_t *ret, int32_t *perf, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  int64_t n_wg = (sh.n + WG - 1) / WG;
  hipLaunchKernelGGL(finished_count_kernel, dim3((unsigned)n_wg), dim3(WG), 0, st, sh.rec, sh.wg_count, sh.n);
  hipLaunchKernelGGL(finished_scan_kernel, dim3(1), dim3(1024), 0, st, sh.wg_count, sh.wg_offset, n_wg, sh.finished_total);
  hipLaunchKernelGGL(finished_scatter_kernel, dim3((unsigned)n_wg), dim3(WG), 0, st, sh.rec, sh.wg_offset, sh.last_return,
                     sh.last_perf, ids, ret, perf, sh.n);
  return hipGetLastError();
}

}  // namespace sgk
