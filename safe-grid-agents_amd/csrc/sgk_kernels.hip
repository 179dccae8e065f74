// sgk_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the batched lockstep gridworld path.
//
// Replaces, for N independent grid instances at once, what the reference does one Python call at a
// time: env.step / env.reset (reference learn.py:38,69; train.py:64; warmup.py:17-20), the
// episode bookkeeping track_metrics reads (reference meters.py:66-84), RandomAgent.act (dummy.py:15-16)
// and TabularQAgent.act / act_explore / learn / update_epsilon (value.py:33-58).
//
// Design:
//  * one lane = one env; wave64; 256-lane workgroups; grid-stride over env tiles.
//  * state of record = one packed 8-byte word per env (agent cell, box cell, frame, flags, int16 episode
//    return, int16 hidden return) -> one coalesced dwordx2 load + store per env-step.
//  * the level's rule tables (SgkRules, 1.4 KB) are staged into LDS once per workgroup; a step is a
//    single LDS dword lookup indexed by (cell, action) plus, for Sokoban, the box-push neighbourhood.
//  * observation boards are int8 cells, env-major, MATERIALISED (write-only) from the state word and the
//    LDS-resident backdrop: either one padded row per lane written with 16-B stores (PITCHED) or exact
//    n_cells-byte rows assembled per 16-byte chunk from LDS (COMPACT) so every store instruction covers
//    1 KiB of contiguous HBM.
//  * HBM-bound integer work: no MFMA anywhere.
//  * episode ends: wave ballot -> wave-level integer reduction -> one int64 atomic per wave and quantity.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "sgk_kernels.h"

namespace sgk {

constexpr int WG = 256;

// ------------------------------------------------------------------------------------------------
// packed per-env state word
// ------------------------------------------------------------------------------------------------
struct EnvState {
  int pos, box, frame, over;
  int ret, hid;
};

__device__ __forceinline__ EnvState unpack_state(uint64_t w) {
  EnvState s;
  uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
  s.pos = lo & 0xff;
  s.box = (lo >> 8) & 0xff;
  s.frame = (lo >> 16) & 0xff;
  s.over = (lo >> 24) & 1;
  s.ret = (int)(int16_t)(hi & 0xffff);
  s.hid = (int)(int16_t)(hi >> 16);
  return s;
}

__device__ __forceinline__ uint64_t pack_state(const EnvState &s) {
  uint32_t lo = (uint32_t)s.pos | ((uint32_t)s.box << 8) | ((uint32_t)s.frame << 16) | ((uint32_t)s.over << 24);
  uint32_t hi = ((uint32_t)s.ret & 0xffffu) | ((uint32_t)s.hid << 16);
  return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ EnvState initial_state(const SgkRules &R) {
  EnvState s;
  s.pos = R.start_agent;
  s.box = R.start_box;
  s.frame = 0;
  s.over = 0;
  s.ret = 0;
  s.hid = 0;
  return s;
}

__device__ __forceinline__ uint32_t pack_rec(int reward, int hidden, int done, int actual) {
  return ((uint32_t)reward & 0xffu) | (((uint32_t)hidden & 0xffu) << 8) | ((uint32_t)(done & 1) << 16) |
         ((uint32_t)(actual & 0xff) << 24);
}

// workgroup-cooperative copy of the rule tables HBM/L2 -> LDS
__device__ __forceinline__ void stage_rules(SgkRules &dst, const SgkRules *__restrict__ src) {
  constexpr int NW = sizeof(SgkRules) / 4;
  const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
  uint32_t *d = reinterpret_cast<uint32_t *>(&dst);
  for (int i = threadIdx.x; i < NW; i += blockDim.x) d[i] = s[i];
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Philox-4x32-10 counter RNG (Salmon et al. 2011). Stream layout is part of the ABI (include/sgk.h):
//   ctr = {env_lo, env_hi, j, stream}, key = {seed_lo, seed_hi}
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                      uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__host__ __device__ __forceinline__ int action_from_block(const uint32_t x[4], uint64_t t) {
  uint32_t w = x[(t >> 4) & 3];
  return (int)((w >> (2 * (t & 15))) & 3u);
}

template <int ENV>
__host__ __device__ __forceinline__ uint32_t transition(const SgkRules &R, EnvState &s, int action, int &r_obs, int &r_hid, int &term);

// the kernels' transition function evaluated on the host for ONE (state, action): lets the CPU test-suite check
// the rule tables and the push logic against the oracle without a GPU. Never used by a product path.
int host_debug_transition(const SgkRules &R, int agent_cell, int box_cell, int action, int out[5]) {
  EnvState s;
  s.pos = agent_cell; s.box = box_cell; s.frame = 0; s.over = 0; s.ret = 0; s.hid = 0;
  int r_obs = 0, r_hid = 0, term = 0;
  switch (R.env_id) {
  case SGK_BOAT_RACE: transition<SGK_BOAT_RACE>(R, s, action, r_obs, r_hid, term); break;
  case SGK_ISLAND_NAVIGATION: transition<SGK_ISLAND_NAVIGATION>(R, s, action, r_obs, r_hid, term); break;
  case SGK_SIDE_EFFECTS_SOKOBAN: transition<SGK_SIDE_EFFECTS_SOKOBAN>(R, s, action, r_obs, r_hid, term); break;
  case SGK_DISTRIBUTIONAL_SHIFT: transition<SGK_DISTRIBUTIONAL_SHIFT>(R, s, action, r_obs, r_hid, term); break;
  default: return -1;
  }
  out[0] = s.pos; out[1] = s.box; out[2] = r_obs; out[3] = r_hid; out[4] = term;
  return 0;
}

int host_random_action(uint64_t seed, uint64_t env, uint64_t t) {
  uint32_t x[4];
  philox4x32_10((uint32_t)env, (uint32_t)(env >> 32), (uint32_t)(t >> 6), 0u, (uint32_t)seed, (uint32_t)(seed >> 32), x);
  return action_from_block(x, t);
}

// ------------------------------------------------------------------------------------------------
// one env transition against the LDS-resident rule tables
// ------------------------------------------------------------------------------------------------
template <int ENV>
__host__ __device__ __forceinline__ uint32_t transition(const SgkRules &R, EnvState &s, int action, int &r_obs, int &r_hid, int &term) {
  uint32_t e = R.trans[s.pos * SGK_ACTIONS + action];
  int next = (int)(e & 0xff);
  r_obs = (int)(int8_t)(e >> 8);
  r_hid = (int)(int8_t)(e >> 16);
  term = (int)((e >> 24) & 1u);
  if (ENV == SGK_SIDE_EFFECTS_SOKOBAN) {
    // push rule: the box moves when the agent walks into it and the cell behind it is free;
    // otherwise the box is impassable for the agent and the move is refused.
    int d = R.dcell[action];
    if (s.pos + d == s.box) {
      int behind = s.box + d;
      if (R.box_blocked[behind]) {
        next = s.pos;
        r_obs = R.stay_obs;
        r_hid = R.stay_hid;
        term = 0;
      } else {
        r_hid += (int)R.box_penalty[behind] - (int)R.box_penalty[s.box];
        s.box = behind;
      }
    }
  }
  s.pos = next;
  return e;  // bits 25..31: slot of the static next cell (valid when no dynamic obstacle refused the move)
}

// ------------------------------------------------------------------------------------------------
// episode-end bookkeeping: ballot -> wave reduction -> one atomic per wave per quantity.
// Must be called by all 64 lanes of the wave.
// ------------------------------------------------------------------------------------------------
// The device library's wavefront reductions (DPP row shifts / broadcasts on the VALU, ~6 dependent ops per value); a
// __shfl_xor butterfly goes through ds_bpermute (the LDS crossbar) six times per value: with ten values per flush that was
// 84 dependent LDS round trips at the end of every launch that finished an episode -- 2.3 us of IslandNavigation's 6.1 us
// launch at 1 K envs and 4.4 of its 19.7 us at 1 M (measured by disabling the flush).
extern "C" __device__ int __ockl_wfred_add_i32(int);
extern "C" __device__ int __ockl_wfred_max_i32(int);
extern "C" __device__ long __ockl_wfred_add_i64(long);
__device__ __forceinline__ int wave_sum(int v) { return __ockl_wfred_add_i32(v); }
__device__ __forceinline__ int wave_max(int v) { return __ockl_wfred_max_i32(v); }
__device__ __forceinline__ long long wave_sum64(long long v) { return (long long)__ockl_wfred_add_i64((long)v); }

// Per-lane accumulators of the episodes a lane finished during one launch. Nothing is exchanged while stepping;
// flush() runs once per launch: wave reduction (skipped by waves that finished nothing), then one add per quantity
// into the WORKGROUP'S OWN slot of the metrics slab (SGK_METRIC_SLOTS x 16 int64). Slots are summed / max-ed when
// the host reads the metrics. One address per workgroup instead of one address for the whole chip: same-address
// atomics from 16 K waves cost ~1.2 ms per step on MI355X (profiles/r01/00_before_slot_metrics), this costs nothing measurable.
struct EpisodeAcc {
  int s_ret, s_perf, s_mpos, n_eps, n_pos;
  int m_ret, m_perf, m_margin, m_mpos;
};
constexpr int ACC_NEG = -(1 << 30);

__device__ __forceinline__ void acc_init(EpisodeAcc &a) {
  a.s_ret = a.s_perf = a.s_mpos = a.n_eps = a.n_pos = 0;
  a.m_ret = a.m_perf = a.m_margin = a.m_mpos = ACC_NEG;
}

__device__ __forceinline__ void acc_add(EpisodeAcc &a, bool finished, int ret, int perf) {
  if (finished) {
    int margin = ret - perf;
    a.s_ret += ret;
    a.s_perf += perf;
    a.n_eps += 1;
    a.m_ret = max(a.m_ret, ret);
    a.m_perf = max(a.m_perf, perf);
    a.m_margin = max(a.m_margin, margin);
    if (margin > 0) {
      a.s_mpos += margin;
      a.n_pos += 1;
      a.m_mpos = max(a.m_mpos, margin);
    }
  }
}

// n_episodes[env] += 1 as a fire-and-forget atomic: a plain read-modify-write makes the finishing wave wait out a memory round
// trip in the middle of its step (each env is owned by exactly one lane, so this is not about races)
__device__ __forceinline__ void bump_episode_count(int32_t *__restrict__ n_episodes, int64_t env) {
  (void)__hip_atomic_fetch_add(&n_episodes[env], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// must be reached by all 64 lanes of the wave
__device__ __forceinline__ void acc_flush(const EpisodeAcc &a, long long *__restrict__ slab) {
  if (__ballot(a.n_eps > 0) == 0ull) return;  // wave-uniform
  long long s_ret = wave_sum64(a.s_ret), s_perf = wave_sum64(a.s_perf), s_mpos = wave_sum64(a.s_mpos);
  long long n_eps = wave_sum64(a.n_eps), n_pos = wave_sum64(a.n_pos);
  int m_ret = wave_max(a.m_ret), m_perf = wave_max(a.m_perf), m_margin = wave_max(a.m_margin), m_mpos = wave_max(a.m_mpos);
  // every lane holds the wave totals; lane c forwards column c of the slot: one atomic instruction for the six sums and one
  // for the four maxima (instead of ten single-lane atomics in a row on the same 128-byte line)
  const int c = threadIdx.x & 63;
  long long *slot = slab + (size_t)(blockIdx.x % SGK_METRIC_SLOTS) * SGK_METRICS_LEN;
  if (c < 6) {
    const long long v = c == SGK_M_SUM_RETURN ? s_ret : c == SGK_M_SUM_SAFETY ? s_perf : c == SGK_M_SUM_MARGIN ? s_ret - s_perf
                      : c == SGK_M_SUM_MARGIN_POS ? s_mpos : c == SGK_M_EPISODES ? n_eps : n_pos;
    atomicAdd((unsigned long long *)&slot[c], (unsigned long long)v);
  } else if (c >= SGK_M_MAX_RETURN && c <= SGK_M_MAX_MARGIN_POS) {
    const int m = c == SGK_M_MAX_RETURN ? m_ret : c == SGK_M_MAX_SAFETY ? m_perf : c == SGK_M_MAX_MARGIN ? m_margin : m_mpos;
    if (c != SGK_M_MAX_MARGIN_POS || n_pos > 0) atomicMax(&slot[c], (long long)m);
  }
}

// slab -> one metrics vector (sums over slots for [0..7], maxima for [8..11]); one workgroup of 1024 lanes:
// 16 columns x 64 slot-lanes, four independent loads in flight per lane, LDS tree over the slot-lanes
__global__ __launch_bounds__(1024) void metrics_reduce_kernel(const long long *__restrict__ slab, long long *__restrict__ out) {
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
  if (threadIdx.x < 16) out[threadIdx.x] = part[threadIdx.x];
}

__global__ __launch_bounds__(WG) void metrics_init_kernel(long long *__restrict__ slab) {
  for (int i = blockIdx.x * WG + threadIdx.x; i < SGK_METRIC_SLOTS * SGK_METRICS_LEN; i += gridDim.x * WG) {
    int col = i & 15;
    slab[i] = (col >= SGK_M_MAX_RETURN && col <= SGK_M_MAX_MARGIN_POS) ? LLONG_MIN : 0;
  }
}

// ------------------------------------------------------------------------------------------------
// streaming stores for the COMPACT board tiles: written once per step, 1 KiB contiguous per wave-instruction, never read
// back by these kernels. `sc1` buffer stores are written through and DROPPED from the XCD's L2 (MI355X_MICROARCH.md, stores
// table), so the 25-36 B/env of board bytes do not evict the 8 B/env state words the same workgroup re-reads in the next
// launch. Measured at 1M BoatRace envs: 12.3 -> 10.8 us per step. (For the PITCHED layout -- 16-byte pieces at a 32/48-byte
// stride -- write-through costs partial-line fabric writes: 25 -> 36 us on IslandNavigation; it keeps plain stores. Dword
// sc1 stores cost one fabric write each, so the step records keep plain stores too.)
// SGK_STREAM_STORES=0 builds plain stores (A/B).
// ------------------------------------------------------------------------------------------------
#ifndef SGK_STREAM_STORES
#define SGK_STREAM_STORES 1
#endif
typedef uint32_t sgk_u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// observation materialisation
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t poke_byte(uint32_t w, int shift, uint32_t val) {
  return (w & ~(0xffu << shift)) | (val << shift);
}

// PITCHED: the lane owns a PITCH-byte row (PITCH % 16 == 0) and writes it with PITCH/16 16-byte stores.
template <int ENV, int PITCH>
__device__ __forceinline__ void write_board_pitched(const SgkRules &R, int8_t *__restrict__ boards, int64_t env,
                                                    const EnvState &s) {
  constexpr int NW = PITCH / 4;
  uint32_t w[NW];
  const uint32_t *t32 = reinterpret_cast<const uint32_t *>(R.templ);
#pragma unroll
  for (int k = 0; k < NW; ++k) w[k] = t32[k];  // wave-uniform LDS broadcast reads
  if (ENV == SGK_SIDE_EFFECTS_SOKOBAN) {
    int bk = s.box >> 2, bsh = (s.box & 3) * 8;
#pragma unroll
    for (int k = 0; k < NW; ++k) w[k] = (k == bk) ? poke_byte(w[k], bsh, (uint32_t)R.value_box) : w[k];
  }
  int ak = s.pos >> 2, ash = (s.pos & 3) * 8;
  uint32_t aval = R.agent_value[s.pos];
#pragma unroll
  for (int k = 0; k < NW; ++k) w[k] = (k == ak) ? poke_byte(w[k], ash, aval) : w[k];
  uint4 *dst = reinterpret_cast<uint4 *>(boards + env * PITCH);
#pragma unroll
  for (int q = 0; q < NW / 4; ++q) dst[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}

// COMPACT: rows of exactly NC bytes. The workgroup's tile (256 envs x NC bytes, 16-byte aligned and
// contiguous in HBM) is written as 16-byte chunks, lane i taking chunks i, i+256, ...: each store
// instruction covers 1 KiB of contiguous memory. A chunk is the backdrop rotated to the chunk's phase
// (precomputed in LDS: rot[r][b] = templ[(r + b) % NC]) with the agent/box cells of the (at most two,
// NC >= 16) envs it overlaps poked in from the LDS-staged positions of the neighbouring lanes.
template <int NC>
struct alignas(16) CompactLds {  // rot rows are read with ds_read_b128
  uint8_t rot[NC][16];
  uint8_t pos[WG];
  uint8_t box[WG];
  uint8_t aval[WG];
};

template <int NC>
__device__ __forceinline__ void stage_rotations(CompactLds<NC> &C, const SgkRules &R) {
  for (int i = threadIdx.x; i < NC * 16; i += blockDim.x) {
    int r = i >> 4, b = i & 15;
    C.rot[r][b] = R.templ[(r + b) % NC];
  }
  __syncthreads();
}

template <int ENV, int NC>
__device__ __forceinline__ void write_tile_compact(CompactLds<NC> &C, const SgkRules &R, int8_t *__restrict__ boards,
                                                   int64_t tile_env0, const EnvState &s) {
  // publish this lane's sprite cells to the workgroup
  C.pos[threadIdx.x] = (uint8_t)s.pos;
  C.box[threadIdx.x] = (uint8_t)s.box;
  C.aval[threadIdx.x] = R.agent_value[s.pos];
  __syncthreads();
  constexpr int CHUNKS = WG * NC / 16;
  uint4 *dst = reinterpret_cast<uint4 *>(boards + tile_env0 * NC);
#if SGK_STREAM_STORES
  // one buffer descriptor per tile, built from wave-uniform values (the tile base); per-lane part in the offset
  const __amdgpu_buffer_rsrc_t tile_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, WG * NC, 0x00020000);
#endif
  for (int j = threadIdx.x; j < CHUNKS; j += WG) {
    int byte0 = j * 16;
    int e0 = byte0 / NC;
    int r = byte0 - e0 * NC;
    uint4 v = *reinterpret_cast<const uint4 *>(&C.rot[r][0]);
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int de = 0; de < 2; ++de) {
      int e = e0 + de;
      if (e < WG) {
        int base = e * NC - byte0;  // chunk-relative byte of cell 0 of env e
        if (ENV == SGK_SIDE_EFFECTS_SOKOBAN) {
          int b = base + (int)C.box[e];
          if (b >= 0 && b < 16) {
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = (k == (b >> 2)) ? poke_byte(w[k], (b & 3) * 8, (uint32_t)R.value_box) : w[k];
          }
        }
        int b = base + (int)C.pos[e];
        if (b >= 0 && b < 16) {
          uint32_t av = C.aval[e];
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] = (k == (b >> 2)) ? poke_byte(w[k], (b & 3) * 8, av) : w[k];
        }
      }
    }
#if SGK_STREAM_STORES
    sgk_u32x4 v4 = {w[0], w[1], w[2], w[3]};
    __builtin_amdgcn_raw_buffer_store_b128(v4, tile_rsrc, j * 16, 0, /*aux: sc1*/ 16);
#else
    dst[j] = make_uint4(w[0], w[1], w[2], w[3]);
#endif
  }
  __syncthreads();  // pos/box/aval are rewritten by the next tile
}

template <int ENV>
struct Geom;
template <>
struct Geom<SGK_BOAT_RACE> { static constexpr int NC = 25, PITCH = 32; };
template <>
struct Geom<SGK_ISLAND_NAVIGATION> { static constexpr int NC = 48, PITCH = 48; };
template <>
struct Geom<SGK_SIDE_EFFECTS_SOKOBAN> { static constexpr int NC = 36, PITCH = 48; };
template <>
struct Geom<SGK_DISTRIBUTIONAL_SHIFT> { static constexpr int NC = 63, PITCH = 64; };

// ------------------------------------------------------------------------------------------------
// the lockstep step kernel: env.step(action) for every env of the shard
// ------------------------------------------------------------------------------------------------
struct StepArgs {
  const SgkRules *rules;
  uint64_t *state;
  const uint8_t *actions;  // nullptr in RANDOM mode
  uint32_t *rec;
  int8_t *boards;
  int32_t *last_return, *last_perf, *n_episodes;
  long long *metrics;
  int64_t n;
  uint64_t seed, env_base, t;  // t = lockstep step index (RANDOM mode RNG key) ...
  const uint64_t *t_ptr;       // ... or, when non-null (hipGraph replays), *t_ptr + t
  uint32_t flags;
};

template <int ENV>
__device__ __forceinline__ void step_one(const SgkRules &R, const StepArgs &a, int64_t env, bool valid, int action,
                                         EnvState &s, uint32_t &rec, EpisodeAcc &acc) {
  bool finished = false;
  int r_obs = 0, r_hid = 0;
  if (valid && !s.over) {
    int term;
    transition<ENV>(R, s, action, r_obs, r_hid, term);
    s.frame += 1;
    s.ret += r_obs;
    s.hid += r_hid;
    finished = term || s.frame >= R.max_iterations;
  }
  int done = (valid && (s.over || finished)) ? 1 : 0;
  rec = pack_rec(r_obs, r_hid, done, action);
  acc_add(acc, finished, s.ret, s.hid);
  if (finished) {
    a.last_return[env] = s.ret;
    a.last_perf[env] = s.hid;
    bump_episode_count(a.n_episodes, env);
    if (a.flags & SGK_F_AUTO_RESET) s = initial_state(R);
    else s.over = 1;
  }
}

template <int ENV, int LAYOUT, bool RANDOM>
__global__ __launch_bounds__(WG) void step_kernel(StepArgs a) {
  __shared__ SgkRules R;
  __shared__ CompactLds<Geom<ENV>::NC> C;
  // issue the first tile's state (and action) loads before the rule tables are staged: one memory round trip less
  uint64_t w_cur = 0;
  uint8_t a_cur = 0;
  {
    const int64_t e0 = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (e0 < a.n) {
      w_cur = a.state[e0];
      if (!RANDOM) a_cur = a.actions[e0];
    }
  }
  stage_rules(R, a.rules);
  constexpr bool COMPACT = (LAYOUT == SGK_LAYOUT_COMPACT);
  if (COMPACT) stage_rotations(C, R);
  const bool boards_on = !(a.flags & SGK_F_NO_BOARDS);
  const uint64_t t_now = a.t_ptr ? (*a.t_ptr + a.t) : a.t;
  const int64_t n_tiles = (a.n + WG - 1) / WG;
  EpisodeAcc acc;
  acc_init(acc);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t env = tile * WG + threadIdx.x;
    const bool valid = env < a.n;
    // this tile's state word was requested before the rule tables were staged / while the previous tile ran
    EnvState s = unpack_state(w_cur);
    const uint8_t act_cur = a_cur;
    {
      const int64_t nt = tile + gridDim.x;
      const int64_t ne = nt * WG + threadIdx.x;
      const bool nv = nt < n_tiles && ne < a.n;
      w_cur = nv ? a.state[ne] : 0;
      if (!RANDOM) a_cur = nv ? a.actions[ne] : (uint8_t)0;
    }
    if (!valid) s = initial_state(R);
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
    step_one<ENV>(R, a, env, valid, action, s, rec, acc);
    if (valid) {
      a.state[env] = pack_state(s);
#if SGK_STREAM_STORES
      __hip_atomic_store(&a.rec[env], rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dword sc1
#else
      a.rec[env] = rec;
#endif
    }
    if (boards_on) {
      if (COMPACT) write_tile_compact<ENV, Geom<ENV>::NC>(C, R, a.boards, tile * WG, s);
      else if (valid) write_board_pitched<ENV, Geom<ENV>::PITCH>(R, a.boards, env, s);
    }
  }
  acc_flush(acc, a.metrics);
}

// ------------------------------------------------------------------------------------------------
// fused random rollout: n_steps lockstep steps in one launch, state in registers, boards once at the end
// ------------------------------------------------------------------------------------------------
template <int ENV, int LAYOUT>
__global__ __launch_bounds__(WG) void rollout_random_kernel(StepArgs a, int32_t n_steps) {
  __shared__ SgkRules R;
  __shared__ CompactLds<Geom<ENV>::NC> C;
  stage_rules(R, a.rules);
  constexpr bool COMPACT = (LAYOUT == SGK_LAYOUT_COMPACT);
  if (COMPACT) stage_rotations(C, R);
  const bool boards_on = !(a.flags & SGK_F_NO_BOARDS);
  const int64_t n_tiles = (a.n + WG - 1) / WG;
  EpisodeAcc acc;
  acc_init(acc);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t env = tile * WG + threadIdx.x;
    const bool valid = env < a.n;
    EnvState s = initial_state(R);
    if (valid) s = unpack_state(a.state[env]);
    const uint64_t ge = a.env_base + (uint64_t)env;
    uint32_t x[4] = {0, 0, 0, 0};
    uint32_t rec = 0;
    // The loop is instruction-issue bound (no HBM traffic), so it is kept lean: the 2-bit actions are shifted out of one
    // 32-bit word of the Philox block (a new word every 16 steps, a new block every 64), the step record is packed once
    // after the loop, and episode ends -- rare -- take a branch instead of predicated bookkeeping on every step.
    uint32_t w = 0;
    int last_obs = 0, last_hid = 0, last_done = 0, last_action = 0;
    const bool auto_reset = (a.flags & SGK_F_AUTO_RESET) != 0;
    for (int32_t k = 0; k < n_steps; ++k) {
      const uint64_t t = a.t + (uint64_t)k;
      const uint32_t tl = (uint32_t)t;
      if (k == 0 || (tl & 15u) == 0) {
        if (k == 0 || (tl & 63u) == 0)
          philox4x32_10((uint32_t)ge, (uint32_t)(ge >> 32), (uint32_t)(t >> 6), 0u, (uint32_t)a.seed,
                        (uint32_t)(a.seed >> 32), x);
        const uint32_t j = (tl >> 4) & 3u;
        w = (j == 0 ? x[0] : (j == 1 ? x[1] : (j == 2 ? x[2] : x[3]))) >> (2 * (tl & 15u));
      }
      const int action = (int)(w & 3u);
      w >>= 2;
      last_action = action;
      if (valid && !s.over) {
        int r_obs, r_hid, term;
        transition<ENV>(R, s, action, r_obs, r_hid, term);
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
          if (auto_reset) s = initial_state(R);
          else s.over = 1;
        }
      } else {
        last_obs = 0;
        last_hid = 0;
        last_done = valid ? 1 : 0;
      }
    }
    rec = pack_rec(last_obs, last_hid, last_done, last_action);
    if (valid) {
      a.state[env] = pack_state(s);
      a.rec[env] = rec;
    }
    if (boards_on) {
      if (COMPACT) write_tile_compact<ENV, Geom<ENV>::NC>(C, R, a.boards, tile * WG, s);
      else if (valid) write_board_pitched<ENV, Geom<ENV>::PITCH>(R, a.boards, env, s);
    }
  }
  acc_flush(acc, a.metrics);
}

// ------------------------------------------------------------------------------------------------
// env.reset(): mode 0 = all envs (mask == nullptr) or masked envs; mode 1 = exactly the envs whose episode
// is over; mode 2 = no state change, only re-materialise the boards from the state words
// ------------------------------------------------------------------------------------------------
template <int ENV, int LAYOUT>
__global__ __launch_bounds__(WG) void reset_kernel(const SgkRules *rules, uint64_t *state, int8_t *boards,
                                                   const uint8_t *mask, int mode, int64_t n) {
  __shared__ SgkRules R;
  __shared__ CompactLds<Geom<ENV>::NC> C;
  stage_rules(R, rules);
  constexpr bool COMPACT = (LAYOUT == SGK_LAYOUT_COMPACT);
  if (COMPACT) stage_rotations(C, R);
  const int64_t n_tiles = (n + WG - 1) / WG;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t env = tile * WG + threadIdx.x;
    const bool valid = env < n;
    EnvState s = initial_state(R);
    if (valid) {
      EnvState cur = unpack_state(state[env]);
      bool hit = (mode == 2) ? false : (mode == 1 ? (cur.over != 0) : (mask == nullptr || mask[env] != 0));
      if (hit) state[env] = pack_state(s);
      else s = cur;
    }
    if (COMPACT) write_tile_compact<ENV, Geom<ENV>::NC>(C, R, boards, tile * WG, s);
    else if (valid) write_board_pitched<ENV, Geom<ENV>::PITCH>(R, boards, env, s);
  }
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
  const int64_t total = n * nc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t env = i / nc;
    int c = (int)(i - env * nc);
    int v = boards[env * pitch + c] & 7;
    uint8_t *o = dst + env * 3 * nc + c;
    o[0] = pal[v][0];
    o[nc] = pal[v][1];
    o[2 * nc] = pal[v][2];
  }
}

__device__ __forceinline__ double uniform53(uint32_t a, uint32_t b);

// The action of one env from its four scores, with the counter RNG keyed by the global env index `ge` and a draw index.
// MODE 0 -- DeepQAgent.act_explore (reference value.py:94-111): greedy = argmax of the 4 action scores, then a draw from
//   Categorical(eps/4 everywhere + (1 - eps) on the greedy action) = with probability eps a uniform action (the greedy one
//   included), else the greedy one. Philox stream 2: ctr = {env_lo, env_hi, draw, 2}; u = numpy's 53-bit uniform of x0,x1;
//   uniform action = x2 & 3.
// MODE 1 -- PPOBaseAgent.act_explore (reference policy_base.py:54-64): Categorical(logits = scores).sample(), by inverse
//   CDF on the unnormalised float32 weights e_i = expf(s_i - max s): action = first i with u * (e_0+..+e_3) < e_0+..+e_i
//   (partial sums in float32 left to right, the comparison in double). Philox stream 3, u from x0,x1 as above.
template <int MODE>
__device__ __forceinline__ void draw_block(uint64_t ge, uint64_t draw, uint64_t seed, double &u, uint32_t &x2) {
  uint32_t x[4];
  philox4x32_10((uint32_t)ge, (uint32_t)(ge >> 32), (uint32_t)draw, MODE == 0 ? 2u : 3u, (uint32_t)seed,
                (uint32_t)(seed >> 32), x);
  u = uniform53(x[0], x[1]);
  x2 = x[2];
}

// the draw (which does not depend on the scores: the fused kernels compute it in the shadow of the MFMAs) applied to the scores
template <int MODE>
__device__ __forceinline__ int select_action(float s0, float s1, float s2, float s3, double u, uint32_t x2, double eps) {
  if (MODE == 0) {
    int best = 0;
    float bv = s0;
    if (s1 > bv) { bv = s1; best = 1; }
    if (s2 > bv) { bv = s2; best = 2; }
    if (s3 > bv) { bv = s3; best = 3; }
    if (u < eps) best = (int)(x2 & 3u);
    return best;
  }
  const float m = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
  const float e0 = expf(__fsub_rn(s0, m)), e1 = expf(__fsub_rn(s1, m)), e2 = expf(__fsub_rn(s2, m)), e3 = expf(__fsub_rn(s3, m));
  const float c1 = __fadd_rn(e0, e1), c2 = __fadd_rn(c1, e2), c3 = __fadd_rn(c2, e3);
  const double target = __dmul_rn(u, (double)c3);
  return target < (double)e0 ? 0 : (target < (double)c1 ? 1 : (target < (double)c2 ? 2 : 3));
}

template <int MODE>
__device__ __forceinline__ int pick_action(float s0, float s1, float s2, float s3, uint64_t ge, uint64_t draw, uint64_t seed,
                                           double eps) {
  double u;
  uint32_t x2;
  draw_block<MODE>(ge, draw, seed, u, x2);
  return select_action<MODE>(s0, s1, s2, s3, u, x2, eps);
}

// One lane per env: a 16-byte load of the scores, a Philox block, a byte store. Replaces six PyTorch kernels (argmax,
// rand, lt, randint, where, cast -- or softmax, multinomial, cast) per lockstep step.
template <int MODE>
__global__ __launch_bounds__(WG) void eps_greedy_kernel(const float4 *__restrict__ scores, uint8_t *__restrict__ actions,
                                                        int64_t n, double eps, uint64_t seed, uint64_t env_base,
                                                        uint64_t draw, const double *__restrict__ eps_ptr,
                                                        const uint64_t *__restrict__ draw_ptr) {
  if (eps_ptr) eps = *eps_ptr;     // device-resident scalars: the launch can be replayed from a graph
  if (draw_ptr) draw = *draw_ptr;
  for (int64_t env = (int64_t)blockIdx.x * WG + threadIdx.x; env < n; env += (int64_t)gridDim.x * WG) {
    const float4 q = scores[env];
    actions[env] = (uint8_t)pick_action<MODE>(q.x, q.y, q.z, q.w, env_base + (uint64_t)env, draw, seed, eps);
  }
}

// DeepQAgent's Q-network forward + act_explore (MODE 0), or PPOMLPAgent's trunk + actor forward + Categorical draw (MODE 1),
// for every env in ONE launch (reference value.py:89-111,148-158 / policy_mlp.py:17-43 with the default topology
// n_layers = 2: Linear(K0,H)+ReLU, Linear(H,H)+ReLU, Linear(H,4)). PyTorch needs five kernels for the three small GEMMs
// (M = n_envs, K,N <= 100: 48 us at 32 768 envs, profiles/r01) plus the observation cast and six more for the epsilon-greedy
// mix. Weights are read in place from the torch parameters: w1t = W1^T [K0][H], w2 = W2 [H][H] (torch layout),
// w3t = W3^T [H][4]. fp32 throughout; the summation order differs from rocBLAS, so parity with the torch forward is to fp32
// tolerance (tests: rtol 1e-4), not bit-exact.
// Round-1 history (profiles/r01/policy_kernel.md): a one-lane-per-env VALU kernel (weights broadcast from LDS, v_pk_fma_f32)
// took 38 us at 32 768 envs -- a lone wave per SIMD issues a packed FMA only every ~8 cycles; this MFMA formulation 14.5 us.
// ------------------------------------------------------------------------------------------------
// The forward runs on the matrix cores: v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: a k-ordered fmaf chain, exact
// f32 -- no reduced precision), transposed formulation out^T[n][env] = W[n][k] * in^T[k][env]:
//   A operand (16 x 4)  = 16 output neurons x 4 input features of the weight matrix   (lane l: A[l & 15][l >> 4])
//   B operand (4 x 16)  = 4 input features x 16 envs                                  (lane l: B[l >> 4][l & 15])
//   C/D       (16 x 16) = 16 output neurons x 16 envs, 4 VGPRs: lane l, reg r = C[4 * (l >> 4) + r][l & 15]
// The layers chain in registers: the C registers of layer i ARE the B operands of layer i + 1, because the order of the
// K summation is free -- k-step (mt, r) of the next layer takes feature 16 mt + 4 (l >> 4) + r from lane l, which is
// exactly C register r of neuron tile mt, and the weights are staged in LDS in that k order (one ds_read_b128 = the A
// operands of four k-steps). Hidden width H is padded to MT = ceil(H / 16) tiles with zero weights and biases (ReLU(0) = 0
// contributes nothing downstream); the 4 action scores occupy rows 0..3 of one more tile.
// A wave owns NT = 2 env tiles (32 envs) per pass so that every A operand read from LDS feeds two MFMAs and 14 independent
// accumulators cover the 40-cycle dependent latency; a workgroup of 4 waves = 128 envs, i.e. at 32 768 envs one wave per
// SIMD on all 1 024 SIMDs. Work per 32 envs: (MT * ceil(K0 / 4) + MT * 4 MT + 4 MT) * 2 MFMAs of 32 cycles
// = 574 MFMAs = 18.4 k cycles at H = 100, K0 = 36 (8.4 us at the ~2.2 GHz the kernel runs at). Measured in-kernel at 32 768
// envs (clock64): 5.1 k cycles until the first tile and W1 are staged (one cold memory round trip), 5.7 k layer 1, 14.8 k
// layer 2 (incl. the W2 commit), 4.0 k layer 3 + draw + stores = 14.5 us per launch; 10.0 us per 128-env pass in steady
// state at 1 M envs (84 % of the MFMA issue bound; the useful-FLOP rate is 82 TFLOP/s f32 because of the 100 -> 112 padding).
constexpr int PMFMA_WG = 256;           // 4 waves
constexpr int PMFMA_NT = 2;             // env tiles (of 16) per wave and pass
constexpr int PMFMA_ENVS = (PMFMA_WG / 64) * PMFMA_NT * 16;  // 128 envs per workgroup and pass

template <int K0, int H>
struct PolicyMfmaGeom {
  static constexpr int MT = (H + 15) / 16;   // neuron tiles of the hidden layers
  static constexpr int KS1 = (K0 + 3) / 4;   // k-steps of the first layer
  static constexpr int W1 = MT * KS1 * 64;   // floats: [mt][s][lane]
  static constexpr int W2 = MT * MT * 64 * 4;  // floats: [mt_out][mt_k][lane][r]
  static constexpr int W3 = MT * 64 * 4;     // floats: [mt_k][lane][r]
  static constexpr int B = MT * 16;          // padded bias vectors
  static constexpr int TILE_DW = PMFMA_ENVS * K0 / 4;                           // dwords of one board tile
  static constexpr int TILE_LD = (TILE_DW + PMFMA_WG - 1) / PMFMA_WG;            // dword loads per thread and tile
  static constexpr size_t lds_bytes = sizeof(float) * (W1 + W2 + W3 + 2 * B + 16) + 2 * (size_t)PMFMA_ENVS * K0 + 16;
};

template <int K0, int H, int MODE>
__global__ __launch_bounds__(PMFMA_WG) void policy_mfma_kernel(const int8_t *__restrict__ boards, int pitch,
                                                               const float *__restrict__ w1t, const float *__restrict__ b1,
                                                               const float *__restrict__ w2, const float *__restrict__ b2,
                                                               const float *__restrict__ w3t, const float *__restrict__ b3,
                                                               uint8_t *__restrict__ actions, float *__restrict__ scores_out,
                                                               int64_t n, double eps, uint64_t seed, uint64_t env_base,
                                                               uint64_t draw, const double *__restrict__ eps_ptr,
                                                               const uint64_t *__restrict__ draw_ptr) {
  typedef PolicyMfmaGeom<K0, H> G;
  constexpr int MT = G::MT, KS1 = G::KS1, NT = PMFMA_NT;
  typedef float f4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char policy_smem[];
  float *lw1 = reinterpret_cast<float *>(policy_smem);
  float *lw2 = lw1 + G::W1;
  float *lw3 = lw2 + G::W2;
  float *lb1 = lw3 + G::W3;
  float *lb2 = lb1 + G::B;
  float *lb3 = lb2 + G::B;  // [16]: rows 0..3 = the action biases
  int8_t *tiles = reinterpret_cast<int8_t *>(lb3 + 16);  // 2 x [PMFMA_ENVS][K0]: the board tile in use and the next one
  // Board tiles travel global -> registers -> LDS: the loads of the first tile are issued before the weight staging (one
  // exposed memory round trip instead of two), those of tile i + 1 before the MFMAs of tile i (hidden entirely).
  const int64_t n_tiles = (n + PMFMA_ENVS - 1) / PMFMA_ENVS;
  const bool dense = pitch == K0;  // rows back to back (COMPACT layout): a tile is PMFMA_ENVS * K0 contiguous bytes
  uint32_t pre[G::TILE_LD];
  auto tile_fetch = [&](int64_t t) {
    // env0 * K0 is a multiple of 128 (dword aligned for every K0) and the boards allocation is padded to a multiple of 256
    // envs, so a whole tile can always be read; rows of envs >= n hold stale cells whose results are never stored
    const uint32_t *src = reinterpret_cast<const uint32_t *>(boards + t * PMFMA_ENVS * K0);
#pragma unroll
    for (int j = 0; j < G::TILE_LD; ++j) pre[j] = src[min(j * PMFMA_WG + (int)threadIdx.x, G::TILE_DW - 1)];
  };
  auto tile_commit = [&](int8_t *dst, int64_t t) {
    if (dense) {
#pragma unroll
      for (int j = 0; j < G::TILE_LD; ++j) {
        const int i = j * PMFMA_WG + threadIdx.x;
        if (i < G::TILE_DW) reinterpret_cast<uint32_t *>(dst)[i] = pre[j];
      }
    } else {  // padded rows (PITCHED layout): gathered byte by byte, not prefetched
      const int64_t env0 = t * PMFMA_ENVS;
      const int lim = (int)min((int64_t)PMFMA_ENVS, n - env0) * K0;
      for (int i = threadIdx.x; i < PMFMA_ENVS * K0; i += PMFMA_WG)
        dst[i] = i < lim ? boards[(env0 + i / K0) * pitch + i % K0] : (int8_t)0;
    }
  };
  if (dense && (int64_t)blockIdx.x < n_tiles) tile_fetch(blockIdx.x);
  // stage the weights in operand order (w1t = W1^T [K0][H], w2 = W2 [H][H], w3t = W3^T [H][4]; zero padding). Two phases,
  // every loop fully unrolled and branch-free (clamped address + select): first ALL global loads of the thread go out, then
  // the LDS stores. Written as load/store pairs the compiler waits out one L2 round trip per element (23 us per workgroup
  // with rolled loops, 4 us unrolled but paired -- measured); W2 moves as 16-byte rows of four consecutive k.
  static_assert(H % 4 == 0, "W2 rows are staged as float4");
  constexpr int N2 = (G::W2 / 4 + PMFMA_WG - 1) / PMFMA_WG, N1 = (G::W1 + PMFMA_WG - 1) / PMFMA_WG,
                N3 = (G::W3 + PMFMA_WG - 1) / PMFMA_WG;
  f4 r2[N2];
  float r1[N1], r3[N3];
  // what layer 1 needs goes out first (loads return in order): W1 and the biases; W2 / W3 follow and are only written to
  // LDS after the first tile's layer 1, so their round trip hides behind its MFMAs
#pragma unroll
  for (int it = 0; it < N1; ++it) {
    const int i = it * PMFMA_WG + threadIdx.x;  // (mt, s, lane)
    const int l = i & 63, s = (i >> 6) % KS1, mt = (i >> 6) / KS1;
    const int nrn = 16 * mt + (l & 15), k = 4 * s + (l >> 4);
    const float v = w1t[min(k, K0 - 1) * H + min(nrn, H - 1)];
    r1[it] = (nrn < H && k < K0) ? v : 0.0f;
  }
  const float rb1 = b1[min((int)threadIdx.x, H - 1)], rb2 = b2[min((int)threadIdx.x, H - 1)];
  const float rb3 = b3[threadIdx.x & 3];
#pragma unroll
  for (int it = 0; it < N2; ++it) {
    const int i = it * PMFMA_WG + threadIdx.x;  // (mo, mk, lane)
    const int l = i & 63, mk = (i >> 6) % MT, mo = (i >> 6) / MT;
    const int nrn = 16 * mo + (l & 15), k = 16 * mk + 4 * (l >> 4);
    const f4 v = *reinterpret_cast<const f4 *>(w2 + min(nrn, H - 1) * H + min(k, H - 4));
    r2[it] = (nrn < H && k < H) ? v : (f4){0.0f, 0.0f, 0.0f, 0.0f};
  }
#pragma unroll
  for (int it = 0; it < N3; ++it) {
    const int i = it * PMFMA_WG + threadIdx.x;  // (mk, lane, r)
    const int r = i & 3, l = (i >> 2) & 63, mk = i >> 8;
    const int a = l & 15, k = 16 * mk + 4 * (l >> 4) + r;
    const float v = w3t[min(k, H - 1) * 4 + (a & 3)];
    r3[it] = (a < 4 && k < H) ? v : 0.0f;
  }
#pragma unroll
  for (int it = 0; it < N1; ++it) {
    const int i = it * PMFMA_WG + threadIdx.x;
    if (i < G::W1) lw1[i] = r1[it];
  }
  static_assert(G::B <= PMFMA_WG, "one thread per padded bias entry");
  if (threadIdx.x < G::B) {
    lb1[threadIdx.x] = (int)threadIdx.x < H ? rb1 : 0.0f;
    lb2[threadIdx.x] = (int)threadIdx.x < H ? rb2 : 0.0f;
  }
  if (threadIdx.x < 16) lb3[threadIdx.x] = threadIdx.x < 4 ? rb3 : 0.0f;
  if (eps_ptr) eps = *eps_ptr;
  if (draw_ptr) draw = *draw_ptr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, grp = lane >> 4;
  if ((int64_t)blockIdx.x < n_tiles) tile_commit(tiles, blockIdx.x);
  __syncthreads();
  int buf = 0;
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int64_t env0 = t * PMFMA_ENVS;
    const int8_t *tile = tiles + buf * (PMFMA_ENVS * K0);
    const int64_t t_next = t + gridDim.x;
    if (dense && t_next < n_tiles) tile_fetch(t_next);
    const int wave_env = wave * NT * 16;  // first env of this wave inside the workgroup tile
    const int64_t env = env0 + wave_env + (lane & 31);  // the env whose action this lane (of lanes 0..31) picks at the end
    double u;
    uint32_t x2;
    draw_block<MODE>(env_base + (uint64_t)env, draw, seed, u, x2);  // VALU work in the shadow of the MFMAs below
    // ---- layer 1: h1^T = relu(W1 x^T + b1) --------------------------------------------------------------
    f4 h1[NT][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f4 bias = *reinterpret_cast<const f4 *>(lb1 + 16 * mt + 4 * grp);
#pragma unroll
      for (int e = 0; e < NT; ++e) h1[e][mt] = bias;
    }
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
      float x[NT];
      const int k = 4 * s + grp;
#pragma unroll
      for (int e = 0; e < NT; ++e) x[e] = (k < K0) ? (float)tile[(wave_env + 16 * e + col) * K0 + k] : 0.0f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float a = lw1[(mt * KS1 + s) * 64 + lane];
#pragma unroll
        for (int e = 0; e < NT; ++e) h1[e][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x[e], h1[e][mt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int e = 0; e < NT; ++e)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) h1[e][mt] = __builtin_elementwise_max(h1[e][mt], (f4){0.0f, 0.0f, 0.0f, 0.0f});
    if (t == (int64_t)blockIdx.x) {  // first pass: W2 / W3 have arrived by now
#pragma unroll
      for (int it = 0; it < N2; ++it) {
        const int i = it * PMFMA_WG + threadIdx.x;
        if (i < G::W2 / 4) reinterpret_cast<f4 *>(lw2)[i] = r2[it];
      }
#pragma unroll
      for (int it = 0; it < N3; ++it) {
        const int i = it * PMFMA_WG + threadIdx.x;
        if (i < G::W3) lw3[i] = r3[it];
      }
      __syncthreads();
    }
    // ---- layer 2: h2^T = relu(W2 h1^T + b2); k-step (mk, r) reads register r of h1 tile mk --------------------
    f4 h2[NT][MT];
#pragma unroll
    for (int mo = 0; mo < MT; ++mo) {
      const f4 bias = *reinterpret_cast<const f4 *>(lb2 + 16 * mo + 4 * grp);
#pragma unroll
      for (int e = 0; e < NT; ++e) h2[e][mo] = bias;
    }
#pragma unroll
    for (int mk = 0; mk < MT; ++mk) {
      f4 a[MT];
#pragma unroll
      for (int mo = 0; mo < MT; ++mo) a[mo] = *reinterpret_cast<const f4 *>(lw2 + ((mo * MT + mk) * 64 + lane) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mo = 0; mo < MT; ++mo)
#pragma unroll
          for (int e = 0; e < NT; ++e)
            h2[e][mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mo][r], h1[e][mk][r], h2[e][mo], 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < NT; ++e)
#pragma unroll
      for (int mo = 0; mo < MT; ++mo) h2[e][mo] = __builtin_elementwise_max(h2[e][mo], (f4){0.0f, 0.0f, 0.0f, 0.0f});
    // ---- layer 3: scores^T = W3 h2^T + b3; rows 0..3 of the tile = lanes 0..15, registers 0..3 ---------------
    f4 sc[NT];
    {
      const f4 bias = *reinterpret_cast<const f4 *>(lb3 + 4 * grp);
#pragma unroll
      for (int e = 0; e < NT; ++e) sc[e] = bias;
    }
#pragma unroll
    for (int mk = 0; mk < MT; ++mk) {
      const f4 a = *reinterpret_cast<const f4 *>(lw3 + (mk * 64 + lane) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < NT; ++e) sc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], h2[e][mk][r], sc[e], 0, 0, 0);
    }
    // lanes 0..15 hold the four scores of env tile e; bring tile 1 to lanes 16..31 so that 32 lanes pick 32 actions at once
    static_assert(NT == 2, "the epilogue pairs two env tiles");
    f4 mine;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float other = __shfl(sc[1][r], lane - 16, 64);
      mine[r] = grp == 1 ? other : sc[0][r];
    }
    if (lane < 32 && env < n) {
      actions[env] = (uint8_t)select_action<MODE>(mine[0], mine[1], mine[2], mine[3], u, x2, eps);
      if (scores_out) reinterpret_cast<float4 *>(scores_out)[env] = make_float4(mine[0], mine[1], mine[2], mine[3]);
    }
    if (t_next < n_tiles) tile_commit(tiles + (buf ^ 1) * (PMFMA_ENVS * K0), t_next);
    __syncthreads();  // the next tile is complete, and nobody still reads the one just used
    buf ^= 1;
  }
}

// PPOBaseAgent.get_discounted_returns (reference policy_base.py:179-186) for a batch of trajectories.
// The reference is an O(T^2) Python double loop per trajectory; its float32 rounding order is kept exactly:
//   d[t] = float32(discount ** t) * r[t];   returns[t] = ((d[t] + d[t+1]) + d[t+2]) + ...   (Python sum(): left to right)
// One wave per trajectory: the wave stages d[] in LDS, then lane t accumulates its own suffix serially -- consecutive
// lanes read consecutive LDS words at every iteration (conflict-free). gamma_pow[t] is computed on the host in double.
constexpr int RET_TMAX = 1024;
__global__ __launch_bounds__(WG) void discounted_returns_kernel(const float *__restrict__ rewards,
                                                                const int32_t *__restrict__ lengths,
                                                                const float *__restrict__ gamma_pow,
                                                                float *__restrict__ returns, int64_t n, int t_max) {
  __shared__ float d[WG / 64][RET_TMAX];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t traj = (int64_t)blockIdx.x * (WG / 64) + wave; traj < n; traj += (int64_t)gridDim.x * (WG / 64)) {
    const int len = lengths ? min(lengths[traj], t_max) : t_max;
    const float *r = rewards + traj * t_max;
    for (int t = lane; t < len; t += 64) d[wave][t] = __fmul_rn(gamma_pow[t], r[t]);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed
    for (int t0 = 0; t0 < len; t0 += 64) {
      const int t = t0 + lane;
      float acc = 0.0f;
      if (t < len) {
        acc = d[wave][t];
        for (int k = t + 1; k < len; ++k) acc = __fadd_rn(acc, d[wave][k]);
        returns[traj * t_max + t] = acc;
      }
    }
    __builtin_amdgcn_wave_barrier();
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
      if ((threadIdx.x & 63) >= off) incl += o;
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
// Tabular Q-learning, one private float64 table per env: Q[env][state][action] (reference value.py:15-58)
// ------------------------------------------------------------------------------------------------
template <int ENV>
__device__ __forceinline__ int state_index(const SgkRules &R, const EnvState &s) {
  return (ENV == SGK_SIDE_EFFECTS_SOKOBAN) ? s.pos * R.n_cells + s.box : s.pos;
}

// np.argmax: first maximum wins
__device__ __forceinline__ int argmax4(double q0, double q1, double q2, double q3) {
  int best = 0;
  double bv = q0;
  if (q1 > bv) { bv = q1; best = 1; }
  if (q2 > bv) { bv = q2; best = 2; }
  if (q3 > bv) { bv = q3; best = 3; }
  return best;
}
__device__ __forceinline__ double pick4(int k, double q0, double q1, double q2, double q3) {
  return k == 0 ? q0 : (k == 1 ? q1 : (k == 2 ? q2 : q3));
}

// numpy's 53-bit uniform from two 32-bit draws (random_sample)
__device__ __forceinline__ double uniform53(uint32_t a, uint32_t b) {
  return (double)((((uint64_t)(a >> 5)) << 26) + (uint64_t)(b >> 6)) / 9007199254740992.0;
}

// Stream 1 (epsilon-greedy draws): one block serves TWO consecutive agent steps of one env:
//   x = philox4x32_10(ctr = {env_lo, env_hi, (t >> 1)_lo, 1}, key); h = t & 1
//   u(t) = uniform53(x[2h], x[2h+1]),  explore action(t) = x[2h] & 3   (bits the 53-bit construction discards)
__device__ __forceinline__ void explore_block(uint64_t seed, uint64_t ge, int64_t t, uint32_t x[4]) {
  philox4x32_10((uint32_t)ge, (uint32_t)(ge >> 32), (uint32_t)((uint64_t)t >> 1), 1u, (uint32_t)seed, (uint32_t)(seed >> 32), x);
}
__device__ __forceinline__ void explore_draw(const uint32_t x[4], int64_t t, double &u, int &action) {
  const bool h = (t & 1) != 0;
  const uint32_t a = h ? x[2] : x[0], b = h ? x[3] : x[1];
  u = uniform53(a, b);
  action = (int)(a & 3u);
}

// epsilon in force at global agent step t (value.py:23-28,54-58): evaluated in Python's operation order
__host__ __device__ __forceinline__ double epsilon_at(double eps0, int64_t anneal, int64_t t) {
  if (t <= 0) return 0.0;
  if (t > anneal - 1) t = anneal - 1;
  if (t <= 0) return 0.0;
  double a = (1 - eps0) * (double)t;
  double b = a / (double)anneal;
  return 1.0 - b;
}
double host_epsilon_at(double eps0, int64_t anneal, int64_t t) { return epsilon_at(eps0, anneal, t); }

// Q <- Q + lr * ((r + discount * v_next) - Q), every operation rounded separately (no FMA contraction)
__device__ __forceinline__ double q_update(double q_sa, double reward, double v_next, double lr, double discount) {
  double target = __dadd_rn(reward, __dmul_rn(discount, v_next));
  double differential = __dsub_rn(target, q_sa);
  return __dadd_rn(q_sa, __dmul_rn(lr, differential));
}

struct TabqArgs {
  const SgkRules *rules;
  uint64_t *state;
  uint32_t *rec;
  int8_t *boards;
  int32_t *last_return, *last_perf, *n_episodes;
  long long *metrics;
  double *table;       // [n][n_states][4]
  uint16_t *s_prev;    // state index the last action was chosen from; 0xffff = env was over
  int64_t n;
  uint64_t seed, env_base;
  int64_t t_agent;     // global agent step (same for every agent: lockstep)
  double lr, discount, eps0;
  int64_t anneal;
  const double *eps_table;  // eps_table[t] for t < anneal (host-computed, bit-identical to the formula); may be null
  int32_t n_states;
  int32_t cheat;
  uint32_t flags;
};

template <int ENV>
__global__ __launch_bounds__(WG) void tabq_act_kernel(TabqArgs a, int explore, uint8_t *__restrict__ actions_out) {
  __shared__ SgkRules R;
  stage_rules(R, a.rules);
  const double eps = explore ? epsilon_at(a.eps0, a.anneal, a.t_agent) : 0.0;
  for (int64_t env = (int64_t)blockIdx.x * WG + threadIdx.x; env < a.n; env += (int64_t)gridDim.x * WG) {
    EnvState s = unpack_state(a.state[env]);
    int si = state_index<ENV>(R, s);
    const double2 *row = reinterpret_cast<const double2 *>(a.table + ((int64_t)env * a.n_states + si) * 4);
    double2 q01 = row[0], q23 = row[1];
    int action = argmax4(q01.x, q01.y, q23.x, q23.y);
    if (explore) {
      uint64_t ge = a.env_base + (uint64_t)env;
      uint32_t x[4];
      explore_block(a.seed, ge, a.t_agent, x);
      double u;
      int ea;
      explore_draw(x, a.t_agent, u, ea);
      if (u < eps) action = ea;
    }
    actions_out[env] = (uint8_t)action;
    a.s_prev[env] = s.over ? (uint16_t)0xffff : (uint16_t)si;
  }
}

template <int ENV>
__global__ __launch_bounds__(WG) void tabq_learn_kernel(TabqArgs a, const uint8_t *__restrict__ actions) {
  __shared__ SgkRules R;
  stage_rules(R, a.rules);
  for (int64_t env = (int64_t)blockIdx.x * WG + threadIdx.x; env < a.n; env += (int64_t)gridDim.x * WG) {
    int sp = a.s_prev[env];
    if (sp == 0xffff) continue;
    EnvState s = unpack_state(a.state[env]);
    uint32_t rec = a.rec[env];
    int action = a.cheat ? (int)(rec >> 24) : (int)(actions[env] & 3);
    double reward = a.cheat ? (double)(int8_t)(rec >> 8) : (double)(int8_t)rec;
    int sn = state_index<ENV>(R, s);
    double *tab = a.table + (int64_t)env * a.n_states * 4;
    const double2 *rown = reinterpret_cast<const double2 *>(tab + sn * 4);
    double2 n01 = rown[0], n23 = rown[1];
    int an = argmax4(n01.x, n01.y, n23.x, n23.y);
    double v_next = pick4(an, n01.x, n01.y, n23.x, n23.y);
    double *cell = tab + sp * 4 + action;
    *cell = q_update(*cell, reward, v_next, a.lr, a.discount);
  }
}

// Fused learning rollout: one wave = 64 private agents whose whole Q-tables live in LDS for the launch,
// lane-minor ([state*4+action][lane], 8-byte elements => lanes l and l+32 are served in different LDS
// passes and every lane hits its own bank pair: conflict-free for arbitrary per-lane states).
template <int ENV>
__global__ __launch_bounds__(64) void tabq_rollout_kernel(TabqArgs a, int64_t n_steps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  SgkRules &R = *reinterpret_cast<SgkRules *>(smem);
  double *Q = reinterpret_cast<double *>(smem + ((sizeof(SgkRules) + 15) / 16) * 16);
  stage_rules(R, a.rules);
  const int lane = threadIdx.x;
  const int S4 = a.n_states * 4;  // HBM row stride: tables are indexed by cell there
  const int L4 = R.n_slots * 4;   // LDS image: only the cells the agent can stand on
  const int64_t n_groups = (a.n + 63) / 64;
  EpisodeAcc acc;
  acc_init(acc);
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int64_t env0 = g * 64;
    const int64_t env = env0 + lane;
    const bool valid = env < a.n;
    const int n_here = (int)min((int64_t)64, a.n - env0);
    // load the 64 tables' reachable rows into the lane-minor LDS image
    {
      const double *src = a.table + env0 * S4;
      const int total = n_here * L4;
      for (int i = lane; i < total; i += 64) {
        int e = i / L4, idx = i - e * L4;
        Q[idx * 64 + e] = src[(int64_t)e * S4 + (int)R.slot_cell[idx >> 2] * 4 + (idx & 3)];
      }
    }
    __syncthreads();
    EnvState s = initial_state(R);
    if (valid) s = unpack_state(a.state[env]);
    const uint64_t ge = a.env_base + (uint64_t)env;
    int si = R.state_slot[s.pos];
    double q0 = Q[(si * 4 + 0) * 64 + lane], q1 = Q[(si * 4 + 1) * 64 + lane];
    double q2 = Q[(si * 4 + 2) * 64 + lane], q3 = Q[(si * 4 + 3) * 64 + lane];
    uint32_t rec = 0;
    uint32_t x[4] = {0, 0, 0, 0};
    for (int64_t k = 0; k < n_steps; ++k) {
      const int64_t t = a.t_agent + k;
      double eps;
      if (a.eps_table) {
        const int64_t tc = t < a.anneal ? t : a.anneal - 1;
        eps = a.eps_table[tc];  // wave-uniform address: one scalar load
      } else {
        eps = epsilon_at(a.eps0, a.anneal, t);
      }
      if (k == 0 || (t & 1) == 0) explore_block(a.seed, ge, t, x);
      double u;
      int ea;
      explore_draw(x, t, u, ea);
      int action = argmax4(q0, q1, q2, q3);
      if (u < eps) action = ea;
      // env.step
      bool finished = false;
      int r_obs = 0, r_hid = 0;
      const bool live = valid && !s.over;
      const int si_prev = si;
      if (live) {
        int term;
        uint32_t e = transition<ENV>(R, s, action, r_obs, r_hid, term);
        si = (int)(e >> 25);  // successor's slot straight from the transition word (no box in these levels)
        s.frame += 1;
        s.ret += r_obs;
        s.hid += r_hid;
        finished = term || s.frame >= R.max_iterations;
      }
      rec = pack_rec(r_obs, r_hid, (valid && (s.over || finished)) ? 1 : 0, action);
      // learn (no terminal masking: value.py:48-50 bootstraps from Q[s'] even when the episode ended)
      double n0 = Q[(si * 4 + 0) * 64 + lane], n1 = Q[(si * 4 + 1) * 64 + lane];
      double n2 = Q[(si * 4 + 2) * 64 + lane], n3 = Q[(si * 4 + 3) * 64 + lane];
      if (live) {
        int an = argmax4(n0, n1, n2, n3);
        double v_next = pick4(an, n0, n1, n2, n3);
        double reward = a.cheat ? (double)r_hid : (double)r_obs;
        double q_sa = pick4(action, q0, q1, q2, q3);
        double q_new = q_update(q_sa, reward, v_next, a.lr, a.discount);
        Q[(si_prev * 4 + action) * 64 + lane] = q_new;
        if (si == si_prev) {  // refused move: the successor row is the row just updated
          if (action == 0) n0 = q_new; else if (action == 1) n1 = q_new; else if (action == 2) n2 = q_new; else n3 = q_new;
        }
      }
      acc_add(acc, finished, s.ret, s.hid);
      if (finished) {  // train.py:62-70: the next episode starts from env.reset()
        a.last_return[env] = s.ret;
        a.last_perf[env] = s.hid;
        bump_episode_count(a.n_episodes, env);
        s = initial_state(R);
        si = R.state_slot[s.pos];
        n0 = Q[(si * 4 + 0) * 64 + lane]; n1 = Q[(si * 4 + 1) * 64 + lane];
        n2 = Q[(si * 4 + 2) * 64 + lane]; n3 = Q[(si * 4 + 3) * 64 + lane];
      }
      q0 = n0; q1 = n1; q2 = n2; q3 = n3;
    }
    if (valid) {
      a.state[env] = pack_state(s);
      a.rec[env] = rec;  // boards are re-materialised by the caller (launch_reset mode 2)
    }
    __syncthreads();
    {
      double *dst = a.table + env0 * S4;
      const int total = n_here * L4;
      for (int i = lane; i < total; i += 64) {
        int e = i / L4, idx = i - e * L4;
        dst[(int64_t)e * S4 + (int)R.slot_cell[idx >> 2] * 4 + (idx & 3)] = Q[idx * 64 + e];
      }
    }
    __syncthreads();
  }
  acc_flush(acc, a.metrics);
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
static int grid_for(int64_t n_tiles, int cap) { return (int)(n_tiles < cap ? (n_tiles < 1 ? 1 : n_tiles) : cap); }

#define SGK_DISPATCH_ENV_LAYOUT(ENVID, LAYOUT, ...)                                                     \
  do {                                                                                                     \
    if ((LAYOUT) == SGK_LAYOUT_COMPACT) {                                                                  \
      switch (ENVID) {                                                                                     \
      case SGK_BOAT_RACE: { constexpr int E = SGK_BOAT_RACE; constexpr int L = SGK_LAYOUT_COMPACT; __VA_ARGS__; } break;         \
      case SGK_ISLAND_NAVIGATION: { constexpr int E = SGK_ISLAND_NAVIGATION; constexpr int L = SGK_LAYOUT_COMPACT; __VA_ARGS__; } break; \
      case SGK_DISTRIBUTIONAL_SHIFT: { constexpr int E = SGK_DISTRIBUTIONAL_SHIFT; constexpr int L = SGK_LAYOUT_COMPACT; __VA_ARGS__; } break; \
      default: { constexpr int E = SGK_SIDE_EFFECTS_SOKOBAN; constexpr int L = SGK_LAYOUT_COMPACT; __VA_ARGS__; } break;         \
      }                                                                                                    \
    } else {                                                                                               \
      switch (ENVID) {                                                                                     \
      case SGK_BOAT_RACE: { constexpr int E = SGK_BOAT_RACE; constexpr int L = SGK_LAYOUT_PITCHED; __VA_ARGS__; } break;         \
      case SGK_ISLAND_NAVIGATION: { constexpr int E = SGK_ISLAND_NAVIGATION; constexpr int L = SGK_LAYOUT_PITCHED; __VA_ARGS__; } break; \
      case SGK_DISTRIBUTIONAL_SHIFT: { constexpr int E = SGK_DISTRIBUTIONAL_SHIFT; constexpr int L = SGK_LAYOUT_PITCHED; __VA_ARGS__; } break; \
      default: { constexpr int E = SGK_SIDE_EFFECTS_SOKOBAN; constexpr int L = SGK_LAYOUT_PITCHED; __VA_ARGS__; } break;         \
      }                                                                                                    \
    }                                                                                                      \
  } while (0)

#define SGK_DISPATCH_ENV(ENVID, ...)                                                  \
  do {                                                                                   \
    switch (ENVID) {                                                                     \
    case SGK_BOAT_RACE: { constexpr int E = SGK_BOAT_RACE; __VA_ARGS__; } break;               \
    case SGK_ISLAND_NAVIGATION: { constexpr int E = SGK_ISLAND_NAVIGATION; __VA_ARGS__; } break; \
    case SGK_DISTRIBUTIONAL_SHIFT: { constexpr int E = SGK_DISTRIBUTIONAL_SHIFT; __VA_ARGS__; } break; \
    default: { constexpr int E = SGK_SIDE_EFFECTS_SOKOBAN; __VA_ARGS__; } break;               \
    }                                                                                    \
  } while (0)

static StepArgs make_step_args(const Shard &sh, const uint8_t *actions, uint32_t flags) {
  StepArgs a;
  a.rules = sh.rules_dev;
  a.state = sh.state;
  a.actions = actions;
  a.rec = sh.rec;
  a.boards = sh.boards;
  a.last_return = sh.last_return;
  a.last_perf = sh.last_perf;
  a.n_episodes = sh.n_episodes;
  a.metrics = (long long *)sh.metric_slab;
  a.n = sh.n;
  a.seed = sh.seed;
  a.env_base = sh.env_base;
  a.t = sh.lockstep_t;
  a.t_ptr = nullptr;
  a.flags = flags;
  return a;
}

// A sub-range [env_off, env_off + count) of the shard as a Shard view (env_off must be a multiple of 256 so that
// board tiles stay aligned). Used to run independent partitions of the batch on concurrent graph branches.
static Shard shard_view(const Shard &sh, int64_t env_off, int64_t count) {
  Shard v = sh;
  v.n = count;
  v.env_base = sh.env_base + (uint64_t)env_off;
  v.state = sh.state + env_off;
  v.rec = sh.rec + env_off;
  v.boards = sh.boards + env_off * sh.pitch;
  v.last_return = sh.last_return + env_off;
  v.last_perf = sh.last_perf + env_off;
  v.n_episodes = sh.n_episodes + env_off;
  return v;
}

hipError_t launch_step_counter(const Shard &sh, const uint64_t *t_dev, uint64_t t_off, uint32_t flags, hipStream_t st,
                               int64_t env_off, int64_t count) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  Shard v = shard_view(sh, env_off, count);
  StepArgs a = make_step_args(v, nullptr, flags);
  a.t = t_off;
  a.t_ptr = t_dev;
  int grid = grid_for((v.n + WG - 1) / WG, v.max_grid);
  SGK_DISPATCH_ENV_LAYOUT(v.env_id, v.layout, step_kernel<E, L, true><<<dim3(grid), dim3(WG), 0, st>>>(a));
  return hipGetLastError();
}

hipError_t launch_step(const Shard &sh, const uint8_t *actions, uint32_t flags, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  StepArgs a = make_step_args(sh, actions, flags);
  int grid = grid_for((sh.n + WG - 1) / WG, sh.max_grid);
  if (actions) {
    SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout, step_kernel<E, L, false><<<dim3(grid), dim3(WG), 0, st>>>(a));
  } else {
    SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout, step_kernel<E, L, true><<<dim3(grid), dim3(WG), 0, st>>>(a));
  }
  return hipGetLastError();
}

hipError_t launch_rollout_random(const Shard &sh, int32_t n_steps, uint32_t flags, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  StepArgs a = make_step_args(sh, nullptr, flags);
  int grid = grid_for((sh.n + WG - 1) / WG, sh.max_grid);
  SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout,
                          rollout_random_kernel<E, L><<<dim3(grid), dim3(WG), 0, st>>>(a, n_steps));
  return hipGetLastError();
}

hipError_t launch_reset(const Shard &sh, const uint8_t *mask, int mode, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  int grid = grid_for((sh.n + WG - 1) / WG, sh.max_grid);
  SGK_DISPATCH_ENV_LAYOUT(sh.env_id, sh.layout,
                          reset_kernel<E, L><<<dim3(grid), dim3(WG), 0, st>>>(sh.rules_dev, sh.state,
                                              sh.boards, mask, mode, sh.n));
  return hipGetLastError();
}

hipError_t launch_metrics_init(const Shard &sh, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  metrics_init_kernel<<<dim3(32), dim3(WG), 0, st>>>((long long *)sh.metric_slab);
  return hipGetLastError();
}

hipError_t launch_metrics_reduce(const Shard &sh, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  metrics_reduce_kernel<<<dim3(1), dim3(1024), 0, st>>>((const long long *)sh.metric_slab, (long long *)sh.metrics);
  return hipGetLastError();
}

hipError_t launch_obs_f32(const Shard &sh, float *dst, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  int64_t total = sh.n * ((sh.n_cells + 3) / 4);
  int grid = grid_for((total + WG - 1) / WG, sh.max_grid * 4);
  hipLaunchKernelGGL(obs_f32_kernel, dim3(grid), dim3(WG), 0, st, sh.boards, dst, sh.n, sh.n_cells, sh.pitch);
  return hipGetLastError();
}

hipError_t launch_eps_greedy(const Shard &sh, int mode, const float *scores, uint8_t *actions, double eps, uint64_t draw,
                             const double *eps_dev, const uint64_t *draw_dev, hipStream_t st) {
  (void)hipGetLastError();
  int grid = grid_for((sh.n + WG - 1) / WG, sh.max_grid);
  const float4 *sc = reinterpret_cast<const float4 *>(scores);
  if (mode == 0)
    eps_greedy_kernel<0><<<dim3(grid), dim3(WG), 0, st>>>(sc, actions, sh.n, eps, sh.seed, sh.env_base, draw, eps_dev, draw_dev);
  else
    eps_greedy_kernel<1><<<dim3(grid), dim3(WG), 0, st>>>(sc, actions, sh.n, eps, sh.seed, sh.env_base, draw, eps_dev, draw_dev);
  return hipGetLastError();
}

hipError_t launch_policy_act(const Shard &sh, int mode, const PolicyWeights &w, uint8_t *actions, float *scores, double eps,
                             uint64_t draw, const double *eps_dev, const uint64_t *draw_dev, hipStream_t st) {
  (void)hipGetLastError();
  int grid = grid_for((sh.n + PMFMA_ENVS - 1) / PMFMA_ENVS, sh.n_cus);
#define SGK_POLICY_LAUNCH_M(K0, MODE)                                                                                      \
  do {                                                                                                                     \
    constexpr size_t lds = PolicyMfmaGeom<K0, 100>::lds_bytes;                                                             \
    static bool lds_opted_in = false; /* > 64 KB of dynamic LDS needs the opt-in, once per kernel */                       \
    if (!lds_opted_in) {                                                                                                   \
      hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&policy_mfma_kernel<K0, 100, MODE>),              \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
      if (ae != hipSuccess) return ae;                                                                                     \
      lds_opted_in = true;                                                                                                 \
    }                                                                                                                      \
    policy_mfma_kernel<K0, 100, MODE><<<dim3(grid), dim3(PMFMA_WG), lds, st>>>(sh.boards, sh.pitch, w.w1t, w.b1, w.w2, w.b2, \
                                                                              w.w3t, w.b3, actions, scores, sh.n, eps,     \
                                                                              sh.seed, sh.env_base, draw, eps_dev,         \
                                                                              draw_dev);                                   \
  } while (0)
#define SGK_POLICY_LAUNCH(K0)                                                                                              \
  do {                                                                                                                     \
    if (mode == 0) SGK_POLICY_LAUNCH_M(K0, 0);                                                                             \
    else SGK_POLICY_LAUNCH_M(K0, 1);                                                                                       \
  } while (0)
  if (w.n_hidden != 100) return hipErrorInvalidValue;
  switch (sh.n_cells) {
  case 25: SGK_POLICY_LAUNCH(25); break;
  case 36: SGK_POLICY_LAUNCH(36); break;
  case 48: SGK_POLICY_LAUNCH(48); break;
  case 63: SGK_POLICY_LAUNCH(63); break;
  default: return hipErrorInvalidValue;
  }
#undef SGK_POLICY_LAUNCH
#undef SGK_POLICY_LAUNCH_M
  return hipGetLastError();
}

hipError_t launch_discounted_returns(const Shard &sh, const float *rewards, const int32_t *lengths, const float *gamma_pow,
                                     float *returns, int64_t n, int t_max, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  int grid = grid_for((n + (WG / 64) - 1) / (WG / 64), sh.max_grid * 2);
  hipLaunchKernelGGL(discounted_returns_kernel, dim3(grid), dim3(WG), 0, st, rewards, lengths, gamma_pow, returns, n, t_max);
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

static TabqArgs make_tabq_args(const Shard &sh, const TabqShard &tq, uint32_t flags) {
  TabqArgs a;
  a.rules = sh.rules_dev;
  a.state = sh.state;
  a.rec = sh.rec;
  a.boards = sh.boards;
  a.last_return = sh.last_return;
  a.last_perf = sh.last_perf;
  a.n_episodes = sh.n_episodes;
  a.metrics = (long long *)sh.metric_slab;
  a.table = tq.table;
  a.s_prev = tq.s_prev;
  a.n = sh.n;
  a.seed = sh.seed;
  a.env_base = sh.env_base;
  a.t_agent = tq.t_agent;
  a.lr = tq.lr;
  a.discount = tq.discount;
  a.eps0 = tq.eps0;
  a.anneal = tq.anneal;
  a.eps_table = tq.eps_table;
  a.n_states = sh.n_states;
  a.cheat = 0;
  a.flags = flags;
  return a;
}

hipError_t launch_tabq_act(const Shard &sh, const TabqShard &tq, int explore, uint8_t *actions_out, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  TabqArgs a = make_tabq_args(sh, tq, 0);
  int grid = grid_for((sh.n + WG - 1) / WG, sh.max_grid);
  SGK_DISPATCH_ENV(sh.env_id, tabq_act_kernel<E><<<dim3(grid), dim3(WG), 0, st>>>(a, explore, actions_out));
  return hipGetLastError();
}

hipError_t launch_tabq_learn(const Shard &sh, const TabqShard &tq, const uint8_t *actions, int cheat, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  TabqArgs a = make_tabq_args(sh, tq, 0);
  a.cheat = cheat;
  int grid = grid_for((sh.n + WG - 1) / WG, sh.max_grid);
  SGK_DISPATCH_ENV(sh.env_id, tabq_learn_kernel<E><<<dim3(grid), dim3(WG), 0, st>>>(a, actions));
  return hipGetLastError();
}

// Sokoban's state is (agent cell, box cell): n_cells^2 rows do not fit LDS -> 0 = "use the per-step kernels"
size_t tabq_rollout_lds_bytes(const Shard &sh) {
  if (sh.n_states != sh.n_cells) return 0;
  return ((sizeof(SgkRules) + 15) / 16) * 16 + (size_t)sh.rules_host.n_slots * 4 * 64 * sizeof(double);
}

hipError_t launch_tabq_rollout(const Shard &sh, const TabqShard &tq, int64_t n_steps, int cheat, hipStream_t st) {
  (void)hipGetLastError();  // drop a stale error another HIP user of this thread may have left
  TabqArgs a = make_tabq_args(sh, tq, 0);
  a.cheat = cheat;
  size_t lds = tabq_rollout_lds_bytes(sh);
  int64_t n_groups = (sh.n + 63) / 64;
  int per_cu = (int)((160u * 1024u) / lds);  // workgroups (= waves) the LDS lets a CU hold
  if (per_cu > 16) per_cu = 16;
  if (per_cu < 1) per_cu = 1;
  int grid = grid_for(n_groups, sh.n_cus * per_cu);
  hipError_t err = hipSuccess;
  SGK_DISPATCH_ENV(sh.env_id, {
    err = hipFuncSetAttribute(reinterpret_cast<const void *>(&tabq_rollout_kernel<E>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err == hipSuccess)
      hipLaunchKernelGGL((tabq_rollout_kernel<E>), dim3(grid), dim3(64), lds, st, a, n_steps);
  });
  if (err != hipSuccess) return err;
  return hipGetLastError();
}

}  // namespace sgk
