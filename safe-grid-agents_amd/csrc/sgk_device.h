// sgk_device.h -- device-side building blocks shared by the kernel files (sgk_step.hip, sgk_policy.hip, sgk_tabq.hip):
// the packed env state word, the counter RNG, the table-driven transition, the episode-metrics accumulators and the two
// board writers. Everything here is a template or __forceinline__; the file is included, never compiled on its own.
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
//  * episode ends: per-lane accumulators -> wave reduction -> one vector atomic per wave on the workgroup's slab slot.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "sgk_kernels.h"
#include "sgk_transition.h"  // EnvState, the counter RNG, transition<ENV>, begin_episode<ENV>, env_actual_action<ENV>


namespace sgk {

constexpr int WG = 256;

// The wave's index inside its workgroup as a value the compiler KNOWS to be wave-uniform. `threadIdx.x >> 6` is uniform but not
// provably so: everything derived from it (tile index, tile pointer, the buffer descriptor of the tile stores) then lives in
// VGPRs, and hipcc wraps every buffer store in a "waterfall" loop (v_readfirstlane x 4, compare, s_and_saveexec, the store,
// loop; cdna_hip_programming.md T20) -- ten extra instructions per store and no overlap between consecutive stores.
__device__ __forceinline__ int wave_index() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Kernel arguments the compiler would fetch lazily (an s_load where they are first used -- for a latency-bound kernel another
// memory round trip behind the first one): naming them as scalar-register inputs of an empty asm at kernel entry makes their
// loads part of the entry block's batch.
template <class P>
__device__ __forceinline__ int keep_in_sgpr(P p) {
  asm volatile("" ::"s"(p));
  return 0;
}
template <class... P>
__device__ __forceinline__ void keep_in_sgprs(P... p) {
  (void)(keep_in_sgpr(p) + ...);
}

// workgroup-cooperative copy of the rule tables HBM/L2 -> LDS
__device__ __forceinline__ void stage_rules(SgkRules &dst, const SgkRules *__restrict__ src) {
  constexpr int NW = sizeof(SgkRules) / 4;
  const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
  uint32_t *d = reinterpret_cast<uint32_t *>(&dst);
  for (int i = threadIdx.x; i < NW; i += blockDim.x) d[i] = s[i];
  __syncthreads();
}

// The rule tables as ONE WAVE's private LDS copy (the per-step kernel, round 5): every lane requests its 16-byte pieces of the
// table right at kernel entry -- next to the tile's state words, so all of a wave's global loads are in flight together -- and
// writes them to the wave's own image. LDS operations of one wave execute in issue order: no workgroup barrier anywhere (the
// barrier-staged form above makes the four waves of a workgroup wait for the slowest one's loads twice per launch; measured
// on synthetic kernels with the step's memory shape, tools/exp_step_latency.hip: 2.69 -> 2.27 us per launch at 65 536 envs,
// 8.6 -> 7.1 at 1 M, 2.42 -> 2.0 at 1 024). The device buffer behind Shard::rules_dev is SGK_RULES_DEV_BYTES long: the
// table padded to SGK_RULES_IMAGE_BYTES, then a blank 64-env COMPACT tile (the level's backdrop 64 times over, n_cells bytes
// each) that the tile writer starts from.
constexpr int SGK_RULES_CHUNKS = (int)((sizeof(SgkRules) + 15) / 16);
constexpr int SGK_RULES_ITS = (SGK_RULES_CHUNKS + 63) / 64;
static_assert(sizeof(SgkRules) <= SGK_RULES_IMAGE_BYTES && SGK_RULES_ITS * 64 * 16 <= SGK_RULES_IMAGE_BYTES, "rule table image");
typedef uint32_t sgk_rules_u32x4 __attribute__((ext_vector_type(4)));
union alignas(16) WaveRulesImage {
  SgkRules r;
  sgk_rules_u32x4 q[SGK_RULES_IMAGE_BYTES / 16];
};
struct WaveRulesLoad {
  sgk_rules_u32x4 v[SGK_RULES_ITS];
  // request this lane's pieces (no wait): a raw buffer load, out-of-range pieces read as zeros
  __device__ __forceinline__ void request(const SgkRules *__restrict__ src) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, SGK_RULES_IMAGE_BYTES, 0x00020000);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < SGK_RULES_ITS; ++it) v[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (lane + 64 * it) * 16, 0, 0);
  }
  // the pieces into the wave's image; the wave's later LDS reads come after these writes in issue order
  __device__ __forceinline__ void commit(WaveRulesImage &img) const {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < SGK_RULES_ITS; ++it) img.q[lane + 64 * it] = v[it];
    __builtin_amdgcn_wave_barrier();
  }
};

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
// n_resets[env] += 1, kept only for the envs whose own draws are keyed by it (HasEnvDraws): every reset -- explicit, masked,
// reset_done, or the auto-reset inside a step -- starts a new draw sequence, so an episode cut short by reset() does not
// replay the draws of the one before it
template <int ENV>
__device__ __forceinline__ void bump_reset_count(int32_t *__restrict__ n_resets, int64_t env) {
  if (HasEnvDraws<ENV>::value) (void)__hip_atomic_fetch_add(&n_resets[env], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// The per-step kernel's episode metrics: the lanes that finished an episode ADD THEIR OWN figures to nine wave-private LDS words
// (ds_add / ds_max: a handful of lanes, nine instructions, nothing to wait for) and the flush reads the words back -- instead of
// nine wavefront reductions over all 64 lanes. It matters because the flush sits on the launch's critical path, at one wave per
// SIMD: in a level whose episodes end all the time (IslandNavigation under random actions: 7 % of the envs per step) every wave
// runs it on every launch. Measured at 65 536 IslandNavigation envs, us per launch (tools/exp_island_step.py,
// profiles/r05/island_step.log): every env idle 2.73; episodes ending, reductions (64-bit sums) at the end of the kernel 4.09;
// int32 reductions issued ahead of the stores 3.78 (3.20 with the flush switched off, 0.07 of the rest being the per-env episode
// arrays); this form 3.53 on a box that runs 0.1 slower. The two global atomics the flush ends with are NOT the cost: a variant
// that owned its slot and used plain stores measured the same.
// The words are 32-bit (ds_add_u32): a wave's sums over ONE launch must stay below 2^31 -- tiles per wave x 64 lanes x the largest
// |episode return| in reward units. The per-step kernels book at most one episode per env and launch: at the largest batch the
// library accepts (2^31 - 512 envs over the 6 144 waves of a full grid = 5 462 tiles per wave) that is 3.5 x 10^5 episodes per
// wave, times at most 1 300 units (tomato watering: 13 tomatoes x 100 steps; the other levels stay within +-150) = 4.5 x 10^8.
// A level whose returns could exceed ~6 000 units would have to flush a wave's words more often (flush() is 64-bit from there on).
struct WaveEpisodeLds {
  int *w;  // [16] wave-private: 0 s_ret, 1 s_perf, 2 s_mpos, 3 n_eps, 4 n_pos, 5 m_ret, 6 m_perf, 7 m_margin, 8 m_mpos
  bool used;  // wave-uniform: some lane of this wave has finished an episode in this launch (the words are initialised then)
  __device__ __forceinline__ void bind(int *words) {
    w = words;
    used = false;
  }
  // called by every lane; the lanes whose env finished an episode on this step contribute
  // (raw ds_add_u32 / ds_max_i32: through atomicAdd() the compiler's atomic optimiser turns every same-address atomic into a
  // wavefront reduction plus a one-lane atomic -- the nine reductions this exists to avoid, and then some: 3.8 -> 6.0 us)
  // A wave in which nobody finished pays one ballot and one scalar branch: in BoatRace that is 99 launches of 100 (an earlier form
  // that initialised the words at entry and read them back in every flush cost that level 0.16 us per launch at 65 536 envs).
  __device__ __forceinline__ void add(bool finished, int ret, int perf) {
    if (__ballot(finished) == 0ull) return;  // wave-uniform
    if (!used) {
      const int c = threadIdx.x & 63;
      if (c < 16) w[c] = c >= 5 ? ACC_NEG : 0;
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the raw atomics below are ordered behind these stores by issue order; the
                                                          // wait keeps the compiler's own counting honest
      used = true;
    }
    if (finished) {
      const int margin = ret - perf, one = 1;
      const uint32_t base = (uint32_t)(uintptr_t)w;  // the words' LDS byte address
      asm volatile("ds_add_u32 %0, %1\n ds_add_u32 %0, %2 offset:4\n ds_add_u32 %0, %3 offset:12\n"
                   "ds_max_i32 %0, %1 offset:20\n ds_max_i32 %0, %2 offset:24\n ds_max_i32 %0, %4 offset:28"
                   :: "v"(base), "v"(ret), "v"(perf), "v"(one), "v"(margin) : "memory");
      if (margin > 0)
        asm volatile("ds_add_u32 %0, %1 offset:8\n ds_add_u32 %0, %2 offset:16\n ds_max_i32 %0, %1 offset:32"
                     :: "v"(base), "v"(margin), "v"(one) : "memory");
    }
  }
  // all 64 lanes; the words go to the workgroup's slot of the metrics slab (one vector atomic for the sums, one for the maxima)
  __device__ __forceinline__ void flush(long long *__restrict__ slab) const {
    if (!used) return;  // wave-uniform
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the raw LDS atomics above are invisible to the compiler's own counting
    __builtin_amdgcn_wave_barrier();
    const int c = threadIdx.x & 63;
    const int n_eps = w[3];  // (the same word in every lane: an LDS broadcast read)
    long long *slot = slab + (size_t)(blockIdx.x % SGK_METRIC_SLOTS) * SGK_METRICS_LEN;
    if (c < 6) {
      const int v = c == SGK_M_SUM_RETURN ? w[0] : c == SGK_M_SUM_SAFETY ? w[1] : c == SGK_M_SUM_MARGIN ? w[0] - w[1]
                  : c == SGK_M_SUM_MARGIN_POS ? w[2] : c == SGK_M_EPISODES ? n_eps : w[4];
      atomicAdd((unsigned long long *)&slot[c], (unsigned long long)(long long)v);
    } else if (c >= SGK_M_MAX_RETURN && c <= SGK_M_MAX_MARGIN_POS) {
      const int m = w[5 + (c - SGK_M_MAX_RETURN)];
      if (c != SGK_M_MAX_MARGIN_POS || w[4] > 0) atomicMax(&slot[c], (long long)m);
    }
  }
};

// ------------------------------------------------------------------------------------------------
// streaming stores for the COMPACT board tiles: written once per step, 1 KiB contiguous per wave-instruction, never read
// back by these kernels. `sc1` buffer stores are written through and DROPPED from the XCD's L2 (MI355X_MICROARCH.md, stores
// table), so the 25-36 B/env of board bytes do not evict the 8 B/env state words the same workgroup re-reads in the next
// launch. Measured at 1M BoatRace envs: 12.3 -> 10.8 us per step. (For the PITCHED layout -- 16-byte pieces at a 32/48-byte
// stride -- write-through costs partial-line fabric writes: 25 -> 36 us on IslandNavigation; it keeps plain stores. Dword
// sc1 stores cost one fabric write each, so the step records keep plain stores too.)
// ------------------------------------------------------------------------------------------------
typedef uint32_t sgk_u32x4 __attribute__((ext_vector_type(4)));
// cache policy of the board-tile stores (raw buffer store aux bits on gfx950: 1 = sc0, 2 = nt, 16 = sc1)
constexpr int BOARD_STORE_AUX = 16;  // write-through
constexpr int RING_STORE_AUX = 18;   // trajectory rings: write-through + non-temporal

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
  const uint32_t *t32 = reinterpret_cast<const uint32_t *>(HasAltBackdrop<ENV>::value ? backdrop_of(R, alt_backdrop<ENV>(R, s)) : R.templ);
#pragma unroll
  for (int k = 0; k < NW; ++k) w[k] = t32[k];  // wave-uniform LDS broadcast reads (per-lane choice of two with two backdrops)
  if (HasSprite2<ENV>::value) {  // a `box` of 255 (whisky drunk) matches no word
    int bk = s.box >> 2, bsh = (s.box & 3) * 8;
#pragma unroll
    for (int k = 0; k < NW; ++k) w[k] = (k == bk) ? poke_byte(w[k], bsh, (uint32_t)sprite2_value<ENV>(R, s)) : w[k];
  }
  if (HasMask<ENV>::value && !alt_backdrop<ENV>(R, s)) {  // the watered tomatoes (on the bucket the alt backdrop shows them all)
    const uint32_t mask = (uint32_t)s.box | ((uint32_t)s.ext << 8);
    for (int t = 0; t < R.n_tomatoes; ++t) {
      const int c = R.tomato_cell[t], tk = c >> 2, tsh = (c & 3) * 8;
      const bool on = (mask >> t) & 1u;
#pragma unroll
      for (int k = 0; k < NW; ++k) w[k] = (on && k == tk) ? poke_byte(w[k], tsh, (uint32_t)R.value_box) : w[k];
    }
  }
  int ak = s.pos >> 2, ash = (s.pos & 3) * 8;
  uint32_t aval = R.agent_value[s.pos];
#pragma unroll
  for (int k = 0; k < NW; ++k) w[k] = (k == ak) ? poke_byte(w[k], ash, aval) : w[k];
  uint4 *dst = reinterpret_cast<uint4 *>(boards + env * PITCH);
#pragma unroll
  for (int q = 0; q < NW / 4; ++q) dst[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}

// COMPACT: rows of exactly NC bytes, written tile-wise as 16-byte chunks so that each store instruction covers up to 1 KiB of
// contiguous memory. A chunk is the backdrop rotated to the chunk's phase (tabulated once per workgroup in LDS:
// rot[r][b] = templ[(r + b) % NC]) with the agent / second-sprite cells of the (at most two, NC >= 16) envs it overlaps poked
// in (WaveTileLds below).
template <int NC>
struct alignas(16) CompactLds {  // rot rows are read with ds_read_b128
  uint8_t rot[NC][16];
  uint8_t rot_alt[NC][16];  // the same for templ_alt (envs with two backdrops; unused elsewhere)
};

template <int NC>
__device__ __forceinline__ void stage_rotations(CompactLds<NC> &C, const SgkRules &R) {
  for (int i = threadIdx.x; i < NC * 16; i += blockDim.x) {
    int r = i >> 4, b = i & 15;
    C.rot[r][b] = R.templ[(r + b) % NC];
    if (R.env_id == SGK_ABSENT_SUPERVISOR || R.env_id == SGK_SAFE_INTERRUPTIBILITY || R.env_id == SGK_TOMATO_WATERING) C.rot_alt[r][b] = R.templ_alt[(r + b) % NC];  // workgroup-uniform
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// WAVE-PRIVATE compact tiles. One wave = 64 consecutive envs = 64 * NC contiguous bytes (a multiple of 16), written as 4 * NC
// 16-byte chunks, lane l taking chunks l, l + 64, ...: every store instruction covers up to 1 KiB of contiguous HBM and a wave
// assembles its OWN tile, so there is no workgroup barrier per tile. What a lane contributes is one packed word: agent cell |
// second sprite cell << 8 | value drawn at the agent's cell << 16 | (board shows templ_alt) << 24 | value drawn at the second
// sprite's cell << 25 (3 bits).
// ------------------------------------------------------------------------------------------------
// (Levels with a mask of two-valued cells -- tomato watering -- put the mask's bits 8..12 where the sprite value goes: the mask
// is (info >> 8 & 0xff) | (info >> 25 & 0x1f) << 8.)
template <int ENV>
__device__ __forceinline__ uint32_t sprite_info(const SgkRules &R, const EnvState &s) {
  return (uint32_t)s.pos | ((uint32_t)s.box << 8) | ((uint32_t)R.agent_value[s.pos] << 16) |
         (HasAltBackdrop<ENV>::value ? ((((uint32_t)alt_backdrop<ENV>(R, s) & 1u) << 24) | (((uint32_t)alt_backdrop<ENV>(R, s) >> 1) << 30)) : 0u) |
         (HasSprite2<ENV>::value ? ((uint32_t)sprite2_value<ENV>(R, s) << 25) : 0u) |
         (HasMask<ENV>::value ? ((uint32_t)s.ext << 25) : 0u);
}
// which backdrop the packed sprite word names: bit 24 | bit 30 << 1 (the third one exists for friend or foe only)
__device__ __forceinline__ int info_backdrop(uint32_t info) { return (int)(((info >> 24) & 1u) | (((info >> 30) & 1u) << 1)); }
__device__ __forceinline__ uint32_t info_mask(uint32_t info) { return ((info >> 8) & 0xffu) | (((info >> 25) & 0x1fu) << 8); }

// ------------------------------------------------------------------------------------------------
// The wave's tile as an IMAGE IN LDS (64 * NC bytes per wave). A lane owns row `lane` of the image and
// (re)draws it with byte stores -- backdrop, second sprite, agent on top --; then the wave reads the image back as 16-byte chunks
// and streams them to HBM (up to 1 KiB of contiguous memory per store instruction, as above). LDS operations of one wave execute
// in issue order, so no barrier is needed (a compiler-only wave barrier keeps the order in the instruction stream). A kernel that
// keeps the image across steps (the streaming rollout) re-draws only the cells a step changed -- two byte stores per sprite. (Round
// 2's register / ds_bpermute assembly of the chunks took ~110 instructions of board work per lane and step against ~15 here and
// was VALU-issue-bound: profiles/r02/01_stream_v1_*; removed in round 4.)
// ------------------------------------------------------------------------------------------------
template <int ENV, int NC>
struct WaveTileLds {
  static constexpr int BYTES = 64 * NC, CHUNKS = 4 * NC, ITS = (CHUNKS + 63) / 64;
  static constexpr bool ALT = HasAltBackdrop<ENV>::value;
  uint8_t *tile;  // this wave's image: tile[lane * NC + cell]

  __device__ __forceinline__ void bind(uint8_t *wave_tile) { tile = wave_tile; }

  // a whole row from one of the two backdrops: 16-byte LDS stores where the rows are 16-byte aligned, bytes elsewhere (rare: a
  // row is re-drawn from scratch only when its backdrop changes)
  __device__ __forceinline__ void copy_row(uint8_t *row, const uint8_t *t) const {
    if (NC % 16 == 0) {
#pragma unroll
      for (int q = 0; q < NC / 16; ++q) *reinterpret_cast<uint4 *>(row + 16 * q) = *reinterpret_cast<const uint4 *>(t + 16 * q);
    } else {
      for (int c = 0; c < NC; ++c) row[c] = t[c];
    }
  }

  // draw every row from scratch: the backdrop (chunk-wise from the rotation table; row-wise where each env picks one of two),
  // then the sprites of `info` (sprite_info of this lane's env)
  __device__ __forceinline__ void draw_all(const CompactLds<NC> &C, const SgkRules &R, uint32_t info) const {
    const int lane = threadIdx.x & 63;
    if (ALT) {
      copy_row(tile + lane * NC, backdrop_of(R, info_backdrop(info)));
    } else {
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        const int j = lane + 64 * it;
        if (ITS * 64 == CHUNKS || j < CHUNKS) {
          const int r = (16 * j) % NC;
          *reinterpret_cast<uint4 *>(tile + 16 * j) = *reinterpret_cast<const uint4 *>(&C.rot[r][0]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    poke(R, info);
  }

  // The barrier-free form of draw_all (the per-step kernel): the blank tile -- the backdrop 64 times over, precomputed on the host
  // behind the rule table (SGK_RULES_IMAGE_BYTES) -- is REQUESTED at kernel entry as 16-byte pieces, one or more per lane, and
  // written to the image when a tile is drawn; levels whose envs pick one of several backdrops copy their row from the wave's
  // rule image instead. No rotation table, no workgroup barrier.
  struct Blank {
    sgk_rules_u32x4 v[ITS];
  };
  __device__ __forceinline__ void request_blank(Blank &b, const SgkRules *__restrict__ rules_dev) const {
    if (ALT) return;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)(reinterpret_cast<const uint8_t *>(rules_dev) + SGK_RULES_IMAGE_BYTES), 0, BYTES, 0x00020000);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < ITS; ++it) b.v[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (lane + 64 * it) * 16, 0, 0);
  }
  // Name the blank tile's registers as used HERE: the wait for their loads then sits where the caller puts this -- before its first
  // store. Loads and stores share one counter on gfx950 (vmcnt): a wait for these loads placed behind the step's state / record
  // stores is also a wait for those stores' trip to memory, in the middle of the kernel.
  __device__ __forceinline__ void blank_arrived(Blank &b) const {
    if (ALT) return;
#pragma unroll
    for (int it = 0; it < ITS; ++it) asm volatile("" : "+v"(b.v[it]));
  }
  __device__ __forceinline__ void draw_from_blank(const Blank &b, const SgkRules &R, uint32_t info) const {
    const int lane = threadIdx.x & 63;
    if (ALT) {
      copy_row(tile + lane * NC, backdrop_of(R, info_backdrop(info)));
    } else {
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        const int j = lane + 64 * it;
        if (ITS * 64 == CHUNKS || j < CHUNKS) *reinterpret_cast<sgk_rules_u32x4 *>(tile + 16 * j) = b.v[it];
      }
    }
    __builtin_amdgcn_wave_barrier();
    poke(R, info);
  }

  // the sprites of this lane's env onto its row: the second sprite first, the agent on top
  __device__ __forceinline__ void poke(const SgkRules &R, uint32_t info) const {
    uint8_t *row = tile + (threadIdx.x & 63) * NC;
    if (HasSprite2<ENV>::value) {
      const int box = (info >> 8) & 0xffu;
      if (box < NC) row[box] = (uint8_t)((info >> 25) & 7u);  // 255: the whisky is drunk / the interruption tile is gone
    }
    if (HasMask<ENV>::value && !((info >> 24) & 1u)) {  // the watered tomatoes over the (all dry) backdrop
      const uint32_t mask = info_mask(info);
      for (int t = 0; t < R.n_tomatoes; ++t)
        if ((mask >> t) & 1u) row[R.tomato_cell[t]] = (uint8_t)R.value_box;
    }
    row[info & 0xffu] = (uint8_t)(info >> 16);
  }

  // a step changed this lane's env from `was` to `now`: re-draw what differs (the whole row when the backdrop changed)
  __device__ __forceinline__ void update(const SgkRules &R, uint32_t was, uint32_t now) const {
    uint8_t *row = tile + (threadIdx.x & 63) * NC;
    if (ALT && info_backdrop(was) != info_backdrop(now)) {
      copy_row(row, backdrop_of(R, info_backdrop(now)));
    } else if (HasMask<ENV>::value) {
      // a level whose cells change by themselves: only the tomatoes that changed state (watered by the agent, dried by their
      // draw: well under one per step) and the cell the agent left are re-drawn; on the bucket the board shows none of it
      if (was != now) {
        const bool alt = (now >> 24) & 1u;
        const uint8_t *t = alt ? R.templ_alt : R.templ;
        const uint32_t m = info_mask(now);
        const int pos = was & 0xffu;
        uint8_t under = t[pos];
        if (!alt) {
          uint32_t changed = info_mask(was) ^ m;
          while (changed) {
            const int k = __ffs((int)changed) - 1;
            changed &= changed - 1u;
            const int c = R.tomato_cell[k];
            row[c] = ((m >> k) & 1u) ? (uint8_t)R.value_box : t[c];
          }
          const uint32_t ti = R.tomato_index[pos];
          if (ti != 255u && ((m >> ti) & 1u)) under = (uint8_t)R.value_box;
        }
        row[pos] = under;
        __builtin_amdgcn_wave_barrier();
        row[now & 0xffu] = (uint8_t)(now >> 16);
      }
      return;
    } else if (was != now) {
      const uint8_t *t = ALT ? backdrop_of(R, info_backdrop(now)) : R.templ;
      const int pos = was & 0xffu;
      row[pos] = t[pos];
      if (HasSprite2<ENV>::value) {
        const int box = (was >> 8) & 0xffu;
        if (box < NC) row[box] = t[box];
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (was != now) poke(R, now);
  }

  // image -> HBM: all 64 * NC bytes, `dst` wave-uniform and 16-byte aligned. AUX = the stores' cache policy: write-through
  // (sc1) for buffers that are rewritten in place; write-through + non-temporal for a trajectory ring, whose bytes nothing
  // on the chip reads again (the pure store probe gains 2.6-3.6 % with it on every box: profiles/r03/write_patterns_*.log)
  template <int AUX = BOARD_STORE_AUX>
  __device__ __forceinline__ void flush(int8_t *dst) const {
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_wave_barrier();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, BYTES, 0x00020000);
    // All the LDS reads first, then the stores: one LDS round trip per flush, not one per 16-byte chunk. No lane is predicated:
    // a lane past the image's last chunk reads the last chunk again and its store is dropped by the buffer's range check
    // (num_records = BYTES) -- a predicated second chunk made the compiler merge the stores behind a waterfall loop.
    uint4 v[ITS];
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const int j = lane + 64 * it;
      const int jr = (ITS * 64 == CHUNKS || j < CHUNKS) ? j : CHUNKS - 1;
      v[it] = *reinterpret_cast<const uint4 *>(tile + 16 * jr);
    }
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      sgk_u32x4 v4 = {v[it].x, v[it].y, v[it].z, v[it].w};
      __builtin_amdgcn_raw_buffer_store_b128(v4, rsrc, (lane + 64 * it) * 16, 0, AUX);
    }
    __builtin_amdgcn_wave_barrier();  // the next re-draw comes after these reads in the instruction stream
  }

  // the first n_bytes of the image only (a batch smaller than a tile whose destination is not padded HBM but host memory)
  __device__ __forceinline__ void flush_prefix(int8_t *dst, int n_bytes) const {
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const int j = lane + 64 * it;
      if (j * 16 < n_bytes) reinterpret_cast<uint4 *>(dst)[j] = *reinterpret_cast<const uint4 *>(tile + 16 * j);
    }
    __builtin_amdgcn_wave_barrier();
  }

  // one-shot form (per-launch kernels): draw, store
  __device__ __forceinline__ void write(const CompactLds<NC> &C, const SgkRules &R, uint32_t info, int8_t *dst) const {
    draw_all(C, R, info);
    flush(dst);
  }
};

// one env's NC-byte row written byte by byte from its own state: the slow path for destinations the tile writer cannot take
// (a trajectory slice whose last tile is partial or whose rows are not 16-byte aligned)
template <int ENV, int NC>
__device__ __noinline__ void write_row_bytes(const SgkRules &R, int8_t *__restrict__ row, const EnvState &s) {
  const uint8_t *backdrop = HasAltBackdrop<ENV>::value ? backdrop_of(R, alt_backdrop<ENV>(R, s)) : R.templ;
  const bool shows_mask = HasMask<ENV>::value && !alt_backdrop<ENV>(R, s);
  const uint32_t mask = (uint32_t)s.box | ((uint32_t)s.ext << 8);
  for (int c = 0; c < NC; ++c) {
    uint8_t v = backdrop[c];
    if (HasSprite2<ENV>::value && c == s.box) v = (uint8_t)sprite2_value<ENV>(R, s);
    if (shows_mask && R.tomato_index[c] != 255 && ((mask >> R.tomato_index[c]) & 1u)) v = (uint8_t)R.value_box;
    if (c == s.pos) v = R.agent_value[c];
    row[c] = (int8_t)v;
  }
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
template <>
struct Geom<SGK_WHISKY_GOLD> { static constexpr int NC = 48, PITCH = 48; };
template <>
struct Geom<SGK_ABSENT_SUPERVISOR> { static constexpr int NC = 48, PITCH = 48; };
template <>
struct Geom<SGK_SAFE_INTERRUPTIBILITY> { static constexpr int NC = 56, PITCH = 64; };
template <>
struct Geom<SGK_CONVEYOR_BELT> { static constexpr int NC = 49, PITCH = 64; };
template <>
struct Geom<SGK_TOMATO_WATERING> { static constexpr int NC = 63, PITCH = 64; };
template <>
struct Geom<SGK_FRIEND_FOE> { static constexpr int NC = 30, PITCH = 32; };

// numpy's 53-bit uniform from two 32-bit draws (random_sample)
__device__ __forceinline__ double uniform53(uint32_t a, uint32_t b) {
  return (double)((((uint64_t)(a >> 5)) << 26) + (uint64_t)(b >> 6)) / 9007199254740992.0;
}

// ------------------------------------------------------------------------------------------------
// one env.step for one lane, shared by the step kernel and the fused policy rollout
// ------------------------------------------------------------------------------------------------
struct StepArgs {
  const SgkRules *rules;
  uint64_t *state;
  const uint8_t *actions;  // nullptr in RANDOM mode
  uint32_t *rec;
  int8_t *boards;
  int32_t *last_return, *last_perf, *n_episodes, *n_resets;
  double *aux;                 // [n][SGK_AUX_DOUBLES] float64 side state that outlives episodes (HasAux levels; nullptr elsewhere)
  long long *metrics;
  int64_t n;
  uint64_t seed, env_base, t;  // t = lockstep step index (RANDOM mode RNG key) ...
  const uint64_t *t_ptr;       // ... or, when non-null (hipGraph replays), *t_ptr + t
  uint32_t flags;
};

// this env's slice of the side state
template <int ENV>
__device__ __forceinline__ double *aux_of(double *aux, int64_t env) {
  return HasAux<ENV>::value ? aux + env * SGK_AUX_DOUBLES : nullptr;
}

template <int ENV>
__device__ __forceinline__ void step_one(const SgkRules &R, const StepArgs &a, int64_t env, bool valid, int action,
                                         EnvState &s, uint32_t &rec, EpisodeAcc &acc) {
  bool finished = false;
  int r_obs = 0, r_hid = 0;
  if (valid && !s.over) {
    int term;
    action = env_actual_action<ENV>(R, s, a.seed, a.env_base + (uint64_t)env, action);  // what the env executes (whisky)
    transition<ENV>(R, s, action, r_obs, r_hid, term, aux_of<ENV>(a.aux, env));
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
    if (a.flags & SGK_F_AUTO_RESET) {
      const int epi = s.epi + 1;  // this reset's index
      bump_reset_count<ENV>(a.n_resets, env);
      s = initial_state(R);
      s.epi = epi;
      begin_episode<ENV>(R, s, a.seed, a.env_base + (uint64_t)env, aux_of<ENV>(a.aux, env));
    } else {
      s.over = 1;
    }
  }
}

// s.epi (the env's reset counter) for the envs whose own draws are keyed by it; a kernel that steps calls this after unpack_state
template <int ENV>
__device__ __forceinline__ void load_episode_index(EnvState &s, const int32_t *__restrict__ n_resets, int64_t env, bool valid) {
  if (HasEnvDraws<ENV>::value && valid) s.epi = n_resets[env];
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
static int grid_for(int64_t n_tiles, int cap) { return (int)(n_tiles < cap ? (n_tiles < 1 ? 1 : n_tiles) : cap); }

#define SGK_DISPATCH_ENV_LAYOUT(ENVID, LAYOUT, ...)                                                     \
  do {                                                                                                     \
    if ((LAYOUT) == SGK_LAYOUT_COMPACT) {                                                                  \
      switch (ENVID) {                                                                                     \
      case SGK_BOAT_RACE: { constexpr int E = SGK_BOAT_RACE; constexpr int L = SGK_LAYOUT_COMPACT; __VA_ARGS__; } break;         \
      case SGK_ISLAND_NAVIGATION: { constexpr int E = SGK_ISLAND_NAVIGATION; constexpr int L = SGK_LAYOUT_COMPACT; __VA_ARGS__; } break; \
      case SGK_DISTRIBUTIONAL_SHIFT: { constexpr int E = SGK_DISTRIBUTIONAL_SHIFT; constexpr int L = SGK_LAYOUT_COMPACT; __VA_ARGS__; } break; \
      case SGK_WHISKY_GOLD: { constexpr int E = SGK_WHISKY_GOLD; constexpr int L = SGK_LAYOUT_COMPACT; __VA_ARGS__; } break; \
      case SGK_ABSENT_SUPERVISOR: { constexpr int E = SGK_ABSENT_SUPERVISOR; constexpr int L = SGK_LAYOUT_COMPACT; __VA_ARGS__; } break; \
      case SGK_SAFE_INTERRUPTIBILITY: { constexpr int E = SGK_SAFE_INTERRUPTIBILITY; constexpr int L = SGK_LAYOUT_COMPACT; __VA_ARGS__; } break; \
      case SGK_CONVEYOR_BELT: { constexpr int E = SGK_CONVEYOR_BELT; constexpr int L = SGK_LAYOUT_COMPACT; __VA_ARGS__; } break; \
      case SGK_TOMATO_WATERING: { constexpr int E = SGK_TOMATO_WATERING; constexpr int L = SGK_LAYOUT_COMPACT; __VA_ARGS__; } break; \
      case SGK_FRIEND_FOE: { constexpr int E = SGK_FRIEND_FOE; constexpr int L = SGK_LAYOUT_COMPACT; __VA_ARGS__; } break; \
      default: { constexpr int E = SGK_SIDE_EFFECTS_SOKOBAN; constexpr int L = SGK_LAYOUT_COMPACT; __VA_ARGS__; } break;         \
      }                                                                                                    \
    } else {                                                                                               \
      switch (ENVID) {                                                                                     \
      case SGK_BOAT_RACE: { constexpr int E = SGK_BOAT_RACE; constexpr int L = SGK_LAYOUT_PITCHED; __VA_ARGS__; } break;         \
      case SGK_ISLAND_NAVIGATION: { constexpr int E = SGK_ISLAND_NAVIGATION; constexpr int L = SGK_LAYOUT_PITCHED; __VA_ARGS__; } break; \
      case SGK_DISTRIBUTIONAL_SHIFT: { constexpr int E = SGK_DISTRIBUTIONAL_SHIFT; constexpr int L = SGK_LAYOUT_PITCHED; __VA_ARGS__; } break; \
      case SGK_WHISKY_GOLD: { constexpr int E = SGK_WHISKY_GOLD; constexpr int L = SGK_LAYOUT_PITCHED; __VA_ARGS__; } break; \
      case SGK_ABSENT_SUPERVISOR: { constexpr int E = SGK_ABSENT_SUPERVISOR; constexpr int L = SGK_LAYOUT_PITCHED; __VA_ARGS__; } break; \
      case SGK_SAFE_INTERRUPTIBILITY: { constexpr int E = SGK_SAFE_INTERRUPTIBILITY; constexpr int L = SGK_LAYOUT_PITCHED; __VA_ARGS__; } break; \
      case SGK_CONVEYOR_BELT: { constexpr int E = SGK_CONVEYOR_BELT; constexpr int L = SGK_LAYOUT_PITCHED; __VA_ARGS__; } break; \
      case SGK_TOMATO_WATERING: { constexpr int E = SGK_TOMATO_WATERING; constexpr int L = SGK_LAYOUT_PITCHED; __VA_ARGS__; } break; \
      case SGK_FRIEND_FOE: { constexpr int E = SGK_FRIEND_FOE; constexpr int L = SGK_LAYOUT_PITCHED; __VA_ARGS__; } break; \
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
    case SGK_WHISKY_GOLD: { constexpr int E = SGK_WHISKY_GOLD; __VA_ARGS__; } break; \
    case SGK_ABSENT_SUPERVISOR: { constexpr int E = SGK_ABSENT_SUPERVISOR; __VA_ARGS__; } break; \
    case SGK_SAFE_INTERRUPTIBILITY: { constexpr int E = SGK_SAFE_INTERRUPTIBILITY; __VA_ARGS__; } break; \
    case SGK_CONVEYOR_BELT: { constexpr int E = SGK_CONVEYOR_BELT; __VA_ARGS__; } break; \
    case SGK_TOMATO_WATERING: { constexpr int E = SGK_TOMATO_WATERING; __VA_ARGS__; } break; \
    case SGK_FRIEND_FOE: { constexpr int E = SGK_FRIEND_FOE; __VA_ARGS__; } break; \
    default: { constexpr int E = SGK_SIDE_EFFECTS_SOKOBAN; __VA_ARGS__; } break;               \
    }                                                                                    \
  } while (0)

static inline StepArgs make_step_args(const Shard &sh, const uint8_t *actions, uint32_t flags) {
  StepArgs a;
  a.rules = sh.rules_dev;
  a.state = sh.state;
  a.actions = actions;
  a.rec = sh.rec;
  a.boards = sh.boards;
  a.last_return = sh.last_return;
  a.last_perf = sh.last_perf;
  a.n_episodes = sh.n_episodes;
  a.n_resets = sh.n_resets;
  a.aux = sh.aux;
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
}  // namespace sgk
