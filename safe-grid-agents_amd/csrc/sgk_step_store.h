// sgk_step_store.h -- the pieces of sgk_step.hip that sgk_learn.hip shares: the replay-ring side output of the step / reset kernels
// (StepStore) and the reset kernel's body over a virtual grid.
#pragma once
#include "sgk_device.h"

namespace sgk {

// ReplayBuffer.add's second half (reference contain.py:15-17, called from value.py:114 right after env.step, learn.py:38-48) fused
// into the step (sgk_step_store): the successor board, the action, the reward (hidden under --cheat) and the terminal flag of every
// env go into slice `slice` of a device replay ring next to the step's usual outputs -- one launch instead of sgk_step +
// sgk_replay_store(phase 1). Also used by reset_kernel for the first half (the board the next action is chosen on -> states ring).
struct StepStore {
  int8_t *boards;       // ring [slices][n][NC] that receives this launch's boards
  uint8_t *actions;     // rings [slices][n]; null for the reset kernel
  int8_t *rewards;
  uint8_t *terminals;
  long long slice;      // slice index ...
  const long long *slice_dev;  // ... or, when non-null (graph replays), (*slice_dev + slice) % ring
  int32_t ring;         // slices of the ring (for the modulo; 0: no modulo)
  int32_t cheat;        // store the hidden reward and the executed action (learn.py:41-47)
  int32_t tiles_ok;     // n * NC % 16 == 0: whole 64-env tiles go through the tile writer
};
__device__ __forceinline__ long long store_slice(const StepStore &st) {
  long long sl = st.slice;
  if (st.slice_dev) sl += *st.slice_dev;
  if (st.ring > 0) sl %= st.ring;
  return sl;
}
// this wave's board tile (in `W`, drawn) -> rows wt * 64 .. of slice `sl` of the ring: the tile writer where the slice's rows are
// 16-byte aligned and the tile is whole, else lane by lane from the LDS image
template <class Tile, int NC>
__device__ __forceinline__ void store_tile(const Tile &W, const StepStore &st, long long sl, int64_t n, int64_t wt, int lane) {
  int8_t *dst = st.boards + ((int64_t)sl * n + wt * 64) * NC;
  if (st.tiles_ok && wt * 64 + 64 <= n) {
    W.template flush<0>(dst);
  } else {
    __builtin_amdgcn_wave_barrier();
    if (wt * 64 + lane < n)
      for (int c = 0; c < NC; ++c) dst[lane * NC + c] = (int8_t)W.tile[lane * NC + c];
    __builtin_amdgcn_wave_barrier();
  }
}

// reset_kernel's body as a device function over a VIRTUAL grid (workgroup vblock of vgrid, 256 lanes each): sgk_step.hip's reset_kernel
// runs it over its own grid, sgk_learn.hip's dqn_adam_reset_kernel over the workgroups behind the Adam ones (sgk_dqn_sgd_step_reset_store)
template <int ENV, int LAYOUT, bool STORE>
__device__ __forceinline__ void reset_body(const SgkRules *rules, uint64_t *state, int8_t *boards, const uint8_t *mask, int mode_flags,
                                           int64_t n, uint64_t seed, uint64_t env_base, int32_t *__restrict__ n_resets,
                                           const double *__restrict__ aux, StepStore st, int vblock, int vgrid) {
  const int mode = mode_flags & 3;
  const bool no_boards = (mode_flags & 4) != 0;  // state words only (the caller steps with SGK_F_NO_BOARDS)
  constexpr int NC = Geom<ENV>::NC;
  constexpr bool COMPACT = (LAYOUT == SGK_LAYOUT_COMPACT);
  // like step_kernel: wave-private rule table and board tile, no workgroup barrier, every load of the wave's first tile -- its
  // state words, its mask bytes, its pieces of the table and of the blank tile -- requested before the first wait (reset_done is
  // one of the four launches of every lockstep step of the drop-in call sequences: its latency is their latency)
  __shared__ WaveRulesImage rules_images[WG / 64];
  __shared__ __attribute__((aligned(16))) uint8_t tile_images[COMPACT ? WG / 64 : 1][COMPACT ? 64 * NC : 16];
  const int lane = threadIdx.x & 63, wave = wave_index();
  const int64_t n_wt = (n + 63) / 64;
  const int64_t wt0 = (int64_t)vblock * (WG / 64) + wave, wstride = (int64_t)vgrid * (WG / 64);
  uint64_t w_cur = 0;
  uint8_t m_cur = 1;
  {
    const int64_t e0 = wt0 * 64 + lane;
    const int64_t e0c = e0 < n ? e0 : n - 1;  // (index clamped into the batch: a wave past the last tile drops the word)
    w_cur = state[e0c];
    if (mask) m_cur = mask[e0c];
  }
  WaveRulesLoad rules_load;
  rules_load.request(rules);
  WaveTileLds<ENV, NC> W;
  W.bind(tile_images[COMPACT ? wave : 0]);
  typename WaveTileLds<ENV, NC>::Blank blank;
  if (COMPACT) W.request_blank(blank, rules);
  const long long st_slice = STORE ? store_slice(st) : 0;
  rules_load.commit(rules_images[wave]);
  const SgkRules &R = rules_images[wave].r;
  if (COMPACT) W.blank_arrived(blank);  // (every load of the wave is in before its first store goes out: one counter for both)
  for (int64_t wt = wt0; wt < n_wt;) {
    const int64_t env = wt * 64 + lane;
    const bool valid = env < n;
    const EnvState cur = unpack_state(w_cur);
    const bool masked = m_cur != 0;
    // the next tile's words, while this one is worked on -- taken over behind the loop's exit (step_kernel has the reason)
    const int64_t wt_next = wt + wstride;
    const bool more = wt_next < n_wt;  // wave-uniform
    uint64_t w_next = 0;
    uint32_t m_next = 1;
    if (more) {
      const int64_t ne = wt_next * 64 + lane;
      if (ne < n) {
        w_next = state[ne];
        if (mask) m_next = mask[ne];
      }
    }
    EnvState s = initial_state(R);
    bool hit = false;
    if (valid) {
      hit = (mode == 2) ? false : (mode == 1 ? (cur.over != 0) : (mask == nullptr || masked));
      if (hit) {
        if (HasEnvDraws<ENV>::value) {  // every reset opens a new draw sequence: the counter is the key
          s.epi = n_resets[env] + 1;
          n_resets[env] = s.epi;
        }
        begin_episode<ENV>(R, s, seed, env_base + (uint64_t)env, HasAux<ENV>::value ? aux + env * SGK_AUX_DOUBLES : nullptr);
        state[env] = pack_state(s);
      } else {
        s = cur;
      }
    }
    // boards: every tile when re-materialising (mode 2) or resetting everything; otherwise only the tiles (rows) a reset
    // touched -- the others already show their envs' states, and rewriting them is most of this kernel's traffic
    const bool all = mode == 2 || (mode == 0 && mask == nullptr);
    if (!no_boards || STORE) {
      if (COMPACT) {
        const bool touched = all || __ballot(hit) != 0ull;
        if (touched || STORE) {
          W.draw_from_blank(blank, R, sprite_info<ENV>(R, s));
          if (touched && !no_boards) W.flush(boards + wt * 64 * NC);
          // (STORE: the boards the next actions are chosen on -- EVERY env's, reset or not -- are the next transitions' states)
          if (STORE) store_tile<WaveTileLds<ENV, NC>, NC>(W, st, st_slice, n, wt, lane);
        }
      } else if (valid) {
        if (!no_boards && (all || hit)) write_board_pitched<ENV, Geom<ENV>::PITCH>(R, boards, env, s);
        if (STORE) write_row_bytes<ENV, NC>(R, st.boards + ((int64_t)st_slice * n + env) * NC, s);
      }
    }
    if (!more) break;
    asm volatile("" : "+v"(w_next), "+v"(m_next));
    w_cur = w_next;
    m_cur = (uint8_t)m_next;
    wt = wt_next;
  }
}

static inline StepStore make_store(const Shard &sh, int8_t *boards_ring, uint8_t *r_actions, int8_t *r_rewards, uint8_t *r_terminals, int64_t slice,
                            const long long *slice_dev, int32_t ring, int cheat) {
  StepStore s;
  s.boards = boards_ring; s.actions = r_actions; s.rewards = r_rewards; s.terminals = r_terminals;
  s.slice = slice; s.slice_dev = slice_dev; s.ring = ring; s.cheat = cheat;
  s.tiles_ok = ((sh.n * sh.n_cells) % 16 == 0 && ((uintptr_t)boards_ring & 15u) == 0) ? 1 : 0;
  return s;
}

}  // namespace sgk
