// sgk_host_debug.cpp -- the kernels' transition code (sgk_transition.h) evaluated on the HOST for one (state, action), plus
// the level tables, behind plain C entry points. Linked into libsgk.so (include/sgk.h: sgk_debug_host_transition,
// sgk_debug_level, sgk_random_action) and, together with sgk_rules.cpp only, into the host-only library the CPU test-suite
// builds with g++ -- under the default reading of every uncertain upstream detail and under each alternative
// (tests/test_switch_variants.py). Never used by a product path.
#include <cstring>

#include "sgk_transition.h"

namespace sgk {

int host_debug_transition(const SgkRules &R, int agent_cell, int box_cell, int action, int out[5]) {
  EnvState s;
  s.pos = agent_cell; s.box = box_cell & 0xff; s.frame = 0; s.over = 0; s.ret = 0; s.hid = 0; s.epi = 0; s.ext = 0; s.draws = 0;
  s.mode = (box_cell >> 8) & 1;  // the debug hook carries the state word's mode bit above the box byte
  int r_obs = 0, r_hid = 0, term = 0;
  switch (R.env_id) {
  case SGK_BOAT_RACE: transition<SGK_BOAT_RACE>(R, s, action, r_obs, r_hid, term); break;
  case SGK_ISLAND_NAVIGATION: transition<SGK_ISLAND_NAVIGATION>(R, s, action, r_obs, r_hid, term); break;
  case SGK_SIDE_EFFECTS_SOKOBAN: transition<SGK_SIDE_EFFECTS_SOKOBAN>(R, s, action, r_obs, r_hid, term); break;
  case SGK_DISTRIBUTIONAL_SHIFT: transition<SGK_DISTRIBUTIONAL_SHIFT>(R, s, action, r_obs, r_hid, term); break;
  case SGK_WHISKY_GOLD: transition<SGK_WHISKY_GOLD>(R, s, action, r_obs, r_hid, term); break;
  case SGK_ABSENT_SUPERVISOR: transition<SGK_ABSENT_SUPERVISOR>(R, s, action, r_obs, r_hid, term); break;
  case SGK_CONVEYOR_BELT: transition<SGK_CONVEYOR_BELT>(R, s, action, r_obs, r_hid, term); break;
  case SGK_TOMATO_WATERING: transition<SGK_TOMATO_WATERING>(R, s, action, r_obs, r_hid, term); break;  // (no drying: s.draws = 0)
  case SGK_FRIEND_FOE: transition<SGK_FRIEND_FOE>(R, s, action, r_obs, r_hid, term); break;  // (level 0, no estimator update)
  case SGK_SAFE_INTERRUPTIBILITY: {
    // the hook is handed the action the AGENT chose: the interruption drape's substitution is part of the kernels' step
    const int executed = env_actual_action<SGK_SAFE_INTERRUPTIBILITY>(R, s, 0, 0, action);
    transition<SGK_SAFE_INTERRUPTIBILITY>(R, s, executed, r_obs, r_hid, term);
    break;
  }
  default: return -1;
  }
  out[0] = s.pos; out[1] = s.box; out[2] = r_obs; out[3] = r_hid; out[4] = term | (s.mode << 1);  // bit 1: the mode bit after the step
  return 0;
}

// one env.step, state word in / state word out: env_actual_action + transition + the bookkeeping of step_one without auto-reset
template <int ENV>
static void host_step_env(const SgkRules &R, EnvState &s, int action, uint64_t seed, uint64_t env, int out[4], double *aux) {
  int r_obs = 0, r_hid = 0, term = 0;
  bool finished = false;
  if (!s.over) {
    action = env_actual_action<ENV>(R, s, seed, env, action);
    transition<ENV>(R, s, action, r_obs, r_hid, term, aux);
    s.frame += 1;
    s.ret += r_obs;
    s.hid += r_hid;
    finished = term || s.frame >= R.max_iterations;
    if (finished) s.over = 1;
  }
  out[0] = r_obs; out[1] = r_hid; out[2] = s.over ? 1 : 0; out[3] = action;
}

int host_debug_step(const SgkRules &R, uint64_t word, int n_resets, int action, uint64_t seed, uint64_t env, uint64_t *word_out,
                    int out[4], double *aux) {
  EnvState s = unpack_state(word);
  s.epi = n_resets;
  switch (R.env_id) {
  case SGK_BOAT_RACE: host_step_env<SGK_BOAT_RACE>(R, s, action, seed, env, out, aux); break;
  case SGK_ISLAND_NAVIGATION: host_step_env<SGK_ISLAND_NAVIGATION>(R, s, action, seed, env, out, aux); break;
  case SGK_SIDE_EFFECTS_SOKOBAN: host_step_env<SGK_SIDE_EFFECTS_SOKOBAN>(R, s, action, seed, env, out, aux); break;
  case SGK_DISTRIBUTIONAL_SHIFT: host_step_env<SGK_DISTRIBUTIONAL_SHIFT>(R, s, action, seed, env, out, aux); break;
  case SGK_WHISKY_GOLD: host_step_env<SGK_WHISKY_GOLD>(R, s, action, seed, env, out, aux); break;
  case SGK_ABSENT_SUPERVISOR: host_step_env<SGK_ABSENT_SUPERVISOR>(R, s, action, seed, env, out, aux); break;
  case SGK_SAFE_INTERRUPTIBILITY: host_step_env<SGK_SAFE_INTERRUPTIBILITY>(R, s, action, seed, env, out, aux); break;
  case SGK_CONVEYOR_BELT: host_step_env<SGK_CONVEYOR_BELT>(R, s, action, seed, env, out, aux); break;
  case SGK_TOMATO_WATERING: host_step_env<SGK_TOMATO_WATERING>(R, s, action, seed, env, out, aux); break;
  case SGK_FRIEND_FOE: host_step_env<SGK_FRIEND_FOE>(R, s, action, seed, env, out, aux); break;
  default: return -1;
  }
  *word_out = pack_state(s);
  return 0;
}

// the state word a reset leaves (reset number `n_resets`): initial_state + begin_episode
uint64_t host_reset_word(const SgkRules &R, uint64_t seed, uint64_t env, int n_resets, const double *aux) {
  EnvState s = initial_state(R);
  s.epi = n_resets;
  switch (R.env_id) {
  case SGK_ABSENT_SUPERVISOR: begin_episode<SGK_ABSENT_SUPERVISOR>(R, s, seed, env); break;
  case SGK_SAFE_INTERRUPTIBILITY: begin_episode<SGK_SAFE_INTERRUPTIBILITY>(R, s, seed, env); break;
  case SGK_FRIEND_FOE: begin_episode<SGK_FRIEND_FOE>(R, s, seed, env, aux); break;
  default: break;
  }
  return pack_state(s);
}

int host_random_action(uint64_t seed, uint64_t env, uint64_t t) {
  uint32_t x[4];
  philox4x32_10((uint32_t)env, (uint32_t)(env >> 32), (uint32_t)(t >> 6), 0u, (uint32_t)seed, (uint32_t)(seed >> 32), x);
  return action_from_block(x, t);
}

// the per-episode coin of reset number `n_resets` (absent supervisor / safe interruptibility): what begin_episode decides
int host_episode_coin(const SgkRules &R, uint64_t seed, uint64_t env, int n_resets) {
  EnvState s = initial_state(R);
  s.epi = n_resets;
  switch (R.env_id) {
  case SGK_ABSENT_SUPERVISOR: begin_episode<SGK_ABSENT_SUPERVISOR>(R, s, seed, env); break;
  case SGK_SAFE_INTERRUPTIBILITY: begin_episode<SGK_SAFE_INTERRUPTIBILITY>(R, s, seed, env); break;
  default: return -1;
  }
  return s.mode;
}

}  // namespace sgk

#ifdef SGK_HOST_ONLY
// the host-only library's C entry points: the same names and meaning as libsgk.so's debug hooks (include/sgk.h)
extern "C" {
#define SGK_HOST_API __attribute__((visibility("default")))
SGK_HOST_API int sgk_debug_host_transition(int env_id, int agent_cell, int box_cell, int action, int32_t out[5]) {
  SgkRules R;
  if (sgk_build_rules(env_id, &R) != 0) return SGK_ERR_INVALID;
  if (agent_cell < 0 || agent_cell >= R.n_cells || action < 0 || action >= SGK_ACTIONS || !out) return SGK_ERR_INVALID;
  int o[5];
  if (sgk::host_debug_transition(R, agent_cell, box_cell, action, o) != 0) return SGK_ERR_INVALID;
  for (int i = 0; i < 5; ++i) out[i] = o[i];
  return SGK_OK;
}
SGK_HOST_API int sgk_debug_host_step(int env_id, uint64_t state_word, int n_resets, int action, uint64_t seed, uint64_t env_index,
                                     uint64_t *state_word_out, int32_t out[4], double *aux_env) {
  SgkRules R;
  if (sgk_build_rules(env_id, &R) != 0 || action < 0 || action >= SGK_ACTIONS || !state_word_out || !out) return SGK_ERR_INVALID;
  int o[4];
  if (sgk::host_debug_step(R, state_word, n_resets, action, seed, env_index, state_word_out, o, aux_env) != 0) return SGK_ERR_INVALID;
  for (int i = 0; i < 4; ++i) out[i] = o[i];
  return SGK_OK;
}
SGK_HOST_API uint64_t sgk_debug_reset_word(int env_id, uint64_t seed, uint64_t env_index, int n_resets, const double *aux_env) {
  SgkRules R;
  if (sgk_build_rules(env_id, &R) != 0) return ~0ull;
  return sgk::host_reset_word(R, seed, env_index, n_resets, aux_env);
}
SGK_HOST_API int sgk_debug_level(int env_id, int32_t dims[4], uint8_t templ[64], uint8_t agent_value[64]) {
  SgkRules R;
  if (sgk_build_rules(env_id, &R) != 0 || !dims || !templ || !agent_value) return SGK_ERR_INVALID;
  dims[0] = R.height; dims[1] = R.width; dims[2] = R.start_agent; dims[3] = R.start_box;
  std::memcpy(templ, R.templ, 64);
  std::memcpy(agent_value, R.agent_value, 64);
  return SGK_OK;
}
SGK_HOST_API int sgk_debug_rules(int env_id, SgkRules *out) { return (out && sgk_build_rules(env_id, out) == 0) ? SGK_OK : SGK_ERR_INVALID; }
SGK_HOST_API int sgk_debug_rules_size(void) { return (int)sizeof(SgkRules); }
SGK_HOST_API int sgk_random_action(uint64_t seed, uint64_t env_index, uint64_t t) { return sgk::host_random_action(seed, env_index, t); }
SGK_HOST_API int sgk_debug_episode_coin(int env_id, uint64_t seed, uint64_t env_index, int n_resets) {
  SgkRules R;
  if (sgk_build_rules(env_id, &R) != 0) return SGK_ERR_INVALID;
  return sgk::host_episode_coin(R, seed, env_index, n_resets);
}
}
#endif
