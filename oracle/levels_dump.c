/* levels_dump.c -- one level table as text (test infrastructure). Compiled twice by oracle/Makefile, once per header
 * (-DLEVELS_HEADER='"levels_oracle.h"' / '"../include/sgk_levels.h"'); tests/test_levels_independent.py compares the two outputs
 * line by line: every level's shape and art, every character constant, reward constant, probability threshold and switch default,
 * the observation value and the colour of every printable character. */
#include <stdio.h>
#include <string.h>

#include LEVELS_HEADER

#define ORC_EXPORT __attribute__((visibility("default")))

static size_t put(char *buf, size_t cap, size_t at, const char *fmt, ...) __attribute__((format(printf, 4, 5)));
#include <stdarg.h>
static size_t put(char *buf, size_t cap, size_t at, const char *fmt, ...) {
  if (at >= cap) return at;
  va_list ap;
  va_start(ap, fmt);
  int n = vsnprintf(buf + at, cap - at, fmt, ap);
  va_end(ap);
  return n < 0 ? at : at + (size_t)n;
}

#define INT(name) at = put(buf, cap, at, #name " = %lld\n", (long long)(name))
#define CHR(name) at = put(buf, cap, at, #name " = '%c'\n", (char)(name))
#define DBL(name) at = put(buf, cap, at, #name " = %.17g\n", (double)(name))

/* fills buf with the table (NUL-terminated), returns the number of bytes needed */
ORC_EXPORT size_t levels_dump(char *buf, size_t cap) {
  size_t at = 0;
  INT(SGK_N_ENVS); INT(SGK_MAX_ITERATIONS); INT(SGK_N_ACTIONS);
  INT(SGK_ACT_UP); INT(SGK_ACT_DOWN); INT(SGK_ACT_LEFT); INT(SGK_ACT_RIGHT); INT(SGK_RNG_STREAM_ENV);
  INT(SGK_ENV_BOAT); INT(SGK_ENV_ISLAND); INT(SGK_ENV_SOKOBAN); INT(SGK_ENV_LAVA); INT(SGK_ENV_WHISKY); INT(SGK_ENV_SUPER);
  INT(SGK_ENV_INTERRUPT); INT(SGK_ENV_BELT); INT(SGK_ENV_TOMATO); INT(SGK_ENV_FOE);
  CHR(SGK_CH_AGENT); CHR(SGK_CH_WALL); CHR(SGK_CH_SPACE); CHR(SGK_CH_GOAL);
  /* boat race */
  CHR(SGK_BOAT_CH_N); CHR(SGK_BOAT_CH_E); CHR(SGK_BOAT_CH_S); CHR(SGK_BOAT_CH_W);
  INT(SGK_BOAT_MOVEMENT_REWARD); INT(SGK_BOAT_CLOCKWISE_REWARD); INT(SGK_BOAT_CLOCKWISE_HIDDEN_REWARD);
  INT(SGK_BOAT_MOVEMENT_IN_HIDDEN); INT(SGK_BOAT_BLOCKED_ON_ARROW_PENALISED);
  /* island navigation */
  CHR(SGK_CH_WATER); INT(SGK_ISLAND_MOVEMENT_REWARD); INT(SGK_ISLAND_FINAL_REWARD); INT(SGK_ISLAND_WATER_HIDDEN);
  INT(SGK_ISLAND_WATER_OBSERVED); INT(SGK_ISLAND_WATER_OVER_AGENT); INT(SGK_ISLAND_VALUE_WALL);
  /* sokoban */
  CHR(SGK_CH_BOX); CHR(SGK_CH_COIN); INT(SGK_SOKOBAN_MOVEMENT_REWARD); INT(SGK_SOKOBAN_GOAL_REWARD);
  INT(SGK_SOKOBAN_HIDDEN_ADJACENT_WALL); INT(SGK_SOKOBAN_HIDDEN_ADJACENT_CORNER); INT(SGK_SOKOBAN_BOX_BLOCKED_BY_GOAL);
  INT(SGK_SOKOBAN_WALL_RULE); INT(SGK_SOKOBAN_VALUE_SET);
  /* lava */
  CHR(SGK_CH_LAVA); INT(SGK_LAVA_MOVEMENT_REWARD); INT(SGK_LAVA_GOAL_REWARD); INT(SGK_LAVA_LAVA_REWARD);
  /* whisky */
  CHR(SGK_CH_WHISKY); INT(SGK_WHISKY_MOVEMENT_REWARD); INT(SGK_WHISKY_GOAL_REWARD); INT(SGK_WHISKY_WHISKY_REWARD);
  INT(SGK_WHISKY_EXPLORATION_U32);
  /* absent supervisor */
  CHR(SGK_CH_SUPERVISOR); CHR(SGK_CH_PUNISHMENT); INT(SGK_SUPER_MOVEMENT_REWARD); INT(SGK_SUPER_FINAL_REWARD);
  INT(SGK_SUPER_PUNISHMENT_REWARD); INT(SGK_SUPER_PRESENT_U32);
  /* safe interruptibility */
  CHR(SGK_CH_INTERRUPTION); CHR(SGK_CH_BUTTON); INT(SGK_INTERRUPT_MOVEMENT_REWARD); INT(SGK_INTERRUPT_GOAL_REWARD);
  INT(SGK_INTERRUPT_PROBABILITY_U32); INT(SGK_INTERRUPT_FORCED_ACTION); INT(SGK_INTERRUPT_HIDDEN_ONLY_UNINTERRUPTED);
  INT(SGK_INTERRUPT_VALUE_SET);
  /* conveyor belt */
  CHR(SGK_CH_OBJECT); CHR(SGK_CH_BELT); CHR(SGK_CH_BELT_END); INT(SGK_BELT_REMOVAL_REWARD); INT(SGK_BELT_HIDDEN_REWARD);
  INT(SGK_BELT_END_OVER_OBJECT); INT(SGK_BELT_OBJECT_BLOCKED_BY_AGENT); INT(SGK_BELT_CURTAIN_COVERS_END);
  /* tomato watering */
  CHR(SGK_CH_TOMATO_WATERED); CHR(SGK_CH_TOMATO_DRY); CHR(SGK_CH_TRANSFORMER); INT(SGK_TOMATO_N); DBL(SGK_TOMATO_REWARD_FACTOR);
  INT(SGK_TOMATO_DRY_U32); INT(SGK_TOMATO_DELUSION_COVERS_ALL_CELLS);
  /* friend or foe */
  CHR(SGK_CH_FOE_GOAL); CHR(SGK_CH_FOE_HIDE); CHR(SGK_CH_FOE_FRIEND); CHR(SGK_CH_FOE_NEUTRAL); CHR(SGK_CH_FOE_ADVERSARY);
  INT(SGK_FOE_FRIEND); INT(SGK_FOE_NEUTRAL); INT(SGK_FOE_ADVERSARY); DBL(SGK_FOE_LEARNING_RATE); INT(SGK_FOE_NEUTRAL_BOX0_U32);
  INT(SGK_FOE_MOVEMENT_REWARD); INT(SGK_FOE_GOAL_REWARD); INT(SGK_FOE_EMPTY_REWARD);
  INT(SGK_RENDER_HWC);
  /* per level: shape, art, the value and the colour of every printable character */
  for (int env = 0; env < SGK_N_ENVS; ++env) {
    int H = 0, W = 0;
    const char *const *art = 0;
    if (sgk_level_shape(env, &H, &W, &art) != 0) {
      at = put(buf, cap, at, "level %d: no shape\n", env);
      continue;
    }
    at = put(buf, cap, at, "level %d shape = %d x %d\n", env, H, W);
    for (int r = 0; r < H; ++r) at = put(buf, cap, at, "level %d art[%d] = \"%s\" (%d chars)\n", env, r, art[r], (int)strlen(art[r]));
    for (int ch = 32; ch < 127; ++ch) {
      int rgb[3] = {-1, -1, -1};
      const int v = sgk_value_of(env, (char)ch), c = sgk_colour_of(env, (char)ch, rgb);
      if (v >= 0 || c == 0) at = put(buf, cap, at, "level %d '%c': value %d, colour %d %d %d\n", env, ch, v, rgb[0], rgb[1], rgb[2]);
    }
  }
  /* the absent supervisor's second art */
  for (int r = 0; r < SGK_SUPER_H; ++r) at = put(buf, cap, at, "level %d art_absent[%d] = \"%s\"\n", SGK_ENV_SUPER, r, SGK_SUPER_ART_ABSENT[r]);
  if (at < cap) buf[at] = 0;
  return at + 1;
}
