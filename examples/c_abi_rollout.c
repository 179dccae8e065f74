/* c_abi_rollout.c -- the C-ABI of libsgk.so used from plain C (no Python, no PyTorch).
 *
 *   gcc -O2 -Iinclude examples/c_abi_rollout.c -o c_abi_rollout -Lsafe-grid-agents_amd/lib -lsgk \
 *       -Wl,-rpath,$PWD/safe-grid-agents_amd/lib
 *   ./c_abi_rollout <env_id> <n_envs> <n_steps> <seed>
 *
 * Random-action lockstep rollout with reset-on-done (the loop shape of reference warmup.py:14-21), first through the
 * step kernel, then the same number of steps through the fused rollout kernel; prints the aggregate episode metrics
 * (what track_metrics accumulates, reference meters.py:76-84) and a checksum of the final boards as one JSON line. */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>

#include "sgk.h"

#define CHECK(call)                                                          \
  do {                                                                       \
    int rc__ = (call);                                                       \
    if (rc__ != SGK_OK) {                                                    \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc__, sgk_last_error()); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

int main(int argc, char **argv) {
  int env_id = argc > 1 ? atoi(argv[1]) : SGK_BOAT_RACE;
  int64_t n = argc > 2 ? atoll(argv[2]) : 4096;
  int steps = argc > 3 ? atoi(argv[3]) : 250;
  uint64_t seed = argc > 4 ? strtoull(argv[4], NULL, 0) : 0x5AFE;

  sgk_env *env = NULL;
  CHECK(sgk_create(env_id, n, 0, seed, &env));
  sgk_info info;
  CHECK(sgk_get_info(env, &info));
  CHECK(sgk_step_random(env, steps, SGK_F_AUTO_RESET));
  CHECK(sgk_rollout_random(env, steps, SGK_F_AUTO_RESET));
  int64_t m[SGK_METRICS_LEN];
  CHECK(sgk_metrics(env, m));
  int8_t *boards = (int8_t *)malloc((size_t)n * info.n_cells);
  CHECK(sgk_copy_boards(env, boards));
  uint64_t checksum = 1469598103934665603ull; /* FNV-1a over the dense boards */
  for (size_t i = 0; i < (size_t)n * info.n_cells; ++i) checksum = (checksum ^ (uint8_t)boards[i]) * 1099511628211ull;
  printf("{\"env_id\": %d, \"n_envs\": %" PRId64 ", \"height\": %d, \"width\": %d, \"steps\": %" PRId64
         ", \"episodes\": %" PRId64 ", \"sum_return\": %" PRId64 ", \"sum_safety\": %" PRId64 ", \"max_return\": %" PRId64
         ", \"boards_fnv1a\": %" PRIu64 "}\n",
         info.env_id, info.n_envs, info.height, info.width, m[SGK_M_STEPS], m[SGK_M_EPISODES], m[SGK_M_SUM_RETURN],
         m[SGK_M_SUM_SAFETY], m[SGK_M_MAX_RETURN], checksum);
  free(boards);
  CHECK(sgk_destroy(env));
  return 0;
}
