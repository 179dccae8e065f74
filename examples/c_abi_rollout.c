/* c_abi_rollout.c -- the C-ABI of libsgk.so used from plain C (no Python, no PyTorch).
 *
 *   gcc -O2 -Iinclude examples/c_abi_rollout.c -o c_abi_rollout -Lsafe-grid-agents_amd/lib -lsgk \
 *       -Wl,-rpath,$PWD/safe-grid-agents_amd/lib
 *   ./c_abi_rollout <env_id> <n_envs> <n_steps> <seed> [shards]
 *
 * Random-action lockstep rollout with reset-on-done (the loop shape of reference warmup.py:14-21): n_steps through the step
 * kernel (one launch per step), n_steps through the fused rollout kernel, n_steps through the streaming rollout kernel
 * (every step's boards and records materialised); prints the aggregate episode metrics (what track_metrics accumulates,
 * reference meters.py:76-84) and a checksum of the final boards as one JSON line.
 *
 * With shards > 1 the batch is cut into contiguous env-id blocks the way a multi-GPU job cuts it -- one sgk_env per block,
 * created with its env_index_base so that the counter RNG is keyed by GLOBAL env index -- and every block's metrics go
 * through the library's RCCL all-reduce (sgk_metrics_allreduced). On a multi-GPU node each block would live in its own
 * process on its own GPU with ONE communicator of `shards` ranks; this example runs the blocks one after the other on
 * device 0, each with a communicator of one rank, and adds the vectors itself: the printed line must not depend on `shards`. */
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
  int shards = argc > 5 ? atoi(argv[5]) : 1;
  if (shards < 1 || shards > n) shards = 1;

  int64_t total[SGK_METRICS_LEN] = {0};
  for (int i = SGK_M_MAX_RETURN; i <= SGK_M_MAX_MARGIN_POS; ++i) total[i] = INT64_MIN;
  uint64_t checksum = 1469598103934665603ull; /* FNV-1a over the dense boards, in env order */
  sgk_info info;
  for (int r = 0; r < shards; ++r) {
    /* rank r's contiguous block [begin, end): remainders go to the lowest ranks (dist.shard_range) */
    const int64_t base = n / shards, rem = n % shards;
    const int64_t begin = r * base + (r < rem ? r : rem), end = begin + base + (r < rem ? 1 : 0);
    sgk_env *env = NULL;
    CHECK(sgk_create_ex(env_id, end - begin, 0, seed, (uint64_t)begin, SGK_LAYOUT_COMPACT, &env));
    CHECK(sgk_get_info(env, &info));
    CHECK(sgk_step_random(env, steps, SGK_F_AUTO_RESET));
    CHECK(sgk_rollout_random(env, steps, SGK_F_AUTO_RESET));
    CHECK(sgk_rollout_random_stream(env, steps, SGK_F_AUTO_RESET, NULL, NULL, 1, 0));
    int64_t m[SGK_METRICS_LEN];
    if (shards > 1) {
      uint8_t id[SGK_COMM_ID_BYTES];
      sgk_comm *comm = NULL;
      CHECK(sgk_comm_unique_id(id));
      CHECK(sgk_comm_create(id, 0, 1, 0, &comm));
      CHECK(sgk_metrics_allreduced(env, comm, m));
      CHECK(sgk_comm_destroy(comm));
    } else {
      CHECK(sgk_metrics(env, m));
    }
    for (int i = 0; i < 8; ++i) total[i] += m[i];
    for (int i = SGK_M_MAX_RETURN; i <= SGK_M_MAX_MARGIN_POS; ++i)
      if (m[i] > total[i]) total[i] = m[i];
    const size_t bytes = (size_t)(end - begin) * info.n_cells;
    int8_t *boards = (int8_t *)malloc(bytes);
    CHECK(sgk_copy_boards(env, boards));
    for (size_t i = 0; i < bytes; ++i) checksum = (checksum ^ (uint8_t)boards[i]) * 1099511628211ull;
    free(boards);
    CHECK(sgk_destroy(env));
  }
  printf("{\"env_id\": %d, \"n_envs\": %" PRId64 ", \"height\": %d, \"width\": %d, \"steps\": %" PRId64
         ", \"episodes\": %" PRId64 ", \"sum_return\": %" PRId64 ", \"sum_safety\": %" PRId64 ", \"max_return\": %" PRId64
         ", \"boards_fnv1a\": %" PRIu64 "}\n",
         env_id, n, info.height, info.width, total[SGK_M_STEPS], total[SGK_M_EPISODES], total[SGK_M_SUM_RETURN],
         total[SGK_M_SUM_SAFETY], total[SGK_M_MAX_RETURN], checksum);
  return 0;
}
