/* sgk.h -- C-ABI of libsgk.so: batched lockstep safety-gridworld step / agent-rollout path on MI355X.
 *
 * The reference (jvmncs/safe-grid-agents) has no FFI; its seam for this path is the Python
 * gym.Env duck type (env.step / env.reset / env._env.episode_return / get_last_performance) and the
 * agent methods called from common/{learn,eval,warmup}.py. Each entry point below names the
 * reference interface it sits under (file:line in /root/reference). The Python host mirror
 * (safe-grid-agents_amd/safe_grid_agents_amd) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions: flat C, plain pointers and sizes. Every function returns SGK_OK (0) or a negative
 * SGK_ERR_* and leaves a message for sgk_last_error() (thread-local, a fixed buffer: reporting an error allocates nothing).
 * No C++ exception leaves the library: every entry point is a function-try-block (SGK_ERR_NOMEM / SGK_ERR_INTERNAL). Handles are opaque and
 * thread-compatible (one thread at a time per handle; different handles may be driven from different threads at the same time: the
 * library serialises its own hipGraph captures against each other and against its own synchronous legacy-stream calls -- on ROCm such a
 * call from ANY thread invalidates a capture in progress on any stream. A handle on a stream of its own or on a caller's non-NULL
 * stream makes none on its path; a handle bound to the device's NULL stream (sgk_use_default_stream: PyTorch's default) waits for
 * that stream in its synchronising entry points, and takes the capture lock while it does. A capture that a caller's own synchronous
 * call, e.g. PyTorch's, invalidated all the same is recorded again). Pointers named *_dev are DEVICE pointers
 * valid on the handle's GPU (e.g. torch tensor.data_ptr()); pointers named *_host are host memory.
 * All work is enqueued on the handle's HIP stream; only functions documented as synchronising wait.
 */
#ifndef SGK_H
#define SGK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGK_ABI_VERSION 4 /* 3: the tabular-Q tables in HBM are state-major (sgk_tabq_table_dev); 4: sgk_tabq_step, sgk_step_store, sgk_reset_done_store, sgk_convq_act, sgk_convq_sample, sgk_convq_rollout, sgk_dqn_sgd_step_reset_store, sgk_dqn_learner's
                             loss_mode / rows / rows_out, SGK_F_SEPARATE_LAUNCHES, the sgk_debug_* hooks are off unless asked for */

#if defined(__GNUC__)
#define SGK_API __attribute__((visibility("default")))
#else
#define SGK_API
#endif

#define SGK_OK 0
#define SGK_ERR_INVALID (-1) /* bad argument */
#define SGK_ERR_HIP (-2)     /* HIP runtime error (message has hipGetErrorString) */
#define SGK_ERR_NOMEM (-3)    /* a host allocation failed inside the library (std::bad_alloc stops at the boundary) */
#define SGK_ERR_NODEVICE (-4) /* no usable GPU: the library has NO CPU fallback */
#define SGK_ERR_INTERNAL (-5) /* any other C++ exception stopped at the boundary: no exception crosses the C-ABI */

/* env ids (reference parsing/parse.py:22-37 ENV_MAP aliases boat / island / sokoban / lava) */
#define SGK_BOAT_RACE 0
#define SGK_ISLAND_NAVIGATION 1
#define SGK_SIDE_EFFECTS_SOKOBAN 2
#define SGK_DISTRIBUTIONAL_SHIFT 3 /* "lava" -> "DistributionalShift-v0", training level */
#define SGK_ABSENT_SUPERVISOR 5    /* "super" -> "AbsentSupervisor-v0": a coin per episode (counter RNG stream 6) decides whether
                                    * the supervisor is present: border cells of the board and the punishment's observed reward */
#define SGK_SAFE_INTERRUPTIBILITY 6 /* "interrupt" -> "SafeInterruptibility-v0": a coin per episode (stream 6) decides whether
                                    * the interruption tile freezes the agent; the button removes the tile */
#define SGK_CONVEYOR_BELT 7         /* "belt" -> "ConveyorBelt-v0" ('vase'): the object is the second sprite (pushed by the agent,
                                    * carried east by the belt every step); state bit `mode` = it has reached the belt's end */
#define SGK_TOMATO_WATERING 8       /* "tomato" -> "TomatoWatering-v0": 13 tomatoes as a bit mask in the state word (the `box` byte +
                                    * five flag bits), each drying with p = 0.05 per step (stream 6); rewards are tomato COUNTS,
                                    * worth sgk_reward_scale() = 0.02 each; standing on the bucket shows every cell watered */
#define SGK_FRIEND_FOE 9            /* "bandit" -> "FriendFoe-v0": per env, three exponential smoothers of the agent's box preference
                                    * live ACROSS episodes (sgk_copy_bandit_policy); a draw per episode (stream 6) picks the
                                    * bandit type (the floor colour), the type's estimate picks the box with the reward */
#define SGK_WHISKY_GOLD 4          /* "whisky" -> "WhiskyGold-v0": the env replaces actions itself once the whisky is drunk
                                    * (counter RNG stream 6); the step record's `actual` byte carries what was executed */

/* flags for sgk_step / sgk_step_random / sgk_rollout_random */
#define SGK_F_AUTO_RESET 1u /* an env whose episode ends is reset in the same step (after its episode is recorded) */
#define SGK_F_NO_BOARDS 2u  /* do not materialise observation boards this call (they go stale until the next writing call) */
#define SGK_F_RING_TILE_MAJOR 8u /* sgk_rollout_random_stream: the trajectory rings are laid out tile-major (see there) */
#define SGK_F_SEPARATE_LAUNCHES 16u /* sgk_tabq_learn_steps: four launches per lockstep step instead of one */
#define SGK_F_MASK_FINISHED 4u /* sgk_policy_rollout: rows of states_out / actions_out of an env whose episode is over are zeros */

/* board layouts (sgk_create_ex) */
#define SGK_LAYOUT_PITCHED 0 /* env-major rows padded to a multiple of 16 B: one lane writes its row with 16-B stores */
#define SGK_LAYOUT_COMPACT 1 /* env-major rows of exactly n_cells bytes; a workgroup assembles its 256-env tile from LDS and
                                writes it as 1-KiB-per-wave streaming stores. Default of sgk_create (faster from ~128 K envs). */

/* memory placement, OR-ed into the `layout` argument of sgk_create_ex */
#define SGK_MEM_HOST_VISIBLE 0x100 /* state words, step records, boards in pinned device-mapped HOST memory: for the
                                      single-env / small-batch case with a host-side agent in the loop (sgk_step_host then
                                      needs no staging copies: one launch + one synchronisation). n_envs <= 65536. */

/* ENVIRONMENT KNOBS the library reads -- all of them (grep getenv over csrc), each once per sgk_create*, none on a hot path:
 *   SGK_NO_GRAPH=1      sgk_step_random issues eager launches instead of replaying a hipGraph (the profiler workloads set it: one
 *                       kernel record per launch).
 *   SGK_MAX_GRID=<n>    workgroups per launch of the grid-stride kernels (default 6 per CU; n >= 64); tools/sweep.py.
 *   SGK_STREAM_GRID=<n> workgroups per launch of the streamed rollout (default 16 per CU; n >= 64); likewise.
 *   SGK_RING_NT=0|1     board tiles into a trajectory ring never / always as non-temporal stores (default: by ring size and slice
 *                       count, sgk_step.hip: ring_stores_nt); the A/B knob behind profiles/r05/ring_nt_ab.log.
 *   SGK_STEP_SERVER=0   host-visible handles of <= 64 envs serve sgk_step_host with one launch per call instead of the resident
 *                       step server (the A/B of tools/bench_single_env.py).
 * The Python host adds SGK_LIB_PATH (another build of libsgk.so), SGK_NO_BUILD=1 (never start a build: set under profilers),
 * SGK_METRICS_COLLECTIVE=torch (metrics all-reduce through torch.distributed instead of the library's RCCL communicator),
 * SGK_DIST_TIMEOUT_S (collective timeout, default 120), SGK_TABQ_HASH_CAPACITY (slots per agent of hashed Q-tables). */

typedef struct sgk_env sgk_env;   /* one shard: N independent grid instances resident on one GPU */
typedef struct sgk_tabq sgk_tabq; /* N private tabular-Q agents bound to an sgk_env */

/* Per-env record written by every step (one coalesced dword per env):
 *   reward         observed reward        -> `reward` of env.step (reference learn.py:38,69)
 *   hidden_reward  info["hidden_reward"]  (reference learn.py:42,73; train.py:73)
 *   done           `done` of env.step
 *   actual_action  info["extra_observations"]["actual_actions"] (reference learn.py:45,76) */
typedef struct sgk_step_rec {
  int8_t reward;
  int8_t hidden_reward;
  uint8_t done;
  uint8_t actual_action;
} sgk_step_rec;

typedef struct sgk_info {
  int32_t env_id;
  int32_t height, width, n_cells; /* observation_space.shape = (1, height, width) (reference value.py:66) */
  int32_t n_actions;              /* action_space.n (reference dummy.py:11, value.py:19) */
  int32_t board_pitch;            /* bytes between consecutive envs' boards */
  int32_t layout;
  int32_t max_iterations;
  int32_t n_states;               /* size of the perfect-hash state index used by sgk_tabq */
  int32_t device;
  int64_t n_envs;
  uint64_t seed;
  uint64_t env_index_base;        /* global index of env 0 of this shard (keys the counter RNG) */
  uint64_t lockstep_t;            /* number of lockstep steps taken since create (keys the counter RNG) */
  int32_t render_hwc;             /* sgk_render_rgb frame layout: 0 = (3, H, W), 1 = (H, W, 3) (a build-time reading, sgk_levels.h) */
  int32_t reserved;
} sgk_info;

/* metrics vector (int64 x SGK_METRICS_LEN): the quantities track_metrics feeds its four meters
 * (reference meters.py:66-84). [0..7] are sums/counts (all-reduce SUM), [8..11] maxima (all-reduce MAX). */
#define SGK_METRICS_LEN 16
#define SGK_M_SUM_RETURN 0
#define SGK_M_SUM_SAFETY 1
#define SGK_M_SUM_MARGIN 2
#define SGK_M_SUM_MARGIN_POS 3
#define SGK_M_EPISODES 4
#define SGK_M_MARGIN_POS_COUNT 5
#define SGK_M_STEPS 6
#define SGK_M_MAX_RETURN 8
#define SGK_M_MAX_SAFETY 9
#define SGK_M_MAX_MARGIN 10
#define SGK_M_MAX_MARGIN_POS 11

SGK_API const char *sgk_last_error(void);
SGK_API int sgk_abi_version(void);
SGK_API int sgk_device_count(int *n_out);

/* ---- lifetime: gym.make(env_name) + env.seed(seed) (reference train.py:51-52) ------------------ */
SGK_API int sgk_create(int env_id, int64_t n_envs, int device, uint64_t seed, sgk_env **out);
SGK_API int sgk_create_ex(int env_id, int64_t n_envs, int device, uint64_t seed, uint64_t env_index_base, int layout,
                  sgk_env **out);
SGK_API int sgk_destroy(sgk_env *h);
/* env.seed(seed) (reference train.py:52): re-keys the counter RNG of later random-action / exploration draws. The
 * envs themselves are deterministic. */
SGK_API int sgk_set_seed(sgk_env *h, uint64_t seed);
SGK_API int sgk_get_info(const sgk_env *h, sgk_info *out);
SGK_API int sgk_set_stream(sgk_env *h, void *hip_stream); /* NULL restores the handle's own stream */
SGK_API int sgk_use_default_stream(sgk_env *h);           /* enqueue on the device's NULL (legacy default) stream */
/* The stream the handle enqueues on. The handle's OWN stream is never destroyed: after sgk_destroy it waits in a per-device pool
 * for the next sgk_create on that device, so a caller-side object that still remembers it (an event to record, a wrapper) finds a
 * valid stream. */
SGK_API void *sgk_get_stream(const sgk_env *h);
SGK_API int sgk_synchronize(sgk_env *h); /* waits for the handle's stream */
/* Stream ordering against another HIP stream of the same device (e.g. torch's current stream, where the policy network
 * runs) without a host wait: sgk_stream_wait makes the handle's stream wait for everything queued on other_stream so far
 * (call before handing it actions produced there); sgk_stream_signal makes other_stream wait for the handle's stream
 * (call before the consumer reads boards / records there). */
SGK_API int sgk_stream_wait(sgk_env *h, void *other_stream);
SGK_API int sgk_stream_signal(sgk_env *h, void *other_stream);

/* ---- env.reset() (reference train.py:64; eval.py:13,23; warmup.py:17) ------------------------- */
/* mask_dev == NULL: every env. Otherwise envs with mask_dev[i] != 0. Episode return, hidden return and
 * frame counter restart; get_last_performance() history is kept. */
SGK_API int sgk_reset(sgk_env *h, const uint8_t *mask_dev);
SGK_API int sgk_reset_done(sgk_env *h); /* resets exactly the envs whose episode is over */

/* ---- env.step(action) (reference learn.py:38,69; eval.py:36; warmup.py:20) -------------------- */
/* actions_dev: uint8[n_envs] in {0..3}. Writes the step records, advances episode state, materialises
 * the successor boards. Stepping an env whose episode is over (and not reset) is a no-op that reports
 * reward 0 / done 1. */
SGK_API int sgk_step(sgk_env *h, const uint8_t *actions_dev, uint32_t flags);
/* Host-buffer form for small N (the single-env drop-in of reference learn.py:38,69): copies the actions in,
 * steps, and copies out the step records, the dense boards [n_envs][n_cells] and env._env.episode_return.
 * Any output pointer may be NULL. Synchronises. */
SGK_API int sgk_step_host(sgk_env *h, const uint8_t *actions_host, uint32_t flags, sgk_step_rec *rec_host,
                  int8_t *boards_host, int32_t *episode_return_host);
/* RandomAgent.act + env.step (reference dummy.py:15-16, warmup.py:19-20): n_steps lockstep steps, one
 * launch per step (replayed from a hipGraph), actions from the counter RNG (Philox-4x32-10, stream 0). */
SGK_API int sgk_step_random(sgk_env *h, int32_t n_steps, uint32_t flags);
/* Builds (captures + instantiates) the hipGraph sgk_step_random(n_steps, flags) replays and uploads the lockstep counter,
 * WITHOUT stepping: a caller that times a region calls this for every chunk size it is going to use, so that no capture
 * falls inside the region. A no-op for chunk sizes that run as eager launches (n_steps < 4, SGK_NO_GRAPH=1). */
SGK_API int sgk_step_random_prepare(sgk_env *h, int32_t n_steps, uint32_t flags);
/* SingleActionAgent.act + env.step (reference dummy.py:19-30): every env repeats ITS action actions_dev[i] for n_steps
 * lockstep steps (one launch per step). */
SGK_API int sgk_step_repeat(sgk_env *h, const uint8_t *actions_dev, int32_t n_steps, uint32_t flags);
/* the same n_steps inside ONE launch: state stays in registers, boards are materialised once at the end */
SGK_API int sgk_rollout_random(sgk_env *h, int32_t n_steps, uint32_t flags);
/* The same n_steps in ONE launch with EVERY step's outputs materialised in HBM, as n_steps calls of sgk_step would leave them
 * one after the other: the successor board of every env (streaming tile stores) and its step record. The batched
 * dqn_warmup / random data collection (reference warmup.py:14-21: every random-action transition is kept). Destinations:
 *   boards_ring_dev == NULL and recs_ring_dev == NULL   the env's own board / record buffers (sgk_boards_dev,
 *       sgk_step_records_dev): each step overwrites the previous one's, the last step's outputs remain;
 *   otherwise trajectory rings the caller owns: boards int8 [ring_slices][n_envs][n_cells] (dense rows, 16-byte aligned),
 *       records sgk_step_rec [ring_slices][n_envs] (either may be NULL); step k of this call goes to slice
 *       (first_slice + k) % ring_slices. The env's own buffers then show the final state, as after sgk_rollout_random.
 *       With SGK_F_RING_TILE_MAJOR in `flags` the rings are TILE-major instead: boards int8 [n_tiles][ring_slices][64][n_cells],
 *       records sgk_step_rec [n_tiles][ring_slices][64], n_tiles = ceil(n_envs / 64) (env e = row e % 64 of tile e / 64; rows
 *       past n_envs in the last tile are padding the caller allocates): the K steps of one 64-env tile are adjacent in
 *       memory, so a wave streams one contiguous run per launch instead of pieces a whole slice apart -- the slice-major
 *       layout is bound by DRAM write locality, this one writes at the rate of the env's own buffers.
 * What a per-step launch pays and this does not: the launch boundary and the state word's round trip through HBM. Results
 * (state, records, boards, episode arrays, metrics) equal n_steps calls of sgk_step_random(1), bit for bit. */
SGK_API int sgk_rollout_random_stream(sgk_env *h, int32_t n_steps, uint32_t flags, int8_t *boards_ring_dev,
                                      sgk_step_rec *recs_ring_dev, int32_t ring_slices, int32_t first_slice);
/* Device memory for trajectory rings (and any other multi-GB buffer a long-running kernel streams into): one contiguous range
 * of `bytes` (rounded up to 2 MiB) mapped through HIP's virtual-memory management from physical chunks of 256 MiB. The streamed
 * rollout writes such a ring at 4.5-4.8 us per step (1 M BoatRace envs, 100 slices) where hipMalloc blocks of the same process
 * measure 4.6-4.9 or 5.6-6.1 depending on the block, for its lifetime. The reference keeps its transitions in a
 * Python deque (contain.py:11-13); this is where the batched form keeps them. sgk_ring_free synchronises the device first. Sizes
 * are rounded to the driver's recommended allocation granularity (never below 2 MiB); both calls leave the thread's current
 * device as they found it. */
SGK_API int sgk_ring_alloc(int32_t device, size_t bytes, void **dev_ptr);
SGK_API int sgk_ring_free(void *dev_ptr);
/* How fast can THIS trajectory ring be written? Runs the streamed rollout's stores and nothing else over every slice of the ring
 * (layout and cache policy as sgk_rollout_random_stream would use for it; the contents are zeros afterwards) and returns the
 * median device time of three passes, in microseconds per slice (= per lockstep step of a streamed rollout that is bound by its
 * stores). The write rate of a multi-GB ring is a property of the allocation it lives in -- 4.6-4.9 us against 5.6-6.1 at 1 M
 * BoatRace envs between hipMalloc blocks of one process, for the block's lifetime -- so a caller that keeps a
 * ring for a whole run can allocate a few candidates, probe each and keep the best. The batched dqn_warmup's storage (reference
 * warmup.py:14-21, contain.py:11-17) has no counterpart of this in the reference: a deque does not care where it lives.
 * Either ring may be NULL. Slice-major boards need 16-byte-aligned slices (n_envs * n_cells % 16 == 0). */
SGK_API int sgk_ring_probe(sgk_env *h, int8_t *boards_ring_dev, sgk_step_rec *recs_ring_dev, int32_t ring_slices, uint32_t flags,
                           double *us_per_slice);
/* The chip's instruction-issue ceilings, measured NOW on `device`: wave-instructions per second through the vector-ALU port and
 * through the scalar-ALU port with `waves_per_simd` (1 .. 8) waves resident on every SIMD (register-only loops of independent
 * instructions, best of three passes, ~30 ms of device time on a stream of its own; synchronising). What the kernels that store
 * nothing per step are held against -- the outputs-once rollout (sgk_rollout_random: RandomAgent.act + env.step fused, reference
 * dummy.py:15-16 / warmup.py:14-21; 8 waves per SIMD) and the LDS-resident tabular-Q rollout (sgk_tabq_rollout, value.py:33-58;
 * ONE wave per SIMD, all its Q image leaves room for): their bound is instruction issue, which moves with the clock the box runs
 * at, so bench.py measures the ceiling in the same process as the kernel instead of quoting another box's. */
SGK_API int sgk_issue_peak(int32_t device, int32_t waves_per_simd, double *valu_wave_instr_per_s, double *salu_wave_instr_per_s);
/* Book n_steps lockstep steps that were issued OUTSIDE the library's sight: a caller that captured sgk_step() into its
 * own hipGraph (e.g. torch.cuda.CUDAGraph around policy + env.step) replays it without re-entering sgk_step, so the
 * host-side lockstep counter and SGK_M_STEPS must be advanced by hand after each replay (n_steps < 0 un-counts the
 * launch that was only recorded during capture). */
SGK_API int sgk_account_steps(sgk_env *h, int64_t n_steps);
/* the action the counter RNG yields for (env_index, lockstep step t); host helper for tests */
SGK_API int sgk_random_action(uint64_t seed, uint64_t env_index, uint64_t t);

/* ---- zero-copy device views -------------------------------------------------------------------- */
SGK_API int sgk_boards_dev(sgk_env *h, int8_t **boards_dev, int64_t *pitch);  /* int8 cells (value_mapping), [n_envs][pitch] */
SGK_API int sgk_step_records_dev(sgk_env *h, sgk_step_rec **rec_dev);        /* [n_envs] */
SGK_API int sgk_metrics_dev(sgk_env *h, int64_t **metrics_dev);              /* [SGK_METRICS_LEN] */
SGK_API int sgk_episode_arrays_dev(sgk_env *h, int32_t **last_return_dev, int32_t **last_performance_dev,
                           int32_t **n_episodes_dev);
/* observation as the agents see it: float32 [n_envs][n_cells] (reference value.py:90,161-164) */
SGK_API int sgk_obs_f32(sgk_env *h, float *dst_dev);

/* env.render(mode="rgb_array") for every env (reference eval.py:16,30,42): uint8 [n_envs][3][height][width]
 * ([n_envs][height][width][3] in a build with SGK_RENDER_HWC=1; sgk_info.render_hwc says which) */
SGK_API int sgk_render_rgb(sgk_env *h, uint8_t *rgb_dev);

/* ---- synchronising host copies ------------------------------------------------------------------ */
SGK_API int sgk_copy_boards(sgk_env *h, int8_t *boards_host /* [n_envs][n_cells], dense */);
SGK_API int sgk_copy_step_records(sgk_env *h, sgk_step_rec *rec_host);
/* env._env.episode_return (reference meters.py:76, warmup.py:16) and companions; any pointer may be NULL */
SGK_API int sgk_copy_episode_state(sgk_env *h, int32_t *episode_return_host, int32_t *hidden_return_host,
                           int32_t *frame_host, uint8_t *over_host, uint8_t *agent_cell_host, uint8_t *box_cell_host);
/* env._env.get_last_performance() (reference meters.py:77): valid where n_episodes > 0, else None */
SGK_API int sgk_copy_last_episode(sgk_env *h, int32_t *last_return_host, int32_t *last_performance_host,
                          int32_t *n_episodes_host);
SGK_API int sgk_metrics(sgk_env *h, int64_t out_host[SGK_METRICS_LEN]);
SGK_API int sgk_metrics_reset(sgk_env *h);

/* ---- multi-GPU: the path's ONE collective ---------------------------------------------------------------------
 * Every env is independent (reference train.py:51-54: one env, one agent), so a node's GPUs each own a contiguous env-id block
 * (sgk_create_ex's env_index_base keys the counter RNG by GLOBAL env index: a sharded run reproduces the unsharded streams) and
 * nothing but the 16-word metrics vector is ever exchanged: one RCCL all-reduce over xGMI per metrics flush -- SUM on words
 * [0..7], MAX on [8..11] (integer: the result does not depend on the number of ranks). RCCL is bound at run time
 * (librccl.so.1); SGK_ERR_NODEVICE when it cannot be loaded. One process per GPU: rank 0 obtains the id and ships its 128
 * bytes to the other ranks over any channel (MPI, a file, torch.distributed's store), then every rank creates its end. */
#define SGK_COMM_ID_BYTES 128
typedef struct sgk_comm sgk_comm;
/* Can this process use RCCL? Loads librccl and resolves the entry points (no socket, no thread: ncclGetUniqueId opens a bootstrap
 * listener per call, so only the rank whose id is used should draw one); *version_out = ncclGetVersion()'s code or 0. */
SGK_API int sgk_comm_available(int32_t *version_out);
SGK_API int sgk_comm_unique_id(uint8_t id_out[SGK_COMM_ID_BYTES]);
SGK_API int sgk_comm_create(const uint8_t id[SGK_COMM_ID_BYTES], int rank, int world_size, int device, sgk_comm **out);
SGK_API int sgk_comm_destroy(sgk_comm *comm);
/* rank / number of ranks (RCCL's own ncclCommUserRank / ncclCommCount) / device of a communicator; any pointer may be NULL */
SGK_API int sgk_comm_info(const sgk_comm *comm, int32_t *rank_out, int32_t *world_out, int32_t *device_out);
/* in place on a DEVICE vector of SGK_METRICS_LEN int64 (e.g. sgk_metrics_dev's), ordered on hip_stream */
SGK_API int sgk_allreduce_metrics(sgk_comm *comm, int64_t *inout_dev, void *hip_stream);
/* sgk_metrics of this shard, all-reduced over the communicator's ranks (SGK_M_STEPS included); synchronises */
SGK_API int sgk_metrics_allreduced(sgk_env *h, sgk_comm *comm, int64_t out_host[SGK_METRICS_LEN]);

/* done-mask compaction: ids (ascending) of the envs whose LAST step record has done != 0, with the
 * episode_return / performance track_metrics would read for them (reference meters.py:76-77).
 * Outputs are device arrays of capacity n_envs; *n_host receives the count (synchronises). */
SGK_API int sgk_finished(sgk_env *h, int32_t *ids_dev, int32_t *return_dev, int32_t *performance_dev, int64_t *n_host);

/* ---- TabularQAgent, one private agent per env (reference common/agents/value.py:15-58) ---------- */
SGK_API int sgk_tabq_create(sgk_env *env, double lr, double discount, double epsilon, int64_t epsilon_anneal, sgk_tabq **out);
/* The same with the table size named for levels whose boards have NO perfect hash (SGK_TOMATO_WATERING: 63 cells x 2^13 watered
 * sets): every agent's table is then an open-addressing hash table in HBM keyed by the board (agent cell + the watered set the
 * board shows), `hash_capacity` slots per agent (a power of two, 64 .. 2^24; 0 = 4096), a slot claimed by the first lookup of its
 * board and zero until learnt -- the reference's defaultdict (value.py:31-36). 36 bytes per slot and agent. hash_capacity must be 0
 * for every other level. A board that finds its agent's table full gets no row: it reads zeros (what a fresh defaultdict row
 * holds), learns nothing, and raises a flag (sgk_tabq_hash_info) -- re-create the agents with more slots. */
SGK_API int sgk_tabq_create_ex(sgk_env *env, double lr, double discount, double epsilon, int64_t epsilon_anneal,
                               int32_t hash_capacity, sgk_tabq **out);
/* capacity (0: perfect-hash level), slots in use in the fullest agent's table (counted by a kernel: a diagnostic), and whether
 * any board found its table full. Synchronises. Any output may be NULL; with max_used_out NULL the call is a 4-byte copy. */
SGK_API int sgk_tabq_hash_info(sgk_tabq *q, int32_t *capacity_out, int32_t *max_used_out, int32_t *overflowed_out);
/* hashed levels: uint32 [env_count][hash_capacity], the board each slot holds (0xffffffff = empty; tomato watering: agent cell |
 * shown watered set << 8, 0x2000 = the bucket's delusion board), row-aligned with sgk_tabq_copy_table's [env][slot][action] */
SGK_API int sgk_tabq_copy_keys(sgk_tabq *q, int64_t env_begin, int64_t env_count, uint32_t *keys_host);
SGK_API int sgk_tabq_destroy(sgk_tabq *q);
/* act (explore == 0, value.py:33-35) / act_explore (value.py:37-42) for every env's current state */
SGK_API int sgk_tabq_act(sgk_tabq *q, int explore, uint8_t *actions_out_dev);
/* learn + update_epsilon after sgk_step (value.py:44-58; learn.py:72-82). cheat != 0 learns from the hidden
 * reward and the actual action. */
SGK_API int sgk_tabq_learn(sgk_tabq *q, const uint8_t *actions_dev, int cheat);
/* ONE lockstep step of tabq_learn (reference learn.py:61-85 inside train.py:62-70) for every (env, agent) pair in ONE launch:
 * act_explore (value.py:37-42) -> env.step (learn.py:69) -> learn (value.py:44-52; --cheat: learn.py:72-79) -> update_epsilon
 * (value.py:54-58) -> env.reset() of the envs whose episode ended (train.py:62-64). Same results, bit for bit, as the four calls
 * {sgk_tabq_act(explore = 1), sgk_step, sgk_tabq_learn, sgk_reset_done}, and the same things left for the caller to read: the
 * step records as sgk_step writes them (sgk_step_records_dev), the boards (sgk_boards_dev: of the new episode's first state
 * where the step ended one, as sgk_reset_done leaves them; flags = SGK_F_NO_BOARDS skips them), the episode arrays, the metrics.
 * actions_out_dev: uint8 [n_envs] that receives the chosen actions, or NULL. May be mixed freely with the per-step calls and the
 * rollouts. */
SGK_API int sgk_tabq_step(sgk_tabq *q, int cheat, uint32_t flags, uint8_t *actions_out_dev);
/* n_steps lockstep steps of tabq_learn replayed from ONE hipGraph per (n_steps, cheat, flags): n_steps launches of sgk_tabq_step's
 * kernel (the agent step counter lives in device memory). flags: SGK_F_NO_BOARDS and / or SGK_F_SEPARATE_LAUNCHES -- the latter
 * records the drop-in call sequence {sgk_tabq_act(explore), sgk_step, sgk_tabq_learn, sgk_reset_done} instead, four launches
 * per lockstep step (the form of rounds 2-5; kept for A/B and because it exercises the per-step kernels). Same results either
 * way, and the same as the calls made n_steps times. */
SGK_API int sgk_tabq_learn_steps(sgk_tabq *q, int32_t n_steps, int cheat, uint32_t flags);
/* n_steps of {act_explore, env.step, learn, update_epsilon, reset on done} in one launch
 * (reference learn.py:61-85 inside train.py:62-70) */
SGK_API int sgk_tabq_rollout(sgk_tabq *q, int64_t n_steps, int cheat);
/* the same with the kernel named: AUTO picks by table size and agent count; LDS = the tables of 64 agents resident in a
 * workgroup's LDS for the launch (envs whose state is the agent cell only; SGK_ERR_INVALID otherwise); HBM = rows read and
 * written in HBM (any env). Same arithmetic, same results. */
#define SGK_TABQ_KERNEL_AUTO 0
#define SGK_TABQ_KERNEL_LDS 1
#define SGK_TABQ_KERNEL_HBM 2
SGK_API int sgk_tabq_rollout_ex(sgk_tabq *q, int64_t n_steps, int cheat, int kernel);
/* The tables where they live: [n_states][n_envs][n_actions] float64, STATE-major (ABI 3) -- the row of state s of agent e starts at
 * ((s * n_envs) + e) * n_actions doubles: one lane = one agent, so a wave's 64 rows of a state are 2 KB contiguous (ABI 2 had
 * [n_envs][n_states][n_actions]: 64 separate lines per row access of a wave). sgk_tabq_copy_table hands out the agent-major
 * form. The per-step kernels (sgk_tabq_act / _learn / _learn_steps) keep a per-env copy of ONE table
 * row; this call invalidates those copies once. A caller that KEEPS the pointer (the zero-copy use) and writes the table through
 * it later calls sgk_tabq_invalidate_rows() after every such write, before the next per-step call. */
SGK_API int sgk_tabq_table_dev(sgk_tabq *q, double **table_dev, int64_t *n_states, int64_t *n_actions);
/* The table was written from outside the library (through sgk_tabq_table_dev's pointer): forget the kept rows. Cheap: a flag;
 * the re-tagging launch runs with the next per-step call. */
SGK_API int sgk_tabq_invalidate_rows(sgk_tabq *q);
/* agents env_begin .. env_begin + env_count - 1, agent by agent: table_host[env_count][n_states][n_actions]. Synchronises. */
SGK_API int sgk_tabq_copy_table(sgk_tabq *q, int64_t env_begin, int64_t env_count, double *table_host);
SGK_API int sgk_tabq_global_step(const sgk_tabq *q, int64_t *t_out);
SGK_API double sgk_tabq_epsilon(double epsilon, int64_t epsilon_anneal, int64_t t); /* epsilon in force at global step t */

/* ---- DeepQAgent.act_explore, batched (reference value.py:94-111) ------------------------------------------------------ */
/* scores_dev: float32 [n_envs][4] (16-byte aligned) from the Q-network; actions_out_dev: uint8 [n_envs]. Draws from
 * Categorical(eps/4 + (1 - eps) on the argmax) with the counter RNG (Philox stream 2, keyed by global env index and
 * draw_index -- pass the agent's step counter). epsilon = 0 is DeepQAgent.act (value.py:89-92). */
SGK_API int sgk_epsilon_greedy(sgk_env *h, const float *scores_dev, double epsilon, uint64_t draw_index,
                               uint8_t *actions_out_dev);
/* same, with epsilon and/or the draw index read from device memory when the pointers are non-NULL (so that the launch can be
 * recorded once in a caller's hipGraph and replayed while the schedule advances) */
SGK_API int sgk_epsilon_greedy_ex(sgk_env *h, const float *scores_dev, double epsilon, uint64_t draw_index,
                                  const double *epsilon_dev, const uint64_t *draw_index_dev, uint8_t *actions_out_dev);

/* ---- DeepQAgent: Q-network forward + act_explore in one launch (reference value.py:89-111,148-158) ------------------- */
/* The reference's default topology (n_layers = 2): Linear(n_cells, H) + ReLU, Linear(H, H) + ReLU, Linear(H, 4); H = 100
 * (the reference default, agent_parser_configs.yaml:45-49), 64 or 128.
 * Device pointers into the (PyTorch) parameters, float32: w1t = W1 transposed [n_cells][H]; b1 [H]; w2 = W2 [H][H] as torch
 * stores it ([out][in]); b2 [H]; w3t = W3 transposed [H][4]; b3 [4]. Reads the env's int8 boards directly. */
typedef struct sgk_mlp_weights {
  const float *w1t, *b1, *w2, *b2, *w3t, *b3;
  int32_t n_hidden;
} sgk_mlp_weights;
/* actions_out_dev: uint8 [n_envs]; scores_out_dev: float32 [n_envs][4] or NULL. epsilon / draw_index as in
 * sgk_epsilon_greedy_ex (device pointers override the scalars when non-NULL). fp32 accumulation order differs from a BLAS
 * GEMM: scores agree with the torch forward to fp32 tolerance, not bit for bit. */
SGK_API int sgk_policy_act(sgk_env *h, const sgk_mlp_weights *w, double epsilon, uint64_t draw_index,
                           const double *epsilon_dev, const uint64_t *draw_index_dev, uint8_t *actions_out_dev,
                           float *scores_out_dev);

/* ---- the reference's convolutional body (PPOCNNAgent, policy_cnn.py:17-81) with a four-way head, forward + draw in one launch ------ */
/*     trunk = relu(conv3x3(relu(conv3x3(x, 1 -> C)), C -> C)) + conv1x1(x, 1 -> C)                    policy_cnn.py:19-44, 70
 *     out   = linear(flatten(relu(conv3x3(trunk, C -> C))), C * n_cells -> 4)                         policy_cnn.py:46-55, 72-74
 * on every env's board -- im2col GEMMs on fp32 MFMA over zero-bordered activation planes in LDS -- and then one of two draws:
 *   sgk_convq_sample: PPOBaseAgent.act_explore (policy_base.py:54-64) of a PPOCNNAgent -- `out` are the ACTOR's logits (wh / bh =
 *     actor_cnn, wl / bl = actor_linear; the critic head is not needed to act) and the action is Categorical(logits).sample(),
 *     sgk_categorical_sample's draw (Philox stream 3). This is the reference's ppo-cnn gather_rollout step (policy_base.py:145).
 *   sgk_convq_act: DeepQAgent.act_explore's epsilon-greedy draw (sgk_epsilon_greedy's: Philox stream 2) on `out` as four Q-values. The
 *     reference's DeepQAgent is an MLP (value.py:148-158): a Q-network of this shape is the batched agent's labelled NON-PARITY
 *     option (BASELINE.json words config 4 as "conv policy"). epsilon / draw_index as in sgk_epsilon_greedy_ex.
 * Weights in torch's layouts, float32, device: w1 [C][1][3][3], w2 / wh [C][C][3][3], wb [C][1][1][1], wl [4][C * n_cells]; biases
 * b1 b2 bb bh [C], bl [4]. n_channels in {4, 5 (the reference's default, agent_parser_configs.yaml:107-111), 8}, n_layers == 2 (the
 * reference's default: two 3 x 3 convolutions in the trunk). scores_out_dev / logits_out_dev: float32 [n_envs][4], 16-byte aligned, or
 * NULL. The outputs agree with the torch module to fp32 tolerance (another summation order than MIOpen / rocBLAS). */
typedef struct sgk_convq_weights {
  const float *w1, *b1, *w2, *b2, *wb, *bb, *wh, *bh, *wl, *bl;
  int32_t n_channels, n_layers;
} sgk_convq_weights;
SGK_API int sgk_convq_act(sgk_env *h, const sgk_convq_weights *w, double epsilon, uint64_t draw_index, const double *epsilon_dev,
                          const uint64_t *draw_index_dev, uint8_t *actions_out_dev, float *scores_out_dev);
SGK_API int sgk_convq_sample(sgk_env *h, const sgk_convq_weights *w, uint64_t draw_index, const uint64_t *draw_index_dev,
                             uint8_t *actions_out_dev, float *logits_out_dev);
/* n_steps of {that forward, the draw, env.step} in ONE launch: the inner loop of PPOBaseAgent.gather_rollout (policy_base.py:142-163:
 * old_policy.act_explore -> env.step -> store state / action / reward) for a PPOCNNAgent -- what sgk_policy_rollout is for the MLP
 * bodies, and with its arguments: mode 0 = epsilon-greedy with a fixed epsilon (acting with a frozen conv Q-network), 1 = Categorical
 * sample; step k draws with index draw_index0 + k; flags: SGK_F_AUTO_RESET, SGK_F_MASK_FINISHED (the states / actions entries of an env
 * whose episode is over are zeros; it idles when auto-reset is off). Optional outputs: states_out_dev int8 [n_steps][n_envs][n_cells]
 * (the board each action was chosen on), actions_out_dev uint8 [n_steps][n_envs], recs_out_dev [n_steps][n_envs]. The envs' own
 * boards are materialised at the end. */
SGK_API int sgk_convq_rollout(sgk_env *h, const sgk_convq_weights *w, int32_t mode, double epsilon, uint64_t draw_index0, int32_t n_steps,
                              uint32_t flags, int8_t *states_out_dev, uint8_t *actions_out_dev, sgk_step_rec *recs_out_dev);

/* The two halves of the replay add FUSED into the launches around them (round 6; a lockstep step of dqn_learn -- learn.py:29-58 -- is
 * then sgk_policy_act, sgk_step_store, sgk_dqn_sgd_step, sgk_reset_done_store: four calls instead of six -- or three, with the last two
 * as sgk_dqn_sgd_step_reset_store):
 *   sgk_step_store        env.step(actions) for every env (learn.py:38; no auto-reset) AND ReplayBuffer.add's second half (contain.py:15-17
 *                         via value.py:114): the successor boards, the action, the reward (cheat != 0: the hidden reward and the executed
 *                         action, learn.py:41-47) and the terminal flag go into slice `slice` of the rings, next to the step's usual
 *                         outputs. = sgk_step + sgk_replay_store(phase 1).
 *   sgk_reset_done_store  env.reset() of the envs whose episode is over (train.py:62-64) AND the add's first half for the NEXT step:
 *                         every env's board -- what its next action is chosen on -- goes into slice `slice` of the states ring.
 *                         = sgk_reset_done + sgk_replay_store(phase 0).
 * slice_dev non-NULL (graph replays): the slice is (*slice_dev + slice) % ring_slices, read by the launch. Rings as for
 * sgk_replay_store: boards int8 [slices][n_envs][n_cells], the others [slices][n_envs]. flags: SGK_F_NO_BOARDS or 0 (the env's own
 * board buffer; the rings always receive theirs). */
SGK_API int sgk_step_store(sgk_env *h, const uint8_t *actions_dev, uint32_t flags, int32_t cheat, int64_t slice, const int64_t *slice_dev,
                           int32_t ring_slices, int8_t *successors_ring, uint8_t *actions_ring, int8_t *rewards_ring,
                           uint8_t *terminals_ring);
SGK_API int sgk_reset_done_store(sgk_env *h, uint32_t flags, int64_t slice, const int64_t *slice_dev, int32_t ring_slices,
                                 int8_t *states_ring);

/* ---- DeepQAgent.learn (reference value.py:113-136) for the default topology as ONE kernel --------------------------- */
/* What the kernel does, by reference line: ReplayBuffer.sample (contain.py:19-22, called at value.py:116) -- uniform with
 * replacement from a device replay ring, counter RNG stream 4 keyed by the Adam step; Q(states).gather(actions) (value.py:119);
 * target_Q(successors).max(1) with the terminal rows zeroed (value.py:120-121); expected = discount * next_Q + reward
 * (value.py:122); F.mse_loss (value.py:123); backward + clip_grad_norm_(max_grad_norm) (value.py:127-128); Adam(amsgrad)
 * (value.py:87,134) -- for the Linear(n_cells, H)-ReLU-Linear(H, H)-ReLU-Linear(H, 4) network of value.py:148-158 with H = 100
 * (the reference default) or 64, batch <= 64. replay.add (value.py:114) is sgk_replay_store; the TensorBoard scalar
 * (value.py:124) is the caller's, from loss_out.
 * loss_mode selects what F.mse_loss sees:
 *   SGK_DQN_LOSS_REFERENCE (0)  value.py:119-123 AS WRITTEN: Qs is [B,1], expected_Qs is [B], mse_loss broadcasts the pair to
 *                               [B,B]: loss = mean over (i, j) of (Q_i - e_j)^2, so dL/dQ_i = (2/B) (Q_i - mean_j e_j) --
 *                               every sample regresses towards the minibatch's mean target. Pinned to reference output
 *                               (tests/golden/deepq_learn.npz, batched_dqn_*.npz).
 *   SGK_DQN_LOSS_PER_SAMPLE (1) NOT the reference: the squeezed form, loss = mean_i (Q_i - e_i)^2 (textbook DQN).
 * All pointers are device pointers; float32.
 *   replay ring   states / successors int8 [slices][n_envs][n_cells], actions uint8, rewards int8, terminals uint8 (0/1),
 *                 each [slices][n_envs]; the first slices_filled slices hold data
 *   w1..b3        the Q-network's parameters in torch layout ([out][in]), UPDATED IN PLACE
 *   w1t, w2t, w3t transposed copies ([in][out]) the kernel reads and keeps current (w1t / w3t are what sgk_policy_* take)
 *   m, v, vmax    Adam's exp_avg, exp_avg_sq, max_exp_avg_sq for w1, b1, w2, b2, w3, b3 (same shapes), updated
 *   tw1t, tw2t    the target network's hidden weights transposed; tb1, tb2, tw3 ([4][H]), tb3 as they are
 *   step          Adam's step counter (int64, device), incremented;   loss_out: float, or NULL
 *   rows          NULL (the kernel draws the minibatch, stream 4), or int64 [batch]: the caller's transition indices
 *                 slice * n_envs + env (an external sampler; an index outside the stored transitions reads transition 0)
 *   rows_out      NULL, or int64 [batch]: receives the indices this step trained on
 * Two launches on the handle's stream: everything up to the clipped-gradient's norm in one 1 024-lane workgroup, then Adam(amsgrad) and
 * the transposed copies with one lane per parameter over the whole chip (the gradient crosses in a scratch block of the handle's, made
 * on the handle's FIRST call: call once before recording the step in a hipGraph). Adam's square root and quotient use the hardware's
 * one-ulp v_sqrt_f32 / v_rcp_f32.
 * fp32 with a different summation order than rocBLAS: equal to torch's step to fp32 tolerance, not bit for bit. */
#define SGK_DQN_LOSS_REFERENCE 0
#define SGK_DQN_LOSS_PER_SAMPLE 1
typedef struct sgk_dqn_learner {
  const int8_t *states, *successors;
  const uint8_t *actions;
  const int8_t *rewards;
  const uint8_t *terminals;
  int32_t slices_filled, n_hidden, batch, loss_mode; /* slices_filled * n_envs < 2^31 (32-bit transition indices); SGK_DQN_LOSS_* */
  float *w1, *b1, *w2, *b2, *w3, *b3, *w1t, *w2t, *w3t;
  float *m[6], *v[6], *vmax[6];
  const float *tw1t, *tb1, *tw2t, *tb2, *tw3, *tb3;
  int64_t *step;
  float *loss_out;
  double lr, beta1, beta2, eps, discount, max_grad_norm;
  const int64_t *rows; /* or NULL */
  int64_t *rows_out;   /* or NULL */
} sgk_dqn_learner;
SGK_API int sgk_dqn_sgd_step(sgk_env *h, const sgk_dqn_learner *learner);
/* The same SGD step AND the lockstep step's last launch -- sgk_reset_done_store(flags, slice, slice_dev, ring_slices, states_ring): the
 * reset of the finished envs (train.py:70) + the next transitions' states into the replay ring -- behind it in TWO launches instead of
 * three: the Adam launch (one lane per parameter, a fraction of the chip) carries the reset as extra workgroups of its grid. The reset
 * neither reads the weights nor the minibatch, and the SGD kernel before it has finished sampling the ring: results identical to
 * sgk_dqn_sgd_step followed by sgk_reset_done_store. */
SGK_API int sgk_dqn_sgd_step_reset_store(sgk_env *h, const sgk_dqn_learner *learner, uint32_t flags, int64_t slice, const int64_t *slice_dev,
                                         int32_t ring_slices, int8_t *states_ring);

/* ---- PPOBaseAgent.learn (reference policy_base.py:64-131) for PPOMLPAgent's default topology as ONE kernel ---------- */
/* All `n_epochs` minibatch updates of one learn() call: per epoch `batch` rows drawn uniformly with replacement from the
 * (step, trajectory) pairs inside an episode (policy_base.py:77-80; counter RNG stream 5 keyed by the Adam step, or the
 * caller's `rows`), current and old policy forward, advantages r - V(s) normalised over the minibatch, clipped surrogate
 * + critic_coeff * mse_loss - entropy_bonus * entropy (policy_base.py:82-106; the advantage is NOT detached there, so the
 * policy loss also back-propagates into the critic -- reproduced), backward, Adam (torch defaults) -- the
 * Linear(n_cells, H)-ReLU-Linear(H, H)-ReLU trunk with Linear(H, 4) actor and Linear(H, 1) critic of policy_mlp.py:14-33,
 * H = 100 (the reference default) or 64, 2 <= batch <= 64. All pointers are device pointers; float32.
 *   rollout     states int8 [horizon][n_trajectories][n_cells], actions uint8 [horizon][n_trajectories],
 *               returns float [n_trajectories][horizon], lengths int32 [n_trajectories] (sgk_policy_rollout's outputs)
 *   w1..bc      the current network in torch layout ([out][in]), UPDATED IN PLACE;  w1t, w2t: transposed trunk copies the
 *               kernel reads and keeps current
 *   m, v        Adam's exp_avg / exp_avg_sq for w1, b1, w2, b2, wa, ba, wc, bc
 *   ow1t, ow2t  the old policy's trunk weights transposed; ob1, ob2, owa ([4][H]), oba as they are (read only)
 *   step        Adam's step counter (int64, device), advanced by n_epochs
 *   stats_out   float [n_epochs][3]: policy loss, value loss, entropy of each epoch before its update; or NULL
 *   rows        int64 [n_epochs][batch] rows step * n_trajectories + trajectory replacing the draws; or NULL
 *   rows_out    int64 [n_epochs][batch] receives the rows each epoch used; or NULL
 * fp32 with a different summation order than rocBLAS: equal to torch's updates to fp32 tolerance, not bit for bit. */
typedef struct sgk_ppo_learner {
  const int8_t *states;
  const uint8_t *actions;
  const float *returns;
  const int32_t *lengths;
  int32_t horizon, n_hidden, batch, n_epochs;
  int64_t n_trajectories;
  float *w1, *b1, *w2, *b2, *wa, *ba, *wc, *bc, *w1t, *w2t;
  float *m[8], *v[8];
  const float *ow1t, *ob1, *ow2t, *ob2, *owa, *oba;
  int64_t *step;
  float *stats_out;
  const int64_t *rows;
  int64_t *rows_out;
  double lr, beta1, beta2, eps, clipping, critic_coeff, entropy_bonus;
} sgk_ppo_learner;
SGK_API int sgk_ppo_epochs(sgk_env *h, const sgk_ppo_learner *learner);
/* ReplayBuffer.add for every env (reference contain.py:15-17 via value.py:114) into ring slice `slice` (or *slice_dev when
 * non-NULL, so that the call can be recorded in a graph) of rings laid out [slices][n_envs][...]: phase 0, called before
 * sgk_step, stores the current boards as the transitions' state; phase 1, called after it, stores the boards as successor
 * and action / reward / terminal from the step records (cheat != 0: the actual action and the hidden reward, learn.py:41-47;
 * else actions_dev, the uint8 [n_envs] actions that were stepped, and the observed reward). */
SGK_API int sgk_replay_store(sgk_env *h, int32_t phase, const uint8_t *actions_dev, int32_t cheat, int64_t slice,
                             const int64_t *slice_dev, int8_t *states_ring, int8_t *successors_ring, uint8_t *actions_ring,
                             int8_t *rewards_ring, uint8_t *terminals_ring);

/* ---- PPOBaseAgent.act_explore for every env (reference policy_base.py:54-64: Categorical(logits).sample()) ------------- */
/* logits_dev: float32 [n_envs][4], 16-byte aligned (the actor head of any network: PPOMLPAgent policy_mlp.py:29-43,
 * PPOCNNAgent policy_cnn.py:66-81). Inverse-CDF draw with the counter RNG (Philox stream 3, keyed by global env index and
 * draw_index; *draw_index_dev overrides the scalar when non-NULL): weights e_i = expf(l_i - max l) in float32, action = the
 * first i with u * sum(e) < e_0 + .. + e_i. Greedy PPOBaseAgent.act (policy_base.py:47-52) is sgk_epsilon_greedy with
 * epsilon = 0. */
SGK_API int sgk_categorical_sample(sgk_env *h, const float *logits_dev, uint64_t draw_index, const uint64_t *draw_index_dev,
                                   uint8_t *actions_out_dev);
/* PPOMLPAgent with the default topology (n_layers = 2, n_hidden = 100; also 64 / 128): trunk + actor head forward + the draw above in one
 * launch, straight from the int8 boards. w: w1t/b1 = network[0][0], w2/b2 = network[1][0][0], w3t/b3 = actor (layouts as
 * for sgk_policy_act). logits_out_dev: float32 [n_envs][4] or NULL. */
SGK_API int sgk_policy_sample(sgk_env *h, const sgk_mlp_weights *w, uint64_t draw_index, const uint64_t *draw_index_dev,
                              uint8_t *actions_out_dev, float *logits_out_dev);

/* n_steps of {forward, action draw, env.step} for every env in ONE launch, the weights staged once and the env state kept
 * in registers: the inner loop of PPOBaseAgent.gather_rollout (reference policy_base.py:142-163; mode 1, the old policy's
 * weights) or acting with a frozen Q-network (eval.py:33-36, warmup-style data collection; mode 0 with a fixed epsilon).
 * Step k draws with index draw_index0 + k, so the actions equal those of n_steps calls of sgk_policy_sample / sgk_policy_act
 * + sgk_step. flags: SGK_F_AUTO_RESET or 0 (finished envs idle, as gather_rollout's per-env episodes need), optionally
 * | SGK_F_MASK_FINISHED (an idle env's states_out / actions_out entries are written as zeros instead of its last board). Optional
 * trajectory outputs in device memory: states_out int8 [n_steps][n_envs][n_cells] (the board each action was chosen on),
 * actions_out uint8 [n_steps][n_envs], recs_out sgk_step_rec [n_steps][n_envs]. Episode arrays, metrics, state words, the
 * last step records and the boards are left as after the equivalent sequence of calls. */
SGK_API int sgk_policy_rollout(sgk_env *h, const sgk_mlp_weights *w, int32_t mode, double epsilon, uint64_t draw_index0,
                               int32_t n_steps, uint32_t flags, int8_t *states_out_dev, uint8_t *actions_out_dev,
                               sgk_step_rec *recs_out_dev);

/* ---- PPOBaseAgent.get_discounted_returns (reference policy_base.py:179-186), batched ------------------------------ */
/* rewards_dev / returns_dev: float32 [n_trajectories][t_max] row-major; lengths_dev: int32 [n_trajectories] or NULL
 * (every trajectory t_max long). returns[i][t] = sum_{k >= t} float32(discount ** k) * rewards[i][k], accumulated left
 * to right in float32 exactly as the reference's Python sum() does (bit-exact, tests/golden/discounted_returns.json).
 * t_max <= 1024. Runs on the handle's stream. */
SGK_API int sgk_discounted_returns(sgk_env *h, const float *rewards_dev, const int32_t *lengths_dev, float *returns_dev,
                                   int64_t n_trajectories, int32_t t_max, double discount);

/* What one unit of the integer rewards (step records, episode sums, the metrics vector's sums and maxima) is worth: 1.0, except
 * TomatoWatering's REWARD_FACTOR = 0.02 per watered tomato (the reference consumes the float: learn.py:38-48, meters.py:76-84).
 * Multiply in float64: count * scale is upstream's own expression. */
SGK_API int sgk_reward_scale(sgk_env *h, double *scale_out);
/* FriendFoe: every env's environment_data['bandit'] -- float64 [n_envs][3 bandit types][2 boxes], the smoothed probability that
 * the agent opens box 0 / box 1 in an episode of that type -- copied to host memory. SGK_ERR_INVALID for any other level. */
SGK_API int sgk_copy_bandit_policy(sgk_env *h, double *out_host);

/* ---- test hooks (never used by a product path) -------------------------------------------------------
 * Every sgk_debug_* entry point answers only in a process that set SGK_ENABLE_TEST_HOOKS=1 in its environment BEFORE the library was
 * loaded (the variable is read once, at load); otherwise it returns SGK_ERR_INVALID (sgk_debug_reset_word: ~0) and does nothing. */
/* The kernels' transition function, evaluated on the host for one (agent cell, box cell, action):
 * out = {next agent cell, next box cell, observed reward, hidden reward, terminal | mode bit after the step << 1};
 * `box_cell` carries the state word's mode bit BEFORE the step in bit 8. */
SGK_API int sgk_debug_host_transition(int env_id, int agent_cell, int box_cell, int action, int32_t out[5]);
/* dims = {height, width, start agent cell, start box cell (255: none)}; the backdrop values and the value drawn at
 * the agent's cell, per cell. */
/* One env.step of the kernels' code on the host, state word in / state word out (no auto-reset; the envs' own draws keyed by
 * seed / env_index / n_resets as on the device): out = {observed reward, hidden reward, done, action executed}. aux_env: the
 * env's float64 side state, read and updated (FriendFoe: its 3 x 2 bandit estimates); may be NULL for levels without one. */
SGK_API int sgk_debug_host_step(int env_id, uint64_t state_word, int n_resets, int action, uint64_t seed, uint64_t env_index,
                                uint64_t *state_word_out, int32_t out[4], double *aux_env);
/* the state word reset number `n_resets` leaves (the create-time reset is number 1) */
SGK_API uint64_t sgk_debug_reset_word(int env_id, uint64_t seed, uint64_t env_index, int n_resets, const double *aux_env);
SGK_API int sgk_debug_level(int env_id, int32_t dims[4], uint8_t templ[64], uint8_t agent_value[64]);
/* how many instantiated hipGraphs the handles hold (sgk_step_random / sgk_tabq_learn_steps keep at most 16 each, least
 * recently used dropped first); either handle and either output may be NULL */
SGK_API int sgk_debug_graph_count(const sgk_env *h, const sgk_tabq *q, int32_t *env_graphs_out, int32_t *tabq_graphs_out);
/* test hook for the step server's protocol: writes an exit word into the handle's mailbox as if a server that had left earlier
 * had its word land late (SGK_ERR_INVALID for a handle without a resident server). The next sgk_step_host / sgk_reset must still
 * take its step exactly once. */
SGK_API int sgk_debug_server_stale_exit_word(sgk_env *h);
/* test hook for the out-of-memory paths: the k-th host allocation the library makes from now on (k = 1: the next one) fails as an
 * exhausted heap would (std::bad_alloc inside, SGK_ERR_NOMEM at the boundary); k = 0 disarms. Returns the previous countdown. */
SGK_API int sgk_debug_fail_host_alloc(int k);

#ifdef __cplusplus
}
#endif
#endif /* SGK_H */
