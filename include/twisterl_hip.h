/*
 * twisterl_hip.h -- C ABI of the MI355X-native twisteRL episode collector.
 *
 * This is the drop-in boundary for ONE hot path of AI4quantum/twisteRL: the vectorised
 * episode collection (Puzzle step/observe/reward/masks loop + policy forward + Gumbel
 * sampling + GAE + merge), i.e. what `Collector::collect(&Box<dyn Env>, &Policy)`
 * (rust/src/collector/collector.rs:92-95) does for `PPOCollector` / `AZCollector`.
 * Plain pointers and sizes only; no torch / pybind types.  All compute entry points run
 * hand-written HIP kernels on the current gfx950 device and FAIL (status != TW_OK, message in
 * tw_last_error()) when no GPU is available -- there is no CPU fallback in this library.
 *
 * Reference interfaces replaced (paths relative to the reference repo):
 *   tw_puzzle_*            PyBaseEnv / Puzzle pymethods   rust/src/python_interface/env.rs:44-160
 *                          over Env for Puzzle            rust/src/envs/puzzle.rs:81-187
 *   tw_policy_create       Policy::new + Linear/EmbeddingBag/Sequential ctors
 *                                                         rust/src/python_interface/policy.rs:28-31,
 *                                                         layers.rs:29-32,45-48, modules.rs:28-32
 *   tw_policy_evaluate     Policy.predict/forward/full_predict
 *                                                         rust/src/python_interface/policy.rs:33-45
 *   tw_ppo_collect         PyBaseCollector.collect -> PPOCollector::collect
 *                                                         rust/src/python_interface/collector.rs:147-151,
 *                                                         rust/src/collector/ppo.rs:108-126
 *   tw_az_collect          ... -> AZCollector::collect    rust/src/collector/az.rs:112-130
 *   tw_evaluate / tw_solve evaluate_py / solve_py          rust/src/python_interface/env.rs:180-207
 *   tw_collected_*         PyCollectedData getters        rust/src/python_interface/collector.rs:55-136
 *   tw_last_error          anyhow::Error -> PyRuntimeError rust/src/python_interface/error_mapping.rs:20-33
 */
#ifndef TWISTERL_HIP_H
#define TWISTERL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TW_ABI_VERSION 6   /* 6: tw_env_vtable grew track_solution / solution / set_state / twists (the rest of `trait Env`), tw_solve_env32 */

/* status codes */
enum {
    TW_OK              = 0,
    TW_ERR_INVALID     = 1,  /* bad argument                                               */
    TW_ERR_UNSUPPORTED = 2,  /* env / policy shape outside what the HIP path implements    */
    TW_ERR_NO_DEVICE   = 3,  /* no usable gfx950 device                                    */
    TW_ERR_HIP         = 4,  /* a HIP runtime call failed (message has the HIP error)      */
    TW_ERR_EMPTY       = 5   /* zero episodes: "No data in collected data chunks to merge" */
};

/* arithmetic of the policy forward inside the collectors */
enum {
    TW_PREC_F32_EXACT = 0,  /* f32 MFMA, k-ordered fma chain: bit-equal to the oracle      */
    TW_PREC_F16       = 1,  /* fp16-input MFMA, f32 accumulate (throughput mode)           */
    TW_PREC_F16X2     = 2   /* f32 operands split into two f16 terms each: f32-equivalent  */
};

int         tw_abi_version(void);
const char *tw_last_error(void);              /* thread-local, valid until the next call    */
int         tw_device_count(void);
int         tw_set_device(int device);
int         tw_set_stream(void *hip_stream);  /* launch on this hipStream_t (NULL = default)*/

typedef struct {
    char     name[128];
    char     arch[64];
    int32_t  compute_units;
    int32_t  wavefront_size;
    uint64_t total_mem_bytes;
    uint64_t lds_bytes_per_block;
} tw_device_info;
int tw_get_device_info(tw_device_info *out);
/* Frees the cached device memory of the current process (trajectory workspace, pooled result arenas). */
int tw_release_cached_memory(void);

/* Diagnostic launch overrides -- test hooks (the parity tests pin launch shapes with them to show that every shape gives
 * the same bytes); process-wide, 0 restores the default.  Not needed by a host of the reference's collector. */
enum {
    TW_OPT_FORCE_GEOM = 0,  /* 8: the 256-episode workgroup shape, 32: the 32-episode shape, 1: small-batch, 0: automatic */
    TW_OPT_NO_PERSIST = 1,  /* 1: never use persistent lanes + episode queue                                               */
    TW_OPT_AZ_TREE_BUDGET_MIN = 4, /* walker kernel: cycles of tree walk after which a walker yields once another one waits; 0: auto */
    TW_OPT_AZ_TREE_BUDGET = 3, /* walker kernel: cycles of tree walk per forward before a walker yields; 0: automatic           */
    TW_OPT_AZ_REUSE = 5,    /* lane-per-episode self-play kernel, how a node whose move takes its parent's move back finds   */
                            /* the stored output of its grandparent (same board): 0 product, 1 no reuse, 3 by the link only, */
                            /* 4 by the link while counting where the path level would have differed; 2 (the path level at   */
                            /* any depth: returns different bytes, kept for the record) only in the TW_ABLATE build -- the   */
                            /* product build refuses it with TW_ERR_INVALID                                                  */
    TW_OPT_AZ_VARIANT = 2   /* self-play with few deep searches: 0 automatic (walker-per-wave kernel where it applies),    */
                            /* 2 always the lane-per-episode kernel; walker kernel with a pinned shape: 3 / 4 / 5 / 6 =    */
                            /* two / one / four / eight walkers per workgroup, + 16 / + 32 = the 16- / 32-column engine,   */
                            /* + 64 = the walkers (and the PPO rollouts' persistent lanes) take the episodes by index      */
                            /*        instead of longest-looking first, + 128 / + 256 = the decoupled shape (engine-only   */
                            /*        waves beside the walkers) pinned on / off, + 512 / + 1024 = the split shape (walkers */
                            /*        and engine as two kernels; automatic from eight episodes per CU on) pinned on / off  */
                            /*        -- + 1024 is what a run under a kernel-serialising profiler (rocprofv3 --pmc) needs; */
                            /*        + 2048 (test hook) launches the split shape's two kernels one after the other        */
};
int tw_set_launch_option(int option, int value);
/* Diagnostic counters of the last self-play launch of this process (MctsArgs::eval_count[0..15]); test hook. */
int tw_debug_counters(uint64_t *out, int n);

/* ---------------------------------------------------------------------------------------- */
/* Env: host object with the PyBaseEnv / Puzzle surface.  Collectors only read its          */
/* descriptor (the reference clones + resets the env per episode, ppo.rs:59-60).            */
/* ---------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t width, height, difficulty, depth_slope, max_depth;
} tw_puzzle_desc;
/* Test hook: start boards (nibble i = tile at cell i; Env::reset of episode episode_offset + i, puzzle.rs:119-133) and the order in
   which the self-play walkers take the episodes (decreasing sum of the tiles' Manhattan distances, ties by index), as tw_az_collect
   computes them on the device; boards_out[n], order_out[n] on the host.  Boards of up to 16 cells. */
int tw_debug_episode_order(const tw_puzzle_desc *env, uint64_t seed, uint64_t episode_offset, uint64_t n, uint64_t *boards_out, uint32_t *order_out);

typedef struct tw_puzzle tw_puzzle;

tw_puzzle *tw_puzzle_create(uint32_t width, uint32_t height, uint32_t difficulty,
                            uint32_t depth_slope, uint32_t max_depth);
tw_puzzle *tw_puzzle_clone(const tw_puzzle *p);
void       tw_puzzle_destroy(tw_puzzle *p);
int        tw_puzzle_get_desc(const tw_puzzle *p, tw_puzzle_desc *out);
uint32_t   tw_puzzle_num_actions(const tw_puzzle *p);
int        tw_puzzle_obs_shape(const tw_puzzle *p, uint32_t out[2]);
int        tw_puzzle_set_difficulty(tw_puzzle *p, uint32_t difficulty);
uint32_t   tw_puzzle_get_difficulty(const tw_puzzle *p);
int        tw_puzzle_set_state(tw_puzzle *p, const int64_t *state, size_t n);
int        tw_puzzle_reset(tw_puzzle *p, uint64_t seed, uint64_t episode);
int        tw_puzzle_step(tw_puzzle *p, uint32_t action);
int        tw_puzzle_masks(const tw_puzzle *p, uint8_t out[4]);
int        tw_puzzle_is_final(const tw_puzzle *p);
float      tw_puzzle_reward(const tw_puzzle *p);
int        tw_puzzle_observe(const tw_puzzle *p, int64_t *out /* width*height */);
int        tw_puzzle_solved(const tw_puzzle *p);
int        tw_puzzle_get_state(const tw_puzzle *p, int64_t *out /* width*height */);
int        tw_puzzle_set_position(tw_puzzle *p, uint32_t x, uint32_t y, int64_t val);
int64_t    tw_puzzle_get_position(const tw_puzzle *p, uint32_t x, uint32_t y);
uint32_t   tw_puzzle_depth(const tw_puzzle *p);

/* ---------------------------------------------------------------------------------------- */
/* Policy: weights in the reference's export layout (src/twisterl/nn/utils.py:17-79)        */
/* ---------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t     in_features, out_features;
    const float *weights;   /* [in][out] row-major == torch_weight.T.flatten() (layers.rs:26)  */
    const float *bias;      /* [out]                                                           */
    uint32_t     apply_relu;
} tw_linear_desc;

typedef struct {
    /* EmbeddingBag, 1-D mode (layers.rs:56-62): out = bias + sum_k vectors[obs[k]], ReLU      */
    uint32_t     obs_size, emb_size;
    const float *emb_vectors;   /* [obs_size][emb_size] == torch_weight.T                      */
    const float *emb_bias;      /* [emb_size]                                                  */
    uint32_t     emb_apply_relu;
    uint32_t n_common; const tw_linear_desc *common;
    uint32_t n_action; const tw_linear_desc *action;
    uint32_t n_value;  const tw_linear_desc *value;
    /* twists (policy.rs:25-26): obs_perms[n_perms][obs_size], act_perms[n_perms][n_actions]   */
    uint32_t       n_perms, n_actions;
    const int32_t *obs_perms;
    const int32_t *act_perms;
} tw_policy_desc;

typedef struct tw_policy tw_policy;

/* Copies the weights to the device (host pointers in `desc` need not outlive the call).  Any stack the reference's
 * Policy holds is accepted (rust/src/nn/modules.rs:28-34): 1-D EmbeddingBag (emb_size % 4 == 0, <= 512; obs_size <= 256),
 * up to 8 Linear layers each in `common`, `action` (ending in n_actions outputs) and `value`, widths <= 512.  The shape of
 * both Puzzle configs -- ONE common Linear of 32 / 64 / 128 / 256 units, emb_size % 32 == 0, linear heads -- runs on the
 * MFMA engines (and has the f16 modes and tw_policy_update_device); every other stack runs the generic engine (vector ALU,
 * same arithmetic, f32 only).  NULL + message in tw_last_error() otherwise. */
tw_policy *tw_policy_create(const tw_policy_desc *desc);
void       tw_policy_destroy(tw_policy *p);
/* Device-to-device policy sync (replaces the per-iteration policy.to_rust() round trip through host lists,
 * reference src/twisterl/nn/policy.py:191-199, rl/algorithm.py:90-93): rebuilds every weight image of `p` from
 * the trainer's parameters where they live.  All pointers are DEVICE pointers in torch layout
 * (nn.Linear.weight = [out][in]): embeddings.weight [emb][obs_size], embeddings.bias [emb], common.0.weight
 * [hidden][emb], common.0.bias [hidden], action.0.weight [n_actions][hidden], action.0.bias, value.0.weight
 * [1][hidden], value.0.bias [1].  Shapes, ReLU flags and twists stay those given to tw_policy_create. */
int tw_policy_update_device(tw_policy *p, const float *emb_w, const float *emb_b, const float *w1, const float *b1,
                            const float *wa, const float *ba, const float *wv, const float *bv);
/* The same for a policy of any Sequential depth (the stacks tw_policy_create ran through its generic engine): the Linear
 * layers' torch parameters in the order common.., action.., value.. (weight [out][in], bias [out]), device pointers. */
int tw_policy_update_device_layers(tw_policy *p, const float *emb_w, const float *emb_b, const float *const *weights,
                                   const float *const *biases, uint32_t n_layers);
uint32_t   tw_policy_num_actions(const tw_policy *p);
uint32_t   tw_policy_num_perms(const tw_policy *p);

enum {
    TW_EVAL_FORWARD      = 0, /* masked logits, -1e10 fill       (policy.rs:56-65)             */
    TW_EVAL_PREDICT      = 1, /* masked softmax, eps 1e-6        (policy.rs:39-49)             */
    TW_EVAL_FULL_PREDICT = 2  /* average over all twists, softmax (policy.rs:102-126)          */
};
/* Batched Policy.{forward,predict,full_predict} on the device.  Host pointers.
 * obs [n][n_obs] ids, masks [n][n_actions] (0/1), perms [n] (-1 = None; NULL = all None;
 * ignored by FULL_PREDICT).  out_actions [n][n_actions], out_values [n]. */
int tw_policy_evaluate(const tw_policy *p, int mode, uint32_t precision,
                       const int32_t *obs, uint32_t n, uint32_t n_obs,
                       const uint8_t *masks, const int32_t *perms,
                       float *out_actions, float *out_values);

/* ---------------------------------------------------------------------------------------- */
/* Collectors                                                                               */
/* ---------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t num_episodes;    /* episodes collected by THIS call (this rank's shard)           */
    uint64_t episode_offset;  /* global index of the first one: RNG is keyed by global index,  */
                              /* so results do not depend on how episodes are sharded          */
    float    gamma, lambda;   /* PPOCollector::new (ppo.rs:30-37)                              */
    uint64_t seed;            /* build extension: the reference RNG is unseedable              */
    uint32_t precision;       /* TW_PREC_*                                                     */
    uint32_t merge_order;     /* 1: reference order [E-1, 0, .., E-2] (collector.rs:40-46)     */
                              /* 0: episode-index order (shards, before the cross-GPU gather)  */
    uint32_t reserve_cus;     /* multi-GPU: compute units the persistent rollout grid leaves    */
                              /* free, so that RCCL's send/recv kernels of the previous chunk's */
                              /* gather can run beside it (0 = use every CU)                    */
} tw_ppo_params;

typedef struct {
    uint64_t num_episodes, episode_offset;
    uint32_t num_mcts_searches;   /* AZCollector::new (az.rs:32-46)                            */
    float    C;
    uint32_t max_expand_depth;
    uint64_t seed;
    uint32_t precision;
    uint32_t merge_order;
    uint32_t reserve_cus;         /* as in tw_ppo_params                                        */
} tw_az_params;

/* ---- any environment: `trait Env` (rust/src/rl/env.rs:18-68) as a table of C functions.  The collector clones the
 * prototype once per episode and never mutates it (rust/src/collector/ppo.rs:59); the environment's code runs on the host,
 * as in the reference (a Rust crate like examples/grid_world, or a Python class behind PyEnv, python_interface/pyenv.rs),
 * the policy forward of all live episodes of a time step is one batched launch on the device.  `reset` receives the collect's
 * seed and the GLOBAL episode index (build extension: the reference's reset draws from thread_rng). */
typedef struct {
    void    *prototype;
    uint32_t num_actions;                 /* Env::num_actions (<= 31)                                   */
    uint32_t n_obs, obs_size;             /* observe() returns EXACTLY n_obs ids for every state, each  */
                                          /* < obs_size (<= 65535): fixed-length observations only; an  */
                                          /* id >= obs_size fails the collect (the reference panics)    */
    void  *(*clone)(void *env);
    void   (*destroy)(void *env);
    void   (*reset)(void *env, uint64_t seed, uint64_t episode);
    void   (*step)(void *env, uint32_t action);
    void   (*observe)(void *env, int32_t *out /* n_obs */);
    void   (*masks)(void *env, uint8_t *out /* num_actions, 0/1 */);
    float  (*reward)(void *env);
    int    (*is_final)(void *env);
    int    (*success)(void *env);         /* Env::success; only tw_evaluate_env / tw_solve_env call it (may be NULL for the collectors) */
    /* The rest of the trait (rust/src/rl/env.rs:30,58-66).  Every one may be NULL = the trait's default body. */
    int      (*track_solution)(void *env);                          /* Env::track_solution (:62); NULL: false.  Asked once per   */
                                                                    /* attempt, before its first move (rust/src/rl/solve.rs:28)  */
    uint32_t (*solution)(void *env, uint32_t *out, uint32_t cap);   /* Env::solution (:65): writes at most `cap` entries, returns */
                                                                    /* how many there are; NULL: empty.  tw_solve_env returns it  */
                                                                    /* instead of the actions it played when the environment      */
                                                                    /* tracks its own (solve.rs:57-64)                            */
    void     (*set_state)(void *env, const int64_t *state, uint32_t n);   /* Env::set_state (:30).  Host-side member: the library  */
                                                                    /* never calls it (solve starts from the prototype AS IT IS)  */
    uint32_t (*twists)(void *env, int32_t *obs_perms, int32_t *act_perms, uint32_t cap);  /* Env::twists (:58-59): up to `cap`       */
                                                                    /* permutations, obs_perms[k][obs_size], act_perms[k][num_actions]; */
                                                                    /* returns how many there are; NULL: none.  Host-side member: */
                                                                    /* the HOST builds the policy with them (tw_policy_create),   */
                                                                    /* as src/twisterl/rl/algorithm.py does from env.twists()     */
} tw_env_vtable;

/* solve / evaluate (rust/src/rl/solve.rs:73-101, rust/src/rl/evaluate.rs:22-89; PyO3 functions
 * collector.solve / collector.evaluate, rust/src/python_interface/env.rs:180-207) */
typedef struct {
    uint32_t deterministic;       /* argmax (1) or weighted sample (0) of the action probs          */
    uint32_t num_searches;        /* best-of-N attempts per episode                                  */
    uint32_t num_mcts_searches;   /* 0: Policy::predict.  > 0: predict_probs_mcts per move (solve.rs:41-47), run on the MCTS kernel */
    float    C;
    uint32_t max_expand_depth;
    uint64_t seed;                /* the reference accepts and ignores `seed` (evaluate.rs:29)       */
    uint32_t precision;
} tw_solve_params;

typedef struct tw_collected tw_collected;

int tw_ppo_collect(const tw_puzzle_desc *env, const tw_policy *policy,
                   const tw_ppo_params *params, tw_collected **out);
int tw_az_collect(const tw_puzzle_desc *env, const tw_policy *policy,
                  const tw_az_params *params, tw_collected **out);
/* PPOCollector::collect for any environment (policies: any shape tw_policy_create accepts; f32).  An episode that has not
 * ended after max_records_per_episode records is an error.  params.reserve_cus is ignored. */
int tw_ppo_collect_env(const tw_env_vtable *env, const tw_policy *policy, const tw_ppo_params *params,
                       uint32_t max_records_per_episode, tw_collected **out);
/* AZCollector::collect for any environment (rust/src/collector/az.rs:51-109): the search trees live on the host (a node owns
 * a clone of the environment, as rust/src/rl/search.rs:20-26), every Policy::full_predict the searches of all episodes want at
 * a moment is one batched launch.  Same rules as tw_ppo_collect_env. */
int tw_az_collect_env(const tw_env_vtable *env, const tw_policy *policy, const tw_az_params *params,
                      uint32_t max_records_per_episode, tw_collected **out);

/* evaluate(): reset + best-of-num_searches solve for episodes [episode_offset, +num_episodes);
 * returns the success rate and the mean total reward, accumulated in episode order. */
int tw_evaluate(const tw_puzzle_desc *env, const tw_policy *policy, const tw_solve_params *params,
                uint64_t num_episodes, uint64_t episode_offset, float *success_rate, float *mean_reward);
/* solve(): best of num_searches attempts from the CURRENT state of `env` (which is not modified).
 * actions_out (capacity actions_cap, may be NULL) receives the best attempt's actions. */
int tw_solve(const tw_puzzle *env, const tw_policy *policy, const tw_solve_params *params,
             float *success, float *reward, uint8_t *actions_out, uint32_t actions_cap, uint32_t *n_actions);
/* evaluate() / solve() for any environment (rust/src/rl/evaluate.rs:22-89, rust/src/rl/solve.rs:17-101): the environment (and,
 * with num_mcts_searches > 0, the search trees) on the host, the policy outputs of all attempts of a moment in one batched
 * launch.  tw_evaluate_env resets a clone of the prototype per episode; tw_solve_env starts every attempt from a clone of the
 * prototype AS IT IS.  An attempt that has not ended after max_steps steps is an error. */
int tw_evaluate_env(const tw_env_vtable *env, const tw_policy *policy, const tw_solve_params *params, uint64_t num_episodes,
                    uint64_t episode_offset, uint32_t max_steps, float *success_rate, float *mean_reward);
int tw_solve_env(const tw_env_vtable *env, const tw_policy *policy, const tw_solve_params *params, uint32_t max_steps,
                 float *success, float *reward, uint8_t *actions_out, uint32_t actions_cap, uint32_t *n_actions);
/* The same with 32-bit entries: what an environment that tracks its own solution returns (Env::solution, Vec<usize>) need not be
 * action indices.  tw_solve_env (8-bit) fails with TW_ERR_INVALID on an entry above 255. */
int tw_solve_env32(const tw_env_vtable *env, const tw_policy *policy, const tw_solve_params *params, uint32_t max_steps,
                   float *success, float *reward, uint32_t *solution_out, uint32_t solution_cap, uint32_t *n_solution);

/* Fields of the result (device-resident, compact, in the order params.merge_order asked for) */
enum {
    TW_F_OBS       = 0,  /* uint8  [n][n_cells]    obs ids (< 256)                             */
    TW_F_LOGITS    = 1,  /* float  [n][n_actions]  PPO: masked logits; AZ: MCTS probs          */
    TW_F_PERMS     = 2,  /* int8   [n]             -1 = None                                   */
    TW_F_VALUES    = 3,  /* float  [n]             (PPO only)                                  */
    TW_F_REWARDS   = 4,  /* float  [n]             (PPO only)                                  */
    TW_F_ACTIONS   = 5,  /* uint8  [n]             (PPO only)                                  */
    TW_F_ADVS      = 6,  /* float  [n]             additional_data["advs"]  (PPO)              */
    TW_F_RETS      = 7,  /* float  [n]             additional_data["rets"]  (PPO)              */
    TW_F_REMAINING = 8,  /* float  [n]             additional_data["remaining_values"] (AZ)    */
    TW_F_EP_LEN    = 9,  /* uint32 [num_episodes]  records per episode, episode-index order    */
    TW_F_EP_START  = 10, /* uint64 [num_episodes]  first record of each episode in the output  */
    TW_F_COUNT     = 11
};

uint64_t tw_collected_num_records(const tw_collected *c);
uint64_t tw_collected_num_episodes(const tw_collected *c);
uint32_t tw_collected_num_cells(const tw_collected *c);
uint32_t tw_collected_num_actions(const tw_collected *c);
int      tw_collected_is_ppo(const tw_collected *c);
uint32_t tw_collected_obs_width(const tw_collected *c);   /* bytes per obs id in TW_F_OBS: 1, or 2 for environments with > 256 ids */
/* device pointer + byte size of a field (NULL/0 when the collector does not produce it) */
void    *tw_collected_device_ptr(const tw_collected *c, int field, size_t *bytes);
int      tw_collected_copy_to_host(const tw_collected *c, int field, void *dst, size_t bytes);

typedef struct {
    float    ms_rollout;    /* fused step+observe+reward+forward+sample kernel (HIP events)     */
    float    ms_scan;       /* episode-length scan kernels                                      */
    float    ms_finalize;   /* GAE + compaction kernel                                          */
    float    ms_total;      /* first launch -> last kernel done                                 */
    uint64_t records;
    uint64_t episodes;
    uint64_t padded_bytes;  /* workspace used by the padded trajectory buffers                  */
    uint32_t rollout_blocks, rollout_threads;
    uint64_t forward_evals; /* policy forwards the searches consumed (AZ: leaf evaluations x    */
                            /* twists) = what the reference computes                            */
    uint64_t speculative_evals; /* AZ, few deep searches: frontier nodes evaluated on otherwise */
                            /* idle MFMA columns before a search asked for them (x twists)      */
    uint64_t reused_evals;  /* AZ, few deep searches: of forward_evals, the outputs a node took  */
                            /* from its grandparent (its move took the parent's move back: same */
                            /* board, same output) instead of a column of a forward (x twists)  */
} tw_collect_stats;
int  tw_collected_stats(const tw_collected *c, tw_collect_stats *out);
/* Releases the result.  Its device memory goes to a per-process pool (freed by tw_release_cached_memory, or when the device
 * runs out of memory inside the library) and is handed to a later collect, which waits for everything the LIBRARY's stream had
 * queued before this call -- on the stream that PRODUCED the result and on the calling thread's stream (tw_set_stream is
 * thread-local; the free may come from another thread).  Work of the caller on other streams that still reads the result must
 * be finished first. */
void tw_collected_free(tw_collected *c);

/* ---- trainer hand-off (replaces the list -> numpy -> tensor path of PPO.data_to_torch / AZ.data_to_torch,
 *      reference src/twisterl/rl/ppo.py:25-61, rl/az.py:28-46).  Every output is a DEVICE pointer supplied by the
 *      caller (e.g. torch tensors) and may be NULL; rows [row_begin, row_begin+row_count) of the collected data.
 *  obs_onehot  float [row_count][obs_size]   np_obs[i, obs_i] = 1.0                        (ppo.py:37-39)
 *  log_probs   float [row_count]             Categorical(logits).log_prob(actions)          (ppo.py:57-59)  PPO only
 *  actions     int64 [row_count]             (ppo.py:47)                                                    PPO only
 *  perms       int64 [row_count]             -1 = None (ppo.py:50-52)
 *  advs        float [row_count]             advantages; normalize_advantage != 0: (a - mean) / (std + 1e-8) with the
 *                                            mean / unbiased std of ALL records of the collect (ppo.py:55-56)  PPO only */
int tw_collected_pack_trainer(const tw_collected *c, uint32_t obs_size, int normalize_advantage, uint64_t row_begin,
                              uint64_t row_count, float *obs_onehot, float *log_probs, int64_t *actions, int64_t *perms,
                              float *advs);
/* mean and unbiased standard deviation (torch.std) of the advantages of the whole collect */
int tw_collected_adv_stats(const tw_collected *c, double *mean, double *std_unbiased);

/* ---------------------------------------------------------------------------------------- */
/* Multi-GPU exchange: one process per GPU, RCCL over xGMI (resolved with dlopen at first use).*/
/* The reference has no distributed backend; its episodes are independent                      */
/* (rust/src/collector/ppo.rs:59,110-124), so ranks collect disjoint episode ranges            */
/* (tw_*_params.episode_offset, merge_order 0) and exchange only:                              */
/*   tw_comm_broadcast_policy  the policy's device image -- the multi-GPU half of              */
/*                             Algorithm.sync_rs_policy (src/twisterl/rl/algorithm.py:90-93)   */
/*   tw_gather_*               finished trajectories -> root in the reference merge order      */
/*                             [E-1, 0, .., E-2] (rust/src/collector/collector.rs:40-46)       */
/* ---------------------------------------------------------------------------------------- */
typedef struct { char bytes[128]; } tw_comm_id;     /* an ncclUniqueId: rank 0 creates it, the host hands it to the other ranks */
typedef struct tw_comm tw_comm;
int  tw_comm_get_unique_id(tw_comm_id *out);
int  tw_comm_init(int rank, int world, const tw_comm_id *id, tw_comm **out);   /* collective; binds the current device */
void tw_comm_destroy(tw_comm *c);
/* How long tw_gather_finish waits for the transfers before it gives up, aborts the communicator (ncclCommAbort) and
 * returns TW_ERR_HIP; 0 (default) = for ever.  Error behaviour of the exchange: whatever a rank can find wrong with the
 * numbers of a step (layout, capacity, the root's allocation) travels with the step's count exchange, so ALL ranks return
 * the error from the same call and nobody is left inside a collective; a failure of an RCCL / HIP call itself aborts
 * that rank's communicator (every later call on it fails) and the peers leave through this timeout. */
int  tw_comm_set_timeout_ms(tw_comm *c, uint32_t ms);
int  tw_comm_rank(const tw_comm *c);
int  tw_comm_world(const tw_comm *c);
/* every rank holds a policy created from the same shapes; afterwards all hold the root's weights */
int  tw_comm_broadcast_policy(tw_comm *c, tw_policy *p, int root);

/* Gather in `steps` pipeline steps (collective; every rank makes the same calls).  In step s rank r submits the
 * trajectories of ITS chunk of episodes -- chunk-major order: the episode ranges of (step 0: rank 0, 1, ..), (step 1:
 * rank 0, 1, ..) ... are consecutive -- collected with merge_order 0; NULL = no episodes in this step.  The transfer of a
 * step runs on the communicator's own stream, beside the collection of the next one (give that collection
 * tw_*_params.reserve_cus so that RCCL's kernels find a compute unit).  Every chunk is received at its final offset in the
 * root's result.  steps > 1: max_records bounds the records of all ranks and steps, max_episode_records those of one
 * episode.  episode_offset: index of the chunk's first episode inside the gathered range [0, total_episodes).
 * The submitted objects must stay alive until tw_gather_finish, which waits, frees the gather and (root only) returns the
 * merged result -- an ordinary tw_collected. */
typedef struct tw_gather tw_gather;
/* Placement of one step's chunks in the root's result -- pure host arithmetic (no device, no RCCL; tw_gather_submit calls
 * it with the counts it has exchanged).  counts[world][TW_GATHER_COUNTS]: records, records of the chunk's last episode,
 * episodes, first global episode, bytes per obs id, actions, status, 0.  *st: steps / max_records / max_episode_records
 * set by the caller, step 0 at the start; front, cap, pos, tail are maintained by the function (front / cap decided in step
 * 0: one step = exact size with the records of episode E-1 first, several = max_records + max_episode_records with that
 * much slack in front).  Out: n_pieces[r] in {0, 1, 2} and pieces[r][2] = records [src_lo, src_hi) of rank r's chunk ->
 * record offset dst of the root's buffers; *tail_rank = the rank whose chunk ends with episode E-1 (last step, else -1).
 * The merged result is records [front - tail, front + pos) once every step is planned. */
enum { TW_GATHER_COUNTS = 8 };
typedef struct { uint64_t src_lo, src_hi, dst; } tw_gather_piece;
typedef struct { uint32_t steps, step; uint64_t max_records, max_episode_records, pos, front, cap, tail; } tw_gather_state;
int tw_gather_plan(tw_gather_state *st, int world, const uint64_t *counts, int32_t *tail_rank, uint32_t *n_pieces,
                   tw_gather_piece *pieces);
int tw_gather_begin(tw_comm *c, int root, uint32_t steps, uint64_t max_records, uint32_t max_episode_records,
                    uint64_t total_episodes, int is_ppo, uint32_t n_cells, tw_gather **out);
int tw_gather_submit(tw_gather *g, const tw_collected *local, uint64_t episode_offset);
int tw_gather_finish(tw_gather *g, tw_collected **merged);

#ifdef __cplusplus
}
#endif
#endif /* TWISTERL_HIP_H */
