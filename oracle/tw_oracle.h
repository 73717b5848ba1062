/*
 * tw_oracle.h -- CPU ORACLE for the twisteRL episode-collection hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and the cpu_baseline leg of bench.py may load it.  The product
 * (twisterl_amd/) never links, imports or calls anything in oracle/.
 *
 * It is a plain-C restatement of the reference algorithm (Rust, /root/reference/rust/src),
 * every function citing the reference file:line it follows.  The reference cannot be
 * compiled in this image (no rustc/cargo), so the oracle is pinned by the reference's own
 * unit-test known answers and the 123-move replay vector (tests/test_oracle_golden.py).
 *
 * Where the reference draws from rand::thread_rng() (unseedable), the oracle draws from a
 * counter-based Philox4x32-10 stream keyed by (seed, episode, index, stream); the HIP
 * product implements the same published generator independently (DESIGN.md "RNG spec").
 */
#ifndef TW_ORACLE_H
#define TW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- RNG -------------------------------------------------------------------------- */
enum {
    TWO_STREAM_SCRAMBLE = 0, /* Puzzle::reset scramble actions        (puzzle.rs:124-131)  */
    TWO_STREAM_GUMBEL   = 1, /* sample_from_logits uniforms           (policy.rs:169-172)  */
    TWO_STREAM_PERM     = 2, /* Policy::get_perm_id                   (policy.rs:67-77)    */
    TWO_STREAM_AZ_ACT   = 3, /* AZCollector root action sample        (az.rs:72)           */
    TWO_STREAM_MCTS     = 4, /* MCTSTree::next_sample                 (search.rs:94-100)   */
    TWO_STREAM_SOLVE    = 5  /* single_solve action sample            (solve.rs:50-54)     */
};
void two_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* deterministic natural log used by the "exact" sampler (same spec as the HIP kernel) */
float two_logf_det(float x);
/* deterministic exp used by the "exact" masked softmax (same spec as the HIP kernels) */
float two_expf_det(float x);
/* 1: masked softmax (predict / full_predict / MCTS priors) uses two_expf_det; 0: libm expf */
void  two_set_det_exp(int on);

/* ---- Puzzle env (envs/puzzle.rs:20-185) -------------------------------------------- */
#define TWO_MAX_CELLS 64
typedef struct {
    int64_t state[TWO_MAX_CELLS];
    int64_t zx, zy;
    int64_t depth;
    int64_t width, height, difficulty, depth_slope, max_depth;
} two_puzzle;

void  two_puzzle_new(two_puzzle *p, int64_t width, int64_t height, int64_t difficulty,
                     int64_t depth_slope, int64_t max_depth);
int   two_puzzle_solved(const two_puzzle *p);
void  two_puzzle_set_state(two_puzzle *p, const int64_t *state, size_t n);
void  two_puzzle_reset(two_puzzle *p, uint64_t seed, uint64_t episode);
void  two_puzzle_step(two_puzzle *p, int64_t action);
void  two_puzzle_masks(const two_puzzle *p, uint8_t out[4]);
int   two_puzzle_is_final(const two_puzzle *p);
float two_puzzle_reward(const two_puzzle *p);
void  two_puzzle_observe(const two_puzzle *p, int64_t *out);

/* ---- NN (nn/layers.rs, nn/modules.rs, nn/policy.rs) -------------------------------- */
typedef struct {
    int in, out;
    const float *w;  /* reference layout: column-major DMatrix(out,in): w[i*out + o] (layers.rs:26) */
    const float *b;
    int relu;
} two_linear;

typedef struct {
    int n_vectors, vec_len;     /* vectors[n_vectors][vec_len] (layers.rs:50-54)              */
    const float *vectors;
    const float *bias; int bias_len;
    int relu;
    int obs_shape[2]; int obs_ndim; int conv_dim;
} two_embbag;

#define TWO_MAX_LAYERS 8
typedef struct {
    two_embbag emb;
    two_linear common[TWO_MAX_LAYERS]; int n_common;
    two_linear action[TWO_MAX_LAYERS]; int n_action;
    two_linear value[TWO_MAX_LAYERS];  int n_value;
    int n_perms, obs_size, n_actions;
    const int32_t *obs_perms;   /* [n_perms][obs_size] */
    const int32_t *act_perms;   /* [n_perms][n_actions] */
} two_policy;

/* arithmetic mode of Linear::forward:
 *  TWO_ARITH_REF   -- reference order: k-ordered, un-fused multiply then add, bias last
 *                     (nalgebra gemv, layers.rs:32)
 *  TWO_ARITH_CHAIN -- k-ordered fused multiply-add chain from 0, bias last; this is the
 *                     order an f32 MFMA accumulates in, so the HIP "exact" mode is bit-equal
 *  TWO_ARITH_F16   -- table rows, weights and layer inputs rounded to binary16 (RNE), products exact,
 *                     accumulation in double, biases and outputs f32: the spec of the HIP library's
 *                     TW_PREC_F16 mode (v_mfma_f32_32x32x16_f16 accumulates in f32 in a hardware-defined
 *                     order, so that mode matches this one to f32 rounding, not bit for bit) */
enum { TWO_ARITH_REF = 0, TWO_ARITH_CHAIN = 1, TWO_ARITH_F16 = 2 };
float two_round_f16(float x);

/* Policy::_raw_predict (policy.rs:79-100); perm < 0 means None */
void two_policy_raw_predict(const two_policy *pol, const int64_t *obs, int n_obs, int perm,
                            int arith, float *logits_out, float *value_out);
/* Policy::forward_with_perm masks step (policy.rs:56-65) given an explicit perm */
void two_policy_forward(const two_policy *pol, const int64_t *obs, int n_obs,
                        const uint8_t *masks, int perm, int arith,
                        float *masked_logits_out, float *value_out);
/* Policy::predict_with_perm (policy.rs:39-49) given an explicit perm */
void two_policy_predict(const two_policy *pol, const int64_t *obs, int n_obs,
                        const uint8_t *masks, int perm, int arith,
                        float *probs_out, float *value_out);
/* Policy::full_predict (policy.rs:102-126) */
void two_policy_full_predict(const two_policy *pol, const int64_t *obs, int n_obs,
                             const uint8_t *masks, int arith, float *probs_out, float *value_out);
/* argmax (policy.rs:130-151) */
int  two_argmax(const float *v, int n);
/* sample_from_logits (policy.rs:169-172) with injected uniforms; det_log selects two_logf_det */
int  two_sample_from_logits(const float *logits, int n, const float *u, int det_log);
/* nn::policy::sample (policy.rs:153-167): weighted index draw with an injected uniform */
int  two_sample_weighted(const float *probs, int n, float u);

/* ---- GAE (collector/ppo.rs:82-92) --------------------------------------------------- */
void two_gae(const float *rews, const float *vals, int n, float gamma, float lambda,
             float *advs, float *rets);

/* ---- collectors --------------------------------------------------------------------- */
typedef struct {
    uint64_t n;          /* number of records                                             */
    int      n_cells;    /* obs ids per record                                            */
    int      n_actions;
    int64_t *obs;        /* [n][n_cells]                                                  */
    float   *logits;     /* [n][n_actions]  (PPO: masked logits; AZ: MCTS probs)           */
    int32_t *perms;      /* [n]  -1 = None                                                */
    float   *values;     /* [n]  (empty for AZ)                                           */
    float   *rewards;    /* [n]  (empty for AZ)                                           */
    int64_t *actions;    /* [n]  (empty for AZ)                                           */
    float   *advs;       /* [n]  additional_data["advs"]  (PPO)                           */
    float   *rets;       /* [n]  additional_data["rets"]  (PPO)                           */
    float   *remaining;  /* [n]  additional_data["remaining_values"] (AZ)                 */
    uint32_t *ep_len;    /* [num_episodes] records per episode, in episode-index order    */
    uint64_t n_episodes;
    int      has_ppo;    /* values/rewards/actions/advs/rets filled                       */
} two_collected;

typedef struct {
    uint64_t num_episodes;     /* episodes collected by this call                          */
    uint64_t episode_offset;   /* global index of this call's first episode (RNG key)      */
    float    gamma, lambda;
    uint64_t seed;
    int      arith;            /* TWO_ARITH_*                                              */
    int      det_log;          /* 1: two_logf_det in the Gumbel sampler, 0: libm logf      */
    int      num_threads;      /* rayon-pool stand-in (ppo.rs:110-124)                     */
    int      merge_order;      /* 1: reference merge order [E-1,0,..,E-2] (collector.rs:40-46); 0: index order */
} two_ppo_params;

/* PPOCollector::collect (ppo.rs:108-126) over Puzzle */
int  two_ppo_collect(const two_puzzle *env, const two_policy *pol, const two_ppo_params *prm,
                     two_collected *out);

typedef struct {
    uint64_t num_episodes, episode_offset;
    uint32_t num_mcts_searches;
    float    C;
    uint32_t max_expand_depth;
    uint64_t seed;
    int      arith;
    int      num_threads;
    int      merge_order;
    int      det_math;         /* 1: two_expf_det in the softmax (bit-equal to the HIP path)   */
} two_az_params;

/* AZCollector::collect (az.rs:112-130) over Puzzle */
int  two_az_collect(const two_puzzle *env, const two_policy *pol, const two_az_params *prm,
                    two_collected *out);
/* predict_probs_mcts (search.rs:104-189) on one state; t = record index of the root (RNG key) */
void two_mcts_probs(const two_puzzle *root, const two_policy *pol, uint32_t num_mcts_searches,
                    float C, uint32_t max_expand_depth, int arith, uint64_t seed,
                    uint64_t episode, uint32_t t, float *probs_out);

void two_collected_free(two_collected *c);

/* ---- solve / evaluate (rl/solve.rs:17-101, rl/evaluate.rs:22-89) --------------------- */
typedef struct {
    int      deterministic;       /* argmax vs weighted sample of the probs (solve.rs:50-54)     */
    uint32_t num_searches;        /* best-of-N attempts (solve.rs:84-98)                          */
    uint32_t num_mcts_searches;   /* 0: policy.predict; >0: predict_probs_mcts (solve.rs:37-48)   */
    float    C;
    uint32_t max_expand_depth;
    uint64_t seed;
    int      arith;
    int      det_math;
} two_solve_params;
/* solve() from the env's CURRENT state; `episode` keys the RNG (attempt a uses episode*N + a).
 * actions_out must hold at least depth+1 entries; returns the number of actions of the best attempt */
int two_solve(const two_puzzle *env, const two_policy *pol, const two_solve_params *prm, uint64_t episode,
              float *success_out, float *reward_out, int64_t *actions_out);
/* evaluate(): reset (seed, episode e) + solve per episode, means in episode order */
void two_evaluate(const two_puzzle *env, const two_policy *pol, const two_solve_params *prm, uint64_t num_episodes,
                  float *success_rate_out, float *mean_reward_out);

/* replay: apply `actions` from `start` (set_state semantics for depth unless depth0>=0),
 * recording for each of the n+1 visited states obs ids, masks, reward, is_final, board. */
void two_replay(const two_puzzle *start, const int64_t *actions, size_t n,
                int64_t *obs_out, uint8_t *masks_out, float *reward_out, uint8_t *final_out,
                int64_t *board_out);

#ifdef __cplusplus
}
#endif
#endif
