/*
 * tw_oracle.c -- CPU ORACLE (test infrastructure only; see tw_oracle.h).
 *
 * Plain-C restatement of the twisteRL collection hot path.  Every function cites the
 * reference source (paths relative to /root/reference/) it follows.  Build with
 * -ffp-contract=off so that the only fused multiply-adds are the explicit fmaf() calls.
 */
#include "tw_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ===================================================================================== */
/* RNG: Philox4x32-10 (Salmon et al., SC'11; Random123 reference constants).            */
/* Stands in for rand::thread_rng() which the reference cannot seed                     */
/* (puzzle.rs:124, policy.rs:72,154,170).                                               */
/* ===================================================================================== */
void two_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void rng_draw(uint64_t seed, uint64_t episode, uint32_t index, uint32_t stream,
                     uint32_t out[4])
{
    uint32_t ctr[4] = { (uint32_t)episode, (uint32_t)(episode >> 32), index, stream };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    two_philox4x32_10(ctr, key, out);
}

/* integer in [0,n): high half of the 32x32 product (exactly uniform for n a power of two).
 * Stands in for Uniform::new(0,n).sample (puzzle.rs:125-129, policy.rs:73). */
static inline uint32_t u32_below(uint32_t word, uint32_t n)
{
    return (uint32_t)(((uint64_t)word * (uint64_t)n) >> 32);
}

/* f32 in [0,1) with 24 random bits: the granularity of rand 0.8.5 `rng.gen::<f32>()`
 * (policy.rs:171). */
static inline float u32_to_unit(uint32_t word)
{
    return (float)(word >> 8) * (1.0f / 16777216.0f);
}

/* Deterministic natural log (Cephes-style degree-8 polynomial, explicit op order).
 * The HIP kernel implements the same operation sequence, so Gumbel noise is bit-equal
 * on both sides.  Domain: positive normal floats, +0 (-> -inf) and +inf (-> +inf). */
float two_logf_det(float x)
{
    if (x == 0.0f) return -INFINITY;
    if (isinf(x)) return INFINITY;
    uint32_t ix; memcpy(&ix, &x, 4);
    int e = (int)(ix >> 23) - 127;
    uint32_t im = (ix & 0x007fffffu) | 0x3f800000u;
    float m; memcpy(&m, &im, 4);
    if (m > 1.41421354f) { m = m * 0.5f; e = e + 1; }
    float f = m - 1.0f;
    float z = f * f;
    float p = 7.0376836292e-2f;
    p = fmaf(p, f, -1.1514610310e-1f);
    p = fmaf(p, f,  1.1676998740e-1f);
    p = fmaf(p, f, -1.2420140846e-1f);
    p = fmaf(p, f,  1.4249322787e-1f);
    p = fmaf(p, f, -1.6668057665e-1f);
    p = fmaf(p, f,  2.0000714765e-1f);
    p = fmaf(p, f, -2.4999993993e-1f);
    p = fmaf(p, f,  3.3333331174e-1f);
    float y = (p * f) * z;
    float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    float r = f + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

/* Deterministic expf (Cephes-style, explicit op order); the HIP kernels implement the same
 * operation sequence, so MCTS priors are bit-equal on CPU and GPU.  Within 2 ulp of libm. */
float two_expf_det(float x)
{
    if (x != x) return x;
    if (x > 88.72283905206835f) return INFINITY;
    if (x < -103.972077083991796f) return 0.0f;
    float fx = fmaf(x, 1.44269504088896341f, 0.5f);
    float fn = floorf(fx);
    float r = fmaf(fn, -0.693359375f, x);
    r = fmaf(fn, 2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p, z, r);
    y = y + 1.0f;
    int n = (int)fn;
    int n1 = n / 2, n2 = n - n1;
    uint32_t b1 = (uint32_t)(n1 + 127) << 23, b2 = (uint32_t)(n2 + 127) << 23;
    float s1, s2; memcpy(&s1, &b1, 4); memcpy(&s2, &b2, 4);
    y = y * s1;
    return y * s2;
}

/* selects libm expf (reference) or two_expf_det inside masked_softmax; set per call by the
 * det_math flag of the AZ parameters / two_set_det_exp */
static _Thread_local int g_det_exp = 0;
void two_set_det_exp(int on) { g_det_exp = on; }

/* ===================================================================================== */
/* Puzzle (envs/puzzle.rs)                                                               */
/* ===================================================================================== */

/* Puzzle::new (puzzle.rs:34-42): identity board, zero at (0,0), depth 1 */
void two_puzzle_new(two_puzzle *p, int64_t width, int64_t height, int64_t difficulty,
                    int64_t depth_slope, int64_t max_depth)
{
    memset(p, 0, sizeof(*p));
    for (int64_t i = 0; i < width * height; ++i) p->state[i] = i;
    p->zx = 0; p->zy = 0; p->depth = 1;
    p->width = width; p->height = height; p->difficulty = difficulty;
    p->depth_slope = depth_slope; p->max_depth = max_depth;
}

/* Puzzle::solved (puzzle.rs:44-50) */
int two_puzzle_solved(const two_puzzle *p)
{
    for (int64_t i = 0; i < p->width * p->height; ++i)
        if (p->state[i] != i) return 0;
    return 1;
}

/* Env::set_state (puzzle.rs:107-117): depth <- max_depth, zero located by first 0 */
void two_puzzle_set_state(two_puzzle *p, const int64_t *state, size_t n)
{
    for (size_t i = 0; i < n; ++i) p->state[i] = state[i];
    p->depth = p->max_depth;
    for (size_t i = 0; i < n; ++i) {
        if (state[i] == 0) {
            p->zx = (int64_t)i % p->width;
            p->zy = (int64_t)i / p->width;
            break;
        }
    }
}

static inline int64_t get_pos(const two_puzzle *p, int64_t x, int64_t y) { return p->state[y * p->width + x]; }
static inline void set_pos(two_puzzle *p, int64_t x, int64_t y, int64_t v) { p->state[y * p->width + x] = v; }

/* Env::step (puzzle.rs:135-160): 0=left 1=up 2=right 3=down of the blank; illegal = no-op;
 * depth always saturating_sub(1) */
void two_puzzle_step(two_puzzle *p, int64_t action)
{
    int64_t zx = p->zx, zy = p->zy;
    if (action == 0 && zx > 0) {
        int64_t v = get_pos(p, zx - 1, zy);
        set_pos(p, zx, zy, v); set_pos(p, zx - 1, zy, 0); p->zx = zx - 1;
    } else if (action == 1 && zy > 0) {
        int64_t v = get_pos(p, zx, zy - 1);
        set_pos(p, zx, zy, v); set_pos(p, zx, zy - 1, 0); p->zy = zy - 1;
    } else if (action == 2 && zx < p->width - 1) {
        int64_t v = get_pos(p, zx + 1, zy);
        set_pos(p, zx, zy, v); set_pos(p, zx + 1, zy, 0); p->zx = zx + 1;
    } else if (action == 3 && zy < p->height - 1) {
        int64_t v = get_pos(p, zx, zy + 1);
        set_pos(p, zx, zy, v); set_pos(p, zx, zy + 1, 0); p->zy = zy + 1;
    }
    p->depth = p->depth > 0 ? p->depth - 1 : 0;
}

/* Env::reset (puzzle.rs:119-133): identity, `difficulty` uniform actions (wall hits are
 * no-ops), then depth = depth_slope*difficulty */
void two_puzzle_reset(two_puzzle *p, uint64_t seed, uint64_t episode)
{
    for (int64_t i = 0; i < p->width * p->height; ++i) p->state[i] = i;
    p->zx = 0; p->zy = 0;
    for (int64_t d = 0; d < p->difficulty; ++d) {
        uint32_t w[4];
        rng_draw(seed, episode, (uint32_t)d, TWO_STREAM_SCRAMBLE, w);
        two_puzzle_step(p, (int64_t)u32_below(w[0], 4));
    }
    p->depth = p->depth_slope * p->difficulty;
}

/* Env::masks (puzzle.rs:162-165) */
void two_puzzle_masks(const two_puzzle *p, uint8_t out[4])
{
    out[0] = p->zx > 0; out[1] = p->zy > 0;
    out[2] = p->zx < p->width - 1; out[3] = p->zy < p->height - 1;
}

/* Env::is_final (puzzle.rs:167-169) */
int two_puzzle_is_final(const two_puzzle *p) { return p->depth == 0 || two_puzzle_solved(p); }

/* Env::reward (puzzle.rs:171-177) */
float two_puzzle_reward(const two_puzzle *p)
{
    if (two_puzzle_solved(p)) return 1.0f;
    if (p->depth == 0) return -0.5f;
    return -0.5f / (float)p->max_depth;
}

/* Env::observe (puzzle.rs:183-185): obs[i] = i*(h*w) + state[i] */
void two_puzzle_observe(const two_puzzle *p, int64_t *out)
{
    int64_t n = p->width * p->height;
    for (int64_t i = 0; i < n; ++i) out[i] = i * n + p->state[i];
}

/* ===================================================================================== */
/* NN (nn/layers.rs, nn/modules.rs, nn/policy.rs)                                        */
/* ===================================================================================== */
static inline float relu_f(float x) { return x > 0.0f ? x : 0.0f; } /* layers.rs:89-91 */

/* Linear::forward (layers.rs:31-37): out = W*x + b (+ReLU).  nalgebra's gemv walks the
 * columns of the column-major W: out = col0*x0; out += col_k*x_k; bias added afterwards. */
/* f32 -> IEEE binary16 (round to nearest even) -> f32: the rounding v_cvt_pk_f16_f32 and the host-side
 * (_Float16) conversion of the HIP library's f16 weight image apply. */
float two_round_f16(float x)
{
    uint32_t u; memcpy(&u, &x, 4);
    const uint32_t sign = u & 0x80000000u;
    uint32_t a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return x;                                   /* inf / nan */
    if (a >= 0x477ff000u) { u = sign | 0x7f800000u; memcpy(&x, &u, 4); return x; }   /* >= 65520 -> inf */
    if (a < 0x38800000u) {                                            /* below 2^-14: f16 subnormal, quantum 2^-24 */
        const float q = rintf(fabsf(x) * 16777216.0f) / 16777216.0f;
        return sign ? -q : q;
    }
    const uint32_t rem = a & 0x1fffu;
    a &= ~0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (a & 0x2000u))) a += 0x2000u;
    a |= sign; memcpy(&x, &a, 4);
    return x;
}

static void linear_forward(const two_linear *l, const float *x, float *out, int arith)
{
    const int n_out = l->out, n_in = l->in;
    if (arith == TWO_ARITH_F16) {
        /* f16 inputs and weights, exact products, wide accumulation, f32 bias (tw_engine16.hpp) */
        for (int o = 0; o < n_out; ++o) {
            double acc = 0.0;
            for (int k = 0; k < n_in; ++k)
                acc += (double)two_round_f16(l->w[(size_t)k * n_out + o]) * (double)two_round_f16(x[k]);
            float v = (float)acc + l->b[o];
            out[o] = l->relu ? relu_f(v) : v;
        }
        return;
    }
    if (arith == TWO_ARITH_CHAIN) {
        for (int o = 0; o < n_out; ++o) out[o] = 0.0f;
        for (int k = 0; k < n_in; ++k) {
            const float xk = x[k];
            const float *col = l->w + (size_t)k * n_out;
            for (int o = 0; o < n_out; ++o) out[o] = fmaf(col[o], xk, out[o]);
        }
    } else {
        if (n_in > 0) {
            const float x0 = x[0];
            for (int o = 0; o < n_out; ++o) out[o] = l->w[o] * x0;
        } else {
            for (int o = 0; o < n_out; ++o) out[o] = 0.0f;
        }
        for (int k = 1; k < n_in; ++k) {
            const float xk = x[k];
            const float *col = l->w + (size_t)k * n_out;
            for (int o = 0; o < n_out; ++o) {
                float prod = col[o] * xk;
                out[o] = out[o] + prod;
            }
        }
    }
    for (int o = 0; o < n_out; ++o) {
        float v = out[o] + l->b[o];
        out[o] = l->relu ? relu_f(v) : v;
    }
}

/* Sequential::forward (modules.rs:28-34).  Returns output length; result in `buf_a`. */
#define TWO_MAX_WIDTH 4096
static int sequential_forward(const two_linear *layers, int n, const float *in, int in_len,
                              float *out, int arith)
{
    float tmp_a[TWO_MAX_WIDTH], tmp_b[TWO_MAX_WIDTH];
    const float *cur = in; int cur_len = in_len;
    float *dst = tmp_a;
    for (int i = 0; i < n; ++i) {
        linear_forward(&layers[i], cur, dst, arith);
        cur = dst; cur_len = layers[i].out;
        dst = (dst == tmp_a) ? tmp_b : tmp_a;
    }
    memcpy(out, cur, sizeof(float) * (size_t)cur_len);
    return cur_len;
}

/* EmbeddingBag::forward (layers.rs:56-86) */
static int embbag_forward(const two_embbag *e, const int64_t *obs, int n_obs, float *out, int arith)
{
    if (arith == TWO_ARITH_F16 && e->obs_ndim == 1) {
        for (int k = 0; k < e->vec_len; ++k) {
            double acc = 0.0;
            for (int i = 0; i < n_obs; ++i) acc += (double)two_round_f16(e->vectors[(size_t)obs[i] * e->vec_len + k]);
            float v = (float)acc + e->bias[k];
            out[k] = e->relu ? relu_f(v) : v;
        }
        return e->bias_len;
    }
    for (int k = 0; k < e->bias_len; ++k) out[k] = e->bias[k];
    if (e->obs_ndim == 1) {
        for (int i = 0; i < n_obs; ++i) {
            const float *v = e->vectors + (size_t)obs[i] * e->vec_len;
            for (int k = 0; k < e->vec_len; ++k) out[k] = out[k] + v[k];
        }
    } else if (e->obs_ndim == 2) {
        const int v_size = e->vec_len;
        for (int i = 0; i < n_obs; ++i) {
            int64_t row = obs[i] / e->obs_shape[1];
            int64_t col = obs[i] % e->obs_shape[1];
            if (e->conv_dim == 1) { int64_t t = row; row = col; col = t; }
            const float *v = e->vectors + (size_t)row * v_size;
            float *o = out + (size_t)col * v_size;
            for (int k = 0; k < v_size; ++k) o[k] = o[k] + v[k];
        }
    }
    if (e->relu) for (int k = 0; k < e->bias_len; ++k) out[k] = relu_f(out[k]);
    return e->bias_len;
}

/* Policy::_raw_predict (policy.rs:79-100) */
void two_policy_raw_predict(const two_policy *pol, const int64_t *obs, int n_obs, int perm,
                            int arith, float *logits_out, float *value_out)
{
    int64_t pobs[TWO_MAX_CELLS * 4];
    const int64_t *use = obs;
    if (perm >= 0) {                                     /* policy.rs:81-83 */
        for (int i = 0; i < n_obs; ++i) pobs[i] = pol->obs_perms[(size_t)perm * pol->obs_size + obs[i]];
        use = pobs;
    }
    float h0[TWO_MAX_WIDTH], h1[TWO_MAX_WIDTH], hv[TWO_MAX_WIDTH], ha[TWO_MAX_WIDTH];
    int n0 = embbag_forward(&pol->emb, use, n_obs, h0, arith);
    int n1 = sequential_forward(pol->common, pol->n_common, h0, n0, h1, arith);   /* :86 */
    int nv = sequential_forward(pol->value, pol->n_value, h1, n1, hv, arith);     /* :89 */
    float vsum = 0.0f;
    for (int i = 0; i < nv; ++i) vsum = vsum + hv[i];   /* .sum() */
    int na = sequential_forward(pol->action, pol->n_action, h1, n1, ha, arith);   /* :92 */
    if (perm >= 0) {                                     /* policy.rs:95-97 */
        for (int i = 0; i < pol->n_actions; ++i)
            logits_out[i] = ha[pol->act_perms[(size_t)perm * pol->n_actions + i]];
    } else {
        for (int i = 0; i < na; ++i) logits_out[i] = ha[i];
    }
    *value_out = vsum;
}

/* Policy::forward_with_perm (policy.rs:56-65) with the perm index supplied by the caller */
void two_policy_forward(const two_policy *pol, const int64_t *obs, int n_obs,
                        const uint8_t *masks, int perm, int arith,
                        float *masked_logits_out, float *value_out)
{
    float logits[256];
    two_policy_raw_predict(pol, obs, n_obs, perm, arith, logits, value_out);
    for (int i = 0; i < pol->n_actions; ++i)
        masked_logits_out[i] = masks[i] ? logits[i] : -1e10f;
}

static void masked_softmax(const float *logits, const uint8_t *masks, int n, float *probs)
{
    /* policy.rs:43-47 / 118-124: no max-subtraction, eps 1e-6 */
    float sum = 0.0f;
    for (int i = 0; i < n; ++i) { probs[i] = masks[i] ? (g_det_exp ? two_expf_det(logits[i]) : expf(logits[i])) : 0.0f; }
    for (int i = 0; i < n; ++i) sum = sum + probs[i];
    for (int i = 0; i < n; ++i) probs[i] = probs[i] / (sum + 0.000001f);
}

/* Policy::predict_with_perm (policy.rs:39-49) */
void two_policy_predict(const two_policy *pol, const int64_t *obs, int n_obs,
                        const uint8_t *masks, int perm, int arith,
                        float *probs_out, float *value_out)
{
    float ml[256];
    two_policy_forward(pol, obs, n_obs, masks, perm, arith, ml, value_out);
    masked_softmax(ml, masks, pol->n_actions, probs_out);
}

/* Policy::full_predict (policy.rs:102-126) */
void two_policy_full_predict(const two_policy *pol, const int64_t *obs, int n_obs,
                             const uint8_t *masks, int arith, float *probs_out, float *value_out)
{
    if (pol->n_perms == 0) {
        /* :103 -> predict; get_perm_id returns None without drawing */
        two_policy_predict(pol, obs, n_obs, masks, -1, arith, probs_out, value_out);
        return;
    }
    float logits[256]; float value = 0.0f;
    const float np = (float)pol->n_perms;
    for (int i = 0; i < pol->n_actions; ++i) logits[i] = 0.0f;
    for (int pi = 0; pi < pol->n_perms; ++pi) {
        float l_pi[256], v_pi;
        two_policy_raw_predict(pol, obs, n_obs, pi, arith, l_pi, &v_pi);
        value = value + v_pi / np;
        for (int i = 0; i < pol->n_actions; ++i) logits[i] = logits[i] + l_pi[i] / np;
    }
    masked_softmax(logits, masks, pol->n_actions, probs_out);
    *value_out = value;
}

/* argmax (policy.rs:130-151): strict '>' => first max wins, NaN never wins, empty => 0 */
int two_argmax(const float *v, int n)
{
    if (n <= 0) return 0;
    int best = 0; float bv = v[0];
    for (int i = 1; i < n; ++i) if (v[i] > bv) { bv = v[i]; best = i; }
    return best;
}

/* sample_from_logits (policy.rs:169-172): argmax_i( l_i - ln(|ln(u_i)|) ) */
int two_sample_from_logits(const float *logits, int n, const float *u, int det_log)
{
    float g[256];
    for (int i = 0; i < n; ++i) {
        float a = det_log ? two_logf_det(u[i]) : logf(u[i]);
        float b = fabsf(a);
        float c = det_log ? two_logf_det(b) : logf(b);
        g[i] = logits[i] - c;
    }
    return two_argmax(g, n);
}

/* nn::policy::sample (policy.rs:153-167) via rand 0.8.5 WeightedIndex semantics:
 * cumulative weights of the first n-1 entries, chosen = u*total, index = number of
 * cumulative weights <= chosen; on invalid weights the reference prints and returns 0. */
int two_sample_weighted(const float *probs, int n, float u)
{
    if (n <= 0) return 0;
    float total = 0.0f; float cum[256];
    for (int i = 0; i < n; ++i) {
        if (!(probs[i] >= 0.0f)) return 0;          /* InvalidWeight -> Err -> 0 */
        total = total + probs[i];
        if (i < n - 1) cum[i] = total;
    }
    if (!(total > 0.0f)) return 0;                   /* AllWeightsZero -> Err -> 0 */
    float chosen = u * total;
    int idx = 0;
    while (idx < n - 1 && cum[idx] <= chosen) ++idx;
    return idx;
}

/* GAE (ppo.rs:82-92) */
void two_gae(const float *rews, const float *vals, int n, float gamma, float lambda,
             float *advs, float *rets)
{
    advs[n - 1] = rews[n - 1] - vals[n - 1];
    rets[n - 1] = rews[n - 1];
    for (int t = n - 2; t >= 0; --t) {
        float inner = lambda * advs[t + 1];
        inner = vals[t + 1] + inner;
        inner = gamma * inner;
        rets[t] = rews[t] + inner;
        advs[t] = rets[t] - vals[t];
    }
}

/* ===================================================================================== */
/* per-episode storage + merge (collector/collector.rs:22-89)                            */
/* ===================================================================================== */
typedef struct {
    uint32_t n, cap;
    int64_t *obs; float *logits; int32_t *perms; float *values; float *rewards;
    int64_t *actions; float *advs; float *rets; float *remaining;
} episode_buf;

static void ep_reserve(episode_buf *e, uint32_t need, int n_cells, int n_actions)
{
    if (need <= e->cap) return;
    uint32_t cap = e->cap ? e->cap * 2 : 16;      /* Vec growth, like the per-episode pushes */
    while (cap < need) cap *= 2;
    e->obs     = (int64_t *)realloc(e->obs, sizeof(int64_t) * (size_t)cap * n_cells);
    e->logits  = (float *)realloc(e->logits, sizeof(float) * (size_t)cap * n_actions);
    e->perms   = (int32_t *)realloc(e->perms, sizeof(int32_t) * cap);
    e->values  = (float *)realloc(e->values, sizeof(float) * cap);
    e->rewards = (float *)realloc(e->rewards, sizeof(float) * cap);
    e->actions = (int64_t *)realloc(e->actions, sizeof(int64_t) * cap);
    e->cap = cap;
}

static void ep_free(episode_buf *e)
{
    free(e->obs); free(e->logits); free(e->perms); free(e->values); free(e->rewards);
    free(e->actions); free(e->advs); free(e->rets); free(e->remaining);
    memset(e, 0, sizeof(*e));
}

/* merge (collector.rs:40-46): pop the LAST chunk, append the others in index order */
static int merge_episodes(episode_buf *eps, uint64_t E, int n_cells, int n_actions,
                          int has_ppo, int merge_order, two_collected *out)
{
    if (E == 0) return -1;   /* "No data in collected data chunks to merge" (collector.rs:41) */
    uint64_t total = 0;
    for (uint64_t e = 0; e < E; ++e) total += eps[e].n;
    memset(out, 0, sizeof(*out));
    out->n = total; out->n_cells = n_cells; out->n_actions = n_actions;
    out->n_episodes = E; out->has_ppo = has_ppo;
    out->obs    = (int64_t *)malloc(sizeof(int64_t) * (size_t)total * n_cells + 8);
    out->logits = (float *)malloc(sizeof(float) * (size_t)total * n_actions + 8);
    out->perms  = (int32_t *)malloc(sizeof(int32_t) * (size_t)total + 8);
    out->ep_len = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)E);
    if (has_ppo) {
        out->values  = (float *)malloc(sizeof(float) * (size_t)total + 8);
        out->rewards = (float *)malloc(sizeof(float) * (size_t)total + 8);
        out->actions = (int64_t *)malloc(sizeof(int64_t) * (size_t)total + 8);
        out->advs    = (float *)malloc(sizeof(float) * (size_t)total + 8);
        out->rets    = (float *)malloc(sizeof(float) * (size_t)total + 8);
    } else {
        out->remaining = (float *)malloc(sizeof(float) * (size_t)total + 8);
    }
    uint64_t pos = 0;
    for (uint64_t i = 0; i < E; ++i) {
        uint64_t e;
        if (merge_order) e = (i == 0) ? E - 1 : i - 1;
        else e = i;
        const episode_buf *b = &eps[e];
        memcpy(out->obs + pos * n_cells, b->obs, sizeof(int64_t) * (size_t)b->n * n_cells);
        memcpy(out->logits + pos * n_actions, b->logits, sizeof(float) * (size_t)b->n * n_actions);
        memcpy(out->perms + pos, b->perms, sizeof(int32_t) * b->n);
        if (has_ppo) {
            memcpy(out->values + pos, b->values, sizeof(float) * b->n);
            memcpy(out->rewards + pos, b->rewards, sizeof(float) * b->n);
            memcpy(out->actions + pos, b->actions, sizeof(int64_t) * b->n);
            memcpy(out->advs + pos, b->advs, sizeof(float) * b->n);
            memcpy(out->rets + pos, b->rets, sizeof(float) * b->n);
        } else {
            memcpy(out->remaining + pos, b->remaining, sizeof(float) * b->n);
        }
        pos += b->n;
    }
    for (uint64_t e = 0; e < E; ++e) out->ep_len[e] = eps[e].n;
    return 0;
}

void two_collected_free(two_collected *c)
{
    free(c->obs); free(c->logits); free(c->perms); free(c->values); free(c->rewards);
    free(c->actions); free(c->advs); free(c->rets); free(c->remaining); free(c->ep_len);
    memset(c, 0, sizeof(*c));
}

/* ===================================================================================== */
/* PPOCollector (collector/ppo.rs)                                                       */
/* ===================================================================================== */
typedef struct {
    const two_puzzle *env; const two_policy *pol; const two_ppo_params *prm;
    episode_buf *eps; atomic_ullong next;
} ppo_job;

/* PPOCollector::single_collect (ppo.rs:54-105) */
static void ppo_single_collect(const two_puzzle *env0, const two_policy *pol,
                               const two_ppo_params *prm, uint64_t episode, episode_buf *eb)
{
    two_puzzle env = *env0;                               /* env.clone()  ppo.rs:59 */
    two_puzzle_reset(&env, prm->seed, episode);           /* env.reset()  ppo.rs:60 */
    const int n_cells = (int)(env.width * env.height);
    const int A = pol->n_actions;
    uint32_t t = 0;
    for (;;) {
        /* get_step_data (ppo.rs:41-52) */
        ep_reserve(eb, t + 1, n_cells, A);
        int64_t *obs = eb->obs + (size_t)t * n_cells;
        two_puzzle_observe(&env, obs);
        uint8_t masks[4]; two_puzzle_masks(&env, masks);
        float reward = two_puzzle_reward(&env);
        int perm = -1;
        if (pol->n_perms > 0) {                           /* get_perm_id  policy.rs:67-77 */
            uint32_t w[4]; rng_draw(prm->seed, episode, t, TWO_STREAM_PERM, w);
            perm = (int)u32_below(w[0], (uint32_t)pol->n_perms);
        }
        float *logits = eb->logits + (size_t)t * A;
        float value;
        two_policy_forward(pol, obs, n_cells, masks, perm, prm->arith, logits, &value);
        float u[256];
        for (int blk = 0; blk * 4 < A; ++blk) {
            uint32_t w[4];
            rng_draw(prm->seed, episode, t, TWO_STREAM_GUMBEL | ((uint32_t)blk << 8), w);
            for (int i = 0; i < 4 && blk * 4 + i < A; ++i) u[blk * 4 + i] = u32_to_unit(w[i]);
        }
        int action = two_sample_from_logits(logits, A, u, prm->det_log);
        eb->values[t] = value; eb->rewards[t] = reward; eb->actions[t] = action; eb->perms[t] = perm;
        eb->n = t + 1;
        if (two_puzzle_is_final(&env)) break;             /* ppo.rs:78 */
        two_puzzle_step(&env, action);                    /* ppo.rs:79 */
        ++t;
    }
    eb->advs = (float *)malloc(sizeof(float) * eb->n);
    eb->rets = (float *)malloc(sizeof(float) * eb->n);
    two_gae(eb->rewards, eb->values, (int)eb->n, prm->gamma, prm->lambda, eb->advs, eb->rets);
}

static void *ppo_worker(void *arg)
{
    ppo_job *job = (ppo_job *)arg;
    for (;;) {
        unsigned long long i = atomic_fetch_add(&job->next, 1ULL);
        if (i >= job->prm->num_episodes) break;
        ppo_single_collect(job->env, job->pol, job->prm, job->prm->episode_offset + i, &job->eps[i]);
    }
    return NULL;
}

/* PPOCollector::collect (ppo.rs:108-126): num_cores==1 serial, else a pool of num_cores
 * threads over 0..num_episodes, results in index order, then merge */
int two_ppo_collect(const two_puzzle *env, const two_policy *pol, const two_ppo_params *prm,
                    two_collected *out)
{
    const uint64_t E = prm->num_episodes;
    if (E == 0) return -1;
    episode_buf *eps = (episode_buf *)calloc((size_t)E, sizeof(episode_buf));
    ppo_job job; job.env = env; job.pol = pol; job.prm = prm; job.eps = eps;
    atomic_init(&job.next, 0ULL);
    int nt = prm->num_threads < 1 ? 1 : prm->num_threads;
    if (nt == 1) {
        ppo_worker(&job);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nt);
        for (int i = 0; i < nt; ++i) pthread_create(&th[i], NULL, ppo_worker, &job);
        for (int i = 0; i < nt; ++i) pthread_join(th[i], NULL);
        free(th);
    }
    int rc = merge_episodes(eps, E, (int)(env->width * env->height), pol->n_actions, 1,
                            prm->merge_order, out);
    for (uint64_t e = 0; e < E; ++e) ep_free(&eps[e]);
    free(eps);
    return rc;
}

/* ===================================================================================== */
/* MCTS (rl/search.rs, rl/tree.rs)                                                       */
/* ===================================================================================== */
#define TWO_MAX_ACT 8
typedef struct {
    two_puzzle state;
    int action_taken;            /* -1 = None */
    float prior; uint32_t visit_count; float value_sum;
    int parent;                  /* -1 = None (tree.rs:18-23) */
    int children[TWO_MAX_ACT]; int n_children;
} mcts_node;

typedef struct { mcts_node *nodes; int n, cap; } mcts_tree;

static int tree_new_node(mcts_tree *t, const mcts_node *val)   /* tree.rs:42-46 */
{
    if (t->n == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 64;
        t->nodes = (mcts_node *)realloc(t->nodes, sizeof(mcts_node) * (size_t)t->cap);
    }
    t->nodes[t->n] = *val;
    t->nodes[t->n].parent = -1; t->nodes[t->n].n_children = 0;
    return t->n++;
}

static int tree_add_child(mcts_tree *t, const mcts_node *val, int idx)   /* tree.rs:48-53 */
{
    int c = tree_new_node(t, val);
    t->nodes[idx].children[t->nodes[idx].n_children++] = c;
    t->nodes[c].parent = idx;
    return c;
}

/* MCTSNode::ucb (search.rs:29-39) */
static float node_ucb(const mcts_node *self, const mcts_node *child, float C)
{
    float q = child->visit_count == 0 ? 0.0f : child->value_sum / (float)child->visit_count;
    float a = sqrtf((float)self->visit_count);
    float b = (float)child->visit_count + 1.0f;
    float c = a / b;
    float d = C * c;
    d = d * child->prior;
    return q + d;
}

/* MCTSTree::backpropagate (search.rs:45-53) */
static void tree_backprop(mcts_tree *t, int idx, float value)
{
    while (idx >= 0) {
        t->nodes[idx].value_sum = t->nodes[idx].value_sum + value;
        t->nodes[idx].visit_count += 1;
        idx = t->nodes[idx].parent;
    }
}

/* MCTSTree::expand (search.rs:56-75) */
static void tree_expand(mcts_tree *t, int idx, const float *priors, int n_actions)
{
    for (int a = 0; a < n_actions; ++a) {
        if (priors[a] <= 0.0f) continue;
        mcts_node c; memset(&c, 0, sizeof(c));
        c.state = t->nodes[idx].state;
        two_puzzle_step(&c.state, a);
        c.action_taken = a; c.prior = priors[a]; c.visit_count = 0; c.value_sum = 0.0f;
        tree_add_child(t, &c, idx);
    }
}

/* MCTSTree::next (search.rs:77-91): first-max UCB over children */
static int tree_next(const mcts_tree *t, int idx, float C)
{
    int best = -1; float best_ucb = -INFINITY;
    const mcts_node *n = &t->nodes[idx];
    for (int i = 0; i < n->n_children; ++i) {
        float u = node_ucb(n, &t->nodes[n->children[i]], C);
        if (u > best_ucb) { best = n->children[i]; best_ucb = u; }
    }
    return best;  /* reference panics if none; unreachable for finite priors */
}

/* predict_probs_mcts (search.rs:104-189) */
void two_mcts_probs(const two_puzzle *root, const two_policy *pol, uint32_t num_mcts_searches,
                    float C, uint32_t max_expand_depth, int arith, uint64_t seed,
                    uint64_t episode, uint32_t t, float *probs_out)
{
    const int A = pol->n_actions;
    const int n_cells = (int)(root->width * root->height);
    mcts_tree tree = {0};
    int64_t obs[TWO_MAX_CELLS]; uint8_t masks[4]; float probs[256], val;

    two_puzzle_observe(root, obs); two_puzzle_masks(root, masks);
    two_policy_full_predict(pol, obs, n_cells, masks, arith, probs, &val);       /* :115 */
    mcts_node r; memset(&r, 0, sizeof(r));
    r.state = *root; r.action_taken = -1; r.prior = 0.0f; r.visit_count = 1; r.value_sum = 0.0f;
    int root_idx = tree_new_node(&tree, &r);                                     /* :120-126 */
    tree_expand(&tree, root_idx, probs, A);                                      /* :129 */

    for (uint32_t it = 0; it < num_mcts_searches; ++it) {                        /* :132 */
        int node = root_idx;
        while (tree.nodes[node].n_children > 0) {                                /* :136-138 */
            int nx = tree_next(&tree, node, C);
            if (nx < 0) break;
            node = nx;
        }
        float value = 0.0f; uint32_t expanded = 0;
        while (expanded < max_expand_depth) {                                    /* :143 */
            const two_puzzle *st = &tree.nodes[node].state;
            value = two_puzzle_reward(st);                                       /* :146 */
            if (two_puzzle_is_final(st)) break;                                  /* :149 */
            float nv;
            two_puzzle_observe(st, obs); two_puzzle_masks(st, masks);
            two_policy_full_predict(pol, obs, n_cells, masks, arith, probs, &nv);/* :154-155 */
            tree_expand(&tree, node, probs, A);                                  /* :156 */
            /* next_sample (search.rs:94-100): weighted draw over the children's priors */
            {
                const mcts_node *n = &tree.nodes[node];
                float pri[TWO_MAX_ACT];
                for (int i = 0; i < n->n_children; ++i) pri[i] = tree.nodes[n->children[i]].prior;
                uint32_t w[4];
                rng_draw(seed, episode, it * max_expand_depth + expanded,
                         TWO_STREAM_MCTS | (t << 8), w);
                int c = two_sample_weighted(pri, n->n_children, u32_to_unit(w[0]));
                if (n->n_children > 0) node = n->children[c];
            }
            value = nv;                                                          /* :158 */
            ++expanded;
        }
        tree_backprop(&tree, node, value);                                       /* :163 */
    }

    for (int i = 0; i < A; ++i) probs_out[i] = 0.0f;                             /* :169 */
    const mcts_node *rn = &tree.nodes[root_idx];
    for (int i = 0; i < rn->n_children; ++i) {
        const mcts_node *c = &tree.nodes[rn->children[i]];
        probs_out[c->action_taken] = (float)c->visit_count;
    }
    float sum = 0.0f;
    for (int i = 0; i < A; ++i) sum = sum + probs_out[i];
    if (sum > 0.0f) { for (int i = 0; i < A; ++i) probs_out[i] = probs_out[i] / sum; }
    else { for (int i = 0; i < A; ++i) probs_out[i] = 1.0f / (float)A; }
    free(tree.nodes);
}

/* ===================================================================================== */
/* AZCollector (collector/az.rs)                                                         */
/* ===================================================================================== */
typedef struct {
    const two_puzzle *env; const two_policy *pol; const two_az_params *prm;
    episode_buf *eps; atomic_ullong next;
} az_job;

/* AZCollector::single_collect (az.rs:51-109) */
static void az_single_collect(const two_puzzle *env0, const two_policy *pol,
                              const two_az_params *prm, uint64_t episode, episode_buf *eb)
{
    two_puzzle env = *env0; two_puzzle_reset(&env, prm->seed, episode);
    const int n_cells = (int)(env.width * env.height);
    const int A = pol->n_actions;
    float total_val = 0.0f; uint32_t t = 0;
    float *total_vals = NULL; uint32_t tv_cap = 0;
    for (;;) {
        ep_reserve(eb, t + 1, n_cells, A);
        if (t + 1 > tv_cap) { tv_cap = tv_cap ? tv_cap * 2 : 16; total_vals = (float *)realloc(total_vals, sizeof(float) * tv_cap); }
        float *probs = eb->logits + (size_t)t * A;
        two_mcts_probs(&env, pol, prm->num_mcts_searches, prm->C, prm->max_expand_depth,
                       prm->arith, prm->seed, episode, t, probs);                /* :69 */
        uint32_t w[4]; rng_draw(prm->seed, episode, t, TWO_STREAM_AZ_ACT, w);
        int action = two_sample_weighted(probs, A, u32_to_unit(w[0]));           /* :72 */
        float val = two_puzzle_reward(&env);                                     /* :73 */
        total_vals[t] = total_val;                                               /* :74 */
        total_val = total_val + val;                                             /* :76 */
        two_puzzle_observe(&env, eb->obs + (size_t)t * n_cells);                 /* :79 */
        eb->perms[t] = -1; eb->values[t] = val; eb->actions[t] = action;
        eb->n = t + 1;
        if (two_puzzle_is_final(&env)) break;                                    /* :84 */
        two_puzzle_step(&env, action);                                           /* :89 */
        ++t;
    }
    eb->remaining = (float *)malloc(sizeof(float) * eb->n);
    for (uint32_t i = 0; i < eb->n; ++i) eb->remaining[i] = total_val - total_vals[i];  /* :93 */
    free(total_vals);
}

static void *az_worker(void *arg)
{
    az_job *job = (az_job *)arg;
    g_det_exp = job->prm->det_math;
    for (;;) {
        unsigned long long i = atomic_fetch_add(&job->next, 1ULL);
        if (i >= job->prm->num_episodes) break;
        az_single_collect(job->env, job->pol, job->prm, job->prm->episode_offset + i, &job->eps[i]);
    }
    return NULL;
}

/* AZCollector::collect (az.rs:112-130) */
int two_az_collect(const two_puzzle *env, const two_policy *pol, const two_az_params *prm,
                   two_collected *out)
{
    const int saved_det = g_det_exp;
    const uint64_t E = prm->num_episodes;
    if (E == 0) return -1;
    episode_buf *eps = (episode_buf *)calloc((size_t)E, sizeof(episode_buf));
    az_job job; job.env = env; job.pol = pol; job.prm = prm; job.eps = eps;
    atomic_init(&job.next, 0ULL);
    int nt = prm->num_threads < 1 ? 1 : prm->num_threads;
    if (nt == 1) {
        az_worker(&job);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nt);
        for (int i = 0; i < nt; ++i) pthread_create(&th[i], NULL, az_worker, &job);
        for (int i = 0; i < nt; ++i) pthread_join(th[i], NULL);
        free(th);
    }
    int rc = merge_episodes(eps, E, (int)(env->width * env->height), pol->n_actions, 0,
                            prm->merge_order, out);
    for (uint64_t e = 0; e < E; ++e) ep_free(&eps[e]);
    free(eps);
    g_det_exp = saved_det;
    return rc;
}

/* ===================================================================================== */
/* solve / evaluate (rl/solve.rs, rl/evaluate.rs)                                        */
/* ===================================================================================== */
/* single_solve (solve.rs:17-71); `key` = RNG episode key of this attempt */
static int single_solve(two_puzzle *env, const two_policy *pol, const two_solve_params *prm, uint64_t key,
                        float *success, float *total_out, int64_t *actions)
{
    const int n_cells = (int)(env->width * env->height);
    const int A = pol->n_actions;
    float total_val = 0.0f; int n = 0; uint32_t t = 0;
    while (!two_puzzle_is_final(env)) {                                          /* :30 */
        float val = two_puzzle_reward(env);                                      /* :31 */
        int64_t obs[TWO_MAX_CELLS]; uint8_t masks[4]; float probs[256], v;
        two_puzzle_observe(env, obs); two_puzzle_masks(env, masks);
        total_val = total_val + val;                                             /* :34 */
        if (prm->num_mcts_searches == 0) {                                       /* :37-38 predict */
            int perm = -1;
            if (pol->n_perms > 0) {
                uint32_t w[4]; rng_draw(prm->seed, key, t, TWO_STREAM_PERM, w);
                perm = (int)u32_below(w[0], (uint32_t)pol->n_perms);
            }
            two_policy_predict(pol, obs, n_cells, masks, perm, prm->arith, probs, &v);
        } else {                                                                 /* :41-47 */
            two_mcts_probs(env, pol, prm->num_mcts_searches, prm->C, prm->max_expand_depth, prm->arith,
                           prm->seed, key, t, probs);
        }
        int action;
        if (prm->deterministic) action = two_argmax(probs, A);                   /* :50-51 */
        else {
            uint32_t w[4]; rng_draw(prm->seed, key, t, TWO_STREAM_SOLVE, w);
            action = two_sample_weighted(probs, A, u32_to_unit(w[0]));           /* :53 */
        }
        two_puzzle_step(env, action);                                            /* :56 */
        actions[n++] = action;                                                   /* :58 */
        ++t;
    }
    total_val = total_val + two_puzzle_reward(env);                              /* :65-66 */
    *success = two_puzzle_solved(env) ? 1.0f : 0.0f;                             /* :68 */
    *total_out = total_val;
    return n;
}

/* solve (solve.rs:73-101): best of num_searches attempts by (success, total) tuple order, strict > */
int two_solve(const two_puzzle *env, const two_policy *pol, const two_solve_params *prm, uint64_t episode,
              float *success_out, float *reward_out, int64_t *actions_out)
{
    const int saved = g_det_exp; g_det_exp = prm->det_math;
    float best_s = 0.0f, best_r = -INFINITY; int best_n = 0;
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(env->depth + 2));
    for (uint32_t a = 0; a < prm->num_searches; ++a) {
        two_puzzle e = *env;                                                     /* :85 clone */
        float s, r;
        int n = single_solve(&e, pol, prm, episode * (uint64_t)prm->num_searches + a, &s, &r, tmp);
        if (s > best_s || (s == best_s && r > best_r)) {                         /* :95 tuple '>' */
            best_s = s; best_r = r; best_n = n;
            if (actions_out) memcpy(actions_out, tmp, sizeof(int64_t) * (size_t)n);
        }
    }
    free(tmp);
    g_det_exp = saved;
    *success_out = best_s; *reward_out = best_r;
    return best_n;
}

/* evaluate (evaluate.rs:22-89), serial accumulation order (:36-52) */
void two_evaluate(const two_puzzle *env, const two_policy *pol, const two_solve_params *prm, uint64_t num_episodes,
                  float *success_rate_out, float *mean_reward_out)
{
    float successes = 0.0f, rewards = 0.0f;
    for (uint64_t e = 0; e < num_episodes; ++e) {
        two_puzzle p = *env;
        two_puzzle_reset(&p, prm->seed, e);                                      /* :39 */
        float s, r;
        two_solve(&p, pol, prm, e, &s, &r, NULL);
        successes = successes + s; rewards = rewards + r;
    }
    *success_rate_out = successes / (float)num_episodes;
    *mean_reward_out = rewards / (float)num_episodes;
}

/* ===================================================================================== */
/* replay helper for replay-parity tests                                                 */
/* ===================================================================================== */
void two_replay(const two_puzzle *start, const int64_t *actions, size_t n,
                int64_t *obs_out, uint8_t *masks_out, float *reward_out, uint8_t *final_out,
                int64_t *board_out)
{
    two_puzzle env = *start;
    const size_t nc = (size_t)(env.width * env.height);
    for (size_t t = 0; t <= n; ++t) {
        two_puzzle_observe(&env, obs_out + t * nc);
        two_puzzle_masks(&env, masks_out + t * 4);
        reward_out[t] = two_puzzle_reward(&env);
        final_out[t] = (uint8_t)two_puzzle_is_final(&env);
        for (size_t i = 0; i < nc; ++i) board_out[t * nc + i] = env.state[i];
        if (t < n) two_puzzle_step(&env, actions[t]);
    }
}
