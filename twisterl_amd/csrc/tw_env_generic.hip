// tw_env_generic.hip -- PPO collection for ANY environment: the reference's `Collector::collect(&Box<dyn Env>, &Policy)`
// (rust/src/collector/collector.rs:92-95, ppo.rs:41-126) takes any implementor of `trait Env` (rust/src/rl/env.rs:18-68) --
// the GridWorld crate of examples/grid_world, Python classes behind PyEnv (python_interface/pyenv.rs) -- not only Puzzle.
//
// An environment whose dynamics this library does not implement on the GPU is user code: it arrives as a table of C
// function pointers (tw_env_vtable = the trait's methods the collector calls) and steps on the host, exactly as the
// reference runs it.  Everything else of the path stays on the device: the policy forward of ALL live episodes of a
// time step is ONE batched launch (policy_eval kernels: MFMA-shape or generic stacks, any obs_size), the result is an
// ordinary tw_collected in HBM.  Per record, as ppo.rs:69-80: observe / masks / reward of the CURRENT state, forward_with_perm
// (twist index from the RNG spec of tw_common.hpp), Gumbel arg-max with the spec's uniforms and deterministic log, push,
// `if is_final break`, step.  Then GAE (ppo.rs:82-92, un-fused) and merge order (collector.rs:40-46).
#include "tw_common.hpp"

#include <cstring>
#include <vector>

using namespace tw;

extern "C" int tw_ppo_collect_env(const tw_env_vtable *env, const tw_policy *policy, const tw_ppo_params *prm,
                                  uint32_t max_records_per_episode, tw_collected **out)
{
    if (!env || !policy || !prm || !out) { set_error("tw_ppo_collect_env: null argument"); return TW_ERR_INVALID; }
    *out = nullptr;
    if (!env->prototype || !env->clone || !env->destroy || !env->reset || !env->step || !env->observe || !env->masks || !env->reward || !env->is_final) {
        set_error("tw_ppo_collect_env: the environment table lacks a method"); return TW_ERR_INVALID;
    }
    if (prm->num_episodes == 0) { set_error("Something went wrong. No data in collected data chunks to merge. "); return TW_ERR_EMPTY; }   // collector.rs:41
    if (prm->precision != TW_PREC_F32_EXACT) { set_error("tw_ppo_collect_env: f32 only"); return TW_ERR_UNSUPPORTED; }
    const PolicyDev *pd = policy_dev(policy);
    const uint32_t A = env->num_actions, NO = env->n_obs;
    if (A == 0 || A > 31 || (int)A != pd->n_actions) { set_error("environment has %u actions, policy has %d (at most 31)", A, pd->n_actions); return TW_ERR_INVALID; }
    if (NO == 0 || NO > 64) { set_error("tw_ppo_collect_env: observations of %u ids (1..64 supported)", NO); return TW_ERR_UNSUPPORTED; }
    if ((int)env->obs_size != pd->obs_size) { set_error("index out of bounds: policy obs_size %d != environment obs ids %u", pd->obs_size, env->obs_size); return TW_ERR_INVALID; }
    if (max_records_per_episode == 0) { set_error("tw_ppo_collect_env: max_records_per_episode must be positive"); return TW_ERR_INVALID; }
    int rc = require_device(); if (rc) return rc;

    const uint64_t E = prm->num_episodes;
    const uint32_t OW = pd->obs_size > 256 ? 2u : 1u;                      // bytes per obs id in the result
    hipStream_t s = current_stream();
    struct Ep {
        void *env = nullptr; bool alive = true;
        std::vector<int32_t> obs; std::vector<float> logits, values, rewards; std::vector<int32_t> actions, perms;
    };
    std::vector<Ep> eps(E);
    auto cleanup = [&]() { for (auto &e : eps) if (e.env) { env->destroy(e.env); e.env = nullptr; } };
    for (uint64_t i = 0; i < E; ++i) {                                     // ppo.rs:59-60: clone + reset per episode
        eps[i].env = env->clone(env->prototype);
        if (!eps[i].env) { cleanup(); set_error("tw_ppo_collect_env: clone() returned null"); return TW_ERR_INVALID; }
        env->reset(eps[i].env, prm->seed, prm->episode_offset + i);
    }
    // device staging for one time step of all live episodes
    const size_t b_obs = (size_t)E * NO * 4, b_m = (size_t)E * A, b_p = (size_t)E * 4, b_la = (size_t)E * A * 4, b_v = (size_t)E * 4;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_obs = 0, o_m = up(b_obs), o_p = o_m + up(b_m), o_la = o_p + up(b_p), o_v = o_la + up(b_la), tot = o_v + up(b_v);
    uint8_t *dev = nullptr;
    hipError_t he = hipMalloc((void **)&dev, tot);
    if (he != hipSuccess) { cleanup(); return hip_fail(he, "hipMalloc(step staging)", __FILE__, __LINE__); }
    std::vector<int32_t> h_obs((size_t)E * NO), h_perm(E); std::vector<uint8_t> h_m((size_t)E * A); std::vector<float> h_la((size_t)E * A), h_v(E);
    std::vector<uint64_t> live; live.reserve(E);
#define TW_HIP_E(call) do { hipError_t _e = (call); if (_e != hipSuccess) { cleanup(); (void)hipFree(dev); return hip_fail(_e, #call, __FILE__, __LINE__); } } while (0)
    for (uint32_t t = 0;; ++t) {
        live.clear();
        for (uint64_t i = 0; i < E; ++i) if (eps[i].alive) live.push_back(i);
        if (live.empty()) break;
        if (t >= max_records_per_episode) { cleanup(); (void)hipFree(dev); set_error("tw_ppo_collect_env: an episode did not end within %u records", max_records_per_episode); return TW_ERR_INVALID; }
        const uint32_t n = (uint32_t)live.size();
        for (uint32_t r = 0; r < n; ++r) {                                  // get_step_data (ppo.rs:46-48)
            Ep &e = eps[live[r]];
            env->observe(e.env, &h_obs[(size_t)r * NO]);
            for (uint32_t c = 0; c < NO; ++c) {
                const int32_t id = h_obs[(size_t)r * NO + c];
                if (id < 0 || id >= pd->obs_size) { cleanup(); (void)hipFree(dev); set_error("index out of bounds: obs id %d, obs_size %d", id, pd->obs_size); return TW_ERR_INVALID; }
            }
            env->masks(e.env, &h_m[(size_t)r * A]);
            e.rewards.push_back(env->reward(e.env));
            int32_t perm = -1;                                               // get_perm_id (policy.rs:67-77)
            if (pd->n_perms > 0) perm = (int32_t)u32_below(rng_draw(prm->seed, prm->episode_offset + live[r], t, STREAM_PERM).x, (uint32_t)pd->n_perms);
            h_perm[r] = perm;
            e.obs.insert(e.obs.end(), &h_obs[(size_t)r * NO], &h_obs[(size_t)r * NO] + NO);
            e.perms.push_back(perm);
        }
        TW_HIP_E(hipMemcpyAsync(dev + o_obs, h_obs.data(), (size_t)n * NO * 4, hipMemcpyHostToDevice, s));
        TW_HIP_E(hipMemcpyAsync(dev + o_m, h_m.data(), (size_t)n * A, hipMemcpyHostToDevice, s));
        TW_HIP_E(hipMemcpyAsync(dev + o_p, h_perm.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        rc = launch_policy_eval(*pd, TW_EVAL_FORWARD, reinterpret_cast<const int32_t *>(dev + o_obs), n, NO, dev + o_m,
                                reinterpret_cast<const int32_t *>(dev + o_p), reinterpret_cast<float *>(dev + o_la), reinterpret_cast<float *>(dev + o_v), s);
        if (rc) { cleanup(); (void)hipFree(dev); return rc; }
        TW_HIP_E(hipMemcpyAsync(h_la.data(), dev + o_la, (size_t)n * A * 4, hipMemcpyDeviceToHost, s));
        TW_HIP_E(hipMemcpyAsync(h_v.data(), dev + o_v, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        TW_HIP_E(hipStreamSynchronize(s));
        for (uint32_t r = 0; r < n; ++r) {
            Ep &e = eps[live[r]];
            const float *l = &h_la[(size_t)r * A];
            // sample_from_logits (policy.rs:169-172) + argmax (:130-151): one uniform per element, first maximum wins, NaN never
            int best = 0; float bv = 0.0f;
            for (uint32_t i = 0; i < A; ++i) {
                const u32x4 w = rng_draw(prm->seed, prm->episode_offset + live[r], t | ((i >> 2) << 24), STREAM_GUMBEL);
                const uint32_t word = (i & 3u) == 0 ? w.x : ((i & 3u) == 1 ? w.y : ((i & 3u) == 2 ? w.z : w.w));
                const float a1 = tw_logf(u32_to_unit(word));
                const float gi = l[i] - tw_logf(__builtin_fabsf(a1));
                if (i == 0) { bv = gi; best = 0; } else if (gi > bv) { bv = gi; best = (int)i; }
            }
            e.logits.insert(e.logits.end(), l, l + A);
            e.values.push_back(h_v[r]);
            e.actions.push_back(best);
            if (env->is_final(e.env)) e.alive = false;                      // ppo.rs:78
            else env->step(e.env, (uint32_t)best);                           // ppo.rs:79
        }
    }
#undef TW_HIP_E
    (void)hipFree(dev);
    cleanup();

    // ---- GAE (ppo.rs:82-92) + merge (collector.rs:40-46) into host images of the compact fields ----------------------
    uint64_t total = 0;
    for (auto &e : eps) total += e.values.size();
    std::vector<uint64_t> order(E);
    for (uint64_t p = 0; p < E; ++p) order[p] = prm->merge_order ? (p == 0 ? E - 1 : p - 1) : p;
    std::vector<uint8_t> f_obs((size_t)total * NO * OW), f_act(total); std::vector<int8_t> f_perm(total);
    std::vector<float> f_lg((size_t)total * A), f_val(total), f_rew(total), f_adv(total), f_ret(total);
    std::vector<uint32_t> f_len(E); std::vector<uint64_t> f_start(E);
    uint64_t pos = 0;
    for (uint64_t p = 0; p < E; ++p) {
        const Ep &e = eps[order[p]];
        const size_t nrec = e.values.size();
        f_len[order[p]] = (uint32_t)nrec; f_start[order[p]] = pos;
        std::vector<float> adv(nrec), ret(nrec);
        adv[nrec - 1] = e.rewards[nrec - 1] - e.values[nrec - 1];
        ret[nrec - 1] = e.rewards[nrec - 1];
        for (size_t tt = nrec - 1; tt-- > 0;) {
            float inner = prm->lambda * adv[tt + 1];
            inner = e.values[tt + 1] + inner;
            inner = prm->gamma * inner;
            ret[tt] = e.rewards[tt] + inner;
            adv[tt] = ret[tt] - e.values[tt];
        }
        for (size_t tt = 0; tt < nrec; ++tt) {
            for (uint32_t c = 0; c < NO; ++c) {
                const int32_t id = e.obs[tt * NO + c];
                if (OW == 1) f_obs[(pos + tt) * NO + c] = (uint8_t)id;
                else { const uint16_t v = (uint16_t)id; memcpy(&f_obs[((pos + tt) * NO + c) * 2], &v, 2); }
            }
            memcpy(&f_lg[(pos + tt) * A], &e.logits[tt * A], A * 4);
            f_perm[pos + tt] = (int8_t)e.perms[tt]; f_val[pos + tt] = e.values[tt]; f_rew[pos + tt] = e.rewards[tt];
            f_act[pos + tt] = (uint8_t)e.actions[tt]; f_adv[pos + tt] = adv[tt]; f_ret[pos + tt] = ret[tt];
        }
        pos += nrec;
    }
    // ---- the result object: one device allocation ---------------------------------------------------------------------
    size_t cur = 0, off[TW_F_COUNT] = {}, bytes[TW_F_COUNT] = {};
    const void *src[TW_F_COUNT] = {};
    auto put = [&](int f, const void *p, size_t b) { src[f] = p; bytes[f] = b; off[f] = cur; cur = (cur + b + 255) / 256 * 256; };
    put(TW_F_OBS, f_obs.data(), f_obs.size()); put(TW_F_LOGITS, f_lg.data(), f_lg.size() * 4); put(TW_F_PERMS, f_perm.data(), f_perm.size());
    put(TW_F_VALUES, f_val.data(), total * 4); put(TW_F_REWARDS, f_rew.data(), total * 4); put(TW_F_ACTIONS, f_act.data(), total);
    put(TW_F_ADVS, f_adv.data(), total * 4); put(TW_F_RETS, f_ret.data(), total * 4); put(TW_F_EP_LEN, f_len.data(), E * 4); put(TW_F_EP_START, f_start.data(), E * 8);
    void *arena = nullptr;
    TW_HIP(hipMalloc(&arena, cur ? cur : 256));
    void *fp[TW_F_COUNT] = {};
    for (int f = 0; f < TW_F_COUNT; ++f) if (bytes[f]) {
        fp[f] = reinterpret_cast<uint8_t *>(arena) + off[f];
        he = hipMemcpyAsync(fp[f], src[f], bytes[f], hipMemcpyHostToDevice, s);
        if (he != hipSuccess) { (void)hipFree(arena); return hip_fail(he, "upload of the collected fields", __FILE__, __LINE__); }
    }
    he = hipStreamSynchronize(s);
    if (he != hipSuccess) { (void)hipFree(arena); return hip_fail(he, "upload of the collected fields", __FILE__, __LINE__); }
    int dev_id = 0; (void)hipGetDevice(&dev_id);
    rc = collected_adopt(arena, cur ? cur : 256, dev_id, 1, NO, A, total, E, fp, bytes, out);
    if (rc) { (void)hipFree(arena); return rc; }
    collected_adopt_obs_width(*out, OW);
    return TW_OK;
}

// ======================================================================================================================
// Self-play, evaluate and solve for ANY environment (rust/src/collector/az.rs:51-109, rust/src/rl/evaluate.rs:22-89,
// rust/src/rl/solve.rs:17-101 over predict_probs_mcts, rust/src/rl/search.rs:104-189).  The search trees live on the host (a
// node owns a clone of the environment, as MCTSNode does); every episode / attempt is a little state machine that runs until
// it needs a policy output for a state, and all the states wanted at that moment are evaluated by ONE batched launch.  Same
// RNG keys and the same f32 operation order as the device kernels (tw_mcts.hip, tw_solve.hip).
namespace {

int sample_weighted_host(const float *w, int n, float u)             // nn::policy::sample (policy.rs:153-167), see tw_common.hpp
{
    if (n <= 0) return 0;
    float total = 0.0f;
    std::vector<float> cum((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (!(w[i] >= 0.0f)) return 0;
        total = total + w[i];
        cum[(size_t)i] = total;
    }
    if (!(total > 0.0f)) return 0;
    const float chosen = u * total;
    int idx = 0;
    while (idx < n - 1 && cum[(size_t)idx] <= chosen) ++idx;
    return idx;
}

struct HostNode {                       // MCTSNode + its place in Tree<T> (search.rs:20-26, tree.rs:14-54)
    void *state; int parent; int action; float prior; uint32_t visit; float value_sum; int first_child, n_children;
};

// predict_probs_mcts (search.rs:104-189) as a resumable machine: run() goes on until the search wants Policy::full_predict of
// a state (returns that state) or the move's visit counts are final (returns nullptr, the probabilities in `mp`).
struct HostMcts {
    const tw_env_vtable *env; uint32_t A, S, MED; float C; uint64_t seed;
    std::vector<HostNode> tree;
    uint32_t it = 0, expanded = 0; int node = -1; float value = 0.0f; bool rooted = false, waiting = false;

    void clear() { for (auto &n : tree) if (n.state) env->destroy(n.state); tree.clear(); it = 0; expanded = 0; node = -1; rooted = false; waiting = false; }
    void expand(int idx, const float *priors)                              // search.rs:56-75: a child per action with prior > 0
    {
        tree[(size_t)idx].first_child = (int)tree.size();
        int cnt = 0;
        for (uint32_t a = 0; a < A; ++a) {
            if (!(priors[a] > 0.0f)) continue;
            void *st = env->clone(tree[(size_t)idx].state);
            env->step(st, a);
            tree.push_back(HostNode{st, idx, (int)a, priors[a], 0u, 0.0f, -1, 0});
            ++cnt;
        }
        tree[(size_t)idx].n_children = cnt;
    }
    void backprop(int idx, float v)                                         // search.rs:45-53
    {
        while (idx >= 0) { HostNode &n = tree[(size_t)idx]; n.value_sum = n.value_sum + v; n.visit += 1u; idx = n.parent; }
    }
    // `root_env`: the state the move starts from; (key, t): the RNG key of the episode / attempt and the move index;
    // probs_in / value_in: the output this search was waiting for (nullptr on the first call of a move)
    void *run(void *root_env, uint64_t key, uint32_t t, const float *probs_in, float value_in, std::vector<float> &mp)
    {
        bool resume = probs_in != nullptr;
        for (;;) {
            if (!rooted) {
                if (!resume) { waiting = true; return root_env; }
                // root (search.rs:110-129): visit_count 1, expanded with the root's priors
                tree.push_back(HostNode{env->clone(root_env), -1, -1, 0.0f, 1u, 0.0f, -1, 0});
                expand(0, probs_in);
                it = 0; resume = false; rooted = true; node = -1;
                continue;
            } else if (resume) {
                // the leaf's output (search.rs:154-159): expand, sample a child by the priors, the value is the network's
                resume = false;
                expand(node, probs_in);
                const HostNode &nd = tree[(size_t)node];
                if (nd.n_children > 0) {                                   // (the reference panics on a node without children here)
                    std::vector<float> pri((size_t)nd.n_children);
                    for (int c = 0; c < nd.n_children; ++c) pri[(size_t)c] = tree[(size_t)(nd.first_child + c)].prior;
                    const u32x4 w = rng_draw(seed, key, it * MED + expanded, (uint32_t)STREAM_MCTS | (t << 8));
                    node = nd.first_child + sample_weighted_host(pri.data(), nd.n_children, u32_to_unit(w.x));
                }
                value = value_in;
                ++expanded;
            } else if (node < 0) {
                if (it == S) {
                    // visit counts -> probs (search.rs:166-188)
                    mp.assign(A, 0.0f);
                    const HostNode &root = tree[0];
                    for (int c = 0; c < root.n_children; ++c) { const HostNode &ch = tree[(size_t)(root.first_child + c)]; mp[(size_t)ch.action] = (float)ch.visit; }
                    float sum = 0.0f;
                    for (uint32_t a = 0; a < A; ++a) sum = sum + mp[a];
                    if (sum > 0.0f) { for (uint32_t a = 0; a < A; ++a) mp[a] = mp[a] / sum; }
                    else { for (uint32_t a = 0; a < A; ++a) mp[a] = 1.0f / (float)A; }
                    clear();
                    return nullptr;
                }
                // descend by UCB (search.rs:133-138, next :77-91, ucb :29-39)
                int idx = 0;
                while (tree[(size_t)idx].n_children > 0) {
                    const HostNode &par = tree[(size_t)idx];
                    int best = -1; float best_ucb = -__builtin_inff();
                    const float sq = sqrtf((float)par.visit);
                    for (int c = 0; c < par.n_children; ++c) {
                        const HostNode &ch = tree[(size_t)(par.first_child + c)];
                        const float q = ch.visit == 0u ? 0.0f : ch.value_sum / (float)ch.visit;
                        float d = sq / ((float)ch.visit + 1.0f);
                        d = C * d;
                        d = d * ch.prior;
                        const float ucb = q + d;
                        if (ucb > best_ucb) { best = par.first_child + c; best_ucb = ucb; }
                    }
                    if (best < 0) break;                                   // all-NaN UCB: the reference panics here
                    idx = best;
                }
                node = idx; value = 0.0f; expanded = 0;
            }
            // leaf phase (search.rs:143-160)
            bool need = false;
            while (expanded < MED) {
                void *st = tree[(size_t)node].state;
                value = env->reward(st);
                if (env->is_final(st)) break;
                need = true; break;
            }
            if (need) { waiting = true; return tree[(size_t)node].state; }
            backprop(node, value);                                          // search.rs:163
            ++it; node = -1;
        }
    }
};

// one batched policy evaluation of `n` host states (mode: TW_EVAL_FULL_PREDICT, or TW_EVAL_PREDICT with a twist per state)
struct HostEvalBatch {
    const tw_env_vtable *env; const PolicyDev *pd; uint32_t A, NO; hipStream_t s;
    uint8_t *dev = nullptr; size_t o_obs = 0, o_m = 0, o_p = 0, o_la = 0, o_v = 0;
    std::vector<int32_t> h_obs, h_perm; std::vector<uint8_t> h_m; std::vector<float> h_la, h_v;
    int init(uint64_t cap)
    {
        auto up = [](size_t x) { return (x + 255) / 256 * 256; };
        o_obs = 0; o_m = up(cap * NO * 4); o_p = o_m + up(cap * A); o_la = o_p + up(cap * 4); o_v = o_la + up(cap * A * 4);
        TW_HIP(hipMalloc((void **)&dev, o_v + up(cap * 4)));
        h_obs.resize(cap * NO); h_perm.resize(cap); h_m.resize(cap * A); h_la.resize(cap * A); h_v.resize(cap);
        return TW_OK;
    }
    ~HostEvalBatch() { if (dev) (void)hipFree(dev); }
    int stage(uint32_t r, void *st, int32_t perm)
    {
        env->observe(st, &h_obs[(size_t)r * NO]);
        for (uint32_t c = 0; c < NO; ++c) {
            const int32_t id = h_obs[(size_t)r * NO + c];
            if (id < 0 || id >= pd->obs_size) { set_error("index out of bounds: obs id %d, obs_size %d", id, pd->obs_size); return TW_ERR_INVALID; }
        }
        env->masks(st, &h_m[(size_t)r * A]);
        h_perm[r] = perm;
        return TW_OK;
    }
    int run(int mode, uint32_t n)
    {
        TW_HIP(hipMemcpyAsync(dev + o_obs, h_obs.data(), (size_t)n * NO * 4, hipMemcpyHostToDevice, s));
        TW_HIP(hipMemcpyAsync(dev + o_m, h_m.data(), (size_t)n * A, hipMemcpyHostToDevice, s));
        if (mode != TW_EVAL_FULL_PREDICT) TW_HIP(hipMemcpyAsync(dev + o_p, h_perm.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        const int rc = launch_policy_eval(*pd, mode, reinterpret_cast<const int32_t *>(dev + o_obs), n, NO, dev + o_m,
                                          mode != TW_EVAL_FULL_PREDICT ? reinterpret_cast<const int32_t *>(dev + o_p) : nullptr,
                                          reinterpret_cast<float *>(dev + o_la), reinterpret_cast<float *>(dev + o_v), s);
        if (rc) return rc;
        TW_HIP(hipMemcpyAsync(h_la.data(), dev + o_la, (size_t)n * A * 4, hipMemcpyDeviceToHost, s));
        TW_HIP(hipMemcpyAsync(h_v.data(), dev + o_v, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        TW_HIP(hipStreamSynchronize(s));
        return TW_OK;
    }
};

int check_env_table(const tw_env_vtable *env, const PolicyDev *pd, const char *who)
{
    if (!env->prototype || !env->clone || !env->destroy || !env->reset || !env->step || !env->observe || !env->masks || !env->reward || !env->is_final) {
        set_error("%s: the environment table lacks a method", who); return TW_ERR_INVALID;
    }
    const uint32_t A = env->num_actions, NO = env->n_obs;
    if (A == 0 || A > 31 || (int)A != pd->n_actions) { set_error("environment has %u actions, policy has %d (at most 31)", A, pd->n_actions); return TW_ERR_INVALID; }
    if (NO == 0 || NO > 64) { set_error("%s: observations of %u ids (1..64 supported)", who, NO); return TW_ERR_UNSUPPORTED; }
    if ((int)env->obs_size != pd->obs_size) { set_error("index out of bounds: policy obs_size %d != environment obs ids %u", pd->obs_size, env->obs_size); return TW_ERR_INVALID; }
    return TW_OK;
}

}  // namespace

extern "C" int tw_az_collect_env(const tw_env_vtable *env, const tw_policy *policy, const tw_az_params *prm,
                                 uint32_t max_records_per_episode, tw_collected **out)
{
    if (!env || !policy || !prm || !out) { set_error("tw_az_collect_env: null argument"); return TW_ERR_INVALID; }
    *out = nullptr;
    if (prm->num_episodes == 0) { set_error("Something went wrong. No data in collected data chunks to merge. "); return TW_ERR_EMPTY; }   // collector.rs:41
    if (prm->precision != TW_PREC_F32_EXACT) { set_error("tw_az_collect_env: f32 only"); return TW_ERR_UNSUPPORTED; }
    const PolicyDev *pd = policy_dev(policy);
    int rc = check_env_table(env, pd, "tw_az_collect_env"); if (rc) return rc;
    if (max_records_per_episode == 0) { set_error("tw_az_collect_env: max_records_per_episode must be positive"); return TW_ERR_INVALID; }
    rc = require_device(); if (rc) return rc;
    const uint32_t A = env->num_actions, NO = env->n_obs;
    const uint64_t E = prm->num_episodes;
    const uint32_t OW = pd->obs_size > 256 ? 2u : 1u;
    struct Ep {
        void *env = nullptr; bool done = false; uint32_t t = 0; HostMcts mc;
        std::vector<int32_t> obs; std::vector<float> probs, vals;
    };
    std::vector<Ep> eps(E);
    auto cleanup = [&]() { for (auto &e : eps) { e.mc.clear(); if (e.env) { env->destroy(e.env); e.env = nullptr; } } };
    for (uint64_t i = 0; i < E; ++i) {                                     // az.rs:56-57
        eps[i].mc.env = env; eps[i].mc.A = A; eps[i].mc.S = prm->num_mcts_searches; eps[i].mc.MED = prm->max_expand_depth; eps[i].mc.C = prm->C; eps[i].mc.seed = prm->seed;
        eps[i].env = env->clone(env->prototype);
        if (!eps[i].env) { cleanup(); set_error("tw_az_collect_env: clone() returned null"); return TW_ERR_INVALID; }
        env->reset(eps[i].env, prm->seed, prm->episode_offset + i);
    }
    HostEvalBatch eb{env, pd, A, NO, current_stream()};
    rc = eb.init(E); if (rc) { cleanup(); return rc; }
    std::vector<uint64_t> want; want.reserve(E);
    std::vector<void *> want_state(E);
    std::vector<float> mp;
    bool overlong = false;
    // runs episode i until its search wants a network output (returns the state) or the episode is over (nullptr)
    auto advance = [&](uint64_t i, const float *probs_in, float value_in) -> void * {
        Ep &e = eps[i];
        for (;;) {
            if (e.done) return nullptr;
            void *st = e.mc.run(e.env, prm->episode_offset + i, e.t, probs_in, value_in, mp);
            if (st) return st;
            probs_in = nullptr;
            // az.rs:72-89: sample the action, store the record, stop at a final state, else step
            const u32x4 w = rng_draw(prm->seed, prm->episode_offset + i, e.t, STREAM_AZ_ACT);
            const int action = sample_weighted_host(mp.data(), (int)A, u32_to_unit(w.x));
            std::vector<int32_t> ob(NO);
            env->observe(e.env, ob.data());
            e.obs.insert(e.obs.end(), ob.begin(), ob.end());
            e.probs.insert(e.probs.end(), mp.begin(), mp.end());
            e.vals.push_back(env->reward(e.env));
            if (env->is_final(e.env)) { e.done = true; return nullptr; }
            if (e.t + 1 >= max_records_per_episode) { e.done = true; overlong = true; return nullptr; }
            env->step(e.env, (uint32_t)action);
            ++e.t;
        }
    };
    for (uint64_t i = 0; i < E; ++i) want_state[i] = advance(i, nullptr, 0.0f);
    for (;;) {
        want.clear();
        for (uint64_t i = 0; i < E; ++i) if (!eps[i].done && want_state[i]) want.push_back(i);
        if (want.empty() || overlong) break;
        const uint32_t n = (uint32_t)want.size();
        for (uint32_t r = 0; r < n && rc == TW_OK; ++r) rc = eb.stage(r, want_state[want[r]], -1);
        if (rc == TW_OK) rc = eb.run(TW_EVAL_FULL_PREDICT, n);
        if (rc) { cleanup(); return rc; }
        for (uint32_t r = 0; r < n; ++r) want_state[want[r]] = advance(want[r], &eb.h_la[(size_t)r * A], eb.h_v[r]);
    }
    if (overlong) { cleanup(); set_error("tw_az_collect_env: an episode did not end within %u records", max_records_per_episode); return TW_ERR_INVALID; }

    // ---- remaining values (az.rs:94-95) + merge (collector.rs:40-46) --------------------------------------------------
    uint64_t total = 0;
    for (auto &e : eps) total += e.vals.size();
    std::vector<uint64_t> order(E);
    for (uint64_t p = 0; p < E; ++p) order[p] = prm->merge_order ? (p == 0 ? E - 1 : p - 1) : p;
    std::vector<uint8_t> f_obs((size_t)total * NO * OW); std::vector<int8_t> f_perm(total, (int8_t)-1);
    std::vector<float> f_lg((size_t)total * A), f_rem(total);
    std::vector<uint32_t> f_len(E); std::vector<uint64_t> f_start(E);
    uint64_t pos = 0;
    for (uint64_t p = 0; p < E; ++p) {
        const Ep &e = eps[order[p]];
        const size_t nrec = e.vals.size();
        f_len[order[p]] = (uint32_t)nrec; f_start[order[p]] = pos;
        float total_val = 0.0f;
        std::vector<float> before(nrec);
        for (size_t tt = 0; tt < nrec; ++tt) { before[tt] = total_val; total_val = total_val + e.vals[tt]; }
        for (size_t tt = 0; tt < nrec; ++tt) {
            for (uint32_t c = 0; c < NO; ++c) {
                const int32_t id = e.obs[tt * NO + c];
                if (OW == 1) f_obs[(pos + tt) * NO + c] = (uint8_t)id;
                else { const uint16_t v = (uint16_t)id; memcpy(&f_obs[((pos + tt) * NO + c) * 2], &v, 2); }
            }
            memcpy(&f_lg[(pos + tt) * A], &e.probs[tt * A], A * 4);
            f_rem[pos + tt] = total_val - before[tt];
        }
        pos += nrec;
    }
    cleanup();
    hipStream_t s = current_stream();
    size_t cur = 0, off[TW_F_COUNT] = {}, bytes[TW_F_COUNT] = {};
    const void *src[TW_F_COUNT] = {};
    auto put = [&](int f, const void *p, size_t b) { src[f] = p; bytes[f] = b; off[f] = cur; cur = (cur + b + 255) / 256 * 256; };
    put(TW_F_OBS, f_obs.data(), f_obs.size()); put(TW_F_LOGITS, f_lg.data(), f_lg.size() * 4); put(TW_F_PERMS, f_perm.data(), f_perm.size());
    put(TW_F_REMAINING, f_rem.data(), total * 4); put(TW_F_EP_LEN, f_len.data(), E * 4); put(TW_F_EP_START, f_start.data(), E * 8);
    void *arena = nullptr;
    TW_HIP(hipMalloc(&arena, cur ? cur : 256));
    void *fp[TW_F_COUNT] = {};
    hipError_t he = hipSuccess;
    for (int f = 0; f < TW_F_COUNT; ++f) if (bytes[f]) {
        fp[f] = reinterpret_cast<uint8_t *>(arena) + off[f];
        he = hipMemcpyAsync(fp[f], src[f], bytes[f], hipMemcpyHostToDevice, s);
        if (he != hipSuccess) { (void)hipFree(arena); return hip_fail(he, "upload of the collected fields", __FILE__, __LINE__); }
    }
    he = hipStreamSynchronize(s);
    if (he != hipSuccess) { (void)hipFree(arena); return hip_fail(he, "upload of the collected fields", __FILE__, __LINE__); }
    int dev_id = 0; (void)hipGetDevice(&dev_id);
    rc = collected_adopt(arena, cur ? cur : 256, dev_id, 0, NO, A, total, E, fp, bytes, out);
    if (rc) { (void)hipFree(arena); return rc; }
    collected_adopt_obs_width(*out, OW);
    return TW_OK;
}

// ---- single_solve / solve / evaluate (solve.rs:17-101, evaluate.rs:22-89) -------------------------------------------------
// All attempts (episode, search) in lockstep; attempt keys, twist draws and action draws as in tw_solve.hip / tw_mcts.hip's
// solve mode.  from_state: every attempt starts from a clone of the prototype as it is (solve), else from reset(seed, episode).
namespace {

int run_attempts_env(const tw_env_vtable *env, const tw_policy *policy, const tw_solve_params *prm, uint64_t n_episodes, uint64_t episode_offset,
                     bool from_state, uint32_t max_steps, std::vector<float> &best_s, std::vector<float> &best_r, std::vector<uint32_t> *best_actions)
{
    if (prm->precision != TW_PREC_F32_EXACT) { set_error("solve: f32 only for this environment"); return TW_ERR_UNSUPPORTED; }
    if (!env->success) { set_error("solve / evaluate: the environment table has no success()"); return TW_ERR_INVALID; }
    const PolicyDev *pd = policy_dev(policy);
    int rc = check_env_table(env, pd, "solve / evaluate"); if (rc) return rc;
    rc = require_device(); if (rc) return rc;
    best_s.assign(n_episodes, 0.0f); best_r.assign(n_episodes, -__builtin_inff());
    if (prm->num_searches == 0 || n_episodes == 0) return TW_OK;
    const uint32_t A = env->num_actions, NO = env->n_obs;
    const uint64_t N = prm->num_searches, NA = n_episodes * N;
    const bool mcts = prm->num_mcts_searches != 0;
    struct Att { void *env = nullptr; bool done = false, track = false; uint32_t t = 0; float total = 0.0f, success = 0.0f; HostMcts mc; std::vector<uint32_t> actions; int32_t perm = -1; };
    std::vector<Att> at(NA);
    auto cleanup = [&]() { for (auto &a : at) { a.mc.clear(); if (a.env) { env->destroy(a.env); a.env = nullptr; } } };
    auto key_of = [&](uint64_t i) { return (episode_offset + i / N) * N + i % N; };
    for (uint64_t i = 0; i < NA; ++i) {
        Att &a = at[i];
        a.mc.env = env; a.mc.A = A; a.mc.S = prm->num_mcts_searches; a.mc.MED = prm->max_expand_depth; a.mc.C = prm->C; a.mc.seed = prm->seed;
        a.env = env->clone(env->prototype);                                  // solve.rs:85 / evaluate.rs:39
        if (!a.env) { cleanup(); set_error("solve: clone() returned null"); return TW_ERR_INVALID; }
        if (!from_state) env->reset(a.env, prm->seed, episode_offset + i / N);
        a.track = env->track_solution && env->track_solution(a.env);        // solve.rs:28: asked once, before the first move
    }
    HostEvalBatch eb{env, pd, A, NO, current_stream()};
    rc = eb.init(NA); if (rc) { cleanup(); return rc; }
    std::vector<void *> want_state(NA, nullptr);
    std::vector<float> mp;
    bool overlong = false;
    // the end of single_solve (solve.rs:62-70): an environment that tracks its own solution hands it over in place of the played actions
    auto finish = [&](Att &a) {
        if (a.track) {
            a.actions.clear();
            if (env->solution) {
                const uint32_t n = env->solution(a.env, nullptr, 0);
                a.actions.resize(n);
                if (n) env->solution(a.env, a.actions.data(), n);
            }
        }
        a.total = a.total + env->reward(a.env); a.success = env->success(a.env) ? 1.0f : 0.0f; a.done = true;
    };
    // the move of attempt i given the action probabilities (solve.rs:31-58)
    auto move = [&](uint64_t i, const float *probs) {
        Att &a = at[i];
        a.total = a.total + env->reward(a.env);
        int action = 0;
        if (prm->deterministic) {                                          // argmax (policy.rs:130-151): first maximum, NaN never
            float bv = probs[0];
            for (uint32_t k = 1; k < A; ++k) if (probs[k] > bv) { bv = probs[k]; action = (int)k; }
        } else {
            const u32x4 w = rng_draw(prm->seed, key_of(i), a.t, STREAM_SOLVE);
            action = sample_weighted_host(probs, (int)A, u32_to_unit(w.x));
        }
        env->step(a.env, (uint32_t)action);
        if (!a.track) a.actions.push_back((uint32_t)action);                // solve.rs:57-59
        ++a.t;
        if (env->is_final(a.env)) finish(a);
        else if (a.t >= max_steps) { a.done = true; overlong = true; }
    };
    // runs attempt i until it wants a policy output (returns the state) or is over (nullptr)
    auto advance = [&](uint64_t i, const float *out_in, float value_in) -> void * {
        Att &a = at[i];
        for (;;) {
            if (a.done) return nullptr;
            if (mcts) {
                void *st = a.mc.run(a.env, key_of(i), a.t, out_in, value_in, mp);
                if (st) return st;
                out_in = nullptr;
                move(i, mp.data());
            } else {
                if (!out_in) {                                             // Policy::predict (policy.rs:34-49): a random twist, soft-max of the masked logits
                    a.perm = -1;
                    if (pd->n_perms > 0) a.perm = (int32_t)u32_below(rng_draw(prm->seed, key_of(i), a.t, STREAM_PERM).x, (uint32_t)pd->n_perms);
                    return a.env;
                }
                move(i, out_in);
                out_in = nullptr;
            }
        }
    };
    for (uint64_t i = 0; i < NA; ++i) {
        if (env->is_final(at[i].env)) finish(at[i]);                        // `while !env.is_final()` never entered (solve.rs:29)
        else want_state[i] = advance(i, nullptr, 0.0f);
    }
    std::vector<uint64_t> want; want.reserve(NA);
    for (;;) {
        want.clear();
        for (uint64_t i = 0; i < NA; ++i) if (!at[i].done && want_state[i]) want.push_back(i);
        if (want.empty() || overlong) break;
        const uint32_t n = (uint32_t)want.size();
        for (uint32_t r = 0; r < n && rc == TW_OK; ++r) rc = eb.stage(r, want_state[want[r]], mcts ? -1 : at[want[r]].perm);
        if (rc == TW_OK) rc = eb.run(mcts ? TW_EVAL_FULL_PREDICT : TW_EVAL_PREDICT, n);
        if (rc) { cleanup(); return rc; }
        for (uint32_t r = 0; r < n; ++r) want_state[want[r]] = advance(want[r], &eb.h_la[(size_t)r * A], eb.h_v[r]);
    }
    if (overlong) { cleanup(); set_error("solve: an attempt did not end within %u steps", max_steps); return TW_ERR_INVALID; }
    std::vector<uint64_t> best_att(n_episodes, (uint64_t)-1);
    for (uint64_t ep = 0; ep < n_episodes; ++ep)
        for (uint64_t k = 0; k < N; ++k) {        // solve.rs:84-98: `if next_val.0 > best.0` on (success, total) tuples
            const Att &a = at[ep * N + k];
            if (a.success > best_s[ep] || (a.success == best_s[ep] && a.total > best_r[ep])) { best_s[ep] = a.success; best_r[ep] = a.total; best_att[ep] = ep * N + k; }
        }
    if (best_actions && n_episodes == 1 && best_att[0] != (uint64_t)-1) *best_actions = at[best_att[0]].actions;
    cleanup();
    return TW_OK;
}

}  // namespace

extern "C" int tw_evaluate_env(const tw_env_vtable *env, const tw_policy *policy, const tw_solve_params *prm, uint64_t num_episodes,
                               uint64_t episode_offset, uint32_t max_steps, float *success_rate, float *mean_reward)
{
    if (!env || !policy || !prm || !success_rate || !mean_reward) { set_error("tw_evaluate_env: null argument"); return TW_ERR_INVALID; }
    if (num_episodes == 0) { *success_rate = __builtin_nanf(""); *mean_reward = __builtin_nanf(""); return TW_OK; }   // 0/0 (evaluate.rs:52)
    std::vector<float> bs, br;
    const int rc = run_attempts_env(env, policy, prm, num_episodes, episode_offset, false, max_steps ? max_steps : 1u, bs, br, nullptr);
    if (rc) return rc;
    float successes = 0.0f, rewards = 0.0f;       // serial accumulation, episode order (evaluate.rs:36-52)
    for (uint64_t e = 0; e < num_episodes; ++e) { successes = successes + bs[e]; rewards = rewards + br[e]; }
    *success_rate = successes / (float)num_episodes;
    *mean_reward = rewards / (float)num_episodes;
    return TW_OK;
}

extern "C" int tw_solve_env32(const tw_env_vtable *env, const tw_policy *policy, const tw_solve_params *prm, uint32_t max_steps,
                              float *success, float *reward, uint32_t *solution_out, uint32_t solution_cap, uint32_t *n_solution)
{
    if (!env || !policy || !prm || !success || !reward) { set_error("tw_solve_env: null argument"); return TW_ERR_INVALID; }
    std::vector<float> bs, br; std::vector<uint32_t> acts;
    const int rc = run_attempts_env(env, policy, prm, 1, 0, true, max_steps ? max_steps : 1u, bs, br, &acts);
    if (rc) return rc;
    *success = bs[0]; *reward = br[0];
    if (n_solution) *n_solution = (uint32_t)acts.size();
    if (solution_out) {
        if (acts.size() > solution_cap) { set_error("tw_solve_env: %zu entries, caller's buffer holds %u", acts.size(), solution_cap); return TW_ERR_INVALID; }
        if (!acts.empty()) memcpy(solution_out, acts.data(), acts.size() * sizeof(uint32_t));
    }
    return TW_OK;
}

extern "C" int tw_solve_env(const tw_env_vtable *env, const tw_policy *policy, const tw_solve_params *prm, uint32_t max_steps,
                            float *success, float *reward, uint8_t *actions_out, uint32_t actions_cap, uint32_t *n_actions)
{
    if (!env || !policy || !prm || !success || !reward) { set_error("tw_solve_env: null argument"); return TW_ERR_INVALID; }
    std::vector<float> bs, br; std::vector<uint32_t> acts;
    const int rc = run_attempts_env(env, policy, prm, 1, 0, true, max_steps ? max_steps : 1u, bs, br, &acts);
    if (rc) return rc;
    *success = bs[0]; *reward = br[0];
    if (n_actions) *n_actions = (uint32_t)acts.size();
    if (actions_out) {
        if (acts.size() > actions_cap) { set_error("tw_solve_env: %zu actions, caller's buffer holds %u", acts.size(), actions_cap); return TW_ERR_INVALID; }
        for (size_t i = 0; i < acts.size(); ++i) {
            if (acts[i] > 255u) { set_error("tw_solve_env: entry %zu of the tracked solution is %u; use tw_solve_env32", i, acts[i]); return TW_ERR_INVALID; }
            actions_out[i] = (uint8_t)acts[i];
        }
    }
    return TW_OK;
}
