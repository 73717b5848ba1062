/*
 * collect_from_c.c -- the collector driven from a compiled host over the C ABI alone (no Python, no torch): what a
 * Rust `impl Collector` does, written in C because this image has no Rust toolchain.
 *
 *   gcc -O2 -I include examples/collect_from_c.c -L twisterl_amd/lib -ltwisterl_hip -Wl,-rpath,'$ORIGIN/../twisterl_amd/lib' -lm
 *   ./collect_from_c [episodes] [seed]
 *
 * Builds a small BasicPolicy-shaped network (81 -> 64 -> 64 -> 4|1, weights from a fixed LCG), collects Puzzle-8 PPO
 * episodes (reference: PPOCollector::collect, rust/src/collector/ppo.rs:108-126) and prints record count and FNV-1a
 * checksums of every field; tests/test_gpu_parity.py compares them with the same collect made through the Python mirror.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "twisterl_hip.h"

static uint64_t lcg_state = 0x9e3779b97f4a7c15ull;
static float lcg_uniform(float bound)           /* U(-bound, bound), 24 random bits */
{
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    const uint32_t r = (uint32_t)(lcg_state >> 40);
    return ((float)r / 16777216.0f * 2.0f - 1.0f) * bound;
}

static float *filled(size_t n, float bound)
{
    float *p = (float *)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) p[i] = lcg_uniform(bound);
    return p;
}

static uint64_t fnv1a(const void *data, size_t n)
{
    const uint8_t *b = (const uint8_t *)data;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}

#define CHECK(call) do { int rc_ = (call); if (rc_ != TW_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, tw_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    const uint64_t episodes = argc > 1 ? strtoull(argv[1], NULL, 10) : 500;
    const uint64_t seed     = argc > 2 ? strtoull(argv[2], NULL, 10) : 7;
    enum { OBS = 81, EMB = 64, HID = 64, ACT = 4 };

    if (tw_abi_version() != TW_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    if (tw_device_count() < 1) { fprintf(stderr, "no GPU: %s\n", tw_last_error()); return 2; }

    /* weights in the reference's export layout (src/twisterl/nn/utils.py:17-79): [in][out] */
    float *emb = filled((size_t)OBS * EMB, 0.111f), *emb_b = filled(EMB, 0.111f);
    float *w1 = filled((size_t)EMB * HID, 0.125f), *b1 = filled(HID, 0.125f);
    float *wa = filled((size_t)HID * ACT, 0.125f), *ba = filled(ACT, 0.125f);
    float *wv = filled(HID, 0.125f), *bv = filled(1, 0.125f);
    tw_linear_desc common = {EMB, HID, w1, b1, 1}, action = {HID, ACT, wa, ba, 0}, value = {HID, 1, wv, bv, 0};
    tw_policy_desc pd;
    memset(&pd, 0, sizeof(pd));
    pd.obs_size = OBS; pd.emb_size = EMB; pd.emb_vectors = emb; pd.emb_bias = emb_b; pd.emb_apply_relu = 1;
    pd.n_common = 1; pd.common = &common; pd.n_action = 1; pd.action = &action; pd.n_value = 1; pd.value = &value;
    pd.n_perms = 0; pd.n_actions = ACT;
    tw_policy *pol = tw_policy_create(&pd);
    if (!pol) { fprintf(stderr, "tw_policy_create: %s\n", tw_last_error()); return 1; }

    tw_puzzle *env = tw_puzzle_create(3, 3, 6, 2, 256);
    tw_puzzle_desc desc;
    CHECK(tw_puzzle_get_desc(env, &desc));

    tw_ppo_params prm;
    memset(&prm, 0, sizeof(prm));
    prm.num_episodes = episodes; prm.episode_offset = 0; prm.gamma = 0.995f; prm.lambda = 0.995f;
    prm.seed = seed; prm.precision = TW_PREC_F32_EXACT; prm.merge_order = 1;
    tw_collected *data = NULL;
    CHECK(tw_ppo_collect(&desc, pol, &prm, &data));

    const uint64_t n = tw_collected_num_records(data);
    printf("records %llu episodes %llu cells %u\n", (unsigned long long)n, (unsigned long long)tw_collected_num_episodes(data),
           tw_collected_num_cells(data));
    static const struct { int field; const char *name; } fields[] = {
        {TW_F_OBS, "obs"}, {TW_F_LOGITS, "logits"}, {TW_F_PERMS, "perms"}, {TW_F_VALUES, "values"}, {TW_F_REWARDS, "rewards"},
        {TW_F_ACTIONS, "actions"}, {TW_F_ADVS, "advs"}, {TW_F_RETS, "rets"}, {TW_F_EP_LEN, "ep_len"}};
    for (size_t i = 0; i < sizeof(fields) / sizeof(fields[0]); ++i) {
        size_t bytes = 0;
        if (!tw_collected_device_ptr(data, fields[i].field, &bytes) || bytes == 0) continue;
        void *host = malloc(bytes);
        CHECK(tw_collected_copy_to_host(data, fields[i].field, host, bytes));
        printf("%s %zu %016llx\n", fields[i].name, bytes, (unsigned long long)fnv1a(host, bytes));
        free(host);
    }
    tw_collect_stats st;
    CHECK(tw_collected_stats(data, &st));
    printf("rollout_ms %.3f blocks %u threads %u\n", st.ms_rollout, st.rollout_blocks, st.rollout_threads);

    tw_collected_free(data);
    tw_puzzle_destroy(env);
    tw_policy_destroy(pol);
    free(emb); free(emb_b); free(w1); free(b1); free(wa); free(ba); free(wv); free(bv);
    return 0;
}
