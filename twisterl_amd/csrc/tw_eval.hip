// tw_eval.hip -- batched Policy.{forward,predict,full_predict} on the device (gfx950).
//
// Serves the PyO3 Policy methods (reference rust/src/python_interface/policy.rs:33-45 over
// rust/src/nn/policy.rs:34-126) for small batches: one workgroup per observation, every
// Linear output evaluated by one thread as the k-ordered fma chain from 0 with the bias added
// last -- the same arithmetic as the MFMA rollout kernel, computed on the VALU, so the two
// kernels cross-check each other bit for bit.  Not a throughput path.
#include "tw_common.hpp"

namespace tw {

constexpr int EVAL_THREADS = 256;
constexpr int EVAL_MAX_ACT = 32;

__global__ void __launch_bounds__(EVAL_THREADS) policy_eval_kernel(const PolicyDev pol, int mode, const int32_t *obs,
                                                                   uint32_t n_obs, const uint8_t *masks,
                                                                   const int32_t *perms, float *out_actions,
                                                                   float *out_values)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *h0 = sm;                       // [emb]
    float *h1 = h0 + pol.emb;             // [hidden]
    float *hl = h1 + pol.hidden;          // [EVAL_MAX_ACT] raw logits of this pass
    float *la = hl + EVAL_MAX_ACT;        // [EVAL_MAX_ACT] accumulated logits
    float *vs = la + EVAL_MAX_ACT;        // [0] value of this pass, [1] accumulated value

    const int tid = threadIdx.x;
    const uint32_t sample = blockIdx.x;
    const int A = pol.n_actions;
    const bool full = (mode == TW_EVAL_FULL_PREDICT) && pol.n_perms > 0;   // policy.rs:103
    const int n_pass = full ? pol.n_perms : 1;
    const float np = (float)pol.n_perms;
    const int bias_row = pol.obs_size;

    if (tid < EVAL_MAX_ACT) la[tid] = 0.0f;
    if (tid == 0) vs[1] = 0.0f;

    for (int pass = 0; pass < n_pass; ++pass) {
        int perm = -1;
        if (full) perm = pass;
        else if (mode != TW_EVAL_FULL_PREDICT && perms != nullptr) perm = perms[sample];
        if (perm >= pol.n_perms) perm = -1;
        __syncthreads();
        // EmbeddingBag (layers.rs:56-62,82-84) on the (twisted) ids (policy.rs:81-83)
        for (int k = tid; k < pol.emb; k += EVAL_THREADS) {
            float v = pol.emb_rows[(size_t)bias_row * pol.emb + k];
            for (uint32_t i = 0; i < n_obs; ++i) {
                int id = obs[(size_t)sample * n_obs + i];
                if (perm >= 0) id = pol.obs_perms[perm * pol.obs_size + id];
                v = v + pol.emb_rows[(size_t)id * pol.emb + k];
            }
            if (pol.emb_relu) v = v > 0.0f ? v : 0.0f;
            h0[k] = v;
        }
        __syncthreads();
        // common Linear (+ReLU)
        for (int o = tid; o < pol.hidden; o += EVAL_THREADS) {
            float acc = 0.0f;
            for (int k = 0; k < pol.emb; ++k) acc = __builtin_fmaf(pol.w1[(size_t)k * pol.hidden + o], h0[k], acc);
            acc = acc + pol.b1[o];
            if (pol.common_relu) acc = acc > 0.0f ? acc : 0.0f;
            h1[o] = acc;
        }
        __syncthreads();
        // heads (policy.rs:89-92)
        if (tid <= A) {
            float acc = 0.0f;
            if (tid < A) {
                for (int k = 0; k < pol.hidden; ++k) acc = __builtin_fmaf(pol.wa[(size_t)k * A + tid], h1[k], acc);
                hl[tid] = acc + pol.ba[tid];
            } else {
                for (int k = 0; k < pol.hidden; ++k) acc = __builtin_fmaf(pol.wv[k], h1[k], acc);
                vs[0] = acc + pol.bv[0];
            }
        }
        __syncthreads();
        if (tid < A) {
            const int src = perm >= 0 ? (int)pol.act_perms[perm * A + tid] : tid;   // policy.rs:95-97
            const float l = hl[src];
            if (full) la[tid] = la[tid] + l / np;                                    // policy.rs:112-114
            else la[tid] = l;
        }
        if (tid == 0) {
            if (full) vs[1] = vs[1] + vs[0] / np;                                    // policy.rs:111
            else vs[1] = vs[0];
        }
    }
    __syncthreads();
    if (tid == 0) {
        const uint8_t *m = masks + (size_t)sample * A;
        float *o = out_actions + (size_t)sample * A;
        if (mode == TW_EVAL_FORWARD) {
            for (int i = 0; i < A; ++i) o[i] = m[i] ? la[i] : -1e10f;                // policy.rs:62
        } else {
            float sum = 0.0f;                                                        // policy.rs:43-47,118-124
            for (int i = 0; i < A; ++i) { o[i] = m[i] ? tw_expf(la[i]) : 0.0f; }
            for (int i = 0; i < A; ++i) sum = sum + o[i];
            for (int i = 0; i < A; ++i) o[i] = o[i] / (sum + 0.000001f);
        }
        out_values[sample] = vs[1];
    }
}

// The same for a generic policy (any Sequential depth, PolicyDev::layers): one workgroup per observation, one thread per
// output of a layer, activations ping-pong through three LDS buffers (the common output stays while the two heads run).
constexpr int EVAL_GEN_W = 512;

__global__ void __launch_bounds__(EVAL_THREADS) policy_eval_generic_kernel(const PolicyDev pol, int mode, const int32_t *obs, uint32_t n_obs,
                                                                           const uint8_t *masks, const int32_t *perms, float *out_actions,
                                                                           float *out_values)
{
    __shared__ float bufs[3][EVAL_GEN_W];
    __shared__ float la[EVAL_MAX_ACT], vs[2];
    const int tid = threadIdx.x;
    const uint32_t sample = blockIdx.x;
    const int A = pol.n_actions;
    const bool full = (mode == TW_EVAL_FULL_PREDICT) && pol.n_perms > 0;
    const int n_pass = full ? pol.n_perms : 1;
    const float np = (float)pol.n_perms;
    if (tid < EVAL_MAX_ACT) la[tid] = 0.0f;
    if (tid == 0) vs[1] = 0.0f;
    auto run_stack = [&](const LayerDev *ls, int n, int src, int keep) -> int {
        int cur = src;
        for (int l = 0; l < n; ++l) {
            int dst = 0;
            while (dst == cur || dst == keep) ++dst;
            const LayerDev L = ls[l];
            for (int o = tid; o < L.out; o += EVAL_THREADS) {
                float acc = 0.0f;
                for (int k = 0; k < L.in; ++k) acc = __builtin_fmaf(L.w[(size_t)k * L.out + o], bufs[cur][k], acc);
                acc = acc + L.b[o];
                bufs[dst][o] = L.relu ? (acc > 0.0f ? acc : 0.0f) : acc;
            }
            __syncthreads();
            cur = dst;
        }
        return cur;
    };
    for (int pass = 0; pass < n_pass; ++pass) {
        int perm = -1;
        if (full) perm = pass;
        else if (mode != TW_EVAL_FULL_PREDICT && perms != nullptr) perm = perms[sample];
        if (perm >= pol.n_perms) perm = -1;
        __syncthreads();
        for (int k = tid; k < pol.emb; k += EVAL_THREADS) {            // EmbeddingBag (layers.rs:56-62,82-84)
            float v = pol.emb_rows[(size_t)pol.obs_size * pol.emb + k];
            for (uint32_t i = 0; i < n_obs; ++i) {
                int id = obs[(size_t)sample * n_obs + i];
                if (perm >= 0) id = pol.obs_perms16 ? (int)pol.obs_perms16[(size_t)perm * pol.obs_size + id] : (int)pol.obs_perms[perm * pol.obs_size + id];
                v = v + pol.emb_rows[(size_t)id * pol.emb + k];
            }
            if (pol.emb_relu) v = v > 0.0f ? v : 0.0f;
            bufs[0][k] = v;
        }
        __syncthreads();
        const int co = run_stack(pol.layers, pol.n_common, 0, -1);                                          // policy.rs:86
        const int vo = run_stack(pol.layers + pol.n_common + pol.n_action, pol.n_value, co, co);            // policy.rs:89
        if (tid == 0) {
            float s = 0.0f;
            for (int i = 0; i < pol.value_out; ++i) s = s + bufs[vo][i];
            vs[0] = s;
        }
        __syncthreads();
        const int ao = run_stack(pol.layers + pol.n_common, pol.n_action, co, co);                          // policy.rs:92
        if (tid < A) {
            const int src = perm >= 0 ? (int)pol.act_perms[perm * A + tid] : tid;                           // policy.rs:95-97
            const float l = bufs[ao][src];
            if (full) la[tid] = la[tid] + l / np; else la[tid] = l;
        }
        if (tid == 0) { if (full) vs[1] = vs[1] + vs[0] / np; else vs[1] = vs[0]; }
    }
    __syncthreads();
    if (tid == 0) {
        const uint8_t *m = masks + (size_t)sample * A;
        float *o = out_actions + (size_t)sample * A;
        if (mode == TW_EVAL_FORWARD) {
            for (int i = 0; i < A; ++i) o[i] = m[i] ? la[i] : -1e10f;
        } else {
            float sum = 0.0f;
            for (int i = 0; i < A; ++i) { o[i] = m[i] ? tw_expf(la[i]) : 0.0f; }
            for (int i = 0; i < A; ++i) sum = sum + o[i];
            for (int i = 0; i < A; ++i) o[i] = o[i] / (sum + 0.000001f);
        }
        out_values[sample] = vs[1];
    }
}

int launch_policy_eval(const PolicyDev &pol, int mode, const int32_t *obs_d, uint32_t n, uint32_t n_obs,
                       const uint8_t *masks_d, const int32_t *perms_d, float *out_actions_d, float *out_values_d,
                       hipStream_t s)
{
    if (n == 0) return TW_OK;
    if (pol.n_actions > EVAL_MAX_ACT - 1) { set_error("evaluate: n_actions %d > %d", pol.n_actions, EVAL_MAX_ACT - 1); return TW_ERR_UNSUPPORTED; }
    if (pol.generic) {
        hipLaunchKernelGGL(policy_eval_generic_kernel, dim3(n), dim3(EVAL_THREADS), 0, s, pol, mode, obs_d, n_obs, masks_d, perms_d, out_actions_d, out_values_d);
        TW_HIP(hipGetLastError());
        return TW_OK;
    }
    const size_t lds = (size_t)(pol.emb + pol.hidden + 2 * EVAL_MAX_ACT + 4) * sizeof(float);
    hipLaunchKernelGGL(policy_eval_kernel, dim3(n), dim3(EVAL_THREADS), lds, s, pol, mode, obs_d, n_obs, masks_d, perms_d,
                       out_actions_d, out_values_d);
    TW_HIP(hipGetLastError());
    return TW_OK;
}

}  // namespace tw
