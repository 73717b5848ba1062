// tw_trainer.hip -- trainer hand-off kernels: what PPO.data_to_torch / AZ.data_to_torch
// (reference src/twisterl/rl/ppo.py:25-61, rl/az.py:28-46) build in Python lists, written straight into
// caller-provided device buffers (torch tensors):
//   * dense one-hot observations  np_obs[i, obs_i] = 1.0            (ppo.py:37-39)   HBM-write bound
//   * old log-probs  Categorical(logits).log_prob(actions)            (ppo.py:57-59)   = l[a] - logsumexp(l)
//   * actions / twist indices widened to int64                        (ppo.py:47,50-52)
//   * advantages, optionally (a - mean) / (std + 1e-8), std unbiased  (ppo.py:55-56)
// All kernels take a row range so a large collect can be handed over in mini-batches.
#include "tw_common.hpp"

namespace tw {

// one wave per row: lane l writes columns 4l..4l+3 (+256 per pass) as ONE 16-byte store
__global__ void __launch_bounds__(256) onehot_kernel(const uint8_t *obs, uint64_t row0, uint64_t rows, int n_cells, int obs_size,
                                                     float *out)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    const int n2 = obs_size / n_cells;              // ids of cell k live in [k*n2, (k+1)*n2)
    for (uint64_t r = wave; r < rows; r += n_waves) {
        const uint8_t *o = obs + (row0 + r) * (uint64_t)n_cells;
        float *dst = out + r * (uint64_t)obs_size;
        for (int c0 = lane * 4; c0 < obs_size; c0 += 256) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int col = c0 + u;
                v[u] = (col < obs_size && (int)o[col / n2] == col) ? 1.0f : 0.0f;
            }
            if (c0 + 3 < obs_size && (obs_size & 3) == 0) *reinterpret_cast<float4 *>(dst + c0) = make_float4(v[0], v[1], v[2], v[3]);
            else
                for (int u = 0; u < 4; ++u) if (c0 + u < obs_size) dst[c0 + u] = v[u];
        }
    }
}

// The usual case (Puzzle-15: 16 cells x 16 ids): a lane's four columns belong to ONE cell (n2 and obs_size multiples of four).  A wave
// takes ROWS rows at a time: every lane reads its cell's id byte of all of them first -- independent loads, one round trip --, then writes
// ROWS 16-byte pieces; the wave's stores cover whole rows (1 KB each).  (The one-row-at-a-time form above waits one load round trip per
// KB written: 9.5 ms for 30 M Puzzle-15 rows, 3.3 TB/s; this one 5.7 ms = 5.4 TB/s, the whole hand-off 6.3 ms -- scripts/bench_trainer_pack.py.)
template <int ROWS>
__global__ void __launch_bounds__(256) onehot4_kernel(const uint8_t *obs, uint64_t row0, uint64_t rows, int n_cells, int obs_size, float *out)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    const int n2 = obs_size / n_cells;
    for (int c0 = lane * 4; c0 < obs_size; c0 += 256) {                 // (one pass up to 256 ids)
        const int cell = c0 / n2;
        for (uint64_t r = wave * ROWS; r < rows; r += n_waves * ROWS) {
            int d[ROWS];
#pragma unroll
            for (int i = 0; i < ROWS; ++i) d[i] = r + i < rows ? (int)obs[(row0 + r + i) * (uint64_t)n_cells + cell] - c0 : -1;
#pragma unroll
            for (int i = 0; i < ROWS; ++i)
                if (r + i < rows)
                    *reinterpret_cast<float4 *>(out + (r + i) * (uint64_t)obs_size + c0) =
                        make_float4(d[i] == 0 ? 1.0f : 0.0f, d[i] == 1 ? 1.0f : 0.0f, d[i] == 2 ? 1.0f : 0.0f, d[i] == 3 ? 1.0f : 0.0f);
        }
    }
}

// generic fallback when obs ids are not "cell k owns [k*n2,(k+1)*n2)" (never the case for Puzzle): zero, then scatter
__global__ void __launch_bounds__(256) onehot_scatter_kernel(const uint8_t *obs, uint64_t row0, uint64_t rows, int n_cells, int obs_size, float *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * (uint64_t)n_cells) return;
    const uint64_t r = i / n_cells;
    out[r * (uint64_t)obs_size + obs[row0 * (uint64_t)n_cells + i]] = 1.0f;
}

// thread per row: log-prob of the taken action, int64 widening, optional advantage normalisation
__global__ void __launch_bounds__(256) ppo_pack_kernel(const float *logits, const uint8_t *actions, const int8_t *perms, const float *advs,
                                                       uint64_t row0, uint64_t rows, int n_actions, float mean, float denom, int normalize,
                                                       float *logp_out, int64_t *acts_out, int64_t *perms_out, float *advs_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    const uint64_t r = row0 + i;
    if (logp_out) {
        const float *l = logits + r * (uint64_t)n_actions;
        float m = l[0];
        for (int k = 1; k < n_actions; ++k) m = l[k] > m ? l[k] : m;
        float s = 0.0f;
        for (int k = 0; k < n_actions; ++k) s = s + tw_expf(l[k] - m);
        logp_out[i] = (l[actions[r]] - m) - tw_logf(s);             // log_softmax(l)[a]
    }
    if (acts_out) acts_out[i] = (int64_t)actions[r];
    if (perms_out) perms_out[i] = (int64_t)perms[r];
    if (advs_out) advs_out[i] = normalize ? (advs[r] - mean) / denom : advs[r];
}

// sum and sum of squared deviations in double (two passes: mean first).  Bit-reproducible: every block writes its partial,
// one workgroup adds the partials in block order (no float atomics anywhere in the library).
constexpr unsigned SUM_BLOCKS = 2048;

__global__ void __launch_bounds__(256) sum_kernel(const float *x, uint64_t n, double shift, int squared, double *partials)
{
    __shared__ double sm[4];
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const double d = (double)x[i] - shift;
        acc += squared ? d * d : d;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

__global__ void __launch_bounds__(64) sum_partials_kernel(const double *partials, unsigned n_blocks, double *out)
{
    // lane l adds partials l, l+64, ... in order; then a fixed-shape tree over the lanes
    double acc = 0.0;
    for (unsigned i = threadIdx.x; i < n_blocks; i += 64) acc += partials[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (threadIdx.x == 0) *out = acc;
}

int launch_onehot(const uint8_t *obs, uint64_t row0, uint64_t rows, int n_cells, int obs_size, float *out, hipStream_t s)
{
    if (rows == 0) return TW_OK;
    if (obs_size % n_cells == 0 && (obs_size / n_cells) % 4 == 0) {
        constexpr int ROWS = 8;
        uint64_t blocks = (rows + 4 * ROWS - 1) / (4 * ROWS);
        if (blocks > 256ull * 32) blocks = 256ull * 32;
        hipLaunchKernelGGL(onehot4_kernel<ROWS>, dim3((unsigned)blocks), dim3(256), 0, s, obs, row0, rows, n_cells, obs_size, out);
    } else if (obs_size % n_cells == 0) {
        uint64_t blocks = (rows + 3) / 4;
        if (blocks > 256ull * 32) blocks = 256ull * 32;
        hipLaunchKernelGGL(onehot_kernel, dim3((unsigned)blocks), dim3(256), 0, s, obs, row0, rows, n_cells, obs_size, out);
    } else {
        TW_HIP(hipMemsetAsync(out, 0, rows * (uint64_t)obs_size * sizeof(float), s));
        const uint64_t n = rows * (uint64_t)n_cells;
        hipLaunchKernelGGL(onehot_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, obs, row0, rows, n_cells, obs_size, out);
    }
    TW_HIP(hipGetLastError());
    return TW_OK;
}

int launch_ppo_pack(const float *logits, const uint8_t *actions, const int8_t *perms, const float *advs, uint64_t row0, uint64_t rows,
                    int n_actions, float mean, float denom, int normalize, float *logp_out, int64_t *acts_out, int64_t *perms_out,
                    float *advs_out, hipStream_t s)
{
    if (rows == 0) return TW_OK;
    hipLaunchKernelGGL(ppo_pack_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, logits, actions, perms, advs, row0, rows,
                       n_actions, mean, denom, normalize, logp_out, acts_out, perms_out, advs_out);
    TW_HIP(hipGetLastError());
    return TW_OK;
}

size_t sum_scratch_doubles() { return SUM_BLOCKS + 1; }

// scratch: sum_scratch_doubles() doubles on the device; the result lands in scratch[0]
int launch_sum(const float *x, uint64_t n, double shift, int squared, double *scratch, hipStream_t s)
{
    TW_HIP(hipMemsetAsync(scratch, 0, sizeof(double), s));
    if (n == 0) return TW_OK;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > SUM_BLOCKS) blocks = SUM_BLOCKS;
    hipLaunchKernelGGL(sum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n, shift, squared, scratch + 1);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, s, scratch + 1, (unsigned)blocks, scratch);
    TW_HIP(hipGetLastError());
    return TW_OK;
}

}  // namespace tw
