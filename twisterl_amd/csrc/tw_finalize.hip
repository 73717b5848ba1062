// tw_finalize.hip -- episode-length scan, GAE and compaction kernels (gfx950).
//
//  * scan: exclusive prefix sum of the per-episode record counts in OUTPUT order.  With
//    merge_order=1 the output order is the reference's merge() order: the LAST episode first,
//    then episodes 0..E-2 (rust/src/collector/collector.rs:40-46).
//  * finalize_ppo: GAE + compaction.  Runs the GAE recurrence exactly as written in
//    rust/src/collector/ppo.rs:82-92 (sequential in t, no contraction: the library is built with
//    -ffp-contract=off) -- one LANE per episode, sixteen (eight) episodes per workgroup; a wave per
//    episode beyond the horizons whose tile fits LDS -- and copies every field of the episode into
//    its compact slot with coalesced loads/stores.  HBM-bound byte moving; algorithmic traffic =
//    N^2+34 B written + one 48-B padded record read per record (each record is read ONCE).
#include "tw_common.hpp"

namespace tw {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS   = 4;                       // episodes per thread
constexpr int SCAN_TILE    = SCAN_THREADS * SCAN_ITEMS;

__device__ inline uint64_t episode_at(uint64_t pos, uint64_t E, int merge_order)
{
    if (!merge_order) return pos;
    return pos == 0 ? E - 1 : pos - 1;
}

__device__ inline uint64_t block_reduce_sum(uint64_t v, uint64_t *smem)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    uint64_t tot = 0;
    for (int w = 0; w < SCAN_THREADS / 64; ++w) tot += smem[w];
    __syncthreads();
    return tot;
}

// phase 1: per-tile sums
__global__ void __launch_bounds__(SCAN_THREADS) scan_tile_sums(const uint32_t *ep_len, uint64_t E, int merge_order,
                                                               uint64_t *tile_sums)
{
    __shared__ uint64_t smem[SCAN_THREADS / 64];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const uint64_t pos = base + i;
        if (pos < E) v += ep_len[episode_at(pos, E, merge_order)];
    }
    const uint64_t tot = block_reduce_sum(v, smem);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// phase 2: exclusive scan of the tile sums by ONE workgroup (n_tiles <= a few thousand)
__global__ void __launch_bounds__(SCAN_THREADS) scan_tile_offsets(uint64_t *tile_sums, uint64_t n_tiles, uint64_t *total)
{
    __shared__ uint64_t smem[SCAN_THREADS];
    __shared__ uint64_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n_tiles; base += SCAN_THREADS) {
        const uint64_t i = base + threadIdx.x;
        const uint64_t v = i < n_tiles ? tile_sums[i] : 0;
        smem[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < SCAN_THREADS; o <<= 1) {          // Hillis-Steele inclusive scan
            const uint64_t add = threadIdx.x >= (unsigned)o ? smem[threadIdx.x - o] : 0;
            __syncthreads();
            smem[threadIdx.x] += add;
            __syncthreads();
        }
        const uint64_t carry = carry_s;
        if (i < n_tiles) tile_sums[i] = carry + smem[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == SCAN_THREADS - 1) carry_s = carry + smem[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

// phase 3: per-episode start offsets
__global__ void __launch_bounds__(SCAN_THREADS) scan_write(const uint32_t *ep_len, uint64_t E, int merge_order,
                                                           const uint64_t *tile_offsets, uint64_t *ep_start)
{
    __shared__ uint64_t smem[SCAN_THREADS];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t l[SCAN_ITEMS]; uint64_t ep[SCAN_ITEMS]; uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const uint64_t pos = base + i;
        ep[i] = pos < E ? episode_at(pos, E, merge_order) : 0;
        l[i]  = pos < E ? ep_len[ep[i]] : 0u;
        v += l[i];
    }
    smem[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < SCAN_THREADS; o <<= 1) {
        const uint64_t add = threadIdx.x >= (unsigned)o ? smem[threadIdx.x - o] : 0;
        __syncthreads();
        smem[threadIdx.x] += add;
        __syncthreads();
    }
    uint64_t run = tile_offsets[blockIdx.x] + smem[threadIdx.x] - v;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < E) ep_start[ep[i]] = run;
        run += l[i];
    }
}

size_t scan_scratch_bytes(uint64_t n_episodes)
{
    const uint64_t n_tiles = (n_episodes + SCAN_TILE - 1) / SCAN_TILE;
    return (size_t)(n_tiles + 1) * sizeof(uint64_t);
}

int launch_scan(const uint32_t *ep_len, uint64_t E, int merge_order, uint64_t *ep_start, uint64_t *total,
                void *scratch, size_t scratch_bytes, hipStream_t s)
{
    const uint64_t n_tiles = (E + SCAN_TILE - 1) / SCAN_TILE;
    if (E == 0 || n_tiles > 0x7fffffffull || scratch_bytes < scan_scratch_bytes(E)) {
        set_error("scan: bad episode count or scratch size");
        return TW_ERR_INVALID;
    }
    uint64_t *tiles = reinterpret_cast<uint64_t *>(scratch);
    hipLaunchKernelGGL(scan_tile_sums, dim3((unsigned)n_tiles), dim3(SCAN_THREADS), 0, s, ep_len, E, merge_order, tiles);
    hipLaunchKernelGGL(scan_tile_offsets, dim3(1), dim3(SCAN_THREADS), 0, s, tiles, n_tiles, total);
    hipLaunchKernelGGL(scan_write, dim3((unsigned)n_tiles), dim3(SCAN_THREADS), 0, s, ep_len, E, merge_order, tiles, ep_start);
    TW_HIP(hipGetLastError());
    return TW_OK;
}

// ---- GAE + compaction ---------------------------------------------------------------------
constexpr int FIN_WAVES = 4;   // episodes per workgroup (one per wave); fewer when the episode tile would not fit 64 KiB of LDS

// bytes of LDS one episode (one wave) needs: [obs staging: 16 B per record, boards below 16 cells only] | rewards | values | advs |
// rets | action/perm word -- rounded to 16 bytes so that the staging of the next wave stays aligned
static __host__ __device__ inline size_t fin_wave_bytes(int t_pad, int n_cells)
{
    return (((size_t)(n_cells == 16 ? 5 : 9) * (size_t)t_pad * 4) + 15) / 16 * 16;
}

// One wave per episode (long horizons; the grouped form below serves the others).  Every 48-byte padded record is FETCHED ONCE: a lane takes a whole record -- its three 16-byte words in three
// loads issued back to back, which cover the same cache lines -- and each word goes where it belongs: word 0 (the board as obs
// bytes) and word 1 (the four logits) straight to their compact rows, word 2 (value, reward, action | twist) into LDS for the
// GAE chain.  (Until round 4 the three words were read in three PASSES over the episode; with more episodes in flight than L2
// holds, each pass fetched every line again: 188 B of HBM traffic per record by counter against 98 algorithmic; now 99.)
__global__ void __launch_bounds__(FIN_WAVES * 64) finalize_ppo_kernel(const PaddedTraj in, const uint64_t *ep_start,
                                                                      uint64_t E, int n_cells, float gamma, float lambda,
                                                                      const CompactTraj out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t fin_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int t_pad = in.t_pad;
    uint8_t *base = fin_lds + (size_t)wave * fin_wave_bytes(t_pad, n_cells);
    uint4 *so = reinterpret_cast<uint4 *>(base);                                            // (n_cells < 16 only)
    float *sr = reinterpret_cast<float *>(base + (n_cells == 16 ? 0 : (size_t)t_pad * 16));    // rewards | values | advs | rets | action/perm word
    float *sv = sr + t_pad, *sa = sv + t_pad, *st = sa + t_pad;
    uint32_t *sz = reinterpret_cast<uint32_t *>(st + t_pad);

    for (uint64_t e = (uint64_t)blockIdx.x * n_waves + wave; e < E; e += (uint64_t)gridDim.x * n_waves) {
        const int      n     = (int)in.ep_len[e];
        const uint64_t src   = e * (uint64_t)t_pad;
        const uint64_t dst   = ep_start[e];
        const uint4 *rec4 = reinterpret_cast<const uint4 *>(in.rec + src);
        uint4 *obs4 = reinterpret_cast<uint4 *>(out.obs), *logits4 = reinterpret_cast<uint4 *>(out.logits);
        for (int t = lane; t < n; t += 64) {
            // the three words of this lane's record, requested back to back: the wave's three loads cover the same 24 cache lines
            // (64 records x 48 B), fetched once and hit in the vector cache twice -- no branch, stores of 16 contiguous bytes per lane
            const uint4 w0 = rec4[3 * t], w1 = rec4[3 * t + 1], w2 = rec4[3 * t + 2];
            sv[t] = __builtin_bit_cast(float, w2.x); sr[t] = __builtin_bit_cast(float, w2.y); sz[t] = w2.z;
            logits4[dst + t] = w1;
            if (n_cells == 16) obs4[dst + t] = w0; else so[t] = w0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // GAE (ppo.rs:82-92): every lane walks the same chain (LDS broadcast reads); lane 0 stores.  A serial chain on purpose:
        // a segmented scan would re-associate the sums (<= 1e-5 of this, not the same bits) for <= 2 ms of a 140 ms step.
        {
            float adv = sr[n - 1] - sv[n - 1];
            float ret = sr[n - 1];
            if (lane == 0) { sa[n - 1] = adv; st[n - 1] = ret; }
            for (int t = n - 2; t >= 0; --t) {
                float inner = lambda * adv;
                inner = sv[t + 1] + inner;
                inner = gamma * inner;
                ret = sr[t] + inner;
                adv = ret - sv[t];
                if (lane == 0) { sa[t] = adv; st[t] = ret; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int t = lane; t < n; t += 64) {
            out.values[dst + t]  = sv[t];
            out.rewards[dst + t] = sr[t];
            out.advs[dst + t]    = sa[t];
            out.rets[dst + t]    = st[t];
            const uint32_t wz = sz[t];
            out.actions[dst + t] = (uint8_t)(wz & 0xffu);
            out.perms[dst + t]   = (int8_t)((wz >> 8) & 0xffu);
        }
        if (n_cells != 16) {                           // compact rows of n_cells bytes out of the staged 16-byte ones
            const uint8_t *sob = reinterpret_cast<const uint8_t *>(so);
            const int nb = n * n_cells;
            for (int i = lane; i < nb; i += 64) {
                const int t = i / n_cells, c = i - t * n_cells;
                out.obs[dst * n_cells + i] = sob[t * 16 + c];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// G episodes per workgroup of four waves (the short-horizon form).  The GAE recurrence is serial in t and must stay so (its sums in the
// reference's order), but episodes are independent: LANE g of one wave walks the chain of episode g, so a workgroup pays one 257-step
// loop for G episodes where the per-wave form above pays one per episode, replicated on 64 lanes (0.63 of its 1.82 ms at the headline
// size were that loop: the same kernel without it ran 1.19 ms).  Phase A: wave w takes episodes w, w + 4, ... -- a lane per record, each
// record fetched once; everything that needs no recurrence goes straight to its compact row, (value, reward) into LDS [episode][t].
// Phase B: the chains, in place (advantage over reward, return over value).  Phase C: wave per episode again, the two columns out.
template <int G>
__global__ void __launch_bounds__(256) finalize_ppo_group_kernel(const PaddedTraj in, const uint64_t *ep_start, uint64_t E, int n_cells,
                                                                  float gamma, float lambda, const CompactTraj out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t fin_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t_pad = in.t_pad, ts = t_pad | 1;                                              // (odd row stride: the chains' lanes spread over the banks)
    float *sv = reinterpret_cast<float *>(fin_lds), *sr = sv + (size_t)G * ts;
    uint4 *so = reinterpret_cast<uint4 *>(fin_lds + (size_t)G * ts * 8);                     // [G][t_pad] obs words (1..15 cells only)
    uint4 *obs4 = reinterpret_cast<uint4 *>(out.obs), *logits4 = reinterpret_cast<uint4 *>(out.logits);
    const uint64_t n_groups = (E + G - 1) / G;
    for (uint64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const uint64_t e0 = grp * G;
        for (int g = wave; g < G && e0 + g < E; g += 4) {
            const uint64_t e = e0 + g;
            const int n = (int)in.ep_len[e];
            const uint64_t dst = ep_start[e];
            const uint4 *rec4 = reinterpret_cast<const uint4 *>(in.rec + e * (uint64_t)t_pad);
            float *v = sv + (size_t)g * ts, *r = sr + (size_t)g * ts;
            // (requesting the loads of THREE such trips before the first store -- more bytes in flight -- measured slower, 1.8 ms against 1.21:
            //  a record's three words share cache lines, and with nine loads per wave in flight the later words miss the vector cache)
            for (int t = lane; t < n; t += 64) {
                const uint4 w0 = rec4[3 * t], w1 = rec4[3 * t + 1], w2 = rec4[3 * t + 2];
                const float val = __builtin_bit_cast(float, w2.x), rew = __builtin_bit_cast(float, w2.y);
                v[t] = val; r[t] = rew;
                out.values[dst + t] = val; out.rewards[dst + t] = rew;
                out.actions[dst + t] = (uint8_t)(w2.z & 0xffu);
                out.perms[dst + t]   = (int8_t)((w2.z >> 8) & 0xffu);
                logits4[dst + t] = w1;
                if (n_cells == 16) obs4[dst + t] = w0; else if (n_cells > 0) so[(size_t)g * t_pad + t] = w0;
            }
        }
        __syncthreads();
        // GAE (ppo.rs:82-92), lane = episode; the wave that runs the chains rotates with the group (the four SIMDs share the work)
        if (wave == (int)(grp & 3u) && lane < G && e0 + lane < E) {
            const int n = (int)in.ep_len[e0 + lane];
            float *v = sv + (size_t)lane * ts, *r = sr + (size_t)lane * ts;
            if (n > 0) {
                float vnext = v[n - 1];
                float ret = r[n - 1];
                float adv = ret - vnext;
                r[n - 1] = adv; v[n - 1] = ret;
                for (int t = n - 2; t >= 0; --t) {
                    const float rt = r[t], vt = v[t];
                    float inner = lambda * adv;
                    inner = vnext + inner;
                    inner = gamma * inner;
                    ret = rt + inner;
                    adv = ret - vt;
                    r[t] = adv; v[t] = ret;
                    vnext = vt;
                }
            }
        }
        __syncthreads();
        for (int g = wave; g < G && e0 + g < E; g += 4) {
            const uint64_t e = e0 + g;
            const int n = (int)in.ep_len[e];
            const uint64_t dst = ep_start[e];
            const float *v = sv + (size_t)g * ts, *r = sr + (size_t)g * ts;
            for (int t = lane; t < n; t += 64) { out.advs[dst + t] = r[t]; out.rets[dst + t] = v[t]; }
            if (n_cells > 0 && n_cells < 16) {         // compact rows of n_cells bytes out of the staged 16-byte ones
                const uint8_t *sob = reinterpret_cast<const uint8_t *>(so + (size_t)g * t_pad);
                const int nb = n * n_cells;
                for (int i = lane; i < nb; i += 64) {
                    const int t = i / n_cells, c = i - t * n_cells;
                    out.obs[dst * n_cells + i] = sob[t * 16 + c];
                }
            }
        }
        __syncthreads();                               // (the next group overwrites the tile)
    }
}

static size_t fin_group_bytes(int g, int t_pad, int n_cells)
{
    return (size_t)g * (size_t)(t_pad | 1) * 8 + ((n_cells == 16 || n_cells == 0) ? 0 : (size_t)g * (size_t)t_pad * 16);     // (0 cells: the obs ids have their own array)
}

template <int G>
static int launch_fin_group(const PaddedTraj &in, const uint64_t *ep_start, uint64_t E, int n_cells, float gamma, float lambda,
                            const CompactTraj &out, hipStream_t s)
{
    const size_t lds_bytes = fin_group_bytes(G, in.t_pad, n_cells);
    uint64_t blocks = (E + G - 1) / G;
    const uint64_t per_cu = (size_t)160 * 1024 / lds_bytes < 8 ? (size_t)160 * 1024 / lds_bytes : 8;
    if (blocks > 256ull * per_cu * 2) blocks = 256ull * per_cu * 2;       // grid-stride the rest
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&finalize_ppo_group_kernel<G>), lds_bytes)) return rc;
    hipLaunchKernelGGL(finalize_ppo_group_kernel<G>, dim3((unsigned)blocks), dim3(256), lds_bytes, s, in, ep_start, E, n_cells, gamma, lambda, out);
    TW_HIP(hipGetLastError());
    return TW_OK;
}

int launch_finalize_ppo(const PaddedTraj &in, const uint64_t *ep_start, uint64_t E, int n_cells, float gamma,
                        float lambda, const CompactTraj &out, hipStream_t s)
{
    if (E == 0) return TW_OK;
    // horizons whose tile of 16 (8) episodes fits 64 KiB of LDS -- 511 (1,023: every horizon the Puzzle takes) records with 16-cell or
    // id-array observations, 170 (341) with staged ones: the grouped form (headline size, 262,144 x 257: 1.21 ms with 16, 1.25 with 8 or 32, against 1.82 per wave;
    // scripts/bench_finalize.py); longer ones: a wave per episode
    if (fin_group_bytes(16, in.t_pad, n_cells) <= 64 * 1024) return launch_fin_group<16>(in, ep_start, E, n_cells, gamma, lambda, out, s);
    if (fin_group_bytes(8, in.t_pad, n_cells) <= 64 * 1024) return launch_fin_group<8>(in, ep_start, E, n_cells, gamma, lambda, out, s);
    const size_t per_wave = fin_wave_bytes(in.t_pad, n_cells);
    int waves = FIN_WAVES;
    while (waves > 1 && waves * per_wave > 64 * 1024) waves >>= 1;           // long horizons: fewer episodes per workgroup
    const size_t lds_bytes = waves * per_wave;
    if (lds_bytes > 64 * 1024) { set_error("finalize: t_pad %d too large for the LDS tile", in.t_pad); return TW_ERR_UNSUPPORTED; }
    uint64_t blocks = (E + waves - 1) / waves;
    if (blocks > 256ull * 16) blocks = 256ull * 16;   // grid-stride the rest
    hipLaunchKernelGGL(finalize_ppo_kernel, dim3((unsigned)blocks), dim3(waves * 64), lds_bytes, s, in, ep_start, E,
                       n_cells, gamma, lambda, out);
    TW_HIP(hipGetLastError());
    return TW_OK;
}

}  // namespace tw
