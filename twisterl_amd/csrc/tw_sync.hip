// tw_sync.hip -- device-to-device policy sync.  The reference re-exports every weight
// through `.cpu().numpy().tolist()` and re-boxes it in Rust on every iteration (src/twisterl/nn/policy.py:191-199,
// nn/utils.py:17-79, rl/algorithm.py:90-93).  Here the trainer's parameters (torch layout: Linear.weight = [out][in])
// are read where they live and every image the kernels consume is rebuilt on the device by ONE launch: thread i of
// segment s computes the source element of output element i (the same index maps tw_policy_create applies on the host).
#include "tw_common.hpp"

namespace tw {

__device__ inline int hid_row_d(int r, int i) { return 32 * r + 2 * ((i & 3) + 4 * (i >> 3)) + ((i >> 2) & 1); }
__device__ inline int rho_d(int r, int hh) { return 8 * (r >> 2) + 4 * hh + (r & 3); }

__global__ void __launch_bounds__(256) policy_sync_kernel(const SyncArgs a)
{
    const unsigned long long gid = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= a.seg_end[17]) return;
    int seg = 0;
    while (gid >= a.seg_end[seg]) ++seg;
    const unsigned long long i = gid - (seg ? a.seg_end[seg - 1] : 0ull);
    const int OS = a.OS, E = a.E, H = a.H, A = a.A, NT = a.NT, NQ = a.NQ;
    auto T  = [&](int id, int k) { return a.emb_w[(size_t)k * OS + id]; };        // table[id][k] = Linear.weight[k][id]
    auto W1 = [&](int k, int n) { return a.w1[(size_t)n * E + k]; };              // W1[k][n]    = Linear.weight[n][k]
    auto WA = [&](int n, int o) { return a.wa[(size_t)o * H + n]; };
    auto WV = [&](int n) { return a.wv[n]; };
    switch (seg) {
    case 0: {   // emb_rows [(OS+2)][E]: rows, bias row, zero row
        const int r = (int)(i / E), k = (int)(i % E);
        a.emb_rows[i] = r < OS ? T(r, k) : (r == OS ? a.emb_b[k] : 0.0f);
    } break;
    case 1: {   // w1p [E][NQ][32][4]
        const int cc = (int)(i & 3), li = (int)((i >> 2) & 31), q = (int)((i >> 7) % NQ), k = (int)((i >> 7) / NQ);
        const int r = 4 * q + cc;
        a.w1p[i] = r < NT ? W1(k, hid_row_d(r, li)) : 0.0f;
    } break;
    case 2: {   // t_img16 [E/16][21*256]: row r (20 floats), position pp
        const int slot = 21 * 256;
        const int ch = (int)(i / slot), o = (int)(i % slot), r = o / 20, pp = o % 20;
        float v = 0.0f;
        if (r < OS + 2 && pp < 16) {
            const int k = ch * 16 + (pp < 8 ? 2 * pp : 2 * (pp - 8) + 1);
            v = r < OS ? T(r, k) : (r == OS ? a.emb_b[k] : 0.0f);
        }
        a.t_img16[i] = v;
    } break;
    case 3: a.b1_d[i] = a.b1[i]; break;
    case 4: {   // wh8 [H][8]
        const int n = (int)(i >> 3), c = (int)(i & 7);
        a.wh8[i] = (A <= 4) ? (c < A ? WA(n, c) : (c == 4 ? WV(n) : 0.0f)) : 0.0f;
    } break;
    case 5: a.bh8[i] = (A <= 4) ? ((int)i < A ? a.ba[i] : (i == 4 ? a.bv[0] : 0.0f)) : 0.0f; break;
    case 6: { const int k = (int)(i / H), n = (int)(i % H); a.w1_nat[i] = W1(k, n); } break;
    case 7: { const int n = (int)(i / A), o = (int)(i % A); a.wa_nat[i] = WA(n, o); } break;
    case 8: a.ba_nat[i] = a.ba[i]; break;
    case 9: a.wv_nat[i] = WV((int)i); break;
    case 10: a.bv_nat[i] = a.bv[0]; break;
    case 11: {  // stage16 [NKT][SP16 KiB] of f16: element index within a stage = piece*512 + lane*8 + jx
        const int per = a.SP16 * 512;
        const int kt = (int)(i / per), o = (int)(i % per), piece = o >> 9, l = (o >> 3) & 63, jx = o & 7, hh = l >> 5, row = l & 31;
        float v = 0.0f;
        if (piece < a.nc16) {
            const int val = 8 * hh + jx, kte = (kt + 1) % a.NKT;
            if (piece < a.n16 && val < a.n16) v = T(piece * a.n16 + val, 32 * kte + row);
        } else if (piece < a.nc16 + 2 * NT) {
            const int q = piece - a.nc16, ht = q >> 1, m = q & 1;
            v = W1(32 * kt + rho_d(8 * m + jx, hh), 32 * ht + row);
        }
        reinterpret_cast<_Float16 *>(a.stage16)[i] = (_Float16)v;
    } break;
    case 12: {  // head16 [NT][2][64][8] f16
        const int g = (int)(i >> 9), l = (int)((i >> 3) & 63), jx = (int)(i & 7), hh = l >> 5, row = l & 31, ht = g >> 1, m = g & 1;
        const int hid = 32 * ht + rho_d(8 * m + jx, hh);
        float v = 0.0f;
        if (row < 8) { if ((row & 3) < A) v = WA(hid, row & 3); }
        else if (row == 8 || row == 12) v = WV(hid);
        reinterpret_cast<_Float16 *>(a.head16)[i] = (_Float16)v;
    } break;
    case 13: { const int kt = (int)(i >> 5), hh = (int)((i >> 4) & 1), r = (int)(i & 15); a.ebias16[i] = a.emb_b[32 * kt + rho_d(r, hh)]; } break;
    case 14: { const int ht = (int)(i >> 5), hh = (int)((i >> 4) & 1), r = (int)(i & 15); a.b1img16[i] = a.b1[32 * ht + rho_d(r, hh)]; } break;
    case 15: a.bh16[i] = (int)i < A ? a.ba[i] : (i == 4 ? a.bv[0] : 0.0f); break;
    case 16: case 17: {   // split-f16 images: x16, hi = f16(16x), lo = f16(16x - hi)
        float v = 0.0f; bool want_lo;
        if (seg == 16) {  // stageS [2*NKT+1][SPS KiB]
            const int per = a.SPS * 512;
            const int st = (int)(i / per), o = (int)(i % per), piece = o >> 9, l = (o >> 3) & 63, jx = o & 7, hh = l >> 5, row = l & 31;
            if (st < 2 * a.NKT) {
                const int kt = st >> 1;
                want_lo = st & 1;
                if (piece < a.nc16) {
                    const int val = 8 * hh + jx, kte = (kt + 1) % a.NKT;
                    if (piece < a.n16 && val < a.n16) v = T(piece * a.n16 + val, 32 * kte + row);
                } else if (piece < a.nc16 + 2 * NT) {
                    const int q = piece - a.nc16, ht = q >> 1, m = q & 1;
                    v = W1(32 * kt + rho_d(8 * m + jx, hh), 32 * ht + row);
                }
            } else {      // heads stage: [hi chunks | lo chunks]
                want_lo = piece >= 2 * NT;
                const int g = want_lo ? piece - 2 * NT : piece;
                if (g < 2 * NT) {
                    const int ht = g >> 1, m = g & 1, hid = 32 * ht + rho_d(8 * m + jx, hh);
                    if (row < 8) { if ((row & 3) < A) v = WA(hid, row & 3); }
                    else if (row == 8 || row == 12) v = WV(hid);
                }
            }
            const float sx = 16.0f * v;
            const _Float16 hi = (_Float16)sx;
            reinterpret_cast<_Float16 *>(a.stageS)[i] = want_lo ? (_Float16)(sx - (float)hi) : hi;
        } else {          // t0S [2*nc16 KiB]: table tile 0, hi chunks then lo chunks
            const int piece = (int)(i >> 9), l = (int)((i >> 3) & 63), jx = (int)(i & 7), hh = l >> 5, row = l & 31;
            want_lo = piece >= a.nc16;
            const int c2 = want_lo ? piece - a.nc16 : piece, val = 8 * hh + jx;
            if (c2 < a.n16 && val < a.n16) v = T(c2 * a.n16 + val, row);
            const float sx = 16.0f * v;
            const _Float16 hi = (_Float16)sx;
            reinterpret_cast<_Float16 *>(a.t0S)[i] = want_lo ? (_Float16)(sx - (float)hi) : hi;
        }
    } break;
    }
}

// policies of any Sequential depth: the embedding rows and, per layer, the natural weights (policy_eval kernels), the padded bias
// and the matrix-core image (EngineV) -- the index maps of create_generic_policy (tw_api.hip)
__global__ void __launch_bounds__(256) policy_sync_generic_kernel(const GenSyncArgs a)
{
    const unsigned long long gid = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const int n_seg = 1 + 3 * a.n_layers;
    if (gid >= a.seg_end[n_seg - 1]) return;
    int seg = 0;
    while (gid >= a.seg_end[seg]) ++seg;
    const unsigned long long i = gid - (seg ? a.seg_end[seg - 1] : 0ull);
    if (seg == 0) {   // emb_rows [(OS+2)][E]: rows, bias row, zero row; table[id][k] = Linear.weight[k][id]
        const int r = (int)(i / a.E), k = (int)(i % a.E);
        a.emb_rows[i] = r < a.OS ? a.emb_w[(size_t)k * a.OS + r] : (r == a.OS ? a.emb_b[k] : 0.0f);
        return;
    }
    const int l = (seg - 1) / 3, what = (seg - 1) % 3;
    const int in = a.in[l], out = a.out[l];
    if (what == 0) {          // natural [in][outp]: W[k][o] = Linear.weight[o][k], 0 in the padding columns
        const int k = (int)(i / a.outp[l]), o = (int)(i % a.outp[l]);
        a.w_nat[l][i] = o < out ? a.w[l][(size_t)o * in + k] : 0.0f;
    } else if (what == 1) {   // bias image [nb * tb * 16]
        a.b_img[l][i] = (int)i < out ? a.b[l][i] : 0.0f;
    } else {                  // matrix-core image [kg * 4][nb][16][tb]: element (k, b, r, t) = W[k][(b * tb + t) * 16 + r], -0.0 outside
        const int tb = a.tb[l], nb = a.nb[l];
        const int t = (int)(i % tb), r = (int)((i / tb) % 16), b = (int)((i / (tb * 16)) % nb), k = (int)(i / ((unsigned long long)tb * 16 * nb));
        const int o = (b * tb + t) * 16 + r;
        a.wm[l][i] = (k < in && o < out) ? a.w[l][(size_t)o * in + k] : -0.0f;
    }
}

int launch_policy_sync_generic(const GenSyncArgs &a, hipStream_t s)
{
    const unsigned long long n = a.seg_end[3 * a.n_layers];
    hipLaunchKernelGGL(policy_sync_generic_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    TW_HIP(hipGetLastError());
    return TW_OK;
}

int launch_policy_sync(const SyncArgs &a, hipStream_t s)
{
    const unsigned long long n = a.seg_end[17];
    hipLaunchKernelGGL(policy_sync_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    TW_HIP(hipGetLastError());
    return TW_OK;
}

}  // namespace tw
