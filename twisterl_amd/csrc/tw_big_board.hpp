// tw_big_board.hpp -- Puzzle boards of 17 .. 64 cells for the device kernels (tw_rollout_big.hip, tw_mcts_big.hip): 25 x 5 bits in a
// 128-bit integer up to 25 cells, one byte per cell in 9 / 16 registers up to 36 / 64; Env::step on either (puzzle.rs:135-160).
#pragma once
#include "tw_common.hpp"

namespace tw {

typedef unsigned __int128 u128;
constexpr int BIG_NC = 25;               // cells of a 5-bit board; 36 and 64: byte boards

// 5 bits per cell in 128 bits: cell i holds tile (b >> 5i) & 31
struct Board5 {
    u128 b;
    __device__ static Board5 ident(int n_cells)
    {
        Board5 r; r.b = 0;
        for (int i = 0; i < n_cells; ++i) r.b |= (u128)(uint32_t)i << (5 * i);
        return r;
    }
    __device__ uint32_t cell(int i) const { return (uint32_t)(b >> (5 * i)) & 31u; }
    __device__ void slide(int zi, int ti)                                   // the tile at cell ti moves to the blank's cell zi
    {
        const u128 tile = (b >> (5 * ti)) & (u128)31;                       // cell zi holds 0
        b = (b & ~((u128)31 << (5 * ti))) | (tile << (5 * zi));
    }
    __device__ void put(int i, uint32_t tile) { b = (b & ~((u128)31 << (5 * i))) | ((u128)(tile & 31u) << (5 * i)); }
    __device__ bool operator==(const Board5 &o) const { return b == o.b; }
};

// one byte per cell, NC / 4 registers.  Cells are addressed with compile-time indices wherever the cell loop is unrolled; the two
// run-time accesses of a step select among the words (a register array indexed at run time would live in scratch)
template <int NC>
struct Board8 {
    static constexpr int NW = (NC + 3) / 4;
    uint32_t w[NW];
    __device__ static Board8 ident(int n_cells)
    {
        Board8 r;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            uint32_t v = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) { const int i = 4 * k + c; if (i < n_cells) v |= (uint32_t)i << (8 * c); }
            r.w[k] = v;
        }
        return r;
    }
    __device__ uint32_t cell(int i) const                                   // (i: a constant after unrolling)
    {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) v = (k == (i >> 2)) ? w[k] : v;
        return (v >> (8 * (i & 3))) & 255u;
    }
    __device__ void slide(int zi, int ti)
    {
        const uint32_t tile = cell(ti);
        const uint32_t clr = ~(255u << (8 * (ti & 3))), put = tile << (8 * (zi & 3));
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            uint32_t v = w[k];
            v = (k == (ti >> 2)) ? (v & clr) : v;
            v = (k == (zi >> 2)) ? (v | put) : v;                           // cell zi holds 0
            w[k] = v;
        }
    }
    __device__ void put(int i, uint32_t tile)
    {
        const uint32_t clr = ~(255u << (8 * (i & 3))), val = (tile & 255u) << (8 * (i & 3));
#pragma unroll
        for (int k = 0; k < NW; ++k) w[k] = (k == (i >> 2)) ? ((w[k] & clr) | val) : w[k];
    }
    __device__ bool operator==(const Board8 &o) const
    {
        uint32_t d = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) d |= w[k] ^ o.w[k];
        return d == 0;
    }
};

// a board from one byte per cell (solve() from a given state)
template <typename Board>
__device__ inline Board big_board_from_cells(const uint8_t *cells, int n_cells, const Board &solved)
{
    Board b = solved;                                   // (cell i of the solved board holds tile i: slide tile by tile into place would be
    for (int i = 0; i < n_cells; ++i) b.put(i, cells[i]);   //  the long way round -- every cell is simply overwritten)
    return b;
}

template <int NC> struct BoardOf { using T = Board8<NC>; };
template <> struct BoardOf<BIG_NC> { using T = Board5; };

template <typename Board>
struct BigLaneT { Board board; int32_t zx, zy, depth; };

template <typename Board>
__device__ inline void big_step(BigLaneT<Board> &s, const PuzzleConsts &c, int action)          // Env::step (puzzle.rs:135-160), as puzzle_step
{
    const int dx = (action == 2 ? 1 : 0) - (action == 0 ? 1 : 0), dy = (action == 3 ? 1 : 0) - (action == 1 ? 1 : 0);
    int nx = s.zx + dx, ny = s.zy + dy;
    const bool ok = (unsigned)nx < (unsigned)c.width && (unsigned)ny < (unsigned)c.height;
    nx = ok ? nx : s.zx; ny = ok ? ny : s.zy;
    const int zi = s.zy * c.width + s.zx, ti = ny * c.width + nx;
    s.board.slide(zi, ti);                                                  // (an illegal move: zi == ti, the board stays)
    s.zx = nx; s.zy = ny;
    s.depth = s.depth > 0 ? s.depth - 1 : 0;
}

// Env::reward / is_final / masks of such a state (puzzle.rs:162-181)
template <typename Board>
__device__ inline float big_reward(const BigLaneT<Board> &s, const Board &ident, const PuzzleConsts &c)
{
    return s.board == ident ? 1.0f : (s.depth == 0 ? -0.5f : c.r_step);
}
template <typename Board>
__device__ inline bool big_final(const BigLaneT<Board> &s, const Board &ident) { return s.depth == 0 || s.board == ident; }
template <typename Board>
__device__ inline uint32_t big_maskbits(const BigLaneT<Board> &s, const PuzzleConsts &c)
{
    return (s.zx > 0 ? 1u : 0u) | (s.zy > 0 ? 2u : 0u) | (s.zx < c.width - 1 ? 4u : 0u) | (s.zy < c.height - 1 ? 8u : 0u);
}

}  // namespace tw
