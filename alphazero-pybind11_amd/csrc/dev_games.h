// Device-side game rules (gfx950): per-game step / valid_moves / terminal /
// canonicalize on bitboards held in registers.  Each game is a policy struct
// consumed by the tree kernels in engine_kernels.h.
//
// Connect4 restates Connect4GS (reference connect4_gs.cc) on two 42-bit
// bitboards, bit index = h*7 + w with h = 0 the TOP row, exactly the [h][w]
// order of the reference's board tensor (connect4_gs.h:16-19), so plane p of the
// canonical tensor is bit-for-cell the bitboard of player p.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev_rng.h"

namespace azmi {

struct Connect4 {
  static constexpr int kGameId = 0;
  static constexpr int P = 2;          // NUM_PLAYERS, connect4_gs.h:12
  static constexpr int M = 7;          // NUM_MOVES,   connect4_gs.h:11
  static constexpr int C = 4, H = 6, W = 7;  // CANONICAL_SHAPE, connect4_gs.h:15
  static constexpr int CANON = C * H * W;
  static constexpr int MAXK = 7;       // children per node
  static constexpr int MAX_TURNS = 42; // board cells
  static constexpr int GROUP = 8;      // lanes cooperating on one game slot
  static constexpr uint64_t kTop = 0x7FULL;
  static constexpr uint64_t kCol0 = 0x810204081ULL;  // bits 0,7,14,21,28,35

  struct State {
    uint64_t bb[2];
    uint32_t turn;
    uint32_t player;
  };

  __host__ __device__ static State initial() { return State{{0, 0}, 0, 0}; }

  // connect4_gs.cc:39-46 — bit w set iff column w's top cell is empty
  __host__ __device__ static uint32_t valid_mask(const State& s) {
    return static_cast<uint32_t>(~(s.bb[0] | s.bb[1]) & kTop);
  }
  __host__ __device__ static uint32_t num_valid(const State& s) { return __builtin_popcount(valid_mask(s)); }
  // i-th legal move in ascending move order
  __host__ __device__ static uint32_t nth_valid(const State& s, uint32_t i) {
    uint32_t m = valid_mask(s);
    for (uint32_t j = 0; j < i; ++j) m &= m - 1;
    return __builtin_ctz(m);
  }

  // connect4_gs.cc:48-58 — drop into the lowest empty row; returns false on a full column
  __host__ __device__ static bool play(State& s, uint32_t mv) {
    const uint64_t col = kCol0 << mv;
    const int cnt = __builtin_popcountll((s.bb[0] | s.bb[1]) & col);
    if (cnt >= H) return false;
    const int h = H - 1 - cnt;
    const uint64_t bit = 1ULL << (h * W + mv);
    if (s.player == 0) s.bb[0] |= bit; else s.bb[1] |= bit;  // no runtime index: keeps State in registers
    s.player = (s.player + 1) & 1;
    ++s.turn;
    return true;
  }

  static constexpr uint64_t rows_of(uint64_t pattern) {
    uint64_t m = 0;
    for (int h = 0; h < H; ++h) m |= pattern << (h * W);
    return m;
  }
  __host__ __device__ static bool four(uint64_t b) {
    constexpr uint64_t lo = rows_of(0x0F);  // columns 0..3 of every row
    constexpr uint64_t hi = rows_of(0x78);  // columns 3..6 of every row
    const uint64_t horiz = b & (b >> 1) & (b >> 2) & (b >> 3) & lo;        // (h, w..w+3), w <= 3
    const uint64_t vert = b & (b >> 7) & (b >> 14) & (b >> 21);            // (h..h+3, w)
    const uint64_t diag1 = b & (b >> 8) & (b >> 16) & (b >> 24) & lo;      // (h+x, w+x), w <= 3
    const uint64_t diag2 = b & (b >> 6) & (b >> 12) & (b >> 18) & hi;      // (h+x, w-x), w >= 3
    return (horiz | vert | diag1 | diag2) != 0;
  }
  // connect4_gs.cc:60-129 — 0 = running, 1 + index of the one-hot score entry otherwise
  // (1 = player 0 won, 2 = player 1 won, 3 = draw); player 0 is tested first.
  __host__ __device__ static uint32_t terminal(const State& s) {
    if (four(s.bb[0])) return 1;
    if (four(s.bb[1])) return 2;
    if (((s.bb[0] | s.bb[1]) & kTop) == kTop) return 3;
    return 0;
  }

  // connect4_gs.cc:33-37 — cells + player take part (build-defined mixing)
  __host__ __device__ static uint64_t key(const State& s) {
    uint64_t k = mix64(s.bb[0] ^ 0xC4C4C4C4ULL);
    k = mix64(k ^ s.bb[1]);
    return mix64(k ^ static_cast<uint64_t>(s.player));
  }

  // connect4_gs.cc:131-149 — element e of the [4,6,7] canonical tensor
  __host__ __device__ static float canonical_at(const State& s, uint32_t e) {
    const uint32_t plane = e / (H * W), cell = e % (H * W);
    if (plane < 2) return static_cast<float>(((plane == 0 ? s.bb[0] : s.bb[1]) >> cell) & 1ULL);
    return (plane - 2 == s.player) ? 1.0f : 0.0f;
  }
};

}  // namespace azmi
