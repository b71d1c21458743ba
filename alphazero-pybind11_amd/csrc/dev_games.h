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

  // Connect4GS::to_bytes image (connect4_gs.cc:172-190): int8 board[2][6][7], int8 player, int32 turn (LE)
  static constexpr uint32_t SERIALIZED = 89;
  __host__ __device__ static State from_bytes(const uint8_t* b) {
    State s{{0, 0}, 0, 0};
    for (uint32_t p = 0; p < 2; ++p)
      for (uint32_t i = 0; i < 42; ++i)
        if (b[p * 42 + i]) { if (p == 0) s.bb[0] |= 1ULL << i; else s.bb[1] |= 1ULL << i; }
    s.player = b[84] & 1u;
    s.turn = uint32_t(b[85]) | uint32_t(b[86]) << 8 | uint32_t(b[87]) << 16 | uint32_t(b[88]) << 24;
    return s;
  }

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

// =====================================================================================================
// Tawlbwrdd — 11x11 tafl (reference tawlbwrdd_gs.h:31-240, tawlbwrdd_gs.cc), on 121-bit bitboards.
// Square index sq = h*11 + w; layer bitboards for defenders and attackers (2 x u64 each), the king
// as a square index.  Move index = sq_from*22 + (column move ? 11 + new_h : new_w)
// (tawlbwrdd_gs.cc:176-214, tafl_helper.h:7-14).
//
// Repetition (tawlbwrdd_gs.cc:253-259, 286-331): the reference keeps a map (board, player) -> count
// that is cleared by every capture.  Here it is the LIST of 64-bit position keys since the last
// capture; count(new position) = 1 + number of equal keys in the list (a 64-bit key collision would
// be needed to differ from the reference's exact board comparison).
// =====================================================================================================
namespace azmi {

// ---- rook slides on a 121-bit board without walking squares (round 4) -------------------------------------------------------------
// The occupancy of the piece's row / column as a W-bit word (the row: one 128-bit shift; the column: H single-bit picks), then the
// squares the piece can reach along it: everything between the piece and the nearest blocker on either side (count-trailing /
// leading-zeros on the blockers above / below it).  Branch-free and the same ~100 instructions for every lane; the square-by-square
// walk it replaces ran up to ten steps of ~30 instructions in each of four directions and as long as the wavefront's busiest lane.
template <int W_>
__host__ __device__ __forceinline__ uint32_t az_row_bits(uint64_t lo, uint64_t hi, int h) {
  const uint32_t sh = static_cast<uint32_t>(h * W_);
  const uint64_t v = sh == 0 ? lo : sh < 64 ? (lo >> sh) | (hi << (64 - sh)) : hi >> (sh - 64);
  return static_cast<uint32_t>(v) & ((1u << W_) - 1u);
}
template <int W_, int H_>
__host__ __device__ __forceinline__ uint32_t az_col_bits(uint64_t lo, uint64_t hi, int w) {
  uint32_t c = 0;
#pragma unroll
  for (int t = 0; t < H_; ++t) {
    const uint32_t sq = static_cast<uint32_t>(t * W_ + w);
    c |= static_cast<uint32_t>(((sq < 64 ? lo >> sq : hi >> (sq - 64)) & 1ULL)) << t;
  }
  return c;
}
// squares of a line of n that a piece on `pos` slides to, given the line's blockers (the piece's own bit may be set: it is masked out)
__host__ __device__ __forceinline__ uint32_t az_ray_mask(uint32_t blockers, int pos, int n) {
  const uint32_t below = (1u << pos) - 1u;
  const uint32_t above = ((1u << n) - 1u) & ~((2u << pos) - 1u);
  const uint32_t up = blockers & above, dn = blockers & below;
  const uint32_t right = up ? (above & ((1u << __builtin_ctz(up)) - 1u)) : above;
  const uint32_t left = dn ? (below & ~((2u << (31 - __builtin_clz(dn))) - 1u)) : below;
  return left | right;
}

struct Tawlbwrdd {
  static constexpr int kGameId = 1;
  static constexpr bool kRelative = false;     // relative_values(), game_state.h:114
  static constexpr int P = 2;
  static constexpr int W = 11, H = 11, SQ = 121;
  static constexpr int M = SQ * (W + H);       // 2662
  static constexpr int C = 7;
  static constexpr int CANON = C * SQ;          // 847
  static constexpr int MAXK = 512;              // children per node (hard cap; > any reachable position)
  static constexpr int MAX_TURNS = 400;         // DEFAULT_MAX_TURNS, tawlbwrdd_gs.h:20
  static constexpr int GROUP = 64;              // one wavefront per game slot
  static constexpr uint32_t kNoKing = 127;
  static constexpr int STATE_WORDS = 5;

  struct State {
    uint64_t def[2], atk[2];
    uint32_t king;     // square of the king or kNoKing
    uint32_t turn, player, rep;  // rep = current_repetition_count_
  };

  // engine state words (engine_kernels_big.h load/store): word 4 = king | turn << 8 | player << 24 | rep << 32
  __host__ __device__ static uint32_t player_from_words(const uint64_t* words, uint32_t S, uint32_t slot) {
    return static_cast<uint32_t>(words[4 * static_cast<size_t>(S) + slot] >> 24) & 1u;
  }

  __host__ __device__ static bool bit(const uint64_t (&b)[2], uint32_t sq) { return ((sq < 64 ? b[0] : b[1]) >> (sq & 63)) & 1ULL; }
  __host__ __device__ static void setb(uint64_t (&b)[2], uint32_t sq) { if (sq < 64) b[0] |= 1ULL << sq; else b[1] |= 1ULL << (sq - 64); }
  __host__ __device__ static void clrb(uint64_t (&b)[2], uint32_t sq) { if (sq < 64) b[0] &= ~(1ULL << sq); else b[1] &= ~(1ULL << (sq - 64)); }

  __host__ __device__ static State initial() {  // Lewis cross, tawlbwrdd_gs.h:91-135
    State s{};
    s.king = 5 * W + 5;
    const int defs[12][2] = {{2,5},{3,5},{4,5},{5,4},{5,3},{5,2},{6,5},{7,5},{8,5},{5,6},{5,7},{5,8}};
    for (auto& d : defs) setb(s.def, d[0] * W + d[1]);
    const int atks[24][2] = {{0,4},{0,5},{0,6},{1,4},{1,5},{1,6},{9,4},{9,5},{9,6},{10,4},{10,5},{10,6},
                             {4,0},{5,0},{6,0},{4,1},{5,1},{6,1},{4,9},{5,9},{6,9},{4,10},{5,10},{6,10}};
    for (auto& a : atks) setb(s.atk, a[0] * W + a[1]);
    s.turn = 0; s.player = 0; s.rep = 1;
    return s;
  }
  __host__ __device__ static bool occupied(const State& s, uint32_t sq) { return bit(s.def, sq) || bit(s.atk, sq) || s.king == sq; }
  // player 0 = attackers, player 1 = king side (tawlbwrdd_gs.h:26-29)
  __host__ __device__ static bool own_piece(const State& s, uint32_t p, uint32_t sq) {
    return p == 0 ? bit(s.atk, sq) : (bit(s.def, sq) || s.king == sq);
  }
  __host__ __device__ static bool empty_at(const State& s, int h, int w) {  // is_valid_square, tawlbwrdd_gs.cc:132-140
    if (w < 0 || w >= W || h < 0 || h >= H) return false;
    return !occupied(s, h * W + w);
  }
  // 22-bit target mask of the piece on sq: bits 0..10 = new_w (row slides), bits 11..21 = new_h
  __host__ __device__ static uint32_t slide_mask(const State& s, uint32_t sq) {
    const int h = sq / W, w = sq % W;
    uint64_t lo = s.def[0] | s.atk[0], hi = s.def[1] | s.atk[1];
    if (s.king != kNoKing) { if (s.king < 64) lo |= 1ULL << s.king; else hi |= 1ULL << (s.king - 64); }
    return az_ray_mask(az_row_bits<W>(lo, hi, h), w, W) | (az_ray_mask(az_col_bits<W, H>(lo, hi, w), h, H) << W);
  }
  // tawlbwrdd_gs.cc:142-174: some piece of the side to move has an empty orthogonal neighbour.  On the bitboards: the mover's
  // pieces shifted one square east / west (without the wrap from one row's end to the next row's start) / south / north, against
  // the empty squares (round 4: the square-by-square scan was ~500 instructions of every simulation's terminal test).
  __host__ __device__ static bool has_valid_moves(const State& s) {
    uint64_t olo = s.def[0] | s.atk[0], ohi = s.def[1] | s.atk[1];
    uint64_t mlo = s.player == 0 ? s.atk[0] : s.def[0], mhi = s.player == 0 ? s.atk[1] : s.def[1];
    if (s.king != kNoKing) {
      const uint64_t kl = s.king < 64 ? 1ULL << s.king : 0ULL, kh = s.king < 64 ? 0ULL : 1ULL << (s.king - 64);
      olo |= kl; ohi |= kh;
      if (s.player == 1) { mlo |= kl; mhi |= kh; }
    }
    // columns 0 and 10 of the 121-bit board (bit sq = h * 11 + w): constants after unrolling
    uint64_t col0lo = 0, col0hi = 0, col10lo = 0, col10hi = 0;
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      const int a = hh * W, b = hh * W + W - 1;
      if (a < 64) col0lo |= 1ULL << a; else col0hi |= 1ULL << (a - 64);
      if (b < 64) col10lo |= 1ULL << b; else col10hi |= 1ULL << (b - 64);
    }
    const uint64_t elo = ~olo, ehi = ~ohi & ((1ULL << (SQ - 64)) - 1ULL);       // empty squares of the board
    const uint64_t east_lo = (mlo << 1) & ~col0lo, east_hi = ((mhi << 1) | (mlo >> 63)) & ~col0hi;
    const uint64_t west_lo = ((mlo >> 1) | (mhi << 63)) & ~col10lo, west_hi = (mhi >> 1) & ~col10hi;
    const uint64_t south_lo = mlo << W, south_hi = (mhi << W) | (mlo >> (64 - W));
    const uint64_t north_lo = (mlo >> W) | (mhi << (64 - W)), north_hi = mhi >> W;
    return (((east_lo | west_lo | south_lo | north_lo) & elo) | ((east_hi | west_hi | south_hi | north_hi) & ehi)) != 0;
  }
  __host__ __device__ static void remove_at(State& s, uint32_t sq) {
    clrb(s.def, sq); clrb(s.atk, sq);
    if (s.king == sq) s.king = kNoKing;
  }
  // custodial capture test, tawlbwrdd_gs.cc:222-244: mover on (fh,fw), victim one step along (dh,dw)
  __host__ __device__ static bool captured(const State& s, uint32_t mover, int fh, int fw, int dh, int dw) {
    const int th = fh + dh, tw = fw + dw;
    if (tw < 0 || tw >= W || th < 0 || th >= H) return false;
    if (!own_piece(s, mover ^ 1u, th * W + tw)) return false;
    const int zh = th + dh, zw = tw + dw;
    if (zw < 0 || zw >= W || zh < 0 || zh >= H) return false;
    return own_piece(s, mover, zh * W + zw);
  }
  // tawlbwrdd_gs.cc:246-321 without the repetition bookkeeping; *captured_any tells the caller to clear it.
  // Returns false on an illegal move (empty source or non-slide).
  __host__ __device__ static bool apply_move(State& s, uint32_t mv, bool* captured_any) {
    *captured_any = false;
    if (mv >= static_cast<uint32_t>(M)) return false;
    uint32_t new_loc = mv % (W + H);
    const bool height_move = new_loc >= static_cast<uint32_t>(W);
    if (height_move) new_loc -= W;
    const uint32_t from = mv / (W + H);
    const int ph = from / W, pw = from % W;
    const int nh = height_move ? static_cast<int>(new_loc) : ph, nw = height_move ? pw : static_cast<int>(new_loc);
    const uint32_t to = nh * W + nw;
    const uint32_t mover = s.player;
    if (!own_piece(s, mover, from)) return false;
    if (s.king == from) s.king = to;
    else if (bit(s.def, from)) { clrb(s.def, from); setb(s.def, to); }
    else { clrb(s.atk, from); setb(s.atk, to); }
    if (captured(s, mover, nh, nw, -1, 0)) { remove_at(s, (nh - 1) * W + nw); *captured_any = true; }
    if (captured(s, mover, nh, nw, 1, 0)) { remove_at(s, (nh + 1) * W + nw); *captured_any = true; }
    if (captured(s, mover, nh, nw, 0, -1)) { remove_at(s, nh * W + nw - 1); *captured_any = true; }
    if (captured(s, mover, nh, nw, 0, 1)) { remove_at(s, nh * W + nw + 1); *captured_any = true; }
    s.player ^= 1u;
    ++s.turn;
    return true;
  }
  // key of (board, player) — what the repetition map is keyed on (tawlbwrdd_gs.h:57-87)
  __host__ __device__ static uint64_t rep_key(const State& s) {
    uint64_t k = mix64(s.def[0] ^ 0x7A77ULL);
    k = mix64(k ^ s.def[1]); k = mix64(k ^ s.atk[0]); k = mix64(k ^ s.atk[1]);
    return mix64(k ^ (static_cast<uint64_t>(s.king) | (static_cast<uint64_t>(s.player) << 8)));
  }
  // position key for the evaluation cache: board, player and repetition count (tawlbwrdd_gs.cc:99-103)
  __host__ __device__ static uint64_t key(const State& s) { return mix64(rep_key(s) ^ (static_cast<uint64_t>(s.rep) << 32)); }

  // tawlbwrdd_gs.cc:345-397 — 0 running, else 1 + index of the winning score entry
  __host__ __device__ static uint32_t terminal(const State& s) {
    if (s.rep >= 3) return 1 + s.player;                     // the side to move is credited
    if (s.king != kNoKing) {
      const int h = s.king / W, w = s.king % W;
      if (h == 0 || h == H - 1 || w == 0 || w == W - 1) return 2;  // king on an edge: defenders
    } else {
      return 1;                                              // no king: attackers
    }
    if (!has_valid_moves(s)) return 1 + (s.player ^ 1u);
    if (s.turn >= static_cast<uint32_t>(MAX_TURNS)) return 3;
    return 0;
  }
  // the canonical tensor of a leaf by one wavefront, plane by plane: the three piece planes from the bitboards, the four flag planes
  // as fills (canonical_at below, element by element, spent a division and a switch on each of the 847 entries)
  __device__ __forceinline__ static void write_canonical_wave(const State& s, float* row, uint32_t lane) {
    const float fl[4] = {s.player == 0 ? 1.0f : 0.0f, s.player == 1 ? 1.0f : 0.0f, (s.rep == 1 || s.rep > 2) ? 1.0f : 0.0f, s.rep >= 2 ? 1.0f : 0.0f};
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const uint32_t sq = it * 64 + lane;
      if (sq < static_cast<uint32_t>(SQ)) {
        row[sq] = s.king == sq ? 1.0f : 0.0f;
        row[SQ + sq] = bit(s.def, sq) ? 1.0f : 0.0f;
        row[2 * SQ + sq] = bit(s.atk, sq) ? 1.0f : 0.0f;
#pragma unroll
        for (int p = 0; p < 4; ++p) row[(3 + p) * SQ + sq] = fl[p];
      }
    }
  }
  // tawlbwrdd_gs.cc:399-453
  __host__ __device__ static float canonical_at(const State& s, uint32_t e) {
    const uint32_t plane = e / SQ, sq = e % SQ;
    switch (plane) {
      case 0: return s.king == sq ? 1.0f : 0.0f;
      case 1: return bit(s.def, sq) ? 1.0f : 0.0f;
      case 2: return bit(s.atk, sq) ? 1.0f : 0.0f;
      case 3: return s.player == 0 ? 1.0f : 0.0f;
      case 4: return s.player == 1 ? 1.0f : 0.0f;
      case 5: return (s.rep == 1 || s.rep > 2) ? 1.0f : 0.0f;
      default: return s.rep >= 2 ? 1.0f : 0.0f;
    }
  }
};

// =====================================================================================================
// Brandubh (7x7) and OpenTafl (11x11): the Tawlbwrdd skeleton plus special squares
// (/root/reference/src/brandubh_gs.cc:118-520, opentafl_gs.cc:99-585).  Corners are king-only and hostile to
// everyone; non-king pieces pass through the empty throne but cannot land on it; the throne is hostile to
// attackers always and to defenders when the king is not on it; the king wins on a CORNER.  Brandubh's
// king is captured like any piece; OpenTafl's needs four hostile sides and is safe on an edge, attackers also
// win by encirclement (flood fill from the rim), the position key includes the turn and canonical plane 7
// is turn / max_turns.  Same state record and interface as Tawlbwrdd, so the wide-game engine
// (engine_kernels_big.h) is instantiated unchanged.
// =====================================================================================================
template <int VARIANT>   // 0 = Brandubh, 1 = OpenTafl
struct TaflX {
  static constexpr bool kOpen = VARIANT == 1;
  static constexpr int kGameId = 2 + VARIANT;
  static constexpr bool kRelative = false;
  static constexpr int P = 2;
  static constexpr int N = kOpen ? 11 : 7;
  static constexpr int W = N, H = N, SQ = N * N, T = N / 2;
  static constexpr int M = SQ * (W + H);
  static constexpr int C = kOpen ? 8 : 7;
  static constexpr int CANON = C * SQ;
  static constexpr int MAXK = kOpen ? 512 : 256;
  static constexpr int MAX_TURNS = kOpen ? 400 : 150;   // opentafl_gs.h:18, brandubh_gs.h:35
  static constexpr int GROUP = 64;
  static constexpr uint32_t kNoKing = 127;
  static constexpr int STATE_WORDS = 5;
  using State = Tawlbwrdd::State;

  __host__ __device__ __forceinline__ static uint32_t player_from_words(const uint64_t* words, uint32_t S, uint32_t slot) {
    return static_cast<uint32_t>(words[4 * static_cast<size_t>(S) + slot] >> 24) & 1u;
  }
  __host__ __device__ __forceinline__ static bool bit(const uint64_t (&b)[2], uint32_t sq) { return ((sq < 64 ? b[0] : b[1]) >> (sq & 63)) & 1ULL; }
  __host__ __device__ __forceinline__ static void setb(uint64_t (&b)[2], uint32_t sq) { if (sq < 64) b[0] |= 1ULL << sq; else b[1] |= 1ULL << (sq - 64); }
  __host__ __device__ __forceinline__ static void clrb(uint64_t (&b)[2], uint32_t sq) { if (sq < 64) b[0] &= ~(1ULL << sq); else b[1] &= ~(1ULL << (sq - 64)); }

  __host__ __device__ __forceinline__ static State initial() {
    State s{};
    s.king = T * W + T;
    if (kOpen) {  // opentafl_gs.h:90-135
      const int defs[12][2] = {{3,5},{4,5},{5,4},{5,3},{6,5},{7,5},{5,6},{5,7},{4,4},{4,6},{6,4},{6,6}};
      for (auto& d : defs) setb(s.def, d[0] * W + d[1]);
      const int atks[24][2] = {{0,3},{0,4},{0,5},{0,6},{0,7},{1,5},{10,3},{10,4},{10,5},{10,6},{10,7},{9,5},
                               {3,0},{4,0},{5,0},{6,0},{7,0},{5,1},{3,10},{4,10},{5,10},{6,10},{7,10},{5,9}};
      for (auto& a : atks) setb(s.atk, a[0] * W + a[1]);
    } else {      // brandubh_gs.h:92-111
      const int defs[4][2] = {{2,3},{3,2},{4,3},{3,4}};
      for (auto& d : defs) setb(s.def, d[0] * W + d[1]);
      const int atks[8][2] = {{1,3},{0,3},{3,1},{3,0},{5,3},{6,3},{3,5},{3,6}};
      for (auto& a : atks) setb(s.atk, a[0] * W + a[1]);
    }
    s.turn = 0; s.player = 0; s.rep = 1;
    return s;
  }
  // a position given as the reference's board tensor: int8 [3][N][N] (king, defenders, attackers) + player + turn
  // (the test helper MakeGS, opentafl_gs_test.cc:97-101: empty repetition map, count 1)
  __host__ __device__ __forceinline__ static State from_board(const uint8_t* b, uint32_t player, uint32_t turn) {
    State s{};
    s.king = kNoKing;
    for (uint32_t sq = 0; sq < static_cast<uint32_t>(SQ); ++sq) {
      if (b[sq]) s.king = sq;
      if (b[SQ + sq]) setb(s.def, sq);
      if (b[2 * SQ + sq]) setb(s.atk, sq);
    }
    s.turn = turn; s.player = player & 1u; s.rep = 1;
    return s;
  }
  __host__ __device__ __forceinline__ static bool corner(int h, int w) { return (h == 0 || h == H - 1) && (w == 0 || w == W - 1); }
  __host__ __device__ __forceinline__ static bool occupied(const State& s, uint32_t sq) { return bit(s.def, sq) || bit(s.atk, sq) || s.king == sq; }
  // (by VALUE: written as `p == 0 ? bit(s.atk, sq) : bit(s.def, sq)` the two loads were merged into one load through a SELECT OF
  // POINTERS with a run-time offset - the whole BigSlot object that holds the state then stays in memory, and with it the by-value
  // kernel arguments it refers to: 1.2 KB of private scratch per lane and flat accesses in every OpenTafl kernel, rounds 2-5)
  __host__ __device__ __forceinline__ static bool own_piece(const State& s, uint32_t p, uint32_t sq) {
    const uint64_t alo = s.atk[0], ahi = s.atk[1], dlo = s.def[0], dhi = s.def[1];
    const uint64_t lo = p == 0 ? alo : dlo, hi = p == 0 ? ahi : dhi;
    const bool on = (((sq < 64 ? lo : hi) >> (sq & 63)) & 1ULL) != 0;
    return on || (p != 0 && s.king == sq);
  }
  // is_valid_square, brandubh_gs.cc:138-154 / opentafl_gs.cc:137-153
  __host__ __device__ __forceinline__ static bool valid_square(const State& s, bool is_king, int h, int w) {
    if (w < 0 || w >= W || h < 0 || h >= H) return false;
    if (corner(h, w)) return is_king;
    return !occupied(s, h * W + w);
  }
  // 2N-bit target mask of the piece on sq: bits 0..N-1 = new_w (row slides), bits N..2N-1 = new_h
  // (is_valid_square: a corner stops everyone but the king, whatever stands on it; the empty throne lets a non-king piece through
  // but not land: not a blocker, its bit is cleared from the landing squares)
  __host__ __device__ __forceinline__ static uint32_t slide_mask(const State& s, uint32_t sq) {
    const int h = sq / W, w = sq % W;
    const bool k = s.king == sq;
    uint64_t lo = s.def[0] | s.atk[0], hi = s.def[1] | s.atk[1];
    if (s.king != kNoKing) { if (s.king < 64) lo |= 1ULL << s.king; else hi |= 1ULL << (s.king - 64); }
    constexpr uint32_t ends = 1u | (1u << (N - 1));
    uint32_t rowb = az_row_bits<W>(lo, hi, h), colb = az_col_bits<W, H>(lo, hi, w);
    if (k) { rowb &= (h == 0 || h == H - 1) ? ~ends : ~0u; colb &= (w == 0 || w == W - 1) ? ~ends : ~0u; }   // (a corner is the king's to land on)
    else { rowb |= (h == 0 || h == H - 1) ? ends : 0u; colb |= (w == 0 || w == W - 1) ? ends : 0u; }
    uint32_t rm = az_ray_mask(rowb, w, W), cm = az_ray_mask(colb, h, H);
    if (!k) { if (h == T) rm &= ~(1u << T); if (w == T) cm &= ~(1u << T); }
    return rm | (cm << W);
  }
  __host__ __device__ __forceinline__ static bool has_valid_moves(const State& s) {  // brandubh_gs.cc:156-213 / opentafl_gs.cc:155-212
    for (uint32_t sq = 0; sq < static_cast<uint32_t>(SQ); ++sq)
      if (own_piece(s, s.player, sq) && slide_mask(s, sq) != 0) return true;
    return false;
  }
  __host__ __device__ __forceinline__ static void remove_at(State& s, uint32_t sq) {
    clrb(s.def, sq); clrb(s.atk, sq);
    if (s.king == sq) s.king = kNoKing;
  }
  // is_hostile_to, brandubh_gs.cc:278-305 / opentafl_gs.cc:277-297: is (h, w) hostile to a piece of player p
  __host__ __device__ __forceinline__ static bool hostile_to(const State& s, uint32_t p, int h, int w) {
    if (corner(h, w)) return true;
    if (h == T && w == T) return p == 1 ? s.king != static_cast<uint32_t>(T * W + T) : true;
    return own_piece(s, p ^ 1u, h * W + w);
  }
  // captured(), brandubh_gs.cc:307-340 / opentafl_gs.cc:299-334: mover sits on (fh, fw), victim one step along (dh, dw)
  __host__ __device__ __forceinline__ static bool captured(const State& s, uint32_t mover, int fh, int fw, int dh, int dw) {
    const int th = fh + dh, tw = fw + dw;
    if (tw < 0 || tw >= W || th < 0 || th >= H) return false;
    const uint32_t tsq = th * W + tw;
    if (kOpen && s.king == tsq) {
      if (th == 0 || th == H - 1 || tw == 0 || tw == W - 1) return false;
      return hostile_to(s, 1, th - 1, tw) && hostile_to(s, 1, th + 1, tw) && hostile_to(s, 1, th, tw - 1) && hostile_to(s, 1, th, tw + 1);
    }
    if (!own_piece(s, mover ^ 1u, tsq)) return false;
    const int zh = th + dh, zw = tw + dw;
    if (zw < 0 || zw >= W || zh < 0 || zh >= H) return false;
    return hostile_to(s, mover ^ 1u, zh, zw);
  }
  __host__ __device__ __forceinline__ static bool apply_move(State& s, uint32_t mv, bool* captured_any, bool unchecked = false) {
    *captured_any = false;
    if (mv >= static_cast<uint32_t>(M)) return false;
    uint32_t new_loc = mv % (W + H);
    const bool height_move = new_loc >= static_cast<uint32_t>(W);
    if (height_move) new_loc -= W;
    const uint32_t from = mv / (W + H);
    const int ph = from / W, pw = from % W;
    const int nh = height_move ? static_cast<int>(new_loc) : ph, nw = height_move ? pw : static_cast<int>(new_loc);
    const uint32_t to = nh * W + nw;
    // `unchecked`: the reference's play_move moves whatever stands on the square and never checks the slide
    // (its own tests rely on that, brandubh_gs_test.cc:10-23, opentafl_gs_test.cc:408-420); captures are judged
    // from the moved piece's side (piece_to_player of the new square).  The engine always plays checked moves.
    uint32_t mover = s.player;
    if (unchecked) {
      if (!occupied(s, from) || (to != from && occupied(s, to))) return false;
      mover = bit(s.atk, from) ? 0u : 1u;
    } else if (!own_piece(s, mover, from)) {
      return false;
    }
    if (s.king == from) s.king = to;
    else if (bit(s.def, from)) { clrb(s.def, from); setb(s.def, to); }
    else { clrb(s.atk, from); setb(s.atk, to); }
    if (captured(s, mover, nh, nw, -1, 0)) { remove_at(s, (nh - 1) * W + nw); *captured_any = true; }
    if (captured(s, mover, nh, nw, 1, 0)) { remove_at(s, (nh + 1) * W + nw); *captured_any = true; }
    if (captured(s, mover, nh, nw, 0, -1)) { remove_at(s, nh * W + nw - 1); *captured_any = true; }
    if (captured(s, mover, nh, nw, 0, 1)) { remove_at(s, nh * W + nw + 1); *captured_any = true; }
    s.player ^= 1u;
    ++s.turn;
    return true;
  }
  __host__ __device__ __forceinline__ static uint64_t rep_key(const State& s) {
    uint64_t k = mix64(s.def[0] ^ (kOpen ? 0x0F7AULL : 0xB7A0ULL));
    k = mix64(k ^ s.def[1]); k = mix64(k ^ s.atk[0]); k = mix64(k ^ s.atk[1]);
    return mix64(k ^ (static_cast<uint64_t>(s.king) | (static_cast<uint64_t>(s.player) << 8)));
  }
  // evaluation-cache key: board, player, repetition count (+ turn for OpenTafl, opentafl_gs.cc:102-108)
  __host__ __device__ __forceinline__ static uint64_t key(const State& s) {
    const uint64_t k = mix64(rep_key(s) ^ (static_cast<uint64_t>(s.rep) << 32));
    return kOpen ? mix64(k ^ (static_cast<uint64_t>(s.turn) << 16)) : k;
  }
  // ---- encirclement (opentafl_gs.cc:466-506): flood fill from the rim over squares without attackers; the
  // defenders can still reach the edge iff the filled region meets a defender or the king.  121-bit boards.
  struct B128 { uint64_t lo, hi; };
  __host__ __device__ __forceinline__ static B128 shl(B128 x, int n) { return B128{x.lo << n, (x.hi << n) | (x.lo >> (64 - n))}; }
  __host__ __device__ __forceinline__ static B128 shr(B128 x, int n) { return B128{(x.lo >> n) | (x.hi << (64 - n)), x.hi >> n}; }
  __host__ __device__ __forceinline__ static B128 band(B128 a, B128 b) { return B128{a.lo & b.lo, a.hi & b.hi}; }
  __host__ __device__ __forceinline__ static B128 bor(B128 a, B128 b) { return B128{a.lo | b.lo, a.hi | b.hi}; }
  __host__ __device__ __forceinline__ static B128 column(int w) {
    B128 m{0, 0};
    for (int h = 0; h < H; ++h) { const int sq = h * W + w; if (sq < 64) m.lo |= 1ULL << sq; else m.hi |= 1ULL << (sq - 64); }
    return m;
  }
  __host__ __device__ __forceinline__ static bool encircled(const State& s) {
    constexpr uint64_t hi_mask = SQ > 64 ? ((1ULL << ((SQ > 64 ? SQ : 65) - 64)) - 1) : 0ULL;
    constexpr uint64_t lo_mask = SQ >= 64 ? ~0ULL : ((1ULL << (SQ < 64 ? SQ : 0)) - 1);
    const B128 all{lo_mask, hi_mask};
    const B128 free_sq{all.lo & ~s.atk[0], all.hi & ~s.atk[1]};
    const B128 col0 = column(0), colN = column(W - 1);
    B128 rim = bor(col0, colN);
    for (int w = 0; w < W; ++w) { const int a = w, b = (H - 1) * W + w;
      if (a < 64) rim.lo |= 1ULL << a; else rim.hi |= 1ULL << (a - 64);
      if (b < 64) rim.lo |= 1ULL << b; else rim.hi |= 1ULL << (b - 64); }
    B128 reach = band(rim, free_sq);
    for (int it = 0; it < SQ; ++it) {
      const B128 left = band(shr(reach, 1), B128{~colN.lo, ~colN.hi});    // sq - 1 (never wraps into column N-1)
      const B128 right = band(shl(reach, 1), B128{~col0.lo, ~col0.hi});   // sq + 1
      const B128 up = shr(reach, W), down = band(shl(reach, W), all);
      const B128 next = bor(reach, band(bor(bor(left, right), bor(up, down)), free_sq));
      if (next.lo == reach.lo && next.hi == reach.hi) break;
      reach = next;
    }
    uint64_t dlo = s.def[0], dhi = s.def[1];
    if (s.king != kNoKing) { if (s.king < 64) dlo |= 1ULL << s.king; else dhi |= 1ULL << (s.king - 64); }
    return ((reach.lo & dlo) | (reach.hi & dhi)) == 0;
  }
  // scores(), brandubh_gs.cc:441-488 / opentafl_gs.cc:430-520 — 0 running, else 1 + index of the winning entry
  __host__ __device__ __forceinline__ static uint32_t terminal(const State& s) {
    if (s.rep >= 3) return 1 + s.player;
    if (s.king != kNoKing) {
      if (corner(s.king / W, s.king % W)) return 2;
    } else {
      return 1;
    }
    if (kOpen && encircled(s)) return 1;
    if (!has_valid_moves(s)) return 1 + (s.player ^ 1u);
    if (s.turn >= static_cast<uint32_t>(MAX_TURNS)) return 3;
    return 0;
  }
  __host__ __device__ __forceinline__ static float canonical_at(const State& s, uint32_t e) {
    const uint32_t plane = e / SQ, sq = e % SQ;
    switch (plane) {
      case 0: return s.king == sq ? 1.0f : 0.0f;
      case 1: return bit(s.def, sq) ? 1.0f : 0.0f;
      case 2: return bit(s.atk, sq) ? 1.0f : 0.0f;
      case 3: return s.player == 0 ? 1.0f : 0.0f;
      case 4: return s.player == 1 ? 1.0f : 0.0f;
      case 5: return (s.rep == 1 || s.rep > 2) ? 1.0f : 0.0f;
      case 6: return s.rep >= 2 ? 1.0f : 0.0f;
      default: return static_cast<float>(s.turn) / static_cast<float>(MAX_TURNS);   // opentafl_gs.cc:574-579
    }
  }
  // (the same tensor by one wavefront, plane by plane: Tawlbwrdd::write_canonical_wave)
  __device__ __forceinline__ static void write_canonical_wave(const State& s, float* row, uint32_t lane) {
    float fl[C - 3];
    fl[0] = s.player == 0 ? 1.0f : 0.0f; fl[1] = s.player == 1 ? 1.0f : 0.0f;
    fl[2] = (s.rep == 1 || s.rep > 2) ? 1.0f : 0.0f; fl[3] = s.rep >= 2 ? 1.0f : 0.0f;
    if constexpr (C > 7) fl[4] = static_cast<float>(s.turn) / static_cast<float>(MAX_TURNS);
#pragma unroll
    for (int it = 0; it < (SQ + 63) / 64; ++it) {
      const uint32_t sq = it * 64 + lane;
      if (sq < static_cast<uint32_t>(SQ)) {
        row[sq] = s.king == sq ? 1.0f : 0.0f;
        row[SQ + sq] = bit(s.def, sq) ? 1.0f : 0.0f;
        row[2 * SQ + sq] = bit(s.atk, sq) ? 1.0f : 0.0f;
#pragma unroll
        for (int p = 0; p < C - 3; ++p) row[(3 + p) * SQ + sq] = fl[p];
      }
    }
  }
};
using Brandubh = TaflX<0>;
using OpenTafl = TaflX<1>;


// ---- the reference's pickle image of a Tafl-family position (tawlbwrdd_gs.cc:10-37, brandubh_gs.cc:11-41, opentafl_gs.cc:13-40):
//   board int8[3][N][N] (king, defenders, attackers) | u16 turn | u16 max_turns | i8 player | u8 current_repetition_count |
//   u32 n | n x (board int8[3][N][N] | u8 player | u8 count)          (little endian)
// The repetition map travels with the position; on the device it becomes the list of 64-bit (board, player) keys the
// rules already use, every entry repeated `count` times.
template <class GM>
struct TaflImage {
  static constexpr uint32_t BB = 3u * GM::SQ, HEADER = BB + 6u, ENTRY = BB + 2u;
  __host__ __device__ static uint32_t bytes(uint32_t entries) { return HEADER + 4u + entries * ENTRY; }
};
template <class GM>
__host__ __device__ __forceinline__ typename GM::State tafl_board_state(const uint8_t* b, uint32_t player, uint32_t turn) {
  typename GM::State s{};
  s.king = GM::kNoKing;
  for (uint32_t sq = 0; sq < static_cast<uint32_t>(GM::SQ); ++sq) {
    if (b[sq]) s.king = sq;
    if (b[GM::SQ + sq]) { if (sq < 64) s.def[0] |= 1ULL << sq; else s.def[1] |= 1ULL << (sq - 64); }
    if (b[2 * GM::SQ + sq]) { if (sq < 64) s.atk[0] |= 1ULL << sq; else s.atk[1] |= 1ULL << (sq - 64); }
  }
  s.turn = turn; s.player = player & 1u; s.rep = 1;
  return s;
}
// parses an image of at most `row_bytes` bytes: the position into `s`, the repetition keys into reps[0 .. nrep).
// false: malformed (size mismatch, another max_turns than the device game's, more keys than `cap`)
template <class GM>
__host__ __device__ inline bool tafl_parse_image(const uint8_t* row, uint32_t row_bytes, typename GM::State& s, uint64_t* reps,
                                                 uint32_t& nrep, uint32_t cap) {
  using I = TaflImage<GM>;
  nrep = 0;
  if (row_bytes < I::bytes(0)) return false;
  const uint8_t* h = row + I::BB;
  const uint32_t turn = uint32_t(h[0]) | uint32_t(h[1]) << 8, max_turns = uint32_t(h[2]) | uint32_t(h[3]) << 8;
  if (max_turns != static_cast<uint32_t>(GM::MAX_TURNS)) return false;
  s = tafl_board_state<GM>(row, h[4], turn);
  s.rep = h[5];
  const uint32_t n = uint32_t(h[6]) | uint32_t(h[7]) << 8 | uint32_t(h[8]) << 16 | uint32_t(h[9]) << 24;
  if (n > 4096u || I::bytes(n) > row_bytes) return false;
  const uint8_t* e = row + I::bytes(0);
  for (uint32_t i = 0; i < n; ++i, e += I::ENTRY) {
    const uint64_t k = GM::rep_key(tafl_board_state<GM>(e, e[I::BB], 0));
    for (uint32_t c = 0; c < e[I::BB + 1]; ++c) {
      if (nrep >= cap) return false;
      reps[nrep++] = k;
    }
  }
  return true;
}

}  // namespace azmi
