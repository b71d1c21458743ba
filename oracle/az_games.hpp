// ORACLE — TEST INFRASTRUCTURE ONLY (see az_rng.hpp header for the usage rule).
// CPU restatement of the reference's game plug-in API (game_state.h:55-139) and
// of Connect4GS (connect4_gs.cc).  Boards are kept in the reference's own dense
// int8 [player][h][w] form on purpose: the device engine uses bitboards, so the
// two implementations share no representation.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace orc {

struct Pcg32;   // az_rng.hpp

constexpr int kMaxValue = 5;  // num_players + 1 <= 5

// game_state.h:55-139 — the subset the search path touches.
struct Game {
  virtual ~Game() = default;
  virtual std::unique_ptr<Game> copy() const = 0;
  virtual uint8_t current_player() const = 0;
  virtual uint32_t current_turn() const = 0;
  virtual uint32_t num_moves() const = 0;
  virtual uint8_t num_players() const = 0;
  virtual void valid_moves(uint8_t* out) const = 0;  // dense [num_moves] 0/1
  virtual void play_move(uint32_t m) = 0;
  // returns false if the game is not over; else fills out[num_players+1]
  virtual bool scores(float* out) const = 0;
  virtual void canonical_shape(int* c, int* h, int* w) const = 0;
  virtual void canonicalized(float* out) const = 0;
  virtual bool relative_values() const { return false; }
  virtual void randomize_start() {}
  // randomize_start() for games whose start is random (StarGambitUnified's variant): the reference draws from an
  // unseedable thread_local engine, here the draw comes from the slot's coin stream (build-defined)
  virtual void randomize_start_from(Pcg32&) { randomize_start(); }
  virtual int num_variants() const { return 0; }
  virtual int get_variant_id() const { return -1; }
  // Build-defined deterministic 64-bit position key over exactly the fields
  // the reference's hash() feeds absl (absl hashes are salted per
  // process, so only the participating fields are contract).
  virtual uint64_t key() const = 0;
  size_t canonical_size() const {
    int c, h, w;
    canonical_shape(&c, &h, &w);
    return static_cast<size_t>(c) * h * w;
  }
};

inline uint64_t mix64(uint64_t x) {  // splitmix64 finalizer
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}

// ---------------------------------------------------------------------------
// Connect4GS — connect4_gs.h:8-14, connect4_gs.cc
// ---------------------------------------------------------------------------
struct Connect4 final : Game {
  static constexpr int W = 7, H = 6;
  int8_t board[2][H][W];
  int8_t player = 0;
  int32_t turn = 0;
  Connect4() { std::memset(board, 0, sizeof(board)); }
  Connect4(const int8_t* b, int8_t p, int32_t t) : player(p), turn(t) {
    std::memcpy(board, b, sizeof(board));
  }
  std::unique_ptr<Game> copy() const override { return std::make_unique<Connect4>(*this); }
  uint8_t current_player() const override { return static_cast<uint8_t>(player); }
  uint32_t current_turn() const override { return static_cast<uint32_t>(turn); }
  uint32_t num_moves() const override { return W; }
  uint8_t num_players() const override { return 2; }
  // connect4_gs.cc:39-46 — a column is legal iff its top cell is empty.
  void valid_moves(uint8_t* out) const override {
    for (int w = 0; w < W; ++w) out[w] = (board[0][0][w] == 0 && board[1][0][w] == 0) ? 1 : 0;
  }
  // connect4_gs.cc:48-58 — lowest empty row, h = H-1 .. 0.
  void play_move(uint32_t m) override {
    for (int h = H - 1; h >= 0; --h) {
      if (board[0][h][m] == 0 && board[1][h][m] == 0) {
        board[player][h][m] = 1;
        player = static_cast<int8_t>((player + 1) % 2);
        ++turn;
        return;
      }
    }
    throw std::runtime_error("Invalid move: You have a bug in your code.");
  }
  // connect4_gs.cc:60-129 — player 0 is checked before player 1; per player:
  // rows, columns, then "\" and "/" diagonals; draw iff no legal move.
  bool scores(float* out) const override {
    out[0] = out[1] = out[2] = 0.0f;
    for (int p = 0; p < 2; ++p) {
      for (int h = 0; h < H; ++h) {
        int run = 0;
        for (int w = 0; w < W; ++w) {
          run = (board[p][h][w] == 1) ? run + 1 : 0;
          if (run == 4) { out[p] = 1; return true; }
        }
      }
      for (int w = 0; w < W; ++w) {
        int run = 0;
        for (int h = 0; h < H; ++h) {
          run = (board[p][h][w] == 1) ? run + 1 : 0;
          if (run == 4) { out[p] = 1; return true; }
        }
      }
      for (int h = 0; h < H - 3; ++h) {
        for (int w = 0; w < W - 3; ++w) {
          bool all = true;
          for (int x = 0; x < 4; ++x) all = all && board[p][h + x][w + x] != 0;
          if (all) { out[p] = 1; return true; }
        }
        for (int w = W - 4; w < W; ++w) {
          bool all = true;
          for (int x = 0; x < 4; ++x) all = all && board[p][h + x][w - x] != 0;
          if (all) { out[p] = 1; return true; }
        }
      }
    }
    for (int w = 0; w < W; ++w)
      if (board[0][0][w] == 0 && board[1][0][w] == 0) return false;
    out[2] = 1;
    return true;
  }
  void canonical_shape(int* c, int* h, int* w) const override { *c = 4; *h = H; *w = W; }
  // connect4_gs.cc:131-149 — planes 0,1 absolute stones; plane 2+player all ones.
  void canonicalized(float* out) const override {
    for (int p = 0; p < 2; ++p)
      for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) out[(p * H + h) * W + w] = board[p][h][w];
    const int me = player + 2, other = (player + 1) % 2 + 2;
    for (int i = 0; i < H * W; ++i) {
      out[me * H * W + i] = 1.0f;
      out[other * H * W + i] = 0.0f;
    }
  }
  // connect4_gs.cc:33-37 — board cells + player participate.
  uint64_t key() const override {
    uint64_t bb[2] = {0, 0};
    for (int p = 0; p < 2; ++p)
      for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w)
          if (board[p][h][w]) bb[p] |= 1ULL << (h * W + w);
    uint64_t k = mix64(bb[0] ^ 0xC4C4C4C4ULL);
    k = mix64(k ^ bb[1]);
    return mix64(k ^ static_cast<uint64_t>(player));
  }
  // connect4_gs.cc:172-190 — 84 B board + i8 player + i32 turn.
  std::string to_bytes() const {
    std::string out(89, '\0');
    std::memcpy(&out[0], board, 84);
    out[84] = static_cast<char>(player);
    std::memcpy(&out[85], &turn, 4);
    return out;
  }
};

}  // namespace orc
