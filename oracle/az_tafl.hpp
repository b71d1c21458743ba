// ORACLE — TEST INFRASTRUCTURE ONLY (see az_rng.hpp header for the usage rule).
// CPU restatement of TawlbwrddGS (tawlbwrdd_gs.h:31-240, tawlbwrdd_gs.cc): 11x11
// tafl, Lewis-cross start, rook slides, custodial capture of any piece (king
// included) between two enemy pieces, threefold repetition, king-on-edge win.
// Dense int8 [layer][h][w] board as in the reference; the device engine uses
// 121-bit bitboards instead.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>

#include "az_games.hpp"

namespace orc {

struct Tawlbwrdd final : Game {
  static constexpr int W = 11, H = 11;
  static constexpr int KING = 0, DEF = 1, ATK = 2;  // layers, tawlbwrdd_gs.h:22-24
  static constexpr int ATK_PLAYER = 0, DEF_PLAYER = 1;
  static constexpr uint32_t NUM_MOVES = W * H * (W + H);
  using Board = std::array<int8_t, 3 * H * W>;
  using RepKey = std::pair<Board, uint8_t>;

  Board board{};
  uint16_t turn = 0, max_turns = 400;
  int8_t player = 0;
  uint8_t rep_count = 1;  // current_repetition_count_
  std::map<RepKey, uint8_t> reps;  // repetition_counts_ (positions since the last capture)

  int8_t& at(int l, int h, int w) { return board[(l * H + h) * W + w]; }
  int8_t at(int l, int h, int w) const { return board[(l * H + h) * W + w]; }

  explicit Tawlbwrdd(uint16_t mt = 400) : max_turns(mt) {  // tawlbwrdd_gs.h:91-135
    board.fill(0);
    at(KING, 5, 5) = 1;
    static const int defs[12][2] = {{2,5},{3,5},{4,5},{5,4},{5,3},{5,2},{6,5},{7,5},{8,5},{5,6},{5,7},{5,8}};
    for (auto& d : defs) at(DEF, d[0], d[1]) = 1;
    static const int atks[24][2] = {{0,4},{0,5},{0,6},{1,4},{1,5},{1,6},{9,4},{9,5},{9,6},{10,4},{10,5},{10,6},
                                    {4,0},{5,0},{6,0},{4,1},{5,1},{6,1},{4,9},{5,9},{6,9},{4,10},{5,10},{6,10}};
    for (auto& a : atks) at(ATK, a[0], a[1]) = 1;
  }

  std::unique_ptr<Game> copy() const override { return std::make_unique<Tawlbwrdd>(*this); }
  uint8_t current_player() const override { return static_cast<uint8_t>(player); }
  uint32_t current_turn() const override { return turn; }
  uint32_t num_moves() const override { return NUM_MOVES; }
  uint8_t num_players() const override { return 2; }

  bool players_piece(uint8_t p, int h, int w) const {  // tawlbwrdd_gs.cc:118-123
    return (p == DEF_PLAYER && (at(KING, h, w) == 1 || at(DEF, h, w) == 1)) ||
           (p == ATK_PLAYER && at(ATK, h, w) == 1);
  }
  bool opponent_piece(uint8_t p, int h, int w) const {  // tawlbwrdd_gs.cc:125-130
    return (p == ATK_PLAYER && (at(KING, h, w) == 1 || at(DEF, h, w) == 1)) ||
           (p == DEF_PLAYER && at(ATK, h, w) == 1);
  }
  bool empty_square(int h, int w) const {  // is_valid_square, tawlbwrdd_gs.cc:132-140
    if (w < 0 || w >= W || h < 0 || h >= H) return false;
    return at(0, h, w) == 0 && at(1, h, w) == 0 && at(2, h, w) == 0;
  }
  uint8_t piece_to_player(int h, int w) const {  // tawlbwrdd_gs.cc:105-116
    if (at(ATK, h, w) == 1) return ATK_PLAYER;
    if (at(KING, h, w) == 1 || at(DEF, h, w) == 1) return DEF_PLAYER;
    throw std::runtime_error("piece to player called on a square without pieces...rip");
  }
  bool has_valid_moves() const {  // tawlbwrdd_gs.cc:142-174
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w)
        if (players_piece(player, h, w))
          if (empty_square(h, w + 1) || empty_square(h, w - 1) || empty_square(h + 1, w) || empty_square(h - 1, w))
            return true;
    return false;
  }
  // tawlbwrdd_gs.cc:176-214 — index (h*W+w)*(W+H) + (column move ? W + new_h : new_w)
  void valid_moves(uint8_t* out) const override {
    std::memset(out, 0, NUM_MOVES);
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w) {
        if (!players_piece(player, h, w)) continue;
        const int base = (h * W + w) * (W + H);
        for (int tw = w + 1; empty_square(h, tw); ++tw) out[base + tw] = 1;
        for (int tw = w - 1; empty_square(h, tw); --tw) out[base + tw] = 1;
        for (int th = h + 1; empty_square(th, w); ++th) out[base + W + th] = 1;
        for (int th = h - 1; empty_square(th, w); --th) out[base + W + th] = 1;
      }
  }
  bool captured(int fh, int fw, int dh, int dw) const {  // tawlbwrdd_gs.cc:222-244
    const int th = fh + dh, tw = fw + dw;
    if (tw < 0 || tw >= W || th < 0 || th >= H) return false;
    const uint8_t from_player = piece_to_player(fh, fw);
    if (!opponent_piece(from_player, th, tw)) return false;
    const uint8_t target_player = piece_to_player(th, tw);
    const int zh = th + dh, zw = tw + dw;
    if (zw < 0 || zw >= W || zh < 0 || zh >= H) return false;
    return opponent_piece(target_player, zh, zw);
  }
  void clear_square(int h, int w) { at(0, h, w) = at(1, h, w) = at(2, h, w) = 0; }

  void play_move(uint32_t move) override {  // tawlbwrdd_gs.cc:246-332
    if (move >= NUM_MOVES) throw std::runtime_error("Invalid move: You have a bug in your code.");
    if (turn == 0) reps[RepKey(board, static_cast<uint8_t>(player))] = 1;
    uint32_t new_loc = move % (W + H);
    const bool height_move = new_loc >= static_cast<uint32_t>(W);
    if (height_move) new_loc -= W;
    const uint32_t piece_loc = move / (W + H);
    const int pw = piece_loc % W, ph = piece_loc / W;
    int nh = ph, nw = pw;
    if (height_move) nh = new_loc; else nw = new_loc;
    for (int l = 0; l < 3; ++l) at(l, nh, nw) = at(l, ph, pw);
    clear_square(ph, pw);
    if (captured(nh, nw, -1, 0)) { clear_square(nh - 1, nw); reps.clear(); }
    if (captured(nh, nw, 1, 0)) { clear_square(nh + 1, nw); reps.clear(); }
    if (captured(nh, nw, 0, -1)) { clear_square(nh, nw - 1); reps.clear(); }
    if (captured(nh, nw, 0, 1)) { clear_square(nh, nw + 1); reps.clear(); }
    player = static_cast<int8_t>((player + 1) % 2);
    ++turn;
    auto& c = reps[RepKey(board, static_cast<uint8_t>(player))];  // 0 if new
    ++c;
    rep_count = c;
  }

  bool scores(float* out) const override {  // tawlbwrdd_gs.cc:345-397
    out[0] = out[1] = out[2] = 0.0f;
    if (rep_count >= 3) { out[player] = 1; return true; }  // side to move wins
    for (int w = 0; w < W; ++w)
      if (at(KING, 0, w) == 1 || at(KING, H - 1, w) == 1) { out[1] = 1; return true; }
    for (int h = 0; h < H; ++h)
      if (at(KING, h, 0) == 1 || at(KING, h, W - 1) == 1) { out[1] = 1; return true; }
    bool king = false;
    for (int i = 0; i < H * W; ++i) king = king || board[KING * H * W + i] == 1;
    if (!king) { out[0] = 1; return true; }
    if (!has_valid_moves()) { out[(player + 1) % 2] = 1; return true; }
    if (turn >= max_turns) { out[2] = 1; return true; }
    return false;
  }

  void canonical_shape(int* c, int* h, int* w) const override { *c = 7; *h = H; *w = W; }
  void canonicalized(float* out) const override {  // tawlbwrdd_gs.cc:399-453
    const int HW = H * W;
    for (int i = 0; i < 3 * HW; ++i) out[i] = board[i];
    const int me = player + 3, other = (player + 1) % 2 + 3;
    for (int i = 0; i < HW; ++i) { out[me * HW + i] = 1; out[other * HW + i] = 0; }
    const float p5 = (rep_count == 1 || rep_count > 2) ? 1.0f : 0.0f;
    const float p6 = (rep_count >= 2) ? 1.0f : 0.0f;
    for (int i = 0; i < HW; ++i) { out[5 * HW + i] = p5; out[6 * HW + i] = p6; }
  }
  // tawlbwrdd_gs.cc:99-103 — board, player and the repetition count participate.
  uint64_t key() const override {
    uint64_t k = 0x7A77ULL;
    for (int l = 0; l < 3; ++l) {
      uint64_t lo = 0, hi = 0;
      for (int i = 0; i < H * W; ++i)
        if (board[l * H * W + i]) { if (i < 64) lo |= 1ULL << i; else hi |= 1ULL << (i - 64); }
      k = mix64(k ^ lo);
      k = mix64(k ^ hi);
    }
    return mix64(k ^ (static_cast<uint64_t>(player) | (static_cast<uint64_t>(rep_count) << 8)));
  }
};

}  // namespace orc
