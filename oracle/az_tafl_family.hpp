// ORACLE — TEST INFRASTRUCTURE ONLY (see az_rng.hpp header for the usage rule).
// CPU restatement of the two rule sets that share TawlbwrddGS's skeleton but add special squares:
//   BrandubhGS  7x7   /root/reference/src/brandubh_gs.h:90-112 (start), brandubh_gs.cc:118-520
//   OpenTaflGS  11x11 /root/reference/src/opentafl_gs.h:88-135 (start), opentafl_gs.cc:99-585
// Common: corners are king-only and hostile to everyone; non-king pieces pass through the empty
// throne but cannot land on it; the throne is hostile to attackers always and to defenders when the
// king is not on it; threefold repetition credits the side to move; king on a CORNER wins.
// Brandubh: the king is captured like any piece (brandubh_gs.cc:309-340).  OpenTafl: the king needs four
// hostile sides and is safe on an edge (opentafl_gs.cc:299-317), attackers also win by encirclement
// (flood fill from the rim, :466-506), hash/equality include the turn (:82-108) and canonical plane 7 is
// turn / max_turns (:574-579).
// Pinned by the 25 rule tests of opentafl_gs_test.cc and brandubh_gs_test.cc (tests/test_oracle_pinned.py).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <stdexcept>
#include <utility>
#include <vector>

#include "az_games.hpp"

namespace orc {

enum class TaflVariant { Brandubh, OpenTafl };

template <TaflVariant V>
struct TaflFamily final : Game {
  static constexpr bool kOpen = V == TaflVariant::OpenTafl;
  static constexpr int N = kOpen ? 11 : 7;
  static constexpr int W = N, H = N, T = N / 2;   // throne at (T, T)
  static constexpr int KING = 0, DEF = 1, ATK = 2;
  static constexpr int ATK_PLAYER = 0, DEF_PLAYER = 1;
  static constexpr int CANON_PLANES = kOpen ? 8 : 7;
  static constexpr uint32_t NUM_MOVES = W * H * (W + H);
  static constexpr uint16_t DEFAULT_MAX_TURNS = kOpen ? 400 : 150;   // opentafl_gs.h:18, brandubh_gs.h:35
  using Board = std::array<int8_t, 3 * H * W>;
  using RepKey = std::pair<Board, uint8_t>;

  Board board{};
  uint16_t turn = 0, max_turns = DEFAULT_MAX_TURNS;
  int8_t player = 0;
  uint8_t rep_count = 1;
  std::map<RepKey, uint8_t> reps;

  int8_t& at(int l, int h, int w) { return board[(l * H + h) * W + w]; }
  int8_t at(int l, int h, int w) const { return board[(l * H + h) * W + w]; }

  explicit TaflFamily(uint16_t mt = DEFAULT_MAX_TURNS) : max_turns(mt) {
    board.fill(0);
    at(KING, T, T) = 1;
    if (kOpen) {  // opentafl_gs.h:90-135
      static const int defs[12][2] = {{3,5},{4,5},{5,4},{5,3},{6,5},{7,5},{5,6},{5,7},{4,4},{4,6},{6,4},{6,6}};
      for (auto& d : defs) at(DEF, d[0], d[1]) = 1;
      static const int atks[24][2] = {{0,3},{0,4},{0,5},{0,6},{0,7},{1,5},{10,3},{10,4},{10,5},{10,6},{10,7},{9,5},
                                      {3,0},{4,0},{5,0},{6,0},{7,0},{5,1},{3,10},{4,10},{5,10},{6,10},{7,10},{5,9}};
      for (auto& a : atks) at(ATK, a[0], a[1]) = 1;
    } else {      // brandubh_gs.h:92-111
      static const int defs[4][2] = {{2,3},{3,2},{4,3},{3,4}};
      for (auto& d : defs) at(DEF, d[0], d[1]) = 1;
      static const int atks[8][2] = {{1,3},{0,3},{3,1},{3,0},{5,3},{6,3},{3,5},{3,6}};
      for (auto& a : atks) at(ATK, a[0], a[1]) = 1;
    }
  }
  // the test helper MakeGS (opentafl_gs_test.cc:97-101): given board, empty repetition map, count 1
  TaflFamily(const int8_t* b, int8_t p, uint16_t t, uint16_t mt) : turn(t), max_turns(mt), player(p) {
    std::memcpy(board.data(), b, board.size());
  }

  std::unique_ptr<Game> copy() const override { return std::make_unique<TaflFamily>(*this); }
  uint8_t current_player() const override { return static_cast<uint8_t>(player); }
  uint32_t current_turn() const override { return turn; }
  uint32_t num_moves() const override { return NUM_MOVES; }
  uint8_t num_players() const override { return 2; }

  static bool corner(int h, int w) { return (h == 0 || h == H - 1) && (w == 0 || w == W - 1); }
  bool players_piece(uint8_t p, int h, int w) const {
    return (p == DEF_PLAYER && (at(KING, h, w) == 1 || at(DEF, h, w) == 1)) || (p == ATK_PLAYER && at(ATK, h, w) == 1);
  }
  bool opponent_piece(uint8_t p, int h, int w) const {
    return (p == ATK_PLAYER && (at(KING, h, w) == 1 || at(DEF, h, w) == 1)) || (p == DEF_PLAYER && at(ATK, h, w) == 1);
  }
  uint8_t piece_to_player(int h, int w) const {
    if (at(ATK, h, w) == 1) return ATK_PLAYER;
    if (at(KING, h, w) == 1 || at(DEF, h, w) == 1) return DEF_PLAYER;
    throw std::runtime_error("piece to player called on a square without pieces...rip");
  }
  bool valid_square(bool is_king, int h, int w) const {  // brandubh_gs.cc:138-154, opentafl_gs.cc:137-153
    if (w < 0 || w >= W || h < 0 || h >= H) return false;
    if (corner(h, w)) return is_king;
    return at(0, h, w) == 0 && at(1, h, w) == 0 && at(2, h, w) == 0;
  }
  // walks one ray; `visit(t)` is called for every landing square; returns true if any exists
  template <class F>
  bool ray(bool is_king, int h, int w, int dh, int dw, F&& visit) const {
    bool any = false;
    int th = h + dh, tw = w + dw;
    while (valid_square(is_king, th, tw)) {
      if (th == T && tw == T && !is_king) { th += dh; tw += dw; continue; }   // pass through the throne
      visit(th, tw);
      any = true;
      th += dh; tw += dw;
    }
    return any;
  }
  bool has_valid_moves() const {  // brandubh_gs.cc:156-213, opentafl_gs.cc:155-212
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w)
        if (players_piece(player, h, w)) {
          const bool k = at(KING, h, w) == 1;
          auto none = [](int, int) {};
          if (ray(k, h, w, 0, 1, none) || ray(k, h, w, 0, -1, none) || ray(k, h, w, 1, 0, none) || ray(k, h, w, -1, 0, none)) return true;
        }
    return false;
  }
  void valid_moves(uint8_t* out) const override {  // brandubh_gs.cc:215-276, opentafl_gs.cc:214-275
    std::memset(out, 0, NUM_MOVES);
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w) {
        if (!players_piece(player, h, w)) continue;
        const bool k = at(KING, h, w) == 1;
        const int base = (h * W + w) * (W + H);
        auto row = [&](int, int tw) { out[base + tw] = 1; };
        auto col = [&](int th, int) { out[base + W + th] = 1; };
        ray(k, h, w, 0, 1, row); ray(k, h, w, 0, -1, row); ray(k, h, w, 1, 0, col); ray(k, h, w, -1, 0, col);
      }
  }
  bool hostile_to(uint8_t p, int h, int w) const {  // brandubh_gs.cc:278-305, opentafl_gs.cc:277-297
    if (corner(h, w)) return true;
    if (h == T && w == T) {
      if (p == DEF_PLAYER) return at(KING, T, T) == 0;
      return true;
    }
    return opponent_piece(p, h, w);
  }
  bool captured(int fh, int fw, int dh, int dw) const {  // brandubh_gs.cc:307-340, opentafl_gs.cc:299-334
    const int th = fh + dh, tw = fw + dw;
    if (tw < 0 || tw >= W || th < 0 || th >= H) return false;
    if (kOpen && at(KING, th, tw) == 1) {
      if (th == 0 || th == H - 1 || tw == 0 || tw == W - 1) return false;
      return hostile_to(DEF_PLAYER, th - 1, tw) && hostile_to(DEF_PLAYER, th + 1, tw) &&
             hostile_to(DEF_PLAYER, th, tw - 1) && hostile_to(DEF_PLAYER, th, tw + 1);
    }
    const uint8_t from_player = piece_to_player(fh, fw);
    if (!opponent_piece(from_player, th, tw)) return false;
    const uint8_t target_player = piece_to_player(th, tw);
    const int zh = th + dh, zw = tw + dw;
    if (zw < 0 || zw >= W || zh < 0 || zh >= H) return false;
    return hostile_to(target_player, zh, zw);
  }
  void clear_square(int h, int w) { at(0, h, w) = at(1, h, w) = at(2, h, w) = 0; }

  void play_move(uint32_t move) override {  // brandubh_gs.cc:342-427, opentafl_gs.cc:336-421
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
    auto& c = reps[RepKey(board, static_cast<uint8_t>(player))];
    ++c;
    rep_count = c;
  }

  bool encircled() const {  // opentafl_gs.cc:466-506, literally (deque of live squares, back first)
    std::deque<std::pair<int, int>> visited, live;
    for (int w = 0; w < W; ++w) live.emplace_back(0, w);
    for (int h = 1; h < H; ++h) live.emplace_back(h, W - 1);
    for (int w = W - 2; w >= 0; --w) live.emplace_back(H - 1, w);
    for (int h = H - 2; h >= 1; --h) live.emplace_back(h, 0);
    auto has = [](const std::deque<std::pair<int, int>>& d, std::pair<int, int> p) {
      for (auto& x : d) if (x == p) return true;
      return false;
    };
    while (!live.empty()) {
      auto [h, w] = live.back();
      live.pop_back();
      visited.emplace_back(h, w);
      if (at(KING, h, w) == 1 || at(DEF, h, w) == 1) return false;
      if (at(ATK, h, w) == 0) {
        const std::pair<int, int> nb[4] = {{h - 1, w}, {h + 1, w}, {h, w - 1}, {h, w + 1}};
        for (auto p : nb) {
          if (p.first < 0 || p.first >= H || p.second < 0 || p.second >= W) continue;
          if (!has(visited, p) && !has(live, p)) live.push_back(p);
        }
      }
    }
    return true;
  }

  bool scores(float* out) const override {  // brandubh_gs.cc:441-488, opentafl_gs.cc:430-520
    out[0] = out[1] = out[2] = 0.0f;
    if (rep_count >= 3) { out[player] = 1; return true; }
    if (at(KING, 0, 0) == 1 || at(KING, H - 1, 0) == 1 || at(KING, 0, W - 1) == 1 || at(KING, H - 1, W - 1) == 1) { out[1] = 1; return true; }
    bool king = false;
    for (int i = 0; i < H * W; ++i) king = king || board[KING * H * W + i] == 1;
    if (!king) { out[0] = 1; return true; }
    if (kOpen && encircled()) { out[0] = 1; return true; }
    if (!has_valid_moves()) { out[(player + 1) % 2] = 1; return true; }
    if (turn >= max_turns) { out[2] = 1; return true; }
    return false;
  }

  void canonical_shape(int* c, int* h, int* w) const override { *c = CANON_PLANES; *h = H; *w = W; }
  void canonicalized(float* out) const override {  // brandubh_gs.cc:490-545, opentafl_gs.cc:522-582
    const int HW = H * W;
    for (int i = 0; i < 3 * HW; ++i) out[i] = board[i];
    const int me = player + 3, other = (player + 1) % 2 + 3;
    for (int i = 0; i < HW; ++i) { out[me * HW + i] = 1; out[other * HW + i] = 0; }
    const float p5 = (rep_count == 1 || rep_count > 2) ? 1.0f : 0.0f;
    const float p6 = (rep_count >= 2) ? 1.0f : 0.0f;
    for (int i = 0; i < HW; ++i) { out[5 * HW + i] = p5; out[6 * HW + i] = p6; }
    if (kOpen) {
      const float t = static_cast<float>(turn) / static_cast<float>(max_turns);
      for (int i = 0; i < HW; ++i) out[7 * HW + i] = t;
    }
  }
  uint64_t key() const override {  // board, player, repetition count (+ turn for OpenTafl, opentafl_gs.cc:102-108)
    uint64_t k = kOpen ? 0x0F7AULL : 0xB7A0ULL;
    for (int l = 0; l < 3; ++l) {
      uint64_t lo = 0, hi = 0;
      for (int i = 0; i < H * W; ++i)
        if (board[l * H * W + i]) { if (i < 64) lo |= 1ULL << i; else hi |= 1ULL << (i - 64); }
      k = mix64(k ^ lo);
      k = mix64(k ^ hi);
    }
    k = mix64(k ^ (static_cast<uint64_t>(player) | (static_cast<uint64_t>(rep_count) << 8)));
    return kOpen ? mix64(k ^ (static_cast<uint64_t>(turn) << 16)) : k;
  }
};

using Brandubh = TaflFamily<TaflVariant::Brandubh>;
using OpenTafl = TaflFamily<TaflVariant::OpenTafl>;

}  // namespace orc
