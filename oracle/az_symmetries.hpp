// ORACLE — TEST INFRASTRUCTURE ONLY (see az_rng.hpp header for the usage rule).
// Training-sample symmetries: restatement of
//   Connect4GS::symmetries            /root/reference/src/connect4_gs.cc:151-170
//   tafl_helper::policyLocation       /root/reference/src/tafl_helper.h:7-14
//   tafl_helper::mirrorWidth          /root/reference/src/tafl_helper.h:16-53
//   tafl_helper::rot90Clockwise       /root/reference/src/tafl_helper.h:55-137
//   tafl_helper::eightSym             /root/reference/src/tafl_helper.h:139-149
//   TawlbwrddGS::symmetries           /root/reference/src/tawlbwrdd_gs.cc:455-458
// Pinned by the known answers of /root/reference/src/tafl_helper_test.cc (Mirror, Rot90 spot
// tables on 5x5; bijection / order / distinctness properties) — tests/test_oracle_pinned.py.
#pragma once
#include <cstdint>
#include <vector>

namespace orc {

struct Sample {              // PlayHistory, game_state.h:31-35
  int channels = 0, height = 0, width = 0;
  std::vector<float> canonical;   // [c][h][w]
  std::vector<float> v;           // [players + 1]
  std::vector<float> pi;          // [num_moves]
  float& at(int c, int h, int w) { return canonical[(size_t(c) * height + h) * width + w]; }
  float at(int c, int h, int w) const { return canonical[(size_t(c) * height + h) * width + w]; }
};

inline int policy_location(int width, int height, int from_h, int from_w, bool height_move, int new_loc) {
  int base = (from_h * width + from_w) * (width + height);   // tafl_helper.h:9-13
  return height_move ? base + width + new_loc : base + new_loc;
}

inline Sample blank_like(const Sample& b) {
  Sample o;
  o.channels = b.channels; o.height = b.height; o.width = b.width;
  o.canonical.assign(b.canonical.size(), 0.f);
  o.pi.assign(b.pi.size(), 0.f);
  o.v = b.v;                                                  // tafl_helper.h:28 / :71
  return o;
}

// tafl_helper.h:16-53 — every cell and every move keeps its row; columns flip.
inline Sample mirror_width(const Sample& b) {
  Sample o = blank_like(b);
  const int H = b.height, W = b.width;
  for (int c = 0; c < b.channels; ++c)
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w) o.at(c, h, W - 1 - w) = b.at(c, h, w);
  for (int fh = 0; fh < H; ++fh)
    for (int fw = 0; fw < W; ++fw) {
      for (int w = 0; w < W; ++w)
        o.pi[policy_location(W, H, fh, W - 1 - fw, false, W - 1 - w)] = b.pi[policy_location(W, H, fh, fw, false, w)];
      for (int h = 0; h < H; ++h)
        o.pi[policy_location(W, H, fh, W - 1 - fw, true, h)] = b.pi[policy_location(W, H, fh, fw, true, h)];
    }
  return o;
}

// tafl_helper.h:55-137.  The reference walks 4-cycles of the quarter board; every assignment there
// is an instance of  out(h, w) = base(H-1-w, h)  and, for the moves of the piece on (h, w),
//   out.pi(h, w, row-slide to column x) = base.pi(H-1-w, h, column-slide to row W-1-x)
//   out.pi(h, w, column-slide to row y) = base.pi(H-1-w, h, row-slide to column y)
// (lines 77-83, 92-124 with (base_h, base_w) running over one representative per cycle, plus the
// centre cell, lines 84-87, 126-135).  Written here per cell.
inline Sample rot90_clockwise(const Sample& b) {
  Sample o = blank_like(b);
  const int H = b.height, W = b.width;   // square boards only (assert at tafl_helper.h:59)
  for (int c = 0; c < b.channels; ++c)
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w) o.at(c, h, w) = b.at(c, H - 1 - w, h);
  for (int h = 0; h < H; ++h)
    for (int w = 0; w < W; ++w) {
      for (int x = 0; x < W; ++x)
        o.pi[policy_location(W, H, h, w, false, x)] = b.pi[policy_location(W, H, H - 1 - w, h, true, W - 1 - x)];
      for (int y = 0; y < H; ++y)
        o.pi[policy_location(W, H, h, w, true, y)] = b.pi[policy_location(W, H, H - 1 - w, h, false, y)];
    }
  return o;
}

// tafl_helper.h:139-149: {base, r, r^2, r^3, m(base), m(r), m(r^2), m(r^3)}
inline std::vector<Sample> eight_sym(const Sample& b) {
  std::vector<Sample> out{b};
  for (int i = 0; i < 3; ++i) out.push_back(rot90_clockwise(out[i]));
  for (int i = 0; i < 4; ++i) out.push_back(mirror_width(out[i]));
  return out;
}

// connect4_gs.cc:151-170: {base, column mirror}; pi is one entry per column.
inline std::vector<Sample> connect4_symmetries(const Sample& b) {
  Sample m = blank_like(b);
  const int H = b.height, W = b.width;
  for (int c = 0; c < b.channels; ++c)
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w) m.at(c, h, w) = b.at(c, h, W - 1 - w);
  for (int w = 0; w < W; ++w) m.pi[w] = b.pi[W - 1 - w];
  return {b, m};
}

}  // namespace orc
