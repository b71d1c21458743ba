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
#include <algorithm>
#include <cstdlib>
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

// StarGambitUnifiedGS::symmetries (star_gambit_gs.cc:2623-2727): {base, mirror about the NW axis} on the 13 x 13 canvas,
// restated as the reference writes it - a SCATTER of every source element into a zeroed sample - so that it shares no
// formulation with the device kernel's gather.
inline std::vector<Sample> stargambit_unified_symmetries(const Sample& b) {
  constexpr int BD = 13, BS = 6, NCH = 36, SPATIAL = 1690, NM = 1709;
  static const int dir_map[6] = {4, 3, 2, 1, 0, 5}, deploy_d[6] = {3, 2, 1, 0, 5, 4}, slot_map[10] = {0, 2, 1, 4, 3, 5, 7, 6, 9, 8},
                   cannon_map[5] = {0, 2, 1, 4, 3};
  Sample m = b;
  std::fill(m.canonical.begin(), m.canonical.end(), 0.0f);
  std::fill(m.pi.begin(), m.pi.end(), 0.0f);
  auto at = [&](std::vector<float>& c, int ch, int r, int col) -> float& { return c[(static_cast<size_t>(ch) * BD + r) * BD + col]; };
  std::vector<float> base = b.canonical;
  for (int ch = 0; ch < NCH; ++ch)
    for (int row = 0; row < BD; ++row)
      for (int col = 0; col < BD; ++col) {
        const int nr = BD - 1 - row, nc = row + col - BS;
        if (nc >= 0 && nc < BD) at(m.canonical, ch, nr, nc) = at(base, ch, row, col);
      }
  std::vector<float> tmp = m.canonical;
  for (int d = 0; d < 6; ++d)
    for (int r = 0; r < BD; ++r) for (int c = 0; c < BD; ++c) at(m.canonical, 9 + d, r, c) = at(tmp, 9 + dir_map[d], r, c);
  for (int k = 0; k < 5; ++k)
    for (int r = 0; r < BD; ++r) for (int c = 0; c < BD; ++c) at(m.canonical, 17 + k, r, c) = at(tmp, 17 + cannon_map[k], r, c);
  for (int a = 0; a < SPATIAL; ++a) {
    const int slot = a % 10, pos = a / 10, row = pos / BD, col = pos % BD;
    const int q = row - BS, r = col - BS, t = -q - r;
    if (std::abs(q) > BS || std::abs(r) > BS || std::abs(t) > BS) { m.pi[a] = b.pi[a]; continue; }
    const int nr = BD - 1 - row, nc = row + col - BS;
    if (nc >= 0 && nc < BD) m.pi[(nr * BD + nc) * 10 + slot_map[slot]] = b.pi[a];
  }
  for (int d = 0; d < 18; ++d) {
    const int t = d / 6, f = d % 6;
    m.pi[SPATIAL + t * 6 + (t == 2 ? deploy_d[f] : dir_map[f])] = b.pi[SPATIAL + d];
  }
  m.pi[NM - 1] = b.pi[NM - 1];
  return {b, m};
}

}  // namespace orc
