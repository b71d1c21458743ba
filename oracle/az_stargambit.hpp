// ORACLE — TEST INFRASTRUCTURE ONLY (see az_rng.hpp header for the usage rule).
// CPU restatement of StarGambitGS<Config> and StarGambitUnifiedGS
// (reference star_gambit_gs.h:22-60, 483-887; star_gambit_gs.cc).  Units are kept as the
// reference keeps them — a growing list of 9-field records, dead units included — and every
// query re-derives the occupied hexes from that list; the device engine uses lane-resident
// packed units and 169-cell occupancy bitboards instead, so the two share no representation.
//
// Build-defined (the reference's source is unseedable): the variant of a new game.  The reference
// draws it from a thread_local std::mt19937 seeded by std::random_device
// (star_gambit_gs.cc:2357-2362); here randomize_start_from(coin) takes one uniform01 draw of the
// slot's coin stream when the game is not pinned (rule in pick_variant below).
#pragma once
#include <array>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "az_games.hpp"
#include "az_rng.hpp"

namespace orc {
namespace sg {

struct Hex { int q, r; };
inline bool operator==(const Hex& a, const Hex& b) { return a.q == b.q && a.r == b.r; }

// star_gambit_gs.h:251-258 — E, NE, NW, W, SW, SE
static const Hex kDir[6] = {{1, 0}, {1, -1}, {0, -1}, {-1, 0}, {-1, 1}, {0, 1}};
inline int rot(int d, int steps) { return (d + steps + 6) % 6; }      // rotate_direction, :267-269
inline int opp(int d) { return (d + 3) % 6; }                           // OPPOSITE_DIRECTION, :261
inline Hex step(const Hex& h, int d) { return Hex{h.q + kDir[d].q, h.r + kDir[d].r}; }   // hex_neighbor
inline bool in_bounds(const Hex& h, int side) {                         // star_gambit_gs.cc:40-44
  return std::abs(h.q) <= side && std::abs(h.r) <= side && std::abs(-h.q - h.r) <= side;
}

enum : uint8_t { FIGHTER = 0, CRUISER = 1, DREAD = 2, PORTAL = 3 };
static const int kMaxHp[4] = {3, 4, 6, 5};        // star_gambit_gs.h:73-76
static const int kMaxMoves[4] = {2, 1, 1, 0};     // :79-82
static const int kCannons[4] = {1, 3, 4, 0};      // :122-124
constexpr int kMaxTurns = 200;                    // :85

struct Unit {   // star_gambit_gs.h:359-371
  uint8_t type, player, slot, hp, facing;
  int8_t q, r;
  uint8_t moves_left, cannons_fired;
  bool alive() const { return hp > 0; }
};

// star_gambit_gs.h:22-60: board side, starting (= maximum) fighters / cruisers / dreadnoughts
struct Config { int side; int start[3]; };
static const Config kConfig[4] = {{5, {3, 1, 0}}, {5, {4, 0, 1}}, {5, {3, 2, 1}}, {6, {4, 3, 2}}};

// get_unit_hexes, star_gambit_gs.cc:88-120 (anchor first; the portal entry is the anchor alone)
inline std::vector<Hex> unit_hexes(int type, const Hex& a, int facing) {
  std::vector<Hex> h{a};
  if (type == CRUISER) h.push_back(step(a, opp(facing)));
  if (type == DREAD) { h.push_back(step(a, rot(opp(facing), 1))); h.push_back(step(a, opp(facing))); }
  return h;
}
// get_portal_hexes, star_gambit_gs.cc:122-141
inline std::vector<Hex> portal_hexes(int player, int side) {
  if (player == 0) return {{0, side}, {1, side - 1}, {-1, side}};
  return {{0, -side}, {-1, -side + 1}, {1, -side}};
}
inline Hex deploy_hex(int player, int side) { return player == 0 ? Hex{0, side - 1} : Hex{0, -side + 1}; }   // :143-152
inline int dread_anchor_dir(int player, int facing) {   // :157-169
  static const int p0[6] = {1, 2, 2, 3, -1, -1}, p1[6] = {0, -1, -1, 4, 5, 5};
  return player == 0 ? p0[facing] : p1[facing];
}
inline bool deploy_facing_ok(int type, int player, int facing) {   // get_valid_deploy_facings, :171-195
  if (type == DREAD) return player == 0 ? facing <= 3 : (facing == 0 || facing >= 3);
  return player == 0 ? (facing >= 1 && facing <= 3) : (facing == 4 || facing == 5 || facing == 0);
}
struct Cannon { int dir_offset, src; };
inline std::vector<Cannon> cannon_info(int type) {   // :201-231
  if (type == FIGHTER) return {{0, 0}};
  if (type == CRUISER) return {{1, 0}, {0, 0}, {-1, 0}};
  if (type == DREAD) return {{1, 2}, {1, 0}, {0, 0}, {0, 1}};
  return {};
}

// One game of one configuration: StarGambitGS<Config>, in its OWN action space
// (ActionSpace<Config>, star_gambit_gs.h:483-590): dim = 2*side+1, spatial = dim*dim*10, then 18 deploys, then end turn.
struct Inner {
  Config cfg;
  std::vector<Unit> units;
  uint8_t reserves[2][4];
  uint8_t player = 0;
  uint32_t turn = 1;
  bool acted = false, over = false;
  int8_t winner = -1;
  std::vector<uint64_t> history;

  int dim() const { return 2 * cfg.side + 1; }
  int spatial() const { return dim() * dim() * 10; }
  int deploy_offset() const { return spatial(); }
  int end_offset() const { return spatial() + 18; }
  int num_moves() const { return spatial() + 19; }

  explicit Inner(const Config& c) : cfg(c) {   // star_gambit_gs.cc:251-290
    for (int p = 0; p < 2; ++p) {
      for (int t = 0; t < 3; ++t) reserves[p][t] = static_cast<uint8_t>(c.start[t]);
      reserves[p][3] = 0;
    }
    for (int p = 0; p < 2; ++p) {
      const Hex a = portal_hexes(p, c.side)[0];
      units.push_back(Unit{PORTAL, static_cast<uint8_t>(p), 0, 5, static_cast<uint8_t>(p == 0 ? 2 : 5),
                           static_cast<int8_t>(a.q), static_cast<int8_t>(a.r), 0, 0});
    }
    history.push_back(position_hash());
  }

  bool turn_one() const { return turn == 1 || turn == 2; }   // star_gambit_gs.h:653
  std::vector<Hex> hexes_of(const Unit& u) const {
    return u.type == PORTAL ? portal_hexes(u.player, cfg.side) : unit_hexes(u.type, Hex{u.q, u.r}, u.facing);
  }
  std::vector<Hex> all_occupied() const {   // :371-390
    std::vector<Hex> o;
    for (const auto& u : units) if (u.alive()) for (const auto& h : hexes_of(u)) o.push_back(h);
    return o;
  }
  bool occupied(const Hex& h, int exclude) const {   // is_hex_occupied, :392-411
    for (size_t i = 0; i < units.size(); ++i) {
      if (static_cast<int>(i) == exclude || !units[i].alive()) continue;
      for (const auto& uh : hexes_of(units[i])) if (uh == h) return true;
    }
    return false;
  }
  int unit_at(const Hex& h) const {   // find_unit_at_hex, :413-433
    for (size_t i = 0; i < units.size(); ++i) {
      if (!units[i].alive()) continue;
      for (const auto& uh : hexes_of(units[i])) if (uh == h) return static_cast<int>(i);
    }
    return -1;
  }
  bool collides(const std::vector<Hex>& hs, int exclude) const {   // would_collide, :435-442
    for (const auto& h : hs) if (!in_bounds(h, cfg.side) || occupied(h, exclude)) return true;
    return false;
  }
  int find_slot(int pl, int type, int slot) const {   // find_unit_by_slot (alive only), :340-358
    for (size_t i = 0; i < units.size(); ++i)
      if (units[i].player == pl && units[i].type == type && units[i].slot == slot && units[i].alive()) return static_cast<int>(i);
    return -1;
  }
  int next_slot(int pl, int type) const {   // :360-369 (dead units count)
    int m = -1;
    for (const auto& u : units) if (u.player == pl && u.type == type && u.slot > m) m = u.slot;
    return m + 1;
  }

  // ---- movement (compute_*_move, :448-600).  `dir` is the per-type move code of the reference's enums.
  struct Move { Hex anchor; int facing; bool valid; };
  Move fighter_move(const Unit& u, int dir) const {
    Move m{{0, 0}, 0, false};
    if (dir < 0 || dir > 2) return m;
    const int d = dir == 0 ? u.facing : dir == 1 ? rot(u.facing, 1) : rot(u.facing, -1);
    m.anchor = step(Hex{u.q, u.r}, d); m.facing = d; m.valid = in_bounds(m.anchor, cfg.side);
    return m;
  }
  Move cruiser_move(const Unit& u, int dir) const {   // 0 rotate-left, 1 fwd-left, 2 forward, 3 fwd-right, 4 rotate-right
    Move m{{0, 0}, 0, false};
    if (dir < 0 || dir > 4) return m;
    const Hex a{u.q, u.r};
    if (dir == 0 || dir == 4) {
      const Hex rear = step(a, opp(u.facing));
      m.facing = rot(u.facing, dir == 0 ? 1 : -1);
      m.anchor = step(rear, m.facing);
    } else {
      m.facing = dir == 1 ? rot(u.facing, 1) : dir == 2 ? u.facing : rot(u.facing, -1);
      m.anchor = step(a, m.facing);
    }
    m.valid = true;
    for (const auto& h : unit_hexes(CRUISER, m.anchor, m.facing)) if (!in_bounds(h, cfg.side)) { m.valid = false; break; }
    return m;
  }
  Move dread_move(const Unit& u, int dir) const {     // 0 pivot-left, 1 slide fwd-left, 2 slide fwd-right, 3 pivot-right
    Move m{{0, 0}, 0, false};
    if (dir < 0 || dir > 3) return m;
    const Hex a{u.q, u.r};
    const int rear = opp(u.facing);
    if (dir == 0) {
      const Hex pivot = step(a, rear);
      m.anchor = step(pivot, rot(opp(rear), 1)); m.facing = rot(u.facing, 1);
    } else if (dir == 1) {
      m.anchor = step(a, rot(u.facing, 1)); m.facing = u.facing;
    } else if (dir == 2) {
      m.anchor = step(a, u.facing); m.facing = u.facing;
    } else {
      const int rr = rot(rear, 1);
      const Hex pivot = step(a, rr);
      m.anchor = step(pivot, rot(opp(rr), -1)); m.facing = rot(u.facing, -1);
    }
    m.valid = true;
    for (const auto& h : unit_hexes(DREAD, m.anchor, m.facing)) if (!in_bounds(h, cfg.side)) { m.valid = false; break; }
    return m;
  }
  Move any_move(const Unit& u, int dir) const {
    return u.type == FIGHTER ? fighter_move(u, dir) : u.type == CRUISER ? cruiser_move(u, dir) : dread_move(u, dir);
  }
  bool move_valid(int ui, int dir) const {   // is_*_move_valid, :606-663
    const Unit& u = units[ui];
    const Move m = any_move(u, dir);
    if (!m.valid) return false;
    return !collides(unit_hexes(u.type, m.anchor, m.facing), ui);
  }
  // has_target_in_range, :669-713
  bool has_target(const Unit& u, int cannon) const {
    const auto cs = cannon_info(u.type);
    if (cannon >= static_cast<int>(cs.size())) return false;
    const auto hs = hexes_of(u);
    if (cs[cannon].src >= static_cast<int>(hs.size())) return false;
    const Hex src = hs[cs[cannon].src];
    const int d = rot(u.facing, cs[cannon].dir_offset);
    const auto occ = all_occupied();
    for (int range = 1; range <= 2; ++range) {
      Hex t = src;
      for (int i = 0; i < range; ++i) t = step(t, d);
      if (!in_bounds(t, cfg.side)) continue;
      if (!line_of_sight(src, d, range, occ)) break;
      const int ti = unit_at(t);
      if (ti >= 0 && units[ti].player != u.player) return true;
    }
    return false;
  }
  static bool line_of_sight(const Hex& from, int d, int dist, const std::vector<Hex>& occ) {   // :233-245
    Hex c = from;
    for (int i = 1; i < dist; ++i) {
      c = step(c, d);
      for (const auto& o : occ) if (o == c) return false;
    }
    return true;
  }
  bool fire_valid(const Unit& u, int cannon) const {   // :715-727
    if (!u.alive() || u.player != player) return false;
    if (cannon < 0 || cannon >= kCannons[u.type]) return false;
    if (u.cannons_fired & (1 << cannon)) return false;
    return has_target(u, cannon);
  }
  Hex deploy_anchor(int type, int facing) const {   // :742-757 / :1056-1069
    const Hex dh = deploy_hex(player, cfg.side);
    if (type == DREAD) return step(dh, dread_anchor_dir(player, facing));
    if (type == CRUISER) return step(dh, facing);
    return dh;
  }
  bool deploy_valid(int type, int facing) const {   // :729-770
    if (type == PORTAL) return false;
    if (reserves[player][type] == 0) return false;
    if (!deploy_facing_ok(type, player, facing)) return false;
    const auto occ = all_occupied();
    for (const auto& h : unit_hexes(type, deploy_anchor(type, facing), facing)) {
      if (!in_bounds(h, cfg.side)) return false;
      for (const auto& o : occ) if (o == h) return false;
    }
    return true;
  }
  bool end_turn_valid() const { return !turn_one() && acted; }   // :772-778

  int encode(int row, int col, int slot) const {   // valid_moves' encode_action, :799-811
    if (player == 1) { row = dim() - 1 - row; col = dim() - 1 - col; }
    return (row * dim() + col) * 10 + slot;
  }
  // valid_moves, :784-923.  SpatialAction slots: 0 fwd, 1 fwd-left, 2 fwd-right, 3 rotate-left, 4 rotate-right,
  // 5 fire fwd, 6 fire fwd-left, 7 fire fwd-right, 8 fire rear-left, 9 fire rear-right (star_gambit_gs.h:454-465)
  void valid_moves(uint8_t* out) const {
    std::memset(out, 0, num_moves());
    if (over) return;
    if (!turn_one()) {
      for (size_t i = 0; i < units.size(); ++i) {
        const Unit& u = units[i];
        if (u.player != player || !u.alive() || u.type == PORTAL) continue;
        const int row = u.q + cfg.side, col = u.r + cfg.side;   // hex_to_2d, :276-279
        const int ui = static_cast<int>(i);
        if (u.moves_left > 0) {
          if (u.type == FIGHTER) {
            if (move_valid(ui, 0)) out[encode(row, col, 0)] = 1;
            if (move_valid(ui, 1)) out[encode(row, col, 1)] = 1;
            if (move_valid(ui, 2)) out[encode(row, col, 2)] = 1;
          } else if (u.type == CRUISER) {
            if (move_valid(ui, 2)) out[encode(row, col, 0)] = 1;
            if (move_valid(ui, 1)) out[encode(row, col, 1)] = 1;
            if (move_valid(ui, 3)) out[encode(row, col, 2)] = 1;
            if (move_valid(ui, 0)) out[encode(row, col, 3)] = 1;
            if (move_valid(ui, 4)) out[encode(row, col, 4)] = 1;
          } else {
            if (move_valid(ui, 1)) out[encode(row, col, 1)] = 1;
            if (move_valid(ui, 2)) out[encode(row, col, 2)] = 1;
            if (move_valid(ui, 0)) out[encode(row, col, 3)] = 1;
            if (move_valid(ui, 3)) out[encode(row, col, 4)] = 1;
          }
        }
        if (u.type == FIGHTER) {
          if (fire_valid(u, 0)) out[encode(row, col, 5)] = 1;
        } else if (u.type == CRUISER) {
          if (fire_valid(u, 1)) out[encode(row, col, 5)] = 1;
          if (fire_valid(u, 0)) out[encode(row, col, 6)] = 1;
          if (fire_valid(u, 2)) out[encode(row, col, 7)] = 1;
        } else {
          if (fire_valid(u, 1)) out[encode(row, col, 6)] = 1;
          if (fire_valid(u, 2)) out[encode(row, col, 7)] = 1;
          if (fire_valid(u, 0)) out[encode(row, col, 8)] = 1;
          if (fire_valid(u, 3)) out[encode(row, col, 9)] = 1;
        }
      }
    }
    for (int t = 0; t < 3; ++t)
      for (int f = 0; f < 6; ++f)
        if (deploy_valid(t, f)) out[deploy_offset() + t * 6 + (player == 1 ? (f + 3) % 6 : f)] = 1;
    if (end_turn_valid()) out[end_offset()] = 1;
  }

  // ---- execution ------------------------------------------------------------------------------------
  void exec_move(int type, int slot, int dir) {   // execute_*_move, :929-972
    const int ui = find_slot(player, type, slot);
    if (ui < 0) return;
    const Move m = any_move(units[ui], dir);
    if (!m.valid) return;
    Unit& u = units[ui];
    u.q = static_cast<int8_t>(m.anchor.q); u.r = static_cast<int8_t>(m.anchor.r); u.facing = static_cast<uint8_t>(m.facing);
    u.moves_left--;
    acted = true;
  }
  void exec_fire(int ui, int cannon) {   // execute_fire, :978-1045
    units[ui].cannons_fired |= static_cast<uint8_t>(1 << cannon);
    acted = true;
    const Unit u = units[ui];
    const auto cs = cannon_info(u.type);
    if (cannon >= static_cast<int>(cs.size())) return;
    const auto hs = hexes_of(u);
    if (cs[cannon].src >= static_cast<int>(hs.size())) return;
    const Hex src = hs[cs[cannon].src];
    const int d = rot(u.facing, cs[cannon].dir_offset);
    const auto occ = all_occupied();
    for (int range = 1; range <= 2; ++range) {
      Hex t = src;
      for (int i = 0; i < range; ++i) t = step(t, d);
      if (!in_bounds(t, cfg.side)) continue;
      if (!line_of_sight(src, d, range, occ)) break;
      const int ti = unit_at(t);
      if (ti >= 0 && ti != ui) {
        const int dmg = range == 1 ? 2 : 1;
        Unit& tu = units[ti];   // apply_damage, :1302-1311
        bool destroyed = false;
        if (dmg >= tu.hp) { tu.hp = 0; destroyed = true; } else tu.hp = static_cast<uint8_t>(tu.hp - dmg);
        if (destroyed) check_game_end();
        return;
      }
    }
  }
  void exec_deploy(int type, int facing) {   // :1051-1087
    history.clear();
    const Hex a = deploy_anchor(type, facing);
    Unit nu{static_cast<uint8_t>(type), player, static_cast<uint8_t>(next_slot(player, type)), static_cast<uint8_t>(kMaxHp[type]),
            static_cast<uint8_t>(facing), static_cast<int8_t>(a.q), static_cast<int8_t>(a.r), 0,
            static_cast<uint8_t>((1 << kCannons[type]) - 1)};
    units.push_back(nu);
    reserves[player][type]--;
    exec_end_turn();
  }
  bool check_repetition() {   // :1246-1261
    history.push_back(position_hash());
    int count = 0;
    for (auto h : history) if (h == history.back()) ++count;
    if (count >= 3) { over = true; winner = 2; return true; }
    return false;
  }
  void exec_end_turn() {   // :1263-1290
    player = static_cast<uint8_t>(1 - player);
    ++turn;
    acted = false;
    if (turn > static_cast<uint32_t>(kMaxTurns)) { over = true; winner = 2; return; }
    if (check_repetition()) return;
    for (auto& u : units)   // reset_turn_state, :1292-1300
      if (u.player == player && u.alive()) { u.moves_left = static_cast<uint8_t>(kMaxMoves[u.type]); u.cannons_fired = 0; }
    std::vector<uint8_t> v(num_moves());
    valid_moves(v.data());
    uint8_t sum = 0;   // Vector<uint8_t>::sum() wraps like the element type
    for (auto x : v) sum = static_cast<uint8_t>(sum + x);
    if (sum == 0) { over = true; winner = static_cast<int8_t>(1 - player); }
  }
  void check_game_end() {   // :1313-1345
    for (const auto& u : units)
      if (u.type == PORTAL && u.hp == 0) { over = true; winner = static_cast<int8_t>(1 - u.player); return; }
    for (int p = 0; p < 2; ++p) {
      bool ships = false, res = false;
      for (const auto& u : units) if (u.player == p && u.alive() && u.type != PORTAL) { ships = true; break; }
      for (int t = 0; t < 3; ++t) if (reserves[p][t] > 0) { res = true; break; }
      if (!ships && !res) { over = true; winner = static_cast<int8_t>(1 - p); return; }
    }
  }
  uint64_t position_hash() const {   // compute_position_hash, :1365-1382
    uint64_t h = static_cast<uint64_t>(player) * 0x9e3779b97f4a7c15ULL;
    for (const auto& u : units) {
      if (!u.alive()) continue;
      const uint64_t uh = static_cast<uint64_t>(u.type) ^ (static_cast<uint64_t>(u.player) << 8) ^ (static_cast<uint64_t>(u.hp) << 12) ^
                          (static_cast<uint64_t>(u.facing) << 20) ^ (static_cast<uint64_t>(u.q + 10) << 28) ^
                          (static_cast<uint64_t>(u.r + 10) << 36);
      h ^= uh * 0x517cc1b727220a95ULL;
    }
    return h;
  }

  void play_move(uint32_t move) {   // :1093-1238
    if (move < static_cast<uint32_t>(deploy_offset())) {
      int slot = move % 10, pos = move / 10, col = pos % dim(), row = pos / dim();
      if (player == 1) { row = dim() - 1 - row; col = dim() - 1 - col; }
      const int q = row - cfg.side, r = col - cfg.side;
      int ui = -1;
      for (size_t i = 0; i < units.size(); ++i) {
        const Unit& u = units[i];
        if (u.player == player && u.alive() && u.q == q && u.r == r && u.type != PORTAL) { ui = static_cast<int>(i); break; }
      }
      if (ui < 0) return;
      const int type = units[ui].type, us = units[ui].slot;
      switch (slot) {
        case 0: if (type == FIGHTER) exec_move(FIGHTER, us, 0); else if (type == CRUISER) exec_move(CRUISER, us, 2); break;
        case 1: exec_move(type, us, 1); break;
        case 2: exec_move(type, us, type == CRUISER ? 3 : 2); break;
        case 3: if (type == CRUISER) exec_move(CRUISER, us, 0); else if (type == DREAD) exec_move(DREAD, us, 0); break;
        case 4: if (type == CRUISER) exec_move(CRUISER, us, 4); else if (type == DREAD) exec_move(DREAD, us, 3); break;
        case 5: if (type == FIGHTER) exec_fire(ui, 0); else if (type == CRUISER) exec_fire(ui, 1); break;
        case 6: if (type == CRUISER) exec_fire(ui, 0); else if (type == DREAD) exec_fire(ui, 1); break;
        case 7: if (type == CRUISER) exec_fire(ui, 2); else if (type == DREAD) exec_fire(ui, 2); break;
        case 8: if (type == DREAD) exec_fire(ui, 0); break;
        case 9: if (type == DREAD) exec_fire(ui, 3); break;
      }
      check_repetition();
    } else if (move < static_cast<uint32_t>(end_offset())) {
      const int rel = static_cast<int>(move) - deploy_offset();
      int type = rel / 6, facing = rel % 6;
      if (player == 1) facing = (facing + 3) % 6;
      exec_deploy(type, facing);
    } else {
      exec_end_turn();
    }
  }
  bool scores(float* out) const {   // :1347-1363
    if (!over) return false;
    out[0] = out[1] = out[2] = 0.0f;
    if (winner == 2) out[2] = 1.0f; else if (winner >= 0 && winner < 2) out[winner] = 1.0f;
    return true;
  }

  // canonicalized, :1384-1669: 32 x dim x dim
  void canonicalized(float* out) const {
    const int D = dim(), S = cfg.side;
    std::memset(out, 0, sizeof(float) * 32 * D * D);
    const bool p1 = player == 1;
    const int me = player, other = 1 - player;
    auto at = [&](int ch, int row, int col) -> float& { return out[(ch * D + row) * D + col]; };
    auto set_hex = [&](int ch, const Hex& h, float v) {
      const Hex x = p1 ? Hex{-h.q, -h.r} : h;
      at(ch, x.q + S, x.r + S) = v;
    };
    auto broadcast = [&](int ch, float v) {
      for (int r = -S; r <= S; ++r) for (int q = -S; q <= S; ++q) if (in_bounds(Hex{q, r}, S)) at(ch, q + S, r + S) = v;
    };
    broadcast(0, 1.0f);
    for (const auto& u : units) {
      if (!u.alive()) continue;
      const int ch = 1 + (u.player == me ? 0 : 4) + u.type;
      for (const auto& h : hexes_of(u)) set_hex(ch, h, 1.0f);
    }
    for (const auto& u : units) {
      if (!u.alive() || u.type == PORTAL) continue;
      const int f = p1 ? (u.facing + 3) % 6 : u.facing;
      for (const auto& h : unit_hexes(u.type, Hex{u.q, u.r}, u.facing)) set_hex(9 + f, h, 1.0f);
    }
    for (const auto& u : units) {
      if (!u.alive()) continue;
      const float v = static_cast<float>(u.hp) / static_cast<float>(kMaxHp[u.type]);
      for (const auto& h : hexes_of(u)) set_hex(15, h, v);
    }
    for (const auto& u : units) {
      if (!u.alive() || u.type == PORTAL) continue;
      const float mm = static_cast<float>(kMaxMoves[u.type]);
      const float v = mm > 0 ? static_cast<float>(u.moves_left) / mm : 0.0f;
      for (const auto& h : unit_hexes(u.type, Hex{u.q, u.r}, u.facing)) set_hex(16, h, v);
    }
    for (const auto& u : units) {
      if (!u.alive() || u.type == PORTAL) continue;
      static const int c_slot[3] = {1, 0, 2}, d_slot[4] = {3, 1, 2, 4};   // cannon index -> observation slot, :1546-1568
      for (int c = 0; c < kCannons[u.type]; ++c) {
        const int slot = u.type == FIGHTER ? 0 : u.type == CRUISER ? c_slot[c] : d_slot[c];
        if (!((u.cannons_fired >> c) & 1)) set_hex(17 + slot, Hex{u.q, u.r}, 1.0f);
      }
    }
    broadcast(22, acted ? 1.0f : 0.0f);
    {
      const uint64_t cur = position_hash();
      int rep = 0;
      for (auto h : history) if (h == cur) ++rep;
      broadcast(23, rep == 0 ? 0.0f : rep == 1 ? 0.5f : 1.0f);
    }
    for (int side = 0; side < 2; ++side) {
      const int p = side == 0 ? me : other;
      for (int t = 0; t < 3; ++t)
        broadcast(24 + side * 3 + t, cfg.start[t] > 0 ? static_cast<float>(reserves[p][t]) / static_cast<float>(cfg.start[t]) : 0.0f);
    }
    for (int side = 0; side < 2; ++side) {
      const int ui = find_slot(side == 0 ? me : other, PORTAL, 0);
      broadcast(30 + side, ui >= 0 ? static_cast<float>(units[ui].hp) / 5.0f : 0.0f);
    }
  }

  // to_bytes / from_bytes, :2253-2338
  std::string to_bytes() const {
    std::string out;
    auto put = [&](const void* p, size_t n) { out.append(static_cast<const char*>(p), n); };
    const uint32_t n = static_cast<uint32_t>(units.size());
    put(&n, 4);
    for (const auto& u : units) {
      const uint8_t b[9] = {u.type, u.player, u.slot, u.hp, u.facing, static_cast<uint8_t>(u.q), static_cast<uint8_t>(u.r), u.moves_left, u.cannons_fired};
      put(b, 9);
    }
    put(reserves, 8);
    out.push_back(static_cast<char>(player));
    put(&turn, 4);
    out.push_back(acted ? 1 : 0); out.push_back(over ? 1 : 0); out.push_back(static_cast<char>(winner));
    const uint32_t hl = static_cast<uint32_t>(history.size());
    put(&hl, 4);
    if (hl) put(history.data(), hl * 8);
    return out;
  }
  void from_bytes(const std::string& d) {
    size_t off = 0;
    auto get = [&](void* p, size_t n) {
      if (off + n > d.size()) throw std::runtime_error("StarGambitGS::from_bytes: short data");
      std::memcpy(p, &d[off], n); off += n;
    };
    uint32_t n = 0; get(&n, 4);
    units.clear();
    for (uint32_t i = 0; i < n; ++i) {
      uint8_t b[9]; get(b, 9);
      units.push_back(Unit{b[0], b[1], b[2], b[3], b[4], static_cast<int8_t>(b[5]), static_cast<int8_t>(b[6]), b[7], b[8]});
    }
    get(reserves, 8);
    uint8_t c; get(&c, 1); player = c;
    get(&turn, 4);
    get(&c, 1); acted = c != 0; get(&c, 1); over = c != 0; get(&c, 1); winner = static_cast<int8_t>(c);
    uint32_t hl = 0; get(&hl, 4);
    history.resize(hl);
    if (hl) get(history.data(), hl * 8);
    if (off != d.size()) throw std::runtime_error("StarGambitGS::from_bytes: trailing bytes");
  }
  bool same(const Inner& o) const {   // operator==, :297-319 (turn, flags and history do not take part)
    if (player != o.player || units.size() != o.units.size() || std::memcmp(reserves, o.reserves, 8) != 0 || acted != o.acted) return false;
    for (size_t i = 0; i < units.size(); ++i) {
      const Unit &a = units[i], &b = o.units[i];
      if (a.type != b.type || a.player != b.player || a.slot != b.slot || a.hp != b.hp || a.facing != b.facing || a.q != b.q ||
          a.r != b.r || a.moves_left != b.moves_left || a.cannons_fired != b.cannons_fired) return false;
    }
    return true;
  }
  // fields of hash(), :321-338: player, acted, every unit record, reserves
  uint64_t key(uint64_t k) const {
    k = mix64(k ^ (static_cast<uint64_t>(player) | (static_cast<uint64_t>(acted) << 8)));
    for (const auto& u : units) {
      const uint64_t w = static_cast<uint64_t>(u.type) | (static_cast<uint64_t>(u.player) << 2) | (static_cast<uint64_t>(u.slot) << 3) |
                         (static_cast<uint64_t>(u.hp) << 6) | (static_cast<uint64_t>(u.facing) << 9) |
                         (static_cast<uint64_t>(u.q + 6) << 12) | (static_cast<uint64_t>(u.r + 6) << 16) |
                         (static_cast<uint64_t>(u.moves_left) << 20) | (static_cast<uint64_t>(u.cannons_fired) << 22);
      k = mix64(k ^ w);
    }
    uint64_t rw = 0;
    for (int p = 0; p < 2; ++p) for (int t = 0; t < 4; ++t) rw |= static_cast<uint64_t>(reserves[p][t]) << (4 * (p * 4 + t));
    return mix64(k ^ rw);
  }
};

}  // namespace sg

// StarGambitGS<Config> as a Game in its own action space (what star_gambit_gs_test.cc exercises).
struct StarGambit final : Game {
  int variant;
  sg::Inner in;
  explicit StarGambit(int v) : variant(v), in(sg::kConfig[v]) {}
  std::unique_ptr<Game> copy() const override { return std::make_unique<StarGambit>(*this); }
  uint8_t current_player() const override { return in.player; }
  uint32_t current_turn() const override { return in.turn; }
  uint32_t num_moves() const override { return static_cast<uint32_t>(in.num_moves()); }
  uint8_t num_players() const override { return 2; }
  void valid_moves(uint8_t* out) const override { in.valid_moves(out); }
  void play_move(uint32_t m) override { in.play_move(m); }
  bool scores(float* out) const override { return in.scores(out); }
  void canonical_shape(int* c, int* h, int* w) const override { *c = 32; *h = in.dim(); *w = in.dim(); }
  void canonicalized(float* out) const override { in.canonicalized(out); }
  bool relative_values() const override { return true; }   // star_gambit_gs.h:628
  uint64_t key() const override { return in.key(0x5347ULL + static_cast<uint64_t>(variant)); }
};

// StarGambitUnifiedGS, star_gambit_gs.h:788-887, star_gambit_gs.cc:2375-2616: every variant on the 13 x 13 canvas,
// 36 channels, 1709 moves.
struct StarGambitUnified final : Game {
  static constexpr int UD = 13, US = 6, SD = 11, UNIFIED_SPATIAL = 1690, SMALL_SPATIAL = 1210, NUM_MOVES = 1709;
  float probs[4];
  int pinned;
  int variant;
  sg::Inner in;

  // build-defined variant draw (see the header): one uniform01 of `coin`, cumulative weights in variant order
  static int pick_variant(const float* p, Pcg32& coin) {
    const float total = ((p[0] + p[1]) + p[2]) + p[3];
    const float x = uniform01(coin) * total;
    int v = 0;
    float acc = p[0];
    while (v < 3 && x >= acc) { ++v; acc += p[v]; }
    return v;
  }
  StarGambitUnified(int pinned_variant, const float* p, int first_variant)
      : pinned(pinned_variant), variant(first_variant), in(sg::kConfig[first_variant]) {
    for (int i = 0; i < 4; ++i) probs[i] = p[i];
  }
  std::unique_ptr<Game> copy() const override { return std::make_unique<StarGambitUnified>(*this); }
  void randomize_start_from(Pcg32& coin) override {   // :2421-2425
    variant = (pinned >= 0 && pinned <= 3) ? pinned : pick_variant(probs, coin);
    in = sg::Inner(sg::kConfig[variant]);
  }
  uint8_t current_player() const override { return in.player; }
  uint32_t current_turn() const override { return in.turn; }
  uint32_t num_moves() const override { return NUM_MOVES; }
  uint8_t num_players() const override { return 2; }
  bool relative_values() const override { return true; }
  int num_variants() const override { return 4; }
  int get_variant_id() const override { return variant; }
  bool small() const { return variant != 3; }
  int to_unified(int a) const {   // :2522-2537
    if (!small()) return a;
    if (a < SMALL_SPATIAL) { const int slot = a % 10, pos = a / 10; return ((pos / SD + 1) * UD + (pos % SD + 1)) * 10 + slot; }
    return UNIFIED_SPATIAL + (a - SMALL_SPATIAL);
  }
  int from_unified(int a) const {   // :2539-2554
    if (!small()) return a;
    if (a < UNIFIED_SPATIAL) { const int slot = a % 10, pos = a / 10; return ((pos / UD - 1) * SD + (pos % UD - 1)) * 10 + slot; }
    return SMALL_SPATIAL + (a - UNIFIED_SPATIAL);
  }
  void valid_moves(uint8_t* out) const override {   // :2560-2572
    std::memset(out, 0, NUM_MOVES);
    std::vector<uint8_t> v(in.num_moves());
    in.valid_moves(v.data());
    for (int i = 0; i < in.num_moves(); ++i) if (v[i]) out[to_unified(i)] = 1;
  }
  void play_move(uint32_t m) override { in.play_move(static_cast<uint32_t>(from_unified(static_cast<int>(m)))); }
  bool scores(float* out) const override { return in.scores(out); }
  void canonical_shape(int* c, int* h, int* w) const override { *c = 36; *h = UD; *w = UD; }
  void canonicalized(float* out) const override {   // :2586-2616
    std::memset(out, 0, sizeof(float) * 36 * UD * UD);
    const int D = in.dim(), off = small() ? 1 : 0;
    std::vector<float> obs(32 * D * D);
    in.canonicalized(obs.data());
    for (int ch = 0; ch < 32; ++ch)
      for (int r = 0; r < D; ++r)
        for (int c = 0; c < D; ++c) out[(ch * UD + r + off) * UD + c + off] = obs[(ch * D + r) * D + c];
    for (int r = 0; r < UD; ++r)
      for (int c = 0; c < UD; ++c)
        if (out[r * UD + c] > 0.5f) out[((32 + variant) * UD + r) * UD + c] = 1.0f;
  }
  uint64_t key() const override { return in.key(0x53475500ULL + static_cast<uint64_t>(variant)); }   // :2399-2402
  std::string to_bytes() const {   // :2451-2465
    std::string out;
    out.append(reinterpret_cast<const char*>(probs), 16);
    const int32_t pv = pinned;
    out.append(reinterpret_cast<const char*>(&pv), 4);
    out.push_back(static_cast<char>(variant));
    const std::string ib = in.to_bytes();
    const uint32_t n = static_cast<uint32_t>(ib.size());
    out.append(reinterpret_cast<const char*>(&n), 4);
    out.append(ib);
    return out;
  }
};

}  // namespace orc
