// Device-side rules of StarGambitUnifiedGS (gfx950) - reference star_gambit_gs.h:483-887, star_gambit_gs.cc.
//
// One wavefront owns one game.  The reference keeps a vector of 9-byte unit records and re-derives every unit's hexes for
// every query (valid_moves is its heaviest function, star_gambit_gs.cc:785-927).  Here:
//   * LANE i HOLDS UNIT i (the reference's units_[i], dead units included, <= 20) as one packed 32-bit word; the scalars of
//     the game (player, turn, flags, reserves, variant) are wave-uniform registers,
//   * occupancy is a pair of 169-bit boards (one per player) over the 13 x 13 canvas, rebuilt from the lanes with LDS
//     atomics whenever it is needed; cell = (q + 6) * 13 + (r + 6) is at once the reference's hex_to_2d position of the
//     UNIFIED canvas, so the small variants need no +1 remap of rows, columns or action indices,
//   * legal moves: every lane tests the <= 10 actions of its own unit against the boards and sets bits of a dense
//     1709-bit map in LDS; lanes 0..17 test the 18 deploys; the tree kernel then enumerates the set bits in ascending
//     move order (what the reference's dense mask implies for Node::add_children),
//   * a move is executed by the lane that owns the unit; hits are found with a ballot over "my unit covers the cell",
//   * the position history (star_gambit_gs.cc:1246-1261) holds the REFERENCE'S OWN 64-bit position hash
//     (compute_position_hash, :1365-1382), so threefold repetition is bit-identical, collisions included; the list itself
//     belongs to the caller (HBM), handed in through a small functor.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev_rng.h"

namespace azmi {

struct SgScratch {   // per-wave LDS of the rules
  unsigned long long occ[6];        // [player][3 words] occupied cells
  unsigned long long vbits[27];     // dense valid-move map (1709 bits)
  unsigned long long masks[19 * 3]; // canonical planes built from units: 8 presence, 6 heading, 5 cannon
  uint32_t uword[24];               // unit words by index
  uint8_t cellunit[176];            // cell -> unit index (0xFF = empty)
};

struct StarGambit {
  static constexpr int kGameId = 4;
  static constexpr bool kRelative = true;        // relative_values(), star_gambit_gs.h:845
  static constexpr int P = 2;
  static constexpr int D = 13, CELLS = 169;      // UNIFIED_BOARD_DIM
  static constexpr int M = 1709;                 // UNIFIED_NUM_MOVES
  static constexpr int SPATIAL = 1690, DEPLOY0 = 1690, END_TURN = 1708;
  static constexpr int C = 36, H = 13, W = 13, CANON = C * H * W;
  static constexpr int MAXK = 128;               // > 56 unit actions + 10 deploys + end turn
  static constexpr int GAME_TURNS = 200;         // MAX_TURNS, star_gambit_gs.h:85
  // engine bound on the ACTIONS of one game (descent depth, pending sample rows, position-history entries): a turn has
  // several actions; 4096 is ~5x the longest of thousands of random games (overflow bit 8 / 2 beyond it)
  static constexpr int MAX_TURNS = 4096;
  static constexpr int GROUP = 64;
  static constexpr int MAX_UNITS = 20;           // 2 portals + 2 x (4 + 3 + 2) in the Battle configuration
  static constexpr int STATE_WORDS = 11;         // 10 words of units + 1 of scalars
  enum : uint32_t { FIGHTER = 0, CRUISER = 1, DREAD = 2, PORTAL = 3 };

  // unit word: type 0:2 | player 2 | slot 3:3 | hp 6:3 | facing 9:3 | q+6 12:4 | r+6 16:4 | moves_left 20:2 | cannons_fired 22:4 | exists 26
  struct State {
    uint32_t unit;       // lane-resident
    uint32_t player, turn;
    uint32_t misc;       // nunits 0:5 | acted 5 | over 6 | winner 7:2 (3 = none) | variant 9:2 | rep 11:2 (occurrences of the position, clipped to 3)
    uint32_t reserves;   // 3 bits per [player][type < 3]
  };
  __device__ __forceinline__ static uint32_t u_type(uint32_t u) { return u & 3u; }
  __device__ __forceinline__ static uint32_t u_player(uint32_t u) { return (u >> 2) & 1u; }
  __device__ __forceinline__ static uint32_t u_slot(uint32_t u) { return (u >> 3) & 7u; }
  __device__ __forceinline__ static uint32_t u_hp(uint32_t u) { return (u >> 6) & 7u; }
  __device__ __forceinline__ static uint32_t u_facing(uint32_t u) { return (u >> 9) & 7u; }
  __device__ __forceinline__ static int u_q(uint32_t u) { return static_cast<int>((u >> 12) & 15u) - 6; }
  __device__ __forceinline__ static int u_r(uint32_t u) { return static_cast<int>((u >> 16) & 15u) - 6; }
  __device__ __forceinline__ static uint32_t u_moves(uint32_t u) { return (u >> 20) & 3u; }
  __device__ __forceinline__ static uint32_t u_cannons(uint32_t u) { return (u >> 22) & 15u; }
  __device__ __forceinline__ static bool u_exists(uint32_t u) { return (u >> 26) & 1u; }
  __device__ __forceinline__ static bool u_alive(uint32_t u) { return u_exists(u) && u_hp(u) > 0; }
  __host__ __device__ __forceinline__ static uint32_t pack_unit(uint32_t type, uint32_t player, uint32_t slot, uint32_t hp, uint32_t facing, int q, int r,
                                                                 uint32_t moves, uint32_t cannons) {
    return type | player << 2 | slot << 3 | hp << 6 | facing << 9 | static_cast<uint32_t>(q + 6) << 12 | static_cast<uint32_t>(r + 6) << 16 |
           moves << 20 | cannons << 22 | 1u << 26;
  }
  __device__ __forceinline__ static uint32_t nunits(const State& s) { return s.misc & 31u; }
  __device__ __forceinline__ static bool acted(const State& s) { return (s.misc >> 5) & 1u; }
  __device__ __forceinline__ static bool over(const State& s) { return (s.misc >> 6) & 1u; }
  __device__ __forceinline__ static uint32_t winner(const State& s) { return (s.misc >> 7) & 3u; }
  __device__ __forceinline__ static uint32_t variant(const State& s) { return (s.misc >> 9) & 3u; }
  __device__ __forceinline__ static uint32_t rep(const State& s) { return (s.misc >> 11) & 3u; }
  __device__ __forceinline__ static void set_acted(State& s, bool v) { s.misc = (s.misc & ~(1u << 5)) | (v ? 1u << 5 : 0u); }
  __device__ __forceinline__ static void set_over(State& s, uint32_t w) { s.misc = (s.misc & ~(7u << 6)) | 1u << 6 | (w & 3u) << 7; }
  __device__ __forceinline__ static void set_rep(State& s, uint32_t c) { s.misc = (s.misc & ~(3u << 11)) | (c > 3u ? 3u : c) << 11; }
  __device__ __forceinline__ static uint32_t reserve(const State& s, uint32_t pl, uint32_t t) { return (s.reserves >> (3 * (pl * 3 + t))) & 7u; }
  __device__ __forceinline__ static int side(const State& s) { return variant(s) == 3 ? 6 : 5; }   // BOARD_SIDE, star_gambit_gs.h:22-60
  __host__ __device__ __forceinline__ static uint32_t start_count(uint32_t v, uint32_t t) {   // STARTING_*, star_gambit_gs.h:22-60
    const uint32_t tab = v == 0 ? 0x013u : v == 1 ? 0x104u : v == 2 ? 0x123u : 0x234u;   // fighters | cruisers << 4 | dreadnoughts << 8
    return (tab >> (4 * t)) & 15u;
  }
  __device__ __forceinline__ static bool turn_one(const State& s) { return s.turn == 1 || s.turn == 2; }   // :653
  __device__ __forceinline__ static uint32_t max_hp(uint32_t t) { return t == 0 ? 3u : t == 1 ? 4u : t == 2 ? 6u : 5u; }
  __device__ __forceinline__ static uint32_t max_moves(uint32_t t) { return t == 0 ? 2u : (t == 3 ? 0u : 1u); }
  __device__ __forceinline__ static uint32_t num_cannons(uint32_t t) { return t == 0 ? 1u : t == 1 ? 3u : t == 2 ? 4u : 0u; }

  // ---- hex geometry (star_gambit_gs.h:251-269): E, NE, NW, W, SW, SE packed as (d + 1) two-bit fields
  __device__ __forceinline__ static int dq(uint32_t d) { return static_cast<int>((1050u >> (2 * d)) & 3u) - 1; }
  __device__ __forceinline__ static int dr(uint32_t d) { return static_cast<int>((2625u >> (2 * d)) & 3u) - 1; }
  __device__ __forceinline__ static uint32_t rot(uint32_t d, int k) { return (d + static_cast<uint32_t>(k + 6)) % 6u; }
  __device__ __forceinline__ static bool inb(int q, int r, int s) {
    const int t = q + r;
    return q >= -s && q <= s && r >= -s && r <= s && t >= -s && t <= s;
  }
  __device__ __forceinline__ static bool on_canvas(int q, int r) { return q >= -6 && q <= 6 && r >= -6 && r <= 6; }
  __device__ __forceinline__ static uint32_t cell_of(int q, int r) { return static_cast<uint32_t>((q + 6) * 13 + (r + 6)); }

  struct Cells { int q[3], r[3]; uint32_t n; };
  // get_unit_hexes / get_portal_hexes, star_gambit_gs.cc:88-141
  __device__ __forceinline__ static Cells cells_of(uint32_t type, uint32_t player, int q, int r, uint32_t f, int s) {
    Cells c;
    c.q[0] = q; c.r[0] = r; c.q[1] = q; c.r[1] = r; c.q[2] = q; c.r[2] = r; c.n = 1;
    if (type == PORTAL) {
      c.n = 3;
      if (player == 0) { c.q[0] = 0; c.r[0] = s; c.q[1] = 1; c.r[1] = s - 1; c.q[2] = -1; c.r[2] = s; }
      else { c.q[0] = 0; c.r[0] = -s; c.q[1] = -1; c.r[1] = -s + 1; c.q[2] = 1; c.r[2] = -s; }
    } else if (type == CRUISER) {
      c.n = 2; c.q[1] = q + dq(rot(f, 3)); c.r[1] = r + dr(rot(f, 3));
    } else if (type == DREAD) {
      c.n = 3;
      c.q[1] = q + dq(rot(f, 4)); c.r[1] = r + dr(rot(f, 4));
      c.q[2] = q + dq(rot(f, 3)); c.r[2] = r + dr(rot(f, 3));
    }
    return c;
  }
  __device__ __forceinline__ static Cells cells_of_unit(uint32_t u, int s) { return cells_of(u_type(u), u_player(u), u_q(u), u_r(u), u_facing(u), s); }

  struct B192 { unsigned long long a, b, c; };
  __device__ __forceinline__ static bool bit(const B192& m, uint32_t cell) {
    const unsigned long long w = cell < 64 ? m.a : cell < 128 ? m.b : m.c;
    return (w >> (cell & 63u)) & 1ull;
  }
  // cross-lane hand-over inside the game's wavefront (every kernel that runs these rules uses 64-thread workgroups): a
  // wavefront-scope fence, see BigSlot::sync (engine_kernels_big.h)
  __device__ __forceinline__ static void lds_sync() {
#ifdef AZMI_WG_SYNC
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#endif
  }
  template <class T>
  __device__ __forceinline__ static T wave_xor(T v) {
    for (int off = 32; off > 0; off >>= 1) v ^= __shfl_xor(v, off, 64);
    return v;
  }

  // occupied cells of both players, from the lanes (get_all_occupied_hexes, star_gambit_gs.cc:371-390)
  __device__ __forceinline__ static void build_occ(const State& s, uint32_t lane, SgScratch& sm, B192& all, B192& p0) {
    if (lane < 6) sm.occ[lane] = 0ull;
    lds_sync();
    if (u_alive(s.unit)) {
      const Cells c = cells_of_unit(s.unit, side(s));
#pragma unroll
      for (uint32_t i = 0; i < 3; ++i)
        if (i < c.n && on_canvas(c.q[i], c.r[i])) {
          const uint32_t cell = cell_of(c.q[i], c.r[i]);
          atomicOr(&sm.occ[u_player(s.unit) * 3 + (cell >> 6)], 1ull << (cell & 63u));
        }
    }
    lds_sync();
    p0 = B192{sm.occ[0], sm.occ[1], sm.occ[2]};
    all = B192{p0.a | sm.occ[3], p0.b | sm.occ[4], p0.c | sm.occ[5]};
  }
  __device__ __forceinline__ static bool occ_at(const B192& m, int q, int r) { return on_canvas(q, r) && bit(m, cell_of(q, r)); }

  // compute_*_move (star_gambit_gs.cc:448-600): `code` is the per-type code of the reference's move enums
  __device__ __forceinline__ static void move_of(uint32_t type, int q, int r, uint32_t f, uint32_t code, int& nq, int& nr, uint32_t& nf) {
    if (type == FIGHTER) {
      nf = code == 0 ? f : code == 1 ? rot(f, 1) : rot(f, -1);
      nq = q + dq(nf); nr = r + dr(nf);
    } else if (type == CRUISER) {   // 0 rotate-left, 1 forward-left, 2 forward, 3 forward-right, 4 rotate-right
      if (code == 0 || code == 4) {
        const int bq = q + dq(rot(f, 3)), br = r + dr(rot(f, 3));   // the rear stays
        nf = rot(f, code == 0 ? 1 : -1);
        nq = bq + dq(nf); nr = br + dr(nf);
      } else {
        nf = code == 1 ? rot(f, 1) : code == 2 ? f : rot(f, -1);
        nq = q + dq(nf); nr = r + dr(nf);
      }
    } else {                        // dreadnought: 0 pivot-left, 1 slide forward-left, 2 slide forward-right, 3 pivot-right
      if (code == 0) { nf = rot(f, 1); nq = q + dq(rot(f, 3)) + dq(nf); nr = r + dr(rot(f, 3)) + dr(nf); }
      else if (code == 1) { nf = f; nq = q + dq(rot(f, 1)); nr = r + dr(rot(f, 1)); }
      else if (code == 2) { nf = f; nq = q + dq(f); nr = r + dr(f); }
      else { nf = rot(f, -1); nq = q + dq(rot(f, 4)) + dq(f); nr = r + dr(rot(f, 4)) + dr(f); }
    }
  }
  // SpatialAction slot -> move code of the type, -1 when the type has no such action (valid_moves :823-865, play_move :1128-1176)
  __device__ __forceinline__ static int move_code(uint32_t type, uint32_t slot) {
    if (type == FIGHTER) return slot < 3 ? static_cast<int>(slot) : -1;
    if (type == CRUISER) return slot == 0 ? 2 : slot == 1 ? 1 : slot == 2 ? 3 : slot == 3 ? 0 : 4;
    if (type == DREAD) return slot == 0 ? -1 : slot == 1 ? 1 : slot == 2 ? 2 : slot == 3 ? 0 : 3;
    return -1;
  }
  // SpatialAction slot (5..9) -> cannon index (valid_moves :867-901, play_move :1178-1217)
  __device__ __forceinline__ static int cannon_of(uint32_t type, uint32_t slot) {
    if (type == FIGHTER) return slot == 5 ? 0 : -1;
    if (type == CRUISER) return slot == 5 ? 1 : slot == 6 ? 0 : slot == 7 ? 2 : -1;
    if (type == DREAD) return slot == 6 ? 1 : slot == 7 ? 2 : slot == 8 ? 0 : slot == 9 ? 3 : -1;
    return -1;
  }
  // get_cannon_info (star_gambit_gs.cc:201-231): source hex and absolute direction of a cannon
  __device__ __forceinline__ static void cannon_geom(uint32_t type, int q, int r, uint32_t f, uint32_t cannon, int& sq, int& sr, uint32_t& d) {
    sq = q; sr = r; d = f;
    if (type == CRUISER) d = cannon == 0 ? rot(f, 1) : cannon == 1 ? f : rot(f, -1);
    if (type == DREAD) {
      d = cannon < 2 ? rot(f, 1) : f;
      if (cannon == 0) { sq = q + dq(rot(f, 3)); sr = r + dr(rot(f, 3)); }      // hexes[2]: the rear hex
      if (cannon == 3) { sq = q + dq(rot(f, 4)); sr = r + dr(rot(f, 4)); }      // hexes[1]
    }
  }
  // has_target_in_range, star_gambit_gs.cc:669-713
  __device__ __forceinline__ static bool has_target(int sq, int sr, uint32_t d, int s, const B192& all, const B192& enemy) {
    const int q1 = sq + dq(d), r1 = sr + dr(d), q2 = q1 + dq(d), r2 = r1 + dr(d);
    if (inb(q1, r1, s) && occ_at(enemy, q1, r1)) return true;
    if (!inb(q2, r2, s)) return false;
    if (occ_at(all, q1, r1)) return false;      // line of sight
    return occ_at(enemy, q2, r2);
  }
  __device__ __forceinline__ static void set_vbit(SgScratch& sm, uint32_t a) { atomicOr(&sm.vbits[a >> 6], 1ull << (a & 63u)); }
  // deploy geometry: is_deploy_valid / execute_deploy, star_gambit_gs.cc:729-770, 1051-1069
  __device__ __forceinline__ static bool deploy_facing_ok(uint32_t type, uint32_t player, uint32_t f) {
    if (type == DREAD) return player == 0 ? f <= 3 : (f == 0 || f >= 3);
    return player == 0 ? (f >= 1 && f <= 3) : (f == 4 || f == 5 || f == 0);
  }
  __device__ __forceinline__ static void deploy_anchor(uint32_t type, uint32_t player, uint32_t f, int s, int& q, int& r) {
    q = 0; r = player == 0 ? s - 1 : -(s - 1);
    if (type == DREAD) {   // get_dreadnought_anchor_dir, :157-169: p0 {1,2,2,3,-,-}, p1 {0,-,-,4,5,5}
      const uint32_t tab = player == 0 ? (1u | 2u << 3 | 2u << 6 | 3u << 9) : (0u | 4u << 9 | 5u << 12 | 5u << 15);
      const uint32_t d = (tab >> (3 * f)) & 7u;
      q += dq(d); r += dr(d);
    } else if (type == CRUISER) {
      q += dq(f); r += dr(f);
    }
  }

  // valid_moves (star_gambit_gs.cc:784-923 through the Unified remap :2560-2572) as bits of sm.vbits; returns their number
  __device__ __forceinline__ static uint32_t gen_valid(const State& s, uint32_t lane, SgScratch& sm) {
    if (lane < 27) sm.vbits[lane] = 0ull;
    B192 all, p0;
    build_occ(s, lane, sm, all, p0);     // (syncs: the zeroed map is visible before the first bit is set)
    if (!over(s)) {
      const int sd = side(s);
      const bool p1 = s.player == 1;
      const B192 enemy = p1 ? p0 : B192{all.a & ~p0.a, all.b & ~p0.b, all.c & ~p0.c};
      const uint32_t u = s.unit;
      if (!turn_one(s) && u_alive(u) && u_player(u) == s.player && u_type(u) != PORTAL) {
        const uint32_t type = u_type(u), f = u_facing(u);
        const int q = u_q(u), r = u_r(u);
        const Cells mine = cells_of_unit(u, sd);
        B192 others = all;
#pragma unroll
        for (uint32_t i = 0; i < 3; ++i)
          if (i < mine.n && on_canvas(mine.q[i], mine.r[i])) {
            const uint32_t cell = cell_of(mine.q[i], mine.r[i]);
            const unsigned long long b = ~(1ull << (cell & 63u));
            if (cell < 64) others.a &= b; else if (cell < 128) others.b &= b; else others.c &= b;
          }
        int row = q + 6, col = r + 6;
        if (p1) { row = 12 - row; col = 12 - col; }     // valid_moves' encode_action, :799-811
        const uint32_t a0 = static_cast<uint32_t>(row * 13 + col) * 10u;
        if (u_moves(u) > 0) {
#pragma unroll
          for (uint32_t slot = 0; slot < 5; ++slot) {
            const int code = move_code(type, slot);
            if (code < 0) continue;
            int nq, nr; uint32_t nf;
            move_of(type, q, r, f, static_cast<uint32_t>(code), nq, nr, nf);
            const Cells nc = cells_of(type, s.player, nq, nr, nf, sd);
            bool ok = true;
#pragma unroll
            for (uint32_t i = 0; i < 3; ++i)
              if (i < nc.n) ok = ok && inb(nc.q[i], nc.r[i], sd) && !occ_at(others, nc.q[i], nc.r[i]);
            if (ok) set_vbit(sm, a0 + slot);
          }
        }
#pragma unroll
        for (uint32_t slot = 5; slot < 10; ++slot) {
          const int cn = cannon_of(type, slot);
          if (cn < 0 || ((u_cannons(u) >> cn) & 1u)) continue;
          int sq, sr; uint32_t d;
          cannon_geom(type, q, r, f, static_cast<uint32_t>(cn), sq, sr, d);
          if (has_target(sq, sr, d, sd, all, enemy)) set_vbit(sm, a0 + slot);
        }
      }
      if (lane < 18) {
        const uint32_t t = lane / 6, f = lane % 6;
        if (reserve(s, s.player, t) > 0 && deploy_facing_ok(t, s.player, f)) {
          int aq, ar_;
          deploy_anchor(t, s.player, f, sd, aq, ar_);
          const Cells nc = cells_of(t, s.player, aq, ar_, f, sd);
          bool ok = true;
#pragma unroll
          for (uint32_t i = 0; i < 3; ++i)
            if (i < nc.n) ok = ok && inb(nc.q[i], nc.r[i], sd) && !occ_at(all, nc.q[i], nc.r[i]);
          if (ok) set_vbit(sm, DEPLOY0 + t * 6 + (p1 ? (f + 3) % 6 : f));
        }
      }
      if (lane == 18 && !turn_one(s) && acted(s)) set_vbit(sm, END_TURN);   // is_end_turn_valid, :772-778
    }
    lds_sync();
    uint32_t cnt = lane < 27 ? static_cast<uint32_t>(__builtin_popcountll(sm.vbits[lane])) : 0u;
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    return cnt;
  }
  // the set bits of sm.vbits as ascending move indices into out[0..k); returns k (all lanes)
  template <class T>
  __device__ __forceinline__ static uint32_t list_valid(uint32_t lane, SgScratch& sm, T* out, uint32_t cap) {
    unsigned long long w = lane < 27 ? sm.vbits[lane] : 0ull;
    const uint32_t cnt = static_cast<uint32_t>(__builtin_popcountll(w));
    uint32_t incl = cnt;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(incl, off, 64);
      if (lane >= static_cast<uint32_t>(off)) incl += o;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    uint32_t pos = incl - cnt;
    while (w) {
      const uint32_t b = static_cast<uint32_t>(__builtin_ctzll(w));
      w &= w - 1;
      if (pos < cap) out[pos] = static_cast<T>(lane * 64 + b);
      ++pos;
    }
    lds_sync();
    return total;
  }
  __device__ __forceinline__ static bool is_valid_bit(const SgScratch& sm, uint32_t mv) { return mv < static_cast<uint32_t>(M) && ((sm.vbits[mv >> 6] >> (mv & 63u)) & 1ull); }

  // compute_position_hash, star_gambit_gs.cc:1365-1382 (the reference's own hash)
  __device__ __forceinline__ static unsigned long long position_hash(const State& s) {
    unsigned long long h = 0ull;
    const uint32_t u = s.unit;
    if (u_alive(u)) {
      const unsigned long long uh = static_cast<unsigned long long>(u_type(u)) ^ (static_cast<unsigned long long>(u_player(u)) << 8) ^
                                    (static_cast<unsigned long long>(u_hp(u)) << 12) ^ (static_cast<unsigned long long>(u_facing(u)) << 20) ^
                                    (static_cast<unsigned long long>(u_q(u) + 10) << 28) ^ (static_cast<unsigned long long>(u_r(u) + 10) << 36);
      h = uh * 0x517cc1b727220a95ULL;
    }
    return wave_xor(h) ^ (static_cast<unsigned long long>(s.player) * 0x9e3779b97f4a7c15ULL);
  }
  // evaluation-cache key over the fields of hash() (star_gambit_gs.cc:321-338, 2399-2402): variant, player, has_taken_action,
  // every unit record in order (dead ones too), reserves - build-defined mixing (absl is salted)
  __device__ __forceinline__ static uint64_t key(const State& s, uint32_t lane) {
    const uint64_t kl = u_exists(s.unit) ? mix64(static_cast<uint64_t>(s.unit) ^ (static_cast<uint64_t>(lane + 1) * 0x9E3779B97F4A7C15ULL)) : 0ull;
    const uint64_t x = wave_xor(kl);
    const uint64_t k = mix64(0x5347556EULL ^ static_cast<uint64_t>(variant(s)) ^ (static_cast<uint64_t>(s.player) << 2) ^
                             (static_cast<uint64_t>(acted(s)) << 3) ^ (static_cast<uint64_t>(nunits(s)) << 4) ^
                             (static_cast<uint64_t>(s.reserves) << 16));
    return mix64(k ^ x);
  }
  // scores(), star_gambit_gs.cc:1347-1363: 0 running, else 1 + index of the one-hot entry (draw = 3)
  __device__ __forceinline__ static uint32_t terminal(const State& s) { return over(s) ? (winner(s) < 3 ? 1u + winner(s) : 3u) : 0u; }

  // StarGambitGS<Config>::StarGambitGS + the Unified wrapper's make_inner_game (star_gambit_gs.cc:251-290, 2411-2419)
  __device__ __forceinline__ static State initial(uint32_t v, uint32_t lane) {
    State s;
    const int sd = v == 3 ? 6 : 5;
    s.unit = lane == 0 ? pack_unit(PORTAL, 0, 0, 5, 2, 0, sd, 0, 0) : lane == 1 ? pack_unit(PORTAL, 1, 0, 5, 5, 0, -sd, 0, 0) : 0u;
    s.player = 0; s.turn = 1;
    s.misc = 2u | 3u << 7 | v << 9 | 1u << 11;     // two units, no winner, the start position seen once
    uint32_t res = 0;
    for (uint32_t pl = 0; pl < 2; ++pl) for (uint32_t t = 0; t < 3; ++t) res |= start_count(v, t) << (3 * (pl * 3 + t));
    s.reserves = res;
    return s;
  }
  // the variant of a new game (randomize_start, star_gambit_gs.cc:2421-2425): build-defined draw from the slot's coin stream,
  // identical to oracle/az_stargambit.hpp pick_variant
  __device__ __forceinline__ static uint32_t pick_variant(int pinned, const float* probs, Pcg32& coin) {
    if (pinned >= 0 && pinned <= 3) return static_cast<uint32_t>(pinned);
    const float total = ((probs[0] + probs[1]) + probs[2]) + probs[3];
    const float x = (canonical01(coin) * 1.0f + 0.0f) * total;
    uint32_t v = 0;
    float acc = probs[0];
    while (v < 3 && x >= acc) { ++v; acc += v == 1 ? probs[1] : v == 2 ? probs[2] : probs[3]; }
    return v;
  }

  // ---- state <-> engine words (gs_words[w][S]): words 0..9 = units 2w | 2w+1 << 32, word 10 = player | turn << 8 | misc << 24 | reserves << 40
  __device__ __forceinline__ static State load_words(const uint64_t* words, uint32_t S, uint32_t slot, uint32_t lane) {
    State s;
    const uint32_t* w32 = reinterpret_cast<const uint32_t*>(words);
    s.unit = lane < static_cast<uint32_t>(MAX_UNITS) ? w32[(static_cast<size_t>(lane >> 1) * S + slot) * 2 + (lane & 1u)] : 0u;
    const uint64_t x = words[static_cast<size_t>(10) * S + slot];
    s.player = static_cast<uint32_t>(x) & 1u; s.turn = static_cast<uint32_t>(x >> 8) & 0xFFFFu;
    s.misc = static_cast<uint32_t>(x >> 24) & 0xFFFFu; s.reserves = static_cast<uint32_t>(x >> 40) & 0x3FFFFu;
    return s;
  }
  __device__ __forceinline__ static void store_words(uint64_t* words, uint32_t S, uint32_t slot, uint32_t lane, const State& s) {
    uint32_t* w32 = reinterpret_cast<uint32_t*>(words);
    if (lane < static_cast<uint32_t>(MAX_UNITS)) w32[(static_cast<size_t>(lane >> 1) * S + slot) * 2 + (lane & 1u)] = s.unit;
    if (lane == 0)
      words[static_cast<size_t>(10) * S + slot] = static_cast<uint64_t>(s.player) | static_cast<uint64_t>(s.turn) << 8 |
                                                   static_cast<uint64_t>(s.misc) << 24 | static_cast<uint64_t>(s.reserves) << 40;
  }
  __host__ __device__ static uint32_t player_from_words(const uint64_t* words, uint32_t S, uint32_t slot) {
    return static_cast<uint32_t>(words[static_cast<size_t>(10) * S + slot]) & 1u;
  }

  // ---- execution -------------------------------------------------------------------------------------------------------
  // check_game_end, star_gambit_gs.cc:1313-1345
  __device__ __forceinline__ static void check_game_end(State& s) {
    const uint32_t u = s.unit;
    const uint64_t dead_portal = __ballot(u_exists(u) && u_type(u) == PORTAL && u_hp(u) == 0);
    if (dead_portal) {
      const uint32_t pl = u_player(static_cast<uint32_t>(__shfl(static_cast<int>(u), static_cast<int>(__builtin_ctzll(dead_portal)), 64)));
      set_over(s, 1u - pl);
      return;
    }
    for (uint32_t pl = 0; pl < 2; ++pl) {
      const bool ships = __ballot(u_alive(u) && u_player(u) == pl && u_type(u) != PORTAL) != 0;
      const bool res = (reserve(s, pl, 0) | reserve(s, pl, 1) | reserve(s, pl, 2)) != 0;
      if (!ships && !res) { set_over(s, 1u - pl); return; }
    }
  }
  // execute_end_turn, star_gambit_gs.cc:1263-1290.  `rep` owns the position history: rep.push(hash) appends and returns
  // the number of entries equal to it (itself included), rep.clear() empties it.
  template <class Rep>
  __device__ __forceinline__ static void end_turn(State& s, uint32_t lane, SgScratch& sm, Rep& rep) {
    s.player ^= 1u;
    ++s.turn;
    set_acted(s, false);
    if (s.turn > static_cast<uint32_t>(GAME_TURNS)) { set_over(s, 2u); return; }
    const uint32_t cnt = rep.push(position_hash(s));
    set_rep(s, cnt);
    if (cnt >= 3) { set_over(s, 2u); return; }
    if (u_alive(s.unit) && u_player(s.unit) == s.player)      // reset_turn_state, :1292-1300
      s.unit = (s.unit & ~(0x3Fu << 20)) | max_moves(u_type(s.unit)) << 20;
    if (gen_valid(s, lane, sm) == 0) set_over(s, 1u - s.player);
  }
  // play_move, star_gambit_gs.cc:1093-1238 through the Unified remap (:2578-2580; identity on the canvas coordinates used here)
  template <class Rep>
  __device__ __forceinline__ static void apply_move(State& s, uint32_t mv, uint32_t lane, SgScratch& sm, Rep& rep) {
    if (mv < static_cast<uint32_t>(SPATIAL)) {
      const uint32_t slot = mv % 10u, pos = mv / 10u;
      int row = static_cast<int>(pos / 13u), col = static_cast<int>(pos % 13u);
      if (s.player == 1) { row = 12 - row; col = 12 - col; }
      const int q = row - 6, r = col - 6;
      const uint32_t u = s.unit;
      const uint64_t who = __ballot(u_alive(u) && u_player(u) == s.player && u_type(u) != PORTAL && u_q(u) == q && u_r(u) == r);
      if (!who) return;                                    // "No unit at this hex": the reference returns before the repetition check
      const uint32_t ul = static_cast<uint32_t>(__builtin_ctzll(who));
      const uint32_t uu = static_cast<uint32_t>(__shfl(static_cast<int>(u), static_cast<int>(ul), 64));
      const uint32_t type = u_type(uu), f = u_facing(uu);
      const int sd = side(s);
      if (slot < 5) {
        const int code = move_code(type, slot);
        if (code >= 0) {                                   // execute_*_move, :929-972 (bounds only: legality is valid_moves' job)
          int nq, nr; uint32_t nf;
          move_of(type, q, r, f, static_cast<uint32_t>(code), nq, nr, nf);
          const Cells nc = cells_of(type, s.player, nq, nr, nf, sd);
          bool ok = true;
#pragma unroll
          for (uint32_t i = 0; i < 3; ++i) if (i < nc.n) ok = ok && inb(nc.q[i], nc.r[i], sd);
          if (ok) {
            if (lane == ul)
              s.unit = (uu & ~(0x7u << 9 | 0xFFu << 12 | 0x3u << 20)) | nf << 9 | static_cast<uint32_t>(nq + 6) << 12 |
                       static_cast<uint32_t>(nr + 6) << 16 | ((u_moves(uu) - 1u) & 3u) << 20;
            set_acted(s, true);
          }
        }
      } else {
        const int cn = cannon_of(type, slot);
        if (cn >= 0) {                                     // execute_fire, :978-1045
          if (lane == ul) s.unit = uu | (1u << (22 + cn));
          set_acted(s, true);
          int sq, sr; uint32_t d;
          cannon_geom(type, q, r, f, static_cast<uint32_t>(cn), sq, sr, d);
          B192 all, p0;
          build_occ(s, lane, sm, all, p0);
          const Cells mine = cells_of_unit(s.unit, sd);
          const bool live = u_alive(s.unit);
          for (int range = 1; range <= 2; ++range) {
            const int tq = sq + range * dq(d), tr = sr + range * dr(d);
            if (!inb(tq, tr, sd)) continue;
            if (range == 2 && occ_at(all, sq + dq(d), sr + dr(d))) break;   // has_line_of_sight, :233-245
            bool covers = false;
#pragma unroll
            for (uint32_t i = 0; i < 3; ++i) covers = covers || (live && i < mine.n && mine.q[i] == tq && mine.r[i] == tr);
            const uint64_t hit = __ballot(covers);
            if (!hit) continue;
            const uint32_t tl = static_cast<uint32_t>(__builtin_ctzll(hit));      // find_unit_at_hex: first unit in order
            if (tl == ul) continue;
            const uint32_t dmg = range == 1 ? 2u : 1u;
            const uint32_t thp = u_hp(static_cast<uint32_t>(__shfl(static_cast<int>(s.unit), static_cast<int>(tl), 64)));
            const uint32_t nhp = dmg >= thp ? 0u : thp - dmg;                     // apply_damage, :1302-1311
            if (lane == tl) s.unit = (s.unit & ~(7u << 6)) | nhp << 6;
            if (nhp == 0) check_game_end(s);
            break;
          }
        }
      }
      const uint32_t cnt = rep.push(position_hash(s));     // check_repetition after every spatial action, :1220-1223, 1246-1261
      set_rep(s, cnt);
      if (cnt >= 3) set_over(s, 2u);
    } else if (mv < static_cast<uint32_t>(END_TURN)) {     // execute_deploy, :1051-1087
      const uint32_t rel = mv - DEPLOY0, type = rel / 6u;
      uint32_t f = rel % 6u;
      if (s.player == 1) f = (f + 3u) % 6u;
      rep.clear();
      int aq, ar_;
      deploy_anchor(type, s.player, f, side(s), aq, ar_);
      const uint32_t u = s.unit;
      uint32_t mx = (u_exists(u) && u_player(u) == s.player && u_type(u) == type) ? u_slot(u) + 1u : 0u;   // get_next_slot, :360-369
      for (int off = 32; off > 0; off >>= 1) mx = max(mx, static_cast<uint32_t>(__shfl_xor(static_cast<int>(mx), off, 64)));
      const uint32_t n = nunits(s);
      if (lane == n) s.unit = pack_unit(type, s.player, mx & 7u, max_hp(type), f, aq, ar_, 0, (1u << num_cannons(type)) - 1u);
      s.misc = (s.misc & ~31u) | ((n + 1u) & 31u);
      const uint32_t sh = 3 * (s.player * 3 + type);
      s.reserves = (s.reserves & ~(7u << sh)) | ((reserve(s, s.player, type) - 1u) & 7u) << sh;
      end_turn(s, lane, sm, rep);
    } else {
      end_turn(s, lane, sm, rep);
    }
  }

  // canonicalized(), star_gambit_gs.cc:1384-1669 + the Unified canvas and variant planes (:2586-2616), written by the wave
  __device__ __forceinline__ static void write_canonical(const State& s, float* row, uint32_t lane, SgScratch& sm) {
    if (lane < 57) sm.masks[lane] = 0ull;
    for (uint32_t i = lane; i < 176; i += 64) sm.cellunit[i] = 0xFF;
    if (lane < 24) sm.uword[lane] = s.unit;
    lds_sync();
    const int sd = side(s);
    const bool p1 = s.player == 1;
    const uint32_t u = s.unit;
    if (u_alive(u)) {
      const uint32_t type = u_type(u);
      const Cells c = cells_of_unit(u, sd);
      const uint32_t pres = (u_player(u) == s.player ? 0u : 4u) + type;
      const uint32_t head = 8u + (p1 ? (u_facing(u) + 3u) % 6u : u_facing(u));
#pragma unroll
      for (uint32_t i = 0; i < 3; ++i)
        if (i < c.n && on_canvas(c.q[i], c.r[i])) {
          const uint32_t cell = cell_of(c.q[i], c.r[i]);
          sm.cellunit[cell] = static_cast<uint8_t>(lane);
          atomicOr(&sm.masks[pres * 3 + (cell >> 6)], 1ull << (cell & 63u));
          if (type != PORTAL) atomicOr(&sm.masks[head * 3 + (cell >> 6)], 1ull << (cell & 63u));
        }
      if (type != PORTAL && on_canvas(u_q(u), u_r(u))) {     // unfired cannons at the anchor hex, :1530-1577
        const uint32_t cell = cell_of(u_q(u), u_r(u));
        for (uint32_t cn = 0; cn < num_cannons(type); ++cn) {
          const uint32_t oslot = type == FIGHTER ? 0u : type == CRUISER ? (cn == 0 ? 1u : cn == 1 ? 0u : 2u) : (cn == 0 ? 3u : cn == 1 ? 1u : cn == 2 ? 2u : 4u);
          if (!((u_cannons(u) >> cn) & 1u)) atomicOr(&sm.masks[(14u + oslot) * 3 + (cell >> 6)], 1ull << (cell & 63u));
        }
      }
    }
    float portal_hp[2];
#pragma unroll
    for (uint32_t k = 0; k < 2; ++k) {     // find_unit_by_slot(player, PORTAL, 0): alive only
      const uint32_t pl = k == 0 ? s.player : 1u - s.player;
      const uint64_t m = __ballot(u_alive(u) && u_type(u) == PORTAL && u_player(u) == pl && u_slot(u) == 0);
      portal_hp[k] = m ? static_cast<float>(u_hp(static_cast<uint32_t>(__shfl(static_cast<int>(u), static_cast<int>(__builtin_ctzll(m)), 64)))) / 5.0f : 0.0f;
    }
    lds_sync();
    const uint32_t v = variant(s);
    const float repv = rep(s) == 0 ? 0.0f : rep(s) == 1 ? 0.5f : 1.0f;
    // plane-outer: which source a plane reads is wave-uniform, the lane's three cells (lane, lane + 64, lane + 128 < 169), their
    // source cells and the on-board test are computed once (the element-outer form spent ~60 vector instructions per element
    // on the index arithmetic and the plane switch: 95 iterations per leaf)
    uint32_t src[3];
    bool on[3], brd[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const uint32_t cell = lane + 64u * j;
      on[j] = cell < 169u;
      src[j] = on[j] ? (p1 ? 168u - cell : cell) : 0u;          // (q, r) -> (-q, -r) for player 1
      brd[j] = on[j] && inb(static_cast<int>(src[j] / 13u) - 6, static_cast<int>(src[j] % 13u) - 6, sd);
    }
    auto put = [&](uint32_t ch, const float (&x)[3]) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (on[j]) row[ch * 169u + lane + 64u * j] = x[j];
    };
    auto put_board_const = [&](uint32_t ch, float val) {       // a scalar broadcast over the board's cells
      float x[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) x[j] = brd[j] ? val : 0.0f;
      put(ch, x);
    };
    put_board_const(0, 1.0f);
    for (uint32_t ch = 1; ch <= 21; ++ch) {
      float x[3] = {0.0f, 0.0f, 0.0f};
      if (ch == 15 || ch == 16) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const uint32_t cu = on[j] ? sm.cellunit[src[j]] : 0xFFu;
          if (cu != 0xFF) {
            const uint32_t w = sm.uword[cu], t = u_type(w);
            if (ch == 15) x[j] = static_cast<float>(u_hp(w)) / static_cast<float>(max_hp(t));
            else if (t != PORTAL) x[j] = static_cast<float>(u_moves(w)) / static_cast<float>(max_moves(t));
          }
        }
      } else {
        const uint32_t mi = ch <= 14 ? ch - 1u : 14u + (ch - 17u);
#pragma unroll
        for (int j = 0; j < 3; ++j) x[j] = (on[j] && ((sm.masks[mi * 3 + (src[j] >> 6)] >> (src[j] & 63u)) & 1ull)) ? 1.0f : 0.0f;
      }
      put(ch, x);
    }
    put_board_const(22, acted(s) ? 1.0f : 0.0f);
    put_board_const(23, repv);
    for (uint32_t ch = 24; ch <= 29; ++ch) {
      const uint32_t k = ch - 24u, pl = k < 3 ? s.player : 1u - s.player, t = k % 3u;
      const uint32_t st = start_count(v, t);
      put_board_const(ch, st > 0 ? static_cast<float>(reserve(s, pl, t)) / static_cast<float>(st) : 0.0f);
    }
    put_board_const(30, portal_hp[0]);
    put_board_const(31, portal_hp[1]);
    for (uint32_t ch = 32; ch < 36; ++ch) put_board_const(ch, (ch - 32u == v) ? 1.0f : 0.0f);
    lds_sync();
  }
};

// ---- the reference's pickle image (star_gambit_gs.cc:2246-2251 inner, 2446-2449 Unified): parsed by the wave -----------
//   Unified: f32 probs[4] | i32 pinned | u8 variant | u32 inner_size | inner
//   inner:   u32 n | n x 9 B (type, player, slot, hp, facing, q, r, moves_left, cannons_fired) | u8 reserves[2][4] | u8 player |
//            u32 turn | u8 acted | u8 over | i8 winner | u32 hist_len | hist_len x u64
// `hist` receives the history; false = malformed / beyond the engine's bounds
__device__ inline bool sg_parse_image(const uint8_t* b, uint32_t bytes, uint32_t lane, StarGambit::State& s, uint64_t* hist, uint32_t& nhist,
                                      uint32_t cap) {
  using G = StarGambit;
  nhist = 0;
  auto rd32 = [&](uint32_t off) { return uint32_t(b[off]) | uint32_t(b[off + 1]) << 8 | uint32_t(b[off + 2]) << 16 | uint32_t(b[off + 3]) << 24; };
  if (bytes < 25u + 4u) return false;
  const uint32_t v = b[20];
  const uint32_t inner_size = rd32(21);
  if (v > 3 || 25u + inner_size != bytes) return false;
  const uint8_t* in = b + 25;
  const uint32_t n = rd32(25);
  if (n > static_cast<uint32_t>(G::MAX_UNITS) || inner_size < 4u + 9u * n + 8u + 8u + 4u) return false;
  bool ok = true;
  s.unit = 0;
  if (lane < n) {
    const uint8_t* r = in + 4 + 9 * lane;
    const int q = static_cast<int8_t>(r[5]), rr = static_cast<int8_t>(r[6]);
    ok = r[0] <= 3 && r[1] <= 1 && r[2] <= 7 && r[3] <= 7 && r[4] <= 5 && q >= -6 && q <= 6 && rr >= -6 && rr <= 6 && r[7] <= 3 && r[8] <= 15;
    s.unit = G::pack_unit(r[0] & 3u, r[1] & 1u, r[2] & 7u, r[3] & 7u, r[4] % 6u, ok ? q : 0, ok ? rr : 0, r[7] & 3u, r[8] & 15u);
  }
  if (__ballot(!ok)) return false;
  const uint8_t* t = in + 4 + 9 * n;
  uint32_t res = 0;
  for (uint32_t pl = 0; pl < 2; ++pl) for (uint32_t ty = 0; ty < 3; ++ty) { if (t[pl * 4 + ty] > 7) return false; res |= uint32_t(t[pl * 4 + ty]) << (3 * (pl * 3 + ty)); }
  s.reserves = res;
  s.player = t[8] & 1u;
  s.turn = uint32_t(t[9]) | uint32_t(t[10]) << 8 | uint32_t(t[11]) << 16 | uint32_t(t[12]) << 24;
  if (s.turn > 0xFFFFu) return false;
  const uint32_t acted = t[13] != 0, over = t[14] != 0;
  const int w = static_cast<int8_t>(t[15]);
  const uint32_t hl = uint32_t(t[16]) | uint32_t(t[17]) << 8 | uint32_t(t[18]) << 16 | uint32_t(t[19]) << 24;
  if (hl > cap || 4u + 9u * n + 8u + 12u + 8u * hl != inner_size) return false;
  s.misc = n | acted << 5 | over << 6 | (w >= 0 && w <= 2 ? static_cast<uint32_t>(w) : 3u) << 7 | v << 9;
  const uint8_t* hp = t + 20;
  for (uint32_t i = lane; i < hl; i += 64) {
    uint64_t x = 0;
    for (int k = 7; k >= 0; --k) x = x << 8 | hp[8 * i + k];
    hist[i] = x;
  }
  nhist = hl;
  G::lds_sync();
  const unsigned long long cur = G::position_hash(s);      // canonical plane 23 counts the current position in the history, :1591-1599
  uint32_t cnt = 0;
  for (uint32_t i = lane; i < hl; i += 64) cnt += hist[i] == cur;
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
  G::set_rep(s, cnt);
  return true;
}

}  // namespace azmi
