// Tree + game-step kernel for games with wide nodes and a large action space (the Tafl family:
// ~113 children per node on average, 2662 moves).  Same algorithm and the same reference citations as
// engine_kernels.h; what changes is the mapping:
//   * ONE WAVEFRONT PER GAME SLOT (workgroup = 64 lanes): control flow is wave-uniform, children are
//     walked in chunks of 64 (consecutive lanes -> consecutive children of the SoA node arrays),
//   * child scores / priors / visit counts are staged in LDS (per-wave scratch) so that the
//     order-sensitive reductions of the reference (sums in child order, sums over the dense
//     [num_moves] vector in move order) are plain in-order LDS sweeps,
//   * legal moves are generated straight from the bitboards by all lanes (one or two squares per
//     lane, wave prefix-sum for the ascending move order the reference's dense mask implies),
//   * the repetition list of the game (keys since the last capture) is staged in LDS once per round;
//     positions along the descent are appended to a path-local list, exactly the effect of the
//     reference's per-simulation gs.copy() of the repetition map (tawlbwrdd_gs.cc:74-78).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev_games.h"
#include "dev_rng.h"
#include "dev_stargambit.h"
#include "engine_kernels.h"
#include "engine_types.h"

namespace azmi {

// StarGambit (dev_stargambit.h) rides in the same engine: its units are lane-resident, its rules are wave-cooperative
// (they need their own LDS scratch), its position history lives in HBM (ar.rep_list / ar.rep_path: one entry per ACTION, up
// to thousands) instead of LDS, and it adds relative values, variants and multi-action turns.  Every difference is an
// `if constexpr (kSG)` branch below; the Tafl instantiations compile to what they were.
#ifdef AZMI_BIG_PROF
// phase timing of the wide-game round (an experiment build: -DAZMI_BIG_PROF): ticks of the 100 MHz wall clock per phase, summed over
// every wave; [0] load [1] process_result (+ move) [2] descent [3] move generation [4] shuffle [5] child records [6] leaf planes + probe
// [7] store [8] rounds
__device__ unsigned long long g_big_prof[16];
#define AZB_PROF_MARK(i) do { const unsigned long long now_ = wall_clock64(); pf_[i] += now_ - pf_t_; pf_t_ = now_; } while (0)
#define AZB_PROF_DECL do {} while (0)
#define AZB_PROF_MEMBERS unsigned long long pf_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pf_t_ = 0;
#else
#define AZB_PROF_MARK(i) do {} while (0)
#define AZB_PROF_DECL do {} while (0)
#define AZB_PROF_MEMBERS
#endif
// pcg32 is an LCG: the state after j draws is A_j * s0 + C_j (A_0 = 1, C_0 = 0; A_{j+1} = A_j * mult, C_{j+1} = C_j * mult + inc), so
// lane j can make draw j of a std::shuffle by itself (expand_node)
struct PcgJumpTable {
  uint64_t a[260], c[260];
  constexpr PcgJumpTable() : a{}, c{} {
    uint64_t A = 1, C = 0;
    for (int j = 0; j < 260; ++j) { a[j] = A; c[j] = C; A = A * Pcg32::kMult; C = C * Pcg32::kMult + Pcg32::kInc; }
  }
};
__device__ const PcgJumpTable kPcgJump{};
template <class GM> struct is_stargambit { static constexpr bool value = false; };
template <> struct is_stargambit<StarGambit> { static constexpr bool value = true; };
struct NoRulesScratch {};
template <class GM, bool SG = is_stargambit<GM>::value> struct RulesScratch { using type = NoRulesScratch; };
template <class GM> struct RulesScratch<GM, true> { using type = SgScratch; };

template <class GM>
struct BigScratch {  // per-wave LDS
  static constexpr int kRepLds = is_stargambit<GM>::value ? 1 : GM::MAX_TURNS + 2;
  uint16_t moves[GM::MAXK];
  alignas(16) float f0[GM::MAXK];     // (16-byte aligned: seq_sum_f0 reads it four floats at a time)
  float f1[GM::MAXK], f2[GM::MAXK];
  uint32_t n[GM::MAXK];
  float dense[GM::M];
  uint64_t glist[kRepLds];  // game repetition list (since the last capture)
  uint64_t plist[kRepLds];  // path-local repetition list of the running descent
  typename RulesScratch<GM>::type rules;
};

#define AZB_SEL(arr, seat) ((seat) == 0 ? arr[0] : arr[P > 1 ? 1 : 0])
template <class GM>
struct BigSlot {
  AZB_PROF_MEMBERS
  static_assert(GM::P == 2, "the wide-game engine is written for two-player games (first-visit value, resign entries)");
  static constexpr int G = 64, P = GM::P, M = GM::M, MAXK = GM::MAXK;
  static constexpr bool kSG = is_stargambit<GM>::value;
  const EngineParams& ep;
  const EngineArrays& ar;
  BigScratch<GM>& sm;
  uint32_t slot, lane;
  Pcg32 rng, coin;
  typename GM::State gs;
  uint8_t flags;
  uint32_t t_root[P], t_bump[P], t_depth[P];
  uint64_t t_tld[P];
  uint32_t cur, plen, ph_rows, glen;
  uint32_t staged_k = 0;          // children staged by stage_root (0: none): the support of sm.dense, see dense_pow
  uint32_t leaf_rep_len = 0;      // find_leaf: entries of sm.plist at the leaf, and whether the game's list still counts
  bool leaf_base_valid = true;    // (a capture on the path clears it); the rollout of a PLAYOUT seat continues from them
  // per-seat search settings of this game's seat permutation (see SlotCtx)
  uint32_t perm, sv_w0[P], sv_w1[P];
  float sv_eps[P], sv_rt[P];
  __device__ __forceinline__ uint32_t seat_visits(uint32_t seat) const { return AZB_SEL(sv_w0, seat); }
  __device__ __forceinline__ uint32_t seat_cap_visits(uint32_t seat) const { return AZB_SEL(sv_w1, seat) & 0xFFFFFFu; }
  __device__ __forceinline__ bool seat_fpu_zero(uint32_t seat) const { return (AZB_SEL(sv_w1, seat) >> 24) & 1u; }
  __device__ __forceinline__ bool seat_eval_random(uint32_t seat) const { return (AZB_SEL(sv_w1, seat) >> 25) & 1u; }
  __device__ __forceinline__ uint32_t seat_group(uint32_t seat) const { return (AZB_SEL(sv_w1, seat) >> 26) & 3u; }
  __device__ __forceinline__ bool seat_eval_playout(uint32_t seat) const { return (AZB_SEL(sv_w1, seat) >> 28) & 1u; }
  __device__ __forceinline__ float seat_eps(uint32_t seat) const { return AZB_SEL(sv_eps, seat); }
  __device__ __forceinline__ float seat_root_temp(uint32_t seat) const { return AZB_SEL(sv_rt, seat); }
  // per-seat Gumbel / resign settings (words 4-7 of the seat record), read on demand
  __device__ __forceinline__ const uint32_t* seat_rec(uint32_t seat) const { return ar.seat_tab + (static_cast<size_t>(perm) * P + seat) * kSeatWords; }
  __device__ __forceinline__ bool seat_gumbel(uint32_t seat) const { return ep.gumbel_on && (seat_rec(seat)[4] & 1u); }
  __device__ __forceinline__ bool seat_gumbel_full(uint32_t seat) const { return (seat_rec(seat)[4] >> 1) & 1u; }
  __device__ __forceinline__ bool seat_gumbel_g3(uint32_t seat) const { return (seat_rec(seat)[4] >> 2) & 1u; }
  __device__ __forceinline__ uint32_t seat_gumbel_m(uint32_t seat) const { return (seat_rec(seat)[4] >> 8) & 0xFFFFu; }
  __device__ __forceinline__ uint32_t seat_resign_need(uint32_t seat) const { return seat_rec(seat)[4] >> 24; }
  __device__ __forceinline__ float seat_sigma_scale(uint32_t seat, uint32_t max_visit) const {   // (c_visit + max N) * c_scale
    const uint32_t* r = seat_rec(seat);
    return (__uint_as_float(r[5]) + static_cast<float>(max_visit)) * __uint_as_float(r[6]);
  }
  __device__ __forceinline__ float seat_resign_threshold(uint32_t seat) const { return __uint_as_float(seat_rec(seat)[7]); }

  __device__ __forceinline__ BigSlot(const EngineParams& e, const EngineArrays& a, BigScratch<GM>& s, uint32_t sl, uint32_t l)
      : ep(e), ar(a), sm(s), slot(sl), lane(l) {}

  __device__ __forceinline__ size_t tree_base(uint32_t seat) const { return (static_cast<size_t>(slot) * P + seat) * ep.cap; }
  __device__ __forceinline__ void raise(uint32_t bit) const { if (lane == 0) { atomicOr(&ar.ctl->overflow, bit); ar.ctl->stop = 1; } }
  // Cross-LANE hand-over inside the slot's wavefront (LDS scratch, node records, history lists written by some lanes and read by
  // others).  The workgroup IS one wavefront, whose LDS and vector-memory operations are performed in order as seen by that
  // wavefront (one LDS, one vector L1 per CU: the AMDGPU memory model needs no instruction for wavefront scope), so this is a
  // wavefront-scope fence + a scheduling barrier for the compiler - NOT a workgroup fence, whose s_waitcnt vmcnt(0) would stall
  // the wave on every store still in flight (AZMI_WG_SYNC restores the round-1 form for comparison).
  __device__ __forceinline__ void sync() const {
#ifdef AZMI_WG_SYNC
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#endif
  }
  __device__ __forceinline__ void set_seat(uint32_t (&arr)[P], uint32_t seat, uint32_t v) {
#pragma unroll
    for (int p = 0; p < P; ++p) if (static_cast<uint32_t>(p) == seat) arr[p] = v;
  }
  template <class T>
  __device__ __forceinline__ T wave_sum(T v) const {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
  }

  // ---- load / store -----------------------------------------------------------------------------------
  __device__ __forceinline__ void load() {
    rng.state = ar.rng[slot]; coin.state = ar.coin[slot]; flags = ar.flags[slot];
    const uint32_t S = ep.S;
    if constexpr (kSG) {
      gs = GM::load_words(ar.gs_words, S, slot, lane);
    } else {
      const uint64_t w0 = ar.gs_words[0 * S + slot], w1 = ar.gs_words[1 * S + slot], w2 = ar.gs_words[2 * S + slot],
                     w3 = ar.gs_words[3 * S + slot], w4 = ar.gs_words[4 * S + slot];
      gs.def[0] = w0; gs.def[1] = w1; gs.atk[0] = w2; gs.atk[1] = w3;
      gs.king = static_cast<uint32_t>(w4) & 0x7Fu; gs.turn = static_cast<uint32_t>(w4 >> 8) & 0xFFFFu;
      gs.player = static_cast<uint32_t>(w4 >> 24) & 1u; gs.rep = static_cast<uint32_t>(w4 >> 32) & 0xFFu;
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const uint32_t t = slot * P + p;
      t_root[p] = ar.root[t]; t_bump[p] = ar.bump[t]; t_depth[p] = ar.depth[t]; t_tld[p] = ar.tld[t];
    }
    cur = ar.cur[slot]; plen = ar.plen[slot]; ph_rows = ar.ph_count[slot]; glen = ar.rep_len[slot];
    perm = ar.perm[slot];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const uint32_t* rec = ar.seat_tab + (static_cast<size_t>(perm) * P + p) * kSeatWords;
      sv_w0[p] = rec[0]; sv_w1[p] = rec[1]; sv_eps[p] = __uint_as_float(rec[2]); sv_rt[p] = __uint_as_float(rec[3]);
    }
    if constexpr (!kSG) {
      const uint64_t* gl = ar.rep_list + static_cast<size_t>(slot) * (GM::MAX_TURNS + 2);
      for (uint32_t i = lane; i < glen; i += G) sm.glist[i] = gl[i];
    }
    sync();
  }
  __device__ __forceinline__ void store(uint8_t sstate) const {
    if constexpr (kSG) GM::store_words(ar.gs_words, ep.S, slot, lane, gs);
    if (lane != 0) return;
    ar.rng[slot] = rng.state; ar.coin[slot] = coin.state; ar.flags[slot] = flags; ar.sstate[slot] = sstate;
    const uint32_t S = ep.S;
    if constexpr (!kSG) {
      ar.gs_words[0 * S + slot] = gs.def[0]; ar.gs_words[1 * S + slot] = gs.def[1];
      ar.gs_words[2 * S + slot] = gs.atk[0]; ar.gs_words[3 * S + slot] = gs.atk[1];
      ar.gs_words[4 * S + slot] = static_cast<uint64_t>(gs.king) | (static_cast<uint64_t>(gs.turn) << 8) |
                                  (static_cast<uint64_t>(gs.player) << 24) | (static_cast<uint64_t>(gs.rep) << 32);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const uint32_t t = slot * P + p;
      ar.root[t] = t_root[p]; ar.bump[t] = t_bump[p]; ar.depth[t] = t_depth[p]; ar.tld[t] = t_tld[p];
    }
    ar.cur[slot] = cur; ar.plen[slot] = plen; ar.ph_count[slot] = ph_rows; ar.rep_len[slot] = glen;
  }
  __device__ __forceinline__ void reset_tree(uint32_t seat) {
    set_seat(t_root, seat, 0); set_seat(t_bump, seat, 1); set_seat(t_depth, seat, 0);
#pragma unroll
    for (int p = 0; p < P; ++p) if (static_cast<uint32_t>(p) == seat) t_tld[p] = 0;
    const size_t tb = tree_base(seat);
    if (lane == 0) { ar.N[tb] = 0; ar.Q[tb] = 0; ar.Pr[tb] = 0; ar.D[tb] = 0; ar.V[tb] = 0; ar.META[tb] = 0; }
    if (ep.gumbel_on) set_gumbel_num_sims(seat, 0);
  }

  // ---- repetition-aware move on a state; `list`/`len` is the list the new position is appended to --------
  // base list = glist[0, glen) unless a capture cleared it (tawlbwrdd_gs.cc:246-332)
  __device__ __forceinline__ uint64_t* game_list() const {
    if constexpr (kSG) return ar.rep_list + static_cast<size_t>(slot) * (GM::MAX_TURNS + 2); else return sm.glist;
  }
  __device__ __forceinline__ uint64_t* path_list() const {
    if constexpr (kSG) return ar.rep_path + static_cast<size_t>(slot) * (GM::MAX_TURNS + 2); else return sm.plist;
  }
  // StarGambit: the position history as the rules see it (dev_stargambit.h apply_move): `list[0, len)` is the list being
  // written - the game's own list (ar.rep_list) when a move is played, the path-local tail (ar.rep_path) during a descent,
  // which also counts the game's list [0, base_len) until a deploy on the path clears it (star_gambit_gs.cc:1054)
  struct SgRep {
    BigSlot& c; uint64_t* list; uint32_t& len; bool& base_valid; uint32_t base_len; bool is_game; bool overflow = false;
    __device__ __forceinline__ void clear() { len = 0; base_valid = false; }
    __device__ __forceinline__ uint32_t push(unsigned long long k) {
      uint32_t cnt = 0;
      for (uint32_t i = c.lane; i < len; i += G) cnt += list[i] == k;
      if (base_valid && !is_game) {
        const uint64_t* gl = c.ar.rep_list + static_cast<size_t>(c.slot) * (GM::MAX_TURNS + 2);
        for (uint32_t i = c.lane; i < base_len; i += G) cnt += gl[i] == k;
      }
      cnt = c.wave_sum(cnt) + 1;
      if (len >= static_cast<uint32_t>(GM::MAX_TURNS + 1)) { overflow = true; return cnt; }
      if (c.lane == 0) list[len] = k;
      ++len;
      c.sync();
      return cnt;
    }
  };
  __device__ __forceinline__ bool step_state(typename GM::State& st, uint32_t mv, uint64_t* list, uint32_t& len, bool& base_valid,
                             uint32_t base_len) {
    if constexpr (kSG) {
      SgRep r{*this, list, len, base_valid, base_len, list == game_list()};
      GM::apply_move(st, mv, lane, sm.rules, r);
      if (r.overflow) { raise(8u); return false; }
      return true;
    } else {
    if (st.turn == 0) {  // the start position enters the map with count 1 at the first move
      if (lane == 0) list[len] = GM::rep_key(st);
      ++len;
      sync();
    }
    bool cap = false;
    if (!GM::apply_move(st, mv, &cap)) return false;
    if (cap) { len = 0; base_valid = false; }
    const uint64_t k = GM::rep_key(st);
    uint32_t cnt = 0;
    for (uint32_t i = lane; i < len; i += G) cnt += list[i] == k;
    if (base_valid && list != sm.glist)
      for (uint32_t i = lane; i < base_len; i += G) cnt += sm.glist[i] == k;
    cnt = wave_sum(cnt) + 1;
    sync();
    if (lane == 0) list[len] = k;
    ++len;
    st.rep = cnt;
    sync();
    return true;
    }
  }

  // ---- Node::add_children: legal moves ascending (from the bitboards), std::shuffle, append -------------
  __device__ __forceinline__ bool expand_node(uint32_t seat, uint32_t node, const typename GM::State& st, uint64_t meta_keep,
                              uint32_t& c0_out, uint32_t& k_out) {
    const size_t tb = tree_base(seat);
    uint32_t base = 0;
    AZB_PROF_DECL;
    if constexpr (kSG) {   // legal moves in ascending order from the dense bit map (dev_stargambit.h)
      GM::gen_valid(st, lane, sm.rules);
      base = GM::list_valid(lane, sm.rules, sm.moves, static_cast<uint32_t>(MAXK));
      if (base > static_cast<uint32_t>(MAXK)) { raise(8u); return false; }
    } else {
    // move generation.  The mover's pieces are compacted first - piece j in ascending square order goes to lane j (two ballots = the
    // mover's bitboard words; a lane's rank is a population count) - so every lane computes at most ONE slide mask and the wave
    // makes one prefix scan (round 4: the kernel used to walk squares lane and lane + 64 in two passes of mask + scan + list)
    {
      const uint32_t sqa = lane, sqb = 64u + lane;
      const bool owna = sqa < static_cast<uint32_t>(GM::SQ) && GM::own_piece(st, st.player, sqa);
      const bool ownb = sqb < static_cast<uint32_t>(GM::SQ) && GM::own_piece(st, st.player, sqb);
      const uint64_t ba = __ballot(owna), bb = __ballot(ownb);
      const uint32_t na = static_cast<uint32_t>(__popcll(ba)), npieces = na + static_cast<uint32_t>(__popcll(bb));
      const uint64_t below = (1ull << lane) - 1ull;
      if (owna) sm.n[__popcll(ba & below)] = sqa;
      if (ownb) sm.n[na + __popcll(bb & below)] = sqb;
      sync();
      for (uint32_t pbase = 0; pbase < npieces; pbase += G) {       // (one turn: a side has at most 24 pieces; a hand-built board may have more)
        const bool on = pbase + lane < npieces;
        const uint32_t sq = on ? sm.n[pbase + lane] : 0u;
        const uint32_t mask = on ? GM::slide_mask(st, sq) : 0u;
        const uint32_t cnt = __builtin_popcount(mask);
        uint32_t incl = cnt;  // inclusive wave scan
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t o = __shfl_up(incl, off, 64);
          if (lane >= static_cast<uint32_t>(off)) incl += o;
        }
        uint32_t pos = base + incl - cnt;
        const uint32_t total = __shfl(incl, 63, 64);
        if (base + total > static_cast<uint32_t>(MAXK)) { raise(8u); return false; }
        uint32_t m = mask;
        while (m) {
          const uint32_t b = __builtin_ctz(m);
          m &= m - 1;
          sm.moves[pos++] = static_cast<uint16_t>(sq * (GM::W + GM::H) + b);
        }
        base += total;
      }
    }
    }
    const uint32_t k = base;
    sync();
    AZB_PROF_MARK(3);
    // std::shuffle (stl_algo.h:3729-3792): for i = 1 .. k-1 in pairs, ONE draw x in [0, (i+1)(i+2)) gives the two swap partners
    // x / (i+2), x % (i+2) (an even k first spends one draw on i = 1 alone).  The draws are independent of the swaps, and draw j only
    // needs the stream advanced j times, which is a multiply-add (kPcgJump): every lane makes the draws of its pairs - state, output
    // function, Lemire's multiply, the division - side by side, and lane 0 is left with the swaps alone (round 4: 9.5 -> ~4 us of a
    // Tawlbwrdd simulation).  Lemire's rejection (probability range / 2^32 per draw, ~2e-4 per expansion) would shift every later
    // draw: any lane that meets its precondition sends the whole expansion down the sequential path below.
    bool shuffled = false;
    static_assert(GM::MAXK <= 512, "the jump table holds 260 draws");
    if (k > 1) {
      const uint64_t s0 = rng.state;
      const uint32_t d0 = (k & 1u) == 0 ? 1u : 0u, i0 = 1u + d0, np = (k - i0) >> 1;
      bool redo = false;
      for (uint32_t j = lane; j < np; j += G) {
        const uint32_t i = i0 + 2u * j, b1 = i + 2u, range = (i + 1u) * b1;
        const uint64_t old = kPcgJump.a[d0 + j] * s0 + kPcgJump.c[d0 + j];
        const uint32_t xs = static_cast<uint32_t>(((old >> 18u) ^ old) >> 27u), rot = static_cast<uint32_t>(old >> 59u);
        const uint32_t out = (xs >> rot) | (xs << ((32u - rot) & 31u));
        const uint64_t product = static_cast<uint64_t>(out) * static_cast<uint64_t>(range);
        if (static_cast<uint32_t>(product) < range) redo = true;
        const uint32_t x = static_cast<uint32_t>(product >> 32);
        sm.n[j] = (x / b1) | ((x % b1) << 16);
      }
      if (__ballot(redo) == 0ull) {
        sync();
        if (lane == 0) {
          uint32_t i = 1;
          if (d0) {       // (range 2: Lemire's threshold is 0, no rejection)
            const uint64_t old = s0;
            const uint32_t xs = static_cast<uint32_t>(((old >> 18u) ^ old) >> 27u), rot = static_cast<uint32_t>(old >> 59u);
            const uint32_t out = (xs >> rot) | (xs << ((32u - rot) & 31u));
            const uint32_t j = static_cast<uint32_t>((static_cast<uint64_t>(out) * 2ull) >> 32);
            const uint16_t t = sm.moves[i]; sm.moves[i] = sm.moves[j]; sm.moves[j] = t;
            ++i;
          }
          for (uint32_t j = 0; j < np; ++j) {
            const uint32_t pk = sm.n[j], p0 = pk & 0xFFFFu, p1 = pk >> 16;
            uint16_t t = sm.moves[i]; sm.moves[i] = sm.moves[p0]; sm.moves[p0] = t; ++i;
            t = sm.moves[i]; sm.moves[i] = sm.moves[p1]; sm.moves[p1] = t; ++i;
          }
        }
        rng.state = kPcgJump.a[d0 + np] * s0 + kPcgJump.c[d0 + np];
        shuffled = true;
        sync();
      }
    }
    if (k > 1 && !shuffled) {
      if (lane == 0) {
        uint32_t i = 1;
        if ((k & 1u) == 0) {
          const uint32_t j = lemire_below(rng, 2);
          const uint16_t t = sm.moves[i]; sm.moves[i] = sm.moves[j]; sm.moves[j] = t;
          ++i;
        }
        while (i != k) {
          const uint32_t sr = i + 1, b1 = sr + 1;
          const uint32_t x = lemire_below(rng, sr * b1);
          const uint32_t p0 = x / b1, p1 = x % b1;
          uint16_t t = sm.moves[i]; sm.moves[i] = sm.moves[p0]; sm.moves[p0] = t; ++i;
          t = sm.moves[i]; sm.moves[i] = sm.moves[p1]; sm.moves[p1] = t; ++i;
        }
      }
      const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(rng.state)));
      const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(rng.state >> 32)));
      rng.state = static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32);
      sync();
    }
    AZB_PROF_MARK(4);
    const uint32_t c0 = AZB_SEL(t_bump, seat);
    // the bump pointer lives in one half of the arena (k_compact ping-pongs between them)
    const uint32_t limit = ep.half_nodes ? ((c0 - 1) / ep.half_nodes + 1) * ep.half_nodes : ep.cap;
    if (c0 + k > limit) { raise(1u); return false; }
    for (uint32_t i = lane; i < k; i += G) {
      const size_t ci = tb + c0 + i;
      ar.N[ci] = 0; ar.Q[ci] = 0.0f; ar.Pr[ci] = 0.0f; ar.D[ci] = 0.0f; ar.V[ci] = 0.0f;
      ar.META[ci] = meta_pack(0, 0, sm.moves[i], 0, 0);
    }
    set_seat(t_bump, seat, c0 + k);
    if (lane == 0) ar.META[tb + node] = meta_pack(c0, k, meta_mv(meta_keep), meta_player(meta_keep), meta_term(meta_keep));
    c0_out = c0; k_out = k;
    sync();
    AZB_PROF_MARK(5);
    return true;
  }

  // in-order sum of sm.f0[0..k): every lane walks the same LDS words (broadcast reads), result uniform
  // (the ADDS are a dependent chain by definition - ((f0[0] + f0[1]) + f0[2]) + ... -, the LDS reads are not: sixteen elements are
  // asked for at once, four broadcast ds_read_b128, instead of one dependent read per element.  Round 4: 11.5 -> ~2 us of a
  // Tawlbwrdd simulation's 78.  Elements at or beyond k are read - f0 is followed by f1 - and never added.)
  __device__ __forceinline__ float seq_sum_f0(uint32_t k) const {
    static_assert(GM::MAXK % 16 == 0, "whole 16-element steps inside f0 / f1");
    float s = 0.0f;
    const float4* const p4 = reinterpret_cast<const float4*>(sm.f0);
    for (uint32_t i = 0; i < k; i += 16) {
      const float4 a = p4[i / 4], b = p4[i / 4 + 1], c = p4[i / 4 + 2], d = p4[i / 4 + 3];
      const float v[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
      if (i + 16 <= k) {
#pragma unroll
        for (int j = 0; j < 16; ++j) s += v[j];
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) if (i + j < k) s += v[j];
      }
    }
    return s;
  }

  // ======================= Gumbel AlphaZero (mcts.cc:24-401), wide-game form ===============================
  // Same per-tree record as SlotCtx (engine_kernels.h); children are walked in chunks of 64 through LDS, the
  // (<= 64) surviving root candidates of the sequential halving are lane-resident.
  enum { kGumTarget = 0, kGumInit = 1, kGumNSurv = 2, kGumPhase = 3, kGumSims = 4, kGumMEff = 5, kGumRemain = 6 };
  __device__ __forceinline__ uint32_t* gum_state(uint32_t seat) const { return ar.gum_state + (static_cast<size_t>(slot) * P + seat) * 8; }
  __device__ __forceinline__ float* gum_g(uint32_t seat) const { return ar.gum_g + (static_cast<size_t>(slot) * P + seat) * ep.gum_stride; }
  __device__ __forceinline__ uint16_t* gum_surv(uint32_t seat) const { return ar.gum_surv + (static_cast<size_t>(slot) * P + seat) * kGumMaxM; }
  __device__ __forceinline__ void reset_gumbel_state(uint32_t seat) const {
    if (lane != 0) return;
    uint32_t* st = gum_state(seat);
    st[kGumInit] = 0; st[kGumNSurv] = 0; st[kGumPhase] = 0; st[kGumSims] = 0; st[kGumMEff] = 0; st[kGumRemain] = 0;
  }
  __device__ __forceinline__ void set_gumbel_num_sims(uint32_t seat, uint32_t n) const {
    if (lane == 0) gum_state(seat)[kGumTarget] = n;
    reset_gumbel_state(seat);
  }
  __device__ __forceinline__ void set_gumbel_target() const {   // play_manager.cc:525-539 / 561-570
    if (!ep.gumbel_on) return;
    const uint32_t cp = gs.player;
    const uint32_t target = (flags & kFlagCapped) ? (ep.fast_gumbel ? seat_cap_visits(cp) : 0u) : seat_visits(cp);
    set_gumbel_num_sims(cp, target);
  }
  __device__ __forceinline__ float gumbel01() { return 0.0f - 1.0f * az_logf(-az_logf(1.0f - canonical01(rng))); }
  __device__ __forceinline__ uint32_t wave_max(uint32_t x) const {
    for (int off = 32; off > 0; off >>= 1) x = max(x, __shfl_xor(x, off, 64));
    return x;
  }
  // init_gumbel_state, mcts.cc:190-227
  __device__ __forceinline__ void init_gumbel_state(uint32_t seat, size_t tb, uint32_t c0, uint32_t k) {
    if (k == 0) return;
    uint32_t* st = gum_state(seat);
    const uint32_t target = st[kGumTarget], depth = AZB_SEL(t_depth, seat);
    const uint32_t remaining = depth < target ? target - depth : 0u;
    if (remaining == 0) return;
    const uint32_t gm = seat_gumbel_m(seat);
    uint32_t m_eff = gm < k ? gm : k;
    m_eff = m_eff < remaining ? m_eff : remaining;
    m_eff = m_eff > 1u ? m_eff : 1u;
    for (uint32_t i = lane; i < k; i += G) sm.f2[i] = ar.Pr[tb + c0 + i];
    sync();
    float* gg = gum_g(seat);
    for (uint32_t i = 0; i < k; ++i) {           // k draws in child order from the slot's stream
      const float g = gumbel01();
      if (lane == 0) { gg[i] = g; sm.f0[i] = g + az_logf(sm.f2[i] + 1e-20f); }
    }
    sync();
    uint16_t* surv = gum_surv(seat);
    for (uint32_t i = lane; i < k; i += G) {     // rank in the descending order of g + log(prior), mcts.cc:214-221
      const float s = sm.f0[i];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < k; ++j) { const float sj = sm.f0[j]; rank += (sj > s || (sj == s && j < i)) ? 1u : 0u; }
      if (rank < m_eff) surv[rank] = static_cast<uint16_t>(i);
    }
    if (lane == 0) {
      st[kGumInit] = 1; st[kGumNSurv] = m_eff; st[kGumPhase] = 0; st[kGumSims] = 0; st[kGumMEff] = m_eff; st[kGumRemain] = remaining;
    }
    sync();
  }
  // score g + log(prior) + sigma * q_hat of survivor `lane` (mcts.cc:241-253, 385-397); lanes >= nsurv get -inf
  __device__ __forceinline__ float survivor_score(uint32_t seat, size_t tb, uint32_t c0, uint32_t nsurv, uint32_t max_visit, uint32_t& ci) const {
    ci = 0;
    if (lane >= nsurv) return -__builtin_inff();
    ci = gum_surv(seat)[lane];
    const size_t idx = tb + c0 + ci;
    const uint32_t n = ar.N[idx];
    const float sigma_scale = seat_sigma_scale(seat, max_visit);
    const float sc = gum_g(seat)[ci] + az_logf(ar.Pr[idx] + 1e-20f) + sigma_scale * (n > 0 ? ar.Q[idx] : 0.0f);
    return sc != sc ? -__builtin_inff() : sc;
  }
  // gumbel_next_root_child (mcts.cc:266-283) with gumbel_advance_phase (mcts.cc:229-264) inlined
  __device__ __forceinline__ uint32_t gumbel_next_root_child(uint32_t seat, size_t tb, uint32_t c0) {
    sync();
    uint32_t* st = gum_state(seat);
    uint16_t* surv = gum_surv(seat);
    uint32_t nsurv = st[kGumNSurv], phase = st[kGumPhase], sims = st[kGumSims];
    const uint32_t m_eff = st[kGumMEff], remain = st[kGumRemain];
    uint32_t num_c, v_per;
    const uint32_t nph = gum_plan(m_eff, remain, phase, num_c, v_per);
    if (phase < nph && sims >= num_c * v_per && phase + 1 < nph) {
      uint32_t next_c, next_v;
      gum_plan(m_eff, remain, phase + 1, next_c, next_v);
      if (next_c < nsurv) {
        const uint32_t my = lane < nsurv ? ar.N[tb + c0 + surv[lane]] : 0u;
        const uint32_t max_visit = wave_max(my);
        uint32_t ci;
        const float score = survivor_score(seat, tb, c0, nsurv, max_visit, ci);
        uint32_t rank = 0;
        for (uint32_t j = 0; j < nsurv; ++j) {
          const float sj = __shfl(score, static_cast<int>(j), 64);
          rank += (sj > score || (sj == score && j < lane)) ? 1u : 0u;
        }
        sync();
        if (lane < nsurv && rank < next_c) surv[rank] = static_cast<uint16_t>(ci);
        sync();
        nsurv = next_c;
      }
      ++phase; sims = 0;
    }
    uint32_t child = 0;
    if (nsurv != 0) { child = surv[sims % nsurv]; ++sims; }
    if (lane == 0) { st[kGumNSurv] = nsurv; st[kGumPhase] = phase; st[kGumSims] = sims; }
    return child;
  }
  // exp terms of softmax(log prior + sigma * completedQ) over the children staged in LDS (n: sm.n, q: f1, p: f2)
  // -> sm.f0[0..k); returns their in-order sum (mcts.cc:285-373 share this)
  __device__ __forceinline__ float gumbel_pi_prime(uint32_t seat, uint32_t k, float node_v) {
    float sum_visits = 0.0f, sum_priors_visited = 0.0f, weighted_num = 0.0f;   // compute_v_mix_from_children, mcts.cc:71-89
    for (uint32_t i = 0; i < k; ++i) {
      const uint32_t ni = sm.n[i];
      sum_visits += static_cast<float>(ni);
      if (ni > 0) { sum_priors_visited += sm.f2[i]; weighted_num += sm.f2[i] * sm.f1[i]; }
    }
    float v_mix = node_v;
    if (!(sum_priors_visited <= 0.0f)) {
      const float weighted_q = weighted_num / sum_priors_visited;
      v_mix = (node_v + sum_visits * weighted_q) / (sum_visits + 1.0f);
    }
    uint32_t mv = 0;
    for (uint32_t i = lane; i < k; i += G) mv = max(mv, sm.n[i]);
    const uint32_t max_visit = wave_max(mv);
    const float sigma_scale = seat_sigma_scale(seat, max_visit);
    float z_max = -__builtin_inff();
    for (uint32_t i = lane; i < k; i += G) {
      const float z = az_logf(sm.f2[i] + 1e-20f) + sigma_scale * (sm.n[i] > 0 ? sm.f1[i] : v_mix);
      sm.f0[i] = z;
      if (z > z_max) z_max = z;
    }
    for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(z_max, off, 64); if (o > z_max) z_max = o; }
    sync();
    for (uint32_t i = lane; i < k; i += G) sm.f0[i] = az_expf(sm.f0[i] - z_max);
    sync();
    return seq_sum_f0(k);
  }
  // gumbel_interior_select, mcts.cc:285-334
  __device__ __forceinline__ uint32_t gumbel_interior_select(uint32_t seat, size_t tb, uint32_t c0, uint32_t k, float node_v) {
    for (uint32_t i = lane; i < k; i += G) {
      const size_t ci = tb + c0 + i;
      sm.n[i] = ar.N[ci]; sm.f1[i] = ar.Q[ci]; sm.f2[i] = ar.Pr[ci];
    }
    sync();
    const float z_sum = gumbel_pi_prime(seat, k, node_v);
    uint32_t sv = 0;
    for (uint32_t i = lane; i < k; i += G) sv += sm.n[i];
    const uint32_t sum_visits = wave_sum(sv);
    const float inv = z_sum > 0 ? (1.0f / z_sum) : 0.0f;
    const float denom = 1.0f + static_cast<float>(sum_visits);
    float best_s = -__builtin_inff();
    uint32_t best_i = 0xFFFFFFFFu;
    for (uint32_t i = lane; i < k; i += G) {
      float sc = sm.f0[i] * inv - static_cast<float>(sm.n[i]) / denom;
      if (sc != sc) sc = -__builtin_inff();
      if (sc > best_s || best_i == 0xFFFFFFFFu) { best_s = sc; best_i = i; }
    }
    for (int off = 1; off < 64; off <<= 1) {
      const float os = __shfl_xor(best_s, off, 64);
      const uint32_t oi = __shfl_xor(best_i, off, 64);
      if (oi != 0xFFFFFFFFu && (best_i == 0xFFFFFFFFu || os > best_s || (os == best_s && oi < best_i))) { best_s = os; best_i = oi; }
    }
    sync();
    return best_i;
  }
  // gumbel_improved_policy into sm.dense (mcts.cc:336-373); the root must be staged (stage_root)
  __device__ __forceinline__ void gumbel_improved_policy(uint32_t seat, uint32_t k, float root_v) {
    dense_zero();
    if (k == 0) return;
    const float z_sum = gumbel_pi_prime(seat, k, root_v);
    if (z_sum <= 0) return;
    for (uint32_t i = lane; i < k; i += G) sm.dense[sm.moves[i]] = sm.f0[i] / z_sum;
    sync();
  }
  // gumbel_final_action (mcts.cc:375-401); the root must be staged; returns the move
  __device__ __forceinline__ uint32_t gumbel_final_action(uint32_t seat, size_t tb, uint32_t c0, uint32_t k) {
    sync();
    const uint32_t* st = gum_state(seat);
    const uint32_t nsurv = st[kGumNSurv];
    if (!st[kGumInit] || nsurv == 0) { probs(0.0f, k); return pick_move(); }
    uint32_t mv = 0;
    for (uint32_t i = lane; i < k; i += G) mv = max(mv, sm.n[i]);
    const uint32_t max_visit = wave_max(mv);
    uint32_t ci;
    float score = survivor_score(seat, tb, c0, nsurv, max_visit, ci);
    uint32_t pos = lane < nsurv ? lane : 0xFFFFu;
    for (int off = 1; off < 64; off <<= 1) {     // first survivor with the strictly largest score
      const float os = __shfl_xor(score, off, 64);
      const uint32_t op = __shfl_xor(pos, off, 64);
      const uint32_t oc = __shfl_xor(ci, off, 64);
      if (op != 0xFFFFu && (pos == 0xFFFFu || os > score || (os == score && op < pos))) { score = os; pos = op; ci = oc; }
    }
    return sm.moves[ci];
  }

  // ---- Node::best_child over k children starting at c0 -------------------------------------------------------
  // (`nif` = Node::n_in_flight per arena node, only given by the WU-UCT batched API; `n_parent` then includes the parent's)
  __device__ __forceinline__ uint32_t select_child(size_t tb, uint32_t c0, uint32_t k, float v_parent, uint32_t n_parent, float fpu_reduction,
                                                   const uint32_t* nif = nullptr) {
    for (uint32_t i = lane; i < k; i += G) {
      const size_t ci = tb + c0 + i;
      const uint32_t n = ar.N[ci];
      const float p = ar.Pr[ci];
      sm.n[i] = n; sm.f1[i] = ar.Q[ci]; sm.f2[i] = p;
      sm.f0[i] = n > 0 ? p : 0.0f;
    }
    sync();
    AZB_PROF_MARK(12);
    // reference: seen_policy += c.policy only for visited children, in child order; adding the 0.0f of an
    // unvisited child leaves the running sum unchanged, so the masked in-order sum is identical
    const float seen = seq_sum_f0(k);
    AZB_PROF_MARK(13);
    const float fpu_value = v_parent - fpu_reduction * sqrtf(seen);
    const float sqrt_n = sqrtf(static_cast<float>(n_parent));
    float best_u = -__builtin_inff();
    uint32_t best_i = 0xFFFFFFFFu;
    for (uint32_t i = lane; i < k; i += G) {
      const uint32_t n = sm.n[i];
      const uint32_t fl = nif ? nif[c0 + i] : 0u;
      float u = (n == 0 ? fpu_value : sm.f1[i]) + ep.cpuct * sm.f2[i] * sqrt_n / static_cast<float>(n + fl + 1);
      if (u != u) u = (i == 0) ? __builtin_inff() : -__builtin_inff();
      if (u > best_u || best_i == 0xFFFFFFFFu) { best_u = u; best_i = i; }
    }
    AZB_PROF_MARK(14);
    for (int off = 1; off < 64; off <<= 1) {
      const float ou = __shfl_xor(best_u, off, 64);
      const uint32_t oi = __shfl_xor(best_i, off, 64);
      if (oi != 0xFFFFFFFFu && (best_i == 0xFFFFFFFFu || ou > best_u || (ou == best_u && oi < best_i))) { best_u = ou; best_i = oi; }
    }
    sync();
    return best_i;
  }

  // ---- MCTS::find_leaf ------------------------------------------------------------------------------------------
  __device__ __forceinline__ bool find_leaf(uint32_t seat, typename GM::State& leaf, uint32_t& term) {
    sync();
    AZB_PROF_DECL;
    const size_t tb = tree_base(seat);
    const uint32_t root = AZB_SEL(t_root, seat);
    cur = root; plen = 0;
    leaf = gs;
    uint32_t path_len = 0;
    bool base_valid = true;
    uint64_t meta = ar.META[tb + cur];
    uint32_t n = ar.N[tb + cur];
    uint32_t* path = ar.path + static_cast<size_t>(slot) * ep.max_depth;
    uint32_t gum_active = 0;
    if (seat_gumbel(seat)) {  // lazy init, mcts.cc:465-472 (MCTS::gumbel_enabled_ of this seat's tree)
      gum_active = gum_state(seat)[kGumInit];
      if (!gum_active && gum_state(seat)[kGumTarget] > 0 && n > 0 && meta_nch(meta) != 0) {
        init_gumbel_state(seat, tb, meta_ch0(meta), meta_nch(meta));
        gum_active = gum_state(seat)[kGumInit];
      }
    }
    while (n > 0 && meta_term(meta) == 0) {
      if (plen >= ep.max_depth) { raise(8u); return false; }
      if (lane == 0) path[plen] = cur;
      ++plen;
      const uint32_t k = meta_nch(meta), c0 = meta_ch0(meta);
      if (k == 0) { raise(8u); return false; }
      const float fpu = (cur == root && seat_fpu_zero(seat)) ? 0.0f : ep.fpu_reduction;
      uint32_t best;
      if (gum_active && cur == root) best = gumbel_next_root_child(seat, tb, c0);
      else if (gum_active && seat_gumbel_full(seat)) best = gumbel_interior_select(seat, tb, c0, k, ar.V[tb + cur]);
      else best = select_child(tb, c0, k, ar.V[tb + cur], n, fpu);
      AZB_PROF_MARK(14);
      cur = c0 + best;
      n = ar.N[tb + cur];
      meta = ar.META[tb + cur];
#ifdef AZMI_BIG_PROF
      asm volatile("s_waitcnt vmcnt(0)" :: "v"(n), "v"(meta) : "memory");
#endif
      AZB_PROF_MARK(10);
      if (!step_state(leaf, meta_mv(meta), path_list(), path_len, base_valid, glen)) { raise(64u); return false; }
      AZB_PROF_MARK(11);
    }
    leaf_rep_len = path_len; leaf_base_valid = base_valid;
#pragma unroll
    for (int p = 0; p < P; ++p) if (static_cast<uint32_t>(p) == seat) t_tld[p] += plen;
    term = meta_term(meta);
    AZB_PROF_MARK(2);
    if (n == 0) {
      term = GM::terminal(leaf);
#ifdef AZMI_BIG_PROF_TERMINAL
      AZB_PROF_MARK(0);      // (experiment: the terminal test's share lands in [0] "load")
#endif
      const uint64_t keep = meta_pack(0, 0, meta_mv(meta), leaf.player, term);
      uint32_t c0, k;
      if (!expand_node(seat, cur, leaf, keep, c0, k)) return false;
    }
    return true;
  }

  // ---- MCTS::find_leaf_batched (WU-UCT), mcts.cc:752-784; see SlotCtx::find_leaf_wu ---------------------------------
  __device__ __forceinline__ bool find_leaf_wu(uint32_t seat, typename GM::State& leaf, uint32_t& term, uint32_t* nif) {
    sync();
    const size_t tb = tree_base(seat);
    const uint32_t root = AZB_SEL(t_root, seat);
    cur = root; plen = 0;
    leaf = gs;
    uint32_t path_len = 0;
    bool base_valid = true;
    uint64_t meta = ar.META[tb + cur];
    uint32_t n = ar.N[tb + cur], nf = nif[cur];
    uint32_t* path = ar.path + static_cast<size_t>(slot) * ep.max_depth;
    while ((n > 0 || nf > 0) && meta_nch(meta) != 0 && meta_term(meta) == 0) {
      if (plen >= ep.max_depth) { raise(8u); return false; }
      if (lane == 0) path[plen] = cur;
      ++plen;
      const uint32_t k = meta_nch(meta), c0 = meta_ch0(meta);
      const float fpu = (cur == root && seat_fpu_zero(seat)) ? 0.0f : ep.fpu_reduction;
      const uint32_t best = select_child(tb, c0, k, ar.V[tb + cur], n + nf, fpu, nif);
      if (lane == 0) nif[cur] = nf + 1;
      cur = c0 + best;
      n = ar.N[tb + cur];
      nf = nif[cur];
      meta = ar.META[tb + cur];
      if (!step_state(leaf, meta_mv(meta), path_list(), path_len, base_valid, glen)) { raise(64u); return false; }
    }
    if (lane == 0) nif[cur] = nf + 1;
#pragma unroll
    for (int p = 0; p < P; ++p) if (static_cast<uint32_t>(p) == seat) t_tld[p] += plen;
    term = meta_term(meta);
    if (n == 0 && meta_nch(meta) == 0) {
      term = GM::terminal(leaf);
      const uint64_t keep = meta_pack(0, 0, meta_mv(meta), leaf.player, term);
      uint32_t c0, k;
      if (!expand_node(seat, cur, leaf, keep, c0, k)) return false;
      for (uint32_t i = lane; i < k; i += G) nif[c0 + i] = 0;
    }
    sync();
    return true;
  }

  // ---- MCTS::add_root_noise over the children whose priors sit in sm.f0[0..k) --------------------------------------
  __device__ __forceinline__ void add_root_noise(uint32_t k, float eps) {
    double sum = 0.0;
    if (ep.shaped && k > 1) {
      const float Nf = static_cast<float>(k);
      for (uint32_t i = lane; i < k; i += G) sm.f1[i] = az_logf(fminf(sm.f0[i], 0.01f) + 1e-20f);
      sync();
      float log_sum = 0.0f;
      for (uint32_t i = 0; i < k; ++i) log_sum += sm.f1[i];
      const float log_mean = log_sum / Nf;
      sync();
      for (uint32_t i = lane; i < k; i += G) sm.f1[i] = fmaxf(0.0f, sm.f1[i] - log_mean);
      sync();
      float shaped_sum = 0.0f;
      for (uint32_t i = 0; i < k; ++i) shaped_sum += sm.f1[i];
      const float uniform = 1.0f / Nf;
      for (uint32_t i = 0; i < k; ++i) {
        const float shaped = sm.f1[i];
        float alpha_prop = (shaped_sum > 0) ? 0.5f * (shaped / shaped_sum + uniform) : uniform;
        alpha_prop = fmaxf(alpha_prop, 1e-6f);
        Gamma dist(kNoiseAlphaRatio * alpha_prop);
        const float g = dist.draw(rng);
        if (lane == 0) sm.f2[i] = g;
        sum += g;
      }
    } else {
      Gamma dist(kNoiseAlphaRatio / static_cast<float>(k));
      for (uint32_t i = 0; i < k; ++i) {
        const float g = dist.draw(rng);
        if (lane == 0) sm.f2[i] = g;
        sum += g;
      }
    }
    sync();
    for (uint32_t i = lane; i < k; i += G)
      sm.f0[i] = sm.f0[i] * (1 - eps) + eps * sm.f2[i] / static_cast<float>(sum);
    sync();
  }

  // ---- playout_eval, game_state.cc:10-54, for the wide games: the scores of a uniformly random rollout from the leaf go
  // to the slot's v row (process_result gives the leaf the uniform policy itself).  One draw of the slot's rollout stream
  // per move picks the lemire_below(#legal)-th legal move in ascending move order; the repetition bookkeeping continues
  // the descent's path-local list, so the rollout sees the same counts as the reference's copied game state.
  __device__ __forceinline__ void playout_eval(const typename GM::State& leaf) {
    Pcg32 roll;
    roll.state = ar.roll[slot];
    typename GM::State sim = leaf;
    uint32_t rep_len = leaf_rep_len;
    bool base_valid = leaf_base_valid;
    uint32_t term = GM::terminal(sim);
    if constexpr (kSG) {
      while (term == 0) {
        const uint32_t k = GM::gen_valid(sim, lane, sm.rules);
        if (k == 0) break;
        const uint32_t r = lemire_below(roll, k);
        // the r-th set bit of the map, in ascending move order
        const unsigned long long w = lane < 27 ? sm.rules.vbits[lane] : 0ull;
        const uint32_t cnt = static_cast<uint32_t>(__builtin_popcountll(w));
        uint32_t in = cnt;
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t o = __shfl_up(in, off, 64);
          if (lane >= static_cast<uint32_t>(off)) in += o;
        }
        const uint32_t lo = in - cnt;
        uint32_t mine = 0xFFFFFFFFu;
        if (r >= lo && r < lo + cnt) {
          unsigned long long m = w;
          for (uint32_t j = lo; j < r; ++j) m &= m - 1;
          mine = lane * 64 + static_cast<uint32_t>(__builtin_ctzll(m));
        }
        const uint64_t owner = __ballot(mine != 0xFFFFFFFFu);
        const uint32_t mv = __shfl(mine, static_cast<int>(__builtin_ctzll(owner)), 64);
        sync();
        if (!step_state(sim, mv, path_list(), rep_len, base_valid, glen)) break;
        term = GM::terminal(sim);
      }
    } else {
    while (term == 0) {
      uint32_t mask[2], cnt[2], incl[2], total[2];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const uint32_t sq = half * 64 + lane;
        mask[half] = (sq < static_cast<uint32_t>(GM::SQ) && GM::own_piece(sim, sim.player, sq)) ? GM::slide_mask(sim, sq) : 0u;
        cnt[half] = __builtin_popcount(mask[half]);
        uint32_t in = cnt[half];
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t o = __shfl_up(in, off, 64);
          if (lane >= static_cast<uint32_t>(off)) in += o;
        }
        incl[half] = in;
        total[half] = __shfl(in, 63, 64);
      }
      const uint32_t k = total[0] + total[1];
      if (k == 0) break;
      const uint32_t r = lemire_below(roll, k);
      uint32_t mine = 0xFFFFFFFFu;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const uint32_t lo = (half ? total[0] : 0u) + incl[half] - cnt[half];
        if (r >= lo && r < lo + cnt[half]) {
          uint32_t m = mask[half];
          for (uint32_t j = lo; j < r; ++j) m &= m - 1;
          mine = (half * 64 + lane) * (GM::W + GM::H) + __builtin_ctz(m);
        }
      }
      const uint64_t owner = __ballot(mine != 0xFFFFFFFFu);
      const uint32_t mv = __shfl(mine, static_cast<int>(__builtin_ctzll(owner)), 64);
      if (!step_state(sim, mv, sm.plist, rep_len, base_valid, glen)) break;
      term = GM::terminal(sim);
    }
    }
    if (lane <= static_cast<uint32_t>(P))
      ar.v[static_cast<size_t>(slot) * (P + 1) + lane] = term ? ((term - 1 == lane) ? 1.0f : 0.0f) : static_cast<float>(1.0 / (P + 1));
    if (lane == 0) ar.roll[slot] = roll.state;
    sync();
  }

  // ---- MCTS::process_result --------------------------------------------------------------------------------------------
  __device__ __forceinline__ void process_result(uint32_t seat, bool from_net, bool root_noise) {
    sync();
    const size_t tb = tree_base(seat);
    const uint32_t root = AZB_SEL(t_root, seat);
    const uint64_t meta = ar.META[tb + cur];
    const uint32_t term = meta_term(meta);
    float val[P + 1];
    if (term != 0) {
#pragma unroll
      for (int i = 0; i <= P; ++i) val[i] = (static_cast<int>(term) - 1 == i) ? 1.0f : 0.0f;
    } else {
      const uint32_t k = meta_nch(meta), c0 = meta_ch0(meta);
      if (from_net) {
#pragma unroll
        for (int i = 0; i <= P; ++i) val[i] = ar.v[static_cast<size_t>(slot) * (P + 1) + i];
      } else {
#pragma unroll
        for (int i = 0; i <= P; ++i) val[i] = static_cast<float>(1.0 / (P + 1));
      }
      if constexpr (GM::kRelative) {   // relative_to_absolute(value, current_->player), mcts.cc:522-524, game_state.h:37-46 (P == 2: a swap)
        if (meta_player(meta) == 1) { const float t = val[0]; val[0] = val[1]; val[1] = t; }
      }
      const float ksum = static_cast<float>(k & 0xFFu);  // dumb_eval: u8 sum wraps (game_state.h:167, shapes.h:14)
      const bool pi_rows = from_net && !seat_eval_playout(seat);   // a PLAYOUT seat's rows hold the rollout's scores only
      const bool is_root = cur == root;
      const float root_temp = seat_root_temp(seat);
      const bool root_pow = is_root && root_temp != 1.0f;
      for (uint32_t i = lane; i < k; i += G) {
        float p;
        if (pi_rows) p = ar.pi[static_cast<size_t>(slot) * M + meta_mv(ar.META[tb + c0 + i])];
        else p = (ksum == 0.0f) ? 0.0f : 1.0f / ksum;
        if (root_pow) p = az_powf(p, 1.0f / root_temp);
        sm.f0[i] = p;
      }
      sync();
      const float sum = seq_sum_f0(k);
      sync();
      for (uint32_t i = lane; i < k; i += G) sm.f0[i] = sm.f0[i] / sum;
      sync();
      if (is_root && root_noise && !seat_gumbel(seat)) add_root_noise(k, seat_eps(seat));
      for (uint32_t i = lane; i < k; i += G) ar.Pr[tb + c0 + i] = sm.f0[i];
    }
    const uint32_t* path = ar.path + static_cast<size_t>(slot) * ep.max_depth;
    const float draw_share = val[P] / static_cast<int32_t>(P);
    for (uint32_t base = 0; base < plen; base += G) {
      const uint32_t i = base + lane;
      if (i < plen) {
        const uint32_t node = (i == plen - 1) ? cur : path[i + 1];
        const uint32_t pp = meta_player(ar.META[tb + path[i]]);
        const size_t ni = tb + node;
        const float vv = ((pp == 0) ? val[0] : val[1]) + draw_share;
        const uint32_t nn = ar.N[ni];
        const float q = ar.Q[ni], d = ar.D[ni];
        ar.Q[ni] = (q * static_cast<float>(nn) + vv) / static_cast<float>(nn + 1);
        ar.D[ni] = (d * static_cast<float>(nn) + val[P]) / static_cast<float>(nn + 1);
        if (nn == 0) ar.V[ni] = ((meta_player(ar.META[ni]) == 0) ? val[0] : val[1]) + draw_share;
        ar.N[ni] = nn + 1;
      }
    }
    if (lane == 0) {
      const size_t ri = tb + root;
      const uint32_t rn = ar.N[ri];
      if (rn == 0) {
        ar.V[ri] = ((meta_player(ar.META[ri]) == 0) ? val[0] : val[1]) + draw_share;
        ar.D[ri] = val[P];
      }
      ar.N[ri] = rn + 1;
      atomicAdd(reinterpret_cast<unsigned long long*>(&ar.c_sims[slot]), 1ull);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) if (static_cast<uint32_t>(p) == seat) t_depth[p] += 1;
    sync();
  }

  // ---- dense [M] helpers on sm.dense ------------------------------------------------------------------------------------
  // f(m, x) for every NONZERO entry of sm.dense in ascending index order (m and x wave-uniform); f returns false to stop.
  // The reference's dense vectors are sums / scans over all M entries in index order (mcts.cc:575-735); an entry that is +-0
  // changes neither a running fp32 sum nor a `sum > choice` test, so walking the ~100 nonzero ones gives the same bits as
  // walking all 2662 - and the move step, whose wave every round's launch waits for, spent most of its time in those walks.
  template <class F>
  __device__ __forceinline__ void dense_for_nonzero(F&& f) const {
    for (uint32_t base = 0; base < static_cast<uint32_t>(M); base += G) {
      const uint32_t m = base + lane;
      const float x = m < static_cast<uint32_t>(M) ? sm.dense[m] : 0.0f;
      unsigned long long mask = __ballot(x != 0.0f);
      while (mask) {
        const int b = __builtin_ctzll(mask);
        mask &= mask - 1;
        if (!f(base + static_cast<uint32_t>(b), __shfl(x, b, 64))) return;
      }
    }
  }
  __device__ __forceinline__ float dense_seq_sum() const {
    float s = 0.0f;
    dense_for_nonzero([&](uint32_t, float x) { s += x; return true; });
    return s;
  }
  __device__ __forceinline__ void dense_zero() { for (uint32_t m = lane; m < static_cast<uint32_t>(M); m += G) sm.dense[m] = 0.0f; sync(); }
  __device__ __forceinline__ void dense_div(float s) { for (uint32_t m = lane; m < static_cast<uint32_t>(M); m += G) sm.dense[m] = sm.dense[m] / s; sync(); }
  __device__ __forceinline__ void dense_pow(float e) {
    if (e == 1.0f) return;
    if (e > 0.0f && staged_k != 0) {
      // every nonzero entry of sm.dense sits at the move of a staged root child (all its builders scatter through sm.moves),
      // and pow(0, e > 0) = 0: the double-precision pow runs over the k children, not over the M entries
      for (uint32_t i = lane; i < staged_k; i += G) { const uint32_t m = sm.moves[i]; sm.dense[m] = az_powf(sm.dense[m], e); }
      sync();
      return;
    }
    for (uint32_t m = lane; m < static_cast<uint32_t>(M); m += G) sm.dense[m] = az_powf(sm.dense[m], e);
    sync();
  }
  // root children staged in LDS: moves, n (sm.n), q (f1), p (f2)
  __device__ __forceinline__ void stage_root(size_t tb, uint32_t c0, uint32_t k) {
    staged_k = k;
    for (uint32_t i = lane; i < k; i += G) {
      const size_t ci = tb + c0 + i;
      sm.moves[i] = static_cast<uint16_t>(meta_mv(ar.META[ci]));
      sm.n[i] = ar.N[ci]; sm.f1[i] = ar.Q[ci]; sm.f2[i] = ar.Pr[ci];
    }
    sync();
  }
  // MCTS::probs into sm.dense (mcts.cc:575-618)
  __device__ __forceinline__ void probs(float temp, uint32_t k) {
    dense_zero();
    for (uint32_t i = lane; i < k; i += G) sm.dense[sm.moves[i]] = static_cast<float>(sm.n[i]);
    sync();
    const float count_sum = dense_seq_sum();
    if (count_sum == 0) {
      sync();
      for (uint32_t i = lane; i < k; i += G) sm.dense[sm.moves[i]] = sm.f2[i];
      sync();
      if (temp != 0.0f) dense_pow(1.0f / temp);
      dense_div(dense_seq_sum());
      return;
    }
    if (temp == 0) {
      float best = 0.0f;
      for (uint32_t m = lane; m < static_cast<uint32_t>(M); m += G) best = fmaxf(best, sm.dense[m]);
      for (int off = 32; off > 0; off >>= 1) best = fmaxf(best, __shfl_xor(best, off, 64));
      uint32_t nb = 0;
      for (uint32_t m = lane; m < static_cast<uint32_t>(M); m += G) nb += sm.dense[m] == best;
      nb = wave_sum(nb);
      const float pv = static_cast<float>(1.0 / static_cast<double>(nb));
      sync();
      for (uint32_t m = lane; m < static_cast<uint32_t>(M); m += G) sm.dense[m] = (sm.dense[m] == best) ? pv : 0.0f;
      sync();
      return;
    }
    dense_div(count_sum);
    dense_pow(1 / temp);
    dense_div(dense_seq_sum());
  }
  // MCTS::probs_pruned into sm.dense (mcts.cc:620-674), temp == 1 or general
  __device__ __forceinline__ void probs_pruned(float temp, uint32_t root_n, uint32_t k) {
    if (root_n <= 1) { probs(temp, k); return; }
    const float explore_scaling = ep.cpuct * sqrtf(static_cast<float>(root_n));
    float best_sel = -1e30f;
    for (uint32_t i = lane; i < k; i += G)
      if (sm.n[i] != 0) {
        const float sel = sm.f1[i] + explore_scaling * sm.f2[i] / static_cast<float>(sm.n[i] + 1);
        if (sel > best_sel) best_sel = sel;
      }
    for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(best_sel, off, 64); if (o > best_sel) best_sel = o; }
    dense_zero();
    for (uint32_t i = lane; i < k; i += G) {
      if (sm.n[i] == 0) continue;
      const float gap = best_sel - sm.f1[i];
      const float nf = static_cast<float>(sm.n[i]);
      const float desired = (gap <= 0) ? nf : explore_scaling * sm.f2[i] / gap - 1.0f;
      const float lo = (0.0f < desired) ? desired : 0.0f;
      sm.dense[sm.moves[i]] = (lo < nf) ? lo : nf;
    }
    sync();
    const float total = dense_seq_sum();
    if (total == 0) { sync(); probs(temp, k); return; }
    if (temp == 0) {
      float best = -__builtin_inff();       // the maximum over all entries (order does not matter for a maximum)
      for (uint32_t m = lane; m < static_cast<uint32_t>(M); m += G) best = (best < sm.dense[m]) ? sm.dense[m] : best;
      for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(best, off, 64); best = (best < o) ? o : best; }
      uint32_t cnt = 0;
      for (uint32_t m = lane; m < static_cast<uint32_t>(M); m += G) cnt += sm.dense[m] == best;
      cnt = wave_sum(cnt);
      sync();
      for (uint32_t m = lane; m < static_cast<uint32_t>(M); m += G) sm.dense[m] = (sm.dense[m] == best) ? 1.0f / static_cast<int32_t>(cnt) : 0.0f;
      sync();
      return;
    }
    dense_div(total);
    if (temp != 1.0f) { dense_pow(1.0f / temp); dense_div(dense_seq_sum()); }
  }
  __device__ __forceinline__ uint32_t pick_move() {  // mcts.cc:717-735 on sm.dense
    const float choice = canonical01(rng) * 1.0f + 0.0f;
    float sum = 0.0f;
    uint32_t picked = 0xFFFFFFFFu, last_pos = 0xFFFFFFFFu;
    dense_for_nonzero([&](uint32_t m, float x) {
      sum += x;
      if (x > 0) last_pos = m;
      if (sum > choice) { picked = m; return false; }
      return true;
    });
    if (picked != 0xFFFFFFFFu) return picked;
    if (last_pos != 0xFFFFFFFFu) return last_pos;     // rounding left the total below `choice`: the last positive entry
    raise(16u);
    return 0;
  }

  __device__ __forceinline__ bool update_root(uint32_t seat, uint32_t move) {
    sync();
    const size_t tb = tree_base(seat);
    const uint32_t root = AZB_SEL(t_root, seat);
    const uint64_t meta = ar.META[tb + root];
    uint32_t k = meta_nch(meta), c0 = meta_ch0(meta);
    if (k == 0 && !expand_node(seat, root, gs, meta, c0, k)) return false;
    uint32_t hit = 0xFFFFFFFFu;
    for (uint32_t i = lane; i < k; i += G)
      if (meta_mv(ar.META[tb + c0 + i]) == move) hit = i;
    for (int off = 32; off > 0; off >>= 1) hit = min(hit, __shfl_xor(hit, off, 64));
    if (hit == 0xFFFFFFFFu) { raise(32u); return false; }
    set_seat(t_root, seat, c0 + hit); set_seat(t_depth, seat, 0);
#pragma unroll
    for (int p = 0; p < P; ++p) if (static_cast<uint32_t>(p) == seat) t_tld[p] = 0;
    if (ep.gumbel_on) reset_gumbel_state(seat);  // mcts.cc:172
    return true;
  }

  __device__ __forceinline__ void reapply_root_prior(uint32_t seat, bool noise) {
    sync();
    const size_t tb = tree_base(seat);
    const uint32_t root = AZB_SEL(t_root, seat);
    if (ar.N[tb + root] == 0) return;
    const uint64_t meta = ar.META[tb + root];
    const uint32_t k = meta_nch(meta), c0 = meta_ch0(meta);
    const float root_temp = seat_root_temp(seat);
    const bool do_temp = root_temp != 1.0f;
    if (!do_temp && !(noise && k > 0)) return;
    for (uint32_t i = lane; i < k; i += G) {
      float p = ar.Pr[tb + c0 + i];
      if (do_temp) p = az_powf(p, 1.0f / root_temp);
      sm.f0[i] = p;
    }
    sync();
    if (do_temp) {
      const float sum = seq_sum_f0(k);
      sync();
      if (sum > 0.0f) for (uint32_t i = lane; i < k; i += G) sm.f0[i] = sm.f0[i] / sum;
      sync();
    }
    if (noise && k > 0) add_root_noise(k, seat_eps(seat));
    for (uint32_t i = lane; i < k; i += G) ar.Pr[tb + c0 + i] = sm.f0[i];
    sync();
  }

  __device__ __forceinline__ void start_game() {
    if constexpr (kSG) {   // randomize_start(): the variant of the new game, then its constructor's history entry (star_gambit_gs.cc:289)
      gs = GM::initial(GM::pick_variant(ep.sg_pinned, ep.sg_probs, coin), lane);
      const unsigned long long h0 = GM::position_hash(gs);
      if (lane == 0) game_list()[0] = h0;
      glen = 1;
    } else {
      gs = GM::initial();
      glen = 0;
    }
    for (uint32_t s = 0; s < static_cast<uint32_t>(P); ++s) reset_tree(s);
    ph_rows = 0;
  }
  __device__ __forceinline__ void draw_capped() {
    const bool capped = ep.cap_rand && (canonical01(coin) < ep.cap_percent);
    flags = capped ? (flags | kFlagCapped) : (flags & ~kFlagCapped);
  }

  __device__ __forceinline__ bool make_move(uint32_t cp) {
    sync();
    const size_t tb = tree_base(cp);
    const bool capped = flags & kFlagCapped;
    const uint32_t root = AZB_SEL(t_root, cp);
    const uint64_t rmeta = ar.META[tb + root];
    const uint32_t k = meta_nch(rmeta), c0 = meta_ch0(rmeta);
    const uint32_t root_n = ar.N[tb + root];
    stage_root(tb, c0, k);

    float temp = ep.start_temp;
    float half_life = ep.half_life;
    if constexpr (kSG) {   // temp_decay_half_life_by_variant, play_manager.cc:290-296
      const uint32_t vid = GM::variant(gs);
      if (vid < ep.n_half_life_v) half_life = vid == 0 ? ep.half_life_v[0] : vid == 1 ? ep.half_life_v[1] : vid == 2 ? ep.half_life_v[2] : ep.half_life_v[3];
    }
    if (half_life != 0) {
      const float lambda = 0.693f / half_life;
      temp -= ep.final_temp;
      temp *= az_expf(-lambda * gs.turn);
      temp += ep.final_temp;
    }
    int resign_entry = -1;
    if (ep.resign_percent > 0 && !(flags & kFlagPlaythrough)) {
      float q = 0, d = 0; bool found = false;  // MCTS::root_value: in-order scan, strict >
      for (uint32_t i = 0; i < k; ++i) {
        const float qi = sm.f1[i];
        if (sm.n[i] > 0 && qi > q) { q = qi; d = ar.D[tb + c0 + i]; found = true; }
      }
      if (!found && root_n > 0) { q = ar.V[tb + root]; d = ar.D[tb + root]; }
      const float w = q - d / static_cast<int32_t>(P);
      const float l = static_cast<float>(1.0 - static_cast<double>(w) - static_cast<double>(d));
      const double resign_val = 1.0 - static_cast<double>(ep.resign_percent);
      int entry = -1;
      if (w > resign_val) entry = static_cast<int>(cp);
      else if (l > resign_val) entry = static_cast<int>((cp + 1) % 2);
      else if (d > resign_val) entry = P;
      if (entry >= 0) {
        if (canonical01(coin) < ep.resign_playthrough) flags |= kFlagPlaythrough;
        else resign_entry = entry;
      }
    }
    // per-seat opt-in resign, play_manager.cc:335-366
    if (ep.seat_resign && resign_entry < 0 && !(flags & kFlagPlaythrough)) {
      const float seat_thresh = seat_resign_threshold(cp);
      if (seat_thresh > -2.0f) {
        float q = 0, d = 0; bool found = false;  // MCTS::root_value
        for (uint32_t i = 0; i < k; ++i) {
          const float qi = sm.f1[i];
          if (sm.n[i] > 0 && qi > q) { q = qi; d = ar.D[tb + c0 + i]; found = true; }
        }
        if (!found && root_n > 0) { q = ar.V[tb + root]; d = ar.D[tb + root]; }
        const float w = q - d / static_cast<int32_t>(P);
        const float l = static_cast<float>(1.0 - static_cast<double>(w) - static_cast<double>(d));
        const float v_self = w - l;
        uint32_t* streak = ar.resign_streak + static_cast<size_t>(slot) * P + cp;
        const uint32_t now = v_self <= seat_thresh ? *streak + 1u : 0u;
        sync();
        if (lane == 0) *streak = now;
        if (now >= seat_resign_need(cp)) resign_entry = static_cast<int>((cp + 1) % 2);
      }
    }
    const uint64_t rng_before = rng.state;
    uint32_t chosen;
    if (seat_gumbel(cp) && !capped) {          // play_manager.cc:367-402
      if (!seat_gumbel_g3(cp)) {
        chosen = gumbel_final_action(cp, tb, c0, k);   // G1 acting
      } else {   // G3 opt-in: sample from improved policy ^ (1 / temp)
        gumbel_improved_policy(cp, k, ar.V[tb + root]);
        if (temp != 1.0f && temp > 0.0f) {
          dense_pow(1.0f / temp);
          const float sg = dense_seq_sum();
          if (sg > 0) dense_div(sg);
        } else if (temp <= 0.0f) {   // arg-max of pi' (first maximum), as a one-hot vector
          float bv = -__builtin_inff(); uint32_t bi = 0xFFFFFFFFu;
          for (uint32_t m = lane; m < static_cast<uint32_t>(M); m += G) { const float x = sm.dense[m]; if (x > bv) { bv = x; bi = m; } }
          for (int off = 1; off < 64; off <<= 1) {
            const float ov = __shfl_xor(bv, off, 64); const uint32_t oi = __shfl_xor(bi, off, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
          }
          sync();
          dense_zero();
          if (lane == 0 && bi != 0xFFFFFFFFu) sm.dense[bi] = 1.0f;
          sync();
        }
        if (dense_seq_sum() > 0) chosen = pick_move();
        else { stage_root(tb, c0, k); chosen = gumbel_final_action(cp, tb, c0, k); }
      }
    } else {
      probs(temp, k);
      chosen = pick_move();
    }
    sync();

    if (ep.log_moves) {
      uint32_t row = 0;
      if (lane == 0) row = atomicAdd(&ar.ctl->log_rows, 1u);
      row = __shfl(row, 0, 64);
      if (row < ep.log_cap) {
        if (lane == 0) {
          uint32_t* r = ar.log_rows + static_cast<size_t>(row) * 8;
          r[0] = slot; r[1] = ar.slot_games[slot]; r[2] = chosen; r[3] = gs.turn; r[4] = cp; r[5] = capped ? 1u : 0u;
          r[6] = static_cast<uint32_t>(rng_before); r[7] = static_cast<uint32_t>(rng_before >> 32);
        }
        uint32_t* cr = ar.log_counts + static_cast<size_t>(row) * M;
        for (uint32_t m = lane; m < static_cast<uint32_t>(M); m += G) cr[m] = 0;
        sync();
        for (uint32_t i = lane; i < k; i += G) cr[sm.moves[i]] = sm.n[i];
      } else {
        raise(4u);
      }
    }
    if (ep.history && !capped) {
      if (ep.gumbel_hist) gumbel_improved_policy(cp, k, ar.V[tb + root]);   // play_manager.cc:411-417
      else if (ep.pruning && seat_eps(cp) > 0) probs_pruned(1.0f, root_n, k); else probs(1.0f, k);
      const uint32_t r = ph_rows;
      if (r < ep.max_hist_rows) {
        if constexpr (kSG) {   // pending rows hold the packed position (16 u64 words); the planes are written when the game's rows are committed
          GM::store_words(reinterpret_cast<uint64_t*>(ar.ph_canon) + (static_cast<size_t>(slot) * ep.max_hist_rows + r) * kSgPendWords, 1, 0, lane, gs);
        } else {
          float* crow = ar.ph_canon + (static_cast<size_t>(slot) * ep.max_hist_rows + r) * GM::CANON;
          for (uint32_t e = lane; e < static_cast<uint32_t>(GM::CANON); e += G) crow[e] = GM::canonical_at(gs, e);
        }
        float* prow = ar.ph_pi + (static_cast<size_t>(slot) * ep.max_hist_rows + r) * M;
        for (uint32_t m = lane; m < static_cast<uint32_t>(M); m += G) prow[m] = sm.dense[m];
        if (lane == 0) {
          uint32_t* pm = ar.ph_meta + (static_cast<size_t>(slot) * ep.max_hist_rows + r) * 2;
          pm[0] = gs.player; pm[1] = gs.turn;
        }
        ph_rows = r + 1;
      } else {
        raise(2u);
      }
      sync();
    }
    {
      const uint32_t dep = AZB_SEL(t_depth, cp);
      const float ald = dep == 0 ? 0.0f : static_cast<float>(AZB_SEL(t_tld, cp)) / static_cast<float>(dep);
      float ent = 0.0f;
      const float kf = static_cast<float>(k);
      if (!(kf <= 1 || root_n <= 1)) {
        const float log_k = az_logf(kf);
        const float total_n = static_cast<float>(root_n);
        for (uint32_t i = lane; i < k; i += G) {
          float t = 0.0f;
          if (sm.n[i] > 0) { const float p = static_cast<float>(sm.n[i]) / total_n; t = p * az_logf(p); }
          sm.f0[i] = t;
        }
        sync();
        float e = 0.0f;
        for (uint32_t i = 0; i < k; ++i) if (sm.n[i] > 0) e -= sm.f0[i];
        ent = e / log_k;
        sync();
      }
      if (lane == 0) {
        const uint32_t S = ep.S;
        if (!capped) { ar.g_dsum[0 * S + slot] += ald; ar.g_dsum[1 * S + slot] += ent; ar.g_cnt[1 * S + slot] += 1; }
        else { ar.g_dsum[2 * S + slot] += ald; ar.g_dsum[3 * S + slot] += ent; ar.g_cnt[2 * S + slot] += 1; }
        ar.g_dsum[4 * S + slot] += k;
        ar.g_cnt[0 * S + slot] += 1;
      }
    }
    for (uint32_t s = 0; s < static_cast<uint32_t>(P); ++s)
      if (!update_root(s, chosen)) return true;
    {
      bool base_valid = true;
      if (!step_state(gs, chosen, game_list(), glen, base_valid, 0)) { raise(64u); return true; }
      if constexpr (!kSG) {
        // persist the game's repetition list (whole list: a capture may have cleared it)
        uint64_t* gl = ar.rep_list + static_cast<size_t>(slot) * (GM::MAX_TURNS + 2);
        for (uint32_t i = lane; i < glen; i += G) gl[i] = sm.glist[i];
      }
    }
    uint32_t term = GM::terminal(gs);
    bool resigned = false;
    if (term == 0 && resign_entry >= 0) { term = static_cast<uint32_t>(resign_entry) + 1; resigned = true; }
    if (term != 0) { end_game(term, resigned); return true; }
    draw_capped();
    set_gumbel_target();
    if (!ep.tree_reuse) {
      for (uint32_t s = 0; s < static_cast<uint32_t>(P); ++s) reset_tree(s);
    } else {
      reapply_root_prior(gs.player, seat_eps(gs.player) > 0 && !(flags & kFlagCapped));
    }
    if (ep.half_nodes && lane == 0) {   // ask k_compact to move a tree whose active half is filling up
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const uint32_t b = t_bump[p];
        const uint32_t used = b - ((b - 1) / ep.half_nodes) * ep.half_nodes;
        if (used > ep.compact_above) ar.compact_flag[slot * P + p] = 1;
      }
    }
    return false;
  }

  __device__ __forceinline__ void end_game(uint32_t term, bool resigned) {
    sync();
    const uint32_t S = ep.S;
    const uint32_t rows = ep.history ? ph_rows : 0u;
    if (rows > 0) {
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(&ar.ctl->hist_rows, static_cast<unsigned long long>(rows));
      base = __shfl(base, 0, 64);
      // ring of hist_cap rows, row i of the run at i % hist_cap (see engine_kernels.h end_game)
      if (base + rows - ar.ctl->hist_read <= ep.hist_cap) {
        const uint32_t game_idx = ar.slot_games[slot];
        // the game's rows are contiguous on the pending side and contiguous modulo the ring on the other: flat copies,
        // eight loads in flight per lane, in two segments when the rows wrap around the end of the ring
        const size_t src0 = static_cast<size_t>(slot) * ep.max_hist_rows;
        const uint32_t first = static_cast<uint32_t>(base % ep.hist_cap);
        const uint32_t n1 = rows < ep.hist_cap - first ? rows : ep.hist_cap - first;
        auto flat_copy = [&](const float* sp, float* dp, size_t n) {
          for (size_t e0 = 0; e0 < n; e0 += 8 * G) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const size_t e = e0 + u * G + lane; if (e < n) t[u] = sp[e]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const size_t e = e0 + u * G + lane; if (e < n) dp[e] = t[u]; }
          }
        };
        for (uint32_t seg = 0; seg < 2; ++seg) {
          const uint32_t sr = seg == 0 ? 0u : n1, nr = seg == 0 ? n1 : rows - n1;
          if (nr == 0) continue;
          const size_t dst0 = seg == 0 ? first : 0u;
          if constexpr (kSG) {
            for (uint32_t r = 0; r < nr; ++r) {
              const typename GM::State ps = GM::load_words(reinterpret_cast<const uint64_t*>(ar.ph_canon) + (src0 + sr + r) * kSgPendWords, 1, 0, lane);
              GM::write_canonical(ps, ar.h_canon + (dst0 + r) * GM::CANON, lane, sm.rules);
            }
          } else {
            flat_copy(ar.ph_canon + (src0 + sr) * GM::CANON, ar.h_canon + dst0 * GM::CANON, static_cast<size_t>(nr) * GM::CANON);
          }
          flat_copy(ar.ph_pi + (src0 + sr) * M, ar.h_pi + dst0 * M, static_cast<size_t>(nr) * M);
          float* dv = ar.h_v + dst0 * (P + 1);
          if constexpr (GM::kRelative) {   // absolute_to_relative(scores, pending.player), play_manager.cc:451-454 (P == 2: a swap)
            for (uint32_t e = lane; e < nr * (P + 1); e += G) {
              const uint32_t rr = e / (P + 1), ent = e % (P + 1);
              const uint32_t pl = ar.ph_meta[(src0 + sr + rr) * 2];
              const uint32_t abs_ent = (ent < static_cast<uint32_t>(P) && pl == 1) ? 1u - ent : ent;
              dv[e] = (abs_ent == term - 1) ? 1.0f : 0.0f;
            }
          } else {
            for (uint32_t e = lane; e < nr * (P + 1); e += G) dv[e] = (e % (P + 1) == term - 1) ? 1.0f : 0.0f;
          }
          for (uint32_t r = lane; r < nr; r += G) {
            const uint32_t* pm = ar.ph_meta + (src0 + sr + r) * 2;
            uint32_t* hm = ar.h_meta + (dst0 + r) * 4;
            hm[0] = slot; hm[1] = game_idx; hm[2] = pm[1]; hm[3] = pm[0];
          }
        }
      } else {
        raise(2u);
      }
    }
    ph_rows = 0;
    if (lane == 0) {
      atomicAdd(&ar.a_scores[static_cast<size_t>(slot) * (P + 1) + (term - 1)], 1.0f);
      if (resigned) atomicAdd(&ar.a_resign[static_cast<size_t>(slot) * (P + 1) + (term - 1)], 1.0f);
      atomicAdd(&ar.a_perm_scores[(static_cast<size_t>(slot) * ep.num_perms + perm) * (P + 1) + (term - 1)], 1.0f);
      atomicAdd(&ar.a_perm_games[static_cast<size_t>(slot) * ep.num_perms + perm], 1u);
      // one round trip for the game's running totals, then accumulations that return nothing (each cell has this
      // one writer, so an atomic add is the same sum as load-add-store without the dependent load)
      double gd[5]; uint32_t gc[3];
#pragma unroll
      for (int j = 0; j < 5; ++j) gd[j] = ar.g_dsum[j * S + slot];
#pragma unroll
      for (int j = 0; j < 3; ++j) gc[j] = ar.g_cnt[j * S + slot];
      atomicAdd(reinterpret_cast<unsigned long long*>(&ar.a_len[slot]), static_cast<unsigned long long>(gs.turn));
      if constexpr (kSG) {   // per-variant tables, play_manager.cc:468-484
        const size_t sv = static_cast<size_t>(slot) * 4 + GM::variant(gs);
        atomicAdd(&ar.a_var_scores[(sv * ep.num_perms + perm) * (P + 1) + (term - 1)], 1.0f);
        atomicAdd(&ar.a_var_games[sv * ep.num_perms + perm], 1u);
        atomicAdd(reinterpret_cast<unsigned long long*>(&ar.a_var_len[sv]), static_cast<unsigned long long>(gs.turn));
#pragma unroll
        for (int j = 0; j < 5; ++j) atomicAdd(&ar.a_var_dsum[sv * 5 + j], gd[j]);
#pragma unroll
        for (int j = 0; j < 3; ++j) atomicAdd(reinterpret_cast<unsigned long long*>(&ar.a_var_cnt[sv * 3 + j]), static_cast<unsigned long long>(gc[j]));
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) { atomicAdd(&ar.a_dsum[j * S + slot], gd[j]); ar.g_dsum[j * S + slot] = 0.0; }
#pragma unroll
      for (int j = 0; j < 3; ++j) { atomicAdd(reinterpret_cast<unsigned long long*>(&ar.a_cnt[j * S + slot]), static_cast<unsigned long long>(gc[j])); ar.g_cnt[j * S + slot] = 0; }
      atomicAdd(&ar.slot_games[slot], 1u);
      const uint32_t pos = atomicAdd(&ar.ctl->ended_count, 1u);
      ar.ended_list[pos] = slot;
    }
  }

  __device__ __forceinline__ uint64_t emit_leaf(const typename GM::State& leaf) const {
    float* row = ar.canon + static_cast<size_t>(slot) * GM::CANON;
    uint64_t key;
    if constexpr (kSG) {
      GM::write_canonical(leaf, row, lane, sm.rules);
      key = GM::key(leaf, lane);
    } else {
      GM::write_canonical_wave(leaf, row, lane);
      key = GM::key(leaf);
    }
    if (lane == 0) ar.leaf_key[slot] = key;
    return key;
  }
  // position-cache probe (play_manager.cc:592-597) by the slot's wavefront: one lane per entry of the 64-entry shard;
  // on a hit the cached (pi[M], v) are copied into the slot's rows
  __device__ __forceinline__ bool cache_lookup(uint64_t key, uint32_t group) const {
    uint32_t sh;
    const CacheView cache = ep.num_groups == 1 ? ar.cache : ar.caches[group];
    const int cslot = wave_shard_find<64>(cache, key, lane, &sh);
    if (lane == 0) {
      unsigned long long* st = cache.stats + static_cast<size_t>(sh) * 4;
      if (cslot < 0) {
        atomicAdd(&st[1], 1ULL);
      } else {
        atomicAdd(&st[0], 1ULL);
        uint32_t* f = cache.freq + static_cast<size_t>(sh) * kWaveCap + cslot;
        if (atomicAdd(f, 1u) >= 3u) atomicSub(f, 1u);
      }
    }
    if (cslot < 0) return false;
    const float* sp = cache.policy + (static_cast<size_t>(sh) * cache.cap + cslot) * M;
    const float* sv = cache.value + (static_cast<size_t>(sh) * cache.cap + cslot) * (P + 1);
    wave_copy_row(ar.pi + static_cast<size_t>(slot) * M, sp, static_cast<uint32_t>(M), lane);
    if (lane <= static_cast<uint32_t>(P)) ar.v[static_cast<size_t>(slot) * (P + 1) + lane] = sv[lane];
    sync();
    return true;
  }
};

// One round for every slot of a wide-node game: one wavefront (= one workgroup) per slot.
// MODE 0: the whole round.  MODE 1 / 2 (round 5): the round SPLIT like the Connect4 engine's - 1 = every slot, but a
// slot whose next backup completes a search (the move behind it: temperature, resignation, the history row, re-rooting, the game
// step, the game's end) or whose game has to start is LISTED (ar.mover_list) and left untouched; 2 = the listed slots alone, the
// whole round for them, launched right behind (their new leaves join the same round's eval list).  The rare, register-hungry steps
// no longer size the kernel every slot runs: at two waves per SIMD the Tafl kernel held 256 registers + 71 spilled (288 B of scratch
// per lane, 72 % of its HBM writes); without make_move / start_game it fits.
template <class GM, bool kPlayout, int MODE = 0>
__device__ __forceinline__ void round_big_body(const EngineParams& ep, const EngineArrays& ar, BigScratch<GM>& sm) {
  uint32_t slot = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  if constexpr (MODE == 2) {
    if (slot >= ar.ctl->mover_count) return;
    slot = ar.mover_list[slot];
  }
  if (slot >= ep.S) return;
  if (ar.ctl->stop) return;
  const uint8_t st = ar.sstate[slot];
  if (st == kSlotDone || st == kSlotEnded) { if (ep.cache_on && lane == 0) ar.cache_keys[slot] = 0; return; }
  if constexpr (MODE == 1) {
    if (st != kSlotWaitEval) {      // kSlotFresh / kSlotRestart: a game start is the move step's
      if (lane == 0) { if (ep.cache_on) ar.cache_keys[slot] = 0; ar.mover_list[atomicAdd(&ar.ctl->mover_count, 1u)] = slot; }
      return;
    }
  }
  BigSlot<GM> c(ep, ar, sm, slot, lane);
#ifdef AZMI_BIG_PROF
  c.pf_t_ = wall_clock64();
#define AZB_CMARK(i) do { const unsigned long long now_ = wall_clock64(); c.pf_[i] += now_ - c.pf_t_; c.pf_t_ = now_; } while (0)
#else
#define AZB_CMARK(i) do {} while (0)
#endif
  c.load();
  AZB_CMARK(0);
  uint32_t inline_sims = 0, insert_key_set = 0;
  bool need_process = (st == kSlotWaitEval);
  if constexpr (MODE == 2) c.flags &= ~kFlagListed;
  if constexpr (MODE != 1) {
    if (!need_process) {
      c.start_game(); c.draw_capped(); c.set_gumbel_target();
      if (ep.gumbel_on && st == kSlotRestart && !ep.tree_reuse) c.set_gumbel_num_sims(c.gs.player, 0);   // see k_round
    }
  }
  for (;;) {
    if (need_process) {
      const uint32_t cp = c.gs.player;
      const bool noise = c.seat_eps(cp) > 0 && !(c.flags & kFlagCapped);
      const uint32_t goal = (c.flags & kFlagCapped) ? c.seat_cap_visits(cp) : c.seat_visits(cp);
      if constexpr (MODE == 1) {
        // process_result adds one to the seat's simulation count: this backup completes the search - the slot goes, as it stands
        // (answers of cache hits of this round already backed up), to the move step
        if (((cp == 0) ? c.t_depth[0] : c.t_depth[GM::P > 1 ? 1 : 0]) + 1u >= goal) {
          c.flags |= kFlagListed;
          if (lane == 0) { if (ep.cache_on) ar.cache_keys[slot] = 0; ar.mover_list[atomicAdd(&ar.ctl->mover_count, 1u)] = slot; }
          c.store(kSlotWaitEval);
          return;
        }
      }
      c.process_result(cp, (c.flags & kFlagLeafNeedsNet) != 0, noise);
      if constexpr (MODE != 1) {
        if (((cp == 0) ? c.t_depth[0] : c.t_depth[GM::P > 1 ? 1 : 0]) >= goal) {
          if (c.make_move(cp)) { if (ep.cache_on && lane == 0) ar.cache_keys[slot] = 0; c.store(kSlotEnded); return; }
        }
      }
    }
    const uint32_t cp = c.gs.player;
    typename GM::State leaf;
    uint32_t term = 0;
    AZB_CMARK(1);
    if (!c.find_leaf(cp, leaf, term)) { if (ep.cache_on && lane == 0) ar.cache_keys[slot] = 0; c.store(kSlotDone); return; }
    const bool playout = kPlayout && term == 0 && c.seat_eval_playout(cp);   // (a terminal leaf's evaluation is never used)
    const bool needs_net = term == 0 && !c.seat_eval_random(cp) && !playout;
    // kFlagLeafNeedsNet = "process_result reads the slot's v row": a net answer or the rollout's
    c.flags = (needs_net || playout) ? (c.flags | kFlagLeafNeedsNet) : (c.flags & ~kFlagLeafNeedsNet);
    if constexpr (kPlayout) { if (playout) c.playout_eval(leaf); }
    if (needs_net) {
      const uint32_t group = c.seat_group(cp);
      // the eval list's ticket is one returning atomic on a word every wave of the launch adds to: without a cache (the leaf goes to
      // the net for certain) it is drawn BEFORE the planes are written, so its round trip rides under their stores
      uint32_t ticket = 0;
      if (!ep.cache_on && lane == 0) ticket = atomicAdd(&ar.ctl->eval_count[group], 1u);
      AZB_CMARK(12 + 3);     // [15]: from the end of the expansion to here (terminal test of the leaf's flags, the ticket's issue)
      const uint64_t key = c.emit_leaf(leaf);
      AZB_CMARK(9);          // [9]: planes + key
      const bool hit = ep.cache_on && c.cache_lookup(key, group);
      if (!hit) {
        if (lane == 0) {
          atomicAdd(reinterpret_cast<unsigned long long*>(&ar.c_evals[slot]), 1ull);     // (this wave is the cell's one writer: the same sum as load-add-store without the dependent load)
          if (ep.cache_on) { ar.cache_keys[slot] = cache_key(key); ticket = atomicAdd(&ar.ctl->eval_count[group], 1u); }
          ar.leaf_group[slot] = static_cast<uint8_t>(group);
          ar.eval_list[static_cast<size_t>(group) * ep.S + ticket] = slot;
        }
        insert_key_set = 1;
        break;
      }
    }
    need_process = true;
    if (++inline_sims >= ep.max_inline) break;
  }
  if (ep.cache_on && !insert_key_set && lane == 0) ar.cache_keys[slot] = 0;
  AZB_CMARK(6);
  c.store(kSlotWaitEval);
  AZB_CMARK(7);
#ifdef AZMI_BIG_PROF
  c.pf_[8] = 1;
#pragma unroll
  for (int i = 0; i < 16; ++i) if (threadIdx.x == static_cast<uint32_t>(i)) atomicAdd(&g_big_prof[i], c.pf_[i]);
#endif
}

template <class GM, bool kPlayout = false>
__global__ __launch_bounds__(64) void k_round_big(EngineParams ep, EngineArrays ar) {
  __shared__ BigScratch<GM> sm;
  round_big_body<GM, kPlayout>(ep, ar, sm);
}
// The same round held to 256 registers, i.e. two waves per SIMD (the default build takes 256 VGPRs + 82 AGPRs = one wave per
// SIMD): the Tafl family's tree phase is wave-slot bound when the four shards' tree kernels meet (2048 waves on 1024 slots),
// and its 360 bytes of scratch spill cost less than the second pass (Tawlbwrdd tree side 165 -> 144 us, 266 -> 274 games/s);
// StarGambit spills 500 bytes and loses 5 %: it keeps the default.
template <class GM, bool kPlayout = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k_round_big_o2(EngineParams ep, EngineArrays ar) {
  __shared__ BigScratch<GM> sm;
  round_big_body<GM, kPlayout>(ep, ar, sm);
}
// The split round (round_big_body MODE 1 / 2): the lean kernel over every slot at two waves per SIMD, then the move step over the
// slots it listed (one wave per SIMD, the whole register file; a few slots per round: one move per seat_visits simulations)
template <class GM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k_round_big_sim(EngineParams ep, EngineArrays ar) {
  __shared__ BigScratch<GM> sm;
  round_big_body<GM, false, 1>(ep, ar, sm);
}
// (StarGambit, round 6: the same lean kernel with the whole register file - at two waves per SIMD its unit-per-lane rules spill)
template <class GM>
__global__ __launch_bounds__(64) void k_round_big_sim1(EngineParams ep, EngineArrays ar) {
  __shared__ BigScratch<GM> sm;
  round_big_body<GM, false, 1>(ep, ar, sm);
}
template <class GM>
__global__ __launch_bounds__(64) void k_round_big_move(EngineParams ep, EngineArrays ar) {
  __shared__ BigScratch<GM> sm;
  round_big_body<GM, false, 2>(ep, ar, sm);
}

// Arena compaction for wide games: copies the subtree under the current root into the idle half of the
// tree's arena in breadth-first order (children of a node stay contiguous and keep their order, so the
// search is unchanged), leaves a forwarding index in the old copy and re-points the slot's pending
// simulation (MCTS::current_ / path_).  One workgroup per flagged tree; HBM-bound: 28 B read + 28 B
// written per live node.
template <class GM>
__device__ __forceinline__ void compact_tree(const EngineParams& ep, const EngineArrays& ar, uint32_t* nif, const uint32_t t,
                                             uint32_t* s_off, uint32_t* s_oldc0, uint32_t* s_scan) {
  constexpr int P = GM::P;
  const uint32_t tid = threadIdx.x;
  const uint32_t slot = t / P, seat = t % P;
  const uint32_t H = ep.half_nodes;
  const size_t tb = static_cast<size_t>(t) * ep.cap;
  uint32_t* N = ar.N + tb; float* Q = ar.Q + tb; float* Pr = ar.Pr + tb; float* D = ar.D + tb; float* V = ar.V + tb;
  uint64_t* META = ar.META + tb;
  const uint32_t old_root = ar.root[t], old_bump = ar.bump[t];
  const uint32_t dst0 = (((old_bump - 1) / H) ^ 1u) * H;
  if (tid == 0) {
    N[dst0] = N[old_root]; Q[dst0] = Q[old_root]; Pr[dst0] = Pr[old_root]; D[dst0] = D[old_root]; V[dst0] = V[old_root];
    META[dst0] = META[old_root];
    META[old_root] = dst0;   // forwarding index
    if (nif) nif[tb + dst0] = nif[tb + old_root];
  }
  __syncthreads();
  uint32_t lo = dst0, hi = dst0 + 1, bump = dst0 + 1;
  bool overflow = false;
  while (lo < hi && !overflow) {
    for (uint32_t base = lo; base < hi && !overflow; base += 256) {
      const uint32_t i = base + tid;
      uint32_t k = 0, oc0 = 0;
      uint64_t m = 0;
      if (i < hi) { m = META[i]; k = meta_nch(m); oc0 = meta_ch0(m); }
      s_scan[tid] = k;
      __syncthreads();
      for (uint32_t d = 1; d < 256; d <<= 1) {   // inclusive Hillis-Steele scan
        const uint32_t v = tid >= d ? s_scan[tid - d] : 0u;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
      }
      const uint32_t total = s_scan[255], off = s_scan[tid] - k;
      if (bump + total > dst0 + H) { overflow = true; break; }   // uniform: total and bump are block-wide values
      if (k > 0) META[i] = (m & ~0xFFFFFFFFull) | static_cast<uint64_t>(bump + off);
      s_off[tid] = off; s_oldc0[tid] = oc0;
      __syncthreads();
      for (uint32_t e = tid; e < total; e += 256) {
        uint32_t a = 0, b = 256;                    // last j with s_off[j] <= e
        while (b - a > 1) { const uint32_t mid = (a + b) >> 1; if (s_off[mid] <= e) a = mid; else b = mid; }
        const uint32_t src = s_oldc0[a] + (e - s_off[a]), dst = bump + e;
        N[dst] = N[src]; Q[dst] = Q[src]; Pr[dst] = Pr[src]; D[dst] = D[src]; V[dst] = V[src];
        META[dst] = META[src];
        META[src] = dst;
        if (nif) nif[tb + dst] = nif[tb + src];
      }
      bump += total;
      __syncthreads();
    }
    lo = hi; hi = bump;
  }
  if (overflow) { if (tid == 0) { atomicOr(&ar.ctl->overflow, 1u); ar.ctl->stop = 1; } return; }
  if (tid == 0) {
    // the slot's pending simulation runs in the tree of the player to move
    if (ar.sstate[slot] == kSlotWaitEval && GM::player_from_words(ar.gs_words, ep.S, slot) == seat) {
      uint32_t* path = ar.path + static_cast<size_t>(slot) * ep.max_depth;
      const uint32_t plen = ar.plen[slot];
      for (uint32_t i = 0; i < plen; ++i) path[i] = static_cast<uint32_t>(META[path[i]]);
      ar.cur[slot] = static_cast<uint32_t>(META[ar.cur[slot]]);
    }
    ar.root[t] = dst0; ar.bump[t] = bump; ar.compact_flag[t] = 0;
  }
  __syncthreads();      // the next tree of this workgroup reuses the scratch
}
// A tree is flagged every few moves, i.e. a handful of the S x P trees per round: a grid of one workgroup per tree spent its
// time being dispatched (16 us in the mix for a kernel whose work is ~1 us).  A few workgroups each look through a stripe of
// the flags at once and compact the trees they find, one after the other.
constexpr uint32_t kCompactBlocks = 32;
template <class GM>
__global__ __launch_bounds__(256) void k_compact(EngineParams ep, EngineArrays ar, uint32_t trees, uint32_t* nif = nullptr) {
  __shared__ uint32_t s_off[256], s_oldc0[256], s_scan[256];
  __shared__ uint32_t s_list[256], s_n;
  if (ar.ctl->stop) return;
  const uint32_t tid = threadIdx.x;
  for (uint32_t base = blockIdx.x; base < trees; base += gridDim.x * 256) {
    if (tid == 0) s_n = 0;
    __syncthreads();
    const uint32_t t = base + tid * gridDim.x;
    if (t < trees && ar.compact_flag[t]) s_list[atomicAdd(&s_n, 1u)] = t;
    __syncthreads();
    const uint32_t n = s_n;
    for (uint32_t i = 0; i < n; ++i) {
      uint32_t tt = s_list[0];              // the smallest tree index not done yet: a fixed order whatever the atomics did
      for (uint32_t j = 0; j < n; ++j) { const uint32_t c = s_list[j]; tt = c < tt ? c : tt; }
      __syncthreads();
      if (tid == 0) for (uint32_t j = 0; j < n; ++j) if (s_list[j] == tt) s_list[j] = 0xFFFFFFFFu;
      __syncthreads();
      compact_tree<GM>(ep, ar, nif, tt, s_off, s_oldc0, s_scan);
    }
    __syncthreads();
  }
}

}  // namespace azmi
