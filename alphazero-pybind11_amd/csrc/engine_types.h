// Plain-data views shared by the host side (engine.hip) and the kernels
// (engine_kernels.h).  Everything the engine owns lives in HBM as
// structure-of-arrays; these structs only carry the base pointers.
#pragma once
#include <stdint.h>

#include "dev_cache.h"

namespace azmi {

// slot life cycle
enum SlotState : uint8_t {
  kSlotFresh = 0,    // no game started yet (PlayManager ctor, play_manager.cc:214-230)
  kSlotWaitEval = 1, // a leaf is pending: next round starts with process_result
  kSlotEnded = 2,    // game finished this round; k_assign decides restart / retire
  kSlotRestart = 3,  // start the next game at the beginning of the round
  kSlotDone = 4,     // retired (games_started >= games_to_play, play_manager.cc:507-509)
  kSlotQueued = 5    // split rounds: the move step left a leaf whose planes are written but which is not on an eval list yet;
                     // the next round's k_sim lists it (and then waits for the answer like kSlotWaitEval)
};

enum SlotFlags : uint8_t { kFlagCapped = 1, kFlagPlaythrough = 2, kFlagLeafNeedsNet = 4,
                           kFlagPendRec = 8 /* ar.pend[slot] describes the pending simulation (split rounds) */,
                           kFlagReqOut = 16 /* pipeline: the pending leaf is a request in the net ring / an answer in the result
                                               granules tagged ar.req_seq[slot]; cleared when the answer is consumed or settled */,
                           kFlagListed = 32 /* pipeline: a tree wavefront handed the slot to the move step (MOVE ring); cleared by the
                                               move step - what is still set when an epoch ends goes to the boundary's move step */ };

// node META word: [31:0] first child (tree-relative), [43:32] child count,
// [55:44] move, [56] player to move at the node, [59:57] terminal code
// (0 = not terminal, 1 + index of the winning entry of the one-hot score vector)
__host__ __device__ inline uint64_t meta_pack(uint32_t ch0, uint32_t nch, uint32_t mv, uint32_t player, uint32_t term) {
  return static_cast<uint64_t>(ch0) | (static_cast<uint64_t>(nch) << 32) | (static_cast<uint64_t>(mv) << 44) |
         (static_cast<uint64_t>(player) << 56) | (static_cast<uint64_t>(term) << 57);
}
__host__ __device__ inline uint32_t meta_ch0(uint64_t m) { return static_cast<uint32_t>(m); }
__host__ __device__ inline uint32_t meta_nch(uint64_t m) { return static_cast<uint32_t>(m >> 32) & 0xFFFu; }
__host__ __device__ inline uint32_t meta_mv(uint64_t m) { return static_cast<uint32_t>(m >> 44) & 0xFFFu; }
__host__ __device__ inline uint32_t meta_player(uint64_t m) { return static_cast<uint32_t>(m >> 56) & 1u; }
__host__ __device__ inline uint32_t meta_term(uint64_t m) { return static_cast<uint32_t>(m >> 57) & 7u; }

struct EngineParams {  // run constants (PlayParams, play_manager.h:60-154, after ctor normalisation)
  uint32_t S;              // concurrent_games
  uint32_t cap;            // nodes per tree arena
  uint32_t games_to_play;
  uint32_t visits[4];      // per seat (single seat permutation)
  uint32_t cap_visits;     // playout_cap_depth
  float cpuct, start_temp, final_temp, half_life;
  float epsilon, root_temp, fpu_reduction;
  float cap_percent, resign_percent, resign_playthrough;
  uint32_t history, tree_reuse, cap_rand, root_fpu_zero, shaped, pruning;
  uint32_t eval_random[4]; // seat uses EvalType::RANDOM (dumb_eval) instead of the net
  uint32_t max_inline;
  uint32_t sim_budget;     // split rounds: k_sim starts no further simulation of a slot once this many 100 MHz ticks of the round have passed (0 = no budget)
  uint32_t hist_cap, log_cap, log_moves;
  uint32_t max_hist_rows;  // pending history rows per slot (= max moves of one game)
  uint32_t max_depth;      // path capacity per slot
  uint32_t cache_on;       // device S3-FIFO position cache enabled (max_cache_size > 0)
  uint32_t trace_slot;     // debug: slot whose RNG events are traced (0xFFFFFFFF = off)
  uint32_t trace_cap;
  uint32_t trace_after;    // debug: events are recorded from this round on (AZMI_TRACE_AFTER)
  // Gumbel AlphaZero (play_manager.h:103-116)
  uint32_t gumbel_on;     // some seat searches with Gumbel (the per-seat settings are in seat_tab)
  uint32_t gumbel_hist;   // PlayParams::gumbel_enabled: history rows carry the improved policy (play_manager.cc:411-417)
  uint32_t fast_gumbel;
  uint32_t seat_resign;   // some seat has a resign threshold (seat_resign_threshold > -2)
  uint32_t gum_stride;     // floats per tree in gum_g (>= max children of a root)
  // wide games: the arena is two halves of `half_nodes`; the live subtree is copied into the idle half
  // (k_compact) once the active half holds more than `compact_above` nodes after a move. 0 = one flat arena.
  uint32_t half_nodes, compact_above;
  uint32_t num_perms, num_groups;   // seat permutations / model groups (play_manager.cc:24-113)
  // games with variants (StarGambitUnifiedGS): the base game's pinned variant / variant weights (star_gambit_gs.h:828-830) and
  // temp_decay_half_life_by_variant (play_manager.h:87-90; n_half_life_v entries)
  int32_t sg_pinned;
  float sg_probs[4];
  uint32_t n_half_life_v;
  float half_life_v[4];
};

// per (permutation, seat) record of ar.seat_tab, 8 words: visits | cap_visits + flags | epsilon | root temp |
// Gumbel word (seat_gum_pack) | gumbel_c_visit | gumbel_c_scale | seat_resign_threshold
constexpr uint32_t kSeatWords = 8;
// bit 0 seat_gumbel_enabled, 1 seat_gumbel_full, 2 seat_gumbel_use_improved_policy; bits 8-23 seat_gumbel_m;
// bits 24-31 max(1, seat_resign_consecutive)
__host__ __device__ inline uint32_t seat_gum_pack(uint32_t enabled, uint32_t full, uint32_t g3, uint32_t m, uint32_t resign_need) {
  return (enabled & 1u) | ((full & 1u) << 1) | ((g3 & 1u) << 2) | ((m & 0xFFFFu) << 8) | ((resign_need & 0xFFu) << 24);
}
__host__ __device__ inline uint32_t seat_w1_pack(uint32_t cap_visits, uint32_t fpu_zero, uint32_t eval_random, uint32_t group,
                                                 uint32_t eval_playout = 0) {
  return (cap_visits & 0xFFFFFFu) | (fpu_zero << 24) | (eval_random << 25) | (group << 26) | (eval_playout << 28);
}

constexpr uint32_t kSgPendWords = 16;   // StarGambit pending history row: the packed position (11 state words), 128 B apart
constexpr uint32_t kPendingWords = 3;   // Connect4 pending history row: stones of player 0, of player 1, turn | player << 32

// One tree node of the lane-group engine (Connect4) — struct Node, mcts.h:14-48, as one 32-byte record: the descent reads a
// node's children as consecutive records (two 16-byte loads per child lane, 7 x 32 B contiguous) instead of five separate
// arrays, which is also one base pointer in scalar registers instead of six.  The wide-game engine keeps the SoA arrays
// below (its wavefront scans ONE field of up to ~250 children at a time).
struct alignas(32) NodeRec {
  uint32_t n;     // Node::n
  float q, pr, d; // Node::q, policy, d
  float v;        // Node::v
  uint32_t pad;
  uint64_t meta;  // children range, move, player, terminal code (meta_pack)
};

// Split rounds: the pending simulation's path as the descent saw it, one record per level (level i <-> lane i of the slot's
// 8-lane group, paths of at most 8 levels): the node chosen at level i with its n, q, d, v, the player to move at its parent,
// and - the same in every lane - the evaluated leaf's META after its expansion; lane j also carries the move of the leaf's
// child j.  k_sim backs the simulation up from this record and the (v, pi) rows alone: one memory round trip.
struct alignas(32) PendRec {
  uint32_t node, n;
  float q, d, v;
  uint32_t pp_mv;       // bits 0-7 player to move at the parent, bits 8-19 move of the leaf's child `lane`
  uint64_t leaf_meta;
};

struct Control {  // small device control block, copied back by azmi_pm_poll
  uint32_t games_started;
  uint32_t games_completed;
  uint32_t stop;
  uint32_t ended_count;
  uint32_t pad_hist;      // (was the 32-bit row counter)
  uint32_t log_rows;
  uint32_t overflow;      // bit0 tree arena, bit1 history, bit2 move log, bit3 path
  uint32_t live_slots;
  uint64_t rounds;
  uint32_t eval_count[4]; // per model group: entries of eval_list[g] written by this round's k_round
  uint32_t pad_read;
  uint32_t mover_count;   // split rounds: entries of mover_list written by this round's k_sim
  // finished-sample ring: rows finished / rows the host has consumed, free-running 64-bit counters (row i of the run lives at
  // i % hist_cap; 32-bit counters would break that sequence when they wrap, after ~14 h at the headline's sample rate)
  unsigned long long hist_rows;
  unsigned long long hist_read;
};

constexpr uint32_t kGumMaxM = 64;   // cap on PlayParams.gumbel_m (reference default 16)

struct EngineArrays {
  Control* ctl;
  uint32_t* ended_list;   // [S]
  PendRec* pend;          // [S][8] split rounds: register image of the pending simulation's path (NULL for the wide games)
  uint32_t* mover_list;   // [S] split rounds: slots k_sim left to the move step (k_round over this list), unordered
  // ---- per slot -----------------------------------------------------------------
  uint64_t* gs_words;     // [STATE_WORDS][S] game state, SoA
  uint64_t* rng;          // [S] tree stream   (mcts.cc:19)
  uint64_t* coin;         // [S] coin stream   (play_manager.cc:261-262)
  uint8_t* sstate;        // [S]
  uint8_t* flags;         // [S]
  uint32_t* cur;          // [S] leaf node of the pending simulation (MCTS::current_)
  uint32_t* plen;         // [S] MCTS::path_.size()
  uint32_t* path;         // [S][max_depth]
  uint32_t* slot_games;   // [S] games completed by the slot
  uint64_t* rep_list;     // [S][max_turns + 2] repetition keys since the last capture (Tafl family)
  uint32_t* rep_len;      // [S]
  // running totals of the slot's current game (GameData, play_manager.h:46-53)
  double* g_dsum;         // [5][S]: leaf depth, entropy, fast leaf depth, fast entropy, valid moves
  uint32_t* g_cnt;        // [3][S]: move_count, full_move_count, fast_move_count
  // committed totals over the slot's finished games (play_manager.cc:463-497)
  float* a_scores;        // [S][P+1]
  float* a_resign;        // [S][P+1]
  uint64_t* a_len;        // [S] sum of game lengths
  double* a_dsum;         // [5][S]
  uint64_t* a_cnt;        // [3][S]
  uint64_t* c_sims;       // [S] simulations finished
  uint64_t* c_evals;      // [S] leaves sent to the net
  // pending history rows of the running game (GameData::partial_history)
  uint32_t* ph_count;     // [S]
  float* ph_canon;        // [S][max_hist_rows][CANON] pending planes (wide games) / [S][max_hist_rows][kPendingWords] u64 packed positions (Connect4)
  float* ph_pi;           // [S][max_hist_rows][M]
  uint32_t* ph_meta;      // [S][max_hist_rows][2]: player, turn
  // ---- per tree (slot * P + seat) -------------------------------------------------
  uint32_t* root;         // MCTS::root_
  uint32_t* bump;         // next free node of the arena
  uint32_t* depth;        // MCTS::depth_
  uint64_t* tld;          // MCTS::total_leaf_depth_
  // ---- node records of the lane-group engine, [trees * cap] (NULL for the wide games) ----------------
  NodeRec* nodes;
  // ---- node arrays of the wide-game engine, [trees * cap] (NULL for Connect4) -------------------------
  uint32_t* N;            // Node::n
  float* Q;               // Node::q
  float* Pr;              // Node::policy
  float* D;               // Node::d
  float* V;               // Node::v
  uint64_t* META;         // children range, move, player, terminal code
  // ---- slot-indexed evaluation I/O ---------------------------------------------------
  float* canon;           // [S][CANON]   leaf planes for the net
  float* v;               // [S][P+1]     net value  (probabilities)
  float* pi;              // [S][M]       net policy (probabilities)
  uint64_t* leaf_key;     // [S]          position key of the pending leaf
  uint64_t* leaf_pos;     // [3][S]       pipeline (Connect4): the pending leaf's packed position (stones p0, stones p1, player); NULL = not kept
  uint32_t* req_seq;      // [S]          pipeline: sequence number of the slot's latest net request (tags its result granules)
  // ---- finished samples (PlayHistory rows) ---------------------------------------------
  float* h_canon;         // [hist_cap][CANON]
  float* h_v;             // [hist_cap][P+1]
  float* h_pi;            // [hist_cap][M]
  uint32_t* h_meta;       // [hist_cap][4]: slot, game_in_slot, turn, player
  // ---- move log (parity hook) -------------------------------------------------------------
  uint32_t* log_rows;     // [log_cap][8]
  uint32_t* log_counts;   // [log_cap][M]
  uint64_t* cache_keys;   // [S] scratch: keys to insert this round (0 = none)
  CacheView cache;        // position cache (s3fifo_cache.h), see dev_cache.h
  uint64_t* trace;        // debug [trace_cap][2]: tag, rng state; trace[0] = event count
  // ---- Gumbel AlphaZero per-tree search state (mcts.h:179-191) ---------------------------------
  uint32_t* gum_state;    // [trees][8]: target, initialized, n_survivors, phase_idx, sims_in_phase, m_eff, remaining
  float* gum_g;           // [trees][gum_stride] Gumbel(0,1) sample per root child
  uint16_t* gum_surv;     // [trees][kGumMaxM] surviving root-child indices, best first
  uint32_t* compact_flag; // [trees] set by make_move, consumed by k_compact
  uint32_t* eval_list;    // [groups][S] slots whose pending leaf needs group g's net this round (unordered)
  uint32_t* seat_tab;     // [perms][P][kSeatWords] per-seat search settings after the reference's normalisation
  uint32_t* perm;         // [S] GameData::perm_index
  uint8_t* leaf_group;    // [S] model group of the pending leaf
  float* a_perm_scores;   // [S][perms][P+1] committed scores per permutation
  uint32_t* a_perm_games; // [S][perms]
  uint64_t* roll;           // [S] rollout stream of EvalType::PLAYOUT seats (game_state.cc:56-59)
  uint32_t* resign_streak;  // [S][P] GameData::resign_streak (kept from one game of the slot to the next, like the reference)
  const CacheView* caches;  // [groups] one S3-FIFO per model group (play_manager.cc:195-203); `cache` = caches[0]
  // ---- StarGambit (games with variants and multi-action turns) ------------------------------------------------------
  uint64_t* rep_path;       // [S][max_turns + 2] position history of the running descent (path-local tail of rep_list)
  // committed per-variant totals of the slot's finished games (play_manager.cc:468-484), NULL for games without variants
  float* a_var_scores;      // [S][4][perms][P+1]  (variant_perm_scores; variant_scores = sum over perms)
  uint32_t* a_var_games;    // [S][4][perms]
  uint64_t* a_var_len;      // [S][4]
  double* a_var_dsum;       // [S][4][5]
  uint64_t* a_var_cnt;      // [S][4][3]
};

}  // namespace azmi
