// ORACLE — TEST INFRASTRUCTURE ONLY (see az_rng.hpp header for the usage rule).
// CPU restatement of PlayManager (play_manager.h:159-441, play_manager.cc) as a
// single-threaded, fully deterministic state machine.
//
// Determinism contract (SURVEY §8c tiers T1/T2, R12, R15):
//  * the reference keeps ONE `thread_local pcg32` per worker thread, shared by
//    every tree that thread touches (mcts.cc:19).  `per_slot_rng=false` restates
//    exactly that for one worker thread (one stream, FIFO queue order).
//    `per_slot_rng=true` gives every game slot its own stream seeded
//    slot_seed(seed, slot); with concurrent_games == 1 both modes coincide, which
//    is how a many-slot device run is compared slot by slot.
//  * the playout-cap / resign-playthrough coins come from an UNSEEDABLE
//    thread_local std::default_random_engine in the reference
//    (play_manager.cc:261-262).  Here they come from a second pcg32 per stream
//    ("coin stream", seeded seed ^ kCoinSalt) through uniform01 — build-defined.
//  * NN evaluation order: the mcts queue is drained, then every waiting slot is
//    evaluated in FIFO order in batches of max_batch_size (py_wrapper.cc:449-504
//    + play_manager.cc:619-642), then pushed back in batch order.
#pragma once
#include <chrono>
#include <deque>
#include <functional>
#include <memory>
#include <optional>
#include <vector>

#include "az_mcts.hpp"
#include "az_s3fifo.hpp"

namespace orc {

enum class EvalType : uint8_t { NN = 0, RANDOM = 1, PLAYOUT = 2 };  // play_manager.h:20

constexpr uint64_t kCoinSalt = 0x5851F42D4C957F2DULL;
constexpr uint64_t kRollSalt = 0x9E3779B97F4A7C15ULL;   // rollout stream of EvalType::PLAYOUT (game_state.cc:56-59 is unseedable)
inline uint64_t slot_seed(uint64_t seed, uint32_t slot) {
  return mix64(seed + 0x9E3779B97F4A7C15ULL * (static_cast<uint64_t>(slot) + 1));
}

struct PlayParams {  // play_manager.h:60-154 (fields the path implements)
  uint32_t games_to_play = 0;
  uint32_t concurrent_games = 0;
  uint32_t max_batch_size = 1;
  uint32_t max_cache_size = 0;
  uint32_t cache_shards = 1;
  std::vector<uint32_t> mcts_visits;
  float cpuct = 2.0f;
  float start_temp = 1.0f;
  float final_temp = 1.0f;
  float temp_decay_half_life = 0;
  std::vector<float> temp_decay_half_life_by_variant;   // play_manager.h:87-90
  bool history_enabled = false;
  bool self_play = false;
  bool tree_reuse = true;
  float epsilon = 0.0f;
  float mcts_root_temp = 1.0f;
  bool playout_cap_randomization = false;
  uint32_t playout_cap_depth = 25;
  float playout_cap_percent = 0.75f;
  float fpu_reduction = 0.0f;
  bool root_fpu_zero = false;
  bool shaped_dirichlet = false;
  bool policy_target_pruning = false;
  bool gumbel_enabled = false;             // play_manager.h:103-116
  uint32_t gumbel_m = 16;
  float gumbel_c_visit = 50.0f, gumbel_c_scale = 1.0f;
  bool gumbel_full = false;
  bool fast_search_uses_gumbel = false;
  float resign_percent = 0.0f;
  float resign_playthrough_percent = 0.0f;
  std::vector<EvalType> eval_type;  // per player; empty = all NN
  // model groups / seat permutations / per-seat overrides, play_manager.h:118-154
  std::vector<uint8_t> model_groups;                    // player -> model group (empty = identity)
  std::vector<std::vector<uint8_t>> seat_perms;         // [perm][seat] -> model group (empty = {model_groups})
  std::vector<std::vector<uint32_t>> seat_visits, seat_cap_visits;
  std::vector<std::vector<float>> seat_epsilon, seat_mcts_root_temp;
  std::vector<std::vector<uint8_t>> seat_root_fpu_zero;
  // per-seat Gumbel / resign overrides, play_manager.h:126-153
  std::vector<std::vector<uint8_t>> seat_gumbel_enabled, seat_gumbel_full, seat_gumbel_use_improved_policy;
  std::vector<std::vector<uint32_t>> seat_gumbel_m, seat_resign_consecutive;
  std::vector<std::vector<float>> seat_gumbel_c_visit, seat_gumbel_c_scale, seat_resign_threshold;
};

struct HistoryRow {  // game_state.h:16-20
  std::vector<float> canonical, v, pi;
};

struct MoveRecord {  // test hook: one entry per played move
  uint32_t slot, game_in_slot, move, turn;
  uint8_t player;
  bool capped;
  uint64_t rng_state;            // tree stream position right before pick_move's draw
  std::vector<uint32_t> counts;  // root counts() right before the move
};

class PlayManager {
 public:
  using Evaluator = std::function<void(const float* canonical, uint32_t n, float* v, float* pi)>;
  using GroupEvaluator = std::function<void(uint32_t group, const float* canonical, uint32_t n, float* v, float* pi)>;

  // perm_base: index of this manager's first game in a larger run (test hook: a one-slot manager standing in for
  // slot s of an S-slot run starts with perm (s % perms), play_manager.cc:218)
  PlayManager(std::unique_ptr<Game> base, PlayParams p, uint64_t seed, bool per_slot_rng, uint32_t perm_base = 0)
      : base_(std::move(base)), params_(std::move(p)), per_slot_rng_(per_slot_rng) {
    const uint32_t P = base_->num_players();
    if (params_.mcts_visits.size() != P)
      throw std::runtime_error("You must specify MCTS visits for each player");  // play_manager.cc:20-22
    normalise(P);
    games_started_ = params_.concurrent_games;
    const uint32_t nstreams = per_slot_rng_ ? params_.concurrent_games : 1;
    tree_rng_.resize(nstreams);
    coin_rng_.resize(nstreams);
    roll_rng_.resize(nstreams);
    for (uint32_t s = 0; s < nstreams; ++s) {
      const uint64_t sd = per_slot_rng_ ? slot_seed(seed, s) : seed;
      tree_rng_[s].seed(sd);
      coin_rng_[s].seed(sd ^ kCoinSalt);
      roll_rng_[s].seed(sd ^ kRollSalt);
    }
    if (params_.max_cache_size > 0) {  // play_manager.cc:195-203: one cache per model group
      const uint32_t per_group = params_.max_cache_size / num_model_groups_;
      for (uint32_t g = 0; g < num_model_groups_; ++g)
        caches_.push_back(std::make_unique<ShardedS3Fifo>(per_group, params_.cache_shards, per_group * 9 / 10,
                                                          base_->num_moves(), P + 1));
    }
    perm_scores_.assign(seat_perms_.size(), std::vector<float>(P + 1, 0.0f));
    perm_games_.assign(seat_perms_.size(), 0u);
    for (int v = 0; v < base_->num_variants(); ++v) {   // play_manager.cc:237-255
      VariantStats vs;
      vs.scores.assign(P + 1, 0.0f);
      vs.perm_scores.assign(seat_perms_.size(), std::vector<float>(P + 1, 0.0f));
      vs.perm_games.assign(seat_perms_.size(), 0u);
      variants_.push_back(std::move(vs));
    }
    awaiting_inference_.resize(num_model_groups_);
    scores_.assign(P + 1, 0.0f);
    resign_scores_.assign(P + 1, 0.0f);
    slots_.resize(params_.concurrent_games);
    for (uint32_t i = 0; i < params_.concurrent_games; ++i) {  // play_manager.cc:214-230
      Slot& g = slots_[i];
      g.gs = base_->copy();
      g.gs->randomize_start_from(coin_rng(i));
      g.perm_index = static_cast<uint8_t>((perm_base + i) % seat_perms_.size());   // play_manager.cc:218
      for (uint32_t j = 0; j < P; ++j) g.mcts.push_back(make_mcts(i, g.perm_index, j));
      g.canonical.assign(base_->canonical_size(), 0.0f);
      g.v.assign(P + 1, 0.0f);
      g.pi.assign(base_->num_moves(), 0.0f);
      awaiting_mcts_.push_back(i);
    }
  }

  // Single-threaded equivalent of N x play() + the GameRunner eval pipeline.
  void run(const Evaluator& nn) {
    run_groups([&nn](uint32_t, const float* c, uint32_t n, float* v, float* pi) { nn(c, n, v, pi); });
  }
  // bench hook (cpu_baseline leg): stop the run loop after this many seconds of wall time (0 = run to games_to_play)
  double time_limit_s = 0.0;
  void run_groups(const GroupEvaluator& nn) {
    const auto t_start = std::chrono::steady_clock::now();
    auto expired = [&]() {
      return time_limit_s > 0.0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() >= time_limit_s;
    };
    while (games_completed_ < params_.games_to_play) {
      uint32_t since_check = 0;
      while (!awaiting_mcts_.empty() && games_completed_ < params_.games_to_play) {
        const uint32_t i = awaiting_mcts_.front();
        awaiting_mcts_.pop_front();
        step(i);
        if (time_limit_s > 0.0 && (++since_check & 1023u) == 0 && expired()) return;
      }
      if (games_completed_ >= params_.games_to_play) break;
      if (expired()) return;
      bool any = false;
      for (auto& q : awaiting_inference_) any = any || !q.empty();
      if (!any) break;  // nothing left to do (all slots retired)
      for (uint32_t g = 0; g < num_model_groups_; ++g) flush_inference(g, nn);
    }
  }
  uint32_t num_model_groups() const { return num_model_groups_; }
  uint32_t num_seat_perms() const { return static_cast<uint32_t>(seat_perms_.size()); }
  const std::vector<float>& perm_scores(uint32_t i) const { return perm_scores_[i]; }
  uint32_t perm_games_completed(uint32_t i) const { return perm_games_[i]; }
  // per-variant tables (play_manager.h:218-275, 425-440); the getters' divisions are restated in variant_stats()
  struct VariantStats {
    std::vector<float> scores;
    std::vector<std::vector<float>> perm_scores;
    std::vector<uint32_t> perm_games;
    uint32_t games = 0;
    uint64_t game_length = 0, total_move_count = 0, full_move_count = 0, fast_move_count = 0;
    double total_avg_leaf_depth = 0, total_search_entropy = 0, fast_total_avg_leaf_depth = 0, fast_total_search_entropy = 0,
           total_valid_moves = 0;
  };
  uint32_t num_tracked_variants() const { return static_cast<uint32_t>(variants_.size()); }
  const VariantStats& variant(uint32_t v) const { return variants_[v]; }
  // out[0..6] = variant_avg_game_length, _avg_leaf_depth, _avg_search_entropy, _fast_avg_leaf_depth, _fast_avg_search_entropy,
  //             _avg_moves_per_turn, _avg_valid_moves
  void variant_stats(uint32_t v, float* out) const {
    const VariantStats& m = variants_[v];
    out[0] = m.games == 0 ? 0 : static_cast<float>(m.game_length) / static_cast<float>(m.games);
    out[1] = m.full_move_count == 0 ? 0 : static_cast<float>(m.total_avg_leaf_depth / static_cast<double>(m.full_move_count));
    out[2] = m.full_move_count == 0 ? 0 : static_cast<float>(m.total_search_entropy / static_cast<double>(m.full_move_count));
    out[3] = m.fast_move_count == 0 ? 0 : static_cast<float>(m.fast_total_avg_leaf_depth / static_cast<double>(m.fast_move_count));
    out[4] = m.fast_move_count == 0 ? 0 : static_cast<float>(m.fast_total_search_entropy / static_cast<double>(m.fast_move_count));
    out[5] = m.game_length == 0 ? 0 : static_cast<float>(m.total_move_count) / static_cast<float>(m.game_length);
    out[6] = m.total_move_count == 0 ? 0 : static_cast<float>(m.total_valid_moves / static_cast<double>(m.total_move_count));
  }

  // ---- stats getters, play_manager.h:173-366 -------------------------------------
  const std::vector<float>& scores() const { return scores_; }
  const std::vector<float>& resign_scores() const { return resign_scores_; }
  uint32_t games_completed() const { return games_completed_; }
  float avg_game_length() const {
    return static_cast<float>(game_length_) / static_cast<float>(games_completed_);
  }
  float avg_leaf_depth() const {
    if (full_move_count_ == 0) return 0;
    return static_cast<float>(total_avg_leaf_depth_ / static_cast<double>(full_move_count_));
  }
  float avg_search_entropy() const {
    if (full_move_count_ == 0) return 0;
    return static_cast<float>(total_search_entropy_ / static_cast<double>(full_move_count_));
  }
  float fast_avg_leaf_depth() const {
    if (fast_move_count_ == 0) return 0;
    return static_cast<float>(fast_total_avg_leaf_depth_ / static_cast<double>(fast_move_count_));
  }
  float fast_avg_search_entropy() const {
    if (fast_move_count_ == 0) return 0;
    return static_cast<float>(fast_total_search_entropy_ / static_cast<double>(fast_move_count_));
  }
  float avg_moves_per_turn() const {
    if (game_length_ == 0) return 0;
    return static_cast<float>(total_move_count_) / static_cast<float>(game_length_);
  }
  float avg_valid_moves() const {
    if (total_move_count_ == 0) return 0;
    return static_cast<float>(total_valid_moves_ / static_cast<double>(total_move_count_));
  }
  size_t hist_count() const { return history_.size(); }
  const std::vector<HistoryRow>& history() const { return history_; }
  const std::vector<MoveRecord>& moves() const { return moves_; }
  uint64_t sims() const { return sims_; }
  uint64_t nn_evals() const { return nn_evals_; }
  uint64_t cache_hits() const { uint64_t n = 0; for (auto& c : caches_) n += c->hits(); return n; }
  uint64_t cache_misses() const { uint64_t n = 0; for (auto& c : caches_) n += c->misses(); return n; }
  bool record_moves = false;
  bool trace_on = false;
  std::vector<std::pair<uint64_t, uint64_t>> trace;

 private:
  struct Pending {  // play_manager.h:28-31
    HistoryRow ph;
    uint8_t player;
  };
  struct Slot {  // GameData, play_manager.h:33-58
    std::unique_ptr<Game> gs;
    uint64_t leaf_hash = 0;
    std::vector<Mcts> mcts;
    std::vector<float> canonical, v, pi;
    std::vector<Pending> partial_history;
    bool initialized = false, capped = false, playthrough = false;
    uint8_t perm_index = 0;   // GameData::perm_index; seat_perm = seat_perms_[perm_index]
    double total_avg_leaf_depth = 0, total_search_entropy = 0;
    double fast_total_avg_leaf_depth = 0, fast_total_search_entropy = 0;
    double total_valid_moves = 0;
    uint32_t move_count = 0, full_move_count = 0, fast_move_count = 0;
    uint32_t games_played = 0;
    std::vector<uint32_t> resign_streak;   // GameData::resign_streak, play_manager.h:54-57 (not cleared between games)
  };

  Pcg32& tree_rng(uint32_t slot) { return tree_rng_[per_slot_rng_ ? slot : 0]; }
  Pcg32& coin_rng(uint32_t slot) { return coin_rng_[per_slot_rng_ ? slot : 0]; }

  // ctor steps 1-5, play_manager.cc:24-113
  void normalise(uint32_t P) {
    std::vector<uint8_t> groups = params_.model_groups;
    if (groups.empty()) for (uint32_t i = 0; i < P; ++i) groups.push_back(static_cast<uint8_t>(i));
    num_model_groups_ = 0;
    for (uint8_t g : groups) num_model_groups_ = std::max<uint32_t>(num_model_groups_, g + 1u);
    std::vector<uint32_t> visits_g(num_model_groups_, 0);
    for (uint32_t i = 0; i < P; ++i) visits_g[groups[i]] = params_.mcts_visits[i];
    if (!params_.eval_type.empty()) {
      eval_types_.assign(num_model_groups_, EvalType::NN);
      for (uint32_t i = 0; i < P; ++i) eval_types_[groups[i]] = params_.eval_type[i];
    }
    seat_perms_ = params_.seat_perms;
    if (seat_perms_.empty()) seat_perms_.push_back(groups);
    const size_t np = seat_perms_.size();
    auto check = [&](size_t outer, const char* name) {
      if (outer != np) throw std::runtime_error(std::string(name) + " outer dimension must match number of seat permutations");
    };
    auto fill = [&](auto& dst, const auto& given, auto deflt, const char* name) {
      if (given.empty()) { dst.resize(np); for (size_t q = 0; q < np; ++q) { dst[q].clear(); for (uint32_t sidx = 0; sidx < P; ++sidx) dst[q].push_back(deflt(q, sidx)); } }
      else {
        check(given.size(), name);
        for (auto& row : given) if (row.size() != P) throw std::runtime_error(std::string(name) + " inner dimension must match number of players");
        dst = given;
      }
    };
    fill(seat_visits_, params_.seat_visits, [&](size_t q, uint32_t sidx) { return visits_g[seat_perms_[q][sidx]]; }, "seat_visits");
    fill(seat_cap_visits_, params_.seat_cap_visits, [&](size_t, uint32_t) { return params_.playout_cap_depth; }, "seat_cap_visits");
    fill(seat_epsilon_, params_.seat_epsilon, [&](size_t, uint32_t) { return params_.epsilon; }, "seat_epsilon");
    fill(seat_root_temp_, params_.seat_mcts_root_temp, [&](size_t, uint32_t) { return params_.mcts_root_temp; }, "seat_mcts_root_temp");
    fill(seat_fpu_zero_, params_.seat_root_fpu_zero, [&](size_t, uint32_t) { return static_cast<uint8_t>(params_.root_fpu_zero ? 1 : 0); }, "seat_root_fpu_zero");
    // play_manager.cc:116-176
    fill(seat_gumbel_enabled_, params_.seat_gumbel_enabled, [&](size_t, uint32_t) { return static_cast<uint8_t>(params_.gumbel_enabled ? 1 : 0); }, "seat_gumbel_enabled");
    fill(seat_gumbel_m_, params_.seat_gumbel_m, [&](size_t, uint32_t) { return params_.gumbel_m; }, "seat_gumbel_m");
    fill(seat_gumbel_c_visit_, params_.seat_gumbel_c_visit, [&](size_t, uint32_t) { return params_.gumbel_c_visit; }, "seat_gumbel_c_visit");
    fill(seat_gumbel_c_scale_, params_.seat_gumbel_c_scale, [&](size_t, uint32_t) { return params_.gumbel_c_scale; }, "seat_gumbel_c_scale");
    fill(seat_gumbel_full_, params_.seat_gumbel_full, [&](size_t, uint32_t) { return static_cast<uint8_t>(params_.gumbel_full ? 1 : 0); }, "seat_gumbel_full");
    fill(seat_gumbel_g3_, params_.seat_gumbel_use_improved_policy, [&](size_t, uint32_t) { return static_cast<uint8_t>(0); }, "seat_gumbel_use_improved_policy");
    fill(seat_resign_threshold_, params_.seat_resign_threshold, [&](size_t, uint32_t) { return -2.0f; }, "seat_resign_threshold");
    fill(seat_resign_consecutive_, params_.seat_resign_consecutive, [&](size_t, uint32_t) { return 1u; }, "seat_resign_consecutive");
  }

  Mcts make_mcts(uint32_t slot, uint8_t perm, uint32_t player) {  // play_manager.cc:602-617
    MctsConfig c;
    c.cpuct = params_.cpuct;
    c.num_players = base_->num_players();
    c.num_moves = base_->num_moves();
    c.epsilon = seat_epsilon_[perm][player];
    c.root_policy_temp = seat_root_temp_[perm][player];
    c.fpu_reduction = params_.fpu_reduction;
    c.relative_values = base_->relative_values();
    c.root_fpu_zero = seat_fpu_zero_[perm][player] != 0;
    c.shaped_dirichlet = params_.shaped_dirichlet;
    c.gumbel_enabled = seat_gumbel_enabled_[perm][player] != 0;   // play_manager.cc:612-616
    c.gumbel_m = seat_gumbel_m_[perm][player];
    c.gumbel_c_visit = seat_gumbel_c_visit_[perm][player];
    c.gumbel_c_scale = seat_gumbel_c_scale_[perm][player];
    c.gumbel_full = seat_gumbel_full_[perm][player] != 0;
    Mcts m(c, &tree_rng(slot));
    if (trace_on) m.trace = &trace;
    return m;
  }

  // capped searches fall back to PUCT (target 0) unless fast_search_uses_gumbel
  template <class SlotT>
  void set_gumbel_target(SlotT& game) {
    const uint8_t cp = game.gs->current_player();
    const uint32_t target = game.capped ? (params_.fast_search_uses_gumbel ? seat_cap_visits_[game.perm_index][cp] : 0u)
                                        : seat_visits_[game.perm_index][cp];
    game.mcts[cp].set_gumbel_num_sims(target);
  }

  EvalType eval_type_for_group(uint32_t group) const {
    return eval_types_.empty() ? EvalType::NN : eval_types_[group];
  }

  // One iteration of the worker loop body, play_manager.cc:277-599.
  void step(uint32_t i) {
    Slot& game = slots_[i];
    const uint32_t P = base_->num_players();
    if (game.initialized) {
      const uint8_t cp = game.gs->current_player();
      Mcts& mcts = game.mcts[cp];
      mcts.process_result(game.v.data(), game.pi.data(), seat_epsilon_[game.perm_index][cp] > 0 && !game.capped);
      ++sims_;
      const uint32_t goal_depth = game.capped ? seat_cap_visits_[game.perm_index][cp] : seat_visits_[game.perm_index][cp];
      if (mcts.depth() >= goal_depth) {
        float temp = params_.start_temp;
        float half_life = params_.temp_decay_half_life;
        if (!params_.temp_decay_half_life_by_variant.empty()) {   // play_manager.cc:290-296
          const int vid = game.gs->get_variant_id();
          if (vid >= 0 && vid < static_cast<int>(params_.temp_decay_half_life_by_variant.size()))
            half_life = params_.temp_decay_half_life_by_variant[vid];
        }
        if (half_life != 0) {  // play_manager.cc:297-304, ln2 hard-coded as 0.693
          const uint32_t t = game.gs->current_turn();
          constexpr float ln2 = 0.693;
          const float lambda = ln2 / half_life;
          temp -= params_.final_temp;
          temp *= az_expf(-lambda * t);
          temp += params_.final_temp;
        }
        std::optional<std::vector<float>> resign_score;
        if (params_.resign_percent > 0 && !game.playthrough) {  // play_manager.cc:306-334
          if (P != 2) throw std::runtime_error("Resigning only works in 2 player games");
          float pred[3];
          mcts.root_value(pred);
          const float w = pred[0], l = pred[1], d = pred[2];
          const double resign_val = 1.0 - params_.resign_percent;
          std::vector<float> tmp(P + 1, 0.0f);
          if (w > resign_val) tmp[cp] = 1.0;
          else if (l > resign_val) tmp[(cp + 1) % 2] = 1.0;
          else if (d > resign_val) tmp[P] = 1.0;
          float tsum = 0;
          for (float x : tmp) tsum += x;
          if (tsum > 0) {
            if (uniform01(coin_rng(i)) < params_.resign_playthrough_percent) game.playthrough = true;
            else resign_score = tmp;
          }
        }
        // per-seat opt-in resign, play_manager.cc:335-366: W - L of the seat at or below its threshold for
        // `consecutive` own moves in a row
        if (!resign_score.has_value() && !game.playthrough) {
          const float seat_thresh = seat_resign_threshold_[game.perm_index][cp];
          if (seat_thresh > -2.0f) {
            if (P != 2) throw std::runtime_error("Per-seat resign only works in 2 player games");
            if (game.resign_streak.empty()) game.resign_streak.assign(P, 0u);
            float pred[3];
            mcts.root_value(pred);
            const float v_self = pred[0] - pred[1];
            if (v_self <= seat_thresh) ++game.resign_streak[cp];
            else game.resign_streak[cp] = 0;
            const uint32_t need = std::max(1u, seat_resign_consecutive_[game.perm_index][cp]);
            if (game.resign_streak[cp] >= need) {
              std::vector<float> tmp(P + 1, 0.0f);
              tmp[(cp + 1) % 2] = 1.0;
              resign_score = tmp;
            }
          }
        }
        // move choice, play_manager.cc:367-406: G1 Gumbel acting, the opt-in G3 acting
        // (seat_gumbel_use_improved_policy) or PUCT sampling
        const uint64_t rng_before = tree_rng(i).state;
        uint32_t chosen_m;
        if (mcts.gumbel_enabled() && !game.capped) {
          if (!seat_gumbel_g3_[game.perm_index][cp]) {
            chosen_m = mcts.gumbel_final_action();
          } else {
            std::vector<float> pi_g = mcts.gumbel_improved_policy();
            if (temp != 1.0f && temp > 0.0f) {
              float sg = 0.0f;
              for (float& x : pi_g) { x = az_powf(x, 1.0f / temp); }
              for (float x : pi_g) sg += x;
              if (sg > 0) for (float& x : pi_g) x /= sg;
            } else if (temp <= 0.0f) {
              uint32_t arg = 0;
              for (uint32_t m = 1; m < pi_g.size(); ++m) if (pi_g[m] > pi_g[arg]) arg = m;
              std::fill(pi_g.begin(), pi_g.end(), 0.0f);
              pi_g[arg] = 1.0f;
            }
            float sg = 0.0f;
            for (float x : pi_g) sg += x;
            if (sg > 0) chosen_m = Mcts::pick_move(pi_g, tree_rng(i));
            else chosen_m = mcts.gumbel_final_action();
          }
        } else {
          const std::vector<float> pi_play = mcts.probs(temp);
          chosen_m = Mcts::pick_move(pi_play, tree_rng(i));
        }
        if (trace_on) trace.push_back({5 | (static_cast<uint64_t>(chosen_m) << 8), tree_rng(i).state});
        if (record_moves) {
          MoveRecord mr;
          mr.slot = i; mr.game_in_slot = game.games_played; mr.move = chosen_m;
          mr.turn = game.gs->current_turn(); mr.player = cp; mr.capped = game.capped;
          mr.rng_state = rng_before;
          mr.counts = mcts.counts();
          moves_.push_back(std::move(mr));
        }
        if (params_.history_enabled && !game.capped) {  // play_manager.cc:407-424
          Pending pd;
          pd.ph.canonical.assign(base_->canonical_size(), 0.0f);
          game.gs->canonicalized(pd.ph.canonical.data());
          pd.ph.v.assign(game.v.size(), 0.0f);
          pd.ph.pi = params_.gumbel_enabled ? mcts.gumbel_improved_policy()   // play_manager.cc:411-417
                     : (params_.policy_target_pruning && seat_epsilon_[game.perm_index][cp] > 0) ? mcts.probs_pruned(1.0)
                                                                              : mcts.probs(1.0);
          pd.player = game.gs->current_player();
          game.partial_history.push_back(std::move(pd));
        }
        if (!game.capped) {  // play_manager.cc:425-435
          game.total_avg_leaf_depth += mcts.avg_leaf_depth();
          game.total_search_entropy += mcts.normalized_root_entropy();
          ++game.full_move_count;
        } else {
          game.fast_total_avg_leaf_depth += mcts.avg_leaf_depth();
          game.fast_total_search_entropy += mcts.normalized_root_entropy();
          ++game.fast_move_count;
        }
        game.total_valid_moves += mcts.num_root_children();
        ++game.move_count;
        for (auto& m : game.mcts) { m.update_root(*game.gs, chosen_m); if (trace_on) trace.push_back({6, tree_rng(i).state}); }  // player order
        game.gs->play_move(chosen_m);
        float sc[kMaxValue];
        bool over = game.gs->scores(sc);
        std::vector<float> scores;
        if (over) scores.assign(sc, sc + P + 1);
        if (!over && resign_score.has_value()) {
          scores = *resign_score;
          over = true;
        } else {
          resign_score.reset();
        }
        if (over) {
          if (params_.history_enabled) {  // play_manager.cc:448-461
            while (!game.partial_history.empty()) {
              Pending& pending = game.partial_history.back();
              if (base_->relative_values()) {
                std::vector<float> rel(P + 1);
                for (uint32_t k = 0; k < P; ++k) rel[k] = scores[(pending.player + k) % P];
                rel[P] = scores[P];
                pending.ph.v = rel;
              } else {
                pending.ph.v = scores;
              }
              history_.push_back(std::move(pending.ph));
              game.partial_history.pop_back();
            }
          }
          // play_manager.cc:463-514
          for (uint32_t k = 0; k <= P; ++k) scores_[k] += scores[k];
          for (uint32_t k = 0; k <= P; ++k) perm_scores_[game.perm_index][k] += scores[k];   // play_manager.cc:466-467
          ++perm_games_[game.perm_index];
          {   // per-variant tables, play_manager.cc:468-484
            const int vid = game.gs->get_variant_id();
            if (vid >= 0 && vid < static_cast<int>(variants_.size())) {
              VariantStats& vs = variants_[vid];
              for (uint32_t k = 0; k <= P; ++k) { vs.scores[k] += scores[k]; vs.perm_scores[game.perm_index][k] += scores[k]; }
              ++vs.games; ++vs.perm_games[game.perm_index];
              vs.game_length += game.gs->current_turn();
              vs.total_avg_leaf_depth += game.total_avg_leaf_depth; vs.total_search_entropy += game.total_search_entropy;
              vs.fast_total_avg_leaf_depth += game.fast_total_avg_leaf_depth; vs.fast_total_search_entropy += game.fast_total_search_entropy;
              vs.total_valid_moves += game.total_valid_moves;
              vs.total_move_count += game.move_count; vs.full_move_count += game.full_move_count; vs.fast_move_count += game.fast_move_count;
            }
          }
          if (resign_score.has_value())
            for (uint32_t k = 0; k <= P; ++k) resign_scores_[k] += (*resign_score)[k];
          ++games_completed_;
          game_length_ += game.gs->current_turn();
          total_avg_leaf_depth_ += game.total_avg_leaf_depth;
          total_search_entropy_ += game.total_search_entropy;
          fast_total_avg_leaf_depth_ += game.fast_total_avg_leaf_depth;
          fast_total_search_entropy_ += game.fast_total_search_entropy;
          total_valid_moves_ += game.total_valid_moves;
          total_move_count_ += game.move_count;
          full_move_count_ += game.full_move_count;
          fast_move_count_ += game.fast_move_count;
          game.total_avg_leaf_depth = game.total_search_entropy = 0;
          game.fast_total_avg_leaf_depth = game.fast_total_search_entropy = 0;
          game.total_valid_moves = 0;
          game.move_count = game.full_move_count = game.fast_move_count = 0;
          ++game.games_played;
          if (games_started_ >= params_.games_to_play) return;  // slot retires
          game.perm_index = static_cast<uint8_t>(games_started_ % seat_perms_.size());   // play_manager.cc:511
          ++games_started_;
          game.gs = base_->copy();
          game.gs->randomize_start_from(coin_rng(i));
          for (uint32_t j = 0; j < P; ++j) game.mcts[j] = make_mcts(i, game.perm_index, j);
        }
        // play_manager.cc:522-555
        game.capped = params_.playout_cap_randomization &&
                      (uniform01(coin_rng(i)) < params_.playout_cap_percent);
        set_gumbel_target(game);  // play_manager.cc:525-539
        if (!params_.tree_reuse) {
          for (uint32_t j = 0; j < P; ++j) game.mcts[j] = make_mcts(i, game.perm_index, j);
        } else {
          const uint8_t next_cp = game.gs->current_player();
          Mcts& next = game.mcts[next_cp];
          if (next.root_n() > 0) {
            next.apply_root_policy_temp();
            if (seat_epsilon_[game.perm_index][next_cp] > 0 && !game.capped) { if (trace_on) trace.push_back({3 | (static_cast<uint64_t>(next.num_root_children()) << 8), tree_rng(i).state}); next.add_root_noise(); if (trace_on) trace.push_back({4, tree_rng(i).state}); }
          }
        }
      }
    } else {  // play_manager.cc:557-571
      game.initialized = true;
      game.capped = params_.playout_cap_randomization &&
                    (uniform01(coin_rng(i)) < params_.playout_cap_percent);
      set_gumbel_target(game);  // play_manager.cc:561-570
    }
    // play_manager.cc:572-598
    const uint8_t cp = game.gs->current_player();
    Mcts& mcts = game.mcts[cp];
    auto leaf = mcts.find_leaf(*game.gs);
    const uint32_t group = seat_perms_[game.perm_index][cp];   // play_manager.cc:577
    const EvalType et = eval_type_for_group(group);
    if (et != EvalType::NN) {
      if (et == EvalType::PLAYOUT) playout_eval(*leaf, roll_rng_[per_slot_rng_ ? i : 0], game.v.data(), game.pi.data());
      else dumb_eval(*leaf, game.v.data(), game.pi.data());
      awaiting_mcts_.push_back(i);
      return;
    }
    leaf->canonicalized(game.canonical.data());
    game.leaf_hash = leaf->key();
    if (!caches_.empty() && caches_[group]->find(game.leaf_hash, game.pi.data(), game.v.data())) {
      awaiting_mcts_.push_back(i);
      return;
    }
    awaiting_inference_[group].push_back(i);
  }

  void flush_inference(uint32_t group, const GroupEvaluator& nn) {
    std::deque<uint32_t>& queue = awaiting_inference_[group];
    const uint32_t P = base_->num_players();
    const uint32_t M = base_->num_moves();
    const size_t csz = base_->canonical_size();
    while (!queue.empty()) {
      const uint32_t n = static_cast<uint32_t>(
          std::min<size_t>(queue.size(), std::max<uint32_t>(1, params_.max_batch_size)));
      std::vector<uint32_t> idx(queue.begin(), queue.begin() + n);
      queue.erase(queue.begin(), queue.begin() + n);
      std::vector<float> batch(n * csz), v(n * (P + 1)), pi(static_cast<size_t>(n) * M);
      for (uint32_t r = 0; r < n; ++r)
        std::copy(slots_[idx[r]].canonical.begin(), slots_[idx[r]].canonical.end(), batch.begin() + r * csz);
      nn(group, batch.data(), n, v.data(), pi.data());
      nn_evals_ += n;
      for (uint32_t r = 0; r < n; ++r) {  // update_inferences, play_manager.cc:619-642
        Slot& g = slots_[idx[r]];
        std::copy(v.begin() + r * (P + 1), v.begin() + (r + 1) * (P + 1), g.v.begin());
        std::copy(pi.begin() + static_cast<size_t>(r) * M, pi.begin() + static_cast<size_t>(r + 1) * M, g.pi.begin());
        if (!caches_.empty()) caches_[group]->insert(g.leaf_hash, g.pi.data(), g.v.data());
      }
      for (uint32_t r = 0; r < n; ++r) awaiting_mcts_.push_back(idx[r]);
    }
  }

  std::unique_ptr<Game> base_;
  PlayParams params_;
  bool per_slot_rng_;
  std::vector<Pcg32> tree_rng_, coin_rng_, roll_rng_;
  std::vector<VariantStats> variants_;
  std::vector<Slot> slots_;
  std::deque<uint32_t> awaiting_mcts_;
  std::vector<std::deque<uint32_t>> awaiting_inference_;   // one queue per model group
  std::vector<std::unique_ptr<ShardedS3Fifo>> caches_;
  uint32_t num_model_groups_ = 1;
  std::vector<EvalType> eval_types_;                       // per model group
  std::vector<std::vector<uint8_t>> seat_perms_, seat_fpu_zero_;
  std::vector<std::vector<uint8_t>> seat_gumbel_enabled_, seat_gumbel_full_, seat_gumbel_g3_;
  std::vector<std::vector<uint32_t>> seat_gumbel_m_, seat_resign_consecutive_;
  std::vector<std::vector<float>> seat_gumbel_c_visit_, seat_gumbel_c_scale_, seat_resign_threshold_;
  std::vector<std::vector<uint32_t>> seat_visits_, seat_cap_visits_;
  std::vector<std::vector<float>> seat_epsilon_, seat_root_temp_;
  std::vector<std::vector<float>> perm_scores_;
  std::vector<uint32_t> perm_games_;
  std::vector<HistoryRow> history_;
  std::vector<MoveRecord> moves_;
  std::vector<float> scores_, resign_scores_;
  uint32_t games_started_ = 0, games_completed_ = 0;
  uint64_t game_length_ = 0;
  double total_avg_leaf_depth_ = 0, total_search_entropy_ = 0;
  double fast_total_avg_leaf_depth_ = 0, fast_total_search_entropy_ = 0;
  double total_valid_moves_ = 0;
  uint64_t total_move_count_ = 0, full_move_count_ = 0, fast_move_count_ = 0;
  uint64_t sims_ = 0, nn_evals_ = 0;
};

}  // namespace orc
