// ORACLE — TEST INFRASTRUCTURE ONLY (see az_rng.hpp header for the usage rule).
// CPU restatement of the reference's per-(game, player) search tree:
// struct Node (mcts.h:14-48) and class MCTS (mcts.h:50-201, mcts.cc).
// Sequential, pointer-free pool of nodes; every floating-point expression keeps
// the reference's operand order and type promotions (build with
// -ffp-contract=off).  Summation order for dense [num_moves] vectors is
// left-to-right (SURVEY §8 note ①: Eigen's packet reductions are not available
// in this image, so that one order is "parity unpinned").
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "az_games.hpp"
#include "az_rng.hpp"

namespace orc {

struct Node {  // mcts.h:14-48
  float q = 0, d = 0, v = 0, policy = 0;
  uint32_t move = 0, n = 0, n_in_flight = 0;
  int8_t player = 0;
  bool terminal = false;         // scores != nullptr
  float scores[kMaxValue] = {};  // *scores
  uint32_t child0 = 0, nchild = 0;  // children are pool[child0 .. child0+nchild)
};

constexpr float kNoiseAlphaRatio = 10.83f;   // mcts.cc:14
constexpr float kGumbelLogFloor = 1e-20f;    // mcts.cc:18

struct MctsConfig {  // ctor arguments, mcts.h:52-57
  float cpuct = 2.0f;
  uint32_t num_players = 2, num_moves = 0;
  float epsilon = 0, root_policy_temp = 1.0f, fpu_reduction = 0;
  bool relative_values = false, root_fpu_zero = false, shaped_dirichlet = false;
  bool gumbel_enabled = false;
  uint32_t gumbel_m = 16;
  float gumbel_c_visit = 50.0f, gumbel_c_scale = 1.0f;
  bool gumbel_full = false;
};

class Mcts {
 public:
  Mcts(const MctsConfig& c, Pcg32* rng) : cfg_(c), re_(rng) { pool_.emplace_back(); root_ = 0; }

  // ---- Node::add_children, mcts.cc:93-101 ---------------------------------
  void add_children(uint32_t node, const uint8_t* valids, uint32_t num_moves) {
    std::vector<uint32_t> moves;
    for (uint32_t w = 0; w < num_moves; ++w)
      if (valids[w] == 1) moves.push_back(w);
    shuffle(moves.data(), moves.size(), *re_);
    const uint32_t c0 = static_cast<uint32_t>(pool_.size());
    for (uint32_t m : moves) {
      Node c;
      c.move = m;
      pool_.push_back(c);
    }
    pool_[node].child0 = c0;
    pool_[node].nchild = static_cast<uint32_t>(moves.size());
  }

  // ---- Node::set_policy_normalized, mcts.cc:109-121 -----------------------
  void set_policy_normalized(uint32_t node, const float* pi, bool apply_temp, float inv_temp) {
    float sum = 0.0f;
    Node& nd = pool_[node];
    for (uint32_t i = 0; i < nd.nchild; ++i) {
      Node& c = pool_[nd.child0 + i];
      float p = pi[c.move];
      if (apply_temp) p = az_powf(p, inv_temp);
      c.policy = p;
      sum += p;
    }
    for (uint32_t i = 0; i < nd.nchild; ++i) pool_[nd.child0 + i].policy /= sum;
  }

  // ---- Node::uct, mcts.cc:123-128 ------------------------------------------
  static float uct(const Node& c, float sqrt_parent_n, float cpuct, float fpu_value) {
    return (c.n == 0 ? fpu_value : c.q) +
           cpuct * c.policy * sqrt_parent_n / static_cast<float>(c.n + c.n_in_flight + 1);
  }

  // ---- Node::best_child, mcts.cc:130-149 -----------------------------------
  uint32_t best_child(uint32_t node, float cpuct, float fpu_reduction) const {
    const Node& nd = pool_[node];
    float seen_policy = 0.0f;
    for (uint32_t i = 0; i < nd.nchild; ++i) {
      const Node& c = pool_[nd.child0 + i];
      if (c.n > 0) seen_policy += c.policy;
    }
    const float fpu_value = nd.v - fpu_reduction * std::sqrt(seen_policy);
    const float sqrt_n = std::sqrt(static_cast<float>(nd.n + nd.n_in_flight));
    if (nd.nchild == 0) throw std::out_of_range("best_child on a node without children");
    uint32_t best_i = 0;
    float best_uct = uct(pool_[nd.child0], sqrt_n, cpuct, fpu_value);
    for (uint32_t i = 1; i < nd.nchild; ++i) {
      const float u = uct(pool_[nd.child0 + i], sqrt_n, cpuct, fpu_value);
      if (u > best_uct) {
        best_uct = u;
        best_i = i;
      }
    }
    return nd.child0 + best_i;
  }

  // ---- MCTS::update_root, mcts.cc:151-173 ----------------------------------
  void update_root(const Game& gs, uint32_t move) {
    depth_ = 0;
    total_leaf_depth_ = 0;
    if (pool_[root_].nchild == 0) {
      std::vector<uint8_t> valids(gs.num_moves());
      gs.valid_moves(valids.data());
      add_children(root_, valids.data(), gs.num_moves());
    }
    const Node& r = pool_[root_];
    uint32_t found = UINT32_MAX;
    for (uint32_t i = 0; i < r.nchild; ++i)
      if (pool_[r.child0 + i].move == move) { found = r.child0 + i; break; }
    if (found == UINT32_MAX)
      throw std::runtime_error("ahh, what is this move: " + std::to_string(move));
    root_ = found;  // the reference moves the subtree into root_ and frees the rest
    reset_gumbel_state();  // mcts.cc:172
  }

  // ---- MCTS::find_leaf, mcts.cc:462-498 ------------------------------------
  std::unique_ptr<Game> find_leaf(const Game& gs) {
    current_ = root_;
    auto leaf = gs.copy();
    if (cfg_.gumbel_enabled && !gumbel_initialized_ && gumbel_num_sims_target_ > 0 && pool_[root_].n > 0 &&
        pool_[root_].nchild != 0) {
      init_gumbel_state();  // lazy init, mcts.cc:465-472
    }
    while (pool_[current_].n > 0 && !pool_[current_].terminal) {
      path_.push_back(current_);
      if (cfg_.gumbel_enabled && gumbel_initialized_ && current_ == root_) {
        current_ = pool_[root_].child0 + static_cast<uint32_t>(gumbel_next_root_child());
      } else if (cfg_.gumbel_enabled && gumbel_initialized_ && cfg_.gumbel_full) {
        current_ = pool_[current_].child0 + static_cast<uint32_t>(gumbel_interior_select(current_));
      } else {
        const float fpu = (current_ == root_ && cfg_.root_fpu_zero) ? 0.0f : cfg_.fpu_reduction;
        current_ = best_child(current_, cfg_.cpuct, fpu);
      }
      leaf->play_move(pool_[current_].move);
    }
    total_leaf_depth_ += path_.size();
    if (pool_[current_].n == 0) {
      pool_[current_].player = static_cast<int8_t>(leaf->current_player());
      float s[kMaxValue];
      if (leaf->scores(s)) {
        pool_[current_].terminal = true;
        for (uint32_t i = 0; i <= cfg_.num_players; ++i) pool_[current_].scores[i] = s[i];
      }
      std::vector<uint8_t> valids(leaf->num_moves());
      leaf->valid_moves(valids.data());
      add_children(current_, valids.data(), leaf->num_moves());
    }
    return leaf;
  }

  // ---- MCTS::process_result, mcts.cc:500-555 --------------------------------
  // `value` is [num_players+1] and is overwritten like the reference's reference
  // parameter (terminal: *scores; relative_values: rotated).
  void process_result(float* value, const float* pi, bool root_noise_enabled) {
    const uint32_t P = cfg_.num_players;
    if (pool_[current_].terminal) {
      for (uint32_t i = 0; i <= P; ++i) value[i] = pool_[current_].scores[i];
    } else {
      if (current_ == root_) {
        set_policy_normalized(current_, pi, cfg_.root_policy_temp != 1.0f,
                              1.0f / cfg_.root_policy_temp);
        if (root_noise_enabled && !cfg_.gumbel_enabled) { tr(1); add_root_noise(); tr(2); }
      } else {
        set_policy_normalized(current_, pi, false, 1.0f);
      }
      if (cfg_.relative_values) {  // game_state.h:36-46
        const uint8_t pl = static_cast<uint8_t>(pool_[current_].player);
        if (pl != 0) {
          float out[kMaxValue];
          for (uint32_t i = 0; i < P; ++i) out[(pl + i) % P] = value[i];
          out[P] = value[P];
          for (uint32_t i = 0; i <= P; ++i) value[i] = out[i];
        }
      }
    }
    const int32_t np = static_cast<int32_t>(P);
    while (!path_.empty()) {
      const uint32_t parent = path_.back();
      path_.pop_back();
      Node& cur = pool_[current_];
      float v = value[pool_[parent].player];
      v += value[P] / np;
      cur.q = (cur.q * static_cast<float>(cur.n) + v) / static_cast<float>(cur.n + 1);
      cur.d = (cur.d * static_cast<float>(cur.n) + value[P]) / static_cast<float>(cur.n + 1);
      if (cur.n == 0) {
        const float leaf_v = value[cur.player] + value[P] / np;
        cur.v = leaf_v;
      }
      ++cur.n;
      current_ = parent;
    }
    Node& r = pool_[root_];
    if (r.n == 0) {
      r.v = value[r.player] + value[P] / np;
      r.d = value[P];
    }
    ++depth_;
    ++r.n;
  }

  // ---- WU-UCT batched search, mcts.cc:752-851 -------------------------------
  // find_leaf_batched: descend through nodes that are visited OR have an evaluation in flight, add the
  // in-flight penalty after each selection, expand only a node nobody expanded yet.
  std::unique_ptr<Game> find_leaf_batched(const Game& gs) {
    InFlight ifl;
    uint32_t cur = root_;
    auto leaf = gs.copy();
    while ((pool_[cur].n > 0 || pool_[cur].n_in_flight > 0) && pool_[cur].nchild != 0 && !pool_[cur].terminal) {
      ifl.path.push_back(cur);
      const float fpu = (cur == root_ && cfg_.root_fpu_zero) ? 0.0f : cfg_.fpu_reduction;
      const uint32_t selected = best_child(cur, cfg_.cpuct, fpu);
      ++pool_[cur].n_in_flight;
      cur = selected;
      leaf->play_move(pool_[cur].move);
    }
    ++pool_[cur].n_in_flight;
    total_leaf_depth_ += ifl.path.size();
    if (pool_[cur].n == 0 && pool_[cur].nchild == 0) {
      pool_[cur].player = static_cast<int8_t>(leaf->current_player());
      float s[kMaxValue];
      if (leaf->scores(s)) {
        pool_[cur].terminal = true;
        for (uint32_t i = 0; i <= cfg_.num_players; ++i) pool_[cur].scores[i] = s[i];
      }
      std::vector<uint8_t> valids(leaf->num_moves());
      leaf->valid_moves(valids.data());
      add_children(cur, valids.data(), leaf->num_moves());
    }
    ifl.leaf = cur;
    in_flight_.push_back(std::move(ifl));
    return leaf;
  }
  // process_result_batched, mcts.cc:786-845: the same backup as process_result over the stored path,
  // releasing the in-flight marks on the way
  void process_result_batched(uint32_t leaf_index, float* value, const float* pi, bool root_noise_enabled) {
    const InFlight& ifl = in_flight_.at(leaf_index);
    path_ = ifl.path;
    current_ = ifl.leaf;
    --pool_[current_].n_in_flight;
    for (uint32_t node : path_) --pool_[node].n_in_flight;
    process_result(value, pi, root_noise_enabled);
  }
  uint32_t in_flight_count() const { return static_cast<uint32_t>(in_flight_.size()); }
  void reset_batch() { in_flight_.clear(); }

  // ---- MCTS::add_root_noise, mcts.cc:403-446 --------------------------------
  void add_root_noise() {
    Node& r = pool_[root_];
    const size_t k = r.nchild;
    std::vector<float> noise(cfg_.num_moves, 0.0f);
    double sum = 0.0;
    if (cfg_.shaped_dirichlet && k > 1) {
      const float N = static_cast<float>(k);
      float log_sum = 0.0f;
      for (uint32_t i = 0; i < k; ++i)
        log_sum += az_logf(std::min(pool_[r.child0 + i].policy, 0.01f) + 1e-20f);
      const float log_mean = log_sum / N;
      float shaped_sum = 0.0f;
      for (uint32_t i = 0; i < k; ++i) {
        const float lp = az_logf(std::min(pool_[r.child0 + i].policy, 0.01f) + 1e-20f);
        shaped_sum += std::max(0.0f, lp - log_mean);
      }
      const float uniform = 1.0f / N;
      for (uint32_t i = 0; i < k; ++i) {
        Node& c = pool_[r.child0 + i];
        const float lp = az_logf(std::min(c.policy, 0.01f) + 1e-20f);
        const float shaped = std::max(0.0f, lp - log_mean);
        float alpha_prop = (shaped_sum > 0) ? 0.5f * (shaped / shaped_sum + uniform) : uniform;
        alpha_prop = std::max(alpha_prop, 1e-6f);
        Gamma dist(kNoiseAlphaRatio * alpha_prop, 1.0f);  // fresh object per child
        noise[c.move] = dist(*re_);
        sum += noise[c.move];
      }
    } else {
      Gamma dist(kNoiseAlphaRatio / static_cast<float>(k), 1.0f);  // one object, cache carries
      for (uint32_t i = 0; i < k; ++i) {
        Node& c = pool_[r.child0 + i];
        noise[c.move] = dist(*re_);
        sum += noise[c.move];
      }
    }
    for (uint32_t i = 0; i < k; ++i) {
      Node& c = pool_[r.child0 + i];
      c.policy = c.policy * (1 - cfg_.epsilon) + cfg_.epsilon * noise[c.move] / static_cast<float>(sum);
    }
  }

  // ---- MCTS::apply_root_policy_temp, mcts.cc:448-460 ------------------------
  void apply_root_policy_temp() {
    if (cfg_.root_policy_temp == 1.0f) return;
    Node& r = pool_[root_];
    float sum = 0.0f;
    for (uint32_t i = 0; i < r.nchild; ++i) {
      Node& c = pool_[r.child0 + i];
      c.policy = az_powf(c.policy, 1.0f / cfg_.root_policy_temp);
      sum += c.policy;
    }
    if (sum > 0.0f)
      for (uint32_t i = 0; i < r.nchild; ++i) pool_[r.child0 + i].policy /= sum;
  }

  // ---- MCTS::root_value, mcts.h:78-100 ---------------------------------------
  void root_value(float* wld) const {
    const Node& r = pool_[root_];
    float q = 0, d = 0;
    bool found = false;
    for (uint32_t i = 0; i < r.nchild; ++i) {
      const Node& c = pool_[r.child0 + i];
      if (c.n > 0 && c.q > q) { q = c.q; d = c.d; found = true; }
    }
    if (!found && r.n > 0) { q = r.v; d = r.d; }
    const float w = q - d / static_cast<int32_t>(cfg_.num_players);
    const double l = 1.0 - w - d;
    wld[0] = w;
    wld[1] = static_cast<float>(l);
    wld[2] = d;
  }

  // ---- counts / root_q_values, mcts.cc:557-573 -------------------------------
  std::vector<uint32_t> counts() const {
    std::vector<uint32_t> out(cfg_.num_moves, 0);
    const Node& r = pool_[root_];
    for (uint32_t i = 0; i < r.nchild; ++i) out[pool_[r.child0 + i].move] = pool_[r.child0 + i].n;
    return out;
  }
  std::vector<float> root_q_values() const {
    std::vector<float> out(cfg_.num_moves, 0.0f);
    const Node& r = pool_[root_];
    for (uint32_t i = 0; i < r.nchild; ++i) out[pool_[r.child0 + i].move] = pool_[r.child0 + i].q;
    return out;
  }

  static float seq_sum(const std::vector<float>& x) {
    float s = 0.0f;
    for (float e : x) s += e;
    return s;
  }
  // x.array().pow(e); pow(x, 1) == x exactly, so e == 1 is skipped (the device does the same).
  static void pow_inplace(std::vector<float>& x, float e) {
    if (e == 1.0f) return;
    for (float& v : x) v = az_powf(v, e);
  }

  // ---- MCTS::probs, mcts.cc:575-618 -------------------------------------------
  std::vector<float> probs(float temp) const {
    const auto cnt = counts();
    const uint32_t M = cfg_.num_moves;
    std::vector<float> p(M, 0.0f);
    float count_sum = 0.0f;
    for (uint32_t m = 0; m < M; ++m) count_sum += static_cast<float>(cnt[m]);
    if (count_sum == 0) {
      const Node& r = pool_[root_];
      for (uint32_t i = 0; i < r.nchild; ++i) p[pool_[r.child0 + i].move] = pool_[r.child0 + i].policy;
      if (temp != 0.0f) pow_inplace(p, 1.0f / temp);
      const float s = seq_sum(p);
      for (float& v : p) v /= s;
      return p;
    }
    if (temp == 0) {
      std::vector<uint32_t> best{0};
      uint32_t best_count = cnt[0];
      for (uint32_t m = 1; m < M; ++m) {
        if (cnt[m] > best_count) { best_count = cnt[m]; best.clear(); best.push_back(m); }
        else if (cnt[m] == best_count) best.push_back(m);
      }
      for (uint32_t m : best) p[m] = static_cast<float>(1.0 / best.size());
      return p;
    }
    for (uint32_t m = 0; m < M; ++m) p[m] = static_cast<float>(cnt[m]);
    float s = seq_sum(p);
    for (float& v : p) v /= s;
    pow_inplace(p, 1 / temp);
    s = seq_sum(p);
    for (float& v : p) v /= s;
    return p;
  }

  // ---- MCTS::probs_pruned, mcts.cc:620-674 -------------------------------------
  std::vector<float> probs_pruned(float temp) const {
    const Node& r = pool_[root_];
    if (r.n <= 1) return probs(temp);
    const uint32_t M = cfg_.num_moves;
    const float explore_scaling = cfg_.cpuct * std::sqrt(static_cast<float>(r.n));
    float best_sel = -1e30f;
    for (uint32_t i = 0; i < r.nchild; ++i) {
      const Node& c = pool_[r.child0 + i];
      if (c.n == 0) continue;
      const float sel = c.q + explore_scaling * c.policy / static_cast<float>(c.n + 1);
      if (sel > best_sel) best_sel = sel;
    }
    std::vector<float> pruned(M, 0.0f);
    for (uint32_t i = 0; i < r.nchild; ++i) {
      const Node& c = pool_[r.child0 + i];
      if (c.n == 0) continue;
      const float explore_gap = best_sel - c.q;
      float desired;
      if (explore_gap <= 0) desired = static_cast<float>(c.n);
      else desired = explore_scaling * c.policy / explore_gap - 1.0f;
      pruned[c.move] = std::min(static_cast<float>(c.n), std::max(0.0f, desired));
    }
    const float total = seq_sum(pruned);
    if (total == 0) return probs(temp);
    if (temp == 0) {
      float best_val = pruned[0];
      for (uint32_t m = 1; m < M; ++m) best_val = std::max(best_val, pruned[m]);
      std::vector<float> result(M, 0.0f);
      int count = 0;
      for (uint32_t m = 0; m < M; ++m) if (pruned[m] == best_val) ++count;
      for (uint32_t m = 0; m < M; ++m) if (pruned[m] == best_val) result[m] = 1.0f / count;
      return result;
    }
    for (float& v : pruned) v /= total;
    if (temp != 1.0f) {
      pow_inplace(pruned, 1.0f / temp);
      const float s = seq_sum(pruned);
      for (float& v : pruned) v /= s;
    }
    return pruned;
  }

  // ---- MCTS::pick_move, mcts.cc:717-735 -----------------------------------------
  static uint32_t pick_move(const std::vector<float>& p, Pcg32& re) {
    const float choice = uniform01(re);
    float sum = 0.0f;
    for (uint32_t m = 0; m < p.size(); ++m) {
      sum += p[m];
      if (sum > choice) return m;
    }
    for (int64_t m = static_cast<int64_t>(p.size()) - 1; m >= 0; --m)
      if (p[m] > 0) return static_cast<uint32_t>(m);
    throw std::runtime_error("this shouldn't be possible.");
  }

  // ---- MCTS::normalized_root_entropy, mcts.cc:737-750 ----------------------------
  float normalized_root_entropy() const {
    const Node& r = pool_[root_];
    const float k = static_cast<float>(r.nchild);
    if (k <= 1 || r.n <= 1) return 0.0f;
    const float log_k = az_logf(k);
    float entropy = 0.0f;
    const float total_n = static_cast<float>(r.n);
    for (uint32_t i = 0; i < r.nchild; ++i) {
      const Node& c = pool_[r.child0 + i];
      if (c.n > 0) {
        const float p = static_cast<float>(c.n) / total_n;
        entropy -= p * az_logf(p);
      }
    }
    return entropy / log_k;
  }

  // ---- MCTS::principal_variation, mcts.cc:676-715 -----------------------------------
  // NB: with Gumbel on and no Gumbel state the reference's root step calls pick_move(probs(0)),
  // which draws from the shared stream; `re` is that stream.
  std::vector<uint32_t> principal_variation(uint32_t depth) {
    std::vector<uint32_t> pv;
    uint32_t node = root_;
    for (uint32_t i = 0; i < depth; ++i) {
      const Node& nd = pool_[node];
      if (nd.nchild == 0) break;
      uint32_t best = UINT32_MAX, best_n = 0;
      if (i == 0 && cfg_.gumbel_enabled) {
        const uint32_t mv = gumbel_final_action();
        for (uint32_t j = 0; j < nd.nchild; ++j)
          if (pool_[nd.child0 + j].move == mv) { best = nd.child0 + j; break; }
      }
      if (best == UINT32_MAX)
      for (uint32_t j = 0; j < nd.nchild; ++j)
        if (pool_[nd.child0 + j].n > best_n) { best_n = pool_[nd.child0 + j].n; best = nd.child0 + j; }
      if (best == UINT32_MAX || pool_[best].n == 0) break;
      pv.push_back(pool_[best].move);
      node = best;
    }
    return pv;
  }


  // ======================= Gumbel AlphaZero, mcts.cc:24-401 =========================
  static constexpr float kGumbelLogFloor = 1e-20f;  // mcts.cc:18

  // seq_halving_phase_plan, mcts.cc:28-66: (num_candidates, visits_per_candidate) per phase
  static std::vector<std::pair<uint32_t, uint32_t>> seq_halving_phase_plan(uint32_t m, uint32_t n) {
    std::vector<std::pair<uint32_t, uint32_t>> phases;
    if (m <= 1) { phases.emplace_back(1u, n); return phases; }
    uint32_t log2m = 0;
    for (uint32_t v = m - 1; v > 0; v >>= 1) ++log2m;
    if (log2m == 0) log2m = 1;
    const uint32_t base_v = std::max<uint32_t>(1u, n / (log2m * m));
    uint32_t sims_used = 0, num_c = m;
    for (uint32_t phase_idx = 0; phase_idx < log2m; ++phase_idx) {
      if (sims_used >= n) break;
      const uint32_t remaining = n - sims_used;
      const bool is_final = (phase_idx == log2m - 1);
      uint32_t v_per = is_final ? std::max<uint32_t>(1u, remaining / num_c) : base_v * (1u << phase_idx);
      if (num_c * v_per > remaining) {
        v_per = remaining / num_c;
        if (v_per == 0) { num_c = remaining; v_per = 1; }
      }
      phases.emplace_back(num_c, v_per);
      sims_used += num_c * v_per;
      num_c = std::max<uint32_t>(1u, num_c / 2);
    }
    return phases;
  }

  // compute_v_mix_from_children, mcts.cc:71-89
  static float v_mix_from_children(float raw_v, const std::vector<float>& qs, const std::vector<uint32_t>& ns,
                                   const std::vector<float>& priors) {
    float sum_visits = 0.0f, sum_priors_visited = 0.0f, weighted_num = 0.0f;
    for (size_t i = 0; i < qs.size(); ++i) {
      sum_visits += static_cast<float>(ns[i]);
      if (ns[i] > 0) { sum_priors_visited += priors[i]; weighted_num += priors[i] * qs[i]; }
    }
    if (sum_priors_visited <= 0.0f) return raw_v;
    const float weighted_q = weighted_num / sum_priors_visited;
    return (raw_v + sum_visits * weighted_q) / (sum_visits + 1.0f);
  }

  void set_gumbel_num_sims(uint32_t n) { gumbel_num_sims_target_ = n; reset_gumbel_state(); }  // mcts.cc:175-178
  bool gumbel_enabled() const { return cfg_.gumbel_enabled; }
  bool gumbel_initialized() const { return gumbel_initialized_; }
  const std::vector<size_t>& gumbel_survivors() const { return gumbel_survivors_; }
  const std::vector<float>& gumbel_g() const { return gumbel_g_; }

  void reset_gumbel_state() {  // mcts.cc:180-188
    gumbel_initialized_ = false;
    gumbel_effective_m_ = 0;
    gumbel_g_.clear(); gumbel_survivors_.clear(); gumbel_phases_.clear();
    gumbel_phase_idx_ = 0; gumbel_sims_in_phase_ = 0;
  }

  void init_gumbel_state() {  // mcts.cc:190-227
    const Node& r = pool_[root_];
    const uint32_t num_legal = r.nchild;
    if (num_legal == 0) return;
    const uint32_t remaining = depth_ < gumbel_num_sims_target_ ? gumbel_num_sims_target_ - depth_ : 0;
    if (remaining == 0) return;
    gumbel_effective_m_ = std::max<uint32_t>(1u, std::min({cfg_.gumbel_m, num_legal, remaining}));
    gumbel_g_.resize(num_legal);
    for (uint32_t i = 0; i < num_legal; ++i) gumbel_g_[i] = gumbel01(*re_);
    std::vector<size_t> idx(num_legal);
    for (uint32_t i = 0; i < num_legal; ++i) idx[i] = i;
    std::partial_sort(idx.begin(), idx.begin() + gumbel_effective_m_, idx.end(), [this, &r](size_t a, size_t b) {
      const float la = az_logf(pool_[r.child0 + a].policy + kGumbelLogFloor);
      const float lb = az_logf(pool_[r.child0 + b].policy + kGumbelLogFloor);
      return gumbel_g_[a] + la > gumbel_g_[b] + lb;
    });
    gumbel_survivors_.assign(idx.begin(), idx.begin() + gumbel_effective_m_);
    gumbel_phases_ = seq_halving_phase_plan(gumbel_effective_m_, remaining);
    gumbel_phase_idx_ = 0;
    gumbel_sims_in_phase_ = 0;
    gumbel_initialized_ = true;
  }

  void gumbel_advance_phase() {  // mcts.cc:229-264
    if (gumbel_phase_idx_ + 1 >= gumbel_phases_.size()) return;
    const uint32_t next_num_c = gumbel_phases_[gumbel_phase_idx_ + 1].first;
    if (next_num_c >= gumbel_survivors_.size()) { ++gumbel_phase_idx_; gumbel_sims_in_phase_ = 0; return; }
    const Node& r = pool_[root_];
    uint32_t max_visit = 0;
    for (size_t ci : gumbel_survivors_) max_visit = std::max(max_visit, pool_[r.child0 + ci].n);
    const float sigma_scale = (cfg_.gumbel_c_visit + static_cast<float>(max_visit)) * cfg_.gumbel_c_scale;
    std::vector<std::pair<float, size_t>> scored;
    for (size_t ci : gumbel_survivors_) {
      const Node& c = pool_[r.child0 + ci];
      const float logit = az_logf(c.policy + kGumbelLogFloor);
      const float q_hat = c.n > 0 ? c.q : 0.0f;
      scored.emplace_back(gumbel_g_[ci] + logit + sigma_scale * q_hat, ci);
    }
    std::partial_sort(scored.begin(), scored.begin() + next_num_c, scored.end(),
                      [](const auto& a, const auto& b) { return a.first > b.first; });
    gumbel_survivors_.resize(next_num_c);
    for (size_t i = 0; i < next_num_c; ++i) gumbel_survivors_[i] = scored[i].second;
    ++gumbel_phase_idx_;
    gumbel_sims_in_phase_ = 0;
  }

  size_t gumbel_next_root_child() {  // mcts.cc:266-283
    if (gumbel_phase_idx_ < gumbel_phases_.size()) {
      const auto& ph = gumbel_phases_[gumbel_phase_idx_];
      if (gumbel_sims_in_phase_ >= ph.first * ph.second) gumbel_advance_phase();
    }
    if (gumbel_survivors_.empty()) return 0;
    const size_t pick = gumbel_sims_in_phase_ % gumbel_survivors_.size();
    ++gumbel_sims_in_phase_;
    return gumbel_survivors_[pick];
  }

  // softmax(log(prior) + sigma * completedQ) over the children of `node`; shared by
  // gumbel_interior_select (mcts.cc:285-334) and gumbel_improved_policy (mcts.cc:336-373)
  void gumbel_pi_prime(uint32_t node, std::vector<float>& z, float& z_sum, std::vector<uint32_t>& ns) const {
    const Node& nd = pool_[node];
    const uint32_t k = nd.nchild;
    uint32_t max_visit = 0;
    std::vector<float> qs(k), priors(k);
    ns.assign(k, 0);
    for (uint32_t i = 0; i < k; ++i) {
      const Node& c = pool_[nd.child0 + i];
      max_visit = std::max(max_visit, c.n);
      qs[i] = c.q; ns[i] = c.n; priors[i] = c.policy;
    }
    const float v_mix = v_mix_from_children(nd.v, qs, ns, priors);
    const float sigma_scale = (cfg_.gumbel_c_visit + static_cast<float>(max_visit)) * cfg_.gumbel_c_scale;
    z.assign(k, 0.0f);
    float z_max = -std::numeric_limits<float>::infinity();
    for (uint32_t i = 0; i < k; ++i) {
      const float completed_q = ns[i] > 0 ? qs[i] : v_mix;
      z[i] = az_logf(priors[i] + kGumbelLogFloor) + sigma_scale * completed_q;
      if (z[i] > z_max) z_max = z[i];
    }
    z_sum = 0.0f;
    for (uint32_t i = 0; i < k; ++i) { z[i] = az_expf(z[i] - z_max); z_sum += z[i]; }
  }

  size_t gumbel_interior_select(uint32_t node) const {  // mcts.cc:285-334
    std::vector<float> z; std::vector<uint32_t> ns; float z_sum;
    gumbel_pi_prime(node, z, z_sum, ns);
    uint32_t sum_visits = 0;
    for (uint32_t n : ns) sum_visits += n;
    const float inv = z_sum > 0 ? (1.0f / z_sum) : 0.0f;
    const float denom = 1.0f + static_cast<float>(sum_visits);
    size_t best = 0;
    float best_score = -std::numeric_limits<float>::infinity();
    for (size_t i = 0; i < z.size(); ++i) {
      const float score = z[i] * inv - static_cast<float>(ns[i]) / denom;
      if (score > best_score) { best_score = score; best = i; }
    }
    return best;
  }

  std::vector<float> gumbel_improved_policy() const {  // mcts.cc:336-373
    std::vector<float> out(cfg_.num_moves, 0.0f);
    const Node& r = pool_[root_];
    if (r.nchild == 0) return out;
    std::vector<float> z; std::vector<uint32_t> ns; float z_sum;
    gumbel_pi_prime(root_, z, z_sum, ns);
    if (z_sum <= 0) return out;
    for (uint32_t i = 0; i < r.nchild; ++i) out[pool_[r.child0 + i].move] = z[i] / z_sum;
    return out;
  }

  uint32_t gumbel_final_action() {  // mcts.cc:375-401
    if (!gumbel_initialized_ || gumbel_survivors_.empty()) return pick_move(probs(0.0f), *re_);
    const Node& r = pool_[root_];
    uint32_t max_visit = 0;
    for (uint32_t i = 0; i < r.nchild; ++i) max_visit = std::max(max_visit, pool_[r.child0 + i].n);
    const float sigma_scale = (cfg_.gumbel_c_visit + static_cast<float>(max_visit)) * cfg_.gumbel_c_scale;
    size_t best = gumbel_survivors_[0];
    float best_score = -std::numeric_limits<float>::infinity();
    for (size_t ci : gumbel_survivors_) {
      const Node& c = pool_[r.child0 + ci];
      const float logit = az_logf(c.policy + kGumbelLogFloor);
      const float q_hat = c.n > 0 ? c.q : 0.0f;
      const float score = gumbel_g_[ci] + logit + sigma_scale * q_hat;
      if (score > best_score) { best_score = score; best = ci; }
    }
    return pool_[r.child0 + best].move;
  }

  std::vector<std::pair<uint64_t, uint64_t>>* trace = nullptr;  // debug: (tag, rng state)
  void tr(uint64_t tag) { if (trace) trace->push_back({tag, re_->state}); }
  uint32_t depth() const { return depth_; }
  float avg_leaf_depth() const {  // mcts.h:112-114
    return depth_ == 0 ? 0.0f : static_cast<float>(total_leaf_depth_) / static_cast<float>(depth_);
  }
  uint32_t num_root_children() const { return pool_[root_].nchild; }
  uint32_t root_n() const { return pool_[root_].n; }
  const MctsConfig& config() const { return cfg_; }
  const Node& root() const { return pool_[root_]; }
  const Node& node(uint32_t i) const { return pool_[i]; }
  size_t pool_size() const { return pool_.size(); }
  // children of the root in stored (shuffled) order: test hook
  std::vector<uint32_t> root_child_moves() const {
    std::vector<uint32_t> out;
    const Node& r = pool_[root_];
    for (uint32_t i = 0; i < r.nchild; ++i) out.push_back(pool_[r.child0 + i].move);
    return out;
  }

 private:
  MctsConfig cfg_;
  Pcg32* re_;
  std::vector<Node> pool_;
  uint32_t root_ = 0;
  uint32_t current_ = 0;
  std::vector<uint32_t> path_;
  struct InFlight { std::vector<uint32_t> path; uint32_t leaf = 0; };  // mcts.h InFlightLeaf
  std::vector<InFlight> in_flight_;
  uint32_t depth_ = 0;
  uint64_t total_leaf_depth_ = 0;
  // per-search Gumbel state, mcts.h:179-191
  bool gumbel_initialized_ = false;
  uint32_t gumbel_num_sims_target_ = 0, gumbel_effective_m_ = 0;
  std::vector<float> gumbel_g_;
  std::vector<size_t> gumbel_survivors_;
  std::vector<std::pair<uint32_t, uint32_t>> gumbel_phases_;
  uint32_t gumbel_phase_idx_ = 0, gumbel_sims_in_phase_ = 0;
};

// game_state.h:160-173 — the deterministic evaluator of the parity tiers.
inline void dumb_eval(const Game& gs, float* value, float* pi) {
  const uint32_t M = gs.num_moves();
  std::vector<uint8_t> valids(M);
  gs.valid_moves(valids.data());
  const int P = gs.num_players();
  for (int i = 0; i <= P; ++i) value[i] = static_cast<float>(1.0 / (P + 1));
  for (uint32_t m = 0; m < M; ++m) pi[m] = 0.0f;
  uint8_t s8 = 0;  // Vector<uint8_t>::sum() wraps mod 256 (shapes.h:14)
  for (uint32_t m = 0; m < M; ++m) s8 = static_cast<uint8_t>(s8 + valids[m]);
  const float sum = s8;
  if (sum == 0.0) return;
  for (uint32_t m = 0; m < M; ++m) pi[m] = static_cast<float>(valids[m]) / sum;
}

// playout_eval, game_state.cc:10-54: policy = uniform over the leaf's legal moves (the same u8-wrapping sum as dumb_eval),
// value = the scores of a uniformly random rollout.  The reference draws from an UNSEEDABLE thread-local
// std::default_random_engine; here the rollout stream is a pcg32 the caller owns and one draw per rollout move picks
// the lemire_below(stream, #legal)-th legal move in ascending move order — build-defined, the same on the device.
inline void playout_eval(const Game& gs, Pcg32& roll, float* value, float* pi) {
  dumb_eval(gs, value, pi);          // the policy half; `value` is overwritten below
  const uint32_t M = gs.num_moves();
  const int P = gs.num_players();
  auto sim = gs.copy();
  std::vector<uint8_t> valids(M);
  float sc[kMaxValue];
  while (!sim->scores(sc)) {
    sim->valid_moves(valids.data());
    std::vector<uint32_t> idx;
    for (uint32_t m = 0; m < M; ++m) if (valids[m]) idx.push_back(m);
    if (idx.empty()) break;
    sim->play_move(idx[lemire_below(roll, static_cast<uint32_t>(idx.size()))]);
  }
  if (sim->scores(sc)) {
    // relative_values games rotate the scores to the leaf player's view (absolute_to_relative); no such game is restated
    for (int i = 0; i <= P; ++i) value[i] = sc[i];
    return;
  }
  for (int i = 0; i <= P; ++i) value[i] = static_cast<float>(1.0 / (P + 1));
}

}  // namespace orc
