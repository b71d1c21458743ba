// ORACLE — TEST INFRASTRUCTURE ONLY (see az_rng.hpp header for the usage rule).
// extern "C" surface over the restatement so tests/ (ctypes) and bench.py's
// cpu_baseline leg can drive it.  Builds to oracle/liboracle.so (oracle/Makefile).
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "az_games.hpp"
#include "az_mcts.hpp"
#include "az_playmanager.hpp"
#include "az_rng.hpp"
#include "az_stargambit.hpp"
#include "az_symmetries.hpp"
#include "az_tafl.hpp"
#include "az_tafl_family.hpp"

using namespace orc;

namespace {
std::unique_ptr<Game> make_game(int game_id) {
  switch (game_id) {
    case 0: return std::make_unique<Connect4>();
    case 1: return std::make_unique<Tawlbwrdd>();
    case 2: return std::make_unique<Brandubh>();
    case 3: return std::make_unique<OpenTafl>();
    case 4: { const float p[4] = {0.25f, 0.25f, 0.25f, 0.25f}; return std::make_unique<StarGambitUnified>(-1, p, 0); }
    default: throw std::runtime_error("unknown game id");
  }
}
struct MctsBox {
  Pcg32 rng;
  std::unique_ptr<Mcts> m;
  std::unique_ptr<Game> leaf;
};
struct PmBox {
  std::unique_ptr<PlayManager> pm;
};
}  // namespace

extern "C" {

// ---------------------------------------------------------------- RNG layer
void orc_pcg32_outputs(uint64_t seed, uint32_t n, uint32_t* out) {
  Pcg32 g; g.seed(seed);
  for (uint32_t i = 0; i < n; ++i) out[i] = g.next();
}
// shuffles {0..n-1} `reps` times in a row from one seeded stream; out[reps*n]
void orc_shuffle_iota(uint64_t seed, uint32_t n, uint32_t reps, uint32_t* out) {
  Pcg32 g; g.seed(seed);
  for (uint32_t r = 0; r < reps; ++r) {
    for (uint32_t i = 0; i < n; ++i) out[r * n + i] = i;
    shuffle(out + r * n, n, g);
  }
}
void orc_uniform01(uint64_t seed, uint32_t n, float* out) {
  Pcg32 g; g.seed(seed);
  for (uint32_t i = 0; i < n; ++i) out[i] = uniform01(g);
}
// fresh_each != 0: a new distribution object per draw (shaped-Dirichlet pattern, mcts.cc:430)
void orc_gamma(uint64_t seed, float alpha, float beta, uint32_t n, int fresh_each, float* out) {
  Pcg32 g; g.seed(seed);
  Gamma dist(alpha, beta);
  for (uint32_t i = 0; i < n; ++i) {
    if (fresh_each) { Gamma d2(alpha, beta); out[i] = d2(g); }
    else out[i] = dist(g);
  }
}
uint64_t orc_gamma_raw(uint64_t state, float alpha, uint32_t n, float* out) {
  Pcg32 g; g.state = state;
  Gamma dist(alpha, 1.0f);
  for (uint32_t i = 0; i < n; ++i) out[i] = dist(g);
  return g.state;
}
void orc_gumbel(uint64_t seed, uint32_t n, float* out) {
  Pcg32 g; g.seed(seed);
  for (uint32_t i = 0; i < n; ++i) out[i] = gumbel01(g);
}

// ---------------------------------------------------------------- games
void* orc_game_new(int game_id) { try { return make_game(game_id).release(); } catch (...) { return nullptr; } }
void* orc_c4_from_board(const int8_t* board, int8_t player, int32_t turn) {
  return new Connect4(board, player, turn);
}
// Tafl family from a given board (test helper MakeGS, opentafl_gs_test.cc:97-101): int8 board[3][N][N]
void* orc_tafl_from_board(int game_id, const int8_t* board, int8_t player, uint32_t turn, uint32_t max_turns) {
  try {
    switch (game_id) {
      case 2: return new Brandubh(board, player, static_cast<uint16_t>(turn), static_cast<uint16_t>(max_turns));
      case 3: return new OpenTafl(board, player, static_cast<uint16_t>(turn), static_cast<uint16_t>(max_turns));
      default: return nullptr;
    }
  } catch (...) { return nullptr; }
}
// StarGambitUnifiedGS(pinned_variant, probs) (py_wrapper.cc:662-667); `first_variant` is the variant the constructor's own
// draw would have produced (the reference draws it from an unseedable engine; a pinned game ignores it)
void* orc_sg_unified_new(int pinned, const float* probs, int first_variant) {
  const int v = (pinned >= 0 && pinned <= 3) ? pinned : first_variant;
  if (v < 0 || v > 3) return nullptr;
  return new StarGambitUnified(pinned, probs, v);
}
// StarGambit{Skirmish,Showdown,Clash,Battle}GS in their own action space (star_gambit_gs.h:752-755)
void* orc_sg_plain_new(int variant) { return (variant < 0 || variant > 3) ? nullptr : new StarGambit(variant); }
static sg::Inner* sg_inner(void* g) {
  if (auto* u = dynamic_cast<StarGambitUnified*>(static_cast<Game*>(g))) return &u->in;
  if (auto* p = dynamic_cast<StarGambit*>(static_cast<Game*>(g))) return &p->in;
  return nullptr;
}
// get_units (star_gambit_gs.cc, UnitInfo): rows of 9 int32 = type, player, slot, hp, facing, q, r, moves_left, cannons_fired
// (dead units included, in creation order); returns the count
uint32_t orc_sg_units(void* g, int32_t* out, uint32_t cap) {
  sg::Inner* in = sg_inner(g);
  if (!in) return 0;
  uint32_t n = 0;
  for (const auto& u : in->units) {
    if (n >= cap) break;
    int32_t* r = out + 9 * n++;
    r[0] = u.type; r[1] = u.player; r[2] = u.slot; r[3] = u.hp; r[4] = u.facing; r[5] = u.q; r[6] = u.r; r[7] = u.moves_left; r[8] = u.cannons_fired;
  }
  return n;
}
// out[0..12] = reserves[2][4], acted, over, winner, history length, variant (-1 for the plain games)
void orc_sg_info(void* g, int32_t* out) {
  sg::Inner* in = sg_inner(g);
  if (!in) return;
  for (int p = 0; p < 2; ++p) for (int t = 0; t < 4; ++t) out[p * 4 + t] = in->reserves[p][t];
  out[8] = in->acted; out[9] = in->over; out[10] = in->winner; out[11] = static_cast<int32_t>(in->history.size());
  out[12] = static_cast<Game*>(g)->get_variant_id();
}
// to_bytes image (star_gambit_gs.cc:2253-2288 / 2451-2465); returns the size, writes at most cap bytes
uint32_t orc_sg_to_bytes(void* g, uint8_t* out, uint32_t cap) {
  std::string b;
  if (auto* u = dynamic_cast<StarGambitUnified*>(static_cast<Game*>(g))) b = u->to_bytes();
  else if (auto* p = dynamic_cast<StarGambit*>(static_cast<Game*>(g))) b = p->in.to_bytes();
  std::memcpy(out, b.data(), std::min<size_t>(cap, b.size()));
  return static_cast<uint32_t>(b.size());
}
// inner from_bytes (star_gambit_gs.cc:2290-2338) into an existing game object; 0 on success
int orc_sg_from_bytes(void* g, const uint8_t* data, uint32_t n) {
  sg::Inner* in = sg_inner(g);
  if (!in) return -1;
  try { in->from_bytes(std::string(reinterpret_cast<const char*>(data), n)); return 0; } catch (...) { return -1; }
}
int orc_game_equal(void* a, void* b) {
  sg::Inner *x = sg_inner(a), *y = sg_inner(b);
  if (!x || !y) return -1;
  if (static_cast<Game*>(a)->get_variant_id() != static_cast<Game*>(b)->get_variant_id()) return 0;
  return x->same(*y) ? 1 : 0;
}
int orc_game_relative_values(void* g) { return static_cast<Game*>(g)->relative_values() ? 1 : 0; }
int orc_game_variant(void* g) { return static_cast<Game*>(g)->get_variant_id(); }
int orc_game_num_variants(void* g) { return static_cast<Game*>(g)->num_variants(); }

void* orc_game_copy(void* g) { return static_cast<Game*>(g)->copy().release(); }
void orc_game_free(void* g) { delete static_cast<Game*>(g); }
int orc_game_play(void* g, uint32_t m) {
  try { static_cast<Game*>(g)->play_move(m); return 0; } catch (...) { return -1; }
}
uint32_t orc_game_num_moves(void* g) { return static_cast<Game*>(g)->num_moves(); }
uint32_t orc_game_num_players(void* g) { return static_cast<Game*>(g)->num_players(); }
uint32_t orc_game_player(void* g) { return static_cast<Game*>(g)->current_player(); }
uint32_t orc_game_turn(void* g) { return static_cast<Game*>(g)->current_turn(); }
void orc_game_valid(void* g, uint8_t* out) { static_cast<Game*>(g)->valid_moves(out); }
int orc_game_scores(void* g, float* out) { return static_cast<Game*>(g)->scores(out) ? 1 : 0; }
void orc_game_canonical_shape(void* g, int* chw) {
  static_cast<Game*>(g)->canonical_shape(chw, chw + 1, chw + 2);
}
void orc_game_canonical(void* g, float* out) { static_cast<Game*>(g)->canonicalized(out); }
uint64_t orc_game_key(void* g) { return static_cast<Game*>(g)->key(); }

// ---------------------------------------------------------------- single tree
struct OrcMctsCfg {
  float cpuct;
  uint32_t num_players, num_moves;
  float epsilon, root_policy_temp, fpu_reduction;
  int32_t relative_values, root_fpu_zero, shaped_dirichlet;
  int32_t gumbel_enabled; uint32_t gumbel_m; float gumbel_c_visit, gumbel_c_scale; int32_t gumbel_full;
};
void* orc_mcts_new(const OrcMctsCfg* c, uint64_t seed) {
  auto* b = new MctsBox();
  b->rng.seed(seed);
  MctsConfig mc;
  mc.cpuct = c->cpuct; mc.num_players = c->num_players; mc.num_moves = c->num_moves;
  mc.epsilon = c->epsilon; mc.root_policy_temp = c->root_policy_temp; mc.fpu_reduction = c->fpu_reduction;
  mc.relative_values = c->relative_values != 0; mc.root_fpu_zero = c->root_fpu_zero != 0;
  mc.shaped_dirichlet = c->shaped_dirichlet != 0;
  mc.gumbel_enabled = c->gumbel_enabled != 0; mc.gumbel_m = c->gumbel_m;
  mc.gumbel_c_visit = c->gumbel_c_visit; mc.gumbel_c_scale = c->gumbel_c_scale; mc.gumbel_full = c->gumbel_full != 0;
  b->m = std::make_unique<Mcts>(mc, &b->rng);
  return b;
}
void orc_mcts_free(void* h) { delete static_cast<MctsBox*>(h); }
// ---- Gumbel hooks (mcts.cc:24-401)
void orc_mcts_set_gumbel_num_sims(void* h, uint32_t n) { static_cast<MctsBox*>(h)->m->set_gumbel_num_sims(n); }
void orc_mcts_gumbel_improved_policy(void* h, float* out) {
  auto v = static_cast<MctsBox*>(h)->m->gumbel_improved_policy();
  std::memcpy(out, v.data(), v.size() * sizeof(float));
}
uint32_t orc_mcts_gumbel_final_action(void* h) { return static_cast<MctsBox*>(h)->m->gumbel_final_action(); }
// survivors as root-child indices, g per root child; returns #survivors (0 if not initialised)
uint32_t orc_mcts_gumbel_state(void* h, uint32_t* survivors, float* g, uint32_t cap) {
  auto& m = *static_cast<MctsBox*>(h)->m;
  if (!m.gumbel_initialized()) return 0;
  const auto& sv = m.gumbel_survivors(); const auto& gg = m.gumbel_g();
  for (uint32_t i = 0; i < sv.size() && i < cap; ++i) survivors[i] = static_cast<uint32_t>(sv[i]);
  for (uint32_t i = 0; i < gg.size() && i < cap; ++i) g[i] = gg[i];
  return static_cast<uint32_t>(sv.size());
}
// seq_halving_phase_plan(m, n) -> out[2*i] = num_c, out[2*i+1] = v_per; returns #phases
uint32_t orc_seq_halving_phase_plan(uint32_t m, uint32_t n, uint32_t* out, uint32_t cap) {
  auto ph = Mcts::seq_halving_phase_plan(m, n);
  for (uint32_t i = 0; i < ph.size() && i < cap; ++i) { out[2 * i] = ph[i].first; out[2 * i + 1] = ph[i].second; }
  return static_cast<uint32_t>(ph.size());
}
float orc_v_mix(float raw_v, const float* q, const uint32_t* n, const float* prior, uint32_t k) {
  return Mcts::v_mix_from_children(raw_v, std::vector<float>(q, q + k), std::vector<uint32_t>(n, n + k),
                                   std::vector<float>(prior, prior + k));
}
// find_leaf; the leaf state is kept inside the box (orc_mcts_leaf to borrow it)
void orc_mcts_find_leaf(void* h, void* game) {
  auto* b = static_cast<MctsBox*>(h);
  b->leaf = b->m->find_leaf(*static_cast<Game*>(game));
}
void* orc_mcts_leaf(void* h) { return static_cast<MctsBox*>(h)->leaf.get(); }
void orc_mcts_process_result(void* h, float* value, const float* pi, int noise) {
  static_cast<MctsBox*>(h)->m->process_result(value, pi, noise != 0);
}
// dumb_eval(gs), game_state.h:160-173
void orc_dumb_eval(void* game, float* v, float* pi) { dumb_eval(*static_cast<Game*>(game), v, pi); }
// playout_eval(gs) with an explicit rollout stream seed (game_state.cc:10-59)
void orc_playout_eval(void* game, uint64_t seed, float* v, float* pi) {
  Pcg32 roll; roll.seed(seed);
  playout_eval(*static_cast<Game*>(game), roll, v, pi);
}
// WU-UCT batched API (mcts.cc:752-851)
void orc_mcts_find_leaf_batched(void* h, void* game) {
  auto* b = static_cast<MctsBox*>(h);
  b->leaf = b->m->find_leaf_batched(*static_cast<Game*>(game));
}
int orc_mcts_process_result_batched(void* h, uint32_t leaf_index, float* value, const float* pi, int noise) {
  try { static_cast<MctsBox*>(h)->m->process_result_batched(leaf_index, value, pi, noise != 0); return 0; }
  catch (const std::out_of_range&) { return -1; }
}
uint32_t orc_mcts_in_flight_count(void* h) { return static_cast<MctsBox*>(h)->m->in_flight_count(); }
void orc_mcts_reset_batch(void* h) { static_cast<MctsBox*>(h)->m->reset_batch(); }
// `sims` x (find_leaf, dumb_eval, process_result) — mcts_test.cc:41-72 pattern
void orc_mcts_search_dumb(void* h, void* game, uint32_t sims, int noise) {
  auto* b = static_cast<MctsBox*>(h);
  Game* gs = static_cast<Game*>(game);
  std::vector<float> v(gs->num_players() + 1), pi(gs->num_moves());
  for (uint32_t s = 0; s < sims; ++s) {
    auto leaf = b->m->find_leaf(*gs);
    dumb_eval(*leaf, v.data(), pi.data());
    b->m->process_result(v.data(), pi.data(), noise != 0);
  }
}
int orc_mcts_update_root(void* h, void* game, uint32_t move) {
  try { static_cast<MctsBox*>(h)->m->update_root(*static_cast<Game*>(game), move); return 0; }
  catch (...) { return -1; }
}
void orc_mcts_counts(void* h, uint32_t* out) {
  auto c = static_cast<MctsBox*>(h)->m->counts();
  std::memcpy(out, c.data(), c.size() * sizeof(uint32_t));
}
void orc_mcts_root_q(void* h, float* out) {
  auto c = static_cast<MctsBox*>(h)->m->root_q_values();
  std::memcpy(out, c.data(), c.size() * sizeof(float));
}
void orc_mcts_probs(void* h, float temp, int pruned, float* out) {
  auto* m = static_cast<MctsBox*>(h)->m.get();
  auto p = pruned ? m->probs_pruned(temp) : m->probs(temp);
  std::memcpy(out, p.data(), p.size() * sizeof(float));
}
uint32_t orc_mcts_pick_move(void* h, const float* p, uint32_t n) {
  auto* b = static_cast<MctsBox*>(h);
  return Mcts::pick_move(std::vector<float>(p, p + n), b->rng);
}
uint32_t orc_mcts_depth(void* h) { return static_cast<MctsBox*>(h)->m->depth(); }
uint32_t orc_mcts_root_n(void* h) { return static_cast<MctsBox*>(h)->m->root_n(); }
float orc_mcts_avg_leaf_depth(void* h) { return static_cast<MctsBox*>(h)->m->avg_leaf_depth(); }
float orc_mcts_entropy(void* h) { return static_cast<MctsBox*>(h)->m->normalized_root_entropy(); }
void orc_mcts_root_value(void* h, float* wld) { static_cast<MctsBox*>(h)->m->root_value(wld); }
uint32_t orc_mcts_principal_variation(void* h, uint32_t depth, uint32_t* out) {
  auto pv = static_cast<MctsBox*>(h)->m->principal_variation(depth);
  for (size_t i = 0; i < pv.size(); ++i) out[i] = pv[i];
  return static_cast<uint32_t>(pv.size());
}
void orc_mcts_add_root_noise(void* h) { static_cast<MctsBox*>(h)->m->add_root_noise(); }
void orc_mcts_apply_root_policy_temp(void* h) { static_cast<MctsBox*>(h)->m->apply_root_policy_temp(); }
// root children in stored order: moves[k], policy[k], n[k], q[k]; returns k
uint32_t orc_mcts_root_children(void* h, uint32_t* moves, float* policy, uint32_t* n, float* q) {
  auto* m = static_cast<MctsBox*>(h)->m.get();
  const Node& r = m->root();
  for (uint32_t i = 0; i < r.nchild; ++i) {
    const Node& c = m->node(r.child0 + i);
    moves[i] = c.move; policy[i] = c.policy; n[i] = c.n; q[i] = c.q;
  }
  return r.nchild;
}
// Node::uct on explicit numbers (mcts_test.cc:14-38 known answers)
float orc_node_uct(float q, float policy, uint32_t n, float sqrt_parent_n, float cpuct, float fpu) {
  Node c; c.q = q; c.policy = policy; c.n = n;
  return Mcts::uct(c, sqrt_parent_n, cpuct, fpu);
}

// ---------------------------------------------------------------- PlayManager
struct OrcPlayParams {
  uint32_t games_to_play, concurrent_games, max_batch_size, max_cache_size, cache_shards;
  uint32_t mcts_visits[4];
  float cpuct, start_temp, final_temp, temp_decay_half_life;
  int32_t history_enabled, tree_reuse;
  float epsilon, mcts_root_temp;
  int32_t playout_cap_randomization;
  uint32_t playout_cap_depth;
  float playout_cap_percent, fpu_reduction;
  int32_t root_fpu_zero, shaped_dirichlet, policy_target_pruning;
  float resign_percent, resign_playthrough_percent;
  int32_t eval_type[4];  // -1 = unset (all NN)
  int32_t gumbel_enabled; uint32_t gumbel_m; float gumbel_c_visit, gumbel_c_scale;
  int32_t gumbel_full, fast_search_uses_gumbel;
  // model groups / seat permutations / per-seat overrides (0 counts = not given)
  uint32_t num_model_groups_given; uint8_t model_groups[4];
  uint32_t num_seat_perms; uint8_t seat_perms[8][4];
  int32_t has_seat_visits, has_seat_cap_visits, has_seat_epsilon, has_seat_mcts_root_temp, has_seat_root_fpu_zero;
  uint32_t seat_visits[8][4], seat_cap_visits[8][4];
  float seat_epsilon[8][4], seat_mcts_root_temp[8][4];
  uint8_t seat_root_fpu_zero[8][4];
  uint32_t perm_base;
  int32_t has_seat_gumbel_enabled, has_seat_gumbel_m, has_seat_gumbel_c_visit, has_seat_gumbel_c_scale, has_seat_gumbel_full,
          has_seat_gumbel_use_improved_policy, has_seat_resign_threshold, has_seat_resign_consecutive;
  uint8_t seat_gumbel_enabled[8][4], seat_gumbel_full[8][4], seat_gumbel_use_improved_policy[8][4];
  uint32_t seat_gumbel_m[8][4], seat_resign_consecutive[8][4];
  float seat_gumbel_c_visit[8][4], seat_gumbel_c_scale[8][4], seat_resign_threshold[8][4];
};
typedef void (*orc_group_eval_fn)(uint32_t group, const float* canonical, uint32_t n, float* v, float* pi, void* user);
typedef void (*orc_eval_fn)(const float* canonical, uint32_t n, float* v, float* pi, void* user);

static void* pm_new_from(std::unique_ptr<Game> base, const OrcPlayParams* c, uint64_t seed, int per_slot_rng, int record_moves,
                         const float* half_life_by_variant, uint32_t n_half_life);
void* orc_pm_new(int game_id, const OrcPlayParams* c, uint64_t seed, int per_slot_rng, int record_moves) {
  try { return pm_new_from(make_game(game_id), c, seed, per_slot_rng, record_moves, nullptr, 0); } catch (...) { return nullptr; }
}
// PlayManager over a caller-made base game (copied, play_manager.cc / py_wrapper.cc:352-360) + temp_decay_half_life_by_variant
void* orc_pm_new_game(void* base_game, const OrcPlayParams* c, uint64_t seed, int per_slot_rng, int record_moves,
                      const float* half_life_by_variant, uint32_t n_half_life) {
  try { return pm_new_from(static_cast<Game*>(base_game)->copy(), c, seed, per_slot_rng, record_moves, half_life_by_variant, n_half_life); }
  catch (...) { return nullptr; }
}
static void* pm_new_from(std::unique_ptr<Game> base, const OrcPlayParams* c, uint64_t seed, int per_slot_rng, int record_moves,
                         const float* half_life_by_variant, uint32_t n_half_life) {
  try {
    const uint32_t P = base->num_players();
    PlayParams p;
    p.games_to_play = c->games_to_play; p.concurrent_games = c->concurrent_games;
    p.max_batch_size = c->max_batch_size; p.max_cache_size = c->max_cache_size;
    p.cache_shards = c->cache_shards ? c->cache_shards : 1;
    p.mcts_visits.assign(c->mcts_visits, c->mcts_visits + P);
    p.cpuct = c->cpuct; p.start_temp = c->start_temp; p.final_temp = c->final_temp;
    p.temp_decay_half_life = c->temp_decay_half_life;
    if (half_life_by_variant) p.temp_decay_half_life_by_variant.assign(half_life_by_variant, half_life_by_variant + n_half_life);
    p.history_enabled = c->history_enabled != 0; p.tree_reuse = c->tree_reuse != 0;
    p.epsilon = c->epsilon; p.mcts_root_temp = c->mcts_root_temp;
    p.playout_cap_randomization = c->playout_cap_randomization != 0;
    p.playout_cap_depth = c->playout_cap_depth; p.playout_cap_percent = c->playout_cap_percent;
    p.fpu_reduction = c->fpu_reduction; p.root_fpu_zero = c->root_fpu_zero != 0;
    p.shaped_dirichlet = c->shaped_dirichlet != 0; p.policy_target_pruning = c->policy_target_pruning != 0;
    p.resign_percent = c->resign_percent; p.resign_playthrough_percent = c->resign_playthrough_percent;
    p.gumbel_enabled = c->gumbel_enabled != 0; p.gumbel_m = c->gumbel_m;
    p.gumbel_c_visit = c->gumbel_c_visit; p.gumbel_c_scale = c->gumbel_c_scale;
    p.gumbel_full = c->gumbel_full != 0; p.fast_search_uses_gumbel = c->fast_search_uses_gumbel != 0;
    if (c->eval_type[0] >= 0)
      for (uint32_t i = 0; i < P; ++i) p.eval_type.push_back(static_cast<EvalType>(c->eval_type[i]));
    for (uint32_t i = 0; i < c->num_model_groups_given; ++i) p.model_groups.push_back(c->model_groups[i]);
    const uint32_t np = c->num_seat_perms;
    for (uint32_t q = 0; q < np; ++q) p.seat_perms.emplace_back(c->seat_perms[q], c->seat_perms[q] + P);
    const uint32_t rows = np ? np : 1;
    for (uint32_t q = 0; q < rows; ++q) {
      if (c->has_seat_visits) p.seat_visits.emplace_back(c->seat_visits[q], c->seat_visits[q] + P);
      if (c->has_seat_cap_visits) p.seat_cap_visits.emplace_back(c->seat_cap_visits[q], c->seat_cap_visits[q] + P);
      if (c->has_seat_epsilon) p.seat_epsilon.emplace_back(c->seat_epsilon[q], c->seat_epsilon[q] + P);
      if (c->has_seat_mcts_root_temp) p.seat_mcts_root_temp.emplace_back(c->seat_mcts_root_temp[q], c->seat_mcts_root_temp[q] + P);
      if (c->has_seat_root_fpu_zero) p.seat_root_fpu_zero.emplace_back(c->seat_root_fpu_zero[q], c->seat_root_fpu_zero[q] + P);
#define ORC_SEAT(name) if (c->has_##name) p.name.emplace_back(c->name[q], c->name[q] + P)
      ORC_SEAT(seat_gumbel_enabled); ORC_SEAT(seat_gumbel_m); ORC_SEAT(seat_gumbel_c_visit); ORC_SEAT(seat_gumbel_c_scale);
      ORC_SEAT(seat_gumbel_full); ORC_SEAT(seat_gumbel_use_improved_policy); ORC_SEAT(seat_resign_threshold);
      ORC_SEAT(seat_resign_consecutive);
#undef ORC_SEAT
    }
    auto* b = new PmBox();
    b->pm = std::make_unique<PlayManager>(std::move(base), p, seed, per_slot_rng != 0, c->perm_base);
    b->pm->record_moves = (record_moves & 1) != 0;
    b->pm->trace_on = (record_moves & 2) != 0;
    return b;
  } catch (...) { return nullptr; }
}
uint64_t orc_pm_trace(void* h, uint64_t* out, uint64_t cap) {
  auto& t = static_cast<PmBox*>(h)->pm->trace;
  uint64_t n = std::min<uint64_t>(cap, t.size());
  for (uint64_t i = 0; i < n; ++i) { out[2 * i] = t[i].first; out[2 * i + 1] = t[i].second; }
  return n;
}
void orc_pm_free(void* h) { delete static_cast<PmBox*>(h); }
int orc_pm_run_groups(void* h, orc_group_eval_fn fn, void* user) {
  try {
    static_cast<PmBox*>(h)->pm->run_groups([fn, user](uint32_t g, const float* c, uint32_t n, float* v, float* pi) {
      if (!fn) throw std::runtime_error("NN evaluator needed but none supplied");
      fn(g, c, n, v, pi, user);
    });
    return 0;
  } catch (...) { return -1; }
}
uint32_t orc_pm_num_groups(void* h) { return static_cast<PmBox*>(h)->pm->num_model_groups(); }
uint32_t orc_pm_num_perms(void* h) { return static_cast<PmBox*>(h)->pm->num_seat_perms(); }
uint32_t orc_pm_perm_scores(void* h, uint32_t perm, float* out) {
  auto& pm = *static_cast<PmBox*>(h)->pm;
  const auto& sc = pm.perm_scores(perm);
  for (size_t i = 0; i < sc.size(); ++i) out[i] = sc[i];
  return pm.perm_games_completed(perm);
}
uint32_t orc_pm_num_variants(void* h) { return static_cast<PmBox*>(h)->pm->num_tracked_variants(); }
// scores[P+1], perm_scores[perms][P+1], perm_games[perms], stats[7] (variant_avg_* getters); returns variant_games_completed
uint32_t orc_pm_variant(void* h, uint32_t v, float* scores, float* perm_scores, uint32_t* perm_games, float* stats) {
  auto& pm = *static_cast<PmBox*>(h)->pm;
  const auto& vs = pm.variant(v);
  for (size_t i = 0; i < vs.scores.size(); ++i) scores[i] = vs.scores[i];
  for (size_t p = 0; p < vs.perm_scores.size(); ++p) {
    for (size_t i = 0; i < vs.perm_scores[p].size(); ++i) perm_scores[p * vs.scores.size() + i] = vs.perm_scores[p][i];
    perm_games[p] = vs.perm_games[p];
  }
  pm.variant_stats(v, stats);
  return vs.games;
}
void orc_pm_set_time_limit(void* h, double seconds) { static_cast<PmBox*>(h)->pm->time_limit_s = seconds; }
int orc_pm_run(void* h, orc_eval_fn fn, void* user) {
  try {
    static_cast<PmBox*>(h)->pm->run([fn, user](const float* c, uint32_t n, float* v, float* pi) {
      if (!fn) throw std::runtime_error("NN evaluator needed but none supplied");
      fn(c, n, v, pi, user);
    });
    return 0;
  } catch (...) { return -1; }
}
void orc_pm_scores(void* h, float* out) {
  auto& s = static_cast<PmBox*>(h)->pm->scores();
  std::memcpy(out, s.data(), s.size() * sizeof(float));
}
void orc_pm_resign_scores(void* h, float* out) {
  auto& s = static_cast<PmBox*>(h)->pm->resign_scores();
  std::memcpy(out, s.data(), s.size() * sizeof(float));
}
uint32_t orc_pm_games_completed(void* h) { return static_cast<PmBox*>(h)->pm->games_completed(); }
// stats[0..6] = avg_game_length, avg_leaf_depth, avg_search_entropy, fast_avg_leaf_depth,
//               fast_avg_search_entropy, avg_moves_per_turn, avg_valid_moves
void orc_pm_stats(void* h, float* out) {
  auto* pm = static_cast<PmBox*>(h)->pm.get();
  out[0] = pm->avg_game_length(); out[1] = pm->avg_leaf_depth(); out[2] = pm->avg_search_entropy();
  out[3] = pm->fast_avg_leaf_depth(); out[4] = pm->fast_avg_search_entropy();
  out[5] = pm->avg_moves_per_turn(); out[6] = pm->avg_valid_moves();
}
// counters[0..4] = sims, nn_evals, cache_hits, cache_misses, hist_count
void orc_pm_counters(void* h, uint64_t* out) {
  auto* pm = static_cast<PmBox*>(h)->pm.get();
  out[0] = pm->sims(); out[1] = pm->nn_evals(); out[2] = pm->cache_hits(); out[3] = pm->cache_misses();
  out[4] = pm->hist_count();
}
uint64_t orc_pm_hist_count(void* h) { return static_cast<PmBox*>(h)->pm->hist_count(); }
void orc_pm_history(void* h, float* canonical, float* v, float* pi) {
  auto& hist = static_cast<PmBox*>(h)->pm->history();
  size_t co = 0, vo = 0, po = 0;
  for (auto& r : hist) {
    std::memcpy(canonical + co, r.canonical.data(), r.canonical.size() * 4); co += r.canonical.size();
    std::memcpy(v + vo, r.v.data(), r.v.size() * 4); vo += r.v.size();
    std::memcpy(pi + po, r.pi.data(), r.pi.size() * 4); po += r.pi.size();
  }
}
uint64_t orc_pm_move_count(void* h) { return static_cast<PmBox*>(h)->pm->moves().size(); }
// rows of 8 u32: slot, game_in_slot, move, turn, player, capped, rng state lo, hi;
// counts[num_moves] per row
void orc_pm_moves(void* h, uint32_t* rows, uint32_t* counts, uint32_t num_moves) {
  auto& mv = static_cast<PmBox*>(h)->pm->moves();
  for (size_t i = 0; i < mv.size(); ++i) {
    rows[i * 8 + 0] = mv[i].slot; rows[i * 8 + 1] = mv[i].game_in_slot; rows[i * 8 + 2] = mv[i].move;
    rows[i * 8 + 3] = mv[i].turn; rows[i * 8 + 4] = mv[i].player; rows[i * 8 + 5] = mv[i].capped;
    rows[i * 8 + 6] = static_cast<uint32_t>(mv[i].rng_state); rows[i * 8 + 7] = static_cast<uint32_t>(mv[i].rng_state >> 32);
    if (counts) std::memcpy(counts + i * num_moves, mv[i].counts.data(), num_moves * sizeof(uint32_t));
  }
}

// ---------------------------------------------------------------- S3-FIFO
void* orc_cache_new(uint32_t max_size, uint32_t shards, uint32_t ghost, uint32_t np, uint32_t nv) {
  return new ShardedS3Fifo(max_size, shards, ghost, np, nv);
}
void orc_cache_free(void* c) { delete static_cast<ShardedS3Fifo*>(c); }
int orc_cache_find(void* c, uint64_t h, float* p, float* v) { return static_cast<ShardedS3Fifo*>(c)->find(h, p, v); }
void orc_cache_insert(void* c, uint64_t h, const float* p, const float* v) { static_cast<ShardedS3Fifo*>(c)->insert(h, p, v); }
// out[0..5] = hits, misses, evictions, reinserts, size, max_size
void orc_cache_stats(void* c, uint64_t* out) {
  auto* s = static_cast<ShardedS3Fifo*>(c);
  out[0] = s->hits(); out[1] = s->misses(); out[2] = s->evictions(); out[3] = s->reinserts();
  out[4] = s->size(); out[5] = s->max_size();
}

// ---------------------------------------------------------------- symmetries
// kind 0: Connect4 {base, mirror}; 1: tafl eightSym; 2: tafl mirrorWidth only; 3: tafl rot90Clockwise only;
// 4: StarGambitUnified {base, NW-axis mirror}.
// canon [C][H][W], pi [num_moves], v [nv]; outputs are [nsym][...]; returns nsym.
uint32_t orc_symmetries(int kind, int channels, int height, int width, uint32_t num_moves, uint32_t nv,
                        const float* canon, const float* v, const float* pi,
                        float* out_canon, float* out_v, float* out_pi) {
  Sample b;
  b.channels = channels; b.height = height; b.width = width;
  b.canonical.assign(canon, canon + size_t(channels) * height * width);
  b.v.assign(v, v + nv);
  b.pi.assign(pi, pi + num_moves);
  std::vector<Sample> syms;
  switch (kind) {
    case 0: syms = connect4_symmetries(b); break;
    case 1: syms = eight_sym(b); break;
    case 2: syms = {mirror_width(b)}; break;
    case 3: syms = {rot90_clockwise(b)}; break;
    case 4: syms = stargambit_unified_symmetries(b); break;
    default: return 0;
  }
  for (size_t s = 0; s < syms.size(); ++s) {
    std::memcpy(out_canon + s * b.canonical.size(), syms[s].canonical.data(), b.canonical.size() * sizeof(float));
    std::memcpy(out_v + s * nv, syms[s].v.data(), nv * sizeof(float));
    std::memcpy(out_pi + s * num_moves, syms[s].pi.data(), size_t(num_moves) * sizeof(float));
  }
  return uint32_t(syms.size());
}

}  // extern "C"
