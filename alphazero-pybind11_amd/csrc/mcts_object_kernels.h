// The reference's stand-alone `MCTS` class (py_wrapper.cc:192-220, mcts.h:50-200) on the device: one tree driven
// call by call (find_leaf / process_result / update_root / queries), built from the same SlotCtx member functions
// the PlayManager round kernel uses, on a one-slot engine (seat 0's tree).  Connect4 (lane-group engine).
#pragma once
#include "engine_kernels.h"

namespace azmi {

// WU-UCT batched API state (mcts.cc:752-851): Node::n_in_flight per arena node + the in_flight_ list.  Every kernel of the
// object that can create nodes clears the marks of the nodes it created, so the array always describes live nodes.
struct WuArrays {
  uint32_t* nif;        // [trees * cap] Node::n_in_flight
  uint32_t* ifl_path;   // [ifl_cap][max_depth] InFlightLeaf::path
  uint32_t* ifl_plen;   // [ifl_cap]
  uint32_t* ifl_cur;    // [ifl_cap] InFlightLeaf::leaf
};

// the root GameState of a call = start position (optional serialized state) + move list
template <class GM>
__device__ __forceinline__ bool mcts_replay_state(const uint8_t* init, const int32_t* moves, uint32_t len, typename GM::State& s) {
  s = init ? GM::from_bytes(init) : GM::initial();
  for (uint32_t i = 0; i < len; ++i) {
    const int32_t mv = moves[i];
    if (mv < 0) break;
    if (mv >= GM::M || !((GM::valid_mask(s) >> mv) & 1u) || !GM::play(s, static_cast<uint32_t>(mv))) return false;
  }
  return true;
}

// MCTS::find_leaf(gs): out_moves[0 .. *out_len) = the moves from gs to the leaf (the leaf GameState is gs + those)
template <class GM>
__global__ void k_mcts_find_leaf(EngineParams ep, EngineArrays ar, uint32_t* nif, const uint8_t* init, const int32_t* moves, uint32_t len,
                                 int32_t* out_moves, uint32_t* out_len, int32_t* status) {
  constexpr int G = GM::GROUP;
  const uint32_t lane = threadIdx.x;
  if (lane >= static_cast<uint32_t>(G)) return;
  SlotCtx<GM> c(ep, ar, 0, lane);
  c.load();
  if (ar.sstate[0] == kSlotFresh) c.start_game();   // first call: empty tree (root = node 0, arena bump = 1)
  typename GM::State st;
  if (!mcts_replay_state<GM>(init, moves, len, st)) { if (lane == 0) *status = -1; return; }
  c.gs = st;
  typename GM::State leaf;
  uint32_t term = 0;
  const uint32_t bump0 = c.t_bump[0];
  const bool ok = c.find_leaf(0, leaf, term);
  for (uint32_t i = bump0 + lane; i < c.t_bump[0]; i += G) nif[c.tree_base(0) + i] = 0;
  if (lane == 0) {
    const size_t tb = c.tree_base(0);
    const uint32_t* path = ar.path;
    for (uint32_t i = 0; i < c.plen; ++i) {
      const uint32_t node = (i + 1 < c.plen) ? path[i + 1] : c.cur;
      out_moves[i] = static_cast<int32_t>(meta_mv(ar.nodes[tb + node].meta));
    }
    *out_len = c.plen;
    *status = ok ? 0 : -2;
  }
  c.store(kSlotWaitEval);
}

// MCTS::process_result(gs, value, pi, root_noise_enabled): value / pi are row 0 of ar.v / ar.pi; value_out receives the
// vector the reference leaves in its by-reference `value` (the cached terminal scores for a terminal leaf)
template <class GM>
__global__ void k_mcts_process_result(EngineParams ep, EngineArrays ar, uint32_t root_noise, float* value_out) {
  constexpr int G = GM::GROUP;
  constexpr int P = GM::P;
  const uint32_t lane = threadIdx.x;
  if (lane >= static_cast<uint32_t>(G)) return;
  SlotCtx<GM> c(ep, ar, 0, lane);
  c.load();
  if (ar.sstate[0] == kSlotFresh) c.start_game();   // first call: empty tree (root = node 0, arena bump = 1)
  const uint32_t term = meta_term(ar.nodes[c.tree_base(0) + c.cur].meta);
  c.process_result(0, true, root_noise != 0);
  if (lane == 0)
    for (int i = 0; i <= P; ++i) value_out[i] = term ? ((static_cast<int>(term) - 1 == i) ? 1.0f : 0.0f) : ar.v[i];
  c.store(kSlotWaitEval);
}

template <class GM>
__global__ void k_mcts_update_root(EngineParams ep, EngineArrays ar, uint32_t* nif, const uint8_t* init, const int32_t* moves, uint32_t len,
                                   uint32_t move, int32_t* status) {
  constexpr int G = GM::GROUP;
  const uint32_t lane = threadIdx.x;
  if (lane >= static_cast<uint32_t>(G)) return;
  SlotCtx<GM> c(ep, ar, 0, lane);
  c.load();
  if (ar.sstate[0] == kSlotFresh) c.start_game();   // first call: empty tree (root = node 0, arena bump = 1)
  typename GM::State st;
  if (!mcts_replay_state<GM>(init, moves, len, st)) { if (lane == 0) *status = -1; return; }
  c.gs = st;
  const uint32_t bump0 = c.t_bump[0];
  const bool ok = c.update_root(0, move);
  for (uint32_t i = bump0 + lane; i < c.t_bump[0]; i += G) nif[c.tree_base(0) + i] = 0;
  if (lane == 0) *status = ok ? 0 : -3;   // -3: "ahh, what is this move"
  c.store(kSlotWaitEval);
}

// ---- WU-UCT batched API (mcts.cc:752-851) kernels ----------------------------------------------------------------
template <class GM>
__global__ void k_mcts_find_leaf_batched(EngineParams ep, EngineArrays ar, WuArrays wu, uint32_t index, const uint8_t* init,
                                         const int32_t* moves, uint32_t len, int32_t* out_moves, uint32_t* out_len, int32_t* status) {
  constexpr int G = GM::GROUP;
  const uint32_t lane = threadIdx.x;
  if (lane >= static_cast<uint32_t>(G)) return;
  SlotCtx<GM> c(ep, ar, 0, lane);
  c.load();
  if (ar.sstate[0] == kSlotFresh) c.start_game();
  typename GM::State st;
  if (!mcts_replay_state<GM>(init, moves, len, st)) { if (lane == 0) *status = -1; return; }
  c.gs = st;
  typename GM::State leaf;
  uint32_t term = 0;
  const bool ok = c.find_leaf_wu(0, leaf, term, wu.nif + c.tree_base(0));
  if (lane == 0) {
    const size_t tb = c.tree_base(0);
    uint32_t* rec = wu.ifl_path + static_cast<size_t>(index) * ep.max_depth;
    for (uint32_t i = 0; i < c.plen; ++i) {
      rec[i] = ar.path[i];
      const uint32_t node = (i + 1 < c.plen) ? ar.path[i + 1] : c.cur;
      out_moves[i] = static_cast<int32_t>(meta_mv(ar.nodes[tb + node].meta));
    }
    wu.ifl_plen[index] = c.plen; wu.ifl_cur[index] = c.cur;
    *out_len = c.plen;
    *status = ok ? 0 : -2;
  }
  c.store(kSlotWaitEval);
}

template <class GM>
__global__ void k_mcts_process_result_batched(EngineParams ep, EngineArrays ar, WuArrays wu, uint32_t index, uint32_t root_noise, float* value_out) {
  constexpr int G = GM::GROUP;
  constexpr int P = GM::P;
  const uint32_t lane = threadIdx.x;
  if (lane >= static_cast<uint32_t>(G)) return;
  SlotCtx<GM> c(ep, ar, 0, lane);
  c.load();
  uint32_t* nif = wu.nif + c.tree_base(0);
  c.cur = wu.ifl_cur[index]; c.plen = wu.ifl_plen[index];
  if (lane == 0) {
    const uint32_t* rec = wu.ifl_path + static_cast<size_t>(index) * ep.max_depth;
    --nif[c.cur];
    for (uint32_t i = 0; i < c.plen; ++i) { ar.path[i] = rec[i]; --nif[rec[i]]; }
  }
  c.sync_lanes();
  const uint32_t term = meta_term(ar.nodes[c.tree_base(0) + c.cur].meta);
  c.process_result(0, true, root_noise != 0);
  if (lane == 0)
    for (int i = 0; i <= P; ++i) value_out[i] = term ? ((static_cast<int>(term) - 1 == i) ? 1.0f : 0.0f) : ar.v[i];
  c.store(kSlotWaitEval);
}

enum MctsQuery : uint32_t {
  kQCounts = 0, kQProbs = 1, kQProbsPruned = 2, kQRootValue = 3, kQRootQ = 4, kQScalars = 5, kQGumbelPolicy = 6,
  kQGumbelFinal = 7, kQAddRootNoise = 8, kQApplyRootTemp = 9, kQPickMove = 10, kQPrincipalVariation = 11, kQSetGumbelSims = 12, kQRootChildren = 13
};

// every read-out / small mutation of the root: dense [M] vectors go to out_f / out_u (lane m writes entry m)
template <class GM>
__global__ void k_mcts_query(EngineParams ep, EngineArrays ar, uint32_t kind, float temp, uint32_t arg, float* out_f, uint32_t* out_u) {
  constexpr int G = GM::GROUP;
  constexpr int P = GM::P;
  constexpr int M = GM::M;
  const uint32_t lane = threadIdx.x;
  if (lane >= static_cast<uint32_t>(G)) return;
  SlotCtx<GM> c(ep, ar, 0, lane);
  c.load();
  if (ar.sstate[0] == kSlotFresh) c.start_game();   // first call: empty tree (root = node 0, arena bump = 1)
  c.sync_lanes();
  const size_t tb = c.tree_base(0);
  const uint32_t root = c.t_root[0];
  const uint64_t rmeta = ar.nodes[tb + root].meta;
  const uint32_t k = meta_nch(rmeta), c0 = meta_ch0(rmeta), root_n = ar.nodes[tb + root].n;
  const size_t ci = tb + c0 + lane;
  uint32_t n_l = 0, mv_l = 0; float q_l = 0, p_l = 0, d_l = 0;
  if (lane < k) { n_l = ar.nodes[ci].n; q_l = ar.nodes[ci].q; p_l = ar.nodes[ci].pr; d_l = ar.nodes[ci].d; mv_l = meta_mv(ar.nodes[ci].meta); }
  const uint32_t cnt_m = c.template scatter_by_move<uint32_t>(k, mv_l, n_l);
  const float pol_m = c.template scatter_by_move<float>(k, mv_l, p_l);
  const bool in = lane < static_cast<uint32_t>(M);
  switch (kind) {
    case kQCounts: if (in) out_u[lane] = cnt_m; break;                                                  // mcts.cc:557-564
    case kQProbs: { const float p = c.probs(temp, cnt_m, pol_m); if (in) out_f[lane] = p; break; }        // mcts.cc:575-618
    case kQProbsPruned: { const float p = c.probs_pruned(temp, root_n, k, mv_l, n_l, q_l, p_l, cnt_m, pol_m); if (in) out_f[lane] = p; break; }
    case kQRootValue: {                                                                                  // mcts.h:78-100
      float q = 0, d = 0; bool found = false;
      for (uint32_t i = 0; i < k; ++i) {
        const uint32_t ni = c.bcast(n_l, i); const float qi = c.bcast(q_l, i), di = c.bcast(d_l, i);
        if (ni > 0 && qi > q) { q = qi; d = di; found = true; }
      }
      if (!found && root_n > 0) { q = ar.nodes[tb + root].v; d = ar.nodes[tb + root].d; }
      const float w = q - d / static_cast<int32_t>(P);
      const float l = static_cast<float>(1.0 - static_cast<double>(w) - static_cast<double>(d));
      if (lane == 0) { out_f[0] = w; out_f[1] = l; out_f[2] = d; }
      break;
    }
    case kQRootQ: { const float qm = c.template scatter_by_move<float>(k, mv_l, q_l); if (in) out_f[lane] = qm; break; }   // mcts.cc:566-573
    case kQScalars: {   // depth(), root_n(), avg_leaf_depth(), normalized_root_entropy(), number of root children
      const uint32_t dep = c.t_depth[0];
      float ent = 0.0f;
      const float kf = static_cast<float>(k);
      if (!(kf <= 1 || root_n <= 1)) {
        const float log_k = az_logf(kf), total_n = static_cast<float>(root_n);
        float term_l = 0.0f;
        if (lane < k && n_l > 0) { const float p = static_cast<float>(n_l) / total_n; term_l = p * az_logf(p); }
        float e = 0.0f;
        for (uint32_t i = 0; i < k; ++i) { const uint32_t ni = c.bcast(n_l, i); const float ti = c.bcast(term_l, i); if (ni > 0) e -= ti; }
        ent = e / log_k;
      }
      if (lane == 0) {
        out_u[0] = dep; out_u[1] = root_n; out_u[2] = k;
        out_f[0] = dep == 0 ? 0.0f : static_cast<float>(c.t_tld[0]) / static_cast<float>(dep);
        out_f[1] = ent;
      }
      break;
    }
    case kQGumbelPolicy: { const float p = c.gumbel_improved_policy(0, k, mv_l, n_l, q_l, p_l, ar.nodes[tb + root].v); if (in) out_f[lane] = p; break; }
    case kQGumbelFinal: { const uint32_t a = c.gumbel_final_action(0, k, mv_l, n_l, q_l, p_l, cnt_m, pol_m); if (lane == 0) out_u[0] = a; break; }
    case kQAddRootNoise: {   // MCTS::add_root_noise on the current root priors
      if (k > 0) { const float p = c.add_root_noise(k, p_l, c.seat_eps(0)); if (lane < k) ar.nodes[ci].pr = p; }
      break;
    }
    case kQApplyRootTemp: {  // MCTS::apply_root_policy_temp, mcts.cc:448-460
      const float rt = c.seat_root_temp(0);
      if (rt != 1.0f && k > 0) {
        float p = lane < k ? az_powf(p_l, 1.0f / rt) : 0.0f;
        const float sum = c.seqsum(p, k);
        if (sum > 0.0f) p = p / sum;
        if (lane < k) ar.nodes[ci].pr = p;
      }
      break;
    }
    case kQPickMove: { const float p = in ? out_f[lane] : 0.0f; const uint32_t m = c.pick_move(p); if (lane == 0) out_u[0] = m; break; }   // mcts.cc:717-735
    case kQPrincipalVariation: {   // mcts.cc:676-715: most-visited child per ply (root: the Gumbel final action when Gumbel is on)
      uint32_t node = root, len = 0;
      for (uint32_t ply = 0; ply < arg; ++ply) {
        const uint64_t m = ar.nodes[tb + node].meta;
        const uint32_t kk = meta_nch(m), cc0 = meta_ch0(m);
        if (kk == 0) break;
        uint32_t nn = 0, mm = 0;
        if (lane < kk) { nn = ar.nodes[tb + cc0 + lane].n; mm = meta_mv(ar.nodes[tb + cc0 + lane].meta); }
        uint32_t best = 0xFFFFu;
        if (ply == 0 && c.seat_gumbel(0)) {
          const uint32_t a = c.gumbel_final_action(0, k, mv_l, n_l, q_l, p_l, cnt_m, pol_m);
          for (uint32_t i = 0; i < kk; ++i) if (c.bcast(mm, i) == a) { best = i; break; }
        }
        if (best == 0xFFFFu) {
          uint32_t best_n = 0;
          for (uint32_t i = 0; i < kk; ++i) { const uint32_t ni = c.bcast(nn, i); if (ni > best_n) { best_n = ni; best = i; } }
        }
        if (best == 0xFFFFu || c.bcast(nn, best) == 0) break;
        const uint32_t best_mv = c.bcast(mm, best);
        if (lane == 0) out_u[1 + len] = best_mv;
        ++len;
        node = cc0 + best;
      }
      if (lane == 0) out_u[0] = len;
      break;
    }
    case kQSetGumbelSims: if (ep.gumbel_on) c.set_gumbel_num_sims(0, arg); break;
    case kQRootChildren:   // test hook: root children in stored (shuffled) order + the stream position
      if (lane < k) { out_u[4 + lane] = mv_l; out_f[lane] = p_l; }
      if (lane == 0) { out_u[0] = k; out_u[1] = static_cast<uint32_t>(c.rng.state); out_u[2] = static_cast<uint32_t>(c.rng.state >> 32); }
      break;
    default: break;
  }
  c.store(kSlotWaitEval);
}

}  // namespace azmi

// =====================================================================================================
// The same object on the wide-game engine (Tafl family): one wavefront, BigSlot member functions.
// The root GameState is rebuilt from its move list with the repetition list (step_state), exactly like
// the engine's own game state.
// =====================================================================================================
#include "engine_kernels_big.h"

namespace azmi {

template <class GM>
__device__ __forceinline__ bool mcts_big_replay(BigSlot<GM>& c, const uint8_t* init, uint32_t init_stride, const int32_t* moves, uint32_t len) {
  if constexpr (is_stargambit<GM>::value) {
    // the root is always given as the reference's pickle image (dev_stargambit.h sg_parse_image): units, flags and the
    // position history; moves are checked against valid_moves() before they are played
    uint64_t* gl = c.game_list();
    if (init) {
      uint32_t n = 0;
      if (!sg_parse_image(init, init_stride, c.lane, c.gs, gl, n, static_cast<uint32_t>(GM::MAX_TURNS))) return false;
      c.glen = n;
    } else {
      c.gs = GM::initial(0, c.lane);
      const unsigned long long h0 = GM::position_hash(c.gs);
      if (c.lane == 0) gl[0] = h0;
      c.glen = 1;
    }
    c.sync();
    for (uint32_t i = 0; i < len; ++i) {
      const int32_t mv = moves[i];
      if (mv < 0) break;
      GM::gen_valid(c.gs, c.lane, c.sm.rules);
      if (!GM::is_valid_bit(c.sm.rules, static_cast<uint32_t>(mv))) return false;
      c.sync();
      bool base_valid = true;
      if (!c.step_state(c.gs, static_cast<uint32_t>(mv), gl, c.glen, base_valid, 0)) return false;
    }
    return true;
  } else {
  c.glen = 0;
  if (init) {       // the reference pickle image (dev_games.h TaflImage): position + repetition keys
    uint32_t n = 0;
    if (!tafl_parse_image<GM>(init, init_stride, c.gs, c.sm.glist, n, static_cast<uint32_t>(GM::MAX_TURNS))) return false;
    c.glen = n;
  } else {
    c.gs = GM::initial();
  }
  constexpr uint32_t SPAN = GM::W + GM::H;
  for (uint32_t i = 0; i < len; ++i) {
    const int32_t mv = moves[i];
    if (mv < 0) break;
    if (mv >= GM::M) return false;
    const uint32_t from = static_cast<uint32_t>(mv) / SPAN, tgt = static_cast<uint32_t>(mv) % SPAN;
    if (!GM::own_piece(c.gs, c.gs.player, from) || !((GM::slide_mask(c.gs, from) >> tgt) & 1u)) return false;
    bool base_valid = true;
    if (!c.step_state(c.gs, static_cast<uint32_t>(mv), c.sm.glist, c.glen, base_valid, 0)) return false;
  }
  return true;
  }
}

template <class GM>
__global__ __launch_bounds__(64) void k_mcts_big_find_leaf(EngineParams ep, EngineArrays ar, uint32_t* nif, const uint8_t* init, uint32_t init_stride,
                                                           const int32_t* moves, uint32_t len, int32_t* out_moves, uint32_t* out_len, int32_t* status) {
  __shared__ BigScratch<GM> sm;
  const uint32_t lane = threadIdx.x;
  BigSlot<GM> c(ep, ar, sm, 0, lane);
  c.load();
  if (ar.sstate[0] == kSlotFresh) c.start_game();
  if (!mcts_big_replay<GM>(c, init, init_stride, moves, len)) { if (lane == 0) *status = -1; return; }
  typename GM::State leaf;
  uint32_t term = 0;
  const uint32_t bump0 = c.t_bump[0];
  const bool ok = c.find_leaf(0, leaf, term);
  for (uint32_t i = bump0 + lane; i < c.t_bump[0]; i += 64) nif[c.tree_base(0) + i] = 0;
  if (lane == 0) {
    const size_t tb = c.tree_base(0);
    for (uint32_t i = 0; i < c.plen; ++i) {
      const uint32_t node = (i + 1 < c.plen) ? ar.path[i + 1] : c.cur;
      out_moves[i] = static_cast<int32_t>(meta_mv(ar.META[tb + node]));
    }
    *out_len = c.plen;
    *status = ok ? 0 : -2;
  }
  c.store(kSlotWaitEval);
}

template <class GM>
__global__ __launch_bounds__(64) void k_mcts_big_process_result(EngineParams ep, EngineArrays ar, uint32_t root_noise, float* value_out) {
  __shared__ BigScratch<GM> sm;
  constexpr int P = GM::P;
  const uint32_t lane = threadIdx.x;
  BigSlot<GM> c(ep, ar, sm, 0, lane);
  c.load();
  const uint32_t term = meta_term(ar.META[c.tree_base(0) + c.cur]);
  c.process_result(0, true, root_noise != 0);
  if (lane == 0)
    for (int i = 0; i <= P; ++i) value_out[i] = term ? ((static_cast<int>(term) - 1 == i) ? 1.0f : 0.0f) : ar.v[i];
  c.store(kSlotWaitEval);
}

template <class GM>
__global__ __launch_bounds__(64) void k_mcts_big_find_leaf_batched(EngineParams ep, EngineArrays ar, WuArrays wu, uint32_t index, const uint8_t* init,
                                                                   uint32_t init_stride, const int32_t* moves, uint32_t len, int32_t* out_moves,
                                                                   uint32_t* out_len, int32_t* status) {
  __shared__ BigScratch<GM> sm;
  const uint32_t lane = threadIdx.x;
  BigSlot<GM> c(ep, ar, sm, 0, lane);
  c.load();
  if (ar.sstate[0] == kSlotFresh) c.start_game();
  if (!mcts_big_replay<GM>(c, init, init_stride, moves, len)) { if (lane == 0) *status = -1; return; }
  typename GM::State leaf;
  uint32_t term = 0;
  const bool ok = c.find_leaf_wu(0, leaf, term, wu.nif + c.tree_base(0));
  if (lane == 0) {
    const size_t tb = c.tree_base(0);
    uint32_t* rec = wu.ifl_path + static_cast<size_t>(index) * ep.max_depth;
    for (uint32_t i = 0; i < c.plen; ++i) {
      rec[i] = ar.path[i];
      const uint32_t node = (i + 1 < c.plen) ? ar.path[i + 1] : c.cur;
      out_moves[i] = static_cast<int32_t>(meta_mv(ar.META[tb + node]));
    }
    wu.ifl_plen[index] = c.plen; wu.ifl_cur[index] = c.cur;
    *out_len = c.plen;
    *status = ok ? 0 : -2;
  }
  c.store(kSlotWaitEval);
}

template <class GM>
__global__ __launch_bounds__(64) void k_mcts_big_process_result_batched(EngineParams ep, EngineArrays ar, WuArrays wu, uint32_t index,
                                                                        uint32_t root_noise, float* value_out) {
  __shared__ BigScratch<GM> sm;
  constexpr int P = GM::P;
  const uint32_t lane = threadIdx.x;
  BigSlot<GM> c(ep, ar, sm, 0, lane);
  c.load();
  uint32_t* nif = wu.nif + c.tree_base(0);
  c.cur = wu.ifl_cur[index]; c.plen = wu.ifl_plen[index];
  if (lane == 0) {
    const uint32_t* rec = wu.ifl_path + static_cast<size_t>(index) * ep.max_depth;
    --nif[c.cur];
    for (uint32_t i = 0; i < c.plen; ++i) { ar.path[i] = rec[i]; --nif[rec[i]]; }
  }
  c.sync();
  const uint32_t term = meta_term(ar.META[c.tree_base(0) + c.cur]);
  c.process_result(0, true, root_noise != 0);
  if (lane == 0)
    for (int i = 0; i <= P; ++i) value_out[i] = term ? ((static_cast<int>(term) - 1 == i) ? 1.0f : 0.0f) : ar.v[i];
  c.store(kSlotWaitEval);
}

template <class GM>
__global__ __launch_bounds__(64) void k_mcts_big_update_root(EngineParams ep, EngineArrays ar, uint32_t* nif, const uint8_t* init, uint32_t init_stride,
                                                             const int32_t* moves, uint32_t len, uint32_t move, int32_t* status) {
  __shared__ BigScratch<GM> sm;
  const uint32_t lane = threadIdx.x;
  BigSlot<GM> c(ep, ar, sm, 0, lane);
  c.load();
  if (ar.sstate[0] == kSlotFresh) c.start_game();
  if (!mcts_big_replay<GM>(c, init, init_stride, moves, len)) { if (lane == 0) *status = -1; return; }
  const uint32_t bump0 = c.t_bump[0];
  const bool ok = c.update_root(0, move);
  for (uint32_t i = bump0 + lane; i < c.t_bump[0]; i += 64) nif[c.tree_base(0) + i] = 0;
  // a long-lived object needs the arena compaction the engine does between rounds
  if (ok && ep.half_nodes && lane == 0) {
    const uint32_t b = c.t_bump[0];
    if (b - ((b - 1) / ep.half_nodes) * ep.half_nodes > ep.compact_above) ar.compact_flag[0] = 1;
  }
  if (lane == 0) *status = ok ? 0 : -3;
  c.store(kSlotFresh + 1);   // kSlotWaitEval, but with no pending simulation to re-point: plen is reset below
  if (lane == 0) ar.plen[0] = 0;
}

template <class GM>
__global__ __launch_bounds__(64) void k_mcts_big_query(EngineParams ep, EngineArrays ar, uint32_t kind, float temp, uint32_t arg, float* out_f, uint32_t* out_u) {
  __shared__ BigScratch<GM> sm;
  constexpr int P = GM::P;
  constexpr uint32_t M = GM::M;
  const uint32_t lane = threadIdx.x;
  BigSlot<GM> c(ep, ar, sm, 0, lane);
  c.load();
  if (ar.sstate[0] == kSlotFresh) c.start_game();
  const size_t tb = c.tree_base(0);
  const uint32_t root = c.t_root[0];
  const uint64_t rmeta = ar.META[tb + root];
  const uint32_t k = meta_nch(rmeta), c0 = meta_ch0(rmeta), root_n = ar.N[tb + root];
  c.stage_root(tb, c0, k);
  auto dense_out = [&]() { for (uint32_t m = lane; m < M; m += 64) out_f[m] = sm.dense[m]; };
  switch (kind) {
    case kQCounts:
      for (uint32_t m = lane; m < M; m += 64) out_u[m] = 0;
      c.sync();
      for (uint32_t i = lane; i < k; i += 64) out_u[sm.moves[i]] = sm.n[i];
      break;
    case kQProbs: c.probs(temp, k); dense_out(); break;
    case kQProbsPruned: c.probs_pruned(temp, root_n, k); dense_out(); break;
    case kQRootValue: {
      float q = 0, d = 0; bool found = false;
      for (uint32_t i = 0; i < k; ++i) { const float qi = sm.f1[i]; if (sm.n[i] > 0 && qi > q) { q = qi; d = ar.D[tb + c0 + i]; found = true; } }
      if (!found && root_n > 0) { q = ar.V[tb + root]; d = ar.D[tb + root]; }
      const float w = q - d / static_cast<int32_t>(P);
      const float l = static_cast<float>(1.0 - static_cast<double>(w) - static_cast<double>(d));
      if (lane == 0) { out_f[0] = w; out_f[1] = l; out_f[2] = d; }
      break;
    }
    case kQRootQ:
      for (uint32_t m = lane; m < M; m += 64) out_f[m] = 0.0f;
      c.sync();
      for (uint32_t i = lane; i < k; i += 64) out_f[sm.moves[i]] = sm.f1[i];
      break;
    case kQScalars: {
      const uint32_t dep = c.t_depth[0];
      float ent = 0.0f;
      const float kf = static_cast<float>(k);
      if (!(kf <= 1 || root_n <= 1)) {
        const float log_k = az_logf(kf), total_n = static_cast<float>(root_n);
        for (uint32_t i = lane; i < k; i += 64) {
          float t = 0.0f;
          if (sm.n[i] > 0) { const float p = static_cast<float>(sm.n[i]) / total_n; t = p * az_logf(p); }
          sm.f0[i] = t;
        }
        c.sync();
        float e = 0.0f;
        for (uint32_t i = 0; i < k; ++i) if (sm.n[i] > 0) e -= sm.f0[i];
        ent = e / log_k;
      }
      if (lane == 0) {
        out_u[0] = dep; out_u[1] = root_n; out_u[2] = k;
        out_f[0] = dep == 0 ? 0.0f : static_cast<float>(c.t_tld[0]) / static_cast<float>(dep);
        out_f[1] = ent;
      }
      break;
    }
    case kQGumbelPolicy: c.gumbel_improved_policy(0, k, ar.V[tb + root]); dense_out(); break;
    case kQGumbelFinal: { const uint32_t a = c.gumbel_final_action(0, tb, c0, k); if (lane == 0) out_u[0] = a; break; }
    case kQAddRootNoise:
      if (k > 0) {
        for (uint32_t i = lane; i < k; i += 64) sm.f0[i] = sm.f2[i];
        c.sync();
        c.add_root_noise(k, c.seat_eps(0));
        for (uint32_t i = lane; i < k; i += 64) ar.Pr[tb + c0 + i] = sm.f0[i];
      }
      break;
    case kQApplyRootTemp: c.reapply_root_prior(0, false); break;
    case kQPickMove: {
      for (uint32_t m = lane; m < M; m += 64) sm.dense[m] = out_f[m];
      c.sync();
      const uint32_t mv = c.pick_move();
      if (lane == 0) out_u[0] = mv;
      break;
    }
    case kQPrincipalVariation: {
      uint32_t node = root, len = 0;
      for (uint32_t ply = 0; ply < arg; ++ply) {
        const uint64_t m = ar.META[tb + node];
        const uint32_t kk = meta_nch(m), cc0 = meta_ch0(m);
        if (kk == 0) break;
        uint32_t best = 0xFFFFFFFFu;
        if (ply == 0 && c.seat_gumbel(0)) {
          const uint32_t a = c.gumbel_final_action(0, tb, c0, k);
          uint32_t hit = 0xFFFFFFFFu;
          for (uint32_t i = lane; i < kk; i += 64) if (meta_mv(ar.META[tb + cc0 + i]) == a) hit = i;
          for (int off = 32; off > 0; off >>= 1) hit = min(hit, __shfl_xor(hit, off, 64));
          best = hit;
        }
        if (best == 0xFFFFFFFFu) {   // first child with the strictly largest visit count
          uint32_t bn = 0, bi = 0xFFFFFFFFu;
          for (uint32_t i = lane; i < kk; i += 64) { const uint32_t ni = ar.N[tb + cc0 + i]; if (ni > bn) { bn = ni; bi = i; } }
          for (int off = 1; off < 64; off <<= 1) {
            const uint32_t on = __shfl_xor(bn, off, 64), oi = __shfl_xor(bi, off, 64);
            if (on > bn || (on == bn && oi < bi)) { bn = on; bi = oi; }
          }
          best = bn == 0 ? 0xFFFFFFFFu : bi;
        }
        if (best == 0xFFFFFFFFu || ar.N[tb + cc0 + best] == 0) break;
        if (lane == 0) out_u[1 + len] = meta_mv(ar.META[tb + cc0 + best]);
        ++len;
        node = cc0 + best;
      }
      if (lane == 0) out_u[0] = len;
      break;
    }
    case kQSetGumbelSims: if (ep.gumbel_on) c.set_gumbel_num_sims(0, arg); break;
    case kQRootChildren:
      for (uint32_t i = lane; i < k && i < 60; i += 64) { out_u[4 + i] = sm.moves[i]; out_f[i] = sm.f2[i]; }
      if (lane == 0) { out_u[0] = k; out_u[1] = static_cast<uint32_t>(c.rng.state); out_u[2] = static_cast<uint32_t>(c.rng.state >> 32); }
      break;
    default: break;
  }
  c.sync();
  c.store(kSlotWaitEval);
}

}  // namespace azmi
