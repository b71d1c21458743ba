// Tree + game-step kernels of the self-play engine (gfx950, wave64).
//
// One lane-group of GM::GROUP lanes owns one game slot; 64/GROUP slots share a
// wavefront.  Per-child work (the N/Q/P walk of PUCT select, prior
// normalisation, expansion, backup levels) is spread over the group's lanes so
// that consecutive lanes touch consecutive 32-byte node records (NodeRec,
// engine_types.h: coalesced reads, one base pointer); per-slot scalars
// (bitboards, pcg32 state) are computed redundantly by every lane of the group,
// which keeps them group-uniform without any cross-lane traffic.  Order-sensitive
// float reductions (the reference sums priors in child order) on the hot path are
// DPP row_shr sweeps (seqsum8), elsewhere ds_bpermute sweeps; arg-max reduces a
// (score, index) key with DPP quad / half-row permutes so the first index wins
// ties exactly like the reference's strict `>` scan.
//
// Reference behaviour restated here (file:line under /root/reference/src):
//   Node::add_children        mcts.cc:93-101      -> expand_node
//   Node::set_policy_normalized mcts.cc:109-121   -> process_result (priors)
//   Node::uct / best_child    mcts.cc:123-149     -> select_child
//   MCTS::update_root         mcts.cc:151-173     -> update_root
//   MCTS::add_root_noise      mcts.cc:403-446     -> add_root_noise
//   MCTS::apply_root_policy_temp mcts.cc:448-460  -> apply_root_policy_temp
//   MCTS::find_leaf           mcts.cc:462-498     -> find_leaf
//   MCTS::process_result      mcts.cc:500-555     -> process_result
//   MCTS::probs/probs_pruned/pick_move/root_value/normalized_root_entropy
//                             mcts.cc:575-674,717-750, mcts.h:78-100
//   PlayManager::play         play_manager.cc:258-600 -> k_round (one worker-loop
//                             iteration per slot per round) + k_assign
//   dumb_eval                 game_state.h:160-173 -> synthesised in process_result
// Build with -ffp-contract=off: the float expressions keep the reference's
// operand order and must not be fused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev_games.h"
#include "dev_rng.h"
#include "engine_types.h"

namespace azmi {

constexpr float kNoiseAlphaRatio = 10.83f;  // mcts.cc:14

// per-seat parameter without a dynamically indexed load: a runtime index into a by-value kernel
// argument forces the whole argument block into scratch memory (every ar.X then costs a load)
__device__ __forceinline__ uint32_t seat_param(const uint32_t (&a)[4], uint32_t seat) {
  return seat == 0 ? a[0] : seat == 1 ? a[1] : seat == 2 ? a[2] : a[3];
}

// seq_halving_phase_plan, mcts.cc:28-66: returns the number of phases; (num_c, v_per) of phase `want`
__device__ __forceinline__ uint32_t gum_plan(uint32_t m, uint32_t n, uint32_t want, uint32_t& num_c_out, uint32_t& v_per_out) {
  num_c_out = 0; v_per_out = 0;
  if (m <= 1) { if (want == 0) { num_c_out = 1; v_per_out = n; } return 1; }
  uint32_t log2m = 0;
  for (uint32_t v = m - 1; v > 0; v >>= 1) ++log2m;
  if (log2m == 0) log2m = 1;
  const uint32_t d0 = n / (log2m * m);
  const uint32_t base_v = d0 > 1u ? d0 : 1u;
  uint32_t sims_used = 0, num_c = m, count = 0;
  for (uint32_t phase_idx = 0; phase_idx < log2m; ++phase_idx) {
    if (sims_used >= n) break;
    const uint32_t remaining = n - sims_used;
    const bool is_final = (phase_idx == log2m - 1);
    const uint32_t fin = remaining / num_c;
    uint32_t v_per = is_final ? (fin > 1u ? fin : 1u) : base_v * (1u << phase_idx);
    if (num_c * v_per > remaining) {
      v_per = remaining / num_c;
      if (v_per == 0) { num_c = remaining; v_per = 1; }
    }
    if (count == want) { num_c_out = num_c; v_per_out = v_per; }
    ++count;
    sims_used += num_c * v_per;
    num_c = (num_c / 2) > 1u ? (num_c / 2) : 1u;
  }
  return count;
}

// lane image of a simulation's path (see PendRec, engine_types.h): level i in lane i, paths of <= 8 levels
struct PathRegs {
  uint32_t node = 0, n = 0, pp = 0, mv = 0;
  float q = 0.0f, d = 0.0f, v = 0.0f;
  uint64_t leaf_meta = 0;
  uint32_t ok = 0;       // the path had at most 8 levels and is complete here
};

#define AZMI_SEL(arr, seat) ((seat) == 0 ? arr[0] : arr[P > 1 ? 1 : 0])
template <class GM>
struct SlotCtx {
  static constexpr int G = GM::GROUP;
  static constexpr int P = GM::P;
  static constexpr int M = GM::M;
  // process_result's first-visit value, make_move's resign entries and the per-seat AZMI_SEL selects are written for two
  // players (every device game); an N-player game needs val[player] there
  static_assert(GM::P == 2, "the lane-group engine is written for two-player games");
  static_assert(GM::MAXK <= G, "single-chunk child handling needs MAXK <= GROUP");
  static_assert(GM::M <= G, "lane-dense move vectors need M <= GROUP");

  const EngineParams& ep;
  const EngineArrays& ar;
  uint32_t slot, lane;
  Pcg32 rng, coin;
  typename GM::State gs;
  uint8_t flags;
  // per-tree scalars, held identically by every lane of the group for the whole round
  // (MCTS::root_, arena bump pointer, MCTS::depth_, MCTS::total_leaf_depth_)
  uint32_t t_root[P], t_bump[P], t_depth[P];
  uint64_t t_tld[P];
  uint32_t cur, plen;  // MCTS::current_, MCTS::path_.size() of the pending simulation
  uint32_t ph_rows;    // GameData::partial_history.size()
  // per-seat search settings of this game's seat permutation (seat_visits_, seat_cap_visits_, seat_epsilon_,
  // seat_mcts_root_temp_, seat_root_fpu_zero_, seat_perm -> model group / eval type; play_manager.cc:57-113)
  uint32_t perm, sv_w0[P], sv_w1[P];
  float sv_eps[P], sv_rt[P];
  __device__ __forceinline__ uint32_t seat_visits(uint32_t seat) const { return AZMI_SEL(sv_w0, seat); }
  __device__ __forceinline__ uint32_t seat_cap_visits(uint32_t seat) const { return AZMI_SEL(sv_w1, seat) & 0xFFFFFFu; }
  __device__ __forceinline__ bool seat_fpu_zero(uint32_t seat) const { return (AZMI_SEL(sv_w1, seat) >> 24) & 1u; }
  __device__ __forceinline__ bool seat_eval_random(uint32_t seat) const { return (AZMI_SEL(sv_w1, seat) >> 25) & 1u; }
  __device__ __forceinline__ uint32_t seat_group(uint32_t seat) const { return (AZMI_SEL(sv_w1, seat) >> 26) & 3u; }
  __device__ __forceinline__ bool seat_eval_playout(uint32_t seat) const { return (AZMI_SEL(sv_w1, seat) >> 28) & 1u; }
  __device__ __forceinline__ float seat_eps(uint32_t seat) const { return AZMI_SEL(sv_eps, seat); }
  __device__ __forceinline__ float seat_root_temp(uint32_t seat) const { return AZMI_SEL(sv_rt, seat); }
  // per-seat Gumbel / resign settings (words 4-7 of the seat record): read on demand, they are off the PUCT path
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

  __device__ __forceinline__ SlotCtx(const EngineParams& e, const EngineArrays& a, uint32_t s, uint32_t l)
      : ep(e), ar(a), slot(s), lane(l) {}

  // ---- group primitives ---------------------------------------------------------
  template <class T>
  __device__ __forceinline__ T bcast(T v, int src) const { return __shfl(v, src, G); }
  // ---- DPP forms of the two reductions of the PUCT select (the descent spends about as long in cross-lane traffic as
  // in its one HBM round trip per level; a ds_bpermute costs an LDS round trip, a DPP move a few cycles) ----------
  template <int CTRL>
  __device__ __forceinline__ static float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
  }
  template <int CTRL>
  __device__ __forceinline__ static uint32_t dpp_u(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xF, 0xF, true));
  }
  // x_0 + x_1 + ... + x_7 in lane order over the 8-lane group; lanes past the data must hold +0.0f.  Lane j takes lane
  // j-1's running sum (row_shr:1) at step j, so the additions happen in exactly the sequential order.
  __device__ __forceinline__ float seqsum8(float x) const {
    static_assert(G == 8, "8-lane groups");
    const uint32_t li = lane & 7u;
    float run = x;
#pragma unroll
    for (uint32_t j = 1; j < 8; ++j) {
      const float t = dpp_f<0x111>(run);          // row_shr:1
      if (li == j) run = t + x;
    }
    return bcast(run, 7);
  }
  __device__ __forceinline__ float seqsum(float x, uint32_t n) const {  // x_0 + x_1 + ... in lane order
    float s = 0.0f;
    for (uint32_t i = 0; i < n; ++i) s += bcast(x, i);
    return s;
  }
  __device__ __forceinline__ size_t tree_base(uint32_t seat) const {
    return (static_cast<size_t>(slot) * P + seat) * ep.cap;
  }
  __device__ __forceinline__ uint32_t tree_id(uint32_t seat) const { return slot * P + seat; }
  __device__ __forceinline__ void raise(uint32_t bit) const {
    if (lane == 0) { atomicOr(&ar.ctl->overflow, bit); ar.ctl->stop = 1; }
  }
  // Lanes of a group hand node data to each other through HBM (e.g. backup writes N/Q by
  // level-lane, select reads them by child-lane).  A workgroup-scope fence makes those stores
  // visible to the other lanes and stops the compiler from forwarding a lane's own stale value.
  __device__ __forceinline__ void sync_lanes() const { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }
  __device__ __forceinline__ void trace(uint64_t tag) const {
    if (slot != ep.trace_slot || lane != 0) return;
    // tags >= 100 are phase marks of the round kernel: they carry the 100 MHz wall clock instead of the stream position.
    // The clock is read BEFORE the hook's own two loads, so a phase does not include the hook's latency (the loads still
    // delay the traced wave by about a microsecond per mark: totals over many marks overstate the untraced kernel).
    const uint64_t val = (tag & 0xFF) >= 100 ? wall_clock64() : rng.state;
    if (ar.ctl->rounds < ep.trace_after) return;
    const uint64_t n = ar.trace[0];
    if (n + 1 < ep.trace_cap) { ar.trace[2 * (n + 1)] = tag; ar.trace[2 * (n + 1) + 1] = val; ar.trace[0] = n + 1; }
  }

  // ---- state load / store ----------------------------------------------------------
  __device__ __forceinline__ void load() {
    rng.state = ar.rng[slot];
    coin.state = ar.coin[slot];
    flags = ar.flags[slot];
    const uint64_t w0 = ar.gs_words[0 * ep.S + slot], w1 = ar.gs_words[1 * ep.S + slot],
                   w2 = ar.gs_words[2 * ep.S + slot];
    gs.bb[0] = w0; gs.bb[1] = w1;
    gs.turn = static_cast<uint32_t>(w2); gs.player = static_cast<uint32_t>(w2 >> 32);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const uint32_t t = slot * P + p;
      t_root[p] = ar.root[t]; t_bump[p] = ar.bump[t]; t_depth[p] = ar.depth[t]; t_tld[p] = ar.tld[t];
    }
    cur = ar.cur[slot]; plen = ar.plen[slot]; ph_rows = ar.ph_count[slot];
    perm = ar.perm[slot];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const uint32_t* rec = ar.seat_tab + (static_cast<size_t>(perm) * P + p) * kSeatWords;
      sv_w0[p] = rec[0]; sv_w1[p] = rec[1]; sv_eps[p] = __uint_as_float(rec[2]); sv_rt[p] = __uint_as_float(rec[3]);
    }
  }
  __device__ __forceinline__ void store(uint8_t sstate) const {
    if (lane != 0) return;
    ar.rng[slot] = rng.state;
    ar.coin[slot] = coin.state;
    ar.flags[slot] = flags;
    ar.sstate[slot] = sstate;
    ar.gs_words[0 * ep.S + slot] = gs.bb[0];
    ar.gs_words[1 * ep.S + slot] = gs.bb[1];
    ar.gs_words[2 * ep.S + slot] = static_cast<uint64_t>(gs.turn) | (static_cast<uint64_t>(gs.player) << 32);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const uint32_t t = slot * P + p;
      ar.root[t] = t_root[p]; ar.bump[t] = t_bump[p]; ar.depth[t] = t_depth[p]; ar.tld[t] = t_tld[p];
    }
    ar.cur[slot] = cur; ar.plen[slot] = plen; ar.ph_count[slot] = ph_rows;
  }

  // ---- tree reset: MCTS{...} construction, play_manager.cc:602-617 --------------------
  __device__ __forceinline__ void reset_tree(uint32_t seat) {
#pragma unroll
    for (int p = 0; p < P; ++p)
      if (static_cast<uint32_t>(p) == seat) { t_root[p] = 0; t_bump[p] = 1; t_depth[p] = 0; t_tld[p] = 0; }
    const size_t tb = tree_base(seat);
    if (lane == 0) { ar.nodes[tb].n = 0; ar.nodes[tb].q = 0; ar.nodes[tb].pr = 0; ar.nodes[tb].d = 0; ar.nodes[tb].v = 0; ar.nodes[tb].meta = 0; }
    if (ep.gumbel_on) set_gumbel_num_sims(seat, 0);   // a new MCTS object: target 0, nothing initialised
  }

  // ======================= Gumbel AlphaZero (mcts.cc:24-401) =========================================
  // Per-tree record in HBM; every lane of the group computes the same scalars, lane 0 stores them.
  enum { kGumTarget = 0, kGumInit = 1, kGumNSurv = 2, kGumPhase = 3, kGumSims = 4, kGumMEff = 5, kGumRemain = 6 };
  __device__ __forceinline__ uint32_t* gum_state(uint32_t seat) const { return ar.gum_state + static_cast<size_t>(tree_id(seat)) * 8; }
  __device__ __forceinline__ float* gum_g(uint32_t seat) const { return ar.gum_g + static_cast<size_t>(tree_id(seat)) * ep.gum_stride; }
  __device__ __forceinline__ uint16_t* gum_surv(uint32_t seat) const { return ar.gum_surv + static_cast<size_t>(tree_id(seat)) * kGumMaxM; }

  __device__ __forceinline__ void reset_gumbel_state(uint32_t seat) const {  // mcts.cc:180-188
    if (lane != 0) return;
    uint32_t* st = gum_state(seat);
    st[kGumInit] = 0; st[kGumNSurv] = 0; st[kGumPhase] = 0; st[kGumSims] = 0; st[kGumMEff] = 0; st[kGumRemain] = 0;
  }
  __device__ __forceinline__ void set_gumbel_num_sims(uint32_t seat, uint32_t n) const {  // mcts.cc:175-178
    if (lane == 0) gum_state(seat)[kGumTarget] = n;
    reset_gumbel_state(seat);
  }
  // play_manager.cc:525-539 / 561-570: the search of the player to move is Gumbel only when it is a
  // full search (or fast_search_uses_gumbel)
  __device__ __forceinline__ void set_gumbel_target() const {
    if (!ep.gumbel_on) return;
    const uint32_t cp = gs.player;
    const uint32_t target = (flags & kFlagCapped) ? (ep.fast_gumbel ? seat_cap_visits(cp) : 0u) : seat_visits(cp);
    set_gumbel_num_sims(cp, target);
  }
  __device__ __forceinline__ float gumbel01() {  // extreme_value_distribution<float>{0,1}, random.tcc:2581-2590
    return 0.0f - 1.0f * az_logf(-az_logf(1.0f - canonical01(rng)));
  }
  __device__ __forceinline__ uint32_t group_max(uint32_t x) const {
    for (int off = 1; off < G; off <<= 1) x = max(x, __shfl_xor(x, off, G));
    return x;
  }
  // init_gumbel_state, mcts.cc:190-227; root children (priors p_l) in lanes [0,k)
  __device__ __forceinline__ void init_gumbel_state(uint32_t seat, uint32_t k, float p_l) {
    if (k == 0) return;
    uint32_t* st = gum_state(seat);
    const uint32_t target = st[kGumTarget], depth = AZMI_SEL(t_depth, seat);
    const uint32_t remaining = depth < target ? target - depth : 0u;
    if (remaining == 0) return;
    const uint32_t gm = seat_gumbel_m(seat);
    uint32_t m_eff = gm < k ? gm : k;
    m_eff = m_eff < remaining ? m_eff : remaining;
    m_eff = m_eff > 1u ? m_eff : 1u;
    float g_l = 0.0f;
    for (uint32_t i = 0; i < k; ++i) { const float gi = gumbel01(); if (lane == i) g_l = gi; }
    const float score_l = g_l + az_logf(p_l + 1e-20f);
    uint32_t rank_l = 0;  // position in the descending order of g + log(prior) (partial_sort, mcts.cc:214-221)
    for (uint32_t j = 0; j < k; ++j) {
      const float sj = bcast(score_l, j);
      if (sj > score_l || (sj == score_l && j < lane)) ++rank_l;
    }
    if (lane < k) {
      gum_g(seat)[lane] = g_l;
      if (rank_l < m_eff) gum_surv(seat)[rank_l] = static_cast<uint16_t>(lane);
    }
    if (lane == 0) {
      st[kGumInit] = 1; st[kGumNSurv] = m_eff; st[kGumPhase] = 0; st[kGumSims] = 0; st[kGumMEff] = m_eff; st[kGumRemain] = remaining;
    }
    sync_lanes();
  }
  // gumbel_next_root_child (mcts.cc:266-283) with gumbel_advance_phase (mcts.cc:229-264) inlined
  __device__ __forceinline__ uint32_t gumbel_next_root_child(uint32_t seat, uint32_t k, uint32_t n_l, float q_l, float p_l) {
    sync_lanes();
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
        uint32_t pos_l = 0xFFFFu;
        for (uint32_t i = 0; i < nsurv; ++i) if (surv[i] == lane) pos_l = i;
        const bool is_s = pos_l != 0xFFFFu;
        const uint32_t max_visit = group_max(is_s ? n_l : 0u);
        const float sigma_scale = seat_sigma_scale(seat, max_visit);
        const float g_l = lane < k ? gum_g(seat)[lane] : 0.0f;
        const float score_l = g_l + az_logf(p_l + 1e-20f) + sigma_scale * (n_l > 0 ? q_l : 0.0f);
        uint32_t rank_l = 0;
        for (uint32_t j = 0; j < k; ++j) {
          const float sj = bcast(score_l, j);
          const uint32_t pj = bcast(pos_l, j);
          if (pj != 0xFFFFu && (sj > score_l || (sj == score_l && pj < pos_l))) ++rank_l;
        }
        sync_lanes();
        if (is_s && rank_l < next_c) surv[rank_l] = static_cast<uint16_t>(lane);
        sync_lanes();
        nsurv = next_c;
      }
      ++phase; sims = 0;
    }
    uint32_t child = 0;
    if (nsurv != 0) { child = surv[sims % nsurv]; ++sims; }
    if (lane == 0) { st[kGumNSurv] = nsurv; st[kGumPhase] = phase; st[kGumSims] = sims; }
    return child;
  }
  // softmax(log prior + sigma * completedQ) over the children in lanes [0,k) (mcts.cc:285-373); returns this
  // lane's exp term, the sum in z_sum
  __device__ __forceinline__ float gumbel_pi_prime(uint32_t seat, uint32_t k, uint32_t n_l, float q_l, float p_l, float node_v, float& z_sum) const {
    float sum_visits = 0.0f, sum_priors_visited = 0.0f, weighted_num = 0.0f;   // compute_v_mix_from_children, mcts.cc:71-89
    for (uint32_t i = 0; i < k; ++i) {
      const uint32_t ni = bcast(n_l, i); const float qi = bcast(q_l, i), pi = bcast(p_l, i);
      sum_visits += static_cast<float>(ni);
      if (ni > 0) { sum_priors_visited += pi; weighted_num += pi * qi; }
    }
    float v_mix = node_v;
    if (!(sum_priors_visited <= 0.0f)) {
      const float weighted_q = weighted_num / sum_priors_visited;
      v_mix = (node_v + sum_visits * weighted_q) / (sum_visits + 1.0f);
    }
    const uint32_t max_visit = group_max(lane < k ? n_l : 0u);
    const float sigma_scale = seat_sigma_scale(seat, max_visit);
    float z = lane < k ? az_logf(p_l + 1e-20f) + sigma_scale * (n_l > 0 ? q_l : v_mix) : -__builtin_inff();
    float z_max = -__builtin_inff();
    for (uint32_t i = 0; i < k; ++i) { const float zi = bcast(z, i); if (zi > z_max) z_max = zi; }
    z = lane < k ? az_expf(z - z_max) : 0.0f;
    z_sum = seqsum(z, k);
    return z;
  }
  // gumbel_interior_select, mcts.cc:285-334
  __device__ __forceinline__ uint32_t gumbel_interior_select(uint32_t seat, uint32_t k, uint32_t n_l, float q_l, float p_l, float node_v) const {
    float z_sum;
    const float z = gumbel_pi_prime(seat, k, n_l, q_l, p_l, node_v, z_sum);
    uint32_t sum_visits = lane < k ? n_l : 0u;
    for (int off = 1; off < G; off <<= 1) sum_visits += __shfl_xor(sum_visits, off, G);
    const float inv = z_sum > 0 ? (1.0f / z_sum) : 0.0f;
    const float denom = 1.0f + static_cast<float>(sum_visits);
    float score = z * inv - static_cast<float>(n_l) / denom;
    if (score != score || lane >= k) score = -__builtin_inff();   // `score > best_score` never picks a NaN
    uint32_t idx = lane < k ? lane : 0xFFFFu;
    for (int off = 1; off < G; off <<= 1) {
      const float os = __shfl_xor(score, off, G);
      const uint32_t oi = __shfl_xor(idx, off, G);
      if (os > score || (os == score && oi < idx)) { score = os; idx = oi; }
    }
    return idx;
  }
  // gumbel_improved_policy, mcts.cc:336-373: dense [M] vector, lane m holds entry m
  __device__ __forceinline__ float gumbel_improved_policy(uint32_t seat, uint32_t k, uint32_t mv_l, uint32_t n_l, float q_l, float p_l, float root_v) const {
    if (k == 0) return 0.0f;
    float z_sum;
    const float z = gumbel_pi_prime(seat, k, n_l, q_l, p_l, root_v, z_sum);
    if (z_sum <= 0) return 0.0f;
    return scatter_by_move<float>(k, mv_l, z / z_sum);
  }
  // gumbel_final_action, mcts.cc:375-401; returns the move
  __device__ __forceinline__ uint32_t gumbel_final_action(uint32_t seat, uint32_t k, uint32_t mv_l, uint32_t n_l, float q_l, float p_l,
                                                          uint32_t cnt_m, float pol_m) {
    sync_lanes();
    const uint32_t* st = gum_state(seat);
    const uint32_t nsurv = st[kGumNSurv];
    if (!st[kGumInit] || nsurv == 0) return pick_move(probs(0.0f, cnt_m, pol_m));
    const uint16_t* surv = gum_surv(seat);
    const uint32_t max_visit = group_max(lane < k ? n_l : 0u);
    const float sigma_scale = seat_sigma_scale(seat, max_visit);
    const float g_l = lane < k ? gum_g(seat)[lane] : 0.0f;
    const float score_l = g_l + az_logf(p_l + 1e-20f) + sigma_scale * (n_l > 0 ? q_l : 0.0f);
    uint32_t best = surv[0];
    float best_score = -__builtin_inff();
    for (uint32_t i = 0; i < nsurv; ++i) {
      const uint32_t ci = surv[i];
      const float sc = bcast(score_l, ci);
      if (sc > best_score) { best_score = sc; best = ci; }
    }
    return bcast(mv_l, best);
  }

  // ---- Node::add_children: legal moves ascending, std::shuffle, append to the arena ----
  // Returns false when the arena is full.  `meta_keep` supplies move/player/term of `node`.
  __device__ __forceinline__ bool expand_node(uint32_t seat, uint32_t node, const typename GM::State& st, uint64_t meta_keep,
                              uint32_t& c0_out, uint32_t& k_out, uint32_t* mv_out = nullptr) {
    const size_t tb = tree_base(seat);
    const uint32_t k = GM::num_valid(st);
    // The legal moves (ascending) as 4-bit fields of one word that every lane of the group holds: std::shuffle
    // (stl_algo.h:3729-3792) then swaps fields of that word — the draws are group-uniform anyway — instead of moving
    // values between lanes (a dependent ds_bpermute chain per swap); each lane finally picks field `lane`.
    static_assert(GM::MAXK <= 8, "eight 4-bit move fields");
    uint32_t packed = 0;
    {
      uint32_t vm = GM::valid_mask(st), pos = 0;
      while (vm) { const uint32_t m = __builtin_ctz(vm); vm &= vm - 1; packed |= m << (4 * pos); ++pos; }
    }
    auto swap_fields = [&](uint32_t a, uint32_t b) {
      const uint32_t fa = (packed >> (4 * a)) & 0xFu, fb = (packed >> (4 * b)) & 0xFu;
      packed = (packed & ~((0xFu << (4 * a)) | (0xFu << (4 * b)))) | (fb << (4 * a)) | (fa << (4 * b));
    };
    if (k > 1) {
      uint32_t i = 1;
      if ((k & 1u) == 0) {
        const uint32_t j = lemire_below(rng, 2);
        swap_fields(i, j);
        ++i;
      }
      while (i != k) {
        const uint32_t swap_range = i + 1, b1 = swap_range + 1;
        const uint32_t x = lemire_below(rng, swap_range * b1);
        const uint32_t p0 = x / b1, p1 = x % b1;
        swap_fields(i, p0);
        ++i;
        swap_fields(i, p1);
        ++i;
      }
    }
    const uint32_t mv = lane < k ? (packed >> (4 * lane)) & 0xFu : 0u;
    const uint32_t c0 = AZMI_SEL(t_bump, seat);
    if (c0 + k > ep.cap) { raise(1u); return false; }
    if (lane < k) {
      const size_t ci = tb + c0 + lane;
      ar.nodes[ci].n = 0; ar.nodes[ci].q = 0.0f; ar.nodes[ci].pr = 0.0f; ar.nodes[ci].d = 0.0f; ar.nodes[ci].v = 0.0f;
      ar.nodes[ci].meta = meta_pack(0, 0, mv, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) if (static_cast<uint32_t>(p) == seat) t_bump[p] = c0 + k;
    if (lane == 0)
      ar.nodes[tb + node].meta = meta_pack(c0, k, meta_mv(meta_keep), meta_player(meta_keep), meta_term(meta_keep));
    c0_out = c0;
    k_out = k;
    if (mv_out) *mv_out = mv;
    return true;
  }

  // ---- Node::best_child ------------------------------------------------------------------
  // children live in lanes [0,k): n_l, q_l, p_l.  Returns the winning lane.
  // (`if_l` = Node::n_in_flight of the child, `n_parent` includes the parent's: only the WU-UCT batched API has them non-zero)
  __device__ __forceinline__ uint32_t select_child(uint32_t k, uint32_t n_l, float q_l, float p_l, float v_parent,
                                   uint32_t n_parent, float fpu_reduction, uint32_t if_l = 0) const {
    // seen_policy: the priors of the visited children added in child order; an unvisited child contributes +0.0f, which
    // leaves the running sum unchanged, so the masked in-order sum is the reference's
    const float seen = seqsum8((lane < k && n_l > 0) ? p_l : 0.0f);
    const float fpu_value = v_parent - fpu_reduction * sqrtf(seen);
    const float sqrt_n = sqrtf(static_cast<float>(n_parent));
    float u = (n_l == 0 ? fpu_value : q_l) + ep.cpuct * p_l * sqrt_n / static_cast<float>(n_l + if_l + 1);
    // a strict `>` scan never replaces the incumbent with a NaN and never leaves a NaN at
    // index 0: map NaN to +inf at lane 0 and -inf elsewhere, then reduce (score, index) keys: largest score, smallest
    // index among equals — an order-free reduction, done with DPP moves (xor 1, xor 2 inside quads, then the half-row
    // mirror to meet the other quad)
    if (u != u) u = (lane == 0) ? __builtin_inff() : -__builtin_inff();
    if (lane >= k) u = -__builtin_inff();
    uint32_t idx = lane < k ? lane : 0xFFFFu;
    {
      float ou = dpp_f<0xB1>(u); uint32_t oi = dpp_u<0xB1>(idx);          // quad_perm [1,0,3,2]
      if (ou > u || (ou == u && oi < idx)) { u = ou; idx = oi; }
      ou = dpp_f<0x4E>(u); oi = dpp_u<0x4E>(idx);                           // quad_perm [2,3,0,1]
      if (ou > u || (ou == u && oi < idx)) { u = ou; idx = oi; }
      ou = dpp_f<0x141>(u); oi = dpp_u<0x141>(idx);                         // row_half_mirror: lane i <-> 7 - i
      if (ou > u || (ou == u && oi < idx)) { u = ou; idx = oi; }
    }
    return idx;
  }

  // ---- MCTS::find_leaf_batched (WU-UCT), mcts.cc:752-784 ----------------------------------------
  // `nif` = Node::n_in_flight per arena node of this tree.  Walks through visited nodes AND nodes with an
  // evaluation in flight, marks every node of the path after selecting below it, expands only a node
  // nobody expanded yet.  No Gumbel branch: the reference's batched descent is plain PUCT.
  __device__ __forceinline__ bool find_leaf_wu(uint32_t seat, typename GM::State& leaf, uint32_t& term, uint32_t* nif) {
    sync_lanes();
    const size_t tb = tree_base(seat);
    const uint32_t root = AZMI_SEL(t_root, seat);
    cur = root; plen = 0;
    leaf = gs;
    uint64_t meta = ar.nodes[tb + cur].meta;
    uint32_t n = ar.nodes[tb + cur].n, nf = nif[cur];
    uint32_t* path = ar.path + static_cast<size_t>(slot) * ep.max_depth;
    while ((n > 0 || nf > 0) && meta_nch(meta) != 0 && meta_term(meta) == 0) {
      if (plen >= ep.max_depth) { raise(8u); return false; }
      if (lane == 0) path[plen] = cur;
      ++plen;
      const uint32_t k = meta_nch(meta), c0 = meta_ch0(meta);
      const size_t ci = tb + c0 + lane;
      uint32_t n_l = 0, if_l = 0; float q_l = 0.0f, p_l = 0.0f; uint64_t m_l = 0;
      if (lane < k) { n_l = ar.nodes[ci].n; q_l = ar.nodes[ci].q; p_l = ar.nodes[ci].pr; m_l = ar.nodes[ci].meta; if_l = nif[c0 + lane]; }
      const float fpu = (cur == root && seat_fpu_zero(seat)) ? 0.0f : ep.fpu_reduction;
      const uint32_t best = select_child(k, n_l, q_l, p_l, ar.nodes[tb + cur].v, n + nf, fpu, if_l);
      if (lane == 0) nif[cur] = nf + 1;
      cur = c0 + best;
      n = bcast(n_l, best);
      nf = bcast(if_l, best);
      meta = bcast(m_l, best);
      GM::play(leaf, meta_mv(meta));
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
      if (lane < k) nif[c0 + lane] = 0;
    }
    sync_lanes();
    return true;
  }

  // ---- MCTS::find_leaf ---------------------------------------------------------------------
  // Descends from the root of `seat`'s tree, expands an unvisited node.  Outputs the leaf
  // state and its terminal code; stores MCTS::current_ / path_ for process_result.
  // LEAN: the instantiation of the per-simulation kernel (k_sim): plain PUCT only, no Gumbel code
  // rec != nullptr: also leaves the path's lane image there (the move step of a split round hands it to the next k_sim)
  template <bool LEAN = false>
  __device__ __forceinline__ bool find_leaf(uint32_t seat, typename GM::State& leaf, uint32_t& term, PathRegs* rec = nullptr) {
    sync_lanes();
    const size_t tb = tree_base(seat);
    const uint32_t root = AZMI_SEL(t_root, seat);
    cur = root; plen = 0;
    leaf = gs;
    uint64_t meta = ar.nodes[tb + cur].meta;
    uint32_t n = ar.nodes[tb + cur].n;
    uint32_t* path = ar.path + static_cast<size_t>(slot) * ep.max_depth;
    uint32_t gum_active = 0;   // (uint32_t: a bool carried across the descent loop is mis-tracked by hipcc in divergent groups)
    if (!LEAN && seat_gumbel(seat)) {  // lazy init, mcts.cc:465-472 (MCTS::gumbel_enabled_ of this seat's tree)
      const uint32_t* st = gum_state(seat);
      gum_active = st[kGumInit];
      if (!gum_active && st[kGumTarget] > 0 && n > 0 && meta_nch(meta) != 0) {
        const uint32_t k0 = meta_nch(meta);
        const float p0 = lane < k0 ? ar.nodes[tb + meta_ch0(meta) + lane].pr : 0.0f;
        init_gumbel_state(seat, k0, p0);
        gum_active = gum_state(seat)[kGumInit];
      }
    }
    while (n > 0 && meta_term(meta) == 0) {
      if (plen >= ep.max_depth) { raise(8u); return false; }
      if (lane == 0) path[plen] = cur;
      ++plen;
      const uint32_t k = meta_nch(meta), c0 = meta_ch0(meta);
      if (k == 0) { raise(8u); return false; }  // reference: children.at(0) throws
      const size_t ci = tb + c0 + lane;
      uint32_t n_l = 0; float q_l = 0.0f, p_l = 0.0f; uint64_t m_l = 0;
      if (lane < k) { n_l = ar.nodes[ci].n; q_l = ar.nodes[ci].q; p_l = ar.nodes[ci].pr; m_l = ar.nodes[ci].meta; }
      float d_l = 0.0f, vv_l = 0.0f;
      if (rec && lane < k) { d_l = ar.nodes[ci].d; vv_l = ar.nodes[ci].v; }
      const float fpu = (cur == root && seat_fpu_zero(seat)) ? 0.0f : ep.fpu_reduction;
      const float v_parent = ar.nodes[tb + cur].v;
      uint32_t best;
      if (!LEAN && gum_active && cur == root) best = gumbel_next_root_child(seat, k, n_l, q_l, p_l);
      else if (!LEAN && gum_active && seat_gumbel_full(seat)) best = gumbel_interior_select(seat, k, n_l, q_l, p_l, v_parent);
      else best = select_child(k, n_l, q_l, p_l, v_parent, n, fpu);
      if (rec) {
        const float s_q = bcast(q_l, best), s_d = bcast(d_l, best), s_v = bcast(vv_l, best);
        const uint32_t s_n = bcast(n_l, best);
        if (plen - 1 == lane) { rec->node = c0 + best; rec->n = s_n; rec->q = s_q; rec->d = s_d; rec->v = s_v; rec->pp = meta_player(meta); }
      }
      cur = c0 + best;
      n = bcast(n_l, best);
      meta = bcast(m_l, best);
      GM::play(leaf, meta_mv(meta));
      trace(108);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) if (static_cast<uint32_t>(p) == seat) t_tld[p] += plen;
    term = meta_term(meta);
    if (rec) { rec->leaf_meta = meta; rec->mv = 0; rec->ok = plen <= static_cast<uint32_t>(G) ? 1u : 0u; }
    if (n == 0) {
      term = GM::terminal(leaf);
      trace(109);
      const uint64_t keep = meta_pack(0, 0, meta_mv(meta), leaf.player, term);
      uint32_t c0, k, mv = 0;
      if (!expand_node(seat, cur, leaf, keep, c0, k, &mv)) return false;
      if (rec) { rec->leaf_meta = meta_pack(c0, k, meta_mv(meta), leaf.player, term); rec->mv = mv; }
    }
    return true;
  }
  // the pending simulation's lane image to / from HBM (split rounds)
  __device__ __forceinline__ void store_pend(const PathRegs& r) const {
    PendRec* p = ar.pend + static_cast<size_t>(slot) * G + lane;
    p->node = r.node; p->n = r.n; p->q = r.q; p->d = r.d; p->v = r.v;
    p->pp_mv = (r.pp & 0xFFu) | (r.mv << 8);
    p->leaf_meta = r.leaf_meta;
  }

  // ---- MCTS::add_root_noise: children in lanes [0,k), priors p_l -----------------------------
  __device__ __forceinline__ float add_root_noise(uint32_t k, float p_l, float eps) {
    float noise_l = 0.0f;
    double sum = 0.0;
    if (ep.shaped && k > 1) {
      const float Nf = static_cast<float>(k);
      const float lp_l = lane < k ? az_logf(fminf(p_l, 0.01f) + 1e-20f) : 0.0f;
      const float log_sum = seqsum(lp_l, k);
      const float log_mean = log_sum / Nf;
      const float sh_l = lane < k ? fmaxf(0.0f, lp_l - log_mean) : 0.0f;
      const float shaped_sum = seqsum(sh_l, k);
      const float uniform = 1.0f / Nf;
      for (uint32_t i = 0; i < k; ++i) {
        const float shaped = bcast(sh_l, i);
        float alpha_prop = (shaped_sum > 0) ? 0.5f * (shaped / shaped_sum + uniform) : uniform;
        alpha_prop = fmaxf(alpha_prop, 1e-6f);
        Gamma dist(kNoiseAlphaRatio * alpha_prop);  // fresh object per child (mcts.cc:430)
        const float g = dist.draw(rng);
        if (lane == i) noise_l = g;
        sum += g;
      }
    } else {
      Gamma dist(kNoiseAlphaRatio / static_cast<float>(k));  // one object, normal cache carries
      for (uint32_t i = 0; i < k; ++i) {
        const float g = dist.draw(rng);
        if (lane == i) noise_l = g;
        sum += g;
      }
    }
    return p_l * (1 - eps) + eps * noise_l / static_cast<float>(sum);
  }

  // ---- MCTS::process_result -------------------------------------------------------------------
  // from_net: read (v, pi) rows written by the net; otherwise synthesise dumb_eval.
  // have_regs: the answer is already in registers (a cache hit of this very round: lane m holds pi[m] in reg_pi, lane
  // i <= P holds v[i] in reg_v), which saves reading back rows the probe stored a moment ago.
  // The kernel is bound by dependent HBM round trips (~2 us each under load), so all loads of one dependency level are
  // issued together: level 1 = leaf META, path entries, value row; level 2 = children moves, parents' META and the
  // path nodes' N/Q/D; level 3 = the prior gather.  The priors half (writes Pr of the leaf's children) and the backup
  // half (reads/writes N, Q, D, V of path nodes) touch disjoint data, so hoisting the backup loads is safe.
  // LEAN (k_sim): the evaluated leaf is never the root, so the root temperature / Dirichlet code is not instantiated
  template <bool LEAN = false>
  __device__ __forceinline__ void process_result(uint32_t seat, bool from_net, bool root_noise, bool have_regs = false,
                                                 float reg_pi = 0.0f, float reg_v = 0.0f) {
    sync_lanes();
    const size_t tb = tree_base(seat);
    const uint32_t root = AZMI_SEL(t_root, seat);
    const uint32_t* path = ar.path + static_cast<size_t>(slot) * ep.max_depth;
    // ---- level 1
    const uint64_t meta = ar.nodes[tb + cur].meta;
    const bool lvl = lane < plen;                       // this lane backs up level `lane` (first chunk of the path)
    const uint32_t parent0 = lvl ? path[lane] : 0u;
    const uint32_t node0 = lvl ? ((lane == plen - 1) ? cur : path[lane + 1]) : 0u;
    float vrow = 0.0f;
    if (from_net && !have_regs && lane <= static_cast<uint32_t>(P)) vrow = ar.v[static_cast<size_t>(slot) * (P + 1) + lane];
    const uint32_t term = meta_term(meta);
    const uint32_t k = meta_nch(meta), c0 = meta_ch0(meta);
    const size_t ci = tb + c0 + lane;
    // ---- level 2
    uint64_t cmeta = 0;
    if (term == 0 && lane < k) cmeta = ar.nodes[ci].meta;
    uint64_t pmeta0 = 0, nmeta0 = 0; uint32_t nn0 = 0; float q0 = 0.0f, d0 = 0.0f;
    if (lvl) {
      const size_t ni = tb + node0;
      pmeta0 = ar.nodes[tb + parent0].meta; nmeta0 = ar.nodes[ni].meta;
      nn0 = ar.nodes[ni].n; q0 = ar.nodes[ni].q; d0 = ar.nodes[ni].d;
    }
    uint32_t rn = 0; uint64_t rmeta = 0;
    if (lane == 0) { rn = ar.nodes[tb + root].n; rmeta = ar.nodes[tb + root].meta; }
    float val[P + 1];
    if (term != 0) {
#pragma unroll
      for (int i = 0; i <= P; ++i) val[i] = (static_cast<int>(term) - 1 == i) ? 1.0f : 0.0f;
    } else {
      float p = 0.0f;
      if (from_net) {
        const float vsrc = have_regs ? reg_v : vrow;
#pragma unroll
        for (int i = 0; i <= P; ++i) val[i] = bcast(vsrc, i);
        // ---- level 3: the prior of this lane's child
        if (have_regs) p = bcast(reg_pi, static_cast<int>(meta_mv(cmeta)));
        else if (lane < k) p = ar.pi[static_cast<size_t>(slot) * M + meta_mv(cmeta)];
        if (lane >= k) p = 0.0f;
      } else {  // dumb_eval: uniform over legal moves, u8 sum wraps (game_state.h:160-173)
#pragma unroll
        for (int i = 0; i <= P; ++i) val[i] = static_cast<float>(1.0 / (P + 1));
        const float ksum = static_cast<float>(k & 0xFFu);
        if (lane < k) p = (ksum == 0.0f) ? 0.0f : 1.0f / ksum;
      }
      const bool is_root = !LEAN && cur == root;
      if constexpr (!LEAN) {
        const float root_temp = seat_root_temp(seat);
        if (is_root && root_temp != 1.0f && lane < k) p = az_powf(p, 1.0f / root_temp);
      }
      const float sum = seqsum8(lane < k ? p : 0.0f);
      p = p / sum;
      if constexpr (!LEAN) {
        if (is_root && root_noise && !seat_gumbel(seat)) { trace(1 | (static_cast<uint64_t>(k) << 8)); p = add_root_noise(k, p, seat_eps(seat)); trace(2); }
      }
      if (lane < k) ar.nodes[ci].pr = p;
    }
    // backup: level i updates node_i (child of path[i]); levels are independent -> one lane each
    const float draw_share = val[P] / static_cast<int32_t>(P);
    for (uint32_t base = 0; base < plen; base += G) {
      const uint32_t i = base + lane;
      if (i < plen) {
        uint32_t node, nn; uint64_t pmeta, nmeta; float q, d;
        if (base == 0) { node = node0; nn = nn0; pmeta = pmeta0; nmeta = nmeta0; q = q0; d = d0; }
        else {
          node = (i == plen - 1) ? cur : path[i + 1];
          pmeta = ar.nodes[tb + path[i]].meta; nmeta = ar.nodes[tb + node].meta;
          nn = ar.nodes[tb + node].n; q = ar.nodes[tb + node].q; d = ar.nodes[tb + node].d;
        }
        const uint32_t pp = meta_player(pmeta);
        const size_t ni = tb + node;
        float vv = (pp == 0) ? val[0] : val[1];
        if (P > 2) vv = val[pp];
        vv += draw_share;
        ar.nodes[ni].q = (q * static_cast<float>(nn) + vv) / static_cast<float>(nn + 1);
        ar.nodes[ni].d = (d * static_cast<float>(nn) + val[P]) / static_cast<float>(nn + 1);
        if (nn == 0) {
          const uint32_t np = meta_player(nmeta);
          ar.nodes[ni].v = ((np == 0) ? val[0] : val[1]) + draw_share;
        }
        ar.nodes[ni].n = nn + 1;
      }
    }
    if (lane == 0) {
      const size_t ri = tb + root;
      // the root is never a path NODE (nodes are children), so its N was not changed by the loop above
      if (rn == 0) {
        const uint32_t rp = meta_player(rmeta);
        ar.nodes[ri].v = ((rp == 0) ? val[0] : val[1]) + draw_share;
        ar.nodes[ri].d = val[P];
      }
      ar.nodes[ri].n = rn + 1;
      ar.c_sims[slot] += 1;
    }
#pragma unroll
    for (int p = 0; p < P; ++p) if (static_cast<uint32_t>(p) == seat) t_depth[p] += 1;
    sync_lanes();
  }

  // ---- playout_eval, game_state.cc:10-54: uniform policy over the leaf's legal moves + the scores of a uniformly random
  // rollout, written to the slot's (v, pi) rows like a net answer.  One draw of the slot's rollout stream per move picks
  // the lemire_below(#legal)-th legal move in ascending order; every lane of the group plays the same rollout.
  __device__ __forceinline__ void playout_eval(const typename GM::State& leaf) {
    Pcg32 roll;
    roll.state = ar.roll[slot];
    typename GM::State sim = leaf;
    uint32_t term = GM::terminal(sim);
    while (term == 0) {
      const uint32_t k = GM::num_valid(sim);
      if (k == 0) break;
      GM::play(sim, GM::nth_valid(sim, lemire_below(roll, k)));
      term = GM::terminal(sim);
    }
    const uint32_t kl = GM::num_valid(leaf);
    const float ksum = static_cast<float>(kl & 0xFFu);   // Vector<uint8_t>::sum() wraps mod 256
    if (lane < static_cast<uint32_t>(M)) {
      const bool legal = (GM::valid_mask(leaf) >> lane) & 1u;
      ar.pi[static_cast<size_t>(slot) * M + lane] = (legal && ksum > 0.0f) ? 1.0f / ksum : 0.0f;
    }
    if (lane <= static_cast<uint32_t>(P))
      ar.v[static_cast<size_t>(slot) * (P + 1) + lane] = term ? ((term - 1 == lane) ? 1.0f : 0.0f) : static_cast<float>(1.0 / (P + 1));
    sync_lanes();
    if (lane == 0) ar.roll[slot] = roll.state;
  }

  // ---- lane-dense [M] vector helpers (lane m holds entry m) ------------------------------------
  template <class T>
  __device__ __forceinline__ T scatter_by_move(uint32_t k, uint32_t mv_l, T x_l) const {
    T out = T(0);
    for (uint32_t i = 0; i < k; ++i) {
      const uint32_t mi = bcast(mv_l, i);
      const T xi = bcast(x_l, i);
      if (lane == mi) out = xi;
    }
    return out;
  }
  __device__ __forceinline__ float pow_entry(float x, float e) const { return e == 1.0f ? x : az_powf(x, e); }

  // MCTS::probs — cnt_m / pol_m are the dense root counts and priors
  __device__ __forceinline__ float probs(float temp, uint32_t cnt_m, float pol_m) const {
    const bool in = lane < M;
    const float count_sum = seqsum(in ? static_cast<float>(cnt_m) : 0.0f, M);
    if (count_sum == 0) {
      float p = in ? pol_m : 0.0f;
      if (temp != 0.0f) p = in ? pow_entry(p, 1.0f / temp) : 0.0f;
      return p / seqsum(p, M);
    }
    if (temp == 0) {
      uint32_t best = in ? cnt_m : 0u;
      for (int off = 1; off < G; off <<= 1) best = max(best, __shfl_xor(best, off, G));
      const bool is_best = in && cnt_m == best;
      uint32_t nbest = is_best ? 1u : 0u;
      for (int off = 1; off < G; off <<= 1) nbest += __shfl_xor(nbest, off, G);
      return is_best ? static_cast<float>(1.0 / static_cast<double>(nbest)) : 0.0f;
    }
    float p = in ? static_cast<float>(cnt_m) : 0.0f;
    p = p / seqsum(p, M);
    p = in ? pow_entry(p, 1 / temp) : 0.0f;
    return p / seqsum(p, M);
  }

  // MCTS::probs_pruned — children in lanes, returns the dense vector
  __device__ __forceinline__ float probs_pruned(float temp, uint32_t root_n, uint32_t k, uint32_t mv_l, uint32_t n_l, float q_l,
                                float p_l, uint32_t cnt_m, float pol_m) const {
    if (root_n <= 1) return probs(temp, cnt_m, pol_m);
    const float explore_scaling = ep.cpuct * sqrtf(static_cast<float>(root_n));
    float best_sel = -1e30f;
    for (uint32_t i = 0; i < k; ++i) {
      const uint32_t ni = bcast(n_l, i);
      const float sel = bcast(q_l, i) + explore_scaling * bcast(p_l, i) / static_cast<float>(ni + 1);
      if (ni != 0 && sel > best_sel) best_sel = sel;
    }
    float pr_l = 0.0f;
    if (lane < k && n_l != 0) {
      const float gap = best_sel - q_l;
      const float nf = static_cast<float>(n_l);
      const float desired = (gap <= 0) ? nf : explore_scaling * p_l / gap - 1.0f;
      const float lo = (0.0f < desired) ? desired : 0.0f;  // std::max(0.0f, desired)
      pr_l = (lo < nf) ? lo : nf;                          // std::min(n, .)
    }
    float pruned = scatter_by_move<float>(k, mv_l, pr_l);
    const bool in = lane < M;
    const float total = seqsum(in ? pruned : 0.0f, M);
    if (total == 0) return probs(temp, cnt_m, pol_m);
    if (temp == 0) {
      float best_val = bcast(pruned, 0);
      for (uint32_t m = 1; m < static_cast<uint32_t>(M); ++m) {
        const float o = bcast(pruned, m);
        best_val = (best_val < o) ? o : best_val;
      }
      const bool is_best = in && pruned == best_val;
      uint32_t cnt = is_best ? 1u : 0u;
      for (int off = 1; off < G; off <<= 1) cnt += __shfl_xor(cnt, off, G);
      return is_best ? 1.0f / static_cast<int32_t>(cnt) : 0.0f;
    }
    pruned = in ? pruned / total : 0.0f;
    if (temp != 1.0f) {
      pruned = in ? pow_entry(pruned, 1.0f / temp) : 0.0f;
      pruned = pruned / seqsum(pruned, M);
    }
    return pruned;
  }

  // MCTS::pick_move — one uniform draw, first m with running sum > choice
  __device__ __forceinline__ uint32_t pick_move(float p_m) {
    const float choice = canonical01(rng) * 1.0f + 0.0f;
    float sum = 0.0f;
    for (uint32_t m = 0; m < static_cast<uint32_t>(M); ++m) {
      sum += bcast(p_m, m);
      if (sum > choice) return m;
    }
    for (int m = M - 1; m >= 0; --m)
      if (bcast(p_m, m) > 0) return static_cast<uint32_t>(m);
    raise(16u);  // reference throws "this shouldn't be possible."
    return 0;
  }

  // ---- MCTS::update_root ----------------------------------------------------------------------------
  __device__ __forceinline__ bool update_root(uint32_t seat, uint32_t move) {
    sync_lanes();
    const size_t tb = tree_base(seat);
    const uint32_t root = AZMI_SEL(t_root, seat);
    const uint64_t meta = ar.nodes[tb + root].meta;
    uint32_t k = meta_nch(meta), c0 = meta_ch0(meta);
    if (k == 0 && !expand_node(seat, root, gs, meta, c0, k)) return false;
    uint32_t hit = 0xFFFFu;
    if (lane < k && meta_mv(ar.nodes[tb + c0 + lane].meta) == move) hit = lane;
    for (int off = 1; off < G; off <<= 1) hit = min(hit, __shfl_xor(hit, off, G));
    if (hit == 0xFFFFu) { raise(32u); return false; }  // "ahh, what is this move"
#pragma unroll
    for (int p = 0; p < P; ++p)
      if (static_cast<uint32_t>(p) == seat) { t_root[p] = c0 + hit; t_depth[p] = 0; t_tld[p] = 0; }
    if (ep.gumbel_on) reset_gumbel_state(seat);  // mcts.cc:172
    return true;
  }

  // ---- MCTS::apply_root_policy_temp + add_root_noise on a reused subtree (play_manager.cc:541-555) ---
  __device__ __forceinline__ void reapply_root_prior(uint32_t seat, bool noise) {
    sync_lanes();
    const size_t tb = tree_base(seat);
    const uint32_t root = AZMI_SEL(t_root, seat);
    if (ar.nodes[tb + root].n == 0) return;
    const uint64_t meta = ar.nodes[tb + root].meta;
    const uint32_t k = meta_nch(meta), c0 = meta_ch0(meta);
    const size_t ci = tb + c0 + lane;
    float p = lane < k ? ar.nodes[ci].pr : 0.0f;
    bool dirty = false;
    const float root_temp = seat_root_temp(seat);
    if (root_temp != 1.0f) {
      if (lane < k) p = az_powf(p, 1.0f / root_temp);
      const float sum = seqsum8(lane < k ? p : 0.0f);
      if (sum > 0.0f) p = p / sum;
      dirty = true;
    }
    if (noise && k > 0) { trace(3 | (static_cast<uint64_t>(k) << 8)); p = add_root_noise(k, p, seat_eps(seat)); trace(4); dirty = true; }
    if (dirty && lane < k) ar.nodes[ci].pr = p;
  }

  // ---- new game: GameData reset + fresh trees (play_manager.cc:214-230, 515-520) ------------------------
  __device__ __forceinline__ void start_game() {
    gs = GM::initial();
    for (uint32_t s = 0; s < static_cast<uint32_t>(P); ++s) reset_tree(s);
    ph_rows = 0;
  }
  __device__ __forceinline__ void draw_capped() {  // play_manager.cc:523-524 / 559-560
    const bool capped = ep.cap_rand && (canonical01(coin) < ep.cap_percent);
    flags = capped ? (flags | kFlagCapped) : (flags & ~kFlagCapped);
  }

  // ---- the "actually play a move" block, play_manager.cc:286-556.  Returns true if the game ended. ------
  __device__ __forceinline__ bool make_move(uint32_t cp) {
    sync_lanes();
    const size_t tb = tree_base(cp);
    const bool capped = flags & kFlagCapped;
    const uint32_t root = AZMI_SEL(t_root, cp);
    const uint64_t rmeta = ar.nodes[tb + root].meta;
    const uint32_t k = meta_nch(rmeta), c0 = meta_ch0(rmeta);
    const uint32_t root_n = ar.nodes[tb + root].n;
    const size_t ci = tb + c0 + lane;
    uint32_t n_l = 0, mv_l = 0; float q_l = 0, p_l = 0, d_l = 0;
    if (lane < k) { n_l = ar.nodes[ci].n; q_l = ar.nodes[ci].q; p_l = ar.nodes[ci].pr; d_l = ar.nodes[ci].d; mv_l = meta_mv(ar.nodes[ci].meta); }
    const uint32_t cnt_m = scatter_by_move<uint32_t>(k, mv_l, n_l);
    const float pol_m = scatter_by_move<float>(k, mv_l, p_l);

    float temp = ep.start_temp;
    if (ep.half_life != 0) {  // play_manager.cc:297-304; ln2 is the literal 0.693
      const float lambda = 0.693f / ep.half_life;
      temp -= ep.final_temp;
      temp *= az_expf(-lambda * gs.turn);
      temp += ep.final_temp;
    }
    // resign, play_manager.cc:305-334
    int resign_entry = -1;
    if (ep.resign_percent > 0 && !(flags & kFlagPlaythrough)) {
      float q = 0, d = 0; bool found = false;  // MCTS::root_value, mcts.h:78-100
      for (uint32_t i = 0; i < k; ++i) {
        const uint32_t ni = bcast(n_l, i); const float qi = bcast(q_l, i), di = bcast(d_l, i);
        if (ni > 0 && qi > q) { q = qi; d = di; found = true; }
      }
      if (!found && root_n > 0) { q = ar.nodes[tb + root].v; d = ar.nodes[tb + root].d; }
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
        float q = 0, d = 0; bool found = false;  // MCTS::root_value, mcts.h:78-100
        for (uint32_t i = 0; i < k; ++i) {
          const uint32_t ni = bcast(n_l, i); const float qi = bcast(q_l, i), di = bcast(d_l, i);
          if (ni > 0 && qi > q) { q = qi; d = di; found = true; }
        }
        if (!found && root_n > 0) { q = ar.nodes[tb + root].v; d = ar.nodes[tb + root].d; }
        const float w = q - d / static_cast<int32_t>(P);
        const float l = static_cast<float>(1.0 - static_cast<double>(w) - static_cast<double>(d));
        const float v_self = w - l;
        uint32_t* streak = ar.resign_streak + static_cast<size_t>(slot) * P + cp;
        const uint32_t now = v_self <= seat_thresh ? *streak + 1u : 0u;
        sync_lanes();
        if (lane == 0) *streak = now;
        if (now >= seat_resign_need(cp)) resign_entry = static_cast<int>((cp + 1) % 2);
      }
    }
    // move choice, play_manager.cc:367-406
    const uint64_t rng_before = rng.state;
    uint32_t chosen;
    if (seat_gumbel(cp) && !capped) {
      if (!seat_gumbel_g3(cp)) {
        chosen = gumbel_final_action(cp, k, mv_l, n_l, q_l, p_l, cnt_m, pol_m);   // G1 acting
      } else {   // G3 opt-in: sample from improved policy ^ (1 / temp)
        const bool in = lane < static_cast<uint32_t>(M);
        float pg = gumbel_improved_policy(cp, k, mv_l, n_l, q_l, p_l, ar.nodes[tb + root].v);
        if (temp != 1.0f && temp > 0.0f) {
          pg = in ? az_powf(pg, 1.0f / temp) : 0.0f;
          const float sg = seqsum(pg, M);
          if (sg > 0) pg = pg / sg;
        } else if (temp <= 0.0f) {   // arg-max of pi' (first maximum), as a one-hot vector
          float bv = in ? pg : -__builtin_inff(); uint32_t bi = in ? lane : 0xFFFFu;
          for (int off = 1; off < G; off <<= 1) {
            const float ov = __shfl_xor(bv, off, G); const uint32_t oi = __shfl_xor(bi, off, G);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
          }
          pg = (lane == bi) ? 1.0f : 0.0f;
        }
        const float sg = seqsum(in ? pg : 0.0f, M);
        if (sg > 0) chosen = pick_move(pg);
        else chosen = gumbel_final_action(cp, k, mv_l, n_l, q_l, p_l, cnt_m, pol_m);
      }
    } else {
      chosen = pick_move(probs(temp, cnt_m, pol_m));
    }
    trace(5 | (static_cast<uint64_t>(chosen) << 8));

    if (ep.log_moves) {
      uint32_t row = 0;
      if (lane == 0) row = atomicAdd(&ar.ctl->log_rows, 1u);
      row = bcast(row, 0);
      if (row < ep.log_cap) {
        if (lane == 0) {
          uint32_t* r = ar.log_rows + static_cast<size_t>(row) * 8;
          r[0] = slot; r[1] = ar.slot_games[slot]; r[2] = chosen; r[3] = gs.turn; r[4] = cp; r[5] = capped ? 1u : 0u;
          r[6] = static_cast<uint32_t>(rng_before); r[7] = static_cast<uint32_t>(rng_before >> 32);
        }
        if (lane < static_cast<uint32_t>(M)) ar.log_counts[static_cast<size_t>(row) * M + lane] = cnt_m;
      } else {
        raise(4u);
      }
    }
    // history sample, play_manager.cc:407-424
    if (ep.history && !capped) {
      const float target = ep.gumbel_hist ? gumbel_improved_policy(cp, k, mv_l, n_l, q_l, p_l, ar.nodes[tb + root].v)  // play_manager.cc:411-417
                           : (ep.pruning && seat_eps(cp) > 0)
                               ? probs_pruned(1.0f, root_n, k, mv_l, n_l, q_l, p_l, cnt_m, pol_m)
                               : probs(1.0f, cnt_m, pol_m);
      const uint32_t r = ph_rows;
      if (r < ep.max_hist_rows) {
        // the pending row keeps the packed position (kPendingWords u64 = 24 B), not its 672 B of planes: the planes are
        // generated from it when the game's rows are committed (end_game), so a row costs one small store here and no
        // read-back there
        if (lane == 0) {
          uint64_t* st = reinterpret_cast<uint64_t*>(ar.ph_canon) + (static_cast<size_t>(slot) * ep.max_hist_rows + r) * kPendingWords;
          st[0] = gs.bb[0]; st[1] = gs.bb[1];
          st[2] = static_cast<uint64_t>(gs.turn) | (static_cast<uint64_t>(gs.player) << 32);
        }
        if (lane < static_cast<uint32_t>(M)) ar.ph_pi[(static_cast<size_t>(slot) * ep.max_hist_rows + r) * M + lane] = target;
        if (lane == 0) {
          uint32_t* pm = ar.ph_meta + (static_cast<size_t>(slot) * ep.max_hist_rows + r) * 2;
          pm[0] = gs.player; pm[1] = gs.turn;
        }
        ph_rows = r + 1;
      } else {
        raise(2u);
      }
    }
    // stats, play_manager.cc:425-435
    {
      const uint32_t dep = AZMI_SEL(t_depth, cp);
      const float ald = dep == 0 ? 0.0f : static_cast<float>(AZMI_SEL(t_tld, cp)) / static_cast<float>(dep);
      float ent = 0.0f;  // MCTS::normalized_root_entropy, mcts.cc:737-750
      const float kf = static_cast<float>(k);
      if (!(kf <= 1 || root_n <= 1)) {
        const float log_k = az_logf(kf);
        const float total_n = static_cast<float>(root_n);
        float term_l = 0.0f;
        if (lane < k && n_l > 0) { const float p = static_cast<float>(n_l) / total_n; term_l = p * az_logf(p); }
        float e = 0.0f;
        for (uint32_t i = 0; i < k; ++i) { const uint32_t ni = bcast(n_l, i); const float ti = bcast(term_l, i); if (ni > 0) e -= ti; }
        ent = e / log_k;
      }
      if (lane == 0) {
        const uint32_t S = ep.S;
        if (!capped) { ar.g_dsum[0 * S + slot] += ald; ar.g_dsum[1 * S + slot] += ent; ar.g_cnt[1 * S + slot] += 1; }
        else { ar.g_dsum[2 * S + slot] += ald; ar.g_dsum[3 * S + slot] += ent; ar.g_cnt[2 * S + slot] += 1; }
        ar.g_dsum[4 * S + slot] += k;
        ar.g_cnt[0 * S + slot] += 1;
      }
    }
    // re-root every seat's tree in seat order, then play the move (play_manager.cc:436-439)
    for (uint32_t s = 0; s < static_cast<uint32_t>(P); ++s)
      { if (!update_root(s, chosen)) return true; trace(6); }
    if (!GM::play(gs, chosen)) { raise(64u); return true; }
    uint32_t term = GM::terminal(gs);
    bool resigned = false;
    if (term == 0 && resign_entry >= 0) { term = static_cast<uint32_t>(resign_entry) + 1; resigned = true; }
    if (term != 0) {
      end_game(term, resigned);
      return true;
    }
    // play_manager.cc:522-555
    draw_capped();
    set_gumbel_target();
    if (!ep.tree_reuse) {
      for (uint32_t s = 0; s < static_cast<uint32_t>(P); ++s) reset_tree(s);
    } else {
      reapply_root_prior(gs.player, seat_eps(gs.player) > 0 && !(flags & kFlagCapped));
    }
    return false;
  }

  // ---- game end: flush history with the final scores, commit totals (play_manager.cc:446-505) -----------
  __device__ __forceinline__ void end_game(uint32_t term, bool resigned) {
    const uint32_t S = ep.S;
    sync_lanes();
    const uint32_t rows = ep.history ? ph_rows : 0u;
    if (rows > 0) {
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(&ar.ctl->hist_rows, static_cast<unsigned long long>(rows));
      base = bcast(base, 0);
      // the finished-sample store is a ring of hist_cap rows (the reference's history queue is unbounded and drained by
      // hist_saver, game_runner.py:729-747): row i of the run lives at i % hist_cap; it overflows only when the host has not
      // consumed enough (hist_rows - hist_read > hist_cap).  hist_rows / hist_read are free-running 64-bit counters.
      if (base + rows - ar.ctl->hist_read <= ep.hist_cap) {
        const uint32_t game_idx = ar.slot_games[slot];
        // The game's rows are contiguous on the pending side and contiguous modulo the ring on the other: a per-row
        // copy loop is a chain of dependent round trips (measured 230 us per game end, which stretched every eighth round
        // of a shard), so the planes are regenerated from packed positions and the policy rows move as flat copies
        // (two segments when the game's rows wrap around the end of the ring).
        const size_t src0 = static_cast<size_t>(slot) * ep.max_hist_rows;
        const uint32_t first = static_cast<uint32_t>(base % ep.hist_cap);
        const uint32_t n1 = rows < ep.hist_cap - first ? rows : ep.hist_cap - first;
        {
          // canonical planes from the packed pending positions: eight rows' positions per round trip, planes written straight
          // into the ring (stores only)
          const uint64_t* ps = reinterpret_cast<const uint64_t*>(ar.ph_canon) + src0 * kPendingWords;
          for (uint32_t r0 = 0; r0 < rows; r0 += G) {
            const uint32_t rr = r0 + lane;
            uint64_t w0 = 0, w1 = 0, w2 = 0;
            if (rr < rows) { w0 = ps[rr * kPendingWords]; w1 = ps[rr * kPendingWords + 1]; w2 = ps[rr * kPendingWords + 2]; }
            const uint32_t nb = (rows - r0) < static_cast<uint32_t>(G) ? (rows - r0) : static_cast<uint32_t>(G);
            for (uint32_t j = 0; j < nb; ++j) {
              typename GM::State st;
              st.bb[0] = bcast(w0, static_cast<int>(j)); st.bb[1] = bcast(w1, static_cast<int>(j));
              const uint64_t t = bcast(w2, static_cast<int>(j));
              st.turn = static_cast<uint32_t>(t); st.player = static_cast<uint32_t>(t >> 32);
              const uint32_t r = r0 + j;
              float* drow = ar.h_canon + static_cast<size_t>(r < n1 ? first + r : r - n1) * GM::CANON;
              for (uint32_t e = lane; e < static_cast<uint32_t>(GM::CANON); e += G) drow[e] = GM::canonical_at(st, e);
            }
          }
          for (uint32_t seg = 0; seg < 2; ++seg) {
            const uint32_t sr = seg == 0 ? 0u : n1, nr = seg == 0 ? n1 : rows - n1;
            const size_t dst0 = seg == 0 ? first : 0u;
            const float* sp = ar.ph_pi + (src0 + sr) * M;
            float* dp = ar.h_pi + dst0 * M;
            const uint32_t np = nr * M;
            for (uint32_t e0 = 0; e0 < np; e0 += 4 * G) {
              float t0 = 0, t1 = 0, t2 = 0, t3 = 0;
              const uint32_t a = e0 + lane, b = a + G, c = b + G, d = c + G;
              if (a < np) t0 = sp[a];
              if (b < np) t1 = sp[b];
              if (c < np) t2 = sp[c];
              if (d < np) t3 = sp[d];
              if (a < np) dp[a] = t0;
              if (b < np) dp[b] = t1;
              if (c < np) dp[c] = t2;
              if (d < np) dp[d] = t3;
            }
            float* dv = ar.h_v + dst0 * (P + 1);
            for (uint32_t e = lane; e < nr * (P + 1); e += G) dv[e] = (e % (P + 1) == term - 1) ? 1.0f : 0.0f;
            for (uint32_t r = lane; r < nr; r += G) {
              const uint32_t* pm = ar.ph_meta + (src0 + sr + r) * 2;
              uint32_t* hm = ar.h_meta + (dst0 + r) * 4;
              hm[0] = slot; hm[1] = game_idx; hm[2] = pm[1]; hm[3] = pm[0];
            }
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
      atomicAdd(&ar.a_perm_scores[(static_cast<size_t>(slot) * ep.num_perms + perm) * (P + 1) + (term - 1)], 1.0f);   // play_manager.cc:466-467
      atomicAdd(&ar.a_perm_games[static_cast<size_t>(slot) * ep.num_perms + perm], 1u);
      // one round trip for the game's running totals, then accumulations that return nothing (each cell has this
      // one writer, so an atomic add is the same sum as load-add-store without the dependent load)
      double gd[5]; uint32_t gc[3];
#pragma unroll
      for (int j = 0; j < 5; ++j) gd[j] = ar.g_dsum[j * S + slot];
#pragma unroll
      for (int j = 0; j < 3; ++j) gc[j] = ar.g_cnt[j * S + slot];
      atomicAdd(reinterpret_cast<unsigned long long*>(&ar.a_len[slot]), static_cast<unsigned long long>(gs.turn));
#pragma unroll
      for (int j = 0; j < 5; ++j) { atomicAdd(&ar.a_dsum[j * S + slot], gd[j]); ar.g_dsum[j * S + slot] = 0.0; }
#pragma unroll
      for (int j = 0; j < 3; ++j) { atomicAdd(reinterpret_cast<unsigned long long*>(&ar.a_cnt[j * S + slot]), static_cast<unsigned long long>(gc[j])); ar.g_cnt[j * S + slot] = 0; }
      atomicAdd(&ar.slot_games[slot], 1u);
      const uint32_t pos = atomicAdd(&ar.ctl->ended_count, 1u);
      ar.ended_list[pos] = slot;
    }
  }

  // ---- leaf hand-off to the net: canonical planes + position key (play_manager.cc:589-598) ----------------
  __device__ __forceinline__ void emit_leaf(const typename GM::State& leaf, uint64_t key) const {
    float* row = ar.canon + static_cast<size_t>(slot) * GM::CANON;
    for (uint32_t e = lane; e < static_cast<uint32_t>(GM::CANON); e += G) row[e] = GM::canonical_at(leaf, e);
    if (lane == 0) {
      ar.leaf_key[slot] = key;
      if (ar.leaf_pos) {     // the packed position: what the asynchronous pipeline's request carries instead of the planes
        ar.leaf_pos[0 * static_cast<size_t>(ep.S) + slot] = leaf.bb[0];
        ar.leaf_pos[1 * static_cast<size_t>(ep.S) + slot] = leaf.bb[1];
        ar.leaf_pos[2 * static_cast<size_t>(ep.S) + slot] = leaf.player;
      }
    }
  }
  // position-cache probe (play_manager.cc:592-597): on a hit the cached (pi, v) land in the slot's rows AND in
  // registers (lane m: pi[m] in hit_pi, lane i <= P: v[i] in hit_v) for a process_result later in this round
  // (before_loads: see wave_shard_find - the pipeline asks for its answer table's granules there; early(cslot, pi, v): called when
  // the shard's keys are back - true = the caller has the answer already (its answer table's granules were valid), the payload
  // round trip is skipped; after_hit(pi, v): called with a shard hit's payload (the pipeline copies it into its answer table))
  struct NoEarly { __device__ __forceinline__ bool operator()(int, float&, float&) const { return false; } };
  struct NoAfter { __device__ __forceinline__ void operator()(float, float) const {} };
  template <class F = NoPreload, class E = NoEarly, class A = NoAfter>
  __device__ __forceinline__ bool cache_lookup(uint64_t key, uint32_t group, float& hit_pi, float& hit_v, F&& before_loads = F(), E&& early = E(),
                                               A&& after_hit = A()) const {
    uint32_t sh;
    const CacheView cache = ep.num_groups == 1 ? ar.cache : ar.caches[group];   // one cache per model group
    const int cslot = wave_shard_find<G>(cache, key, lane, &sh, before_loads);
    if (lane == 0) {  // hits / misses / freq; the ghost "reinserts" statistic is not kept for in-round probes
      unsigned long long* st = cache.stats + static_cast<size_t>(sh) * 4;
      if (cslot < 0) {
        atomicAdd(&st[1], 1ULL);
      } else {
        atomicAdd(&st[0], 1ULL);
        // freq = min(freq + 1, 3) with two atomics that return nothing (no round trip); add-then-clamp ends at <= 3 for
        // every interleaving of concurrent probes of the same entry
        uint32_t* f = cache.freq + static_cast<size_t>(sh) * kWaveCap + cslot;
        atomicAdd(f, 1u);
        atomicMin(f, 3u);
      }
    }
    static_assert(M <= G, "one lane per policy entry");
    bool have = early(cslot, hit_pi, hit_v);
    if (!have) {
      if (cslot < 0) return false;
      const float* sp = cache.policy + (static_cast<size_t>(sh) * cache.cap + cslot) * M;
      const float* sv = cache.value + (static_cast<size_t>(sh) * cache.cap + cslot) * (P + 1);
      hit_pi = lane < static_cast<uint32_t>(M) ? sp[lane] : 0.0f;
      hit_v = lane <= static_cast<uint32_t>(P) ? sv[lane] : 0.0f;
      after_hit(hit_pi, hit_v);
    }
    if (lane < static_cast<uint32_t>(M)) ar.pi[static_cast<size_t>(slot) * M + lane] = hit_pi;
    if (lane <= static_cast<uint32_t>(P)) ar.v[static_cast<size_t>(slot) * (P + 1) + lane] = hit_v;
    return true;
  }
};

// One round of PlayManager::play for every slot (play_manager.cc:272-599).
// kPlayout: the instantiation for engines with an EvalType::PLAYOUT seat carries the rollout code; the common one does not
// kMover: the move step of a SPLIT round — the kernel runs over the slots k_sim listed in ar.mover_list (game starts, the
// simulation that completes a search + the move, root leaves: everything rare and register-hungry), beside or behind the
// net launch of the round; a leaf it sends to the net is only written (kSlotQueued), the next round's k_sim lists it.
// round_slot: the round of ONE slot by its lane group (lane = 0 .. GROUP-1); returns the slot state it stored (kSlotWaitEval,
// kSlotQueued, kSlotEnded, kSlotDone) or 0xFF when it had nothing to do.  round_body maps a thread to its slot (kMover: through
// ar.mover_list); the asynchronous pipeline's mover wavefronts (pipeline.hip) call round_slot on the slots their tokens name.
template <class GM, bool kPlayout = false, bool kMover = false>
__device__ __forceinline__ uint32_t round_slot(const EngineParams& ep, const EngineArrays& ar, const uint32_t slot, const uint32_t lane) {
  constexpr int G = GM::GROUP;
  constexpr int P = GM::P;
  if (slot >= ep.S) return 0xFFu;
  if (ar.ctl->stop) return 0xFFu;
  const uint8_t st = ar.sstate[slot];
  // cache_keys[slot] = key of the leaf this slot sends to the net this round (0 = none): the next round's
  // k_cache_insert stores the net's answer under it (PlayManager::update_inferences -> insert_many)
  if (st == kSlotDone || st == kSlotEnded) { if (ep.cache_on && lane == 0) ar.cache_keys[slot] = 0; return 0xFFu; }
  if constexpr (!kMover) {
    if (st == kSlotQueued) {   // a leaf the asynchronous pipeline left unsent (k_pipe_settle after an error): on this round's eval list, as in k_sim
      if (lane == 0) {
        const uint32_t group = ar.leaf_group[slot];
        if (ep.cache_on) ar.cache_keys[slot] = cache_key(ar.leaf_key[slot]);
        ar.eval_list[static_cast<size_t>(group) * ep.S + atomicAdd(&ar.ctl->eval_count[group], 1u)] = slot;
        ar.sstate[slot] = kSlotWaitEval;
      }
      return kSlotWaitEval;
    }
  }
  SlotCtx<GM> c(ep, ar, slot, lane);
  c.trace(100);
  c.load();
  if constexpr (kMover) c.flags &= ~kFlagListed;      // (set by the pipeline's tree wavefronts when they hand a slot to the move step)
  c.trace(101);
  uint32_t inline_sims = 0, insert_key_set = 0;
  PathRegs prec;           // split rounds: the lane image of the simulation this step leaves pending
  bool need_process = (st == kSlotWaitEval);
  if (!need_process) {  // kSlotFresh / kSlotRestart
    c.start_game();
    c.draw_capped();
    c.set_gumbel_target();
    // a game restarted in a slot goes through play_manager.cc:522-546: with tree_reuse off the trees are
    // rebuilt AFTER set_gumbel_num_sims, so the first search of that game has no Gumbel target
    if (ep.gumbel_on && st == kSlotRestart && !ep.tree_reuse) c.set_gumbel_num_sims(c.gs.player, 0);
  }
  bool have_regs = false;        // the pending answer is a cache hit of this round, still in registers
  float reg_pi = 0.0f, reg_v = 0.0f;
  for (;;) {
    if (need_process) {
      const uint32_t cp = c.gs.player;
      const bool noise = c.seat_eps(cp) > 0 && !(c.flags & kFlagCapped);
      c.process_result(cp, (c.flags & kFlagLeafNeedsNet) != 0, noise, have_regs, reg_pi, reg_v);
      have_regs = false;
      c.trace(102);
      const uint32_t goal = (c.flags & kFlagCapped) ? c.seat_cap_visits(cp) : c.seat_visits(cp);
      if (AZMI_SEL(c.t_depth, cp) >= goal) {
        if (c.make_move(cp)) { if (ep.cache_on && lane == 0) ar.cache_keys[slot] = 0; c.store(kSlotEnded); c.trace(107); return kSlotEnded; }
        c.trace(103);
      }
    }
    const uint32_t cp = c.gs.player;
    typename GM::State leaf;
    uint32_t term = 0;
    if (!c.find_leaf(cp, leaf, term, kMover ? &prec : nullptr)) { if (ep.cache_on && lane == 0) ar.cache_keys[slot] = 0; c.store(kSlotDone); return kSlotDone; }
    c.trace(104 | (static_cast<uint64_t>(c.plen) << 8));
    const bool playout = kPlayout && term == 0 && c.seat_eval_playout(cp);   // (a terminal leaf's evaluation is never used)
    const bool needs_net = term == 0 && !c.seat_eval_random(cp) && !playout;
    const uint32_t group = c.seat_group(cp);
    // kFlagLeafNeedsNet = "process_result reads the slot's (v, pi) rows": a net answer or the rollout's
    c.flags = (needs_net || playout) ? (c.flags | kFlagLeafNeedsNet) : (c.flags & ~kFlagLeafNeedsNet);
    if constexpr (kPlayout) { if (playout) c.playout_eval(leaf); }
    if (needs_net) {
      const uint64_t key = GM::key(leaf);
      const bool hit = ep.cache_on && c.cache_lookup(key, group, reg_pi, reg_v);
      c.trace(105 | (static_cast<uint64_t>(hit ? 1 : 0) << 8));
      have_regs = hit;
      if (!hit) {
        c.emit_leaf(leaf, key);      // the planes are only needed when the net is (the cached answer replaces them)
        if (lane == 0) {
          ar.c_evals[slot] += 1;
          ar.leaf_group[slot] = static_cast<uint8_t>(group);
          if constexpr (!kMover) {
            if (ep.cache_on) ar.cache_keys[slot] = cache_key(key);
            ar.eval_list[static_cast<size_t>(group) * ep.S + atomicAdd(&ar.ctl->eval_count[group], 1u)] = slot;
          }
        }
        insert_key_set = kMover ? 2 : 1;
        break;
      }
    }
    need_process = true;
    if (++inline_sims >= ep.max_inline) break;
    if constexpr (kMover) break;     // the move step stops at its first leaf: what follows is k_sim's work, and the step must stay shorter than the net launch it rides in
  }
  if (ep.cache_on && insert_key_set != 1 && lane == 0) ar.cache_keys[slot] = 0;
  if constexpr (kMover) {
    if (prec.ok) { c.store_pend(prec); c.flags |= kFlagPendRec; } else c.flags &= ~kFlagPendRec;
  }
  c.store(insert_key_set == 2 ? kSlotQueued : kSlotWaitEval);
  c.trace(106);
  return insert_key_set == 2 ? kSlotQueued : kSlotWaitEval;
}
template <class GM, bool kPlayout = false, bool kMover = false>
__device__ __forceinline__ void round_body(const EngineParams& ep, const EngineArrays& ar, const uint32_t gtid) {
  constexpr int G = GM::GROUP;
  uint32_t slot = gtid / G;
  const uint32_t lane = gtid % G;
  if constexpr (kMover) {
    if (slot >= ar.ctl->mover_count) return;
    slot = ar.mover_list[slot];
  }
  (void)round_slot<GM, kPlayout, kMover>(ep, ar, slot, lane);
}
template <class GM, bool kPlayout = false, bool kMover = false>
__global__ __launch_bounds__(256, 1) void k_round(EngineParams ep, EngineArrays ar) {
  round_body<GM, kPlayout, kMover>(ep, ar, blockIdx.x * blockDim.x + threadIdx.x);
}

// The per-simulation kernel of a SPLIT round (lane-group engine, plain PUCT seats): backup of the pending evaluation, descent,
// expansion, position-cache probe, leaf hand-off — up to max_inline times — for every slot whose next step is an ordinary
// simulation.  Everything else (a game start, the simulation that completes a search and the move behind it, a leaf that
// is the root: temperature / Dirichlet noise) is left untouched and listed for the move step (k_round<kMover>).  Without
// that code the kernel needs half the registers of k_round, so two to four times as many waves share a SIMD, and no
// wave waits for a neighbour slot's 20 us move.
template <class GM>
__global__ __launch_bounds__(256, 2) void k_sim(EngineParams ep, EngineArrays ar) {
  constexpr int G = GM::GROUP;
  constexpr int P = GM::P;
  const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t slot = gtid / G, lane = gtid % G;
  if (slot >= ep.S) return;
  if (ar.ctl->stop) return;
  const uint8_t st = ar.sstate[slot];
  if (st == kSlotDone || st == kSlotEnded) { if (ep.cache_on && lane == 0) ar.cache_keys[slot] = 0; return; }
  if (st == kSlotQueued) {     // the move step's leaf: on this round's eval list, its answer is cached next round
    if (lane == 0) {
      const uint32_t group = ar.leaf_group[slot];
      if (ep.cache_on) ar.cache_keys[slot] = cache_key(ar.leaf_key[slot]);
      ar.eval_list[static_cast<size_t>(group) * ep.S + atomicAdd(&ar.ctl->eval_count[group], 1u)] = slot;
      ar.sstate[slot] = kSlotWaitEval;
    }
    return;
  }
  auto defer = [&]() {
    if (lane == 0) {
      if (ep.cache_on) ar.cache_keys[slot] = 0;
      ar.mover_list[atomicAdd(&ar.ctl->mover_count, 1u)] = slot;
    }
  };
  if (st != kSlotWaitEval) { defer(); return; }          // kSlotFresh / kSlotRestart: a game start
  SlotCtx<GM> c(ep, ar, slot, lane);
  const uint64_t t_start = ep.sim_budget ? wall_clock64() : 0;
  c.trace(100);
  c.load();
  static_assert(P == 2 && G == 8, "the register-resident path below is written for two players and 8-lane groups");
  const uint32_t cp = c.gs.player;                 // no move happens in this kernel: one tree, one player to move
  const size_t tb = c.tree_base(cp);
  const uint32_t root = AZMI_SEL(c.t_root, cp);
  const uint32_t goal = (c.flags & kFlagCapped) ? c.seat_cap_visits(cp) : c.seat_visits(cp);
  // the next backup completes the search (a move follows), the evaluated leaf is the root (temperature, noise), or the
  // pending simulation has no lane image (a path deeper than 8 levels): the move step's business
  if (!(c.flags & kFlagPendRec) || AZMI_SEL(c.t_depth, cp) + 1 >= goal || c.cur == root) { defer(); return; }
  const float fpu_root = c.seat_fpu_zero(cp) ? 0.0f : ep.fpu_reduction;
  uint32_t* const path = ar.path + static_cast<size_t>(slot) * ep.max_depth;
  // ---- the pending simulation in registers: level i <-> lane i: the node chosen at level i as the descent saw it (n, q,
  // d, v) and the player to move at its parent; lf_*: the evaluated leaf after its expansion (children in lanes).  The
  // backup works on these, stores its results without waiting and FORWARDS them (fw_*, fl_*) to the next descent, which
  // patches whatever it loads for those nodes.  (Only the immediately preceding simulation can have stores in flight:
  // every descent waits for loads it issued behind them.)  One round trip at the start of the round brings in the
  // slot's state, the image of the simulation left pending last round (ar.pend) and its (v, pi) rows.
  const PendRec pr_in = ar.pend[static_cast<size_t>(slot) * G + lane];
  float reg_pi = 0.0f, reg_v = 0.0f;   // lane m: pi[m]; lane i <= P: v[i] (the net's rows, or a cache hit of this round)
  if (c.flags & kFlagLeafNeedsNet) {
    if (lane < static_cast<uint32_t>(GM::M)) reg_pi = ar.pi[static_cast<size_t>(slot) * GM::M + lane];
    if (lane <= static_cast<uint32_t>(P)) reg_v = ar.v[static_cast<size_t>(slot) * (P + 1) + lane];
  }
  uint32_t root_n; float root_v; uint64_t root_meta;
  { const NodeRec* rr = ar.nodes + tb + root; root_n = rr->n; root_v = rr->v; root_meta = rr->meta; }
  uint32_t lv_node = pr_in.node, lv_n = pr_in.n, lv_pp = pr_in.pp_mv & 0xFFu;
  float lv_q = pr_in.q, lv_d = pr_in.d, lv_v = pr_in.v;
  uint64_t lf_meta = pr_in.leaf_meta;
  uint32_t lf_mv = pr_in.pp_mv >> 8;
  uint32_t lf_c0 = meta_ch0(lf_meta), lf_k = meta_nch(lf_meta), lf_term = meta_term(lf_meta), lf_player = meta_player(lf_meta);
  bool fw = false;               // fw_* / fl_* describe the simulation backed up last (in this round)
  uint32_t fw_node = 0, fw_n = 0, fw_plen = 0;
  float fw_q = 0.0f, fw_d = 0.0f, fw_v = 0.0f;
  uint32_t fl_node = 0xFFFFFFFFu, fl_mv = 0;
  uint64_t fl_meta = 0;
  float fl_pr = 0.0f;
  uint32_t inline_sims = 0, insert_key_set = 0, sims_done = 0;
  bool rec_ok = true;
  c.trace(101);
  for (;;) {
    if (AZMI_SEL(c.t_depth, cp) + 1 >= goal || c.cur == root || !rec_ok) {   // (never true in the first pass: checked above)
      if (ep.cache_on && lane == 0) ar.cache_keys[slot] = 0;
      if (lane == 0) ar.mover_list[atomicAdd(&ar.ctl->mover_count, 1u)] = slot;
      break;
    }
    {
      // ---- MCTS::process_result (mcts.cc:500-555) on the lane-resident path
      const bool from_net = (c.flags & kFlagLeafNeedsNet) != 0;
      float val[P + 1];
      if (lf_term != 0) {
#pragma unroll
        for (int i = 0; i <= P; ++i) val[i] = (static_cast<int>(lf_term) - 1 == i) ? 1.0f : 0.0f;
      } else {
        float p = 0.0f;
        if (from_net) {
#pragma unroll
          for (int i = 0; i <= P; ++i) val[i] = c.bcast(reg_v, i);
          p = c.bcast(reg_pi, static_cast<int>(lf_mv));
          if (lane >= lf_k) p = 0.0f;
        } else {                  // dumb_eval: uniform over legal moves, u8 sum wraps (game_state.h:160-173)
#pragma unroll
          for (int i = 0; i <= P; ++i) val[i] = static_cast<float>(1.0 / (P + 1));
          const float ksum = static_cast<float>(lf_k & 0xFFu);
          if (lane < lf_k) p = (ksum == 0.0f) ? 0.0f : 1.0f / ksum;
        }
        const float sum = c.seqsum8(lane < lf_k ? p : 0.0f);
        p = p / sum;
        if (lane < lf_k) ar.nodes[tb + lf_c0 + lane].pr = p;
        fl_pr = p;
      }
      fl_node = c.cur; fl_mv = lf_mv; fl_meta = lf_meta;
      const float draw_share = val[P] / static_cast<int32_t>(P);
      const uint32_t plen = c.plen;
      float nq = 0.0f, nd = 0.0f, nv = lv_v;
      if (lane < plen) {
        const float vv = ((lv_pp == 0) ? val[0] : val[1]) + draw_share;
        nq = (lv_q * static_cast<float>(lv_n) + vv) / static_cast<float>(lv_n + 1);
        nd = (lv_d * static_cast<float>(lv_n) + val[P]) / static_cast<float>(lv_n + 1);
        NodeRec* nr = ar.nodes + tb + lv_node;
        nr->q = nq; nr->d = nd;
        if (lv_n == 0) { nv = ((lf_player == 0) ? val[0] : val[1]) + draw_share; nr->v = nv; }   // only the leaf can be a first visit
        nr->n = lv_n + 1;
      }
      fw_node = lv_node; fw_n = lv_n + 1; fw_q = nq; fw_d = nd; fw_v = nv; fw_plen = plen; fw = true;
      root_n += 1;
      if (lane == 0) ar.nodes[tb + root].n = root_n;
#pragma unroll
      for (int p = 0; p < P; ++p) if (static_cast<uint32_t>(p) == cp) c.t_depth[p] += 1;
      sims_done += 1;
    }
    c.trace(102);
    // ---- MCTS::find_leaf (mcts.cc:462-498), plain PUCT, with the forwarded values patched in
    typename GM::State leaf = c.gs;
    uint32_t cur = root, plen = 0, n = root_n;
    uint64_t meta = root_meta;
    float v_cur = root_v;
    bool prefix = true;
    while (n > 0 && meta_term(meta) == 0) {
      if (plen >= ep.max_depth) { c.raise(8u); if (ep.cache_on && lane == 0) ar.cache_keys[slot] = 0; c.store(kSlotDone); return; }
      if (lane == 0) path[plen] = cur;
      const uint32_t k = meta_nch(meta), c0 = meta_ch0(meta);
      if (k == 0) { c.raise(8u); if (ep.cache_on && lane == 0) ar.cache_keys[slot] = 0; c.store(kSlotDone); return; }
      uint32_t n_l = 0; float q_l = 0.0f, p_l = 0.0f, d_l = 0.0f, v_l = 0.0f; uint64_t m_l = 0;
      if (cur == fl_node) {                 // the children of the leaf evaluated a moment ago: all in registers
        if (lane < k) { p_l = fl_pr; m_l = meta_pack(0, 0, fl_mv, 0, 0); }
      } else {
        if (lane < k) {
          const NodeRec* cr = ar.nodes + tb + c0 + lane;
          n_l = cr->n; q_l = cr->q; p_l = cr->pr; d_l = cr->d; v_l = cr->v; m_l = cr->meta;
        }
        if (prefix && plen < fw_plen && plen < 8u) {   // one child of this node was updated by the last backup
          const uint32_t t_node = c.bcast(fw_node, static_cast<int>(plen));
          const uint32_t t_n = c.bcast(fw_n, static_cast<int>(plen));
          const float t_q = c.bcast(fw_q, static_cast<int>(plen)), t_d = c.bcast(fw_d, static_cast<int>(plen)), t_v = c.bcast(fw_v, static_cast<int>(plen));
          if (c0 + lane == t_node) { n_l = t_n; q_l = t_q; d_l = t_d; v_l = t_v; if (t_node == fl_node) m_l = fl_meta; }
        }
      }
      const float fpu = (cur == root) ? fpu_root : ep.fpu_reduction;
      const uint32_t best = c.select_child(k, n_l, q_l, p_l, v_cur, n, fpu);
      const uint32_t nxt = c0 + best;
      const uint32_t s_n = c.bcast(n_l, static_cast<int>(best));
      const float s_q = c.bcast(q_l, static_cast<int>(best)), s_d = c.bcast(d_l, static_cast<int>(best)), s_v = c.bcast(v_l, static_cast<int>(best));
      const uint64_t s_m = c.bcast(m_l, static_cast<int>(best));
      if (plen < 8u) {
        if (lane == plen) { lv_node = nxt; lv_n = s_n; lv_q = s_q; lv_d = s_d; lv_v = s_v; lv_pp = meta_player(meta); }
        prefix = prefix && plen < fw_plen && nxt == c.bcast(fw_node, static_cast<int>(plen));
      } else {
        prefix = false;
      }
      cur = nxt; n = s_n; meta = s_m; v_cur = s_v;
      GM::play(leaf, meta_mv(meta));
      ++plen;
      c.trace(108);
    }
    c.cur = cur; c.plen = plen;
    rec_ok = plen <= 8u;
#pragma unroll
    for (int p = 0; p < P; ++p) if (static_cast<uint32_t>(p) == cp) c.t_tld[p] += plen;
    uint32_t term = meta_term(meta);
    lf_k = meta_nch(meta); lf_c0 = meta_ch0(meta); lf_mv = 0; lf_meta = meta;
    if (n == 0) {
      term = GM::terminal(leaf);
      const uint64_t keep = meta_pack(0, 0, meta_mv(meta), leaf.player, term);
      if (!c.expand_node(cp, cur, leaf, keep, lf_c0, lf_k, &lf_mv)) { if (ep.cache_on && lane == 0) ar.cache_keys[slot] = 0; c.store(kSlotDone); return; }
      lf_meta = meta_pack(lf_c0, lf_k, meta_mv(meta), leaf.player, term);
    }
    lf_term = term; lf_player = leaf.player;
    c.trace(109);
    const bool needs_net = term == 0 && !c.seat_eval_random(cp);
    const uint32_t group = c.seat_group(cp);
    c.flags = needs_net ? (c.flags | kFlagLeafNeedsNet) : (c.flags & ~kFlagLeafNeedsNet);
    if (needs_net) {
      const uint64_t key = GM::key(leaf);
      const bool hit = ep.cache_on && c.cache_lookup(key, group, reg_pi, reg_v);
      c.trace(105 | (static_cast<uint64_t>(hit ? 1 : 0) << 8));
      if (!hit) {
        c.emit_leaf(leaf, key);
        if (lane == 0) {
          ar.c_evals[slot] += 1;
          if (ep.cache_on) ar.cache_keys[slot] = cache_key(key);
          ar.leaf_group[slot] = static_cast<uint8_t>(group);
          ar.eval_list[static_cast<size_t>(group) * ep.S + atomicAdd(&ar.ctl->eval_count[group], 1u)] = slot;
        }
        insert_key_set = 1;
        break;
      }
    }
    // the answer of this leaf is at hand (terminal, RANDOM evaluator, cache hit): the slot goes on, unless the round's
    // budget is used up - in simulations, or in time: the kernel lasts as long as its slowest wave, and the results do
    // not depend on where a round ends
    if (++inline_sims >= ep.max_inline || (ep.sim_budget && static_cast<uint32_t>(wall_clock64() - t_start) > ep.sim_budget)) {
      if (ep.cache_on && lane == 0) ar.cache_keys[slot] = 0;
      break;
    }
  }
  (void)insert_key_set;
  if (lane == 0 && sims_done) ar.c_sims[slot] += sims_done;
  // the pending simulation's lane image for the next round (or for nobody, when the path was too deep: the move step reads memory)
  if (rec_ok) {
    PathRegs r;
    r.node = lv_node; r.n = lv_n; r.q = lv_q; r.d = lv_d; r.v = lv_v; r.pp = lv_pp; r.mv = lf_mv; r.leaf_meta = lf_meta;
    c.store_pend(r);
    c.flags |= kFlagPendRec;
  } else {
    c.flags &= ~kFlagPendRec;
  }
  c.store(kSlotWaitEval);
  c.trace(106);
}

// Deterministic restart / retire of the slots whose game ended in the previous round:
// slots are served in slot order against games_started (play_manager.cc:506-513).
__device__ __forceinline__ void assign_body(const EngineParams& ep, const EngineArrays& ar, uint32_t count_round) {
  Control* ctl = ar.ctl;
  __shared__ uint32_t s_n, s_base;
  if (threadIdx.x == 0) { s_n = ctl->ended_count; s_base = ctl->games_started; }
  __syncthreads();
  const uint32_t n = s_n, base = s_base;
  if (threadIdx.x == 0 && count_round) {
    ctl->rounds += 1;
    ctl->eval_count[0] = 0; ctl->eval_count[1] = 0; ctl->eval_count[2] = 0; ctl->eval_count[3] = 0;
    ctl->mover_count = 0;
  }
  if (n == 0) return;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t s = ar.ended_list[i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n; ++j) rank += (ar.ended_list[j] < s) ? 1u : 0u;
    ar.sstate[s] = (base + rank < ep.games_to_play) ? kSlotRestart : kSlotDone;
    ar.perm[s] = (base + rank) % ep.num_perms;   // game.perm_index = games_started_ % seat_perms_.size(), play_manager.cc:511
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t room = base < ep.games_to_play ? ep.games_to_play - base : 0u;
    const uint32_t restarted = n < room ? n : room;
    ctl->games_started = base + restarted;
    ctl->games_completed += n;
    ctl->live_slots -= (n - restarted);
    ctl->ended_count = 0;
    if (ctl->games_completed >= ep.games_to_play || ctl->live_slots == 0) ctl->stop = 1;
  }
}

#ifndef AZMI_KERNELS_NO_ASSIGN   // (a plain, non-template kernel: one translation unit of the library defines it)
__global__ void k_assign(EngineParams ep, EngineArrays ar, uint32_t count_round) { assign_body(ep, ar, count_round); }
#endif

// Start of a round with the position cache on: the leaves the net evaluated in the previous round go into
// the cache (PlayManager::update_inferences -> insert_many, play_manager.cc:631-640; one wave per leaf, batch
// order per shard, keys left by k_round), and — in the same launch, they are independent — the extra last block
// does the restart / retire bookkeeping.
template <class GM>
__global__ __launch_bounds__(256) void k_cache_insert(EngineParams ep, EngineArrays ar, const uint64_t* keys, uint32_t off, uint32_t n,
                                                      uint32_t assign_block, uint32_t count_round, uint32_t group) {
  if (blockIdx.x == assign_block) { assign_body(ep, ar, count_round); return; }
  __shared__ uint32_t s_sid[kApplyMax];
  const CacheView cache = ep.num_groups == 1 ? ar.cache : ar.caches[group];
  cache_apply_batch(cache, keys + off, ar.pi + static_cast<size_t>(off) * GM::M, ar.v + static_cast<size_t>(off) * (GM::P + 1), n, s_sid,
                    ep.num_groups == 1 ? nullptr : ar.leaf_group + off, group);
}

}  // namespace azmi
