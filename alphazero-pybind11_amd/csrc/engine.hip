// Host side of libazmi.so: owns the HBM arrays of one PlayManager, launches the
// round kernels, and implements the C ABI declared in include/azmi.h.
// There is deliberately no CPU code path: every entry point that computes
// requires a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "../../include/azmi.h"
#include <atomic>
#include <mutex>

#include "cache_host.h"
#include "engine_host.h"
#include "engine_kernels.h"
#include "leafnet_c4.h"
#include "engine_kernels_big.h"
#include "mcts_object_kernels.h"

using namespace azmi;

namespace {

thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
}  // namespace
int azmi_host_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
namespace {
#define HIP_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(e_ == hipErrorOutOfMemory ? AZMI_ERR_OOM : AZMI_ERR_NO_DEVICE, "%s: %s", #expr, \
                  hipGetErrorString(e_));                                                 \
  } while (0)

bool game_info(int game, GameInfo* gi) {
  switch (game) {
    case AZMI_GAME_CONNECT4:
      *gi = GameInfo{Connect4::P, Connect4::M, Connect4::C, Connect4::H, Connect4::W, Connect4::MAXK,
                     Connect4::MAX_TURNS, 3, Connect4::MAXK};
      return true;
    case AZMI_GAME_TAWLBWRDD:
      *gi = GameInfo{Tawlbwrdd::P, Tawlbwrdd::M, Tawlbwrdd::C, Tawlbwrdd::H, Tawlbwrdd::W, Tawlbwrdd::MAXK,
                     Tawlbwrdd::MAX_TURNS, Tawlbwrdd::STATE_WORDS, 160};
      return true;
    case AZMI_GAME_BRANDUBH:
      *gi = GameInfo{Brandubh::P, Brandubh::M, Brandubh::C, Brandubh::H, Brandubh::W, Brandubh::MAXK,
                     Brandubh::MAX_TURNS, Brandubh::STATE_WORDS, 48};
      return true;
    case AZMI_GAME_OPENTAFL:
      *gi = GameInfo{OpenTafl::P, OpenTafl::M, OpenTafl::C, OpenTafl::H, OpenTafl::W, OpenTafl::MAXK,
                     OpenTafl::MAX_TURNS, OpenTafl::STATE_WORDS, 160};
      return true;
    case AZMI_GAME_STARGAMBIT:   // max_turns = the engine's bound on the ACTIONS of a game (dev_stargambit.h)
      *gi = GameInfo{StarGambit::P, StarGambit::M, StarGambit::C, StarGambit::H, StarGambit::W, StarGambit::MAXK,
                     StarGambit::MAX_TURNS, StarGambit::STATE_WORDS, 32};
      return true;
    default:
      return false;
  }
}

}  // namespace

namespace {

__global__ void k_seed(EngineArrays ar, uint32_t S, uint64_t seed) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  Pcg32 g;
  const uint64_t sd = slot_seed(seed, s);
  g.seed(sd);
  ar.rng[s] = g.state;
  g.seed(sd ^ kCoinSalt);
  ar.coin[s] = g.state;
  g.seed(sd ^ kRollSalt);
  ar.roll[s] = g.state;
}

// start of a round: cache insert of last round's leaves + the restart/retire bookkeeping in one launch
template <class GM>
void launch_pre_round(azmi_pm* pm, hipStream_t st) {
  if (!pm->ep.cache_on) { k_assign<<<1, 256, 0, st>>>(pm->ep, pm->ar, 1u); return; }
  for (uint32_t g = 0; g < pm->ep.num_groups; ++g)
    for (uint32_t off = 0; off < pm->ep.S; off += kApplyMax) {
      const uint32_t m = std::min<uint32_t>(kApplyMax, pm->ep.S - off);
      const uint32_t nb = (m + 3) / 4;
      const bool with_assign = g == 0 && off == 0;
      k_cache_insert<GM><<<nb + (with_assign ? 1u : 0u), 256, 0, st>>>(pm->ep, pm->ar, pm->ar.cache_keys, off, m,
                                                                      with_assign ? nb : 0xFFFFFFFFu, 1u, g);
    }
}

// defer_moves: split rounds only - leave the move step to the fused net launch that follows (launch_net_move)
int launch_round(azmi_pm* pm, hipStream_t st, bool defer_moves = false) {
  const uint32_t threads = 256;
  switch (pm->game) {
    case AZMI_GAME_CONNECT4: {
      launch_pre_round<Connect4>(pm, st);
      const uint32_t slots_per_block = threads / Connect4::GROUP;
      const uint32_t blocks = (pm->ep.S + slots_per_block - 1) / slots_per_block;
      if (pm->any_playout) k_round<Connect4, true><<<blocks, threads, 0, st>>>(pm->ep, pm->ar);
      else if (pm->split_rounds) {
        // split round: the lean per-simulation kernel over every slot, then the move step over the slots it listed
        k_sim<Connect4><<<blocks, threads, 0, st>>>(pm->ep, pm->ar);
        if (!defer_moves) k_round<Connect4, false, true><<<blocks, threads, 0, st>>>(pm->ep, pm->ar);
      }
      else k_round<Connect4><<<blocks, threads, 0, st>>>(pm->ep, pm->ar);
      break;
    }
    case AZMI_GAME_TAWLBWRDD:
      launch_pre_round<Tawlbwrdd>(pm, st);
      if (pm->any_playout) k_round_big<Tawlbwrdd, true><<<pm->ep.S, 64, 0, st>>>(pm->ep, pm->ar);
      else if (pm->big_split) { k_round_big_sim<Tawlbwrdd><<<pm->ep.S, 64, 0, st>>>(pm->ep, pm->ar); k_round_big_move<Tawlbwrdd><<<pm->ep.S, 64, 0, st>>>(pm->ep, pm->ar); }
      else k_round_big_o2<Tawlbwrdd><<<pm->ep.S, 64, 0, st>>>(pm->ep, pm->ar);
      if (pm->ep.half_nodes) k_compact<Tawlbwrdd><<<std::min(pm->ep.S * pm->gi.P, kCompactBlocks), 256, 0, st>>>(pm->ep, pm->ar, pm->ep.S * pm->gi.P);
      break;
    case AZMI_GAME_BRANDUBH:
      launch_pre_round<Brandubh>(pm, st);
      if (pm->any_playout) k_round_big<Brandubh, true><<<pm->ep.S, 64, 0, st>>>(pm->ep, pm->ar);
      else if (pm->big_split) { k_round_big_sim<Brandubh><<<pm->ep.S, 64, 0, st>>>(pm->ep, pm->ar); k_round_big_move<Brandubh><<<pm->ep.S, 64, 0, st>>>(pm->ep, pm->ar); }
      else k_round_big_o2<Brandubh><<<pm->ep.S, 64, 0, st>>>(pm->ep, pm->ar);
      if (pm->ep.half_nodes) k_compact<Brandubh><<<std::min(pm->ep.S * pm->gi.P, kCompactBlocks), 256, 0, st>>>(pm->ep, pm->ar, pm->ep.S * pm->gi.P);
      break;
    case AZMI_GAME_OPENTAFL:
      launch_pre_round<OpenTafl>(pm, st);
      if (pm->any_playout) k_round_big<OpenTafl, true><<<pm->ep.S, 64, 0, st>>>(pm->ep, pm->ar);
      else if (pm->big_split) { k_round_big_sim<OpenTafl><<<pm->ep.S, 64, 0, st>>>(pm->ep, pm->ar); k_round_big_move<OpenTafl><<<pm->ep.S, 64, 0, st>>>(pm->ep, pm->ar); }
      else k_round_big_o2<OpenTafl><<<pm->ep.S, 64, 0, st>>>(pm->ep, pm->ar);
      if (pm->ep.half_nodes) k_compact<OpenTafl><<<std::min(pm->ep.S * pm->gi.P, kCompactBlocks), 256, 0, st>>>(pm->ep, pm->ar, pm->ep.S * pm->gi.P);
      break;
    case AZMI_GAME_STARGAMBIT:
      launch_pre_round<StarGambit>(pm, st);
      if (pm->any_playout) k_round_big<StarGambit, true><<<pm->ep.S, 64, 0, st>>>(pm->ep, pm->ar);
      else if (pm->big_split) { k_round_big_sim1<StarGambit><<<pm->ep.S, 64, 0, st>>>(pm->ep, pm->ar); k_round_big_move<StarGambit><<<pm->ep.S, 64, 0, st>>>(pm->ep, pm->ar); }
      else k_round_big<StarGambit><<<pm->ep.S, 64, 0, st>>>(pm->ep, pm->ar);
      if (pm->ep.half_nodes) k_compact<StarGambit><<<std::min(pm->ep.S * pm->gi.P, kCompactBlocks), 256, 0, st>>>(pm->ep, pm->ar, pm->ep.S * pm->gi.P);
      break;
    default:
      return fail(AZMI_ERR_INVALID, "game %d has no device kernels", pm->game);
  }
  HIP_TRY(hipGetLastError());
  return AZMI_OK;
}

// The net + move-step launch of a split round (Connect4 engine, one model group): workgroups [0, net_tiles) carry the tiles
// of the round's eval list through the leaf net (leafnet_c4.h), the workgroups behind them run the round's move step
// (round_body<kMover> over ar.mover_list).  The two are independent - a listed mover was not evaluated this round, its
// new leaf is only queued for the next one - so the 20-60 us move chains hide behind the net instead of stretching the
// tree kernel of every round.
__global__ __launch_bounds__(256, 2) void k_net_move(azmi_net_dev::NetDesc nd, azmi_net_dev::NetPtrs np, EngineParams ep, EngineArrays ar,
                                                     uint32_t net_tiles) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_fused[];
  if (blockIdx.x < net_tiles)
    azmi_net_dev::c4::tile<azmi_net_dev::c4::TileSmall, 4, 4, 16>(nd, np, ar.canon, ar.v, ar.pi, ep.S, ar.eval_list, &ar.ctl->eval_count[0], blockIdx.x, lds_fused);
  else
    round_body<Connect4, false, true>(ep, ar, (blockIdx.x - net_tiles) * blockDim.x + threadIdx.x);
}

}  // namespace
// the two between-epoch steps of the asynchronous pipeline (pipeline.hip): the move step of a split round over the slots the
// tree side listed, as a launch of its own, and the restart / retire bookkeeping
unsigned long long azmi_host_pipe_l0_hits(azmi_pm* pm, hipStream_t st);    // pipeline.hip
int azmi_host_launch_move_step(azmi_pm* pm, hipStream_t st) {
  const uint32_t blocks = (pm->ep.S * Connect4::GROUP + 255u) / 256u;
  k_round<Connect4, false, true><<<blocks, 256, 0, st>>>(pm->ep, pm->ar);
  HIP_TRY(hipGetLastError());
  return AZMI_OK;
}
int azmi_host_launch_assign(azmi_pm* pm, hipStream_t st, uint32_t count_round) {
  k_assign<<<1, 256, 0, st>>>(pm->ep, pm->ar, count_round);
  HIP_TRY(hipGetLastError());
  return AZMI_OK;
}
namespace {

int read_ctl(azmi_pm* pm, hipStream_t st, Control* out, bool settle) {
  if (settle) k_assign<<<1, 256, 0, st>>>(pm->ep, pm->ar, 0u);
  HIP_TRY(hipMemcpyAsync(out, pm->ar.ctl, sizeof(Control), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (out->overflow)
    return fail(AZMI_ERR_OVERFLOW, "device engine stopped: overflow mask 0x%x (1 tree arena, 2 history, 4 move log, "
                "8 path/children, 16 pick_move, 32 unknown move, 64 illegal move)", out->overflow);
  return AZMI_OK;
}

template <class T>
int d2h(std::vector<T>& dst, const T* src, size_t n, hipStream_t st) {
  dst.resize(n);
  HIP_TRY(hipMemcpyAsync(dst.data(), src, n * sizeof(T), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return AZMI_OK;
}

// ---- batched rules replay (parity tier T0) --------------------------------------------------------
template <class GM>
__global__ void k_replay(const uint8_t* init, const int32_t* moves, uint32_t n, uint32_t len, uint8_t* valid, float* scores,
                         float* canonical, uint32_t* player, uint32_t* turn, uint64_t* key, int32_t* status) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  typename GM::State s = GM::initial();
  if (init) s = GM::from_bytes(init + static_cast<size_t>(g) * GM::SERIALIZED);
  int32_t stt = 0;
  for (uint32_t i = 0; i < len; ++i) {
    const int32_t mv = moves[static_cast<size_t>(g) * len + i];
    if (mv < 0) break;
    if (mv >= GM::M || !((GM::valid_mask(s) >> mv) & 1u) || !GM::play(s, static_cast<uint32_t>(mv))) { stt = -1; break; }
  }
  if (status) status[g] = stt;
  if (valid) for (int m = 0; m < GM::M; ++m) valid[static_cast<size_t>(g) * GM::M + m] = (GM::valid_mask(s) >> m) & 1u;
  if (scores) {
    const uint32_t t = GM::terminal(s);
    for (int i = 0; i <= GM::P; ++i)
      scores[static_cast<size_t>(g) * (GM::P + 1) + i] = t == 0 ? -1.0f : (static_cast<int>(t) - 1 == i ? 1.0f : 0.0f);
  }
  if (canonical) for (int e = 0; e < GM::CANON; ++e) canonical[static_cast<size_t>(g) * GM::CANON + e] = GM::canonical_at(s, e);
  if (player) player[g] = s.player;
  if (turn) turn[g] = s.turn;
  if (key) key[g] = GM::key(s);
}

// playout_eval / playout_eval_batch (game_state.cc:10-95) for a batch of states given as start position + move list: one
// thread per state, its rollout stream seeded with seeds[g]
template <class GM>
__global__ void k_playout(const uint8_t* init, const int32_t* moves, uint32_t n, uint32_t len, const uint64_t* seeds, float* v, float* pi,
                          int32_t* status) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  typename GM::State s = GM::initial();
  if (init) s = GM::from_bytes(init + static_cast<size_t>(g) * GM::SERIALIZED);
  int32_t stt = 0;
  for (uint32_t i = 0; i < len; ++i) {
    const int32_t mv = moves[static_cast<size_t>(g) * len + i];
    if (mv < 0) break;
    if (mv >= GM::M || !((GM::valid_mask(s) >> mv) & 1u) || !GM::play(s, static_cast<uint32_t>(mv))) { stt = -1; break; }
  }
  status[g] = stt;
  if (stt) return;
  const uint32_t kl = GM::num_valid(s);
  const float ksum = static_cast<float>(kl & 0xFFu);
  for (int m = 0; m < GM::M; ++m)
    pi[static_cast<size_t>(g) * GM::M + m] = (((GM::valid_mask(s) >> m) & 1u) && ksum > 0.0f) ? 1.0f / ksum : 0.0f;
  Pcg32 roll;
  roll.seed(seeds[g]);
  uint32_t term = GM::terminal(s);
  while (term == 0) {
    const uint32_t k = GM::num_valid(s);
    if (k == 0) break;
    GM::play(s, GM::nth_valid(s, lemire_below(roll, k)));
    term = GM::terminal(s);
  }
  for (int i = 0; i <= GM::P; ++i)
    v[static_cast<size_t>(g) * (GM::P + 1) + i] = term ? ((static_cast<int>(term) - 1 == i) ? 1.0f : 0.0f) : static_cast<float>(1.0 / (GM::P + 1));
}

// StarGambit replay / rollout: ONE WAVEFRONT per game (its rules are wave-cooperative, dev_stargambit.h); the position
// history of game g lives in row g of `hist` (hist_stride entries)
struct SgListRep {
  uint64_t* list; uint32_t& len; uint32_t cap; uint32_t lane; bool overflow = false;
  __device__ __forceinline__ void clear() { len = 0; }
  __device__ __forceinline__ uint32_t push(unsigned long long k) {
    uint32_t cnt = 0;
    for (uint32_t i = lane; i < len; i += 64) cnt += list[i] == k;
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (len >= cap) { overflow = true; return cnt + 1; }
    if (lane == 0) list[len] = k;
    ++len;
    StarGambit::lds_sync();
    return cnt + 1;
  }
};
__device__ __forceinline__ bool sg_start(const uint8_t* init, uint32_t stride, uint32_t g, uint32_t lane, StarGambit::State& s, uint64_t* hist,
                                         uint32_t& nh, uint32_t cap) {
  if (!init) {
    s = StarGambit::initial(0, lane);
    const unsigned long long h0 = StarGambit::position_hash(s);
    if (lane == 0) hist[0] = h0;
    nh = 1;
    StarGambit::lds_sync();
    return true;
  }
  const uint8_t* row = init + static_cast<size_t>(g) * stride;
  const uint32_t inner = uint32_t(row[21]) | uint32_t(row[22]) << 8 | uint32_t(row[23]) << 16 | uint32_t(row[24]) << 24;
  return sg_parse_image(row, 25u + inner, lane, s, hist, nh, cap);
}
// flags bit 0: play_move as the reference does (no validity check; a move that names no unit is ignored)
__global__ __launch_bounds__(64) void k_replay_sg(const uint8_t* init, uint32_t init_stride, const int32_t* moves, uint32_t n, uint32_t len,
                                                   uint64_t* hist, uint32_t hist_stride, uint8_t* valid, float* scores, float* canonical,
                                                   uint32_t* player, uint32_t* turn, uint64_t* key, int32_t* status, uint32_t flags) {
  using G = StarGambit;
  __shared__ SgScratch sm;
  const uint32_t g = blockIdx.x, lane = threadIdx.x;
  if (g >= n) return;
  G::State s;
  uint64_t* hl = hist + static_cast<size_t>(g) * hist_stride;
  uint32_t nh = 0;
  int32_t stt = sg_start(init, init_stride, g, lane, s, hl, nh, hist_stride) ? 0 : -1;
  SgListRep rep{hl, nh, hist_stride, lane};
  for (uint32_t i = 0; i < len && stt == 0; ++i) {
    const int32_t mv = moves[static_cast<size_t>(g) * len + i];
    if (mv < 0) break;
    if (mv >= G::M) { stt = -1; break; }
    if (!(flags & 1u)) {
      G::gen_valid(s, lane, sm);
      const bool ok = G::is_valid_bit(sm, static_cast<uint32_t>(mv));
      G::lds_sync();
      if (!ok) { stt = -1; break; }
    }
    G::apply_move(s, static_cast<uint32_t>(mv), lane, sm, rep);
    if (rep.overflow) { stt = -1; break; }
  }
  if (status && lane == 0) status[g] = stt;
  if (valid) {
    G::gen_valid(s, lane, sm);
    for (uint32_t m = lane; m < static_cast<uint32_t>(G::M); m += 64) valid[static_cast<size_t>(g) * G::M + m] = G::is_valid_bit(sm, m) ? 1 : 0;
    G::lds_sync();
  }
  if (scores && lane <= static_cast<uint32_t>(G::P)) {
    const uint32_t t = G::terminal(s);
    // over with no winner recorded (only reachable through a hand-made image): all zeros, like the reference's scores()
    scores[static_cast<size_t>(g) * (G::P + 1) + lane] = t == 0 ? -1.0f : ((G::winner(s) < 3 && t - 1 == lane) ? 1.0f : 0.0f);
  }
  if (canonical) G::write_canonical(s, canonical + static_cast<size_t>(g) * G::CANON, lane, sm);
  const uint64_t k = G::key(s, lane);
  if (lane == 0) {
    if (player) player[g] = s.player;
    if (turn) turn[g] = s.turn;
    if (key) key[g] = k;
  }
}
// the state itself for the Python objects: to_bytes image of game g after its moves (row of out_stride bytes, size in out_len)
__global__ __launch_bounds__(64) void k_sg_image(const uint8_t* init, uint32_t init_stride, const int32_t* moves, uint32_t n, uint32_t len,
                                                  uint64_t* hist, uint32_t hist_stride, uint8_t* out, uint32_t out_stride, uint32_t* out_len,
                                                  int32_t* status, uint32_t flags) {
  using G = StarGambit;
  __shared__ SgScratch sm;
  const uint32_t g = blockIdx.x, lane = threadIdx.x;
  if (g >= n) return;
  G::State s;
  uint64_t* hl = hist + static_cast<size_t>(g) * hist_stride;
  uint32_t nh = 0;
  int32_t stt = sg_start(init, init_stride, g, lane, s, hl, nh, hist_stride) ? 0 : -1;
  SgListRep rep{hl, nh, hist_stride, lane};
  for (uint32_t i = 0; i < len && stt == 0; ++i) {
    const int32_t mv = moves[static_cast<size_t>(g) * len + i];
    if (mv < 0) break;
    if (mv >= G::M) { stt = -1; break; }
    if (!(flags & 1u)) {
      G::gen_valid(s, lane, sm);
      const bool ok = G::is_valid_bit(sm, static_cast<uint32_t>(mv));
      G::lds_sync();
      if (!ok) { stt = -1; break; }
    }
    G::apply_move(s, static_cast<uint32_t>(mv), lane, sm, rep);
    if (rep.overflow) { stt = -1; break; }
  }
  if (lane == 0) status[g] = stt;
  // inner image (star_gambit_gs.cc:2253-2288) behind a 25-byte Unified header whose probs / pinned fields the caller fills in
  uint8_t* row = out + static_cast<size_t>(g) * out_stride;
  const uint32_t nu = G::nunits(s);
  const uint32_t inner = 4u + 9u * nu + 8u + 12u + 8u * nh;
  if (25u + inner > out_stride) { if (lane == 0) { status[g] = -2; out_len[g] = 0; } return; }
  auto wr32 = [&](uint8_t* p, uint32_t x) { p[0] = uint8_t(x); p[1] = uint8_t(x >> 8); p[2] = uint8_t(x >> 16); p[3] = uint8_t(x >> 24); };
  uint8_t* in = row + 25;
  if (lane < nu) {
    const uint32_t u = s.unit;
    uint8_t* r = in + 4 + 9 * lane;
    r[0] = uint8_t(G::u_type(u)); r[1] = uint8_t(G::u_player(u)); r[2] = uint8_t(G::u_slot(u)); r[3] = uint8_t(G::u_hp(u)); r[4] = uint8_t(G::u_facing(u));
    r[5] = uint8_t(int8_t(G::u_q(u))); r[6] = uint8_t(int8_t(G::u_r(u))); r[7] = uint8_t(G::u_moves(u)); r[8] = uint8_t(G::u_cannons(u));
  }
  for (uint32_t i = lane; i < nh; i += 64) {
    uint8_t* p = in + 4 + 9 * nu + 20 + 8 * i;
    const uint64_t x = hl[i];
    for (int k = 0; k < 8; ++k) p[k] = uint8_t(x >> (8 * k));
  }
  if (lane == 0) {
    for (int i = 0; i < 20; ++i) row[i] = 0;
    row[20] = uint8_t(G::variant(s));
    wr32(row + 21, inner);
    wr32(in, nu);
    uint8_t* t = in + 4 + 9 * nu;
    for (uint32_t pl = 0; pl < 2; ++pl) { for (uint32_t ty = 0; ty < 3; ++ty) t[pl * 4 + ty] = uint8_t(G::reserve(s, pl, ty)); t[pl * 4 + 3] = 0; }
    t[8] = uint8_t(s.player);
    wr32(t + 9, s.turn);
    t[13] = G::acted(s) ? 1 : 0; t[14] = G::over(s) ? 1 : 0;
    t[15] = uint8_t(int8_t(G::winner(s) < 3 ? int(G::winner(s)) : -1));
    wr32(t + 16, nh);
    out_len[g] = 25u + inner;
  }
}
// playout_eval (game_state.cc:10-54) for StarGambit: pi uniform over the legal moves, v the scores of a uniformly random rollout
__global__ __launch_bounds__(64) void k_playout_sg(const uint8_t* init, uint32_t init_stride, const int32_t* moves, uint32_t n, uint32_t len,
                                                    uint64_t* hist, uint32_t hist_stride, const uint64_t* seeds, float* v, float* pi, int32_t* status) {
  using G = StarGambit;
  __shared__ SgScratch sm;
  const uint32_t g = blockIdx.x, lane = threadIdx.x;
  if (g >= n) return;
  G::State s;
  uint64_t* hl = hist + static_cast<size_t>(g) * hist_stride;
  uint32_t nh = 0;
  int32_t stt = sg_start(init, init_stride, g, lane, s, hl, nh, hist_stride) ? 0 : -1;
  SgListRep rep{hl, nh, hist_stride, lane};
  for (uint32_t i = 0; i < len && stt == 0; ++i) {
    const int32_t mv = moves[static_cast<size_t>(g) * len + i];
    if (mv < 0) break;
    G::gen_valid(s, lane, sm);
    const bool ok = mv < G::M && G::is_valid_bit(sm, static_cast<uint32_t>(mv));
    G::lds_sync();
    if (!ok) { stt = -1; break; }
    G::apply_move(s, static_cast<uint32_t>(mv), lane, sm, rep);
    if (rep.overflow) { stt = -1; break; }
  }
  if (lane == 0) status[g] = stt;
  if (stt) return;
  const uint32_t kl = G::gen_valid(s, lane, sm);
  const float ksum = static_cast<float>(kl & 0xFFu);
  for (uint32_t m = lane; m < static_cast<uint32_t>(G::M); m += 64)
    pi[static_cast<size_t>(g) * G::M + m] = (G::is_valid_bit(sm, m) && ksum > 0.0f) ? 1.0f / ksum : 0.0f;
  G::lds_sync();
  Pcg32 roll;
  roll.seed(seeds[g]);
  uint32_t term = G::terminal(s);
  while (term == 0) {
    const uint32_t k = G::gen_valid(s, lane, sm);
    if (k == 0) break;
    const uint32_t r = lemire_below(roll, k);
    const unsigned long long w = lane < 27 ? sm.vbits[lane] : 0ull;
    const uint32_t cnt = static_cast<uint32_t>(__builtin_popcountll(w));
    uint32_t in = cnt;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(in, off, 64); if (lane >= static_cast<uint32_t>(off)) in += o; }
    const uint32_t lo = in - cnt;
    uint32_t mine = 0xFFFFFFFFu;
    if (r >= lo && r < lo + cnt) { unsigned long long m = w; for (uint32_t j = lo; j < r; ++j) m &= m - 1; mine = lane * 64 + static_cast<uint32_t>(__builtin_ctzll(m)); }
    const uint64_t owner = __ballot(mine != 0xFFFFFFFFu);
    const uint32_t mv = __shfl(mine, static_cast<int>(__builtin_ctzll(owner)), 64);
    G::lds_sync();
    G::apply_move(s, mv, lane, sm, rep);
    if (rep.overflow) break;
    term = G::terminal(s);
  }
  if (lane <= static_cast<uint32_t>(G::P))
    v[static_cast<size_t>(g) * (G::P + 1) + lane] = term ? ((term - 1 == lane) ? 1.0f : 0.0f) : static_cast<float>(1.0 / (G::P + 1));
}

// Tafl-family replay: one thread per game, repetition list in a global scratch row per game
// start position of game g: the game's initial position, or the reference pickle image in row g of `init` (dev_games.h
// TaflImage) with its repetition keys; false = malformed image
template <class GM>
__device__ bool tafl_start(const uint8_t* init, uint32_t stride, uint32_t g, typename GM::State& s, uint64_t* reps, uint32_t& nrep, uint32_t cap) {
  nrep = 0;
  if (!init) { s = GM::initial(); return true; }
  return tafl_parse_image<GM>(init + static_cast<size_t>(g) * stride, stride, s, reps, nrep, cap);
}
template <class GM>
__global__ void k_replay_tafl(const uint8_t* init, uint32_t init_stride, const int32_t* moves, uint32_t n, uint32_t len,
                              uint64_t* rep_scratch, uint32_t rep_stride,
                              uint8_t* valid, float* scores, float* canonical, uint32_t* player, uint32_t* turn,
                              uint64_t* key, int32_t* status, uint32_t flags) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const bool unchecked = (flags & 1u) && GM::kGameId != Tawlbwrdd::kGameId;
  constexpr uint32_t SPAN = GM::W + GM::H;
  uint64_t* reps = rep_scratch + static_cast<size_t>(g) * rep_stride;
  uint32_t nrep = 0;
  typename GM::State s;
  int32_t stt = tafl_start<GM>(init, init_stride, g, s, reps, nrep, rep_stride) ? 0 : -1;
  if (stt != 0) s = GM::initial();
  for (uint32_t i = 0; i < len && stt == 0; ++i) {
    const int32_t mv = moves[static_cast<size_t>(g) * len + i];
    if (mv < 0) break;
    bool legal = mv < GM::M;
    if (legal && !unchecked) {
      const uint32_t from = static_cast<uint32_t>(mv) / SPAN, tgt = static_cast<uint32_t>(mv) % SPAN;
      legal = GM::own_piece(s, s.player, from) && ((GM::slide_mask(s, from) >> tgt) & 1u);
    }
    if (!legal) { stt = -1; break; }
    typename GM::State before = s;
    bool cap = false;
    bool ok;
    if constexpr (GM::kGameId == Tawlbwrdd::kGameId) ok = GM::apply_move(s, static_cast<uint32_t>(mv), &cap);
    else ok = GM::apply_move(s, static_cast<uint32_t>(mv), &cap, unchecked);
    if (!ok) { stt = -1; break; }
    if (before.turn == 0) { reps[0] = GM::rep_key(before); nrep = 1; }   // tawlbwrdd_gs.cc:253-259
    if (cap) nrep = 0;
    const uint64_t k = GM::rep_key(s);
    uint32_t cnt = 1;
    for (uint32_t j = 0; j < nrep; ++j) cnt += reps[j] == k;
    if (nrep < rep_stride) reps[nrep++] = k;
    s.rep = cnt;
  }
  if (status) status[g] = stt;
  if (valid) {
    uint8_t* vr = valid + static_cast<size_t>(g) * GM::M;
    for (int m = 0; m < GM::M; ++m) vr[m] = 0;
    for (uint32_t sq = 0; sq < static_cast<uint32_t>(GM::SQ); ++sq) {
      if (!GM::own_piece(s, s.player, sq)) continue;
      const uint32_t mask = GM::slide_mask(s, sq);
      for (uint32_t b = 0; b < SPAN; ++b) if ((mask >> b) & 1u) vr[sq * SPAN + b] = 1;
    }
  }
  if (scores) {
    const uint32_t t = GM::terminal(s);
    for (int i = 0; i <= GM::P; ++i)
      scores[static_cast<size_t>(g) * (GM::P + 1) + i] = t == 0 ? -1.0f : (static_cast<int>(t) - 1 == i ? 1.0f : 0.0f);
  }
  if (canonical) for (int e = 0; e < GM::CANON; ++e) canonical[static_cast<size_t>(g) * GM::CANON + e] = GM::canonical_at(s, e);
  if (player) player[g] = s.player;
  if (turn) turn[g] = s.turn;
  if (key) key[g] = GM::key(s);
}

// playout_eval / playout_eval_batch for the Tafl family: one thread per state; the repetition list of the game record and
// of the rollout lives in the thread's scratch row (rep_stride >= record length + max_turns + 2 entries)
template <class GM>
__global__ void k_playout_tafl(const uint8_t* init, uint32_t init_stride, const int32_t* moves, uint32_t n, uint32_t len,
                               uint64_t* rep_scratch, uint32_t rep_stride, const uint64_t* seeds, float* v, float* pi, int32_t* status) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  constexpr uint32_t SPAN = GM::W + GM::H;
  uint64_t* reps = rep_scratch + static_cast<size_t>(g) * rep_stride;
  uint32_t nrep = 0;
  typename GM::State s;
  if (!tafl_start<GM>(init, init_stride, g, s, reps, nrep, rep_stride)) { if (status) status[g] = -1; return; }
  auto step = [&](uint32_t mv, bool unchecked) -> bool {     // one move with the reference's repetition bookkeeping
    typename GM::State before = s;
    bool cap = false, ok;
    if constexpr (GM::kGameId == Tawlbwrdd::kGameId) ok = GM::apply_move(s, mv, &cap);
    else ok = GM::apply_move(s, mv, &cap, unchecked);
    if (!ok) return false;
    if (before.turn == 0) { reps[0] = GM::rep_key(before); nrep = 1; }
    if (cap) nrep = 0;
    const uint64_t k = GM::rep_key(s);
    uint32_t cnt = 1;
    for (uint32_t j = 0; j < nrep; ++j) cnt += reps[j] == k;
    if (nrep < rep_stride) reps[nrep++] = k;
    s.rep = cnt;
    return true;
  };
  int32_t stt = 0;
  const bool unchecked = GM::kGameId != Tawlbwrdd::kGameId;       // the objects of these games replay like the reference's play_move
  for (uint32_t i = 0; i < len; ++i) {
    const int32_t mv = moves[static_cast<size_t>(g) * len + i];
    if (mv < 0) break;
    if (mv >= GM::M || !step(static_cast<uint32_t>(mv), unchecked)) { stt = -1; break; }
  }
  status[g] = stt;
  if (stt) return;
  auto count_moves = [&]() {
    uint32_t k = 0;
    for (uint32_t sq = 0; sq < static_cast<uint32_t>(GM::SQ); ++sq)
      if (GM::own_piece(s, s.player, sq)) k += __builtin_popcount(GM::slide_mask(s, sq));
    return k;
  };
  {   // policy: uniform over the leaf's legal moves, the u8 sum of the mask wraps mod 256 like dumb_eval
    float* pr = pi + static_cast<size_t>(g) * GM::M;
    for (int m = 0; m < GM::M; ++m) pr[m] = 0.0f;
    const float ksum = static_cast<float>(count_moves() & 0xFFu);
    if (ksum > 0.0f)
      for (uint32_t sq = 0; sq < static_cast<uint32_t>(GM::SQ); ++sq) {
        if (!GM::own_piece(s, s.player, sq)) continue;
        const uint32_t mask = GM::slide_mask(s, sq);
        for (uint32_t b = 0; b < SPAN; ++b) if ((mask >> b) & 1u) pr[sq * SPAN + b] = 1.0f / ksum;
      }
  }
  Pcg32 roll;
  roll.seed(seeds[g]);
  uint32_t term = GM::terminal(s);
  while (term == 0) {
    const uint32_t k = count_moves();
    if (k == 0) break;
    uint32_t r = lemire_below(roll, k), mv = 0;
    for (uint32_t sq = 0; sq < static_cast<uint32_t>(GM::SQ); ++sq) {    // the r-th legal move in ascending move order
      if (!GM::own_piece(s, s.player, sq)) continue;
      uint32_t mask = GM::slide_mask(s, sq);
      const uint32_t c = __builtin_popcount(mask);
      if (r >= c) { r -= c; continue; }
      for (uint32_t j = 0; j < r; ++j) mask &= mask - 1;
      mv = sq * SPAN + __builtin_ctz(mask);
      break;
    }
    if (!step(mv, false)) break;
    term = GM::terminal(s);
  }
  for (int i = 0; i <= GM::P; ++i)
    v[static_cast<size_t>(g) * (GM::P + 1) + i] = term ? ((static_cast<int>(term) - 1 == i) ? 1.0f : 0.0f) : static_cast<float>(1.0 / (GM::P + 1));
}

// ---- RNG probe: the device RNG layer on its own (parity tier "RNG") -------------------------------
__global__ void k_rng_probe(int kind, uint64_t seed, float param, uint32_t n, uint32_t reps, uint32_t* out_u, float* out_f) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  Pcg32 g;
  g.seed(seed);
  if (kind == 0) {
    for (uint32_t i = 0; i < n; ++i) out_u[i] = g.next();
  } else if (kind == 1) {  // std::shuffle of iota(n), `reps` times from one stream (one lane, plain arrays)
    for (uint32_t r = 0; r < reps; ++r) {
      uint32_t* a = out_u + static_cast<size_t>(r) * n;
      for (uint32_t i = 0; i < n; ++i) a[i] = i;
      if (n > 1) {
        uint32_t i = 1;
        if ((n & 1u) == 0) { const uint32_t j = lemire_below(g, 2); const uint32_t t = a[i]; a[i] = a[j]; a[j] = t; ++i; }
        while (i != n) {
          const uint32_t sr = i + 1, b1 = sr + 1;
          const uint32_t x = lemire_below(g, sr * b1);
          const uint32_t p0 = x / b1, p1 = x % b1;
          uint32_t t = a[i]; a[i] = a[p0]; a[p0] = t; ++i;
          t = a[i]; a[i] = a[p1]; a[p1] = t; ++i;
        }
      }
    }
  } else if (kind == 2) {
    for (uint32_t i = 0; i < n; ++i) out_f[i] = canonical01(g) * 1.0f + 0.0f;
  } else if (kind == 3) {  // one gamma object across draws (mcts.cc:435-440)
    Gamma d(param);
    for (uint32_t i = 0; i < n; ++i) out_f[i] = d.draw(g);
  } else if (kind == 4) {  // fresh gamma object per draw (mcts.cc:430)
    for (uint32_t i = 0; i < n; ++i) { Gamma d(param); out_f[i] = d.draw(g); }
  }
}

}  // namespace

extern "C" {

int azmi_rng_probe(int device, int kind, uint64_t seed, float param, uint32_t n, uint32_t reps, void* out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(AZMI_ERR_NO_DEVICE, "no HIP device: libazmi has no CPU path");
  if (kind < 0 || kind > 4 || !out) return fail(AZMI_ERR_INVALID, "bad rng probe arguments");
  HIP_TRY(hipSetDevice(device));
  const size_t count = static_cast<size_t>(n) * (kind == 1 ? std::max(reps, 1u) : 1u);
  void* d = nullptr;
  HIP_TRY(hipMalloc(&d, std::max<size_t>(count, 1) * 4));
  k_rng_probe<<<1, 64>>>(kind, seed, param, n, std::max(reps, 1u), static_cast<uint32_t*>(d), static_cast<float*>(d));
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out, d, count * 4, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(AZMI_ERR_NO_DEVICE, "rng probe: %s", hipGetErrorString(e));
  return AZMI_OK;
}

const char* azmi_last_error(void) { return g_err.c_str(); }
int azmi_abi_version(void) { return AZMI_ABI_VERSION; }
int azmi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void azmi_play_params_default(azmi_play_params* p) {  // play_manager.h:60-154
  std::memset(p, 0, sizeof(*p));
  p->max_batch_size = 1;
  p->cache_shards = 1;
  p->cpuct = 2.0f;
  p->start_temp = 1.0f;
  p->final_temp = 1.0f;
  p->tree_reuse = 1;
  p->mcts_root_temp = 1.0f;
  p->playout_cap_depth = 25;
  p->playout_cap_percent = 0.75f;
  p->gumbel_m = 16;
  p->gumbel_c_visit = 50.0f;
  p->gumbel_c_scale = 1.0f;
}
void azmi_engine_opts_default(azmi_engine_opts* o) {
  std::memset(o, 0, sizeof(*o));
  o->seed = 20240601ULL;  // mcts_test.cc:515
}

int azmi_game_info(int game, uint32_t* num_players, uint32_t* num_moves, uint32_t chw[3]) {
  GameInfo gi;
  if (!game_info(game, &gi)) return fail(AZMI_ERR_INVALID, "unknown game id %d", game);
  if (num_players) *num_players = gi.P;
  if (num_moves) *num_moves = gi.M;
  if (chw) { chw[0] = gi.C; chw[1] = gi.H; chw[2] = gi.W; }
  return AZMI_OK;
}

namespace {
// PlayManager ctor normalisation of model groups, seat permutations and the per-seat override matrices
// (play_manager.cc:24-113): mcts_visits / eval_type are given per PLAYER, folded per model group (the last player
// of a group wins), then expanded per (permutation, seat).
struct SeatTables {
  uint32_t num_groups = 1, num_perms = 1, max_visits = 0;
  bool all_random = true, any_random = false;
  bool any_gumbel = false, any_seat_resign = false, any_playout = false;
  uint32_t nn_groups = 0;        // bit g: a seat of model group g evaluates with the net
  std::vector<uint32_t> words;   // [perm][seat][kSeatWords]
};
int build_seat_tables(const azmi_play_params* p, uint32_t P, SeatTables* out) {
  uint8_t groups[AZMI_MAX_PLAYERS];
  if (p->num_model_groups_given == 0) for (uint32_t i = 0; i < P; ++i) groups[i] = static_cast<uint8_t>(i);
  else {
    if (p->num_model_groups_given != P) return fail(AZMI_ERR_INVALID, "model_groups must be empty or have one entry per player");
    for (uint32_t i = 0; i < P; ++i) groups[i] = p->model_groups[i];
  }
  uint32_t ng = 0;
  for (uint32_t i = 0; i < P; ++i) ng = std::max<uint32_t>(ng, groups[i] + 1u);
  if (ng > AZMI_MAX_GROUPS) return fail(AZMI_ERR_INVALID, "at most %d model groups", AZMI_MAX_GROUPS);
  uint32_t visits_g[AZMI_MAX_GROUPS] = {0, 0, 0, 0};
  int32_t eval_g[AZMI_MAX_GROUPS] = {AZMI_EVAL_NN, AZMI_EVAL_NN, AZMI_EVAL_NN, AZMI_EVAL_NN};
  for (uint32_t i = 0; i < P; ++i) {
    visits_g[groups[i]] = p->mcts_visits[i];
    if (p->num_eval_type) eval_g[groups[i]] = p->eval_type[i];
  }
  uint32_t np = p->num_seat_perms;
  uint8_t perms[AZMI_MAX_PERMS][AZMI_MAX_PLAYERS];
  if (np == 0) { np = 1; for (uint32_t s = 0; s < P; ++s) perms[0][s] = groups[s]; }
  else {
    if (np > AZMI_MAX_PERMS) return fail(AZMI_ERR_INVALID, "at most %d seat permutations", AZMI_MAX_PERMS);
    for (uint32_t q = 0; q < np; ++q)
      for (uint32_t s = 0; s < P; ++s) {
        if (p->seat_perms[q][s] >= ng) return fail(AZMI_ERR_INVALID, "seat_perms refers to model group %u but there are %u", unsigned(p->seat_perms[q][s]), ng);
        perms[q][s] = p->seat_perms[q][s];
      }
  }
  out->num_groups = ng; out->num_perms = np;
  out->words.assign(static_cast<size_t>(np) * P * kSeatWords, 0u);
  out->max_visits = p->playout_cap_randomization ? p->playout_cap_depth : 0;
  for (uint32_t q = 0; q < np; ++q)
    for (uint32_t s = 0; s < P; ++s) {
      const uint32_t g = perms[q][s];
      const uint32_t visits = p->has_seat_visits ? p->seat_visits[q][s] : visits_g[g];
      const uint32_t capv = p->has_seat_cap_visits ? p->seat_cap_visits[q][s] : p->playout_cap_depth;
      const float eps = p->has_seat_epsilon ? p->seat_epsilon[q][s] : p->epsilon;
      const float rt = p->has_seat_mcts_root_temp ? p->seat_mcts_root_temp[q][s] : p->mcts_root_temp;
      const uint32_t fz = p->has_seat_root_fpu_zero ? (p->seat_root_fpu_zero[q][s] != 0) : (p->root_fpu_zero != 0);
      const bool playout = eval_g[g] == AZMI_EVAL_PLAYOUT;
      const bool rnd = eval_g[g] == AZMI_EVAL_RANDOM;   // all_random below: "no seat needs a net" (RANDOM or PLAYOUT)
      out->any_playout = out->any_playout || playout;
      if (capv > 0xFFFFFFu) return fail(AZMI_ERR_INVALID, "seat_cap_visits too large");
      out->all_random = out->all_random && (rnd || playout); out->any_random = out->any_random || rnd || playout;
      if (!(rnd || playout)) out->nn_groups |= 1u << g;
      out->max_visits = std::max(out->max_visits, visits);
      if (p->playout_cap_randomization) out->max_visits = std::max(out->max_visits, capv);
      // per-seat Gumbel / resign overrides, play_manager.cc:116-176
      const uint32_t gum = p->has_seat_gumbel_enabled ? (p->seat_gumbel_enabled[q][s] != 0) : (p->gumbel_enabled != 0);
      const uint32_t gfull = p->has_seat_gumbel_full ? (p->seat_gumbel_full[q][s] != 0) : (p->gumbel_full != 0);
      const uint32_t g3 = p->has_seat_gumbel_use_improved_policy ? (p->seat_gumbel_use_improved_policy[q][s] != 0) : 0u;
      const uint32_t gm = p->has_seat_gumbel_m ? p->seat_gumbel_m[q][s] : p->gumbel_m;
      const float gcv = p->has_seat_gumbel_c_visit ? p->seat_gumbel_c_visit[q][s] : p->gumbel_c_visit;
      const float gcs = p->has_seat_gumbel_c_scale ? p->seat_gumbel_c_scale[q][s] : p->gumbel_c_scale;
      const float rth = p->has_seat_resign_threshold ? p->seat_resign_threshold[q][s] : -2.0f;
      const uint32_t rneed = std::max<uint32_t>(1u, p->has_seat_resign_consecutive ? p->seat_resign_consecutive[q][s] : 1u);
      if (gum && gm > kGumMaxM) return fail(AZMI_ERR_INVALID, "gumbel_m %u exceeds the engine limit %u", gm, kGumMaxM);
      if (rneed > 255u) return fail(AZMI_ERR_INVALID, "seat_resign_consecutive %u exceeds the engine limit 255", rneed);
      if (rth > -2.0f && P != 2) return fail(AZMI_ERR_INVALID, "Per-seat resign only works in 2 player games");
      out->any_gumbel = out->any_gumbel || gum;
      out->any_seat_resign = out->any_seat_resign || rth > -2.0f;
      uint32_t* w = &out->words[(static_cast<size_t>(q) * P + s) * kSeatWords];
      w[0] = visits; w[1] = seat_w1_pack(capv, fz, rnd ? 1u : 0u, g, playout ? 1u : 0u);
      std::memcpy(&w[2], &eps, 4); std::memcpy(&w[3], &rt, 4);
      w[4] = seat_gum_pack(gum, gfull, g3, gm, rneed);
      std::memcpy(&w[5], &gcv, 4); std::memcpy(&w[6], &gcs, 4); std::memcpy(&w[7], &rth, 4);
    }
  return AZMI_OK;
}
}  // namespace

namespace {
int pm_create_impl(int game, const azmi_play_params* params, const azmi_engine_opts* opts_in, azmi_cache* const* ext_caches,
                   uint32_t num_ext, bool use_ext, azmi_pm** out);
}
int azmi_pm_create(int game, const azmi_play_params* params, const azmi_engine_opts* opts_in, azmi_pm** out) {
  return pm_create_impl(game, params, opts_in, nullptr, 0, false, out);
}
// PlayManager(gs, params, caches), play_manager.cc:644-649: max_cache_size is forced to 0 and the given caches are used
int azmi_pm_create_with_caches(int game, const azmi_play_params* params, const azmi_engine_opts* opts_in, azmi_cache* const* caches,
                               uint32_t num_caches, azmi_pm** out) {
  if (num_caches && !caches) return fail(AZMI_ERR_INVALID, "null argument");
  return pm_create_impl(game, params, opts_in, caches, num_caches, true, out);
}
namespace {
int pm_create_impl(int game, const azmi_play_params* params, const azmi_engine_opts* opts_in, azmi_cache* const* ext_caches,
                   uint32_t num_ext, bool use_ext, azmi_pm** out) {
  if (!params || !out) return fail(AZMI_ERR_INVALID, "null argument");
  GameInfo gi;
  if (!game_info(game, &gi)) return fail(AZMI_ERR_INVALID, "unknown game id %d", game);
  azmi_engine_opts opts;
  if (opts_in) opts = *opts_in; else azmi_engine_opts_default(&opts);
  // play_manager.cc:20-22
  if (params->num_mcts_visits != gi.P) return fail(AZMI_ERR_INVALID, "You must specify MCTS visits for each player");
  if (params->concurrent_games == 0) return fail(AZMI_ERR_INVALID, "concurrent_games must be > 0");
  if (params->num_eval_type != 0 && params->num_eval_type != gi.P)
    return fail(AZMI_ERR_INVALID, "eval_type must be empty or have one entry per player");
  SeatTables seats;
  { const int rc_seats = build_seat_tables(params, gi.P, &seats); if (rc_seats != AZMI_OK) return rc_seats; }
  if (params->resign_percent > 0 && gi.P != 2) return fail(AZMI_ERR_INVALID, "Resigning only works in 2 player games");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(AZMI_ERR_NO_DEVICE, "no HIP device: libazmi has no CPU path");
  if (opts.device < 0 || opts.device >= ndev) return fail(AZMI_ERR_INVALID, "device %d out of range", opts.device);
  HIP_TRY(hipSetDevice(opts.device));

  auto pm = new azmi_pm();
  pm->game = game; pm->device = opts.device; pm->gi = gi; pm->params = *params;
  EngineParams& ep = pm->ep;
  ep.S = params->concurrent_games;
  ep.games_to_play = params->games_to_play;
  const uint32_t max_visits = seats.max_visits;
  for (uint32_t i = 0; i < gi.P; ++i) ep.visits[i] = params->mcts_visits[i];   // kept for reference; kernels read ar.seat_tab
  ep.cap_visits = params->playout_cap_depth;
  ep.num_perms = seats.num_perms; ep.num_groups = seats.num_groups;
  pm->all_random = seats.all_random;
  pm->nn_groups = seats.nn_groups;
  pm->any_playout = seats.any_playout;
  pm->split_rounds = game == AZMI_GAME_CONNECT4 && !seats.any_gumbel && !seats.any_playout && getenv("AZMI_NO_SPLIT") == nullptr;
  pm->big_split = (game == AZMI_GAME_TAWLBWRDD || game == AZMI_GAME_BRANDUBH || game == AZMI_GAME_OPENTAFL || game == AZMI_GAME_STARGAMBIT) && !seats.any_playout && getenv("AZMI_NO_BIG_SPLIT") == nullptr;
  // (StarGambit, round 6: its simulation kernel keeps the whole register file - k_round_big_sim1, one wave per SIMD: held to 256
  // registers it spills 232 of them)
  ep.cpuct = params->cpuct; ep.start_temp = params->start_temp; ep.final_temp = params->final_temp;
  ep.half_life = params->temp_decay_half_life;
  ep.n_half_life_v = 0;
  if (game == AZMI_GAME_STARGAMBIT) {
    if (params->num_temp_decay_half_life_by_variant > 4) { delete pm; return fail(AZMI_ERR_INVALID, "temp_decay_half_life_by_variant: at most 4 entries"); }
    ep.n_half_life_v = params->num_temp_decay_half_life_by_variant;
    for (uint32_t i = 0; i < 4; ++i) ep.half_life_v[i] = i < ep.n_half_life_v ? params->temp_decay_half_life_by_variant[i] : 0.0f;
    const bool given = opts.sg_variant_probs[0] != 0 || opts.sg_variant_probs[1] != 0 || opts.sg_variant_probs[2] != 0 || opts.sg_variant_probs[3] != 0;
    ep.sg_pinned = given || opts.sg_pinned_variant != 0 ? opts.sg_pinned_variant : -1;   // zeroed opts = the constructor's defaults
    for (uint32_t i = 0; i < 4; ++i) ep.sg_probs[i] = given ? opts.sg_variant_probs[i] : 0.25f;
  }
  ep.epsilon = params->epsilon; ep.root_temp = params->mcts_root_temp; ep.fpu_reduction = params->fpu_reduction;
  ep.cap_percent = params->playout_cap_percent; ep.resign_percent = params->resign_percent;
  ep.resign_playthrough = params->resign_playthrough_percent;
  ep.history = params->history_enabled != 0; ep.tree_reuse = params->tree_reuse != 0;
  ep.cap_rand = params->playout_cap_randomization != 0; ep.root_fpu_zero = params->root_fpu_zero != 0;
  ep.shaped = params->shaped_dirichlet != 0; ep.pruning = params->policy_target_pruning != 0;
  for (uint32_t i = 0; i < gi.P; ++i) ep.eval_random[i] = 0;
  ep.gumbel_on = seats.any_gumbel ? 1u : 0u;
  ep.gumbel_hist = params->gumbel_enabled != 0;
  ep.fast_gumbel = params->fast_search_uses_gumbel != 0;
  ep.seat_resign = seats.any_seat_resign ? 1u : 0u;
  ep.gum_stride = gi.maxk;
  // simulations a slot may finish inside one round without the net.  A round lasts as long as its SLOWEST slot, so inline
  // continuation pays only where a simulation is cheap next to the net launch (Connect4 with NN seats: 3 - measured 2588 / 2637 /
  // 2559 / 2502 games/s at 2 / 3 / 4 / 5 with the 3-board net tiles; 4 was the optimum behind the slower 6-board tiles).  In the wide-game engine a
  // simulation costs ~100 us: with NN seats one chain of cache hits / terminal leaves would stretch the round for every slot
  // (measured: Tawlbwrdd 4.49 -> 5.01 M sims/s, StarGambit 1.74 -> 2.14 M sims/s with 1 instead of 4); with RANDOM / PLAYOUT
  // seats only there is no net to wait for and 4 stands.
  ep.max_inline = opts.max_inline ? opts.max_inline : (seats.all_random ? 4u : game == AZMI_GAME_CONNECT4 ? 3u : 1u);
  pm->max_inline_explicit = opts.max_inline != 0;
  ep.sim_budget = getenv("AZMI_SIM_BUDGET_US") ? static_cast<uint32_t>(100.0 * atof(getenv("AZMI_SIM_BUDGET_US"))) : 0u;
  ep.max_hist_rows = gi.max_turns;
  ep.max_depth = gi.max_turns + 2;
  // every search expands at most one node (<= maxk children) per simulation and a tree is
  // searched on at most ceil(max_turns/2) turns; re-rooting an empty tree expands once more.
  // Connect4: cap_branch == maxk, a hard bound.  Wide games: sized for the average branching factor and
  // capped at 16 M nodes per tree; running out raises overflow bit 1 (arena compaction is the next step).
  uint64_t cap64 = (static_cast<uint64_t>((gi.max_turns + 1) / 2) * max_visits + gi.max_turns) * gi.cap_branch + 8;
  ep.half_nodes = 0; ep.compact_above = 0;
  if (gi.cap_branch != gi.maxk) {
    // wide games: two halves of 4 searches' worth of nodes each; the live subtree is copied into the idle
    // half (k_compact) when a move leaves less than one search of room in the active one
    const uint64_t growth = (static_cast<uint64_t>(max_visits) + 12 + (opts.max_inline ? opts.max_inline : 4)) * (gi.cap_branch * 3 / 2);
    const uint64_t half = std::min<uint64_t>(std::max<uint64_t>(4 * growth, 64u << 10), 8u << 20);
    ep.half_nodes = static_cast<uint32_t>(half);
    ep.compact_above = static_cast<uint32_t>(half > growth ? half - growth : half / 2);
    cap64 = 2 * half;
    if (const char* e = getenv("AZMI_COMPACT_ABOVE")) ep.compact_above = static_cast<uint32_t>(atoi(e));  // test hook: 0 = compact after every move
  }
  if (cap64 > 0xFFFFFFF0ULL) { delete pm; return fail(AZMI_ERR_INVALID, "tree arena too large"); }
  ep.cap = static_cast<uint32_t>(cap64);
  ep.log_moves = opts.log_moves != 0;
  ep.log_cap = ep.log_moves ? (opts.move_log_capacity ? opts.move_log_capacity
                                                       : (params->games_to_play + ep.S) * gi.max_turns) : 0;
  // finished-sample ring (rows): by default every row of the run, (games_to_play + S) * max_turns, computed in 64 bits and
  // bounded to 8 GiB of rows (never less than two full games per slot) — a longer run has to drain it
  // (build_history_batch / azmi_pm_history_consume), like the reference's hist_saver drains the unbounded queue
  if (ep.history && opts.history_capacity) {
    ep.hist_cap = opts.history_capacity;
  } else if (ep.history) {
    const uint64_t row_bytes = 4ull * (static_cast<uint64_t>(gi.C) * gi.H * gi.W + gi.P + 1 + gi.M + 4);
    // rows one game can produce: max_turns, except StarGambit whose 4096 is a hard bound on ACTIONS far above any real game
    // (self-play games: ~220 sample rows; thousands of random games peak below 800) - the default ring counts 1024 per game
    const uint64_t game_rows = game == AZMI_GAME_STARGAMBIT ? 1024u : gi.max_turns;
    const uint64_t all_rows = (static_cast<uint64_t>(params->games_to_play) + ep.S) * game_rows;
    const uint64_t bound = std::max<uint64_t>(2ull * ep.S * game_rows, (8ull << 30) / row_bytes);
    ep.hist_cap = static_cast<uint32_t>(std::min<uint64_t>(std::min(all_rows, bound), 0x7FFFFFFFull));
  } else {
    ep.hist_cap = 0;
  }

  const uint32_t S = ep.S, P = gi.P, M = gi.M, CANON = gi.C * gi.H * gi.W;
  const size_t T = static_cast<size_t>(S) * P, NODES = T * ep.cap;
  EngineArrays& ar = pm->ar;
  int rc = AZMI_OK;
#define A(field, count, zero) if (rc == AZMI_OK) rc = pm->alloc(ar.field, (count), (zero))
  A(ctl, 1, true);
  A(ended_list, S, true);
  A(mover_list, S, true);
  A(pend, game == AZMI_GAME_CONNECT4 ? static_cast<size_t>(S) * Connect4::GROUP : 0, true);
  A(eval_list, static_cast<size_t>(S) * ep.num_groups, true);
  A(seat_tab, seats.words.size(), false);
  A(perm, S, true);
  A(leaf_group, S, true);
  A(a_perm_scores, static_cast<size_t>(S) * ep.num_perms * (P + 1), true);
  A(a_perm_games, static_cast<size_t>(S) * ep.num_perms, true);
  A(gs_words, static_cast<size_t>(gi.state_words) * S, true);
  A(rng, S, true); A(coin, S, true);
  A(sstate, S, true); A(flags, S, true);
  A(cur, S, true); A(plen, S, true);
  A(path, static_cast<size_t>(S) * ep.max_depth, true);
  A(slot_games, S, true);
  A(rep_list, game != AZMI_GAME_CONNECT4 ? static_cast<size_t>(S) * (gi.max_turns + 2) : 0, true);
  A(rep_len, S, true);
  A(g_dsum, 5 * static_cast<size_t>(S), true); A(g_cnt, 3 * static_cast<size_t>(S), true);
  A(a_scores, static_cast<size_t>(S) * (P + 1), true); A(a_resign, static_cast<size_t>(S) * (P + 1), true);
  A(a_len, S, true); A(a_dsum, 5 * static_cast<size_t>(S), true); A(a_cnt, 3 * static_cast<size_t>(S), true);
  A(c_sims, S, true); A(c_evals, S, true);
  A(ph_count, S, true);
  A(ph_canon, ep.history ? static_cast<size_t>(S) * ep.max_hist_rows * (game == AZMI_GAME_CONNECT4 ? 2 * kPendingWords : game == AZMI_GAME_STARGAMBIT ? 2 * kSgPendWords : CANON) : 0, false);
  A(ph_pi, ep.history ? static_cast<size_t>(S) * ep.max_hist_rows * M : 0, false);
  A(ph_meta, ep.history ? static_cast<size_t>(S) * ep.max_hist_rows * 2 : 0, false);
  A(root, T, true); A(bump, T, true); A(depth, T, true); A(tld, T, true);
  if (game == AZMI_GAME_CONNECT4) {
    A(nodes, NODES, false);
  } else {
    A(N, NODES, false); A(Q, NODES, false); A(Pr, NODES, false); A(D, NODES, false); A(V, NODES, false);
    A(META, NODES, false);
  }
  A(canon, static_cast<size_t>(S) * CANON, true);
  A(v, static_cast<size_t>(S) * (P + 1), true);
  A(pi, static_cast<size_t>(S) * M, true);
  A(leaf_key, S, true);
  if (game == AZMI_GAME_CONNECT4) { A(leaf_pos, 3 * static_cast<size_t>(S), true); A(req_seq, S, true); }
  A(h_canon, static_cast<size_t>(ep.hist_cap) * CANON, false);
  A(h_v, static_cast<size_t>(ep.hist_cap) * (P + 1), false);
  A(h_pi, static_cast<size_t>(ep.hist_cap) * M, false);
  A(h_meta, static_cast<size_t>(ep.hist_cap) * 4, false);
  A(log_rows, static_cast<size_t>(ep.log_cap) * 8, false);
  A(log_counts, static_cast<size_t>(ep.log_cap) * M, false);
  // position cache: play_manager.cc:195-203 (one model group): ghost = 9/10 of the capacity.  The
  // reference's cache_shards (<= 255) exists to spread a mutex; here a shard is one wavefront's worth
  // of entries (64) and the unit of parallelism of the insert kernel.
  bool any_ext = false;
  if (use_ext) {
    // caches_[group] is indexed by model group (play_manager.cc:619-642): the list must cover every group; None entries
    // (groups that never reach the net) are allowed
    if (num_ext != 0 && num_ext < ep.num_groups) { delete pm; return fail(AZMI_ERR_INVALID, "caches: %u entries for %u model groups", num_ext, ep.num_groups); }
    for (uint32_t g = 0; g < std::min<uint32_t>(num_ext, ep.num_groups); ++g) {
      const azmi_cache* c = ext_caches[g];
      if (!c) continue;
      any_ext = true;
      if (c->device != opts.device) { delete pm; return fail(AZMI_ERR_INVALID, "caches[%u] lives on another device", g); }
      if (c->c.np != M || c->c.nv != P + 1) { delete pm; return fail(AZMI_ERR_INVALID, "caches[%u]: num_policy / num_value do not match the game", g); }
      if (c->c.cap != kWaveCap) {
        delete pm;
        return fail(AZMI_ERR_INVALID, "caches[%u]: the engine probes 64-entry shards; create the cache with shards = max_size / 64 "
                    "(ShardedS3FIFOCache.for_engine)", g);
      }
    }
  }
  ep.cache_on = use_ext ? any_ext : params->max_cache_size > 0;
  if (ep.cache_on) {
    // wave-resident shards: 64 entries each (dev_cache.h); max_cache_size is rounded down to a multiple of 64
    // one cache per model group, max_cache_size / num_model_groups entries each (play_manager.cc:195-203)
    const uint32_t shards = use_ext ? 1u : std::max<uint32_t>(1, params->max_cache_size / ep.num_groups / kWaveCap);
    pm->cache_shards = shards;
    A(cache_keys, S, true);
    pm->group_caches.resize(ep.num_groups);
    pm->group_cache_counted.assign(ep.num_groups, 1);
    for (uint32_t g = 0; g < ep.num_groups && rc == AZMI_OK; ++g) {
      if (use_ext && g < num_ext && ext_caches[g]) { pm->group_caches[g] = ext_caches[g]->c; continue; }
      if (use_ext) pm->group_cache_counted[g] = 0;   // a None entry: one private 64-entry shard stands in, outside the statistics
      if (cache_alloc(pm->group_caches[g], pm->allocs, shards * kWaveCap, shards, static_cast<uint32_t>(static_cast<uint64_t>(shards) * kWaveCap * 9 / 10), M, P + 1) != hipSuccess) {
        delete pm;
        return fail(AZMI_ERR_OOM, "position cache allocation failed");
      }
    }
    if (rc == AZMI_OK) {
      ar.cache = pm->group_caches[0];
      CacheView* dev_views = nullptr;
      rc = pm->alloc(dev_views, ep.num_groups, false);
      if (rc == AZMI_OK && hipMemcpy(dev_views, pm->group_caches.data(), sizeof(CacheView) * ep.num_groups, hipMemcpyHostToDevice) != hipSuccess) rc = AZMI_ERR_NO_DEVICE;
      ar.caches = dev_views;
    }
  }
  ep.trace_slot = getenv("AZMI_TRACE_SLOT") ? static_cast<uint32_t>(atoi(getenv("AZMI_TRACE_SLOT"))) : 0xFFFFFFFFu;
  ep.trace_cap = ep.trace_slot != 0xFFFFFFFFu ? (1u << 16) : 1u;
  ep.trace_after = getenv("AZMI_TRACE_AFTER") ? static_cast<uint32_t>(atoi(getenv("AZMI_TRACE_AFTER"))) : 0u;
  A(trace, 2 * static_cast<size_t>(ep.trace_cap), true);
  if (ep.half_nodes) A(compact_flag, T, true);
  if (ep.gumbel_on) {
    A(gum_state, static_cast<size_t>(S) * P * 8, true);
    A(gum_g, static_cast<size_t>(S) * P * ep.gum_stride, true);
    A(gum_surv, static_cast<size_t>(S) * P * kGumMaxM, true);
  }
  A(resign_streak, static_cast<size_t>(S) * P, true);
  A(roll, S, true);
  if (game == AZMI_GAME_STARGAMBIT) {
    A(rep_path, static_cast<size_t>(S) * (gi.max_turns + 2), true);
    A(a_var_scores, static_cast<size_t>(S) * 4 * ep.num_perms * (P + 1), true);
    A(a_var_games, static_cast<size_t>(S) * 4 * ep.num_perms, true);
    A(a_var_len, static_cast<size_t>(S) * 4, true);
    A(a_var_dsum, static_cast<size_t>(S) * 4 * 5, true);
    A(a_var_cnt, static_cast<size_t>(S) * 4 * 3, true);
  }
#undef A
  if (rc != AZMI_OK) { delete pm; return rc; }
  if (hipDeviceSynchronize() != hipSuccess) { delete pm; return fail(AZMI_ERR_NO_DEVICE, "device sync failed"); }
  if (hipStreamCreateWithFlags(&pm->stream, hipStreamNonBlocking) != hipSuccess) {
    delete pm;
    return fail(AZMI_ERR_NO_DEVICE, "hipStreamCreate failed");
  }
  Control c{};
  c.games_started = S;  // play_manager.cc:15
  c.live_slots = S;
  if (hipMemcpy(ar.ctl, &c, sizeof(c), hipMemcpyHostToDevice) != hipSuccess) { delete pm; return fail(AZMI_ERR_NO_DEVICE, "ctl init failed"); }
  if (hipMemcpy(ar.seat_tab, seats.words.data(), seats.words.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { delete pm; return fail(AZMI_ERR_NO_DEVICE, "seat table upload failed"); }
  {
    std::vector<uint32_t> perm0(S);
    for (uint32_t i = 0; i < S; ++i) perm0[i] = i % ep.num_perms;   // gd.perm_index = i % seat_perms_.size(), play_manager.cc:218
    if (hipMemcpy(ar.perm, perm0.data(), S * 4, hipMemcpyHostToDevice) != hipSuccess) { delete pm; return fail(AZMI_ERR_NO_DEVICE, "perm init failed"); }
  }
  pm->pending_g.resize(ep.num_groups);
  k_seed<<<(S + 255) / 256, 256, 0, pm->stream>>>(ar, S, opts.seed);
  if (hipStreamSynchronize(pm->stream) != hipSuccess) { delete pm; return fail(AZMI_ERR_NO_DEVICE, "seed kernel failed"); }
  pm->host_v.assign(static_cast<size_t>(S) * (P + 1), 0.0f);
  pm->host_pi.assign(static_cast<size_t>(S) * M, 0.0f);
  pm->last = pm->stream;
  *out = pm;
  return AZMI_OK;
}
}  // namespace

void azmi_pm_destroy(azmi_pm* pm) {
  if (!pm) return;
  (void)hipSetDevice(pm->device);
  (void)hipDeviceSynchronize();
  delete pm;
}

int azmi_pm_round(azmi_pm* pm, void* stream) {
  if (!pm) return fail(AZMI_ERR_INVALID, "null pm");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  return launch_round(pm, pm->pick(stream));
}

namespace {
constexpr uint32_t kGraphRounds = 16;
// the net on this engine's leaf batch: only the rows k_round listed (Connect4 engine), else the whole batch
int pm_net_forward(azmi_pm* pm, uint32_t group, azmi_net* net, hipStream_t st) {
  int rc;
  if (pm->game == AZMI_GAME_CONNECT4 || pm->ep.num_groups > 1 || pm->ep.cache_on)
    rc = azmi_net_forward_rows(net, pm->ar.canon, pm->ar.v, pm->ar.pi, pm->ar.eval_list + static_cast<size_t>(group) * pm->ep.S,
                               &pm->ar.ctl->eval_count[group], pm->ep.S, st);
  else
    rc = azmi_net_forward(net, pm->ar.canon, pm->ar.v, pm->ar.pi, pm->ep.S, st);
  if (rc != AZMI_OK) return fail(rc, "%s", azmi_net_last_error());
  return AZMI_OK;
}
// split round + a Connect4-family bf16 net + one model group: the move step rides in the net launch (k_net_move)
bool can_fuse(const azmi_pm* pm, const azmi_net* net, azmi_net_c4_view* view) {
  return pm->split_rounds && pm->ep.num_groups == 1 && !pm->all_random && getenv("AZMI_NO_FUSE") == nullptr && azmi_net_c4_view_get(net, view) != 0 && view->x3 == 0;
}
int launch_net_move(azmi_pm* pm, const azmi_net_c4_view& view, hipStream_t st) {
  // (per device: the attribute belongs to the device's copy of the kernel; the tile's LDS need is a constant of the geometry)
  static std::mutex reserved_mu;
  static std::vector<int> reserved_devices;
  {
    std::lock_guard<std::mutex> l(reserved_mu);
    if (std::find(reserved_devices.begin(), reserved_devices.end(), pm->device) == reserved_devices.end()) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_net_move), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(view.lds_bytes)));
      reserved_devices.push_back(pm->device);
    }
  }
  const uint32_t net_tiles = (pm->ep.S + azmi_net_dev::c4::TileSmall::TBW - 1) / azmi_net_dev::c4::TileSmall::TBW;
  const uint32_t move_blocks = (pm->ep.S * Connect4::GROUP + 255) / 256;
  k_net_move<<<net_tiles + move_blocks, 256, view.lds_bytes, st>>>(view.nd, view.np, pm->ep, pm->ar, net_tiles);
  HIP_TRY(hipGetLastError());
  return AZMI_OK;
}
int one_round_with_net(azmi_pm* pm, azmi_net* net, hipStream_t st) {
  azmi_net_c4_view view;
  if (can_fuse(pm, net, &view)) {
    const int rc = launch_round(pm, st, true);
    return rc != AZMI_OK ? rc : launch_net_move(pm, view, st);
  }
  int rc = launch_round(pm, st);
  if (rc != AZMI_OK) return rc;
  for (uint32_t g = 0; g < pm->ep.num_groups; ++g) {   // one net given: it serves every model group
    rc = pm_net_forward(pm, g, net, st);
    if (rc != AZMI_OK) return rc;
  }
  return AZMI_OK;
}
int ensure_graph(azmi_pm* pm, azmi_net* net, hipStream_t st) {
  if (pm->graph_exec && pm->graph_stream == st && pm->graph_net == net) return AZMI_OK;
  if (pm->graph_exec) { (void)hipGraphExecDestroy(pm->graph_exec); pm->graph_exec = nullptr; }
  hipGraph_t graph = nullptr;
  if (azmi_net_reserve_stream(net, st, pm->ep.S) != AZMI_OK) return fail(AZMI_ERR_OOM, "%s", azmi_net_last_error());
  HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  int rc = AZMI_OK;
  for (uint32_t r = 0; r < kGraphRounds && rc == AZMI_OK; ++r) rc = one_round_with_net(pm, net, st);
  const hipError_t e = hipStreamEndCapture(st, &graph);
  if (rc != AZMI_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess) return fail(AZMI_ERR_NO_DEVICE, "hipStreamEndCapture: %s", hipGetErrorString(e));
  const hipError_t e2 = hipGraphInstantiate(&pm->graph_exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e2 != hipSuccess) { pm->graph_exec = nullptr; return fail(AZMI_ERR_NO_DEVICE, "hipGraphInstantiate: %s", hipGetErrorString(e2)); }
  pm->graph_stream = st;
  pm->graph_net = net;
  return AZMI_OK;
}
}  // namespace

// holds a stream for about `us` microseconds (wall_clock64 ticks at 100 MHz)
__global__ void k_delay(uint32_t us) {
  const uint64_t t0 = wall_clock64();
  while (wall_clock64() - t0 < static_cast<uint64_t>(us) * 100u) __builtin_amdgcn_s_sleep(32);
}

int azmi_run_rounds(azmi_pm* const* pms, azmi_net* net, uint32_t k, uint32_t rounds, void* const* streams) {
  if (!pms || !net || !streams || k == 0) return fail(AZMI_ERR_INVALID, "null argument");
  for (uint32_t i = 0; i < k; ++i)      // PlayManager::stop(): the workers leave their loop (play_manager.cc:272)
    if (pms[i] && pms[i]->stopped.load(std::memory_order_relaxed)) return AZMI_OK;
  if (const char* sg = getenv("AZMI_STAGGER_US")) {     // experiment: odd shards start half a cycle late
    const uint32_t us = static_cast<uint32_t>(atoi(sg));
    for (uint32_t i = 1; i < k && us; i += 2) k_delay<<<1, 64, 0, pms[i]->pick(streams[i])>>>(us);
  }
  // hipGraph replay of 16 rounds per launch is available (AZMI_GRAPH=1) but off by default: measured equal to plain
  // launches on this workload (2300 vs 2304 games/s — the host is ahead of the GPU either way), and capturing and
  // instantiating a graph per engine costs about a second at start-up
  const bool use_graph = getenv("AZMI_GRAPH") != nullptr && getenv("AZMI_NO_GRAPH") == nullptr;
  uint32_t done = 0;
  if (use_graph) {
    for (uint32_t i = 0; i < k; ++i) {
      const int rc = ensure_graph(pms[i], net, pms[i]->pick(streams[i]));
      if (rc != AZMI_OK) return rc;
    }
    for (; done + kGraphRounds <= rounds; done += kGraphRounds)
      for (uint32_t i = 0; i < k; ++i) HIP_TRY(hipGraphLaunch(pms[i]->graph_exec, pms[i]->graph_stream));
  }
  for (; done < rounds; ++done)
    for (uint32_t i = 0; i < k; ++i) {
      const int rc = one_round_with_net(pms[i], net, pms[i]->pick(streams[i]));
      if (rc != AZMI_OK) return rc;
    }
  return AZMI_OK;
}

// The round loop with one net PER MODEL GROUP (gating / benchmark matches, game_runner.py:2184-2332: two models, seats
// swapped by the permutations): nets[g] evaluates the leaves of model group g; NULL = that group needs no net (RANDOM /
// PLAYOUT evaluator).
int azmi_run_rounds_groups(azmi_pm* const* pms, azmi_net* const* nets, uint32_t num_nets, uint32_t k, uint32_t rounds, void* const* streams) {
  if (!pms || !nets || !streams || k == 0) return fail(AZMI_ERR_INVALID, "null argument");
  for (uint32_t i = 0; i < k; ++i)
    if (pms[i]->ep.num_groups > num_nets) return fail(AZMI_ERR_INVALID, "engine %u has %u model groups but %u nets were given", i, pms[i]->ep.num_groups, num_nets);
  for (uint32_t r = 0; r < rounds; ++r)
    for (uint32_t i = 0; i < k; ++i) {
      azmi_pm* pm = pms[i];
      hipStream_t st = pm->pick(streams[i]);
      int rc = launch_round(pm, st);
      if (rc != AZMI_OK) return rc;
      for (uint32_t g = 0; g < pm->ep.num_groups; ++g) {
        if (!nets[g]) continue;
        rc = pm_net_forward(pm, g, nets[g], st);
        if (rc != AZMI_OK) return rc;
      }
    }
  return AZMI_OK;
}

int azmi_pm_net_forward(azmi_pm* pm, azmi_net* net, void* stream) {
  if (!pm || !net) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  for (uint32_t g = 0; g < pm->ep.num_groups; ++g) {   // the same net for every model group
    const int rc = pm_net_forward(pm, g, net, pm->pick(stream));
    if (rc != AZMI_OK) return rc;
  }
  return AZMI_OK;
}

int azmi_pm_round_net(azmi_pm* pm, azmi_net* net, void* stream, uint32_t part) {
  if (!pm || !net) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  hipStream_t st = pm->pick(stream);
  azmi_net_c4_view view;
  const bool fused = can_fuse(pm, net, &view);
  if (part == 0 || part == 1) {          // the tree half of the round
    const int rc = launch_round(pm, st, fused);
    if (rc != AZMI_OK || part == 1) return rc;
  }
  if (fused) return launch_net_move(pm, view, st);
  for (uint32_t g = 0; g < pm->ep.num_groups; ++g) {
    const int rc = pm_net_forward(pm, g, net, st);
    if (rc != AZMI_OK) return rc;
  }
  return AZMI_OK;
}

int azmi_pm_net_forward_group(azmi_pm* pm, uint32_t group, azmi_net* net, void* stream) {
  if (!pm || !net) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  if (group >= pm->ep.num_groups) return fail(AZMI_ERR_INVALID, "model group %u out of range", group);
  return pm_net_forward(pm, group, net, pm->pick(stream));
}

int azmi_pm_groups(azmi_pm* pm, uint32_t* num_model_groups, uint32_t* num_seat_perms) {
  if (!pm) return fail(AZMI_ERR_INVALID, "null pm");
  if (num_model_groups) *num_model_groups = pm->ep.num_groups;
  if (num_seat_perms) *num_seat_perms = pm->ep.num_perms;
  return AZMI_OK;
}

int azmi_pm_perm_scores(azmi_pm* pm, uint32_t perm, float* out_scores, uint32_t* games_completed) {
  if (!pm) return fail(AZMI_ERR_INVALID, "null pm");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  if (perm >= pm->ep.num_perms) return fail(AZMI_ERR_INVALID, "seat permutation %u out of range", perm);
  const uint32_t S = pm->ep.S, V = pm->gi.P + 1, NP = pm->ep.num_perms;
  std::vector<float> a; std::vector<uint32_t> g;
  int rc = d2h(a, pm->ar.a_perm_scores, static_cast<size_t>(S) * NP * V, pm->last); if (rc) return rc;
  rc = d2h(g, pm->ar.a_perm_games, static_cast<size_t>(S) * NP, pm->last); if (rc) return rc;
  if (out_scores) {
    for (uint32_t i = 0; i < V; ++i) out_scores[i] = 0.0f;
    for (uint32_t s = 0; s < S; ++s) for (uint32_t i = 0; i < V; ++i) out_scores[i] += a[(static_cast<size_t>(s) * NP + perm) * V + i];
  }
  if (games_completed) { uint32_t n = 0; for (uint32_t s = 0; s < S; ++s) n += g[static_cast<size_t>(s) * NP + perm]; *games_completed = n; }
  return AZMI_OK;
}

int azmi_pm_io_buffers(azmi_pm* pm, float** dev_canonical, float** dev_v, float** dev_pi) {
  if (!pm) return fail(AZMI_ERR_INVALID, "null pm");
  if (dev_canonical) *dev_canonical = pm->ar.canon;
  if (dev_v) *dev_v = pm->ar.v;
  if (dev_pi) *dev_pi = pm->ar.pi;
  return AZMI_OK;
}

int azmi_pm_poll(azmi_pm* pm, void* stream, uint32_t* games_completed, uint32_t* live_slots) {
  if (!pm) return fail(AZMI_ERR_INVALID, "null pm");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  Control c;
  const int rc = read_ctl(pm, pm->pick(stream), &c, true);
  if (rc != AZMI_OK) return rc;
  if (games_completed) *games_completed = c.games_completed;
  if (live_slots) *live_slots = c.stop ? 0u : c.live_slots;
  return AZMI_OK;
}

// ---- PlayManager::stop / stopped / queue sizes / one slot's GameState (play_manager.h:177-186, 285-324) ----------
int azmi_pm_stop(azmi_pm* pm) {
  if (!pm) return fail(AZMI_ERR_INVALID, "null pm");
  pm->stopped.store(true, std::memory_order_relaxed);
  return AZMI_OK;
}
int azmi_pm_stopped(azmi_pm* pm, int* out) {
  if (!pm || !out) return fail(AZMI_ERR_INVALID, "null argument");
  *out = pm->stopped.load(std::memory_order_relaxed) ? 1 : 0;
  return AZMI_OK;
}
// awaiting_inference_count(): leaves waiting to be handed out by build_batch / pop_games; awaiting_mcts_count(): live
// slots that hold their answer (or need none) and wait for the next round
int azmi_pm_queue_counts(azmi_pm* pm, uint32_t* awaiting_inference, uint32_t* awaiting_mcts) {
  if (!pm) return fail(AZMI_ERR_INVALID, "null pm");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  Control c;
  const int rc = read_ctl(pm, pm->last, &c, true);
  if (rc != AZMI_OK) return rc;
  uint32_t pend = 0;
  for (auto& q : pm->pending_g) pend += static_cast<uint32_t>(q.size());
  const uint32_t live = c.stop ? 0u : c.live_slots;
  if (awaiting_inference) *awaiting_inference = pend;
  if (awaiting_mcts) *awaiting_mcts = live > pend + pm->outstanding ? live - pend - pm->outstanding : 0u;
  return AZMI_OK;
}
// game_data(i).gs: the packed state words of slot `slot` (Connect4: stones of player 0, stones of player 1,
// turn | player << 32; Tafl family: defenders lo/hi, attackers lo/hi, king | turn << 8 | player << 24 | repetitions << 32)
int azmi_pm_slot_state(azmi_pm* pm, uint32_t slot, uint64_t* words, uint32_t cap, uint32_t* n) {
  if (!pm || !words || !n) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  if (slot >= pm->ep.S) return fail(AZMI_ERR_RANGE, "game index %u out of range", slot);
  const uint32_t W = pm->gi.state_words;
  if (cap < W + 1) return fail(AZMI_ERR_INVALID, "words too small");
  HIP_TRY(hipStreamSynchronize(pm->last));
  for (uint32_t w = 0; w < W; ++w)
    HIP_TRY(hipMemcpy(words + w, pm->ar.gs_words + static_cast<size_t>(w) * pm->ep.S + slot, 8, hipMemcpyDeviceToHost));
  uint32_t perm = 0;
  HIP_TRY(hipMemcpy(&perm, pm->ar.perm + slot, 4, hipMemcpyDeviceToHost));
  words[W] = perm;      // GameData::perm_index, play_manager.h:41
  *n = W + 1;
  return AZMI_OK;
}

// game_data(i).gs of a game with a position history (StarGambit: position_history_, star_gambit_gs.h:745): the slot's
// history entries (the reference's own position hashes since the last deploy)
int azmi_pm_slot_history(azmi_pm* pm, uint32_t slot, uint64_t* out, uint32_t cap, uint32_t* n) {
  if (!pm || !out || !n) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  if (slot >= pm->ep.S) return fail(AZMI_ERR_RANGE, "game index %u out of range", slot);
  if (pm->game == AZMI_GAME_CONNECT4) { *n = 0; return AZMI_OK; }
  HIP_TRY(hipStreamSynchronize(pm->last));
  uint32_t len = 0;
  HIP_TRY(hipMemcpy(&len, pm->ar.rep_len + slot, 4, hipMemcpyDeviceToHost));
  if (len > cap) return fail(AZMI_ERR_INVALID, "history of %u entries does not fit %u", len, cap);
  if (len) HIP_TRY(hipMemcpy(out, pm->ar.rep_list + static_cast<size_t>(slot) * (pm->gi.max_turns + 2), static_cast<size_t>(len) * 8, hipMemcpyDeviceToHost));
  *n = len;
  return AZMI_OK;
}

// game_data(i).canonical(): the planes of the leaf slot `slot` is waiting on (host array [C,H,W])
int azmi_pm_slot_canonical(azmi_pm* pm, uint32_t slot, float* out) {
  if (!pm || !out) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  if (slot >= pm->ep.S) return fail(AZMI_ERR_RANGE, "game index %u out of range", slot);
  const size_t CANON = static_cast<size_t>(pm->gi.C) * pm->gi.H * pm->gi.W;
  HIP_TRY(hipStreamSynchronize(pm->last));
  HIP_TRY(hipMemcpy(out, pm->ar.canon + slot * CANON, CANON * 4, hipMemcpyDeviceToHost));
  return AZMI_OK;
}

int azmi_pm_play(azmi_pm* pm, void* stream) {
  if (!pm) return fail(AZMI_ERR_INVALID, "null pm");
  if (!pm->all_random) return fail(AZMI_ERR_STATE, "azmi_pm_play needs EvalType::RANDOM on every seat; drive NN seats with azmi_pm_round");
  hipStream_t st = pm->pick(stream);
  for (;;) {
    if (pm->stopped.load(std::memory_order_relaxed)) return AZMI_OK;   // play_manager.cc:272
    // several threads may sit in play() at once (the reference starts mcts_workers of them): each takes the engine for a
    // chunk of rounds, all of them return when the games are done
    std::lock_guard<std::recursive_mutex> lock_(pm->mu);
    for (int r = 0; r < 32; ++r) {
      const int rc = launch_round(pm, st);
      if (rc != AZMI_OK) return rc;
    }
    Control c;
    const int rc = read_ctl(pm, st, &c, true);
    if (rc != AZMI_OK) return rc;
    if (c.stop) return AZMI_OK;
  }
}

int azmi_pm_scores(azmi_pm* pm, float* out) {
  if (!pm || !out) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  std::vector<float> a;
  const uint32_t V = pm->gi.P + 1;
  const int rc = d2h(a, pm->ar.a_scores, static_cast<size_t>(pm->ep.S) * V, pm->last);
  if (rc != AZMI_OK) return rc;
  for (uint32_t i = 0; i < V; ++i) out[i] = 0.0f;
  for (uint32_t s = 0; s < pm->ep.S; ++s) for (uint32_t i = 0; i < V; ++i) out[i] += a[static_cast<size_t>(s) * V + i];
  return AZMI_OK;
}
int azmi_pm_resign_scores(azmi_pm* pm, float* out) {
  if (!pm || !out) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  std::vector<float> a;
  const uint32_t V = pm->gi.P + 1;
  const int rc = d2h(a, pm->ar.a_resign, static_cast<size_t>(pm->ep.S) * V, pm->last);
  if (rc != AZMI_OK) return rc;
  for (uint32_t i = 0; i < V; ++i) out[i] = 0.0f;
  for (uint32_t s = 0; s < pm->ep.S; ++s) for (uint32_t i = 0; i < V; ++i) out[i] += a[static_cast<size_t>(s) * V + i];
  return AZMI_OK;
}

// the sums behind azmi_pm_stats, for callers that combine several engines (shards / ranks) into one set of averages:
// out[10] = game_length, games completed, total / full / fast move counts, leaf depth, entropy, fast leaf depth,
//           fast entropy, valid moves (the accumulators of play_manager.h:398-424)
int azmi_pm_stat_sums(azmi_pm* pm, double* out) {
  if (!pm || !out) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  const uint32_t S = pm->ep.S;
  std::vector<uint64_t> len, cnt; std::vector<double> ds; std::vector<uint32_t> games;
  int rc = d2h(len, pm->ar.a_len, S, pm->last); if (rc) return rc;
  rc = d2h(cnt, pm->ar.a_cnt, 3 * static_cast<size_t>(S), pm->last); if (rc) return rc;
  rc = d2h(ds, pm->ar.a_dsum, 5 * static_cast<size_t>(S), pm->last); if (rc) return rc;
  rc = d2h(games, pm->ar.slot_games, S, pm->last); if (rc) return rc;
  for (int i = 0; i < 10; ++i) out[i] = 0.0;
  for (uint32_t s = 0; s < S; ++s) {
    out[0] += static_cast<double>(len[s]); out[1] += games[s];
    for (int j = 0; j < 3; ++j) out[2 + j] += static_cast<double>(cnt[static_cast<size_t>(j) * S + s]);
    for (int j = 0; j < 5; ++j) out[5 + j] += ds[static_cast<size_t>(j) * S + s];
  }
  return AZMI_OK;
}

uint32_t azmi_pm_num_variants(azmi_pm* pm) { return pm && pm->game == AZMI_GAME_STARGAMBIT ? 4u : 0u; }   // num_variants(), star_gambit_gs.h:863
int azmi_pm_variant_sums(azmi_pm* pm, uint32_t variant, float* perm_scores, uint32_t* perm_games, double* sums) {
  if (!pm || !perm_scores || !perm_games || !sums) return fail(AZMI_ERR_INVALID, "null argument");
  if (variant >= azmi_pm_num_variants(pm)) return fail(AZMI_ERR_RANGE, "variant %u out of range", variant);
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  const uint32_t S = pm->ep.S, NP = pm->ep.num_perms, V = pm->gi.P + 1;
  std::vector<float> sc; std::vector<uint32_t> gm; std::vector<uint64_t> len, cnt; std::vector<double> ds;
  int rc = d2h(sc, pm->ar.a_var_scores, static_cast<size_t>(S) * 4 * NP * V, pm->last); if (rc) return rc;
  rc = d2h(gm, pm->ar.a_var_games, static_cast<size_t>(S) * 4 * NP, pm->last); if (rc) return rc;
  rc = d2h(len, pm->ar.a_var_len, static_cast<size_t>(S) * 4, pm->last); if (rc) return rc;
  rc = d2h(ds, pm->ar.a_var_dsum, static_cast<size_t>(S) * 4 * 5, pm->last); if (rc) return rc;
  rc = d2h(cnt, pm->ar.a_var_cnt, static_cast<size_t>(S) * 4 * 3, pm->last); if (rc) return rc;
  for (uint32_t i = 0; i < NP * V; ++i) perm_scores[i] = 0.0f;
  for (uint32_t i = 0; i < NP; ++i) perm_games[i] = 0;
  for (int i = 0; i < 10; ++i) sums[i] = 0.0;
  for (uint32_t s = 0; s < S; ++s) {      // slot order: a deterministic sum
    const size_t sv = static_cast<size_t>(s) * 4 + variant;
    for (uint32_t q = 0; q < NP; ++q) {
      for (uint32_t i = 0; i < V; ++i) perm_scores[q * V + i] += sc[(sv * NP + q) * V + i];
      perm_games[q] += gm[sv * NP + q];
      sums[1] += gm[sv * NP + q];
    }
    sums[0] += static_cast<double>(len[sv]);
    for (int j = 0; j < 3; ++j) sums[2 + j] += static_cast<double>(cnt[sv * 3 + j]);
    for (int j = 0; j < 5; ++j) sums[5 + j] += ds[sv * 5 + j];
  }
  return AZMI_OK;
}

int azmi_pm_stats(azmi_pm* pm, float* out) {  // play_manager.h:288-315
  if (!pm || !out) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  const uint32_t S = pm->ep.S;
  std::vector<uint64_t> len, cnt; std::vector<double> ds; std::vector<uint32_t> games;
  int rc = d2h(len, pm->ar.a_len, S, pm->last); if (rc) return rc;
  rc = d2h(cnt, pm->ar.a_cnt, 3 * static_cast<size_t>(S), pm->last); if (rc) return rc;
  rc = d2h(ds, pm->ar.a_dsum, 5 * static_cast<size_t>(S), pm->last); if (rc) return rc;
  rc = d2h(games, pm->ar.slot_games, S, pm->last); if (rc) return rc;
  uint64_t game_length = 0, mc[3] = {0, 0, 0}, completed = 0; double d[5] = {0, 0, 0, 0, 0};
  for (uint32_t s = 0; s < S; ++s) {
    game_length += len[s]; completed += games[s];
    for (int j = 0; j < 3; ++j) mc[j] += cnt[static_cast<size_t>(j) * S + s];
    for (int j = 0; j < 5; ++j) d[j] += ds[static_cast<size_t>(j) * S + s];
  }
  out[0] = static_cast<float>(game_length) / static_cast<float>(completed);
  out[1] = mc[1] == 0 ? 0.0f : static_cast<float>(d[0] / static_cast<double>(mc[1]));
  out[2] = mc[1] == 0 ? 0.0f : static_cast<float>(d[1] / static_cast<double>(mc[1]));
  out[3] = mc[2] == 0 ? 0.0f : static_cast<float>(d[2] / static_cast<double>(mc[2]));
  out[4] = mc[2] == 0 ? 0.0f : static_cast<float>(d[3] / static_cast<double>(mc[2]));
  out[5] = game_length == 0 ? 0.0f : static_cast<float>(mc[0]) / static_cast<float>(game_length);
  out[6] = mc[0] == 0 ? 0.0f : static_cast<float>(d[4] / static_cast<double>(mc[0]));
  return AZMI_OK;
}

// cache_hits / misses / evictions / reinserts / size / max_size summed over the model groups' caches (play_manager.h:325-366)
int azmi_pm_cache_stats(azmi_pm* pm, uint64_t out[6]) {
  if (!pm || !out) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  for (int i = 0; i < 6; ++i) out[i] = 0;
  if (!pm->ep.cache_on) return AZMI_OK;
  for (uint32_t g = 0; g < pm->ep.num_groups; ++g) {
    if (!pm->group_cache_counted[g]) continue;
    const CacheView& c = pm->group_caches[g];
    std::vector<unsigned long long> st; std::vector<uint32_t> state;
    int rc = d2h(st, c.stats, static_cast<size_t>(c.shards) * 4, pm->last); if (rc) return rc;
    rc = d2h(state, c.state, static_cast<size_t>(c.shards) * 8, pm->last); if (rc) return rc;
    for (uint32_t s = 0; s < c.shards; ++s) {
      for (int j = 0; j < 4; ++j) out[j] += st[static_cast<size_t>(s) * 4 + j];
      out[4] += state[static_cast<size_t>(s) * 8 + kSize];
    }
    out[5] += static_cast<uint64_t>(c.cap) * c.shards;
  }
  // the pipeline's in-epoch answer table answers probes the S3-FIFO shard counted as misses (pipe_types.h)
  const unsigned long long l0 = azmi_host_pipe_l0_hits(pm, pm->last);
  out[0] += l0; out[1] -= std::min<uint64_t>(out[1], l0);
  return AZMI_OK;
}

int azmi_pm_counters(azmi_pm* pm, uint64_t* out) {
  if (!pm || !out) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  const uint32_t S = pm->ep.S;
  std::vector<uint64_t> sims, evals;
  int rc = d2h(sims, pm->ar.c_sims, S, pm->last); if (rc) return rc;
  rc = d2h(evals, pm->ar.c_evals, S, pm->last); if (rc) return rc;
  Control c;
  rc = read_ctl(pm, pm->last, &c, true); if (rc) return rc;
  out[0] = out[1] = 0;
  for (uint32_t s = 0; s < S; ++s) { out[0] += sims[s]; out[1] += evals[s]; }
  uint64_t cs[6];
  rc = azmi_pm_cache_stats(pm, cs); if (rc) return rc;
  out[2] = cs[0]; out[3] = cs[1];
  out[4] = c.hist_rows - pm->hist_read;   // free-running 64-bit counters: the difference is the live row count
  out[5] = c.rounds;
  return AZMI_OK;
}

namespace {
// tells the device how far the host has consumed the finished-sample ring (ordered behind the rounds already queued)
int publish_hist_read(azmi_pm* pm) {
  HIP_TRY(hipMemcpyAsync(&pm->ar.ctl->hist_read, &pm->hist_read, sizeof(pm->hist_read), hipMemcpyHostToDevice, pm->last));
  HIP_TRY(hipStreamSynchronize(pm->last));
  return AZMI_OK;
}
}  // namespace

int azmi_pm_pop_history(azmi_pm* pm, float* canonical, float* v, float* pi, uint32_t cap, uint32_t* n) {
  if (!pm || !n) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  Control c;
  int rc = read_ctl(pm, pm->last, &c, true); if (rc) return rc;
  const uint32_t avail = static_cast<uint32_t>(c.hist_rows - pm->hist_read);
  const uint32_t take = std::min(avail, cap);
  const uint32_t CANON = pm->gi.C * pm->gi.H * pm->gi.W, V = pm->gi.P + 1, M = pm->gi.M;
  if (take) {
    const uint32_t first = static_cast<uint32_t>(pm->hist_read % pm->ep.hist_cap);
    const uint32_t n1 = std::min(take, pm->ep.hist_cap - first);
    for (int seg = 0; seg < 2; ++seg) {     // the window may wrap around the end of the ring
      const size_t r0 = seg == 0 ? first : 0, nr = seg == 0 ? n1 : take - n1, o = seg == 0 ? 0 : n1;
      if (!nr) continue;
      HIP_TRY(hipMemcpy(canonical + o * CANON, pm->ar.h_canon + r0 * CANON, nr * CANON * 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(v + o * V, pm->ar.h_v + r0 * V, nr * V * 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(pi + o * M, pm->ar.h_pi + r0 * M, nr * M * 4, hipMemcpyDeviceToHost));
    }
    pm->hist_read += take;
    rc = publish_hist_read(pm); if (rc) return rc;
  }
  *n = take;
  return AZMI_OK;
}

int azmi_pm_history_device(azmi_pm* pm, float** dev_canonical, float** dev_v, float** dev_pi, uint32_t** dev_meta,
                           uint32_t* rows) {
  if (!pm) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  Control c;
  int rc = read_ctl(pm, pm->last, &c, true); if (rc) return rc;
  if (dev_canonical) *dev_canonical = pm->ar.h_canon;
  if (dev_v) *dev_v = pm->ar.h_v;
  if (dev_pi) *dev_pi = pm->ar.h_pi;
  if (dev_meta) *dev_meta = pm->ar.h_meta;
  if (rows) *rows = static_cast<uint32_t>(std::min<unsigned long long>(c.hist_rows, pm->ep.hist_cap));   // rows of the run while nothing has wrapped; use the window call for a ring
  return AZMI_OK;
}

int azmi_pm_history_window(azmi_pm* pm, uint32_t* first_row, uint32_t* rows, uint32_t* capacity) {
  if (!pm) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  Control c;
  int rc = read_ctl(pm, pm->last, &c, true); if (rc) return rc;
  if (first_row) *first_row = pm->ep.hist_cap ? static_cast<uint32_t>(pm->hist_read % pm->ep.hist_cap) : 0;
  if (rows) *rows = static_cast<uint32_t>(c.hist_rows - pm->hist_read);
  if (capacity) *capacity = pm->ep.hist_cap;
  return AZMI_OK;
}

int azmi_pm_history_consume(azmi_pm* pm, uint32_t rows) {
  if (!pm) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  Control c;
  int rc = read_ctl(pm, pm->last, &c, true); if (rc) return rc;
  if (rows > c.hist_rows - pm->hist_read) return fail(AZMI_ERR_INVALID, "history_consume: %u rows asked, %u unread", rows, static_cast<uint32_t>(c.hist_rows - pm->hist_read));
  pm->hist_read += rows;
  return publish_hist_read(pm);
}

int azmi_pm_move_log(azmi_pm* pm, uint32_t* rows, uint32_t* counts, uint32_t cap, uint32_t* n) {
  if (!pm || !n) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  if (!pm->ep.log_moves) return fail(AZMI_ERR_STATE, "move log was not enabled (azmi_engine_opts.log_moves)");
  Control c;
  int rc = read_ctl(pm, pm->last, &c, true); if (rc) return rc;
  const uint32_t take = std::min(std::min(c.log_rows, pm->ep.log_cap), cap);
  if (take && rows) HIP_TRY(hipMemcpy(rows, pm->ar.log_rows, static_cast<size_t>(take) * 8 * 4, hipMemcpyDeviceToHost));
  if (take && counts) HIP_TRY(hipMemcpy(counts, pm->ar.log_counts, static_cast<size_t>(take) * pm->gi.M * 4, hipMemcpyDeviceToHost));
  *n = take;
  return AZMI_OK;
}

// debug: keys of the leaves the last round sent to the net (0 = none), one per slot
int azmi_debug_eval_keys(azmi_pm* pm, uint64_t* out, uint32_t cap, uint32_t* n) {
  if (!pm || !out || !n) return fail(AZMI_ERR_INVALID, "null argument");
  if (!pm->ep.cache_on) return fail(AZMI_ERR_STATE, "cache is off");
  std::vector<uint64_t> k;
  const int rc = d2h(k, pm->ar.cache_keys, pm->ep.S, pm->last);
  if (rc) return rc;
  const uint32_t cnt = std::min<uint32_t>(cap, pm->ep.S);
  std::memcpy(out, k.data(), static_cast<size_t>(cnt) * 8);
  *n = cnt;
  return AZMI_OK;
}

int azmi_debug_trace(azmi_pm* pm, uint64_t* out, uint32_t cap, uint32_t* n) {
  std::vector<uint64_t> t;
  const int rc = d2h(t, pm->ar.trace, 2 * static_cast<size_t>(pm->ep.trace_cap), pm->last);
  if (rc) return rc;
  const uint32_t cnt = static_cast<uint32_t>(std::min<uint64_t>(t[0], cap));
  std::memcpy(out, t.data() + 2, static_cast<size_t>(cnt) * 16);
  *n = cnt;
  return AZMI_OK;
}

int azmi_pm_slot_games(azmi_pm* pm, uint32_t* out) {
  if (!pm || !out) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  std::vector<uint32_t> g;
  const int rc = d2h(g, pm->ar.slot_games, pm->ep.S, pm->last);
  if (rc) return rc;
  std::memcpy(out, g.data(), g.size() * 4);
  return AZMI_OK;
}

// ---- host-buffer compatibility path ------------------------------------------------------------------
int azmi_pm_build_batch(azmi_pm* pm, float* batch, uint32_t cap, uint32_t* indices, uint32_t* n) {
  return azmi_pm_build_batch_group(pm, 0xFFFFFFFFu, batch, cap, indices, n);
}

// group == 0xFFFFFFFF: leaves of any model group (single-evaluator callers)
int azmi_pm_build_batch_group(azmi_pm* pm, uint32_t group, float* batch, uint32_t cap, uint32_t* indices, uint32_t* n) {
  if (!pm || !indices || !n) return fail(AZMI_ERR_INVALID, "null argument");   // batch == NULL: pop_games_upto (indices only)
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  *n = 0;
  if (cap == 0 || pm->stopped.load(std::memory_order_relaxed)) return AZMI_OK;
  if (group != 0xFFFFFFFFu && group >= pm->ep.num_groups) return fail(AZMI_ERR_INVALID, "model group %u out of range", group);
  const uint32_t S = pm->ep.S, CANON = pm->gi.C * pm->gi.H * pm->gi.W;
  hipStream_t st = pm->pick(AZMI_STREAM_ENGINE);
  int guard = 0;
  auto any_pending = [&]() { for (auto& q : pm->pending_g) if (!q.empty()) return true; return false; };
  auto mine = [&]() -> std::deque<uint32_t>* {
    if (group != 0xFFFFFFFFu) return pm->pending_g[group].empty() ? nullptr : &pm->pending_g[group];
    for (auto& q : pm->pending_g) if (!q.empty()) return &q;
    return nullptr;
  };
  while (!any_pending()) {
    if (pm->outstanding != 0) return AZMI_OK;  // rows handed out, answers not back yet
    Control c;
    int rc = read_ctl(pm, st, &c, true); if (rc) return rc;
    if (c.stop) return AZMI_OK;
    rc = launch_round(pm, st); if (rc) return rc;
    // the round's eval lists name exactly the leaves that need a network answer, per model group (a slot whose
    // last leaf was a cache hit or terminal is not listed); sorted so the hand-out order is deterministic
    Control c2;
    rc = read_ctl(pm, st, &c2, false); if (rc) return rc;
    for (uint32_t g = 0; g < pm->ep.num_groups; ++g) {
      const uint32_t cnt = std::min<uint32_t>(c2.eval_count[g], S);
      if (cnt == 0) continue;
      std::vector<uint32_t> lst;
      rc = d2h(lst, pm->ar.eval_list + static_cast<size_t>(g) * S, cnt, st); if (rc) return rc;
      std::sort(lst.begin(), lst.end());
      for (uint32_t sidx : lst) pm->pending_g[g].push_back(sidx);
    }
    if (++guard > (1 << 20)) return fail(AZMI_ERR_STATE, "build_batch made no progress");
  }
  std::deque<uint32_t>* q = mine();
  if (!q) return AZMI_OK;   // other groups have pending leaves, this one has none right now
  const uint32_t take = std::min<uint32_t>(cap, static_cast<uint32_t>(q->size()));
  for (uint32_t r = 0; r < take; ++r) {
    const uint32_t s = q->front();
    q->pop_front();
    indices[r] = s;
    if (batch) HIP_TRY(hipMemcpyAsync(batch + static_cast<size_t>(r) * CANON, pm->ar.canon + static_cast<size_t>(s) * CANON,
                           CANON * 4, hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  pm->outstanding += take;
  *n = take;
  return AZMI_OK;
}

int azmi_pm_update_inferences(azmi_pm* pm, const uint32_t* indices, uint32_t n, const float* v, const float* pi) {
  if (!pm || (n && (!indices || !v || !pi))) return fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  const uint32_t S = pm->ep.S, V = pm->gi.P + 1, M = pm->gi.M;
  if (n > pm->outstanding) return fail(AZMI_ERR_STATE, "update_inferences: more rows than build_batch handed out");
  // rows go straight to the slot-indexed device buffers: rows of other slots (cache hits written by the round
  // kernel) must not be touched, so there is no whole-buffer mirror upload
  for (uint32_t r = 0; r < n; ++r) {
    if (indices[r] >= S) return fail(AZMI_ERR_INVALID, "slot index out of range");
    HIP_TRY(hipMemcpyAsync(pm->ar.v + static_cast<size_t>(indices[r]) * V, v + static_cast<size_t>(r) * V, V * 4, hipMemcpyHostToDevice, pm->stream));
    HIP_TRY(hipMemcpyAsync(pm->ar.pi + static_cast<size_t>(indices[r]) * M, pi + static_cast<size_t>(r) * M, M * 4, hipMemcpyHostToDevice, pm->stream));
  }
  HIP_TRY(hipStreamSynchronize(pm->stream));
  pm->outstanding -= n;
  return AZMI_OK;
}

int azmi_game_replay(int game, int device, const int32_t* moves, uint32_t n, uint32_t len, uint8_t* valid,
                     float* scores, float* canonical, uint32_t* player, uint32_t* turn, uint64_t* key,
                     int32_t* status) {
  return azmi_game_replay_from(game, device, nullptr, 0, moves, n, len, valid, scores, canonical, player, turn, key, status);
}

int azmi_game_replay_from(int game, int device, const uint8_t* init, uint32_t init_stride, const int32_t* moves,
                          uint32_t n, uint32_t len, uint8_t* valid, float* scores, float* canonical,
                          uint32_t* player, uint32_t* turn, uint64_t* key, int32_t* status) {
  return azmi_game_replay_ex(game, device, init, init_stride, moves, n, len, valid, scores, canonical, player, turn, key, status, 0u);
}

namespace {
// start-position rows: Connect4 = the 89-byte to_bytes image; Tafl family = the reference pickle image (dev_games.h TaflImage),
// rows zero-padded to a common stride.  *extra_reps = the most repetition keys any row brings along.
int check_init_rows(int game, const uint8_t* init, uint32_t init_stride, uint32_t n, uint32_t* extra_reps) {
  *extra_reps = 0;
  if (!init) return AZMI_OK;
  if (game == AZMI_GAME_CONNECT4) {
    if (init_stride != Connect4::SERIALIZED) return fail(AZMI_ERR_INVALID, "start positions: Connect4 images are %u bytes", Connect4::SERIALIZED);
    return AZMI_OK;
  }
  if (game == AZMI_GAME_STARGAMBIT) {   // rows hold one StarGambitUnifiedGS::to_bytes image each, zero-padded; the history rides along
    if (init_stride < 25u + 24u) return fail(AZMI_ERR_INVALID, "start positions: a StarGambit image is at least 49 bytes, got %u", init_stride);
    for (uint32_t g = 0; g < n; ++g) {
      const uint8_t* row = init + static_cast<size_t>(g) * init_stride;
      const uint32_t inner = uint32_t(row[21]) | uint32_t(row[22]) << 8 | uint32_t(row[23]) << 16 | uint32_t(row[24]) << 24;
      if (25ull + inner > init_stride) return fail(AZMI_ERR_INVALID, "start position %u: image longer than the row", g);
      const uint32_t nu = uint32_t(row[25]) | uint32_t(row[26]) << 8 | uint32_t(row[27]) << 16 | uint32_t(row[28]) << 24;
      if (nu > 20u || 9ull * nu + 24ull > inner) return fail(AZMI_ERR_INVALID, "start position %u: malformed image", g);
      const uint8_t* hl = row + 25 + 9 * nu + 20;
      *extra_reps = std::max(*extra_reps, uint32_t(hl[0]) | uint32_t(hl[1]) << 8 | uint32_t(hl[2]) << 16 | uint32_t(hl[3]) << 24);
    }
    return AZMI_OK;
  }
  const uint32_t sq = game == AZMI_GAME_BRANDUBH ? Brandubh::SQ : 121u, bb = 3u * sq, header = bb + 6u, entry = bb + 2u;
  if (init_stride < header + 4u) return fail(AZMI_ERR_INVALID, "start positions: a Tafl image is at least %u bytes, got %u", header + 4u, init_stride);
  for (uint32_t g = 0; g < n; ++g) {
    const uint8_t* row = init + static_cast<size_t>(g) * init_stride;
    const uint8_t* h = row + header;
    const uint32_t cnt = uint32_t(h[0]) | uint32_t(h[1]) << 8 | uint32_t(h[2]) << 16 | uint32_t(h[3]) << 24;
    if (cnt > 4096u || header + 4u + static_cast<uint64_t>(cnt) * entry > init_stride)
      return fail(AZMI_ERR_INVALID, "start position %u: repetition entry count mismatch", g);
    uint32_t keys = 0;
    for (uint32_t i = 0; i < cnt; ++i) keys += row[header + 4u + static_cast<size_t>(i) * entry + bb + 1u];
    *extra_reps = std::max(*extra_reps, keys);
  }
  return AZMI_OK;
}
}  // namespace

int azmi_playout_eval(int game, int device, const uint8_t* init, uint32_t init_stride, const int32_t* moves, uint32_t n, uint32_t len,
                      const uint64_t* seeds, float* v, float* pi) {
  GameInfo gi;
  if (!game_info(game, &gi)) return fail(AZMI_ERR_INVALID, "unknown game id %d", game);
  if (!seeds || !v || !pi || (!moves && n * len)) return fail(AZMI_ERR_INVALID, "null argument");
  uint32_t extra_reps = 0;
  { const int rc_init = check_init_rows(game, init, init_stride, n, &extra_reps); if (rc_init != AZMI_OK) return rc_init; }
  if (n == 0) return AZMI_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(AZMI_ERR_NO_DEVICE, "no HIP device: libazmi has no CPU path");
  HIP_TRY(hipSetDevice(device));
  std::vector<void*> tmp;
  auto dalloc = [&](auto*& p, size_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(count * sizeof(*p), 4));
    if (e == hipSuccess) { tmp.push_back(q); p = static_cast<std::remove_reference_t<decltype(p)>>(q); }
    return e;
  };
  auto cleanup = [&]() { for (void* q : tmp) (void)hipFree(q); };
#define TRY3(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(AZMI_ERR_NO_DEVICE, "%s: %s", #x, hipGetErrorString(e_)); } } while (0)
  int32_t* d_moves = nullptr; uint8_t* d_init = nullptr; uint64_t* d_seeds = nullptr; float *d_v = nullptr, *d_pi = nullptr; int32_t* d_status = nullptr;
  const uint32_t V = gi.P + 1;
  TRY3(dalloc(d_moves, static_cast<size_t>(n) * len));
  if (len) TRY3(hipMemcpy(d_moves, moves, static_cast<size_t>(n) * len * 4, hipMemcpyHostToDevice));
  if (init) { TRY3(dalloc(d_init, static_cast<size_t>(n) * init_stride)); TRY3(hipMemcpy(d_init, init, static_cast<size_t>(n) * init_stride, hipMemcpyHostToDevice)); }
  TRY3(dalloc(d_seeds, n)); TRY3(hipMemcpy(d_seeds, seeds, static_cast<size_t>(n) * 8, hipMemcpyHostToDevice));
  TRY3(dalloc(d_v, static_cast<size_t>(n) * V)); TRY3(dalloc(d_pi, static_cast<size_t>(n) * gi.M)); TRY3(dalloc(d_status, n));
  if (game == AZMI_GAME_CONNECT4) {
    k_playout<Connect4><<<(n + 63) / 64, 64>>>(d_init, d_moves, n, len, d_seeds, d_v, d_pi, d_status);
  } else if (game == AZMI_GAME_STARGAMBIT) {
    uint64_t* d_rep = nullptr;
    const uint32_t stride = len + gi.max_turns + 4 + extra_reps;
    TRY3(dalloc(d_rep, static_cast<size_t>(n) * stride));
    k_playout_sg<<<n, 64>>>(d_init, init_stride, d_moves, n, len, d_rep, stride, d_seeds, d_v, d_pi, d_status);
  } else {
    uint64_t* d_rep = nullptr;
    const uint32_t stride = len + gi.max_turns + 4 + extra_reps;
    TRY3(dalloc(d_rep, static_cast<size_t>(n) * stride));
    if (game == AZMI_GAME_TAWLBWRDD) k_playout_tafl<Tawlbwrdd><<<(n + 63) / 64, 64>>>(d_init, init_stride, d_moves, n, len, d_rep, stride, d_seeds, d_v, d_pi, d_status);
    else if (game == AZMI_GAME_BRANDUBH) k_playout_tafl<Brandubh><<<(n + 63) / 64, 64>>>(d_init, init_stride, d_moves, n, len, d_rep, stride, d_seeds, d_v, d_pi, d_status);
    else k_playout_tafl<OpenTafl><<<(n + 63) / 64, 64>>>(d_init, init_stride, d_moves, n, len, d_rep, stride, d_seeds, d_v, d_pi, d_status);
  }
  TRY3(hipGetLastError());
  TRY3(hipDeviceSynchronize());
  std::vector<int32_t> st(n);
  TRY3(hipMemcpy(st.data(), d_status, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost));
  TRY3(hipMemcpy(v, d_v, static_cast<size_t>(n) * V * 4, hipMemcpyDeviceToHost));
  TRY3(hipMemcpy(pi, d_pi, static_cast<size_t>(n) * gi.M * 4, hipMemcpyDeviceToHost));
#undef TRY3
  cleanup();
  for (uint32_t i = 0; i < n; ++i) if (st[i]) return fail(AZMI_ERR_INVALID, "illegal move in the game record of state %u", i);
  return AZMI_OK;
}

int azmi_game_replay_ex(int game, int device, const uint8_t* init, uint32_t init_stride, const int32_t* moves,
                        uint32_t n, uint32_t len, uint8_t* valid, float* scores, float* canonical,
                        uint32_t* player, uint32_t* turn, uint64_t* key, int32_t* status, uint32_t flags) {
  GameInfo gi;
  if (!game_info(game, &gi)) return fail(AZMI_ERR_INVALID, "unknown game id %d", game);
  if (!moves && n * len) return fail(AZMI_ERR_INVALID, "null moves");
  uint32_t extra_reps = 0;
  { const int rc_init = check_init_rows(game, init, init_stride, n, &extra_reps); if (rc_init != AZMI_OK) return rc_init; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(AZMI_ERR_NO_DEVICE, "no HIP device: libazmi has no CPU path");
  HIP_TRY(hipSetDevice(device));
  const uint32_t CANON = gi.C * gi.H * gi.W, V = gi.P + 1;
  int32_t* d_moves = nullptr; uint8_t* d_valid = nullptr; float *d_scores = nullptr, *d_canon = nullptr;
  uint32_t *d_player = nullptr, *d_turn = nullptr; uint64_t* d_key = nullptr; int32_t* d_status = nullptr;
  std::vector<void*> tmp;
  auto dalloc = [&](void** p, size_t bytes) { hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 4)); if (e == hipSuccess) tmp.push_back(*p); return e; };
  auto cleanup = [&]() { for (void* q : tmp) (void)hipFree(q); };
#define TRY2(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(AZMI_ERR_NO_DEVICE, "%s: %s", #x, hipGetErrorString(e_)); } } while (0)
  TRY2(dalloc(reinterpret_cast<void**>(&d_moves), static_cast<size_t>(n) * len * 4));
  TRY2(hipMemcpy(d_moves, moves, static_cast<size_t>(n) * len * 4, hipMemcpyHostToDevice));
  uint8_t* d_init = nullptr;
  if (init && n) {
    TRY2(dalloc(reinterpret_cast<void**>(&d_init), static_cast<size_t>(n) * init_stride));
    TRY2(hipMemcpy(d_init, init, static_cast<size_t>(n) * init_stride, hipMemcpyHostToDevice));
  }
  if (valid) TRY2(dalloc(reinterpret_cast<void**>(&d_valid), static_cast<size_t>(n) * gi.M));
  if (scores) TRY2(dalloc(reinterpret_cast<void**>(&d_scores), static_cast<size_t>(n) * V * 4));
  if (canonical) TRY2(dalloc(reinterpret_cast<void**>(&d_canon), static_cast<size_t>(n) * CANON * 4));
  if (player) TRY2(dalloc(reinterpret_cast<void**>(&d_player), static_cast<size_t>(n) * 4));
  if (turn) TRY2(dalloc(reinterpret_cast<void**>(&d_turn), static_cast<size_t>(n) * 4));
  if (key) TRY2(dalloc(reinterpret_cast<void**>(&d_key), static_cast<size_t>(n) * 8));
  if (status) TRY2(dalloc(reinterpret_cast<void**>(&d_status), static_cast<size_t>(n) * 4));
  if (n) {
    switch (game) {
      case AZMI_GAME_CONNECT4:
        k_replay<Connect4><<<(n + 255) / 256, 256>>>(d_init, d_moves, n, len, d_valid, d_scores, d_canon, d_player, d_turn, d_key, d_status);
        break;
      case AZMI_GAME_TAWLBWRDD:
      case AZMI_GAME_BRANDUBH:
      case AZMI_GAME_OPENTAFL: {
        uint64_t* d_rep = nullptr;
        const uint32_t stride = len + 2 + extra_reps;
        TRY2(dalloc(reinterpret_cast<void**>(&d_rep), static_cast<size_t>(n) * stride * 8));
        if (game == AZMI_GAME_TAWLBWRDD)
          k_replay_tafl<Tawlbwrdd><<<(n + 63) / 64, 64>>>(d_init, init_stride, d_moves, n, len, d_rep, stride, d_valid, d_scores, d_canon, d_player, d_turn, d_key, d_status, flags);
        else if (game == AZMI_GAME_BRANDUBH)
          k_replay_tafl<Brandubh><<<(n + 63) / 64, 64>>>(d_init, init_stride, d_moves, n, len, d_rep, stride, d_valid, d_scores, d_canon, d_player, d_turn, d_key, d_status, flags);
        else
          k_replay_tafl<OpenTafl><<<(n + 63) / 64, 64>>>(d_init, init_stride, d_moves, n, len, d_rep, stride, d_valid, d_scores, d_canon, d_player, d_turn, d_key, d_status, flags);
        break;
      }
      case AZMI_GAME_STARGAMBIT: {
        uint64_t* d_rep = nullptr;
        const uint32_t stride = len + 4 + extra_reps;
        TRY2(dalloc(reinterpret_cast<void**>(&d_rep), static_cast<size_t>(n) * stride * 8));
        k_replay_sg<<<n, 64>>>(d_init, init_stride, d_moves, n, len, d_rep, stride, d_valid, d_scores, d_canon, d_player, d_turn, d_key, d_status, flags);
        break;
      }
      default: cleanup(); return fail(AZMI_ERR_INVALID, "game %d has no device kernels", game);
    }
    TRY2(hipGetLastError());
    TRY2(hipDeviceSynchronize());
  }
  if (valid) TRY2(hipMemcpy(valid, d_valid, static_cast<size_t>(n) * gi.M, hipMemcpyDeviceToHost));
  if (scores) TRY2(hipMemcpy(scores, d_scores, static_cast<size_t>(n) * V * 4, hipMemcpyDeviceToHost));
  if (canonical) TRY2(hipMemcpy(canonical, d_canon, static_cast<size_t>(n) * CANON * 4, hipMemcpyDeviceToHost));
  if (player) TRY2(hipMemcpy(player, d_player, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost));
  if (turn) TRY2(hipMemcpy(turn, d_turn, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost));
  if (key) TRY2(hipMemcpy(key, d_key, static_cast<size_t>(n) * 8, hipMemcpyDeviceToHost));
  if (status) TRY2(hipMemcpy(status, d_status, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost));
#undef TRY2
  cleanup();
  return AZMI_OK;
}

// StarGambitUnifiedGS::to_bytes (star_gambit_gs.cc:2451-2465) of n states given as start image + moves: rows of out_stride
// bytes (probs / pinned_variant fields zero: they belong to the caller's object), sizes in out_len
int azmi_sg_image(int device, const uint8_t* init, uint32_t init_stride, const int32_t* moves, uint32_t n, uint32_t len,
                  uint8_t* out, uint32_t out_stride, uint32_t* out_len, int32_t* status, uint32_t flags) {
  if (!out || !out_len || !status || (!moves && n * len)) return fail(AZMI_ERR_INVALID, "null argument");
  uint32_t extra_reps = 0;
  { const int rc_init = check_init_rows(AZMI_GAME_STARGAMBIT, init, init_stride, n, &extra_reps); if (rc_init != AZMI_OK) return rc_init; }
  if (n == 0) return AZMI_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(AZMI_ERR_NO_DEVICE, "no HIP device: libazmi has no CPU path");
  HIP_TRY(hipSetDevice(device));
  std::vector<void*> tmp;
  auto dalloc = [&](void** p, size_t bytes) { hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 4)); if (e == hipSuccess) tmp.push_back(*p); return e; };
  auto cleanup = [&]() { for (void* q : tmp) (void)hipFree(q); };
#define TRY4(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(AZMI_ERR_NO_DEVICE, "%s: %s", #x, hipGetErrorString(e_)); } } while (0)
  int32_t* d_moves = nullptr; uint8_t* d_init = nullptr; uint64_t* d_rep = nullptr; uint8_t* d_out = nullptr; uint32_t* d_len = nullptr; int32_t* d_status = nullptr;
  TRY4(dalloc(reinterpret_cast<void**>(&d_moves), static_cast<size_t>(n) * len * 4));
  if (len) TRY4(hipMemcpy(d_moves, moves, static_cast<size_t>(n) * len * 4, hipMemcpyHostToDevice));
  if (init) { TRY4(dalloc(reinterpret_cast<void**>(&d_init), static_cast<size_t>(n) * init_stride)); TRY4(hipMemcpy(d_init, init, static_cast<size_t>(n) * init_stride, hipMemcpyHostToDevice)); }
  const uint32_t stride = len + 4 + extra_reps;
  TRY4(dalloc(reinterpret_cast<void**>(&d_rep), static_cast<size_t>(n) * stride * 8));
  TRY4(dalloc(reinterpret_cast<void**>(&d_out), static_cast<size_t>(n) * out_stride));
  TRY4(dalloc(reinterpret_cast<void**>(&d_len), static_cast<size_t>(n) * 4));
  TRY4(dalloc(reinterpret_cast<void**>(&d_status), static_cast<size_t>(n) * 4));
  k_sg_image<<<n, 64>>>(d_init, init_stride, d_moves, n, len, d_rep, stride, d_out, out_stride, d_len, d_status, flags);
  TRY4(hipGetLastError());
  TRY4(hipDeviceSynchronize());
  TRY4(hipMemcpy(out, d_out, static_cast<size_t>(n) * out_stride, hipMemcpyDeviceToHost));
  TRY4(hipMemcpy(out_len, d_len, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost));
  TRY4(hipMemcpy(status, d_status, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost));
#undef TRY4
  cleanup();
  return AZMI_OK;
}

// ---- stand-alone MCTS object (py_wrapper.cc:192-220) on a one-slot engine -------------------------------------
struct azmi_mcts {
  azmi_pm* pm = nullptr;
  uint8_t* d_init = nullptr; int32_t* d_moves = nullptr; int32_t* d_out_moves = nullptr;
  uint32_t* d_len = nullptr; int32_t* d_status = nullptr; float* d_f = nullptr; uint32_t* d_u = nullptr;
  uint32_t moves_cap = 0, vec = 0;
  uint32_t init_bytes = 0;               // size of the image in d_init (0 = the game's initial position)
  WuArrays wu{};                         // Node::n_in_flight + MCTS::in_flight_ (mcts.h:24,171)
  uint32_t ifl_count = 0, ifl_cap = 0;
};

namespace {
// capacity of the start-position buffer: Connect4's 89 bytes; a Tafl pickle image with up to 512 repetition entries
uint32_t mcts_init_bytes(int game) {
  if (game == AZMI_GAME_STARGAMBIT) return 25u + 4u + 9u * 20u + 20u + 8u * (StarGambit::MAX_TURNS + 2);
  return game == AZMI_GAME_CONNECT4 ? Connect4::SERIALIZED : game == AZMI_GAME_BRANDUBH ? TaflImage<Brandubh>::bytes(512) : TaflImage<OpenTafl>::bytes(512);
}
}  // namespace

int azmi_mcts_create(int game, const azmi_mcts_config* cfg, uint64_t seed, int device, azmi_mcts** out) {
  if (!cfg || !out) return fail(AZMI_ERR_INVALID, "null argument");
  GameInfo gi;
  if (!game_info(game, &gi)) return fail(AZMI_ERR_INVALID, "unknown game id %d", game);
  if (cfg->num_players != gi.P || cfg->num_moves != gi.M) return fail(AZMI_ERR_INVALID, "MCTS(num_players, num_moves) do not match the game");
  // MCTS(..., relative_values, ...) (mcts.h:54): on the device the rotation is part of the game's instantiation
  if ((cfg->relative_values != 0) != (game == AZMI_GAME_STARGAMBIT))
    return fail(AZMI_ERR_INVALID, "relative_values must be the game's relative_values() (true for StarGambit only)");
  azmi_play_params p;
  azmi_play_params_default(&p);
  p.games_to_play = 1; p.concurrent_games = 1; p.max_batch_size = 1;
  p.num_mcts_visits = gi.P;
  const uint32_t sims = cfg->max_simulations ? cfg->max_simulations : 50000u;
  // Connect4: arena = (21 * visits + 42) * 7 nodes >= sims * 7.  Wide games: two halves of 4 x (visits + 16) x 240 nodes,
  // compacted after update_root when the active half fills up
  for (uint32_t i = 0; i < gi.P; ++i) p.mcts_visits[i] = game == AZMI_GAME_CONNECT4 ? (sims + 20) / 21 : std::min<uint32_t>(sims, 8000u);
  p.cpuct = cfg->cpuct; p.epsilon = cfg->epsilon; p.mcts_root_temp = cfg->root_policy_temp; p.fpu_reduction = cfg->fpu_reduction;
  p.root_fpu_zero = cfg->root_fpu_zero; p.shaped_dirichlet = cfg->shaped_dirichlet;
  p.gumbel_enabled = cfg->gumbel_enabled; p.gumbel_m = cfg->gumbel_m; p.gumbel_c_visit = cfg->gumbel_c_visit;
  p.gumbel_c_scale = cfg->gumbel_c_scale; p.gumbel_full = cfg->gumbel_full;
  p.num_model_groups_given = gi.P;
  for (uint32_t i = 0; i < gi.P; ++i) p.model_groups[i] = 0;
  azmi_engine_opts o;
  azmi_engine_opts_default(&o);
  o.seed = seed; o.device = device;
  auto m = new azmi_mcts();
  int rc = azmi_pm_create(game, &p, &o, &m->pm);
  if (rc != AZMI_OK) { delete m; return rc; }
  // the slot's stream is the object's stream; seed it directly (not through slot_seed) so that `seed` means what
  // MCTS::seed_thread_rng(seed) means in the reference tests
  {
    Pcg32 g; g.seed(seed);
    const uint64_t st = g.state;
    if (hipMemcpy(m->pm->ar.rng, &st, 8, hipMemcpyHostToDevice) != hipSuccess) { azmi_pm_destroy(m->pm); delete m; return fail(AZMI_ERR_NO_DEVICE, "rng init failed"); }
  }
  m->moves_cap = gi.max_turns + 8;
  m->vec = std::max<uint32_t>(gi.M, 64u);
  auto A = [&](auto*& ptr, size_t n) { return m->pm->alloc(ptr, n, true); };
  rc = A(m->d_init, mcts_init_bytes(game)); if (rc == AZMI_OK) rc = A(m->d_moves, m->moves_cap); if (rc == AZMI_OK) rc = A(m->d_out_moves, m->moves_cap);
  if (rc == AZMI_OK) rc = A(m->d_len, 1); if (rc == AZMI_OK) rc = A(m->d_status, 1);
  if (rc == AZMI_OK) rc = A(m->d_f, m->vec); if (rc == AZMI_OK) rc = A(m->d_u, m->vec + 64);
  m->ifl_cap = 1024;
  if (rc == AZMI_OK) rc = A(m->wu.nif, static_cast<size_t>(gi.P) * m->pm->ep.cap);
  if (rc == AZMI_OK) rc = A(m->wu.ifl_path, static_cast<size_t>(m->ifl_cap) * m->pm->ep.max_depth);
  if (rc == AZMI_OK) rc = A(m->wu.ifl_plen, m->ifl_cap); if (rc == AZMI_OK) rc = A(m->wu.ifl_cur, m->ifl_cap);
  if (rc != AZMI_OK) { azmi_pm_destroy(m->pm); delete m; return rc; }
  *out = m;
  return AZMI_OK;
}

void azmi_mcts_destroy(azmi_mcts* m) {
  if (!m) return;
  azmi_pm_destroy(m->pm);
  delete m;
}

namespace {
int mcts_upload_state(azmi_mcts* m, const uint8_t* init, uint32_t init_bytes, const int32_t* moves, uint32_t len, hipStream_t st) {
  if (len > m->moves_cap) return fail(AZMI_ERR_INVALID, "game record too long");
  if (init) {
    uint32_t extra = 0;
    const int rc = check_init_rows(m->pm->game, init, init_bytes, 1, &extra);
    if (rc != AZMI_OK) return rc;
    if (init_bytes > mcts_init_bytes(m->pm->game)) return fail(AZMI_ERR_INVALID, "start position: image too large (%u bytes)", init_bytes);
  }
  m->init_bytes = init ? init_bytes : 0;
  if (init) HIP_TRY(hipMemcpyAsync(m->d_init, init, init_bytes, hipMemcpyHostToDevice, st));
  if (len) HIP_TRY(hipMemcpyAsync(m->d_moves, moves, static_cast<size_t>(len) * 4, hipMemcpyHostToDevice, st));
  return AZMI_OK;
}
int mcts_check(azmi_mcts* m, hipStream_t st) {
  Control c;
  return read_ctl(m->pm, st, &c, false);
}
}  // namespace

int azmi_mcts_find_leaf(azmi_mcts* m, const uint8_t* init, uint32_t init_bytes, const int32_t* moves, uint32_t len,
                        int32_t* leaf_moves, uint32_t cap, uint32_t* leaf_len) {
  if (!m || !leaf_len || (len && !moves)) return fail(AZMI_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(m->pm->device));
  hipStream_t st = m->pm->stream;
  int rc = mcts_upload_state(m, init, init_bytes, moves, len, st); if (rc) return rc;
  const uint8_t* di = init ? m->d_init : nullptr;
  switch (m->pm->game) {
    case AZMI_GAME_CONNECT4: k_mcts_find_leaf<Connect4><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu.nif, di, m->d_moves, len, m->d_out_moves, m->d_len, m->d_status); break;
    case AZMI_GAME_TAWLBWRDD: k_mcts_big_find_leaf<Tawlbwrdd><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu.nif, di, init_bytes, m->d_moves, len, m->d_out_moves, m->d_len, m->d_status); break;
    case AZMI_GAME_BRANDUBH: k_mcts_big_find_leaf<Brandubh><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu.nif, di, init_bytes, m->d_moves, len, m->d_out_moves, m->d_len, m->d_status); break;
    case AZMI_GAME_STARGAMBIT: k_mcts_big_find_leaf<StarGambit><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu.nif, di, init_bytes, m->d_moves, len, m->d_out_moves, m->d_len, m->d_status); break;
    default: k_mcts_big_find_leaf<OpenTafl><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu.nif, di, init_bytes, m->d_moves, len, m->d_out_moves, m->d_len, m->d_status); break;
  }
  int32_t status = 0; uint32_t n = 0;
  HIP_TRY(hipMemcpyAsync(&status, m->d_status, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(&n, m->d_len, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (status == -1) return fail(AZMI_ERR_INVALID, "illegal move in the game record");
  rc = mcts_check(m, st); if (rc) return rc;
  if (status != 0) return fail(AZMI_ERR_OVERFLOW, "find_leaf failed (tree arena or path capacity)");
  if (n > cap) return fail(AZMI_ERR_INVALID, "leaf_moves too small");
  if (n && leaf_moves) HIP_TRY(hipMemcpy(leaf_moves, m->d_out_moves, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost));
  *leaf_len = n;
  return AZMI_OK;
}

int azmi_mcts_process_result(azmi_mcts* m, const float* value, const float* pi, int root_noise_enabled, float* value_out) {
  if (!m || !value || !pi) return fail(AZMI_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(m->pm->device));
  hipStream_t st = m->pm->stream;
  const uint32_t V = m->pm->gi.P + 1, M = m->pm->gi.M;
  HIP_TRY(hipMemcpyAsync(m->pm->ar.v, value, V * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(m->pm->ar.pi, pi, M * 4, hipMemcpyHostToDevice, st));
  const uint32_t rn = root_noise_enabled ? 1u : 0u;
  switch (m->pm->game) {
    case AZMI_GAME_CONNECT4: k_mcts_process_result<Connect4><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, rn, m->d_f); break;
    case AZMI_GAME_TAWLBWRDD: k_mcts_big_process_result<Tawlbwrdd><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, rn, m->d_f); break;
    case AZMI_GAME_BRANDUBH: k_mcts_big_process_result<Brandubh><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, rn, m->d_f); break;
    case AZMI_GAME_STARGAMBIT: k_mcts_big_process_result<StarGambit><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, rn, m->d_f); break;
    default: k_mcts_big_process_result<OpenTafl><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, rn, m->d_f); break;
  }
  float tmp[8];
  HIP_TRY(hipMemcpyAsync(tmp, m->d_f, V * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (value_out) std::memcpy(value_out, tmp, V * 4);
  return mcts_check(m, st);
}

// ---- WU-UCT batched API, mcts.cc:752-851 -----------------------------------------------------------------------
int azmi_mcts_find_leaf_batched(azmi_mcts* m, const uint8_t* init, uint32_t init_bytes, const int32_t* moves, uint32_t len,
                                int32_t* leaf_moves, uint32_t cap, uint32_t* leaf_len) {
  if (!m || !leaf_len || (len && !moves)) return fail(AZMI_ERR_INVALID, "null argument");
  if (m->ifl_count >= m->ifl_cap) return fail(AZMI_ERR_OVERFLOW, "%u leaves in flight: call reset_batch", m->ifl_count);
  HIP_TRY(hipSetDevice(m->pm->device));
  hipStream_t st = m->pm->stream;
  int rc = mcts_upload_state(m, init, init_bytes, moves, len, st); if (rc) return rc;
  const uint8_t* di = init ? m->d_init : nullptr;
  const uint32_t idx = m->ifl_count;
  switch (m->pm->game) {
    case AZMI_GAME_CONNECT4: k_mcts_find_leaf_batched<Connect4><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu, idx, di, m->d_moves, len, m->d_out_moves, m->d_len, m->d_status); break;
    case AZMI_GAME_TAWLBWRDD: k_mcts_big_find_leaf_batched<Tawlbwrdd><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu, idx, di, init_bytes, m->d_moves, len, m->d_out_moves, m->d_len, m->d_status); break;
    case AZMI_GAME_BRANDUBH: k_mcts_big_find_leaf_batched<Brandubh><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu, idx, di, init_bytes, m->d_moves, len, m->d_out_moves, m->d_len, m->d_status); break;
    case AZMI_GAME_STARGAMBIT: k_mcts_big_find_leaf_batched<StarGambit><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu, idx, di, init_bytes, m->d_moves, len, m->d_out_moves, m->d_len, m->d_status); break;
    default: k_mcts_big_find_leaf_batched<OpenTafl><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu, idx, di, init_bytes, m->d_moves, len, m->d_out_moves, m->d_len, m->d_status); break;
  }
  int32_t status = 0; uint32_t n = 0;
  HIP_TRY(hipMemcpyAsync(&status, m->d_status, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(&n, m->d_len, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (status == -1) return fail(AZMI_ERR_INVALID, "illegal move in the game record");
  rc = mcts_check(m, st); if (rc) return rc;
  if (status != 0) return fail(AZMI_ERR_OVERFLOW, "find_leaf_batched failed (tree arena or path capacity)");
  if (n > cap) return fail(AZMI_ERR_INVALID, "leaf_moves too small");
  if (n && leaf_moves) HIP_TRY(hipMemcpy(leaf_moves, m->d_out_moves, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost));
  *leaf_len = n;
  ++m->ifl_count;
  return AZMI_OK;
}

int azmi_mcts_process_result_batched(azmi_mcts* m, uint32_t leaf_index, const float* value, const float* pi, int root_noise_enabled,
                                     float* value_out) {
  if (!m || !value || !pi) return fail(AZMI_ERR_INVALID, "null argument");
  if (leaf_index >= m->ifl_count) return fail(AZMI_ERR_RANGE, "leaf_index %u out of range (%u in flight)", leaf_index, m->ifl_count);
  HIP_TRY(hipSetDevice(m->pm->device));
  hipStream_t st = m->pm->stream;
  const uint32_t V = m->pm->gi.P + 1, M = m->pm->gi.M;
  HIP_TRY(hipMemcpyAsync(m->pm->ar.v, value, V * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(m->pm->ar.pi, pi, M * 4, hipMemcpyHostToDevice, st));
  const uint32_t rn = root_noise_enabled ? 1u : 0u;
  switch (m->pm->game) {
    case AZMI_GAME_CONNECT4: k_mcts_process_result_batched<Connect4><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu, leaf_index, rn, m->d_f); break;
    case AZMI_GAME_TAWLBWRDD: k_mcts_big_process_result_batched<Tawlbwrdd><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu, leaf_index, rn, m->d_f); break;
    case AZMI_GAME_BRANDUBH: k_mcts_big_process_result_batched<Brandubh><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu, leaf_index, rn, m->d_f); break;
    case AZMI_GAME_STARGAMBIT: k_mcts_big_process_result_batched<StarGambit><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu, leaf_index, rn, m->d_f); break;
    default: k_mcts_big_process_result_batched<OpenTafl><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu, leaf_index, rn, m->d_f); break;
  }
  float tmp[8];
  HIP_TRY(hipMemcpyAsync(tmp, m->d_f, V * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (value_out) std::memcpy(value_out, tmp, V * 4);
  return mcts_check(m, st);
}

int azmi_mcts_in_flight_count(const azmi_mcts* m, uint32_t* out) {
  if (!m || !out) return fail(AZMI_ERR_INVALID, "null argument");
  *out = m->ifl_count;
  return AZMI_OK;
}

int azmi_mcts_reset_batch(azmi_mcts* m) {
  if (!m) return fail(AZMI_ERR_INVALID, "null argument");
  m->ifl_count = 0;
  return AZMI_OK;
}

int azmi_mcts_update_root(azmi_mcts* m, const uint8_t* init, uint32_t init_bytes, const int32_t* moves, uint32_t len, uint32_t move) {
  if (!m || (len && !moves)) return fail(AZMI_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(m->pm->device));
  hipStream_t st = m->pm->stream;
  int rc = mcts_upload_state(m, init, init_bytes, moves, len, st); if (rc) return rc;
  const uint8_t* di = init ? m->d_init : nullptr;
  const uint32_t trees = m->pm->gi.P;
  switch (m->pm->game) {
    case AZMI_GAME_CONNECT4: k_mcts_update_root<Connect4><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu.nif, di, m->d_moves, len, move, m->d_status); break;
    case AZMI_GAME_TAWLBWRDD:
      k_mcts_big_update_root<Tawlbwrdd><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu.nif, di, init_bytes, m->d_moves, len, move, m->d_status);
      if (m->pm->ep.half_nodes) k_compact<Tawlbwrdd><<<std::min<uint32_t>(trees, kCompactBlocks), 256, 0, st>>>(m->pm->ep, m->pm->ar, trees, m->wu.nif);
      break;
    case AZMI_GAME_BRANDUBH:
      k_mcts_big_update_root<Brandubh><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu.nif, di, init_bytes, m->d_moves, len, move, m->d_status);
      if (m->pm->ep.half_nodes) k_compact<Brandubh><<<std::min<uint32_t>(trees, kCompactBlocks), 256, 0, st>>>(m->pm->ep, m->pm->ar, trees, m->wu.nif);
      break;
    case AZMI_GAME_STARGAMBIT:
      k_mcts_big_update_root<StarGambit><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu.nif, di, init_bytes, m->d_moves, len, move, m->d_status);
      if (m->pm->ep.half_nodes) k_compact<StarGambit><<<std::min<uint32_t>(trees, kCompactBlocks), 256, 0, st>>>(m->pm->ep, m->pm->ar, trees, m->wu.nif);
      break;
    default:
      k_mcts_big_update_root<OpenTafl><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, m->wu.nif, di, init_bytes, m->d_moves, len, move, m->d_status);
      if (m->pm->ep.half_nodes) k_compact<OpenTafl><<<std::min<uint32_t>(trees, kCompactBlocks), 256, 0, st>>>(m->pm->ep, m->pm->ar, trees, m->wu.nif);
      break;
  }
  int32_t status = 0;
  HIP_TRY(hipMemcpyAsync(&status, m->d_status, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (status == -1) return fail(AZMI_ERR_INVALID, "illegal move in the game record");
  if (status == -3) {   // the device raised its "unknown move" bit; clear it so the object stays usable
    Control c; HIP_TRY(hipMemcpy(&c, m->pm->ar.ctl, sizeof(c), hipMemcpyDeviceToHost));
    c.overflow &= ~32u; if (!c.overflow) c.stop = 0;
    HIP_TRY(hipMemcpy(m->pm->ar.ctl, &c, sizeof(c), hipMemcpyHostToDevice));
    return fail(AZMI_ERR_INVALID, "ahh, what is this move: %u", move);
  }
  return mcts_check(m, st);
}

int azmi_mcts_query(azmi_mcts* m, uint32_t kind, float temp, uint32_t arg, const float* in_f, float* out_f, uint32_t* out_u) {
  if (!m) return fail(AZMI_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(m->pm->device));
  hipStream_t st = m->pm->stream;
  if (kind == kQPickMove) {
    if (!in_f) return fail(AZMI_ERR_INVALID, "pick_move needs a probability vector");
    HIP_TRY(hipMemcpyAsync(m->d_f, in_f, static_cast<size_t>(m->pm->gi.M) * 4, hipMemcpyHostToDevice, st));
  }
  if (kind == kQPrincipalVariation && arg > 60) arg = 60;
  switch (m->pm->game) {
    case AZMI_GAME_CONNECT4: k_mcts_query<Connect4><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, kind, temp, arg, m->d_f, m->d_u); break;
    case AZMI_GAME_TAWLBWRDD: k_mcts_big_query<Tawlbwrdd><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, kind, temp, arg, m->d_f, m->d_u); break;
    case AZMI_GAME_BRANDUBH: k_mcts_big_query<Brandubh><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, kind, temp, arg, m->d_f, m->d_u); break;
    case AZMI_GAME_STARGAMBIT: k_mcts_big_query<StarGambit><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, kind, temp, arg, m->d_f, m->d_u); break;
    default: k_mcts_big_query<OpenTafl><<<1, 64, 0, st>>>(m->pm->ep, m->pm->ar, kind, temp, arg, m->d_f, m->d_u); break;
  }
  if (out_f) HIP_TRY(hipMemcpyAsync(out_f, m->d_f, static_cast<size_t>(m->vec) * 4, hipMemcpyDeviceToHost, st));
  if (out_u) HIP_TRY(hipMemcpyAsync(out_u, m->d_u, static_cast<size_t>(m->vec) * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return mcts_check(m, st);
}

}  // extern "C"

#ifdef AZMI_BIG_PROF
extern "C" int azmi_debug_big_prof(unsigned long long* out) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(azmi::g_big_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  unsigned long long z[16] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(azmi::g_big_prof), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
