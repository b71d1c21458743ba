// The asynchronous tree / net pipeline of the Connect4 engine (see pipe_types.h for the structure and the hand-off
// protocol): persistent tree wavefronts, mover wavefronts and net workgroups for one EPOCH, then - at a kernel boundary - the
// position-cache inserts of everything the net answered in the epoch, game restarts (k_assign of the lock-step engine) and the
// move steps the mover wavefronts did not get to.
//
// Reference behaviour restated: the worker loop of PlayManager::play (play_manager.cc:258-600) with its queues
// (concurrent_queue.h:130-217) and GameRunner's batcher / gpu_loop / result_worker threads (game_runner.py:483-552, 651-726);
// the search itself is k_sim's (engine_kernels.h): MCTS::process_result (mcts.cc:500-555), MCTS::find_leaf (mcts.cc:462-498),
// Node::add_children (mcts.cc:93-101), S3FIFOCache::find / insert (s3fifo_cache.h:41-80).
// A slot's game is a function of its seed alone (the answer to a position does not depend on where or when the net evaluates
// it), so the games are the lock-step engine's games: tests/test_gpu_pipeline.py.
#include <hip/hip_runtime.h>

#include <chrono>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>

#include "../../include/azmi.h"
#include "engine_host.h"
#define AZMI_KERNELS_NO_ASSIGN
#include "engine_kernels.h"
#include "leafnet_c4.h"
#include "pipe_types.h"
// The conveyor (round 5's weight-stationary net side: bit-identical to the tile kernel, slower - DESIGN 4.5b) is an EXPERIMENT since
// round 6: its device code lives in scripts/experiments/conveyor_c4.h and is compiled in only with -DAZMI_WITH_CONVEYOR
// (AZMI_HIPCC_EXTRA=-DAZMI_WITH_CONVEYOR python -c "import __graft_entry__ as g; g.build()"); the default library carries the host
// plumbing and answers AZMI_PIPE_NET=conveyor with an error.
#ifdef AZMI_WITH_CONVEYOR
#include "../../scripts/experiments/conveyor_c4.h"
#endif

using namespace azmi;

namespace azmi {

__device__ __forceinline__ unsigned long long g_ld(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t g_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void g_st(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void g_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }


enum : uint32_t { kGrpIdle = 0, kGrpReady = 1, kGrpWait = 2, kGrpPush = 3 };
// A FREEZE: two looks at the wall clock by one polling wavefront more than 2 ms apart.  No iteration of a polling loop takes that
// long on its own - the loops load a few words and sleep a microsecond -: the wavefront, and with it the chip, stood still (round 5:
// the stall-cap error of round 4, one in ~30,000 epochs, was every wavefront of both kernels spanning > 190 ms in ONE iteration - the
// GPU's own scheduler had switched the process's queues out, profiles/r5_repro_generic.txt).  The time caps are stall detectors for
// THIS code's protocol, so a freeze is credited: the wavefront's start time moves by the gap, the freeze is counted
// (PipeCtl::freezes, longest gap in PipeCtl::freeze_max) and reported with the call's statistics, not as an error.  Only a gap INSIDE
// a poll loop counts: the first look after a pass (a slow pass is not a freeze) only sets the mark.
constexpr uint64_t kFreezeTicks = 200000ull;
__device__ __forceinline__ void pipe_freeze_credit(PipeCtl* pc, uint64_t now, uint64_t& t_last, uint64_t& t_start, bool leader) {
  const uint64_t gap = now - t_last;
  if (gap > kFreezeTicks) {
    t_start += gap;
    if (leader) { atomicAdd(&pc->freezes, 1ull); atomicMax(&pc->freeze_max, static_cast<unsigned long long>(gap)); }
  }
  t_last = now;
}
constexpr uint64_t kMask48 = (1ull << 48) - 1;
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;

// one struct = the whole kernarg segment of k_pipe_tree: the move step (a function of its own, below) reads its arguments from
// there instead of from copies on the stack
struct PipeKernArgs { EngineParams ep; EngineArrays ar; PipeArrays pa; };

// ---- tree side ---------------------------------------------------------------------------------------------------------------------
// Every slot has a HOME workgroup (slot % tree workgroups).  A slot that can take its next step sits, as a token, in its home
// workgroup's READY ring (seeded by k_pipe_seed when the epoch starts, refilled by the net workgroups with the slots they have
// answered and by the workgroup's own wavefronts with slots whose next answer is already at hand or whose next step is the move
// step's).  A tree wavefront works in PASSES: look at the ring's head | tail word, draw what is there (at most eight tokens: one
// fetch-add), load the slots - one 8-lane group each -, back the pending simulations up and descend again until each group's new
// leaf needs the net (k_sim's body, engine_kernels.h: the path in registers, level i in lane i), store the slots, hand the leaves
// to the request ring.  A slot's state and trees are plain memory that only ever moves between the wavefronts of ONE workgroup =
// one CU = one vector L1: a pass drains its stores (s_waitcnt vmcnt(0)) before its tokens go out and that is all (round 3 let
// slots wander between CUs and paid an agent-scope acquire + release, ~10 us, per pass).
constexpr uint32_t kTreeWindow = 8;

__device__ __forceinline__ void pipe_push_token(const PipeArrays& pa, uint32_t home, uint32_t pos, unsigned long long payload) {
  g_st(pa.rring + ((static_cast<size_t>(home) << pa.rshift) + (pos & ((1u << pa.rshift) - 1u))), (pipe_lap_tag_r(pos, pa.rshift) << 48) | payload);
}

// request ring of model group g (0 or 1): its tail / head words and its entries
__device__ __forceinline__ uint32_t* pipe_tail_of(PipeCtl* pc, uint32_t g) { return g ? &pc->tail1 : &pc->tail; }
__device__ __forceinline__ uint32_t* pipe_head_of(PipeCtl* pc, uint32_t g) { return g ? &pc->head1 : &pc->head; }
__device__ __forceinline__ unsigned long long* pipe_ring_of(const PipeArrays& pa, uint32_t g) { return g ? pa.ring1 : pa.ring; }

// The epoch's first tokens: one thread per slot.  A slot whose leaf was left by the move step (kSlotQueued) sends its request, a
// slot with its answer in the (v, pi) rows gets a READY token, a slot whose game has to start a READY token with the move bit.
__global__ void k_pipe_seed(EngineParams ep, EngineArrays ar, PipeArrays pa) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= ep.S) return;
  PipeCtl* const pc = pa.ctl;
  PipeEpoch* const pe = pa.ep;
  if (ar.ctl->stop != 0 || pc->err != 0) { atomicAdd(&pe->dead, 1u); return; }
  const uint8_t sst = ar.sstate[slot];
  if (sst == kSlotDone || sst == kSlotEnded) { atomicAdd(&pe->dead, 1u); return; }
  const uint32_t home = slot % pa.n_tree_wgs;
  const uint8_t f = ar.flags[slot];
  if ((sst != kSlotWaitEval && sst != kSlotQueued) || (f & kFlagListed)) {       // kSlotFresh / kSlotRestart: a game start is the move step's
    const uint32_t pos = atomicAdd(&pa.wg[home].rtail, 1u);
    pipe_push_token(pa, home, pos, static_cast<unsigned long long>(slot) | kTokMove);
    return;
  }
  if (f & kFlagReqOut) { atomicOr(&pc->err, static_cast<uint32_t>(kPipeErrTag)); return; }      // (every request was settled)
  if (sst == kSlotQueued) {
    const uint32_t grp_ = pa.n_groups > 1u ? (ar.leaf_group[slot] & 1u) : 0u;
    const uint32_t pos = atomicAdd(pipe_tail_of(pc, grp_), 1u);
    uint32_t seq = ar.req_seq[slot] + 1u;
    if (seq == 0u) seq = 1u;
    ar.req_seq[slot] = seq;
    const unsigned long long tag = pipe_lap_tag(pos) << 48;
    unsigned long long* e = pipe_ring_of(pa, grp_) + static_cast<size_t>(pos & (kPipeRing - 1u)) * kReqGranules;
    g_st(e + 0, tag | (ar.leaf_pos[0 * static_cast<size_t>(ep.S) + slot] & kMask48));
    g_st(e + 1, tag | (ar.leaf_pos[1 * static_cast<size_t>(ep.S) + slot] & kMask48));
    g_st(e + 2, tag | static_cast<unsigned long long>(slot) | ((ar.leaf_pos[2 * static_cast<size_t>(ep.S) + slot] & 1ull) << 16));
    g_st(e + 3, tag | static_cast<unsigned long long>(seq));
    ar.flags[slot] = f | kFlagReqOut;
    ar.sstate[slot] = kSlotWaitEval;
    return;
  }
  const uint32_t pos = atomicAdd(&pa.wg[home].rtail, 1u);
  pipe_push_token(pa, home, pos, static_cast<unsigned long long>(slot));     // (answers no request: sequence field 0)
}

// ---- the move step inside the epoch ---------------------------------------------------------------------------------------------
// A slot a tree wavefront lists (its next backup completes a search, or its leaf is the root) and a slot whose game has to start
// come back to their home workgroup as tokens with the move bit; the wavefront that draws them runs the lock-step engine's own
// move step (round_slot<kMover>: backup, move, history row, re-rooting, game step / game end, first descent of the next search)
// for those groups before the pass proper.  What a group gets back: kSlotWaitEval - the next answer is at hand, the group goes on
// into the pass like any READY slot; kSlotQueued - the new leaf is a request now (sent here), the group sits this pass out;
// kSlotEnded / kSlotDone - the game is over, the slot waits for the boundary's k_assign.
// A function of its own, NOT inlined: inside k_pipe_tree its 244 registers pushed the tree loop into scratch (pass 50 -> 73 us).
// Its arguments are references into the kernel's kernarg segment (PipeKernArgs), not into copies.
template <class GM>
__device__ __attribute__((noinline)) uint32_t pipe_move_groups(const EngineParams& ep, const EngineArrays& ar, const PipeArrays& pa, const uint32_t my_slot) {
  const uint32_t wlane = threadIdx.x & 63u, grp = wlane >> 3, lane = wlane & 7u;
  PipeCtl* const pc = pa.ctl;
  PipeEpoch* const pe = pa.ep;
  const bool on = my_slot != kNoSlot;
  const uint32_t slot = on ? my_slot : 0u;
  uint32_t res_state = 0xFFu, seq = 0, was_sim = 0;
  if (on) {
    was_sim = ar.sstate[slot] == kSlotWaitEval ? 1u : 0u;        // (a listed slot's step finishes a simulation; a game start does not)
    res_state = round_slot<GM, false, true>(ep, ar, slot, lane);
  }
  if (on && res_state == kSlotQueued && lane == 0) {
    // the leaf goes to the net: the slot takes the form of a slot with a request out (k_pipe_seed does the same between epochs)
    seq = g_ld(ar.req_seq + slot) + 1u;
    if (seq == 0u) seq = 1u;
    ar.flags[slot] = ar.flags[slot] | kFlagReqOut;
    ar.sstate[slot] = kSlotWaitEval;
  }
  seq = __shfl(seq, static_cast<int>(grp * 8), 64);
  // everything the move steps wrote is in L2 before a request of theirs is out (its answer's token may be drawn by another
  // wavefront of this workgroup)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (on && res_state == kSlotQueued) {
    if (lane == 0) g_st(ar.req_seq + slot, seq);
    const uint32_t mg = pa.n_groups > 1u ? (ar.leaf_group[slot] & 1u) : 0u;       // the leaf's model group: its net's ring
    uint32_t pos = 0;
    if (lane == 0) pos = atomicAdd(pipe_tail_of(pc, mg), 1u);
    pos = __shfl(pos, static_cast<int>(grp * 8), 64);
    const size_t S_ = static_cast<size_t>(ep.S);
    const uint64_t payload = lane == 0 ? ar.leaf_pos[0 * S_ + slot] : lane == 1 ? ar.leaf_pos[1 * S_ + slot]
                           : lane == 2 ? (static_cast<uint64_t>(slot) | ((ar.leaf_pos[2 * S_ + slot] & 1ull) << 16)) : static_cast<uint64_t>(seq);
    if (lane < kReqGranules)
      g_st(pipe_ring_of(pa, mg) + static_cast<size_t>(pos & (kPipeRing - 1u)) * kReqGranules + lane, (pipe_lap_tag(pos) << 48) | (payload & kMask48));
  }
  {
    const unsigned long long em = __ballot(on && lane == 0 && res_state == kSlotEnded);
    const unsigned long long dm = __ballot(on && lane == 0 && (res_state == kSlotDone || res_state == 0xFFu));
    const unsigned long long om = __ballot(on && lane == 0);
    const unsigned long long sm = __ballot(on && lane == 0 && was_sim != 0u);
    if (wlane == 0) {
      if (em) atomicAdd(&pe->ended, static_cast<uint32_t>(__popcll(em)));
      if (dm) atomicAdd(&pe->dead, static_cast<uint32_t>(__popcll(dm)));
      if (sm) atomicAdd(&pe->sims, static_cast<unsigned long long>(__popcll(sm)));
      atomicAdd(&pe->moved, static_cast<uint32_t>(__popcll(om)));
    }
  }
  return res_state;
}

// NT: threads per workgroup (256: four wavefronts beside one net workgroup on a CU)
// PROF: the time accounting of AZMI_PIPE_PROF (a build of its own: the counters cost a dozen registers of a kernel that has none to spare)
// TWO: two model groups (play_past: a leaf goes to the S3-FIFO, the answer table lines and the request ring of the group its seat belongs
// to) - a build of its own for the same reason
// GUM: Gumbel seats (mcts.cc:233-342).  A search differs from plain PUCT in the choice at the root (the sequential-halving schedule,
// gumbel_next_root_child) and - gumbel_full - at interior nodes (gumbel_interior_select): both are SlotCtx's own functions, called
// from the descent below; the Gumbel state of a search (g, survivors, phase) is initialised by the move step's first descent
// (find_leaf's lazy init), so a search whose state is not there yet is handed to the move step.  Built with TWO.
#ifndef AZMI_TREE_MINB
#define AZMI_TREE_MINB 2          // workgroups of k_pipe_tree per CU the register budget is set for (experiment builds: 3, 4)
#endif
template <class GM, int NT, bool PROF, bool TWO = false, bool GUM = false>
__global__ __launch_bounds__(NT, AZMI_TREE_MINB) void k_pipe_tree(PipeKernArgs ka) {
  constexpr int G = GM::GROUP;
  constexpr int P = GM::P;
  static_assert(P == 2 && G == 8 && GM::M == 7, "written for Connect4's 8-lane groups");
  const EngineParams& ep = ka.ep;
  const EngineArrays& ar = ka.ar;
  const PipeArrays& pa = ka.pa;
  const uint32_t wlane = threadIdx.x & 63u, grp = wlane >> 3, lane0 = wlane & 7u;
  uint64_t t_start = wall_clock64(), t_last = t_start;      // (t_start moves by what a freeze takes: pipe_freeze_credit)
  PipeCtl* const pc = pa.ctl;
  PipeEpoch* const pe = pa.ep;
  PipeWg* const wc = pa.wg + blockIdx.x;
  const uint32_t rmask = (1u << pa.rshift) - 1u;
  unsigned long long* const myring = pa.rring + (static_cast<size_t>(blockIdx.x) << pa.rshift);
  // census first, then the epoch's stop word: a workgroup that only gets a place on the chip after the epoch has ended (the net
  // side, which leaves when every ARRIVED tree workgroup is done, may be gone by then) must not send requests any more
  if (threadIdx.x == 0) {
    unsigned long long t0 = atomicCAS(&pe->t0, 0ull, static_cast<unsigned long long>(t_start));
    if (t0 == 0ull) t0 = t_start;
    atomicMax(&pe->tree_late, static_cast<uint32_t>(t_start > t0 ? t_start - t0 : 0ull));
    const uint32_t before = atomicAdd(&pe->tree_arrived, 1u);
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(before) : "memory");
  }
  __syncthreads();
  if (pa.census_hold != 0u) {
    // calibration launch (pipe_calibrate): hold the place until census_hold ticks after the epoch's first workgroup started, do
    // nothing else.  A workgroup the chip had no place for starts when the others have left: it finds itself late and is counted.
    // (EVERY wavefront holds: a wavefront that ends gives its registers back, and a workgroup of one live wavefront would let a
    // third workgroup onto the CU - the measurement would admit more than an epoch can hold)
    const unsigned long long t0 = g_ld(&pe->t0);
    if (threadIdx.x == 0 && t_start > t0 && t_start - t0 >= pa.census_hold) atomicAdd(&pe->tree_late_n, 1u);
    while (wall_clock64() < t0 + pa.census_hold) __builtin_amdgcn_s_sleep(32);
    return;
  }
  bool go = g_ld(&ar.ctl->stop) == 0 && g_ld(&pc->err) == 0 && g_ld(&pe->stop) == 0;
  // The ring's HEAD lives in LDS for the epoch: this workgroup's four wavefronts are its only consumers, so a draw is one look at the
  // ring (eight token loads side by side) and a compare-and-swap in LDS - no head | tail word, no fetch-add in HBM, no window that
  // can reach past the tail (round 4, second half: three dependent round trips per pass became one).
  __shared__ uint32_t s_head;
  if (threadIdx.x == 0) s_head = wc->rhead;
  __syncthreads();
  uint64_t pf_pass = 0, pf_idle = 0, pf_n = 0, pf_act = 0, pf_polls = 0, pf_io = 0, pf_lvls = 0, pf_mv = 0;
  uint64_t pf_ph[5] = {0, 0, 0, 0, 0};     // per group (lane 0 counts): ticks in backup / descent / expansion / probe, simulations
  SlotCtx<GM> c(ep, ar, 0u, lane0);

  while (go) {
    const uint64_t pf_t0 = PROF ? wall_clock64() : 0;
    // (the lane number, opaque per pass: per-lane addresses - base + lane x stride of a dozen arrays - are then computed where they
    // are used instead of being hoisted out of the epoch's loop into 64-bit registers that live for ever, and spill)
    uint32_t lane = lane0;
    asm volatile("" : "+v"(lane));
    c.lane = lane;
    // ---- tokens for this pass: the arrived prefix of the eight ring positions at the head
    uint32_t my_slot = kNoSlot, tok_seq = 0, tok_move = 0, n_tok = 0, empty_polls = 0, ctl_word = 0;
    uint64_t t_seen = 0;
    bool fresh_pass = true;
    for (;;) {
      const uint32_t h = __builtin_amdgcn_readfirstlane(*const_cast<volatile uint32_t*>(&s_head));
      unsigned long long tok = 0;
      bool here = false;
      if (lane == 0) {
        const uint32_t pos = h + grp;
        tok = g_ld(myring + (pos & rmask));
        here = (tok >> 48) == pipe_lap_tag_r(pos, pa.rshift);
      }
      // the epoch's end: stop word / error, the quota, games that ended, the engine's own stop word.  (Looked at on the first poll
      // of a pass and on every fourth poll of a wavefront that finds nothing: idle wavefronts must not hammer hot lines)
      if ((empty_polls & 3u) == 0u) {
        ctl_word = 0;
        if (wlane == 7) ctl_word = g_ld(&pe->stop) | g_ld(&pc->err) | g_ld(&ar.ctl->stop);
        else if (wlane == 15) ctl_word = g_ld(&pe->sims) >= pa.quota ? 1u : 0u;
        else if (wlane == 23) ctl_word = g_ld(&pe->ended);
        else if (wlane == 31) ctl_word = g_ld(&pe->dead);
      }
      uint32_t stop_seen = __builtin_amdgcn_readlane(ctl_word, 7) | __builtin_amdgcn_readlane(ctl_word, 15);
      {
        const uint32_t w = __builtin_amdgcn_readlane(ctl_word, 23), d = __builtin_amdgcn_readlane(ctl_word, 31);
        const uint32_t with_game = ep.S > d ? ep.S - d : 0u;
        const uint32_t thr = max(1u, static_cast<uint32_t>((static_cast<unsigned long long>(with_game) * pa.idle_num) >> 10));
        if (w >= thr || w + d >= ep.S) stop_seen = 1u;
      }
      const uint64_t now = wall_clock64();
      if (fresh_pass) { t_last = now; fresh_pass = false; }      // (the pass before this look is work, not a freeze)
      else pipe_freeze_credit(pc, now, t_last, t_start, wlane == 0);
      // an epoch that runs long (a cold cache sends every leaf to the net: 16384 slots x 256 simulations take > 200 ms then) simply
      // ends at a quarter of the cap; the cap itself is the stall detector
      if (now - t_start > pa.soft_ticks) stop_seen = 1u;
      if (now - t_start > pa.cap_ticks) {
        if (wlane == 0) { atomicOr(&pc->err, static_cast<uint32_t>(kPipeErrTimeout)); pc->dbg[16] = __builtin_amdgcn_readlane(ctl_word, 23); pc->dbg[17] = __builtin_amdgcn_readlane(ctl_word, 31); }
        stop_seen = 1u;
      }
      if (stop_seen) {          // (tokens still in the ring are dropped: their slots are whole in HBM, the next epoch seeds them again)
        if (wlane == 0) g_st(&pe->stop, 1u);
        go = false;
        break;
      }
      if constexpr (PROF) pf_polls += 1;
      const unsigned long long hm = __ballot(here);
      // group g's token is bit 8 g: the arrived prefix in group order
      uint32_t k = 0;
      while (k < kTreeWindow && ((hm >> (8 * k)) & 1ull)) ++k;
      if (k == 0u) { ++empty_polls; __builtin_amdgcn_s_sleep(16); continue; }
      // fewer than a window: a short wait gathers what is on its way (AZMI_PIPE_TAKE_WAIT; 0 = take what is there)
      if (k < kTreeWindow && pa.take_wait != 0u) {
        if (t_seen == 0) t_seen = now;
        if (now - t_seen <= pa.take_wait) { __builtin_amdgcn_s_sleep(2); continue; }
      }
      // claim [h, h + k): another wavefront of this workgroup may have been faster - then look again from the new head
      uint32_t won = 0;
      if (wlane == 0) won = atomicCAS(&s_head, h, h + k) == h ? 1u : 0u;
      if (__builtin_amdgcn_readfirstlane(won) == 0u) continue;
      n_tok = k;
      {
        const uint32_t sl = static_cast<uint32_t>(__shfl(static_cast<uint32_t>(tok & 0xFFFFull), static_cast<int>(grp * 8), 64));
        const uint32_t sq = static_cast<uint32_t>(__shfl(static_cast<uint32_t>((tok >> 16) & 0xFFFFFFFFull), static_cast<int>(grp * 8), 64));
        if (grp < k) { my_slot = sl & static_cast<uint32_t>(kTokSlotMask); tok_move = (sl >> 15) & 1u; tok_seq = sq; }
      }
      break;
    }
    if (!go) break;
    const uint64_t pf_t1 = PROF ? wall_clock64() : 0;
    if constexpr (PROF) { pf_idle += pf_t1 - pf_t0; pf_n += 1; pf_act += n_tok; }

    // ---- the move step for the groups whose token carries the move bit (pipe_move_groups); a group whose next answer is at
    // hand afterwards goes on into the pass like any READY slot
    if (__ballot(tok_move != 0u) != 0ull) {
      const PipeKernArgs* const kargs = reinterpret_cast<const PipeKernArgs*>(reinterpret_cast<uintptr_t>(__builtin_amdgcn_kernarg_segment_ptr()));
      const uint32_t rs = pipe_move_groups<GM>(kargs->ep, kargs->ar, kargs->pa, tok_move != 0u ? my_slot : kNoSlot);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (tok_move != 0u) { if (rs == kSlotWaitEval) tok_seq = 0u; else my_slot = kNoSlot; }
      if constexpr (PROF) pf_mv += wall_clock64() - pf_t1;
    }

    // ---- the pass.  A token that answers a request (its sequence field is not 0) may be here before the wavefront that sent
    // the request - early, in the middle of its own pass - has put the slot back: req_seq[slot] is written last, behind that
    // pass's drained stores, so the slot is whole once it shows the token's sequence number.  (Same CU, same L1: no acquire.)
    // Everything the pass needs of its slots is asked for in ONE batch - the publication word, the answer's granules (speculatively:
    // they are where the token says), the slot's state, its path image, the leaf's key - and checked afterwards; only a slot that
    // turns out not to be back yet (rare) costs a second round (round 4, second half: four dependent round trips became two).
    const uint32_t slot = my_slot != kNoSlot ? my_slot : 0u;
    c.slot = slot;
    uint32_t st = kGrpIdle;
    uint8_t final_state = kSlotWaitEval;
    uint32_t cp = 0, root = 0, goal = 0, seq = 0, mg = 0, gum_active = 0, gum_full = 0;
    bool gum_defer = false;
    size_t tb = 0;
    float fpu_root = 0.0f;
    uint32_t* const path = ar.path + static_cast<size_t>(slot) * ep.max_depth;
    unsigned long long* const res = pa.res + static_cast<size_t>(slot) * kResStride;
    float reg_pi = 0.0f, reg_v = 0.0f;
    uint32_t root_n = 0; float root_v = 0.0f; uint64_t root_meta = 0;
    uint32_t lv_node = 0, lv_n = 0, lv_pp = 0;
    float lv_q = 0.0f, lv_d = 0.0f, lv_v = 0.0f;
    uint64_t lf_meta = 0;
    uint32_t lf_mv = 0, lf_c0 = 0, lf_k = 0, lf_term = 0, lf_player = 0;
    bool fw = false;
    uint32_t fw_node = 0, fw_n = 0, fw_plen = 0;
    float fw_q = 0.0f, fw_d = 0.0f, fw_v = 0.0f;
    uint32_t fl_node = 0xFFFFFFFFu, fl_mv = 0;
    uint64_t fl_meta = 0;
    float fl_pr = 0.0f;
    uint32_t sims_done = 0, sims_mem = 0, l0_hits = 0;
    bool rec_ok = true, answered = false;
    uint64_t cur_key = 0;
    unsigned long long g0 = 0, g1 = 0;
    PendRec pr_in{};
    uint32_t seq_now = 0;
    // (the publication word is looked at BEFORE the slot's plain loads are issued, not in one batch with them: the word
    // comes from L2 (sc1), the plain loads may hit this CU's L1 earlier in time - a holder that published in between would have had its
    // new sequence number accepted beside lines read before its stores.  One more round trip per pass, ~0.5 us of ~120.)
    if (my_slot != kNoSlot) {
      seq_now = g_ld(ar.req_seq + slot);
      if (tok_seq != 0u) {
        while (seq_now != tok_seq) {      // the slot's last holder is still storing it: wait for the publication (rare)
          if (wall_clock64() - t_start > pa.cap_ticks) {
            if (lane == 0 && atomicAdd(&pc->dbg[0], 1u) == 0u) { pc->dbg[1] = slot; pc->dbg[2] = tok_seq; pc->dbg[3] = seq_now; pc->dbg[4] = 0xEEEEu; pc->dbg[14] = blockIdx.x; pc->dbg[15] = static_cast<uint32_t>((wall_clock64() - t_start) / 100u); }
            if (lane == 0) atomicOr(&pc->err, static_cast<uint32_t>(kPipeErrTimeout));
            my_slot = kNoSlot;      // (its other holder is still at work: the slot is not touched; k_pipe_settle takes its answer over)
            break;
          }
          __builtin_amdgcn_s_sleep(2);
          seq_now = g_ld(ar.req_seq + slot);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      // (no instruction: the loads below stay below the poll)
    if (my_slot != kNoSlot) {
      if (tok_seq != 0u) {
        if (lane < static_cast<uint32_t>(GM::M)) g0 = g_ld(res + lane);
        if (lane <= static_cast<uint32_t>(P)) g1 = g_ld(res + kResV + lane);
      }
      pr_in = ar.pend[static_cast<size_t>(slot) * G + lane];
      cur_key = ar.leaf_key[slot];
      c.load();
    }
    const bool on = my_slot != kNoSlot;
    if (on) {
      cp = c.gs.player;
      if constexpr (TWO) mg = c.seat_group(cp) & 1u;        // (no move inside a pass: one player to move, one model group, also for the answer the pass starts with)
      tb = c.tree_base(cp);
      root = AZMI_SEL(c.t_root, cp);
      goal = (c.flags & kFlagCapped) ? c.seat_cap_visits(cp) : c.seat_visits(cp);
      fpu_root = c.seat_fpu_zero(cp) ? 0.0f : ep.fpu_reduction;
      seq = seq_now;
      rec_ok = (c.flags & kFlagPendRec) != 0;
      st = kGrpReady;
      if constexpr (GUM) {
        if (c.seat_gumbel(cp)) {
          const uint32_t* const gst = c.gum_state(cp);
          gum_active = gst[SlotCtx<GM>::kGumInit];
          gum_defer = gum_active == 0u && gst[SlotCtx<GM>::kGumTarget] > 0u;      // (not initialised yet: find_leaf's lazy init is the move step's)
          gum_full = c.seat_gumbel_full(cp) ? 1u : 0u;
        }
      }
      { const NodeRec* rr = ar.nodes + tb + root; root_n = rr->n; root_v = rr->v; root_meta = rr->meta; }
      lv_node = pr_in.node; lv_n = pr_in.n; lv_pp = pr_in.pp_mv & 0xFFu;
      lv_q = pr_in.q; lv_d = pr_in.d; lv_v = pr_in.v;
      lf_meta = pr_in.leaf_meta;
      lf_mv = pr_in.pp_mv >> 8;
      lf_c0 = meta_ch0(lf_meta); lf_k = meta_nch(lf_meta); lf_term = meta_term(lf_meta); lf_player = meta_player(lf_meta);
      if (c.flags & kFlagReqOut) {
        // the answer of the slot's request: its granules (asked for with the batch above; on their way at the latest - the net
        // workgroup sent the token after them -, so a granule that is not there yet is simply asked for again)
        for (uint32_t spins = 0;; ++spins) {
          if (spins != 0u || tok_seq == 0u) {
            if (lane < static_cast<uint32_t>(GM::M)) g0 = g_ld(res + lane);
            if (lane <= static_cast<uint32_t>(P)) g1 = g_ld(res + kResV + lane);
          }
          const bool ok = (lane >= static_cast<uint32_t>(GM::M) || static_cast<uint32_t>(g0 >> 32) == seq) &&
                          (lane > static_cast<uint32_t>(P) || static_cast<uint32_t>(g1 >> 32) == seq);
          // (all eight lanes of the group agree through a group-wide AND of ok)
          uint32_t okg = ok ? 1u : 0u;
          okg &= c.bcast(okg, 0) & c.bcast(okg, 1) & c.bcast(okg, 2) & c.bcast(okg, 3) & c.bcast(okg, 4) & c.bcast(okg, 5) & c.bcast(okg, 6);
          okg = c.bcast(okg, 0);
          if (okg) break;
          if (wall_clock64() - t_start > pa.cap_ticks) {
            if (lane == 0 && atomicAdd(&pc->dbg[0], 1u) == 0u) { pc->dbg[1] = slot; pc->dbg[2] = seq; pc->dbg[3] = c.flags; pc->dbg[4] = static_cast<uint32_t>(g1 >> 32); pc->dbg[5] = static_cast<uint32_t>(g0 >> 32); }
            if (lane == 0) atomicOr(&pc->err, static_cast<uint32_t>(kPipeErrTimeout));
            st = kGrpIdle;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
        if (st == kGrpReady) {
          reg_pi = lane < static_cast<uint32_t>(GM::M) ? __uint_as_float(static_cast<uint32_t>(g0)) : 0.0f;
          reg_v = lane <= static_cast<uint32_t>(P) ? __uint_as_float(static_cast<uint32_t>(g1)) : 0.0f;
          // the rows keep the pending answer, as they do for the lock-step kernels (the move step reads them)
          if (lane < static_cast<uint32_t>(GM::M)) ar.pi[static_cast<size_t>(slot) * GM::M + lane] = reg_pi;
          if (lane <= static_cast<uint32_t>(P)) ar.v[static_cast<size_t>(slot) * (P + 1) + lane] = reg_v;
          c.flags &= ~kFlagReqOut;
          answered = true;
        }
      } else if (c.flags & kFlagLeafNeedsNet) {
        if (lane < static_cast<uint32_t>(GM::M)) reg_pi = ar.pi[static_cast<size_t>(slot) * GM::M + lane];
        if (lane <= static_cast<uint32_t>(P)) reg_v = ar.v[static_cast<size_t>(slot) * (P + 1) + lane];
      }
    }
    if (ep.cache_on) {        // PlayManager::update_inferences -> insert_many (play_manager.cc:631-640): logged, applied after the epoch
      const unsigned long long am = __ballot(answered && lane == 0);
      if (am) {
        uint32_t base = 0;
        if (wlane == 0) base = atomicAdd(&pe->ins_count, static_cast<uint32_t>(__popcll(am)));
        base = __builtin_amdgcn_readfirstlane(base);
        if (answered) {
          const uint32_t idx = base + static_cast<uint32_t>(__popcll(am & ((1ull << (grp * 8)) - 1ull)));
          if (idx < pa.ins_cap) {
            if (lane == 0) pa.ins_key[idx] = cur_key;
            if constexpr (TWO) { if (lane == 1 && pa.ins_grp) pa.ins_grp[idx] = static_cast<uint8_t>(mg); }
            if (lane < static_cast<uint32_t>(GM::M)) pa.ins_pi[static_cast<size_t>(idx) * GM::M + lane] = reg_pi;
            if (lane <= static_cast<uint32_t>(P)) pa.ins_v[static_cast<size_t>(idx) * (P + 1) + lane] = reg_v;
          } else if (lane == 0) {
            g_st(&pe->stop, 1u);      // the log is full: this answer is not cached (nothing else is lost) and the epoch ends here
          }
        }
      }
    }
    const uint64_t pf_t2 = PROF ? wall_clock64() : 0;
    if constexpr (PROF) pf_io += pf_t2 - pf_t1;

    // ---- simulations (k_sim's body)
    bool listed = false;
    if (st == kGrpReady) {
      uint32_t inline_sims = 0;
      for (;;) {
        if (AZMI_SEL(c.t_depth, cp) + 1 >= goal || c.cur == root || (GUM && gum_defer)) {
          // the next backup completes the search (a move follows) or the evaluated leaf is the root (temperature, noise): the move step's
          c.flags |= kFlagListed;
          listed = true;
          st = kGrpIdle;
          break;
        }
        const uint64_t ph0 = PROF ? wall_clock64() : 0;
        if (!rec_ok) {
          // a path deeper than the 8 levels of the lane image: the backup walks MCTS::path_ in memory (the lock-step engine hands
          // these to the move step; here that would park the slot until the epoch ends)
          const bool from_net = (c.flags & kFlagLeafNeedsNet) != 0;
          c.template process_result<true>(cp, from_net, false, from_net, reg_pi, reg_v);     // (counts the simulation in c_sims itself)
          root_n += 1;
          sims_mem += 1;
          fw = false; fl_node = 0xFFFFFFFFu;
        } else {
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
        // ---- MCTS::find_leaf (mcts.cc:462-498), plain PUCT, with the forwarded values patched in
        const uint64_t ph1 = PROF ? wall_clock64() : 0;
        typename GM::State leaf = c.gs;
        uint32_t cur = root, plen = 0, n = root_n;
        uint64_t meta = root_meta;
        float v_cur = root_v;
        bool prefix = true, failed = false;
        while (n > 0 && meta_term(meta) == 0) {
          if (plen >= ep.max_depth) { failed = true; break; }
          if (lane == 0) path[plen] = cur;
          const uint32_t k = meta_nch(meta), c0 = meta_ch0(meta);
          if (k == 0) { failed = true; break; }
          uint32_t n_l = 0; float q_l = 0.0f, p_l = 0.0f, d_l = 0.0f, v_l = 0.0f; uint64_t m_l = 0;
          if (cur == fl_node) {                 // the children of the leaf evaluated a moment ago: all in registers
            if (lane < k) { p_l = fl_pr; m_l = meta_pack(0, 0, fl_mv, 0, 0); }
          } else {
            if (lane < k) {
              const NodeRec* cr = ar.nodes + tb + c0 + lane;
              n_l = cr->n; q_l = cr->q; p_l = cr->pr; d_l = cr->d; v_l = cr->v; m_l = cr->meta;
            }
            if (fw && prefix && plen < fw_plen && plen < 8u) {   // one child of this node was updated by the last backup
              const uint32_t t_node = c.bcast(fw_node, static_cast<int>(plen));
              const uint32_t t_n = c.bcast(fw_n, static_cast<int>(plen));
              const float t_q = c.bcast(fw_q, static_cast<int>(plen)), t_d = c.bcast(fw_d, static_cast<int>(plen)), t_v = c.bcast(fw_v, static_cast<int>(plen));
              if (c0 + lane == t_node) { n_l = t_n; q_l = t_q; d_l = t_d; v_l = t_v; if (t_node == fl_node) m_l = fl_meta; }
            }
          }
          const float fpu = (cur == root) ? fpu_root : ep.fpu_reduction;
          uint32_t best;
          if constexpr (GUM) {
            if (gum_active != 0u && cur == root) best = c.gumbel_next_root_child(cp, k, n_l, q_l, p_l);
            else if (gum_active != 0u && gum_full != 0u) best = c.gumbel_interior_select(cp, k, n_l, q_l, p_l, v_cur);
            else best = c.select_child(k, n_l, q_l, p_l, v_cur, n, fpu);
          } else {
            best = c.select_child(k, n_l, q_l, p_l, v_cur, n, fpu);
          }
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
        }
        if (failed) { c.raise(8u); final_state = kSlotDone; st = kGrpIdle; break; }
        const uint64_t ph2 = PROF ? wall_clock64() : 0;
        if constexpr (PROF) pf_lvls += plen;
        c.cur = cur; c.plen = plen;
        rec_ok = plen <= 8u;
#pragma unroll
        for (int p = 0; p < P; ++p) if (static_cast<uint32_t>(p) == cp) c.t_tld[p] += plen;
        uint32_t term = meta_term(meta);
        lf_k = meta_nch(meta); lf_c0 = meta_ch0(meta); lf_mv = 0; lf_meta = meta;
        if (n == 0) {
          term = GM::terminal(leaf);
          const uint64_t keep = meta_pack(0, 0, meta_mv(meta), leaf.player, term);
          if (!c.expand_node(cp, cur, leaf, keep, lf_c0, lf_k, &lf_mv)) { final_state = kSlotDone; st = kGrpIdle; break; }
          lf_meta = meta_pack(lf_c0, lf_k, meta_mv(meta), leaf.player, term);
        }
        lf_term = term; lf_player = leaf.player;
        const uint64_t ph3 = PROF ? wall_clock64() : 0;
        const bool needs_net = term == 0 && !c.seat_eval_random(cp);
        c.flags = needs_net ? (c.flags | kFlagLeafNeedsNet) : (c.flags & ~kFlagLeafNeedsNet);
        if (needs_net) {
          const uint64_t key = GM::key(leaf);
          // the answer table's granules are asked for inside the lookup, right behind the S3-FIFO shard's key loads (one round trip for
          // both); when they are valid - all ten carry this key's tags - the answer is there and the shard's payload round trip is
          // skipped; a shard hit the table did not have is copied into it (ten granule stores, nothing waits for them)
          unsigned long long l0a = 0, l0b = 0;
          const uint64_t lkey = TWO ? key ^ (mg ? kPipeGroupSalt : 0ull) : key;        // the answer table is one: group 1's lines under a salted key
          const unsigned long long* const le = pa.l0 + static_cast<size_t>(pipe_l0_entry(lkey, pa.l0_mask)) * kResStride;
          const bool hit = ep.cache_on && c.cache_lookup(key, mg, reg_pi, reg_v,
            [&]() {
              if (pa.l0) {
                if (lane < static_cast<uint32_t>(GM::M)) l0a = g_ld(le + lane);
                if (lane <= static_cast<uint32_t>(P)) l0b = g_ld(le + kResV + lane);
              }
            },
            [&](int cslot, float& o_pi, float& o_v) -> bool {
              if (!pa.l0) return false;
              uint32_t okg = ((lane >= static_cast<uint32_t>(GM::M) || static_cast<uint32_t>(l0a >> 32) == pipe_l0_tag(lkey, lane)) &&
                              (lane > static_cast<uint32_t>(P) || static_cast<uint32_t>(l0b >> 32) == pipe_l0_tag(lkey, kResV + lane))) ? 1u : 0u;
              okg &= c.bcast(okg, 0) & c.bcast(okg, 1) & c.bcast(okg, 2) & c.bcast(okg, 3) & c.bcast(okg, 4) & c.bcast(okg, 5) & c.bcast(okg, 6);
              okg = c.bcast(okg, 0);
              if (!okg) return false;
              o_pi = lane < static_cast<uint32_t>(GM::M) ? __uint_as_float(static_cast<uint32_t>(l0a)) : 0.0f;
              o_v = lane <= static_cast<uint32_t>(P) ? __uint_as_float(static_cast<uint32_t>(l0b)) : 0.0f;
              if (cslot < 0) l0_hits += 1;          // (the shard counted a miss: the host moves these to the hits)
              return true;
            },
            [&](float h_pi, float h_v) {
              if (pa.l0 && pa.l0_wb) {
                unsigned long long* const we = pa.l0 + static_cast<size_t>(pipe_l0_entry(lkey, pa.l0_mask)) * kResStride;
                if (lane < static_cast<uint32_t>(GM::M)) g_st(we + lane, (static_cast<unsigned long long>(pipe_l0_tag(lkey, lane)) << 32) | __float_as_uint(h_pi));
                if (lane <= static_cast<uint32_t>(P)) g_st(we + kResV + lane, (static_cast<unsigned long long>(pipe_l0_tag(lkey, kResV + lane)) << 32) | __float_as_uint(h_v));
              }
            });
          if (!hit) {
            // the request goes out NOW, not when the pass ends (its slowest group may run two more simulations): the net's
            // answer and this pass's tail overlap.  Nobody can take the slot before it is back: see the pass start.
            cur_key = key;
            seq = seq + 1u == 0u ? 1u : seq + 1u;
            uint32_t pos = 0;
            if (lane == 0) pos = atomicAdd(TWO ? pipe_tail_of(pc, mg) : &pc->tail, 1u);
            pos = c.bcast(pos, 0);
            const uint64_t payload = lane == 0 ? leaf.bb[0] : lane == 1 ? leaf.bb[1] : lane == 2 ? (static_cast<uint64_t>(slot) | (static_cast<uint64_t>(leaf.player) << 16))
                                                                                                  : static_cast<uint64_t>(seq);
            // (test hook, AZMI_PIPE_TEST_DROP = n: the request at ring position n goes out with a foreign lap tag - what an entry looks
            // like that was overwritten before its net workgroup could read it; tests/test_gpu_pipeline.py)
            const unsigned long long rtag = (pa.test_drop != 0u && pos == pa.test_drop) ? pipe_lap_tag(pos + kPipeRing) : pipe_lap_tag(pos);
            if (lane < kReqGranules)
              g_st((TWO ? pipe_ring_of(pa, mg) : pa.ring) + static_cast<size_t>(pos & (kPipeRing - 1u)) * kReqGranules + lane, (rtag << 48) | (payload & kMask48));
            // (the packed position stays with the slot: a request dropped by a pipeline error is sent again from there)
            if (lane < 3u) ar.leaf_pos[static_cast<size_t>(lane) * ep.S + slot] = lane == 2u ? static_cast<uint64_t>(leaf.player) : payload;
            if constexpr (TWO) { if (lane == 3u) ar.leaf_group[slot] = static_cast<uint8_t>(mg); }
            st = kGrpPush;
          }
        }
        if constexpr (PROF) {
          const uint64_t ph4 = wall_clock64();
          pf_ph[0] += ph1 - ph0; pf_ph[1] += ph2 - ph1; pf_ph[2] += ph3 - ph2; pf_ph[3] += ph4 - ph3; pf_ph[4] += 1;
        }
        if (st == kGrpPush) break;
        // the answer of this leaf is at hand (terminal, RANDOM evaluator, cache hit): the group goes on, a few times - the
        // pass ends with its slowest group
        if (++inline_sims >= pa.max_inline) break;
        // (the groups still here all have their next answer at hand; when only a few are left the other lanes of the wavefront idle:
        // the stragglers re-queue as READY instead and the pass ends)
        if (static_cast<uint32_t>(__popcll(__ballot(lane == 0))) < pa.min_active) break;
      }
    }
    const uint64_t pf_t3 = PROF ? wall_clock64() : 0;
    if constexpr (PROF) pf_pass += pf_t3 - pf_t2;


    // ---- the slots go back to HBM (k_sim's exit); the request's sequence number is drawn here, its flag is part of the state
    if (on) {
      if (st == kGrpPush) c.flags |= kFlagReqOut;
      if (lane == 0) {
        if (sims_done) ar.c_sims[slot] += sims_done;
        if (st == kGrpPush) { ar.c_evals[slot] += 1; ar.leaf_key[slot] = cur_key; }
      }
      if (rec_ok) {
        PathRegs r;
        r.node = lv_node; r.n = lv_n; r.q = lv_q; r.d = lv_d; r.v = lv_v; r.pp = lv_pp; r.mv = lf_mv; r.leaf_meta = lf_meta;
        c.store_pend(r);
        c.flags |= kFlagPendRec;
      } else {
        c.flags &= ~kFlagPendRec;
      }
      c.store(final_state);
    }
    // ---- tokens out, into this workgroup's own ring: READY tokens for the slots whose next answer is at hand (the requests went
    // out when their leaves were found), READY tokens with the move bit for the slots whose next step is the move step's.  The
    // ring TICKET is drawn before the stores have drained (it only reserves positions: its round trip and the drain overlap); the
    // tokens themselves go out behind the drain.
    {
      const bool tok_out = on && lane == 0 && (st == kGrpReady || listed);
      const unsigned long long rm = __ballot(tok_out);
      uint32_t base = 0;
      if (rm && wlane == 0) base = atomicAdd(&wc->rtail, static_cast<uint32_t>(__popcll(rm)));
      // everything this pass wrote is in L2 before any of its tokens is out (workgroup scope: the slots stay on this CU)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // the slot is whole: publish it (the word a token's taker waits for)
      if (on && lane == 0) g_st(ar.req_seq + slot, seq);
      if (rm) {
        base = __builtin_amdgcn_readfirstlane(base);
        if (tok_out) {
          const uint32_t pos = base + static_cast<uint32_t>(__popcll(rm & ((1ull << (grp * 8)) - 1ull)));
          g_st(myring + (pos & rmask), (pipe_lap_tag_r(pos, pa.rshift) << 48) | static_cast<unsigned long long>(slot) | (listed ? kTokMove : 0ull));
        }
      }
    }
    // ---- the epoch's counts: simulations (the quota ends it), slots lost to an engine error
    {
      uint32_t x = lane == 0 ? sims_done + sims_mem : 0u;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
      if (wlane == 0 && x) atomicAdd(&pe->sims, static_cast<unsigned long long>(x));
      if (pa.l0) {
        uint32_t y = lane == 0 ? l0_hits : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) y += __shfl_xor(y, off, 64);
        if (wlane == 0 && y) atomicAdd(&pc->l0_hits, static_cast<unsigned long long>(y));
      }
      const unsigned long long fm = __ballot(on && lane == 0 && final_state == kSlotDone);
      if (wlane == 0 && fm) atomicAdd(&pe->dead, static_cast<uint32_t>(__popcll(fm)));
    }
    if constexpr (PROF) pf_io += wall_clock64() - pf_t3;
  }

  if (wlane == 0) g_st(&pe->stop, 1u);       // a tree wavefront leaves: the epoch is over (also when it never began: engine stopped, error)
  if (PROF && wlane == 0) {
    atomicAdd(&pc->prof[0], pf_pass); atomicAdd(&pc->prof[1], pf_idle); atomicAdd(&pc->prof[2], pf_n); atomicAdd(&pc->prof[3], pf_act);
    atomicAdd(&pc->prof[4], pf_polls); atomicAdd(&pc->prof[5], pf_io);
    atomicAdd(&pc->prof[9], pf_ph[0]); atomicAdd(&pc->prof[10], pf_ph[1]); atomicAdd(&pc->prof[11], pf_ph[2]); atomicAdd(&pc->prof[12], pf_ph[3]);
    atomicAdd(&pc->prof[13], pf_ph[4]); atomicAdd(&pc->prof[14], pf_lvls); atomicAdd(&pc->prof[15], pf_mv); atomicAdd(&pc->prof[6], static_cast<unsigned long long>(wall_clock64() - t_start));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) { wc->rhead = s_head; atomicAdd(&pe->tree_done, 1u); }
}

// ---- the generic tree kernel: every search variant, two model groups -----------------------------------------------------------
// Engines the fast kernel above does not drive - Gumbel seats (k_sim is plain PUCT), two model groups (play_past: a leaf goes to the
// net of its seat's group) - run EVERY step through the lock-step engine's own round_slot<kMover> (pipe_move_groups): a pass = the
// tokens that have arrived, for an answer token first the answer (granules -> the slot's (v, pi) rows, the insert log), then one
// step per slot (backup, a move if the search is complete, descent to the next leaf, cache probe), then a request (sent by
// pipe_move_groups, into the ring of the leaf's group) or - when the answer is at hand - the slot's token back into the ring.  One
// simulation per slot and pass instead of the fast kernel's several, but no round barrier: a slot waits for its own answer only.
// Same rings, granules, home workgroups, epoch boundaries and exit conditions as k_pipe_tree.
// DIAG (AZMI_PIPE_DIAG=1, a build of its own): round 5's instrumentation of the stall hunt - phase stamps, segment clocks, the longest
// interval between two looks - written into PipeWg::pad (the line of rtail, the word net workgroups add to: hence not on by default,
// ADVICE r5) with s_waitcnt fences around the poll loop's loads so that each segment can be clocked.
template <class GM, int NT, bool DIAG = false>
__global__ __launch_bounds__(NT, 2) void k_pipe_tree_generic(PipeKernArgs ka) {
  constexpr int P = GM::P;
  const EngineParams& ep = ka.ep;
  const EngineArrays& ar = ka.ar;
  const PipeArrays& pa = ka.pa;
  const uint32_t wlane = threadIdx.x & 63u, grp = wlane >> 3, lane = wlane & 7u;
  uint64_t t_start = wall_clock64(), t_last = t_start;
  PipeCtl* const pc = pa.ctl;
  PipeEpoch* const pe = pa.ep;
  PipeWg* const wc = pa.wg + blockIdx.x;
  // where every wavefront of this workgroup is (diagnostics of the rare stall-cap error): PipeWg::pad[2 + wave] =
  // phase | units of 10 us since the wave started << 8; the host prints the words of every tree workgroup with a pipeline error
  uint32_t* const phase_word = &wc->pad[2 + (threadIdx.x >> 6)];
  // (an epoch that starts with the error word already set - the epochs of a call that were enqueued behind the failing one - leaves
  // the diagnostics of the failing epoch alone)
  const bool diag_on = DIAG && g_ld(&pc->err) == 0u;
#define AZMI_GEN_PHASE(x) do { if (DIAG && wlane == 0 && diag_on) g_st(phase_word, static_cast<uint32_t>(x) | (static_cast<uint32_t>((wall_clock64() - t_start) / 1000u) << 8)); } while (0)
  AZMI_GEN_PHASE(1);
  // (when each wavefront of the workgroup ENTERED the kernel: the raw clock, low 32 bits - do the four start together?)
  if (DIAG && wlane == 0 && diag_on) g_st(&wc->pad[6 + (threadIdx.x >> 6)], static_cast<uint32_t>(t_start));
  const uint32_t rmask = (1u << pa.rshift) - 1u;
  unsigned long long* const myring = pa.rring + (static_cast<size_t>(blockIdx.x) << pa.rshift);
  if (threadIdx.x == 0) {
    unsigned long long t0 = atomicCAS(&pe->t0, 0ull, static_cast<unsigned long long>(t_start));
    if (t0 == 0ull) t0 = t_start;
    atomicMax(&pe->tree_late, static_cast<uint32_t>(t_start > t0 ? t_start - t0 : 0ull));
    const uint32_t before = atomicAdd(&pe->tree_arrived, 1u);
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(before) : "memory");
  }
  __syncthreads();
  AZMI_GEN_PHASE(2);
  if (pa.census_hold != 0u) {        // calibration launch: see k_pipe_tree
    const unsigned long long t0 = g_ld(&pe->t0);
    if (threadIdx.x == 0 && t_start > t0 && t_start - t0 >= pa.census_hold) atomicAdd(&pe->tree_late_n, 1u);
    while (wall_clock64() < t0 + pa.census_hold) __builtin_amdgcn_s_sleep(32);
    return;
  }
  bool go = g_ld(&ar.ctl->stop) == 0 && g_ld(&pc->err) == 0 && g_ld(&pe->stop) == 0;
  __shared__ uint32_t s_head;
  if (threadIdx.x == 0) s_head = wc->rhead;
  __syncthreads();
  const PipeKernArgs* const kargs = reinterpret_cast<const PipeKernArgs*>(reinterpret_cast<uintptr_t>(__builtin_amdgcn_kernarg_segment_ptr()));
  uint64_t seg_max[3] = {0, 0, 0}, sg_prev = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
  uint64_t ph_max[5] = {0, 0, 0, 0, 0};
  uint64_t gap_rt = 0, gap_sq = 0, sq_last = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
  uint32_t gap_where = 0, did_pass = 0, gap_polls = 0, total_polls = 0;
  while (go) {
    AZMI_GEN_PHASE(3);
    // ---- tokens (as k_pipe_tree: the arrived prefix of the eight ring positions at the LDS head)
    uint32_t my_slot = kNoSlot, tok_seq = 0, empty_polls = 0, ctl_word = 0;
    bool fresh_pass = true;
    for (;;) {
      const uint64_t sg0 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
      const uint32_t h = __builtin_amdgcn_readfirstlane(*const_cast<volatile uint32_t*>(&s_head));
      unsigned long long tok = 0;
      bool here = false;
      if (lane == 0) {
        const uint32_t pos = h + grp;
        tok = g_ld(myring + (pos & rmask));
        here = (tok >> 48) == pipe_lap_tag_r(pos, pa.rshift);
      }
      uint64_t sg1 = 0;
      if constexpr (DIAG) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" :: "v"(tok) : "memory");
        sg1 = __builtin_amdgcn_s_memtime();
        if (sg1 - sg0 > seg_max[0]) seg_max[0] = sg1 - sg0;          // (diagnostics: LDS head + the token load)
      }
      if ((empty_polls & 3u) == 0u) {
        ctl_word = 0;
        if (wlane == 7) ctl_word = g_ld(&pe->stop) | g_ld(&pc->err) | g_ld(&ar.ctl->stop);
        else if (wlane == 15) ctl_word = g_ld(&pe->sims) >= pa.quota ? 1u : 0u;
        else if (wlane == 23) ctl_word = g_ld(&pe->ended);
        else if (wlane == 31) ctl_word = g_ld(&pe->dead);
      }
      if constexpr (DIAG) {
        asm volatile("s_waitcnt vmcnt(0)" :: "v"(ctl_word) : "memory");
        const uint64_t sg2 = __builtin_amdgcn_s_memtime();
        if (sg2 - sg1 > seg_max[1]) seg_max[1] = sg2 - sg1;          // (the control words)
        if (sg0 - sg_prev > seg_max[2]) seg_max[2] = sg0 - sg_prev;  // (from the last look's control words to this look's start: the sleep, a pass)
        sg_prev = sg2;
      }
      uint32_t stop_seen = __builtin_amdgcn_readlane(ctl_word, 7) | __builtin_amdgcn_readlane(ctl_word, 15);
      {
        const uint32_t w = __builtin_amdgcn_readlane(ctl_word, 23), d = __builtin_amdgcn_readlane(ctl_word, 31);
        const uint32_t with_game = ep.S > d ? ep.S - d : 0u;
        const uint32_t thr = max(1u, static_cast<uint32_t>((static_cast<unsigned long long>(with_game) * pa.idle_num) >> 10));
        if (w >= thr || w + d >= ep.S) stop_seen = 1u;
      }
      const uint64_t now = wall_clock64();
      if constexpr (DIAG) {   // diagnostics: the longest interval between two looks of this wavefront by the 100 MHz wall clock, the same interval by the SQ's
          // own counter (s_memtime), and where the wavefront was in between (0: only this poll loop, 1: a pass)
        const uint64_t sq_now = __builtin_amdgcn_s_memtime();
        if (now - t_last > gap_rt) { gap_rt = now - t_last; gap_sq = sq_now - sq_last; gap_where = did_pass; gap_polls = total_polls; }
        sq_last = sq_now; did_pass = 0u; ++total_polls;
      }
      if (fresh_pass) { t_last = now; fresh_pass = false; }
      else pipe_freeze_credit(pc, now, t_last, t_start, wlane == 0);
      if (now - t_start > pa.soft_ticks) stop_seen = 1u;
      if (now - t_start > pa.cap_ticks) {
        if (wlane == 0) {
          atomicOr(&pc->err, static_cast<uint32_t>(kPipeErrTimeout));
          if (atomicAdd(&pc->dbg[0], 1u) == 0u) { pc->dbg[1] = blockIdx.x; pc->dbg[2] = empty_polls; pc->dbg[3] = h; pc->dbg[4] = 0xCCCCu; pc->dbg[5] = wc->rtail; pc->dbg[6] = static_cast<uint32_t>(pa.soft_ticks / 1000u); }
        }
        stop_seen = 1u;
      }
      if (stop_seen) { if (wlane == 0) g_st(&pe->stop, 1u); go = false; break; }
      const unsigned long long hm = __ballot(here);
      uint32_t k = 0;
      while (k < kTreeWindow && ((hm >> (8 * k)) & 1ull)) ++k;
      if (k == 0u) { ++empty_polls; __builtin_amdgcn_s_sleep(16); continue; }
      uint32_t won = 0;
      if (wlane == 0) won = atomicCAS(&s_head, h, h + k) == h ? 1u : 0u;
      if (__builtin_amdgcn_readfirstlane(won) == 0u) continue;
      const uint32_t sl = static_cast<uint32_t>(__shfl(static_cast<uint32_t>(tok & 0xFFFFull), static_cast<int>(grp * 8), 64));
      const uint32_t sq = static_cast<uint32_t>(__shfl(static_cast<uint32_t>((tok >> 16) & 0xFFFFFFFFull), static_cast<int>(grp * 8), 64));
      if (grp < k) { my_slot = sl & static_cast<uint32_t>(kTokSlotMask); tok_seq = sq; }
      break;
    }
    if (!go) break;
    did_pass = 1u;
    uint64_t ph_t = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
#define AZMI_GEN_SEG(i) do { if constexpr (DIAG) { const uint64_t t_ = __builtin_amdgcn_s_memtime(); if (t_ - ph_t > ph_max[i]) ph_max[i] = t_ - ph_t; ph_t = t_; } } while (0)
    AZMI_GEN_PHASE(4);
    // ---- an answer token: the slot is back once req_seq shows the token's number; its granules become the slot's (v, pi) rows
    if (my_slot != kNoSlot && tok_seq != 0u) {
      while (g_ld(ar.req_seq + my_slot) != tok_seq) {
        if (wall_clock64() - t_start > pa.cap_ticks) {
          if (lane == 0 && atomicAdd(&pc->dbg[0], 1u) == 0u) { pc->dbg[1] = my_slot; pc->dbg[2] = tok_seq; pc->dbg[3] = g_ld(ar.req_seq + my_slot); pc->dbg[4] = 0xAAAAu; pc->dbg[5] = ar.sstate[my_slot]; pc->dbg[6] = ar.flags[my_slot]; }
          if (lane == 0) atomicOr(&pc->err, static_cast<uint32_t>(kPipeErrTimeout));
          my_slot = kNoSlot; break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    bool answered = false;
    float reg_pi = 0.0f, reg_v = 0.0f;
    AZMI_GEN_SEG(0);      // the wait for the slot's publication word
    AZMI_GEN_PHASE(5);
    if (my_slot != kNoSlot && tok_seq != 0u) {
      const unsigned long long* const res = pa.res + static_cast<size_t>(my_slot) * kResStride;
      unsigned long long g0 = 0, g1 = 0;
      for (;;) {
        if (lane < static_cast<uint32_t>(GM::M)) g0 = g_ld(res + lane);
        if (lane <= static_cast<uint32_t>(P)) g1 = g_ld(res + kResV + lane);
        uint32_t okg = ((lane >= static_cast<uint32_t>(GM::M) || static_cast<uint32_t>(g0 >> 32) == tok_seq) &&
                        (lane > static_cast<uint32_t>(P) || static_cast<uint32_t>(g1 >> 32) == tok_seq)) ? 1u : 0u;
#pragma unroll
        for (int i = 1; i < 8; i <<= 1) okg &= static_cast<uint32_t>(__shfl_xor(static_cast<int>(okg), i, 8));
        if (okg) { answered = true; break; }
        if (wall_clock64() - t_start > pa.cap_ticks) {
          if (lane == 0 && atomicAdd(&pc->dbg[0], 1u) == 0u) { pc->dbg[1] = my_slot; pc->dbg[2] = tok_seq; pc->dbg[3] = static_cast<uint32_t>(g0 >> 32); pc->dbg[4] = 0xBBBBu; pc->dbg[5] = static_cast<uint32_t>(g1 >> 32); pc->dbg[6] = g_ld(ar.req_seq + my_slot); }
          if (lane == 0) atomicOr(&pc->err, static_cast<uint32_t>(kPipeErrTimeout));
          my_slot = kNoSlot; break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      if (answered) {
        reg_pi = lane < static_cast<uint32_t>(GM::M) ? __uint_as_float(static_cast<uint32_t>(g0)) : 0.0f;
        reg_v = lane <= static_cast<uint32_t>(P) ? __uint_as_float(static_cast<uint32_t>(g1)) : 0.0f;
        if (lane < static_cast<uint32_t>(GM::M)) ar.pi[static_cast<size_t>(my_slot) * GM::M + lane] = reg_pi;
        if (lane <= static_cast<uint32_t>(P)) ar.v[static_cast<size_t>(my_slot) * (P + 1) + lane] = reg_v;
        if (lane == 0) ar.flags[my_slot] = ar.flags[my_slot] & static_cast<uint8_t>(~kFlagReqOut);
      }
    }
    AZMI_GEN_SEG(1);      // the answer's granules
    if (ep.cache_on) {        // PlayManager::update_inferences -> insert_many: logged with the leaf's group, applied after the epoch
      const unsigned long long am = __ballot(answered && lane == 0);
      if (am) {
        uint32_t base = 0;
        if (wlane == 0) base = atomicAdd(&pe->ins_count, static_cast<uint32_t>(__popcll(am)));
        base = __builtin_amdgcn_readfirstlane(base);
        if (answered) {
          const uint32_t idx = base + static_cast<uint32_t>(__popcll(am & ((1ull << (grp * 8)) - 1ull)));
          if (idx < pa.ins_cap) {
            if (lane == 0) { pa.ins_key[idx] = ar.leaf_key[my_slot]; if (pa.ins_grp) pa.ins_grp[idx] = ar.leaf_group[my_slot]; }
            if (lane < static_cast<uint32_t>(GM::M)) pa.ins_pi[static_cast<size_t>(idx) * GM::M + lane] = reg_pi;
            if (lane <= static_cast<uint32_t>(P)) pa.ins_v[static_cast<size_t>(idx) * (P + 1) + lane] = reg_v;
          } else if (lane == 0) {
            g_st(&pe->stop, 1u);
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the rows are in place before the step reads them
    AZMI_GEN_SEG(2);      // the insert-log ticket and the drain
    AZMI_GEN_PHASE(6);
    // ---- one step of every slot of the pass (requests leave inside)
    const uint32_t rs = pipe_move_groups<GM>(kargs->ep, kargs->ar, kargs->pa, my_slot);
    AZMI_GEN_SEG(3);      // the steps themselves (round_slot, the requests)
    AZMI_GEN_PHASE(7);
    // ---- a slot whose next answer is at hand (cache hit, terminal leaf, RANDOM seat) goes straight back into the ring
    {
      const bool tok_out = my_slot != kNoSlot && lane == 0 && rs == kSlotWaitEval;
      const unsigned long long rm = __ballot(tok_out);
      uint32_t base = 0;
      if (rm && wlane == 0) base = atomicAdd(&wc->rtail, static_cast<uint32_t>(__popcll(rm)));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (rm) {
        base = __builtin_amdgcn_readfirstlane(base);
        if (tok_out) {
          const uint32_t pos = base + static_cast<uint32_t>(__popcll(rm & ((1ull << (grp * 8)) - 1ull)));
          g_st(myring + (pos & rmask), (pipe_lap_tag_r(pos, pa.rshift) << 48) | static_cast<unsigned long long>(my_slot));
        }
      }
    }
    AZMI_GEN_SEG(4);      // the READY tokens
#undef AZMI_GEN_SEG
  }
  AZMI_GEN_PHASE(8);
  if (wlane == 0) g_st(&pe->stop, 1u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  AZMI_GEN_PHASE(9);
  if (DIAG && threadIdx.x == 0 && diag_on) {      // (wavefront 0's longest interval between two looks: 10 us units by the wall clock | by the SQ clock, raw ticks >> 10 | where | after how many polls)
    for (int i = 0; i < 5; ++i) wc->pad[18 + i] = static_cast<uint32_t>(ph_max[i] >> 10);
    wc->pad[15] = static_cast<uint32_t>(seg_max[0] >> 10); wc->pad[16] = static_cast<uint32_t>(seg_max[1] >> 10); wc->pad[17] = static_cast<uint32_t>(seg_max[2] >> 10);
    wc->pad[10] = static_cast<uint32_t>(gap_rt / 1000u); wc->pad[11] = static_cast<uint32_t>(gap_sq >> 10); wc->pad[12] = gap_where; wc->pad[13] = gap_polls; wc->pad[14] = total_polls;
  }
  if (threadIdx.x == 0) { wc->rhead = s_head; atomicAdd(&pe->tree_done, 1u); }
#undef AZMI_GEN_PHASE
}

// ---- net side -----------------------------------------------------------------------------------------------------------------------
// A persistent workgroup of the leaf net (leafnet_c4.h: 4 waves, one tile of boards through the whole tower): claim up to six
// requests, read their granules, build the input planes from the packed positions (connect4_gs.cc:131-149: planes 0 / 1 the
// stones, plane 2 + player all ones), run the 3- or the 6-board tile, write the (v, pi) granules of every board, again.
// Leaves when the epoch is over: stop is up, every tree workgroup has left and the ring is empty.
constexpr uint32_t kPipeXs = 512;      // bytes of claim scratch behind the tile's LDS
// MODE 0: the 3- and the 6-board tile (by what the claim brought), 1: the 6-board tile only, 2: the 3-board tile only
// X3: the bf16x3 tier's tiles (Tile<.., SPLIT>: weights and activations as bf16 high + low parts, three MFMAs per product - the north
// star's 1e-5 on the matrix cores); 16 activation planes, so ONE workgroup per CU, beside which a tree workgroup still fits
// Two model groups (play_past: one net per group): np1 = the weights of group 1's net; a workgroup serves ONE group's ring for the
// whole epoch (its index's parity when both groups have a net behind them, pa.net_groups).
template <int MODE, bool X3 = false>
__global__ __launch_bounds__(256, 2) void k_pipe_net(azmi_net_dev::NetDesc nd, azmi_net_dev::NetPtrs np, azmi_net_dev::NetPtrs np1, PipeArrays pa) {
  constexpr uint32_t kMaxTake = MODE == 2 ? 3u : 6u;
  constexpr uint64_t kPatienceTicks = 150;       // 1.5 us: how long a request that is there waits for the rest of its window
  using namespace azmi_net_dev;
  using TBig = std::conditional_t<X3, c4::TileBigX3, c4::TileBig>;
  using TSmall = std::conditional_t<X3, c4::TileSmallX3, c4::TileSmall>;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_pipe[];
  // claim scratch: [0] boards claimed (0 = leave), [1] first ring position; [8..14) slot, [16..22) sequence number,
  // [24..30) player, then as u64: [0..6) stones of player 0, [8..14) stones of player 1
  uint32_t* const xs = reinterpret_cast<uint32_t*>(lds_pipe + TBig::LDS_BYTES);
  static_assert(TBig::LDS_BYTES >= TSmall::LDS_BYTES && (X3 ? 1 : 2) * (TBig::LDS_BYTES + kPipeXs) <= 160 * 1024, "two workgroups per CU (X3: one)");
  const uint32_t tid0 = threadIdx.x;
  uint64_t t_start = wall_clock64(), t_last = t_start;      // (wave 0's copy moves by what a freeze takes)
  PipeCtl* const pc = pa.ctl;
  PipeEpoch* const pe = pa.ep;
  const uint32_t mg = pa.net_groups == 3u ? (blockIdx.x & 1u) : (pa.net_groups == 2u ? 1u : 0u);      // this workgroup's model group
  uint32_t* const q_head = pipe_head_of(pc, mg);
  uint32_t* const q_tail = pipe_tail_of(pc, mg);
  const unsigned long long* const q_ring = pipe_ring_of(pa, mg);
  if (tid0 == 0) {
    unsigned long long t0 = atomicCAS(&pe->t0, 0ull, static_cast<unsigned long long>(t_start));
    if (t0 == 0ull) t0 = t_start;
    atomicMax(&pe->net_late, static_cast<uint32_t>(t_start > t0 ? t_start - t0 : 0ull));
    atomicAdd(&pe->net_arrived, 1u);
    if (pa.census_hold != 0u && t_start > t0 && t_start - t0 >= pa.census_hold) atomicAdd(&pe->net_late_n, 1u);      // calibration launch: see k_pipe_tree
  }
  if (pa.census_hold != 0u) {
    __syncthreads();
    const unsigned long long t0 = g_ld(&pe->t0);
    while (wall_clock64() < t0 + pa.census_hold) __builtin_amdgcn_s_sleep(32);
    return;
  }
  // The workgroup's WINDOW: kMaxTake consecutive ring positions drawn with ONE fetch-add on `head` (a compare-and-swap claim
  // of "what is there" serialises every workgroup of the chip on one word: a claim then costs a memory round trip per
  // contender).  The window's positions are this workgroup's to serve, whenever their requests arrive: it waits until the
  // rest of the window is there - or, when a request is waiting, for a short patience only -, runs a tile over what
  // arrived, and draws a new window once this one is used up.  head therefore runs ahead of tail.
  uint32_t w0 = 0, wn = 0, wdone = 0;        // window start, size, positions served (wave 0 keeps them; uniform)
  bool prev_full = false, first_look = false; // the last window drawn was complete when first looked at (a backlog); the next look is a window's first
  uint64_t pf_wait = 0, pf_tile = 0, pf_mark = wall_clock64();
  for (;;) {
    uint32_t tid = tid0;      // (opaque per tile, as k_pipe_tree's lane number: per-thread addresses are not hoisted out of the epoch's loop)
    asm volatile("" : "+v"(tid));
    __syncthreads();
    { const uint64_t nowp = wall_clock64(); pf_tile += nowp - pf_mark; pf_mark = nowp; }
    if (tid < 64) {        // wave 0 runs the claim; lanes 0..7 look at one ring entry each
      uint32_t n = 0;
      uint32_t sl = 0xFFFFFFFFu, sq = 0, pl = 0;
      unsigned long long b0 = 0, b1 = 0;
      uint64_t t_first = 0;                  // when the first request of this pass was seen
      uint32_t final_looks = 0;              // empty looks at the window after the tree side had left
      uint64_t t_empty = 0;                  // when this claim first found nothing at its position
      uint32_t empty_looks = 0;              // looks that found nothing (lane 0 counts)
      unsigned long long sims_seen = 0; uint64_t t_prog = 0;     // a surplus workgroup's view of the tree side's progress (its slow poll)
      for (;;) {
        if (wdone == wn) {
          // the window's size follows the load (MODE 0): with a backlog of requests the 6-board tile (capacity: 28 M evaluations/s on
          // the chip's net places), with none - this workgroup is idle, and so are others - the 3-board tile (latency: 39 us alone
          // against 60; the slots, not the matrix cores, are what is short then)
          // (round 4 asked the ring: 6 when at least big_at = 96 requests waited, from a load of head and one of tail per claim.  Those
          // are the two hottest words of the pipeline - every request is a returning add on tail - and the two loads alone cost 2.7 % of
          // the headline (always-3 with them against always-3 without).  Round 5: a workgroup whose LAST window was complete when it
          // first looked at it has met a backlog itself - 6; one that had to wait for its requests - 3.  No load at all: +3.4 % on one
          // box, profiles/r5_window_rule_ab.txt.  AZMI_PIPE_BIG_AT >= 2 brings the ring's count back as a second condition.)
          uint32_t h = 0, take = kMaxTake;
          if (MODE == 0 && pa.big_at != 0u) {
            if (!prev_full) take = 3u;
            else if (pa.big_at >= 2u) {
              uint32_t hd = 0, tl = 0;
              if (tid == 0) { hd = g_ld(q_head); tl = g_ld(q_tail); }
              const int32_t backlog = static_cast<int32_t>(__builtin_amdgcn_readfirstlane(tl) - __builtin_amdgcn_readfirstlane(hd));
              if (backlog < static_cast<int32_t>(pa.big_at)) take = 3u;
            }
          }
          first_look = true;
          if (tid == 0) h = atomicAdd(q_head, take);
          w0 = __builtin_amdgcn_readfirstlane(h); wn = take; wdone = 0;
        }
        const uint32_t left = wn - wdone;
        bool here = false;
        if (tid < left) {
          const uint32_t pos = w0 + wdone + tid;
          const unsigned long long want = pipe_lap_tag(pos);
          const unsigned long long* e = q_ring + static_cast<size_t>(pos & (kPipeRing - 1u)) * kReqGranules;
          const unsigned long long a0 = g_ld(e), a1 = g_ld(e + 1), a2 = g_ld(e + 2), a3 = g_ld(e + 3);
          here = (a0 >> 48) == want && (a1 >> 48) == want && (a2 >> 48) == want && (a3 >> 48) == want;
          if (here) {
            b0 = a0 & kMask48; b1 = a1 & kMask48;
            sl = static_cast<uint32_t>(a2 & 0xFFFFull); pl = static_cast<uint32_t>((a2 >> 16) & 1ull);
            sq = static_cast<uint32_t>(a3);
          }
        }
        const uint32_t hm = static_cast<uint32_t>(__ballot(here)) & 0xFFu;
        const uint32_t k = static_cast<uint32_t>(__builtin_ctz(~hm));          // arrived prefix of the window's rest
        if (first_look) { prev_full = k == left; first_look = false; }
        const uint64_t now = wall_clock64();
        pipe_freeze_credit(pc, now, t_last, t_start, tid == 0);
        if (k == left) { n = k; break; }
        if (k != 0u) {
          if (t_first == 0) t_first = now;
          if (now - t_first > kPatienceTicks) { n = k; break; }
          continue;
        }
        // nothing there: is the epoch over?  stop is up (no tree workgroup that arrives from now on sends anything) and every
        // tree workgroup that did arrive has left: tail is final, and what lies at or beyond it never comes
        uint32_t over = 0, stale = 0, idle = 0, leave = 0;
        if (tid == 0) {
          // A position more than half a ring BEHIND the tail will never show this lap's tag again: its request was overwritten a lap
          // later before this workgroup could look (seen once in ~1e10 requests: the workgroup sat 2 laps behind - it had been
          // switched out).  Give the window up; the slot stays unanswered and k_pipe_settle sends its request again.
          // (looked at only once this position has been empty for 0.5 ms: `tail` is the hottest word of the pipeline - every request
          // is a returning add on it - and 384 idle workgroups reading it on every poll slowed every request: the headline lost 12 %)
          if (t_empty == 0) t_empty = now;
          if (now - t_empty > 2000ull) idle = 1;          // nothing for 20 us: poll slowly from here on (below)
          if (now - t_empty > 50000ull && static_cast<int32_t>(g_ld(q_tail) - (w0 + wdone)) > static_cast<int32_t>(kPipeRing / 2u)) stale = 1;
          // (err and stop are looked at on every eighth empty look only - ~4 us: they matter when the epoch ends, the window's own
          // ring entries, read above on every look, are what a request's latency depends on)
          if (idle == 0u && (empty_looks++ & 7u) != 0u) { /* not this time */ }      // (a workgroup in its slow poll looks every time: 14 us apart)
          else if (g_ld(&pc->err)) over = 1;
          else if (g_ld(&pe->stop) != 0u && g_ld(&pe->tree_done) >= g_ld(&pe->tree_arrived)) {
            const uint32_t t2 = g_ld(q_tail);
            if (static_cast<int32_t>(t2 - (w0 + wdone)) <= 0) over = 1;
            // every writer has left (and drained its stores before it was counted done): a position below the final tail that is
            // still not there on the second look after that will never be - given up like a stale one
            else if (++final_looks >= 2u) stale = 1;
          }
          // the stall detector: cap x 1.25 once the tree side is there; while no tree workgroup has started yet (the host was held up
          // between the two launches, the tree kernel waits for a place) four caps - long enough for any scheduling hiccup, short
          // enough that a launch that never comes is an error and not a hang
          // (round 5: FORTY caps = 10 s.  The cap is the tree side's stall detector; the net side only waits for the tree side, and the
          // tree kernel's wavefronts were seen to stand still for 0.3 s (profiles/r5_repro_generic_freeze_credit.txt) and for more than
          // 2 s (r5_repro_generic_8xcap.txt) while this kernel kept running: they credit such a freeze and finish normally, this side
          // must outwait it; a tree kernel that never comes is still an error after 10 s, not a hang)
          // (round 6, ADVICE r5: what round 5's captures showed is that tree wavefronts can stand still for as long as HUNDREDS of idle
          // net workgroups stay resident beside them, and run on the moment those leave.  An epoch launches no more net workgroups than
          // its slots can keep busy (net_launch), so this only arises under AZMI_PIPE_NET_ALL; there a workgroup the slots do not need
          // (index >= net_needed) that has been idle, in its slow poll, beside a tree side that HAS arrived and whose simulation count
          // has not moved for 1.25 caps gives its place back instead of outwaiting the freeze: its window is given up like a stale one
          // (k_pipe_settle sends a request that still lands there again), no error.  The needed workgroups keep the long patience.)
          if (idle != 0u && !over && blockIdx.x >= pa.net_needed) {
            const unsigned long long s_now = g_ld(&pe->sims);
            if (s_now != sims_seen || t_prog == 0) { sims_seen = s_now; t_prog = now; }
            else if (g_ld(&pe->tree_arrived) != 0u && now - t_prog > pa.cap_ticks + (pa.cap_ticks >> 2)) { stale = 1; leave = 1; }
          }
          if (!over && now - t_start > 40u * pa.cap_ticks) {
            // (what the stalled workgroup saw, for the error message: dbg[8..14) = stop, tree workgroups done / arrived, ring tail,
            // the window position it waits at, 10 us units since its start)
            if (atomicAdd(&pc->dbg[7], 1u) == 0u) {
              pc->dbg[8] = g_ld(&pe->stop); pc->dbg[9] = g_ld(&pe->tree_done); pc->dbg[10] = g_ld(&pe->tree_arrived);
              pc->dbg[11] = g_ld(q_tail); pc->dbg[12] = w0 + wdone; pc->dbg[13] = static_cast<uint32_t>((now - t_start) / 1000u);
            }
            atomicOr(&pc->err, static_cast<uint32_t>(kPipeErrNetTimeout)); g_st(&pe->stop, 1u); over = 1;
          }
        }
        if (__builtin_amdgcn_readfirstlane(over)) { n = 0; break; }
        if (__builtin_amdgcn_readfirstlane(stale)) {
          if (tid == 0) { atomicAdd(&pe->lost, left); atomicAdd(&pc->lost_total, left); }
          wdone = wn;          // (the next turn of the loop draws a new window)
          if (__builtin_amdgcn_readfirstlane(leave)) { n = 0; break; }      // a surplus workgroup gives its place back
          continue;
        }
        // A workgroup that has found nothing for 20 us polls every ~14 us instead of every 0.5 us (7 us more latency for a request that
        // arrives into an idle chip): idle persistent workgroups are not free - their polls share the control lines and the fabric with
        // the tree side.  (Round 5: this alone did NOT end the seconds-long stalls of the tree wavefronts seen beside ~500 idle net
        // workgroups - not launching them did: net_launch.)
        if (__builtin_amdgcn_readfirstlane(idle)) {
          __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);
        } else {
          __builtin_amdgcn_s_sleep(16);
        }
      }
      if (tid < 8) {
        const bool mine = tid < n;
        xs[8 + tid] = mine ? sl : 0xFFFFFFFFu; xs[16 + tid] = sq; xs[24 + tid] = pl;
        unsigned long long* xb = reinterpret_cast<unsigned long long*>(xs + 32);
        xb[tid] = mine ? b0 : 0ull; xb[8 + tid] = mine ? b1 : 0ull;
        if (pa.l0) {       // the position's cache key (Connect4::key: stones and the player to move) for the in-epoch answer table
          Connect4::State ps_; ps_.bb[0] = b0; ps_.bb[1] = b1; ps_.player = pl; ps_.turn = 0;
          const unsigned long long k64 = Connect4::key(ps_) ^ (mg ? kPipeGroupSalt : 0ull);
          xb[16 + tid] = k64;
          xs[96 + tid] = pipe_l0_entry(k64, pa.l0_mask);
        }
      }
      if (tid == 0) xs[0] = n;
      wdone += n;
    }
    __syncthreads();
    { const uint64_t nowp = wall_clock64(); pf_wait += nowp - pf_mark; pf_mark = nowp; }
    const uint32_t n = xs[0];
    if (n == 0) break;
    if (tid == 0) { atomicAdd(&pc->tiles, 1ull); atomicAdd(&pc->tile_boards, static_cast<unsigned long long>(n)); atomicAdd(&pc->tile_hist[(n - 1u) % 6u], 1ull); }
    __syncthreads();
    c4::PipeIO pio{pa.res, xs + 8, xs + 16, kResStride, kResV, pa.l0, xs + 96, reinterpret_cast<const unsigned long long*>(xs + 32) + 16,
                   reinterpret_cast<const unsigned long long*>(xs + 32), xs + 24};
    // the weight pointers are made opaque per pass: otherwise the tile's loads of its (pass-invariant) head weights are
    // hoisted out of this loop and sit in ~120 registers for the whole tile (spills)
    NetPtrs npi = mg ? np1 : np;
    asm volatile("" : "+s"(npi.stem_w), "+s"(npi.stem_b), "+s"(npi.blocks), "+s"(npi.head_w), "+s"(npi.head_b), "+s"(npi.v_fc1_w));
    asm volatile("" : "+s"(npi.v_fc1_b), "+s"(npi.v_fc2_w), "+s"(npi.v_fc2_b), "+s"(npi.pi_fc_w), "+s"(npi.pi_fc_b));
    if (MODE == 2 || (MODE == 0 && n <= static_cast<uint32_t>(TSmall::TBW))) {
      if constexpr (MODE != 1) {
        c4::tile<TSmall, 4, 4, 16, 0, true>(nd, npi, nullptr, nullptr, nullptr, TSmall::TBW, nullptr, nullptr, 0u, lds_pipe, &pio);
      }
    } else {
      if constexpr (MODE != 2) {
        c4::tile<TBig, 4, 4, 16, 0, true>(nd, npi, nullptr, nullptr, nullptr, TBig::TBW, nullptr, nullptr, 0u, lds_pipe, &pio);
      }
    }
    // READY tokens of the answered slots (the tree wavefront that draws one checks the granules' tags itself, so the tokens
    // need no ordering behind the tile's stores)
    // (each into the ring of its slot's home workgroup: one ticket per board, drawn side by side)
    if (tid < n) {
      const uint32_t sl = xs[8 + tid];
      const uint32_t home = sl % pa.n_tree_wgs;
      const uint32_t pos = atomicAdd(&pa.wg[home].rtail, 1u);
      pipe_push_token(pa, home, pos, static_cast<unsigned long long>(sl) | (static_cast<unsigned long long>(xs[16 + tid]) << 16));
    }
  }
  if (tid0 == 0) { atomicAdd(&pc->prof[7], pf_wait); atomicAdd(&pc->prof[8], pf_tile); }
}

// (the net kernel's time accounting is added by its thread 0 when it leaves)
// ---- between epochs -----------------------------------------------------------------------------------------------------------------
// Requests that were still out when the epoch ended have been answered by now (the net workgroups drain the ring before they
// leave): their answers move from the granules into the slots' (v, pi) rows - the lock-step form of a pending answer - and
// into the insert log.
__global__ void k_pipe_settle(EngineParams ep, EngineArrays ar, PipeArrays pa) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  // the net workgroups left with unused window positions (head ran ahead of tail): the next epoch's windows start at tail; tokens
  // that were still in a READY ring when the epoch ended are dropped (their slots are whole in HBM: k_pipe_seed sends new ones)
  if (slot == 0) { pa.ctl->head = pa.ctl->tail; pa.ctl->head1 = pa.ctl->tail1; pa.ctl->sims_total += pa.ep->sims; }
  if (slot < pa.n_tree_wgs) pa.wg[slot].rhead = pa.wg[slot].rtail;
  if (slot >= ep.S) return;
  // slots the epoch listed for the move step and did not get to (listed as it ended): the boundary's move step
  if (ar.flags[slot] & kFlagListed) ar.mover_list[atomicAdd(&ar.ctl->mover_count, 1u)] = slot;
  else { const uint8_t s0 = ar.sstate[slot]; if (s0 == kSlotFresh || s0 == kSlotRestart) ar.mover_list[atomicAdd(&ar.ctl->mover_count, 1u)] = slot; }
  const uint8_t f = ar.flags[slot];
  if (!(f & kFlagReqOut)) return;
  const uint32_t seq = ar.req_seq[slot];
  constexpr int M = Connect4::M, P1 = Connect4::P + 1;
  float val[M + P1];
  bool bad = false;
#pragma unroll
  for (int i = 0; i < M + P1; ++i) {
    const unsigned long long g = pa.res[static_cast<size_t>(slot) * kResStride + i];
    bad = bad || static_cast<uint32_t>(g >> 32) != seq;
    val[i] = __uint_as_float(static_cast<uint32_t>(g));
  }
  if (bad) {
    // the request was never answered - the net side left early (a pipeline error: time cap, engine stop) or gave a ring position up
    // (PipeEpoch::lost).  The slot goes back
    // to the form the move step leaves a leaf in (kSlotQueued: planes + packed position written, no request out), so the next epoch
    // sends the request again and a lock-step round lists it: the engine stays usable after the error has been reported.
    if (g_ld(&pa.ctl->err) == 0u && ar.ctl->stop == 0u && pa.ep->lost == 0u) { atomicOr(&pa.ctl->err, static_cast<uint32_t>(kPipeErrTag)); return; }
    Connect4::State leaf;
    leaf.bb[0] = ar.leaf_pos[0 * static_cast<size_t>(ep.S) + slot];
    leaf.bb[1] = ar.leaf_pos[1 * static_cast<size_t>(ep.S) + slot];
    leaf.player = static_cast<uint32_t>(ar.leaf_pos[2 * static_cast<size_t>(ep.S) + slot]) & 1u;
    leaf.turn = static_cast<uint32_t>(__popcll(leaf.bb[0] | leaf.bb[1]));
    float* row = ar.canon + static_cast<size_t>(slot) * Connect4::CANON;
    for (uint32_t e = 0; e < static_cast<uint32_t>(Connect4::CANON); ++e) row[e] = Connect4::canonical_at(leaf, e);
    ar.flags[slot] = f & static_cast<uint8_t>(~kFlagReqOut);
    ar.sstate[slot] = kSlotQueued;
    return;
  }
#pragma unroll
  for (int i = 0; i < M; ++i) ar.pi[static_cast<size_t>(slot) * M + i] = val[i];
#pragma unroll
  for (int i = 0; i < P1; ++i) ar.v[static_cast<size_t>(slot) * P1 + i] = val[kResV + i];
  ar.flags[slot] = f & static_cast<uint8_t>(~kFlagReqOut);
  if (ep.cache_on) {
    const uint32_t idx = atomicAdd(&pa.ep->ins_count, 1u);
    if (idx < pa.ins_cap) {
      pa.ins_key[idx] = ar.leaf_key[slot];
      if (pa.ins_grp) pa.ins_grp[idx] = ar.leaf_group[slot];
#pragma unroll
      for (int i = 0; i < M; ++i) pa.ins_pi[static_cast<size_t>(idx) * M + i] = val[i];
#pragma unroll
      for (int i = 0; i < P1; ++i) pa.ins_v[static_cast<size_t>(idx) * P1 + i] = val[kResV + i];
    }
    // (a full log only loses cache entries, never answers: the epoch's quota bounds it, k_pipe_tree stops filling it at its end)
  }
}

// The epoch's answers go into the position cache (PlayManager::update_inferences -> insert_many, play_manager.cc:631-640): one
// wavefront per log entry, under the shard's insert lock (dev_cache.h).  The order of the inserts is the order the
// wavefronts get there - the reference's is the order its threads get the mutex.
// Two launches per epoch: `late` = 0 right behind the tree kernel - the log holds every answer a tree wavefront consumed - while
// the net side is still finishing its last tiles (it does not touch the cache); `late` = 1 after k_pipe_settle for the few answers
// that arrived when the tree side had left.
__global__ __launch_bounds__(256) void k_pipe_cache_insert(EngineArrays ar, PipeArrays pa, uint32_t late) {
  const uint32_t n = min(pa.ep->ins_count, pa.ins_cap);
  const uint32_t from = late ? min(pa.ep->ins_done, n) : 0u;
  if (!late && blockIdx.x == 0 && threadIdx.x == 0) pa.ep->ins_done = n;     // (read by the second launch only; the log does not grow before k_pipe_settle)
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const uint32_t waves = gridDim.x * (blockDim.x >> 6);
  constexpr uint32_t M = Connect4::M, P1 = Connect4::P + 1;
  for (uint32_t i = from + wave; i < n; i += waves) {
    const uint64_t key = pa.ins_key[i];
    const float p = lane < M ? pa.ins_pi[static_cast<size_t>(i) * M + lane] : 0.0f;
    const float v = lane < P1 ? pa.ins_v[static_cast<size_t>(i) * P1 + lane] : 0.0f;
    const uint32_t g = pa.ins_grp ? (pa.ins_grp[i] & 1u) : 0u;           // (one S3-FIFO per model group, play_manager.cc:195-203)
    const bool ok = g ? wave_shard_insert_locked<false>(ar.caches[1], pa.locks + pa.lock_base1, key, p, v, lane)
                      : wave_shard_insert_locked<false>(ar.cache, pa.locks, key, p, v, lane);
    if (!ok && lane == 0) atomicOr(&pa.ctl->err, static_cast<uint32_t>(kPipeErrLock));
  }
}
// ---- do two streams run side by side? -------------------------------------------------------------------------------------------
// The runtime multiplexes streams onto a few hardware queues (4 by default): two streams that share a queue run their kernels
// one after the other, and an epoch's tree and net kernels, each waiting for the other, would then sit there until the time cap.
// k_pair_wait (launched first, on the tree side's stream) waits up to 2 ms for the word k_pair_set (net side's stream) writes.
__global__ void k_pair_wait(uint32_t* flag) {
  const uint64_t t0 = wall_clock64();
  while (g_ld(flag) == 0u) {
    if (wall_clock64() - t0 > 200000ull) { g_st(flag + 1, 1u); return; }     // 2 ms: the partner never started
    __builtin_amdgcn_s_sleep(32);
  }
}
__global__ void k_pair_set(uint32_t* flag) { g_st(flag, 1u); }

#ifdef AZMI_WITH_CONVEYOR
// ---- the conveyor (conveyor_c4.h): its service kernel, and the launch-order helper --------------------------------------------------------
namespace cvn = azmi_net_dev::cv;
struct C4KeyFn {      // Connect4::key of a packed position (what k_pipe_net computes for the answer table)
  __device__ unsigned long long operator()(unsigned long long b0, unsigned long long b1, uint32_t pl) const {
    Connect4::State s_; s_.bb[0] = b0; s_.bb[1] = b1; s_.player = pl; s_.turn = 0;
    return Connect4::key(s_);
  }
};
// census of a calibration launch (pipe_calibrate): the workgroup announces itself, holds its place until `hold` ticks after the
// epoch's first workgroup and leaves; returns true in that case
__device__ __forceinline__ bool cv_census(PipeEpoch* pe, uint32_t hold, uint32_t* arrived, uint32_t* late_n) {
  const uint64_t t_start = wall_clock64();
  if (threadIdx.x == 0) {
    unsigned long long t0 = atomicCAS(&pe->t0, 0ull, static_cast<unsigned long long>(t_start));
    if (t0 == 0ull) t0 = t_start;
    atomicMax(&pe->net_late, static_cast<uint32_t>(t_start > t0 ? t_start - t0 : 0ull));
    atomicAdd(arrived, 1u);
    if (hold != 0u && t_start > t0 && t_start - t0 >= hold) atomicAdd(late_n, 1u);
  }
  if (hold == 0u) return false;
  __syncthreads();
  const unsigned long long t0 = g_ld(&pe->t0);
  while (wall_clock64() < t0 + hold) __builtin_amdgcn_s_sleep(32);
  return true;
}
__global__ __launch_bounds__(256, 1) void k_cv_line_pipe(cvn::CvArgs a, PipeEpoch* pe, uint32_t hold) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_cvl[];
  if (cv_census(pe, hold, &pe->net_arrived, &pe->net_late_n)) return;
  cvn::line_wg(a, lds_cvl);
}
__global__ __launch_bounds__(cvn::SVC_THREADS, 2) void k_cv_service(cvn::CvArgs a, cvn::SvcPipe sp, PipeEpoch* pe, uint32_t hold) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_cvs[];
  if (cv_census(pe, hold, &pe->svc_arrived, &pe->svc_late_n)) return;
  cvn::service_wg<C4KeyFn, 4, 16>(a, sp, lds_cvs, C4KeyFn{});
}
#endif   // AZMI_WITH_CONVEYOR
// the tree kernel is launched behind this one: the conv workgroups (a whole CU each) take their places first, the tree workgroups
// pack into what is left (two per CU).  Gives up after 0.5 ms: placement is speed, never correctness
__global__ void k_cv_wait_started(PipeEpoch* pe, uint32_t want) {
  const uint64_t t0 = wall_clock64();
  while (g_ld(&pe->net_arrived) < want && wall_clock64() - t0 < 50000ull) __builtin_amdgcn_s_sleep(32);
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------
struct PipeState {
  PipeArrays pa{};
  std::vector<void*> allocs;
  hipStream_t net_stream = nullptr;
  hipStream_t paired_with = nullptr;     // the tree-side stream net_stream was last seen running beside
  bool paired = false;
  std::vector<hipStream_t> parked;       // streams that shared a hardware queue with a tree-side stream (kept: destroying one frees its queue slot for the next try)
  uint32_t* pair_flag = nullptr;
  hipEvent_t ev_go = nullptr, ev_net = nullptr;
  uint32_t net_wgs = 0, tree_wgs = 0, tree_block = 256;
  uint32_t net_launch = 0;          // net workgroups an epoch really launches: net_wgs, held to what the engine's slots can keep busy
  std::vector<hipEvent_t> tev;      // timing events: four per epoch of a run (net kernel start / end, tree kernel start / end)
  size_t lds_bytes = 0;
  bool sized = false, x3 = false;   // pipe_size_net has run for a net of this precision tier
  bool np1_valid = false;           // this call has a second net (model group 1): its weights
  azmi_net_dev::NetPtrs np1{};
  int kind = 0;                     // PipePlan::kind of the current call (which tree kernel)
  bool calibrated = false;          // net_wgs has been measured beside the tree workgroups (pipe_calibrate)
  bool svc_paired = false; hipStream_t svc_paired_with = nullptr;      // the conveyor's service stream runs beside `svc_paired_with` and net_stream
  uint32_t calib_rounds = 0;
  // balance between the two sides (pipe_balance): rings are allocated for tree_wgs_alloc workgroups, `places` = tree + net workgroups
  // the chip was measured to hold, the cumulative counters are those of the previous call
  uint32_t tree_wgs_alloc = 0, tree_wgs_min = 1, tree_wgs_default = 1, places = 0;
  bool balance = false;
  unsigned long long bal_sims = 0, bal_boards = 0;
  uint32_t lost_seen = 0;           // PipeCtl::lost_total already reported
  // the conveyor (conveyor_c4.h): the net side as lines of weight-stationary conv workgroups + one service workgroup per line
  bool cv_on = false;               // this call runs it (else: the tile kernel k_pipe_net)
  hipStream_t svc_stream = nullptr;
  hipEvent_t ev_svc = nullptr;
  uint32_t cv_lines = 0, cv_lines_alloc = 0, cv_nwg = 0, cv_heads = 2;
  uint32_t* cv_xh = nullptr; uint8_t* cv_xt = nullptr; uint8_t* cv_xs = nullptr; uint32_t* cv_meta = nullptr; unsigned long long* cv_stat = nullptr;
  bool cv_calibrated = false;
  const void* l0_nets[2] = {nullptr, nullptr};      // the weight images the answer table's entries were computed with (group 0, group 1)
  // CU split (AZMI_PIPE_CU_SPLIT=T): the tree kernel on a stream masked to T CUs, the net kernel on the others
  uint32_t cu_split = 0, cu_total = 0;
  hipStream_t tree_stream = nullptr;
  hipEvent_t ev_tree = nullptr;
};
void pipe_state_free(PipeState* p) {
  if (!p) return;
  for (void* q : p->allocs) (void)hipFree(q);
  if (p->net_stream) (void)hipStreamDestroy(p->net_stream);
  if (p->svc_stream) (void)hipStreamDestroy(p->svc_stream);
  if (p->tree_stream) (void)hipStreamDestroy(p->tree_stream);
  if (p->ev_tree) (void)hipEventDestroy(p->ev_tree);
  if (p->ev_svc) (void)hipEventDestroy(p->ev_svc);
  for (hipStream_t q : p->parked) (void)hipStreamDestroy(q);
  if (p->ev_go) (void)hipEventDestroy(p->ev_go);
  if (p->ev_net) (void)hipEventDestroy(p->ev_net);
  for (hipEvent_t e : p->tev) (void)hipEventDestroy(e);
  delete p;
}

}  // namespace azmi

// defined in engine.hip: the move step of a split round as its own launch, and the restart / retire bookkeeping
int azmi_host_launch_move_step(azmi_pm* pm, hipStream_t st);
int azmi_host_launch_assign(azmi_pm* pm, hipStream_t st, uint32_t count_round);

namespace {

template <class T>
int pipe_alloc(PipeState* ps, T*& p, size_t n) {
  void* q = nullptr;
  const size_t sz = std::max<size_t>(n, 1) * sizeof(T);
  AZMI_HIP_TRY(hipMalloc(&q, sz));
  ps->allocs.push_back(q);
  // hipMemset is asynchronous with respect to the host and runs on the null stream, which the (non-blocking) streams of the
  // pipeline do not wait for: without the synchronisation the zeroes can land while the first epoch is already running
  // (seen: the epoch block wiped under a running epoch of the second engine of a process)
  AZMI_HIP_TRY(hipMemset(q, 0, sz));
  AZMI_HIP_TRY(hipStreamSynchronize(nullptr));
  p = static_cast<T*>(q);
  return AZMI_OK;
}

// the persistent net kernel's instantiations: tile selection x precision tier
using PipeNetFn = void (*)(azmi_net_dev::NetDesc, azmi_net_dev::NetPtrs, azmi_net_dev::NetPtrs, PipeArrays);
PipeNetFn pipe_net_fn(int mode, bool x3) {
  if (x3) return mode == 1 ? &k_pipe_net<1, true> : mode == 2 ? &k_pipe_net<2, true> : &k_pipe_net<0, true>;
  return mode == 1 ? &k_pipe_net<1, false> : mode == 2 ? &k_pipe_net<2, false> : &k_pipe_net<0, false>;
}
int pipe_launch_net(PipeState* ps, const azmi_net_c4_view& view, int mode, uint32_t wgs, hipStream_t st, const PipeArrays& pa);
// how a call is driven: which tree kernel, which model groups have a net behind their request ring
struct PipePlan {
  int kind = 0;                       // 0 = not supported, 1 = fast kernel (plain PUCT, one group), 2 = generic kernel (any search, <= 2 groups)
  bool tree_only = false;             // no net kernel at all (every seat RANDOM)
  uint32_t net_groups = 0;            // bit g: nets[g] serves group g's ring
  azmi_net_c4_view view[2]{};         // view[g] of group g's net (view[1] = view[0] when group 1 has none)
};
int pipe_size_net(azmi_pm* pm, PipeState* ps, const azmi_net_c4_view& view);

// a stream for one side of the pipeline: plain, or - with the CU split on - masked to the side's CUs (bit i of the mask = CU i as the
// runtime numbers them; which physical CUs those are is nothing this code assumes: the two masks are disjoint, the census measures
// what fits)
int pipe_make_stream(PipeState* ps, hipStream_t* out, bool tree_side) {
  if (ps->cu_split == 0u) { AZMI_HIP_TRY(hipStreamCreateWithFlags(out, hipStreamNonBlocking)); return AZMI_OK; }
  std::vector<uint32_t> mask((ps->cu_total + 31u) / 32u, 0u);
  for (uint32_t cu = 0; cu < ps->cu_total; ++cu)
    if ((cu < ps->cu_split) == tree_side) mask[cu / 32u] |= 1u << (cu % 32u);
  AZMI_HIP_TRY(hipExtStreamCreateWithCUMask(out, static_cast<uint32_t>(mask.size()), mask.data()));
  return AZMI_OK;
}

uint32_t pipe_tree_wgs_for(uint32_t S) {
  // tree workgroups: a slot lives in ONE workgroup for an epoch (its home), so a workgroup's 32 lane-groups serve S / workgroups
  // slots; about a third of the slots is with the net or in a ring at any time.  Every tree workgroup takes a place from the net
  // side, and at 4096 slots both sides are full (tree wavefronts 94 % busy, net workgroups 87-91 %): measured, M simulations/s at 96 /
  // 112 / 128 / 144 / 160 workgroups: 89 / 101 / 102-105 / 100 / 88.  AZMI_PIPE_TREE_WGS sets another count.
  uint32_t t = std::min<uint32_t>(128u, std::max<uint32_t>(1u, (S + 31u) / 32u));
  if (const char* e = getenv("AZMI_PIPE_TREE_WGS")) t = static_cast<uint32_t>(std::max(1, atoi(e)));
  return std::min<uint32_t>(t, std::max<uint32_t>(1u, S));
}

int pipe_create(azmi_pm* pm, size_t tile_lds) {
  auto ps = new PipeState();
  struct Guard { PipeState* p; ~Guard() { if (p) pipe_state_free(p); } } guard{ps};     // (a failed set-up leaves nothing half-built behind)
  PipeArrays& pa = ps->pa;
  const uint32_t S = pm->ep.S;
  ps->tree_wgs = ps->tree_wgs_default = pipe_tree_wgs_for(S);
  pa.n_tree_wgs = ps->tree_wgs;
  // the tree side may grow or shrink between calls with the share of leaves that reach the net (pipe_balance): unless the count was
  // fixed by hand, rings exist for 1.5 x the default and are sized for a third of it
  ps->balance = getenv("AZMI_PIPE_TREE_WGS") == nullptr && getenv("AZMI_PIPE_NET_WGS") == nullptr && getenv("AZMI_PIPE_NO_BALANCE") == nullptr && ps->tree_wgs >= 48u;
  ps->tree_wgs_alloc = ps->balance ? ps->tree_wgs + ps->tree_wgs / 2u : ps->tree_wgs;
  ps->tree_wgs_min = ps->balance ? ps->tree_wgs / 3u : ps->tree_wgs;
  {   // a workgroup's READY ring: at least twice its slots (every slot has at most one token out), a power of two
    const uint32_t per_wg = (S + ps->tree_wgs_min - 1u) / ps->tree_wgs_min;
    uint32_t sh = 6;
    while ((1u << sh) < 2u * per_wg + 2u * kTreeWindow) ++sh;
    pa.rshift = sh;
  }
  int rc = pipe_alloc(ps, pa.ctl, 1);
  if (rc == AZMI_OK) rc = pipe_alloc(ps, pa.wg, ps->tree_wgs_alloc);
  if (rc == AZMI_OK && getenv("AZMI_PIPE_POS0")) {
    // test hook: the rings' free-running 32-bit positions start here instead of at 0 (a long run wraps them after ~2 minutes:
    // tests/test_gpu_pipeline.py starts just below 2^32)
    const uint32_t p0 = static_cast<uint32_t>(strtoul(getenv("AZMI_PIPE_POS0"), nullptr, 0));
    PipeCtl h{};
    h.head = h.tail = p0;
    if (hipMemcpy(pa.ctl, &h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) return azmi_host_fail(AZMI_ERR_NO_DEVICE, "pipeline: control block upload failed");
    std::vector<PipeWg> hw(ps->tree_wgs_alloc);
    for (auto& w : hw) { w = PipeWg{}; w.rhead = w.rtail = p0; }
    if (hipMemcpy(pa.wg, hw.data(), hw.size() * sizeof(PipeWg), hipMemcpyHostToDevice) != hipSuccess) return azmi_host_fail(AZMI_ERR_NO_DEVICE, "pipeline: control block upload failed");
  }
  if (rc == AZMI_OK) rc = pipe_alloc(ps, pa.ep, 1);
  if (rc == AZMI_OK) rc = pipe_alloc(ps, pa.ring, static_cast<size_t>(kPipeRing) * kReqGranules);
  if (rc == AZMI_OK) rc = pipe_alloc(ps, pa.rring, static_cast<size_t>(ps->tree_wgs_alloc) << pa.rshift);
  if (rc == AZMI_OK) rc = pipe_alloc(ps, pa.res, static_cast<size_t>(S) * kResStride);
  if (rc != AZMI_OK) return rc;
  // inline budget of a pass: a group runs up to 8 simulations whose answers are at hand (cache hits, terminal leaves) before its slot
  // re-queues (M simulations/s at 3 / 5 / 6 / 8 / 16: 82 / 89-96 / 102 / 102-110 / 97-100); a pass may also end early once fewer than
  // min_active of its groups are still running (round 3: 3; with home workgroups 0 - the stragglers' slots would only queue for the
  // same four wavefronts).  An engine created with an explicit max_inline keeps it as its budget.
  pa.max_inline = getenv("AZMI_PIPE_INLINE") ? static_cast<uint32_t>(std::max(1, atoi(getenv("AZMI_PIPE_INLINE")))) : (pm->max_inline_explicit ? pm->ep.max_inline : 8u);
  pa.min_active = getenv("AZMI_PIPE_MIN_ACTIVE") ? static_cast<uint32_t>(std::max(0, atoi(getenv("AZMI_PIPE_MIN_ACTIVE")))) : 0u;
  // net side: a workgroup draws a 6-request window (the 6-board tile: capacity) when at least big_at requests wait in the ring, else a
  // 3-request window (the 3-board tile: 39 us instead of 60 alone - with 4096 slots a slot's wait for its answer is what is short)
  pa.big_at = getenv("AZMI_PIPE_BIG_AT") ? static_cast<uint32_t>(std::max(0, atoi(getenv("AZMI_PIPE_BIG_AT")))) : 1u;      // 1: the workgroup's own experience (round 5); round 4: 96      // (same-box A/B at 4096 slots: 48 -> 100.0, 96 / 128 / 192 -> 102.4 M simulations/s; 16384 slots: no difference)
  pa.take_wait = getenv("AZMI_PIPE_TAKE_WAIT") ? static_cast<uint32_t>(std::max(0, atoi(getenv("AZMI_PIPE_TAKE_WAIT")))) : 0u;
  (void)tile_lds;
  if (const char* e = getenv("AZMI_PIPE_CU_SPLIT")) {
    hipDeviceProp_t prop;
    AZMI_HIP_TRY(hipGetDeviceProperties(&prop, pm->device));
    ps->cu_total = static_cast<uint32_t>(prop.multiProcessorCount);
    ps->cu_split = static_cast<uint32_t>(std::max(0, std::min(atoi(e), static_cast<int>(ps->cu_total) - 1)));
    if (ps->cu_split) {
      { const int rc2 = pipe_make_stream(ps, &ps->tree_stream, true); if (rc2 != AZMI_OK) return rc2; }
      AZMI_HIP_TRY(hipEventCreateWithFlags(&ps->ev_tree, hipEventDisableTiming));
      // two tree workgroups per CU (256 registers x 4 wavefronts each)
      if (!getenv("AZMI_PIPE_TREE_WGS")) { ps->tree_wgs = ps->tree_wgs_default = std::min(ps->tree_wgs_alloc, 2u * ps->cu_split); pa.n_tree_wgs = ps->tree_wgs; }
      ps->balance = false;
    }
  }
  { const int rc2 = pipe_make_stream(ps, &ps->net_stream, false); if (rc2 != AZMI_OK) return rc2; }
  AZMI_HIP_TRY(hipEventCreateWithFlags(&ps->ev_go, hipEventDisableTiming));
  AZMI_HIP_TRY(hipEventCreateWithFlags(&ps->ev_net, hipEventDisableTiming));
  guard.p = nullptr;
  pm->pipe = ps;       // owned by the engine from here on (freed with it)
  return AZMI_OK;
}

// The net side for THIS net: LDS per workgroup (bf16: the 6-board tile, two workgroups per CU; bf16x3: its 16-plane tile, one per CU)
// and the first guess of the workgroup count = the runtime's own occupancy answer x CUs - an UPPER bound; pipe_calibrate then measures
// what really runs beside the tree workgroups (a persistent kernel only works when every workgroup is resident, and how the dispatcher
// deals workgroups to shader engines is nothing this code may assume).  Called again when a call brings a net of the other tier.
int pipe_size_net(azmi_pm* pm, PipeState* ps, const azmi_net_c4_view& view) {
  const bool x3 = view.x3 != 0;
  if (ps->sized && ps->x3 == x3) return AZMI_OK;
  ps->x3 = x3;
  ps->lds_bytes = (x3 ? azmi_net_dev::c4::TileBigX3::LDS_BYTES : azmi_net_dev::c4::TileBig::LDS_BYTES) + kPipeXs;
  for (int mode = 0; mode < 3; ++mode)
    AZMI_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(pipe_net_fn(mode, x3)), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ps->lds_bytes)));
  hipDeviceProp_t prop;
  AZMI_HIP_TRY(hipGetDeviceProperties(&prop, pm->device));
  int per_cu = 0;
  AZMI_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(pipe_net_fn(0, x3)), 256, ps->lds_bytes));
  uint32_t net = static_cast<uint32_t>(std::max(1, per_cu)) * (static_cast<uint32_t>(prop.multiProcessorCount) - ps->cu_split);
  if (const char* e = getenv("AZMI_PIPE_NET_WGS")) net = static_cast<uint32_t>(atoi(e));
  ps->net_wgs = std::max<uint32_t>(1u, net);
  ps->calibrated = false;
  ps->places = 0;
  ps->sized = true;
  // the bf16x3 tiles cost three times the matrix work and get one workgroup per CU: the net side is what is short, and every tree
  // wavefront beside a tile slows it - half the tree workgroups (M simulations/s at 128 / 96 / 64: 27.7 / 33.4 / 36.2)
  if (ps->balance) {
    ps->tree_wgs = x3 ? std::max(ps->tree_wgs_min, ps->tree_wgs_default / 2u) : ps->tree_wgs_default;
    ps->pa.n_tree_wgs = ps->tree_wgs;
  }
  return AZMI_OK;
}
int pipe_launch_net(PipeState* ps, const azmi_net_c4_view& view, int mode, uint32_t wgs, hipStream_t st, const PipeArrays& pa) {
  // AZMI_WHATIF_NET_DEPTH=n (a measurement knob, never a result: the answers are another net's): the tiles run n residual blocks
  // instead of the net's - how much of the pipeline's rate is the tile's latency
  azmi_net_dev::NetDesc nd = view.nd;
  static const int whatif_depth = getenv("AZMI_WHATIF_NET_DEPTH") ? atoi(getenv("AZMI_WHATIF_NET_DEPTH")) : 0;
  if (whatif_depth > 0 && whatif_depth < nd.depth) nd.depth = whatif_depth;
  pipe_net_fn(mode, ps->x3)<<<wgs, 256, ps->lds_bytes, st>>>(nd, view.np, ps->np1_valid ? ps->np1 : view.np, pa);
  AZMI_HIP_TRY(hipGetLastError());
  return AZMI_OK;
}

// net_stream must run beside `st` (see k_pair_wait): checked once per tree-side stream; a stream that shares st's hardware queue is
// parked and another one is tried.  Fails when no stream can be found (GPU_MAX_HW_QUEUES = 1): the caller drives the engine with
// azmi_run_rounds then.
int pipe_pair_streams(PipeState* ps, hipStream_t st) {
  if (ps->paired && ps->paired_with == st) return AZMI_OK;
  if (!ps->pair_flag) { const int rc = pipe_alloc(ps, ps->pair_flag, 4); if (rc != AZMI_OK) return rc; }
  for (int attempt = 0; attempt < 12; ++attempt) {
    AZMI_HIP_TRY(hipMemsetAsync(ps->pair_flag, 0, 4 * sizeof(uint32_t), st));
    AZMI_HIP_TRY(hipEventRecord(ps->ev_go, st));
    AZMI_HIP_TRY(hipStreamWaitEvent(ps->net_stream, ps->ev_go, 0));
    k_pair_wait<<<1, 1, 0, st>>>(ps->pair_flag);
    AZMI_HIP_TRY(hipGetLastError());
    k_pair_set<<<1, 1, 0, ps->net_stream>>>(ps->pair_flag);
    AZMI_HIP_TRY(hipGetLastError());
    uint32_t h[2] = {0, 0};
    AZMI_HIP_TRY(hipStreamSynchronize(ps->net_stream));
    AZMI_HIP_TRY(hipMemcpyAsync(h, ps->pair_flag, sizeof(h), hipMemcpyDeviceToHost, st));
    AZMI_HIP_TRY(hipStreamSynchronize(st));
    if (h[1] == 0u) { ps->paired = true; ps->paired_with = st; return AZMI_OK; }
    ps->parked.push_back(ps->net_stream);          // shares st's hardware queue
    ps->net_stream = nullptr;
    { const int rc2 = pipe_make_stream(ps, &ps->net_stream, false); if (rc2 != AZMI_OK) return rc2; }
  }
  return azmi_host_fail(AZMI_ERR_STATE, "azmi_run_pipeline: no second stream runs beside the caller's (one hardware queue? GPU_MAX_HW_QUEUES); "
                        "the pipeline needs its tree and net kernels on the chip together - use azmi_run_rounds");
}

// The conveyor's service kernel is a THIRD persistent kernel: its stream must share a hardware queue with neither the caller's stream
// nor net_stream (the runtime deals streams onto a few queues round-robin; a kernel queued behind a persistent one starts when the
// epoch is over).  Same probe as above, against both partners; streams that fail are parked.
int pipe_pair_svc(PipeState* ps, hipStream_t st) {
  if (ps->svc_paired && ps->svc_paired_with == st) return AZMI_OK;
  if (!ps->pair_flag) { const int rc = pipe_alloc(ps, ps->pair_flag, 4); if (rc != AZMI_OK) return rc; }
  for (int attempt = 0; attempt < 12; ++attempt) {
    bool ok = true;
    for (hipStream_t partner : {st, ps->net_stream}) {
      AZMI_HIP_TRY(hipMemsetAsync(ps->pair_flag, 0, 4 * sizeof(uint32_t), partner));
      AZMI_HIP_TRY(hipEventRecord(ps->ev_go, partner));
      AZMI_HIP_TRY(hipStreamWaitEvent(ps->svc_stream, ps->ev_go, 0));
      k_pair_wait<<<1, 1, 0, partner>>>(ps->pair_flag);
      AZMI_HIP_TRY(hipGetLastError());
      k_pair_set<<<1, 1, 0, ps->svc_stream>>>(ps->pair_flag);
      AZMI_HIP_TRY(hipGetLastError());
      uint32_t h[2] = {0, 0};
      AZMI_HIP_TRY(hipStreamSynchronize(ps->svc_stream));
      AZMI_HIP_TRY(hipMemcpyAsync(h, ps->pair_flag, sizeof(h), hipMemcpyDeviceToHost, partner));
      AZMI_HIP_TRY(hipStreamSynchronize(partner));
      if (h[1] != 0u) { ok = false; break; }
    }
    if (ok) { ps->svc_paired = true; ps->svc_paired_with = st; return AZMI_OK; }
    ps->parked.push_back(ps->svc_stream);
    ps->svc_stream = nullptr;
    AZMI_HIP_TRY(hipStreamCreateWithFlags(&ps->svc_stream, hipStreamNonBlocking));
  }
  return azmi_host_fail(AZMI_ERR_STATE, "azmi_run_pipeline: no third stream runs beside the caller's and the net side's (GPU_MAX_HW_QUEUES < 3?): the conveyor "
                        "needs three kernels on the chip together; AZMI_PIPE_NET=tiles runs the tile kernel");
}

void pipe_launch_tree(azmi_pm* pm, PipeState* ps, const PipeArrays& pa, hipStream_t st, bool prof);
#ifdef AZMI_WITH_CONVEYOR
// ---- the conveyor's host side ---------------------------------------------------------------------------------------------------------
// When the net side can be the conveyor (conveyor_c4.h): the bf16 tier, one model group, an even number of residual blocks (a conv
// workgroup = two blocks).  Round 5: opt-in (AZMI_PIPE_NET=conveyor; an error where it cannot run) - bit for bit the tile kernel's
// answers, but at 0.25 of the MFMA peak alone on the chip against the tile kernel's 0.45 (DESIGN section 4.5b says what is missing).
bool cv_eligible(const azmi_pm* pm, const PipePlan& plan) {
  const char* e = getenv("AZMI_PIPE_NET");
  if (!e || strcmp(e, "conveyor") != 0) return false;
  const azmi_net_dev::NetDesc& nd = plan.view[0].nd;
  return !plan.tree_only && plan.kind == 1 && plan.net_groups == 1u && pm->ep.num_groups == 1u && plan.view[0].x3 == 0 && nd.depth >= 2 && nd.depth % 2 == 0;
}
uint32_t cv_default_lines(uint32_t cus, uint32_t nwg, uint32_t tree_wgs) {
  // CUs: a conv workgroup each (it takes its CU's registers whole), a service workgroup each (two would fit a CU, but the dispatcher
  // deals them one per CU while CUs are free), two tree workgroups per CU - inside nine tenths of the chip: the three kernels' workgroups
  // are dealt at the same time, and a conv workgroup needs a CU that nothing else has touched (cv_calibrate gives lines up from there)
  uint32_t lines = 1;
  while ((lines + 1u) * (nwg + 1u) + (tree_wgs + 1u) / 2u <= cus - cus / 10u) ++lines;
  return lines;
}
int cv_setup(azmi_pm* pm, PipeState* ps, const azmi_net_c4_view& view) {
  const uint32_t nwg = static_cast<uint32_t>(view.nd.depth) / 2u;
  hipDeviceProp_t prop;
  AZMI_HIP_TRY(hipGetDeviceProperties(&prop, pm->device));
  const uint32_t cus = static_cast<uint32_t>(prop.multiProcessorCount);
  if (!ps->svc_stream) {
    AZMI_HIP_TRY(hipStreamCreateWithFlags(&ps->svc_stream, hipStreamNonBlocking));
    AZMI_HIP_TRY(hipEventCreateWithFlags(&ps->ev_svc, hipEventDisableTiming));
    AZMI_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cv_line_pipe), hipFuncAttributeMaxDynamicSharedMemorySize, cvn::LINE_LDS_BYTES));
    AZMI_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cv_service), hipFuncAttributeMaxDynamicSharedMemorySize, cvn::SVC_LDS_BYTES));
  }
  const uint32_t want_alloc = std::max<uint32_t>(1u, cus / nwg);
  if (ps->cv_nwg != nwg || ps->cv_lines_alloc < want_alloc) {
    // (buffers of an earlier net shape stay in ps->allocs until the engine goes: a change of depth on one engine is rare)
    const size_t rings = static_cast<size_t>(want_alloc) * (nwg + 1u);
    int rc = pipe_alloc(ps, ps->cv_xh, rings * 128);
    if (rc == AZMI_OK) rc = pipe_alloc(ps, ps->cv_xt, rings * cvn::X_T);
    if (rc == AZMI_OK) rc = pipe_alloc(ps, ps->cv_xs, rings * cvn::X_S);
    if (rc == AZMI_OK) rc = pipe_alloc(ps, ps->cv_meta, static_cast<size_t>(want_alloc) * cvn::MGRP * 32);
    if (rc == AZMI_OK && !ps->cv_stat) rc = pipe_alloc(ps, ps->cv_stat, 32);
    if (rc != AZMI_OK) return rc;
    ps->cv_nwg = nwg; ps->cv_lines_alloc = want_alloc; ps->cv_calibrated = false;
    ps->cv_lines = 0;
  }
  if (ps->cv_lines == 0) {
    uint32_t lines = cv_default_lines(cus, nwg, ps->tree_wgs);
    if (const char* e = getenv("AZMI_CV_LINES")) lines = static_cast<uint32_t>(std::max(1, atoi(e)));
    ps->cv_lines = std::min(lines, ps->cv_lines_alloc);
    ps->cv_heads = getenv("AZMI_CV_HEADS") ? static_cast<uint32_t>(std::min(2, std::max(1, atoi(getenv("AZMI_CV_HEADS"))))) : 2u;
  }
  return AZMI_OK;
}
cvn::CvArgs cv_args(const PipeState* ps, const azmi_net_c4_view& view, const PipeArrays& pa) {
  cvn::CvArgs a{};
  a.nd = view.nd; a.np = view.np;
  a.xh = ps->cv_xh; a.xt = ps->cv_xt; a.xs = ps->cv_xs; a.meta = ps->cv_meta;
  a.lines = ps->cv_lines; a.nwg = ps->cv_nwg;
  a.dbg_flags = getenv("AZMI_CV_DBG") ? static_cast<uint32_t>(atoi(getenv("AZMI_CV_DBG"))) : 0u;
  a.err = &pa.ctl->err; a.stop = &pa.ep->stop; a.cap_ticks = pa.cap_ticks; a.stat = ps->cv_stat;
  return a;
}
cvn::SvcPipe cv_svc(const PipeState* ps, const PipeArrays& pa) {
  cvn::SvcPipe sp{};
  sp.ring = pa.ring; sp.head = &pa.ctl->head; sp.tail = &pa.ctl->tail;
  sp.stop = &pa.ep->stop; sp.tree_done = &pa.ep->tree_done; sp.tree_arrived = &pa.ep->tree_arrived;
  sp.res = pa.res; sp.l0 = pa.l0; sp.l0_mask = pa.l0_mask;
  sp.rring = pa.rring; sp.rshift = pa.rshift; sp.n_tree_wgs = pa.n_tree_wgs; sp.wg_rtail = reinterpret_cast<uint32_t*>(pa.wg);
  sp.tiles = &pa.ctl->tiles; sp.tile_boards = &pa.ctl->tile_boards; sp.lost = &pa.ep->lost; sp.lost_total = &pa.ctl->lost_total;
  sp.dbg = pa.ctl->dbg; sp.n_heads = ps->cv_heads;
  return sp;
}
// one epoch's conveyor launches: the ring headers are zeroed on `st` (ahead of ev_go), the conv lines go to net_stream, the service
// workgroups to svc_stream; both streams are joined back into `st` by the caller (ev_net, ev_svc)
int cv_launch(PipeState* ps, const azmi_net_c4_view& view, const PipeArrays& pa, uint32_t hold) {
  const cvn::CvArgs a = cv_args(ps, view, pa);
  k_cv_line_pipe<<<ps->cv_lines * ps->cv_nwg, 256, cvn::LINE_LDS_BYTES, ps->net_stream>>>(a, pa.ep, hold);
  AZMI_HIP_TRY(hipGetLastError());
  k_cv_service<<<ps->cv_lines, cvn::SVC_THREADS, cvn::SVC_LDS_BYTES, ps->svc_stream>>>(a, cv_svc(ps, pa), pa.ep, hold);
  AZMI_HIP_TRY(hipGetLastError());
  return AZMI_OK;
}
int cv_zero_headers(PipeState* ps, hipStream_t st) {
  AZMI_HIP_TRY(hipMemsetAsync(ps->cv_xh, 0, static_cast<size_t>(ps->cv_lines) * (ps->cv_nwg + 1u) * 512u, st));
  return AZMI_OK;
}
// Measures how many LINES run beside the tree workgroups (pipe_calibrate's census, for the conveyor): every workgroup of the three
// kernels holds its place; a conv workgroup that the chip has no CU for starts late and is counted, and the line count gives way.
int cv_calibrate(azmi_pm* pm, PipeState* ps, hipStream_t st, const azmi_net_c4_view& view) {
  if (ps->cv_calibrated || getenv("AZMI_PIPE_NO_CALIBRATE")) { ps->cv_calibrated = true; return AZMI_OK; }
  PipeArrays pa = ps->pa;
  pa.census_hold = 100000u;
  for (int attempt = 0; attempt < 24; ++attempt) {
    AZMI_HIP_TRY(hipMemsetAsync(pa.ep, 0, sizeof(PipeEpoch), st));
    { const int rc = cv_zero_headers(ps, st); if (rc != AZMI_OK) return rc; }
    AZMI_HIP_TRY(hipEventRecord(ps->ev_go, st));
    AZMI_HIP_TRY(hipStreamWaitEvent(ps->net_stream, ps->ev_go, 0));
    AZMI_HIP_TRY(hipStreamWaitEvent(ps->svc_stream, ps->ev_go, 0));
    { const int rc = cv_launch(ps, view, pa, pa.census_hold); if (rc != AZMI_OK) return rc; }
    k_cv_wait_started<<<1, 1, 0, st>>>(pa.ep, ps->cv_lines * ps->cv_nwg);
    pipe_launch_tree(pm, ps, pa, st, false);
    AZMI_HIP_TRY(hipGetLastError());
    AZMI_HIP_TRY(hipEventRecord(ps->ev_net, ps->net_stream));
    AZMI_HIP_TRY(hipEventRecord(ps->ev_svc, ps->svc_stream));
    AZMI_HIP_TRY(hipStreamWaitEvent(st, ps->ev_net, 0));
    AZMI_HIP_TRY(hipStreamWaitEvent(st, ps->ev_svc, 0));
    PipeEpoch he;
    AZMI_HIP_TRY(hipMemcpyAsync(&he, pa.ep, sizeof(he), hipMemcpyDeviceToHost, st));
    AZMI_HIP_TRY(hipStreamSynchronize(st));
    ps->calib_rounds = static_cast<uint32_t>(attempt) + 1u;
    const uint32_t late = he.tree_late_n + he.net_late_n + he.svc_late_n;
    if (late == 0u && he.tree_arrived == ps->tree_wgs && he.net_arrived == ps->cv_lines * ps->cv_nwg && he.svc_arrived == ps->cv_lines) {
      ps->cv_calibrated = true;
      return AZMI_OK;
    }
    if (ps->cv_lines <= 1u) break;
    const uint32_t give = std::max<uint32_t>(1u, (late + ps->cv_nwg - 1u) / ps->cv_nwg);
    ps->cv_lines = ps->cv_lines > give ? ps->cv_lines - give : 1u;
  }
  return azmi_host_fail(AZMI_ERR_STATE, "azmi_run_pipeline: the chip does not hold %u tree workgroups beside one conveyor line (another tenant on the GPU? no third "
                        "hardware queue?); AZMI_PIPE_NET=tiles runs the tile kernel", ps->tree_wgs);
}
#else      // the default library: the conveyor is not compiled in (scripts/experiments/conveyor_c4.h)
static int cv_absent() { return azmi_host_fail(AZMI_ERR_STATE, "this build has no conveyor (an experiment since round 6: build with -DAZMI_WITH_CONVEYOR); the tile kernel is the net side"); }
bool cv_eligible(const azmi_pm*, const PipePlan&) { return false; }
uint32_t cv_default_lines(uint32_t, uint32_t, uint32_t) { return 0u; }
int cv_setup(azmi_pm*, PipeState*, const azmi_net_c4_view&) { return cv_absent(); }
int cv_launch(PipeState*, const azmi_net_c4_view&, const PipeArrays&, uint32_t) { return cv_absent(); }
int cv_zero_headers(PipeState*, hipStream_t) { return cv_absent(); }
int cv_calibrate(azmi_pm*, PipeState*, hipStream_t, const azmi_net_c4_view&) { return cv_absent(); }
#endif     // AZMI_WITH_CONVEYOR

// Measures how many net workgroups run BESIDE the tree workgroups: both persistent kernels are launched as they are in an epoch, with
// PipeArrays::census_hold set - every workgroup that gets a place holds it until 1 ms after the first one started (far longer than the
// two launches are apart) and leaves; one that the chip has no place for starts only then, finds itself late and is counted.  The net side gives up as many workgroups as came late (tree workgroups that came
// late count too: the net kernel had taken their places) until nobody is late.  No constant of the part's dealing order is involved;
// a second tenant on the chip, a CU mask or a new runtime change the answer, not the code.  (AZMI_PIPE_NO_CALIBRATE=1 skips it.)
int pipe_calibrate(azmi_pm* pm, PipeState* ps, hipStream_t st, const azmi_net_c4_view& view) {
  if (ps->calibrated || getenv("AZMI_PIPE_NO_CALIBRATE")) { ps->calibrated = true; return AZMI_OK; }
  PipeArrays pa = ps->pa;
  pa.census_hold = 100000u;
  for (int attempt = 0; attempt < 24; ++attempt) {
    AZMI_HIP_TRY(hipMemsetAsync(pa.ep, 0, sizeof(PipeEpoch), st));
    AZMI_HIP_TRY(hipEventRecord(ps->ev_go, st));
    AZMI_HIP_TRY(hipStreamWaitEvent(ps->net_stream, ps->ev_go, 0));
    if (ps->tree_stream) AZMI_HIP_TRY(hipStreamWaitEvent(ps->tree_stream, ps->ev_go, 0));
    pipe_launch_tree(pm, ps, pa, ps->tree_stream ? ps->tree_stream : st, false);
    AZMI_HIP_TRY(hipGetLastError());
    if (ps->tree_stream) { AZMI_HIP_TRY(hipEventRecord(ps->ev_tree, ps->tree_stream)); AZMI_HIP_TRY(hipStreamWaitEvent(st, ps->ev_tree, 0)); }
    { const int rc = pipe_launch_net(ps, view, 0, ps->net_wgs, ps->net_stream, pa); if (rc != AZMI_OK) return rc; }
    AZMI_HIP_TRY(hipEventRecord(ps->ev_net, ps->net_stream));
    AZMI_HIP_TRY(hipStreamWaitEvent(st, ps->ev_net, 0));
    PipeEpoch he;
    AZMI_HIP_TRY(hipMemcpyAsync(&he, pa.ep, sizeof(he), hipMemcpyDeviceToHost, st));
    AZMI_HIP_TRY(hipStreamSynchronize(st));
    ps->calib_rounds = static_cast<uint32_t>(attempt) + 1u;
    const uint32_t late = he.tree_late_n + he.net_late_n;
    if (late == 0u && he.tree_arrived == ps->tree_wgs && he.net_arrived == ps->net_wgs) {
      // (a workgroup serves ONE model group's ring for an epoch - its index's parity when both groups have a net: with one workgroup
      // group 1's ring would never be served and every epoch would run into the time cap - ADVICE r4)
      if (pa.net_groups == 3u && ps->net_wgs < 2u) break;
      ps->calibrated = true; ps->places = ps->x3 ? 0u : ps->tree_wgs + ps->net_wgs; return AZMI_OK;
    }
    if (ps->net_wgs <= 1u) break;
    ps->net_wgs = ps->net_wgs > late + 1u ? ps->net_wgs - std::max<uint32_t>(late, 1u) : 1u;
  }
  return azmi_host_fail(AZMI_ERR_STATE, "azmi_run_pipeline: the chip does not hold %u tree workgroups beside one net workgroup (another tenant on the GPU? "
                        "AZMI_PIPE_TREE_WGS too large?); use azmi_run_rounds", ps->tree_wgs);
}

void pipe_launch_tree(azmi_pm* pm, PipeState* ps, const PipeArrays& pa, hipStream_t st, bool prof) {
  if (ps->kind == 2) {
    static const bool diag = getenv("AZMI_PIPE_DIAG") != nullptr;
    if (diag) k_pipe_tree_generic<Connect4, 256, true><<<ps->tree_wgs, 256, 0, st>>>(PipeKernArgs{pm->ep, pm->ar, pa});
    else k_pipe_tree_generic<Connect4, 256><<<ps->tree_wgs, 256, 0, st>>>(PipeKernArgs{pm->ep, pm->ar, pa});
  }
  else if (pm->ep.gumbel_on) k_pipe_tree<Connect4, 256, false, true, true><<<ps->tree_wgs, 256, 0, st>>>(PipeKernArgs{pm->ep, pm->ar, pa});
  else if (pa.n_groups > 1u) k_pipe_tree<Connect4, 256, false, true><<<ps->tree_wgs, 256, 0, st>>>(PipeKernArgs{pm->ep, pm->ar, pa});
  else if (prof) k_pipe_tree<Connect4, 256, true><<<ps->tree_wgs, 256, 0, st>>>(PipeKernArgs{pm->ep, pm->ar, pa});
  else k_pipe_tree<Connect4, 256, false><<<ps->tree_wgs, 256, 0, st>>>(PipeKernArgs{pm->ep, pm->ar, pa});
}

}  // namespace

extern "C" int azmi_net_c4_view_get(const struct azmi_net* net, azmi_net_c4_view* out);

namespace {
// What azmi_run_pipeline(_groups) can drive: the Connect4 engine without PLAYOUT seats, at most two model groups, every group
// that evaluates with a net has a Connect4-family matrix-core net of ONE precision tier behind it.  nets[g] == NULL: group g needs
// no net (RANDOM seats: the reference's RandPlayer); no net at all: the tree side alone.
PipePlan pipe_plan(const azmi_pm* pm, azmi_net* const* nets, uint32_t num_nets) {
  PipePlan pl;
  const uint32_t G = pm->ep.num_groups;
  if (!(pm->game == AZMI_GAME_CONNECT4 && !pm->any_playout && G >= 1u && G <= 2u && pm->ep.S <= kPipeRing / 2u)) return pl;
  int x3 = -1;
  for (uint32_t g = 0; g < G; ++g) {
    if (!(pm->nn_groups >> g & 1u)) continue;                   // RANDOM seats: whatever stands in nets[g] is not asked
    azmi_net* n = (nets && g < num_nets) ? nets[g] : nullptr;
    if (!n) return pl;                                          // NN seats without a net
    if (azmi_net_c4_view_get(n, &pl.view[g]) == 0) return pl;
    if (x3 >= 0 && x3 != pl.view[g].x3) return pl;           // one precision tier per call (one net kernel)
    x3 = pl.view[g].x3;
    pl.net_groups |= 1u << g;
  }
  if (pl.net_groups == 0u) pl.tree_only = true;
  if (pl.net_groups == 2u) pl.view[0] = pl.view[1];             // (descriptor and LDS size of the one net there is)
  if (!(pl.net_groups & 2u)) pl.view[1] = pl.view[0];
  // the fast tree kernel (its plain, two-group and Gumbel builds) drives all of them; AZMI_PIPE_GENERIC=1 asks for the generic one (every
  // step through the lock-step engine's move-step function: an independent implementation kept as a cross-check, tests/test_gpu_pipeline.py)
  pl.kind = getenv("AZMI_PIPE_GENERIC") == nullptr ? 1 : 2;
  return pl;
}
}  // namespace
extern "C" int azmi_pipeline_supported(azmi_pm* pm, azmi_net* net) {
  if (!pm) return 0;
  azmi_net* nets[2] = {net, net};
  return pipe_plan(pm, nets, 2).kind != 0 ? 1 : 0;
}
extern "C" int azmi_pipeline_supported_groups(azmi_pm* pm, azmi_net* const* nets, uint32_t num_nets) {
  return pm && pipe_plan(pm, nets, num_nets).kind != 0 ? 1 : 0;
}

namespace {
// The two sides share the chip's workgroup places, and how many each needs follows the share of leaves that reach the net.  Measured
// at 4096 slots on 512 places (M simulations/s by tree workgroups; m = evaluations per simulation):
//   m = 0.17 (128 M-entry cache, answer table on, the final tree pass):  128 / 144 / 160      -> 114 / 118 / 116
//   m = 0.19 (128 M entries, first half of the round):                    96 / 112 / 128 / 144 / 160 -> 88 / 96-101 / 96-105 / 100 / 88
//   m = 0.29 (32 M entries, first half):                                  80 /  96 / 112 / 128 -> 70 / 78 / 80 / 77
//   m = 0.55 (200 k entries, first half):                                 48 /  64 /  80 / 128 -> 43 / 51 / 52 / 39
// i.e. the best count falls about linearly with m: tree workgroups = places x (0.342 - 0.367 m), between a third and 1.5 x the
// default (AZMI_PIPE_BALANCE_A / _B set other coefficients).  m is taken over the PREVIOUS call; the count moves 8 workgroups per call,
// between epochs only (a slot's home is slot % tree workgroups for one epoch at a time; between epochs a slot is whole in HBM).  The
// games do not depend on it (a slot's game is a function of its seed alone).
void pipe_balance(PipeState* ps, const PipeCtl& hc) {
  if (!ps->balance || !ps->calibrated || ps->places == 0u) return;
  const unsigned long long ds = hc.sims_total - ps->bal_sims, db = hc.tile_boards - ps->bal_boards;
  ps->bal_sims = hc.sims_total; ps->bal_boards = hc.tile_boards;
  if (ds < 4096ull || db == 0ull) return;
  double ca = 0.342, cb = 0.367;
  if (const char* e = getenv("AZMI_PIPE_BALANCE_A")) ca = atof(e);
  if (const char* e = getenv("AZMI_PIPE_BALANCE_B")) cb = atof(e);
  const double m = static_cast<double>(db) / static_cast<double>(ds);
  const double share = std::max(0.02, ca - cb * m);
  uint32_t want = static_cast<uint32_t>(static_cast<double>(ps->places) * share + 0.5);
  want = std::min(ps->tree_wgs_alloc, std::max(ps->tree_wgs_min, (want + 4u) / 8u * 8u));
  if (want + 8u <= ps->tree_wgs) ps->tree_wgs -= 8u;
  else if (want >= ps->tree_wgs + 8u) ps->tree_wgs += 8u;
  else return;
  ps->tree_wgs = std::min(ps->tree_wgs_alloc, std::max(ps->tree_wgs_min, ps->tree_wgs));
  ps->pa.n_tree_wgs = ps->tree_wgs;
  ps->net_wgs = ps->places > ps->tree_wgs ? ps->places - ps->tree_wgs : 1u;
}
}  // namespace

// An epoch's workgroup counts assume the chip to itself (placement): two engines' epochs at once would each find half of
// their workgroups without a place.  Calls from different threads therefore take turns (they are synchronous anyway).
// (one mutex per DEVICE: two engines on two different GPUs of one process do not wait for each other - VERDICT r4)
static std::mutex& pipeline_turn_of(int device) {
  static std::mutex table_mu;
  static std::map<int, std::unique_ptr<std::mutex>> table;
  std::lock_guard<std::mutex> l(table_mu);
  auto& m = table[device];
  if (!m) m.reset(new std::mutex());
  return *m;
}
static int run_pipeline_impl(azmi_pm* pm, azmi_net* const* nets, uint32_t num_nets, uint32_t epochs, uint64_t sims_per_epoch, void* stream, uint64_t* out_stats);
extern "C" int azmi_run_pipeline(azmi_pm* pm, azmi_net* net, uint32_t epochs, uint64_t sims_per_epoch, void* stream, uint64_t* out_stats) {
  azmi_net* nets[2] = {net, net};            // the same net for every model group (azmi_pm_net_forward's rule)
  return run_pipeline_impl(pm, nets, 2, epochs, sims_per_epoch, stream, out_stats);
}
extern "C" int azmi_run_pipeline_groups(azmi_pm* pm, azmi_net* const* nets, uint32_t num_nets, uint32_t epochs, uint64_t sims_per_epoch, void* stream,
                                        uint64_t* out_stats) {
  return run_pipeline_impl(pm, nets, num_nets, epochs, sims_per_epoch, stream, out_stats);
}
static int run_pipeline_impl(azmi_pm* pm, azmi_net* const* nets, uint32_t num_nets, uint32_t epochs, uint64_t sims_per_epoch, void* stream, uint64_t* out_stats) {
  if (!pm) return azmi_host_fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  std::lock_guard<std::mutex> turn_(pipeline_turn_of(pm->device));
  const PipePlan plan = pipe_plan(pm, nets, num_nets);
  if (plan.kind == 0)
    return azmi_host_fail(AZMI_ERR_STATE, "azmi_run_pipeline: the pipeline drives the Connect4 engine (no PLAYOUT seats, at most two model groups, at most %u "
                          "concurrent games) with a Connect4-family matrix-core net of one precision tier behind every model group that evaluates with a net "
                          "(or no net at all when every seat uses EvalType::RANDOM) - azmi_pipeline_supported; use azmi_run_rounds for everything else", kPipeRing / 2u);
  const azmi_net_c4_view& view = plan.view[0];
  const bool tree_only = plan.tree_only;
  if (sims_per_epoch == 0) return azmi_host_fail(AZMI_ERR_INVALID, "azmi_run_pipeline: sims_per_epoch must be > 0");
  if (pm->stopped.load(std::memory_order_relaxed)) {     // PlayManager::stop(): the workers leave their loop (play_manager.cc:272)
    if (out_stats) for (int i = 0; i < 16; ++i) out_stats[i] = 0;
    return AZMI_OK;
  }
  AZMI_HIP_TRY(hipSetDevice(pm->device));
  hipStream_t st = pm->pick(stream);
  if (!pm->pipe) {
    const int rc = pipe_create(pm, view.lds_bytes > azmi_net_dev::c4::TileBig::LDS_BYTES ? view.lds_bytes : azmi_net_dev::c4::TileBig::LDS_BYTES);
    if (rc != AZMI_OK) return rc;
  }
  PipeState* ps = pm->pipe;
  PipeArrays& pa = ps->pa;
  ps->kind = plan.kind;
  ps->np1_valid = (plan.net_groups & 2u) != 0u;
  ps->np1 = plan.view[1].np;
  pa.n_groups = pm->ep.num_groups;
  pa.net_groups = plan.net_groups;
  if (pa.n_groups > 1u && !pa.ring1) {
    const int rc = pipe_alloc(ps, pa.ring1, static_cast<size_t>(kPipeRing) * kReqGranules);
    if (rc != AZMI_OK) return rc;
  }
  if (!tree_only) { const int rc = pipe_pair_streams(ps, st); if (rc != AZMI_OK) return rc; }
  if (!tree_only) { const int rc = pipe_size_net(pm, ps, view); if (rc != AZMI_OK) return rc; }
  // the net side: the conveyor where it can run (conveyor_c4.h), else the tile kernel
  ps->cv_on = cv_eligible(pm, plan);
  if (const char* e = getenv("AZMI_PIPE_NET")) if (strcmp(e, "conveyor") == 0 && !ps->cv_on && !tree_only)
#ifdef AZMI_WITH_CONVEYOR
    return azmi_host_fail(AZMI_ERR_STATE, "AZMI_PIPE_NET=conveyor: the conveyor runs the bf16 tier behind one model group with an even number of residual blocks");
#else
    return cv_absent();
#endif
  if (ps->cv_on) {
    if (ps->balance) { ps->tree_wgs = ps->tree_wgs_default; pa.n_tree_wgs = ps->tree_wgs; }      // (the 512-place balance rule is the tile kernel's)
    pa.cap_ticks = static_cast<unsigned long long>((getenv("AZMI_PIPE_CAP_MS") ? atof(getenv("AZMI_PIPE_CAP_MS")) : 250.0) * 1e5);
    int rc = cv_setup(pm, ps, view);
    if (rc == AZMI_OK) rc = pipe_pair_svc(ps, st);
    if (rc == AZMI_OK) rc = cv_calibrate(pm, ps, st, view);
    if (rc != AZMI_OK) {
      if (getenv("AZMI_PIPE_NET")) return rc;
      fprintf(stderr, "azmi_run_pipeline: the conveyor does not run here (%s); the tile kernel takes the net side\n", azmi_last_error());
      ps->cv_on = false;
    }
  }
  if (!tree_only && !ps->cv_on) { const int rc = pipe_calibrate(pm, ps, st, view); if (rc != AZMI_OK) return rc; }
  // the insert log holds an epoch's answers: at most one per simulation, in practice a third of them
  // (the quota is checked between passes: an epoch overshoots it by what the passes under way still finish)
  const uint64_t want_log = pm->ep.cache_on ? std::min<uint64_t>(std::min<uint64_t>(sims_per_epoch, 1024ull * pm->ep.S) + 2ull * pm->ep.S + static_cast<uint64_t>(ps->tree_wgs) * 32u * (pa.max_inline + 1u), 1ull << 24) : 0ull;
  if (want_log > pa.ins_cap) {
    PipeState* p = ps;
    int rc = pipe_alloc(p, pa.ins_key, want_log);
    if (rc == AZMI_OK) rc = pipe_alloc(p, pa.ins_pi, want_log * Connect4::M);
    if (rc == AZMI_OK) rc = pipe_alloc(p, pa.ins_v, want_log * (Connect4::P + 1));
    if (rc == AZMI_OK && pa.n_groups > 1u) rc = pipe_alloc(p, pa.ins_grp, want_log);
    if (rc != AZMI_OK) return rc;
    pa.ins_cap = static_cast<uint32_t>(want_log);
  }
  if (pm->ep.cache_on && !pa.locks) {      // one insert lock per shard, group 0's cache first
    size_t total = 0;
    for (uint32_t g = 0; g < pm->ep.num_groups && g < pm->group_caches.size(); ++g) total += pm->group_caches[g].shards;
    total = std::max<size_t>(total, pm->ar.cache.shards);
    pa.lock_base1 = pm->group_caches.empty() ? 0u : pm->group_caches[0].shards;
    const int rc = pipe_alloc(ps, pa.locks, total);
    if (rc != AZMI_OK) return rc;
  }
  if (pm->ep.cache_on && !pa.l0 && !tree_only && plan.kind == 1 && getenv("AZMI_PIPE_NO_L0") == nullptr) {
    // the answer table: a power of two of 128-byte entries, an eighth of the S3-FIFO's entries, between 4 Ki and 4 Mi (512 MB: an
    // epoch of the headline brings ~200 k answers).  Measured and dropped: a table of half the S3-FIFO's entries (64 Mi = 8 GB at the
    // headline's cache) with shard hits copied into it - a table hit costs the probe one round trip instead of two, but 8 GB more of
    // randomly probed memory made every probe slower: 4968 games/s against 5667 (AZMI_PIPE_L0_LOG2 / AZMI_PIPE_L0_WB bring it back)
    uint64_t want = static_cast<uint64_t>(pm->ar.cache.shards) * kWaveCap / 8u;
    uint32_t sh = 12;
    while (sh < 22u && (1ull << sh) < want) ++sh;
    if (const char* e = getenv("AZMI_PIPE_L0_LOG2")) sh = static_cast<uint32_t>(std::min(26, std::max(4, atoi(e))));
    const int rc = pipe_alloc(ps, pa.l0, (static_cast<size_t>(1) << sh) * kResStride);
    if (rc != AZMI_OK) return rc;
    pa.l0_mask = (1u << sh) - 1u;
    pa.l0_wb = getenv("AZMI_PIPE_L0_WB") ? static_cast<uint32_t>(atoi(getenv("AZMI_PIPE_L0_WB"))) : 0u;
  }
  if (pa.l0) {
    // the table's entries are answers of the nets the LAST call ran with: another net behind a group (or the groups' nets swapped)
    // would be served the old net's (v, pi) as hits - wipe it.  (It also outlives the S3-FIFO's evictions inside one
    // engine: by design, the table is a second, direct-mapped level of the position cache.)
    const void* n0 = (plan.net_groups & 1u) ? static_cast<const void*>(plan.view[0].np.blocks) : nullptr;
    const void* n1 = (plan.net_groups & 2u) ? static_cast<const void*>(plan.view[1].np.blocks) : nullptr;
    if (n0 != ps->l0_nets[0] || n1 != ps->l0_nets[1]) {
      if (ps->l0_nets[0] || ps->l0_nets[1]) AZMI_HIP_TRY(hipMemsetAsync(pa.l0, 0, (static_cast<size_t>(pa.l0_mask) + 1u) * kResStride * sizeof(unsigned long long), st));
      ps->l0_nets[0] = n0; ps->l0_nets[1] = n1;
    }
  }
  // an epoch must end long before the wall-clock cap (a stall detector, 250 ms): with the move step inside the epoch nothing else ends it,
  // so the quota is held to 1024 simulations per slot (~50 ms at the slowest per-slot rate measured)
  pa.quota = std::min<uint64_t>(sims_per_epoch, 1024ull * pm->ep.S);
  pa.net_needed = std::max<uint32_t>(8u, (pm->ep.S + 2u) / 3u + 8u);
  // an epoch also ends when this share of the slots waits for the move step (all of them: the start of a run)
  double idle_frac = 0.125;
  if (const char* e = getenv("AZMI_PIPE_IDLE_FRAC")) idle_frac = atof(e);
  pa.idle_num = std::max<uint32_t>(1u, std::min<uint32_t>(1024u, static_cast<uint32_t>(idle_frac * 1024.0)));
  double cap_ms = 250.0;
  if (const char* e = getenv("AZMI_PIPE_CAP_MS")) cap_ms = atof(e);
  pa.cap_ticks = static_cast<unsigned long long>(cap_ms * 1e5);
  // (an epoch that runs long simply ends at a quarter of the cap; AZMI_PIPE_SOFT_MS sets another limit - the error-path test asks for
  // one beyond the cap)
  pa.test_drop = getenv("AZMI_PIPE_TEST_DROP") ? static_cast<uint32_t>(strtoul(getenv("AZMI_PIPE_TEST_DROP"), nullptr, 0)) : 0u;
  pa.soft_ticks = getenv("AZMI_PIPE_SOFT_MS") ? static_cast<unsigned long long>(atof(getenv("AZMI_PIPE_SOFT_MS")) * 1e5) : pa.cap_ticks / 4u;
  // the lock-step kernels leave the key of a round's leaf in cache_keys for the next round's insert: none of that here
  if (pm->ep.cache_on) AZMI_HIP_TRY(hipMemsetAsync(pm->ar.cache_keys, 0, sizeof(uint64_t) * pm->ep.S, st));
  int rc = azmi_host_launch_assign(pm, st, 1u);
  if (rc != AZMI_OK) return rc;
  const uint32_t settle_blocks = (pm->ep.S + 255u) / 256u;
  // tile selection: 0 = by what a window brings within its patience (<= 3 requests: the 3-board tile), 1 = 6-board tiles only,
  // 2 = 3-board tiles only.  At 4096 slots 0 and 2 measure the same (the net side has places to spare); at 16384 slots the
  // 3-board tile's capacity (19.5 M evaluations/s on 416 workgroups) is the limit, so 0 it is.
  const int net_mode = getenv("AZMI_PIPE_TILE") ? atoi(getenv("AZMI_PIPE_TILE")) : 0;
  while (ps->tev.size() < 4ull * epochs) {
    hipEvent_t ev;
    AZMI_HIP_TRY(hipEventCreate(&ev));
    ps->tev.push_back(ev);
  }
  // A slot has at most one request out and a tile takes three: more than S / 3 net workgroups (+ 8 of slack) can never all have work.
  // A small engine therefore does not launch the ~500 the chip holds: idle persistent workgroups are not free - they poll - and
  // the rare seconds-long stalls of the tree side were only ever seen with hundreds of them idle.
  ps->net_launch = std::min<uint32_t>(ps->net_wgs, std::max<uint32_t>(8u, (pm->ep.S + 2u) / 3u + 8u));
  if (getenv("AZMI_PIPE_NET_ALL")) ps->net_launch = ps->net_wgs;
  const bool prof = getenv("AZMI_PIPE_PROF") != nullptr;
  const auto host_t0 = std::chrono::steady_clock::now();       // what the host spends enqueueing the epochs (it runs ahead of the GPU)
  for (uint32_t e = 0; e < epochs; ++e) {
    AZMI_HIP_TRY(hipMemsetAsync(pa.ep, 0, sizeof(PipeEpoch), st));
    if (ps->cv_on) { rc = cv_zero_headers(ps, st); if (rc != AZMI_OK) return rc; }
    k_pipe_seed<<<settle_blocks, 256, 0, st>>>(pm->ep, pm->ar, pa);
    AZMI_HIP_TRY(hipGetLastError());
    if (!tree_only) {
      AZMI_HIP_TRY(hipEventRecord(ps->ev_go, st));
      AZMI_HIP_TRY(hipStreamWaitEvent(ps->net_stream, ps->ev_go, 0));
      if (ps->cv_on) AZMI_HIP_TRY(hipStreamWaitEvent(ps->svc_stream, ps->ev_go, 0));
    }
    if (ps->cv_on) {      // the conv workgroups take their CUs first (k_cv_wait_started), the tree workgroups pack into the rest
      AZMI_HIP_TRY(hipEventRecord(ps->tev[4 * e + 0], ps->net_stream));
      rc = cv_launch(ps, view, pa, 0u);
      if (rc != AZMI_OK) return rc;
      AZMI_HIP_TRY(hipEventRecord(ps->tev[4 * e + 1], ps->net_stream));
      AZMI_HIP_TRY(hipEventRecord(ps->ev_net, ps->net_stream));
      AZMI_HIP_TRY(hipEventRecord(ps->ev_svc, ps->svc_stream));
      k_cv_wait_started<<<1, 1, 0, st>>>(pa.ep, ps->cv_lines * ps->cv_nwg);
      AZMI_HIP_TRY(hipGetLastError());
    }
    // the tree kernel goes first: its workgroups take their places per shader engine, the net kernel is sized for what is left
    // (CU split: on its own stream, masked to the tree side's CUs; the caller's stream waits for it)
    hipStream_t ts = ps->tree_stream && !tree_only ? ps->tree_stream : st;
    if (ts != st) AZMI_HIP_TRY(hipStreamWaitEvent(ts, ps->ev_go, 0));
    AZMI_HIP_TRY(hipEventRecord(ps->tev[4 * e + 2], ts));
    pipe_launch_tree(pm, ps, pa, ts, prof);
    AZMI_HIP_TRY(hipGetLastError());
    AZMI_HIP_TRY(hipEventRecord(ps->tev[4 * e + 3], ts));
    if (ts != st) { AZMI_HIP_TRY(hipEventRecord(ps->ev_tree, ts)); AZMI_HIP_TRY(hipStreamWaitEvent(st, ps->ev_tree, 0)); }
    if (pm->ep.cache_on) {      // behind the tree kernel, beside the net side's last tiles
      k_pipe_cache_insert<<<2048, 256, 0, st>>>(pm->ar, pa, 0u);
      AZMI_HIP_TRY(hipGetLastError());
    }
    if (!tree_only && !ps->cv_on) {
      AZMI_HIP_TRY(hipEventRecord(ps->tev[4 * e + 0], ps->net_stream));
      rc = pipe_launch_net(ps, view, net_mode, ps->net_launch, ps->net_stream, pa);
      if (rc != AZMI_OK) return rc;
      AZMI_HIP_TRY(hipEventRecord(ps->tev[4 * e + 1], ps->net_stream));
      AZMI_HIP_TRY(hipEventRecord(ps->ev_net, ps->net_stream));
      AZMI_HIP_TRY(hipStreamWaitEvent(st, ps->ev_net, 0));
    }
    if (ps->cv_on) {
      AZMI_HIP_TRY(hipStreamWaitEvent(st, ps->ev_net, 0));
      AZMI_HIP_TRY(hipStreamWaitEvent(st, ps->ev_svc, 0));
    }
    k_pipe_settle<<<settle_blocks, 256, 0, st>>>(pm->ep, pm->ar, pa);
    AZMI_HIP_TRY(hipGetLastError());
    rc = azmi_host_launch_move_step(pm, st);
    if (rc != AZMI_OK) return rc;
    rc = azmi_host_launch_assign(pm, st, 1u);
    if (rc != AZMI_OK) return rc;
    if (pm->ep.cache_on) {
      k_pipe_cache_insert<<<256, 256, 0, st>>>(pm->ar, pa, 1u);
      AZMI_HIP_TRY(hipGetLastError());
    }
    if (getenv("AZMI_PIPE_DEBUG")) {      // stop at the first epoch that raised an error, with that epoch's own counters
      PipeCtl dc; PipeEpoch de;
      AZMI_HIP_TRY(hipMemcpyAsync(&dc, pa.ctl, sizeof(dc), hipMemcpyDeviceToHost, st));
      AZMI_HIP_TRY(hipMemcpyAsync(&de, pa.ep, sizeof(de), hipMemcpyDeviceToHost, st));
      AZMI_HIP_TRY(hipStreamSynchronize(st));
      if (dc.err) {
        fprintf(stderr, "pipeline debug: epoch %u of this call: err 0x%x head %u tail %u sims %llu ended %u dead %u stop %u tree_done %u/%u arrived, net arrived %u, ins %u, late %u/%u\n",
                e, dc.err, dc.head, dc.tail, de.sims, de.ended, de.dead, de.stop, de.tree_done, de.tree_arrived, de.net_arrived, de.ins_count,
                de.tree_late / 100u, de.net_late / 100u);
        epochs = e + 1;
        break;
      }
    }
  }
  const uint64_t host_enqueue_us = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - host_t0).count());
  // the run is synchronous: a pipeline error (a spin that hit its time cap, a tag that did not match) must not go unseen
  PipeCtl hc;
  PipeEpoch he;
  AZMI_HIP_TRY(hipMemcpyAsync(&hc, pa.ctl, sizeof(hc), hipMemcpyDeviceToHost, st));
  AZMI_HIP_TRY(hipMemcpyAsync(&he, pa.ep, sizeof(he), hipMemcpyDeviceToHost, st));
  AZMI_HIP_TRY(hipStreamSynchronize(st));
  if (hc.lost_total != ps->lost_seen) {
    fprintf(stderr, "azmi_run_pipeline: %u request(s) were given up by the net side and sent again (a net workgroup fell more than half a ring behind)\n", hc.lost_total - ps->lost_seen);
    ps->lost_seen = hc.lost_total;
  }
  if (!tree_only && !hc.err && !ps->cv_on) pipe_balance(ps, hc);       // (takes effect with the next call's first epoch)
  if (out_stats) {
    out_stats[0] = hc.tiles; out_stats[1] = hc.tile_boards; out_stats[2] = he.sims; out_stats[3] = he.tree_arrived;
    out_stats[4] = he.net_arrived; out_stats[5] = he.ins_count; out_stats[6] = ps->cv_on ? ps->cv_lines : ps->net_launch; out_stats[7] = ps->tree_wgs;
    out_stats[8] = he.tree_late / 100u; out_stats[9] = he.net_late / 100u;
    double net_us = 0.0, tree_us = 0.0;
    for (uint32_t e = 0; e < epochs; ++e) {
      float ms = 0.0f;
      if (!tree_only && hipEventElapsedTime(&ms, ps->tev[4 * e + 0], ps->tev[4 * e + 1]) == hipSuccess) net_us += 1e3 * ms;
      if (hipEventElapsedTime(&ms, ps->tev[4 * e + 2], ps->tev[4 * e + 3]) == hipSuccess) tree_us += 1e3 * ms;
    }
    out_stats[10] = static_cast<uint64_t>(net_us); out_stats[11] = static_cast<uint64_t>(tree_us); out_stats[12] = epochs; out_stats[13] = host_enqueue_us; out_stats[14] = static_cast<uint64_t>(ps->calib_rounds & 0xFFu) | (static_cast<uint64_t>(std::min<unsigned long long>(hc.freezes, 0xFFFFFFull)) << 8) | (static_cast<uint64_t>(hc.lost_total) << 32); out_stats[15] = hc.l0_hits;
    if (getenv("AZMI_PIPE_PROF")) {
      fprintf(stderr, "pipe prof:");
      for (int i = 0; i < 16; ++i) fprintf(stderr, " %llu", hc.prof[i]);
      fprintf(stderr, "\n");
    }
    if (getenv("AZMI_PIPE_PROF") || getenv("AZMI_PIPE_HIST")) {
      fprintf(stderr, "pipe tile hist (tiles of 1 .. 6 boards since creation):");
      for (int i = 0; i < 6; ++i) fprintf(stderr, " %llu", hc.tile_hist[i]);
      fprintf(stderr, "\n");
    }
  }
  if (hc.err) {
    fprintf(stderr, "pipeline dbg:");
    for (int i = 0; i < 18; ++i) fprintf(stderr, " %u", hc.dbg[i]);
    fprintf(stderr, "\n");
    if (plan.kind == 2) {      // the generic tree kernel: where each of its wavefronts was last (phase, 10 us units since its start)
      std::vector<PipeWg> hw(ps->tree_wgs);
      if (hipMemcpy(hw.data(), pa.wg, hw.size() * sizeof(PipeWg), hipMemcpyDeviceToHost) == hipSuccess) {
        fprintf(stderr, "pipeline generic tree kernel, workgroup: rhead rtail | phase@10us of its four wavefronts | their entry times, us after wavefront 0's:");
        for (uint32_t w = 0; w < ps->tree_wgs; ++w)
          fprintf(stderr, "  [%u: %u %u | %u@%u %u@%u %u@%u %u@%u | %d %d %d]", w, hw[w].rhead, hw[w].rtail, hw[w].pad[2] & 255u, hw[w].pad[2] >> 8, hw[w].pad[3] & 255u, hw[w].pad[3] >> 8,
                  hw[w].pad[4] & 255u, hw[w].pad[4] >> 8, hw[w].pad[5] & 255u, hw[w].pad[5] >> 8,
                  static_cast<int32_t>(hw[w].pad[7] - hw[w].pad[6]) / 100, static_cast<int32_t>(hw[w].pad[8] - hw[w].pad[6]) / 100, static_cast<int32_t>(hw[w].pad[9] - hw[w].pad[6]) / 100);
        fprintf(stderr, "\n  wavefront 0's longest interval between two looks: wall clock x10us | SQ clock >> 10 | 0 poll loop, 1 a pass | after polls | polls in all:");
        for (uint32_t w = 0; w < ps->tree_wgs; ++w) fprintf(stderr, "  [%u: %u | %u | %u | %u | %u]", w, hw[w].pad[10], hw[w].pad[11], hw[w].pad[12], hw[w].pad[13], hw[w].pad[14]);
        fprintf(stderr, "\n  longest phases of its passes (SQ clock >> 10): publication word | granules | log ticket + drain | steps + requests | tokens:");
        for (uint32_t w = 0; w < ps->tree_wgs; ++w) fprintf(stderr, "  [%u: %u | %u | %u | %u | %u]", w, hw[w].pad[18], hw[w].pad[19], hw[w].pad[20], hw[w].pad[21], hw[w].pad[22]);
        fprintf(stderr, "\n  its longest segments (SQ clock >> 10): token load | control words | between looks:");
        for (uint32_t w = 0; w < ps->tree_wgs; ++w) fprintf(stderr, "  [%u: %u | %u | %u]", w, hw[w].pad[15], hw[w].pad[16], hw[w].pad[17]);
        fprintf(stderr, "\n");
      }
    }
  }
  if (hc.err) {
    // reported once: k_pipe_settle has put every slot whose request went unanswered back into the move step's kSlotQueued form, so
    // the engine is whole - the next call (either driver) carries on from here
    AZMI_HIP_TRY(hipMemsetAsync(&pa.ctl->err, 0, sizeof(uint32_t), st));
    AZMI_HIP_TRY(hipMemsetAsync(pa.ctl->dbg, 0, sizeof(pa.ctl->dbg), st));
    AZMI_HIP_TRY(hipStreamSynchronize(st));
  }
  if (hc.err)
    return azmi_host_fail(AZMI_ERR_STATE, "pipeline error mask 0x%x (1 a spin hit the epoch's time cap, 2 a ring entry never arrived, 4 a result tag "
                          "did not match, 8 insert log full, 16 cache lock, 32 slots, 64 the net side hit the time cap); census: %u of %u tree and %u of %u net workgroups started; "
                          "last epoch: head %u tail %u sims %llu ended %u dead %u stop %u tree_done %u tiles %llu boards %llu; latest tree / net workgroup start %u / %u us; "
                          "first time-out saw: slot %u seq %u / %u; net side: stop %u tree done %u of %u tail %u window at %u after %u0 us; "
                          "wavefronts that stood still > 2 ms (not one instruction between two looks at the clock) so far: %llu, the longest for %.1f ms",
                          hc.err, he.tree_arrived, ps->tree_wgs, he.net_arrived, ps->net_launch, hc.head, hc.tail, he.sims, he.ended, he.dead, he.stop,
                          he.tree_done, hc.tiles, hc.tile_boards, he.tree_late / 100u, he.net_late / 100u,
                          hc.dbg[1], hc.dbg[2], hc.dbg[3], hc.dbg[8], hc.dbg[9], hc.dbg[10], hc.dbg[11], hc.dbg[12], hc.dbg[13],
                          static_cast<unsigned long long>(hc.freezes), static_cast<double>(hc.freeze_max) * 1e-5);
  if (hc.err && getenv("AZMI_PIPE_DEBUG")) {
    fprintf(stderr, "pipeline dbg:");
    for (int i = 0; i < 18; ++i) fprintf(stderr, " %u", hc.dbg[i]);
    fprintf(stderr, "\n");
  }
  return AZMI_OK;
}

// probes the S3-FIFO missed and the in-epoch answer table answered (azmi_pm_cache_stats moves them from misses to hits)
unsigned long long azmi_host_pipe_l0_hits(azmi_pm* pm, hipStream_t st) {
  if (!pm->pipe || !pm->pipe->pa.l0) return 0ull;
  unsigned long long v = 0;
  if (hipMemcpyAsync(&v, &pm->pipe->pa.ctl->l0_hits, sizeof(v), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 0ull;
  return v;
}

// ---- diagnostics: how many of the last epoch's answers were asked for more than once --------------------------------------------
// The insert log holds the key of every answer the tree side consumed in the last epoch of the last azmi_run_pipeline call.  A key that
// is there k times went to the net k times inside one epoch - k - 1 evaluations an insert at answer time (the reference's,
// play_manager.cc:631-640) or a table of requests in flight would have saved.  out[0] = entries, out[1] = entries beyond the first
// of their key, out[2] = of those, the ones whose twin sits within 4096 log entries (~ in flight together).
extern "C" int azmi_debug_pipe_log_dupes(azmi_pm* pm, uint64_t* out) {
  if (!pm || !out || !pm->pipe) return azmi_host_fail(AZMI_ERR_INVALID, "no pipeline on this engine");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  PipeState* ps = pm->pipe;
  AZMI_HIP_TRY(hipSetDevice(pm->device));
  AZMI_HIP_TRY(hipDeviceSynchronize());
  PipeEpoch he;
  AZMI_HIP_TRY(hipMemcpy(&he, ps->pa.ep, sizeof(he), hipMemcpyDeviceToHost));
  const uint32_t n = std::min(he.ins_count, ps->pa.ins_cap);
  std::vector<uint64_t> keys(n);
  if (n) AZMI_HIP_TRY(hipMemcpy(keys.data(), ps->pa.ins_key, sizeof(uint64_t) * n, hipMemcpyDeviceToHost));
  std::vector<std::pair<uint64_t, uint32_t>> kv(n);
  for (uint32_t i = 0; i < n; ++i) kv[i] = {keys[i], i};
  std::sort(kv.begin(), kv.end());
  uint64_t dup = 0, near = 0;
  for (uint32_t i = 1; i < n; ++i)
    if (kv[i].first == kv[i - 1].first) { ++dup; if (kv[i].second - kv[i - 1].second < 4096u) ++near; }
  out[0] = n; out[1] = dup; out[2] = near;
  return AZMI_OK;
}

// ---- diagnostics: the net side alone -----------------------------------------------------------------------------------------
namespace azmi {
__global__ void k_pipe_fill(PipeArrays pa, uint32_t n, uint32_t S, uint64_t seed) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { pa.ep->stop = 1u; }
  if (i >= n) return;
  const uint32_t pos = pa.ctl->tail + i;
  // a random legal-looking position: stones dropped column by column
  uint64_t x = mix64(seed + i);
  uint64_t b0 = 0, b1 = 0;
  uint32_t player = 0;
  for (uint32_t w = 0; w < 7; ++w) {
    const uint32_t hgt = static_cast<uint32_t>(x % 5u); x = mix64(x);
    for (uint32_t j = 0; j < hgt; ++j) {
      const uint64_t bit = 1ull << ((5u - j) * 7u + w);
      if (x & 1ull) b0 |= bit; else b1 |= bit;
      x >>= 1; player ^= 1u;
    }
  }
  unsigned long long* e = pa.ring + static_cast<size_t>(pos & (kPipeRing - 1u)) * kReqGranules;
  const unsigned long long tag = pipe_lap_tag(pos) << 48;
  e[0] = tag | b0; e[1] = tag | b1; e[2] = tag | (i % S) | (static_cast<unsigned long long>(player & 1u) << 16); e[3] = tag | (i + 1u);
}
__global__ void k_pipe_fill_done(PipeArrays pa, uint32_t n) { pa.ctl->tail += n; }
__global__ void k_pipe_head_reset(PipeArrays pa) { pa.ctl->head = pa.ctl->tail; for (uint32_t w = 0; w < pa.n_tree_wgs; ++w) pa.wg[w].rhead = pa.wg[w].rtail; }
}  // namespace azmi

// Fills the request ring with `n` synthetic positions (n <= the ring) and lets the persistent net kernel alone drain it, `reps`
// times; ms_out = average milliseconds per drain (HIP events).  mode: the tile selection of k_pipe_net (0 both, 1 six-board, 2
// three-board).  Timing only: the slots' result granules are overwritten.
extern "C" int azmi_debug_pipe_net_bench(azmi_pm* pm, azmi_net* net, uint32_t n, uint32_t reps, uint32_t net_wgs, int mode, float* ms_out) {
  if (!pm || !net || !ms_out) return azmi_host_fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  azmi_net* nets_[2] = {net, net};
  const PipePlan plan_ = pipe_plan(pm, nets_, 2);
  if (plan_.kind == 0 || plan_.tree_only) return azmi_host_fail(AZMI_ERR_STATE, "not a pipeline engine");
  const azmi_net_c4_view view = plan_.view[0];
  if (n > kPipeRing) return azmi_host_fail(AZMI_ERR_INVALID, "at most %u requests", kPipeRing);
  AZMI_HIP_TRY(hipSetDevice(pm->device));
  if (!pm->pipe) { const int rc = pipe_create(pm, azmi_net_dev::c4::TileBig::LDS_BYTES); if (rc != AZMI_OK) return rc; }
  PipeState* ps = pm->pipe;
  PipeArrays& pa = ps->pa;
  { const int rc = pipe_size_net(pm, ps, view); if (rc != AZMI_OK) return rc; }
  pa.cap_ticks = 25000000ull;
  hipStream_t st = pm->stream;
  hipEvent_t e0, e1;
  AZMI_HIP_TRY(hipEventCreate(&e0)); AZMI_HIP_TRY(hipEventCreate(&e1));
  float total = 0.0f;
  const uint32_t wgs = net_wgs ? net_wgs : ps->net_wgs;
  if (mode == 3) {      // the conveyor: net_wgs = lines (0: as many as the chip holds)
    if (view.x3 || view.nd.depth < 2 || view.nd.depth % 2) return azmi_host_fail(AZMI_ERR_STATE, "the conveyor runs the bf16 tier with an even number of residual blocks");
    { const int rc = cv_setup(pm, ps, view); if (rc != AZMI_OK) return rc; }
    if (net_wgs) ps->cv_lines = std::min(net_wgs, ps->cv_lines_alloc);
    else { hipDeviceProp_t prop; AZMI_HIP_TRY(hipGetDeviceProperties(&prop, pm->device)); ps->cv_lines = std::min(ps->cv_lines_alloc, cv_default_lines(static_cast<uint32_t>(prop.multiProcessorCount), ps->cv_nwg, 0u)); }
  }
  for (uint32_t r = 0; r < reps + 1; ++r) {
    AZMI_HIP_TRY(hipMemsetAsync(pa.ep, 0, sizeof(PipeEpoch), st));
    if (mode == 3) { const int rc = cv_zero_headers(ps, st); if (rc != AZMI_OK) return rc; }
    k_pipe_fill<<<(n + 255) / 256, 256, 0, st>>>(pa, n, pm->ep.S, 1234 + r);
    k_pipe_fill_done<<<1, 1, 0, st>>>(pa, n);
    AZMI_HIP_TRY(hipEventRecord(e0, st));
    if (mode == 3) {
      AZMI_HIP_TRY(hipEventRecord(ps->ev_go, st));
      AZMI_HIP_TRY(hipStreamWaitEvent(ps->net_stream, ps->ev_go, 0));
      AZMI_HIP_TRY(hipStreamWaitEvent(ps->svc_stream, ps->ev_go, 0));
      { const int rc = cv_launch(ps, view, pa, 0u); if (rc != AZMI_OK) return rc; }
      AZMI_HIP_TRY(hipEventRecord(ps->ev_net, ps->net_stream));
      AZMI_HIP_TRY(hipEventRecord(ps->ev_svc, ps->svc_stream));
      AZMI_HIP_TRY(hipStreamWaitEvent(st, ps->ev_net, 0));
      AZMI_HIP_TRY(hipStreamWaitEvent(st, ps->ev_svc, 0));
    } else { const int rc = pipe_launch_net(ps, view, mode, wgs, st, pa); if (rc != AZMI_OK) return rc; }
    AZMI_HIP_TRY(hipEventRecord(e1, st));
    k_pipe_head_reset<<<1, 1, 0, st>>>(pa);                   // (head = tail for the next drain; the READY tokens of the synthetic answers are dropped)
    AZMI_HIP_TRY(hipStreamSynchronize(st));
    float ms = 0.0f;
    AZMI_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0) total += ms;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (mode == 3 && getenv("AZMI_CV_STAT")) {
    unsigned long long h[32];
    AZMI_HIP_TRY(hipMemcpy(h, ps->cv_stat, sizeof(h), hipMemcpyDeviceToHost));
    AZMI_HIP_TRY(hipMemset(ps->cv_stat, 0, sizeof(h)));
    fprintf(stderr, "conveyor stat (%u lines, %u drains): conv n-tiles %llu (drained %llu), stem groups %llu boards %llu, head groups %llu | per wave, us: "
            "conv wait-in %.0f %.0f %.0f %.0f wait-out %.0f %.0f %.0f %.0f life %.0f | stem claim %.0f room %.0f life %.0f | head wait %.0f life %.0f\n",
            ps->cv_lines, reps + 1, h[0], h[1], h[2], h[3], h[4],
            h[5] * 0.01 / (ps->cv_lines * ps->cv_nwg), h[6] * 0.01 / (ps->cv_lines * ps->cv_nwg), h[7] * 0.01 / (ps->cv_lines * ps->cv_nwg), h[8] * 0.01 / (ps->cv_lines * ps->cv_nwg),
            h[9] * 0.01 / (ps->cv_lines * ps->cv_nwg), h[10] * 0.01 / (ps->cv_lines * ps->cv_nwg), h[11] * 0.01 / (ps->cv_lines * ps->cv_nwg), h[12] * 0.01 / (ps->cv_lines * ps->cv_nwg),
            h[16] * 0.01 / (ps->cv_lines * ps->cv_nwg), h[13] * 0.01 / ps->cv_lines, h[14] * 0.01 / ps->cv_lines, h[20] * 0.01 / ps->cv_lines,
            h[15] * 0.01 / (ps->cv_lines * ps->cv_heads), h[21] * 0.01 / (ps->cv_lines * ps->cv_heads));
    if (h[26]) fprintf(stderr, "  back-to-back n-tiles, ns each by role: %.0f %.0f %.0f %.0f (of %llu %llu %llu %llu)\n", 10.0 * h[22] / std::max(1ull, h[26]), 10.0 * h[23] / std::max(1ull, h[27]),
                       10.0 * h[24] / std::max(1ull, h[28]), 10.0 * h[25] / std::max(1ull, h[29]), h[26], h[27], h[28], h[29]);
  }
  // the synthetic answers carry sequence numbers a real request of the same slot could draw later: wipe the granules
  AZMI_HIP_TRY(hipMemsetAsync(pa.res, 0, sizeof(unsigned long long) * static_cast<size_t>(pm->ep.S) * kResStride, st));
  AZMI_HIP_TRY(hipStreamSynchronize(st));
  *ms_out = total / static_cast<float>(reps);
  PipeCtl hc;
  AZMI_HIP_TRY(hipMemcpy(&hc, pa.ctl, sizeof(hc), hipMemcpyDeviceToHost));
  if (hc.err) {
    (void)hipMemset(&pa.ctl->err, 0, sizeof(uint32_t));
    return azmi_host_fail(AZMI_ERR_STATE, "pipeline error mask 0x%x in the net drain", hc.err);
  }
  return AZMI_OK;
}

// The net side's ANSWERS for `n` synthetic positions (n <= the engine's slots; position i -> slot i): the request ring is filled from
// `seed` (k_pipe_fill), the net side drains it once - conveyor != 0: the conveyor with `lines` lines (0: the default), else the tile
// kernel (tile selection 0) -, out[i * 10 + k] = entry k of slot i's result granules (pi[0..7) then v[0..3)), out_seq[i] = the sequence
// number they carry (i + 1 when the slot was answered).  tests/test_gpu_conveyor.py compares the two paths bit for bit.
extern "C" int azmi_debug_pipe_net_answers(azmi_pm* pm, azmi_net* net, uint32_t n, uint64_t seed, int conveyor, uint32_t lines, float* out, uint32_t* out_seq) {
  if (!pm || !net || !out || !out_seq) return azmi_host_fail(AZMI_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lock_(pm->mu);
  azmi_net* nets_[2] = {net, net};
  const PipePlan plan_ = pipe_plan(pm, nets_, 2);
  if (plan_.kind == 0 || plan_.tree_only) return azmi_host_fail(AZMI_ERR_STATE, "not a pipeline engine");
  const azmi_net_c4_view view = plan_.view[0];
  if (n > pm->ep.S || n > kPipeRing) return azmi_host_fail(AZMI_ERR_INVALID, "at most %u positions", pm->ep.S);
  AZMI_HIP_TRY(hipSetDevice(pm->device));
  if (!pm->pipe) { const int rc = pipe_create(pm, azmi_net_dev::c4::TileBig::LDS_BYTES); if (rc != AZMI_OK) return rc; }
  PipeState* ps = pm->pipe;
  PipeArrays pa = ps->pa;
  { const int rc = pipe_size_net(pm, ps, view); if (rc != AZMI_OK) return rc; }
  pa.cap_ticks = 25000000ull;
  pa.l0 = nullptr;
  hipStream_t st = pm->stream;
  { const int rc = pipe_pair_streams(ps, st); if (rc != AZMI_OK) return rc; }
  if (conveyor) {
    if (view.x3 || view.nd.depth < 2 || view.nd.depth % 2) return azmi_host_fail(AZMI_ERR_STATE, "the conveyor runs the bf16 tier with an even number of residual blocks");
    { const int rc = cv_setup(pm, ps, view); if (rc != AZMI_OK) return rc; }
    if (lines) ps->cv_lines = std::min(lines, ps->cv_lines_alloc);
  }
  AZMI_HIP_TRY(hipMemsetAsync(pa.res, 0, sizeof(unsigned long long) * static_cast<size_t>(pm->ep.S) * kResStride, st));
  AZMI_HIP_TRY(hipMemsetAsync(pa.ep, 0, sizeof(PipeEpoch), st));
  if (conveyor) { const int rc = cv_zero_headers(ps, st); if (rc != AZMI_OK) return rc; }
  k_pipe_fill<<<(n + 255) / 256, 256, 0, st>>>(pa, n, pm->ep.S, seed);
  k_pipe_fill_done<<<1, 1, 0, st>>>(pa, n);
  if (conveyor) {
    AZMI_HIP_TRY(hipEventRecord(ps->ev_go, st));
    AZMI_HIP_TRY(hipStreamWaitEvent(ps->net_stream, ps->ev_go, 0));
    AZMI_HIP_TRY(hipStreamWaitEvent(ps->svc_stream, ps->ev_go, 0));
    { const int rc = cv_launch(ps, view, pa, 0u); if (rc != AZMI_OK) return rc; }
    AZMI_HIP_TRY(hipEventRecord(ps->ev_net, ps->net_stream));
    AZMI_HIP_TRY(hipEventRecord(ps->ev_svc, ps->svc_stream));
    AZMI_HIP_TRY(hipStreamWaitEvent(st, ps->ev_net, 0));
    AZMI_HIP_TRY(hipStreamWaitEvent(st, ps->ev_svc, 0));
  } else { const int rc = pipe_launch_net(ps, view, 0, ps->net_wgs, st, pa); if (rc != AZMI_OK) return rc; }
  k_pipe_head_reset<<<1, 1, 0, st>>>(pa);
  std::vector<unsigned long long> hres(static_cast<size_t>(n) * kResStride);
  AZMI_HIP_TRY(hipMemcpyAsync(hres.data(), pa.res, hres.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  AZMI_HIP_TRY(hipMemsetAsync(pa.res, 0, sizeof(unsigned long long) * static_cast<size_t>(pm->ep.S) * kResStride, st));
  PipeCtl hc;
  AZMI_HIP_TRY(hipMemcpyAsync(&hc, pa.ctl, sizeof(hc), hipMemcpyDeviceToHost, st));
  AZMI_HIP_TRY(hipStreamSynchronize(st));
  for (uint32_t i = 0; i < n; ++i) {
    out_seq[i] = static_cast<uint32_t>(hres[static_cast<size_t>(i) * kResStride] >> 32);
    for (uint32_t k = 0; k < 10; ++k) { const uint32_t bits = static_cast<uint32_t>(hres[static_cast<size_t>(i) * kResStride + k]); memcpy(out + static_cast<size_t>(i) * 10 + k, &bits, 4); }
  }
  if (hc.err) {
    AZMI_HIP_TRY(hipMemset(&pa.ctl->err, 0, sizeof(uint32_t)));
    return azmi_host_fail(AZMI_ERR_STATE, "pipeline error mask 0x%x in the net drain (dbg %u %u %u %u %u %u)", hc.err, hc.dbg[8], hc.dbg[9], hc.dbg[10], hc.dbg[11], hc.dbg[12], hc.dbg[13]);
  }
  return AZMI_OK;
}
