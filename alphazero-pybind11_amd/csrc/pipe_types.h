// Asynchronous tree / net pipeline of the Connect4 engine (pipeline.hip): the structures its kernels share.
//
// The reference's worker loop has no global barrier: every game advances on its own through the queues between the MCTS
// workers and the batcher threads (play_manager.cc:258-600, concurrent_queue.h:130-217).  The lock-step round of
// engine.hip (cache insert -> k_sim -> k_net_move, one after the other, a round as long as its slowest slot) gave that up.
// Here it is back, on the device: for one EPOCH
//   * tree wavefronts (persistent, k_pipe_tree) take the slot tokens that have arrived in their WORKGROUP's READY ring (a slot has a
//     home workgroup for the epoch, not a home wavefront), run simulation after simulation for them - and the rare, register-hungry
//     move step (round_slot<kMover>: the simulation that completes a search and the move behind it, a game start) as a non-inlined
//     function when a token carries the move bit - and hand every leaf that needs the net to
//   * net workgroups (persistent, k_pipe_net) that pull 3- or 6-board tiles off a request ring as soon as leaves exist
//     (continuous batching), write the (v, pi) answers as result granules, drop them into the in-epoch answer table and put the
//     answered slots back into their home workgroups' READY rings.
// The sides talk through HBM only:
//   * request ring: a tree group takes a ticket (tail) and writes its leaf as kReqGranules 8-byte granules
//     {tag16 | payload48}: stones of player 0 | stones of player 1 | slot, player | sequence number; a net workgroup draws a
//     window [head, head + n) with one fetch-add and re-reads its granules until every tag is the tag of the ring lap;
//   * result granules: per slot kResStride 8-byte granules {seq32 | float bits}, pi[0..M) then v[0..P]; the slot's next tree
//     group polls them until every tag is the sequence number of its request;
//   * READY rings, ONE PER TREE WORKGROUP: one 8-byte token {tag16 | seq32 | move1 | slot15} per slot that can take its next step.
// Every granule and token is ONE naturally aligned 8-byte agent-scope atomic store / load (write-through, L1-bypassing): the
// data is its own flag, no ordering between granules is assumed (cdna_hip_programming.md Guideline 16, form R2).
// Round 4: a slot has a HOME workgroup (slot % tree workgroups) for the whole epoch.  Its state, its path image and its trees are
// plain memory that only the four wavefronts of that workgroup - one CU, one vector L1 - ever touch inside an epoch, so a pass
// needs no agent-scope acquire / release any more (round 3 paid ~10 us of buffer_wbl2 / buffer_inv per 75 us pass for slots that
// wandered between CUs, and more with every tree workgroup added): a wavefront's stores are drained (s_waitcnt vmcnt(0)) before
// its tokens go out, which is all workgroup scope asks for.  The move step (round_slot<kMover>) is run by the home workgroup too
// (a token with the `move` bit): no mover wavefronts, no MOVE ring, no hand-over at all.
// What still needs a kernel boundary: the position-cache inserts of the epoch's answers (an insert log), game restarts
// (k_assign).  Every spin is bounded by a wall-clock cap; errors are a word (PipeErr) the host reads after every call.
#pragma once
#include <stdint.h>

namespace azmi {

constexpr uint32_t kReqGranules = 4;     // ring entry = 32 bytes
constexpr uint32_t kResStride = 16;      // result granules per slot: M + P + 1 = 10 used, padded to one 128-byte line
constexpr uint32_t kResV = 7;            // first value granule (Connect4: pi in [0, 7), v in [7, 10))
constexpr uint32_t kPipeRing = 32768;    // ring entries (a power of two, twice the most slots of an engine: one request / one READY token per slot at most)
enum PipeErr : uint32_t { kPipeErrTimeout = 1, kPipeErrRing = 2, kPipeErrTag = 4, kPipeErrLog = 8, kPipeErrLock = 16, kPipeErrSlots = 32, kPipeErrNetTimeout = 64 };

__host__ __device__ inline uint64_t pipe_lap_tag(uint32_t pos) { return static_cast<uint64_t>(((pos / kPipeRing) & 0x7FFFu) + 1u); }
// the per-workgroup READY rings have 1 << shift entries
// in-epoch answer table: entry of a key, tag of granule k of that key's entry (splitmix finaliser; k + 1 keeps granule 0's tag apart from the index)
__host__ __device__ inline uint64_t pipe_mix64(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x; }
constexpr unsigned long long kPipeGroupSalt = 0xA5A5A5A55A5A5A5AULL;      // model group 1's lines of the one answer table: its keys under this salt
__host__ __device__ inline uint32_t pipe_l0_entry(uint64_t key, uint32_t mask) { return static_cast<uint32_t>(pipe_mix64(key ^ 0x5851F42D4C957F2DULL)) & mask; }
__host__ __device__ inline uint32_t pipe_l0_tag(uint64_t key, uint32_t k) { return static_cast<uint32_t>(pipe_mix64(key + 0x9E3779B97F4A7C15ULL * (k + 1u)) >> 32) | 1u; }
__host__ __device__ inline uint64_t pipe_lap_tag_r(uint32_t pos, uint32_t shift) { return static_cast<uint64_t>(((pos >> shift) & 0x7FFFu) + 1u); }
constexpr unsigned long long kTokMove = 0x8000ull;     // token: the slot's next step is the move step's (a game start, the simulation that completes a search)
constexpr unsigned long long kTokSlotMask = 0x7FFFull;

struct PipeWg {           // one 128-byte line per tree workgroup: its READY ring's positions (free-running)
  uint32_t rhead;         // positions drawn by the workgroup's wavefronts
  uint32_t rtail;         // tickets handed out (seed kernel, net workgroups, the workgroup's own wavefronts)
  uint32_t pad[30];
};
static_assert(sizeof(PipeWg) == 128, "one line");

struct PipeCtl {          // zeroed when the pipeline is created; lives across epochs.  Every hot word on a 128-byte line of its own
  uint32_t head; uint32_t pad0[31];             // ring entries claimed by net workgroups (free-running: position = value % kPipeRing)
  uint32_t tail; uint32_t pad1[31];             // ring tickets handed out to tree wavefronts (free-running)
  uint32_t head1; uint32_t pad4[31];            // the same two words for the request ring of model group 1 (the generic tree kernel
  uint32_t tail1; uint32_t pad5[31];            // routes a leaf to the ring of its seat's group: one net per group, play_past)
  uint32_t err;           // PipeErr bits, sticky: every pipeline kernel leaves at once when it is set.  Polled by every waiting
  uint32_t pad6[31];      // workgroup: a line of its own (round 5; it shared the line of the tile counters below)
  // freezes credited by pipe_freeze_credit (count, longest gap in ticks): words of their own (they shared prof[14] / [15]
  // with the PROF build's level / move counters, so a profiled run reported garbage as freezes)
  unsigned long long freezes, freeze_max;
  unsigned long long tile_hist[6];  // net tiles by the boards they carried (1 .. 6): printed by AZMI_PIPE_PROF
  uint32_t pad7[16];
  uint32_t pad2a, pad2;
  unsigned long long tiles;         // net tiles run
  unsigned long long tile_boards;   // boards in them
  unsigned long long epochs;
  unsigned long long sims_total;    // simulations of all epochs so far (k_pipe_settle adds an epoch's count)
  unsigned long long l0_hits;       // probes the S3-FIFO missed and the in-epoch answer table answered
  uint32_t lost_total, pad3;        // requests given up on and sent again (see PipeEpoch::lost), all epochs
  uint32_t dbg[18];       // diagnostics of the first time-out: slot, sequence number, group state, the tags seen
  // tree-side time accounting (100 MHz ticks / counts, summed over wavefronts): [0] in simulation passes, [1] polling with no
  // group ready, [2] passes, [3] groups active in them, [4] polls, [5] in the request step, [6] wavefront lifetimes, [7] net: ticks
  // waiting for requests, [8] net: ticks in tiles
  unsigned long long prof[16];
};
static_assert(sizeof(PipeCtl) == 1024, "eight lines");

struct PipeEpoch {        // an allocation of its own, zeroed before every epoch (one memset); three 128-byte lines by who touches them
  unsigned long long sims; uint32_t pad0[30];   // simulations finished in this epoch (tree wavefronts add, their idle polls read)
  // ---- line 1: words the net workgroups POLL while they wait (written a few times per epoch)
  unsigned long long t0;  // wall clock of the epoch's first workgroup
  uint32_t stop;          // the epoch is over: quota reached, enough slots wait for the move step, time cap passed, or an error
  uint32_t tree_done;     // tree workgroups that have stored their slots and left
  uint32_t tree_arrived, net_arrived;   // census: workgroups that started
  uint32_t tree_late, net_late;   // census: the latest start of a tree / net workgroup, in 100 MHz ticks after the first workgroup of the epoch
  uint32_t tree_late_n, net_late_n;   // calibration launches: workgroups that only started when the others had left
  uint32_t svc_arrived, svc_late_n;   // conveyor: its service workgroups that started / that only started when the others had left (calibration)
  uint32_t pad1[20];
  // ---- line 2: words the tree wavefronts WRITE all the time (round 5: off the polled line - an atomic on a line that hundreds of idle
  // workgroups read every microsecond waits behind them)
  uint32_t ins_count;     // entries of the insert log
  uint32_t ended;         // games that ended in this epoch: their slots idle until the boundary's k_assign restarts or retires them
  uint32_t dead;          // slots without a game when the epoch began (retired) or lost to an engine error
  uint32_t ins_done;      // insert-log entries already applied (the first insert launch runs while the net side drains)
  uint32_t moved;         // move steps run by the mover wavefronts in this epoch
  uint32_t lost;          // ring positions a net workgroup gave up on in this epoch (their requests were overwritten a lap later before
                          // it could look: the workgroup had been switched out): k_pipe_settle re-queues those slots, no error
  uint32_t pad2[26];
};
static_assert(sizeof(PipeEpoch) == 384, "memset block: three lines");

struct PipeArrays {
  PipeCtl* ctl;
  PipeEpoch* ep;
  unsigned long long* ring;   // [kPipeRing][kReqGranules]
  unsigned long long* ring1;  // the request ring of model group 1 (NULL: one group)
  uint32_t n_groups;          // model groups that send requests (1 or 2)
  uint32_t net_groups;        // bit g: group g has a net behind its ring in this call
  uint32_t lock_base1;        // first insert lock of group 1's cache in `locks`
  uint8_t* ins_grp;           // [ins_cap] model group of an insert-log entry (NULL: all group 0)
  unsigned long long* rring;  // [n_tree_wgs][1 << rshift] READY rings: {tag16 | seq32 | move | slot}
  PipeWg* wg;                 // [n_tree_wgs]
  uint32_t rshift;            // log2 of a READY ring's entries (>= twice the slots of a workgroup)
  uint32_t big_at;            // net side, tile selection 0: 0 = always a 6-request window; 1 = 6 after a window that was complete at first look, else 3; >= 2: that, and at least this many requests waiting in the ring
  uint32_t census_hold;       // != 0: a calibration launch - every workgroup holds its place this many ticks and leaves (pipe_calibrate)
  uint32_t take_wait;         // ticks a wavefront that found fewer than kTreeWindow tokens waits for more before it starts its pass
  uint32_t max_inline;        // simulations a group may finish in one pass without the net (cache hits, terminal leaves) before its slot re-queues
  uint32_t min_active;        // ... and a pass ends early once fewer than this many of its eight groups are still running (the others idle meanwhile)
  unsigned long long* res;    // [S][kResStride]
  // in-epoch answer table (round 4): a direct-mapped table of self-validating granules {tag32(key, k) | float bits}, entry =
  // mix64(key) & l0_mask, written by the NET workgroup the moment it has an answer (the reference inserts at update_inferences time,
  // play_manager.cc:631-640; the S3-FIFO insert proper waits for the epoch boundary) and probed by the tree side beside the S3-FIFO
  // shard.  No lock, no ordering: a reader accepts an entry only when all ten granules carry the tag of ITS key.  NULL = off.
  unsigned long long* l0;     // [l0_mask + 1][kResStride]
  uint32_t l0_mask;
  uint32_t test_drop;         // test hook: the request at this ring position goes out with a foreign lap tag (0 = off)
  uint32_t l0_wb;             // 1: a shard hit the table did not have is copied into it
  // insert log: (key, pi, v) of every answer consumed in the epoch; applied to the position cache between epochs
  uint64_t* ins_key;          // [ins_cap]
  float* ins_pi;              // [ins_cap][M]
  float* ins_v;               // [ins_cap][P + 1]
  uint32_t ins_cap;
  uint32_t* locks;            // [cache shards] insert locks of the position cache (0 = free)
  uint32_t n_tree_wgs;
  uint32_t net_needed;            // net workgroups the slots can keep busy (S / 3 + 8): one with a higher index that sits idle beside a tree side that has arrived and makes no progress for 1.25 caps leaves (k_pipe_net)
  unsigned long long quota;       // simulations per epoch
  uint32_t idle_num;              // the epoch also ends when `ended` reaches idle_num / 1024 of the slots that have a game
  unsigned long long cap_ticks;   // hard time cap of an epoch in 100 MHz ticks (a stall detector: an error)
  unsigned long long soft_ticks;  // an epoch that has run this long ends like one that reached its quota
};

}  // namespace azmi
