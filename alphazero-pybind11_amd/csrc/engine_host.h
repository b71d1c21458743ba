// Host-side state of one PlayManager engine, shared by the translation units of libazmi.so that launch kernels on it
// (engine.hip: the round loop and the C ABI; pipeline.hip: the asynchronous tree / net pipeline).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/azmi.h"
#include "engine_types.h"

// sets the thread's azmi_last_error() text and returns `code` (defined in engine.hip)
int azmi_host_fail(int code, const char* fmt, ...);
#define AZMI_HIP_TRY(expr)                                                                \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return azmi_host_fail(e_ == hipErrorOutOfMemory ? AZMI_ERR_OOM : AZMI_ERR_NO_DEVICE, "%s: %s", #expr, \
                            hipGetErrorString(e_));                                       \
  } while (0)

namespace azmi {
struct GameInfo {
  uint32_t P, M, C, H, W, maxk, max_turns, state_words;
  uint32_t cap_branch;  // children per expansion the tree arena is sized for (== maxk when that is affordable)
};

// the asynchronous pipeline's own HBM (pipeline.hip), created on first use
struct PipeState;
void pipe_state_free(PipeState* p);
}  // namespace azmi

struct azmi_pm {
  int game = 0;
  int device = 0;
  azmi::GameInfo gi{};
  azmi_play_params params{};
  azmi::EngineParams ep{};
  azmi::EngineArrays ar{};
  std::vector<void*> allocs;
  size_t bytes = 0;
  hipStream_t stream = nullptr;  // engine-owned stream (AZMI_STREAM_ENGINE)
  hipStream_t last = nullptr;    // stream of the most recent round: result queries order themselves behind it
  hipStream_t pick(void* s) { last = (s == AZMI_STREAM_ENGINE) ? stream : static_cast<hipStream_t>(s); return last; }
  unsigned long long hist_read = 0;
  uint32_t cache_shards = 0;
  std::vector<azmi::CacheView> group_caches;   // host copies of the per-model-group cache views
  std::vector<uint8_t> group_cache_counted;  // 0: stand-in for a `None` entry of an external cache list (not in the statistics)
  bool all_random = false;               // no seat needs a net (EvalType::RANDOM / PLAYOUT everywhere)
  bool any_playout = false;              // some seat uses EvalType::PLAYOUT
  uint32_t nn_groups = 0;                // bit g: a seat of model group g evaluates with the net
  bool big_split = false;                // wide games, no PLAYOUT seats: k_round_big_sim (StarGambit: k_round_big_sim1) + k_round_big_move instead of the one-kernel round (engine_kernels_big.h)
  bool split_rounds = false;             // Connect4, plain PUCT seats: k_sim + move step instead of the one k_round (engine_kernels.h)
  bool max_inline_explicit = false;      // azmi_pm_options.max_inline was given (the pipeline then keeps it instead of its own default)
  std::vector<std::deque<uint32_t>> pending_g;   // host-buffer path: pending leaves per model group
  // hipGraph of kGraphRounds x (round kernels + net) for azmi_run_rounds: one graph launch instead of
  // ~5 kernel launches per round keeps the host ahead of the GPU
  hipGraphExec_t graph_exec = nullptr;
  hipStream_t graph_stream = nullptr;
  azmi_net* graph_net = nullptr;
  // host-buffer compatibility path
  std::deque<uint32_t> pending;        // slots whose leaf waits for the net
  std::vector<float> host_v, host_pi;  // mirrors of the slot-indexed rows
  uint32_t outstanding = 0;
  std::atomic<bool> stopped{false};
  // The reference's callers reach one PlayManager from several Python threads (mcts_workers x play(), batcher threads with
  // build_batch / update_inferences, the main thread with counters): every entry point that touches the engine's host state
  // takes this lock, so such callers are serialised instead of racing.  (Recursive: entry points call each other.)
  azmi::PipeState* pipe = nullptr;   // the asynchronous tree / net pipeline (pipeline.hip), created by its first run
  std::recursive_mutex mu;    // PlayManager::stop(), play_manager.h:177 (may be set from another thread)

  template <class T>
  int alloc(T*& p, size_t n, bool zero) {
    void* q = nullptr;
    const size_t sz = std::max<size_t>(n, 1) * sizeof(T);
    AZMI_HIP_TRY(hipMalloc(&q, sz));
    allocs.push_back(q);
    bytes += sz;
    if (zero) {     // (hipMemset is asynchronous to the host, on the null stream: nothing on another stream may run ahead of it)
      AZMI_HIP_TRY(hipMemset(q, 0, sz));
      AZMI_HIP_TRY(hipStreamSynchronize(nullptr));
    }
    p = static_cast<T*>(q);
    return AZMI_OK;
  }
  ~azmi_pm() {
    if (pipe) azmi::pipe_state_free(pipe);
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    for (void* q : allocs) (void)hipFree(q);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

