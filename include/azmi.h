/* azmi.h — C ABI of the MI355X-native self-play engine (libazmi.so).
 *
 * Drop-in boundary for the reference's PlayManager/MCTS hot path.  Every entry
 * point below replaces one member of the reference's pybind11 surface
 * (/root/reference/src/py_wrapper.cc) or of the C++ class behind it; the
 * reference file:line each one stands in for is cited next to it.  Plain
 * pointers and sizes only: no torch, pybind11 or HIP types cross this line
 * (`stream` is a hipStream_t passed as void*; `dev_*` pointers are HIP device
 * addresses, e.g. torch.Tensor.data_ptr()).
 *
 * All functions return 0 on success or a negative azmi_status; the message of
 * the last error on the calling thread is available from azmi_last_error().
 * The library has NO CPU fallback: without a HIP device azmi_pm_create fails
 * with AZMI_ERR_NO_DEVICE.
 */
#ifndef AZMI_H_
#define AZMI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZMI_ABI_VERSION 1

typedef enum azmi_status {
  AZMI_OK = 0,
  AZMI_ERR_INVALID = -1,   /* bad argument (reference: std::runtime_error -> RuntimeError) */
  AZMI_ERR_NO_DEVICE = -2, /* no HIP device / HIP runtime error */
  AZMI_ERR_OOM = -3,
  AZMI_ERR_OVERFLOW = -4,  /* a device-side arena / history ring ran out of room */
  AZMI_ERR_STATE = -5,
  AZMI_ERR_RANGE = -6      /* index out of range (reference: std::out_of_range -> IndexError) */
} azmi_status;

/* game ids (reference GAME_REGISTRY, config.py:17-35) */
typedef enum azmi_game { AZMI_GAME_CONNECT4 = 0, AZMI_GAME_TAWLBWRDD = 1, AZMI_GAME_BRANDUBH = 2, AZMI_GAME_OPENTAFL = 3,
                         /* StarGambitUnifiedGS (star_gambit_gs.h:788-887): four variants on the 13 x 13 canvas, 36 planes, 1709 moves,
                          * relative values, several actions per turn */
                         AZMI_GAME_STARGAMBIT = 4 } azmi_game;

/* EvalType, play_manager.h:20 */
typedef enum azmi_eval_type { AZMI_EVAL_NN = 0, AZMI_EVAL_RANDOM = 1, AZMI_EVAL_PLAYOUT = 2 } azmi_eval_type;

#define AZMI_MAX_PLAYERS 4
#define AZMI_MAX_PERMS 8    /* seat permutations (2 players: 2; the reference cycles or enumerates them, game_runner.py:2208-2231) */
#define AZMI_MAX_GROUPS 4   /* model groups = distinct networks in one PlayManager */

/* POD mirror of struct PlayParams (play_manager.h:60-154): same names, same
 * meaning, same defaults (azmi_play_params_default).  Vectors become fixed
 * arrays + a count. */
typedef struct azmi_play_params {
  uint32_t games_to_play;
  uint32_t concurrent_games;
  uint32_t max_batch_size;
  uint32_t max_cache_size;
  uint32_t cache_shards;
  uint32_t num_mcts_visits;                 /* must equal num_players (play_manager.cc:20-22) */
  uint32_t mcts_visits[AZMI_MAX_PLAYERS];
  float cpuct;
  float start_temp;
  float final_temp;
  float temp_decay_half_life;
  int32_t history_enabled;
  int32_t self_play;
  int32_t tree_reuse;
  float epsilon;
  float mcts_root_temp;
  int32_t playout_cap_randomization;
  uint32_t playout_cap_depth;
  float playout_cap_percent;
  float fpu_reduction;
  int32_t root_fpu_zero;
  int32_t shaped_dirichlet;
  int32_t policy_target_pruning;
  float resign_percent;
  float resign_playthrough_percent;
  uint32_t num_eval_type;                   /* 0 = all NN */
  int32_t eval_type[AZMI_MAX_PLAYERS];
  /* Gumbel AlphaZero (play_manager.h:103-116; mcts.cc:24-401): full searches use Gumbel-Top-k +
   * sequential halving at the root, the improved policy as the training target and the
   * deterministic final action; capped searches stay PUCT unless fast_search_uses_gumbel. */
  int32_t gumbel_enabled;
  uint32_t gumbel_m;                        /* default 16, at most 64 here */
  float gumbel_c_visit;                     /* default 50 */
  float gumbel_c_scale;                     /* default 1 */
  int32_t gumbel_full;
  int32_t fast_search_uses_gumbel;
  /* model groups and seat permutations (play_manager.h:118-154; normalised as play_manager.cc:24-113 does):
   * model_groups[player] = network index (0 entries = identity, i.e. one group per player);
   * seat_perms[perm][seat] = model group sitting in that seat for games with perm_index == perm
   * (0 perms = one permutation equal to model_groups); game i starts with perm i % num_seat_perms.
   * mcts_visits / eval_type are per PLAYER and become per-group (last player of a group wins), then per seat. */
  uint32_t num_model_groups_given;
  uint8_t model_groups[AZMI_MAX_PLAYERS];
  uint32_t num_seat_perms;
  uint8_t seat_perms[AZMI_MAX_PERMS][AZMI_MAX_PLAYERS];
  /* per-permutation x per-seat overrides [num_seat_perms][num_players]; has_* = 0 -> filled from the globals */
  int32_t has_seat_visits, has_seat_cap_visits, has_seat_epsilon, has_seat_mcts_root_temp, has_seat_root_fpu_zero;
  uint32_t seat_visits[AZMI_MAX_PERMS][AZMI_MAX_PLAYERS];
  uint32_t seat_cap_visits[AZMI_MAX_PERMS][AZMI_MAX_PLAYERS];
  float seat_epsilon[AZMI_MAX_PERMS][AZMI_MAX_PLAYERS];
  float seat_mcts_root_temp[AZMI_MAX_PERMS][AZMI_MAX_PLAYERS];
  uint8_t seat_root_fpu_zero[AZMI_MAX_PERMS][AZMI_MAX_PLAYERS];
  /* per-seat Gumbel and resign overrides, play_manager.h:126-153 / play_manager.cc:116-176 (same has_* convention; defaults:
   * the global gumbel_* fields, use_improved_policy 0, resign threshold -2 = off, consecutive 1) */
  int32_t has_seat_gumbel_enabled, has_seat_gumbel_m, has_seat_gumbel_c_visit, has_seat_gumbel_c_scale, has_seat_gumbel_full,
          has_seat_gumbel_use_improved_policy, has_seat_resign_threshold, has_seat_resign_consecutive;
  uint8_t seat_gumbel_enabled[AZMI_MAX_PERMS][AZMI_MAX_PLAYERS];
  uint8_t seat_gumbel_full[AZMI_MAX_PERMS][AZMI_MAX_PLAYERS];
  uint8_t seat_gumbel_use_improved_policy[AZMI_MAX_PERMS][AZMI_MAX_PLAYERS];
  uint32_t seat_gumbel_m[AZMI_MAX_PERMS][AZMI_MAX_PLAYERS];
  uint32_t seat_resign_consecutive[AZMI_MAX_PERMS][AZMI_MAX_PLAYERS];
  float seat_gumbel_c_visit[AZMI_MAX_PERMS][AZMI_MAX_PLAYERS];
  float seat_gumbel_c_scale[AZMI_MAX_PERMS][AZMI_MAX_PLAYERS];
  float seat_resign_threshold[AZMI_MAX_PERMS][AZMI_MAX_PLAYERS];
  /* temp_decay_half_life_by_variant (play_manager.h:87-90, play_manager.cc:290-296): entry get_variant_id() replaces
   * temp_decay_half_life when it exists; games without variants (id -1) ignore it */
  uint32_t num_temp_decay_half_life_by_variant;
  float temp_decay_half_life_by_variant[4];
} azmi_play_params;

/* engine-only knobs that have no reference counterpart */
typedef struct azmi_engine_opts {
  uint64_t seed;          /* slot s uses pcg32 stream slot_seed(seed, s) */
  int32_t device;         /* HIP device ordinal */
  uint32_t max_inline;    /* simulations a slot may finish inside one round without the net
                             (terminal leaves, RANDOM eval, cache hits); 0 = default */
  uint32_t history_capacity; /* rows of the device history buffer; 0 = sized from params */
  int32_t log_moves;      /* keep a per-move log (parity tests) */
  uint32_t move_log_capacity;
  /* the base GameState of PlayManager(gs, params) where it has constructor arguments: StarGambitUnifiedGS(pinned_variant,
   * probs) (py_wrapper.cc:662-667; defaults -1 and four times 0.25).  Every new game draws its variant from the slot's coin
   * stream unless pinned (build-defined draw: the reference's engine is unseedable, star_gambit_gs.cc:2357-2362) */
  int32_t sg_pinned_variant;
  float sg_variant_probs[4];
} azmi_engine_opts;

typedef struct azmi_pm azmi_pm;

/* `stream` arguments are hipStream_t values used as given (NULL is the HIP null stream);
 * AZMI_STREAM_ENGINE selects the engine's own non-blocking stream. Result queries order
 * themselves behind the stream of the most recent round/play/poll call. */
#define AZMI_STREAM_ENGINE ((void*)(intptr_t)-1)

void azmi_play_params_default(azmi_play_params* p);      /* PlayParams{} defaults, play_manager.h:60-154 */
void azmi_engine_opts_default(azmi_engine_opts* o);
const char* azmi_last_error(void);
int azmi_abi_version(void);
int azmi_device_count(void);

/* static game facts: GameState::num_moves/num_players + CANONICAL_SHAPE (py_wrapper.cc:157-189, 562-580) */
int azmi_game_info(int game, uint32_t* num_players, uint32_t* num_moves, uint32_t chw[3]);

/* PlayManager(gs, params) — py_wrapper.cc:352-354, play_manager.cc:12-256 */
int azmi_pm_create(int game, const azmi_play_params* params, const azmi_engine_opts* opts, azmi_pm** out);
/* PlayManager(gs, params, caches) — py_wrapper.cc:355-360, play_manager.cc:644-649: params.max_cache_size is ignored and model
 * group g uses caches[g] (an azmi_cache from azmi_cache_create, see below; NULL entries = no cache for that group).  The
 * caches are not owned: they must outlive the engine, and they keep their contents from one engine to the next.  The engine
 * probes 64-entry shards, so a cache must have been created with shards = max_size / 64. */
typedef struct azmi_cache azmi_cache;
int azmi_pm_create_with_caches(int game, const azmi_play_params* params, const azmi_engine_opts* opts, azmi_cache* const* caches,
                               uint32_t num_caches, azmi_pm** out);
void azmi_pm_destroy(azmi_pm* pm);

/* ---- device fast path (what PlayManager::play + GameRunner's batcher/result_worker do,
 *      play_manager.cc:258-600 + game_runner.py:651-726, without leaving HBM) ---------------
 * One round = every live slot: consume its (v, pi) row -> backup -> maybe play a move ->
 * descend to the next leaf -> write that leaf's canonical planes into row `slot` of the
 * leaf batch.  Rows are slot-indexed, so the batch shape is always
 * [concurrent_games, C, H, W] and a round needs no host round-trip. */
int azmi_pm_round(azmi_pm* pm, void* stream);
/* device pointers of the slot-indexed buffers: canonical [S,C,H,W] (engine writes, net reads),
 * v [S,P+1] and pi [S,M] (net writes probabilities, engine reads) — replaces
 * build_batch / update_inferences, py_wrapper.cc:449-504 / play_manager.cc:619-642 */
int azmi_pm_io_buffers(azmi_pm* pm, float** dev_canonical, float** dev_v, float** dev_pi);
/* play() for EvalType::RANDOM on every seat: rounds until games_completed >= games_to_play */
int azmi_pm_play(azmi_pm* pm, void* stream);
/* copies the small control block back (sync on `stream`); mirrors remaining_games()/games_completed() */
int azmi_pm_poll(azmi_pm* pm, void* stream, uint32_t* games_completed, uint32_t* live_slots);

/* stop() / stopped(), play_manager.h:177-178: play() and build_batch return at their next check; may be called from another thread */
int azmi_pm_stop(azmi_pm* pm);
int azmi_pm_stopped(azmi_pm* pm, int* out);
/* awaiting_inference_count() / awaiting_mcts_count(), play_manager.h:316-323 */
int azmi_pm_queue_counts(azmi_pm* pm, uint32_t* awaiting_inference, uint32_t* awaiting_mcts);
/* game_data(i).gs, play_manager.h:286: the packed state of the game in slot i.  Connect4: words = {stones of player 0, stones
 * of player 1 (bit h*7+w), turn | player << 32}; Tafl family: {defenders lo, hi, attackers lo, hi (bit = square),
 * king square | turn << 8 | player << 24 | repetition count << 32}; one more word follows: GameData::perm_index, the seat
 * permutation the slot's game runs under (play_manager.h:41).  `cap` >= state words + 1.  AZMI_ERR_RANGE for a bad index. */
int azmi_pm_slot_state(azmi_pm* pm, uint32_t slot, uint64_t* words, uint32_t cap, uint32_t* n);
/* StarGambit: state words = 20 packed units (two per word: type 0:2 | player 2 | slot 3:3 | hp 6:3 | facing 9:3 | q+6 12:4 |
 * r+6 16:4 | moves_left 20:2 | cannons_fired 22:4 | exists 26) + one word player | turn << 8 | misc << 24 (units 0:5, acted 5, over 6,
 * winner 7:2, variant 9:2) | reserves << 40 (3 bits per [player][type]); its position history (star_gambit_gs.h:745) comes
 * from azmi_pm_slot_history: the slot's entries since the last deploy (the reference's own position hashes); 0 entries for Connect4 */
int azmi_pm_slot_history(azmi_pm* pm, uint32_t slot, uint64_t* out, uint32_t cap, uint32_t* n);
/* game_data(i).canonical(), py_wrapper.cc:279-288: the leaf planes slot i is waiting on, to a HOST array [C,H,W] */
int azmi_pm_slot_canonical(azmi_pm* pm, uint32_t slot, float* out);

/* ---- results (all synchronise on the engine's stream first) -------------------------------- */
int azmi_pm_scores(azmi_pm* pm, float* out /* [P+1] */);          /* scores(), play_manager.h:173 */
int azmi_pm_resign_scores(azmi_pm* pm, float* out /* [P+1] */);   /* resign_scores(), :174 */
/* out[7] = avg_game_length, avg_leaf_depth, avg_search_entropy, fast_avg_leaf_depth,
 *          fast_avg_search_entropy, avg_moves_per_turn, avg_valid_moves (play_manager.h:288-315) */
int azmi_pm_stats(azmi_pm* pm, float* out);
/* the sums behind those averages, to combine several engines (shards of one GPU, ranks) exactly: out[10] = sum of game
 * lengths, games completed, total / full / fast move counts, sums of leaf depth, search entropy, fast leaf depth, fast
 * entropy, valid moves (play_manager.h:398-424) */
int azmi_pm_stat_sums(azmi_pm* pm, double* out);
/* out[6] = simulations, leaf evaluations requested from the net, cache hits, cache misses,
 *          history rows available, rounds */
int azmi_pm_counters(azmi_pm* pm, uint64_t* out);
/* out[6] = cache_hits, cache_misses, cache_evictions, cache_reinserts, cache_size, cache_max_size summed over the model
 *          groups' caches (play_manager.h:325-366) */
int azmi_pm_cache_stats(azmi_pm* pm, uint64_t out[6]);
/* build_history_batch, py_wrapper.cc:393-424: copies up to `cap` finished rows to HOST arrays
 * canonical [cap,C,H,W], v [cap,P+1], pi [cap,M]; returns rows written in *n */
int azmi_pm_pop_history(azmi_pm* pm, float* canonical, float* v, float* pi, uint32_t cap, uint32_t* n);
/* same rows, left in HBM: device pointers + row count (for the RCCL sample gather) */
int azmi_pm_history_device(azmi_pm* pm, float** dev_canonical, float** dev_v, float** dev_pi,
                           uint32_t** dev_meta, uint32_t* rows);
/* The finished-sample store is a RING of `capacity` rows (the reference's history queue is unbounded and drained by
 * GameRunner.hist_saver, game_runner.py:729-747): the unread rows are `rows` rows starting at physical row `first_row`
 * of the azmi_pm_history_device arrays, wrapping modulo `capacity`.  azmi_pm_history_consume releases the oldest `rows`
 * unread rows (azmi_pm_pop_history does both for host arrays).  The engine stops with AZMI_ERR_OVERFLOW (mask 2) when a
 * finished game finds less room than its rows. */
int azmi_pm_history_window(azmi_pm* pm, uint32_t* first_row, uint32_t* rows, uint32_t* capacity);
int azmi_pm_history_consume(azmi_pm* pm, uint32_t rows);
/* per-move log (opts.log_moves): rows [n,8] = slot, game_in_slot, move, turn, player, capped,
 * pcg32 state (lo, hi) of the slot's tree stream right before pick_move's draw;
 * counts [n,M] = root counts() right before the move.  Parity-test hook. */
int azmi_pm_move_log(azmi_pm* pm, uint32_t* rows, uint32_t* counts, uint32_t cap, uint32_t* n);
/* games each slot has completed, [S] */
int azmi_pm_slot_games(azmi_pm* pm, uint32_t* out);

/* ---- host-buffer compatibility path (exact reference signatures) ---------------------------
 * build_batch(group, batch) — py_wrapper.cc:449-504: runs rounds until a leaf batch is
 * pending, copies the live rows to the HOST array batch[cap,C,H,W], returns their slot ids.
 * batch == NULL hands out the indices only: pop_game / pop_games_upto (play_manager.h:186-192). */
int azmi_pm_build_batch(azmi_pm* pm, float* batch, uint32_t cap, uint32_t* indices, uint32_t* n);
/* update_inferences(group, indices, v, pi) — play_manager.cc:619-642, HOST arrays */
int azmi_pm_update_inferences(azmi_pm* pm, const uint32_t* indices, uint32_t n, const float* v, const float* pi);

/* per-variant tables of a game with variants (play_manager.h:218-275; play_manager.cc:237-255, 468-484).
 * azmi_pm_num_variants: num_tracked_variants().  azmi_pm_variant_sums(v): perm_scores [perms][P+1] (variant_perm_scores; their
 * sum over perms = variant_scores), perm_games [perms], sums[10] = game_length, games, total / full / fast move counts, leaf
 * depth, entropy, fast leaf depth, fast entropy, valid moves - the accumulators behind the variant_avg_* getters */
uint32_t azmi_pm_num_variants(azmi_pm* pm);
int azmi_pm_variant_sums(azmi_pm* pm, uint32_t variant, float* perm_scores, uint32_t* perm_games, double* sums);

/* ---- GameState / MCTS single-object surface on the device (py_wrapper.cc:157-220) ----------
 * Batched over `n` independent states so one launch covers many objects. Used by the
 * parity tests of the rules kernels (SURVEY tier T0). `moves` is [n, len] row-major; a
 * negative entry stops that game early.  Outputs (any may be NULL):
 *   valid [n,M] u8, scores [n,P+1] f32 (all -1 if not over), canonical [n,C,H,W] f32,
 *   player [n] u32, turn [n] u32, key [n] u64, status [n] i32 (0 ok, -1 illegal move). */
int azmi_game_replay(int game, int device, const int32_t* moves, uint32_t n, uint32_t len,
                     uint8_t* valid, float* scores, float* canonical, uint32_t* player,
                     uint32_t* turn, uint64_t* key, int32_t* status);
/* the same from given start positions: `init` is [n, init_stride] bytes, one serialized state per
 * game in the reference's pickle image (Connect4GS::to_bytes, connect4_gs.cc:172-190: int8
 * board[2][6][7], int8 player, int32 turn = 89 bytes; Connect4GS(board, player, turn) ctor,
 * py_wrapper.cc:563-581).  Brandubh / OpenTafl: int8 board[3][N][N] (king, defenders, attackers), int8 player,
 * int32 turn = 3*N*N + 5 bytes, with an empty repetition map (the reference tests' MakeGS helper,
 * opentafl_gs_test.cc:97-101).  StarGambit: StarGambitUnifiedGS::to_bytes (star_gambit_gs.cc:2446-2449 around the inner image
 * :2246-2251), any length up to init_stride, rows padded with zeros; its replay/playout kernels take one wavefront per game.
 * NULL = the game's initial position (StarGambit: the Skirmish variant). */
int azmi_game_replay_from(int game, int device, const uint8_t* init, uint32_t init_stride, const int32_t* moves,
                          uint32_t n, uint32_t len, uint8_t* valid, float* scores, float* canonical,
                          uint32_t* player, uint32_t* turn, uint64_t* key, int32_t* status);
/* playout_eval(gs) / playout_eval_batch(states), game_state.cc:10-95 (Python: py_wrapper.cc:726-770): for each of n states
 * (start position + move list, as in azmi_game_replay_from) pi = uniform over the legal moves, v = the scores of a uniformly
 * random rollout.  The reference's rollout RNG is an unseedable thread-local engine; here state i uses a pcg32 stream
 * seeded with seeds[i].  v [n,P+1], pi [n,M] are HOST arrays.  All five games; EvalType::PLAYOUT seats of an engine run the same
 * rollout on the device from the slot's third pcg32 stream. */
int azmi_playout_eval(int game, int device, const uint8_t* init, uint32_t init_stride, const int32_t* moves, uint32_t n, uint32_t len,
                      const uint64_t* seeds, float* v, float* pi);
/* flags bit 0 (Brandubh / OpenTafl): apply moves the way the reference's play_move does — no ownership or slide
 * check, captures judged from the moved piece's side (brandubh_gs.cc:342-427; the reference's own tests rely on
 * it).  Without the flag a move that valid_moves() does not list ends the game record with status -1. */
int azmi_game_replay_ex(int game, int device, const uint8_t* init, uint32_t init_stride, const int32_t* moves,
                        uint32_t n, uint32_t len, uint8_t* valid, float* scores, float* canonical,
                        uint32_t* player, uint32_t* turn, uint64_t* key, int32_t* status, uint32_t flags);

/* StarGambitUnifiedGS::to_bytes (star_gambit_gs.cc:2451-2465; py::pickle of the game objects, py_wrapper.cc:77-83) of n
 * states given as start image + moves (as azmi_game_replay_from; flags as azmi_game_replay_ex): out [n, out_stride] bytes,
 * out_len [n] image sizes, status [n] (0 ok, -1 illegal move / malformed start, -2 image longer than out_stride).  The
 * variant_probs / pinned_variant fields of the header (bytes 0-19) are left zero: they are constructor arguments the caller keeps. */
int azmi_sg_image(int device, const uint8_t* init, uint32_t init_stride, const int32_t* moves, uint32_t n, uint32_t len,
                  uint8_t* out, uint32_t out_stride, uint32_t* out_len, int32_t* status, uint32_t flags);

/* ---- training-sample symmetries (GameState::symmetries(PlayHistory), py_wrapper.cc:178;
 *      connect4_gs.cc:151-170, tafl_helper.h:16-149, tawlbwrdd_gs.cc:455-458) ------------------
 * `count` samples in: canon [count,C,H,W], v [count,P+1], pi [count,M]; out: the NUM_SYMMETRIES
 * images of every sample, sample-major, in the reference's order (Connect4 {base, mirror}; Tafl
 * {base, r, r^2, r^3, m(base), m(r), m(r^2), m(r^3)}): out_canon [count,NS,C,H,W], out_v
 * [count,NS,P+1], out_pi [count,NS,M].  Pointers are DEVICE pointers (asynchronous on `stream`)
 * unless host_buffers != 0 (then HOST arrays, staged through HBM, synchronous). */
uint32_t azmi_num_symmetries(int game);
int azmi_symmetries(int game, int device, uint32_t count, const float* canon, const float* v, const float* pi,
                    float* out_canon, float* out_v, float* out_pi, int host_buffers, void* stream);
/* the same for any square Tafl board (Brandubh 7, Tawlbwrdd 11, OpenTafl 11 share tafl_helper.h) */
int azmi_tafl_symmetries(uint32_t board, uint32_t channels, uint32_t num_values, int device, uint32_t count,
                         const float* canon, const float* v, const float* pi, float* out_canon, float* out_v,
                         float* out_pi, int host_buffers, void* stream);
const char* azmi_symmetries_last_error(void);

/* ---- the stand-alone MCTS class (py_wrapper.cc:192-220, mcts.h:50-200): one search tree driven call by call.
 * All five games.  A GameState argument is passed as start position (`init`, NULL = initial position, else the
 * game's serialized image as in azmi_game_replay_from) + the moves played from it.  The object owns one pcg32
 * stream (the reference shares a thread_local one across all trees of a thread). */
typedef struct azmi_mcts azmi_mcts;
typedef struct azmi_mcts_config {   /* ctor arguments, mcts.h:52-72 */
  float cpuct; uint32_t num_players, num_moves;
  float epsilon, root_policy_temp, fpu_reduction;
  int32_t relative_values, root_fpu_zero, shaped_dirichlet, gumbel_enabled;
  uint32_t gumbel_m; float gumbel_c_visit, gumbel_c_scale; int32_t gumbel_full;
  uint32_t max_simulations;         /* arena size: simulations over the object's lifetime (0 = 50000) */
} azmi_mcts_config;
int azmi_mcts_create(int game, const azmi_mcts_config* cfg, uint64_t seed, int device, azmi_mcts** out);
void azmi_mcts_destroy(azmi_mcts* m);
/* find_leaf(gs): leaf_moves[0 .. *leaf_len) are the moves from gs to the leaf (leaf GameState = gs + those moves) */
int azmi_mcts_find_leaf(azmi_mcts* m, const uint8_t* init, uint32_t init_bytes, const int32_t* moves, uint32_t len,
                        int32_t* leaf_moves, uint32_t cap, uint32_t* leaf_len);
/* process_result(gs, value, pi, root_noise_enabled); value_out [P+1] = the reference's in-out `value` afterwards */
int azmi_mcts_process_result(azmi_mcts* m, const float* value, const float* pi, int root_noise_enabled, float* value_out);
/* update_root(gs, move); returns AZMI_ERR_INVALID with "ahh, what is this move" for a move the root does not have */
int azmi_mcts_update_root(azmi_mcts* m, const uint8_t* init, uint32_t init_bytes, const int32_t* moves, uint32_t len, uint32_t move);
/* WU-UCT batched search (mcts.h:124-129, mcts.cc:752-851; Python: py_wrapper.cc:212-215).  find_leaf_batched descends with the
 * in-flight penalty (Node::n_in_flight in PUCT's denominator and parent count), appends an entry to the in-flight list and returns
 * the leaf like find_leaf; the entry's index is in_flight_count() - 1.  process_result_batched(leaf_index, ...) backs one entry up
 * (AZMI_ERR_RANGE for an index that is not in the list: the reference's in_flight_.at() -> IndexError); reset_batch() clears the
 * list (at most 1024 entries between resets).  The batched descent is plain PUCT also when Gumbel is enabled, like the reference. */
int azmi_mcts_find_leaf_batched(azmi_mcts* m, const uint8_t* init, uint32_t init_bytes, const int32_t* moves, uint32_t len,
                                int32_t* leaf_moves, uint32_t cap, uint32_t* leaf_len);
int azmi_mcts_process_result_batched(azmi_mcts* m, uint32_t leaf_index, const float* value, const float* pi, int root_noise_enabled,
                                     float* value_out);
int azmi_mcts_in_flight_count(const azmi_mcts* m, uint32_t* out);
int azmi_mcts_reset_batch(azmi_mcts* m);
/* read-outs and small mutations, kind: 0 counts (u32 [M]) 1 probs(temp) 2 probs_pruned(temp) (f32 [M]) 3 root_value (f32 [3])
 * 4 root_q_values (f32 [M]) 5 scalars: u = {depth, root_n, root children}, f = {avg_leaf_depth, normalized_root_entropy}
 * 6 gumbel_improved_policy (f32 [M]) 7 gumbel_final_action (u[0]) 8 add_root_noise 9 apply_root_policy_temp
 * 10 pick_move: in_f = pi [M], u[0] = move 11 principal_variation(depth = arg): u[0] = length, u[1..] = moves
 * 12 set_gumbel_num_sims(arg).  out_f / out_u are HOST arrays of at least max(M, 64) entries (may be NULL if unused). */
int azmi_mcts_query(azmi_mcts* m, uint32_t kind, float temp, uint32_t arg, const float* in_f, float* out_f, uint32_t* out_u);

/* ---- leaf policy/value network (the reference's NNArch forward + NNWrapper.process,
 *      neural_net.py:448-510, 800-823) as one fused MFMA kernel ---------------------------------
 * `blob` is the BatchNorm-folded, MFMA-fragment-ordered weight image produced by
 * alphazero/hip_net.py::fold() (layout documented there and in DESIGN.md); it is copied to HBM.
 * forward(): canonical [batch,C,H,W] f32 -> v [batch,P+1], pi [batch,M] f32 probabilities, all
 * device pointers, asynchronous on `stream`. */
typedef struct azmi_net_desc {
  int32_t in_channels, height, width;   /* CANONICAL_SHAPE */
  int32_t channels, depth, kernel_size; /* NNArgs.num_channels / depth / kernel_size */
  int32_t head_channels, v_hidden;      /* NNArgs.head_channels / v_fc_hidden */
  int32_t num_moves, num_players;
  /* head options (neural_net.py:56-75, 341-427). policy_channels > 0 selects the spatial policy head
   * (POLICY_SHAPE[0]; Tafl family 22, Brandubh 14) and its kernel; 0 = flat FC policy head.
   * bf16 kernels instantiated: 6x7 flat head (64 trunk / 32 head channels), 11x11 and 7x7 spatial head (64 / 64 channels;
   * narrower nets are passed zero-padded to 64 by fold()).  Any other shape: AZMI_ERR_INVALID, or precision = 1. */
  int32_t v_head_convs, pi_head_convs, v_fc_layers, policy_channels;
  /* 0 = bf16 operands on the matrix cores, fp32 accumulation (what the reference runs under autocast,
   * neural_net.py:811-813) — the throughput path.  1 = plain fp32 arithmetic, layer by layer: the precision
   * the 1e-5 parity tier refers to, for any net shape; blob layout in csrc/leafnet_f32.hip.  2 = "bf16x3" (6x7 flat-head
   * family only): the bf16 tile with weights and activations split into bf16 high + low parts, three MFMAs per product -
   * within 1e-5 of the fp32 outputs (measured 4e-7 / 5e-6) at a third of the bf16 rate; blob: the bf16 layout with the
   * stem's fragments twice (high, low) and three chunks [W_hi][W_hi][W_lo] per tap and for the head 1x1 (hip_net.fold). */
  int32_t precision;
  /* spatial head with GLOBAL actions (num_moves > policy_channels * H * W: StarGambit's 18 deploys + end turn,
   * neural_net.py:413-426, 486-493): hidden width of pi_global (NNArgs.pi_fc_hidden); the global logits come from the
   * average-pooled policy features through Linear - ReLU - Linear - LayerNorm and join the softmax behind the spatial block.
   * 0 when the net has no global actions.  bf16 kernel: also instantiated for 13x13 (configs/star_gambit_unified.yaml). */
  int32_t pi_hidden;
} azmi_net_desc;
typedef struct azmi_net azmi_net;
size_t azmi_net_blob_bytes(const azmi_net_desc* desc);
int azmi_net_create(const azmi_net_desc* desc, const void* blob, size_t blob_bytes, int device, azmi_net** out);
void azmi_net_destroy(azmi_net* net);
int azmi_net_forward(azmi_net* net, const float* dev_canonical, float* dev_v, float* dev_pi, uint32_t batch, void* stream);
/* the same on a subset of the rows: dev_rows[0 .. *dev_row_count) (device memory, read when the kernel runs) are
 * the row indices to evaluate; rows not listed are left untouched.  The engine lists the slots whose pending
 * leaf really needs the net (cache hits, terminal leaves and retired slots do not). */
int azmi_net_forward_rows(azmi_net* net, const float* dev_canonical, float* dev_v, float* dev_pi, const uint32_t* dev_rows,
                          const uint32_t* dev_row_count, uint32_t max_rows, void* stream);
/* NNWrapper.process on HOST arrays (neural_net.py:800-823 behind game_runner.py:651-726's batcher -> gpu_loop -> result_worker):
 * canonical [n,C,H,W] -> v [n,P+1], pi [n,M]; copies in, runs the net, copies out, synchronously, on a stream and staging
 * buffers private to the calling thread (any number of host threads may call it at once).  The signature is that of a
 * host PlayManager's evaluator callback (user = the azmi_net*), so a CPU-side PlayManager can be served by the GPU
 * through host buffers without an interpreter in the loop.  Errors are reported by filling v / pi with NaN. */
void azmi_net_eval_host(const float* canonical, uint32_t n, float* v, float* pi, void* net);
/* azmi_net_forward_rows on an engine's own leaf batch and eval list (what azmi_run_rounds does after each round) */
int azmi_pm_net_forward(azmi_pm* pm, azmi_net* net, void* stream);
/* one engine round with its leaf evaluation, as azmi_run_rounds issues it: part 0 = the whole round, 1 = the tree half only
 * (cache insert, simulations), 2 = the net half only (leaf net over the eval list; on the split-round Connect4 engine the
 * round's move step rides in the same launch).  Parts 1 and 2 exist so that a caller can put timing events between them. */
int azmi_pm_round_net(azmi_pm* pm, azmi_net* net, void* stream, uint32_t part);
/* the same for one model group: only the leaves whose seat belongs to `group` (play_manager.cc:577, 597) */
int azmi_pm_net_forward_group(azmi_pm* pm, uint32_t group, azmi_net* net, void* stream);
/* num_model_groups() / num_seat_perms(), play_manager.h:210-211 */
int azmi_pm_groups(azmi_pm* pm, uint32_t* num_model_groups, uint32_t* num_seat_perms);
/* perm_scores(idx) / perm_games_completed(idx), play_manager.h:212-217: out_scores [P+1] */
int azmi_pm_perm_scores(azmi_pm* pm, uint32_t perm, float* out_scores, uint32_t* games_completed);
/* build_batch / update_inferences restricted to one model group (py_wrapper.cc:449-504, play_manager.cc:619-642) */
int azmi_pm_build_batch_group(azmi_pm* pm, uint32_t group, float* batch, uint32_t cap, uint32_t* indices, uint32_t* n);
const char* azmi_net_last_error(void);

/* ---- device position cache on its own: S3FIFOCache / ShardedS3FIFOCache (s3fifo_cache.h:15-318,
 *      Python classes at py_wrapper.cc:222-259).  max_size / ghost_size are totals over `shards`;
 *      shard = hash % shards.  insert_many keeps batch order inside a shard (eviction order is the
 *      reference's); the finds of one call are concurrent.  All pointers are HOST arrays.
 *      stats out[6] = hits, misses, evictions, reinserts, size, max_size. */
int azmi_cache_create(uint32_t max_size, uint32_t shards, uint32_t ghost_size, uint32_t num_policy, uint32_t num_value,
                      int device, azmi_cache** out);
void azmi_cache_destroy(azmi_cache* cache);
int azmi_cache_insert_many(azmi_cache* cache, const uint64_t* hashes, const float* policy, const float* value, uint32_t n);
int azmi_cache_find_many(azmi_cache* cache, const uint64_t* hashes, uint32_t n, uint8_t* hit, float* policy, float* value);
int azmi_cache_stats(azmi_cache* cache, uint64_t out[6]);
const char* azmi_cache_last_error(void);

/* ---- native round driver (the counterpart of GameRunner's batcher / gpu_loop / result_worker
 *      threads, game_runner.py:483-552, 651-726): `rounds` times, for each of the `k` engines in turn:
 *      one round on streams[i], then the net on that engine's slot-indexed leaf batch on the same
 *      stream.  Engines on different streams overlap: while one engine's leaf batch is on the matrix
 *      cores, another engine's tree kernel runs on the CUs the net leaves free.  Launch-only
 *      (asynchronous); poll with azmi_pm_poll. */
int azmi_run_rounds(azmi_pm* const* pms, azmi_net* net, uint32_t k, uint32_t rounds, void* const* streams);
/* ---- asynchronous tree / net pipeline (Connect4 engine; plain PUCT seats + one model group on the fast tree kernel, Gumbel seats and
 *      two model groups on the generic one; bf16 or bf16x3 Connect4-family net): what
 *      azmi_run_rounds does, without its lock step.  The reference's worker loop has no global barrier - every game advances
 *      on its own through the queues between PlayManager::play's workers and GameRunner's batcher threads
 *      (play_manager.cc:258-600, concurrent_queue.h:130-217, game_runner.py:483-552) - and neither has this: for one EPOCH
 *      persistent tree wavefronts simulate their slots and hand the leaves that need the net to persistent net workgroups
 *      (request ring / tagged result granules in HBM, csrc/pipe_types.h); an epoch ends after `sims_per_epoch` simulations
 *      (held to 1024 per slot; an epoch also ends a quarter of the way to the 250 ms stall cap, or when an eighth of the slots'
 *      games have ended in it), then game restarts, the moves the epoch's end cut off and the position-cache inserts of its answers
 *      run at a kernel boundary.  A slot lives in ONE tree workgroup (slot % tree workgroups) for an epoch - searches AND moves -, so
 *      nothing but requests, answers and READY tokens crosses between CUs inside an epoch.  `epochs` epochs, synchronous (returns
 *      when they are done).  net == NULL: an engine whose seats all use EvalType::RANDOM runs on the tree kernel alone.
 *      A pipeline error (a spin that hit the time cap: another tenant holds the chip, the host stalled between the two launches)
 *      is reported here ONCE and cleared: slots whose requests went unanswered are back in the move step's kSlotQueued form, so
 *      the next call of either driver carries on.  The k-th game of a slot is that of azmi_run_rounds for the same seed; with a finite
 *      games_to_play, WHICH slots receive the last restarts depends on the order in which games end inside an epoch (as the reference's
 *      workers race for games_started_, play_manager.cc:506-513), so the SET of games is only the same while no game restarts.
 *      out_stats[14] = the calibration launches of this call (bits 0-7) | freezes - tree wavefronts that stood still for more than
 *      2 ms between two looks of their poll loop (seen with hundreds of idle net workgroups beside a small engine);
 *      credited against the time caps, not errors - since creation (bits 8-31) | requests given up and sent again since creation
 *      (bits 32-63).  An epoch launches min(the measured places, S / 3 + 8) net workgroups: a slot has at most one request out and
 *      a tile takes three.
 *      Worst case of ONE epoch: the tree side's stall detector is the 250 ms cap (AZMI_PIPE_CAP_MS); the net side only waits for
 *      the tree side and outwaits it for 40 caps = 10 s before it raises error bit 64 itself - a tree kernel that never starts
 *      (the host held up between the two launches, no place on the chip) is an error after 10 s, not a hang.  A net workgroup beyond
 *      the S / 3 + 8 the slots can use (only launched under AZMI_PIPE_NET_ALL) that idles beside a tree side without progress for
 *      1.25 caps leaves instead.
 *      out_stats (may be NULL): [0] net tiles run since the pipeline was created, [1] boards in them, [2] simulations of the
 *      last epoch, [3] / [4] tree / net workgroups that started in it, [5] its insert-log entries, [6] / [7] net / tree
 *      workgroups launched, [8] / [9] the latest start of a tree / net workgroup after the epoch's first, in microseconds (all
 *      of them must be on the chip together: a late one found its place only when another left), [10] / [11] the summed
 *      durations of this call's net / tree kernels in microseconds (HIP events on their streams), [12] the epochs of this
 *      call, [13] the host time spent enqueueing them in microseconds (the host runs ahead of the GPU), [14] the calibration
 *      launches that measured [6] (0 = not measured: tree side alone); 16 entries.
 *      After azmi_pm_stop both this call and azmi_run_rounds return AZMI_OK at once and run nothing (play_manager.cc:272). */
int azmi_run_pipeline(azmi_pm* pm, azmi_net* net, uint32_t epochs, uint64_t sims_per_epoch, void* stream, uint64_t* out_stats);
/* the same with one net PER MODEL GROUP (azmi_run_rounds_groups' counterpart: play_past, game_runner.py:2184-2332 - two models, seats
 * swapped by the permutations): nets[g] evaluates the leaves of group g (its own request ring, its own S3-FIFO), NULL = that group
 * needs no net (RANDOM seats: the reference's RandPlayer).  At most two groups; every net a Connect4-family matrix-core net of ONE
 * precision tier.  Engines the fast tree kernel does not cover - Gumbel seats, two model groups - run on the generic tree kernel
 * (every step through the lock-step engine's own move-step function, one simulation per slot and pass, no round barrier). */
int azmi_run_pipeline_groups(azmi_pm* pm, azmi_net* const* nets, uint32_t num_nets, uint32_t epochs, uint64_t sims_per_epoch, void* stream,
                             uint64_t* out_stats);
int azmi_pipeline_supported_groups(azmi_pm* pm, azmi_net* const* nets, uint32_t num_nets);
/* diagnostics: the pipeline's persistent net kernel alone, draining `n` synthetic requests (n <= 32768, the request ring) `reps` times with
 * `net_wgs` workgroups (0 = the pipeline's own count) and tile selection `mode` (0 = 3- and 6-board tiles, 1 = 6-board, 2 =
 * 3-board); *ms_out = milliseconds per drain.  Timing only (scripts/pipe_net_timing.py -> profiles/). */
int azmi_debug_pipe_net_bench(azmi_pm* pm, azmi_net* net, uint32_t n, uint32_t reps, uint32_t net_wgs, int mode, float* ms_out);
/* diagnostics: the net side's ANSWERS for `n` synthetic positions (n <= the engine's concurrent games; position i is asked for slot i,
 * sequence number i + 1), drained once by the conveyor (conveyor != 0, csrc/conveyor_c4.h; `lines` lines, 0 = its default; mode 3 of
 * azmi_debug_pipe_net_bench times it, net_wgs = lines) or by the tile kernel: out[10 i + k] = pi[k] (k < 7), v[k - 7] of position i,
 * out_seq[i] = the sequence number its result granules carry.  The two paths answer bit for bit alike (tests/test_gpu_conveyor.py). */
int azmi_debug_pipe_net_answers(azmi_pm* pm, azmi_net* net, uint32_t n, uint64_t seed, int conveyor, uint32_t lines, float* out, uint32_t* out_seq);
/* diagnostics: out[0] = answers the tree side consumed in the last epoch of the last azmi_run_pipeline call (its insert log), out[1] =
 * those whose key is in the log more than once (evaluations an insert at answer time - PlayManager::update_inferences,
 * play_manager.cc:631-640 - or a table of requests in flight would have saved), out[2] = of those, twins within 4096 log entries */
int azmi_debug_pipe_log_dupes(azmi_pm* pm, uint64_t* out);
/* 1 when azmi_run_pipeline can drive this engine with this net (net may be NULL for an engine whose seats all use
 * EvalType::RANDOM: the tree kernel alone), 0 otherwise (then azmi_run_rounds is the driver) */
int azmi_pipeline_supported(azmi_pm* pm, azmi_net* net);
/* the same loop with one net per MODEL GROUP (gating / benchmark matches between two models, game_runner.py:2184-2332):
 * nets[g] evaluates the leaves of group g, NULL = the group needs no net (RANDOM / PLAYOUT evaluator) */
int azmi_run_rounds_groups(azmi_pm* const* pms, azmi_net* const* nets, uint32_t num_nets, uint32_t k, uint32_t rounds, void* const* streams);

/* ---- the path's one exchange step: finished self-play samples go to rank 0 over RCCL / xGMI ---------------------------
 * Replaces the reference's hist_saver (game_runner.py:729-747: one process, a queue) for one process per GPU.  No collective runs
 * during the search.  librccl is loaded (dlopen) by the first of these calls; a one-GPU process that never gathers does not need it.
 *   azmi_comm_available  AZMI_OK when librccl and every entry point used here load in this process (no communicator is made): what
 *                        the ranks agree on BEFORE any of them enters ncclCommInitRank - a rank that would fail there strands the
 *                        others inside it
 *   azmi_comm_unique_id  rank 0: the 128 bytes every rank passes to azmi_comm_create (carried there by the launcher's own channel:
 *                        bench.py broadcasts them with torch.distributed)
 *   azmi_comm_create     ncclCommInitRank on `device`
 *   azmi_gather_counts   every rank: its row count in, all ranks' counts out (one ncclAllGather of 8 bytes per rank; synchronises
 *                        `stream`)
 *   azmi_gather_rows     every rank: `num_parts` device arrays of counts[rank] rows, row_bytes[i] bytes per row; rank 0 receives them
 *                        in rank order, unpadded, into dst[i] (device memory for sum(counts) rows; ignored on the other ranks):
 *                        grouped ncclSend / ncclRecv, its own rows by a device copy.  Asynchronous on `stream`. */
typedef struct azmi_comm azmi_comm;
int azmi_comm_available(void);
int azmi_comm_unique_id(void* out128);
int azmi_comm_create(const void* id128, int rank, int world, int device, azmi_comm** out);
void azmi_comm_destroy(azmi_comm* comm);
int azmi_gather_counts(azmi_comm* comm, uint64_t n_local, uint64_t* out_counts, void* stream);
int azmi_gather_rows(azmi_comm* comm, const void* const* src, const uint64_t* row_bytes, uint32_t num_parts, const uint64_t* counts,
                     void* const* dst, void* stream);

/* The device RNG layer on its own (parity tier "RNG"): runs `thread_local pcg32 re` + the
 * libstdc++ algorithm the reference applies to it (mcts.cc:19,100,430-440,718) on the GPU.
 * kind 0: n raw pcg32 outputs (u32)      1: std::shuffle of iota(n), reps times (u32 [reps,n])
 *      2: n uniform_real<float>(0,1)      3: n gamma_distribution<float>(param,1), one object
 *      4: same, a fresh object per draw.  `out` is a HOST array of n (kind 1: reps*n) 4-byte items. */
int azmi_rng_probe(int device, int kind, uint64_t seed, float param, uint32_t n, uint32_t reps, void* out);

#ifdef __cplusplus
}
#endif
#endif /* AZMI_H_ */
