"""ctypes front-end of the CPU oracle (oracle/liboracle.so) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
REF_RNG = os.path.join(ORACLE_DIR, "_ref", "librngref.so")

GAME_CONNECT4, GAME_TAWLBWRDD, GAME_BRANDUBH, GAME_OPENTAFL, GAME_STARGAMBIT = 0, 1, 2, 3, 4


def build():
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".hpp", ".cpp"))]
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs):
        return
    subprocess.run(["make", "-C", ORACLE_DIR, "liboracle.so"], check=True, capture_output=True)


build()
lib = C.CDLL(LIB)
for name in ("orc_game_new", "orc_c4_from_board", "orc_tafl_from_board", "orc_game_copy", "orc_mcts_new", "orc_mcts_leaf", "orc_pm_new",
             "orc_cache_new", "orc_sg_unified_new", "orc_sg_plain_new", "orc_pm_new_game"):
    getattr(lib, name).restype = C.c_void_p
lib.orc_game_key.restype = C.c_uint64
lib.orc_pm_hist_count.restype = C.c_uint64
lib.orc_pm_move_count.restype = C.c_uint64
lib.orc_mcts_avg_leaf_depth.restype = C.c_float
lib.orc_mcts_entropy.restype = C.c_float
lib.orc_node_uct.restype = C.c_float
lib.orc_node_uct.argtypes = [C.c_float, C.c_float, C.c_uint32, C.c_float, C.c_float, C.c_float]


def ref_rng():
    """The real-reference RNG leg (reference pcg header + libstdc++), or None if not built."""
    return C.CDLL(REF_RNG) if os.path.exists(REF_RNG) else None


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ RNG layer
def rng_call(which, fn, *args, n, dtype):
    out = np.zeros(n, dtype)
    getattr(which, fn)(*args, _p(out))
    return out


def pcg32_outputs(seed, n, which=None):
    return rng_call(which or lib, ("ref_" if which else "orc_") + "pcg32_outputs", C.c_uint64(seed), C.c_uint32(n), n=n, dtype=np.uint32)


def shuffle_iota(seed, n, reps=1, which=None):
    return rng_call(which or lib, ("ref_" if which else "orc_") + "shuffle_iota", C.c_uint64(seed), C.c_uint32(n), C.c_uint32(reps),
                    n=n * reps, dtype=np.uint32).reshape(reps, n)


def uniform01(seed, n, which=None):
    return rng_call(which or lib, ("ref_" if which else "orc_") + "uniform01", C.c_uint64(seed), C.c_uint32(n), n=n, dtype=np.float32)


def gamma(seed, alpha, n, fresh_each=False, which=None):
    return rng_call(which or lib, ("ref_" if which else "orc_") + "gamma", C.c_uint64(seed), C.c_float(alpha), C.c_float(1.0),
                    C.c_uint32(n), C.c_int(int(fresh_each)), n=n, dtype=np.float32)


def gumbel(seed, n, which=None):
    return rng_call(which or lib, ("ref_" if which else "orc_") + "gumbel", C.c_uint64(seed), C.c_uint32(n), n=n, dtype=np.float32)


# ------------------------------------------------------------------ games
class Game:
    def __init__(self, game_id=GAME_CONNECT4, handle=None, owned=True):
        self.h = C.c_void_p(handle if handle is not None else lib.orc_game_new(game_id))
        self.owned = owned
        assert self.h.value

    @classmethod
    def connect4_from_board(cls, board, player, turn):
        b = np.ascontiguousarray(board, dtype=np.int8)
        assert b.shape == (2, 6, 7)
        return cls(handle=lib.orc_c4_from_board(_p(b), C.c_int8(player), C.c_int32(turn)))

    @classmethod
    def tafl_from_board(cls, game_id, board, player, turn=10, max_turns=400):
        """MakeGS of opentafl_gs_test.cc:97-101: int8 board [3,N,N], empty repetition map, count 1."""
        b = np.ascontiguousarray(board, dtype=np.int8)
        h = lib.orc_tafl_from_board(C.c_int(game_id), _p(b), C.c_int8(player), C.c_uint32(turn), C.c_uint32(max_turns))
        assert h
        return cls(handle=h)

    @classmethod
    def sg_unified(cls, pinned=-1, probs=(0.25, 0.25, 0.25, 0.25), first_variant=0):
        """StarGambitUnifiedGS(pinned_variant, probs) (py_wrapper.cc:662-667); `first_variant` stands for the constructor's own
        unseedable draw when the game is not pinned."""
        pr = np.ascontiguousarray(probs, np.float32)
        h = lib.orc_sg_unified_new(C.c_int(pinned), _p(pr), C.c_int(first_variant))
        assert h
        return cls(handle=h)

    @classmethod
    def sg_plain(cls, variant):
        """StarGambit{Skirmish,Showdown,Clash,Battle}GS in their own action space."""
        h = lib.orc_sg_plain_new(C.c_int(variant))
        assert h
        return cls(handle=h)

    def sg_units(self):
        """rows of (type, player, slot, hp, facing, q, r, moves_left, cannons_fired), dead units included"""
        out = np.zeros((64, 9), np.int32)
        n = lib.orc_sg_units(self.h, _p(out), C.c_uint32(64))
        return out[:n].copy()

    def sg_info(self):
        out = np.zeros(13, np.int32)
        lib.orc_sg_info(self.h, _p(out))
        return dict(reserves=out[:8].reshape(2, 4).copy(), acted=int(out[8]), over=int(out[9]), winner=int(out[10]),
                    history_len=int(out[11]), variant=int(out[12]))

    def sg_to_bytes(self):
        buf = np.zeros(1 << 16, np.uint8)
        n = lib.orc_sg_to_bytes(self.h, _p(buf), C.c_uint32(buf.size))
        return buf[:n].tobytes()

    def sg_from_bytes(self, data):
        """inner-game from_bytes (star_gambit_gs.cc:2290-2338) into this object"""
        b = np.frombuffer(bytes(data), np.uint8).copy()
        if lib.orc_sg_from_bytes(self.h, _p(b), C.c_uint32(b.size)) != 0:
            raise RuntimeError("StarGambitGS::from_bytes failed")

    def equals(self, other): return lib.orc_game_equal(self.h, other.h) == 1
    def relative_values(self): return bool(lib.orc_game_relative_values(self.h))
    def variant(self): return int(lib.orc_game_variant(self.h))
    def num_variants(self): return int(lib.orc_game_num_variants(self.h))

    def __del__(self):
        if getattr(self, "owned", False) and self.h:
            lib.orc_game_free(self.h)
            self.h = None

    def copy(self):
        return Game(handle=lib.orc_game_copy(self.h))

    def play(self, m):
        if lib.orc_game_play(self.h, C.c_uint32(m)) != 0:
            raise RuntimeError("Invalid move: You have a bug in your code.")

    def num_moves(self): return lib.orc_game_num_moves(self.h)
    def num_players(self): return lib.orc_game_num_players(self.h)
    def player(self): return lib.orc_game_player(self.h)
    def turn(self): return lib.orc_game_turn(self.h)
    def key(self): return lib.orc_game_key(self.h)

    def valid(self):
        out = np.zeros(self.num_moves(), np.uint8)
        lib.orc_game_valid(self.h, _p(out))
        return out

    def scores(self):
        out = np.zeros(self.num_players() + 1, np.float32)
        return out if lib.orc_game_scores(self.h, _p(out)) else None

    def canonical(self):
        chw = (C.c_int * 3)()
        lib.orc_game_canonical_shape(self.h, chw)
        out = np.zeros(tuple(chw), np.float32)
        lib.orc_game_canonical(self.h, _p(out))
        return out


# ------------------------------------------------------------------ single tree
class MctsCfg(C.Structure):
    _fields_ = [("cpuct", C.c_float), ("num_players", C.c_uint32), ("num_moves", C.c_uint32), ("epsilon", C.c_float),
                ("root_policy_temp", C.c_float), ("fpu_reduction", C.c_float), ("relative_values", C.c_int32),
                ("root_fpu_zero", C.c_int32), ("shaped_dirichlet", C.c_int32), ("gumbel_enabled", C.c_int32),
                ("gumbel_m", C.c_uint32), ("gumbel_c_visit", C.c_float), ("gumbel_c_scale", C.c_float),
                ("gumbel_full", C.c_int32)]


class Mcts:
    def __init__(self, cpuct, num_players, num_moves, epsilon=0.0, root_policy_temp=1.0, fpu_reduction=0.0,
                 relative_values=False, root_fpu_zero=False, shaped_dirichlet=False, gumbel_enabled=False, gumbel_m=16,
                 gumbel_c_visit=50.0, gumbel_c_scale=1.0, gumbel_full=False, seed=0):
        cfg = MctsCfg(cpuct, num_players, num_moves, epsilon, root_policy_temp, fpu_reduction, int(relative_values),
                      int(root_fpu_zero), int(shaped_dirichlet), int(gumbel_enabled), gumbel_m, gumbel_c_visit,
                      gumbel_c_scale, int(gumbel_full))
        self.h = C.c_void_p(lib.orc_mcts_new(C.byref(cfg), C.c_uint64(seed)))
        self.P, self.M = num_players, num_moves

    def __del__(self):
        if self.h:
            lib.orc_mcts_free(self.h)
            self.h = None

    def set_gumbel_num_sims(self, n):
        lib.orc_mcts_set_gumbel_num_sims(self.h, C.c_uint32(n))

    def gumbel_improved_policy(self):
        out = np.zeros(self.M, np.float32); lib.orc_mcts_gumbel_improved_policy(self.h, _p(out)); return out

    def gumbel_final_action(self):
        return int(lib.orc_mcts_gumbel_final_action(self.h))

    def gumbel_state(self):
        sv = np.zeros(512, np.uint32); g = np.zeros(512, np.float32)
        n = lib.orc_mcts_gumbel_state(self.h, _p(sv), _p(g), C.c_uint32(512))
        return sv[:n].copy(), g

    def search_dumb(self, game, sims, noise=False):
        lib.orc_mcts_search_dumb(self.h, game.h, C.c_uint32(sims), C.c_int(int(noise)))

    def find_leaf(self, game):
        lib.orc_mcts_find_leaf(self.h, game.h)
        return Game(handle=lib.orc_mcts_leaf(self.h), owned=False)

    def process_result(self, value, pi, noise=False):
        v = np.ascontiguousarray(value, np.float32).copy()
        p = np.ascontiguousarray(pi, np.float32)
        lib.orc_mcts_process_result(self.h, _p(v), _p(p), C.c_int(int(noise)))
        return v

    def find_leaf_batched(self, game):
        lib.orc_mcts_find_leaf_batched(self.h, game.h)
        return Game(handle=lib.orc_mcts_leaf(self.h), owned=False)

    def process_result_batched(self, leaf_index, value, pi, noise=False):
        v = np.ascontiguousarray(value, np.float32).copy()
        p = np.ascontiguousarray(pi, np.float32)
        if lib.orc_mcts_process_result_batched(self.h, C.c_uint32(leaf_index), _p(v), _p(p), C.c_int(int(noise))) != 0:
            raise IndexError("leaf_index out of range")
        return v

    def in_flight_count(self): return lib.orc_mcts_in_flight_count(self.h)
    def reset_batch(self): lib.orc_mcts_reset_batch(self.h)

    def update_root(self, game, move):
        if lib.orc_mcts_update_root(self.h, game.h, C.c_uint32(move)) != 0:
            raise RuntimeError("ahh, what is this move")

    def counts(self):
        out = np.zeros(self.M, np.uint32); lib.orc_mcts_counts(self.h, _p(out)); return out

    def root_q(self):
        out = np.zeros(self.M, np.float32); lib.orc_mcts_root_q(self.h, _p(out)); return out

    def probs(self, temp, pruned=False):
        out = np.zeros(self.M, np.float32)
        lib.orc_mcts_probs(self.h, C.c_float(temp), C.c_int(int(pruned)), _p(out))
        return out

    def pick_move(self, p):
        p = np.ascontiguousarray(p, np.float32)
        return lib.orc_mcts_pick_move(self.h, _p(p), C.c_uint32(len(p)))

    def depth(self): return lib.orc_mcts_depth(self.h)
    def root_n(self): return lib.orc_mcts_root_n(self.h)
    def avg_leaf_depth(self): return lib.orc_mcts_avg_leaf_depth(self.h)
    def entropy(self): return lib.orc_mcts_entropy(self.h)

    def root_value(self):
        out = np.zeros(3, np.float32); lib.orc_mcts_root_value(self.h, _p(out)); return out

    def add_root_noise(self): lib.orc_mcts_add_root_noise(self.h)
    def apply_root_policy_temp(self): lib.orc_mcts_apply_root_policy_temp(self.h)

    def principal_variation(self, depth=5):
        out = np.zeros(max(depth, 1), np.uint32)
        n = lib.orc_mcts_principal_variation(self.h, C.c_uint32(depth), _p(out))
        return out[:n].copy()

    def root_children(self):
        mv = np.zeros(4096, np.uint32); pol = np.zeros(4096, np.float32); n = np.zeros(4096, np.uint32); q = np.zeros(4096, np.float32)
        k = lib.orc_mcts_root_children(self.h, _p(mv), _p(pol), _p(n), _p(q))
        return mv[:k], pol[:k], n[:k], q[:k]


def dumb_eval(game):
    """dumb_eval(gs) -> (value [P+1], pi [M]), game_state.h:160-173"""
    v = np.zeros(game.num_players() + 1, np.float32); pi = np.zeros(game.num_moves(), np.float32)
    lib.orc_dumb_eval(game.h, _p(v), _p(pi))
    return v, pi


def playout_eval(game, seed):
    """playout_eval(gs) with the rollout stream seeded by `seed` -> (value [P+1], pi [M]), game_state.cc:10-59"""
    v = np.zeros(game.num_players() + 1, np.float32); pi = np.zeros(game.num_moves(), np.float32)
    lib.orc_playout_eval(game.h, C.c_uint64(seed), _p(v), _p(pi))
    return v, pi


def node_uct(q, policy, n, sqrt_parent_n, cpuct, fpu):
    return lib.orc_node_uct(q, policy, n, sqrt_parent_n, cpuct, fpu)


# ------------------------------------------------------------------ PlayManager
class OrcPlayParams(C.Structure):
    _fields_ = [("games_to_play", C.c_uint32), ("concurrent_games", C.c_uint32), ("max_batch_size", C.c_uint32),
                ("max_cache_size", C.c_uint32), ("cache_shards", C.c_uint32), ("mcts_visits", C.c_uint32 * 4),
                ("cpuct", C.c_float), ("start_temp", C.c_float), ("final_temp", C.c_float),
                ("temp_decay_half_life", C.c_float), ("history_enabled", C.c_int32), ("tree_reuse", C.c_int32),
                ("epsilon", C.c_float), ("mcts_root_temp", C.c_float), ("playout_cap_randomization", C.c_int32),
                ("playout_cap_depth", C.c_uint32), ("playout_cap_percent", C.c_float), ("fpu_reduction", C.c_float),
                ("root_fpu_zero", C.c_int32), ("shaped_dirichlet", C.c_int32), ("policy_target_pruning", C.c_int32),
                ("resign_percent", C.c_float), ("resign_playthrough_percent", C.c_float), ("eval_type", C.c_int32 * 4),
                ("gumbel_enabled", C.c_int32), ("gumbel_m", C.c_uint32), ("gumbel_c_visit", C.c_float),
                ("gumbel_c_scale", C.c_float), ("gumbel_full", C.c_int32), ("fast_search_uses_gumbel", C.c_int32),
                ("num_model_groups_given", C.c_uint32), ("model_groups", C.c_uint8 * 4),
                ("num_seat_perms", C.c_uint32), ("seat_perms", (C.c_uint8 * 4) * 8),
                ("has_seat_visits", C.c_int32), ("has_seat_cap_visits", C.c_int32), ("has_seat_epsilon", C.c_int32),
                ("has_seat_mcts_root_temp", C.c_int32), ("has_seat_root_fpu_zero", C.c_int32),
                ("seat_visits", (C.c_uint32 * 4) * 8), ("seat_cap_visits", (C.c_uint32 * 4) * 8),
                ("seat_epsilon", (C.c_float * 4) * 8), ("seat_mcts_root_temp", (C.c_float * 4) * 8),
                ("seat_root_fpu_zero", (C.c_uint8 * 4) * 8), ("perm_base", C.c_uint32),
                ("has_seat_gumbel_enabled", C.c_int32), ("has_seat_gumbel_m", C.c_int32), ("has_seat_gumbel_c_visit", C.c_int32),
                ("has_seat_gumbel_c_scale", C.c_int32), ("has_seat_gumbel_full", C.c_int32),
                ("has_seat_gumbel_use_improved_policy", C.c_int32), ("has_seat_resign_threshold", C.c_int32),
                ("has_seat_resign_consecutive", C.c_int32),
                ("seat_gumbel_enabled", (C.c_uint8 * 4) * 8), ("seat_gumbel_full", (C.c_uint8 * 4) * 8),
                ("seat_gumbel_use_improved_policy", (C.c_uint8 * 4) * 8),
                ("seat_gumbel_m", (C.c_uint32 * 4) * 8), ("seat_resign_consecutive", (C.c_uint32 * 4) * 8),
                ("seat_gumbel_c_visit", (C.c_float * 4) * 8), ("seat_gumbel_c_scale", (C.c_float * 4) * 8),
                ("seat_resign_threshold", (C.c_float * 4) * 8)]


GROUP_EVAL_FN = C.CFUNCTYPE(None, C.c_uint32, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p)


def fill_seat_fields(c, pp):
    """model_groups / seat_perms / seat_* matrices of a PlayParams-shaped object into a ctypes params struct
    (shared by the oracle and, through alphazero._capi, the engine: same field names)."""
    groups = list(getattr(pp, "model_groups", []) or [])
    c.num_model_groups_given = len(groups)
    for i, g in enumerate(groups):
        c.model_groups[i] = int(g)
    perms = [list(p) for p in (getattr(pp, "seat_perms", []) or [])]
    c.num_seat_perms = len(perms)
    for q, row in enumerate(perms):
        for sidx, g in enumerate(row):
            c.seat_perms[q][sidx] = int(g)
    for name, cast in (("seat_visits", int), ("seat_cap_visits", int), ("seat_epsilon", float), ("seat_mcts_root_temp", float),
                       ("seat_root_fpu_zero", int), ("seat_gumbel_enabled", int), ("seat_gumbel_m", int),
                       ("seat_gumbel_c_visit", float), ("seat_gumbel_c_scale", float), ("seat_gumbel_full", int),
                       ("seat_gumbel_use_improved_policy", int), ("seat_resign_threshold", float), ("seat_resign_consecutive", int)):
        mat = getattr(pp, name, []) or []
        setattr(c, "has_" + name, int(bool(mat)))
        for q, row in enumerate(mat):
            for sidx, x in enumerate(row):
                getattr(c, name)[q][sidx] = cast(x)


EVAL_FN = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p)


def params_from(pp, num_players):
    """Build the oracle's params from an `alphazero.PlayParams`-shaped object (same field names)."""
    c = OrcPlayParams()
    for name in ("games_to_play", "concurrent_games", "max_batch_size", "max_cache_size", "cache_shards",
                 "playout_cap_depth"):
        setattr(c, name, int(getattr(pp, name)))
    for name in ("cpuct", "start_temp", "final_temp", "temp_decay_half_life", "epsilon", "mcts_root_temp",
                 "playout_cap_percent", "fpu_reduction", "resign_percent", "resign_playthrough_percent"):
        setattr(c, name, float(getattr(pp, name)))
    for name in ("history_enabled", "tree_reuse", "playout_cap_randomization", "root_fpu_zero", "shaped_dirichlet",
                 "policy_target_pruning"):
        setattr(c, name, int(bool(getattr(pp, name))))
    for i, v in enumerate(pp.mcts_visits):
        c.mcts_visits[i] = int(v)
    c.gumbel_enabled = int(bool(getattr(pp, "gumbel_enabled", False)))
    c.gumbel_m = int(getattr(pp, "gumbel_m", 16))
    c.gumbel_c_visit = float(getattr(pp, "gumbel_c_visit", 50.0))
    c.gumbel_c_scale = float(getattr(pp, "gumbel_c_scale", 1.0))
    c.gumbel_full = int(bool(getattr(pp, "gumbel_full", False)))
    c.fast_search_uses_gumbel = int(bool(getattr(pp, "fast_search_uses_gumbel", False)))
    for i in range(4):
        c.eval_type[i] = -1
    for i, e in enumerate(pp.eval_type):
        c.eval_type[i] = int(e)
    fill_seat_fields(c, pp)
    return c


class PlayManager:
    def __init__(self, game_id, pp, seed, per_slot_rng=True, record_moves=True, num_players=2, perm_base=0):
        """game_id: one of GAME_*, or an oracle `Game` object used as the base game (copied)."""
        self.c = params_from(pp, num_players)
        self.c.perm_base = int(perm_base)
        if isinstance(game_id, Game):
            g = game_id
            hl = np.ascontiguousarray(getattr(pp, "temp_decay_half_life_by_variant", []) or [], np.float32)
            self.h = C.c_void_p(lib.orc_pm_new_game(g.h, C.byref(self.c), C.c_uint64(seed), int(per_slot_rng), int(record_moves),
                                                    _p(hl) if hl.size else None, C.c_uint32(hl.size)))
        else:
            self.h = C.c_void_p(lib.orc_pm_new(game_id, C.byref(self.c), C.c_uint64(seed), int(per_slot_rng), int(record_moves)))
            g = Game(game_id)
        if not self.h.value:
            raise RuntimeError("oracle PlayManager construction failed")
        self.P, self.M = g.num_players(), g.num_moves()
        self.chw = g.canonical().shape

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_pm_free(self.h)
            self.h = None

    def run(self, evaluator=None):
        """evaluator(canonical[n,C,H,W]) -> (v[n,P+1], pi[n,M]) for NN seats."""
        if evaluator is None:
            rc = lib.orc_pm_run(self.h, None, None)
        else:
            chw, P, M = self.chw, self.P, self.M

            def cb(canon, n, v, pi, _user):
                x = np.ctypeslib.as_array(canon, shape=(n,) + tuple(chw))
                vv, pp = evaluator(x.copy())
                np.ctypeslib.as_array(v, shape=(n, P + 1))[:] = vv
                np.ctypeslib.as_array(pi, shape=(n, M))[:] = pp

            self._cb = EVAL_FN(cb)
            rc = lib.orc_pm_run(self.h, self._cb, None)
        if rc != 0:
            raise RuntimeError("oracle PlayManager.run failed")

    def set_time_limit(self, seconds):
        """bench hook: run() returns after `seconds` of wall time even if games_to_play is not reached."""
        lib.orc_pm_set_time_limit(self.h, C.c_double(seconds))

    def run_native(self, fn_ptr, user):
        """run() with a NATIVE evaluator: fn_ptr(canonical, n, v, pi, user) is a C function pointer (e.g. libazmi's
        azmi_net_eval_host), so no Python runs inside the loop and the call releases the GIL."""
        if lib.orc_pm_run(self.h, C.c_void_p(fn_ptr), C.c_void_p(user)) != 0:
            raise RuntimeError("oracle PlayManager.run failed")

    def run_groups(self, evaluator):
        """evaluator(group, canonical[n,C,H,W]) -> (v, pi): one evaluator per model group (play_manager.cc:577-597)."""
        chw, P, M = self.chw, self.P, self.M

        def cb(group, canon, n, v, pi, _user):
            x = np.ctypeslib.as_array(canon, shape=(n,) + tuple(chw))
            vv, pp = evaluator(int(group), x.copy())
            np.ctypeslib.as_array(v, shape=(n, P + 1))[:] = vv
            np.ctypeslib.as_array(pi, shape=(n, M))[:] = pp

        self._gcb = GROUP_EVAL_FN(cb)
        if lib.orc_pm_run_groups(self.h, self._gcb, None) != 0:
            raise RuntimeError("oracle PlayManager.run_groups failed")

    def num_model_groups(self): return int(lib.orc_pm_num_groups(self.h))
    def num_seat_perms(self): return int(lib.orc_pm_num_perms(self.h))

    def perm_scores(self, perm):
        out = np.zeros(self.P + 1, np.float32)
        n = lib.orc_pm_perm_scores(self.h, C.c_uint32(perm), _p(out))
        return out, int(n)

    def num_tracked_variants(self): return int(lib.orc_pm_num_variants(self.h))

    def variant(self, v):
        """per-variant tables (play_manager.h:218-275): dict(scores, games, perm_scores, perm_games, stats[7])"""
        np_ = self.num_seat_perms()
        sc = np.zeros(self.P + 1, np.float32); ps = np.zeros((np_, self.P + 1), np.float32); pg = np.zeros(np_, np.uint32)
        st = np.zeros(7, np.float32)
        n = lib.orc_pm_variant(self.h, C.c_uint32(v), _p(sc), _p(ps), _p(pg), _p(st))
        return dict(scores=sc, games=int(n), perm_scores=ps, perm_games=pg, stats=st)

    def scores(self):
        out = np.zeros(self.P + 1, np.float32); lib.orc_pm_scores(self.h, _p(out)); return out

    def resign_scores(self):
        out = np.zeros(self.P + 1, np.float32); lib.orc_pm_resign_scores(self.h, _p(out)); return out

    def games_completed(self): return lib.orc_pm_games_completed(self.h)

    def stats(self):
        out = np.zeros(7, np.float32); lib.orc_pm_stats(self.h, _p(out)); return out

    def counters(self):
        out = np.zeros(5, np.uint64); lib.orc_pm_counters(self.h, _p(out))
        return dict(zip(("sims", "evals", "cache_hits", "cache_misses", "hist_rows"), (int(x) for x in out)))

    def history(self):
        n = int(lib.orc_pm_hist_count(self.h))
        c = np.zeros((n,) + tuple(self.chw), np.float32); v = np.zeros((n, self.P + 1), np.float32); p = np.zeros((n, self.M), np.float32)
        if n:
            lib.orc_pm_history(self.h, _p(c), _p(v), _p(p))
        return c, v, p

    def moves(self):
        n = int(lib.orc_pm_move_count(self.h))
        rows = np.zeros((n, 8), np.uint32); counts = np.zeros((n, self.M), np.uint32)
        if n:
            lib.orc_pm_moves(self.h, _p(rows), _p(counts), C.c_uint32(self.M))
        return rows, counts


def seq_halving_phase_plan(m, n):
    out = np.zeros(64, np.uint32)
    lib.orc_seq_halving_phase_plan.restype = C.c_uint32
    k = lib.orc_seq_halving_phase_plan(C.c_uint32(m), C.c_uint32(n), _p(out), C.c_uint32(32))
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(k)]


def v_mix(raw_v, q, n, prior):
    q = np.ascontiguousarray(q, np.float32); n = np.ascontiguousarray(n, np.uint32); prior = np.ascontiguousarray(prior, np.float32)
    lib.orc_v_mix.restype = C.c_float
    return float(lib.orc_v_mix(C.c_float(raw_v), _p(q), _p(n), _p(prior), C.c_uint32(len(q))))


def slot_seed(seed, slot):
    """orc::slot_seed — mix64(seed + golden * (slot + 1))."""
    M = (1 << 64) - 1
    x = (seed + 0x9E3779B97F4A7C15 * (slot + 1)) & M
    x = (x + 0x9E3779B97F4A7C15) & M
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
    return x ^ (x >> 31)


# ------------------------------------------------------------------ S3-FIFO
class Cache:
    def __init__(self, max_size, shards, ghost, num_policy, num_value):
        self.h = C.c_void_p(lib.orc_cache_new(max_size, shards, ghost, num_policy, num_value))
        self.np_, self.nv = num_policy, num_value

    def __del__(self):
        if self.h:
            lib.orc_cache_free(self.h)
            self.h = None

    def find(self, key):
        p = np.zeros(self.np_, np.float32); v = np.zeros(self.nv, np.float32)
        hit = lib.orc_cache_find(self.h, C.c_uint64(key), _p(p), _p(v))
        return (p, v) if hit else None

    def insert(self, key, p, v):
        p = np.ascontiguousarray(p, np.float32); v = np.ascontiguousarray(v, np.float32)
        lib.orc_cache_insert(self.h, C.c_uint64(key), _p(p), _p(v))

    def stats(self):
        out = np.zeros(6, np.uint64); lib.orc_cache_stats(self.h, _p(out))
        return dict(zip(("hits", "misses", "evictions", "reinserts", "size", "max_size"), (int(x) for x in out)))


# ------------------------------------------------------------------ symmetries
SYM_CONNECT4, SYM_TAFL_EIGHT, SYM_TAFL_MIRROR, SYM_TAFL_ROT90, SYM_STARGAMBIT = 0, 1, 2, 3, 4


def symmetries(kind, canon, v, pi):
    """orc::connect4_symmetries / eight_sym / mirror_width / rot90_clockwise on ONE sample.
    canon [C,H,W], v [nv], pi [M] -> (canon [ns,C,H,W], v [ns,nv], pi [ns,M])."""
    canon = np.ascontiguousarray(canon, np.float32); v = np.ascontiguousarray(v, np.float32)
    pi = np.ascontiguousarray(pi, np.float32)
    ns = {SYM_CONNECT4: 2, SYM_TAFL_EIGHT: 8, SYM_STARGAMBIT: 2}.get(kind, 1)
    oc = np.zeros((ns,) + canon.shape, np.float32); ov = np.zeros((ns, v.size), np.float32)
    op = np.zeros((ns, pi.size), np.float32)
    lib.orc_symmetries.restype = C.c_uint32
    got = lib.orc_symmetries(C.c_int(kind), C.c_int(canon.shape[0]), C.c_int(canon.shape[1]), C.c_int(canon.shape[2]),
                             C.c_uint32(pi.size), C.c_uint32(v.size), _p(canon), _p(v), _p(pi), _p(oc), _p(ov), _p(op))
    assert got == ns
    return oc, ov, op
