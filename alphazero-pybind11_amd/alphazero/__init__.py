"""`alphazero` — host-side mirror of the reference's pybind11 module
(/root/reference/src/py_wrapper.cc:108-788) for the self-play hot path, backed by
the MI355X engine in libazmi.so (include/azmi.h).

Same names, argument meaning and error behaviour as the reference for
PlayParams / EvalType / PlayManager and the game classes' static facts, so
`game_runner.py`-style callers work unchanged; plus the device fast path the
reference lacks (`PlayManager.round`, `io_tensors`).

Put the directory that contains this package on sys.path:
    sys.path.insert(0, "<repo>/alphazero-pybind11_amd"); import alphazero
"""
import ctypes as C
import enum
import os
import struct

import numpy as np

from . import _capi
from ._capi import lib, check


def __getattr__(name):  # torch-dependent pieces are imported on first use
    if name == "HipLeafNet":
        from .hip_net import HipLeafNet
        return HipLeafNet
    raise AttributeError(name)

__all__ = [
    "EvalType", "PlayParams", "PlayManager", "GameState", "Connect4GS", "TawlbwrddGS", "S3FIFOCache", "ShardedS3FIFOCache",
    "tracy_is_enabled", "tracy_frame_mark",
]


class EvalType(enum.IntEnum):  # py_wrapper.cc:290-293
    NN = 0
    RANDOM = 1
    PLAYOUT = 2


class PlayParams:
    """py_wrapper.cc:295-349 / play_manager.h:60-154 — same field names and defaults."""

    def __init__(self):
        self.games_to_play = 0
        self.concurrent_games = 0
        self.max_batch_size = 1
        self.max_cache_size = 0
        self.cache_shards = 1
        self.queue_shards = 1
        self.eval_pipelines = 1
        self.mcts_visits = []
        self.cpuct = 2.0
        self.start_temp = 1.0
        self.final_temp = 1.0
        self.temp_decay_half_life = 0.0
        self.temp_decay_half_life_by_variant = []
        self.history_enabled = False
        self.self_play = False
        self.tree_reuse = True
        self.epsilon = 0.0
        self.mcts_root_temp = 1.0
        self.playout_cap_randomization = False
        self.playout_cap_depth = 25
        self.playout_cap_percent = 0.75
        self.fpu_reduction = 0.0
        self.root_fpu_zero = False
        self.shaped_dirichlet = False
        self.policy_target_pruning = False
        self.gumbel_enabled = False
        self.gumbel_m = 16
        self.gumbel_c_visit = 50.0
        self.gumbel_c_scale = 1.0
        self.gumbel_full = False
        self.fast_search_uses_gumbel = False
        self.resign_percent = 0.0
        self.resign_playthrough_percent = 0.0
        self.eval_type = []
        self.model_groups = []
        self.seat_perms = []
        self.seat_visits = []
        self.seat_cap_visits = []
        self.seat_epsilon = []
        self.seat_mcts_root_temp = []
        self.seat_root_fpu_zero = []
        self.seat_gumbel_enabled = []
        self.seat_gumbel_m = []
        self.seat_gumbel_c_visit = []
        self.seat_gumbel_c_scale = []
        self.seat_gumbel_full = []
        self.seat_gumbel_use_improved_policy = []
        self.seat_resign_threshold = []
        self.seat_resign_consecutive = []

    # fields the device engine does not implement yet: anything but the default is an error.
    # (temp_decay_half_life_by_variant is NOT one of them: play_manager.cc:289-296 applies entry get_variant_id() only when
    # that id is >= 0, and every device game answers -1 (game_state.h:79), so the reference itself ignores the list there.)
    _UNSUPPORTED = ()

    def _to_c(self, num_players):
        c = _capi.PlayParamsC()
        lib.azmi_play_params_default(C.byref(c))
        for name in self._UNSUPPORTED:
            if getattr(self, name):
                raise RuntimeError(f"PlayParams.{name} is not supported by the MI355X engine yet")
        # model groups / seat permutations / per-seat matrices: same validation messages as play_manager.cc:57-113
        groups = [int(g) for g in self.model_groups]
        if groups and len(groups) != num_players:
            raise RuntimeError("model_groups must be empty or have one entry per player")
        perms = [[int(g) for g in row] for row in self.seat_perms]
        if len(perms) > _capi.AZMI_MAX_PERMS:
            raise RuntimeError(f"at most {_capi.AZMI_MAX_PERMS} seat permutations")
        n_perms = len(perms) if perms else 1
        c.num_model_groups_given = len(groups)
        for i, g in enumerate(groups):
            c.model_groups[i] = g
        c.num_seat_perms = len(perms)
        for q, row in enumerate(perms):
            if len(row) != num_players:
                raise RuntimeError("seat_perms inner dimension must match number of players")
            for sidx, g in enumerate(row):
                c.seat_perms[q][sidx] = g
        for name, cast in (("seat_visits", int), ("seat_cap_visits", int), ("seat_epsilon", float),
                           ("seat_mcts_root_temp", float), ("seat_root_fpu_zero", int),
                           ("seat_gumbel_enabled", int), ("seat_gumbel_m", int), ("seat_gumbel_c_visit", float),
                           ("seat_gumbel_c_scale", float), ("seat_gumbel_full", int), ("seat_gumbel_use_improved_policy", int),
                           ("seat_resign_threshold", float), ("seat_resign_consecutive", int)):
            mat = [list(row) for row in getattr(self, name)]
            setattr(c, "has_" + name, int(bool(mat)))
            if not mat:
                continue
            if len(mat) != n_perms:
                raise RuntimeError(f"{name} outer dimension must match number of seat permutations")
            for q, row in enumerate(mat):
                if len(row) != num_players:
                    raise RuntimeError(f"{name} inner dimension must match number of players")
                for sidx, x in enumerate(row):
                    getattr(c, name)[q][sidx] = cast(x)
        c.games_to_play = int(self.games_to_play)
        c.concurrent_games = int(self.concurrent_games)
        c.max_batch_size = int(self.max_batch_size)
        c.max_cache_size = int(self.max_cache_size)
        c.cache_shards = int(self.cache_shards)
        visits = list(self.mcts_visits)
        if len(visits) > _capi.AZMI_MAX_PLAYERS:
            raise RuntimeError("You must specify MCTS visits for each player")
        c.num_mcts_visits = len(visits)
        for i, v in enumerate(visits):
            c.mcts_visits[i] = int(v)
        c.cpuct = self.cpuct
        c.start_temp = self.start_temp
        c.final_temp = self.final_temp
        c.temp_decay_half_life = self.temp_decay_half_life
        hl = [float(x) for x in self.temp_decay_half_life_by_variant]     # play_manager.h:87-90 (entry get_variant_id())
        if len(hl) > 4:
            raise RuntimeError("temp_decay_half_life_by_variant: at most 4 variants")
        c.num_temp_decay_half_life_by_variant = len(hl)
        for i, x in enumerate(hl):
            c.temp_decay_half_life_by_variant[i] = x
        c.history_enabled = int(bool(self.history_enabled))
        c.self_play = int(bool(self.self_play))
        c.tree_reuse = int(bool(self.tree_reuse))
        c.epsilon = self.epsilon
        c.mcts_root_temp = self.mcts_root_temp
        c.playout_cap_randomization = int(bool(self.playout_cap_randomization))
        c.playout_cap_depth = int(self.playout_cap_depth)
        c.playout_cap_percent = self.playout_cap_percent
        c.fpu_reduction = self.fpu_reduction
        c.root_fpu_zero = int(bool(self.root_fpu_zero))
        c.shaped_dirichlet = int(bool(self.shaped_dirichlet))
        c.policy_target_pruning = int(bool(self.policy_target_pruning))
        c.resign_percent = self.resign_percent
        c.resign_playthrough_percent = self.resign_playthrough_percent
        c.gumbel_enabled = int(bool(self.gumbel_enabled))
        c.gumbel_m = int(self.gumbel_m)
        c.gumbel_c_visit = float(self.gumbel_c_visit)
        c.gumbel_c_scale = float(self.gumbel_c_scale)
        c.gumbel_full = int(bool(self.gumbel_full))
        c.fast_search_uses_gumbel = int(bool(self.fast_search_uses_gumbel))
        ets = list(self.eval_type)
        c.num_eval_type = len(ets)
        for i, e in enumerate(ets[: _capi.AZMI_MAX_PLAYERS]):
            c.eval_type[i] = int(e)
        return c


class PlayHistory:
    """One training sample (game_state.h:31-35, py_wrapper.cc:111-155): canonical [C,H,W], v [P+1], pi [M]."""

    def __init__(self, canonical, v, pi):
        if canonical is None or v is None or pi is None:
            raise TypeError("PlayHistory(canonical, v, pi): arguments must not be None")
        self._c = np.array(canonical, dtype=np.float32, order="C")
        self._v = np.array(v, dtype=np.float32).reshape(-1)
        self._pi = np.array(pi, dtype=np.float32).reshape(-1)
        if self._c.ndim != 3:
            raise RuntimeError("PlayHistory: canonical must have 3 dimensions")

    def canonical(self):
        return memoryview(self._c)

    def v(self):
        return self._v

    def pi(self):
        return self._pi


def symmetries_batch(game_cls, canonical, v, pi, device=0, stream=None):
    """All NUM_SYMMETRIES images of a batch of samples in one launch (azmi_symmetries).
    numpy in -> numpy out ([n, NS, ...]); torch CUDA tensors in -> torch CUDA tensors out, no host copy."""
    ns = game_cls.NUM_SYMMETRIES()
    P, M, chw = game_cls._info()
    if hasattr(canonical, "is_cuda"):
        import torch
        if not (canonical.is_cuda and v.is_cuda and pi.is_cuda):
            raise RuntimeError("symmetries_batch: torch inputs must be CUDA tensors (numpy arrays take the staged path)")
        c = canonical.contiguous().float(); vv = v.contiguous().float(); pp = pi.contiguous().float()
        n = c.shape[0]
        if tuple(c.shape[1:]) != tuple(chw) or tuple(vv.shape) != (n, P + 1) or tuple(pp.shape) != (n, M):
            raise RuntimeError("Improper sample shapes")
        oc = torch.empty((n, ns) + tuple(chw), dtype=torch.float32, device=c.device)
        ov = torch.empty((n, ns, P + 1), dtype=torch.float32, device=c.device)
        op = torch.empty((n, ns, M), dtype=torch.float32, device=c.device)
        st = torch.cuda.current_stream(c.device).cuda_stream if stream is None else stream
        rc = lib.azmi_symmetries(game_cls.GAME_ID, c.device.index, n, c.data_ptr(), vv.data_ptr(), pp.data_ptr(),
                                 oc.data_ptr(), ov.data_ptr(), op.data_ptr(), 0, C.c_void_p(st))
        if rc:
            raise RuntimeError(lib.azmi_symmetries_last_error().decode())
        return oc, ov, op
    c = np.ascontiguousarray(canonical, np.float32); vv = np.ascontiguousarray(v, np.float32)
    pp = np.ascontiguousarray(pi, np.float32)
    n = c.shape[0]
    if tuple(c.shape[1:]) != tuple(chw) or vv.shape != (n, P + 1) or pp.shape != (n, M):
        raise RuntimeError("Improper sample shapes")
    oc = np.zeros((n, ns) + tuple(chw), np.float32); ov = np.zeros((n, ns, P + 1), np.float32)
    op = np.zeros((n, ns, M), np.float32)
    rc = lib.azmi_symmetries(game_cls.GAME_ID, device, n, c.ctypes.data, vv.ctypes.data, pp.ctypes.data,
                             oc.ctypes.data, ov.ctypes.data, op.ctypes.data, 1, None)
    if rc:
        raise RuntimeError(lib.azmi_symmetries_last_error().decode())
    return oc, ov, op


def tafl_symmetries(board, canonical, v, pi, device=0):
    """eightSym for any square Tafl board (azmi_tafl_symmetries), numpy [n,...] in/out."""
    c = np.ascontiguousarray(canonical, np.float32); vv = np.ascontiguousarray(v, np.float32)
    pp = np.ascontiguousarray(pi, np.float32)
    n, ch = c.shape[0], c.shape[1]
    oc = np.zeros((n, 8) + c.shape[1:], np.float32); ov = np.zeros((n, 8, vv.shape[1]), np.float32)
    op = np.zeros((n, 8, pp.shape[1]), np.float32)
    if c.shape[2:] != (board, board) or pp.shape[1] != board * board * 2 * board:
        raise RuntimeError("Improper sample shapes")
    rc = lib.azmi_tafl_symmetries(board, ch, vv.shape[1], device, n, c.ctypes.data, vv.ctypes.data, pp.ctypes.data,
                                  oc.ctypes.data, ov.ctypes.data, op.ctypes.data, 1, None)
    if rc:
        raise RuntimeError(lib.azmi_symmetries_last_error().decode())
    return oc, ov, op


class GameState:
    """A game position (py_wrapper.cc:157-189).  The object is a start position plus the moves played
    from it; every query (valid_moves, scores, canonicalized, ...) is answered by the device rules
    kernels (azmi_game_replay_from) and memoised until the next play_move()."""

    GAME_ID = -1
    _REPLAY_FLAGS = 0          # bit 0: reference-style unchecked play_move (Brandubh / OpenTafl objects)

    def __init__(self):
        self._init = None      # serialized start position (reference pickle image) or None = initial()
        self._moves = []
        self._snap = None
        self._device = 0

    # ---- statics ---------------------------------------------------------------------------
    @classmethod
    def _info(cls):
        p, m = C.c_uint32(), C.c_uint32()
        chw = (C.c_uint32 * 3)()
        check(lib.azmi_game_info(cls.GAME_ID, C.byref(p), C.byref(m), chw))
        return p.value, m.value, tuple(chw)

    @classmethod
    def NUM_PLAYERS(cls):
        return cls._info()[0]

    @classmethod
    def NUM_MOVES(cls):
        return cls._info()[1]

    @classmethod
    def CANONICAL_SHAPE(cls):
        return cls._info()[2]

    # ---- instance surface ------------------------------------------------------------------
    def _state(self):
        if self._snap is None:
            P, M, chw = self._info()
            mv = np.array([self._moves + [-1]], dtype=np.int32)
            out = dict(valid=np.zeros((1, M), np.uint8), scores=np.zeros((1, P + 1), np.float32),
                       canonical=np.zeros((1,) + tuple(chw), np.float32), player=np.zeros(1, np.uint32),
                       turn=np.zeros(1, np.uint32), key=np.zeros(1, np.uint64), status=np.zeros(1, np.int32))
            init = None if self._init is None else np.frombuffer(self._init, np.uint8)
            check(lib.azmi_game_replay_ex(self.GAME_ID, self._device, None if init is None else init.ctypes.data,
                                          0 if init is None else init.size, mv.ctypes.data, 1, mv.shape[1],
                                          out["valid"].ctypes.data, out["scores"].ctypes.data, out["canonical"].ctypes.data,
                                          out["player"].ctypes.data, out["turn"].ctypes.data, out["key"].ctypes.data,
                                          out["status"].ctypes.data, self._REPLAY_FLAGS))
            if out["status"][0] != 0:
                raise RuntimeError("illegal move in the game record")
            self._snap = out
        return self._snap

    def copy(self):
        g = self.__class__.__new__(self.__class__)
        g.__dict__.update(self.__dict__)
        g._moves = list(self._moves)
        return g

    def __eq__(self, other):
        if not isinstance(other, GameState) or other.GAME_ID != self.GAME_ID:
            return False
        a, b = self._state(), other._state()
        return bool(a["player"][0] == b["player"][0] and a["turn"][0] == b["turn"][0]
                    and np.array_equal(a["canonical"], b["canonical"]))

    __hash__ = None

    def current_turn(self):
        return int(self._state()["turn"][0])

    def current_player(self):
        return int(self._state()["player"][0])

    def num_players(self):
        return self.NUM_PLAYERS()

    def num_moves(self):
        return self.NUM_MOVES()

    def num_symmetries(self):
        return self.NUM_SYMMETRIES()

    def relative_values(self):   # game_state.h:114
        return False

    def randomize_start(self):   # game_state.h:73 — a no-op except for StarGambitUnifiedGS (which re-draws its variant)
        return None

    def num_variants(self):      # game_state.h:76
        return 0

    def get_variant_id(self):    # game_state.h:79
        return -1

    def valid_moves(self):
        return self._state()["valid"][0].copy()

    def play_move(self, move):
        move = int(move)
        if not 0 <= move < self.NUM_MOVES():
            raise RuntimeError(f"move {move} out of range")
        self._moves.append(move)
        self._snap = None

    def scores(self):
        sc = self._state()["scores"][0]
        return None if sc[0] < 0 else sc.copy()

    def canonicalized(self):
        return self._state()["canonical"][0].copy()

    def symmetries(self, base):
        """GameState::symmetries(PlayHistory) -> list of NUM_SYMMETRIES PlayHistory (device gather)."""
        oc, ov, op = symmetries_batch(self.__class__, base._c[None], base._v[None], base._pi[None], device=self._device)
        return [PlayHistory(oc[0, i], ov[0, i], op[0, i]) for i in range(oc.shape[1])]


class Connect4GS(GameState):  # py_wrapper.cc:562-586
    GAME_ID = 0

    def __init__(self, board=None, player=0, turn=0):
        super().__init__()
        if board is not None:
            b = np.ascontiguousarray(board, dtype=np.int8)
            if b.shape != (2, 6, 7):
                raise RuntimeError("Improper connect 4 board shape")
            self._init = b.tobytes() + np.int8(player).tobytes() + np.int32(turn).tobytes()

    @staticmethod
    def NUM_SYMMETRIES():
        return 2

    def play_move(self, move):
        """connect4_gs.cc:48-58: a move into a full column throws at once (the object stays as it was)."""
        super().play_move(move)
        try:
            self._state()
        except RuntimeError:
            self._moves.pop()
            self._snap = None
            raise RuntimeError("Invalid move: You have a bug in your code.") from None

    def to_bytes(self):          # connect4_gs.cc:172-178
        st = self._state()
        board = (st["canonical"][0, :2] != 0).astype(np.int8)
        return board.tobytes() + np.int8(st["player"][0]).tobytes() + np.int32(st["turn"][0]).tobytes()

    @classmethod
    def from_bytes(cls, data):   # connect4_gs.cc:180-190
        if len(data) != 89:
            raise ValueError("Connect4GS::from_bytes: wrong size")
        g = cls()
        g._init = bytes(data)
        return g

    def __reduce__(self):        # ADD_GS_PICKLE, py_wrapper.cc:77-83
        return (_c4_from_bytes, (self.to_bytes(),))

    def __str__(self):           # connect4_gs.cc:192-208
        st = self._state()
        out = "Current Player: %d\n" % st["player"][0]
        for h in range(6):
            out += "".join("X" if st["canonical"][0, 0, h, w] == 1 else "O" if st["canonical"][0, 1, h, w] == 1 else "."
                           for w in range(7)) + "\n"
        return out + "\n"


def _c4_from_bytes(data):
    return Connect4GS.from_bytes(data)


_ENGINE_STREAM = C.c_void_p(-1)  # AZMI_STREAM_ENGINE


class MCTS:
    """The stand-alone search tree (py_wrapper.cc:192-220, mcts.h:50-200) on the device: find_leaf / process_result /
    update_root and the read-outs, call by call; pass game=<GS class> for anything but Connect4.  `seed` seeds the object's pcg32 stream (the
    reference shares one unseedable-from-Python thread_local stream; its C++ tests call MCTS::seed_thread_rng)."""

    def __init__(self, cpuct, num_players, num_moves, epsilon=0.0, root_policy_temp=1.0, fpu_reduction=0.0,
                 relative_values=False, root_fpu_zero=False, shaped_dirichlet=False, gumbel_enabled=False, gumbel_m=16,
                 gumbel_c_visit=50.0, gumbel_c_scale=1.0, gumbel_full=False, *, game=None, seed=None, device=0,
                 max_simulations=0):
        game = Connect4GS if game is None else (game if isinstance(game, type) else type(game))
        cfg = _capi.MctsConfigC(cpuct, num_players, num_moves, epsilon, root_policy_temp, fpu_reduction, int(relative_values),
                                int(root_fpu_zero), int(shaped_dirichlet), int(gumbel_enabled), gumbel_m, gumbel_c_visit,
                                gumbel_c_scale, int(gumbel_full), int(max_simulations))
        if seed is None:
            seed = int.from_bytes(__import__("os").urandom(8), "little")     # std::random_device, mcts.cc:19
        h = C.c_void_p()
        check(lib.azmi_mcts_create(game.GAME_ID, C.byref(cfg), C.c_uint64(seed), int(device), C.byref(h)))
        self._h, self._game, self._P, self._M = h, game, num_players, num_moves
        self._gumbel = bool(gumbel_enabled)
        self._vec = max(num_moves, 64)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.azmi_mcts_destroy(self._h)
            self._h = None

    @staticmethod
    def _gs_args(gs):
        init = None if gs._init is None else np.frombuffer(gs._init, np.uint8)
        mv = np.array(gs._moves, dtype=np.int32)
        return init, mv

    def find_leaf(self, gs):
        init, mv = self._gs_args(gs)
        out = np.zeros(512, np.int32)
        n = C.c_uint32()
        check(lib.azmi_mcts_find_leaf(self._h, None if init is None else init.ctypes.data, 0 if init is None else init.size,
                                      mv.ctypes.data if mv.size else None, mv.size, out.ctypes.data, out.size, C.byref(n)))
        leaf = gs.copy()
        for m in out[: n.value]:
            leaf._moves.append(int(m))
        leaf._snap = None
        return leaf

    def process_result(self, gs, value, pi, root_noise_enabled=False):
        """value is updated in place like the reference's by-reference argument (terminal leaves: the cached scores)."""
        v = np.ascontiguousarray(value, dtype=np.float32); p = np.ascontiguousarray(pi, dtype=np.float32)
        if v.shape != (self._P + 1,) or p.shape != (self._M,):
            raise RuntimeError("process_result: value must be [num_players + 1] and pi [num_moves]")
        out = np.zeros(self._P + 1, np.float32)
        check(lib.azmi_mcts_process_result(self._h, v.ctypes.data, p.ctypes.data, int(bool(root_noise_enabled)), out.ctypes.data))
        if isinstance(value, np.ndarray) and value.dtype == np.float32:
            value[...] = out
        return out

    def update_root(self, gs, move):
        init, mv = self._gs_args(gs)
        check(lib.azmi_mcts_update_root(self._h, None if init is None else init.ctypes.data, 0 if init is None else init.size,
                                        mv.ctypes.data if mv.size else None, mv.size, int(move)))

    def _query(self, kind, temp=0.0, arg=0, in_f=None):
        f = np.zeros(self._vec, np.float32); u = np.zeros(self._vec + 64, np.uint32)
        inp = None if in_f is None else np.ascontiguousarray(in_f, dtype=np.float32)
        check(lib.azmi_mcts_query(self._h, kind, float(temp), int(arg), None if inp is None else inp.ctypes.data, f.ctypes.data, u.ctypes.data))
        return f, u

    def counts(self): return self._query(0)[1][: self._M].copy()
    def probs(self, temp): return self._query(1, temp)[0][: self._M].copy()
    def probs_pruned(self, temp): return self._query(2, temp)[0][: self._M].copy()
    def root_value(self): return self._query(3)[0][:3].copy()
    def root_q_values(self): return self._query(4)[0][: self._M].copy()
    def depth(self): return int(self._query(5)[1][0])
    def root_n(self): return int(self._query(5)[1][1])
    def avg_leaf_depth(self): return float(self._query(5)[0][0])
    def normalized_root_entropy(self): return float(self._query(5)[0][1])
    def gumbel_enabled(self): return self._gumbel
    def gumbel_improved_policy(self): return self._query(6)[0][: self._M].copy()
    def gumbel_final_action(self): return int(self._query(7)[1][0])
    def add_root_noise(self): self._query(8)
    def apply_root_policy_temp(self): self._query(9)
    def set_gumbel_num_sims(self, n): self._query(12, arg=n)

    def principal_variation(self, depth=5):
        u = self._query(11, arg=depth)[1]
        return u[1: 1 + int(u[0])].copy()

    def pick_move(self, pi):
        """MCTS::pick_move (static in the reference, drawing from the thread's stream): draws from this object's stream."""
        return int(self._query(10, in_f=pi)[1][0])

    # ---- WU-UCT batched API (py_wrapper.cc:212-215, mcts.cc:752-851) ----
    def find_leaf_batched(self, gs):
        init, mv = self._gs_args(gs)
        out = np.zeros(512, np.int32)
        n = C.c_uint32()
        check(lib.azmi_mcts_find_leaf_batched(self._h, None if init is None else init.ctypes.data, 0 if init is None else init.size,
                                              mv.ctypes.data if mv.size else None, mv.size, out.ctypes.data, out.size, C.byref(n)))
        leaf = gs.copy()
        for m in out[: n.value]:
            leaf._moves.append(int(m))
        leaf._snap = None
        return leaf

    def process_result_batched(self, gs, leaf_index, value, pi, root_noise_enabled=False):
        v = np.ascontiguousarray(value, dtype=np.float32); p = np.ascontiguousarray(pi, dtype=np.float32)
        if v.shape != (self._P + 1,) or p.shape != (self._M,):
            raise RuntimeError("process_result_batched: value must be [num_players + 1] and pi [num_moves]")
        out = np.zeros(self._P + 1, np.float32)
        check(lib.azmi_mcts_process_result_batched(self._h, int(leaf_index), v.ctypes.data, p.ctypes.data, int(bool(root_noise_enabled)),
                                                   out.ctypes.data))
        if isinstance(value, np.ndarray) and value.dtype == np.float32:
            value[...] = out
        return out

    def in_flight_count(self):
        n = C.c_uint32()
        check(lib.azmi_mcts_in_flight_count(self._h, C.byref(n)))
        return n.value

    def reset_batch(self):
        check(lib.azmi_mcts_reset_batch(self._h))


def dumb_eval(gs):
    """game_state.h:160-173: uniform policy over the legal moves (u8 sum wraps), value 1/(P+1) each."""
    valids = gs.valid_moves()
    P = gs.num_players()
    v = np.full(P + 1, np.float32(1.0 / (P + 1)), np.float32)
    ssum = np.float32(int(valids.sum()) & 0xFF)
    pi = np.zeros(valids.size, np.float32) if ssum == 0 else (valids.astype(np.float32) / ssum)
    return v, pi


def playout_eval_batch(states, seeds=None):
    """playout_eval_batch(list of GameState) -> (values, policies), py_wrapper.cc:748-770 / game_state.cc:62-95: uniform policy
    over the legal moves and the scores of one uniformly random rollout per state, all states in one kernel launch.  The
    reference seeds its rollout engines from std::random_device; `seeds` (one 64-bit seed per state) makes a run repeatable."""
    states = list(states)
    if not states:
        return [], []
    cls = type(states[0])
    if any(type(g) is not cls for g in states):
        raise RuntimeError("playout_eval_batch: all states must be of the same game")
    P, M, _ = cls._info()
    n = len(states)
    if seeds is None:
        seeds = np.frombuffer(os.urandom(8 * n), dtype=np.uint64)
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    if seeds.shape != (n,):
        raise RuntimeError("playout_eval_batch: one seed per state")
    length = max(len(g._moves) for g in states) + 1
    mv = np.full((n, length), -1, np.int32)
    for i, g in enumerate(states):
        mv[i, : len(g._moves)] = g._moves
    init = None
    if any(g._init is not None for g in states):
        blank = cls().to_bytes() if hasattr(cls, "to_bytes") else None
        rows = [g._init if g._init is not None else blank for g in states]
        if any(r is None for r in rows):
            raise RuntimeError("playout_eval_batch: this game has no serialized start position")
        stride = max(len(r) for r in rows)
        init = np.frombuffer(b"".join(r + bytes(stride - len(r)) for r in rows), np.uint8)
    v = np.zeros((n, P + 1), np.float32); pi = np.zeros((n, M), np.float32)
    check(lib.azmi_playout_eval(cls.GAME_ID, states[0]._device, None if init is None else init.ctypes.data,
                                0 if init is None else init.size // n, mv.ctypes.data, n, length, seeds.ctypes.data,
                                v.ctypes.data, pi.ctypes.data))
    return [v[i].copy() for i in range(n)], [pi[i].copy() for i in range(n)]


def playout_eval(gs, seed=None):
    """playout_eval(gs) -> (value, pi), py_wrapper.cc:726-746"""
    v, pi = playout_eval_batch([gs], None if seed is None else [seed])
    return v[0], pi[0]


class _TaflBoardGS(GameState):
    """Tafl-family objects.  A start position other than the game's own is the REFERENCE's pickle image
    (tawlbwrdd_gs.cc:10-37, brandubh_gs.cc:11-41, opentafl_gs.cc:13-40):
        board int8[3][N][N] (king, defenders, attackers) | u16 turn | u16 max_turns | i8 player | u8 current_repetition_count |
        u32 n | n x (board | u8 player | u8 count)                       (little endian)
    so a state pickled by the reference loads here and the other way round, repetition map included."""
    BOARD = 0
    MAX_TURNS = 0
    _REPLAY_FLAGS = 1          # play_move like the reference: no ownership / slide check (valid_moves is the validator)

    def __init__(self, max_turns=None):
        super().__init__()
        if max_turns is not None and max_turns != self.MAX_TURNS:
            raise RuntimeError(f"the MI355X engine implements {type(self).__name__} with the default max_turns = {self.MAX_TURNS}")

    @classmethod
    def _image(cls, board, player, turn, rep_count=1, entries=()):
        b = np.ascontiguousarray(board, dtype=np.int8)
        if b.shape != (3, cls.BOARD, cls.BOARD):
            raise RuntimeError("Improper tafl board shape")
        out = [b.tobytes(), np.uint16(turn).tobytes(), np.uint16(cls.MAX_TURNS).tobytes(), np.int8(player).tobytes(),
               np.uint8(rep_count).tobytes(), np.uint32(len(entries)).tobytes()]
        for eb, ep, ec in entries:
            out += [eb, np.uint8(ep).tobytes(), np.uint8(ec).tobytes()]
        return b"".join(out)

    @classmethod
    def from_board(cls, board, player, turn=0):
        """the C++ test helper MakeGS (opentafl_gs_test.cc:97-101): a board with an empty repetition map, count 1"""
        g = cls()
        g._init = cls._image(board, player, turn)
        return g

    @classmethod
    def from_bytes(cls, data):
        data = bytes(data)
        bb = 3 * cls.BOARD * cls.BOARD
        if len(data) < bb + 10:
            raise RuntimeError(f"{cls.__name__}::from_bytes: data too short")
        n = int(np.frombuffer(data[bb + 6: bb + 10], np.uint32)[0])
        if bb + 10 + n * (bb + 2) != len(data):
            raise RuntimeError(f"{cls.__name__}::from_bytes: repetition entry count mismatch")
        if int(np.frombuffer(data[bb + 2: bb + 4], np.uint16)[0]) != cls.MAX_TURNS:
            raise RuntimeError(f"the MI355X engine implements {cls.__name__} with the default max_turns = {cls.MAX_TURNS}")
        g = cls()
        g._init = data
        return g

    def to_bytes(self):
        """The reference's to_bytes image of the current position.  The repetition map holds boards, the device keeps
        64-bit keys, so the boards are read back by replaying every prefix of the game record (one batched device call)
        and applying the reference's bookkeeping: the start position is interned by the first move of a game, a capture
        clears the map, every move counts the position it reaches (tawlbwrdd_gs.cc:253-259, 286-331)."""
        if not self._moves and self._init is not None:
            return self._init
        P, M, chw = self._info()
        L = len(self._moves)
        mv = np.full((L + 1, L + 1), -1, np.int32)
        for i in range(L + 1):
            mv[i, :i] = self._moves[:i]
        canon = np.zeros((L + 1,) + tuple(chw), np.float32)
        player = np.zeros(L + 1, np.uint32); turn = np.zeros(L + 1, np.uint32); status = np.zeros(L + 1, np.int32)
        init = None if self._init is None else np.frombuffer(self._init * (L + 1), np.uint8)
        check(lib.azmi_game_replay_ex(self.GAME_ID, self._device, None if init is None else init.ctypes.data,
                                      0 if init is None else len(self._init), mv.ctypes.data, L + 1, L + 1,
                                      None, None, canon.ctypes.data, player.ctypes.data, turn.ctypes.data, None, status.ctypes.data,
                                      self._REPLAY_FLAGS))
        if status.any():
            raise RuntimeError("illegal move in the game record")
        boards = (canon[:, :3] != 0).astype(np.int8)
        pieces = boards.reshape(L + 1, -1).sum(1)
        bb = 3 * self.BOARD * self.BOARD
        reps = []                                   # (board bytes, player) in insertion order, with multiplicity
        rep_count = 1
        if self._init is not None:
            rep_count = self._init[bb + 5]
            n = int(np.frombuffer(self._init[bb + 6: bb + 10], np.uint32)[0])
            for e in range(n):
                o = bb + 10 + e * (bb + 2)
                reps += [(self._init[o: o + bb], self._init[o + bb])] * self._init[o + bb + 1]
        for i in range(1, L + 1):
            if turn[i - 1] == 0:
                reps = [(boards[i - 1].tobytes(), int(player[i - 1]))]
            if pieces[i] < pieces[i - 1]:
                reps = []
            key = (boards[i].tobytes(), int(player[i]))
            reps.append(key)
            rep_count = reps.count(key)
        entries, seen = [], {}
        for key in reps:
            if key in seen:
                entries[seen[key]][2] += 1
            else:
                seen[key] = len(entries)
                entries.append([key[0], key[1], 1])
        return self._image(boards[L], int(player[L]), int(turn[L]), rep_count, entries)

    def __reduce__(self):        # ADD_GS_PICKLE, py_wrapper.cc:77-83
        return (_tafl_from_bytes, (type(self).__name__, self.to_bytes()))

    @staticmethod
    def NUM_SYMMETRIES():
        return 8

    @classmethod
    def POLICY_SHAPE(cls):
        return (2 * cls.BOARD, cls.BOARD, cls.BOARD)

    def _rep_count(self):
        st = self._state()
        c = st["canonical"][0]
        return 3 if (c[5, 0, 0] and c[6, 0, 0] and st["turn"][0] > 0) else 2 if c[6, 0, 0] else 1 if c[5, 0, 0] else 0

    def __str__(self):           # brandubh_gs.cc:547-581 / opentafl_gs.cc:589-623 without the colour escapes
        st = self._state()
        c = st["canonical"][0]
        out = "Current Player: %d\nCurrent Turn: %d out of %d\n" % (st["player"][0], st["turn"][0], self.MAX_TURNS)
        for h in range(self.BOARD):
            out += "".join("@" if c[0, h, w] == 1 else "O" if c[1, h, w] == 1 else "X" if c[2, h, w] == 1 else "."
                           for w in range(self.BOARD)) + "\n"
        return out + "\n"


def _tafl_from_bytes(name, data):
    return {"TawlbwrddGS": TawlbwrddGS, "BrandubhGS": BrandubhGS, "OpenTaflGS": OpenTaflGS}[name].from_bytes(data)


class TawlbwrddGS(_TaflBoardGS):  # py_wrapper.cc:549-560
    GAME_ID = 1
    BOARD = 11
    MAX_TURNS = 400
    _REPLAY_FLAGS = 0

    def __str__(self):           # tawlbwrdd_gs.cc:460-484
        st = self._state()
        c = st["canonical"][0]
        out = "Current Player: %d\nCurrent Turn: %d out of 400\nCurrent Repetition Count: %d\n" % (st["player"][0], st["turn"][0], self._rep_count())
        for h in range(11):
            out += "".join("@" if c[0, h, w] == 1 else "O" if c[1, h, w] == 1 else "X" if c[2, h, w] == 1 else "."
                           for w in range(11)) + "\n"
        return out + "\n"


class BrandubhGS(_TaflBoardGS):  # py_wrapper.cc:527-536
    GAME_ID = 2
    BOARD = 7
    MAX_TURNS = 150


class OpenTaflGS(_TaflBoardGS):  # py_wrapper.cc:538-547
    GAME_ID = 3
    BOARD = 11
    MAX_TURNS = 400


class UnitInfo:
    """UnitInfo of get_units() (star_gambit_gs.h:424-433, py_wrapper.cc StarGambit bindings)."""
    __slots__ = ("player", "type", "slot", "hp", "anchor_q", "anchor_r", "facing", "moves_left")

    def __init__(self, player, type, slot, hp, anchor_q, anchor_r, facing, moves_left):
        self.player, self.type, self.slot, self.hp = player, type, slot, hp
        self.anchor_q, self.anchor_r, self.facing, self.moves_left = anchor_q, anchor_r, facing, moves_left

    def __repr__(self):
        return (f"UnitInfo(player={self.player}, type={self.type}, slot={self.slot}, hp={self.hp}, "
                f"anchor=({self.anchor_q}, {self.anchor_r}), facing={self.facing}, moves_left={self.moves_left})")


class StarGambitUnifiedGS(GameState):
    """StarGambitUnifiedGS(pinned_variant=-1, probs=[.25]*4) (py_wrapper.cc:662-667; star_gambit_gs.h:788-887): the four
    configurations on the 13 x 13 canvas.  The object is the reference's to_bytes image of a start position plus the moves
    played from it; the device rules (csrc/dev_stargambit.h, one wavefront per object) answer every query.

    Build-defined: the reference draws the variant of an unpinned game from an unseedable thread-local engine
    (star_gambit_gs.cc:2357-2362); here the constructor / randomize_start() draw it from Python's `random` module, and a
    PlayManager draws every game's variant from the slot's coin stream."""

    GAME_ID = 4
    _REPLAY_FLAGS = 1          # play_move does not validate, like the reference's (star_gambit_gs.cc:1093-1238)
    _VARIANT_NAMES = ("Skirmish", "Showdown", "Clash", "Battle")

    def __init__(self, pinned_variant=-1, probs=(0.25, 0.25, 0.25, 0.25)):
        super().__init__()
        self._pinned = int(pinned_variant)
        self._probs = tuple(float(x) for x in probs)
        if len(self._probs) != 4:
            raise TypeError("probs must have four entries")
        self._start_variant(self._pick_variant())

    def _pick_variant(self):
        if 0 <= self._pinned <= 3:
            return self._pinned
        import random
        return random.choices(range(4), weights=self._probs)[0]

    def _start_variant(self, v):
        """the start position of variant v: two portals, full reserves, the position seen once (star_gambit_gs.cc:251-290)"""
        side = 6 if v == 3 else 5
        start = ((3, 1, 0), (4, 0, 1), (3, 2, 1), (4, 3, 2))[v]
        units = bytes([3, 0, 0, 5, 2, 0, side, 0, 0]) + bytes([3, 1, 0, 5, 5, 0, (-side) & 0xFF, 0, 0])

        def uh(t, pl, hp, f, q, r):
            return ((t ^ (pl << 8) ^ (hp << 12) ^ (f << 20) ^ ((q + 10) << 28) ^ ((r + 10) << 36)) * 0x517cc1b727220a95) & (2**64 - 1)
        h0 = uh(3, 0, 5, 2, 0, side) ^ uh(3, 1, 5, 5, 0, -side)     # compute_position_hash with player 0 to move, :1365-1382
        inner = (struct.pack("<I", 2) + units + bytes(start) + b"\0" + bytes(start) + b"\0" + struct.pack("<BIBBbI", 0, 1, 0, 0, -1, 1)
                 + struct.pack("<Q", h0))
        self._init = self._header(v, len(inner)) + inner
        self._moves = []
        self._snap = None
        self._img = None

    def _header(self, variant, inner_size):
        return struct.pack("<4fiBI", *self._probs, self._pinned, variant, inner_size)

    @staticmethod
    def NUM_SYMMETRIES():
        return 2

    @classmethod
    def POLICY_SHAPE(cls):         # star_gambit_gs.h:818-819; neural_net.py:281 keys the spatial policy head on it
        return (10, 13, 13)

    def relative_values(self): return True           # star_gambit_gs.h:845
    def num_variants(self): return 4                 # :863
    def get_variant_id(self): return self._image()[20]

    def randomize_start(self):                       # star_gambit_gs.cc:2421-2425
        self._start_variant(self._pick_variant())

    def _image(self):
        """to_bytes image of the current position (device: azmi_sg_image); cached per move count (MCTS.find_leaf appends moves)"""
        cached = getattr(self, "_img", None)
        if cached is not None and cached[0] == len(self._moves):
            return cached[1]
        self._img = (len(self._moves), self._make_image())
        return self._img[1]

    def _make_image(self):
        if True:
            if not self._moves:
                return bytes(self._init)
            else:
                mv = np.array([self._moves + [-1]], dtype=np.int32)
                init = np.frombuffer(self._init, np.uint8)
                cap = len(self._init) + 9 * 20 + 8 * (len(self._moves) + 4) + 64
                out = np.zeros(cap, np.uint8); n = np.zeros(1, np.uint32); st = np.zeros(1, np.int32)
                check(lib.azmi_sg_image(self._device, init.ctypes.data, init.size, mv.ctypes.data, 1, mv.shape[1], out.ctypes.data, cap,
                                        n.ctypes.data, st.ctypes.data, self._REPLAY_FLAGS))
                if st[0] != 0:
                    raise RuntimeError("illegal move in the game record")
                b = bytearray(out[: int(n[0])].tobytes())
                b[:20] = struct.pack("<4fi", *self._probs, self._pinned)
                return bytes(b)

    def to_bytes(self):                              # star_gambit_gs.cc:2451-2465
        return self._image()

    @classmethod
    def from_bytes(cls, data):                       # star_gambit_gs.cc:2467-2516
        data = bytes(data)
        if len(data) < 25:
            raise RuntimeError("StarGambitUnifiedGS::from_bytes: short data")
        probs = struct.unpack_from("<4f", data, 0)
        pinned, variant, inner_size = struct.unpack_from("<iBI", data, 16)
        if variant > 3:
            raise RuntimeError("StarGambitUnifiedGS::from_bytes: bad variant_id")
        if 25 + inner_size > len(data):
            raise RuntimeError("StarGambitUnifiedGS::from_bytes: short inner")
        if 25 + inner_size != len(data):
            raise RuntimeError("StarGambitUnifiedGS::from_bytes: trailing bytes")
        g = cls.__new__(cls)
        GameState.__init__(g)
        g._pinned, g._probs = int(pinned), tuple(float(x) for x in probs)
        g._init, g._moves, g._snap, g._img = data, [], None, None
        g._state()        # the device parses the inner image: malformed images raise here
        return g

    def __reduce__(self):                            # ADD_GS_PICKLE, py_wrapper.cc:77-83
        return (_sg_from_bytes, (type(self).__name__, self.to_bytes()))

    def _fields(self):
        b = self._image()
        n = struct.unpack_from("<I", b, 25)[0]
        units = [struct.unpack_from("<5B2b2B", b, 29 + 9 * i) for i in range(n)]
        off = 29 + 9 * n
        reserves = list(b[off:off + 8])
        player, turn, acted, over, winner, hist = struct.unpack_from("<BIBBbI", b, off + 8)
        return units, reserves, player, turn, bool(acted), bool(over), winner, hist

    def get_units(self):                             # alive units only, star_gambit_gs.cc:2102-2118
        return [UnitInfo(u[1], u[0], u[2], u[3], u[5], u[6], u[4], u[7]) for u in self._fields()[0] if u[3] > 0]

    def has_taken_action(self):                      # star_gambit_gs.h:654
        return self._fields()[4]

    def __eq__(self, other):                         # operator==, star_gambit_gs.cc:297-319, 2392-2397: turn and history take no part
        if not isinstance(other, StarGambitUnifiedGS):
            return False
        a, b = self._fields(), other._fields()
        return self.get_variant_id() == other.get_variant_id() and a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and a[4] == b[4]

    __hash__ = None

    def dump(self):                                  # star_gambit_gs.cc:1807-1830 (header lines)
        units, res, player, turn, acted, over, winner, _ = self._fields()
        return (f"Turn: {turn}, Player: {player}{' (acted)' if acted else ''}\n"
                f"P0 reserves: F={res[0]} C={res[1]} D={res[2]}\nP1 reserves: F={res[4]} C={res[5]} D={res[6]}\n"
                + "".join(f"P{u[1]} {'FCDP'[u[0]]}{u[2] + 1} hp={u[3]} at ({u[5]}, {u[6]}) facing {u[4]}\n" for u in units if u[3] > 0))

    __str__ = dump

    # symmetries(PlayHistory): GameState.symmetries -> azmi_symmetries(AZMI_GAME_STARGAMBIT): {base, NW-axis mirror},
    # star_gambit_gs.cc:2623-2727


def _pinned_sg(name, variant):
    def __init__(self):
        StarGambitUnifiedGS.__init__(self, variant)
    return type(name, (StarGambitUnifiedGS,), {"__init__": __init__, "__doc__": f"StarGambitUnifiedGS pinned to {name[17:-2]} (star_gambit_gs.h:893-911)"})


StarGambitUnifiedSkirmishGS = _pinned_sg("StarGambitUnifiedSkirmishGS", 0)
StarGambitUnifiedShowdownGS = _pinned_sg("StarGambitUnifiedShowdownGS", 1)
StarGambitUnifiedClashGS = _pinned_sg("StarGambitUnifiedClashGS", 2)
StarGambitUnifiedBattleGS = _pinned_sg("StarGambitUnifiedBattleGS", 3)


class _StarGambitPlainGS:
    """StarGambit{Skirmish,Showdown,Clash,Battle}GS (star_gambit_gs.h:752-755): one configuration in its OWN action space
    (ActionSpace<Config>, :483-590) and its own 32-plane canonical form.  A view of the unified device game: actions map
    through to_unified_action (star_gambit_gs.cc:2522-2554), the planes are the 32 x D x D window of the canvas."""

    VARIANT = 0

    def __init__(self):
        self._u = StarGambitUnifiedGS(self.VARIANT)

    @classmethod
    def _dims(cls):
        side = 6 if cls.VARIANT == 3 else 5
        d = 2 * side + 1
        return side, d, d * d * 10

    @classmethod
    def NUM_PLAYERS(cls): return 2
    @classmethod
    def NUM_MOVES(cls): return cls._dims()[2] + 19
    @classmethod
    def CANONICAL_SHAPE(cls): return (32, cls._dims()[1], cls._dims()[1])
    @classmethod
    def POLICY_SHAPE(cls): return (10, cls._dims()[1], cls._dims()[1])
    @staticmethod
    def NUM_SYMMETRIES(): return 2

    def _to_unified(self, a):
        side, d, sp = self._dims()
        if self.VARIANT == 3:
            return a
        if a < sp:
            return ((a // 10 // d + 1) * 13 + (a // 10 % d + 1)) * 10 + a % 10
        return 1690 + (a - sp)

    def copy(self):
        g = self.__class__.__new__(self.__class__)
        g._u = self._u.copy()
        return g

    def __eq__(self, other): return isinstance(other, _StarGambitPlainGS) and self._u == other._u
    __hash__ = None
    def current_player(self): return self._u.current_player()
    def current_turn(self): return self._u.current_turn()
    def num_players(self): return 2
    def num_moves(self): return self.NUM_MOVES()
    def num_symmetries(self): return 2
    def relative_values(self): return True           # star_gambit_gs.h:628
    def randomize_start(self): return None
    def num_variants(self): return 0
    def get_variant_id(self): return -1
    def scores(self): return self._u.scores()
    def get_units(self): return self._u.get_units()
    def has_taken_action(self): return self._u.has_taken_action()
    def dump(self): return self._u.dump()
    __str__ = dump

    def valid_moves(self):
        side, d, sp = self._dims()
        uv = self._u.valid_moves()
        if self.VARIANT == 3:
            return uv
        out = np.zeros(sp + 19, np.uint8)
        out[:sp] = uv[:1690].reshape(13, 13, 10)[1:1 + d, 1:1 + d].reshape(-1)
        out[sp:] = uv[1690:]
        return out

    def play_move(self, move):
        move = int(move)
        if not 0 <= move < self.NUM_MOVES():
            raise RuntimeError(f"move {move} out of range")
        self._u.play_move(self._to_unified(move))

    def canonicalized(self):
        side, d, _ = self._dims()
        off = 0 if self.VARIANT == 3 else 1
        return self._u.canonicalized()[:32, off:off + d, off:off + d].copy()

    def to_bytes(self):                              # the inner image, star_gambit_gs.cc:2253-2288
        return self._u.to_bytes()[25:]

    @classmethod
    def from_bytes(cls, data):
        data = bytes(data)
        g = cls.__new__(cls)
        g._u = StarGambitUnifiedGS.from_bytes(struct.pack("<4fiBI", 0.25, 0.25, 0.25, 0.25, cls.VARIANT, cls.VARIANT, len(data)) + data)
        return g

    def __reduce__(self):
        return (_sg_from_bytes, (type(self).__name__, self.to_bytes()))


class StarGambitSkirmishGS(_StarGambitPlainGS): VARIANT = 0
class StarGambitShowdownGS(_StarGambitPlainGS): VARIANT = 1
class StarGambitClashGS(_StarGambitPlainGS): VARIANT = 2
class StarGambitBattleGS(_StarGambitPlainGS): VARIANT = 3
StarGambitGS = StarGambitSkirmishGS                   # the legacy name, star_gambit_gs.h:758


def _sg_from_bytes(name, data):
    return globals()[name].from_bytes(data)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class GameData:
    """GameData (py_wrapper.cc:265-288) of one slot: gs(), valid_moves(), canonical() read the device state; v() and pi()
    are the host rows push_inference(i) hands to the engine."""

    def __init__(self, pm, i):
        self._pm, self._i = pm, i
        self._v = np.zeros(pm._P + 1, np.float32)
        self._pi = np.zeros(pm._M, np.float32)

    def v(self): return self._v
    def pi(self): return self._pi

    def canonical(self):
        out = np.zeros(tuple(self._pm._chw), np.float32)
        check(lib.azmi_pm_slot_canonical(self._pm._h, self._i, out.ctypes.data))
        return out

    @property
    def perm_index(self):
        """GameData::perm_index (play_manager.h:41): the seat permutation this slot's game runs under."""
        w = np.zeros(8, np.uint64); n = C.c_uint32()
        check(lib.azmi_pm_slot_state(self._pm._h, self._i, w.ctypes.data, 8, C.byref(n)))
        return int(w[n.value - 1])

    def gs(self):
        pm = self._pm
        game = pm._game
        if issubclass(game, StarGambitUnifiedGS):     # the to_bytes image from the packed units, the scalars and the position history
            w = np.zeros(16, np.uint64); n = C.c_uint32()
            check(lib.azmi_pm_slot_state(pm._h, self._i, w.ctypes.data, 16, C.byref(n)))
            hist = np.zeros(4100, np.uint64); hn = C.c_uint32()
            check(lib.azmi_pm_slot_history(pm._h, self._i, hist.ctypes.data, hist.size, C.byref(hn)))
            sc = int(w[10]); misc = (sc >> 24) & 0xFFFF; res = (sc >> 40) & 0x3FFFF
            units = b""
            for i in range(misc & 31):
                u = (int(w[i // 2]) >> (32 * (i % 2))) & 0xFFFFFFFF
                units += struct.pack("<5B2b2B", u & 3, (u >> 2) & 1, (u >> 3) & 7, (u >> 6) & 7, (u >> 9) & 7, ((u >> 12) & 15) - 6,
                                     ((u >> 16) & 15) - 6, (u >> 20) & 3, (u >> 22) & 15)
            reserves = bytes([(res >> (3 * (p * 3 + t))) & 7 if t < 3 else 0 for p in range(2) for t in range(4)])
            winner = (misc >> 7) & 3
            inner = (struct.pack("<I", misc & 31) + units + reserves + struct.pack("<BIBBbI", sc & 1, (sc >> 8) & 0xFFFF, (misc >> 5) & 1,
                                                                                    (misc >> 6) & 1, winner if winner < 3 else -1, hn.value)
                     + hist[: hn.value].tobytes())
            base = pm._sg_base
            return game.from_bytes(struct.pack("<4fiBI", *base[1], base[0], (misc >> 9) & 3, len(inner)) + inner)
        w = np.zeros(8, np.uint64); n = C.c_uint32()
        check(lib.azmi_pm_slot_state(pm._h, self._i, w.ctypes.data, 8, C.byref(n)))
        if game is Connect4GS:
            board = np.zeros((2, 6, 7), np.int8)
            for p in range(2):
                bits = int(w[p])
                for c in range(42):
                    board[p].flat[c] = (bits >> c) & 1
            return Connect4GS(board, int(w[2]) >> 32, int(w[2]) & 0xFFFFFFFF)
        if issubclass(game, _TaflBoardGS):
            sq = game.BOARD * game.BOARD
            d = int(w[0]) | (int(w[1]) << 64); a = int(w[2]) | (int(w[3]) << 64)
            king = int(w[4]) & 0xFF
            board = np.zeros((3, game.BOARD, game.BOARD), np.int8)
            for c in range(sq):
                board[1].flat[c] = (d >> c) & 1
                board[2].flat[c] = (a >> c) & 1
            if king < sq:
                board[0].flat[king] = 1
            return game.from_board(board, (int(w[4]) >> 24) & 0xFF, (int(w[4]) >> 8) & 0xFFFF)
        raise RuntimeError("game_data(i).gs() is not available for this game (its state carries a repetition history)")

    def valid_moves(self):
        return self.gs().valid_moves()


class PlayManager:
    """py_wrapper.cc:351-504 over libazmi.

    Reference methods: play, build_batch, update_inferences, build_history_batch, scores,
    resign_scores, games_completed, remaining_games, params, avg_* getters, hist_count.
    Device fast path (no reference counterpart): round(), io_tensors(), poll().
    """

    def __init__(self, gs, params, caches=None, seed=None, device=0, max_inline=0, log_moves=False, history_capacity=0):
        if gs is None:
            raise TypeError("PlayManager(): gs must not be None")  # py::arg().none(false)
        game_id = gs.GAME_ID
        is_sg = (gs if isinstance(gs, type) else type(gs)).GAME_ID == StarGambitUnifiedGS.GAME_ID
        if not isinstance(gs, type) and not is_sg and (gs._moves or gs._init is not None):
            raise RuntimeError("the MI355X PlayManager starts every game from the game's initial position")
        nplayers = type(gs).NUM_PLAYERS() if not isinstance(gs, type) else gs.NUM_PLAYERS()
        self._game = gs if isinstance(gs, type) else type(gs)
        self._params = params
        cparams = params._to_c(nplayers)
        opts = _capi.EngineOptsC()
        lib.azmi_engine_opts_default(C.byref(opts))
        if seed is not None:
            opts.seed = int(seed)
        opts.device = int(device)
        opts.max_inline = int(max_inline)
        opts.log_moves = int(bool(log_moves))
        opts.history_capacity = int(history_capacity)
        if is_sg:   # every game re-draws its variant (randomize_start): only the base game's constructor arguments matter
            base = gs() if isinstance(gs, type) else gs
            self._sg_base = (base._pinned, base._probs)
            opts.sg_pinned_variant = base._pinned
            for i in range(4):
                opts.sg_variant_probs[i] = base._probs[i]
            if not any(base._probs):
                raise RuntimeError("StarGambitUnifiedGS: the variant weights are all zero")
        h = C.c_void_p()
        self._caches = None
        if caches is None:
            check(lib.azmi_pm_create(game_id, C.byref(cparams), C.byref(opts), C.byref(h)))
        else:   # PlayManager(gs, params, caches): py_wrapper.cc:355-360, one (possibly None) cache per model group
            self._caches = list(caches)     # the engine borrows them: keep them alive as long as the engine
            arr = (C.c_void_p * max(1, len(self._caches)))()
            for i, c in enumerate(self._caches):
                if c is not None and not isinstance(c, ShardedS3FIFOCache):
                    raise TypeError("caches must hold ShardedS3FIFOCache objects or None")
                if c is not None:
                    c._ensure_engine_layout()
                arr[i] = None if c is None else c._h
            check(lib.azmi_pm_create_with_caches(game_id, C.byref(cparams), C.byref(opts), arr, len(self._caches), C.byref(h)))
        self._h = h
        self._eager = False
        self._slot_rows = {}
        self._P, self._M, self._chw = self._game._info()
        self._S = int(params.concurrent_games)
        self._last_stream = _ENGINE_STREAM

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:   # `lib` is already None while the interpreter shuts down
            lib.azmi_pm_destroy(h)
            self._h = None

    # ---- reference surface --------------------------------------------------------------
    def params(self):
        return self._params

    def play(self):
        """PlayManager::play (play_manager.cc:258-600).  Engines whose seats need no net (RANDOM / PLAYOUT) run to completion
        here; any number of threads may call it at once, like the reference's mcts_workers.  With NN seats the engine is
        driven by whoever calls build_batch / update_inferences (the reference's batcher threads), so a worker thread that
        calls play() just waits for the games to finish or for stop(), as the reference's workers do when they find
        their queue empty."""
        rc = lib.azmi_pm_play(self._h, _ENGINE_STREAM)
        if rc == -5:                                       # AZMI_ERR_STATE: NN seats
            import time
            while self.remaining_games() > 0 and not self.stopped():
                time.sleep(0.0005)
            return
        check(rc)

    def games_completed(self):
        done, live = C.c_uint32(), C.c_uint32()
        check(lib.azmi_pm_poll(self._h, self._stream_arg(None), C.byref(done), C.byref(live)))
        return done.value

    def stop(self):                                        # play_manager.h:177
        check(lib.azmi_pm_stop(self._h))

    def stopped(self):                                     # play_manager.h:178
        f = C.c_int()
        check(lib.azmi_pm_stopped(self._h, C.byref(f)))
        return bool(f.value)

    def set_eager(self, e):
        """play_manager.h:277: the reference's batcher hands over partial batches while this is set; the engine's
        build_batch never waits for a batch to fill, so the flag is only remembered."""
        self._eager = bool(e)

    def _queue_counts(self):
        a, b = C.c_uint32(), C.c_uint32()
        check(lib.azmi_pm_queue_counts(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def awaiting_inference_count(self): return self._queue_counts()[0]    # play_manager.h:317-323
    def awaiting_mcts_count(self): return self._queue_counts()[1]         # play_manager.h:316

    def remaining_games(self):
        if self.stopped():                                 # play_manager.h:179-182
            return 0
        done, live = C.c_uint32(), C.c_uint32()
        check(lib.azmi_pm_poll(self._h, self._stream_arg(None), C.byref(done), C.byref(live)))
        if live.value == 0:
            return 0
        return max(0, int(self._params.games_to_play) - done.value)

    def scores(self):
        out = np.zeros(self._P + 1, np.float32)
        check(lib.azmi_pm_scores(self._h, out.ctypes.data))
        return out

    def resign_scores(self):
        out = np.zeros(self._P + 1, np.float32)
        check(lib.azmi_pm_resign_scores(self._h, out.ctypes.data))
        return out

    def _stats(self):
        out = np.zeros(7, np.float32)
        check(lib.azmi_pm_stats(self._h, out.ctypes.data))
        return out

    def stat_sums(self):
        """The accumulators behind the avg_* figures (play_manager.h:398-424), for exact aggregation over several engines."""
        out = np.zeros(10, np.float64)
        check(lib.azmi_pm_stat_sums(self._h, out.ctypes.data))
        return dict(zip(("game_length", "games", "moves", "full_moves", "fast_moves", "leaf_depth", "entropy", "fast_leaf_depth",
                         "fast_entropy", "valid_moves"), (float(x) for x in out)))

    def avg_game_length(self): return float(self._stats()[0])
    def avg_leaf_depth(self): return float(self._stats()[1])
    def avg_search_entropy(self): return float(self._stats()[2])
    def fast_avg_leaf_depth(self): return float(self._stats()[3])
    def fast_avg_search_entropy(self): return float(self._stats()[4])
    def avg_moves_per_turn(self): return float(self._stats()[5])
    def avg_valid_moves(self): return float(self._stats()[6])

    def counters(self):
        out = np.zeros(6, np.uint64)
        check(lib.azmi_pm_counters(self._h, out.ctypes.data))
        return dict(zip(("sims", "evals", "cache_hits", "cache_misses", "hist_rows", "rounds"), (int(x) for x in out)))

    def hist_count(self):
        return self.counters()["hist_rows"]

    def _cache_stats(self):
        out = np.zeros(6, np.uint64)
        check(lib.azmi_pm_cache_stats(self._h, out.ctypes.data))
        return [int(x) for x in out]

    # play_manager.h:325-366: sums over the model groups' caches
    def cache_hits(self): return self._cache_stats()[0]
    def cache_misses(self): return self._cache_stats()[1]
    def cache_evictions(self): return self._cache_stats()[2]
    def cache_reinserts(self): return self._cache_stats()[3]
    def cache_size(self): return self._cache_stats()[4]
    def cache_max_size(self):
        if self._caches is not None:      # external caches report the size they were created with (see _ensure_engine_layout)
            return sum(c.max_size() for c in self._caches if c is not None)
        return self._cache_stats()[5]

    # per-variant tables (play_manager.h:218-275): only for games with variants (StarGambitUnifiedGS)
    def num_tracked_variants(self): return int(lib.azmi_pm_num_variants(self._h))

    def _variant(self, v):
        v = int(v)
        if not 0 <= v < self.num_tracked_variants():
            raise IndexError("variant index out of range")
        npm = self.num_seat_perms()
        ps = np.zeros((npm, self._P + 1), np.float32); pg = np.zeros(npm, np.uint32); sums = np.zeros(10, np.float64)
        check(lib.azmi_pm_variant_sums(self._h, v, ps.ctypes.data, pg.ctypes.data, sums.ctypes.data))
        return ps, pg, sums

    def variant_sums(self, v): return self._variant(v)[2]
    def variant_scores(self, v): return self._variant(v)[0].sum(0)
    def variant_games_completed(self, v): return int(self._variant(v)[1].sum())
    def variant_perm_scores(self, v, p): return self._variant(v)[0][int(p)].copy()
    def variant_perm_games_completed(self, v, p): return int(self._variant(v)[1][int(p)])

    @staticmethod
    def _ratio(a, b, single=False):
        if b == 0:
            return 0.0
        return float(np.float32(a) / np.float32(b)) if single else float(np.float32(a / b))

    # the variant_avg_* getters with the reference's own divisions (play_manager.h:233-275)
    def variant_avg_game_length(self, v): s = self.variant_sums(v); return self._ratio(s[0], s[1], True)
    def variant_avg_leaf_depth(self, v): s = self.variant_sums(v); return self._ratio(s[5], s[3])
    def variant_avg_search_entropy(self, v): s = self.variant_sums(v); return self._ratio(s[6], s[3])
    def variant_fast_avg_leaf_depth(self, v): s = self.variant_sums(v); return self._ratio(s[7], s[4])
    def variant_fast_avg_search_entropy(self, v): s = self.variant_sums(v); return self._ratio(s[8], s[4])
    def variant_avg_moves_per_turn(self, v): s = self.variant_sums(v); return self._ratio(s[2], s[0], True)
    def variant_avg_valid_moves(self, v): s = self.variant_sums(v); return self._ratio(s[9], s[2])

    # ---- the raw queue interface (play_manager.h:186-192, 285-286): pop leaf indices, read the slot through
    # game_data(i), write its v() / pi() rows, push_inference(i)
    def pop_games_upto(self, group, n):
        idx = np.zeros(max(int(n), 1), np.uint32)
        cnt = C.c_uint32()
        check(lib.azmi_pm_build_batch_group(self._h, int(group), None, int(n), idx.ctypes.data, C.byref(cnt)))
        return [int(i) for i in idx[: cnt.value]]

    def pop_game(self, group):
        got = self.pop_games_upto(group, 1)
        return got[0] if got else None

    def game_data(self, i):
        i = int(i)
        if not 0 <= i < self._S:
            raise IndexError("game index out of range")
        gd = self._slot_rows.get(i)
        if gd is None:
            gd = self._slot_rows[i] = GameData(self, i)
        return gd

    def push_inference(self, i):
        gd = self.game_data(i)
        self.update_inferences(0, [int(i)], gd._v[None, :], gd._pi[None, :])
    def _groups(self):
        g, p = C.c_uint32(), C.c_uint32()
        check(lib.azmi_pm_groups(self._h, C.byref(g), C.byref(p)))
        return g.value, p.value

    def num_model_groups(self): return self._groups()[0]   # play_manager.h:210
    def num_seat_perms(self): return self._groups()[1]     # play_manager.h:211

    def perm_scores(self, idx):                            # play_manager.h:212-214
        out = np.zeros(self._P + 1, np.float32)
        check(lib.azmi_pm_perm_scores(self._h, int(idx), out.ctypes.data, None))
        return out

    def perm_games_completed(self, idx):                   # play_manager.h:215-217
        n = C.c_uint32()
        check(lib.azmi_pm_perm_scores(self._h, int(idx), None, C.byref(n)))
        return n.value

    def build_batch(self, group, batch, shard=0):
        """py_wrapper.cc:449-504: fills the caller's [max_batch, C, H, W] float32 array, returns slot ids."""
        arr = np.asarray(batch) if not hasattr(batch, "numpy") else batch.numpy()
        if arr.ndim != 4 or tuple(arr.shape[1:]) != tuple(self._chw) or arr.dtype != np.float32 or not arr.flags.c_contiguous:
            raise RuntimeError("Improper batch size")
        idx = np.zeros(arr.shape[0], np.uint32)
        n = C.c_uint32()
        cap = min(arr.shape[0], int(self._params.max_batch_size)) if self._params.max_batch_size else arr.shape[0]
        check(lib.azmi_pm_build_batch_group(self._h, int(group), arr.ctypes.data, cap, idx.ctypes.data, C.byref(n)))
        return [int(i) for i in idx[: n.value]]

    def update_inferences(self, group, indices, v, pi):
        idx = np.ascontiguousarray(indices, dtype=np.uint32)
        v = _f32(v); pi = _f32(pi)
        if v.shape != (len(idx), self._P + 1) or pi.shape != (len(idx), self._M):
            raise RuntimeError("Eigen is angry!!!")  # shapes.h:4-6 bounds assertion
        check(lib.azmi_pm_update_inferences(self._h, idx.ctypes.data, len(idx), v.ctypes.data, pi.ctypes.data))

    def build_history_batch(self, canonical, v, pi):
        """py_wrapper.cc:393-424: fills the caller arrays with finished samples, returns rows written."""
        c = np.asarray(canonical); vv = np.asarray(v); p = np.asarray(pi)
        for a in (c, vv, p):
            if a.dtype != np.float32 or not a.flags.c_contiguous:
                raise RuntimeError("history buffers must be C-contiguous float32")
        n = C.c_uint32()
        check(lib.azmi_pm_pop_history(self._h, c.ctypes.data, vv.ctypes.data, p.ctypes.data, c.shape[0], C.byref(n)))
        return n.value

    # ---- device fast path ------------------------------------------------------------------
    def _stream_arg(self, stream):
        """None -> the stream of the previous call (initially the engine's own); an int (0 included, the HIP
        null stream) -> that hipStream_t."""
        if stream is None:
            return self._last_stream
        self._last_stream = C.c_void_p(int(stream))
        return self._last_stream

    def round(self, stream=None):
        """One engine round on `stream` (a hipStream_t as int, e.g. torch.cuda.current_stream().cuda_stream)."""
        check(lib.azmi_pm_round(self._h, self._stream_arg(stream)))

    def net_forward(self, net, stream=None):
        """The HIP leaf net on this engine's leaf batch, restricted to the rows the last round listed as needing
        an evaluation (azmi_pm_net_forward); `net` is a HipLeafNet."""
        if isinstance(net, (list, tuple)):     # one HipLeafNet per model group
            for g, one in enumerate(net):
                check(lib.azmi_pm_net_forward_group(self._h, g, one._h, self._stream_arg(stream)))
        else:
            check(lib.azmi_pm_net_forward(self._h, net._h, self._stream_arg(stream)))

    def round_net(self, net, stream=None, part=0):
        """One round with its leaf evaluation exactly as the native loop (run_rounds) issues it (azmi_pm_round_net):
        part 0 = all of it, 1 = the tree half, 2 = the net half (with the move step of a split round fused in)."""
        check(lib.azmi_pm_round_net(self._h, net._h, self._stream_arg(stream), int(part)))

    def poll(self, stream=None):
        done, live = C.c_uint32(), C.c_uint32()
        check(lib.azmi_pm_poll(self._h, self._stream_arg(stream), C.byref(done), C.byref(live)))
        return done.value, live.value

    def io_pointers(self):
        c, v, p = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib.azmi_pm_io_buffers(self._h, C.byref(c), C.byref(v), C.byref(p)))
        return c.value, v.value, p.value

    def io_tensors(self):
        """torch views (no copy) of the slot-indexed canonical / v / pi buffers in HBM."""
        import torch
        from ._torch_view import device_tensor
        c, v, p = self.io_pointers()
        S = self._S
        dev = torch.device("cuda", torch.cuda.current_device())
        return (device_tensor(c, (S,) + tuple(self._chw), torch.float32, dev),
                device_tensor(v, (S, self._P + 1), torch.float32, dev),
                device_tensor(p, (S, self._M), torch.float32, dev))

    def history_device_tensors(self, dev=None):
        """Finished samples left in HBM as torch views: canonical [n,C,H,W], v [n,P+1], pi [n,M], meta [n,4] u32."""
        import torch
        from ._torch_view import device_tensor
        c, v, p, m = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        rows = C.c_uint32()
        check(lib.azmi_pm_history_device(self._h, C.byref(c), C.byref(v), C.byref(p), C.byref(m), C.byref(rows)))
        n = rows.value
        # these are views of the ring's first rows: only the whole story while nothing has been consumed and nothing has wrapped
        first, live, cap = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib.azmi_pm_history_window(self._h, C.byref(first), C.byref(live), C.byref(cap)))
        if first.value != 0 or live.value != n:
            raise RuntimeError("history_device_tensors: rows of this engine's sample ring have been consumed or overwritten; "
                               "read it with take_history_device()")
        dev = dev or torch.device("cuda", torch.cuda.current_device())
        if n == 0:
            z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
            return z(0, *self._chw), z(0, self._P + 1), z(0, self._M), torch.zeros((0, 4), dtype=torch.int32, device=dev)
        return (device_tensor(c.value, (n,) + tuple(self._chw), torch.float32, dev),
                device_tensor(v.value, (n, self._P + 1), torch.float32, dev),
                device_tensor(p.value, (n, self._M), torch.float32, dev),
                device_tensor(m.value, (n, 4), torch.int32, dev))

    def take_history_device(self, dev=None, consume=True):
        """The UNREAD finished samples as device tensors (canonical, v, pi, meta) — copies out of the engine's ring
        (azmi_pm_history_window: the window may wrap) — and, with `consume`, their release (azmi_pm_history_consume): the
        device-side counterpart of build_history_batch for the RCCL sample gather and for long self-play streams."""
        import torch
        from ._torch_view import device_tensor
        first, rows, cap = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib.azmi_pm_history_window(self._h, C.byref(first), C.byref(rows), C.byref(cap)))
        n, f, cp = rows.value, first.value, cap.value
        dev = dev or torch.device("cuda", torch.cuda.current_device())
        if n == 0:
            z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
            return z(0, *self._chw), z(0, self._P + 1), z(0, self._M), torch.zeros((0, 4), dtype=torch.int32, device=dev)
        c, v, p, m = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib.azmi_pm_history_device(self._h, C.byref(c), C.byref(v), C.byref(p), C.byref(m), None))
        full = (device_tensor(c.value, (cp,) + tuple(self._chw), torch.float32, dev),
                device_tensor(v.value, (cp, self._P + 1), torch.float32, dev),
                device_tensor(p.value, (cp, self._M), torch.float32, dev),
                device_tensor(m.value, (cp, 4), torch.int32, dev))
        n1 = min(n, cp - f)
        out = tuple(torch.cat([t[f:f + n1], t[:n - n1]], 0) if n1 < n else t[f:f + n].clone() for t in full)
        if consume:
            # the copies above run on torch's current stream, asynchronously: the rows may only be released to the engine
            # (whose later rounds overwrite them) once they have been read
            torch.cuda.current_stream(dev).synchronize()
            check(lib.azmi_pm_history_consume(self._h, n))
        return out

    def move_log(self):
        cap = (int(self._params.games_to_play) + self._S) * 512
        rows = np.zeros((cap, 8), np.uint32)
        counts = np.zeros((cap, self._M), np.uint32)
        n = C.c_uint32()
        check(lib.azmi_pm_move_log(self._h, rows.ctypes.data, counts.ctypes.data, cap, C.byref(n)))
        return rows[: n.value].copy(), counts[: n.value].copy()

    def slot_games(self):
        out = np.zeros(self._S, np.uint32)
        check(lib.azmi_pm_slot_games(self._h, out.ctypes.data))
        return out

    def history(self):
        """All finished samples not yet popped, as numpy arrays (canonical, v, pi)."""
        n = self.hist_count()
        c = np.zeros((n,) + tuple(self._chw), np.float32)
        v = np.zeros((n, self._P + 1), np.float32)
        p = np.zeros((n, self._M), np.float32)
        got = self.build_history_batch(c, v, p) if n else 0
        return c[:got], v[:got], p[:got]


class ShardedS3FIFOCache:
    """py_wrapper.cc:250-259 + find/insert of S3FIFOCache (:222-248), on the device (azmi_cache_*)."""

    def __init__(self, max_size, shards, ghost_size, num_policy, num_value, device=0):
        h = C.c_void_p()
        rc = lib.azmi_cache_create(max_size, shards, ghost_size, num_policy, num_value, device, C.byref(h))
        if rc != 0:
            raise RuntimeError(lib.azmi_cache_last_error().decode())
        self._h, self._np, self._nv = h, num_policy, num_value
        self._args = (int(max_size), int(shards), int(ghost_size), int(device))
        self._nominal_max = None      # set when the cache was re-laid out for an engine (see _ensure_engine_layout)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.azmi_cache_destroy(self._h)
            self._h = None

    def _ensure_engine_layout(self):
        """Called when the cache is handed to a PlayManager: the engine probes 64-entry shards, the reference's callers pass
        any `shards` (cache_utils.create_sharded_cache defaults to 1).  A cache that has not been used yet is re-created in
        the engine's layout behind the same object (capacity rounded down to a multiple of 64; max_size() keeps reporting
        the requested size, like the reference's own shard rounding); one that already holds entries cannot be converted."""
        max_size, shards, ghost, device = self._args
        if max_size // max(1, shards) == 64:
            return
        st = self._stats()
        if st[0] or st[1] or st[4]:
            raise RuntimeError("this cache has been used with another shard layout; create it with ShardedS3FIFOCache.for_engine "
                               "before the first use to share it with a PlayManager")
        new_shards = max(1, max_size // 64)
        h = C.c_void_p()
        rc = lib.azmi_cache_create(new_shards * 64, new_shards, max(0, min(ghost, new_shards * 64)), self._np, self._nv, device, C.byref(h))
        if rc != 0:
            raise RuntimeError(lib.azmi_cache_last_error().decode())
        lib.azmi_cache_destroy(self._h)
        self._h = h
        self._nominal_max = (max_size // max(1, shards)) * max(1, shards)
        self._args = (new_shards * 64, new_shards, ghost, device)

    @classmethod
    def for_engine(cls, max_size, num_policy, num_value, device=0):
        """A cache in the layout the engine probes (64-entry shards, ghost = 9/10 like play_manager.cc:195-203), to be
        passed as PlayManager(gs, params, caches=[...]) and kept from one PlayManager to the next."""
        shards = max(1, int(max_size) // 64)
        return cls(shards * 64, shards, shards * 64 * 9 // 10, num_policy, num_value, device)

    def insert_many(self, hashes, policies, values):
        h = np.ascontiguousarray(hashes, np.uint64)
        p = np.ascontiguousarray(policies, np.float32).reshape(len(h), self._np)
        v = np.ascontiguousarray(values, np.float32).reshape(len(h), self._nv)
        if lib.azmi_cache_insert_many(self._h, h.ctypes.data, p.ctypes.data, v.ctypes.data, len(h)) != 0:
            raise RuntimeError(lib.azmi_cache_last_error().decode())

    def insert(self, hash, policy, value):
        self.insert_many([hash], [policy], [value])

    def find_many(self, hashes):
        h = np.ascontiguousarray(hashes, np.uint64)
        hit = np.zeros(len(h), np.uint8)
        p = np.zeros((len(h), self._np), np.float32)
        v = np.zeros((len(h), self._nv), np.float32)
        if lib.azmi_cache_find_many(self._h, h.ctypes.data, len(h), hit.ctypes.data, p.ctypes.data, v.ctypes.data) != 0:
            raise RuntimeError(lib.azmi_cache_last_error().decode())
        return hit.astype(bool), p, v

    def find(self, hash, num_policy=None, num_value=None):
        hit, p, v = self.find_many([hash])
        return (p[0], v[0]) if hit[0] else None

    def _stats(self):
        out = np.zeros(6, np.uint64)
        lib.azmi_cache_stats(self._h, out.ctypes.data)
        return [int(x) for x in out]

    def hits(self): return self._stats()[0]
    def misses(self): return self._stats()[1]
    def evictions(self): return self._stats()[2]
    def reinserts(self): return self._stats()[3]
    def size(self): return self._stats()[4]
    def max_size(self): return self._nominal_max if self._nominal_max is not None else self._stats()[5]


def hash_game_state(gs):
    """hash_game_state(gs), game_state.h:141-156 / py_wrapper.cc:260-263: the 64-bit position key the caches use.  absl's
    hash is salted per process, so only equality semantics are contract; this is the engine's deterministic key over
    the same fields."""
    return int(gs._state()["key"][0])


def S3FIFOCache(max_size, ghost_size, num_policy, num_value, device=0):
    """py_wrapper.cc:222-248 — the single-shard cache."""
    return ShardedS3FIFOCache(max_size, 1, ghost_size, num_policy, num_value, device)


def run_rounds(pms, net, rounds, streams):
    """azmi_run_rounds: native round driver over several engines (one stream each) sharing one HipLeafNet."""
    k = len(pms)
    arr_pm = (C.c_void_p * k)(*[pm._h for pm in pms])
    arr_st = (C.c_void_p * k)(*[C.c_void_p(int(s)) for s in streams])
    for pm, s in zip(pms, streams):
        pm._last_stream = C.c_void_p(int(s))
    check(lib.azmi_run_rounds(arr_pm, net._h, k, int(rounds), arr_st))


def pipeline_supported(pm, net):
    """azmi_pipeline_supported: True when azmi_run_pipeline can drive this engine with this net (the Connect4 engine with plain
    PUCT seats, one model group, NN seats and a bf16 Connect4-family HipLeafNet - or `net=None` for an engine whose seats all use
    EvalType.RANDOM: the tree side alone -, at most 16384 concurrent games)."""
    return bool(lib.azmi_pipeline_supported(pm._h, None if net is None else net._h))


def run_pipeline(pm, net, epochs, sims_per_epoch, stream=None):
    """azmi_run_pipeline: `epochs` epochs of the asynchronous tree / net pipeline on one engine (persistent tree wavefronts and
    net workgroups side by side; moves, game ends and cache inserts between epochs).  Synchronous.  Returns a dict of
    pipeline counters: net tiles run and boards in them since the pipeline was created, simulations / insert-log entries of
    the last epoch, workgroups launched and started, kernel and host-enqueue times of this call."""
    st = pm._stream_arg(stream)
    out = (C.c_uint64 * 16)()
    check(lib.azmi_run_pipeline(pm._h, None if net is None else net._h, int(epochs), int(sims_per_epoch), st, out))
    return _pipe_stats(out)


def _pipe_stats(out):
    d = dict(zip(_PIPE_KEYS, (int(x) for x in out)))
    # (slot 14 carries two 32-bit counts: the calibration launches of this call, and the requests the net side has given up on and
    # the boundary has sent again since the engine was created - PipeCtl::lost_total)
    d["lost_total"] = d["calibration_rounds"] >> 32
    d["freezes"] = (d["calibration_rounds"] >> 8) & 0xFFFFFF      # polling wavefronts that stood still for > 2 ms since creation (credited, not errors)
    d["calibration_rounds"] &= 0xFF
    return d


_PIPE_KEYS = ("tiles", "tile_boards", "last_epoch_sims", "tree_wgs_started", "net_wgs_started", "last_epoch_inserts", "net_wgs", "tree_wgs",
              "tree_latest_start_us", "net_latest_start_us", "net_kernel_us", "tree_kernel_us", "epochs", "host_enqueue_us", "calibration_rounds",
              "answer_table_hits")


def pipeline_supported_groups(pm, nets):
    """azmi_pipeline_supported_groups: True when azmi_run_pipeline_groups can drive this engine with nets[g] behind model group g
    (None = a group of RANDOM seats)."""
    arr = (C.c_void_p * len(nets))(*[None if n is None else n._h for n in nets])
    return bool(lib.azmi_pipeline_supported_groups(pm._h, arr, len(nets)))


def run_pipeline_groups(pm, nets, epochs, sims_per_epoch, stream=None):
    """azmi_run_pipeline_groups: run_pipeline with one net per model group (play_past: the new model behind group 0, the past one
    behind group 1; None = the reference's RandPlayer)."""
    st = pm._stream_arg(stream)
    out = (C.c_uint64 * 16)()
    arr = (C.c_void_p * len(nets))(*[None if n is None else n._h for n in nets])
    check(lib.azmi_run_pipeline_groups(pm._h, arr, len(nets), int(epochs), int(sims_per_epoch), st, out))
    return _pipe_stats(out)


def pipeline_log_duplicates(pm):
    """azmi_debug_pipe_log_dupes: (answers consumed in the last epoch, of those asked for more than once, of those within 4096 log
    entries of their twin) - what inserting at answer time / coalescing requests in flight would save."""
    out = (C.c_uint64 * 3)()
    check(lib.azmi_debug_pipe_log_dupes(pm._h, out))
    return int(out[0]), int(out[1]), int(out[2])


def run_rounds_groups(pms, nets, rounds, streams):
    """azmi_run_rounds_groups: `nets[g]` (HipLeafNet or None) evaluates the leaves of model group g."""
    k = len(pms)
    arr_pm = (C.c_void_p * k)(*[pm._h for pm in pms])
    arr_net = (C.c_void_p * len(nets))(*[None if n is None else n._h for n in nets])
    arr_st = (C.c_void_p * k)(*[C.c_void_p(int(s)) for s in streams])
    for pm, s in zip(pms, streams):
        pm._last_stream = C.c_void_p(int(s))
    check(lib.azmi_run_rounds_groups(arr_pm, arr_net, len(nets), k, int(rounds), arr_st))


def game_replay(game_cls, moves, device=0):
    """Batched rules replay on the device (azmi_game_replay): moves [n, len] int32, -1 padded."""
    moves = np.ascontiguousarray(moves, dtype=np.int32)
    n, ln = moves.shape
    P, M, chw = game_cls._info()
    out = dict(
        valid=np.zeros((n, M), np.uint8), scores=np.zeros((n, P + 1), np.float32),
        canonical=np.zeros((n,) + tuple(chw), np.float32), player=np.zeros(n, np.uint32),
        turn=np.zeros(n, np.uint32), key=np.zeros(n, np.uint64), status=np.zeros(n, np.int32),
    )
    check(lib.azmi_game_replay(game_cls.GAME_ID, device, moves.ctypes.data, n, ln, out["valid"].ctypes.data,
                               out["scores"].ctypes.data, out["canonical"].ctypes.data, out["player"].ctypes.data,
                               out["turn"].ctypes.data, out["key"].ctypes.data, out["status"].ctypes.data))
    return out


def rng_probe(kind, seed, n, param=0.0, reps=1, device=0):
    """azmi_rng_probe: kind in {"pcg32","shuffle","uniform","gamma","gamma_fresh"}."""
    kinds = {"pcg32": 0, "shuffle": 1, "uniform": 2, "gamma": 3, "gamma_fresh": 4}
    k = kinds[kind]
    dtype = np.uint32 if k < 2 else np.float32
    out = np.zeros(n * (reps if k == 1 else 1), dtype)
    check(lib.azmi_rng_probe(device, k, seed, param, n, reps, out.ctypes.data))
    return out.reshape(reps, n) if k == 1 else out


# Tracy stubs — py_wrapper.cc:772-787: must exist even as no-ops
def tracy_is_enabled():
    return False


def tracy_frame_mark():
    return None


def _tracy_zone_begin(name, file, line):
    return None


def _tracy_zone_end():
    return None


def _tracy_set_thread_name(name):
    return None
