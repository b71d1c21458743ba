"""The compiled pybind11 binding (integration/azmi_pybind.cpp -> alphazero-pybind11_amd/azmi_pybind*.so): the class INTEGRATION.md tells a
maintainer to add next to the reference's PlayManager binding (/root/reference/src/py_wrapper.cc:108-330), compiled here against the
C ABI alone.  CPU: the module loads and exposes the bound surface, and fails loudly without a device; GPU: it plays the games the
ctypes module plays (same C ABI underneath: equal scores, equal history rows)."""
import numpy as np
import pytest


def _mods():
    import alphazero as az          # (first: it shares PyTorch's HIP runtime with libazmi.so)
    import azmi_pybind as ap
    return az, ap


def test_compiled_module_loads_and_exposes_the_bound_surface():
    az, ap = _mods()
    for name in ("play", "build_batch", "update_inferences", "build_history_batch", "scores", "games_completed", "remaining_games",
                 "stop", "stopped", "awaiting_inference_count", "awaiting_mcts_count", "resign_scores", "avg_game_length", "avg_leaf_depth",
                 "avg_search_entropy", "fast_avg_leaf_depth", "fast_avg_search_entropy", "avg_moves_per_turn", "avg_valid_moves", "stat_sums",
                 "hist_count", "cache_hits", "cache_misses", "cache_evictions", "cache_reinserts", "cache_size", "cache_max_size",
                 "num_model_groups", "num_seat_perms", "perm_scores", "perm_games_completed", "num_tracked_variants"):
        assert hasattr(ap.DevicePlayManager, name), name
    p = ap.PlayParams()
    p.games_to_play, p.concurrent_games, p.mcts_visits, p.eval_type = 4, 4, [10, 10], [1, 1]
    assert p.mcts_visits == [10, 10] and p.cpuct == 2.0 and p.tree_reuse is True
    assert ap.GAME_CONNECT4 == 0


def test_no_device_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    az, ap = _mods()
    p = ap.PlayParams()
    p.games_to_play, p.concurrent_games, p.mcts_visits, p.eval_type = 4, 4, [10, 10], [1, 1]
    with pytest.raises(RuntimeError):
        ap.DevicePlayManager(ap.GAME_CONNECT4, p, 1)


def _params(obj, games, slots, sims, eval_type):
    obj.games_to_play, obj.concurrent_games, obj.max_batch_size = games, slots, slots
    obj.mcts_visits, obj.eval_type = [sims, sims], eval_type
    obj.history_enabled = True
    obj.model_groups = [0, 0]           # one network on both seats (set_model_groups(), game_runner.py:773-787)
    obj.cpuct, obj.start_temp, obj.final_temp = 1.25, 1.0, 1.0
    return obj


@pytest.mark.gpu
def test_random_evaluator_games_equal_the_ctypes_module():
    az, ap = _mods()
    seed = 4242
    a = ap.DevicePlayManager(ap.GAME_CONNECT4, _params(ap.PlayParams(), 32, 16, 60, [1, 1]), seed)
    a.play()
    b = az.PlayManager(az.Connect4GS(), _params(az.PlayParams(), 32, 16, 60, [az.EvalType.RANDOM, az.EvalType.RANDOM]), seed=seed)
    b.play()
    assert a.games_completed() == b.games_completed() == 32 and a.remaining_games() == 0
    assert np.array_equal(np.asarray(a.scores()), np.asarray(b.scores()))
    assert a.num_players == 2 and a.num_moves == 7 and a.canonical_shape == (4, 6, 7)
    # the read-out surface: every getter of the compiled binding against the ctypes module's (same C ABI underneath)
    for name in ("avg_game_length", "avg_leaf_depth", "avg_search_entropy", "fast_avg_leaf_depth", "fast_avg_search_entropy",
                 "avg_moves_per_turn", "avg_valid_moves", "hist_count", "cache_hits", "cache_misses", "cache_evictions", "cache_reinserts",
                 "cache_size", "cache_max_size", "num_model_groups", "num_seat_perms", "num_tracked_variants", "awaiting_inference_count",
                 "awaiting_mcts_count", "stopped"):
        assert getattr(a, name)() == getattr(b, name)(), name
    assert np.array_equal(np.asarray(a.resign_scores()), np.asarray(b.resign_scores()))
    assert np.array_equal(np.asarray(a.stat_sums()), np.asarray(list(b.stat_sums().values())))
    for q in range(a.num_seat_perms()):
        assert np.array_equal(np.asarray(a.perm_scores(q)), np.asarray(b.perm_scores(q))) and a.perm_games_completed(q) == b.perm_games_completed(q)
    assert a.simulations() == b.counters()["sims"] and a.leaf_evaluations() == b.counters()["evals"]


@pytest.mark.gpu
def test_host_buffer_loop_through_the_compiled_binding():
    """game_runner.py's loop (build_batch -> net -> update_inferences, :651-726) with a uniform evaluator, once through the compiled
    binding and once through the ctypes module: the same leaves come out, the same games are played, the same history rows pop"""
    az, ap = _mods()
    seed, S = 77, 8

    def drive(pm, n_players, n_moves):
        batch = np.zeros((S, 4, 6, 7), np.float32)
        v = np.full((S, n_players + 1), 1.0 / (n_players + 1), np.float32)
        pi = np.full((S, n_moves), 1.0 / n_moves, np.float32)
        turns = 0
        while pm.remaining_games() > 0 and turns < 100000:
            idx = pm.build_batch(0, batch, 0)
            if len(idx):          # (build_batch itself advances the engine until leaves need the net; empty = the games are over)
                pm.update_inferences(0, list(idx), v[:len(idx)], pi[:len(idx)])
            turns += 1
        c = np.zeros((4096, 4, 6, 7), np.float32); hv = np.zeros((4096, 3), np.float32); hp = np.zeros((4096, 7), np.float32)
        n = pm.build_history_batch(c, hv, hp)
        return np.asarray(pm.scores()).copy(), int(n), c[:n].copy(), hv[:n].copy(), hp[:n].copy()

    a = ap.DevicePlayManager(ap.GAME_CONNECT4, _params(ap.PlayParams(), 16, S, 30, [0, 0]), seed)
    b = az.PlayManager(az.Connect4GS(), _params(az.PlayParams(), 16, S, 30, [az.EvalType.NN, az.EvalType.NN]), seed=seed)
    ra, rb = drive(a, 2, 7), drive(b, 2, 7)
    assert ra[1] == rb[1] > 0 and np.array_equal(ra[0], rb[0])
    for x, y in zip(ra[2:], rb[2:]):
        assert np.array_equal(x, y)
