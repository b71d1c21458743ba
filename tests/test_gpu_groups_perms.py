"""-m gpu: model groups, seat permutations and the per-seat override matrices (play_manager.cc:24-113, 214-230,
283-286, 466-467, 511, 575-597) — the `play_past` configuration of game_runner.py:2184-2332 — against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def az():
    import alphazero
    return alphazero


def _base(az, **kw):
    pp = az.PlayParams()
    pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
    pp.history_enabled = True
    pp.cpuct, pp.fpu_reduction = 1.25, 0.25
    for k, v in kw.items():
        setattr(pp, k, v)
    return pp


def _slotwise(az, oracle, game_cls, gid, pp, seed):
    """one game per slot (no restarts): slot s plays with permutation s % perms, like a one-slot oracle with perm_base=s"""
    S = pp.concurrent_games
    assert pp.games_to_play == S
    pm = az.PlayManager(game_cls(), pp, seed=seed, log_moves=True)
    pm.play()
    rows, counts = pm.move_log()
    nperm = pm.num_seat_perms()
    want = [np.zeros(3, np.float32) for _ in range(nperm)]
    games = [0] * nperm
    for s in range(S):
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games = 1, 1
        o = oracle.PlayManager(gid, one, oracle.slot_seed(seed, s), per_slot_rng=False, perm_base=s)
        o.run()
        orows, ocounts = o.moves()
        sel = rows[:, 0] == s
        assert np.array_equal(rows[sel][:, 1:], orows[:, 1:]), s
        assert np.array_equal(counts[sel], ocounts), s
        want[s % nperm] += o.scores(); games[s % nperm] += 1
    for q in range(nperm):
        assert np.array_equal(pm.perm_scores(q), want[q]) and pm.perm_games_completed(q) == games[q]
    return pm


def test_default_is_one_group_per_player(az):
    pp = _base(az, games_to_play=2, concurrent_games=2, mcts_visits=[8, 8])
    pm = az.PlayManager(az.Connect4GS(), pp)
    assert pm.num_model_groups() == 2 and pm.num_seat_perms() == 1       # play_manager.cc:24-31, 47-51
    pp.model_groups = [0, 0]
    pm = az.PlayManager(az.Connect4GS(), pp)
    assert pm.num_model_groups() == 1


def test_play_past_configuration_connect4(az, oracle):
    """two model groups, both seatings, different search budgets / noise / root settings per seat"""
    pp = _base(az, games_to_play=8, concurrent_games=8, mcts_visits=[30, 50], model_groups=[0, 1],
               seat_perms=[[0, 1], [1, 0]], epsilon=0.0)
    pm = _slotwise(az, oracle, az.Connect4GS, oracle.GAME_CONNECT4, pp, seed=61)
    assert pm.num_model_groups() == 2 and pm.num_seat_perms() == 2
    # visits follow the group into the other seat: perm 1 searches 50 sims in seat 0
    rows, counts = pm.move_log()
    first = {int(r[0]): int(c.sum()) for r, c in zip(rows, counts) if r[3] == 0}
    assert first[0] == 29 and first[1] == 49
    pp = _base(az, games_to_play=8, concurrent_games=8, mcts_visits=[24, 24], model_groups=[0, 1], seat_perms=[[0, 1], [1, 0]],
               seat_visits=[[20, 36], [36, 20]], seat_epsilon=[[0.25, 0.0], [0.0, 0.25]], seat_mcts_root_temp=[[1.25, 1.0], [1.0, 1.25]],
               seat_root_fpu_zero=[[1, 0], [0, 1]], shaped_dirichlet=True, policy_target_pruning=True,
               playout_cap_randomization=True, playout_cap_percent=0.4, seat_cap_visits=[[6, 10], [10, 6]])
    _slotwise(az, oracle, az.Connect4GS, oracle.GAME_CONNECT4, pp, seed=62)


def test_play_past_configuration_tawlbwrdd(az, oracle):
    pp = _base(az, games_to_play=4, concurrent_games=4, mcts_visits=[16, 24], model_groups=[0, 1], seat_perms=[[0, 1], [1, 0]],
               seat_epsilon=[[0.25, 0.0], [0.0, 0.25]])
    _slotwise(az, oracle, az.TawlbwrddGS, oracle.GAME_TAWLBWRDD, pp, seed=63)


def test_group_routing_with_two_evaluators(az, oracle):
    """build_batch(group) hands out only that group's leaves and update_inferences feeds them back; two different
    synthetic evaluators (one per model group), both seatings; per-group caches on.  Equals the oracle driven by the
    same pair of evaluators."""
    def evaluator(group, canon):
        n = canon.shape[0]
        flat = canon.reshape(n, -1)
        w = np.linspace(0.5, 1.5, flat.shape[1], dtype=np.float32) * (1.0 + 0.5 * group)
        s = flat @ w
        v = np.stack([0.3 + 0.1 * np.sin(s), 0.3 - 0.1 * np.sin(s), np.full(n, 0.4)], 1).astype(np.float32)
        pi = np.abs(np.sin(s[:, None] * np.arange(1, 8, dtype=np.float32) + group)) + 0.05
        return v, (pi / pi.sum(1, keepdims=True)).astype(np.float32)

    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.max_batch_size = 6, 6, 6
    pp.mcts_visits = [24, 32]
    pp.model_groups, pp.seat_perms = [0, 1], [[0, 1], [1, 0]]
    pp.cpuct, pp.fpu_reduction, pp.history_enabled = 1.25, 0.25, True
    pp.max_cache_size = 4096
    seed = 64
    pm = az.PlayManager(az.Connect4GS(), pp, seed=seed, log_moves=True)
    batch = np.zeros((6, 4, 6, 7), np.float32)
    seen = [0, 0]
    while pm.remaining_games() > 0:
        for g in range(pm.num_model_groups()):       # the GameRunner loop: one batcher per model group
            idx = pm.build_batch(g, batch)
            if not idx:
                continue
            seen[g] += len(idx)
            v, pi = evaluator(g, batch[: len(idx)])
            pm.update_inferences(g, idx, v, pi)
    assert seen[0] > 0 and seen[1] > 0
    rows, counts = pm.move_log()
    for s in range(6):
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games, one.max_cache_size = 1, 1, 0
        o = oracle.PlayManager(oracle.GAME_CONNECT4, one, oracle.slot_seed(seed, s), per_slot_rng=False, perm_base=s)
        o.run_groups(evaluator)
        orows, ocounts = o.moves()
        sel = rows[:, 0] == s
        assert np.array_equal(rows[sel][:, 2:], orows[:, 2:]), s
        assert np.array_equal(counts[sel], ocounts), s


def test_shape_errors_match_the_reference(az):
    pp = _base(az, games_to_play=1, concurrent_games=1, mcts_visits=[8, 8], model_groups=[0, 1], seat_perms=[[0, 1], [1, 0]])
    pp.seat_visits = [[8, 8]]
    with pytest.raises(RuntimeError, match="seat_visits outer dimension must match number of seat permutations"):
        az.PlayManager(az.Connect4GS(), pp)
    pp.seat_visits = [[8, 8], [8]]
    with pytest.raises(RuntimeError, match="seat_visits inner dimension must match number of players"):
        az.PlayManager(az.Connect4GS(), pp)
