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
    # the same case on the asynchronous pipeline (round 4: azmi_run_pipeline_groups drives Gumbel seats and two model groups; with
    # RANDOM seats it is the tree kernel alone): the very same games, permutation tables included
    nets = [None] * pm.num_model_groups()
    pq = az.PlayManager(game_cls(), pp, seed=seed, log_moves=True)
    if az.pipeline_supported_groups(pq, nets):
        n = 0
        while pq.remaining_games() > 0 and n < 20000:
            az.run_pipeline_groups(pq, nets, 2, S * 64)
            n += 1
            if pq.poll()[1] == 0:
                break
        rows2, counts2 = pq.move_log()
        o1, o2 = np.lexsort((rows[:, 2], rows[:, 1], rows[:, 0])), np.lexsort((rows2[:, 2], rows2[:, 1], rows2[:, 0]))
        assert np.array_equal(rows[o1], rows2[o2]) and np.array_equal(counts[o1], counts2[o2]), "the pipeline plays other games than the lock-step engine"
        for q in range(nperm):
            assert np.array_equal(pq.perm_scores(q), want[q]) and pq.perm_games_completed(q) == games[q]
    else:
        assert game_cls is not az.Connect4GS, "every Connect4 case of this file runs on the pipeline too"
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


# ---- per-seat Gumbel and resign matrices (play_manager.h:126-153, play_manager.cc:116-176, 335-402, 602-617) ---------
def test_per_seat_gumbel_connect4(az, oracle):
    """one seat searches with Gumbel (its own m / c_visit / c_scale / full), the other with PUCT; the seats swap with the
    permutation.  Global gumbel_enabled = False: history rows carry probs(1), not the improved policy."""
    pp = _base(az, games_to_play=8, concurrent_games=8, mcts_visits=[32, 40], model_groups=[0, 1], seat_perms=[[0, 1], [1, 0]],
               seat_gumbel_enabled=[[1, 0], [0, 1]], seat_gumbel_m=[[4, 16], [16, 8]], seat_gumbel_c_visit=[[50.0, 50.0], [50.0, 20.0]],
               seat_gumbel_c_scale=[[1.0, 1.0], [1.0, 0.5]], seat_gumbel_full=[[1, 0], [0, 0]], epsilon=0.25)
    _slotwise(az, oracle, az.Connect4GS, oracle.GAME_CONNECT4, pp, seed=71)
    # global Gumbel with one seat opted out and the other acting by sampling the improved policy (G3), three temperature regimes
    for temps in ((1.0, 1.0), (0.6, 0.6), (0.0, 0.0)):
        pp = _base(az, games_to_play=6, concurrent_games=6, mcts_visits=[28, 28], model_groups=[0, 1], seat_perms=[[0, 1], [1, 0]],
                   gumbel_enabled=True, gumbel_m=8, seat_gumbel_enabled=[[1, 0], [1, 1]],
                   seat_gumbel_use_improved_policy=[[1, 0], [0, 1]], start_temp=temps[0], final_temp=temps[1])
        _slotwise(az, oracle, az.Connect4GS, oracle.GAME_CONNECT4, pp, seed=72)


def test_per_seat_gumbel_wide_games(az, oracle):
    pp = _base(az, games_to_play=3, concurrent_games=3, mcts_visits=[20, 24], model_groups=[0, 1], seat_perms=[[0, 1], [1, 0]],
               seat_gumbel_enabled=[[1, 0], [0, 1]], seat_gumbel_m=[[8, 16], [16, 4]], seat_gumbel_use_improved_policy=[[1, 0], [0, 0]],
               start_temp=0.8, final_temp=0.8)
    _slotwise(az, oracle, az.TawlbwrddGS, oracle.GAME_TAWLBWRDD, pp, seed=73)
    pp = _base(az, games_to_play=4, concurrent_games=4, mcts_visits=[16, 16], gumbel_enabled=True, gumbel_m=6,
               seat_gumbel_full=[[1, 0]], seat_gumbel_c_scale=[[0.5, 2.0]])
    _slotwise(az, oracle, az.BrandubhGS, oracle.GAME_BRANDUBH, pp, seed=74)


def test_per_seat_resign(az, oracle):
    """seat_resign_threshold / seat_resign_consecutive (play_manager.cc:335-366): with the uniform evaluator W - L stays
    near 0, so a threshold of 0.1 trips after `consecutive` own moves and the opponent is credited."""
    pp = _base(az, games_to_play=8, concurrent_games=8, mcts_visits=[20, 20], model_groups=[0, 1], seat_perms=[[0, 1], [1, 0]],
               seat_resign_threshold=[[0.1, -2.0], [-2.0, 0.1]], seat_resign_consecutive=[[3, 1], [1, 2]])
    pm = _slotwise(az, oracle, az.Connect4GS, oracle.GAME_CONNECT4, pp, seed=75)
    rows, _ = pm.move_log()
    lens = [int((rows[:, 0] == s).sum()) for s in range(8)]
    assert max(lens) <= 6                       # seat 0 resigns on its 3rd move (perm 0), seat 1 on its 2nd (perm 1)
    assert pm.scores().sum() == 8 and pm.resign_scores().sum() == 8
    pp = _base(az, games_to_play=3, concurrent_games=3, mcts_visits=[12, 12], seat_resign_threshold=[[-2.0, 0.2]],
               seat_resign_consecutive=[[1, 4]])
    _slotwise(az, oracle, az.BrandubhGS, oracle.GAME_BRANDUBH, pp, seed=76)


def test_resign_streak_is_kept_across_games_of_a_slot(az, oracle):
    """GameData::resign_streak is not cleared when the slot starts its next game (play_manager.cc:498-520): one slot,
    three games, compared with the oracle's one-slot run."""
    pp = _base(az, games_to_play=3, concurrent_games=1, mcts_visits=[16, 16], seat_resign_threshold=[[0.1, -2.0]],
               seat_resign_consecutive=[[2, 1]])
    seed = 77
    pm = az.PlayManager(az.Connect4GS(), pp, seed=seed, log_moves=True)
    pm.play()
    o = oracle.PlayManager(oracle.GAME_CONNECT4, pp, oracle.slot_seed(seed, 0), per_slot_rng=False)
    o.run()
    rows, counts = pm.move_log()
    orows, ocounts = o.moves()
    assert np.array_equal(rows[:, 1:], orows[:, 1:]) and np.array_equal(counts, ocounts)
    assert np.array_equal(pm.scores(), o.scores()) and np.array_equal(pm.resign_scores(), o.resign_scores())
    games = [int((rows[:, 1] == g).sum()) for g in range(3)]
    assert games[0] == 3 and games[1] == 1 and games[2] == 1      # later games resign at seat 0's first move: the streak carried over


def test_per_seat_matrix_validation(az):
    pp = _base(az, games_to_play=2, concurrent_games=2, mcts_visits=[8, 8], seat_gumbel_m=[[4, 4], [4, 4]])
    with pytest.raises(RuntimeError, match="seat_gumbel_m outer dimension must match number of seat permutations"):
        az.PlayManager(az.Connect4GS(), pp)
    pp = _base(az, games_to_play=2, concurrent_games=2, mcts_visits=[8, 8], seat_resign_threshold=[[0.1]])
    with pytest.raises(RuntimeError, match="seat_resign_threshold inner dimension must match number of players"):
        az.PlayManager(az.Connect4GS(), pp)
    pp = _base(az, games_to_play=2, concurrent_games=2, mcts_visits=[8, 8], seat_gumbel_enabled=[[1, 0]], seat_gumbel_m=[[100, 4]])
    with pytest.raises(RuntimeError, match="gumbel_m"):
        az.PlayManager(az.Connect4GS(), pp)
