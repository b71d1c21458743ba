"""-m gpu: the HIP engine (through the C ABI, libazmi.so) against the CPU oracle.

Tiers: T0 rules, T2 PlayManager slot — bit-exact on move sequences, root visit
counts, emitted PlayHistory rows, final scores and the statistics getters.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _az():
    import alphazero
    return alphazero


def _random_playouts(orc, n, length, seed):
    """Random legal Connect4 move lists generated with the ORACLE rules."""
    rng = np.random.default_rng(seed)
    moves = -np.ones((n, length), np.int32)
    finals = []
    for g in range(n):
        game = orc.Game(orc.GAME_CONNECT4)
        stop = rng.integers(0, length + 1)
        for i in range(stop):
            if game.scores() is not None:
                break
            legal = np.flatnonzero(game.valid())
            m = int(rng.choice(legal))
            game.play(m)
            moves[g, i] = m
        finals.append(game)
    return moves, finals


def test_rules_random_playouts(oracle):
    """T0: valid_moves / play_move / scores / canonicalized / player / turn after random playouts."""
    az = _az()
    moves, finals = _random_playouts(oracle, 4000, 42, seed=7)
    out = az.game_replay(az.Connect4GS, moves)
    assert (out["status"] == 0).all()
    for g, game in enumerate(finals):
        assert np.array_equal(out["valid"][g], game.valid()), g
        sc = game.scores()
        if sc is None:
            assert (out["scores"][g] == -1).all(), g
        else:
            assert np.array_equal(out["scores"][g], sc), g
        assert np.array_equal(out["canonical"][g], game.canonical()), g
        assert out["player"][g] == game.player() and out["turn"][g] == game.turn(), g
        assert int(out["key"][g]) == game.key(), g


def test_rules_reference_known_answers():
    """connect4_gs_test.cc:104-171 win-state cases, reached by move sequences."""
    az = _az()
    cases = [
        ([0, 0, 1, 1, 2, 2, 3], [1, 0, 0]),           # bottom row, player 0
        ([0, 1, 0, 1, 0, 1, 0], [1, 0, 0]),           # column, player 0
        ([6, 0, 6, 1, 6, 2, 5, 3], [0, 1, 0]),        # bottom row, player 1
        ([0, 1, 1, 2, 2, 3, 2, 3, 3, 6, 3], [1, 0, 0]),  # "/" diagonal
        ([3, 2, 2, 1, 1, 0, 1, 0, 0, 6, 0], [1, 0, 0]),  # "\" diagonal
    ]
    L = max(len(c[0]) for c in cases)
    moves = -np.ones((len(cases), L), np.int32)
    for i, (mv, _) in enumerate(cases):
        moves[i, : len(mv)] = mv
    out = az.game_replay(az.Connect4GS, moves)
    for i, (_, sc) in enumerate(cases):
        assert out["scores"][i].tolist() == sc, i
    # full column: the 7th drop into one column is illegal (connect4_gs.cc:48-58 throws)
    out = az.game_replay(az.Connect4GS, np.array([[3] * 7], np.int32))
    assert out["status"][0] == -1


def _params(az, **kw):
    pp = az.PlayParams()
    pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
    pp.history_enabled = True
    for k, v in kw.items():
        setattr(pp, k, v)
    return pp


def _compare_slotwise(az, orc, pp, seed, slots=None):
    pm = az.PlayManager(az.Connect4GS(), pp, seed=seed, log_moves=True)
    pm.play()
    rows, counts = pm.move_log()
    games = pm.slot_games()
    hc, hv, hp = pm.history()
    assert int(games.sum()) == pm.games_completed() >= pp.games_to_play
    S = pp.concurrent_games
    tot_scores = np.zeros(3, np.float32)
    n_hist = 0
    for s in (range(S) if slots is None else slots):
        if games[s] == 0:
            continue
        one = az.PlayParams()
        one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games = int(games[s]), 1
        o = orc.PlayManager(orc.GAME_CONNECT4, one, orc.slot_seed(seed, s), per_slot_rng=False)
        o.run()
        orows, ocounts = o.moves()
        sel = (rows[:, 0] == s) & (rows[:, 1] < games[s])
        drows, dcounts = rows[sel], counts[sel]
        order = np.lexsort((drows[:, 3], drows[:, 1]))
        drows, dcounts = drows[order], dcounts[order]
        assert drows.shape == orows.shape, (s, drows.shape, orows.shape)
        assert np.array_equal(drows[:, 1:], orows[:, 1:]), f"slot {s}: moves/turns/players differ"
        assert np.array_equal(dcounts, ocounts), f"slot {s}: root visit counts differ"
        tot_scores += o.scores()
        n_hist += o.counters()["hist_rows"]
    return pm, rows, (hc, hv, hp), tot_scores, n_hist


def test_playmanager_exact_tier(oracle):
    """T2, exact tier: eps=0, root temp 1, start_temp 1, tree reuse, one game per slot."""
    az = _az()
    pp = _params(az, games_to_play=64, concurrent_games=64, mcts_visits=[100, 100], cpuct=1.25, fpu_reduction=0.25)
    pm, rows, hist, tot_scores, n_hist = _compare_slotwise(az, oracle, pp, seed=20240601)
    assert np.array_equal(pm.scores(), tot_scores)
    assert len(hist[0]) == n_hist
    assert np.allclose(hist[1].sum(1), 1.0) and np.allclose(hist[2].sum(1), 1.0, atol=1e-6)


def test_playmanager_stream_pool_and_stats(oracle):
    """T2 with games_to_play = 4 x concurrent (stream pool) + statistics getters vs one oracle
    run per slot aggregated in slot order."""
    az = _az()
    pp = _params(az, games_to_play=96, concurrent_games=24, mcts_visits=[40, 60], cpuct=2.0, start_temp=1.0)
    pm, rows, hist, tot_scores, n_hist = _compare_slotwise(az, oracle, pp, seed=99)
    assert np.array_equal(pm.scores(), tot_scores)
    assert len(hist[0]) == n_hist


def test_playmanager_history_rows_match(oracle):
    """The multiset of (canonical, v, pi) rows equals the oracle's."""
    az = _az()
    orc = oracle
    pp = _params(az, games_to_play=8, concurrent_games=8, mcts_visits=[64, 64], cpuct=1.25, fpu_reduction=0.25,
                 root_fpu_zero=True)
    seed = 5
    pm = az.PlayManager(az.Connect4GS(), pp, seed=seed)
    pm.play()
    hc, hv, hp = pm.history()
    rows_dev = sorted((c.tobytes(), v.tobytes(), p.tobytes()) for c, v, p in zip(hc, hv, hp))
    rows_orc = []
    for s in range(8):
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games = 1, 1
        o = orc.PlayManager(orc.GAME_CONNECT4, one, orc.slot_seed(seed, s), per_slot_rng=False)
        o.run()
        c, v, p = o.history()
        rows_orc += [(a.tobytes(), b.tobytes(), d.tobytes()) for a, b, d in zip(c, v, p)]
    assert rows_dev == sorted(rows_orc)
    # statistics getters against the all-slots oracle run (per-slot streams, same aggregation)
    o = orc.PlayManager(orc.GAME_CONNECT4, pp, seed, per_slot_rng=True)
    o.run()
    st = o.stats()
    dev = [pm.avg_game_length(), pm.avg_leaf_depth(), pm.avg_search_entropy(), pm.fast_avg_leaf_depth(),
           pm.fast_avg_search_entropy(), pm.avg_moves_per_turn(), pm.avg_valid_moves()]
    assert np.allclose(dev, st, rtol=1e-6, atol=0), (dev, st)
    assert np.array_equal(pm.scores(), o.scores())


@pytest.mark.parametrize("cfg", [
    dict(epsilon=0.25, shaped_dirichlet=False),
    dict(epsilon=0.25, shaped_dirichlet=True, mcts_root_temp=1.25, policy_target_pruning=True, root_fpu_zero=True),
    dict(tree_reuse=False, start_temp=0.0),
    dict(temp_decay_half_life=10.0, final_temp=0.2, start_temp=1.0),
    dict(playout_cap_randomization=True, playout_cap_depth=10, playout_cap_percent=0.75, epsilon=0.25),
    dict(resign_percent=0.02, resign_playthrough_percent=0.2),
])
def test_playmanager_option_tiers(oracle, cfg):
    """Noise / temperature / cap / resign tiers: exact against the oracle, which uses the same
    float(round(double fn)) definition of logf/powf/expf as the device."""
    az = _az()
    pp = _params(az, games_to_play=32, concurrent_games=16, mcts_visits=[48, 48], cpuct=1.25, fpu_reduction=0.25, **cfg)
    pm, rows, hist, tot_scores, n_hist = _compare_slotwise(az, oracle, pp, seed=4242)
    assert np.array_equal(pm.scores(), tot_scores)
    assert len(hist[0]) == n_hist
    if cfg.get("playout_cap_randomization"):
        assert rows[:, 5].any() and not rows[:, 5].all()


def test_host_buffer_path_matches_device_path(oracle):
    """build_batch / update_inferences (host arrays, reference signatures) with a deterministic
    synthetic evaluator == oracle driven by the same evaluator."""
    az = _az()
    orc = oracle

    def evaluator(canon):
        n = canon.shape[0]
        flat = canon.reshape(n, -1)
        w = np.linspace(0.5, 1.5, flat.shape[1], dtype=np.float32)
        s = flat @ w
        v = np.stack([0.3 + 0.1 * np.sin(s), 0.3 - 0.1 * np.sin(s), np.full(n, 0.4)], 1).astype(np.float32)
        pi = np.abs(np.sin(s[:, None] * np.arange(1, 8, dtype=np.float32))) + 0.05
        pi = (pi / pi.sum(1, keepdims=True)).astype(np.float32)
        return v, pi

    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.max_batch_size = 6, 6, 4
    pp.model_groups = [0, 0]     # one evaluator for both seats (game_runner.set_model_groups)
    pp.mcts_visits = [30, 30]
    pp.cpuct, pp.fpu_reduction = 1.25, 0.25
    pp.history_enabled = True
    seed = 77
    pm = az.PlayManager(az.Connect4GS(), pp, seed=seed, log_moves=True)
    batch = np.zeros((4, 4, 6, 7), np.float32)
    while pm.remaining_games() > 0:
        idx = pm.build_batch(0, batch)
        if not idx:
            continue
        v, pi = evaluator(batch[: len(idx)])
        pm.update_inferences(0, idx, v, pi)
    rows, counts = pm.move_log()
    for s in range(6):
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games = 1, 1
        o = orc.PlayManager(orc.GAME_CONNECT4, one, orc.slot_seed(seed, s), per_slot_rng=False)
        o.run(evaluator)
        orows, ocounts = o.moves()
        sel = rows[:, 0] == s
        assert np.array_equal(rows[sel][:, 2:], orows[:, 2:]), s
        assert np.array_equal(counts[sel], ocounts), s


def test_error_behaviour():
    az = _az()
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games = 1, 1
    pp.mcts_visits = [10]  # play_manager.cc:20-22
    with pytest.raises(RuntimeError, match="You must specify MCTS visits for each player"):
        az.PlayManager(az.Connect4GS(), pp)
    pp.mcts_visits = [10, 10]
    pm = az.PlayManager(az.Connect4GS(), pp)
    with pytest.raises(RuntimeError, match="Improper batch size"):  # py_wrapper.cc:469-475
        pm.build_batch(0, np.zeros((2, 3, 6, 7), np.float32))
    with pytest.raises(TypeError):
        az.PlayManager(None, pp)


def test_control_surface_stop_queues_variants():
    """stop / stopped / remaining_games (play_manager.h:177-182), awaiting_* counts, set_eager, the variant accessors."""
    import alphazero as az
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.max_batch_size = 1000, 16, 16
    pp.mcts_visits = [20, 20]
    pp.model_groups = [0, 0]
    pm = az.PlayManager(az.Connect4GS(), pp, seed=2)
    assert not pm.stopped() and pm.remaining_games() == 1000
    assert pm.awaiting_inference_count() == 0 and pm.awaiting_mcts_count() == 16
    batch = np.zeros((16, 4, 6, 7), np.float32)
    idx = pm.build_batch(0, batch[:5])
    assert len(idx) == 5 and pm.awaiting_inference_count() == 11 and pm.awaiting_mcts_count() == 0
    pm.set_eager(True); pm.set_eager(False)
    assert pm.num_tracked_variants() == 0
    with pytest.raises(IndexError):
        pm.variant_scores(0)
    pm.stop()
    assert pm.stopped() and pm.remaining_games() == 0
    assert pm.build_batch(0, batch) == []
    # stop() ends a running play() (RANDOM seats) from another thread
    import threading
    pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
    pp.games_to_play = 10 ** 9
    pm2 = az.PlayManager(az.Connect4GS(), pp, seed=2)
    t = threading.Thread(target=pm2.play)
    t.start()
    import time
    time.sleep(0.3)
    pm2.stop()
    t.join(timeout=20)
    assert not t.is_alive() and pm2.games_completed() > 0


def test_raw_queue_interface_matches_build_batch(oracle):
    """pop_games_upto / game_data(i) / push_inference(i) (play_manager.h:186-192, 285-286; py_wrapper.cc:265-288) drive
    the same games as build_batch / update_inferences; game_data(i).gs() is the game in slot i."""
    import alphazero as az

    def evaluator(canon):
        n = canon.shape[0]
        s = (canon.reshape(n, -1).astype(np.float64) * np.linspace(0.5, 1.5, 168)).sum(1).astype(np.float32)
        v = np.stack([0.3 + 0.1 * np.sin(s), 0.3 - 0.1 * np.sin(s), np.full(n, 0.4)], 1).astype(np.float32)
        pi = np.abs(np.sin(s[:, None] * np.arange(1, 8, dtype=np.float32))) + 0.05
        return v, (pi / pi.sum(1, keepdims=True)).astype(np.float32)

    def params():
        pp = az.PlayParams()
        pp.games_to_play, pp.concurrent_games, pp.max_batch_size = 6, 6, 6
        pp.mcts_visits = [25, 25]
        pp.model_groups = [0, 0]
        pp.cpuct, pp.fpu_reduction = 1.25, 0.25
        return pp

    a = az.PlayManager(az.Connect4GS(), params(), seed=13, log_moves=True)
    batch = np.zeros((6, 4, 6, 7), np.float32)
    while a.remaining_games() > 0:
        idx = a.build_batch(0, batch)
        if idx:
            v, pi = evaluator(batch[: len(idx)])
            a.update_inferences(0, idx, v, pi)
    b = az.PlayManager(az.Connect4GS(), params(), seed=13, log_moves=True)
    checked_gs = False
    while b.remaining_games() > 0:
        idx = b.pop_games_upto(0, 4)
        if not idx:
            continue
        for i in idx:
            gd = b.game_data(i)
            canon = gd.canonical()
            if not checked_gs:                      # the slot's own game: stones on the board = its turn counter
                g = gd.gs()
                assert int(np.asarray(g.canonicalized())[:2].sum()) == g.current_turn()
                assert np.array_equal(gd.valid_moves(), g.valid_moves())
                checked_gs = True
            v, pi = evaluator(canon[None])
            gd.v()[:] = v[0]; gd.pi()[:] = pi[0]
            b.push_inference(i)
    assert b.pop_game(0) is None
    assert np.array_equal(a.move_log()[0], b.move_log()[0]) and np.array_equal(a.move_log()[1], b.move_log()[1])
    with pytest.raises(IndexError):
        b.game_data(6)


def test_hash_game_state_equality_semantics():
    """hash_game_state (game_state.h:141-156): equal states hash equal, transpositions too, different states differ."""
    import alphazero as az
    a = az.Connect4GS(); b = az.Connect4GS()
    for m in (3, 4, 2, 5):
        a.play_move(m)
    for m in (2, 5, 3, 4):
        b.play_move(m)
    c = az.Connect4GS(); c.play_move(3)
    assert az.hash_game_state(a) == az.hash_game_state(b) == az.hash_game_state(a.copy())
    assert az.hash_game_state(a) != az.hash_game_state(c)


# ---- EvalType::PLAYOUT and playout_eval (game_state.cc:10-95, play_manager.cc:580-582) ------------------------------
def test_playout_eval_matches_oracle_and_reference_statistics(oracle):
    import alphazero as az
    states, ogames = [], []
    rng = np.random.default_rng(3)
    for i in range(64):
        g = az.Connect4GS(); og = oracle.Game(oracle.GAME_CONNECT4)
        for _ in range(int(rng.integers(0, 12))):
            legal = np.flatnonzero(g.valid_moves())
            if g.scores() is not None or len(legal) == 0:
                break
            m = int(rng.choice(legal)); g.play_move(m); og.play(m)
        if g.scores() is not None:
            continue
        states.append(g); ogames.append(og)
    seeds = np.arange(1000, 1000 + len(states), dtype=np.uint64)
    vs, pis = az.playout_eval_batch(states, seeds)
    for g, og, sd, v, pi in zip(states, ogames, seeds, vs, pis):
        ov, opi = oracle.playout_eval(og, int(sd))
        assert np.array_equal(v, ov) and np.array_equal(pi, opi)
        assert v.sum() == 1.0 and set(np.unique(v)) <= {0.0, 1.0}                 # a terminal score vector
        valid = np.asarray(g.valid_moves(), dtype=np.float32)
        assert np.array_equal(pi, valid / valid.sum())                             # uniform over the legal moves
    v1, pi1 = az.playout_eval(states[0], seed=5)
    assert np.array_equal(v1, az.playout_eval(states[0], seed=5)[0])
    # random Connect4 playouts from the empty board: first player ~55.6 %, second ~44.2 %, draws ~0.25 %
    n = 4000
    vs, _ = az.playout_eval_batch([az.Connect4GS()] * n, np.arange(n, dtype=np.uint64))
    mean = np.mean(vs, 0)
    assert abs(mean[0] - 0.556) < 0.03 and abs(mean[1] - 0.442) < 0.03 and mean[2] < 0.01


def test_playout_seats_match_oracle(oracle):
    """EvalType.PLAYOUT seats (alone and against a RANDOM seat): same moves / counts / scores as the oracle, whose rollout
    stream is the slot's third pcg32 stream."""
    import alphazero as az
    for evals in ([az.EvalType.PLAYOUT, az.EvalType.PLAYOUT], [az.EvalType.PLAYOUT, az.EvalType.RANDOM]):
        pp = az.PlayParams()
        pp.games_to_play, pp.concurrent_games, pp.max_batch_size = 6, 6, 6
        pp.mcts_visits = [40, 30]
        pp.eval_type = evals
        pp.cpuct, pp.fpu_reduction, pp.history_enabled = 1.25, 0.25, True
        seed = 91
        pm = az.PlayManager(az.Connect4GS(), pp, seed=seed, log_moves=True)
        pm.play()
        rows, counts = pm.move_log()
        total = np.zeros(3, np.float32)
        for s in range(6):
            one = az.PlayParams(); one.__dict__.update(pp.__dict__)
            one.games_to_play, one.concurrent_games = 1, 1
            o = oracle.PlayManager(oracle.GAME_CONNECT4, one, oracle.slot_seed(seed, s), per_slot_rng=False)
            o.run()
            orows, ocounts = o.moves()
            sel = rows[:, 0] == s
            assert np.array_equal(rows[sel][:, 1:], orows[:, 1:]), s
            assert np.array_equal(counts[sel], ocounts), s
            total += o.scores()
        assert np.array_equal(pm.scores(), total)
