"""-m gpu: Tawlbwrdd rules kernels (bitboards) against the oracle (dense boards): tier T0."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _playouts(orc, n, length, seed):
    rng = np.random.default_rng(seed)
    moves = -np.ones((n, length), np.int32)
    finals = []
    for g in range(n):
        game = orc.Game(orc.GAME_TAWLBWRDD)
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
    import alphazero as az
    moves, finals = _playouts(oracle, 300, 160, seed=11)
    out = az.game_replay(az.TawlbwrddGS, moves)
    assert (out["status"] == 0).all()
    n_term = 0
    for g, game in enumerate(finals):
        assert np.array_equal(out["valid"][g], game.valid()), g
        sc = game.scores()
        if sc is None:
            assert (out["scores"][g] == -1).all(), g
        else:
            n_term += 1
            assert np.array_equal(out["scores"][g], sc), g
        assert np.array_equal(out["canonical"][g], game.canonical()), g
        assert out["player"][g] == game.player() and out["turn"][g] == game.turn(), g
    assert n_term > 5


def test_threefold_repetition_and_captures(oracle):
    """tawlbwrdd_gs_test.cc:9-53 (threefold repetition -> side to move credited) + a custodial capture."""
    import alphazero as az
    W = 11

    def mv(fh, fw, th, tw):
        base = (fh * W + fw) * 22
        return base + (W + th if fw == tw else tw)

    seq = [mv(0, 4, 0, 3), mv(2, 5, 2, 4), mv(0, 3, 0, 4), mv(2, 4, 2, 5)] * 2
    out = az.game_replay(az.TawlbwrddGS, np.array([seq], np.int32))
    assert out["scores"][0].tolist() == [1, 0, 0]
    assert out["canonical"][0][5].all() and out["canonical"][0][6].all()  # rep count 3: both planes set
    # attacker (1,4)->(2,4) next to defender (2,5), attacker already... capture needs two hostile sides:
    g = oracle.Game(oracle.GAME_TAWLBWRDD)
    cap = [mv(1, 4, 2, 4), mv(5, 2, 4, 2), mv(1, 6, 2, 6)]  # attackers sandwich the defender on (2,5)
    for m in cap:
        g.play(m)
    out = az.game_replay(az.TawlbwrddGS, np.array([cap], np.int32))
    assert np.array_equal(out["canonical"][0], g.canonical())
    assert out["canonical"][0][1].sum() == 11  # one of the 12 defenders is gone
    # an illegal move (jumping over a piece) is reported, like the reference's throw
    out = az.game_replay(az.TawlbwrddGS, np.array([[mv(0, 4, 0, 8)]], np.int32))
    assert out["status"][0] == -1


def _compare(az, orc, pp, seed, S):
    pm = az.PlayManager(az.TawlbwrddGS(), pp, seed=seed, log_moves=True)
    pm.play()
    rows, counts = pm.move_log()
    games = pm.slot_games()
    tot = np.zeros(3, np.float32)
    nhist = 0
    for s in range(S):
        if games[s] == 0:
            continue
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games = int(games[s]), 1
        o = orc.PlayManager(orc.GAME_TAWLBWRDD, one, orc.slot_seed(seed, s), per_slot_rng=False)
        o.run()
        orows, ocounts = o.moves()
        sel = (rows[:, 0] == s) & (rows[:, 1] < games[s])
        d, dc = rows[sel], counts[sel]
        assert d.shape == orows.shape, (s, d.shape, orows.shape)
        bad = np.flatnonzero((d[:, 1:] != orows[:, 1:]).any(1))
        assert bad.size == 0, (s, bad[:3], d[bad[:3]], orows[bad[:3]])
        assert np.array_equal(dc, ocounts), s
        tot += o.scores()
        nhist += o.counters()["hist_rows"]
    assert np.array_equal(pm.scores(), tot)
    return pm, nhist


def test_playmanager_exact_tier(oracle):
    """T2: Tawlbwrdd, RANDOM evaluator, eps = 0: moves, visit counts, RNG positions, scores."""
    import alphazero as az
    pp = az.PlayParams()
    pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
    pp.games_to_play, pp.concurrent_games = 6, 6
    pp.mcts_visits = [24, 24]
    pp.cpuct, pp.fpu_reduction = 1.25, 0.25
    pp.history_enabled = True
    pm, nhist = _compare(az, oracle, pp, seed=777, S=6)
    hc, hv, hp = pm.history()
    assert len(hc) == nhist and np.allclose(hp.sum(1), 1.0, atol=1e-5)


def test_known_answer_game(oracle):
    """seed 777, 30 visits, one game -> attackers win after 101 plies (through the oracle pin)."""
    import alphazero as az
    pp = az.PlayParams()
    pp.eval_type = [1, 1]
    pp.games_to_play, pp.concurrent_games = 1, 1
    pp.mcts_visits = [30, 30]
    pp.cpuct = 2.0
    pp.history_enabled = True
    o = oracle.PlayManager(oracle.GAME_TAWLBWRDD, pp, 777, per_slot_rng=False)
    o.run()
    assert o.scores().tolist() == [1, 0, 0] and o.stats()[0] == 101
    # the device uses slot_seed(seed, 0) for slot 0: compare against the oracle run with that stream
    pm, _ = _compare(az, oracle, pp, seed=777, S=1)


@pytest.mark.parametrize("cfg", [
    dict(epsilon=0.25, shaped_dirichlet=True, mcts_root_temp=1.25, policy_target_pruning=True, root_fpu_zero=True),
    dict(tree_reuse=False, start_temp=0.0),
    dict(temp_decay_half_life=12.0, final_temp=0.2, start_temp=1.2, epsilon=0.25),
])
def test_playmanager_option_tiers(oracle, cfg):
    import alphazero as az
    pp = az.PlayParams()
    pp.eval_type = [1, 1]
    pp.games_to_play, pp.concurrent_games = 4, 4
    pp.mcts_visits = [16, 16]
    pp.cpuct, pp.fpu_reduction = 1.25, 0.25
    pp.history_enabled = True
    for k, v in cfg.items():
        setattr(pp, k, v)
    _compare(az, oracle, pp, seed=31337, S=4)


def test_arena_compaction_is_invisible_to_the_search(oracle, monkeypatch):
    """k_compact (ping-pong arena halves) forced after EVERY move (AZMI_COMPACT_ABOVE=0): the games must still
    equal the oracle's move for move, count for count — the copy keeps child order and re-points the pending path."""
    import alphazero as az
    monkeypatch.setenv("AZMI_COMPACT_ABOVE", "0")
    pp = az.PlayParams()
    pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
    pp.history_enabled = True
    pp.games_to_play, pp.concurrent_games, pp.mcts_visits = 6, 6, [40, 40]
    pp.cpuct, pp.fpu_reduction, pp.epsilon, pp.shaped_dirichlet = 1.25, 0.25, 0.25, True
    seed = 31
    pm = az.PlayManager(az.TawlbwrddGS(), pp, seed=seed, log_moves=True)
    pm.play()
    rows, counts = pm.move_log()
    games = pm.slot_games()
    assert int(games.sum()) == 6
    for s in range(6):
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games = 1, 1
        o = oracle.PlayManager(oracle.GAME_TAWLBWRDD, one, oracle.slot_seed(seed, s), per_slot_rng=False)
        o.run()
        orows, ocounts = o.moves()
        sel = rows[:, 0] == s
        assert np.array_equal(rows[sel][:, 1:], orows[:, 1:]), s
        assert np.array_equal(counts[sel], ocounts), s
