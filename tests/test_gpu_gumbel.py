"""-m gpu: Gumbel AlphaZero search on the HIP engine (mcts.cc:24-401, play_manager.cc:367-417,
525-539) against the CPU oracle: move sequences, root visit counts, RNG stream position per move and the
emitted PlayHistory rows (the improved policy pi') are compared bit for bit.  Both sides define
logf/expf as float(double fn) (numerics)."""
import numpy as np
import pytest

from test_gpu_parity_connect4 import _az, _compare_slotwise, _params

pytestmark = pytest.mark.gpu


def _history_multiset(c, v, p):
    return sorted((a.tobytes(), b.tobytes(), d.tobytes()) for a, b, d in zip(c, v, p))


@pytest.mark.parametrize("cfg", [
    dict(),                                                                   # G1 acting, full searches only
    dict(gumbel_m=4, mcts_visits=[13, 13]),                                   # mcts_test.cc:801-822 schedule
    dict(gumbel_full=True),                                                   # pi'-matching below the root
    dict(gumbel_c_visit=20.0, gumbel_c_scale=0.5, gumbel_m=3),
    dict(epsilon=0.25, mcts_root_temp=1.25, root_fpu_zero=True),              # Dirichlet only on reused roots
    dict(playout_cap_randomization=True, playout_cap_depth=10, playout_cap_percent=0.6, epsilon=0.25),
    dict(playout_cap_randomization=True, playout_cap_depth=12, playout_cap_percent=0.5, fast_search_uses_gumbel=True),
    dict(tree_reuse=False),                                                   # fresh trees lose the target: PUCT + pick_move(probs(0))
    dict(resign_percent=0.05, resign_playthrough_percent=0.3, temp_decay_half_life=8.0, final_temp=0.3),
])
def test_gumbel_playmanager_tiers(oracle, cfg):
    az = _az()
    base = dict(games_to_play=32, concurrent_games=16, mcts_visits=[40, 40], cpuct=1.25, fpu_reduction=0.25,
                gumbel_enabled=True)
    base.update(cfg)
    pp = _params(az, **base)
    seed = 777
    pm, rows, hist, tot_scores, n_hist = _compare_slotwise(az, oracle, pp, seed=seed)
    assert np.array_equal(pm.scores(), tot_scores)
    assert len(hist[0]) == n_hist
    # history rows: pi target = gumbel_improved_policy (play_manager.cc:411-417), exact
    games = pm.slot_games()
    rows_orc = []
    for s in range(pp.concurrent_games):
        if games[s] == 0:
            continue
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games = int(games[s]), 1
        o = oracle.PlayManager(oracle.GAME_CONNECT4, one, oracle.slot_seed(seed, s), per_slot_rng=False)
        o.run()
        rows_orc += _history_multiset(*o.history())
    assert _history_multiset(*hist) == sorted(rows_orc)
    assert np.allclose(hist[2].sum(1), 1.0, atol=1e-5)


def test_gumbel_visit_schedule_on_device(oracle):
    """With m=4 and 13 sims every full search visits exactly {5,5,1,1} root children
    (SequentialHalvingVisitDistribution, mcts_test.cc:801-822) when the tree is fresh."""
    az = _az()
    pp = _params(az, games_to_play=8, concurrent_games=8, mcts_visits=[13, 13], gumbel_enabled=True, gumbel_m=4,
                 tree_reuse=True, history_enabled=False)
    pm = az.PlayManager(az.Connect4GS(), pp, seed=3, log_moves=True)
    pm.play()
    rows, counts = pm.move_log()
    first = counts[rows[:, 3] == 0]          # turn 0: fresh tree, no reused visits
    assert len(first) == 8
    for c in first:
        assert sorted((int(x) for x in c if x > 0), reverse=True) == [5, 5, 1, 1] and c.sum() == 12


def test_gumbel_m_limit():
    az = _az()
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.mcts_visits = 1, 1, [10, 10]
    pp.gumbel_enabled, pp.gumbel_m = True, 65
    with pytest.raises(RuntimeError, match="gumbel_m"):
        az.PlayManager(az.Connect4GS(), pp)


@pytest.mark.parametrize("cfg", [
    dict(),                                                        # configs/tawlbwrdd.yaml:24-25: gumbel on, capped searches PUCT
    dict(gumbel_m=8, gumbel_full=True),
    dict(playout_cap_randomization=True, playout_cap_depth=8, playout_cap_percent=0.5, epsilon=0.25, shaped_dirichlet=True),
    dict(playout_cap_randomization=True, playout_cap_depth=8, playout_cap_percent=0.5, fast_search_uses_gumbel=True),
])
def test_gumbel_tawlbwrdd_tiers(oracle, cfg):
    """Gumbel on the wide-game engine (one wave per slot, lane-resident survivors): moves, visit counts, RNG
    position and pi' history rows equal the oracle's."""
    az = _az()
    pp = az.PlayParams()
    pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
    pp.history_enabled = True
    pp.games_to_play, pp.concurrent_games, pp.mcts_visits = 4, 4, [24, 24]
    pp.cpuct, pp.fpu_reduction, pp.gumbel_enabled = 1.25, 0.25, True
    for k, v in cfg.items():
        setattr(pp, k, v)
    seed = 99
    pm = az.PlayManager(az.TawlbwrddGS(), pp, seed=seed, log_moves=True)
    pm.play()
    rows, counts = pm.move_log()
    hist = pm.history()
    rows_orc = []
    for s in range(4):
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games = 1, 1
        o = oracle.PlayManager(oracle.GAME_TAWLBWRDD, one, oracle.slot_seed(seed, s), per_slot_rng=False)
        o.run()
        orows, ocounts = o.moves()
        sel = rows[:, 0] == s
        assert np.array_equal(rows[sel][:, 1:], orows[:, 1:]), s
        assert np.array_equal(counts[sel], ocounts), s
        rows_orc += _history_multiset(*o.history())
    assert _history_multiset(*hist) == sorted(rows_orc)
