"""-m gpu: the reference's own PlayManager test cases (play_manager_test.cc:12-308) against the device engine, case by case.
(The constructor cases that must throw are CPU tests: tests/test_abi.py; MultiThreaded has no counterpart — the engine has
no worker threads.)"""
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def az():
    import alphazero
    return alphazero


def _params(az, games, concurrent, **kw):
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games = games, concurrent
    pp.mcts_visits = [10, 10]
    pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
    for k, v in kw.items():
        setattr(pp, k, v)
    return pp


def test_basic(az):                                   # TEST(PlayManager, Basic), :12-29
    pm = az.PlayManager(az.Connect4GS(), _params(az, 32, 8, history_enabled=True, playout_cap_randomization=True))
    pm.play()
    assert pm.games_completed() == 32 and pm.scores().sum() == 32


def test_stop_early(az):                              # StopEarly, :58-72
    pm = az.PlayManager(az.Connect4GS(), _params(az, 1000000, 8))
    t = threading.Thread(target=pm.play)
    t.start()
    time.sleep(0.05)
    pm.stop()
    t.join(timeout=5)
    assert not t.is_alive() and pm.stopped() and pm.remaining_games() == 0


def test_seat_overrides_default_fill(az):             # SeatOverridesDefaultFill, :113-123; NewSeatFieldDefaultsFromEmpty, :240-251
    az.PlayManager(az.Connect4GS(), _params(az, 4, 2, epsilon=0.25))
    az.PlayManager(az.Connect4GS(), _params(az, 4, 2))


def test_per_seat_overrides(az):                      # PerSeatOverrides, :125-147
    pm = az.PlayManager(az.Connect4GS(), _params(az, 32, 8, seat_perms=[[0, 1], [1, 0]], seat_epsilon=[[0.25, 0.0], [0.0, 0.25]],
                                                  seat_mcts_root_temp=[[1.25, 1.0], [1.0, 1.25]], seat_root_fpu_zero=[[1, 0], [0, 1]],
                                                  seat_visits=[[10, 10], [10, 10]]))
    pm.play()
    assert pm.games_completed() == 32
    assert pm.perm_games_completed(0) + pm.perm_games_completed(1) == 32


def test_init_order_per_perm(az):                     # InitOrderPerPermMctsSettings, :149-183: game i starts under perm i % perms
    pm = az.PlayManager(az.Connect4GS(), _params(az, 100, 4, seat_perms=[[0, 1], [1, 0]], seat_epsilon=[[0.25, 0.0], [0.0, 0.5]],
                                                  seat_mcts_root_temp=[[1.25, 1.0], [1.0, 1.5]], seat_root_fpu_zero=[[1, 0], [0, 1]],
                                                  seat_visits=[[10, 10], [10, 10]]))
    pm.round()                                        # the slots exist after the first round
    for i in range(4):
        assert pm.game_data(i).perm_index == i % 2


def test_runs_with_g3_opt_in_and_resign(az):          # RunsWithG3OptInAndResign, :253-277
    pm = az.PlayManager(az.Connect4GS(), _params(az, 16, 4, seat_gumbel_use_improved_policy=[[0, 0]],
                                                  seat_resign_threshold=[[-1.5, -1.5]], seat_resign_consecutive=[[1, 1]]))
    pm.play()
    assert pm.games_completed() == 16 and pm.resign_scores().sum() == 0      # a threshold below any W - L never fires


def test_aggressive_resign_terminates_games(az):      # AggressiveResignTerminatesGames, :279-308
    pm = az.PlayManager(az.Connect4GS(), _params(az, 8, 4, seat_resign_threshold=[[2.0, 2.0]], seat_resign_consecutive=[[1, 1]]),
                        log_moves=True)
    pm.play()
    assert pm.games_completed() == 8
    assert pm.resign_scores().sum() == 8              # every game ends by resignation ...
    rows, _ = pm.move_log()
    assert len(rows) == 8 and (rows[:, 3] == 0).all()  # ... at the very first move
    assert pm.scores()[1] == 8                        # the first mover resigns: the opponent is credited


def test_seat_visits_overrides_visit_count_with_worker_threads(az):
    """test_cache.py:543-592: PLAYOUT evaluators, one model group on both seats, seat_visits 200 vs 1 swapped by the
    permutation, TWO worker threads in play() at once: the seat with 200 visits wins more in both permutations."""
    pp = _params(az, 64, 8, max_batch_size=8, mcts_visits=[200, 1], eval_type=[az.EvalType.PLAYOUT] * 2, model_groups=[0, 0],
                 seat_perms=[[0, 0], [0, 0]], seat_visits=[[200, 1], [1, 200]], cpuct=1.25, fpu_reduction=0.25)
    pm = az.PlayManager(az.Connect4GS(), pp, seed=4)
    workers = [threading.Thread(target=pm.play) for _ in range(2)]
    for t in workers: t.start()
    for t in workers: t.join()
    assert pm.games_completed() == 64
    p0, p1 = pm.perm_scores(0), pm.perm_scores(1)
    assert pm.perm_games_completed(0) + pm.perm_games_completed(1) == 64
    assert p0[0] > p0[1] and p1[1] > p1[0]


def test_play_with_nn_seats_waits_for_the_batcher(az):
    """the GameRunner shape (game_runner.py:651-726): worker threads sit in play() while a batcher thread drives
    build_batch / update_inferences; everybody returns when the games are done."""
    pp = _params(az, 12, 6, max_batch_size=6, mcts_visits=[16, 16], eval_type=[], model_groups=[0, 0])
    pm = az.PlayManager(az.Connect4GS(), pp, seed=6)
    workers = [threading.Thread(target=pm.play) for _ in range(3)]
    for t in workers: t.start()
    batch = np.zeros((6, 4, 6, 7), np.float32)
    while pm.remaining_games() > 0:
        idx = pm.build_batch(0, batch)
        if idx:
            n = len(idx)
            pm.update_inferences(0, idx, np.full((n, 3), 1 / 3, np.float32), np.full((n, 7), 1 / 7, np.float32))
    for t in workers: t.join(timeout=10)
    assert not any(t.is_alive() for t in workers) and pm.games_completed() == 12
