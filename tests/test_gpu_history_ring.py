"""-m gpu: the finished-sample store is a ring (the reference's history queue is unbounded and drained by hist_saver,
game_runner.py:729-747): a run that is drained while it plays produces, through a ring much smaller than the run, exactly
the rows of the same run with room for everything; an undrained ring stops the engine loudly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(az, games, S, sims=40):
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.max_batch_size = games, S, S
    pp.mcts_visits = [sims, sims]
    pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
    pp.history_enabled = True
    return pp


def _drain(pm, chw, M):
    c = np.zeros((4096,) + chw, np.float32); v = np.zeros((4096, 3), np.float32); p = np.zeros((4096, M), np.float32)
    n = pm.build_history_batch(c, v, p)
    return c[:n].copy(), v[:n].copy(), p[:n].copy()


def _rows_sorted(parts):
    c = np.concatenate([x[0] for x in parts]); v = np.concatenate([x[1] for x in parts]); p = np.concatenate([x[2] for x in parts])
    flat = np.concatenate([c.reshape(len(c), -1), v, p], 1)
    return flat[np.lexsort(flat.T[::-1])]


@pytest.mark.parametrize("game", ["connect4", "brandubh"])
def test_a_drained_ring_yields_the_rows_of_an_unbounded_run(game):
    import alphazero as az
    G = az.Connect4GS if game == "connect4" else az.BrandubhGS
    S, games = 8, 96 if game == "connect4" else 24
    chw, M = tuple(G.CANONICAL_SHAPE()), G.NUM_MOVES()
    ref = az.PlayManager(G(), _params(az, games, S), seed=77)
    ref.play()
    want = _rows_sorted([_drain(ref, chw, M) for _ in range(8)])
    cap = 2 * S * 42 if game == "connect4" else 320            # far smaller than the run's rows; wraps several times
    pm = az.PlayManager(G(), _params(az, games, S), seed=77, history_capacity=cap)
    got = []
    while pm.remaining_games() > 0 or pm.games_completed() < games:
        for _ in range(32):
            pm.round()
        got.append(_drain(pm, chw, M))
        if pm.games_completed() >= games:
            break
    got.append(_drain(pm, chw, M))
    have = _rows_sorted(got)
    assert len(want) > cap, "the test must wrap the ring"
    assert have.shape == want.shape and np.array_equal(have, want)


def test_take_history_device_follows_the_ring():
    import torch
    import alphazero as az
    S, games = 8, 64
    ref = az.PlayManager(az.Connect4GS(), _params(az, games, S), seed=5)
    ref.play()
    want = _rows_sorted([_drain(ref, (4, 6, 7), 7) for _ in range(4)])
    pm = az.PlayManager(az.Connect4GS(), _params(az, games, S), seed=5, history_capacity=2 * S * 42)
    got = []
    while pm.games_completed() < games:
        for _ in range(32):
            pm.round()
        c, v, p, meta = pm.take_history_device(torch.device("cuda", 0))
        got.append((c.cpu().numpy(), v.cpu().numpy(), p.cpu().numpy()))
    c, v, p, meta = pm.take_history_device(torch.device("cuda", 0))
    got.append((c.cpu().numpy(), v.cpu().numpy(), p.cpu().numpy()))
    assert np.array_equal(_rows_sorted(got), want)
    assert pm.hist_count() == 0


def test_an_undrained_ring_stops_the_engine_loudly():
    import alphazero as az
    pm = az.PlayManager(az.Connect4GS(), _params(az, 64, 8), seed=3, history_capacity=60)
    with pytest.raises(RuntimeError, match="overflow mask 0x2"):
        pm.play()
