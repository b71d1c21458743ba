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
