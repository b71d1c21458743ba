"""GPU: GameState::symmetries on the device (azmi_symmetries / azmi_tafl_symmetries) vs the oracle and
the reference's own known-answer tables (tafl_helper_test.cc).  Pure permutations -> bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def az():
    import alphazero
    return alphazero


@pytest.fixture(scope="module")
def oracle():
    import oracle_api
    return oracle_api


def _ploc(n, fh, fw, hm, loc):
    return (fh * n + fw) * (2 * n) + (n if hm else 0) + loc


def test_connect4_batch_matches_oracle(az, oracle):
    rng = np.random.default_rng(11)
    n = 257
    c = rng.random((n, 4, 6, 7), dtype=np.float32); v = rng.random((n, 3), dtype=np.float32)
    pi = rng.random((n, 7), dtype=np.float32)
    oc, ov, op = az.symmetries_batch(az.Connect4GS, c, v, pi)
    assert oc.shape == (n, 2, 4, 6, 7) and ov.shape == (n, 2, 3) and op.shape == (n, 2, 7)
    for i in range(0, n, 16):
        ec, ev, ep = oracle.symmetries(oracle.SYM_CONNECT4, c[i], v[i], pi[i])
        assert np.array_equal(oc[i], ec) and np.array_equal(ov[i], ev) and np.array_equal(op[i], ep)
    assert np.array_equal(oc[:, 1], c[:, :, :, ::-1]) and np.array_equal(op[:, 1], pi[:, ::-1])


def test_tawlbwrdd_batch_matches_oracle(az, oracle):
    rng = np.random.default_rng(12)
    n = 33
    c = rng.random((n, 7, 11, 11), dtype=np.float32); v = rng.random((n, 3), dtype=np.float32)
    pi = rng.random((n, 2662), dtype=np.float32)
    oc, ov, op = az.symmetries_batch(az.TawlbwrddGS, c, v, pi)
    assert oc.shape == (n, 8, 7, 11, 11)
    for i in range(n):
        ec, ev, ep = oracle.symmetries(oracle.SYM_TAFL_EIGHT, c[i], v[i], pi[i])
        assert np.array_equal(oc[i], ec) and np.array_equal(ov[i], ev) and np.array_equal(op[i], ep)


def test_reference_spot_tables_5x5(az):
    """TEST(TaflHelper, Mirror) and TEST(TaflHelper, Rot90), tafl_helper_test.cc:73-241: image 4 of
    eightSym is mirrorWidth(base), image 1 is rot90Clockwise(base)."""
    n = 5
    canon = np.zeros((1, 3, n, n), np.float32); pi = np.zeros((1, n * n * 2 * n), np.float32)
    canon[0, 0, 2, 1] = 1; canon[0, 1, 2, 3] = 1; canon[0, 1, 4, 1] = 1; canon[0, 2, 2, 2] = 1
    for hm, loc in ((False, 0), (False, 2), (True, 0), (True, 1), (True, 3)):
        pi[0, _ploc(n, 2, 1, hm, loc)] = 1
    for hm, loc in ((False, 1), (False, 4), (True, 0), (True, 3)):
        pi[0, _ploc(n, 2, 2, hm, loc)] = 1
    v = np.array([[0.5, -0.25, 0.75]], np.float32)
    oc, ov, op = az.tafl_symmetries(n, canon, v, pi)

    def expect(cells, wm, hm_):
        c = np.zeros((3, n, n), np.float32); p = np.zeros(n * n * 2 * n, np.float32)
        for x in cells:
            c[x] = 1
        for (fh, fw), locs in wm.items():
            for x in locs:
                p[_ploc(n, fh, fw, False, x)] = 1
        for (fh, fw), locs in hm_.items():
            for y in locs:
                p[_ploc(n, fh, fw, True, y)] = 1
        return c, p

    mc, mp = expect([(0, 2, 3), (1, 2, 1), (1, 4, 3), (2, 2, 2)], {(2, 3): (2, 4), (2, 2): (0, 3)}, {(2, 3): (0, 1, 3), (2, 2): (0, 3)})
    rc, rp = expect([(0, 1, 2), (1, 3, 2), (1, 1, 0), (2, 2, 2)], {(1, 2): (1, 3, 4), (2, 2): (1, 4)}, {(1, 2): (0, 2), (2, 2): (1, 4)})
    assert np.array_equal(oc[0, 4], mc) and np.array_equal(op[0, 4], mp)
    assert np.array_equal(oc[0, 1], rc) and np.array_equal(op[0, 1], rp)
    assert np.array_equal(oc[0, 0], canon[0]) and np.array_equal(op[0, 0], pi[0])
    assert all(np.array_equal(ov[0, i], v[0]) for i in range(8))


@pytest.mark.parametrize("n", [5, 7, 11])
def test_group_properties_and_oracle(az, oracle, n):
    """MakeDistinct samples (tafl_helper_test.cc:16-37): bijection, rot^4 = id, mirror^2 = id, 8 distinct
    images, and equality with the oracle for every board size the Tafl family uses."""
    pi = np.arange(1, n * n * 2 * n + 1, dtype=np.float32)[None]
    canon = np.arange(1, 3 * n * n + 1, dtype=np.float32).reshape(1, 3, n, n)
    v = np.array([[0.1, 0.2, 0.3]], np.float32)
    oc, ov, op = az.tafl_symmetries(n, canon, v, pi)
    ec, ev, ep = oracle.symmetries(oracle.SYM_TAFL_EIGHT, canon[0], v[0], pi[0])
    assert np.array_equal(oc[0], ec) and np.array_equal(op[0], ep) and np.array_equal(ov[0], ev)
    for i in range(8):
        assert np.array_equal(np.sort(op[0, i]), pi[0])
        for j in range(i + 1, 8):
            assert not np.array_equal(op[0, i], op[0, j])
    # rot90 of image 3 is the identity again; mirror of image 4 is the base
    oc2, _, op2 = az.tafl_symmetries(n, oc[0, 3:5], np.repeat(v, 2, 0), op[0, 3:5])
    assert np.array_equal(oc2[0, 1], canon[0]) and np.array_equal(op2[0, 1], pi[0])
    assert np.array_equal(oc2[1, 4], canon[0]) and np.array_equal(op2[1, 4], pi[0])


def test_gamestate_symmetries_and_device_tensors(az, oracle):
    import torch
    rng = np.random.default_rng(5)
    c = rng.random((4, 6, 7), dtype=np.float32); pi = rng.random(7, dtype=np.float32); v = np.array([0, 1, 0], np.float32)
    syms = az.Connect4GS().symmetries(az.PlayHistory(c, v, pi))          # game_runner.py:1084-1092 usage
    assert len(syms) == 2
    ec, ev, ep = oracle.symmetries(oracle.SYM_CONNECT4, c, v, pi)
    for i, s in enumerate(syms):
        assert np.array_equal(np.array(s.canonical()), ec[i]) and np.array_equal(np.array(s.v()), ev[i]) and np.array_equal(np.array(s.pi()), ep[i])
    # device-resident path: torch CUDA tensors in and out, no host staging
    n = 1000
    tc = torch.rand(n, 7, 11, 11, device="cuda"); tv = torch.rand(n, 3, device="cuda"); tp = torch.rand(n, 2662, device="cuda")
    oc, ov, op = az.symmetries_batch(az.TawlbwrddGS, tc, tv, tp)
    torch.cuda.synchronize()
    hc, hv, hp = az.symmetries_batch(az.TawlbwrddGS, tc.cpu().numpy(), tv.cpu().numpy(), tp.cpu().numpy())
    assert np.array_equal(oc.cpu().numpy(), hc) and np.array_equal(ov.cpu().numpy(), hv) and np.array_equal(op.cpu().numpy(), hp)
    assert torch.equal(oc[:, 0], tc) and torch.equal(op[:, 0], tp)


def test_empty_and_bad_shapes(az):
    c = np.zeros((0, 4, 6, 7), np.float32); v = np.zeros((0, 3), np.float32); pi = np.zeros((0, 7), np.float32)
    oc, ov, op = az.symmetries_batch(az.Connect4GS, c, v, pi)
    assert oc.shape == (0, 2, 4, 6, 7)
    with pytest.raises(RuntimeError):
        az.symmetries_batch(az.Connect4GS, np.zeros((1, 4, 6, 6), np.float32), np.zeros((1, 3), np.float32), np.zeros((1, 7), np.float32))
