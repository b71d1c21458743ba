"""CPU: pins the oracle (oracle/) against the reference's known answers.

Sources: tests/golden/known_answers.json (outputs of the reference itself recorded in SURVEY §8c and
the known-answer assertions of the reference's own gtest files) and, when built, the real libstdc++ +
reference pcg header (oracle/_ref/librngref.so).
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "known_answers.json")))


def test_rng_known_answers(oracle):
    assert oracle.pcg32_outputs(12345, 4).tolist() == KA["pcg32_seed_12345_first4"]
    assert oracle.shuffle_iota(12345, 7)[0].tolist() == KA["shuffle_iota7_seed_12345"]
    assert np.allclose(oracle.uniform01(12345, 2), KA["uniform_real_float_seed_12345_first2"], rtol=1e-7)
    assert np.array_equal(oracle.uniform01(12345, 1), np.float32([1411482639 / 2 ** 32]))
    assert np.allclose(oracle.gamma(12345, np.float32(10.83) / np.float32(7), 2), KA["gamma_10.83_over_7_seed_12345_first2"], rtol=1e-7)
    assert np.allclose(oracle.gumbel(12345, 2), KA["extreme_value_seed_12345_first2"], rtol=1e-6)


def test_rng_against_real_libstdcxx(oracle):
    """Integer paths are bit-exact; gamma / gumbel differ from glibc's float libm by a few ulp in < 0.2 % / 2 % of draws."""
    ref = oracle.ref_rng()
    if ref is None:
        pytest.skip("oracle/_ref/librngref.so not built (reference tree absent)")
    assert np.array_equal(oracle.pcg32_outputs(7, 100000), oracle.pcg32_outputs(7, 100000, which=ref))
    for n in (1, 2, 3, 4, 5, 6, 7, 8, 33, 112, 113, 250, 1000):
        assert np.array_equal(oracle.shuffle_iota(99 + n, n, 100), oracle.shuffle_iota(99 + n, n, 100, which=ref)), n
    assert np.array_equal(oracle.uniform01(5, 500000), oracle.uniform01(5, 500000, which=ref))
    for alpha in (10.83 / 7, 10.83 / 2, 0.3, 0.77, 5.0):
        for fresh in (False, True):
            a = oracle.gamma(77, alpha, 100000, fresh)
            b = oracle.gamma(77, alpha, 100000, fresh, which=ref)
            assert (a != b).mean() < 2e-3, (alpha, fresh)
            assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)) < 2e-6
    a, b = oracle.gumbel(77, 100000), oracle.gumbel(77, 100000, which=ref)
    assert (a != b).mean() < 0.03 and np.max(np.abs(a - b)) < 1e-5


def test_node_uct_known_answers(oracle):
    ka = KA["node_uct"]
    for n, sqrt_n, fpu, want in ka["cases"]:
        got = oracle.node_uct(0.0, ka["policy"], n, sqrt_n, ka["cpuct"], fpu)
        assert np.float32(got) == pytest.approx(want, rel=4e-7)  # EXPECT_FLOAT_EQ = 4 ulp


def test_node_best_child_known_answer(oracle):
    """mcts_test.cc:14-38: priors [.1,1.2,.3,.4,.5,.6,.7], root.n = 1 -> best child is move 1."""
    m = oracle.Mcts(2.0, 2, 7, seed=1)
    g = oracle.Game(oracle.GAME_CONNECT4)
    m.find_leaf(g)
    # raw (un-normalised) priors are what update_policy sets; normalisation keeps the argmax
    m.process_result([1 / 3, 1 / 3, 1 / 3], [0.1, 1.2, 0.3, 0.4, 0.5, 0.6, 0.7])
    m.find_leaf(g)
    m.process_result([1 / 3, 1 / 3, 1 / 3], [1 / 7] * 7)
    assert int(np.argmax(m.counts())) == 1


def test_mcts_known_answers(oracle):
    ka = KA["mcts_connect4"]
    g = oracle.Game(oracle.GAME_CONNECT4)
    for mv in ka["moves"]:
        g.play(mv)
    m = oracle.Mcts(ka["cpuct"], 2, 7, seed=ka["seed"])
    m.search_dumb(g, ka["sims"])
    assert m.counts().tolist() == ka["counts"]
    assert m.pick_move(m.probs(0.0)) == ka["pick_move_probs0"]
    kb = KA["mcts_connect4_second_position"]
    for seed in (1, 2, 3, 20240601):
        g = oracle.Game(oracle.GAME_CONNECT4)
        for mv in kb["moves"]:
            g.play(mv)
        m = oracle.Mcts(kb["cpuct"], 2, 7, seed=seed)
        m.search_dumb(g, kb["sims"])
        assert m.pick_move(m.probs(0.0)) == kb["pick_move_probs0"]


def _pm(oracle, game, ka):
    import types
    pp = types.SimpleNamespace(
        games_to_play=ka["games_to_play"], concurrent_games=ka["concurrent_games"], max_batch_size=1, max_cache_size=0,
        cache_shards=1, mcts_visits=ka["mcts_visits"], cpuct=2.0, start_temp=1.0, final_temp=1.0, temp_decay_half_life=0.0,
        history_enabled=True, tree_reuse=True, epsilon=0.0, mcts_root_temp=1.0, playout_cap_randomization=False,
        playout_cap_depth=25, playout_cap_percent=0.75, fpu_reduction=0.0, root_fpu_zero=False, shaped_dirichlet=False,
        policy_target_pruning=False, resign_percent=0.0, resign_playthrough_percent=0.0, eval_type=[1, 1])
    pm = oracle.PlayManager(game, pp, ka["seed"], per_slot_rng=False)
    pm.run()
    return pm


def test_playmanager_known_answers(oracle):
    ka = KA["playmanager_connect4"]
    pm = _pm(oracle, oracle.GAME_CONNECT4, ka)
    assert pm.scores().tolist() == ka["scores"]
    st = pm.stats()
    assert st[0] == ka["avg_game_length"]
    assert st[1] == pytest.approx(ka["avg_leaf_depth"], abs=5e-6)
    assert pm.counters()["hist_rows"] == ka["hist_count"]
    kt = KA["playmanager_tawlbwrdd"]
    pm = _pm(oracle, oracle.GAME_TAWLBWRDD, kt)
    assert pm.scores().tolist() == kt["scores"]
    st = pm.stats()
    assert st[0] == kt["game_length"]
    assert st[1] == pytest.approx(kt["avg_leaf_depth"], abs=5e-7)


# ---- connect4_gs_test.cc, re-expressed -------------------------------------------------------------
def _board(cells):
    b = np.zeros((2, 6, 7), np.int8)
    for p, h, w in cells:
        b[p, h, w] = 1
    return b


def test_connect4_valid_moves(oracle):  # connect4_gs_test.cc:54-72
    g = oracle.Game(oracle.GAME_CONNECT4)
    assert g.valid().tolist() == [1] * 7
    g = oracle.Game.connect4_from_board(_board([(0, 0, 3), (1, 0, 5)]), 0, 0)
    assert g.valid().tolist() == [1, 1, 1, 0, 1, 0, 1]


def test_connect4_play_move_stacks(oracle):  # connect4_gs_test.cc:75-101
    g = oracle.Game(oracle.GAME_CONNECT4)
    for i in range(6):
        g.play(3)
        c = g.canonical()
        assert c[i % 2, 5 - i, 3] == 1 and g.turn() == i + 1 and g.player() == (i + 1) % 2
    with pytest.raises(RuntimeError, match="Invalid move: You have a bug in your code."):
        g.play(3)


def test_connect4_win_states(oracle):  # connect4_gs_test.cc:104-171
    F = oracle.Game.connect4_from_board
    assert oracle.Game(oracle.GAME_CONNECT4).scores() is None
    row = [(0, 3, 0), (0, 3, 1), (0, 3, 2), (0, 3, 3)]
    assert F(_board(row), 0, 0).scores().tolist() == [1, 0, 0]
    assert F(_board(row[:2] + row[3:]), 0, 0).scores() is None
    col = [(1, 1, 2), (1, 2, 2), (1, 3, 2), (1, 4, 2)]
    base = row[:2] + row[3:]
    assert F(_board(base + col), 0, 0).scores().tolist() == [0, 1, 0]
    assert F(_board(base + [c for c in col if c != (1, 2, 2)]), 0, 0).scores() is None
    d1 = [(0, 1, 1), (0, 2, 2), (0, 3, 3), (0, 4, 4)]
    assert F(_board(d1), 0, 0).scores().tolist() == [1, 0, 0]
    assert F(_board([c for c in d1 if c != (0, 2, 2)]), 0, 0).scores() is None
    d2 = [(1, 1, 3), (1, 2, 2), (1, 3, 1), (1, 4, 0)]
    assert F(_board(d2), 0, 0).scores().tolist() == [0, 1, 0]
    assert F(_board([c for c in d2 if c != (1, 2, 2)]), 0, 0).scores() is None
    full_top = [(w % 2, 0, w) for w in range(7)]
    assert F(_board(full_top), 0, 0).scores().tolist() == [0, 0, 1]


def test_connect4_canonical(oracle):  # connect4_gs_test.cc:174-225
    g = oracle.Game(oracle.GAME_CONNECT4)
    want = np.zeros((4, 6, 7), np.float32); want[2] = 1
    assert np.array_equal(g.canonical(), want)
    g.play(0)
    want = np.zeros((4, 6, 7), np.float32); want[0, 5, 0] = 1; want[3] = 1
    assert np.array_equal(g.canonical(), want)
    g.play(0)
    want = np.zeros((4, 6, 7), np.float32); want[0, 5, 0] = 1; want[1, 4, 0] = 1; want[2] = 1
    assert np.array_equal(g.canonical(), want)


def test_tawlbwrdd_threefold_repetition(oracle):
    """tawlbwrdd_gs_test.cc:9-53: shuffling a piece back and forth -> third repetition ends the game,
    the side to move is credited (tawlbwrdd_gs.cc:350-364)."""
    g = oracle.Game(oracle.GAME_TAWLBWRDD)
    W = 11

    def mv(fh, fw, th, tw):
        base = (fh * W + fw) * 22
        return base + (W + th if fw == tw else tw)

    seq = [mv(0, 4, 0, 3), mv(2, 5, 2, 4), mv(0, 3, 0, 4), mv(2, 4, 2, 5)]
    for rep in range(2):
        for m in seq:
            assert g.scores() is None
            g.play(m)
    assert g.scores() is not None and g.scores().tolist() == [1, 0, 0]


def test_s3fifo_reference_behaviour(oracle):
    """s3fifo_cache_test.cc: promotion S->M on re-reference, ghost re-admission to Main, 2-bit freq cap."""
    c = oracle.Cache(4, 1, 4, 2, 1)
    for k in range(1, 5):
        c.insert(k, [k, k], [k])
    assert c.find(1) is not None        # freq(1) = 1
    c.insert(5, [5, 5], [5])            # evicts from Small: 1 is promoted to Main, 2 is evicted -> ghost
    assert c.find(2) is None and c.find(1) is not None and c.stats()["evictions"] == 1
    assert c.stats()["reinserts"] == 1  # the miss on 2 hit the ghost list
    c.insert(2, [2, 2], [2])            # ghost hit -> admitted to Main
    assert c.find(2)[0].tolist() == [2, 2]
    z = oracle.Cache(0, 1, 0, 2, 1)     # capacity 0: inserts are ignored
    z.insert(1, [1, 1], [1])
    assert z.find(1) is None and z.stats()["size"] == 0
    for _ in range(10):
        c.find(1)
    assert c.stats()["hits"] >= 10


# ---------------------------------------------------------------- symmetries (tafl_helper_test.cc)
def _ploc(n, fh, fw, height_move, new_loc):  # tafl_helper.h:7-14
    return (fh * n + fw) * (2 * n) + (n if height_move else 0) + new_loc


def _spot_base():  # the 5x5 sample of TEST(TaflHelper, Mirror / Rot90), tafl_helper_test.cc:76-103
    n = 5
    canon = np.zeros((3, n, n), np.float32); pi = np.zeros(n * n * 2 * n, np.float32)
    canon[0, 2, 1] = 1; canon[1, 2, 3] = 1; canon[1, 4, 1] = 1; canon[2, 2, 2] = 1
    for hm, loc in ((False, 0), (False, 2), (True, 0), (True, 1), (True, 3)):
        pi[_ploc(n, 2, 1, hm, loc)] = 1
    for hm, loc in ((False, 1), (False, 4), (True, 0), (True, 3)):
        pi[_ploc(n, 2, 2, hm, loc)] = 1
    return n, canon, np.array([0.25, -0.5, 0.125], np.float32), pi


def _expect(n, canon_cells, width_moves, height_moves):
    canon = np.zeros((3, n, n), np.float32); pi = np.zeros(n * n * 2 * n, np.float32)
    for c in canon_cells:
        canon[c] = 1
    for (fh, fw), locs in width_moves.items():
        for x in locs:
            pi[_ploc(n, fh, fw, False, x)] = 1
    for (fh, fw), locs in height_moves.items():
        for y in locs:
            pi[_ploc(n, fh, fw, True, y)] = 1
    return canon, pi


def test_symmetry_mirror_known_answer(oracle):  # tafl_helper_test.cc:73-157
    n, canon, v, pi = _spot_base()
    oc, ov, op = oracle.symmetries(oracle.SYM_TAFL_MIRROR, canon, v, pi)
    ec, ep = _expect(n, [(0, 2, 3), (1, 2, 1), (1, 4, 3), (2, 2, 2)],
                     {(2, 3): (2, 4), (2, 2): (0, 3)}, {(2, 3): (0, 1, 3), (2, 2): (0, 3)})
    assert np.array_equal(oc[0], ec) and np.array_equal(op[0], ep) and np.array_equal(ov[0], v)


def test_symmetry_rot90_known_answer(oracle):  # tafl_helper_test.cc:159-241
    n, canon, v, pi = _spot_base()
    oc, ov, op = oracle.symmetries(oracle.SYM_TAFL_ROT90, canon, v, pi)
    ec, ep = _expect(n, [(0, 1, 2), (1, 3, 2), (1, 1, 0), (2, 2, 2)],
                     {(1, 2): (1, 3, 4), (2, 2): (1, 4)}, {(1, 2): (0, 2), (2, 2): (1, 4)})
    assert np.array_equal(oc[0], ec) and np.array_equal(op[0], ep) and np.array_equal(ov[0], v)


def _distinct(n):  # MakeDistinct, tafl_helper_test.cc:16-37
    pi = np.arange(1, n * n * 2 * n + 1, dtype=np.float32)
    canon = np.arange(1, 3 * n * n + 1, dtype=np.float32).reshape(3, n, n)
    return canon, np.array([0.1, 0.2, 0.3], np.float32), pi


def _equivariant(n, out_pi, base_pi, xf):  # ExpectGeometricEquivariance, tafl_helper_test.cc:328-353
    span = 2 * n
    for m in range(n * n * span):
        nl = m % span; hm = nl >= n; nl -= n if hm else 0
        fh, fw = divmod(m // span, n)
        th, tw = (nl, fw) if hm else (fh, nl)
        nf, nt = xf(fh, fw), xf(th, tw)
        if nf == nt:
            continue
        nm = _ploc(n, nf[0], nf[1], False, nt[1]) if nf[0] == nt[0] else _ploc(n, nf[0], nf[1], True, nt[0])
        assert out_pi[nm] == base_pi[m], (n, m)


def test_symmetry_group_properties(oracle):  # tafl_helper_test.cc:268-326, 355-375
    for n in (5, 7, 11):
        canon, v, pi = _distinct(n)
        rc, rv, rp = oracle.symmetries(oracle.SYM_TAFL_ROT90, canon, v, pi)
        assert np.array_equal(np.sort(rp[0]), np.sort(pi)) and np.array_equal(rv[0], v)
        c4, p4 = rc[0], rp[0]
        for _ in range(3):
            c4, _, p4 = oracle.symmetries(oracle.SYM_TAFL_ROT90, c4, v, p4); c4, p4 = c4[0], p4[0]
        assert np.array_equal(c4, canon) and np.array_equal(p4, pi)
        _equivariant(n, rp[0], pi, lambda h, w: (w, n - 1 - h))
        mc, mv, mp = oracle.symmetries(oracle.SYM_TAFL_MIRROR, canon, v, pi)
        assert np.array_equal(np.sort(mp[0]), np.sort(pi))
        m2c, _, m2p = oracle.symmetries(oracle.SYM_TAFL_MIRROR, mc[0], v, mp[0])
        assert np.array_equal(m2c[0], canon) and np.array_equal(m2p[0], pi)
        _equivariant(n, mp[0], pi, lambda h, w: (h, n - 1 - w))
    canon, v, pi = _distinct(7)
    ec, ev, ep = oracle.symmetries(oracle.SYM_TAFL_EIGHT, canon, v, pi)
    assert ec.shape[0] == 8 and np.array_equal(ec[0], canon) and np.array_equal(ep[0], pi)
    for i in range(8):
        assert np.array_equal(np.sort(ep[i]), np.sort(pi)) and np.array_equal(ev[i], v)
        for j in range(i + 1, 8):
            assert not np.array_equal(ep[i], ep[j])


def test_symmetry_connect4(oracle):  # connect4_gs.cc:151-170
    rng = np.random.default_rng(3)
    canon = rng.random((4, 6, 7), dtype=np.float32); pi = rng.random(7, dtype=np.float32)
    v = np.array([1, 0, 0], np.float32)
    oc, ov, op = oracle.symmetries(oracle.SYM_CONNECT4, canon, v, pi)
    assert np.array_equal(oc[0], canon) and np.array_equal(op[0], pi)
    assert np.array_equal(oc[1], canon[:, :, ::-1]) and np.array_equal(op[1], pi[::-1]) and np.array_equal(ov[1], v)


# ---------------------------------------------------------------- Gumbel (test_gumbel.py, mcts_test.cc:706-900)
def test_gumbel_phase_plan_known_answers(oracle):
    # test_gumbel.py:250-256 (paper figure 1) and :268-296
    ph = oracle.seq_halving_phase_plan(16, 200)
    assert ph == [(16, 3), (8, 6), (4, 12), (2, 28)]
    assert sum(c * v for c, v in ph) == 200 and sum(v for _, v in ph) == 49
    assert oracle.seq_halving_phase_plan(1, 10) == [(1, 10)]
    assert oracle.seq_halving_phase_plan(4, 12) == [(4, 1), (2, 4)]          # mcts_test.cc:802-805
    assert sum(c * v for c, v in oracle.seq_halving_phase_plan(8, 8)) == 8
    for m in (2, 3, 4, 5, 7, 8, 16, 32):
        for n in (m, m + 1, 2 * m, 50, 200, 800):
            ph = oracle.seq_halving_phase_plan(m, n)
            tot = sum(c * v for c, v in ph)
            assert n - ph[-1][0] <= tot <= n, (m, n, ph)


def test_gumbel_v_mix_known_answers(oracle):  # test_gumbel.py:170-195
    pr = [0.25] * 4
    assert oracle.v_mix(0.7, [0, 0, 0, 0], [0, 0, 0, 0], pr) == pytest.approx(0.7)
    assert oracle.v_mix(0.5, [0, 0, 0.8, 0], [0, 0, 1, 0], pr) == pytest.approx(0.65)
    assert oracle.v_mix(0.1, [0.9] * 4, [100] * 4, pr) == pytest.approx(0.898, abs=0.005)


def _gumbel_mcts(oracle, m, full=False, seed=1):  # make_gumbel_mcts, mcts_test.cc:713-729
    return oracle.Mcts(2.0, 2, 7, gumbel_enabled=True, gumbel_m=m, gumbel_full=full, seed=seed)


def test_gumbel_mcts_known_answers(oracle):
    gs = oracle.Game(oracle.GAME_CONNECT4)
    # SequentialHalvingVisitDistribution, mcts_test.cc:801-822
    for seed in (1, 2, 3, 12345):
        m = _gumbel_mcts(oracle, 4, seed=seed)
        m.set_gumbel_num_sims(13)
        m.search_dumb(gs, 13)
        cnt = m.counts()
        assert sorted((int(c) for c in cnt if c > 0), reverse=True) == [5, 5, 1, 1] and cnt.sum() == 12
    # RunsCleanlyAtLowVisits / ImprovedPolicySumsToOne / FinalActionIsValidMove, mcts_test.cc:744-853
    for n in (4, 8, 16, 32):
        m = _gumbel_mcts(oracle, 16, seed=n)
        m.set_gumbel_num_sims(n)
        m.search_dumb(gs, n)
        assert np.isfinite(m.root_q()).all()
        pi = m.gumbel_improved_policy()
        assert abs(pi.sum() - 1.0) < 1e-4 and (pi >= 0).all()
        assert 0 <= m.gumbel_final_action() < 7
    # SkipsDirichletNoise, mcts_test.cc:768-798: first eval with noise requested keeps the uniform prior
    m = _gumbel_mcts(oracle, 16)
    m.set_gumbel_num_sims(64)
    leaf = m.find_leaf(gs)
    m.process_result(np.full(3, 1 / 3, np.float32), np.full(7, 1 / 7, np.float32), noise=True)
    assert np.allclose(m.gumbel_improved_policy(), 1 / 7, atol=1e-5)
    # FallbackToPuctWhenSimsTargetZero, mcts_test.cc:886-900: target 0 -> plain PUCT, same tree as a PUCT search
    a = _gumbel_mcts(oracle, 16, seed=9); a.set_gumbel_num_sims(0); a.search_dumb(gs, 100)
    b = oracle.Mcts(2.0, 2, 7, seed=9); b.search_dumb(gs, 100)
    assert np.array_equal(a.counts(), b.counts())
    sv, _ = a.gumbel_state()
    assert len(sv) == 0
    # gumbel_full (interior pi'-matching) runs and keeps the root schedule
    f = _gumbel_mcts(oracle, 4, full=True, seed=5); f.set_gumbel_num_sims(13); f.search_dumb(gs, 13)
    assert sorted((int(c) for c in f.counts() if c > 0), reverse=True) == [5, 5, 1, 1]


def test_gumbel_effective_m_and_prior_tables(oracle):
    """the remaining tables of the reference's test_gumbel.py, read off the oracle's MCTS on Connect4 (7 legal moves):
    effective m = max(1, min(gumbel_m, legal moves, simulations)) (test_gumbel.py:298-318, mcts.cc:190-227) as the number of
    Gumbel-top-k survivors after the lazy initialisation, and pi' = the prior while nothing has been visited
    (test_improved_policy_no_visits_equals_prior, test_gumbel.py:207-218)."""
    gs = oracle.Game(oracle.GAME_CONNECT4)
    for gumbel_m, sims, want in ((16, 32, 7),      # capped by the legal moves
                                 (16, 240, 7),
                                 (16, 5, 4),       # capped by the simulations LEFT when the root has been evaluated (mcts.cc:190-199: target - depth)
                                 (4, 240, 4),      # no cap: gumbel_m
                                 (1, 10, 1), (0, 10, 1)):   # never below one
        m = oracle.Mcts(2.0, 2, 7, gumbel_enabled=True, gumbel_m=gumbel_m, seed=3)
        m.set_gumbel_num_sims(sims)
        m.find_leaf(gs)
        m.process_result(np.full(3, 1 / 3, np.float32), np.full(7, 1 / 7, np.float32))
        m.find_leaf(gs)                 # the first descent below an evaluated root initialises the Gumbel state
        sv, _ = m.gumbel_state()
        assert len(sv) == want, (gumbel_m, sims, len(sv))
        assert len(set(int(x) for x in sv)) == len(sv) and all(0 <= int(x) < 7 for x in sv)     # top-k WITHOUT replacement (test_gumbel.py:338-345)
    prior = np.array([0.05, 0.3, 0.1, 0.25, 0.1, 0.15, 0.05], np.float32)
    m = oracle.Mcts(2.0, 2, 7, gumbel_enabled=True, gumbel_m=16, seed=4)
    m.set_gumbel_num_sims(64)
    m.find_leaf(gs)
    m.process_result(np.array([0.2, 0.5, 0.3], np.float32), prior)
    assert np.allclose(m.gumbel_improved_policy(), prior, atol=1e-6)


# ---------------------------------------------------------------- Brandubh / OpenTafl (opentafl_gs_test.cc, brandubh_gs_test.cc)
def test_opentafl_rule_known_answers(oracle):
    import tafl_cases as tc
    for case in tc.OPENTAFL_CASES:
        name, pieces, player, turn, move, exp = case
        g = oracle.Game.tafl_from_board(oracle.GAME_OPENTAFL, tc.board(pieces), player, turn=turn, max_turns=400)
        tc.check_case(g, case)
    g = oracle.Game(oracle.GAME_OPENTAFL)       # StartingPosition, opentafl_gs_test.cc:295-312
    c = g.canonical()
    assert g.player() == 0 and c[0].sum() == 1 and c[1].sum() == 12 and c[2].sum() == 24 and c[0, 5, 5] == 1
    assert c.shape == (8, 11, 11) and (c[7] == 0).all()
    for m in tc.OPENTAFL_REPETITION:            # RepetitionCount, opentafl_gs_test.cc:10-78
        g.play(m)
    assert np.array_equal(g.scores(), [1, 0, 0])
    assert np.allclose(g.canonical()[7], 8 / 400)


def test_brandubh_known_answers(oracle):
    import tafl_cases as tc
    g = oracle.Game(oracle.GAME_BRANDUBH)
    c = g.canonical()
    assert c.shape == (7, 7, 7) and c[0, 3, 3] == 1 and c[1].sum() == 4 and c[2].sum() == 8
    for m in tc.BRANDUBH_REPETITION:            # brandubh_gs_test.cc:10-55 (the test moves one piece back and forth
        g.play(m)                               # for both sides: play_move does not check ownership)
    assert np.array_equal(g.scores(), [1, 0, 0])
    # rules shared with OpenTafl, on the 7x7 board: corner king-only, throne pass-through, king escapes to a corner
    b = tc.board([(tc.A, 3, 1), (tc.K, 6, 5)], n=7)
    g = oracle.Game.tafl_from_board(oracle.GAME_BRANDUBH, b, 0, turn=4, max_turns=150)
    v = g.valid()
    assert v[tc.mv(3, 1, False, 3, n=7)] == 0 and v[tc.mv(3, 1, False, 4, n=7)] == 1
    g = oracle.Game.tafl_from_board(oracle.GAME_BRANDUBH, b, 1, turn=4, max_turns=150)
    g.play(tc.mv(6, 5, False, 6, n=7))
    assert np.array_equal(g.scores(), [0, 1, 0])
    # Brandubh king is captured custodially (no 4-side rule): brandubh_gs.cc:309-340
    b = tc.board([(tc.K, 2, 2), (tc.A, 2, 1), (tc.A, 2, 5)], n=7)
    g = oracle.Game.tafl_from_board(oracle.GAME_BRANDUBH, b, 0, turn=4, max_turns=150)
    g.play(tc.mv(2, 5, False, 3, n=7))
    assert np.array_equal(g.scores(), [1, 0, 0])


# ---- WU-UCT batched search: the reference's own four tests (mcts_test.cc:572-702) on the restatement --------------
def _orc_batched(orc, m, g, total, batch):
    sims = 0
    while sims < total:
        b = min(batch, total - sims)
        for i in range(b):
            leaf = m.find_leaf_batched(g)
            v, pi = orc.dumb_eval(leaf)
            m.process_result_batched(i, v, pi)
        m.reset_batch()
        sims += b


def test_batched_search_reference_cases():
    import oracle_api as orc
    g = orc.Game(orc.GAME_CONNECT4)
    for mv in (1, 6, 3, 6):
        g.play(mv)
    m = orc.Mcts(2.0, 2, 7, seed=3)
    _orc_batched(orc, m, g, 800, 8)
    assert m.pick_move(m.probs(0.0)) == 2                     # BatchedBasic
    g2 = orc.Game(orc.GAME_CONNECT4)
    for mv in (3, 0, 3, 0, 3, 1):
        g2.play(mv)
    m = orc.Mcts(2.0, 2, 7, seed=3)
    _orc_batched(orc, m, g2, 100, 4)
    assert m.pick_move(m.probs(0.0)) == 3                     # BatchedTerminal
    a = orc.Mcts(2.0, 2, 7, seed=12345); b = orc.Mcts(2.0, 2, 7, seed=12345)
    a.search_dumb(g, 800)
    _orc_batched(orc, b, g, 800, 1)
    assert a.counts().tolist() == b.counts().tolist() == [62, 21, 631, 21, 22, 21, 21]   # BatchedSingleEquivalent (+ the SURVEY §8c answer)
    g0 = orc.Game(orc.GAME_CONNECT4)
    m = orc.Mcts(2.0, 2, 7, seed=5)
    m.search_dumb(g0, 1)
    keys = []
    for _ in range(4):
        keys.append(m.find_leaf_batched(g0).canonical().tobytes())
    v, pi = orc.dumb_eval(g0)
    for i in range(4):
        m.process_result_batched(i, v, pi)
    with pytest.raises(IndexError):
        m.process_result_batched(4, v, pi)
    m.reset_batch()
    assert m.in_flight_count() == 0 and len(set(keys)) >= 3   # WUUCTDiversity


@pytest.mark.parametrize("fname,spec_fn", [("nn_connect4_6b64c.npz", "connect4_spec"), ("nn_tawlbwrdd_4b64c.npz", "tawlbwrdd_spec"),
                                            ("nn_opentafl_4b64c.npz", "opentafl_spec"), ("nn_brandubh_4b32c.npz", "brandubh_spec")])
def test_torch_leafnet_reproduces_the_reference_nnarch_fixtures(fname, spec_fn):
    """the package's PyTorch restatement of NNArch (alphazero/torch_net.py: the fp32 reference of the HIP nets) loads the
    reference's state_dict and reproduces the reference's outputs bit for bit on the CPU (fixtures: make_nn_fixture.py)."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "alphazero-pybind11_amd"))
    from alphazero import torch_net
    fx = np.load(os.path.join(HERE, "golden", fname))
    net = torch_net.LeafNet(getattr(torch_net, spec_fn)())
    net.load_state_dict({k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("sd.")})
    v, pi = net.eval().process(torch.from_numpy(fx["input"]))
    assert np.abs(v.numpy() - fx["v"]).max() <= 1e-6 and np.abs(pi.numpy() - fx["pi"]).max() <= 1e-6


def test_reference_s3fifo_cases_on_the_oracle(oracle):
    """s3fifo_cache_test.cc's 24 single-thread cases (tests/s3fifo_cases.py) on the oracle's restatement."""
    import s3fifo_cases
    ran = s3fifo_cases.run_all(lambda mx, gh, np_, nv, shards=1: oracle.Cache(mx, shards, gh, np_, nv))
    assert len(ran) == 21
