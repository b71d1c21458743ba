"""-m gpu: the stand-alone `MCTS` class on the device (py_wrapper.cc:192-220) against the reference's known answer
(counts [62 21 631 21 22 21 21]) and the oracle's Mcts, call by call."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def az():
    import alphazero
    return alphazero


def _search(az, m, gs, sims, noise=False):
    for _ in range(sims):
        leaf = m.find_leaf(gs)
        v, pi = az.dumb_eval(leaf)
        m.process_result(gs, v, pi, noise)


def test_reference_known_answer(az):
    """MCTS::seed_thread_rng(12345); Connect4 after 1,6,3,6; MCTS{2,2,7}; 800 x (find_leaf, dumb_eval, process_result)."""
    gs = az.Connect4GS()
    for mv in (1, 6, 3, 6):
        gs.play_move(mv)
    m = az.MCTS(2.0, 2, 7, seed=12345)
    _search(az, m, gs, 800)
    assert m.counts().tolist() == [62, 21, 631, 21, 22, 21, 21]
    assert m.pick_move(m.probs(0.0)) == 2
    assert m.depth() == 800 and m.root_n() == 800


@pytest.mark.parametrize("cfg", [
    dict(cpuct=1.25, fpu_reduction=0.25),
    dict(cpuct=1.25, fpu_reduction=0.25, epsilon=0.25, root_policy_temp=1.25, root_fpu_zero=True, shaped_dirichlet=True),
    dict(cpuct=2.0, gumbel_enabled=True, gumbel_m=4),
])
def test_call_by_call_parity_with_oracle(az, oracle, cfg):
    seed = 99
    noise = cfg.get("epsilon", 0) > 0
    kw = dict(cfg); cpuct = kw.pop("cpuct")
    m = az.MCTS(cpuct, 2, 7, seed=seed, **kw)
    o = oracle.Mcts(cpuct, 2, 7, seed=seed, **kw)
    gs = az.Connect4GS(); og = oracle.Game(oracle.GAME_CONNECT4)
    rng = np.random.default_rng(4)
    for ply in range(6):
        if cfg.get("gumbel_enabled"):
            m.set_gumbel_num_sims(40); o.set_gumbel_num_sims(40)
        for _ in range(40):
            leaf = m.find_leaf(gs); oleaf = o.find_leaf(og)
            assert np.array_equal(leaf.canonicalized(), oleaf.canonical())
            v, pi = az.dumb_eval(leaf)
            pi = (pi * (1 + 0.3 * np.sin(np.arange(7) + ply))).astype(np.float32); pi /= pi.sum()
            out = m.process_result(gs, v.copy(), pi, noise)
            oout = o.process_result(v.copy(), pi, noise)
            assert np.array_equal(out, oout)
        assert np.array_equal(m.counts(), o.counts())
        assert np.array_equal(m.root_q_values(), o.root_q())
        assert np.array_equal(m.probs(1.0), o.probs(1.0)) and np.array_equal(m.probs(0.0), o.probs(0.0))
        assert np.array_equal(m.probs_pruned(1.0), o.probs(1.0, pruned=True))
        assert np.array_equal(m.root_value(), o.root_value())
        assert m.depth() == o.depth() and m.root_n() == o.root_n()
        assert m.avg_leaf_depth() == pytest.approx(o.avg_leaf_depth(), rel=1e-6)
        assert m.normalized_root_entropy() == pytest.approx(o.entropy(), rel=1e-6)
        assert np.array_equal(m.principal_variation(5), o.principal_variation(5))
        if cfg.get("gumbel_enabled"):
            assert np.array_equal(m.gumbel_improved_policy(), o.gumbel_improved_policy())
            move = m.gumbel_final_action()
            assert move == o.gumbel_final_action()
        else:
            move = m.pick_move(m.probs(1.0))
            assert move == o.pick_move(o.probs(1.0))
        m.update_root(gs, move); o.update_root(og, move)
        gs.play_move(move); og.play(move)
        if gs.scores() is not None:
            break
        if noise:
            m.apply_root_policy_temp(); o.apply_root_policy_temp()
            if m.root_n() > 0:
                m.add_root_noise(); o.add_root_noise()


def test_errors(az):
    m = az.MCTS(2.0, 2, 7, seed=1)
    gs = az.Connect4GS()
    _search(az, m, gs, 5)
    with pytest.raises(RuntimeError, match="ahh, what is this move"):
        m.update_root(gs, 9)
    m.update_root(gs, 3)           # the object is still usable afterwards
    with pytest.raises(RuntimeError):
        az.MCTS(2.0, 2, 8, seed=1)
    with pytest.raises(IndexError):          # in_flight_.at(leaf_index), mcts.cc:789
        v, pi = az.dumb_eval(gs)
        m.process_result_batched(gs, 0, v, pi)


@pytest.mark.parametrize("name,cfg", [
    ("TawlbwrddGS", dict(cpuct=1.25, fpu_reduction=0.25)),
    ("TawlbwrddGS", dict(cpuct=1.25, epsilon=0.25, root_policy_temp=1.25, shaped_dirichlet=True)),
    ("BrandubhGS", dict(cpuct=2.0, gumbel_enabled=True, gumbel_m=8)),
    ("OpenTaflGS", dict(cpuct=1.25, fpu_reduction=0.25, root_fpu_zero=True)),
])
def test_tafl_family_call_by_call_parity(az, oracle, name, cfg):
    """The MCTS object on the wide-game engine (one wavefront, repetition-aware replay of the root state)."""
    Game = getattr(az, name)
    gid = {"TawlbwrddGS": oracle.GAME_TAWLBWRDD, "BrandubhGS": oracle.GAME_BRANDUBH, "OpenTaflGS": oracle.GAME_OPENTAFL}[name]
    M = Game.NUM_MOVES()
    seed = 21
    noise = cfg.get("epsilon", 0) > 0
    kw = dict(cfg); cpuct = kw.pop("cpuct")
    m = az.MCTS(cpuct, 2, M, seed=seed, game=Game, max_simulations=400, **kw)
    o = oracle.Mcts(cpuct, 2, M, seed=seed, **kw)
    gs = Game(); og = oracle.Game(gid)
    for ply in range(4):
        if cfg.get("gumbel_enabled"):
            m.set_gumbel_num_sims(20); o.set_gumbel_num_sims(20)
        for _ in range(20):
            leaf = m.find_leaf(gs); oleaf = o.find_leaf(og)
            assert np.array_equal(leaf.canonicalized(), oleaf.canonical())
            v, pi = az.dumb_eval(leaf)
            assert np.array_equal(m.process_result(gs, v.copy(), pi, noise), o.process_result(v.copy(), pi, noise))
        assert np.array_equal(m.counts(), o.counts())
        assert np.array_equal(m.root_q_values(), o.root_q())
        assert np.array_equal(m.probs(1.0), o.probs(1.0)) and np.array_equal(m.probs(0.0), o.probs(0.0))
        assert np.array_equal(m.probs_pruned(1.0), o.probs(1.0, pruned=True))
        assert np.array_equal(m.root_value(), o.root_value())
        assert m.depth() == o.depth() and m.root_n() == o.root_n()
        assert np.array_equal(m.principal_variation(4), o.principal_variation(4))
        if cfg.get("gumbel_enabled"):
            assert np.array_equal(m.gumbel_improved_policy(), o.gumbel_improved_policy())
            move = m.gumbel_final_action(); assert move == o.gumbel_final_action()
        else:
            move = m.pick_move(m.probs(1.0)); assert move == o.pick_move(o.probs(1.0))
        m.update_root(gs, move); o.update_root(og, move)
        gs.play_move(move); og.play(move)
        if noise:
            m.apply_root_policy_temp(); o.apply_root_policy_temp()
            if m.root_n() > 0:
                m.add_root_noise(); o.add_root_noise()


# ---- WU-UCT batched API (mcts.cc:752-851; reference tests mcts_test.cc:572-697) ----------------------------------
def _batched_search(az, m, gs, total, batch):
    sims = 0
    while sims < total:
        b = min(batch, total - sims)
        for i in range(b):
            leaf = m.find_leaf_batched(gs)
            assert m.in_flight_count() == i + 1
            v, pi = az.dumb_eval(leaf)
            m.process_result_batched(gs, i, v, pi)
        m.reset_batch()
        assert m.in_flight_count() == 0
        sims += b


def test_batched_basic_and_terminal(az):
    """mcts_test.cc BatchedBasic (:572-596) and BatchedTerminal (:599-626)."""
    gs = az.Connect4GS()
    for mv in (1, 6, 3, 6):
        gs.play_move(mv)
    m = az.MCTS(2.0, 2, 7, seed=3)
    _batched_search(az, m, gs, 800, 8)
    assert m.pick_move(m.probs(0.0)) == 2 and int(m.counts().sum()) == 799
    gs = az.Connect4GS()
    for mv in (3, 0, 3, 0, 3, 1):
        gs.play_move(mv)
    m = az.MCTS(2.0, 2, 7, seed=3)
    _batched_search(az, m, gs, 100, 4)
    assert m.pick_move(m.probs(0.0)) == 3


def test_batched_single_equals_unbatched(az):
    """mcts_test.cc BatchedSingleEquivalent (:629-667); with one seed the two are the same search, bit for bit."""
    gs = az.Connect4GS()
    for mv in (1, 6, 3, 6):
        gs.play_move(mv)
    a = az.MCTS(2.0, 2, 7, seed=12345); b = az.MCTS(2.0, 2, 7, seed=12345)
    _search(az, a, gs, 800)
    _batched_search(az, b, gs, 800, 1)
    assert a.counts().tolist() == b.counts().tolist() == [62, 21, 631, 21, 22, 21, 21]
    assert np.array_equal(a.root_q_values(), b.root_q_values())


def test_wu_uct_diversity(az):
    """mcts_test.cc WUUCTDiversity (:670-702): four leaves found in one batch explore different children."""
    gs = az.Connect4GS()
    m = az.MCTS(2.0, 2, 7, seed=5)
    _search(az, m, gs, 1)
    leaves = [m.find_leaf_batched(gs) for _ in range(4)]
    for i in range(4):
        v, pi = az.dumb_eval(gs)
        m.process_result_batched(gs, i, v, pi)
    m.reset_batch()
    assert len({str(l) for l in leaves}) >= 3


@pytest.mark.parametrize("name,batch,cfg", [
    ("Connect4GS", 8, dict(cpuct=1.25, fpu_reduction=0.25)),
    ("Connect4GS", 5, dict(cpuct=1.25, fpu_reduction=0.25, epsilon=0.25, root_policy_temp=1.25, root_fpu_zero=True, shaped_dirichlet=True)),
    ("TawlbwrddGS", 4, dict(cpuct=1.25, fpu_reduction=0.25)),
    ("BrandubhGS", 6, dict(cpuct=2.0, root_fpu_zero=True)),
])
def test_batched_parity_with_oracle(az, oracle, name, batch, cfg):
    """find all leaves of a batch first, then back them up in a scrambled order: every leaf, every returned value vector and
    every root read-out equals the oracle's, over several moves with update_root in between."""
    Game = getattr(az, name)
    gid = {"Connect4GS": oracle.GAME_CONNECT4, "TawlbwrddGS": oracle.GAME_TAWLBWRDD, "BrandubhGS": oracle.GAME_BRANDUBH}[name]
    M = Game.NUM_MOVES()
    noise = cfg.get("epsilon", 0) > 0
    kw = dict(cfg); cpuct = kw.pop("cpuct")
    extra = {} if name == "Connect4GS" else dict(game=Game, max_simulations=400)
    m = az.MCTS(cpuct, 2, M, seed=17, **extra, **kw)
    o = oracle.Mcts(cpuct, 2, M, seed=17, **kw)
    gs = Game(); og = oracle.Game(gid)
    rng = np.random.default_rng(8)
    rounds = 10 if name == "Connect4GS" else 5
    for ply in range(4):
        for _ in range(rounds):
            evals = []
            for i in range(batch):
                leaf = m.find_leaf_batched(gs); oleaf = o.find_leaf_batched(og)
                assert np.array_equal(leaf.canonicalized(), oleaf.canonical())
                v, pi = az.dumb_eval(leaf)
                w = (1 + 0.3 * np.sin(np.arange(M) * 0.37 + i)).astype(np.float32)
                pi = (pi * w).astype(np.float32); pi /= max(pi.sum(), 1e-30)
                evals.append((v, pi))
            assert m.in_flight_count() == o.in_flight_count() == batch
            for i in rng.permutation(batch):
                v, pi = evals[i]
                assert np.array_equal(m.process_result_batched(gs, int(i), v.copy(), pi, noise),
                                      o.process_result_batched(int(i), v.copy(), pi, noise))
            m.reset_batch(); o.reset_batch()
        assert np.array_equal(m.counts(), o.counts())
        assert np.array_equal(m.root_q_values(), o.root_q())
        assert np.array_equal(m.probs(1.0), o.probs(1.0))
        assert np.array_equal(m.root_value(), o.root_value())
        assert m.depth() == o.depth() and m.root_n() == o.root_n()
        assert m.avg_leaf_depth() == pytest.approx(o.avg_leaf_depth(), rel=1e-6)
        move = m.pick_move(m.probs(1.0)); assert move == o.pick_move(o.probs(1.0))
        m.update_root(gs, move); o.update_root(og, move)
        gs.play_move(move); og.play(move)
        if gs.scores() is not None:
            break


# ---- the reference's principal_variation tests (test_principal_variation.py:19-152), same calls, same assertions -----------
def _ref_make_mcts(az, Game, gumbel_enabled=False):
    # the 14-argument constructor overload, positionally (py_wrapper.cc:196-197)
    return az.MCTS(1.25, Game.NUM_PLAYERS(), Game.NUM_MOVES(), 0.0, 1.0, 0.25, Game().relative_values(), False, False,
                   bool(gumbel_enabled), 16, 50.0, 1.0, False)


def _ref_uniform_sims(mcts, gs, n):
    P = gs.num_players()
    uniform_v = np.full(P + 1, 1.0 / (P + 1))                 # float64, like the reference test
    for i in range(n):
        leaf = mcts.find_leaf(gs)
        if leaf.scores() is not None:
            v = np.array(leaf.scores()); pi = np.zeros(gs.num_moves())
        else:
            v = uniform_v; pi = np.ones(gs.num_moves()) / gs.num_moves()
        mcts.process_result(gs, v, pi, i == 0)


def test_reference_principal_variation_cases(az):
    Game = az.Connect4GS
    assert list(_ref_make_mcts(az, Game).principal_variation(0)) == []            # depth zero
    assert list(_ref_make_mcts(az, Game).principal_variation(5)) == []            # no simulations
    gs = Game(); m = _ref_make_mcts(az, Game); _ref_uniform_sims(m, gs, 100)       # root = visit arg-max under PUCT
    pv = list(m.principal_variation(5))
    assert len(pv) >= 1 and pv[0] == int(np.argmax(np.array(m.counts())))
    gs = Game(); m = _ref_make_mcts(az, Game); _ref_uniform_sims(m, gs, 200)       # depth cap
    short, long_ = list(m.principal_variation(3)), list(m.principal_variation(10))
    assert len(short) <= 3 and len(long_) >= len(short)
    gs = Game(); m = _ref_make_mcts(az, Game); _ref_uniform_sims(m, gs, 3)         # stops at unvisited nodes
    assert len(list(m.principal_variation(10))) <= 3
    gs = Game(); m = _ref_make_mcts(az, Game, gumbel_enabled=True)                 # Gumbel: the root follows gumbel_final_action
    m.set_gumbel_num_sims(64); _ref_uniform_sims(m, gs, 64)
    pv = list(m.principal_variation(3))
    assert len(pv) >= 1 and pv[0] == int(m.gumbel_final_action())
    gs = Game(); m = _ref_make_mcts(az, Game); _ref_uniform_sims(m, gs, 30)        # valid move ids
    assert all(0 <= int(x) < gs.num_moves() for x in m.principal_variation(4))
    gs = Game(); m = _ref_make_mcts(az, Game); _ref_uniform_sims(m, gs, 50)        # first move legal at the start position
    pv = list(m.principal_variation(1))
    assert len(pv) == 1 and np.array(gs.valid_moves())[pv[0]] == 1


# ---- the reference's mcts_test.cc property and Gumbel cases (:127-278, :744-940) on the device MCTS object ---------------------
def _dumb_search(az, m, gs, upto, noise=False):
    while m.depth() < upto:
        leaf = m.find_leaf(gs)
        v, pi = az.dumb_eval(leaf)
        m.process_result(gs, v, pi, noise)


def test_reference_mcts_property_cases(az):
    gs = az.Connect4GS()
    m = az.MCTS(2, 2, 7, 0, 1.4, 0.25, seed=1)                 # RootValueSetOnFirstEval, :127-157
    _dumb_search(az, m, gs, 31)
    assert int((m.counts() > 0).sum()) >= 3
    normal = az.MCTS(2, 2, 7, 0.0, 1.0, 0.25, False, False, seed=2)   # RootFpuZero, :198-230
    fpu0 = az.MCTS(2, 2, 7, 0.0, 1.0, 0.25, False, True, seed=2)
    _dumb_search(az, normal, gs, 50); _dumb_search(az, fpu0, gs, 50)
    assert int((fpu0.counts() > 0).sum()) >= int((normal.counts() > 0).sum())
    m = az.MCTS(2, 2, 7, 0.25, 1.4, 0.0, False, False, False, seed=3)  # PolicyTargetPruning, :233-276
    _dumb_search(az, m, gs, 1, noise=True)
    _dumb_search(az, m, gs, 200)
    regular, pruned = m.probs(1.0), m.probs_pruned(1.0)
    assert abs(regular.sum() - 1) < 1e-5 and abs(pruned.sum() - 1) < 1e-5
    assert pruned.max() + 0.01 >= regular.max()
    assert (pruned[regular == 0] == 0).all()                   # pruning creates no mass at unvisited moves (:563-568)
    nn = az.MCTS(2, 2, 7, 0.0, 1.0, 0.0, False, False, False, seed=4)
    _dumb_search(az, nn, gs, 100)
    assert abs(nn.probs_pruned(1.0).sum() - 1) < 1e-5


def test_reference_shaped_dirichlet_distribution_cases(az):
    """mcts_test.cc:279-355 (ShapedDirichletDistribution, ShapedDirichletAlphaDistribution) on the device object - the one MCTS test of
    the reference that was only covered draw by draw.  With no visits probs(1.0) returns the root's priors
    (mcts.cc:575-591), i.e. the priors after add_root_noise (mcts.cc:403-441).  The reference's assertions - a valid distribution,
    every legal move noised in every trial - and, beyond them, the noise's first two moments over 400 seeds: dumb_eval's uniform
    policy makes every shaped alpha equal (shaped_sum = 0), so the noise is a symmetric Dirichlet(A / 7 each, A = NOISE_ALPHA_RATIO =
    10.83) and a prior is 0.75 / 7 + 0.25 x Dir_i: mean 1 / 7, variance 0.25^2 x a (A - a) / (A^2 (A + 1)), a = A / 7."""
    for shaped in (True, False):                                              # ShapedDirichletDistribution, :279-314
        gs = az.Connect4GS()
        m = az.MCTS(2, 2, 7, 0.25, 1.4, 0.0, False, False, shaped, seed=11)
        leaf = m.find_leaf(gs); v, pi = az.dumb_eval(leaf); m.process_result(gs, v, pi, True)
        p = m.probs(1.0)
        assert (p >= 0).all() and abs(float(p.sum()) - 1.0) < 1e-4
    gs = az.Connect4GS()                                                        # ShapedDirichletAlphaDistribution, :317-355
    for mv in (3, 0, 3, 0):
        gs.play_move(mv)
    valid = gs.valid_moves()
    trials = 400                                                                # (the reference: 50)
    rows = []
    for t in range(trials):
        m = az.MCTS(2, 2, 7, 0.25, 1.4, 0.0, False, False, True, seed=1000 + t)
        leaf = m.find_leaf(gs); v, pi = az.dumb_eval(leaf); m.process_result(gs, v, pi, True)
        p = m.probs(1.0)
        assert abs(float(p.sum()) - 1.0) < 1e-4 and (p >= 0).all()
        assert ((p > 0) == (valid == 1)).all(), f"trial {t}: every legal move gets noise, no other move any mass"
        rows.append(p)
    x = np.stack(rows).astype(np.float64)
    assert len({tuple(r) for r in x.round(7)}) == trials                        # 400 seeds, 400 different noise draws
    A, eps = 10.83, 0.25
    a = A / 7
    var = eps * eps * a * (A - a) / (A * A * (A + 1))
    se = (var / trials) ** 0.5
    assert np.abs(x.mean(0) - 1.0 / 7).max() < 5 * se, (x.mean(0), se)
    assert 0.75 * var < x.var(0).min() and x.var(0).max() < 1.3 * var, (x.var(0), var)


def _gumbel_mcts(az, m, full=False, seed=7):                   # make_gumbel_mcts, :713-728
    return az.MCTS(2.0, 2, 7, 0.0, 1.0, 0.0, False, False, False, True, m, 50.0, 1.0, full, seed=seed)


def _run_gumbel(az, mcts, gs, n):                              # run_gumbel_search, :730-738
    mcts.set_gumbel_num_sims(n)
    for _ in range(n):
        leaf = mcts.find_leaf(gs)
        v, pi = az.dumb_eval(leaf)
        mcts.process_result(gs, v, pi)


def test_reference_gumbel_mcts_cases(az):
    for n in (4, 8, 16):                                       # RunsCleanlyAtLowVisits, :744-765
        gs = az.Connect4GS(); m = _gumbel_mcts(az, 16); _run_gumbel(az, m, gs, n)
        assert np.isfinite(m.root_q_values()).all() and abs(m.gumbel_improved_policy().sum() - 1) < 1e-4
        fa = m.gumbel_final_action(); assert fa < 7 and gs.valid_moves()[fa] == 1
    gs = az.Connect4GS(); m = _gumbel_mcts(az, 16); m.set_gumbel_num_sims(64)   # SkipsDirichletNoise, :768-798
    leaf = m.find_leaf(gs); v, pi = az.dumb_eval(leaf); m.process_result(gs, v, pi, True)
    assert np.allclose(m.gumbel_improved_policy(), 1.0 / 7, atol=1e-5)
    gs = az.Connect4GS(); m = _gumbel_mcts(az, 4); _run_gumbel(az, m, gs, 13)    # SequentialHalvingVisitDistribution, :801-822
    c = m.counts(); nz = sorted((int(x) for x in c if x > 0), reverse=True)
    assert nz == [5, 5, 1, 1] and int(c.sum()) == 12
    gs = az.Connect4GS(); m = _gumbel_mcts(az, 16); _run_gumbel(az, m, gs, 32)   # ImprovedPolicySumsToOne, :825-839
    p = m.gumbel_improved_policy(); assert abs(p.sum() - 1) < 1e-4 and (p >= 0).all()
    gs = az.Connect4GS(); m = _gumbel_mcts(az, 4); _run_gumbel(az, m, gs, 16)    # FinalActionIsValidMove, :842-852
    fa1 = m.gumbel_final_action(); assert fa1 < 7 and gs.valid_moves()[fa1] == 1
    m.set_gumbel_num_sims(0); m.set_gumbel_num_sims(16)                           # TreeReuseResamplesGumbel, :855-884
    for _ in range(16):
        leaf = m.find_leaf(gs); v, pi = az.dumb_eval(leaf); m.process_result(gs, v, pi)
    assert gs.valid_moves()[m.gumbel_final_action()] == 1
    gs = az.Connect4GS(); m = _gumbel_mcts(az, 16); m.set_gumbel_num_sims(0)      # FallbackToPuctWhenSimsTargetZero, :887-904
    for _ in range(100):
        leaf = m.find_leaf(gs); v, pi = az.dumb_eval(leaf); m.process_result(gs, v, pi)
    fa = m.gumbel_final_action(); assert fa < 7 and gs.valid_moves()[fa] == 1
    gs = az.Connect4GS()                                                            # FindsWinningMove, :907-924
    for mv in (3, 0, 3, 0, 3, 1): gs.play_move(mv)
    m = _gumbel_mcts(az, 16); _run_gumbel(az, m, gs, 128)
    assert m.gumbel_final_action() == 3
    gs = az.Connect4GS(); m = _gumbel_mcts(az, 16, full=True); _run_gumbel(az, m, gs, 64)   # FullGumbelInteriorRunsCleanly, :927-937
    assert abs(m.gumbel_improved_policy().sum() - 1) < 1e-4 and gs.valid_moves()[m.gumbel_final_action()] == 1


def test_direct_mcts_consumer_shape(az):
    """The call pattern of the reference's direct-MCTS consumers (play.py:274-343, cache_utils.cached_inference): leaves are
    collected with find_leaf_batched, terminal / playout / cached leaves are backed up at once, the rest after one batched
    evaluation that also fills an S3FIFOCache keyed by hash_game_state."""
    gs = az.Connect4GS()
    for mv in (3, 3, 2, 4):
        gs.play_move(mv)
    mcts = az.MCTS(1.25, 2, 7, 0.25, 1.25, 0.25, False, True, True, seed=9)
    cache = az.S3FIFOCache(4096, 3686, 7, 3)
    P1, M = gs.num_players() + 1, gs.num_moves()
    sims, evals = 0, 0
    while sims < 160:
        pending = []
        for attempt in range(16):
            if len(pending) >= 8:
                break
            leaf = mcts.find_leaf_batched(gs)
            idx = mcts.in_flight_count() - 1
            noise = sims == 0 and attempt == 0
            if leaf.scores() is not None:
                mcts.process_result_batched(gs, idx, np.array(leaf.scores()), np.zeros(M), noise); sims += 1
                continue
            h = az.hash_game_state(leaf)
            hit = cache.find(h, M, P1)
            if hit is not None:
                mcts.process_result_batched(gs, idx, np.array(hit[1]), np.array(hit[0]), noise); sims += 1
                continue
            pending.append((idx, h, noise, leaf))
        for idx, h, noise, leaf in pending:                # the "GPU batch": a playout evaluation per leaf
            v, pi = az.playout_eval(leaf, seed=h & 0xFFFF)
            cache.insert(h, pi, v)
            mcts.process_result_batched(gs, idx, v, pi, noise)
            evals += 1
        sims += len(pending)
        mcts.reset_batch()
    assert mcts.depth() == sims and int(mcts.counts().sum()) == sims - 1
    assert cache.size() > 0 and cache.hits() + cache.misses() >= evals
    pv = list(mcts.principal_variation(3))
    assert len(pv) >= 1 and gs.valid_moves()[pv[0]] == 1
