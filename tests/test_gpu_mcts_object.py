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
    with pytest.raises(RuntimeError, match="not implemented"):
        m.find_leaf_batched(gs, 4)


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
