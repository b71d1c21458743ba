"""-m gpu: Brandubh and OpenTafl on the device (rules kernels + the wide-game engine) against the reference's own
rule tests (opentafl_gs_test.cc / brandubh_gs_test.cc, as data in tafl_cases.py) and the oracle."""
import numpy as np
import pytest

import tafl_cases as tc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def az():
    import alphazero
    return alphazero


class _DevGame:
    """adapter: alphazero GameState object -> the tiny interface tafl_cases.check_case uses"""
    def __init__(self, gs): self.gs = gs
    def play(self, m): self.gs.play_move(m)
    def valid(self): return self.gs.valid_moves()
    def scores(self): return self.gs.scores()
    def canonical(self): return self.gs.canonicalized()


def test_opentafl_reference_rule_cases_on_device(az):
    for case in tc.OPENTAFL_CASES:
        name, pieces, player, turn, move, exp = case
        gs = az.OpenTaflGS.from_board(tc.board(pieces), player, turn)
        tc.check_case(_DevGame(gs), case)
    rep = az.OpenTaflGS()                      # RepetitionCount, opentafl_gs_test.cc:10-78
    for m in tc.OPENTAFL_REPETITION:
        rep.play_move(m)
    assert np.array_equal(rep.scores(), [1, 0, 0]) and np.allclose(rep.canonicalized()[7], 8 / 400)
    gs = az.OpenTaflGS()
    c = gs.canonicalized()
    assert gs.current_player() == 0 and c[0].sum() == 1 and c[1].sum() == 12 and c[2].sum() == 24 and c[0, 5, 5] == 1
    assert gs.num_moves() == 2662 and az.OpenTaflGS.CANONICAL_SHAPE() == (8, 11, 11) and az.OpenTaflGS.POLICY_SHAPE() == (22, 11, 11)


def test_brandubh_cases_on_device(az):
    rep = az.BrandubhGS()                      # RepetitionCount, brandubh_gs_test.cc:10-55 (moves a piece of either side)
    for m in tc.BRANDUBH_REPETITION:
        rep.play_move(m)
    assert np.array_equal(rep.scores(), [1, 0, 0])
    gs = az.BrandubhGS()
    c = gs.canonicalized()
    assert c.shape == (7, 7, 7) and c[0, 3, 3] == 1 and c[1].sum() == 4 and c[2].sum() == 8 and gs.num_moves() == 686
    b = tc.board([(tc.A, 3, 1), (tc.K, 6, 5)], n=7)
    v = az.BrandubhGS.from_board(b, 0, 4).valid_moves()
    assert v[tc.mv(3, 1, False, 3, n=7)] == 0 and v[tc.mv(3, 1, False, 4, n=7)] == 1
    g = az.BrandubhGS.from_board(b, 1, 4); g.play_move(tc.mv(6, 5, False, 6, n=7))
    assert np.array_equal(g.scores(), [0, 1, 0])
    b = tc.board([(tc.K, 2, 2), (tc.A, 2, 1), (tc.A, 2, 5)], n=7)
    g = az.BrandubhGS.from_board(b, 0, 4); g.play_move(tc.mv(2, 5, False, 3, n=7))
    assert np.array_equal(g.scores(), [1, 0, 0])


@pytest.mark.parametrize("name", ["BrandubhGS", "OpenTaflGS"])
def test_rules_random_walks_match_oracle(az, oracle, name):
    """T0: valid_moves / scores / canonicalized / player / turn along seeded random legal walks, batched replay."""
    Game = getattr(az, name)
    gid = oracle.GAME_BRANDUBH if name == "BrandubhGS" else oracle.GAME_OPENTAFL
    rng = np.random.default_rng(17)
    n, L = 300, 120
    moves = -np.ones((n, L), np.int32)
    finals = []
    for g in range(n):
        game = oracle.Game(gid)
        for i in range(int(rng.integers(0, L + 1))):
            if game.scores() is not None:
                break
            m = int(rng.choice(np.flatnonzero(game.valid())))
            game.play(m); moves[g, i] = m
        finals.append(game)
    out = az.game_replay(Game, moves)
    assert (out["status"] == 0).all()
    ended = 0
    for g, game in enumerate(finals):
        assert np.array_equal(out["valid"][g], game.valid()), g
        sc = game.scores()
        if sc is None:
            assert (out["scores"][g] == -1).all(), g
        else:
            ended += 1
            assert np.array_equal(out["scores"][g], sc), g
        assert np.array_equal(out["canonical"][g], game.canonical()), g
        assert out["player"][g] == game.player() and out["turn"][g] == game.turn(), g
    assert ended > 0


@pytest.mark.parametrize("name,cfg", [
    ("BrandubhGS", dict()),
    ("BrandubhGS", dict(epsilon=0.25, shaped_dirichlet=True, mcts_root_temp=1.25, policy_target_pruning=True)),
    ("BrandubhGS", dict(gumbel_enabled=True, gumbel_m=8)),
    ("OpenTaflGS", dict()),
    ("OpenTaflGS", dict(playout_cap_randomization=True, playout_cap_depth=8, playout_cap_percent=0.5, epsilon=0.25)),
])
def test_playmanager_parity(az, oracle, name, cfg):
    """T2 on the wide-game engine instantiated for Brandubh / OpenTafl: moves, visit counts, RNG position, history rows."""
    Game = getattr(az, name)
    gid = oracle.GAME_BRANDUBH if name == "BrandubhGS" else oracle.GAME_OPENTAFL
    pp = az.PlayParams()
    pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
    pp.history_enabled = True
    pp.games_to_play, pp.concurrent_games, pp.mcts_visits = 6, 6, [24, 24]
    pp.cpuct, pp.fpu_reduction = 1.25, 0.25
    for k, v in cfg.items():
        setattr(pp, k, v)
    seed = 404
    pm = az.PlayManager(Game(), pp, seed=seed, log_moves=True)
    pm.play()
    rows, counts = pm.move_log()
    hc, hv, hp = pm.history()
    tot = np.zeros(3, np.float32)
    rows_orc = []
    for s in range(6):
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games = 1, 1
        o = oracle.PlayManager(gid, one, oracle.slot_seed(seed, s), per_slot_rng=False)
        o.run()
        orows, ocounts = o.moves()
        sel = rows[:, 0] == s
        assert np.array_equal(rows[sel][:, 1:], orows[:, 1:]), s
        assert np.array_equal(counts[sel], ocounts), s
        tot += o.scores()
        c, v, p = o.history()
        rows_orc += [(a.tobytes(), b.tobytes(), d.tobytes()) for a, b, d in zip(c, v, p)]
    assert np.array_equal(pm.scores(), tot)
    assert sorted((a.tobytes(), b.tobytes(), d.tobytes()) for a, b, d in zip(hc, hv, hp)) == sorted(rows_orc)


def test_symmetries_match_oracle(az, oracle):
    rng = np.random.default_rng(2)
    for Game, n, C in ((az.BrandubhGS, 7, 7), (az.OpenTaflGS, 11, 8)):
        c = rng.random((3, C, n, n), dtype=np.float32); v = rng.random((3, 3), dtype=np.float32)
        pi = rng.random((3, n * n * 2 * n), dtype=np.float32)
        oc, ov, op = az.symmetries_batch(Game, c, v, pi)
        for i in range(3):
            ec, ev, ep = oracle.symmetries(oracle.SYM_TAFL_EIGHT, c[i], v[i], pi[i])
            assert np.array_equal(oc[i], ec) and np.array_equal(op[i], ep) and np.array_equal(ov[i], ev)


def test_opentafl_selfplay_on_the_hip_net():
    """OpenTafl self-play end to end on the device: engine rounds + the bf16 MFMA net (8 input planes) with Gumbel search as in
    configs/open_tafl.yaml; every game finishes, every history row is a distribution over legal-looking moves."""
    import torch
    import alphazero as az
    from alphazero import torch_net
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.max_batch_size = 48, 48, 48
    pp.mcts_visits = [24, 24]
    pp.model_groups = [0, 0]
    pp.gumbel_enabled, pp.gumbel_m = True, 8
    pp.history_enabled = True
    pp.cpuct = 1.25
    net = torch_net.random_init(torch_net.opentafl_spec(), seed=3)
    hip = az.HipLeafNet(net)
    pm = az.PlayManager(az.OpenTaflGS(), pp, seed=5, log_moves=True)
    st = torch.cuda.Stream()
    for _ in range(400):
        az.run_rounds([pm], hip, 64, [st.cuda_stream])
        if pm.poll(st.cuda_stream)[1] == 0:
            break
    assert pm.games_completed() == 48 and pm.scores().sum() == 48
    n = pm.hist_count()
    canon = np.zeros((n, 8, 11, 11), np.float32); v = np.zeros((n, 3), np.float32); pi = np.zeros((n, 2662), np.float32)
    assert pm.build_history_batch(canon, v, pi) == n and n == len(pm.move_log()[0])
    assert np.abs(pi.sum(1) - 1).max() < 1e-4 and (v.sum(1) == 1).all()
    src = pi.reshape(n, 121, 22).sum(2).reshape(n, 11, 11)          # target mass per from-square
    own = np.where(canon[:, 3, 0, 0][:, None, None] == 1, canon[:, 2], canon[:, 0] + canon[:, 1])   # attackers move for player 0
    assert (src[own == 0] == 0).all()                               # only the mover's pieces carry policy mass
    c = pm.counters()
    assert c["evals"] > 0 and c["sims"] >= n * 23


def test_brandubh_selfplay_on_the_hip_net():
    """configs/brandubh.yaml end to end on the device: the wide-game engine + the bf16 MFMA net (4b32c zero-padded to the 64-channel
    kernels, 7x7) through the self-play harness; every game finishes, every history row is a distribution whose mass sits on
    legal moves of the stored position."""
    import alphazero as az
    from alphazero import selfplay, torch_net
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.max_batch_size = 64, 32, 32
    pp.mcts_visits = [20, 20]
    pp.model_groups = [0, 0]
    pp.history_enabled = True
    pp.cpuct, pp.epsilon = 1.25, 0.25
    net = torch_net.random_init(torch_net.brandubh_spec(), seed=4)
    res, (canon, v, pi) = selfplay.self_play(az.BrandubhGS(), pp, az.HipLeafNet(net), engines=2, seed=9)
    canon, v, pi = (t.cpu().numpy() for t in (canon, v, pi))
    n = len(canon)
    assert res.games == 64 and res.samples == n and abs(sum(res.win_rates) - 1) < 1e-6
    assert n > 64 and canon.shape[1:] == (7, 7, 7) and pi.shape == (n, 686)
    assert np.abs(pi.sum(1) - 1).max() < 1e-4 and (np.abs(v.sum(1) - 1) < 1e-6).all()
    assert res.leaf_evaluations > 0 and res.simulations >= 19 * n


# ---- test_canon_symmetry.py:131-190: a symmetry-mirrored OpenTafl game stays a perfect image ------------------------------------
_TW = _TH = 11
_TWH = _TW + _TH


def _t_decode(a):
    sq, rem = a // _TWH, a % _TWH
    fh, fw = sq // _TW, sq % _TW
    return ("row", fh, fw, rem) if rem < _TW else ("col", fh, fw, rem - _TW)


def _t_encode(kind, fh, fw, nl):
    return (fh * _TW + fw) * _TWH + (nl if kind == "row" else _TW + nl)


def _t_mirror_w(a):
    kind, fh, fw, nl = _t_decode(a)
    fw2 = _TW - 1 - fw
    return _t_encode("row", fh, fw2, _TW - 1 - nl) if kind == "row" else _t_encode("col", fh, fw2, nl)


def _t_rot90_cw(a):
    kind, fh, fw, nl = _t_decode(a)
    fh2, fw2 = fw, _TH - 1 - fh
    return _t_encode("col", fh2, fw2, nl) if kind == "row" else _t_encode("row", fh2, fw2, _TH - 1 - nl)


@pytest.mark.parametrize("name,move_map,board_map", [
    ("mirror_w", _t_mirror_w, lambda bp: bp[:, :, ::-1]),
    ("rot90_cw", _t_rot90_cw, lambda bp: np.rot90(bp, k=-1, axes=(1, 2))),
])
def test_reference_tafl_mirrored_game_stays_an_image(az, name, move_map, board_map):
    """the reference's own test of the Tafl rules under the two generating symmetries, on the device OpenTaflGS objects
    (12 games instead of 60: every call is a kernel launch)."""
    rng = np.random.default_rng(11)
    steps = 0
    board = lambda g: np.array(g.canonicalized())[:3]
    for _ in range(12):
        ga, gb = az.OpenTaflGS(), az.OpenTaflGS()
        assert np.array_equal(board_map(board(ga)), board(gb))
        for _ in range(50):
            va = np.flatnonzero(np.array(ga.valid_moves()))
            if len(va) == 0:
                break
            a = int(rng.choice(va)); b = move_map(a)
            assert np.array(gb.valid_moves())[b], f"{name}: mirrored move illegal"
            ga.play_move(a); gb.play_move(b)
            steps += 1
            assert np.array_equal(board_map(board(ga)), board(gb)), f"{name}: desync"
            if ga.scores() is not None:
                break
    assert steps > 100


def test_symmetry_kernel_policy_permutation_equals_the_reference_move_maps(az):
    """the device eightSym (tafl_helper.h:139-149) moves a one-hot policy exactly where test_canon_symmetry.py's independently
    written move maps send the move: symmetry 1 is the mirror, symmetry 2 the clockwise quarter turn of symmetry 0."""
    rng = np.random.default_rng(5)
    moves = rng.choice(2662, 40, replace=False)
    canon = np.zeros((len(moves), 8, 11, 11), np.float32); v = np.zeros((len(moves), 3), np.float32)
    pi = np.zeros((len(moves), 2662), np.float32); pi[np.arange(len(moves)), moves] = 1
    oc, ov, op = az.tafl_symmetries(11, canon, v, pi)
    where = op.argmax(2)                                    # [n, 8]: where each symmetry sends the move
    assert (op.sum(2) == 1).all() and (where[:, 0] == moves).all()
    images = {int(m): set(int(x) for x in where[i]) for i, m in enumerate(moves)}
    for m in moves:
        orbit, frontier = {int(m)}, [int(m)]                # the orbit under the two generators
        while frontier:
            a = frontier.pop()
            for f in (_t_mirror_w, _t_rot90_cw):
                b = f(a)
                if b not in orbit:
                    orbit.add(b); frontier.append(b)
        assert images[int(m)] == orbit


@pytest.mark.parametrize("name", ["TawlbwrddGS", "BrandubhGS", "OpenTaflGS"])
def test_playout_eval_matches_oracle(az, oracle, name):
    """playout_eval_batch (game_state.cc:10-95) for the Tafl family: uniform policy over the legal moves and the outcome of a
    random rollout with the repetition rule in force, equal to the oracle's rollout on the same stream."""
    Game = getattr(az, name)
    gid = {"TawlbwrddGS": oracle.GAME_TAWLBWRDD, "BrandubhGS": oracle.GAME_BRANDUBH, "OpenTaflGS": oracle.GAME_OPENTAFL}[name]
    rng = np.random.default_rng(12)
    states, ogames = [], []
    for i in range(10):
        g = Game(); og = oracle.Game(gid)
        for _ in range(int(rng.integers(0, 14))):
            legal = np.flatnonzero(g.valid_moves())
            if g.scores() is not None or len(legal) == 0:
                break
            m = int(rng.choice(legal)); g.play_move(m); og.play(m)
        if g.scores() is None:
            states.append(g); ogames.append(og)
    seeds = np.arange(500, 500 + len(states), dtype=np.uint64)
    vs, pis = az.playout_eval_batch(states, seeds)
    outcomes = np.zeros(3)
    for g, og, sd, v, pi in zip(states, ogames, seeds, vs, pis):
        ov, opi = oracle.playout_eval(og, int(sd))
        assert np.array_equal(v, ov) and np.array_equal(pi, opi)
        valid = np.asarray(g.valid_moves(), dtype=np.float32)
        assert np.array_equal(pi > 0, valid > 0) and abs(pi.sum() - 1) < 1e-5
        outcomes += v
    assert outcomes.sum() == len(states)


@pytest.mark.parametrize("name,evals", [
    ("BrandubhGS", ("PLAYOUT", "PLAYOUT")),
    ("BrandubhGS", ("PLAYOUT", "RANDOM")),
    ("OpenTaflGS", ("RANDOM", "PLAYOUT")),
    ("TawlbwrddGS", ("PLAYOUT", "PLAYOUT")),
])
def test_playout_seats_match_oracle(az, oracle, name, evals):
    """EvalType.PLAYOUT seats of the wide-game engine (play_manager.cc:580-582, game_state.cc:10-54): the rollout starts at
    the leaf with the descent's repetition counts and draws from the slot's rollout stream; moves, visit counts and scores
    equal the oracle's."""
    Game = getattr(az, name)
    gid = {"TawlbwrddGS": oracle.GAME_TAWLBWRDD, "BrandubhGS": oracle.GAME_BRANDUBH, "OpenTaflGS": oracle.GAME_OPENTAFL}[name]
    n = 3 if name == "TawlbwrddGS" else 4
    pp = az.PlayParams()
    pp.eval_type = [getattr(az.EvalType, e) for e in evals]
    pp.games_to_play, pp.concurrent_games, pp.max_batch_size = n, n, n
    pp.mcts_visits = [12, 10]
    pp.cpuct, pp.fpu_reduction = 1.25, 0.25
    seed = 77
    pm = az.PlayManager(Game(), pp, seed=seed, log_moves=True)
    pm.play()
    rows, counts = pm.move_log()
    tot = np.zeros(3, np.float32)
    for s in range(n):
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games = 1, 1
        o = oracle.PlayManager(gid, one, oracle.slot_seed(seed, s), per_slot_rng=False)
        o.run()
        orows, ocounts = o.moves()
        sel = rows[:, 0] == s
        assert np.array_equal(rows[sel][:, 1:], orows[:, 1:]), s
        assert np.array_equal(counts[sel], ocounts), s
        tot += o.scores()
    assert np.array_equal(pm.scores(), tot)
