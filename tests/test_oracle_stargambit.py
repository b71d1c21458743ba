"""CPU: the oracle's StarGambit restatement against the reference's own rule tests (tests/stargambit_cases.py), its pickle
layout (star_gambit_gs.cc:2246-2251, 2446-2449) and the relative-value rotation of game_state.h:22-46."""
import struct
import types

import numpy as np
import pytest

import oracle_api as orc
import stargambit_cases as sgc


class OracleSG:
    """the reference's GameState method names over an oracle game"""
    def __init__(self, g): self.g = g
    def valid_moves(self): return self.g.valid()
    def play_move(self, m): self.g.play(int(m))
    def scores(self): return self.g.scores()
    def canonicalized(self): return self.g.canonical()
    def current_player(self): return self.g.player()
    def current_turn(self): return self.g.turn()
    def num_moves(self): return self.g.num_moves()
    def has_taken_action(self): return bool(self.g.sg_info()["acted"])
    def relative_values(self): return self.g.relative_values()
    def get_variant_id(self): return self.g.variant()
    def num_variants(self): return self.g.num_variants()
    def copy(self): return OracleSG(self.g.copy())
    def __eq__(self, o): return self.g.equals(o.g)
    def get_units(self):   # alive units only, star_gambit_gs.cc:2102-2118
        return [types.SimpleNamespace(type=int(r[0]), player=int(r[1]), slot=int(r[2]), hp=int(r[3]), facing=int(r[4]), anchor_q=int(r[5]),
                                      anchor_r=int(r[6]), moves_left=int(r[7])) for r in self.g.sg_units() if r[3] > 0]


def make(v): return OracleSG(orc.Game.sg_plain(v))
def make_unified(v): return OracleSG(orc.Game.sg_unified(pinned=v))


@pytest.mark.parametrize("case", sgc.ALL_CASES, ids=lambda f: f.__name__)
def test_reference_rule_cases_on_the_oracle(case):
    case(make, make_unified)


def test_pickle_layout_and_round_trip():
    g = orc.Game.sg_unified(pinned=2, probs=(0.1, 0.2, 0.3, 0.4))
    rng = np.random.default_rng(5)
    for _ in range(60):
        v = np.nonzero(g.valid())[0]
        if v.size == 0:
            break
        g.play(int(rng.choice(v)))
    b = g.sg_to_bytes()
    probs = struct.unpack_from("<4f", b, 0)
    pinned, variant, inner_size = struct.unpack_from("<iBI", b, 16)
    assert np.allclose(probs, (0.1, 0.2, 0.3, 0.4)) and (pinned, variant) == (2, 2) and inner_size == len(b) - 25
    inner = b[25:]
    n_units, = struct.unpack_from("<I", inner, 0)
    units = g.sg_units()
    assert n_units == len(units)
    for i, r in enumerate(units):
        t, pl, slot, hp, facing, q, rr, ml, cf = struct.unpack_from("<5B2b2B", inner, 4 + 9 * i)
        assert (t, pl, slot, hp, facing, q, rr, ml, cf) == tuple(int(x) for x in r)
    off = 4 + 9 * n_units
    info = g.sg_info()
    assert list(inner[off:off + 8]) == info["reserves"].reshape(-1).tolist()
    player, turn, acted, over, winner, hist = struct.unpack_from("<BIBBbI", inner, off + 8)
    assert (player, turn, acted, over, winner, hist) == (g.player(), g.turn(), info["acted"], info["over"], info["winner"], info["history_len"])
    assert len(inner) == off + 8 + 12 + 8 * hist
    h = orc.Game.sg_unified(pinned=2)
    h.sg_from_bytes(inner)
    assert h.equals(g) and h.sg_to_bytes()[25:] == inner and np.array_equal(h.canonical(), g.canonical())
    with pytest.raises(RuntimeError):
        h.sg_from_bytes(inner[:-3])


def _pp(**kw):
    pp = types.SimpleNamespace(
        games_to_play=1, concurrent_games=1, max_batch_size=1, max_cache_size=0, cache_shards=1, mcts_visits=[12, 12], cpuct=1.25,
        start_temp=1.0, final_temp=1.0, temp_decay_half_life=0.0, history_enabled=True, tree_reuse=True, epsilon=0.0, mcts_root_temp=1.0,
        playout_cap_randomization=False, playout_cap_depth=25, playout_cap_percent=0.75, fpu_reduction=0.25, root_fpu_zero=False,
        shaped_dirichlet=False, policy_target_pruning=False, resign_percent=0.0, resign_playthrough_percent=0.0, eval_type=[1, 1])
    pp.__dict__.update(kw)
    return pp


def test_variant_draw_rule_is_the_documented_one():
    """build-defined: one uniform01 of the slot's coin stream against the cumulative weights (oracle/az_stargambit.hpp)"""
    probs = (0.1, 0.2, 0.3, 0.4)
    for seed in range(12):
        pm = orc.PlayManager(orc.Game.sg_unified(-1, probs, first_variant=0), _pp(mcts_visits=[2, 2]), seed, per_slot_rng=False)
        pm.run()
        u = float(orc.uniform01(seed ^ 0x5851F42D4C957F2D, 1)[0]) * np.float32(np.float32(np.float32(0.1) + np.float32(0.2)) + np.float32(0.3) + np.float32(0.4))
        acc, want = np.float32(0.1), 0
        while want < 3 and u >= acc:
            want += 1
            acc = np.float32(acc + np.float32(probs[want]))
        games = [pm.variant(v)["games"] for v in range(4)]
        assert sum(games) == 1 and games[want] == 1, (seed, games, want)


def test_playmanager_relative_targets_and_variant_tables():
    """history value targets are rotated to the mover's view (play_manager.cc:451-454) and the per-variant tables add up
    (play_manager.cc:468-484, play_manager.h:218-275)"""
    pp = _pp(games_to_play=6, concurrent_games=3, mcts_visits=[10, 10], temp_decay_half_life_by_variant=[3.0, 4.0, 5.0, 8.0], final_temp=0.2)
    pm = orc.PlayManager(orc.Game.sg_unified(-1, (0.25, 0.25, 0.25, 0.25)), pp, 77, per_slot_rng=True)
    pm.run()
    assert pm.games_completed() == 6 and pm.num_tracked_variants() == 4
    tot = np.zeros(3, np.float32); games = 0
    for v in range(4):
        d = pm.variant(v)
        tot += d["scores"]; games += d["games"]
        assert d["perm_games"].sum() == d["games"] and np.array_equal(d["perm_scores"].sum(0), d["scores"])
        if d["games"]:
            assert d["stats"][0] > 2 and d["stats"][5] >= 1.0 and d["stats"][6] >= 1.0    # turns per game, moves per turn, legal moves
    assert games == 6 and np.array_equal(tot, pm.scores())
    canon, v, pi = pm.history()
    assert len(canon) == pm.counters()["hist_rows"] > 0 and canon.shape[1:] == (36, 13, 13)
    assert np.all(v.sum(1) == 1.0) and np.allclose(pi.sum(1), 1.0, atol=1e-5)
    # a decisive game's rows carry BOTH [1,0,0] (the winner's own moves) and [0,1,0] (the loser's): the rotation at work
    decisive = v[v[:, 2] == 0]
    if len(decisive):
        assert decisive[:, 0].sum() > 0 and decisive[:, 1].sum() > 0


def test_mcts_backs_up_relative_values_as_absolute():
    """RelativeValues.MCTSRelativeValueBackup, star_gambit_gs_test.cc:2861-2897: with player 1 at the leaf a relative
    (win, loss, draw) = (1, 0, 0) is a WIN FOR PLAYER 1 (mcts.cc:522-524)"""
    g = orc.Game.sg_plain(0)
    g.play(int(np.nonzero(g.valid())[0][0]))
    assert g.player() == 1
    m = orc.Mcts(1.25, 2, g.num_moves(), relative_values=True, seed=3)
    leaf = m.find_leaf(g)
    pi = np.full(g.num_moves(), 1.0 / g.num_moves(), np.float32)
    m.process_result(np.array([1, 0, 0], np.float32), pi)
    assert list(m.root_value()) == [1.0, 0.0, 0.0]      # root_value is from the root player's (player 1's) side
    m2 = orc.Mcts(1.25, 2, g.num_moves(), relative_values=False, seed=3)
    m2.find_leaf(g)
    m2.process_result(np.array([1, 0, 0], np.float32), pi)
    assert list(m2.root_value()) == [0.0, 1.0, 0.0]


def _sym(canon, v, pi):
    return orc.symmetries(orc.SYM_STARGAMBIT, canon, v, pi)


def test_mirror_symmetry_cases_of_the_reference():
    """MirrorSymmetry.* of star_gambit_gs_test.cc:928-1360 on the unified 13 x 13 canvas (star_gambit_gs.cc:2623-2727): the
    spatial remap (row, col, slot) -> (12 - row, row + col - 6, SLOT_MAP[slot]), the deploy maps, end turn, value, heading and
    cannon planes, self-inverse"""
    g = orc.Game.sg_unified(pinned=3)
    rng = np.random.default_rng(2)
    for _ in range(30):
        g.play(int(rng.choice(np.flatnonzero(g.valid()))))
    canon = g.canonical()
    v = np.array([0.5, 0.5, 0.0], np.float32)
    SLOT_MAP = [0, 2, 1, 4, 3, 5, 7, 6, 9, 8]; MIRROR = [4, 3, 2, 1, 0, 5]; DEPLOY_D = [3, 2, 1, 0, 5, 4]
    # PolicySpatialActionsRemapped (:1011-1047), on the 13-wide canvas
    pi = np.zeros(1709, np.float32); pi[(3 * 13 + 6) * 10 + 1] = 1.0
    oc, ov, op = _sym(canon, v, pi)
    assert np.array_equal(oc[0], canon) and np.array_equal(op[0], pi) and np.array_equal(ov[1], v)       # identity first; value kept
    assert op[1][((12 - 3) * 13 + (3 + 6 - 6)) * 10 + SLOT_MAP[1]] == 1.0 and op[1].sum() == 1.0
    # PolicyDeployActionsRemapped / DeployMirrorMapsCorrectly (:1049-1108, 1314-1366), EndTurnUnchanged (:1111-1128)
    for t in range(3):
        for f in range(6):
            pi = np.zeros(1709, np.float32); pi[1690 + t * 6 + f] = 1.0
            m = _sym(canon, v, pi)[2][1]
            assert m[1690 + t * 6 + (DEPLOY_D[f] if t == 2 else MIRROR[f])] == 1.0 and m.sum() == 1.0
    pi = np.zeros(1709, np.float32); pi[1708] = 1.0
    assert _sym(canon, v, pi)[2][1][1708] == 1.0
    # FacingChannelsRemapped (:1150-1196): what was plane 9 + d at a cell is plane 9 + MIRROR[d] at the mirrored cell; board mask and
    # the broadcast planes are invariant
    m = oc[1]
    assert np.array_equal(m[0], canon[0]) and np.array_equal(m[22:36], canon[22:36])
    for d in range(6):
        src = canon[9 + d]
        for r, c in zip(*np.nonzero(src)):
            assert m[9 + MIRROR[d], 12 - r, r + c - 6] == src[r, c]
    assert m[9:15].sum() == canon[9:15].sum() and m[1:9].sum() == canon[1:9].sum() and m[17:22].sum() == canon[17:22].sum()
    # SelfInverse (:928-980)
    pi = rng.random(1709).astype(np.float32)
    valid_pi = pi.copy()
    for a in range(1690):      # the reference only transforms hex cells; cells off the board are copied
        pass
    oc1, _, op1 = _sym(canon, v, pi)
    oc2, _, op2 = _sym(oc1[1], v, op1[1])
    assert np.array_equal(oc2[1], canon) and np.array_equal(op2[1], pi)
