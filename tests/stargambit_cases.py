"""The reference's StarGambit rule tests as ONE script over any implementation of the game API.

Cases follow /root/reference/src/star_gambit_gs_test.cc and star_gambit_unified_gs_test.cc (cited per case): the move
notation of its parse_move helper (:22-221), its scripted games and its known answers, re-expressed as data and Python
asserts.  `make(variant)` returns a game in the variant's OWN action space (StarGambit{Skirmish,Showdown,Clash,Battle}GS),
`make_unified(pinned)` a StarGambitUnifiedGS; both expose the reference's method names (valid_moves, play_move, scores,
canonicalized, current_player, current_turn, get_units, copy, ==, has_taken_action, num_moves).
tests/test_oracle_stargambit.py runs it on the CPU oracle, tests/test_gpu_stargambit.py on the device objects."""
import numpy as np

SIDE = {0: 5, 1: 5, 2: 5, 3: 6}
START = {0: (3, 1, 0), 1: (4, 0, 1), 2: (3, 2, 1), 3: (4, 3, 2)}
FACING = {"e": 0, "ne": 1, "nw": 2, "w": 3, "sw": 4, "se": 5}
MOVE_SLOT = {"f": 0, "fl": 1, "fr": 2, "l": 3, "r": 4}
FIRE_SLOT = {"f": 5, "l": 6, "fl": 6, "r": 7, "fr": 7, "rl": 8, "rr": 9}
TYPE = {"f": 0, "c": 1, "d": 2}
DIRS = [(1, 0), (1, -1), (0, -1), (-1, 0), (-1, 1), (0, 1)]


def dim(v): return 2 * SIDE[v] + 1
def spatial(v): return dim(v) * dim(v) * 10
def deploy_offset(v): return spatial(v)
def end_offset(v): return spatial(v) + 18
def num_moves(v): return spatial(v) + 19
def encode_deploy(v, t, facing): return deploy_offset(v) + t * 6 + facing


def in_bounds(q, r, side): return abs(q) <= side and abs(r) <= side and abs(-q - r) <= side


def parse_move(game, v, text):
    """parse_move, star_gambit_gs_test.cc:64-221 (None = not parseable / no such unit)"""
    w = text.split()
    p1 = game.current_player() == 1
    D = dim(v)

    def enc(u, slot):
        row, col = u.anchor_q + SIDE[v], u.anchor_r + SIDE[v]
        if p1:
            row, col = D - 1 - row, D - 1 - col
        return (row * D + col) * 10 + slot

    def unit(tag):
        t, s = TYPE[tag[0]], int(tag[1:]) - 1
        for u in game.get_units():
            if u.player == game.current_player() and u.type == t and u.slot == s:
                return u
        return None

    if w[0] == "e":
        return end_offset(v)
    if w[0] == "m":
        u = unit(w[1])
        return None if u is None else enc(u, MOVE_SLOT[w[2]])
    if w[0] == "f":
        u = unit(w[1])
        if u is None:
            return None
        return enc(u, 5 if w[1][0] == "f" else FIRE_SLOT[w[2]])
    if w[0] == "d":
        f = FACING[w[2]]
        return encode_deploy(v, TYPE[w[1]], (f + 3) % 6 if p1 else f)
    return None


def play(game, v, text):
    """play_notation, :1958-1969: parse, require validity, play"""
    a = parse_move(game, v, text)
    if a is None or game.valid_moves()[a] == 0:
        return False
    game.play_move(a)
    return True


def first_valid(valids, lo, hi):
    nz = np.nonzero(np.asarray(valids)[lo:hi])[0]
    return None if nz.size == 0 else int(nz[0]) + lo


# ------------------------------------------------------------------------------------------------ cases
def case_initial_state(make, make_unified):   # GameState.InitialState / InitialUnitsArePortals / ScoresNotOverInitially, :414-516
    for v in range(4):
        g = make(v)
        assert (g.current_player(), g.current_turn(), g.num_moves()) == (0, 1, num_moves(v))
        assert not g.has_taken_action()
        us = g.get_units()
        assert [u.type for u in us] == [3, 3] and [u.player for u in us] == [0, 1]
        s = SIDE[v]
        assert (us[0].anchor_q, us[0].anchor_r, us[0].facing, us[0].hp) == (0, s, 2, 5)     # portal hexes, :361-410
        assert (us[1].anchor_q, us[1].anchor_r, us[1].facing, us[1].hp) == (0, -s, 5, 5)
        assert g.scores() is None
        assert g.canonicalized().shape == (32, dim(v), dim(v))


def case_turn_one_is_deploy_only(make, make_unified):   # ValidMovesOnTurnOne, TurnOneIsDeployOnly, ValidDeployFacings, :434-456, 531-540, 658-666
    for v in range(4):
        g = make(v)
        va = np.asarray(g.valid_moves())
        assert va[:deploy_offset(v)].sum() == 0 and va[end_offset(v)] == 0
        want = np.zeros(18, np.uint8)
        for t in range(3):
            if START[v][t] == 0:
                continue
            for f in ((0, 1, 2, 3) if t == 2 else (1, 2, 3)):
                want[t * 6 + f] = 1
        assert np.array_equal(va[deploy_offset(v):end_offset(v)], want), (v, va[deploy_offset(v):end_offset(v)])


def case_deploy_switches_player(make, make_unified):   # DeployAction, CopyEquality, :458-501
    g = make(0)
    c = g.copy()
    assert g == c
    a = first_valid(g.valid_moves(), deploy_offset(0), end_offset(0))
    g.play_move(a)
    assert (g.current_player(), g.current_turn()) == (1, 2)
    assert not (g == c)
    u = g.get_units()[2]
    assert (u.type, u.player, u.slot, u.hp, u.anchor_q, u.anchor_r, u.facing, u.moves_left) == (0, 0, 0, 3, 0, 4, 1, 0)   # deploy hex, :632-656


def case_p1_deploy_facings(make, make_unified):   # DreadnoughtDeployAllFacingsP0/P1, :2140-2195
    g = make(3)
    va = np.asarray(g.valid_moves())
    assert all(va[deploy_offset(3) + 12 + f] == 1 for f in (0, 1, 2, 3))
    g.play_move(first_valid(va, deploy_offset(3), deploy_offset(3) + 6))
    assert g.current_player() == 1
    va = np.asarray(g.valid_moves())
    for world in (0, 3, 4, 5):
        assert va[deploy_offset(3) + 12 + (world + 3) % 6] == 1
    for world in (4, 5, 0):          # fighters / cruisers of player 1, star_gambit_gs.cc:187-191
        assert va[deploy_offset(3) + (world + 3) % 6] == 1 and va[deploy_offset(3) + 6 + (world + 3) % 6] == 1
    assert va[deploy_offset(3):end_offset(3)].sum() == 3 + 3 + 4


def case_notation_game(make, make_unified):   # FullGame.PlayWithNotation / DeployAndMoveSequence / CruiserDeployAndMove, :1971-2037
    g = make(0)
    assert play(g, 0, "d f ne") and (g.current_player(), g.current_turn()) == (1, 2)
    assert play(g, 0, "d f se") and (g.current_player(), g.current_turn()) == (0, 3)
    assert play(g, 0, "m f1 f") and g.has_taken_action() and g.current_player() == 0
    f = [u for u in g.get_units() if u.type == 0 and u.player == 0][0]
    assert (f.anchor_q, f.anchor_r, f.facing, f.moves_left) == (1, 3, 1, 1)    # (0,4) + NE
    assert play(g, 0, "m f1 f")
    f = [u for u in g.get_units() if u.type == 0 and u.player == 0][0]
    assert (f.anchor_q, f.anchor_r, f.moves_left) == (2, 2, 0)
    assert not play(g, 0, "m f1 f")                     # no moves left
    assert play(g, 0, "e") and g.current_player() == 1 and not g.has_taken_action()
    g = make(0)
    assert play(g, 0, "d c ne") and play(g, 0, "d f se")
    c = [u for u in g.get_units() if u.type == 1][0]
    assert (c.anchor_q, c.anchor_r, c.facing) == (1, 3, 1)       # rear on the deploy hex, front one step along the facing
    va = np.asarray(g.valid_moves())
    assert any(va[a] for a in range(spatial(0)) if a % 10 < 5)
    assert play(g, 0, "m c1 f")
    c = [u for u in g.get_units() if u.type == 1][0]
    assert (c.anchor_q, c.anchor_r, c.facing, c.moves_left) == (2, 2, 1, 0)


def case_no_fire_without_target(make, make_unified):   # FireValidation.NoFireWithoutTarget, :1497-1530
    g = make(0)
    for _ in range(2):
        g.play_move(first_valid(g.valid_moves(), deploy_offset(0), deploy_offset(0) + 6))
    va = np.asarray(g.valid_moves())
    assert not any(va[a] for a in range(spatial(0)) if a % 10 >= 5)


def case_fire_and_damage(make, make_unified):   # FireAvailableWhenTargetInRange + FullGame.FireWhenInRange, :1532-1590, 2074-2133
    g = make(0)
    assert play(g, 0, "d f nw") and play(g, 0, "d f se")
    fired = None
    for _ in range(30):
        va = np.asarray(g.valid_moves())
        fire = [a for a in range(spatial(0)) if a % 10 >= 5 and va[a]]
        if fire:
            before = {(u.player, u.type, u.slot): u.hp for u in g.get_units()}
            shooter = g.current_player()
            g.play_move(fire[0])
            after = {(u.player, u.type, u.slot): u.hp for u in g.get_units()}
            fired = (shooter, before, after)
            break
        fwd = [a for a in range(spatial(0)) if a % 10 == 0 and va[a]]
        g.play_move(fwd[0] if fwd else int(np.nonzero(va)[0][0]))
        if g.scores() is not None:
            break
    assert fired is not None
    shooter, before, after = fired
    hit = [(k, before[k] - after.get(k, 0)) for k in before if before[k] != after.get(k, 0)]
    assert len(hit) == 1 and hit[0][0][0] == 1 - shooter and hit[0][1] in (1, 2), hit   # one enemy unit, 1 (range 2) or 2 (range 1)
    # fighters that walk straight at each other along NW / SE first see each other at range 2
    assert hit[0][1] == 1


def case_cannon_fires_once(make, make_unified):   # is_fire_valid's cannons_fired bit, star_gambit_gs.cc:715-727
    g = make(0)
    assert play(g, 0, "d f nw") and play(g, 0, "d f se")
    for _ in range(40):
        va = np.asarray(g.valid_moves())
        fire = [a for a in range(spatial(0)) if a % 10 == 5 and va[a]]
        if fire:
            g.play_move(fire[0])
            assert np.asarray(g.valid_moves())[fire[0]] == 0
            return
        fwd = [a for a in range(spatial(0)) if a % 10 == 0 and va[a]]
        g.play_move(fwd[0] if fwd else int(np.nonzero(va)[0][0]))
    raise AssertionError("no fire found")


def case_fighter_and_cruiser_move_options(make, make_unified):   # MovementConstraints, :1593-1680
    g = make(0)
    assert play(g, 0, "d f nw") and play(g, 0, "d c se")
    va = np.asarray(g.valid_moves())
    moves = sorted(a % 10 for a in range(spatial(0)) if va[a] and a % 10 < 5)
    assert moves == [0, 1, 2], moves                           # fighter: forward, forward-left, forward-right
    assert play(g, 0, "e") is False                            # nothing done yet: end turn is not legal
    assert play(g, 0, "m f1 f") and play(g, 0, "e")
    va = np.asarray(g.valid_moves())
    moves = sorted(a % 10 for a in range(spatial(0)) if va[a] and a % 10 < 5)
    assert moves == [0, 1, 2, 3, 4], moves                     # cruiser: all five


def case_slot_numbering(make, make_unified):   # SlotNumbering.MultipleUnitsGetDifferentSlots, :1683-1757
    g = make(0)
    assert play(g, 0, "d f ne") and play(g, 0, "d f se")
    assert play(g, 0, "m f1 f") and play(g, 0, "m f1 f") and play(g, 0, "e")
    assert play(g, 0, "m f1 f") and play(g, 0, "m f1 f") and play(g, 0, "e")
    assert play(g, 0, "d f ne")
    fs = sorted(u.slot for u in g.get_units() if u.player == 0 and u.type == 0)
    assert fs == [0, 1]


def case_threefold_repetition(make, make_unified):   # ThreefoldRepetition.DrawOnThirdOccurrence, :2711-2754
    g = make(0)
    assert play(g, 0, "d c ne") and play(g, 0, "d c sw")
    assert g.scores() is None
    cycle = ["m c1 l", "e", "m c1 l", "e", "m c1 r", "e", "m c1 r", "e"]
    for t in cycle:
        assert play(g, 0, t), t
    assert g.scores() is None
    for t in cycle:
        assert play(g, 0, t), t
    sc = g.scores()
    assert sc is not None and list(sc) == [0.0, 0.0, 1.0]
    assert np.asarray(g.valid_moves()).sum() == 0            # game over: no legal moves, star_gambit_gs.cc:789-791
    assert g.canonicalized().shape == (32, 11, 11)           # CanonicalizedWorksAfterGameEnd, :2591-2615


def case_history_cleared_on_deploy(make, make_unified):   # HistoryClearedOnDeploy + rep channel, :2621-2668, 2756-2783
    g = make(0)
    obs = g.canonicalized()
    hexes = [(q + 5, r + 5) for q in range(-5, 6) for r in range(-5, 6) if in_bounds(q, r, 5)]
    assert all(obs[23, a, b] == 0.5 for a, b in hexes)
    assert obs[23].sum() == 0.5 * 91 and obs[0].sum() == 91
    assert play(g, 0, "d f ne") and play(g, 0, "d f sw")
    for t in ("m f1 f", "e", "m f1 f", "e"):
        assert play(g, 0, t)
    g.play_move(first_valid(g.valid_moves(), deploy_offset(0), end_offset(0)))
    assert g.scores() is None
    assert g.canonicalized()[23, 5, 5] == 0.5


def case_repetition_channel_counts(make, make_unified):   # canonical channel 23 = {0, .5, 1}, star_gambit_gs.cc:1586-1599
    g = make(0)
    assert play(g, 0, "d c ne") and play(g, 0, "d c sw")
    cycle = ["m c1 l", "e", "m c1 l", "e", "m c1 r", "e", "m c1 r", "e"]
    for t in cycle:
        assert play(g, 0, t)
    assert g.canonicalized()[23, 5, 5] == 1.0                # the position after both deploys, seen for the second time


def case_first_valid_game_ends(make, make_unified):   # FullGame.CompleteGameToVictory / EndToEnd / TerminalStates.ScoresSumToOne, :1760-1811, 2039-2071, 2503-2527
    for v in range(4):
        g = make(v)
        for _ in range(3000):
            va = np.asarray(g.valid_moves())
            if va.sum() == 0 or g.scores() is not None:
                break
            g.play_move(int(np.nonzero(va)[0][0]))
        sc = g.scores()
        assert sc is not None and float(np.sum(sc)) == 1.0 and sorted(sc) == [0.0, 0.0, 1.0]


def case_p1_observation(make, make_unified):   # P1Canonicalization.*, :811-925
    g = make(0)
    g.play_move(encode_deploy(0, 0, 1))
    g.play_move(encode_deploy(0, 0, 4))       # canonical SW = world NE for player 1
    g.play_move(end_offset(0))                # the reference's test ends the turn without acting: play_move does not validate
    assert g.current_player() == 1
    obs = g.canonicalized()
    assert (obs[4] > 0).sum() == 3 and (obs[8] > 0).sum() == 3
    # own portal of the player to move is drawn where player 0 sees its own: (q, r) -> (-q, -r)
    assert obs[4, 0 + 5, 5 + 5] == 1 and obs[4, 1 + 5, 4 + 5] == 1 and obs[4, -1 + 5, 5 + 5] == 1
    f1 = [u for u in g.get_units() if u.player == 1 and u.type == 0][0]
    assert (f1.anchor_q, f1.anchor_r, f1.facing) == (0, -4, 1)
    assert obs[1, 0 + 5, 4 + 5] == 1 and obs[9 + 4, 0 + 5, 4 + 5] == 1       # my fighter, heading rotated by 180 degrees
    va = np.asarray(g.valid_moves())
    sp = [a for a in range(spatial(0)) if va[a]]
    assert sp and all((a // 10) == (5 * 11 + 9) for a in sp)                  # at the rotated anchor cell (row 5, col 9)
    # forward (1, -5) and forward-left (0, -5) are the own portal; the deploy hex is taken and nothing was done yet:
    # forward-right is the only legal move of the position
    assert [a % 10 for a in sp] == [2] and va.sum() == 1
    g.play_move(sp[0])
    f1 = [u for u in g.get_units() if u.player == 1 and u.type == 0][0]
    assert (f1.anchor_q, f1.anchor_r, f1.facing, f1.moves_left) == (1, -4, 0, 1)


def case_observation_channels(make, make_unified):   # CharacterizationObservation + canonicalized(), star_gambit_gs.cc:1384-1669
    g = make(2)   # Clash: 3 fighters, 2 cruisers, 1 dreadnought
    assert play(g, 2, "d d nw") and play(g, 2, "d c se")
    obs = g.canonicalized()
    d = [u for u in g.get_units() if u.type == 2][0]
    assert (d.anchor_q, d.anchor_r, d.facing) == (0, 3, 2)          # deploy hex (0,4) + NW
    cells = [(0, 3), (0, 4), (1, 3)]                                   # anchor, rear (SE of the anchor), rear-right (rot(rear, 1) = E)
    for q, r in cells:
        assert obs[1 + 2, q + 5, r + 5] == 1 and obs[9 + 2, q + 5, r + 5] == 1 and obs[15, q + 5, r + 5] == 1.0
        assert obs[16, q + 5, r + 5] == 1.0      # turn 3: the dreadnought's move was reset at the start of the owner's turn
    assert obs[1 + 2].sum() == 3
    assert [obs[17 + s, 0 + 5, 3 + 5] for s in range(5)] == [0, 1, 1, 1, 1]   # four unfired dreadnought cannons at the anchor
    assert sum(obs[17 + s].sum() for s in range(5)) == 4                       # enemy cruiser: all cannons spent by the deploy
    assert obs[5 + 1].sum() == 2 and obs[22].sum() == 0
    assert np.isclose(obs[24, 5, 5], 1.0) and obs[25, 5, 5] == 1.0 and obs[26, 5, 5] == 0.0
    assert obs[27, 5, 5] == 1.0 and obs[28, 5, 5] == 0.5 and obs[29, 5, 5] == 1.0
    assert obs[30, 5, 5] == 1.0 and obs[31, 5, 5] == 1.0


def case_portal_kill_wins(make, make_unified):   # TerminalStates.WinnerGetsOne, :2529-2558 + check_game_end, star_gambit_gs.cc:1313-1345
    g = make(0)
    assert play(g, 0, "d f nw") and play(g, 0, "d f se")
    # player 0's fighter walks up the q = 0 column to the enemy portal and shoots it down; player 1 shuffles its fighter sideways
    portal_hp = 5
    for _ in range(400):
        if g.scores() is not None:
            break
        va = np.asarray(g.valid_moves())
        if g.current_player() == 0:
            fire = [a for a in range(spatial(0)) if a % 10 == 5 and va[a]]
            fwd = [a for a in range(spatial(0)) if a % 10 == 0 and va[a]]
            if fire:
                g.play_move(fire[0])
            elif fwd:
                g.play_move(fwd[0])
            elif va[end_offset(0)]:
                g.play_move(end_offset(0))
            else:
                g.play_move(int(np.nonzero(va)[0][0]))
        else:
            side = [a for a in range(spatial(0)) if a % 10 in (1, 2) and va[a]]
            if not g.has_taken_action() and side:
                g.play_move(side[0])
            elif va[end_offset(0)]:
                g.play_move(end_offset(0))
            else:
                g.play_move(int(np.nonzero(va)[0][0]))
        ps = [u for u in g.get_units() if u.type == 3 and u.player == 1]
        portal_hp = ps[0].hp if ps else 0
    sc = g.scores()
    assert sc is not None
    if portal_hp == 0:
        assert list(sc) == [1.0, 0.0, 0.0]


def case_unified_shapes_and_remap(make, make_unified):   # star_gambit_unified_gs_test.cc:50-386
    for v in range(4):
        u, p = make_unified(v), make(v)
        assert u.num_moves() == 1709 and u.get_variant_id() == v and u.num_variants() == 4
        assert u.relative_values() and p.relative_values()
        rng = np.random.default_rng(100 + v)
        for step in range(400):
            uv, pv = np.asarray(u.valid_moves()), np.asarray(p.valid_moves())
            assert uv.shape == (1709,)
            idx = np.nonzero(pv)[0]
            # to_unified_action, star_gambit_gs.cc:2522-2537
            if v == 3:
                mapped = idx
            else:
                mapped = np.array([(((a // 10) // 11 + 1) * 13 + ((a // 10) % 11 + 1)) * 10 + a % 10 if a < 1210 else 1690 + (a - 1210)
                                   for a in idx], np.int64)
            assert np.array_equal(np.nonzero(uv)[0], np.sort(mapped)), (v, step)
            uo, po = u.canonicalized(), p.canonicalized()
            assert uo.shape == (36, 13, 13)
            off = 0 if v == 3 else 1
            D = dim(v)
            assert np.array_equal(uo[:32, off:off + D, off:off + D], po)
            border = uo[:32].copy(); border[:, off:off + D, off:off + D] = 0
            assert border.sum() == 0
            for k in range(4):
                assert np.array_equal(uo[32 + k], uo[0] if k == v else np.zeros((13, 13), np.float32))
            if pv.sum() == 0:
                break
            j = int(rng.integers(len(idx)))
            p.play_move(int(idx[j])); u.play_move(int(mapped[j]))
            assert (u.current_player(), u.current_turn()) == (p.current_player(), p.current_turn())
            su, sp = u.scores(), p.scores()
            assert (su is None) == (sp is None) and (su is None or list(su) == list(sp))


def case_unit_and_action_space_constants(make, make_unified):
    """UnitProperties.MaxHP / NumCannons / UnitSize (:1369-1397), ActionSpace.*ActionCounts / BoardSizes (:1401-1462),
    CharacterizationFighter / Cruiser / Dreadnought unit shapes (:2206-2240, 2410-2460) - the reference checks free functions
    and class constants that have no binding; here the same numbers are read off what the game shows: the action-space
    size, the hex mask of the observation, and a freshly deployed unit's hit points, cannon planes and cells."""
    max_hp, cannons, size = {0: 3, 1: 4, 2: 6}, {0: 1, 1: 3, 2: 4}, {0: 1, 1: 2, 2: 3}
    for v, (moves, hexes) in {0: (1229, 91), 1: (1229, 91), 2: (1229, 91), 3: (1709, 127)}.items():
        g = make(v)
        assert g.num_moves() == moves == dim(v) * dim(v) * 10 + 18 + 1
        assert int(np.asarray(g.canonicalized())[0].sum()) == hexes == 3 * SIDE[v] ** 2 + 3 * SIDE[v] + 1
    for t in range(3):
        v = 3                                            # Battle starts with every unit type in reserve
        g = make(v)
        before = np.asarray(g.canonicalized())
        a = first_valid(g.valid_moves(), deploy_offset(v) + 6 * t, deploy_offset(v) + 6 * t + 6)
        g.play_move(a)
        u = [x for x in g.get_units() if x.type == t][0]
        assert u.hp == max_hp[t] and u.player == 0
        obs = np.asarray(g.canonicalized())               # player 1's view now: player 0's new unit is an OPPONENT unit
        opp_type_plane = 5 + t                            # channels 1-4 mine (F, C, D, portal), 5-8 the opponent's
        assert int(obs[opp_type_plane].sum() - before[1 + t].sum()) == size[t]
        # unfired cannon slots (channels 17-21) are shown for the viewer's OWN units, at the anchor hex only: once player 1 has
        # deployed too, player 0 sees all the cannons of its fresh unit
        g.play_move(first_valid(g.valid_moves(), deploy_offset(v), end_offset(v)))
        assert g.current_player() == 0
        assert int(np.asarray(g.canonicalized())[17:22].sum()) == cannons[t]


ALL_CASES = [v for k, v in sorted(globals().items()) if k.startswith("case_")]
