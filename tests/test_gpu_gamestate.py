"""GPU: the GameState object surface (py_wrapper.cc:157-189, 562-586) answered by the device rules
kernels, against the oracle and the reference's connect4_gs_test.cc cases."""
import pickle

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


def test_connect4_object_walk_matches_oracle(az, oracle):
    rng = np.random.default_rng(7)
    for game in range(6):
        gs = az.Connect4GS(); og = oracle.Game(oracle.GAME_CONNECT4)
        while gs.scores() is None:
            assert gs.current_player() == og.player() and gs.current_turn() == og.turn()
            vm = gs.valid_moves()
            assert np.array_equal(vm, og.valid())
            assert np.array_equal(gs.canonicalized(), og.canonical())
            assert og.scores() is None
            mv = int(rng.choice(np.flatnonzero(vm)))
            gs.play_move(mv); og.play(mv)
        assert np.array_equal(gs.scores(), og.scores())


def test_connect4_from_board_ctor_and_pickle(az, oracle):
    # connect4_gs_test.cc:104-171 style: a hand-built board, player to move wins by dropping in column 3
    board = np.zeros((2, 6, 7), np.int8)
    board[0, 5, 0] = board[0, 5, 1] = board[0, 5, 2] = 1
    board[1, 4, 0] = board[1, 4, 1] = board[1, 4, 2] = 1
    gs = az.Connect4GS(board, 0, 6)
    assert gs.current_player() == 0 and gs.current_turn() == 6 and gs.scores() is None
    assert np.array_equal(gs.valid_moves(), np.ones(7, np.uint8))
    assert np.array_equal(gs.canonicalized()[:2], board.astype(np.float32))
    twin = gs.copy()
    assert twin == gs
    gs.play_move(3)
    assert twin != gs
    assert np.array_equal(gs.scores(), np.array([1, 0, 0], np.float32))
    assert twin.scores() is None
    blob = pickle.dumps(gs)
    back = pickle.loads(blob)
    assert back == gs and back.to_bytes() == gs.to_bytes() and len(gs.to_bytes()) == 89
    assert str(gs).startswith("Current Player: 1\n") and str(gs).count("\n") == 8
    with pytest.raises(RuntimeError):
        az.Connect4GS(np.zeros((2, 6, 6), np.int8), 0, 0)
    full = az.Connect4GS()
    for _ in range(6):
        full.play_move(0)
    assert full.valid_moves()[0] == 0
    with pytest.raises(RuntimeError, match="Invalid move: You have a bug in your code."):   # connect4_gs.cc:48-58
        full.play_move(0)
    assert full.current_turn() == 6 and full.valid_moves()[0] == 0                          # the object is unchanged


def test_tawlbwrdd_object_walk_matches_oracle(az, oracle):
    rng = np.random.default_rng(8)
    gs = az.TawlbwrddGS(); og = oracle.Game(oracle.GAME_TAWLBWRDD)
    assert gs.num_symmetries() == 8 and gs.num_moves() == 2662 and gs.num_players() == 2
    assert gs.relative_values() is False and gs.num_variants() == 0 and gs.get_variant_id() == -1
    for ply in range(40):
        if gs.scores() is not None:
            break
        vm = gs.valid_moves()
        assert np.array_equal(vm, og.valid())
        assert np.array_equal(gs.canonicalized(), og.canonical())
        mv = int(rng.choice(np.flatnonzero(vm)))
        gs.play_move(mv); og.play(mv)
    assert gs.current_turn() == og.turn() and gs.current_player() == og.player()
    s = str(gs)
    assert s.startswith("Current Player:") and "@" in s


@pytest.mark.parametrize("name", ["TawlbwrddGS", "BrandubhGS", "OpenTaflGS"])
def test_tafl_objects_pickle_and_copy_round_trip(az, name):
    """py::pickle on the game classes (py_wrapper.cc:77-83): a pickled object comes back equal, with its repetition history
    (the object is its start position + move list, so the history travels with it); copies are independent."""
    import copy
    g = getattr(az, name)()
    rng = np.random.default_rng(1)
    for _ in range(9):
        legal = np.flatnonzero(g.valid_moves())
        g.play_move(int(rng.choice(legal)))
    back = pickle.loads(pickle.dumps(g))
    assert back == g and az.hash_game_state(back) == az.hash_game_state(g)
    assert np.array_equal(back.canonicalized(), g.canonicalized()) and np.array_equal(back.valid_moves(), g.valid_moves())
    assert back.current_turn() == g.current_turn() == 9
    c = copy.deepcopy(g)
    c.play_move(int(np.flatnonzero(c.valid_moves())[0]))
    assert c != g and g.current_turn() == 9


# ---- the reference's pickle contract (test_game_pickle.py:19-170) for the four device games ---------------------------------
def _play_first_valid(gs, n):
    for _ in range(n):
        if gs.scores() is not None:
            break
        idx = np.flatnonzero(np.asarray(gs.valid_moves()))
        if len(idx) == 0:
            break
        gs.play_move(int(idx[0]))


def _assert_state_equal(a, b):
    assert np.array_equal(np.asarray(a.canonicalized()), np.asarray(b.canonicalized()))
    assert a.current_player() == b.current_player() and a.current_turn() == b.current_turn()
    assert np.array_equal(np.asarray(a.valid_moves()), np.asarray(b.valid_moves()))
    sa, sb = a.scores(), b.scores()
    assert (sa is None) == (sb is None)
    if sa is not None:
        assert np.array_equal(np.asarray(sa), np.asarray(sb))


@pytest.mark.parametrize("name", ["Connect4GS", "BrandubhGS", "OpenTaflGS", "TawlbwrddGS"])
def test_reference_pickle_contract(az, name):
    make = getattr(az, name)
    for moves in (0, 4, 20):                                  # initial state, short play, long play
        gs = make(); _play_first_valid(gs, moves)
        _assert_state_equal(gs, pickle.loads(pickle.dumps(gs)))
    gs = make()                                               # chain: pickle, play, pickle, play ...
    for _ in range(3):
        _play_first_valid(gs, 3)
        gs = pickle.loads(pickle.dumps(gs))
    _assert_state_equal(gs, pickle.loads(pickle.dumps(gs)))
    orig = make(); _play_first_valid(orig, 5)                 # continued play is identical (repetition counters travel)
    back = pickle.loads(pickle.dumps(orig))
    for _ in range(5):
        if orig.scores() is not None:
            break
        m = int(np.flatnonzero(np.asarray(orig.valid_moves()))[0])
        orig.play_move(m); back.play_move(m)
        _assert_state_equal(orig, back)


# ---- the reference's connect4_gs_test.cc (:9-230) on the device Connect4GS object, case by case ---------------------------------
def test_reference_connect4_gs_cases(az):
    G = az.Connect4GS
    # Equals (:9-29): transpositions are equal states
    x, y = G(), G()
    assert x == y
    x.play_move(0); assert x != y
    y.play_move(0); assert x == y
    x, y = G(), G()
    for m in (0, 1, 2): x.play_move(m)
    for m in (2, 1, 0): y.play_move(m)
    assert x == y
    # Copy (:32-51)
    x = G(); y = x.copy(); assert x == y
    for m in (0, 1, 2): y.play_move(m)
    assert x != y
    z = y.copy(); assert y == z and x != z
    for m in (2, 1, 0): x.play_move(m)
    assert x == y and x == z
    # ValidMoves (:54-71)
    assert np.asarray(G().valid_moves()).tolist() == [1] * 7
    board = np.zeros((2, 6, 7), np.int8); board[0, 0, 3] = 1; board[1, 0, 5] = 1
    assert np.asarray(G(board, 0, 0).valid_moves()).tolist() == [1, 1, 1, 0, 1, 0, 1]
    # PlayMove (:75-101): stones stack from the bottom row up; a full column throws the reference's message
    board = np.zeros((2, 6, 7), np.int8)
    x = G(); assert x == G(board, 0, 0)
    i = 0
    for h in range(5, 0, -2):
        x.play_move(3); board[0, h, 3] = 1; i += 1; assert x == G(board, 1, i)
        x.play_move(3); board[1, h - 1, 3] = 1; i += 1; assert x == G(board, 0, i)
    with pytest.raises(RuntimeError, match="Invalid move: You have a bug in your code."):
        x.play_move(3)
    # WinState (:104-171): rows, columns, both diagonals, draw on a full top row
    assert G().scores() is None
    def score(b):
        s = G(b, 0, 0).scores()
        return None if s is None else np.asarray(s).tolist()
    b = np.zeros((2, 6, 7), np.int8)
    b[0, 3, 0:4] = 1; assert score(b) == [1, 0, 0]
    b[0, 3, 2] = 0; assert score(b) is None
    b[1, 1:5, 2] = 1; assert score(b) == [0, 1, 0]
    b[1, 2, 2] = 0; assert score(b) is None
    for k in (1, 2, 3, 4): b[0, k, k] = 1
    assert score(b) == [1, 0, 0]
    b[0, 2, 2] = 0; assert score(b) is None
    b[1, 1, 3] = b[1, 2, 2] = b[1, 3, 1] = b[1, 4, 0] = 1; assert score(b) == [0, 1, 0]
    b[1, 2, 2] = 0; assert score(b) is None
    b = np.zeros((2, 6, 7), np.int8)
    for w in range(7): b[w % 2, 0, w] = 1
    assert score(b) == [0, 0, 1]
    # Canonicalize (:174-227): stones are absolute planes 0/1, plane 2 + player is all ones
    x = G(); e = np.zeros((4, 6, 7), np.float32); e[2] = 1
    assert np.array_equal(np.asarray(x.canonicalized()), e)
    x.play_move(0); e = np.zeros((4, 6, 7), np.float32); e[0, 5, 0] = 1; e[3] = 1
    assert np.array_equal(np.asarray(x.canonicalized()), e)
    x.play_move(0); e = np.zeros((4, 6, 7), np.float32); e[0, 5, 0] = 1; e[1, 4, 0] = 1; e[2] = 1
    assert np.array_equal(np.asarray(x.canonicalized()), e)
