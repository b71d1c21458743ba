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


# ---- the Tafl pickle BYTE LAYOUT (tawlbwrdd_gs.cc:10-37, brandubh_gs.cc:11-41, opentafl_gs.cc:13-40) -----------------
import struct


def _parse_tafl_image(data, N):
    bb = 3 * N * N
    board = np.frombuffer(data[:bb], np.int8).reshape(3, N, N)
    turn, max_turns, player, rep, n = struct.unpack_from("<HHbBI", data, bb)
    assert len(data) == bb + 10 + n * (bb + 2)
    entries = []
    for e in range(n):
        o = bb + 10 + e * (bb + 2)
        entries.append((data[o: o + bb], data[o + bb], data[o + bb + 1]))
    return board, turn, max_turns, player, rep, entries


@pytest.mark.parametrize("name,N,max_turns", [("TawlbwrddGS", 11, 400), ("BrandubhGS", 7, 150), ("OpenTaflGS", 11, 400)])
def test_tafl_to_bytes_is_the_reference_image(az, name, N, max_turns):
    G = getattr(az, name)
    g = G()
    img = g.to_bytes()                          # initial position: header only, repetition count 1, empty map
    board, turn, mt, player, rep, entries = _parse_tafl_image(img, N)
    assert (turn, mt, player, rep, entries) == (0, max_turns, 0, 1, [])
    assert np.array_equal(board, (g.canonicalized()[:3] != 0).astype(np.int8)) and board[0].sum() == 1
    # a quiet shuffle: attacker out and back, defender out and back -> the start position again, and again
    def first_move_of(gs, to_back=None):
        return int(np.flatnonzero(gs.valid_moves())[0])
    seq = []
    a1 = first_move_of(g); g.play_move(a1); seq.append(a1)
    d1 = first_move_of(g); g.play_move(d1); seq.append(d1)
    pieces = lambda gs: (gs.canonicalized()[:3] != 0).sum()
    assert pieces(g) == pieces(G())                                        # no capture so far
    board, turn, mt, player, rep, entries = _parse_tafl_image(g.to_bytes(), N)
    assert (turn, player, rep) == (2, 0, 1)
    # the map: the start position (interned by the first move) and the two positions reached, one each
    assert len(entries) == 3 and [e[2] for e in entries] == [1, 1, 1] and [e[1] for e in entries] == [0, 1, 0]
    assert entries[0][0] == (G().canonicalized()[:3] != 0).astype(np.int8).tobytes()
    assert entries[2][0] == board.tobytes()
    # round trip through the image keeps the bytes and the object
    back = G.from_bytes(g.to_bytes())
    assert back.to_bytes() == g.to_bytes() and back == g
    assert pickle.loads(pickle.dumps(g)).to_bytes() == g.to_bytes()


@pytest.mark.parametrize("name,N", [("BrandubhGS", 7), ("TawlbwrddGS", 11)])
def test_a_reference_image_with_a_repetition_map_loads_and_counts_on(az, name, N):
    """an image as the reference would have written it mid-game (hand-built per the layout): the repetition map travels, so
    re-reaching a recorded position raises its count, and the third occurrence ends the game exactly as for the object
    that played the whole record itself (tawlbwrdd_gs.cc:350-364)."""
    G = getattr(az, name)
    g = G()
    # a move pair that can be undone: piece from a to b, later b to a (a rook slide along an open line is reversible)
    def reverse(m):                        # move index = (h*N + w) * 2N + (column move ? N + new_h : new_w)
        frm, tgt = divmod(m, 2 * N)
        h, w = divmod(frm, N)
        nh, nw = (tgt - N, w) if tgt >= N else (h, tgt)
        return (nh * N + nw) * 2 * N + (N + h if nw == w else w)
    pieces = lambda gs: int((gs.canonicalized()[:3] != 0).sum())
    cycle = None
    for m1 in np.flatnonzero(g.valid_moves())[:12]:
        a = g.copy(); a.play_move(int(m1))
        for m2 in np.flatnonzero(a.valid_moves())[:12]:
            b = a.copy(); b.play_move(int(m2))
            if pieces(b) != pieces(g) or not b.valid_moves()[reverse(int(m1))]:
                continue
            c = b.copy(); c.play_move(reverse(int(m1)))
            if pieces(c) == pieces(g) and c.valid_moves()[reverse(int(m2))]:
                d = c.copy(); d.play_move(reverse(int(m2)))
                if pieces(d) == pieces(g) and np.array_equal(d.canonicalized()[:3], g.canonicalized()[:3]):
                    cycle = [int(m1), int(m2), reverse(int(m1)), reverse(int(m2))]
                    break
        if cycle:
            break
    assert cycle is not None
    for m in cycle:
        assert g.valid_moves()[m] == 1
        g.play_move(m)
    board, turn, mt, player, rep, entries = _parse_tafl_image(g.to_bytes(), N)
    assert turn == 4 and rep == 2 and g.scores() is None        # the start position for the second time
    assert sorted(e[2] for e in entries) == [1, 1, 1, 2]
    # hand-build the same image and load it
    img = board.tobytes() + struct.pack("<HHbBI", 4, mt, 0, 2, len(entries)) + b"".join(e[0] + bytes([e[1], e[2]]) for e in entries)
    assert img == g.to_bytes()
    h = G.from_bytes(img)
    assert h == g and np.array_equal(h.canonicalized(), g.canonicalized())
    for m in cycle:                                             # once more round: third occurrence
        g.play_move(m); h.play_move(m)
        assert np.array_equal(h.canonicalized(), g.canonicalized()) and h.current_turn() == g.current_turn()
    sg, sh = g.scores(), h.scores()
    assert sg is not None and np.array_equal(sg, sh) and np.asarray(sg).tolist() == [1.0, 0.0, 0.0]   # side to move wins
    assert h.to_bytes() == g.to_bytes()
    # a capture clears the map (tawlbwrdd_gs.cc:286-318): entry lists never outlive it -> covered by the random-play round trips
    with pytest.raises(RuntimeError, match="data too short"):
        G.from_bytes(img[: 3 * N * N + 4])
    with pytest.raises(RuntimeError, match="repetition entry count mismatch"):
        G.from_bytes(img[:-1])


def test_tafl_images_feed_the_search_and_the_rollouts(az):
    """a position loaded from an image is a first-class start position: MCTS.find_leaf and playout_eval accept it"""
    G = az.BrandubhGS
    g = G()
    for _ in range(6):
        g.play_move(int(np.flatnonzero(g.valid_moves())[0]))
    h = G.from_bytes(g.to_bytes())
    v1, p1 = az.playout_eval(g, seed=7)
    v2, p2 = az.playout_eval(h, seed=7)
    assert np.array_equal(v1, v2) and np.array_equal(p1, p2)
    t = az.MCTS(1.25, 2, G.NUM_MOVES(), game=G, seed=3)
    leaf = t.find_leaf(h)
    assert leaf.current_turn() >= h.current_turn()
