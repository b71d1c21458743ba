"""Known-answer rule cases of the reference's opentafl_gs_test.cc (25 tests) and brandubh_gs_test.cc, as data.
Each case: board pieces, side to move, turn, optional move, and the assertions the reference test makes.
Used twice: to pin the oracle (tests/test_oracle_pinned.py, CPU) and to check the device rules kernels
(tests/test_gpu_tafl_family.py) through the same table."""
import numpy as np

K, D, A = 0, 1, 2          # layers: king, defender, attacker
ATK, DEF, DRAW = 0, 1, 2   # players / score entries
N = 11


def mv(fh, fw, height_move, new_loc, n=N):  # Mv(), opentafl_gs_test.cc:92-95
    return (fh * n + fw) * (2 * n) + (n + new_loc if height_move else new_loc)


def board(pieces, n=N):
    b = np.zeros((3, n, n), np.int8)
    for layer, h, w in pieces:
        b[layer, h, w] = 1
    return b


def _box():
    p = []
    for x in range(3, 8):
        p += [(A, 3, x), (A, 7, x), (A, x, 3), (A, x, 7)]
    return sorted(set(p))


# (name, pieces, player, turn, move or None, {"cells": [(layer,h,w,expected)], "valid": [(move, expected)], "score": entry or None/"none"})
OPENTAFL_CASES = [
    ("CaptureBetweenTwoEnemies", [(A, 3, 2), (D, 3, 3), (A, 3, 7), (K, 8, 8)], ATK, 10, mv(3, 7, False, 4),
     dict(cells=[(D, 3, 3, 0), (A, 3, 4, 1)])),
    ("CaptureAgainstCorner", [(D, 0, 1), (A, 5, 2), (K, 8, 8)], ATK, 10, mv(5, 2, True, 0), dict(cells=[(D, 0, 1, 0)])),
    ("CaptureAgainstEmptyThrone", [(D, 5, 4), (A, 5, 1), (K, 8, 8)], ATK, 10, mv(5, 1, False, 3), dict(cells=[(D, 5, 4, 0)])),
    ("ThroneNotHostileToDefenderWithKing", [(K, 5, 5), (D, 5, 4), (A, 5, 1)], ATK, 10, mv(5, 1, False, 3), dict(cells=[(D, 5, 4, 1)])),
    ("EdgeIsNotHostile", [(D, 1, 5), (A, 7, 5), (K, 8, 8)], ATK, 10, mv(7, 5, True, 2), dict(cells=[(D, 1, 5, 1)])),
    ("CaptureInTwoDirectionsAtOnce", [(D, 3, 2), (A, 3, 1), (D, 2, 3), (A, 1, 3), (A, 3, 7), (K, 8, 8)], ATK, 10, mv(3, 7, False, 3),
     dict(cells=[(D, 3, 2, 0), (D, 2, 3, 0)])),
    ("MovingBetweenTwoEnemiesIsSafe", [(A, 3, 2), (A, 3, 4), (D, 7, 3), (K, 8, 8)], DEF, 10, mv(7, 3, True, 3),
     dict(cells=[(D, 3, 3, 1), (A, 3, 2, 1), (A, 3, 4, 1)])),
    ("NonKingPassesThroughThroneButCannotLand", [(A, 5, 2), (K, 8, 8)], ATK, 10, None,
     dict(valid=[(mv(5, 2, False, 5), 0), (mv(5, 2, False, 6), 1)])),
    ("KingMayLandOnEmptyThrone", [(K, 5, 2)], DEF, 10, None, dict(valid=[(mv(5, 2, False, 5), 1)])),
    ("EncirclementIsAttackerWin", _box() + [(K, 5, 5)], DEF, 10, None, dict(score=ATK)),
    ("NotEncircledGameContinues", [p for p in _box() if p != (A, 3, 5)] + [(K, 5, 5)], DEF, 10, None, dict(score="none")),
    ("KingCapturedOnFourSides", [(K, 3, 3), (A, 2, 3), (A, 4, 3), (A, 3, 2), (A, 3, 7)], ATK, 10, mv(3, 7, False, 4), dict(score=ATK)),
    ("KingNotCapturedOnEdge", [(K, 0, 3), (A, 0, 2), (A, 1, 3), (A, 0, 7)], ATK, 10, mv(0, 7, False, 4), dict(cells=[(K, 0, 3, 1)])),
    ("RookMovementSlidesAndIsBlocked", [(A, 3, 0), (A, 3, 5), (K, 8, 8)], ATK, 10, None,
     dict(valid=[(mv(3, 0, False, 4), 1), (mv(3, 0, False, 5), 0), (mv(3, 0, False, 6), 0)])),
    ("DiagonalSandwichDoesNotCapture", [(D, 3, 3), (A, 2, 2), (A, 4, 7), (K, 8, 8)], ATK, 10, mv(4, 7, False, 4), dict(cells=[(D, 3, 3, 1)])),
    ("KingParticipatesInCapture", [(A, 3, 3), (D, 3, 2), (K, 3, 7)], DEF, 10, mv(3, 7, False, 4), dict(cells=[(A, 3, 3, 0)])),
    ("NonKingCannotLandOnCorner", [(A, 0, 3), (K, 8, 8)], ATK, 10, None, dict(valid=[(mv(0, 3, False, 0), 0), (mv(0, 3, False, 1), 1)])),
    ("ThroneHostileToAttackersWhenEmpty", [(A, 5, 4), (D, 5, 1), (K, 8, 8)], DEF, 10, mv(5, 1, False, 3), dict(cells=[(A, 5, 4, 0)])),
    ("ThroneHostileToAttackersWithKing", [(K, 5, 5), (A, 5, 4), (D, 5, 1)], DEF, 10, mv(5, 1, False, 3), dict(cells=[(A, 5, 4, 0)])),
    ("KingEscapesToCorner", [(K, 0, 3)], DEF, 10, mv(0, 3, False, 0), dict(score=DEF)),
    ("KingCapturedNextToThrone", [(K, 5, 4), (A, 4, 4), (A, 6, 4), (A, 5, 7)], ATK, 10, mv(5, 7, False, 3), dict(score=ATK)),
    ("LoneSurroundedEdgeKingLoses", [(K, 0, 5), (A, 0, 4), (A, 1, 5), (A, 0, 8)], ATK, 10, mv(0, 8, False, 6), dict(score=ATK)),
    ("NoLegalMovesLoses", [(A, 0, 1), (D, 0, 2), (D, 1, 1), (K, 5, 5)], ATK, 10, None, dict(score=DEF)),
    ("MaxTurnsIsADraw", [(K, 5, 5), (D, 5, 4), (A, 0, 5), (A, 10, 5)], ATK, 400, None, dict(score=DRAW)),
]

# RepetitionCount: from the start position, the side to move after the third repeat is credited (expected {1,0,0})
OPENTAFL_REPETITION = [(9 * 11 + 5) * 22 + 11 + 8, (7 * 11 + 5) * 22 + 4, (8 * 11 + 5) * 22 + 11 + 9, (7 * 11 + 4) * 22 + 5] * 2
BRANDUBH_REPETITION = [(3 * 7 + 5) * 14 + 7 + 4, (4 * 7 + 5) * 14 + 7 + 3] * 4   # brandubh_gs_test.cc:10-23


def check_case(game, case):
    """`game` exposes play(m), valid() -> u8[M], scores() -> f32[3] or None, canonical() -> [C,N,N]."""
    name, pieces, player, turn, move, exp = case
    if move is not None:
        game.play(move)
    c = game.canonical()
    for layer, h, w, want in exp.get("cells", []):
        assert c[layer, h, w] == want, (name, layer, h, w, c[layer, h, w])
    if "valid" in exp:
        v = game.valid()
        for m, want in exp["valid"]:
            assert v[m] == want, (name, m, v[m])
    if "score" in exp:
        s = game.scores()
        if exp["score"] == "none":
            assert s is None, (name, s)
        else:
            assert s is not None and s[exp["score"]] == 1.0 and s.sum() == 1.0, (name, s)
