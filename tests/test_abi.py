"""CPU: the C-ABI library loads and exports every symbol include/azmi.h declares; without a GPU the
compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "azmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(azmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    lib = C.CDLL(os.path.join(ROOT, "alphazero-pybind11_amd", "libazmi.so"))
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    from alphazero import _capi
    assert set(names) <= set(_capi.SYMBOLS) | {"azmi_debug_trace"}
    assert lib.azmi_abi_version() == 1


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import alphazero as az
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.mcts_visits = 1, 1, [10, 10]
    with pytest.raises(RuntimeError, match="no HIP device"):
        az.PlayManager(az.Connect4GS(), pp)
    with pytest.raises(RuntimeError, match="no HIP device"):
        az.rng_probe("pcg32", 1, 4)


def test_host_side_validation_matches_reference_messages():
    import alphazero as az
    pp = az.PlayParams()
    assert pp.cpuct == 2.0 and pp.tree_reuse is True and pp.playout_cap_depth == 25  # play_manager.h:60-154
    assert int(az.EvalType.NN) == 0 and int(az.EvalType.RANDOM) == 1 and int(az.EvalType.PLAYOUT) == 2
    assert az.Connect4GS.NUM_PLAYERS() == 2 and az.Connect4GS.NUM_MOVES() == 7
    assert az.Connect4GS.CANONICAL_SHAPE() == (4, 6, 7)
    assert az.tracy_is_enabled() is False


def test_bench_cpu_baseline_leg_runs_on_the_oracle():
    """bench.py's cpu_baseline leg (the only place outside tests/ and smoke() that may touch oracle/): a short multi-threaded
    sample returns a consistent record."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd"))
    import alphazero as az
    import bench
    r = bench.cpu_baseline(az, 100, 2.0, 64, None, 0, threads=2)      # no GPU here: the tree-only leg (EvalType.RANDOM) + configs[0]
    assert r["kind"] == "port" and r["unit"] == "games/s" and r["cores"] == 2
    assert r["value"] > 0 and r["sims_per_s"] > 0 and r["value"] == r["tree_only"]["games_per_s"]
    assert r["configs0"]["random_eval_games_per_s"] > 0 and "64 concurrent games, 100 sims" in r["configs0"]["workload"]


def _pm_params(az, **kw):
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games = 4, 2
    pp.mcts_visits = [10, 10]
    pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
    for k, v in kw.items():
        setattr(pp, k, v)
    return pp


@pytest.mark.parametrize("field,value", [
    ("seat_epsilon", [[0.25, 0.0], [0.0, 0.25]]),                 # play_manager_test.cc:74  SeatEpsilonWrongOuterDimThrows
    ("seat_epsilon", [[0.25, 0.0, 0.1]]),                          # :87  SeatEpsilonWrongInnerDimThrows
    ("seat_visits", [[10, 10], [10, 10], [10, 10]]),               # :100 SeatVisitsWrongDimThrows
    ("seat_gumbel_use_improved_policy", [[1, 0], [0, 1]]),         # :190 SeatGumbelUseImprovedPolicyWrongDimThrows
    ("seat_gumbel_use_improved_policy", [[1, 0, 1]]),              # :203 ...InnerDimThrows
    ("seat_resign_threshold", [[-0.9, -0.9], [-0.9, -0.9]]),       # :216 SeatResignThresholdWrongDimThrows
    ("seat_resign_consecutive", [[3, 3, 3]]),                      # :228 SeatResignConsecutiveWrongDimThrows
])
def test_reference_seat_matrix_dimension_errors(field, value):
    """the reference's own PlayManager constructor tests for mis-shaped per-seat matrices (play_manager_test.cc:74-238): the
    host side rejects them with the reference's message (play_manager.cc:57-68) before anything touches a device."""
    sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd"))
    import alphazero as az
    with pytest.raises(RuntimeError, match=f"{field} (outer dimension must match number of seat permutations|inner dimension must match number of players)"):
        az.PlayManager(az.Connect4GS(), _pm_params(az, **{field: value}))


def test_reference_playparams_binding_roundtrips():
    """test_cache.py:491-541 and test_temp_decay.py:19-24: list-valued PlayParams fields read back what was set, the removed
    `add_noise` field stays removed, epsilon defaults to 0."""
    sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd"))
    import alphazero as az
    p = az.PlayParams()
    for name, value in (("seat_visits", [[1, 480], [480, 1]]), ("seat_epsilon", [[0.25, 0.0], [0.0, 0.25]]),
                        ("seat_mcts_root_temp", [[1.25, 1.0], [1.0, 1.25]]), ("seat_root_fpu_zero", [[1, 0], [0, 1]]),
                        ("temp_decay_half_life_by_variant", [8.0, 12.0, 0.0, 24.0])):
        assert getattr(p, name) == []
        setattr(p, name, value)
        assert getattr(p, name) == value
    assert not hasattr(p, "add_noise")
    assert p.epsilon == 0.0
