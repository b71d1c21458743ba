"""An EXPERIMENT since round 6 (not collected by `pytest tests/`): needs a library built with -DAZMI_WITH_CONVEYOR
(AZMI_HIPCC_EXTRA=-DAZMI_WITH_CONVEYOR python -c "import __graft_entry__ as g; g.build()"), then
    python -m pytest scripts/experiments/test_conveyor_experiment.py -m gpu -p no:cacheprovider --rootdir tests -c /dev/null
-m gpu: the conveyor (scripts/experiments/conveyor_c4.h: the Connect4 leaf net as lines of weight-stationary conv wavefronts + service waves)
answers every position bit for bit as the tile kernel of csrc/leafnet_c4.h does - the same MFMAs on the same operands in the same
order per accumulator, the same epilogue and head expressions (neural_net.py:233-263, 448-510, 800-823) - so every fixture and
parity tier of the tile carries over.  Both paths drain the same synthetic request ring (azmi_debug_pipe_net_answers)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _answers(az, pm, hip, n, seed, conveyor, lines=0):
    from alphazero._capi import lib, check
    out = np.zeros((n, 10), np.float32)
    seq = np.zeros(n, np.uint32)
    check(lib.azmi_debug_pipe_net_answers(pm._h, hip._h, n, seed, int(conveyor), lines, out.ctypes.data_as(C.c_void_p), seq.ctypes.data_as(C.c_void_p)))
    return out, seq


def _engine(az, S):
    pp = az.PlayParams()
    pp.games_to_play = pp.concurrent_games = pp.max_batch_size = S
    pp.mcts_visits = [16, 16]
    pp.model_groups = [0, 0]
    return az.PlayManager(az.Connect4GS(), pp, seed=5)


@pytest.mark.parametrize("depth,n,lines", [(2, 3, 1), (2, 48, 1), (6, 96, 1), (6, 1000, 7), (4, 333, 3), (6, 4096, 0)])
def test_conveyor_answers_bit_for_bit_like_the_tile_kernel(depth, n, lines):
    import alphazero as az
    from alphazero import torch_net
    spec = torch_net.connect4_spec(depth=depth)
    hip = az.HipLeafNet(torch_net.random_init(spec, seed=40 + depth), spec)
    pm = _engine(az, max(n, 64))
    want, wseq = _answers(az, pm, hip, n, 77, conveyor=False)
    got, gseq = _answers(az, pm, hip, n, 77, conveyor=True, lines=lines)
    assert np.array_equal(wseq, np.arange(1, n + 1, dtype=np.uint32)), "the tile kernel did not answer every position"
    assert np.array_equal(gseq, wseq), f"unanswered by the conveyor: {np.nonzero(gseq != wseq)[0][:10]}"
    assert np.all(np.isfinite(got))
    bad = np.nonzero(np.any(got.view(np.uint32) != want.view(np.uint32), axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} of {n} positions differ, first {bad[:5]}: {got[bad[0]]} vs {want[bad[0]]}"
    # and they are what the plain launch of the tile answers (the whole-batch kernel behind HipLeafNet.process)
    assert abs(float(got[:, :7].sum(1).mean()) - 1.0) < 1e-5 and abs(float(got[:, 7:].sum(1).mean()) - 1.0) < 1e-5


def test_the_pipeline_on_the_conveyor_plays_the_lockstep_engines_games(monkeypatch):
    """AZMI_PIPE_NET=conveyor: the asynchronous pipeline with the conveyor as its net side (three kernels side by side: tree, conv
    lines, service) plays, move for move and visit count for visit count, what the lock-step engine plays."""
    import alphazero as az
    from alphazero import torch_net
    from test_gpu_pipeline import _pipeline_games, _lockstep_games, _sorted_log
    from test_gpu_t3_nn_in_the_loop import _selfplay_params
    monkeypatch.setenv("AZMI_PIPE_NET", "conveyor")
    monkeypatch.setenv("AZMI_CV_LINES", "8")
    spec = torch_net.connect4_spec()
    hip = az.HipLeafNet(torch_net.random_init(spec, seed=21), spec)
    S, seed = 128, 9001
    pp = _selfplay_params(az, S, 100, cache=1 << 16)
    pa, (ra, ca), stats = _pipeline_games(az, pp, seed, hip, sims_per_epoch=S * 40)
    monkeypatch.delenv("AZMI_PIPE_NET")
    pb, (rb, cb) = _lockstep_games(az, pp, seed, hip)
    assert pa.games_completed() == pb.games_completed() == S
    ra, ca = _sorted_log(ra, ca)
    rb, cb = _sorted_log(rb, cb)
    assert np.array_equal(ra, rb), "moves / pcg32 positions differ between the conveyor pipeline and the lock-step engine"
    assert np.array_equal(ca, cb), "visit counts differ"
    assert stats["tiles"] > 0 and stats["net_wgs"] == 8
