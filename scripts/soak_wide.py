"""Soak of the wide-game engine paths added late in round 1: Brandubh self-play on the bf16 MFMA net (zero-padded 32-channel
net, 7 boards per workgroup) and PLAYOUT seats on the three Tafl games.  Prints one line per run; any engine error raises."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alphazero-pybind11_amd"))
import numpy as np
import alphazero as az
from alphazero import selfplay, torch_net

t0 = time.time()
pp = az.PlayParams()
pp.games_to_play, pp.concurrent_games, pp.max_batch_size = 4096, 1024, 1024
pp.mcts_visits = [100, 100]
pp.model_groups = [0, 0]
pp.history_enabled = True
pp.cpuct, pp.epsilon = 1.25, 0.25
pp.playout_cap_randomization, pp.playout_cap_depth, pp.playout_cap_percent = True, 16, 0.75
net = torch_net.random_init(torch_net.brandubh_spec(), seed=4)
res, (c, v, p) = selfplay.self_play(az.BrandubhGS(), pp, az.HipLeafNet(net), engines=4, seed=1)
assert res.games == 4096 and abs(sum(res.win_rates) - 1) < 1e-6 and float((p.sum(1) - 1).abs().max()) < 1e-4
print("brandubh net self-play: %d games, %d samples, %d sims, %d evals, length %.1f, win rates %s, %.1f s" % (
    res.games, res.samples, res.simulations, res.leaf_evaluations, res.game_length, np.round(res.win_rates, 3).tolist(), time.time() - t0))

for Game, n in ((az.BrandubhGS, 256), (az.OpenTaflGS, 64), (az.TawlbwrddGS, 64)):
    t0 = time.time()
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.max_batch_size = n, n, n
    pp.mcts_visits = [30, 30]
    pp.eval_type = [az.EvalType.PLAYOUT, az.EvalType.PLAYOUT]
    pp.history_enabled = True
    res, (c, v, p) = selfplay.self_play(Game(), pp, None, engines=2, seed=2)
    assert res.games == n and abs(sum(res.win_rates) - 1) < 1e-6 and np.abs(p.sum(1) - 1).max() < 1e-4
    print("%s PLAYOUT seats: %d games, %d samples, length %.1f, win rates %s, %.1f s" % (
        Game.__name__, res.games, res.samples, res.game_length, np.round(res.win_rates, 3).tolist(), time.time() - t0))
