"""debug driver of the pipeline: a few epochs, one at a time, counters printed"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import alphazero as az
from alphazero import torch_net
S = int(os.environ.get("S", 128)); sims = int(os.environ.get("SIMS", 100)); cache = int(os.environ.get("CACHE", 0))
pp = az.PlayParams()
pp.games_to_play, pp.concurrent_games, pp.max_batch_size = S, S, S
pp.mcts_visits = [sims, sims]; pp.model_groups = [0, 0]; pp.cpuct, pp.fpu_reduction = 1.25, 0.25
pp.history_enabled = True; pp.max_cache_size = cache
spec = torch_net.connect4_spec()
hip = az.HipLeafNet(torch_net.random_init(spec, seed=21), spec)
pm = az.PlayManager(az.Connect4GS(), pp, seed=1)
st = torch.cuda.Stream()
for e in range(int(os.environ.get("EPOCHS", 12))):
    t0 = time.perf_counter()
    try:
        s = az.run_pipeline(pm, hip, 1, S * int(os.environ.get("Q", 40)), st.cuda_stream)
    except RuntimeError as ex:
        print("epoch", e, "FAILED", ex); break
    print("epoch", e, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), s, pm.counters(), pm.poll(st.cuda_stream), flush=True)
