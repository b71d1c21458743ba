"""debug: second engine / playout cap in the pipeline, one epoch per call"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import alphazero as az
from alphazero import torch_net
import bench
S = int(os.environ.get("S", 4096)); sims = 800
spec = torch_net.connect4_spec()
hip = az.HipLeafNet(torch_net.random_init(spec, seed=0), spec)
st = torch.cuda.Stream()
for trial, cap in enumerate([int(x) for x in os.environ.get("CAPS", "0,1").split(",")]):
    pp = bench.selfplay_params(az, S, sims, 1 << 30, cache=32_000_000, playout_cap=bool(cap))
    pm = az.PlayManager(az.Connect4GS(), pp, seed=977, history_capacity=S * 42 * 4)
    print("engine", trial, "cap", cap, flush=True)
    N = int(os.environ.get("N", 1))
    if os.environ.get("POLL_FIRST"):
        print(" poll first", pm.poll(), pm.counters()["sims"], flush=True)
    for e in range(int(os.environ.get("EPOCHS", 8))):
        try:
            s = az.run_pipeline(pm, hip, N, S * 64, st.cuda_stream)
            if os.environ.get("TAKE"): pm.take_history_device(torch.device("cuda", 0))
        except RuntimeError as ex:
            print(" epoch", e, "FAILED", str(ex)[-400:]); break
        print(" epoch", e, {k: s[k] for k in ("tiles", "tile_boards", "last_epoch_sims", "tree_kernel_us", "net_kernel_us")}, pm.counters()["sims"], pm.poll(st.cuda_stream), flush=True)
    del pm
