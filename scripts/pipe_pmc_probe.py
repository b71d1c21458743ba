"""what happens to one pipeline epoch under rocprofv3 --pmc (dispatches serialised): the tree kernel runs alone, no answer ever
arrives, its wall-clock cap ends it and the library reports the sticky timeout bit.  Evidence for why k_pipe_tree's counter
traffic is not collected (profiles/r3_pmc_pipeline_probe.txt)."""
import os, sys
os.environ.setdefault("AZMI_PIPE_CAP_MS", "40")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd"))
import torch
import alphazero as az
from alphazero import torch_net
import bench
pp = bench.selfplay_params(az, 512, 100, 1 << 20, cache=1 << 16)
spec = torch_net.connect4_spec()
hip = az.HipLeafNet(torch_net.random_init(spec, seed=0), spec)
pm = az.PlayManager(az.Connect4GS(), pp, seed=1)
try:
    st = az.run_pipeline(pm, hip, 2, 512 * 16)
    print("epochs ran:", st)
except RuntimeError as e:
    print("pipeline under --pmc:", str(e)[:400])
