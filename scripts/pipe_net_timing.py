"""the persistent net kernel alone: drain N requests with W workgroups, tile mode M -> us per drain, evals/s; beside the
standalone launch (azmi_net_forward) of the same row count"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import alphazero as az
from alphazero import torch_net
from alphazero._capi import lib, check
pp = az.PlayParams(); pp.games_to_play = pp.concurrent_games = pp.max_batch_size = 4096; pp.mcts_visits = [50, 50]; pp.model_groups = [0, 0]
spec = torch_net.connect4_spec()
hip = az.HipLeafNet(torch_net.random_init(spec, seed=0), spec)
pm = az.PlayManager(az.Connect4GS(), pp, seed=1)
ms = C.c_float()
for mode in (1, 2, 0):
    for wgs in (128, 256, 384, 512):
        for n in (768, 2304, 4608, 8192):
            check(lib.azmi_debug_pipe_net_bench(pm._h, hip._h, n, 5, wgs, mode, C.byref(ms)))
            print("mode %d wgs %3d n %4d: %7.1f us  %.2f Mevals/s  %.1f us/tile-pass" % (mode, wgs, n, ms.value * 1e3, n / ms.value / 1e3,
                  ms.value * 1e3 / max(1.0, n / ((3 if mode == 2 else 6) * wgs))), flush=True)
dev = torch.device("cuda", 0)
for n in (768, 2304, 4608, 8192):
    x = torch.randint(0, 2, (n, 4, 6, 7), device=dev).float(); v = torch.empty(n, 3, device=dev); p = torch.empty(n, 7, device=dev)
    for _ in range(3): hip.forward(x, v, p)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): hip.forward(x, v, p)
    e1.record(); torch.cuda.synchronize()
    print("standalone launch n %4d: %7.1f us  %.2f Mevals/s" % (n, e0.elapsed_time(e1) * 100, n / (e0.elapsed_time(e1) * 100)), flush=True)
