"""the net side ALONE on a pre-filled request ring (azmi_debug_pipe_net_bench): the conveyor (mode 3, `lines` lines, AZMI_CV_HEADS head
waves per line) beside the tile kernel (mode 0) -> us per drain, evaluations/s, fraction of the 2.5 PFLOP/s bf16 MFMA peak"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import alphazero as az
from alphazero import torch_net
from alphazero._capi import lib, check
pp = az.PlayParams(); pp.games_to_play = pp.concurrent_games = pp.max_batch_size = 4096; pp.mcts_visits = [50, 50]; pp.model_groups = [0, 0]
spec = torch_net.connect4_spec()
hip = az.HipLeafNet(torch_net.random_init(spec, seed=0), spec)
pm = az.PlayManager(az.Connect4GS(), pp, seed=1)
ms = C.c_float()
FLOP = 37.7e6
def run(mode, wgs, n, tag):
    check(lib.azmi_debug_pipe_net_bench(pm._h, hip._h, n, 4, wgs, mode, C.byref(ms)))
    print("%-22s n %5d: %8.1f us  %6.2f M evaluations/s  %.3f of MFMA peak" % (tag, n, ms.value * 1e3, n / ms.value / 1e3, n / ms.value * 1e3 * FLOP / 2.5e15), flush=True)
for n in (3072, 12288, 32768):
    run(0, 0, n, "tiles, 512 workgroups")
lines_list = [int(x) for x in os.environ.get("CV_LINES", "16,32,48,64,72").split(",")]
for lines in lines_list:
    for n in (12288, 32768):
        run(3, lines, n, "conveyor, %d lines" % lines)
