"""the persistent net kernel ALONE on a pre-filled request ring (azmi_debug_pipe_net_bench): the launch rocprofv3 --pmc can
count.  Counter collection serialises dispatches, and an epoch of the pipeline needs its tree and net kernels co-resident, so
the HBM-side traffic of k_pipe_net is taken here: N positions drained by the bench's workgroup count, bytes per position =
counter / N (scripts/gpu_round6_profiles.sh)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd"))
import alphazero as az
from alphazero import torch_net
from alphazero._capi import lib, check
N = int(os.environ.get("N", 4608)); WGS = int(os.environ.get("WGS", 416)); REPS = int(os.environ.get("REPS", 20))
pp = az.PlayParams(); pp.games_to_play = pp.concurrent_games = pp.max_batch_size = 4096; pp.mcts_visits = [50, 50]; pp.model_groups = [0, 0]
spec = torch_net.connect4_spec()
hip = az.HipLeafNet(torch_net.random_init(spec, seed=0), spec)
pm = az.PlayManager(az.Connect4GS(), pp, seed=1)
ms = C.c_float()
check(lib.azmi_debug_pipe_net_bench(pm._h, hip._h, N, REPS, WGS, 0, C.byref(ms)))
print("k_pipe_net alone: n %d wgs %d reps %d: %.1f us per drain, %.2f M positions/s" % (N, WGS, REPS, ms.value * 1e3, N / ms.value / 1e3))
