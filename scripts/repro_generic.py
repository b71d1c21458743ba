"""bounded reproduction loop for an intermittent pipeline error on the generic tree kernel (AZMI_PIPE_GENERIC=1): the failing test's
configuration, N engines one after the other; stops at the first error and prints it"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import alphazero as az
from alphazero import torch_net
from test_gpu_t3_nn_in_the_loop import _selfplay_params
if os.environ.get("FAST") != "1": os.environ["AZMI_PIPE_GENERIC"] = "1"      # FAST=1: the default tree kernel (its Gumbel build here) instead
spec = torch_net.connect4_spec()
net = az.HipLeafNet(torch_net.random_init(spec, seed=51), spec)
N = int(os.environ.get("N", 60))
for it in range(N):
    S = 64
    pp = _selfplay_params(az, S, 48, cache=1 << 14)
    pp.gumbel_enabled, pp.gumbel_m, pp.gumbel_full = True, 8, True
    pm = az.PlayManager(az.Connect4GS(), pp, seed=949 + it, log_moves=True)
    st = torch.cuda.Stream()
    n = 0
    stats = None
    try:
        while pm.remaining_games() > 0 and n < 16000:
            stats = az.run_pipeline_groups(pm, [net], int(os.environ.get("E", 4)), S * 16, st.cuda_stream); n += 1
            if pm.poll(st.cuda_stream)[1] == 0: break
    except RuntimeError as e:
        print("iteration %d call %d: %s" % (it, n, str(e)[:900]), flush=True)
        break
    torch.cuda.synchronize()
    if stats and stats.get("freezes"): print("iteration %d: %d freeze(s) credited (wavefronts that stood still > 2 ms), %d calls, no error" % (it, stats["freezes"], n), flush=True)
    if it % 50 == 0: print("iteration", it, "ok", pm.games_completed(), flush=True)
print("done")
