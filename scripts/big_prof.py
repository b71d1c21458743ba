"""phase timing of the wide-game round kernel (a -DAZMI_BIG_PROF build: AZMI_HIPCC_EXTRA=-DAZMI_BIG_PROF python -c 'import __graft_entry__ as g; g.build()'):
Tawlbwrdd 2048 x 400 on 4 lock-step shards with the HIP net, ticks per phase summed over every wave / rounds"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd"))
import torch
import alphazero as az
from alphazero import torch_net, _capi
import bench
lib = _capi.lib if hasattr(_capi, "lib") else az.lib
GAME = os.environ.get("GAME", "tawlbwrdd")      # tawlbwrdd | stargambit | opentafl
S, K = int(os.environ.get("S", 1024 if GAME == "stargambit" else 2048)), 4
SIMS = int(os.environ.get("SIMS", 800 if GAME == "stargambit" else 400))
spec = {"tawlbwrdd": torch_net.tawlbwrdd_spec, "stargambit": torch_net.stargambit_spec, "opentafl": torch_net.opentafl_spec}[GAME]()
Game = {"tawlbwrdd": az.TawlbwrddGS, "stargambit": az.StarGambitUnifiedGS, "opentafl": az.OpenTaflGS}[GAME]
net = az.HipLeafNet(torch_net.random_init(spec, seed=0), spec)
pms, sts = [], []
for k in range(K):
    pp = bench.selfplay_params(az, S // K, SIMS, 1 << 30, cache=(200_000 // K if GAME == "stargambit" else 0), gumbel=bool(os.environ.get("GUMBEL")))
    pms.append(az.PlayManager(Game(), pp, seed=11 + k)); sts.append(torch.cuda.Stream())
sp = [s.cuda_stream for s in sts]
az.run_rounds(pms, net, 1024, sp); torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
lib.azmi_debug_big_prof.restype = C.c_int; lib.azmi_debug_big_prof.argtypes = [C.c_void_p]
lib.azmi_debug_big_prof(out)
import time
t0 = time.perf_counter()
az.run_rounds(pms, net, 1024, sp); torch.cuda.synchronize()
dt = time.perf_counter() - t0
lib.azmi_debug_big_prof(out)
n = max(1, out[8])
names = ["load", "process_result(+move)", "descent (rest: entry, Gumbel init, exit)", "move generation", "shuffle", "child records", "list entry, exit", "store"]
tot = 0.0
for i, nm in enumerate(names):
    us = out[i] / n / 100.0; tot += us
    print("%-24s %7.2f us per wave-round" % (nm, us))
for i, nm in ((12, "descent: children -> LDS"), (13, "descent: in-order seen sum"), (14, "descent: scores + arg-max"), (9, "planes + key"), (10, "descent: N / META of the chosen child"), (11, "descent: step_state"), (15, "expansion end -> planes start")):
    us = out[i] / n / 100.0; tot += us
    print("%-40s %7.2f us per wave-round" % (nm, us))
print("sum %.1f us; %d wave-rounds in %.2f s = %.2f M simulations/s" % (tot, n, dt, n / dt / 1e6))
