"""throughput of the wide-game lock-step engine for A/B builds on ONE box (AZMI_LIB=<path to a libazmi.so>): Tawlbwrdd 2048 x 400,
4 shards, HIP net; prints simulations/s over ROUNDS rounds after a warm-up"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd"))
import torch
import alphazero as az
from alphazero import torch_net
import bench
S, K, R = int(os.environ.get("S", 2048)), 4, int(os.environ.get("ROUNDS", 2048))
game = os.environ.get("GAME", "tawlbwrdd")
spec = getattr(torch_net, game + "_spec")()
Game = {"tawlbwrdd": az.TawlbwrddGS, "brandubh": az.BrandubhGS, "opentafl": az.OpenTaflGS}[game]
net = az.HipLeafNet(torch_net.random_init(spec, seed=0), spec)
pms, sts = [], []
for k in range(K):
    pp = bench.selfplay_params(az, S // K, 400, 1 << 30, cache=0, gumbel=bool(os.environ.get("GUMBEL")))
    pms.append(az.PlayManager(Game(), pp, seed=11 + k)); sts.append(torch.cuda.Stream())
sp = [s.cuda_stream for s in sts]
az.run_rounds(pms, net, 1024, sp); torch.cuda.synchronize()
def sims(): return sum(pm.counters()["sims"] for pm in pms)
for rep in range(int(os.environ.get("REPS", 2))):
    a = sims(); t0 = time.perf_counter()
    az.run_rounds(pms, net, R, sp); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s %s: %.3f M simulations/s (%d rounds in %.2f s)" % (os.environ.get("AZMI_LIB", "libazmi.so").split("/")[-1], game, (sims() - a) / dt / 1e6, R, dt), flush=True)
