"""throughput probe of the gating shape (play_past, game_runner.py:2184-2332): Connect4 S x SIMS, two nets behind two model groups, seats
swapped by the permutations.  DRIVER=rounds: azmi_run_rounds_groups on 4 engine shards; DRIVER=pipeline: azmi_run_pipeline_groups on one
engine (AZMI_PIPE_GENERIC=1 forces the generic tree kernel).  GUMBEL=1: Gumbel roots (one group's worth: the same net twice)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import alphazero as az
from alphazero import torch_net
import bench
S = int(os.environ.get("S", 4096)); sims = int(os.environ.get("SIMS", 800)); cache = int(os.environ.get("CACHE", 32_000_000))
Q = int(os.environ.get("Q", 64)); E = int(os.environ.get("E", 100)); BLOCKS = int(os.environ.get("BLOCKS", 6)); PRE = float(os.environ.get("PRE", 1.0))
driver = os.environ.get("DRIVER", "pipeline")
K = 1 if driver == "pipeline" else 4
spec = torch_net.connect4_spec()
nets = [az.HipLeafNet(torch_net.random_init(spec, seed=0), spec), az.HipLeafNet(torch_net.random_init(spec, seed=1), spec)]
pms, sts = [], []
for k in range(K):
    pp = bench.selfplay_params(az, S // K, sims, 1 << 30, cache=cache // K, gumbel=bool(os.environ.get("GUMBEL")))
    pp.model_groups, pp.seat_perms = [0, 1], [[0, 1], [1, 0]]
    pp.history_enabled = False                      # (a gating match keeps scores, not samples)
    pms.append(az.PlayManager(az.Connect4GS(), pp, seed=20240601 + k))
    sts.append(torch.cuda.Stream())
def step():
    if driver == "pipeline":
        return az.run_pipeline_groups(pms[0], nets, E, S * Q, sts[0].cuda_stream)
    az.run_rounds_groups(pms, nets, 8 * E, [s.cuda_stream for s in sts])
    return {}
def tot():
    g = sm = ev = 0
    for pm, st in zip(pms, sts):
        c = pm.counters(); g += pm.poll(st.cuda_stream)[0]; sm += c["sims"]; ev += c["evals"]
    return g, sm, ev
t_pre = time.perf_counter()
while tot()[0] < PRE * S:
    s = step()
print(driver, "preroll %.1fs" % (time.perf_counter() - t_pre), s, flush=True)
a = tot(); t0 = time.perf_counter()
for b in range(BLOCKS):
    s = step()
    c = tot(); t1 = time.perf_counter(); dt = t1 - t0
    print("%s T %s N %s block %d: %.0f games/s %.1f Msims/s %.1f Mevals/s" % (driver, s.get("tree_wgs"), s.get("net_wgs"), b, (c[0] - a[0]) / dt, (c[1] - a[1]) / dt / 1e6, (c[2] - a[2]) / dt / 1e6), flush=True)
    a = c; t0 = t1
