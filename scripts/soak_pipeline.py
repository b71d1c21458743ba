"""Soak of the asynchronous pipeline in the shapes VERDICT r5 item 7 names (profiles/r6_soak_*.txt): N engine iterations (one iteration
= one azmi_run_pipeline(_groups) call of E epochs) of
  SHAPE=net_random       two model groups, a net behind group 0 and RANDOM seats in group 1 (the RandPlayer baseline match), 256 slots
  SHAPE=gumbel_two_nets  Gumbel roots, two different nets behind two model groups (play_past), 512 slots
  SHAPE=slots_16384      the headline engine with 16384 concurrent games (one group, PUCT)
  SHAPE=headline         the bench's engine itself (4096 x 800, 128 M-entry cache), epochs of 256 simulations per slot
A pipeline error RAISES (no retry here): the run stops at the first one and prints it.  Credited freezes, lost requests and every
error are counted in the last line."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import alphazero as az
from alphazero import torch_net
import bench
shape = os.environ.get("SHAPE", "net_random")
N = int(os.environ.get("N", 1000))
spec = torch_net.connect4_spec()
net0 = az.HipLeafNet(torch_net.random_init(spec, seed=0), spec)
if shape == "net_random":
    S, sims, E, Q = 256, 200, 2, 32
    pp = bench.selfplay_params(az, S, sims, 1 << 30, cache=1 << 16)
    pp.model_groups, pp.seat_perms = [0, 1], [[0, 1], [1, 0]]
    pp.eval_type = [az.EvalType.NN, az.EvalType.RANDOM]
    nets = [net0, None]
elif shape == "gumbel_two_nets":
    S, sims, E, Q = 512, 200, 2, 32
    pp = bench.selfplay_params(az, S, sims, 1 << 30, cache=1 << 18, gumbel=True)
    pp.model_groups, pp.seat_perms = [0, 1], [[0, 1], [1, 0]]
    nets = [net0, az.HipLeafNet(torch_net.random_init(spec, seed=1), spec)]
elif shape == "headline":          # the bench's own engine: 4096 x 800, 128 M-entry cache
    S, sims, E, Q = 4096, 800, 4, 256
    pp = bench.selfplay_params(az, S, sims, 1 << 30, cache=128_000_000)
    nets = None
else:
    S, sims, E, Q = 16384, 800, 2, 64
    pp = bench.selfplay_params(az, S, sims, 1 << 30, cache=32_000_000)
    nets = None
pp.history_enabled = False
pm = az.PlayManager(az.Connect4GS(), pp, seed=20240601)
st = torch.cuda.Stream()
errors, t0, last = 0, time.perf_counter(), {}
for it in range(N):
    try:
        last = az.run_pipeline(pm, net0, E, S * Q, st.cuda_stream) if nets is None else az.run_pipeline_groups(pm, nets, E, S * Q, st.cuda_stream)
    except RuntimeError as e:
        errors += 1
        print("iteration %d: %s" % (it, str(e)[:600]), flush=True)
        break
    if it % 100 == 0:
        c = pm.counters()
        print("%s iteration %d: games %d sims %.1f M evals %.1f M freezes %d lost %d  T %s N %s  %.0f s" % (
            shape, it, pm.poll(st.cuda_stream)[0], c["sims"] / 1e6, c["evals"] / 1e6, last.get("freezes", 0), last.get("lost_total", 0), last.get("tree_wgs"), last.get("net_wgs"), time.perf_counter() - t0), flush=True)
c = pm.counters()
print("%s: %d iterations of %d epochs, %d slots: errors %d, freezes %d, lost requests %d; %d games, %.1f M simulations, %.1f M evaluations in %.0f s" % (
    shape, it + 1, E, S, errors, last.get("freezes", 0), last.get("lost_total", 0), pm.poll(st.cuda_stream)[0], c["sims"] / 1e6, c["evals"] / 1e6, time.perf_counter() - t0), flush=True)
sys.exit(1 if errors else 0)
