"""quick StarGambit engine timing: S games x sims, K shards, HIP net; prints sims/s, evals/s, games finished"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alphazero-pybind11_amd"))
import torch
import alphazero as az
from alphazero import torch_net
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 800
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
cache = int(sys.argv[5]) if len(sys.argv) > 5 else 200000
rand = len(sys.argv) > 6 and sys.argv[6] == "random"
inline = int(os.environ.get("SG_INLINE", "0"))
dev = torch.device("cuda", 0)
spec = torch_net.stargambit_spec()
hip = az.HipLeafNet(torch_net.random_init(spec, seed=0), spec)
pms = []
for i in range(K):
    pp = az.PlayParams()
    pp.concurrent_games, pp.games_to_play, pp.max_batch_size = S // K, 1 << 30, S // K
    pp.mcts_visits = [sims, sims]
    pp.cpuct, pp.fpu_reduction = 1.25, 0.25
    pp.start_temp, pp.final_temp, pp.temp_decay_half_life_by_variant = 1.2, 0.2, [3.0, 4.0, 5.0, 8.0]
    pp.history_enabled, pp.epsilon, pp.mcts_root_temp, pp.root_fpu_zero, pp.shaped_dirichlet = True, 0.25, 1.25, True, True
    pp.policy_target_pruning = True
    pp.max_cache_size = cache // K
    pp.model_groups = [0, 0]
    if rand:
        pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
    pms.append(az.PlayManager(az.StarGambitUnifiedGS(), pp, seed=20240601 + i, history_capacity=(S // K) * 1200, max_inline=inline))
streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
sps = [st.cuda_stream for st in streams]
def tot():
    g = sum(pm.games_completed() for pm in pms); c = [pm.counters() for pm in pms]
    return g, sum(x["sims"] for x in c), sum(x["evals"] for x in c), sum(x["cache_hits"] for x in c)
az.run_rounds(pms, hip, 64, sps); torch.cuda.synchronize()
for rep in range(3):
    g0, s0, e0, h0 = tot()
    t0 = time.perf_counter()
    az.run_rounds(pms, hip, rounds, sps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    g1, s1, e1, h1 = tot()
    for pm in pms:
        pm.take_history_device()
    print(f"rep {rep}: {dt:.2f}s {rounds / dt:.0f} rounds/s  sims/s {(s1 - s0) / dt:.3e}  evals/s {(e1 - e0) / dt:.3e}  hits/s {(h1 - h0) / dt:.3e}  games {g1 - g0} ({(g1 - g0) / dt:.1f}/s)", flush=True)
