"""the pipeline's tree kernel ALONE (every seat EvalType.RANDOM: no request ever leaves, no net kernel is launched): Connect4
S x SIMS with the bench flags.  What `rocprofv3 --pmc` can look at (one persistent kernel, no partner), and what the tree side's
latency chain costs without a net workgroup on its CU.  Prints simulations/s per block; AZMI_PIPE_PROF=1 adds the pass accounting."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import alphazero as az
import bench
S = int(os.environ.get("S", 4096)); sims = int(os.environ.get("SIMS", 800))
Q = int(os.environ.get("Q", 64)); E = int(os.environ.get("E", 100)); BLOCKS = int(os.environ.get("BLOCKS", 4))
pp = bench.selfplay_params(az, S, sims, 1 << 30, cache=0)
pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
pm = az.PlayManager(az.Connect4GS(), pp, seed=20240601, history_capacity=S * 42 * 4)
assert az.pipeline_supported(pm, None)
st = torch.cuda.Stream()
az.run_pipeline(pm, None, E, S * Q, st.cuda_stream); pm.take_history_device(torch.device("cuda", 0))
a = pm.counters()["sims"]; t0 = time.perf_counter()
for b in range(BLOCKS):
    s = az.run_pipeline(pm, None, E, S * Q, st.cuda_stream); pm.take_history_device(torch.device("cuda", 0))
    c = pm.counters()["sims"]; t1 = time.perf_counter()
    print("block %d: %.1f Msims/s  games %d  tree kernel %.2f ms/epoch (wall %.2f)" % (b, (c - a) / (t1 - t0) / 1e6, pm.poll(st.cuda_stream)[0], s["tree_kernel_us"] / E / 1e3, (t1 - t0) / E * 1e3), flush=True)
    a, t0 = c, t1
if os.environ.get("STATS_OUT"):
    import json
    n0 = pm.counters()["sims"]
    az.run_pipeline(pm, None, E, S * Q, st.cuda_stream)
    json.dump({"sims_per_epoch": (pm.counters()["sims"] - n0) / E, "S": S, "sims": sims, "quota": S * Q}, open(os.environ["STATS_OUT"], "w"))
