"""quick throughput probe of the pipeline (not the bench): Connect4 S x SIMS, bench flags, prints per-block rates"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import alphazero as az
from alphazero import torch_net
import bench
S = int(os.environ.get("S", 4096)); sims = int(os.environ.get("SIMS", 800)); cache = int(os.environ.get("CACHE", 32_000_000))
Q = int(os.environ.get("Q", 64)); E = int(os.environ.get("E", 300)); BLOCKS = int(os.environ.get("BLOCKS", 12)); PRE = float(os.environ.get("PRE", 1.0))
pp = bench.selfplay_params(az, S, sims, 1 << 30, cache=cache)
spec = torch_net.connect4_spec()
hip = az.HipLeafNet(torch_net.random_init(spec, seed=0), spec, **({"precision": os.environ["PRECISION"]} if os.environ.get("PRECISION") else {}))
pm = az.PlayManager(az.Connect4GS(), pp, seed=20240601, history_capacity=S * 42 * 4, max_inline=int(os.environ.get('MAXI', 0)))
st = torch.cuda.Stream()
def tot():
    c = pm.counters(); return pm.poll(st.cuda_stream)[0], c["sims"], c["evals"], c["cache_hits"], c["cache_misses"]
t_pre = time.perf_counter()
while tot()[0] < PRE * S:
    s = az.run_pipeline(pm, hip, E, S * Q, st.cuda_stream); pm.take_history_device(torch.device("cuda", 0))
print("preroll %.1fs" % (time.perf_counter() - t_pre), s, flush=True)
a = tot(); t0 = time.perf_counter(); tiles0, boards0 = s["tiles"], s["tile_boards"]
for b in range(BLOCKS):
    s = az.run_pipeline(pm, hip, E, S * Q, st.cuda_stream); pm.take_history_device(torch.device("cuda", 0))
    c = tot(); t1 = time.perf_counter(); dt = t1 - t0
    print("T %d N %d " % (s["tree_wgs"], s["net_wgs"]) + "block %d: %.0f games/s %.1f Msims/s %.1f Mevals/s hit %.3f  ms/epoch %.2f  boards/tile %.2f late %d/%d us tiles_total %d" % (
        b, (c[0] - a[0]) / dt, (c[1] - a[1]) / dt / 1e6, (c[2] - a[2]) / dt / 1e6, (c[3] - a[3]) / max(1, c[3] - a[3] + c[4] - a[4]),
        dt / E * 1e3, (s["tile_boards"] - boards0) / max(1, s["tiles"] - tiles0), s["tree_latest_start_us"], s["net_latest_start_us"], s["tiles"]), flush=True)
    a = c; t0 = t1; tiles0, boards0 = s["tiles"], s["tile_boards"]
if os.environ.get("DUPES"):
    for q in (64, 256, 1024):
        az.run_pipeline(pm, hip, 4, S * q, st.cuda_stream); pm.take_history_device(torch.device("cuda", 0))
        n, dup, near = az.pipeline_log_duplicates(pm)
        print("insert log of one epoch of %d sims/slot: %d answers, %d (%.2f %%) asked for more than once, %d (%.2f %%) within 4096 entries of their twin" % (q, n, dup, 100.0 * dup / max(1, n), near, 100.0 * near / max(1, n)), flush=True)
