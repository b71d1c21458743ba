"""What does L2 traffic of the leaf-net tile's kind cost the tree kernel?  The pipeline's tree side ALONE (RANDOM seats, as
pipe_tree_only.py) while W workgroups of scripts/micro/l2_stream.hip read one L2-resident megabyte over and over on another stream.
Prints simulations/s and the stream's TB/s per setting (W = 0: nothing beside it)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import alphazero as az
import bench
lib = C.CDLL(os.path.join(ROOT, "scripts", "micro", "libl2stream.so"))
lib.l2_stream_launch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_void_p, C.c_void_p]
S, sims, Q, E = 4096, 800, 64, int(os.environ.get("E", 40))
pp = bench.selfplay_params(az, S, sims, 1 << 30, cache=0)
pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
os.environ["AZMI_PIPE_TREE_WGS"] = os.environ.get("TREE_WGS", "144")
pm = az.PlayManager(az.Connect4GS(), pp, seed=20240601, history_capacity=S * 42 * 4)
st, s2 = torch.cuda.Stream(), torch.cuda.Stream()
buf = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda"); out = torch.zeros(4, dtype=torch.int64, device="cuda")
for _ in range(3): az.run_pipeline(pm, None, E, S * Q, st.cuda_stream); pm.take_history_device(torch.device("cuda", 0))
for W in [int(x) for x in os.environ.get("WGS", "0,184,368,736,0").split(",")]:
    torch.cuda.synchronize(); out.zero_(); torch.cuda.synchronize()
    a = pm.counters()["sims"]; t0 = time.perf_counter()
    ms = float(os.environ.get("MS", 400))
    if W: lib.l2_stream_launch(buf.data_ptr(), buf.numel(), W, ms, out.data_ptr(), s2.cuda_stream)
    t_a = time.perf_counter(); n_calls = 0
    while time.perf_counter() - t_a < ms * 1e-3 * 0.9:
        az.run_pipeline(pm, None, 4, S * Q, st.cuda_stream); n_calls += 1
    c = pm.counters()["sims"]; t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("stream workgroups %4d: tree side %.1f M simulations/s over %d calls; the stream moved %.2f TB/s" % (W, (c - a) / (t1 - t0) / 1e6, n_calls, out[0].item() / (ms * 1e-3) / 1e12 if W else 0.0), flush=True)
    pm.take_history_device(torch.device("cuda", 0))
