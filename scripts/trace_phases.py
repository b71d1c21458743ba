"""Phase timing of the round kernel for one slot (debug hook AZMI_TRACE_SLOT / AZMI_TRACE_AFTER): prints, per phase of
k_round, the mean / p50 / p90 time in microseconds over the traced rounds of a steady-state bench-like run.
The hook itself costs the traced wave about a microsecond per mark (two loads, three stores), which lands in the NEXT
phase: use the numbers to compare phases and versions, and the rocprof kernel time for absolute durations."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("AZMI_TRACE_SLOT", "17")
os.environ.setdefault("AZMI_TRACE_AFTER", "30000")
import torch
import alphazero as az
from alphazero import torch_net, _capi
import bench

K, Se = 4, int(os.environ.get("TRACE_SLOTS_PER_SHARD", "1024"))
pms, streams = [], []
for i in range(K):
    pp = bench.selfplay_params(az, Se, 800, Se * 16, cache=8_000_000)
    pms.append(az.PlayManager(az.Connect4GS(), pp, seed=20240601 + 104729 * i, max_inline=0))
    streams.append(torch.cuda.Stream())
hip = az.HipLeafNet(torch_net.random_init(torch_net.connect4_spec(), seed=0), torch_net.connect4_spec())
sps = [s.cuda_stream for s in streams]
az.run_rounds(pms, hip, int(os.environ.get('TRACE_ROUNDS', '33000')), sps)
torch.cuda.synchronize()
lib = _capi.lib
lib.azmi_debug_trace.restype = C.c_int
lib.azmi_debug_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
buf = np.zeros((1 << 16, 2), np.uint64); n = C.c_uint32()
assert lib.azmi_debug_trace(pms[0]._h, buf.ctypes.data, 1 << 16, C.byref(n)) == 0
ev = buf[: n.value]
tags = (ev[:, 0] & 0xFF).astype(int); arg = (ev[:, 0] >> 8).astype(int); clk = ev[:, 1].astype(np.int64)
names = {100: "start", 101: "load", 102: "process_result", 103: "make_move", 104: "find_leaf", 105: "key+cache probe", 106: "emit/store/end", 107: "game end", 108: "  descent level", 109: "  terminal+expand"}
dur = {}
rounds = []
prev = None
for t, a, c in zip(tags, arg, clk):
    if t < 100:
        continue
    if t == 100:
        prev = c; r0 = c
        continue
    if prev is None:
        continue
    dur.setdefault(t, []).append((c - prev) / 100.0)     # 100 MHz -> us
    prev = c
    if t in (106, 107):
        rounds.append((c - r0) / 100.0)
print("events", n.value, "rounds traced", len(rounds))
for t in sorted(dur):
    d = np.array(dur[t])
    print(f"{names[t]:>18}: n={len(d):6d} mean {d.mean():7.2f} us  p50 {np.percentile(d, 50):7.2f}  p90 {np.percentile(d, 90):7.2f}  max {d.max():7.2f}")
r = np.array(rounds)
print(f"{'slot total/round':>18}: mean {r.mean():7.2f} us  p50 {np.percentile(r, 50):7.2f}  p90 {np.percentile(r, 90):7.2f}  max {r.max():7.2f}")
print("cache hit fraction of probes", arg[tags == 105].mean() if (tags == 105).any() else None)
