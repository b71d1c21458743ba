python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys
sys.path.insert(0, "alphazero-pybind11_amd")
import torch, alphazero as az
from alphazero import torch_net
spec = torch_net.connect4_spec(); net = torch_net.random_init(spec, seed=0)
dev = torch.device("cuda", 0)
for prec in ("bf16", "bf16x3"):
    hip = az.HipLeafNet(net, spec, precision=prec)
    for n in (384, 768, 1536, 3072, 6144):
        x = torch.randint(0, 2, (n, 4, 6, 7), device=dev).float(); v = torch.empty(n, 3, device=dev); p = torch.empty(n, 7, device=dev)
        for _ in range(3): hip.forward(x, v, p)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): hip.forward(x, v, p)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        print("%s n %4d: %7.1f us  %.2f Mevals/s  %.3f PFLOP/s (%.0f%s)" % (prec, n, us, n / us, n * 37.7e6 * (3 if prec == "bf16x3" else 1) / us / 1e9, n * 37.7e6 / us / 1e9 * 1000, " TF useful"), flush=True)
PY
AZMI_BENCH_SECONDARY=tier_1e5 timeout -k 10 500 python bench.py --worker --steps 3 --warmup 1 --preroll-factor 0.3 --profile-window --no-cpu-baseline > gpurun_out/x3_bench.json 2> gpurun_out/x3_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/x3_bench.json').read().strip().splitlines()[-1])
print(d['config'].get('tier_1e5'))
PY
