cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w scripts/c4_tile_timing.hip -o /tmp/c4t || exit 1
C4T_NO_CLOCK=1 /tmp/c4t 714 2904 2>&1 | grep -E "rows|full"
timeout -k 10 400 python -m pytest tests/test_gpu_leafnet.py -x -q 2>&1 | tail -3
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys
sys.path.insert(0, "alphazero-pybind11_amd")
import torch, alphazero as az
from alphazero import torch_net
dev = torch.device("cuda", 0)
for name, spec, shape, mf in (("tawlbwrdd", torch_net.tawlbwrdd_spec(), (7, 11, 11), 93.1e6), ("stargambit", torch_net.stargambit_spec(), (36, 13, 13), 135.3e6)):
    net = torch_net.random_init(spec, seed=0)
    hip = az.HipLeafNet(net, spec)
    P1 = spec.num_players + 1
    for n in (256, 512, 2048):
        x = torch.randint(0, 2, (n,) + shape, device=dev).float(); v = torch.empty(n, P1, device=dev); p = torch.empty(n, spec.num_moves, device=dev)
        for _ in range(3): hip.forward(x, v, p)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): hip.forward(x, v, p)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        print("%s n %4d: %7.1f us  %.3f PFLOP/s" % (name, n, us, n * mf / us / 1e9), flush=True)
PY
