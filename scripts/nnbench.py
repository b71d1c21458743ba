import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alphazero-pybind11_amd"))
import torch, alphazero as az
from alphazero import torch_net
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
game = sys.argv[3] if len(sys.argv) > 3 else "connect4"      # connect4 | tawlbwrdd | stargambit
spec = {"connect4": torch_net.connect4_spec, "tawlbwrdd": torch_net.tawlbwrdd_spec, "stargambit": torch_net.stargambit_spec}[game]()
mflop = {"connect4": 37.7, "tawlbwrdd": 93.1, "stargambit": 135.3}[game]
net = torch_net.random_init(spec, seed=0).cuda()
hn = az.HipLeafNet(net, spec)
x = (torch.rand(B, *spec.in_shape, device="cuda") < 0.3).float()
v = torch.empty(B, 3, device="cuda"); pi = torch.empty(B, spec.num_moves, device="cuda")
for _ in range(20): hn.forward(x, v, pi)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): hn.forward(x, v, pi)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
print(f"{game} B={B} {us:.1f} us/launch  {B*mflop*1e6/us/1e6:.1f} TFLOP/s")
