"""CPU, world_size 2 (gloo): the N>1 sample-gather path of bench.py / alphazero.gather."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from alphazero import gather
    n = 5 if rank == 0 else 3
    canon = torch.full((n, 4, 6, 7), float(rank + 1))
    v = torch.arange(n * 3, dtype=torch.float32).reshape(n, 3) + 100 * rank
    pi = torch.full((n, 7), 1.0 / 7)
    res = gather.gather_rows_to_rank0([canon, v, pi], rank, world)
    empty = gather.gather_rows_to_rank0([canon[:0], v[:0], pi[:0]], rank, world)
    ragged = gather.gather_rows_to_rank0([canon[: (0 if rank == 0 else 2)]], rank, world)
    if rank == 0:
        q.put((res[0].shape, res[0][:5].mean().item(), res[0][5:].mean().item(), res[1][5:, 0].tolist(), empty[0].shape[0],
               ragged[0].shape[0], ragged[0].mean().item()))
    else:
        assert res is None and empty is None
    dist.destroy_process_group()


def test_gather_rows_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    shape, m0, m1, v1, n_empty, n_ragged, ragged_mean = out
    assert tuple(shape) == (8, 4, 6, 7) and m0 == 1.0 and m1 == 2.0
    assert v1 == [100.0, 103.0, 106.0]
    assert n_empty == 0 and n_ragged == 2 and ragged_mean == 2.0
