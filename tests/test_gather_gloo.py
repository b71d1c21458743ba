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


def _worker8(rank, world, port, q):
    """world size 8 (the node the driver scales to): ragged and zero row counts at Tawlbwrdd's row size (847 + 3 + 2662 floats =
    14 KB per row), then the statistics vector of bench.py (scores, resign scores, ten accumulators) with StarGambit's
    per-variant tables (games + scores per variant) in ONE all-reduce."""
    sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from alphazero import gather
    counts = [7, 0, 3, 0, 0, 11, 1, 5]
    n = counts[rank]
    canon = torch.full((n, 7, 11, 11), float(rank))
    v = torch.full((n, 3), float(rank) + 0.5)
    pi = torch.zeros((n, 2662))
    if n:
        pi[:, rank] = 1.0                       # a one-hot marker: row r of rank k keeps its place
    res = gather.gather_rows_to_rank0([canon, v, pi], rank, world)
    stats = torch.arange(3 + 3 + 10 + 4 * 4, dtype=torch.float64) * (rank + 1)
    dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    tmax = torch.tensor([float(rank)], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        owner = res[2].argmax(1).tolist()
        q.put((tuple(res[0].shape), tuple(res[2].shape), owner, res[0].mean((1, 2, 3)).tolist(), res[1][:, 0].tolist(), stats.tolist(), tmax.item()))
    else:
        assert res is None
    dist.destroy_process_group()


def test_gather_rows_and_statistics_world8():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    cshape, pshape, owner, cmean, v0, stats, tmax = out
    counts = [7, 0, 3, 0, 0, 11, 1, 5]
    expect_owner = [r for r, c in enumerate(counts) for _ in range(c)]
    assert cshape == (27, 7, 11, 11) and pshape == (27, 2662)
    assert owner == expect_owner                              # rank order, nothing lost, nothing from the padding
    assert cmean == [float(r) for r in expect_owner] and v0 == [r + 0.5 for r in expect_owner]
    assert stats == [36.0 * i for i in range(32)]             # sum over ranks of (rank + 1) = 36
    assert tmax == 7.0


def _worker_native_fail(rank, world, port, q, failing_rank):
    """bench.py's road choice when the native communicator cannot be made on one rank (rank 0 raised before the broadcast
    the others were waiting in, and the job hung): every rank must leave NativeGather with the same exception after the same
    collectives, agree on the fallback, and gather through torch.distributed."""
    sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["AZMI_COMM_TEST_FAIL_RANK"] = str(failing_rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from alphazero import gather
    ng, why = None, ""
    try:
        ng = gather.NativeGather(rank, world, 0)
    except RuntimeError as e:
        why = str(e)
    ok = torch.tensor([1.0 if ng is not None else 0.0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)                  # bench.py's agreement: same collective on every rank
    rows = torch.full((2 + rank, 3), float(rank))
    res = gather.gather_rows_to_rank0([rows], rank, world)     # the fallback road
    q.put((rank, ng is None, "cannot be made on every rank" in why, ok.item(), None if res is None else tuple(res[0].shape)))
    dist.destroy_process_group()


def test_native_gather_failure_is_agreed_on_by_every_rank():
    for failing_rank in (0, 1):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 33500 + (os.getpid() % 2000) + failing_rank
        procs = [ctx.Process(target=_worker_native_fail, args=(r, 2, port, q, failing_rank)) for r in range(2)]
        for p in procs:
            p.start()
        outs = sorted(q.get(timeout=120) for _ in range(2))
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        assert outs[0] == (0, True, True, 0.0, (5, 3)) and outs[1] == (1, True, True, 0.0, None), outs
