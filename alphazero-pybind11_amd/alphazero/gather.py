"""The path's one exchange step: finished self-play samples go to rank 0.

Self-play shards by game slot with no communication during search; once per drain each rank
contributes its new PlayHistory rows.  Row counts are exchanged with one all_gather of a scalar,
then the rows travel with a padded gather (rank 0 receives, everyone else sends) — on GPUs this is
RCCL over xGMI (`backend="nccl"`), in the CPU tests it is gloo.  Messages are small
(Connect4: 712 B/row), so no ring tuning is involved.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist


def gather_rows_to_rank0(parts, rank, world, group=None):
    """parts: list of tensors with the same leading dimension n_local (may be 0).
    Returns on rank 0 a list of concatenated tensors (rank order), elsewhere None."""
    dev = parts[0].device
    n_local = parts[0].shape[0]
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([n_local], dtype=torch.int64, device=dev), group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    if n_max == 0:
        return [p[:0].clone() for p in parts] if rank == 0 else None
    out = []
    for p in parts:
        padded = torch.zeros((n_max,) + tuple(p.shape[1:]), dtype=p.dtype, device=dev)
        padded[:n_local] = p
        if rank == 0:
            bufs = [torch.empty_like(padded) for _ in range(world)]
            dist.gather(padded, bufs, dst=0, group=group)
            out.append(torch.cat([b[:c] for b, c in zip(bufs, counts)], 0))
        else:
            dist.gather(padded, None, dst=0, group=group)
    return out if rank == 0 else None


def gather_history_to_rank0(pm, dev, rank, world):
    """Gathers the rows `pm` has finished since the previous call. Returns total rows on rank 0."""
    c, v, p, meta = pm.take_history_device(dev)      # the unread rows of the engine's ring, released once copied out
    parts = [c, v, p]
    res = gather_rows_to_rank0(parts, rank, world)
    if rank == 0:
        pm._gathered = getattr(pm, "_gathered", [])
        pm._gathered.append(res)
        return int(res[0].shape[0])
    return 0


class NativeGather:
    """The same exchange behind the C ABI (csrc/gather.hip: azmi_comm_* / azmi_gather_counts / azmi_gather_rows over librccl, no
    torch collective in the data path): one ncclAllGather of the row counts, then the rows UNPADDED to rank 0 - grouped
    ncclSend / ncclRecv per array, rank 0's own rows by a device copy - so rank 0's extra memory is exactly the gathered rows
    (gather_rows_to_rank0 pads every rank to the largest count).  The 128-byte RCCL id travels over the launcher's own process
    group (one broadcast, any backend).  GPU only; the gloo tests keep using gather_rows_to_rank0."""

    def __init__(self, rank, world, device, group=None):
        from ._capi import lib, check
        self._lib, self._check = lib, check
        self.rank, self.world = int(rank), int(world)
        dev = torch.device("cuda", int(device))
        # Every rank runs the SAME sequence of collectives whatever fails where (rank 0 used to raise before the broadcast
        # while the others waited in it): (1) rank 0 broadcasts {status byte | 128-byte id} - status 0 and a zeroed id when it could not
        # make one; (2) a MIN all-reduce of "librccl loads here AND rank 0's id is good"; only when every rank said yes does any of them
        # enter ncclCommInitRank (a rank missing there would strand the others inside it).  A failure raises on EVERY rank.
        mine_ok, why = 1, ""
        try:
            if os.environ.get("AZMI_COMM_TEST_FAIL_RANK") == str(self.rank):      # test hook (tests/test_gather_gloo.py): this rank has no librccl
                raise RuntimeError("librccl not found (forced by AZMI_COMM_TEST_FAIL_RANK)")
            check(lib.azmi_comm_available())
        except Exception as e:          # noqa: BLE001 (carried into the agreement below, then raised on every rank)
            mine_ok, why = 0, str(e)[:160]
        msg = torch.zeros(129, dtype=torch.uint8)
        if self.rank == 0 and mine_ok:
            buf = (C.c_uint8 * 128)()
            try:
                check(lib.azmi_comm_unique_id(buf))
                msg = torch.tensor([1] + list(buf), dtype=torch.uint8)
            except Exception as e:      # noqa: BLE001
                mine_ok, why = 0, str(e)[:160]
        if self.world > 1:
            on_dev = dist.get_backend(group) == "nccl"
            t = msg.to(dev) if on_dev else msg
            dist.broadcast(t, src=0, group=group)
            msg = t.cpu()
            ok = torch.tensor([float(mine_ok and int(msg[0]) == 1)], device=dev if on_dev else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            all_ok = bool(ok.item() > 0.5)
        else:
            all_ok = bool(mine_ok and int(msg[0]) == 1)
        if not all_ok:
            raise RuntimeError("NativeGather: the RCCL communicator cannot be made on every rank" + (f" (rank {self.rank}: {why})" if why else ""))
        idb = (C.c_uint8 * 128)(*msg[1:].tolist())
        h = C.c_void_p()
        check(lib.azmi_comm_create(idb, self.rank, self.world, int(device), C.byref(h)))
        self._h, self._dev = h, dev

    def __del__(self):
        if getattr(self, "_h", None) and self._lib is not None:
            self._lib.azmi_comm_destroy(self._h)
            self._h = None

    def gather_rows_to_rank0(self, parts, stream=None):
        """parts: CONTIGUOUS device tensors with the same leading dimension (may be 0).  Rank 0: the list of gathered tensors (rank
        order), elsewhere None.  Returns after the copies are enqueued on `stream` (default: torch's current stream)."""
        lib, check = self._lib, self._check
        st = C.c_void_p(stream if stream is not None else torch.cuda.current_stream(self._dev).cuda_stream)
        n_local = int(parts[0].shape[0])
        counts = (C.c_uint64 * self.world)()
        check(lib.azmi_gather_counts(self._h, n_local, counts, st))
        total = sum(int(c) for c in counts)
        parts = [p.contiguous() for p in parts]
        k = len(parts)
        row_bytes = (C.c_uint64 * k)(*[p.element_size() * (p.numel() // max(1, p.shape[0])) if p.shape[0] else p.element_size() * int(torch.tensor(p.shape[1:]).prod()) for p in parts])
        src = (C.c_void_p * k)(*[p.data_ptr() for p in parts])
        out = None
        dst = (C.c_void_p * k)()
        if self.rank == 0:
            out = [torch.empty((total,) + tuple(p.shape[1:]), dtype=p.dtype, device=self._dev) for p in parts]
            dst = (C.c_void_p * k)(*[o.data_ptr() for o in out])
        if total:
            check(lib.azmi_gather_rows(self._h, src, row_bytes, k, counts, dst, st))
        self.last_counts = [int(c) for c in counts]
        return out
