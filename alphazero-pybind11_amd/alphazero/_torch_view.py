"""Zero-copy torch views of engine-owned HBM (plumbing only)."""
import numpy as np


class _Span:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {
            "shape": tuple(int(s) for s in shape), "typestr": typestr, "data": (int(ptr), False),
            "version": 2, "strides": None,
        }


_TYPESTR = {"torch.float32": "<f4", "torch.uint32": "<u4", "torch.int32": "<i4", "torch.uint8": "|u1"}


def device_tensor(ptr, shape, dtype, device):
    import torch
    if not ptr:
        raise RuntimeError("null device pointer")
    return torch.as_tensor(_Span(ptr, shape, _TYPESTR[str(dtype)]), device=device)
