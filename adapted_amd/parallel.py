"""Multi-GPU layout of ``adapted detect``: one process per GPU, reads sharded in WHOLE
minibatches (the LLR path normalises per minibatch -- reference adapted/detect/normalize.py:
15-22 -- so results do not depend on the GPU count), no collective on the data path, and ONE
gather of the fixed-width result rows (544 B each) to the writer rank.  With the "nccl"
backend (= RCCL over xGMI on ROCm) the gather runs on device tensors; the same code runs on
"gloo"/CPU tensors (used by the world_size-2 CPU tests)."""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import numpy as np


def world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_minibatches(n_minibatches: int, world_size: int, rank: int) -> range:
    """Contiguous block of minibatch indices owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_minibatches, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def shard_reads(n_reads: int, minibatch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """[start, stop) read range of `rank`: whole minibatches, same membership as a 1-GPU run."""
    n_mb = (n_reads + minibatch - 1) // minibatch
    r = shard_minibatches(n_mb, world_size, rank)
    return min(r.start * minibatch, n_reads), min(r.stop * minibatch, n_reads)


def gather_rows(rows, dst: int = 0, group=None, always: bool = False):
    """Gather per-rank row blocks (numpy structured array or uint8 torch tensor [n, row_bytes])
    to `dst` in rank order.  Returns the concatenation on dst, None elsewhere.  A group of one returns its rows as they
    are unless `always` asks for the collective all the same (how the RCCL branch is exercised on a one-GPU box)."""
    import torch
    import torch.distributed as dist

    from . import lib as _lib

    _lib.assert_one_runtime()  # (rows of the library's device context go to torch.distributed's)

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not always):
        return rows
    ws, rk = dist.get_world_size(group), dist.get_rank(group)
    backend = dist.get_backend(group)
    as_numpy = isinstance(rows, np.ndarray)
    if as_numpy:
        dtype = rows.dtype
        # (explicit width: reshape(0, -1) of an empty block is an error, and a rank may well end up with no rows)
        t = torch.from_numpy(np.ascontiguousarray(rows).view(np.uint8).reshape(rows.shape[0], dtype.itemsize).copy())
    else:
        t = rows
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t = t.to(dev)
    width = t.shape[1]
    counts = torch.zeros(ws, dtype=torch.int64, device=dev)
    counts[rk] = t.shape[0]
    dist.all_reduce(counts, group=group)
    nmax = int(counts.max().item())
    pad = torch.zeros((nmax, width), dtype=torch.uint8, device=dev)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(ws)] if rk == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    more = None
    if as_numpy and "open_pores_more" in (dtype.names or ()):
        # open_pores lists longer than a row holds travel beside the fixed-width rows (rare, small): tokens of this
        # process's registry -> the lists -> tokens of the destination's registry
        from . import lib as _lib

        # (the sender's copies leave its registry: the destination registers them again under its own tokens)
        mine = [(int(i), _lib._OPEN_PORES_MORE.pop(int(rows[i]["open_pores_more"]))) for i in np.flatnonzero(rows["n_open_pores"] > _lib.MAX_OPEN_PORES)]
        more = [None] * ws if rk == dst else None
        dist.gather_object(mine, more, dst=dst, group=group)
    if rk != dst:
        return None
    out = torch.cat([bufs[r][: int(counts[r].item())] for r in range(ws)], dim=0)
    if as_numpy:
        res = out.cpu().numpy().reshape(-1).view(dtype)
        if more is not None:
            base = 0
            for r in range(ws):
                for i, arr in more[r]:
                    res[base + i]["open_pores_more"] = _lib.register_open_pores(arr)
                base += int(counts[r].item())
        return res
    return out
