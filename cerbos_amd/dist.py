"""Multi-GPU plumbing: one process per GPU, independent request shards, and exactly one
collective - the one-time broadcast of the lowered policy image (RCCL over xGMI on the GPU
box, gloo in the CPU tests).  Requests are independent (internal/engine/engine.go:296-304),
so there is no data-path collective."""
from __future__ import annotations

import numpy as np


def broadcast_image(blob: bytes | None, src: int = 0, device: str = "cpu"):
    """Rank `src` passes the lowered image, the others pass None.  Returns a uint8 tensor on
    `device` holding the image on every rank (two broadcasts: length, then payload)."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank()
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    img = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    if rank == src:
        img.copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
    dist.broadcast(img, src=src)
    return img


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, order-preserving split (like outputs[wo.index], engine.go:332)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def image_checksum(img) -> int:
    a = np.frombuffer(bytes(img.cpu().numpy().tobytes()), dtype=np.uint8)
    return int(np.bitwise_xor.reduce(a.astype(np.uint64) * (np.arange(a.size, dtype=np.uint64) % 251 + 1)))
