"""Cell-range sharding across GPUs.

Cells are independent (every per-cell structure is cleared between cells in the reference,
src/quant.rs:880, 937, 967, 1313-1321), so the quant path shards with no data-path collective:
rank r takes a contiguous range of chunks, balanced by BYTES (a collated file is ordered
largest-cells-first, src/collate.rs:272-274), quantifies it on its own GPU, and rank 0 concatenates
the CSR shards in cell order (host-side gather).
"""
from __future__ import annotations

import numpy as np

from .afquant import QuantResult


def shard_ranges(chunk_nbytes, world: int):
    """Greedy prefix split into `world` contiguous ranges of about equal bytes.
    Returns [(c0, c1)] * world (empty ranges allowed when there are fewer cells than ranks)."""
    nb = np.asarray(chunk_nbytes, dtype=np.float64)
    n = len(nb)
    csum = np.concatenate(([0.0], np.cumsum(nb)))
    total = csum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        c = int(np.searchsorted(csum, target, side="left"))
        # choose the boundary closer to the target
        if c > 0 and c <= n and abs(csum[c - 1] - target) <= abs(csum[min(c, n)] - target):
            c -= 1
        cuts.append(min(max(c, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def concat_results(parts) -> QuantResult:
    """Concatenate per-rank QuantResults (already in cell order) into one."""
    parts = [p for p in parts if p is not None and p.n_cells > 0]
    if not parts:
        z = np.zeros(0)
        return QuantResult(0, np.zeros(1, np.uint64), z.astype(np.uint32), z.astype(np.float32), z.astype(np.uint64),
                           z.astype(np.uint32), z.astype(np.uint8))
    ptr = [np.zeros(1, np.uint64)]
    base = np.uint64(0)
    for p in parts:
        ptr.append(p.cell_ptr[1:] + base)
        base = base + p.cell_ptr[-1]
    cat = lambda f: np.concatenate([getattr(p, f) for p in parts])
    return QuantResult(parts[0].first_cell_index, np.concatenate(ptr), cat("gene"), cat("val"), cat("bc"),
                       cat("nrec"), cat("flags"))


def gather_results(local: QuantResult | None, dist, dst: int = 0):
    """Host-side gather of the shards on `dst` (the only communication the path has)."""
    world = dist.get_world_size()
    payload = None if local is None else (local.first_cell_index, np.asarray(local.cell_ptr), np.asarray(local.gene).copy(),
                                          np.asarray(local.val).copy(), local.bc, local.nrec, local.flags)
    out = [None] * world if dist.get_rank() == dst else None
    dist.gather_object(payload, out, dst=dst)
    if dist.get_rank() != dst:
        return None
    return concat_results([None if p is None else QuantResult(*p) for p in out])
