"""Task split = the only place the libcsc path shards.

Mirrors the archiver's task construction and dispatch for a single file:
  * `split_single_file`  -- csarc.cpp:532-543 (`-p N`: slices of max(esize/N, 1 MiB) + 4 bytes)
  * `dispatch_order`     -- csarc.cpp:355 (tasks sorted by size, descending; ties keep file order
                            for <= 16 tasks)
  * `assign`             -- task i of the dispatch order -> rank i mod world (one process per GPU)
  * `gather_results`     -- per-task {size, digest/bytes} to rank 0 in task-id order (= `csarc -t1`
                            layout).  Tasks are independent streams: no collective touches the data
                            path; torch.distributed only moves these small records.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def split_single_file(esize: int, split_count: int) -> List[Tuple[int, int]]:
    split_count = max(1, split_count)
    s = esize // split_count
    s = max(s, 1048576) + 4
    out, off = [], 0
    while off < esize:
        b = min(s, esize - off)
        out.append((off, b))
        off += b
    return out


def dispatch_order(tasks: Sequence[Tuple[int, int]]) -> List[int]:
    """task ids in the order compress_mt hands them out (size desc, stable)"""
    return sorted(range(len(tasks)), key=lambda i: -tasks[i][1])


def assign(tasks: Sequence[Tuple[int, int]], world: int) -> List[List[int]]:
    """rank -> list of task ids"""
    out = [[] for _ in range(world)]
    for k, tid in enumerate(dispatch_order(tasks)):
        out[k % world].append(tid)
    return out


def gather_results(local: dict, world: int, rank: int):
    """local: {task_id: record}.  Returns the merged dict on every rank (records are small)."""
    if world == 1:
        return dict(local)
    import torch.distributed as dist
    box = [None] * world
    dist.all_gather_object(box, local)
    merged = {}
    for part in box:
        merged.update(part)
    return merged
