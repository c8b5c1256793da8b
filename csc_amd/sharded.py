"""csc_amd.sharded -- `csarc a` over the GPUs of one node, one process per GPU.

The libcsc path shards only at the archiver's task boundary (csarc.cpp:532-557): tasks are independent
streams with private dictionaries.  The reference runs them on up to 8 worker threads that take the
next task of the size-sorted list (csarc.cpp:348-355) and hand their archive blocks to one writer.
Here a worker is a rank with its own MI355X:

  1. every rank plans the same tasks from the same file names (C++: CSAMI_AddShardEncode) and encodes
     tasks `rank, rank + world, ...` of the dispatch order on its GPU -- all of them at once, one
     workgroup per task stream.  No communication;
  2. the ONE exchange on this path: the finished task streams (with their block tables and fragment
     checksums, one opaque blob per rank) travel to rank 0.  Lengths by `all_gather`, payload by grouped
     point-to-point send/recv -- over RCCL/xGMI the blobs are device tensors, so rank 0 receives on all its
     links at once (xGMI is point-to-point: 7 peers, 7 links); over gloo (CPU tests) they are host tensors;
  3. rank 0 appends the tasks in task-id order and packs the index (C++: CSAMI_AddShardAssemble).
     The archive is byte for byte what CSA_Add on one GPU -- and `csarc a -t1` -- writes.

Nothing here computes anything; torch.distributed is the transport.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

from . import csa


def _dev(backend: str):
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")


def gather_blobs(blob: bytes, dst: int = 0, group=None) -> Optional[List[bytes]]:
    """every rank's `blob` on rank `dst` (list indexed by rank), None elsewhere"""
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world == 1:
        return [blob]
    dev = _dev(dist.get_backend(group))
    lens = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(lens, torch.tensor([len(blob)], dtype=torch.int64, device=dev), group=group)
    lens = [int(x.item()) for x in lens]
    if rank != dst:
        if len(blob):
            t = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(dev)
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, t, dst, group)]):
                w.wait()
        return None
    bufs = {r: torch.empty(lens[r], dtype=torch.uint8, device=dev) for r in range(world) if r != dst and lens[r]}
    if bufs:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, bufs[r], r, group) for r in sorted(bufs)]):
            w.wait()
    return [blob if r == dst else (bufs[r].cpu().numpy().tobytes() if r in bufs else b"") for r in range(world)]


def add(arcname: str, filenames: Sequence[str], group=None, **opts):
    """`csarc a [opts] arcname filenames...` with the tasks spread over the ranks of `group`.
    Call on every rank (process group initialised, one GPU per rank).  -> (rc, stats of this rank)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return csa.add(arcname, filenames, **opts)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = _dev(dist.get_backend(group))
    rc, blob, stats = csa.add_shard_encode(filenames, rank, world, **opts)
    # every rank must agree on failure before anything is exchanged
    flag = torch.tensor([rc], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    worst = int(flag.item())
    if worst == 0:
        flag = torch.tensor([rc], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        worst = int(flag.item())
    if worst != 0:
        return worst, stats
    blobs = gather_blobs(blob, 0, group)
    out = torch.zeros(1, dtype=torch.int32, device=dev)
    if rank == 0:
        rc, wstats = csa.add_shard_assemble(arcname, filenames, blobs, **opts)
        stats = dict(stats, archive_bytes=wstats["archive_bytes"], index_raw_size=wstats["index_raw_size"],
                     index_compressed_size=wstats["index_compressed_size"], n_entries=wstats["n_entries"],
                     n_tasks=wstats["n_tasks"], n_blocks=wstats["n_blocks"])
        out[0] = rc
    dist.broadcast(out, 0, group=group)
    return int(out.item()), stats
