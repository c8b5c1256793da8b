"""N > 1 path on CPU: world_size 2 over gloo.  Each rank encodes the tasks `assign` gives it (with
the oracle standing in for the GPU encoder -- this test is about sharding and gathering, not the
kernels) and rank 0 checks that the gathered per-task streams equal a single-process run, in
task-id order (the `csarc -t1` layout), for every world size."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import ctypes as C, hashlib, os, sys
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import torch.distributed as dist
    from csc_amd import corpus, tasks
    from csc_amd.capi import CscLib
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    orc = CscLib(os.path.join(%r, "oracle", "liborc.so"))
    orc.lib.orc_zero_alloc.restype = C.c_void_p
    za = orc.lib.orc_zero_alloc()
    TOTAL, SPLIT = 5 * 1048576 + 12345, 4
    tl = tasks.split_single_file(TOTAL, SPLIT)
    mine = tasks.assign(tl, world)[rank]
    local = {}
    for tid in mine:
        off, n = tl[tid]
        data = corpus.fill("text", 77, off, n).tobytes()
        rc, s = orc.encode(data, 3, 64 << 20, alloc=za)          # dict = min(dict, task size), csa_worker.cpp:35
        assert rc == 0
        local[tid] = (len(s), hashlib.sha256(s).hexdigest())
    merged = tasks.gather_results(local, world, rank)
    dist.barrier()
    if rank == 0:
        assert sorted(merged) == list(range(len(tl)))
        print("RESULT", [merged[t] for t in sorted(merged)])
    dist.destroy_process_group()
""") % (ROOT, ROOT, ROOT)


def run(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29531 + world))
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(WORKER)
        path = f.name
    try:
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                              "--master-addr", "127.0.0.1", "--master-port", str(29541 + world), path],
                             env=env, capture_output=True, text=True, timeout=600)
    finally:
        os.unlink(path)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    assert line, out.stdout[-2000:] + out.stderr[-2000:]
    return line[0]


def test_two_ranks_equal_one_rank():
    assert run(1) == run(2)


def test_assignment_covers_every_task_once():
    from csc_amd import tasks
    tl = tasks.split_single_file(10 ** 9, 8)
    for world in (1, 2, 4, 8):
        a = tasks.assign(tl, world)
        assert sorted(t for r in a for t in r) == list(range(8))
        assert max(len(r) for r in a) - min(len(r) for r in a) <= 1
    # largest first: the short last slice is dispatched last (csarc.cpp:355)
    assert tasks.dispatch_order(tl)[-1] == 7


EXCHANGE = textwrap.dedent("""
    import hashlib, os, sys
    sys.path.insert(0, %r)
    import torch.distributed as dist
    from csc_amd import sharded
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    def blob_of(r):                      # ragged on purpose: rank 1 has nothing to send
        n = 0 if r == 1 else 1000003 * (r + 1) + 17
        return bytes((i * 131 + r * 7) & 0xFF for i in range(min(n, 4096))) * (n // 4096) + b"x" * (n %% 4096) if n else b""
    got = sharded.gather_blobs(blob_of(rank), 0)
    if rank == 0:
        assert got is not None and len(got) == world
        for r in range(world):
            assert got[r] == blob_of(r), r
        print("RESULT ok", [len(b) for b in got])
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()
""") % (ROOT,)


def _launch(src, world, port):
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(src)
        path = f.name
    try:
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                              "--master-addr", "127.0.0.1", "--master-port", str(port), path],
                             capture_output=True, text=True, timeout=600)
    finally:
        os.unlink(path)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def test_shard_blobs_reach_rank0_ragged():
    """the one exchange of the sharded Add (csc_amd/sharded.py): variable-length blobs, an empty one among them"""
    for world in (2, 3):
        assert "RESULT ok" in _launch(EXCHANGE, world, 29560 + world)
