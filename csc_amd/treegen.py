"""A seeded MANY-TASK workload for the archiver path: a file tree whose multi-file task split (csarc.cpp:545-557: files sorted by
their 4-character extension, a new task at every extension change once the task holds 64 KiB) yields thousands of independent
libcsc streams -- the regime in which the GPUs of a node add throughput (a single file splits into at most 127 tasks, and one
MI355X runs those at once).  Bytes come from the corpus generator (csrc/corpus.c), names / sizes / kinds from the file number
alone, mtimes and modes are fixed: the archive `csarc a -t1` writes for the tree is a constant, recorded by
tools/make_golden_tree.py in tests/golden/tree_workload.json.
"""
from __future__ import annotations

import os

from . import corpus

MTIME = 1700000000
KINDS = ("text", "exe", "delta", "silesia")
SPECS = {   # name -> (files, files per extension, base size, size spread)
    "tree": (4096, 2, 384 << 10, 256 << 10),        # 2048 tasks, ~2.1 GB
    "tree_mid": (1024, 2, 384 << 10, 256 << 10),    # 512 tasks, ~0.5 GB (development)
    "tree_small": (256, 2, 96 << 10, 64 << 10),     # 128 tasks, ~32 MB (tests)
}


def _b36(n: int, width: int) -> str:
    s = ""
    for _ in range(width):
        s = "0123456789abcdefghijklmnopqrstuvwxyz"[n % 36] + s
        n //= 36
    return s


def files(spec: str):
    """[(relative path, kind, seed, size)] of the tree, in file-number order"""
    n, per_ext, base, spread = SPECS[spec]
    out = []
    for i in range(n):
        g = i // per_ext
        ext = "x" + _b36(g, 3)                                   # 4 characters: all of them count (csarc.cpp:508-509)
        size = base + (i * 2654435761 % spread)
        out.append((f"t/d{i % 64:02d}/f{i:04d}.{ext}", KINDS[g % 4], 5000 + i, size))
    return out


def total_bytes(spec: str) -> int:
    return sum(f[3] for f in files(spec))


def materialize(root: str, spec: str) -> int:
    """write the tree under `root` (idempotent: an existing file of the right size is kept); -> total bytes"""
    tot = 0
    dirs = set()
    for rel, kind, seed, size in files(spec):
        path = os.path.join(root, rel)
        d = os.path.dirname(path)
        if d not in dirs:
            os.makedirs(d, exist_ok=True)
            dirs.add(d)
        if not (os.path.isfile(path) and os.path.getsize(path) == size):
            with open(path, "wb") as f:
                f.write(corpus.fill(kind, seed, 0, size).tobytes())
        os.chmod(path, 0o644)
        os.utime(path, (MTIME, MTIME))
        tot += size
    for d in sorted(dirs | {os.path.join(root, "t")}, key=len, reverse=True):
        os.chmod(d, 0o755)
        os.utime(d, (MTIME, MTIME))
    return tot
