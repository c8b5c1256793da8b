"""Seeded file trees + `csarc a` option sets shared by tools/make_golden_csa.py (which records what
the REFERENCE archiver writes for them), tests/test_oracle_csa.py (oracle container on CPU) and
tests/test_gpu_csa.py (the product container on the GPU).

Everything that ends up in the archive is pinned: file bytes come from the seeded corpus
generator, mtimes and modes are set explicitly, names are relative to the tree root (tests run
with cwd = that root) and the archive name has a fixed length (its length enters the index size,
csa_indexpack.cpp:129-134)."""
import os

import cases

MTIME = 1700000000          # 2023-11-14 22:13:20 UTC
ARCNAME = "out.csa"


def _f(kind, seed, n, off=0):
    return [[kind, seed, off, n]] if n else []


# name -> {"files": {relpath: input spec}, "dirs": [...], "args": (filenames), "opts": {...}}
CSA_CASES = {
    # directories, an empty file, a file without extension, one task per extension group
    "mixed_tree": {
        "files": {"d/a.txt": _f("text", 1, 300000), "d/b.txt": _f("text", 2, 70000), "d/sub/c.exe": _f("exe", 3, 200000),
                  "d/sub/e.dat": _f("delta", 4, 150000), "d/empty": [], "d/noext": _f("random", 5, 5000)},
        "args": ["d"], "opts": {"level": 2, "dict_size": 1 << 20, "recurse": True},
    },
    # > 16 files and > 16 tasks: std::sort's introsort order, sizes straddling the 64 KiB rules
    "many_files": {
        "files": {**{f"m/f{i:02d}.{'abcdefghijklmnopqrstuvwx'[i % 24]}{i % 3}": _f(("text", "exe", "delta", "entropy8")[i % 4], 100 + i,
                                                                               (70000, 66000, 3000, 131072, 65536, 65537)[i % 6] + 17 * (i % 5))
                     for i in range(40)},
                  **{f"m/same{i}.log": _f("text", 7, 80000, off=i * 1000) for i in range(20)},
                  **{f"m/tiny{i:02d}.cfg": _f("text", 9, 100 + i) for i in range(20)}},
        "args": ["m"], "opts": {"level": 1, "dict_size": 256 << 10, "recurse": True},
    },
    # single file, -p3 (slice = esize/3 + 4)
    "single_split3": {
        "files": {"big.bin": _f("silesia", 6, 5 << 20)},
        "args": ["big.bin"], "opts": {"level": 2, "dict_size": 4 << 20, "split_count": 3},
    },
    # single file, 21 slices of 1 MiB + 4: equal keys through introsort, 127-fragment field
    "single_split_many": {
        "files": {"wiki.xml": _f("text", 40, 21 << 20)},
        "args": ["wiki.xml"], "opts": {"level": 1, "dict_size": 1 << 20, "split_count": 64},
    },
    # csarc.cpp:516-530 quirk: an empty file sorted behind the only non-empty one becomes "the" single file
    "single_shadowed_by_empty": {
        "files": {"q/a.bin": _f("exe", 8, 100000), "q/z.zzz": []},
        "args": ["q"], "opts": {"level": 2, "dict_size": 1 << 20, "recurse": True},
    },
    # the same with the empty file sorted first: a normal single-file archive
    "single_after_empty": {
        "files": {"q/z.bin": _f("exe", 8, 100000), "q/a.aaa": []},
        "args": ["q"], "opts": {"level": 2, "dict_size": 1 << 20, "recurse": True},
    },
    # explicit names without -r, level 3 (advanced parser), default dictionary
    "named_files_m3": {
        "files": {"x.txt": _f("text", 11, 400000), "y.txt": _f("text", 12, 90000), "skip.txt": _f("text", 13, 1000)},
        "args": ["x.txt", "y.txt"], "opts": {"level": 3},
    },
    # level 5 (binary tree), task crossing the 2 MiB chunk boundary, files ending exactly on it
    "chunk_edges_m5": {
        "files": {"c/p0.raw": _f("delta", 21, 2 << 20), "c/p1.raw": _f("delta", 22, 1 << 20), "c/p2.raw": [],
                  "c/p3.raw": _f("text", 23, (1 << 20) + 5)},
        "args": ["c"], "opts": {"level": 5, "dict_size": 8 << 20, "recurse": True},
    },
    # nothing but directories and empty files: an archive with no task
    "no_data": {
        "files": {"e/one": [], "e/two.txt": []},
        "args": ["e"], "opts": {"level": 2, "recurse": True},
    },
}

# cases small enough for the (slow, one stream at a time) CPU oracle in the default CPU suite
CPU_CASES = [c for c in CSA_CASES if c != "single_split_many"]


def make_tree(root, case):
    """materialise CSA_CASES[case] under root; returns {relpath: bytes}"""
    spec = CSA_CASES[case]
    content = {}
    dirs = set()
    for rel, parts in spec["files"].items():
        path = os.path.join(root, rel)
        d = os.path.dirname(rel)
        while d:
            dirs.add(d)
            d = os.path.dirname(d)
        os.makedirs(os.path.dirname(path) or root, exist_ok=True)
        data = cases.build(parts)
        with open(path, "wb") as f:
            f.write(data)
        os.chmod(path, 0o644)
        os.utime(path, (MTIME, MTIME))
        content[rel] = data
    for d in sorted(dirs, key=len, reverse=True):
        os.chmod(os.path.join(root, d), 0o755)
        os.utime(os.path.join(root, d), (MTIME, MTIME))
    return content


def csarc_argv(case):
    """the `csarc a` command line of the case (without the program name)"""
    o = CSA_CASES[case]["opts"]
    argv = ["a", f"-m{o.get('level', 2)}"]
    if "dict_size" in o:
        d = o["dict_size"]
        argv.append(f"-d{d >> 20}m" if d % (1 << 20) == 0 else f"-d{d >> 10}k")
    if o.get("recurse"):
        argv.append("-r")
    if o.get("split_count", 1) != 1:
        argv.append(f"-p{o['split_count']}")
    return argv + ["-f", ARCNAME] + list(CSA_CASES[case]["args"])
