"""Seeded synthetic corpora (csrc/corpus.c): stand-ins for enwik8/9, silesia.tar and the
EXE/text/delta mix that BASELINE.json names -- the real files are not available offline.

`fill(kind, seed, offset, n)` is random-access, so rank r of a `-p` split can produce its own
task slice (csarc.cpp:532-543) without generating the bytes before it.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

KINDS = {"text": 0, "exe": 1, "delta": 2, "random": 3, "entropy8": 4, "silesia": 5, "mix5": 6}
# seeds fixed by SURVEY.md section 8(d)
SEED_ENWIK8 = 0xC5C00001
SEED_ENWIK9 = 0xC5C00002
SEED_EXE = 0xC5C00003
SEED_DELTA = 0xC5C00004

_lib = None


def _load():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcsc_corpus.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run __graft_entry__.build()")
        _lib = C.CDLL(path)
        _lib.csc_corpus_fill.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
        _lib.csc_corpus_fill.restype = None
    return _lib


def fill(kind, seed: int, offset: int, n: int) -> np.ndarray:
    """bytes [offset, offset+n) of stream (kind, seed) as a uint8 array"""
    k = KINDS[kind] if isinstance(kind, str) else int(kind)
    out = np.empty(n, dtype=np.uint8)
    if n:
        _load().csc_corpus_fill(k, seed, offset, out.ctypes.data, n)
    return out


def task_slices(total: int, split: int):
    """The archiver's single-file -p split (csarc.cpp:532-543): (offset, size) per task."""
    from .tasks import split_single_file
    return split_single_file(total, split)


# real corpora, when the box has them (SURVEY.md section 8d / BASELINE.md section 3): $CSC_CORPUS_DIR/<name>
STANDINS = {"enwik8": ("text", SEED_ENWIK8, 10 ** 8), "enwik9": ("text", SEED_ENWIK9, 10 ** 9),
            "silesia.tar": ("silesia", 6, 211957760), "mix5": ("mix5", SEED_EXE, 10 ** 10)}


class Source:
    """`name` (enwik8 / enwik9 / silesia.tar / mix5) as a random-access byte source: the real file under $CSC_CORPUS_DIR when it
    is there, the seeded stand-in otherwise.  `label` says which (bench.py prints it in `data`); the committed reference digests
    (tests/golden/) are of the stand-ins, so `synthetic` gates every comparison with them."""

    def __init__(self, name: str):
        kind, seed, size = STANDINS[name]
        self.name, self.kind, self.seed = name, kind, seed
        d = os.environ.get("CSC_CORPUS_DIR")
        path = os.path.join(d, name) if d else None
        if path and os.path.isfile(path) and os.path.getsize(path) > 0:
            self.mm = np.memmap(path, dtype=np.uint8, mode="r")
            self.size, self.synthetic = int(self.mm.shape[0]), False
            self.label = f"real file {path} ({self.size} bytes, $CSC_CORPUS_DIR)"
        else:
            self.mm, self.size, self.synthetic = None, size, True
            self.label = f"synthetic (seeded {name} stand-in, csc_amd/csrc/corpus.c kind={kind} seed={seed:#x})"

    def read(self, offset: int, n: int) -> np.ndarray:
        n = max(0, min(n, self.size - offset))
        if self.mm is not None:
            return np.array(self.mm[offset:offset + n])
        return fill(self.kind, self.seed, offset, n)
