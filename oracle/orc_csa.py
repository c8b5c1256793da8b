"""oracle/orc_csa.py -- TEST INFRASTRUCTURE, not product.

CPU restatement of the reference ARCHIVER's container logic: what `csarc a`
writes around the libcsc streams and what `csarc x/t/l` read back.  Only tests/ may import this
file; the product container lives in csc_amd/csrc/csa_archive.cpp and never calls it.

Pinned against the reference itself: tests/test_oracle_csa.py builds archives with
oracle/_ref/csarc_ref (the reference's own sources + oracle/zero_heap.cpp) and compares them byte
for byte with `create()` below driven by the liborc encoder.

Every function cites the reference lines it follows (paths relative to /root/reference/src).
The libcsc streams themselves are produced/consumed through callables the caller passes in
(`encode(data, dict_size, level) -> (stream_bytes, write_sizes)`, `decode(stream) -> bytes`), so
this file holds no codec.
"""
import os
import stat as _stat

import numpy as np

MAGIC_NUM = 0x20130331
HEADER_SIZE = 24
BLOCK_CAP = 1048576          # csa_io.h:158
DUMMY_FILENAME = "****"      # csa_common.h:79

# ----------------------------------------------------------------------------------------------
# adler32 -- archiver/csa_adler32.cpp:63-129 (zlib's algorithm, seed passed by the caller; the
# archiver starts every fragment at 0, not 1: csarc.cpp:541,553 push checksum 0, csa_io.h:250).
# ----------------------------------------------------------------------------------------------
BASE = 65521


def adler32(adler: int, data) -> int:
    a = adler & 0xFFFF
    b = (adler >> 16) & 0xFFFF
    buf = np.frombuffer(memoryview(data), dtype=np.uint8)
    step = 1 << 20           # sum((n-i)*byte) < 2^20 * 2^20 * 2^8: fits u64
    for s in range(0, len(buf), step):
        blk = buf[s:s + step].astype(np.uint64)
        n = len(blk)
        w = np.arange(n, 0, -1, dtype=np.uint64)
        b = (b + n * a + int((blk * w).sum())) % BASE
        a = (a + int(blk.sum())) % BASE
    return a | (b << 16)


# ----------------------------------------------------------------------------------------------
# time stamps -- archiver/csa_common.cpp:3-25 (decimal_time), :27-39 (unix_time)
# ----------------------------------------------------------------------------------------------
def decimal_time(tt: int) -> int:
    if tt == -1:
        tt = 0
    t = tt
    second = t % 60
    minute = t // 60 % 60
    hour = t // 3600 % 24
    t //= 86400
    term = t // 1461
    t %= 1461
    t += (t >= 59)
    t += (t >= 425)
    t += (t >= 1157)
    year = term * 4 + t // 366 + 1970
    t %= 366
    t += (t >= 60) * 2
    t += (t >= 123)
    t += (t >= 185)
    t += (t >= 278)
    t += (t >= 340)
    month = t // 31 + 1
    day = t % 31 + 1
    return year * 10000000000 + month * 100000000 + day * 1000000 + hour * 10000 + minute * 100 + second


def unix_time(date: int) -> int:
    if date <= 0:
        return -1
    days = [0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334]
    year = date // 10000000000 % 10000
    month = (date // 100000000 % 100 - 1) % 12
    day = date // 1000000 % 100
    hour = date // 10000 % 100
    mn = date // 100 % 100
    sec = date % 100
    return ((day - 1 + days[month] + (1 if (year % 4 == 0 and month > 1) else 0)
             + ((year - 1970) * 1461 + 1) // 4) * 86400 + hour * 3600 + mn * 60 + sec)


# ----------------------------------------------------------------------------------------------
# std::sort -- the archiver orders files (csarc.cpp:512) and tasks (:355,:430) with std::sort, which
# is NOT stable; equal keys (e.g. the equal-size slices of a -p split) come out in the order GNU
# libstdc++'s introsort leaves them.  Third-party dependency, absent from /root/reference:
# libstdc++ (GCC 11.4, bits/stl_algo.h `std::__sort`): introsort with threshold 16, median-of-3
# moved to *first, unguarded Hoare partition, depth limit 2*floor(log2 n) with heapsort fallback,
# then one final insertion pass.  Restated from its published algorithm;
# tests/test_oracle_csa.py checks it against g++'s own std::sort on random key multisets.
# ----------------------------------------------------------------------------------------------
def std_sort(a: list, less) -> None:
    n = len(a)
    if n == 0:
        return
    _introsort_loop(a, 0, n, 2 * (n.bit_length() - 1), less)
    if n > 16:
        _insertion_sort(a, 0, 16, less)
        for i in range(16, n):
            _unguarded_linear_insert(a, i, less)
    else:
        _insertion_sort(a, 0, n, less)


def _swap(a, i, j):
    a[i], a[j] = a[j], a[i]


def _introsort_loop(a, first, last, depth, less):
    while last - first > 16:
        if depth == 0:
            _heap_sort(a, first, last, less)
            return
        depth -= 1
        mid = first + (last - first) // 2
        _move_median_to_first(a, first, first + 1, mid, last - 1, less)
        cut = _unguarded_partition(a, first + 1, last, first, less)
        _introsort_loop(a, cut, last, depth, less)
        last = cut


def _move_median_to_first(a, result, x, y, z, less):
    if less(a[x], a[y]):
        if less(a[y], a[z]):
            _swap(a, result, y)
        elif less(a[x], a[z]):
            _swap(a, result, z)
        else:
            _swap(a, result, x)
    elif less(a[x], a[z]):
        _swap(a, result, x)
    elif less(a[y], a[z]):
        _swap(a, result, z)
    else:
        _swap(a, result, y)


def _unguarded_partition(a, first, last, pivot, less):
    while True:
        while less(a[first], a[pivot]):
            first += 1
        last -= 1
        while less(a[pivot], a[last]):
            last -= 1
        if not first < last:
            return first
        _swap(a, first, last)
        first += 1


def _unguarded_linear_insert(a, last, less):
    val = a[last]
    nxt = last - 1
    while less(val, a[nxt]):
        a[last] = a[nxt]
        last = nxt
        nxt -= 1
    a[last] = val


def _insertion_sort(a, first, last, less):
    for i in range(first + 1, last):
        if less(a[i], a[first]):
            val = a[i]
            a[first + 1:i + 1] = a[first:i]
            a[first] = val
        else:
            _unguarded_linear_insert(a, i, less)


def _adjust_heap(a, first, hole, length, value, less):
    top = hole
    child = hole
    while child < (length - 1) // 2:
        child = 2 * (child + 1)
        if less(a[first + child], a[first + child - 1]):
            child -= 1
        a[first + hole] = a[first + child]
        hole = child
    if (length & 1) == 0 and child == (length - 2) // 2:
        child = 2 * (child + 1)
        a[first + hole] = a[first + child - 1]
        hole = child - 1
    parent = (hole - 1) // 2
    while hole > top and less(a[first + parent], value):
        a[first + hole] = a[first + parent]
        hole = parent
        parent = (hole - 1) // 2
    a[first + hole] = value


def _heap_sort(a, first, last, less):
    # std::__partial_sort(first, last, last): make_heap over the whole range, then sort_heap
    length = last - first
    if length >= 2:
        parent = (length - 2) // 2
        while True:
            _adjust_heap(a, first, parent, length, a[first + parent], less)
            if parent == 0:
                break
            parent -= 1
    while last - first > 1:
        last -= 1
        value = a[last]
        a[last] = a[first]
        _adjust_heap(a, first, 0, last - first, value, less)


# ----------------------------------------------------------------------------------------------
# archive-block writer -- archiver/csa_io.h:145-201 (AsyncWriter::flush/Write), :546-582 (one
# (off,size) record per flushed block), :596-603 (Finish flushes the partial block).  The block
# table depends only on the sequence of Write sizes.
# ----------------------------------------------------------------------------------------------
class BlockCoalescer:
    def __init__(self):
        self.cap = BLOCK_CAP
        self.cur = 0
        self.blocks = []

    def write(self, size: int):
        if self.cur + size > self.cap:            # csa_io.h:188-192
            if self.cur > 0:
                self.blocks.append(self.cur)
            self.cur = 0
            self.cap = max(BLOCK_CAP, size)
        self.cur += size

    def finish(self):
        if self.cur > 0:
            self.blocks.append(self.cur)
        self.cur = 0
        return self.blocks


# ----------------------------------------------------------------------------------------------
# file selection -- csarc.cpp:16-35 (ispath, unix: case sensitive), :807-816 (isselected)
# ----------------------------------------------------------------------------------------------
def ispath(a: str, b: str) -> bool:
    ia = ib = 0
    while ia < len(a):
        ca = a[ia]
        cb = b[ib] if ib < len(b) else "\0"
        if ca == "*":
            while True:
                if ispath(a[ia + 1:], b[ib:]):
                    return True
                if ib >= len(b):
                    return False
                ib += 1
        elif ca == "?":
            if cb == "\0":
                return False
        elif ca == cb and ca == "/" and ia + 1 == len(a):
            return True
        elif ca != cb:
            return False
        ia += 1
        ib += 1
    return ib >= len(b) or b[ib] == "/"


def isselected(filenames, name: str) -> bool:
    if not filenames:
        return True
    return any(ispath(f, name) for f in filenames)


# ----------------------------------------------------------------------------------------------
# directory scan -- csarc.cpp:709-750 (unix scandir), :798-805 (addfile)
# ----------------------------------------------------------------------------------------------
def scan(filenames, recurse: bool) -> dict:
    index = {}

    def addfile(name, edate, esize, eattr):
        if not isselected(filenames, name):
            return
        index[name] = {"edate": edate, "esize": esize, "eattr": eattr, "ext": b"\0\0\0\0", "frags": []}

    def scandir(filename):
        while len(filename) > 1 and filename.endswith("/"):
            filename = filename[:-1]
        try:
            sb = os.lstat(filename)
        except OSError:
            return
        if _stat.S_ISREG(sb.st_mode):
            addfile(filename, decimal_time(int(sb.st_mtime)), sb.st_size, ord("u") + (sb.st_mode << 8))
        if _stat.S_ISDIR(sb.st_mode):
            addfile("/" if filename == "/" else filename + "/", decimal_time(int(sb.st_mtime)), 0,
                    ord("u") + (sb.st_mode << 8))
            if recurse:
                for d in os.listdir(filename):
                    scandir(filename + ("" if filename == "/" else "/") + d)

    for f in filenames:
        scandir(f)
    return index


def file_ext(name: str) -> bytes:
    # csarc.cpp:498-509: up to 4 chars after the last '.', lower-cased, zero padded
    dot = name.rfind(".")
    slash = name.rfind("/")
    if dot < 0 or (slash >= 0 and dot < slash):
        return b"\0\0\0\0"
    e = name[dot + 1:dot + 5].lower().encode("latin-1", "replace")
    return (e + b"\0\0\0\0")[:4]


# ----------------------------------------------------------------------------------------------
# task split -- csarc.cpp:490-557 (Add), :76-92 (comparators), :355 (largest task first)
# ----------------------------------------------------------------------------------------------
def plan_tasks(index: dict, split_count: int):
    names = sorted(index.keys(), key=lambda s: s.encode("latin-1", "surrogateescape"))  # std::map order
    itlist = []
    for n in names:
        if n.endswith("/"):
            continue
        index[n]["ext"] = file_ext(n)
        itlist.append(n)

    def by_ext(a, b):
        ea, eb = index[a]["ext"], index[b]["ext"]
        if ea != eb:
            return ea < eb
        if index[a]["esize"] > 64 * 1024 or index[b]["esize"] > 64 * 1024:
            return index[a]["esize"] < index[b]["esize"]
        return a.encode("latin-1", "surrogateescape") < b.encode("latin-1", "surrogateescape")

    std_sort(itlist, by_ext)

    # csarc.cpp:516-530 -- `sit` keeps moving while the count of non-empty files is still 1, so an
    # empty file sorted after the only non-empty one becomes the "single file" (and yields no task)
    valid = 0
    single = False
    sit = None
    for n in itlist:
        if index[n]["esize"] > 0:
            valid += 1
        if valid == 1:
            single = True
            sit = n
        elif valid > 1:
            single = False
            break

    tasks = []
    if single:
        esize = index[sit]["esize"]
        split = esize // split_count
        split = 1048576 if split < 1048576 else split
        split += 4
        off = 0
        while off < esize:
            bsize = min(split, esize - off)
            tasks.append({"files": [[sit, off, bsize]], "total": bsize})
            off += bsize
    else:
        cur = {"files": [], "total": 0}
        for i, n in enumerate(itlist):
            if i and index[n]["ext"] != index[itlist[i - 1]]["ext"] and cur["total"] > 64 * 1024:
                tasks.append(cur)
                cur = {"files": [], "total": 0}
            cur["files"].append([n, 0, index[n]["esize"]])
            cur["total"] += index[n]["esize"]
        if cur["total"]:
            tasks.append(cur)

    std_sort(tasks, lambda a, b: a["total"] > b["total"])
    return tasks


# ----------------------------------------------------------------------------------------------
# index -- archiver/csa_indexpack.cpp:67-94 (file entry), :127-150 (task blocks), :166-189 (PackIndex)
# ----------------------------------------------------------------------------------------------
def _p4(v):
    return int(v & 0xFFFFFFFF).to_bytes(4, "little")


def _p8(v):
    return int(v & 0xFFFFFFFFFFFFFFFF).to_bytes(8, "little")


def _name_bytes(n: str) -> bytes:
    return n.encode("latin-1", "surrogateescape") if isinstance(n, str) else n


def pack_index(index: dict, abindex: dict, arcname: str) -> bytes:
    out = bytearray()
    total = 4
    names = sorted(index.keys(), key=_name_bytes)
    out += _p4(len(names))
    for n in names:
        e = index[n]
        nb = _name_bytes(n)
        total += 4 + len(nb) + 3 * 8 + 1 + len(e["frags"]) * 32
        out += _p4(len(nb)) + nb + _p8(e["edate"]) + _p8(e["esize"]) + _p8(e["eattr"])
        out.append(len(e["frags"]) & 0xFF)
        for f in e["frags"]:
            out += _p4(f["bid"]) + _p4(f["checksum"]) + _p8(f["posblock"]) + _p8(f["size"]) + _p8(f["posfile"])
    total += 4
    out += _p4(len(abindex))
    for tid in sorted(abindex.keys()):
        blocks = abindex[tid]
        # csa_indexpack.cpp:129-134 still counts `4 + filename.size()` although :141-143 no longer
        # writes the name: the buffer is that much too long; with a zero-filling heap the tail is 0.
        total += 8 + 4 + len(_name_bytes(arcname)) + 4 + len(blocks) * 16
        out += _p8(tid) + _p4(len(blocks))
        for off, size in blocks:
            out += _p8(off) + _p8(size)
    assert len(out) <= total
    return bytes(out) + bytes(total - len(out))


def unpack_index(buf: bytes):
    pos = 0

    def g4():
        nonlocal pos
        v = int.from_bytes(buf[pos:pos + 4], "little")
        pos += 4
        return v

    def g8(signed=False):
        nonlocal pos
        v = int.from_bytes(buf[pos:pos + 8], "little", signed=signed)
        pos += 8
        return v

    index = {}
    for _ in range(g4()):
        ln = g4()
        name = buf[pos:pos + ln].decode("latin-1")
        pos += ln
        e = {"edate": g8(True), "esize": g8(True), "eattr": g8(True), "frags": []}
        nfr = int.from_bytes(buf[pos:pos + 1], "little", signed=True)    # int8_t: csa_indexpack.cpp:105
        pos += 1
        for _ in range(max(nfr, 0)):
            e["frags"].append({"bid": g4(), "checksum": g4(), "posblock": g8(), "size": g8(), "posfile": g8()})
        index[name] = e
    abindex = {}
    for _ in range(g4()):
        tid = g8()
        abindex[tid] = [(g8(), g8()) for _ in range(g4())]
    return index, abindex, pos


# ----------------------------------------------------------------------------------------------
# csarc a -- csarc.cpp:472-575 (Add), :338-409 (compress_mt with one worker == task-id order),
# csa_worker.cpp:23-56 (one task), csa_io.h:215-272 (file reader: posblock, adler32 per file),
# csarc.cpp:219-288 (compress_index + header)
# ----------------------------------------------------------------------------------------------
def create(arcname: str, filenames, *, level=2, dict_size=32000000, recurse=False, split_count=1,
           encode=None, read_file=None) -> bytes:
    if read_file is None:
        def read_file(name, off, size):
            with open(name, "rb") as f:
                f.seek(off)
                return f.read(size)
    index = scan(filenames, recurse)
    tasks = plan_tasks(index, max(split_count, 1))
    body = bytearray(HEADER_SIZE)
    abindex = {}
    for tid, t in enumerate(tasks):
        cum = 0
        parts = []
        frags = []
        for name, off, size in t["files"]:
            try:
                data = read_file(name, off, size)
            except OSError:
                frags.append((name, off, 0, 0, 0))           # csa_io.h:232-236: size 0, posblock stays 0
                continue
            frags.append((name, off, len(data), cum, adler32(0, data)))
            cum += len(data)
            parts.append(data)
        raw = b"".join(parts)
        stream, wsizes = encode(raw, min(dict_size, t["total"]), level)
        co = BlockCoalescer()
        for s in wsizes:
            co.write(s)
        blocks = []
        pos = len(body)                       # one worker: a task's blocks are appended back to back
        for bs in co.finish():
            blocks.append((pos, bs))
            pos += bs
        assert pos - len(body) == len(stream)
        body += stream
        abindex[tid] = blocks
        for name, off, size, posblock, cks in frags:
            index[name]["frags"].append({"bid": tid, "checksum": cks, "posblock": posblock, "size": size,
                                         "posfile": off})
    raw_index = pack_index(index, abindex, arcname)
    index_pos = len(body)
    istream, _ = encode(raw_index, 256 * 1024, 2)            # csarc.cpp:251 (dict NOT clamped to the size)
    body += istream
    body[8:24] = _p8(index_pos) + _p4(len(istream)) + _p4(len(raw_index))
    body[0:3] = b"CSA"
    body[3:7] = _p4(MAGIC_NUM)
    body[7:8] = b"1"
    return bytes(body)


# ----------------------------------------------------------------------------------------------
# csarc l/t/x -- csarc.cpp:577-598 (check_header), :290-336 (decompress_index), :667-700 (Test task
# build), :411-470 (decompress_mt), csa_io.h:286-377 (file writer: posblock gaps, adler32 verify)
# ----------------------------------------------------------------------------------------------
def parse(arc: bytes, decode):
    if len(arc) < HEADER_SIZE or arc[0:3] != b"CSA" or arc[7:8] != b"1" \
            or int.from_bytes(arc[3:7], "little") != MAGIC_NUM:
        raise ValueError("Invalid csarc file")
    index_pos = int.from_bytes(arc[8:16], "little")
    csize = int.from_bytes(arc[16:20], "little")
    rsize = int.from_bytes(arc[20:24], "little")
    raw = decode(arc[index_pos:index_pos + csize])
    raw = raw[:rsize]
    index, abindex, used = unpack_index(raw)
    return {"index_pos": index_pos, "index_csize": csize, "index_rsize": rsize, "index_raw": raw,
            "index": index, "abindex": abindex, "index_used": used}


def extract(arc: bytes, decode, only=None):
    """-> ({name: bytes}, [names whose adler32 did not verify])"""
    info = parse(arc, decode)
    tasks = {}
    for name in sorted(info["index"].keys(), key=_name_bytes):
        if only and not isselected(only, name):
            continue
        for fr in info["index"][name]["frags"]:
            if fr["size"]:
                tasks.setdefault(fr["bid"], []).append((fr["posblock"], name, fr))
    files = {n: bytearray(e["esize"]) for n, e in info["index"].items()
             if not n.endswith("/") and (not only or isselected(only, n))}
    bad = []
    for bid, frs in tasks.items():
        stream = b"".join(arc[o:o + s] for o, s in info["abindex"][bid])
        raw = decode(stream)
        for posblock, name, fr in sorted(frs, key=lambda x: x[0]):
            piece = raw[posblock:posblock + fr["size"]]
            if adler32(0, piece) != fr["checksum"]:
                bad.append(name)
            files[name][fr["posfile"]:fr["posfile"] + fr["size"]] = piece
    return {n: bytes(v) for n, v in files.items()}, bad
