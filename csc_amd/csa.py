"""csc_amd.csa -- Python mirror of the `.csa` container entry points (include/csa_mi355x.h).

The four operations of the reference's `csarc` (`class CSArc`, archiver/csarc.cpp:37-70) with its
options (ParseArg, csarc.cpp:137-208), served by `libcsc_mi355x.so`: streams and adler32 on the
GPU, container logic in C++.  Nothing here computes anything; it marshals arguments.
"""
import ctypes as C
from typing import Iterable, List, Optional, Sequence

from . import load


class CSAOptions(C.Structure):
    _fields_ = [
        ("level", C.c_int), ("dict_size", C.c_uint32), ("recurse", C.c_int), ("overwrite", C.c_int),
        ("verbose", C.c_int), ("mt_count", C.c_int), ("split_count", C.c_int), ("to_dir", C.c_char_p),
        ("device_streams", C.c_int), ("hbm_budget", C.c_uint64), ("task_bytes", C.c_uint64),
    ]


class CSAStats(C.Structure):
    _fields_ = [
        ("raw_bytes", C.c_uint64), ("archive_bytes", C.c_uint64), ("index_raw_size", C.c_uint64),
        ("index_compressed_size", C.c_uint64), ("n_entries", C.c_uint32), ("n_tasks", C.c_uint32),
        ("n_blocks", C.c_uint32), ("verify_failures", C.c_uint32), ("seconds_total", C.c_double),
        ("seconds_encode", C.c_double), ("peak_streams", C.c_uint32), ("reserved", C.c_uint32),
        ("seconds_io", C.c_double), ("seconds_setup", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class CSAFrag(C.Structure):
    _fields_ = [("bid", C.c_uint32), ("checksum", C.c_uint32), ("posblock", C.c_uint64),
                ("size", C.c_uint64), ("posfile", C.c_uint64)]


LIST_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.POINTER(CSAFrag))

SYMBOLS = ["CSA_OptionsInit", "CSA_Add", "CSA_Extract", "CSA_Test", "CSA_List", "CSA_ReadIndex",
           "CSA_Adler32", "CSAMI_Adler32Device", "CSA_DecimalTime", "CSA_UnixTime",
           "CSAMI_AddShardEncode", "CSAMI_FreeBlob", "CSAMI_AddShardAssemble", "CSAMI_PlanInfo", "CSAMI_PlanShards", "CSA_IndexRoundTrip"]

CSA_MAX_FRAGMENTS = 127
CSA_TOO_MANY_FRAGMENTS = -94
CSA_UNSAFE_NAME = -93

_lib = None


def lib():
    """the product library with the CSA_* prototypes bound"""
    global _lib
    if _lib is None:
        L = load().lib
        names = C.POINTER(C.c_char_p)
        L.CSA_OptionsInit.argtypes = [C.POINTER(CSAOptions)]
        L.CSA_OptionsInit.restype = None
        for fn in (L.CSA_Add, L.CSA_Extract, L.CSA_Test):
            fn.argtypes = [C.c_char_p, names, C.c_int, C.POINTER(CSAOptions), C.POINTER(CSAStats)]
            fn.restype = C.c_int
        L.CSA_List.argtypes = [C.c_char_p, names, C.c_int, LIST_FN, C.c_void_p]
        L.CSA_List.restype = C.c_int
        L.CSA_ReadIndex.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
        L.CSA_ReadIndex.restype = C.c_int64
        L.CSA_Adler32.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64]
        L.CSA_Adler32.restype = C.c_uint32
        L.CSAMI_Adler32Device.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32)]
        L.CSAMI_Adler32Device.restype = C.c_int
        L.CSA_DecimalTime.argtypes = [C.c_int64]
        L.CSA_DecimalTime.restype = C.c_int64
        L.CSA_UnixTime.argtypes = [C.c_int64]
        L.CSA_UnixTime.restype = C.c_int64
        L.CSAMI_AddShardEncode.argtypes = [names, C.c_int, C.POINTER(CSAOptions), C.c_int, C.c_int,
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(CSAStats)]
        L.CSAMI_AddShardEncode.restype = C.c_int
        L.CSAMI_FreeBlob.argtypes = [C.c_void_p]
        L.CSAMI_FreeBlob.restype = None
        L.CSAMI_AddShardAssemble.argtypes = [C.c_char_p, names, C.c_int, C.POINTER(CSAOptions), C.POINTER(C.c_void_p),
                                             C.POINTER(C.c_uint64), C.c_int, C.POINTER(CSAStats)]
        L.CSAMI_AddShardAssemble.restype = C.c_int
        L.CSAMI_PlanInfo.argtypes = [names, C.c_int, C.POINTER(CSAOptions), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.CSAMI_PlanInfo.restype = C.c_int
        L.CSA_IndexRoundTrip.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.CSA_IndexRoundTrip.restype = C.c_int64
        _lib = L
    return _lib


def _names(filenames: Iterable[str]):
    enc = [f.encode() if isinstance(f, str) else bytes(f) for f in filenames]
    arr = (C.c_char_p * max(len(enc), 1))(*enc)
    return arr, len(enc)


def options(level: int = 2, dict_size: int = 32000000, recurse: bool = False, overwrite: bool = False,
            verbose: bool = False, mt_count: int = 1, split_count: int = 1, to_dir: Optional[str] = None,
            device_streams: int = 0, hbm_budget: int = 0, task_bytes: int = 0) -> CSAOptions:
    o = CSAOptions()
    lib().CSA_OptionsInit(C.byref(o))
    o.level, o.dict_size, o.recurse, o.overwrite = level, dict_size, int(recurse), int(overwrite)
    o.verbose, o.mt_count, o.split_count = int(verbose), mt_count, split_count
    if to_dir is not None:
        o.to_dir = to_dir.encode()
    o.device_streams, o.hbm_budget, o.task_bytes = device_streams, hbm_budget, task_bytes
    return o


def add(arcname: str, filenames: Sequence[str], **opts):
    """`csarc a [opts] arcname filenames...` -> (rc, stats dict)"""
    o, st = options(**opts), CSAStats()
    arr, n = _names(filenames)
    rc = lib().CSA_Add(arcname.encode(), arr, n, C.byref(o), C.byref(st))
    return rc, st.as_dict()


def plan_info(filenames: Sequence[str], **opts):
    """the plan `add` would execute (host only) -> (rc, number of task streams, largest fragment count of any file)"""
    o = options(**opts)
    arr, n = _names(filenames)
    nt, mf = C.c_uint32(), C.c_uint32()
    rc = lib().CSAMI_PlanInfo(arr, n, C.byref(o), C.byref(nt), C.byref(mf))
    return rc, int(nt.value), int(mf.value)


def index_round_trip(raw: bytes):
    """bounds-checked parse + re-pack of a raw index buffer -> (rc, bytes): rc = re-packed size, -1 unparsable, CSA_UNSAFE_NAME"""
    buf = (C.c_uint8 * max(len(raw), 1)).from_buffer_copy(raw or b"\0")
    n = lib().CSA_IndexRoundTrip(buf, len(raw), None, 0)
    if n < 0:
        return int(n), b""
    out = (C.c_uint8 * max(n, 1))()
    lib().CSA_IndexRoundTrip(buf, len(raw), out, n)
    return int(n), bytes(out[:n])


def add_shard_encode(filenames: Sequence[str], rank: int, world: int, **opts):
    """this rank's tasks of `csarc a [opts] arc filenames...` encoded on its GPU -> (rc, blob bytes, stats dict)"""
    o, st = options(**opts), CSAStats()
    arr, n = _names(filenames)
    ptr, ln = C.c_void_p(), C.c_uint64()
    rc = lib().CSAMI_AddShardEncode(arr, n, C.byref(o), rank, world, C.byref(ptr), C.byref(ln), C.byref(st))
    blob = b""
    if rc == 0:
        blob = C.string_at(ptr, ln.value)
        lib().CSAMI_FreeBlob(ptr)
    return rc, blob, st.as_dict()


def plan_shards(filenames: Sequence[str], world: int, **opts):
    """the deal add_shard_encode uses, without encoding anything (host only): -> (rank of every task, cost estimate of every task)"""
    o = options(**opts)
    arr, n = _names(filenames)
    L = lib()
    L.CSAMI_PlanShards.restype = C.c_int
    nt = L.CSAMI_PlanShards(arr, n, C.byref(o), world, None, None, 0)
    if nt < 0:
        raise RuntimeError(f"CSAMI_PlanShards failed rc={nt}")
    ro, co = (C.c_uint32 * max(nt, 1))(), (C.c_double * max(nt, 1))()
    L.CSAMI_PlanShards(arr, n, C.byref(o), world, ro, co, nt)
    return list(ro[:nt]), list(co[:nt])


def add_shard_assemble(arcname: str, filenames: Sequence[str], blobs: Sequence[bytes], **opts):
    """every rank's blob -> the archive `csarc a -t1` writes; -> (rc, stats dict)"""
    o, st = options(**opts), CSAStats()
    arr, n = _names(filenames)
    keep = [C.create_string_buffer(b, len(b)) if len(b) else C.create_string_buffer(1) for b in blobs]
    ptrs = (C.c_void_p * max(len(keep), 1))(*[C.cast(k, C.c_void_p) for k in keep])
    lens = (C.c_uint64 * max(len(keep), 1))(*[len(b) for b in blobs])
    rc = lib().CSAMI_AddShardAssemble(arcname.encode(), arr, n, C.byref(o), ptrs, lens, len(blobs), C.byref(st))
    return rc, st.as_dict()


def extract(arcname: str, filenames: Sequence[str] = (), **opts):
    """`csarc x [opts] arcname [names...]` -> (rc, stats dict)"""
    o, st = options(**opts), CSAStats()
    arr, n = _names(filenames)
    rc = lib().CSA_Extract(arcname.encode(), arr, n, C.byref(o), C.byref(st))
    return rc, st.as_dict()


def test(arcname: str, filenames: Sequence[str] = (), **opts):
    """`csarc t [opts] arcname [names...]` -> (rc, stats dict)"""
    o, st = options(**opts), CSAStats()
    arr, n = _names(filenames)
    rc = lib().CSA_Test(arcname.encode(), arr, n, C.byref(o), C.byref(st))
    return rc, st.as_dict()


def list_entries(arcname: str, filenames: Sequence[str] = ()) -> Optional[List[dict]]:
    """`csarc l -v arcname [names...]` -> entries in name order, or None on a bad archive"""
    out = []

    def cb(ctx, name, esize, edate, eattr, nfrags, frags):
        out.append({"name": name.decode("latin-1"), "esize": esize, "edate": edate, "eattr": eattr,
                    "frags": [{"bid": frags[i].bid, "checksum": frags[i].checksum, "posblock": frags[i].posblock,
                               "size": frags[i].size, "posfile": frags[i].posfile} for i in range(nfrags)]})

    fn = LIST_FN(cb)
    arr, n = _names(filenames)
    rc = lib().CSA_List(arcname.encode(), arr, n, fn, None)
    return out if rc == 0 else None


def read_index(arcname: str) -> Optional[bytes]:
    n = lib().CSA_ReadIndex(arcname.encode(), None, 0)
    if n < 0:
        return None
    buf = (C.c_uint8 * max(n, 1))()
    lib().CSA_ReadIndex(arcname.encode(), buf, n)
    return bytes(buf[:n])


def adler32(data, adler: int = 0) -> int:
    b = bytes(data)
    return int(lib().CSA_Adler32(adler, b, len(b)))


def adler32_device(device_ptr: int, nbytes: int, adler: int = 0) -> int:
    out = C.c_uint32()
    rc = lib().CSAMI_Adler32Device(adler, C.c_void_p(device_ptr), nbytes, C.byref(out))
    if rc != 0:
        raise RuntimeError(f"CSAMI_Adler32Device failed: {rc}")
    return int(out.value)
