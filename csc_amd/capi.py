"""ctypes view of the libcsc C boundary (CSCEnc_* / CSCDec_* + ISeq*Stream).

This is the Python host-side mirror of the reference's public interface
(`/root/reference/src/libcsc/csc_enc.h:11-30`, `csc_dec.h:8-21`,
`csc_common.h:19-63`, `Types.h:137-154,220-231`): same names, same argument
meaning, same error codes.  `CscLib` binds ANY shared library exporting that
ABI -- the product library `libcsc_mi355x.so` by default; the tests also point
it at the checkers under `oracle/` to compare the three byte for byte.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional

CSC_PROP_SIZE = 10
DECODE_ERROR = -96
WRITE_ERROR = -97
READ_ERROR = -98
CSC_WRITE_ABORT = C.c_size_t(-1).value


class CSCProps(C.Structure):
    """`csc_common.h:19-63` -- layout-identical (40 bytes on x86-64)."""
    _fields_ = [
        ("dict_size", C.c_size_t),
        ("csc_blocksize", C.c_uint32),
        ("raw_blocksize", C.c_uint32),
        ("hash_bits", C.c_uint8),
        ("hash_width", C.c_uint8),
        ("bt_hash_bits", C.c_uint8),
        ("bt_size", C.c_uint32),
        ("bt_cyc", C.c_uint32),
        ("good_len", C.c_uint8),
        ("lz_mode", C.c_uint8),
        ("DLTFilter", C.c_uint8),
        ("TXTFilter", C.c_uint8),
        ("EXEFilter", C.c_uint8),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


READ_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t))
WRITE_FN = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t)
PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64)
ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
FREE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)


class ISeqInStream(C.Structure):
    _fields_ = [("Read", READ_FN)]


class ISeqOutStream(C.Structure):
    _fields_ = [("Write", WRITE_FN)]


class ICompressProgress(C.Structure):
    _fields_ = [("Progress", PROGRESS_FN)]


class ISzAlloc(C.Structure):
    _fields_ = [("Alloc", ALLOC_FN), ("Free", FREE_FN)]


class BytesReader:
    """An `ISeqInStream` over a bytes-like object.  `max_read` forces short reads
    (the reference turns every Read into one chunk: `csc_enc.cpp:170-171`)."""

    def __init__(self, data, max_read: Optional[int] = None, fail_at: Optional[int] = None):
        self.data = memoryview(data).cast("B") if not isinstance(data, memoryview) else data
        self.pos = 0
        self.max_read = max_read
        self.fail_at = fail_at
        self.calls = []
        self._fn = READ_FN(self._read)
        self.stream = ISeqInStream(self._fn)

    def _read(self, p, buf, psize):
        want = psize[0]
        self.calls.append(want)
        if self.fail_at is not None and self.pos >= self.fail_at:
            return -1
        n = min(want, len(self.data) - self.pos)
        if self.max_read is not None:
            n = min(n, self.max_read)
        if n:
            C.memmove(buf, (C.c_char * n).from_buffer(self._slice(n)), n)
        self.pos += n
        psize[0] = n
        return 0

    def _slice(self, n):
        return bytearray(self.data[self.pos:self.pos + n])

    def ptr(self):
        return C.byref(self.stream)


class BytesWriter:
    """An `ISeqOutStream` collecting into a bytearray.  `fail_after` makes Write
    return short after that many bytes (-> WRITE_ERROR -97); `abort_after` makes
    it return CSC_WRITE_ABORT (`csc_dec.cpp:768-769`)."""

    def __init__(self, fail_after: Optional[int] = None, abort_after: Optional[int] = None):
        self.out = bytearray()
        self.fail_after = fail_after
        self.abort_after = abort_after
        self.sizes = []
        self._fn = WRITE_FN(self._write)
        self.stream = ISeqOutStream(self._fn)

    def _write(self, p, buf, size):
        self.sizes.append(size)
        if self.fail_after is not None and len(self.out) + size > self.fail_after:
            return 0
        if self.abort_after is not None and len(self.out) + size > self.abort_after:
            return CSC_WRITE_ABORT
        self.out += C.string_at(buf, size)
        return size

    def ptr(self):
        return C.byref(self.stream)


class CscLib:
    """Binds one shared library exporting the libcsc C ABI."""

    SYMBOLS = [
        "CSCEncProps_Init", "CSCEnc_WriteProperties", "CSCEnc_EstMemUsage", "CSCEnc_Create",
        "CSCEnc_Destroy", "CSCEnc_Encode", "CSCEnc_Encode_Flush", "CSCDec_ReadProperties",
        "CSCDec_Create", "CSCDec_Destroy", "CSCDec_Decode",
    ]

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} is missing -- build it first (python -c 'import __graft_entry__ as g; g.build()')")
        self.path = path
        self.lib = L = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_NOW", 2))
        L.CSCEncProps_Init.argtypes = [C.POINTER(CSCProps), C.c_uint32, C.c_int]
        L.CSCEncProps_Init.restype = None
        L.CSCEnc_WriteProperties.argtypes = [C.POINTER(CSCProps), C.POINTER(C.c_uint8), C.c_int]
        L.CSCEnc_WriteProperties.restype = None
        L.CSCEnc_EstMemUsage.argtypes = [C.POINTER(CSCProps)]
        L.CSCEnc_EstMemUsage.restype = C.c_uint64
        L.CSCEnc_Create.argtypes = [C.POINTER(CSCProps), C.c_void_p, C.c_void_p]
        L.CSCEnc_Create.restype = C.c_void_p
        L.CSCEnc_Destroy.argtypes = [C.c_void_p]
        L.CSCEnc_Destroy.restype = None
        L.CSCEnc_Encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.CSCEnc_Encode.restype = C.c_int
        L.CSCEnc_Encode_Flush.argtypes = [C.c_void_p]
        L.CSCEnc_Encode_Flush.restype = C.c_int
        L.CSCDec_ReadProperties.argtypes = [C.POINTER(CSCProps), C.POINTER(C.c_uint8)]
        L.CSCDec_ReadProperties.restype = None
        L.CSCDec_Create.argtypes = [C.POINTER(CSCProps), C.c_void_p, C.c_void_p]
        L.CSCDec_Create.restype = C.c_void_p
        L.CSCDec_Destroy.argtypes = [C.c_void_p]
        L.CSCDec_Destroy.restype = None
        L.CSCDec_Decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.CSCDec_Decode.restype = C.c_int

    # -- thin mirrors ---------------------------------------------------
    def props_init(self, dict_size: int, level: int) -> CSCProps:
        p = CSCProps()
        self.lib.CSCEncProps_Init(C.byref(p), dict_size & 0xFFFFFFFF, level)
        return p

    def write_properties(self, props: CSCProps) -> bytes:
        buf = (C.c_uint8 * CSC_PROP_SIZE)()
        self.lib.CSCEnc_WriteProperties(C.byref(props), buf, 0)
        return bytes(buf)

    def read_properties(self, head: bytes) -> CSCProps:
        p = CSCProps()
        buf = (C.c_uint8 * CSC_PROP_SIZE).from_buffer_copy(head[:CSC_PROP_SIZE])
        self.lib.CSCDec_ReadProperties(C.byref(p), buf)
        return p

    def est_mem_usage(self, props: CSCProps) -> int:
        return int(self.lib.CSCEnc_EstMemUsage(C.byref(props)))

    # -- whole-buffer helpers in the caller order of csa_worker.cpp:35-50 ---
    def encode(self, data, level: int = 2, dict_size: int = 64000000, *, props: Optional[CSCProps] = None,
               alloc=None, max_read: Optional[int] = None, writer: Optional[BytesWriter] = None,
               progress: Optional[Callable[[int, int], None]] = None, clamp_dict: bool = True,
               reader: Optional[BytesReader] = None, after_create: Optional[Callable[[int], None]] = None):
        """props -> Create -> caller writes the 10-byte header -> Encode -> Flush -> Destroy.
        Returns (rc, stream_bytes).  `clamp_dict` mirrors csa_worker.cpp:35 / csc.cpp:133-134."""
        n = len(data)
        if props is None:
            d = min(dict_size, n) if clamp_dict else dict_size
            props = self.props_init(d, level)
        w = writer or BytesWriter()
        r = reader or BytesReader(data, max_read=max_read)
        h = self.lib.CSCEnc_Create(C.byref(props), C.cast(w.ptr(), C.c_void_p), alloc)
        if not h:
            raise MemoryError("CSCEnc_Create returned NULL")
        w.out += self.write_properties(props)
        if after_create is not None:
            after_create(h)          # (tests: a development hook of a test-only library gets the handle before the first byte)
        prog = None
        pfn = None
        if progress is not None:
            pfn = PROGRESS_FN(lambda p, a, b: (progress(a, b), 0)[1])
            prog = ICompressProgress(pfn)
        rc = self.lib.CSCEnc_Encode(h, C.cast(r.ptr(), C.c_void_p), C.byref(prog) if prog else None)
        rc2 = self.lib.CSCEnc_Encode_Flush(h)
        self.lib.CSCEnc_Destroy(h)
        return (rc if rc < 0 else rc2), bytes(w.out)

    def decode(self, stream: bytes, *, alloc=None, writer: Optional[BytesWriter] = None,
               max_read: Optional[int] = None):
        """caller reads the 10-byte header -> ReadProperties -> Create -> Decode -> Destroy
        (csa_worker.cpp:75-83).  Returns (rc, raw_bytes); rc None if Create failed."""
        props = self.read_properties(stream[:CSC_PROP_SIZE])
        r = BytesReader(stream[CSC_PROP_SIZE:], max_read=max_read)
        w = writer or BytesWriter()
        h = self.lib.CSCDec_Create(C.byref(props), C.cast(r.ptr(), C.c_void_p), alloc)
        if not h:
            return None, b""
        rc = self.lib.CSCDec_Decode(h, C.cast(w.ptr(), C.c_void_p), None)
        self.lib.CSCDec_Destroy(h)
        return rc, bytes(w.out)
