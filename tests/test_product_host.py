"""CPU tier for the PRODUCT library: it loads, exports every symbol include/csc_mi355x.h declares,
its host-side pieces (props arithmetic, header I/O, the host decoder) agree with the reference
vectors -- and it refuses to encode without a GPU instead of falling back to anything."""
import ctypes as C
import json
import os
import re
import subprocess

import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
PROPS = json.load(open(os.path.join(G, "props.json")))


def test_exports_every_declared_symbol(prod):
    hdr = open(os.path.join(ROOT, "include", "csc_mi355x.h")).read()
    names = set(re.findall(r"^[A-Za-z_][\w \*]*?\b(CSC(?:Enc|Dec|EncProps|MI)_\w+)\s*\(", hdr, flags=re.M))
    assert {"CSCEncProps_Init", "CSCEnc_WriteProperties", "CSCEnc_EstMemUsage", "CSCEnc_Create", "CSCEnc_Destroy",
            "CSCEnc_Encode", "CSCEnc_Encode_Flush", "CSCDec_ReadProperties", "CSCDec_Create", "CSCDec_Destroy",
            "CSCDec_Decode"} <= names
    for n in names:
        assert hasattr(prod.lib, n), n


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "csc_mi355x.h"\nint main(void){CSCProps p; CSCEncProps_Init(&p, 1u<<20, 3); return sizeof(CSCProps)==40 ? 0 : 1;}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                    "-o", str(tmp_path / "t.o")], check=True)


def test_struct_layout():
    from csc_amd.capi import CSCProps
    assert C.sizeof(CSCProps) == 40          # 40 bytes on x86-64
    assert CSCProps.bt_size.offset == 20 and CSCProps.good_len.offset == 28 and CSCProps.EXEFilter.offset == 32


@pytest.mark.parametrize("key", sorted(PROPS))
def test_props_match_reference(prod, key):
    from csc_amd.capi import CSCProps
    d, lv = key.split("/")
    p = CSCProps()
    p.bt_cyc = 7
    prod.lib.CSCEncProps_Init(C.byref(p), int(d), int(lv))
    want = PROPS[key]
    assert p.as_dict() == want["props"]
    assert prod.est_mem_usage(p) == want["est_mem"]
    assert prod.write_properties(p).hex() == want["header_hex"]


@pytest.mark.parametrize("name,level", [("mix_types", 3), ("text_300k", 2), ("exe_300k", 5), ("delta_200k", 1),
                                        ("window_wrap_32k", 3), ("periodic_5000x200", 4), ("empty", 3), ("one_byte", 5)])
def test_host_decoder_on_oracle_streams(prod, orc, zalloc, name, level):
    spec, dict_size, clamp, max_read = cases.STREAM_CASES[name]
    data = cases.build(spec)
    rc, s = orc.encode(data, level, dict_size, alloc=zalloc, clamp_dict=clamp, max_read=max_read)
    assert rc == 0
    assert prod.decode(s) == (0, data)
    assert prod.decode(s, alloc=zalloc) == (0, data)                    # custom ISzAlloc
    # MemIO::ReadBlock issues ONE Read per payload and rejects a short one (csc_memio.cpp:47-50): same here
    assert prod.decode(s, max_read=1000) == orc.decode(s, alloc=zalloc, max_read=1000)


def test_decoder_error_paths(prod, orc, zalloc):
    from csc_amd.capi import BytesWriter
    data = cases.build(cases.STREAM_CASES["mix_types"][0])
    s = orc.encode(data, 3, 1 << 20, alloc=zalloc)[1]
    rc, out = prod.decode(s, writer=BytesWriter(fail_after=100000))
    assert rc == -97
    rc, out = prod.decode(s, writer=BytesWriter(abort_after=100000))       # CSC_WRITE_ABORT ends silently (csc_dec.cpp:768)
    assert rc == 0 and len(out) < len(data)
    assert prod.decode(s[:10] + b"\x00" * 50)[0] is None                    # garbage after the header: Create fails
    bad = bytearray(s); bad[0] = 0x7F                                       # dict_size > 1 GiB
    assert prod.decode(bytes(bad))[0] is None
    for cut in (len(s) - 3, len(s) // 2):
        assert prod.decode(s[:cut]) == orc.decode(s[:cut], alloc=zalloc)


def test_no_gpu_means_no_encoder(prod):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from csc_amd.capi import BytesWriter
    p = prod.props_init(1 << 20, 3)
    w = BytesWriter()
    h = prod.lib.CSCEnc_Create(C.byref(p), C.cast(w.ptr(), C.c_void_p), None)
    assert not h, "CSCEnc_Create must fail loudly without a HIP device -- there is no CPU fallback"
    assert prod.lib.CSCMI_DeviceCheck() < 0


def test_frozen_tables_match_oracle(orc):
    hdr = open(os.path.join(ROOT, "csc_amd", "csrc", "csc_tables.h")).read()
    def arr(name):
        body = hdr[hdr.index(name):]
        body = body[body.index("{") + 1:body.index("}")]
        return [int(x) for x in body.replace("\n", " ").split(",") if x.strip()]
    p2 = (C.c_uint32 * 512)(); lt = (C.c_uint32 * 513)()
    orc.lib.orc_tables(p2, lt)
    assert arr("kP2Bits[512]") == list(p2)
    assert arr("kLogTable[513]") == list(lt)
    assert list(p2[:3]) == [1280, 1077, 982] and list(lt[:3]) == [300, 458, 532] and lt[512] == 1300


def test_product_never_touches_the_oracle():
    """the product tree must not import, link or mention oracle/ (parity would be void otherwise)"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "csc_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".inc", ".h", ".c")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liborc" not in txt and "oracle/" not in txt.replace("oracle/_ref", "").replace("`oracle/`", ""), f
    out = subprocess.run(["ldd", os.path.join(ROOT, "csc_amd", "libcsc_mi355x.so")], capture_output=True, text=True).stdout
    assert "liborc" not in out and "libcsc_ref" not in out and "libamdhip64" in out
