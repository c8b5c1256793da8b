"""CPU tier for the PRODUCT library: it loads, exports every symbol include/csc_mi355x.h declares,
its host-side pieces (props arithmetic, header I/O) agree with the reference
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
    # the decoder runs on the device too: same refusal
    from csc_amd.capi import BytesReader
    r = BytesReader(bytes(100))
    q = prod.props_init(1 << 20, 3)
    assert not prod.lib.CSCDec_Create(C.byref(q), C.cast(r.ptr(), C.c_void_p), None)


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


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_names_the_kernel_form_the_library_launches():
    """bench.py labels a split's dominant kernel from the same threshold launch_encode_runs_multi uses (round 5 advice)"""
    src = open(os.path.join(ROOT, "csc_amd", "csrc", "csc_kernels_dp4.inc")).read()
    k = int(re.search(r"constexpr uint32_t kD4MultiMax = (\d+);", src).group(1))
    assert _bench_module().D4_MULTI_MAX == k


def test_bench_summary_is_compact_and_carries_the_other_configs():
    """the `summary` object bench.py emits LAST: <= 600 characters, with configs[2] / configs[4] / -p8 / many-stream values"""
    b = _bench_module()
    line = {"value": 2.2, "roofline": {"frac": 1.1e-5}, "cpu_baseline": {"value": 8.0}, "bit_exact_vs_cpu_baseline": True,
            "other_configs": {"silesia.tar_m5_d256m": {"value": 0.57, "roofline": {"frac": 9e-6}, "cpu_baseline": {"value": 4.5}, "bit_exact_vs_cpu_baseline": True},
                              "mix5_m2_d1024m": {"value": 2.6, "roofline": {"frac": 3e-5}, "cpu_baseline": {"value": 8.0}, "bit_exact_vs_cpu_baseline": True}},
            "p8_on_one_gpu": {"value": 17.1, "roofline": {"frac": 9e-5}, "cpu_baseline": {"value": 51.6}, "bit_exact_vs_reference": True},
            "multi_stream": [{"streams": 127, "value": 255.0, "hbm_roofline_frac": 1.3e-3, "bit_exact_vs_reference_digest": True, "decode": {"value": 480.0, "roundtrip_ok": True}},
                             {"streams": 954, "value": 465.0, "hbm_roofline_frac": 2.4e-3, "bit_exact_vs_reference_digest": True, "decode": {"value": 670.0, "roundtrip_ok": True}}]}
    s = b.summary_of(line)
    assert len(json.dumps(s)) <= 600
    assert s["m5"] == [0.57, 9e-6, 4.5, True] and s["m2"][0] == 2.6 and s["p8"][0] == 17.1 and s["p954"][0] == 465.0 and s["p954_dec"] == [670.0, True]


def test_library_exports_only_its_api():
    """a drop-in for libcsc.a must not leak C++ symbols (kernel launch stubs, helpers) into the program it is linked into:
    the dynamic symbol table holds the C API -- CSCEnc* / CSCDec* / CSCEncProps_* (the reference's eleven), CSA_*, the CSCMI_* /
    CSAMI_* extensions -- and nothing else (csc_amd/csrc/exports.map)"""
    for lib in ("csc_amd/libcsc_mi355x.so", "tests/stage/libcsc_stage.so"):
        path = os.path.join(ROOT, lib)
        if not os.path.exists(path):
            continue
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        names = [l.split()[-1] for l in out.splitlines() if l.strip()]
        bad = [n for n in names if not re.match(r"^(CSCEnc|CSCDec|CSCEncProps_|CSCMI_|CSA_|CSAMI_|CSCST_)", n)]
        assert not bad, (lib, bad[:10])
        assert any(n == "CSCEnc_Encode" for n in names)
