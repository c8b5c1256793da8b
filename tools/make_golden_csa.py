#!/usr/bin/env python3
"""Generate tests/golden/csa.json from the REFERENCE ARCHIVER ITSELF (oracle/_ref/csarc_ref, built by
oracle/Makefile from /root/reference/src/archiver + libcsc with the zero-filling heap of
oracle/zero_heap.cpp).  Run in the development container only; what is committed is data: archive
sizes and digests (small archives in hex) for the seeded trees of tests/csa_cases.py.

  python tools/make_golden_csa.py
"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, csa_cases

REF = os.path.join(ROOT, "oracle", "_ref", "csarc_ref")
out = {}
for name in csa_cases.CSA_CASES:
    with tempfile.TemporaryDirectory() as td:
        content = csa_cases.make_tree(td, name)
        subprocess.run([REF] + csa_cases.csarc_argv(name), cwd=td, check=True, capture_output=True)
        arc = open(os.path.join(td, csa_cases.ARCNAME), "rb").read()
        t = subprocess.run([REF, "t", csa_cases.ARCNAME], cwd=td, capture_output=True)
        assert t.returncode == 0 and b"failed" not in t.stderr, name
        ent = {"archive_size": len(arc), "archive_sha256": cases.digest(arc),
               "index_pos": int.from_bytes(arc[8:16], "little"), "index_csize": int.from_bytes(arc[16:20], "little"),
               "index_rsize": int.from_bytes(arc[20:24], "little"),
               "raw_bytes": sum(len(v) for v in content.values())}
        if len(arc) <= 1500:
            ent["archive_hex"] = arc.hex()
        out[name] = ent
        print(name, ent["raw_bytes"], "->", len(arc), flush=True)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "csa.json"), "w"), indent=1, sort_keys=True)
