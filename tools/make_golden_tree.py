#!/usr/bin/env python3
"""Record what the REFERENCE archiver (oracle/_ref/csarc_ref = archiver/*.cpp + libcsc/*.cpp where they lie, zero-filling heap)
writes for the seeded many-task trees of csc_amd/treegen.py: `csarc_ref a -r -m3 -d64m -t1 out.csa t`, size + SHA-256 of the archive
into tests/golden/tree_workload.json.  CPU; "tree_small" takes seconds, "tree" (2.1 GB) several minutes.
python tools/make_golden_tree.py [tree_small] [tree]"""
import hashlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from csc_amd import treegen
path = os.path.join(ROOT, "tests", "golden", "tree_workload.json")
out = json.load(open(path)) if os.path.exists(path) else {"what": "csarc_ref a -r -m3 -d64m -t1 out.csa t over csc_amd/treegen.py trees", "level": 3, "dict": 64 << 20, "trees": {}}
for spec in sys.argv[1:] or ["tree_small"]:
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        tot = treegen.materialize(d, spec)
        t0 = time.time()
        subprocess.run([os.path.join(ROOT, "oracle", "_ref", "csarc_ref"), "a", "-r", "-m3", "-d64m", "-t1", "out.csa", "t"], cwd=d, check=True,
                       stdout=subprocess.DEVNULL)
        dt = time.time() - t0
        h = hashlib.sha256()
        with open(os.path.join(d, "out.csa"), "rb") as f:
            for blk in iter(lambda: f.read(1 << 22), b""):
                h.update(blk)
        out["trees"][spec] = {"files": len(treegen.files(spec)), "input_bytes": tot, "archive_bytes": os.path.getsize(os.path.join(d, "out.csa")),
                              "sha256": h.hexdigest(), "reference_seconds": round(dt, 1)}
        print(spec, out["trees"][spec], flush=True)
        json.dump(out, open(path, "w"), indent=1)
