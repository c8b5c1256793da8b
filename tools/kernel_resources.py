#!/usr/bin/env python3
"""Development aid (CPU only): the register / LDS / scratch budget of every __global__ in the shipped library, read out of its gfx950
code objects (llvm-objdump --offloading + llvm-readelf --notes) plus the number of scratch_* / v_readlane / v_writelane instructions in
each kernel's body (llvm-objdump -d).  python tools/kernel_resources.py [lib.so] > profiles/rNN_kernel_resources.md"""
import hashlib, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
lib = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.join(ROOT, "csc_amd", "libcsc_mi355x.so")
tmp = tempfile.mkdtemp()
try:
    work = os.path.join(tmp, os.path.basename(lib))
    shutil.copy(lib, work)
    subprocess.run([LLVM + "/llvm-objdump", "--offloading", work], check=True, capture_output=True)
    rows = []
    for f in sorted(os.listdir(tmp)):
        if "hipv4-amdgcn" not in f:
            continue
        path = os.path.join(tmp, f)
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", path], check=True, capture_output=True, text=True).stdout
        dis = subprocess.run([LLVM + "/llvm-objdump", "-d", path], check=True, capture_output=True, text=True).stdout
        body = {}
        cur = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1); body[cur] = [0, 0, 0, 0]; continue
            if cur is None:
                continue
            t = line.split()
            if len(t) < 1:
                continue
            op = t[0]
            if not re.match(r"^[a-z_0-9]+$", op):
                continue
            body[cur][0] += 1
            if op.startswith("scratch_"):
                body[cur][1] += 1
            if op in ("v_readlane_b32", "v_writelane_b32"):
                body[cur][2] += 1
            if op.startswith("s_cbranch") or op == "s_branch":
                body[cur][3] += 1
        for blk in notes.split("  - .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = g("name")
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("cscmi::", "")
            b = body.get(name, [0, 0, 0, 0])
            rows.append((dem, g("vgpr_count"), blk.split()[0], g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"),
                         g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("max_flat_workgroup_size"), b[0], b[1], b[2], b[3]))
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()
    print(f"# Kernel resources of `{os.path.relpath(lib, ROOT)}` (sha256 {sha[:16]}…)\n")
    print("Read out of the gfx950 code objects (`llvm-readelf --notes`; body counts from `llvm-objdump -d`).  `lane ops` = v_readlane + v_writelane")
    print("(SGPR spills live in VGPR lanes: each spill / reload is one of these).  LDS is the static part; dynamic LDS comes on top at launch.\n")
    print("| kernel | VGPR | AGPR | VGPR spills | SGPR | SGPR spills | scratch B/lane | static LDS B | max wg | instructions | scratch ops | lane ops | branches |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    seen = set()
    for r in sorted(rows, key=lambda r: r[0]):
        if r in seen:
            continue
        seen.add(r)
        print("| `" + r[0] + "` | " + " | ".join(str(x) for x in r[1:]) + " |")
finally:
    shutil.rmtree(tmp)
