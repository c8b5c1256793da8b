#!/usr/bin/env python3
"""Development aid (CPU only): what LLVM's uniformity analysis says about the decode kernel.

The decoder's speed hangs on its stream context living in SCALAR registers; one lane-dependent branch whose two sides meet in
a block of the packet loop turns every phi of that block -- i.e. the context -- into per-lane data (csc_dec_kernels.hip, DNOINL).
This compiles the kernel to LLVM IR for gfx950, runs `opt -passes=print<uniformity>` and prints, per kernel: cycles with a
divergent exit, divergent / uniform phi counts, and the divergent phis that are ROOTS (no divergent operand: made divergent by
control flow) with the lane-dependent branches found in the IR.   python3 tools/dec_uniformity.py [-v]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "csc_amd", "csrc", next((a for a in sys.argv[1:] if a.endswith(".hip")), "csc_dec_kernels.hip"))
LLVM = "/opt/rocm/lib/llvm/bin"
verbose = "-v" in sys.argv
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
with tempfile.TemporaryDirectory() as td:
    ll = os.path.join(td, "dec.ll")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "-emit-llvm", "--offload-device-only", *extra, SRC, "-o", ll],
                   check=True, stderr=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(LLVM, "opt"), "-passes=print<uniformity>", "-disable-output", ll], stderr=subprocess.PIPE, text=True)
    txt = r.stderr
for part in txt.split("UniformityInfo for function ")[1:]:
    name = part.split("'")[1]
    lines = part.split("\n")
    cyc = [l for l in lines if l.startswith("  depth=")]
    defs = {}
    for l in lines:
        m = re.match(r"\s*(DIVERGENT:)?\s*(%\d+) = (.*)", l)
        if m: defs[m.group(2)] = (bool(m.group(1)), m.group(3))
    dphi = sum(1 for d, r_ in defs.values() if d and r_.startswith("phi"))
    uphi = sum(1 for d, r_ in defs.values() if not d and r_.startswith("phi"))
    dterm = sum(1 for l in lines if "DIVERGENT:" in l and re.search(r"\bbr i1\b|\bswitch\b", l))
    print(f"{name}: {len(cyc)} cycles with a divergent exit, phis divergent/uniform {dphi}/{uphi}, divergent terminators {dterm}")
    if verbose:
        for c in cyc: print("   ", c[:160])
        for k, (d, rhs) in defs.items():
            if d and not any(defs.get(o, (False, ""))[0] for o in set(re.findall(r"%\d+", rhs))): print("    root", k, rhs[:140])
