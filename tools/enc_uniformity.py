#!/usr/bin/env python3
"""Development aid (CPU only): LLVM's uniformity analysis over the role functions of ONE single-stream encode kernel
(binary-tree form + wide-bucket form: -DCSCMI_TU=2 -DCSCMI_DEV_ONE compiles nothing else, 25 s).  Printing the analysis of the
kernel bodies themselves takes opt tens of minutes, so the output is read as it comes and opt is stopped behind the last role
function.  See tools/dec_uniformity.py for why this matters.   python3 tools/enc_uniformity.py [-v] [function-name-part ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "csc_amd", "csrc", "csc_kernels.hip")
verbose = "-v" in sys.argv
TU = "1" if "-1" in sys.argv else "2"      # -1: the level-3 pipeline form (csc_kernels_dp4.inc) instead of the binary-tree / wide-bucket forms
want = [a for a in sys.argv[1:] if not a.startswith("-")] or ["_nl"]
with tempfile.TemporaryDirectory() as td:
    ll = os.path.join(td, "k.ll")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DCSCMI_TU=" + TU, "-DCSCMI_DEV_ONE", "-S", "-emit-llvm",
                    "--offload-device-only", SRC, "-o", ll], check=True, stderr=subprocess.DEVNULL)
    p = subprocess.Popen(["/opt/rocm/lib/llvm/bin/opt", "-passes=print<uniformity>", "-disable-output", ll], stderr=subprocess.PIPE, text=True)
    parts, cur = [], None
    for line in p.stderr:
        if line.startswith("UniformityInfo for function "):
            name = line.split("'")[1]
            if "k_encode_runs" in name: break
            cur = [name]; parts.append(cur)
        elif cur is not None: cur.append(line.rstrip("\n"))
    p.kill()
for part in parts:
    name, lines = part[0], part[1:]
    if not any(w in name for w in want): continue
    cyc = [l for l in lines if l.startswith("  depth=")]
    defs = {}
    for l in lines:
        m = re.match(r"\s*(DIVERGENT:)?\s*(%\d+) = (.*)", l)
        if m: defs[m.group(2)] = (bool(m.group(1)), m.group(3))
    dphi = sum(1 for d, r in defs.values() if d and r.startswith("phi"))
    uphi = sum(1 for d, r in defs.values() if not d and r.startswith("phi"))
    dterm = sum(1 for l in lines if "DIVERGENT:" in l and re.search(r"\bbr i1\b|\bswitch\b", l))
    print(f"{name}: {len(cyc)} cycles with a divergent exit ({sum(1 for c in cyc if c.startswith('  depth=1'))} outermost), phis divergent/uniform {dphi}/{uphi}, divergent terminators {dterm}")
    if verbose:
        for k, (d, rhs) in defs.items():
            if d and not any(defs.get(o, (False, ""))[0] for o in set(re.findall(r"%\d+", rhs))): print("    root", k, rhs[:140])
