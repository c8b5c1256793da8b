"""The serial loops must keep their scalars in scalar registers: one lane-dependent branch whose two sides meet in a block of such a
loop makes per-lane data of everything merged there ("one lane-dependent branch ...").  This runs LLVM's
uniformity analysis over the IR of the decode kernel and of the single-stream level-5 / level-2 encode kernels (CPU only: hipcc
cross-compiles, `opt` analyses) and pins what round 5 reached, so that an `if (lane == 0)` slipping back into a packet / node /
coder loop fails a test instead of costing 40 % unnoticed."""
import os, re, shutil, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")) or not os.path.exists("/opt/rocm/lib/llvm/bin/opt"),
                                reason="needs hipcc and opt")


def run(tool, *args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *args], capture_output=True, text=True, timeout=900).stdout
    res = {}
    for line in out.splitlines():
        m = re.match(r"(\S+): (\d+) cycles with a divergent exit(?: \((\d+) outermost\))?, phis divergent/uniform (\d+)/(\d+), divergent terminators (\d+)", line)
        if m:
            res[m.group(1)] = dict(cycles=int(m.group(2)), outer=int(m.group(3) or 0), dphi=int(m.group(4)), uphi=int(m.group(5)), dterm=int(m.group(6)))
    return res


def find(res, part):
    hits = [v for k, v in res.items() if part in k]
    assert hits, f"no function *{part}* in the analysis output: {sorted(res)}"
    return hits[0]


def test_decoder_packet_loops_are_uniform():
    r = run("dec_uniformity.py")
    fast = find(r, "dlz_fast")
    assert fast["cycles"] == 0 and fast["dterm"] == 0, fast          # no lane-dependent branch at all in the fast packet loop
    k = find(r, "k_decode_runEPNS")
    assert k["dphi"] * 3 < k["uphi"], k                              # round 4: 3 363 divergent / 641 uniform; now 524 / 3 116
    assert k["cycles"] <= 12, k                                      # (the set-up / write-back loops in front of and behind the state machine)


def test_encoder_coder_and_parser_loops_are_uniform():
    r = run("enc_uniformity.py")
    for name in ("bt_coder_nl", "hp_coder_nl"):
        c = find(r, name)
        assert c["cycles"] == 0 and c["dphi"] * 20 < c["uphi"], (name, c)
    p = find(r, "bt_parser_nl")
    assert p["outer"] == 0 and p["dphi"] * 4 < p["uphi"], p          # round 5, first half: 988 divergent / 197 uniform, 179 divergent loops
