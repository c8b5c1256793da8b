"""Seeded inputs shared by tools/make_golden.py (which records what the REFERENCE produces for
them) and the parity tests (which replay them through the oracle and the HIP path)."""
import hashlib

from csc_amd import corpus


def _c(kind, seed, off, n):
    return corpus.fill(kind, seed, off, n).tobytes()


def build(spec):
    """spec: list of parts; a part is ["zeros", n] | ["pattern", hex, repeat] | [kind, seed, offset, n]"""
    out = bytearray()
    for part in spec:
        if part[0] == "zeros":
            out += bytes(part[1])
        elif part[0] == "pattern":
            out += bytes.fromhex(part[1]) * part[2]
        else:
            out += _c(part[0], part[1], part[2], part[3])
    return bytes(out)


LEVELS = (1, 2, 3, 4, 5)

# name -> (input spec, dict_size, clamp_dict, max_read)
STREAM_CASES = {
    "empty": ([], 1 << 20, True, None),
    "one_byte": ([["pattern", "78", 1]], 1 << 20, True, None),
    "zeros_8k": ([["zeros", 8192]], 1 << 20, True, None),
    "abcdefgh_64k": ([["pattern", "6162636465666768", 8192]], 1 << 20, True, None),
    "random_64k": ([["random", 11, 0, 65536]], 1 << 20, True, None),
    "text_20k": ([["text", 1, 0, 20000]], 1 << 20, True, None),
    "text_300k": ([["text", 1, 0, 300000]], 1 << 20, True, None),
    "exe_300k": ([["exe", 2, 0, 300000]], 1 << 20, True, None),
    "delta_200k": ([["delta", 3, 0, 200000]], 1 << 20, True, None),
    "entropy8_100k": ([["entropy8", 5, 0, 100000]], 1 << 20, True, None),
    "mix_types": ([["text", 1, 0, 100000], ["random", 4, 0, 30000], ["exe", 2, 0, 100000], ["delta", 3, 0, 70000],
                   ["entropy8", 5, 0, 40000], ["text", 1, 500000, 50000], ["random", 4, 90000, 100]], 1 << 20, True, None),
    "dup_blocks": ([["random", 21, 0, 40000], ["text", 1, 0, 30000], ["random", 21, 0, 40000], ["entropy8", 5, 0, 20000],
                    ["random", 21, 8192, 16384]], 1 << 20, True, None),
    "ragged_tail_511": ([["text", 9, 0, 8192 * 3 + 511]], 1 << 20, True, None),
    "short_reads_8191": ([["text", 7, 0, 200000], ["exe", 8, 0, 100000]], 1 << 20, True, 8191),
    "short_reads_511": ([["text", 7, 0, 40000]], 1 << 20, True, 511),
    "window_wrap_32k": ([["text", 7, 0, 600000], ["exe", 8, 0, 300000]], 32768, False, None),
    "window_wrap_100k": ([["text", 7, 0, 1500000], ["exe", 8, 0, 700000], ["text", 9, 0, 500000]], 100000, False, None),
    "periodic_5000x200": ([["text", 11, 0, 5000]] * 200, 1 << 22, True, None),
    "zeros_5m": ([["zeros", 5000000]], 1 << 22, True, None),
    "silesia_like_3m": ([["silesia", 6, 0, 3 << 20]], 16 << 20, True, None),
    "text_4m_d16m": ([["text", 40, 0, 4 << 20]], 16 << 20, True, None),
}
# the slow ones only at the levels of the BASELINE configs
HEAVY = {"window_wrap_100k": (2, 3, 5), "silesia_like_3m": (3, 5), "text_4m_d16m": (3, 5), "zeros_5m": (1, 3, 5)}


def levels_for(name):
    return HEAVY.get(name, LEVELS)


def digest(b):
    return hashlib.sha256(b).hexdigest()
