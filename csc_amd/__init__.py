"""csc_amd -- MI355X-native libcsc encode/decode path (drop-in for fusiyuan2010/CSC's libcsc).

The product is `libcsc_mi355x.so` (HIP kernels for gfx950 + the reference's C ABI); this package is
its Python host-side mirror: `capi.CscLib` binds the C ABI with ctypes, `corpus` generates the
seeded synthetic corpora.  There is no CPU fallback: `load()` raises if the library is missing and
`CSCEnc_Create` returns NULL (loudly) without a HIP device.
"""
import os

from .capi import CscLib, CSCProps, BytesReader, BytesWriter  # noqa: F401

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcsc_mi355x.so")


def load() -> "CscLib":
    """Bind the product library (built in-tree by __graft_entry__.build()).

    PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64; two HIP runtimes in one
    process cannot both own the GPU.  Importing torch first lets the dynamic loader satisfy this
    library's `libamdhip64.so.7` dependency with the copy torch already mapped, so torch tensors,
    torch.distributed (RCCL) and these kernels share ONE runtime.  Stand-alone C/C++ callers simply
    get /opt/rocm's runtime."""
    try:
        import torch  # noqa: F401  (plumbing only: device memory, streams, torch.distributed)
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    return CscLib(LIB_PATH)
