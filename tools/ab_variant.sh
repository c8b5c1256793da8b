#!/bin/bash
# Development aid: a variant of the product library from the WORKING TREE with extra compiler flags for the level-3 code object (-DCSCMI_TU=1), e.g. the
# sensitivity pads of csc_kernels_dp4.inc (-DD5_PAD_SPINE=64: that many idle cycles in every straight-line step of the spine).  The other objects are taken
# from csc_amd/csrc/build as they are.  tools/ab_variant.sh <name> <flags...>  ->  csc_amd/csrc/build/ab/<name>.so   (then tools/gpu_ab2.py m3 cur <name> ...)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root/csc_amd/csrc"
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DCSCMI_TU=1 "$@" -c csc_kernels.hip -o build/ab/${name}_k.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o build/ab/${name}.so build/ab/${name}_k.o build/csc_kernels_other.o build/csc_host.o build/csc_dec_kernels.o build/csc_dec_device.o build/csa_kernels.o build/csa_archive.o -lpthread
echo "built $name ($*)"
