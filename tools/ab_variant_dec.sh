#!/bin/bash
# Development aid: a variant of the product library from the WORKING TREE with extra compiler flags for the DECODER's code object (csc_dec_kernels.hip), e.g.
# -DDEC_AB_COPY2 / -DDEC_AB_TOKEN (the two-wavefront question, profiles/r06_dec_two_wave.md).  The other objects are taken from csc_amd/csrc/build as they are.
# tools/ab_variant_dec.sh <name> <flags...>  ->  csc_amd/csrc/build/ab/<name>.so   (then tools/gpu_ab2.py dec cur <name> ...)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root/csc_amd/csrc"
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c csc_dec_kernels.hip -o build/ab/${name}_d.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o build/ab/${name}.so build/csc_kernels.o build/csc_kernels_other.o build/csc_host.o build/ab/${name}_d.o build/csc_dec_device.o build/csa_kernels.o build/csa_archive.o -lpthread
echo "built $name ($*)"
