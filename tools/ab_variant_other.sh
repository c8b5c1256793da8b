#!/bin/bash
# Development aid: a variant of the product library from the WORKING TREE with extra compiler flags for the code object of the level-1/2/5 forms (-DCSCMI_TU=2:
# k_encode_runs_hp / _bt / the one-wavefront forms).  The other objects are taken from csc_amd/csrc/build as they are.
# tools/ab_variant_other.sh <name> <flags...>  ->  csc_amd/csrc/build/ab/<name>.so   (then tools/gpu_ab2.py m2,m5 cur <name> ...)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root/csc_amd/csrc"
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DCSCMI_TU=2 "$@" -c csc_kernels.hip -o build/ab/${name}_o.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o build/ab/${name}.so build/csc_kernels.o build/ab/${name}_o.o build/csc_host.o build/csc_dec_kernels.o build/csc_dec_device.o build/csa_kernels.o build/csa_archive.o -lpthread
echo "built $name ($*)"
