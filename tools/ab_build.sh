#!/bin/bash
# Development aid: build the product library of another git ref into build/ab/<name>.so, to compare two versions on the SAME GPU box
# (boxes differ by a few per cent).  tools/ab_build.sh <ref> <name>;  then gpurun -- python tools/gpu_ab.py <name> ...
set -e
ref=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$ref" csc_amd/csrc include | tar -x -C "$tmp"
make -C "$tmp/csc_amd/csrc" -j4 ../libcsc_mi355x.so >/dev/null 2>&1
mkdir -p "$root/csc_amd/csrc/build/ab"
cp "$tmp/csc_amd/libcsc_mi355x.so" "$root/csc_amd/csrc/build/ab/$name.so"
rm -rf "$tmp"
echo "built $name from $ref"
