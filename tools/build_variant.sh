#!/bin/bash
# build a variant of libtbg_hip.so with extra -D flags for ONE translation unit (the other objects come from the product build's cache)
# usage: tools/build_variant.sh <out.so> <unit.hip> [-DNAME=VALUE ...]
set -e
cd "$(dirname "$0")/.."
python -c "from textboxgan_amd.build import build_native; build_native(verbose=False)"
out=$1; unit=$2; shift 2
mkdir -p "$(dirname "$out")"
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c textboxgan_amd/csrc/$unit -o $tmp/unit.o
objs=$(ls textboxgan_amd/csrc/.obj/*.o | grep -v "/$unit\.")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" $objs $tmp/unit.o
rm -rf $tmp
echo "$out"
