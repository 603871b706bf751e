#!/bin/bash
# Builds an A/B variant of libmmgen with extra compiler flags / defines: tools/build_variant.sh <name> "<extra flags>"
# -> build_ab/libmmgen_<name>.so (same ABI; select it with MMGEN_LIB=...).  The objects are built under /tmp: only the library lands in
# build_ab/ (git-ignored; it travels to the GPU box with the snapshot, so keep it to the few libraries of the current experiment).
set -e
name=$1; extra=$2
root=$(cd $(dirname $0)/.. && pwd)
d=/tmp/mmgen_ab/$name
mkdir -p $d/pkg $root/build_ab
rm -rf $d/pkg/csrc && cp -r $root/mega-minecraft_amd/csrc $d/pkg/csrc && rm -f $d/pkg/csrc/*.o
rm -rf $d/include && cp -r $root/include $d/include
make -C $d/pkg/csrc -j8 ../libmmgen.so FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function $extra" 2>&1 | grep -E "error|warning: v|Error" || true
cp $d/pkg/libmmgen.so $root/build_ab/libmmgen_$name.so
echo built build_ab/libmmgen_$name.so
