#!/bin/bash
# Builds tools/variants/libosg_<name>.so: the product library with osg_mcts_wave.hip (or the file named
# by SRC=) compiled with extra -D flags, for A/B measurements of kernel variants on the GPU box.
#   tools/build_variant.sh thr2 -DOSG_THR_MODE=2
set -e
cd "$(dirname "$0")/../open_spiel_amd/csrc"
name=$1; shift
src=${SRC:-osg_mcts_wave}
out=../../tools/variants
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value "$@" -c $src.hip -o $out/${src}_$name.o
objs=$(ls *.o | grep -v "^$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libosg_$name.so $objs $out/${src}_$name.o
rm -f $out/${src}_$name.o
echo built $out/libosg_$name.so
