#!/bin/bash
# Which values of a kernel does the compiler consider lane-varying?  Emits the device IR of one .hip file and
# runs LLVM's uniformity analysis over it; prints, per kernel whose mangled name matches $2, the number of
# divergent values / branches and the loops with a divergent exit.  A wave-uniform algorithm (one wavefront
# per search root, scalar position) should show none of its own state there.
#   tools/uniformity.sh open_spiel_amd/csrc/osg_mcts_wave.hip 'HexTILi3EEELb1ELb1'
set -e
src=${1:?source file}; pat=${2:-.}
dir=$(dirname "$src")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I"$dir" -Iinclude -S -emit-llvm \
  --cuda-device-only "$src" -o /tmp/uniformity.ll 2>/dev/null
/opt/rocm/lib/llvm/bin/opt -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -passes='print<uniformity>' -disable-output \
  /tmp/uniformity.ll 2> /tmp/uniformity.txt
awk -v pat="$pat" '
  /^UniformityInfo for function/ { if (name != "" && show) report(); name=$0; show = ($0 ~ pat); div=0; br=0; cyc="" ; next }
  /DIVERGENT:.* br / { br++ }
  /DIVERGENT:/ { div++ }
  /^  depth=/ { cyc = cyc "\n    " substr($0, 1, 60) }
  function report() { print name; print "  divergent values " div ", divergent branches " br; if (cyc != "") print "  cycles with divergent exit:" cyc }
  END { if (show) report() }' /tmp/uniformity.txt
