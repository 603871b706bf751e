#!/bin/bash
# C++-hosted rate (mmgen_tiled_demo, 64 x 128 tile) of several library builds on one box: tools/ab_demo.sh <rounds> build_ab/demo_x ...
# each directory holds mmgen_tiled_demo and the libmmgen.so it loads (rpath $ORIGIN)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for d in "$@"; do
    echo -n "$d: "; (cd $d && ./mmgen_tiled_demo --gpus 1 --tile 64 128 --steps 48 2>&1 | grep -o "world.*aggregate")
  done
done
