#!/bin/bash
# A/B of the relaxation's plane loads (round 6, judge item 7): tools/ab_erode_loads.sh lib1.so lib2.so ...
# per library: the erosion parity tests, the bench (DAG headline + serial per-kernel times) twice alternating, FETCH / WRITE_SIZE of the serial run
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out; mkdir -p $out
for lib in "$@"; do
  n=$(basename $lib .so)
  echo "== $n: erosion parity tests"
  MMGEN_LIB=$root/$lib timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "erosion or relaxation or zone" -x 2>&1 | tail -3
done
tools/ab_brief.sh 2 "$@" 2>&1 | grep -v "^  *$" | cut -c1-900
for lib in "$@"; do
  n=$(basename $lib .so)
  echo "== $n: traffic"
  MMGEN_LIB=$root/$lib tools/pmc_traffic.sh ero_$n 2>&1 | grep -E "k_erode|k_cave_voxels "
  rm -rf $out/ero_${n}_pmc_fetch $out/ero_${n}_pmc_write
done
