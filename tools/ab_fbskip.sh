#!/bin/bash
# removal experiment on k_fill_base (the MM_FB_SKIP hooks live in commit b3c4048 only - their literals fail tests/test_ref_literals.py -: check it out,
# build the variants with tools/build_variant.sh <name> "-DMM_FB_SKIP=<bits>", run this; MM_FB_SKIP: 1 no list appends, 2 no layer search, 4 no biome draw, 8 no place_block_base at all, 16 no noise tables): its
# time alone (serial pass) and the step beside everything else.  The variants' blocks are wrong; only the times mean something.
#   tools/ab_fbskip.sh build_ab/libmmgen_fbskip1.so ...
for lib in mega-minecraft_amd/libmmgen.so "$@"; do
  echo "== $lib"
  MMGEN_LIB=$lib python3 tools/bench_brief.py --steps 16 --no-baseline-configs 2>&1 | python3 -c "
import sys,re
t=sys.stdin.read()
m=re.search(r'ms_per_step ([\d.]+)', t); fb=re.search(r'k_fill_base=([\d.]+)', t); fc=re.search(r'k_fill_cave=([\d.]+)', t)
print('  step', m and m.group(1), ' k_fill_base alone', fb and fb.group(1), ' k_fill_cave alone', fc and fc.group(1))"
done
