#!/bin/bash
# A/B timing of libmmgen builds with the same ABI: prints chunks/s and per-kernel ms per step of the full-pipeline bench for each library.
# usage: tools/ab_variants.sh [bench args --] lib1.so lib2.so ...      (default bench args: a 36x36 tile, 6 steps)
args="--tile-nx 36 --tile-nz 36 --steps 6 --warmup 1 --cpu-side 0"
if [[ "$*" == *" -- "* ]]; then args="${*%% -- *}"; set -- ${*#* -- }; fi
for lib in "$@"; do
  echo "== $lib"
  MMGEN_LIB=$lib python3 bench.py $args | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_ms'])"
done
