#!/bin/bash
# A/B timing of libmmgen builds with the same ABI: prints chunks/s and per-kernel ms per step of the full-pipeline bench for each library.
# usage: [AB_ARGS="--tile-nx 64 --tile-nz 128 --steps 4 --warmup 1 --cpu-side 0"] tools/ab_variants.sh lib1.so lib2.so ...
args=${AB_ARGS:---tile-nx 36 --tile-nz 36 --steps 6 --warmup 1 --cpu-side 0}
for lib in "$@"; do
  echo "== $lib"
  MMGEN_LIB=$lib python3 bench.py $args | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_ms'])"
done
