#!/bin/bash
# A/B timing of libmmgen builds with the same ABI: prints per-kernel ms of the config-2 bench for each library given.
for lib in "$@"; do
  echo "== $lib"
  MMGEN_LIB=$lib python3 bench.py --steps 8 --warmup 2 --cpu-sample 0 --full-extra 0 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['kernels_ms'])"
done
