#!/bin/bash
# repeats the GPU erosion tests with each library given: tools/dbg_ero_loop.sh <rounds> lib...
rounds=$1; shift
for lib in "$@"; do
  ok=0; bad=0
  for r in $(seq 1 $rounds); do
    if MMGEN_LIB=$lib timeout 300 python -m pytest tests -m gpu -x -q -k "erosion or relaxation or zones or config4 or frozen" > /tmp/ero_loop.log 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); grep "mmgen:" /tmp/ero_loop.log | head -4; fi
  done
  echo "== $lib: $ok passed, $bad failed"
done
