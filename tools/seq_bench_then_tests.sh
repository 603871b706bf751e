#!/bin/bash
for i in 1 2 3; do
  timeout 200 python bench.py > /dev/null 2>&1; timeout 120 python bench.py --serial --cpu-side 0 > /dev/null 2>&1; timeout 120 python bench.py --extras --cpu-side 0 > /dev/null 2>&1
  t0=$(date +%s)
  timeout 420 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|rror|Thread|File \"|Timeout" | tail -30
  echo "rep $i rc=${PIPESTATUS[0]} $(( $(date +%s) - t0 )) s"
done
