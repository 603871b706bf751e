#!/bin/bash
# per-step spread of the DAG headline for each library: tools/ab_steps.sh <rounds> lib...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    MMGEN_LIB=$lib python3 bench.py --cpu-side 0 --no-cpp-host --no-streaming --no-kernel-events --steps 48 2>/dev/null | grep -o '{"metric.*' | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$lib', 'mean', j['ms_per_step'], 'min', j['ms_per_step_min'], 'median', j['ms_per_step_median'], 'max', j['ms_per_step_max'])"
  done
done
