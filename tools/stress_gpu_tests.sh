#!/bin/bash
# Repeats GPU tests under a per-iteration timeout to expose intermittent stalls.
# usage: tools/stress_gpu_tests.sh <iterations> <timeout_s> <pytest -k expression | ALL> [MMGEN_LIB]
n=$1; to=$2; k=$3; lib=$4
for i in $(seq 1 $n); do
  t0=$(date +%s.%N)
  if [ "$k" = ALL ]; then
    env ${lib:+MMGEN_LIB=$lib} timeout $to python -m pytest tests -m gpu -x -q --durations=3 2>&1 | grep -E "passed|failed|rror|s call|Thread|File|Timeout" | tail -40 | tr '\n' ' '
  else
    env ${lib:+MMGEN_LIB=$lib} timeout $to python -m pytest tests -m gpu -x -q -k "$k" 2>&1 | tail -1 | tr '\n' ' '
  fi
  rc=${PIPESTATUS[0]}
  t1=$(date +%s.%N)
  echo " | iteration $i rc=$rc $(python3 -c "print(round($t1 - $t0, 1))") s"
done
