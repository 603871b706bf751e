#!/bin/bash
# same-box A/B of library builds: full bench (DAG headline + serial attribution), the libraries alternating, `rounds` times
# usage: tools/ab_brief.sh <rounds> lib1.so lib2.so ...      (env per run via AB_ENV="VAR=1")
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    echo "== $lib (round $r)"
    env $AB_ENV MMGEN_LIB=$lib python3 tools/bench_brief.py --steps 32 2>&1 | grep -v "^roofline"
  done
done
