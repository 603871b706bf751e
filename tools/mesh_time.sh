#!/bin/bash
# kernel times of the mesher for each library given (rocprofv3 --kernel-trace --stats of the default bench, which meshes a 32x32 region)
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export MMGEN_LIB=$root/$lib
  rm -rf /tmp/ms; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ms -- python3 $root/bench.py --no-cpp-host --no-streaming --steps 2 --warmup 1 --cpu-side 0 --extras --tile-nx 36 --tile-nz 36 > /tmp/b.log 2>&1
  python3 - <<PY
import csv,glob
print("== $lib")
for r in csv.DictReader(open(glob.glob("/tmp/ms/*/*kernel_stats.csv")[0])):
    if "mesh" in r["Name"]: print("  ", r["Name"][:20], r["Calls"], round(float(r["AverageNs"])/1000,1), "us")
PY
done
