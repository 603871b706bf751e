#!/bin/bash
# per-kernel stats of the full-pipeline bench (32x32 tile) for each library given: rocprofv3 --kernel-trace --stats
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  export MMGEN_LIB=$root/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/fs_$tag -- python3 $root/bench.py --workload full --steps 3 --warmup 1 --no-kernel-events > $root/gpurun_out/fs_$tag.log 2>&1
  echo "== $lib"
  grep -o '"value": [0-9.]*' $root/gpurun_out/fs_$tag.log | head -1
  python3 - <<PY
import csv,glob
f=glob.glob("$root/gpurun_out/fs_$tag/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print(f"  {r['Name'].split('(')[0][:36]:36s} calls={r['Calls']:>4s} avg_ms={float(r['AverageNs'])/1e6:8.3f} pct={r['Percentage']}")
PY
done
