#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every kernel of the serial bench (two separate --pmc passes): tools/pmc_traffic.sh <tag>
tag=${1:-t}
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --steps 3 --warmup 1 --cpu-side 0 --serial --no-kernel-events --no-cpp-host --no-streaming --no-baseline-configs"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -- $B > $out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -- $B > $out/${tag}_pmc_write.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in ("fetch", "write"):
    for f in glob.glob("$out/${tag}_pmc_%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mm::", "").split("<")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, cs in sorted(acc.items()):
    if not k.startswith("k_"): continue
    f = cs["FETCH_SIZE"] / max(cnt[(k, "FETCH_SIZE")], 1); w = cs["WRITE_SIZE"] / max(cnt[(k, "WRITE_SIZE")], 1)
    print(f"{k:26s} fetch {2 * f * 1024 / 1e9:7.3f} GB  write {w * 1024 / 1e9:7.3f} GB  hbm_bytes {(2 * f + w) * 1024 / 1e9:7.3f} GB   (per launch; 2 x FETCH_SIZE: gfx950 correction)")
PY
