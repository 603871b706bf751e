#!/bin/bash
# SQ counter passes of the config-2 bench (one rocprofv3 --pmc run per line of counters; never combined with other trace domains).
# usage: tools/pmc_passes.sh <tag>   → gpurun_out/<tag>_pmcN/ ...
tag=${1:-pmc}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
i=0
while read -r counters; do
  [ -z "$counters" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $root/gpurun_out/${tag}_pmc$i -- python3 $root/bench.py --steps 3 --warmup 1 --cpu-sample 0 --full-extra 0 --no-kernel-events > $root/gpurun_out/${tag}_pmc$i.log 2>&1
  echo "pass $i ($counters): rc=$?"
done <<LIST
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU
SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE
LIST
python3 - <<PY
import csv,glob,collections,os
root="$root"; tag="$tag"
for d in sorted(glob.glob(f"{root}/gpurun_out/{tag}_pmc*/")):
    for f in glob.glob(d+"**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
        for k in acc:
            if not k.startswith("mm::k_") and "k_" not in k: continue
            print(k[:40], {c: round(v/n[(k,c)]) for c,v in acc[k].items()})
PY
