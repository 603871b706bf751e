#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/<tag>_* (copy what is to be judged into profiles/):
#   <tag>_config2_stats   rocprofv3 --kernel-trace --stats of the default bench command (config 2)
#   <tag>_full_stats      same for --workload full
#   <tag>_pmc_{sq,fetch,write}   three separate --pmc passes of the config-2 bench (never combined with other trace domains)
#   <tag>_config2_pmc.json       per-kernel per-launch averages; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction)
tag=${1:-r01g}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_config2_stats -- python3 $root/bench.py --steps 5 --warmup 1 --cpu-sample 0 --full-extra 0 > $out/${tag}_config2_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_full_stats -- python3 $root/bench.py --workload full --steps 3 --warmup 1 --no-kernel-events > $out/${tag}_full_stats.log 2>&1
B="python3 $root/bench.py --steps 3 --warmup 1 --cpu-sample 0 --full-extra 0 --no-kernel-events"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_pmc_sq -- $B > $out/${tag}_pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -- $B > $out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -- $B > $out/${tag}_pmc_write.log 2>&1
BF="python3 $root/bench.py --workload full --steps 2 --warmup 1 --no-kernel-events"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_full_pmc_fetch -- $BF > $out/${tag}_full_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_full_pmc_write -- $BF > $out/${tag}_full_pmc_write.log 2>&1
python3 - <<PY
import csv, glob, json, collections
out, tag = "$out", "$tag"
# full pipeline: HBM bytes per kernel launch (32x32-chunk tile)
facc = collections.defaultdict(lambda: collections.defaultdict(float)); fcnt = collections.Counter()
for d in ("fetch", "write"):
    for f in glob.glob(f"{out}/{tag}_full_pmc_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mm::", "").split("<")[0]
            facc[k][r["Counter_Name"]] += float(r["Counter_Value"]); fcnt[(k, r["Counter_Name"])] += 1
fres = {}
for k, cs in facc.items():
    if not k.startswith("k_"): continue
    e = {c: round(v / fcnt[(k, c)]) for c, v in cs.items()}
    e["hbm_bytes"] = (2 * e.get("FETCH_SIZE", 0) + e.get("WRITE_SIZE", 0)) * 1024
    fres[k] = e
json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --workload full --steps 2 --warmup 1 --no-kernel-events (two separate passes)",
           "workload": "full pipeline, 32x32-chunk tile (38x38 ring-extended grid for caves / placements)",
           "units": "per launch averages, KiB as reported; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024", "kernels": fres}, open(f"{out}/{tag}_full_pmc.json", "w"), indent=1)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in ("sq", "fetch", "write"):
    for f in glob.glob(f"{out}/{tag}_pmc_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mm::", "").split("<")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
res = {}
for k, cs in acc.items():
    if not k.startswith("k_"): continue
    e = {c: round(v / cnt[(k, c)]) for c, v in cs.items()}
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e: e["hbm_bytes"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
    if "GRBM_GUI_ACTIVE" in e:
        e["gpu_cycles"] = e["GRBM_GUI_ACTIVE"] // 8
        e["valu_insts_per_simd_cycle"] = round(e.get("SQ_INSTS_VALU", 0) / 1024 / max(e["gpu_cycles"], 1), 4)
    res[k] = e
json.dump({"command": "rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --full-extra 0 --no-kernel-events (three separate passes: SQ/GRBM, FETCH_SIZE, WRITE_SIZE)",
           "workload": "config2, 256 chunks per launch",
           "units": "per launch averages; FETCH_SIZE/WRITE_SIZE in KiB as reported; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md §HBM); gpu_cycles = GRBM_GUI_ACTIVE / 8 XCDs",
           "kernels": res}, open(f"{out}/{tag}_config2_pmc.json", "w"), indent=1)
for k, e in res.items(): print(k, e)
PY
for t in config2 full; do f=$(ls $out/${tag}_${t}_stats/*/*kernel_stats.csv | head -1); echo "== $t"; head -8 $f | cut -c1-60,150-260; done
grep -o '{"metric.*' $out/${tag}_config2_stats.log | cut -c1-300
grep -o '{"metric.*' $out/${tag}_full_stats.log | cut -c1-200
