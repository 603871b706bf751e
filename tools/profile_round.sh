#!/bin/bash
# Regenerates the rocprofv3 evidence of the full-pipeline bench under gpurun_out/<tag>_* (copy what is to be judged into profiles/):
#   <tag>_full_stats/            rocprofv3 --kernel-trace --stats of `bench.py` (the headline command, fewer steps)
#   <tag>_pmc_{sq,fetch,write}/  three separate --pmc passes of the same command (never combined with other trace domains)
#   <tag>_full_pmc.json          per-kernel per-launch averages + lib_sha16 of the library that ran;
#                                hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)
# usage: tools/profile_round.sh <tag> [stats|pmc|all] [extra bench args]
tag=${1:-r02a}
what=${2:-all}
shift $(( $# < 2 ? $# : 2 ))
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --steps 3 --warmup 1 --cpu-side 0 --serial --no-kernel-events --no-cpp-host --no-streaming --no-baseline-configs $*"      # serial schedule: one kernel at a time, counters and durations attribute cleanly
if [ "$what" = stats ] || [ "$what" = all ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_full_stats -- $B > $out/${tag}_full_stats.log 2>&1
  f=$(ls $out/${tag}_full_stats/*/*kernel_stats.csv | head -1); cp $f $out/${tag}_full_kernel_stats.csv
  python3 - <<PY
import csv
for r in list(csv.DictReader(open("$out/${tag}_full_kernel_stats.csv")))[:14]:
    print(f"  {r['Name'].split('(')[0][:40]:40s} calls={r['Calls']:>5s} avg_ms={float(r['AverageNs'])/1e6:9.4f} pct={r['Percentage']}")
PY
  grep -o '{"metric.*' $out/${tag}_full_stats.log | cut -c1-400
fi
if [ "$what" = pmc ] || [ "$what" = all ]; then
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_pmc_sq -- $B > $out/${tag}_pmc_sq.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/${tag}_pmc_sq2 -- $B > $out/${tag}_pmc_sq2.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU --output-format csv -d $out/${tag}_pmc_sq3 -- $B > $out/${tag}_pmc_sq3.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -- $B > $out/${tag}_pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -- $B > $out/${tag}_pmc_write.log 2>&1
  python3 - <<PY
import csv, glob, json, collections, hashlib
out, tag, root = "$out", "$tag", "$root"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in ("sq", "sq2", "sq3", "fetch", "write"):
    for f in glob.glob(f"{out}/{tag}_pmc_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mm::", "").replace("(anonymous namespace)::", "").split("<")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
res = {}
for k, cs in acc.items():
    if not k.startswith("k_"): continue
    e = {c: round(v / cnt[(k, c)]) for c, v in cs.items()}
    e["launches_seen"] = cnt[(k, "FETCH_SIZE")] or max(cnt[(k, c)] for c in cs)
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e: e["hbm_bytes"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
    if "GRBM_GUI_ACTIVE" in e:
        e["gpu_cycles"] = e["GRBM_GUI_ACTIVE"] // 8
        e["valu_insts_per_simd_cycle"] = round(e.get("SQ_INSTS_VALU", 0) / 1024 / max(e["gpu_cycles"], 1), 4)
    if e.get("SQ_THREAD_CYCLES_VALU") and e.get("SQ_ACTIVE_INST_VALU"):
        e["valu_active_lanes_avg"] = round(e["SQ_THREAD_CYCLES_VALU"] / e["SQ_ACTIVE_INST_VALU"], 2)      # of 64
    if e.get("SQC_ICACHE_REQ"):
        e["icache_miss_frac"] = round(e.get("SQC_ICACHE_MISSES", 0) / e["SQC_ICACHE_REQ"], 4)
    if e.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in e: e[c.lower() + "_frac"] = round(e[c] / e["SQ_WAVE_CYCLES"], 4)
    res[k] = e
sha = hashlib.sha256(open(f"{root}/mega-minecraft_amd/libmmgen.so", "rb").read()).hexdigest()[:16]
json.dump({"command": "rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- $B  (separate passes: SQ/GRBM, SQ wave states + LDS, FETCH_SIZE, WRITE_SIZE)",
           "lib_sha16": sha,
           "units": "per launch averages; FETCH_SIZE/WRITE_SIZE in KiB as reported; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section); gpu_cycles = GRBM_GUI_ACTIVE / 8 XCDs; valu_insts_per_simd_cycle = SQ_INSTS_VALU / 1024 SIMDs / gpu_cycles (peak 0.5 for wave64)",
           "kernels": res}, open(f"{out}/{tag}_full_pmc.json", "w"), indent=1)
for k, e in sorted(res.items(), key=lambda kv: -kv[1].get("gpu_cycles", 0))[:8]:
    print(k, {c: e[c] for c in ("gpu_cycles", "valu_insts_per_simd_cycle", "valu_active_lanes_avg", "icache_miss_frac", "hbm_bytes", "sq_wait_any_frac", "sq_wait_inst_any_frac", "sq_active_inst_any_frac") if c in e})
PY
fi
