#!/bin/bash
# streaming tick after the single-copy layout upload / single read-back: scheduler tests, the demo's bench line, the tick trace
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_schedulers.py tests/test_mesh.py -q -m gpu -x 2>&1 | tail -3
timeout 900 python3 -m pytest tests/test_gpu_features.py -q -m gpu -x -k "zone_cache or stage_dag or region_batched or flag_combinations or full_pipeline" 2>&1 | tail -3
for i in 1 2 3; do
./mega-minecraft_amd/mmgen_region_terrain_demo --bench > $out/r06t_streaming.json
python3 -c "
import json; j=json.load(open('$out/r06t_streaming.json')); print('walk ms/tick', j['device_resident']['walk']['ms_per_step'], 'load ms', j['device_resident']['initial_load']['ms'], 'host', j['host_chunks_packed_d2h']['walk']['ms_per_step'], 'nozone', j['device_resident_without_zone_cache']['walk']['ms_per_step'])"
done
R=$root; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $out/tick -- $R/mega-minecraft_amd/mmgen_region_terrain_demo --bench > /dev/null 2>&1; cd $R
python3 tools/tick_trace.py $out/tick 20 2>&1 | tee $out/r06t_tick_trace.txt | tail -40; rm -rf $out/tick
