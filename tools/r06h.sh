#!/bin/bash
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r06h_gputest.log
./mega-minecraft_amd/mmgen_region_terrain_demo --bench > gpurun_out/r06h_streaming.json
python3 -c "
import json; j=json.load(open('gpurun_out/r06h_streaming.json')); print('walk ms/tick', j['device_resident']['walk']['ms_per_step'], 'load ms', j['device_resident']['initial_load']['ms'], 'host', j['host_chunks_packed_d2h']['walk']['ms_per_step'], 'nozone', j['device_resident_without_zone_cache']['walk']['ms_per_step'])"
MMGEN_CAVE_WIDE_MAX_ROWS=0 ./mega-minecraft_amd/mmgen_region_terrain_demo --bench | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('narrow kernel: walk ms/tick', j['device_resident']['walk']['ms_per_step'])"
R=$PWD; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tick -- $R/mega-minecraft_amd/mmgen_region_terrain_demo --bench > /dev/null 2>&1; cd $R
python tools/tick_trace.py gpurun_out/tick 20 2>&1 | tee gpurun_out/r06h_tick_trace.txt | tail -36; rm -rf gpurun_out/tick
python bench.py --no-cpp-host --no-streaming --cpu-side 0 2>/dev/null | python3 -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench', j['ms_per_step'], json.dumps(j['baseline_configs'])[:1200])"
