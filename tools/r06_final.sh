python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r06z_gputest.log
./mega-minecraft_amd/mmgen_region_terrain_demo --bench > gpurun_out/r06z_streaming.json
python3 -c "
import json; j=json.load(open('gpurun_out/r06z_streaming.json')); print('walk ms/tick', j['device_resident']['walk']['ms_per_step'], 'load ms', j['device_resident']['initial_load']['ms'], 'host', j['host_chunks_packed_d2h']['walk']['ms_per_step'], 'nozone', j['device_resident_without_zone_cache']['walk']['ms_per_step'])"
./mega-minecraft_amd/mmgen_region_terrain_demo 0 0 | tail -4
R=$PWD; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tick -- $R/mega-minecraft_amd/mmgen_region_terrain_demo --bench > /dev/null 2>&1; cd $R
python tools/tick_trace.py gpurun_out/tick 20 2>&1 | tee gpurun_out/r06z_tick_trace.txt | tail -36; rm -rf gpurun_out/tick
tools/r06z.sh
MMGEN_BENCH_ONE_GPU_DRYRUN=1 python bench.py --gpus 8 --steps 2 --warmup 1 > gpurun_out/r06z_bench_dryrun8.json 2>/dev/null; python3 -c "
import json; j=json.loads([l for l in open('gpurun_out/r06z_bench_dryrun8.json') if l.startswith('{')][-1]); print('dryrun8', j.get('tiles_bit_exact'), j.get('chunks_bit_exact'))"
