#!/bin/bash
# round-6 batch f: suite on the new library, counter-probe / cave-fill-shape A/B, slices, streaming
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r06f_gputest.log
tools/ab_brief.sh 2 mega-minecraft_amd/libmmgen.so build_ab/libmmgen_oldprobe.so build_ab/libmmgen_fc512_nof1.so 2>&1 | tee gpurun_out/r06f_ab_probe.txt | grep -E "^==|ms_per_step|k_cave_voxels"
for lib in mega-minecraft_amd/libmmgen.so build_ab/libmmgen_fc512_nof1.so; do
  for sl in 2 3 4; do
    echo "== $lib --slices $sl"
    MMGEN_LIB=$lib python3 bench.py --cpu-side 0 --no-cpp-host --no-streaming --no-kernel-events --no-baseline-configs --steps 32 --slices $sl 2>/dev/null | grep -o '{"metric.*' | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('  mean', j['ms_per_step'], 'median', j['ms_per_step_median'], 'value', j['value'], j.get('tiles_bit_exact'))"
  done
done 2>&1 | tee gpurun_out/r06f_slices.txt
./mega-minecraft_amd/mmgen_region_terrain_demo --bench > gpurun_out/r06f_streaming.json; python3 -c "
import json; j=json.load(open('gpurun_out/r06f_streaming.json')); print('walk ms/tick', j['device_resident']['walk']['ms_per_step'], 'load ms', j['device_resident']['initial_load']['ms'], 'host', j['host_chunks_packed_d2h']['walk']['ms_per_step'])"
