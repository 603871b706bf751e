python -m pytest tests -x -q -m gpu 2>&1 | tail -3
./mega-minecraft_amd/mmgen_region_terrain_demo --bench | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('small units: walk ms/tick', j['device_resident']['walk']['ms_per_step'], 'load', j['device_resident']['initial_load']['ms'])"
MMGEN_CB_SMALL_MAX_CHUNKS=0 ./mega-minecraft_amd/mmgen_region_terrain_demo --bench | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('64-col units: walk ms/tick', j['device_resident']['walk']['ms_per_step'], 'load', j['device_resident']['initial_load']['ms'])"
./mega-minecraft_amd/mmgen_region_terrain_demo --bench | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('small units: walk ms/tick', j['device_resident']['walk']['ms_per_step'])"
MMGEN_CB_SMALL_MAX_CHUNKS=0 ./mega-minecraft_amd/mmgen_region_terrain_demo --bench | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('64-col units: walk ms/tick', j['device_resident']['walk']['ms_per_step'])"
