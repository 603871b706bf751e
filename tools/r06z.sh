#!/bin/bash
# round-6 final evidence on the frozen library
tools/profile_round.sh r06z all 2>&1 | tail -30 | tee gpurun_out/r06z_profile_round.log
python bench.py > gpurun_out/r06z_bench_default.json 2> gpurun_out/r06z_bench_default.err; tail -c 600 gpurun_out/r06z_bench_default.json
python bench.py --serial --no-cpp-host --no-streaming --cpu-side 0 > gpurun_out/r06z_bench_serial.json 2>/dev/null
python bench.py --extras --no-cpp-host --no-streaming --cpu-side 0 > gpurun_out/r06z_bench_extras.json 2>/dev/null
tools/trace_step.sh r06z_dag 45 > /dev/null 2>&1; cp gpurun_out/r06z_dag_trace.txt gpurun_out/r06z_trace_dag_kernels.txt; rm -rf gpurun_out/r06z_dag_trace gpurun_out/r06z_full_stats gpurun_out/r06z_pmc_*
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r06z_gputest.log
MMGEN_BENCH_ONE_GPU_DRYRUN=1 python bench.py --gpus 8 --steps 2 --warmup 1 --tile-nx 64 --tile-nz 128 > gpurun_out/r06z_bench_dryrun8.json 2> gpurun_out/r06z_bench_dryrun8.err; python3 -c "
import json; j=json.loads([l for l in open('gpurun_out/r06z_bench_dryrun8.json') if l.startswith('{')][-1]); print('dryrun8', j.get('tiles_bit_exact'), j.get('chunks_bit_exact'), j.get('dry_run'))"
