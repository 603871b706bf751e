#!/bin/bash
# round-6 final evidence on the frozen library: kernel stats + PMC passes of the serial bench (tile launches only), then the default bench
# (which attaches the VALU roofline from the PMC summary of the same library hash), serial / extras lines, the DAG trace, the GPU suite
# and the 8-rank rehearsal on one GPU
tools/profile_round.sh r06z all 2>&1 | tail -24 | tee gpurun_out/r06z_profile_round.log
cp gpurun_out/r06z_full_pmc.json profiles/r06z_full_pmc.json
python bench.py > gpurun_out/r06z_bench_default.json 2> gpurun_out/r06z_bench_default.err; tail -c 400 gpurun_out/r06z_bench_default.json
python bench.py > gpurun_out/r06z_bench_default_2.json 2>/dev/null
python bench.py --serial --no-cpp-host --no-streaming --cpu-side 0 > gpurun_out/r06z_bench_serial.json 2>/dev/null
python bench.py --extras --no-cpp-host --no-streaming --cpu-side 0 > gpurun_out/r06z_bench_extras.json 2>/dev/null
tools/trace_step.sh r06z_dag 45 > /dev/null 2>&1; cp gpurun_out/r06z_dag_trace.txt gpurun_out/r06z_trace_dag_kernels.txt
rm -rf gpurun_out/r06z_dag_trace gpurun_out/r06z_full_stats gpurun_out/r06z_pmc_*
