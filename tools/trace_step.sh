#!/bin/bash
# Kernel trace of a few DAG-schedule bench steps -> gpurun_out/<tag>_trace.txt (timeline of the last step: tools/trace_overlap.py)
# usage: tools/trace_step.sh <tag> [n_last_kernels] [extra bench args]
tag=${1:-t}; n=${2:-45}; shift 2
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_trace -- python3 $root/bench.py --steps 3 --warmup 1 --cpu-side 0 --no-kernel-events --no-cpp-host --no-streaming --no-baseline-configs "$@" > $out/${tag}_trace.log 2>&1
# (--no-cpp-host: bench.py would otherwise run mmgen_tiled_demo as a child, which the profiler traces too; the largest trace is the bench's own)
f=$(ls -S $out/${tag}_trace/*/*kernel_trace.csv | head -1)
# one steady-state step (between two k_heightfield launches; the checksum kernels behind the last step are torch's and not shown)
python3 $root/tools/trace_one_step.py $f 2 > $out/${tag}_trace.txt
python3 $root/tools/trace_overlap.py $f $n | tail -2 >> $out/${tag}_trace.txt
cat $out/${tag}_trace.txt
