#!/bin/bash
# Per-kernel time table of a bench step under rocprofv3 (--kernel-trace --stats only):
#   gpurun -- 'bash tools/prof_step.sh <tag> <config> [extra bench flags]'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=${1:-prof}; cfg=${2:-c2}; shift 2
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$cfg -o $cfg -- \
  python bench.py --config $cfg --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-probe --no-fit-loop "$@" > $out/bench_${cfg}_under_rocprof.json 2> $out/rocprof_$cfg.err
rm -f $out/stats_$cfg/*kernel_trace.csv $out/stats_$cfg/*agent_info.csv
# 5 warm-up + 3 x 20 timed graph replays + 10 launches each of the two roofline kernels (counted as fractions of a step)
python tools/kernel_breakdown.py $out/stats_$cfg/${cfg}_kernel_stats.csv 65 0.5
