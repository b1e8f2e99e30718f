#!/bin/bash
# Per-config rocprofv3 kernel statistics of bench.py (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/profile_configs.sh <tag> c1 c2 c3 c4 c5'
# writes gpurun_out/<tag>/stats_<cfg>/<cfg>_kernel_stats.csv and the bench line printed under the profiler.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/$tag
for c in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stats_$c -o $c -- \
    python bench.py --config $c --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline \
    > gpurun_out/$tag/bench_${c}_under_rocprof.json 2> gpurun_out/$tag/rocprof_$c.err
  rm -f gpurun_out/$tag/stats_$c/*kernel_trace.csv gpurun_out/$tag/stats_$c/*agent_info.csv
done
ls -R gpurun_out/$tag | head -40
