#!/bin/bash
# Collects the round's judged artefacts on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1800 -- 'bash tools/collect_profiles.sh'
# then, back in the container:
#   python tools/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write c2 <tag>     # -> profiles/traffic.json
#   and copy gpurun_out/{bench_c2.json,bench_c2_under_rocprof.json,prof_stats/b_kernel_stats.csv} into profiles/.
# --pmc passes are separate runs with --kernel-trace only (no sys/hip/hsa trace domains), as the pool requires.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o b -- python bench.py \
  > gpurun_out/bench_c2_under_rocprof.json 2> gpurun_out/rocprof.err
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_$( [ $c = FETCH_SIZE ] && echo fetch || echo write )
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o b -- \
    python bench.py --no-graph --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
done
tail -c 400 gpurun_out/bench_c2.json
