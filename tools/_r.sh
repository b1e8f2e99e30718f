cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o b -- python bench.py --no-cpu-baseline > gpurun_out/bench_c2_under_rocprof.json 2> gpurun_out/rocprof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o b -- python bench.py --no-graph --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o b -- python bench.py --no-graph --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
find gpurun_out -name "*.csv" -newer tools/_r.sh | head -20
