#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/r04h
mkdir -p $out
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "finishing or partials or user_head or attpool" 2>&1 | tail -30 > $out/pytest_kernels.log
timeout 900 python -m pytest tests/test_nrms_model.py -m gpu -x -q 2>&1 | tail -30 > $out/pytest_model.log
timeout 900 python -m pytest tests/test_multi_rank_gpu.py tests/test_full_size_parity.py -m gpu -x -q -k "graph_collectives or two_rank_data_parallel or overlapped or c2_full or c1_full" 2>&1 | tail -30 > $out/pytest_multi.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg > $out/bench_c2.json 2> $out/bench_c2.err
for c in c1 c4 c5; do
  python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg > $out/bench_$c.json 2> $out/bench_$c.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c2 -o c2 -- \
  python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg --no-roofline > /dev/null 2> $out/rocprof_c2.err
rm -f $out/stats_c2/*kernel_trace.csv $out/stats_c2/*agent_info.csv
cat $out/pytest_kernels.log $out/pytest_model.log $out/pytest_multi.log | tail -60
python tools/show_bench.py $out 2>&1 | tail
