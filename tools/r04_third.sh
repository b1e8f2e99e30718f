#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/r04g
mkdir -p $out
timeout 900 python -m pytest tests/test_multi_rank_gpu.py tests/test_rccl_single_rank_gpu.py -m gpu -x -q -k "rank_local or graph_collectives or c5_full_width or rccl or fit_keeps" 2>&1 | tail -40 > $out/pytest_multi.log
timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -x -q -k "split or c5" 2>&1 | tail -15 > $out/pytest_fullsize.log
timeout 900 python -m pytest tests/test_bench_contract_gpu.py -m gpu -x -q 2>&1 | tail -25 > $out/pytest_bench.log
python bench.py --steps 20 --warmup 5 > $out/bench_c2.json 2> $out/bench_c2.err
cat $out/pytest_multi.log $out/pytest_fullsize.log $out/pytest_bench.log
python tools/show_bench.py $out 2>&1 | tail
