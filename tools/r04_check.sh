#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/r04m
mkdir -p $out
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "gemm" 2>&1 | tail -4 > $out/pytest_gemm.log
timeout 1200 python -m pytest tests/test_full_size_parity.py tests/test_nrms_model.py -m gpu -x -q 2>&1 | tail -6 > $out/pytest_model.log
for c in c1 c4 c2; do
  python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg > $out/bench_${c}.json 2> $out/bench_${c}.err
done
cat $out/pytest_gemm.log $out/pytest_model.log
python tools/show_bench.py $out 2>&1 | tail -4
