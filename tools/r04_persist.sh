#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/r04k
mkdir -p $out
EBN_GEMM_PERSIST=1 timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "gemm_all_layouts or gemm_is_asym or dense_backward" 2>&1 | tail -8 > $out/pytest_persist.log
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "gemm_all_layouts" 2>&1 | tail -4 > $out/pytest_default.log
for c in c1 c2 c4; do
  EBN_GEMM_PERSIST=0 python tools/gemm_shapes_probe.py $c 2>&1 | grep -v amdgpu.ids | grep "^n " > $out/gemm_${c}_p0.log
  EBN_GEMM_PERSIST=1 python tools/gemm_shapes_probe.py $c 2>&1 | grep -v amdgpu.ids | grep "^n " > $out/gemm_${c}_p1.log
done
for c in c1 c2 c4; do
  EBN_GEMM_PERSIST=1 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg > $out/bench_${c}_persist.json 2> $out/bench_${c}_persist.err
  python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg > $out/bench_${c}.json 2> $out/bench_${c}.err
done
cat $out/pytest_persist.log $out/pytest_default.log
for c in c1 c2 c4; do echo "== $c persist 0"; cat $out/gemm_${c}_p0.log; echo "== $c persist 1"; cat $out/gemm_${c}_p1.log; done
python tools/show_bench.py $out 2>&1 | tail -8
