#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/r04i
mkdir -p $out
timeout 1200 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "gemm or finishing or partials or dense_backward" 2>&1 | tail -30 > $out/pytest_kernels.log
timeout 900 python -m pytest tests/test_nrms_model.py -m gpu -x -q -k "finishing or trajectory or training" 2>&1 | tail -30 > $out/pytest_model.log
for c in c2 c4; do
  python tools/gemm_shapes_probe.py $c 2>&1 | grep -v amdgpu.ids > $out/gemm_${c}.log
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg > $out/bench_c2.json 2> $out/bench_c2.err
for c in c1 c4; do
  python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg > $out/bench_$c.json 2> $out/bench_$c.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c2 -o c2 -- \
  python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg --no-roofline > /dev/null 2> $out/rocprof_c2.err
rm -f $out/stats_c2/*kernel_trace.csv $out/stats_c2/*agent_info.csv
cat $out/pytest_kernels.log $out/pytest_model.log | tail -40
grep -E "n dW|n U=|n dY" $out/gemm_c2.log $out/gemm_c4.log
python tools/show_bench.py $out 2>&1 | tail
