#!/bin/bash
# generic A/B: variants/base.so against the in-tree build.  gpurun -- 'bash tools/r04_ab.sh "<pytest -k expr>" cfg...'
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04v; mkdir -p $out
V=$PWD/ebnerd-benchmark_amd/csrc/variants
k="$1"; shift
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_nrms_model.py -m gpu -q -x -k "$k" 2>&1 | tail -3 | tee $out/ab_tests.log
b() { cfg=$1; shift; env "$@" python bench.py --config $cfg --steps 200 --warmup 20 --no-probe --no-fit-loop --no-cpu-baseline --no-split-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '$*', d['ms_per_step'])"; }
{
for i in 1 2; do
for cfg in "$@"; do
  b $cfg EBNERD_HIP_LIB=$V/base.so
  b $cfg X=0
done
done
} 2>&1 | tee $out/ab.log
