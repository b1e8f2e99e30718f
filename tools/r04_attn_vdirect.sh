#!/bin/bash
# A/B of the group backward with V taken straight from global memory (3 LDS tiles per wave, 5 workgroups per CU) and an
# occupancy control (padded LDS: 3 workgroups per CU).  gpurun -- 'bash tools/r04_attn_vdirect.sh'
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04v; mkdir -p $out
V=$PWD/ebnerd-benchmark_amd/csrc/variants
{
for n in 800 1760 3200; do
  python tools/tail_probe.py $n 30 a
  EBNERD_HIP_LIB=$V/attn_vdirect.so python tools/tail_probe.py $n 30 a
  EBNERD_HIP_LIB=$V/attn_pad3.so python tools/tail_probe.py $n 30 a
  python tools/tail_probe.py $n 30 a
done
} 2>&1 | grep -v Warn | tee $out/probe.log
EBNERD_HIP_LIB=$V/attn_vdirect.so timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "attention" 2>&1 | tail -5 | tee $out/tests.log
for i in 1 2; do
  python bench.py --config c2 --steps 200 --warmup 20 --no-probe --no-fit-loop --no-cpu-baseline --no-split-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 base   ', d['ms_per_step'])"
  EBNERD_HIP_LIB=$V/attn_vdirect.so python bench.py --config c2 --steps 200 --warmup 20 --no-probe --no-fit-loop --no-cpu-baseline --no-split-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 vdirect', d['ms_per_step'])"
done 2>&1 | tee $out/bench.log
