#!/bin/bash
# In-step A/B of the attention backward stagger.  gpurun -- 'bash tools/r04_attn_stagger_step.sh'
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04v; mkdir -p $out
V=$PWD/ebnerd-benchmark_amd/csrc/variants
b() { cfg=$1; shift; env "$@" python bench.py --config $cfg --steps 200 --warmup 20 --no-probe --no-fit-loop --no-cpu-baseline --no-split-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '$*', d['ms_per_step'])"; }
{
for i in 1 2; do
for cfg in c2 c4; do
  b $cfg X=0
  b $cfg EBN_ATTN_STAGGER=8
  b $cfg EBN_ATTN_STAGGER=7 EBN_ATTN_STAGGER_MOD=5 EBNERD_HIP_LIB=$V/attn_vdirect.so
done
done
} 2>&1 | tee $out/stagger_step.log
