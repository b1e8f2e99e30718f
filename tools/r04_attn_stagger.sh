#!/bin/bash
# Sweep of the start-up stagger of the attention kernels.  gpurun -- 'bash tools/r04_attn_stagger.sh'
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04v; mkdir -p $out
V=$PWD/ebnerd-benchmark_amd/csrc/variants
run() { echo "== $*"; env "$@" python tools/tail_probe.py $N 30 a 2>&1 | grep "attn"; }
{
for N in 800 1760; do
  run X=0
  for sh in 8 3; do for u in 2 4 6 8 12; do run EBN_ATTN_STAGGER=$u EBN_ATTN_STAGGER_SHIFT=$sh; done; done
  for u in 3 5 7; do run EBN_ATTN_STAGGER=$u EBN_ATTN_STAGGER_MOD=5 EBNERD_HIP_LIB=$V/attn_vdirect.so; done
  for md in 4 8; do for u in 2 4 8; do run EBN_ATTN_FWD_STAGGER=$u EBN_ATTN_FWD_STAGGER_MOD=$md; done; done
  run X=0
done
} 2>&1 | tee $out/stagger2.log
