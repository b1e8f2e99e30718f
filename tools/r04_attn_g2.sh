#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04v; mkdir -p $out
V=$PWD/ebnerd-benchmark_amd/csrc/variants
run() { echo "== $*"; env "$@" python tools/tail_probe.py $N 30 a 2>&1 | grep "attn"; }
{
for N in 800 1760; do
  run X=0
  run EBNERD_HIP_LIB=$V/attn_g2v.so
  for md in 5 10; do for u in 2 4; do run EBNERD_HIP_LIB=$V/attn_g2v.so EBN_ATTN_STAGGER=$u EBN_ATTN_STAGGER_MOD=$md; done; done
  run EBNERD_HIP_LIB=$V/attn_vdirect.so EBN_ATTN_STAGGER=7 EBN_ATTN_STAGGER_MOD=5
  run X=0
done
} 2>&1 | tee $out/g2.log
