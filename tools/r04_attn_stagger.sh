#!/bin/bash
# Sweep of the start-up stagger of the group-form attention backward (EBN_ATTN_STAGGER = units of 512 cycles, _SHIFT, _MOD).
#   gpurun -- 'bash tools/r04_attn_stagger.sh'
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04v; mkdir -p $out
run() { echo "== $*"; env "$@" python tools/tail_probe.py $N 30 a 2>&1 | grep "attn"; }
{
for N in 800 1760; do
  run EBN_ATTN_STAGGER=0
  for sh in 8 3; do for u in 2 4 6 8 12; do run EBN_ATTN_STAGGER=$u EBN_ATTN_STAGGER_SHIFT=$sh; done; done
  for md in 4 5 6; do run EBN_ATTN_STAGGER=7 EBN_ATTN_STAGGER_MOD=$md; done
  run EBN_ATTN_STAGGER=0
done
} 2>&1 | tee $out/stagger.log
