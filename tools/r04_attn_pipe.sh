#!/bin/bash
# Pipelined persistent group backward: parity + timing.  gpurun -- 'bash tools/r04_attn_pipe.sh'
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04v; mkdir -p $out
run() { echo "== $*"; env "$@" python tools/tail_probe.py $N 30 a 2>&1 | grep "attn bwd"; }
EBN_ATTN_BWD_PIPE=1 timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "attention" 2>&1 | tail -5 | tee $out/pipe_tests.log
{
for N in 800 1760 3200; do
  run X=0
  run EBN_ATTN_STAGGER=8
  run EBN_ATTN_BWD_PIPE=1
  run EBN_ATTN_BWD_PIPE=1 EBN_ATTN_STAGGER=4
  run EBN_ATTN_BWD_PIPE=1 EBN_ATTN_STAGGER=8
  run EBN_ATTN_BWD_PIPE=1 EBN_ATTN_BWD_PIPE_SLOTS=768
  run EBN_ATTN_BWD_PIPE=1 EBN_ATTN_BWD_PIPE_SLOTS=2048
  run X=0
done
} 2>&1 | tee $out/pipe.log
