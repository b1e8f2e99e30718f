#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r04f
{
for rows in 24000 52800; do
  EBN_GEMM_DIRECT=0 python tools/direct_gemm_probe.py $rows
  for depth in 2 3; do
    EBN_GEMM_DIRECT_DEPTH=$depth python tools/direct_gemm_probe.py $rows
  done
  for rc in "4 7" "3 7" "2 7" "1 7" "4 5" "3 5" "2 5" "4 4" "3 4" "2 4"; do
    set -- $rc
    EBN_GEMM_DIRECT_R=$1 EBN_GEMM_DIRECT_C=$2 python tools/direct_gemm_probe.py $rows
  done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04f/direct_sweep.log
