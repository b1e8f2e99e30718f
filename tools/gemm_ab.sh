#!/bin/bash
# A/B of library variants on the GEMM shapes of a config: gpurun -- 'bash tools/gemm_ab.sh <cfg> <variant.so>...'
cd "${GRAFT_REPO_ROOT:-.}"
cfg=$1; shift
mkdir -p gpurun_out/ab
{
python tools/gemm_shapes_probe.py $cfg
for v in "$@"; do echo "== $v"; EBNERD_HIP_LIB=$PWD/ebnerd-benchmark_amd/csrc/variants/$v python tools/gemm_shapes_probe.py $cfg; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab/gemm_$cfg.log
