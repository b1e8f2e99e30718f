#!/bin/bash
# A/B of library variants on the tail kernels: gpurun -- 'bash tools/attn_ab.sh <which> <variant.so>...'
cd "${GRAFT_REPO_ROOT:-.}"
which=$1; shift
mkdir -p gpurun_out/ab
{
python tools/tail_probe.py 800 30 $which
for v in "$@"; do EBNERD_HIP_LIB=$PWD/ebnerd-benchmark_amd/csrc/variants/$v python tools/tail_probe.py 800 30 $which; done
python tools/tail_probe.py 800 30 $which
} 2>&1 | tee gpurun_out/ab/probe.log
