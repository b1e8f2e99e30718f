#!/bin/bash
# A/B of a library variant on the bench step: tools/lib_ab.sh <config> <variant.so> [precision]   (default library vs variant, interleaved)
cd "${GRAFT_REPO_ROOT:-.}"
cfg=$1; v=$2; prec=${3:-exact}
one() { python bench.py --config $cfg --precision $prec --steps 200 --warmup 20 --no-probe --no-fit-loop --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in 1 2; do
  one default
  EBNERD_HIP_LIB=$PWD/ebnerd-benchmark_amd/csrc/variants/$v one $v
done
