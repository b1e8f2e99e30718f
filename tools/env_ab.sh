#!/bin/bash
# A/B of an environment switch on the bench step: tools/env_ab.sh <config> <VAR> [precision]   (VAR unset vs VAR=1, interleaved)
cd "${GRAFT_REPO_ROOT:-.}"
cfg=$1; var=$2; prec=${3:-exact}
for i in 1 2; do
  python bench.py --config $cfg --precision $prec --steps 200 --warmup 20 --no-probe --no-fit-loop --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var unset', d['ms_per_step'])"
  env $var=1 python bench.py --config $cfg --precision $prec --steps 200 --warmup 20 --no-probe --no-fit-loop --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=1   ', d['ms_per_step'])"
done
