#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04v; mkdir -p $out
run() { echo "== $*"; env "$@" python tools/tail_probe.py $N $L a 2>&1 | grep "attn"; }
{
N=256; L=50
run X=0
for md in 3 5; do for u in 4 8 12; do run EBN_ATTN2_STAGGER=$u EBN_ATTN2_STAGGER_MOD=$md; done; done
run X=0
N=800; L=30
run X=0
run EBN_ATTN_STAGGER=0
N=1760
run X=0
run EBN_ATTN_STAGGER=0
} 2>&1 | tee $out/attn2_stagger.log
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "attention" 2>&1 | tail -3 | tee $out/tests2.log
for cfg in c2 c4 c1; do python bench.py --config $cfg --steps 200 --warmup 20 --no-probe --no-fit-loop --no-cpu-baseline --no-split-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'])"; done | tee $out/bench2.log
