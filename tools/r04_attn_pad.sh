#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04v; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "attention" 2>&1 | tail -3 | tee $out/pad_tests.log
for N in 800 1760 3200; do python tools/tail_probe.py $N 30 a 2>&1 | grep "attn"; done | tee $out/pad_probe.log
for cfg in c2 c4 c5; do python bench.py --config $cfg --steps 200 --warmup 20 --no-probe --no-fit-loop --no-cpu-baseline --no-split-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'])"; done | tee $out/pad_bench.log
