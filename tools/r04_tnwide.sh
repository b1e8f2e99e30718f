#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/r04l
mkdir -p $out
for c in c1 c4; do
  python tools/gemm_shapes_probe.py $c 2>&1 | grep "dWqkv" > $out/gemm_${c}_base.log
  for wgs in 256 512; do
    EBN_GEMM_DIRECT_TN_ANYPAD=1 EBN_GEMM_DIRECT_TN_MAXN=1280 EBN_GEMM_DIRECT_TN_WGS=$wgs python tools/gemm_shapes_probe.py $c 2>&1 | grep "dWqkv" > $out/gemm_${c}_wide_$wgs.log
  done
  python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg > $out/bench_${c}.json 2> $out/bench_${c}.err
  EBN_GEMM_DIRECT_TN_ANYPAD=1 EBN_GEMM_DIRECT_TN_MAXN=1280 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg > $out/bench_${c}_wide.json 2> $out/bench_${c}_wide.err
  EBN_GEMM_DIRECT_TN_ANYPAD=1 EBN_GEMM_DIRECT_TN_MAXN=1280 EBN_GEMM_DIRECT_TN_WGS=512 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg > $out/bench_${c}_wide512.json 2> $out/bench_${c}_wide512.err
done
EBN_GEMM_DIRECT_TN_ANYPAD=1 EBN_GEMM_DIRECT_TN_MAXN=1280 timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "gemm_all_layouts" 2>&1 | tail -3
cat $out/gemm_*.log
python tools/show_bench.py $out 2>&1 | tail -8
