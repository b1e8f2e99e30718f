#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/r04j
mkdir -p $out
{
for rows in 24000 52800; do
  EBN_GEMM_DIRECT=0 python tools/tn_gemm_probe.py $rows
  python tools/tn_gemm_probe.py $rows
  for wgs in 256 512 768; do for rc in "5 5" "4 5" "3 5" "5 4" "4 4" "3 4"; do
    set -- $rc
    EBN_GEMM_DIRECT_TN_WGS=$wgs EBN_GEMM_DIRECT_TN_R=$1 EBN_GEMM_DIRECT_TN_C=$2 python tools/tn_gemm_probe.py $rows
  done; done
done
} 2>&1 | grep -v amdgpu.ids | tee $out/tn_sweep.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg > $out/bench_c2.json 2> $out/bench_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c2 -o c2 -- \
  python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg --no-roofline > /dev/null 2> $out/rocprof_c2.err
rm -f $out/stats_c2/*kernel_trace.csv $out/stats_c2/*agent_info.csv
grep -h "grad_finish\|tn_kernel" $out/stats_c2/*kernel_stats.csv | cut -c1-160
python tools/show_bench.py $out 2>&1 | tail -3
