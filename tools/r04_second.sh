#!/bin/bash
# round 4, second GPU pass: the LDS-free AttLayer2 GEMM and the duplicate-combining table-gradient accumulation
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/r04b
mkdir -p $out
timeout 600 python -m pytest tests/test_multi_rank_gpu.py -m gpu -x -q -k "rank_local" 2>&1 | tail -80 > $out/pytest_ranklocal.log
timeout 1200 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "gemm or duplicate_combining or tall or dense_backward" 2>&1 | tail -30 > $out/pytest_kernels.log
timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -x -q -k "c1 or c4 or c2" 2>&1 | tail -15 > $out/pytest_fullsize.log
for c in c2 c4; do
  EBN_GEMM_DIRECT=1 python tools/gemm_shapes_probe.py $c 2>&1 | grep -v amdgpu.ids > $out/gemm_${c}_direct.log
  EBN_GEMM_DIRECT=0 python tools/gemm_shapes_probe.py $c 2>&1 | grep -v amdgpu.ids > $out/gemm_${c}_lds.log
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop > $out/bench_c2.json 2> $out/bench_c2.err
EBN_GEMM_DIRECT=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop > $out/bench_c2_nodirect.json 2> $out/bench_c2_nodirect.err
for c in c1 c4; do
  for ids in uniform zipf; do
    python bench.py --config $c --ids $ids --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop > $out/bench_${c}_${ids}.json 2> $out/bench_${c}_${ids}.err
    python bench.py --config $c --ids $ids --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --atomic-table-grad > $out/bench_${c}_${ids}_atomic.json 2> $out/bench_${c}_${ids}_atomic.err
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c2 -o c2 -- \
  python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop > $out/bench_c2_under_rocprof.json 2> $out/rocprof_c2.err
rm -f $out/stats_c2/*kernel_trace.csv $out/stats_c2/*agent_info.csv
for c in c1 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_${c}_zipf -o $c -- \
    python bench.py --config $c --ids zipf --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-roofline > /dev/null 2> $out/rocprof_${c}_zipf.err
  rm -f $out/stats_${c}_zipf/*kernel_trace.csv $out/stats_${c}_zipf/*agent_info.csv
done
cat $out/pytest_ranklocal.log | tail -60
cat $out/pytest_kernels.log $out/pytest_fullsize.log
cat $out/gemm_c2_direct.log $out/gemm_c2_lds.log
python tools/show_bench.py $out 2>&1 | tail -40
