#!/bin/bash
# Collects the round's judged artefacts on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 3000 -- 'bash tools/collect_round.sh r02'
# then, back in the container: python tools/store_round.py r02     (copies the summaries into profiles/)
# --pmc passes are separate runs with --kernel-trace only (no sys/hip/hsa trace domains), as the pool requires.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=${1:-r06}
out=gpurun_out/$tag
mkdir -p $out
# the probe-only summaries behind every roofline* figure (bench.py keeps them when asked): per config, rocprofv3 --kernel-trace --stats and the
# FETCH_SIZE / WRITE_SIZE passes over `bench.py --kernel-probe` -- only the two roofline kernels' launches, no calibration copies
export EBN_PROBE_KEEP_DIR=$PWD/$out/probe
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $out/pytest_gpu.log
# the driver's command first (c2 headline with the c1 / c3 / c4 / c5 legs, CPU baseline included), then every other config on its own
python bench.py > $out/bench_default.json 2> $out/bench_default.err
python bench.py --steps 20 --warmup 5 --legs "" > $out/bench_c2.json 2> $out/bench_c2.err
# the multi-rank branch on a one-rank RCCL group (what the 8-GPU node will run, executed here)
timeout 900 python bench.py --gpus 1 --force-dist --steps 20 --warmup 5 > $out/bench_c2_force_dist_1rank_rccl.json 2> $out/bench_c2_force_dist.err
for c in c1 c3 c4 c5 c5h50; do
  nocpu=--no-cpu-baseline; [ $c = c3 ] && nocpu=   # c3 carries its own cpu_baseline leg (a second of CPU work)
  python bench.py --config $c --steps 20 --warmup 5 $nocpu --no-split-leg > $out/bench_$c.json 2> $out/bench_$c.err
done
# SURVEY.md 8(d) "Z" inputs (Zipf ids + 15 % padded history slots): every trainable / sharded config, and the A/B of the table-gradient
# accumulation on them (duplicate-combining default vs one atomic per element)
for c in c1 c2 c4 c5; do
  python bench.py --config $c --ids zipf --steps 20 --warmup 5 --no-cpu-baseline --no-fit-loop --no-split-leg > $out/bench_${c}_zipf.json 2> $out/bench_${c}_zipf.err
done
for c in c1 c4; do for ids in uniform zipf; do
  python bench.py --config $c --ids $ids --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg --atomic-table-grad > $out/bench_${c}_${ids}_atomic.json 2> $out/bench_${c}_${ids}_atomic.err
done; done
# the opt-in second precision (bf16x6 split projections): its own line, next to an exact line from the same box
python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --precision split > $out/bench_c2_split.json 2> $out/bench_c2_split.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c2_split -o c2_split -- \
  python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --precision split > $out/bench_c2_split_under_rocprof.json 2> $out/rocprof_c2_split.err
rm -f $out/stats_c2_split/*kernel_trace.csv $out/stats_c2_split/*agent_info.csv
# two ranks sharing the one GPU over gloo: functional dry run of the multi-rank paths (not a scaling number)
for c in c2 c4 c5; do
  timeout 600 python bench.py --gpus 2 --config $c --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_${c}_2ranks_gloo.json 2> $out/bench_${c}_2ranks_gloo.err
done
# per-kernel statistics of the same bench command, per config
for c in c2 c1 c3 c4 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$c -o $c -- \
    python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg > $out/bench_${c}_under_rocprof.json 2> $out/rocprof_$c.err
  rm -f $out/stats_$c/*kernel_trace.csv $out/stats_$c/*agent_info.csv
done
# HBM traffic counters (c2): separate --pmc passes, kernel by kernel (no graphs: counters are per dispatch)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o b -- \
    python bench.py --no-graph --no-roofline --no-probe --no-fit-loop --no-split-leg --no-cpu-baseline --steps 10 --warmup 2 --repeats 1 > /dev/null 2> $out/pmc_$c.err
  rm -f $out/pmc_$c/*kernel_trace.csv $out/pmc_$c/*agent_info.csv
done
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_mfma_a -o b -- \
  python bench.py --no-graph --no-roofline --no-probe --no-fit-loop --no-split-leg --no-cpu-baseline --steps 10 --warmup 2 --repeats 1 > /dev/null 2> $out/pmc_mfma_a.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out/pmc_mfma_b -o b -- \
  python bench.py --no-graph --no-roofline --no-probe --no-fit-loop --no-split-leg --no-cpu-baseline --steps 10 --warmup 2 --repeats 1 > /dev/null 2> $out/pmc_mfma_b.err
rm -f $out/pmc_mfma_*/*kernel_trace.csv $out/pmc_mfma_*/*agent_info.csv
bash tools/trace_kernel.sh c2 "256, 64, 4, false, false, true, 0" > $out/gemm_launch_trace.txt 2>&1
cat $out/pytest_gpu.log
ls $out
