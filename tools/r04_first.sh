#!/bin/bash
# round 4, first GPU pass: the new parity tests, the default bench line, and the U / Z (Zipf + padded history) lines
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/r04a
mkdir -p $out
timeout 1500 python -m pytest tests/test_full_size_parity.py tests/test_multi_rank_gpu.py -m gpu -x -q 2>&1 | tail -15 > $out/pytest_new.log
python bench.py --steps 20 --warmup 5 > $out/bench_c2.json 2> $out/bench_c2.err
for c in c1 c4; do
  for ids in uniform zipf; do
    python bench.py --config $c --ids $ids --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop > $out/bench_${c}_${ids}_atomics.json 2> $out/bench_${c}_${ids}_atomics.err
    python bench.py --config $c --ids $ids --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --segmented-table-grad > $out/bench_${c}_${ids}_segmented.json 2> $out/bench_${c}_${ids}_segmented.err
  done
done
for c in c2 c5; do
  python bench.py --config $c --ids zipf --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop > $out/bench_${c}_zipf.json 2> $out/bench_${c}_zipf.err
done
for c in c1 c4; do for ids in uniform zipf; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_${c}_${ids} -o $c -- \
    python bench.py --config $c --ids $ids --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-roofline > /dev/null 2> $out/rocprof_${c}_${ids}.err
  rm -f $out/stats_${c}_${ids}/*kernel_trace.csv $out/stats_${c}_${ids}/*agent_info.csv
done; done
cat $out/pytest_new.log
python tools/show_bench.py $out 2>&1 | tail -40
