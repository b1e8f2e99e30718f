#!/bin/bash
# Re-collects ONLY the probe summaries behind every config's roofline figures (profiles/<tag>_probe/): gpurun -- 'bash tools/collect_probe_only.sh r06'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=${1:-r06}; out=gpurun_out/$tag; mkdir -p $out/probe
export EBN_PROBE_KEEP_DIR=$PWD/$out/probe
for c in c2 c1 c3 c4 c5 c5h50; do
  python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-fit-loop --no-split-leg --legs "" > $out/bench_${c}_probe_run.json 2> $out/bench_${c}_probe_run.err
  python - <<P $out/bench_${c}_probe_run.json $c
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], d["ms_per_step"], "roofline avg_launch_us", d["roofline"]["avg_launch_us"], "gather", d["roofline_gather"]["avg_launch_us"], d["roofline"]["traffic_source"][:30])
P
done
