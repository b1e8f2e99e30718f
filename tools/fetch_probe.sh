#!/bin/bash
# FETCH_SIZE of the GEMM kernels of an eager c2 step, per value of EBN_GEMM_COL_GROUP: gpurun -- 'bash tools/fetch_probe.sh 0 4 10'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for g in "$@"; do
  d=gpurun_out/fetch_$g; rm -rf $d; mkdir -p $d
  EBN_GEMM_COL_GROUP=$g rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $d -o b -- \
    python bench.py --no-graph --no-roofline --no-cpu-baseline --steps 6 --warmup 2 --repeats 1 > /dev/null 2> $d/err.log
  python - "$d" "$g" <<'P'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    m = re.search(r"gemm_f32_kernel<[^>]*>", r["Kernel_Name"])
    if m: acc[m.group(0)].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"col_group {sys.argv[2]:>3}  {k:55s} FETCH_SIZE x2 = {2 * 1024 * sum(v) / len(v) / 1e6:8.1f} MB  (n={len(v)})")
P
  rm -rf $d
done
