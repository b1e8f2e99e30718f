#!/bin/bash
# Per-kernel medians of tools/split_gemm_probe.py under rocprofv3 --kernel-trace:  gpurun -- 'bash tools/split_prof.sh <tag> [probe args]'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=${1:-sp}; shift
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/sp -o sp -- python tools/split_gemm_probe.py "$@" > $out/probe.txt 2>&1
cat $out/probe.txt | grep -v "^$" | tail -8
python - "$out/sp/sp_kernel_trace.csv" <<'P'
import collections, csv, sys
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if any(k in n for k in ("split_planes", "bf16x6", "split_reduce", "gemm_f32_kernel")):
        key = (n.replace("void (anonymous namespace)::", "").split("(")[0][:48], r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    v = sorted(v)
    print(f"{k[0]:50s} grid {k[1]:>8s} {k[2]:>5s} {k[3]:>3s}  n={len(v):3d}  median {v[len(v) // 2]:8.1f} us  min {v[0]:8.1f}")
P
rm -f $out/sp/*.csv
