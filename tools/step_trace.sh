#!/bin/bash
# Kernel-by-kernel trace of ONE training step (launch order, duration, gap to the previous kernel), eager launches.
#   gpurun -- 'bash tools/step_trace.sh <cfg>'   -> gpurun_out/step_trace_<cfg>.txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
cfg=${1:-c2}
d=gpurun_out/step_trace_$cfg
rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python bench.py --config $cfg --no-graph --no-roofline --no-cpu-baseline --steps 6 --warmup 3 --repeats 1 > /dev/null 2> $d/err.log
python - "$d" <<'P' > gpurun_out/step_trace_$cfg.txt
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last step = from the last copy3/expand kernel to the last adam kernel
names = [r["Kernel_Name"] for r in rows]
ends = [i for i, n in enumerate(names) if "adam_keras" in n]
starts = [i for i, n in enumerate(names) if "copy3_kernel" in n or "expand_titles" in n]
last = ends[-1]
first = max(i for i in starts if i < last)
step = rows[first:last + 1]
t0 = int(step[0]["Start_Timestamp"])
pe = None
tot = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").split("(")[0][:72]
    gap = (s - pe) / 1e3 if pe else 0.0
    tot += (e - s) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {n}")
    pe = e
print(f"kernels {len(step)}, sum of durations {tot:.1f} us, span {(pe - t0) / 1e3:.1f} us")
P
rm -rf $d
tail -60 gpurun_out/step_trace_$cfg.txt
