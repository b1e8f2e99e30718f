#!/bin/bash
# Per-launch durations (us) of one kernel over a bench.py run, in launch order:  trace_kernel.sh <config> <kernel-name-substring>
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p /tmp/tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python bench.py --config $1 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg > /tmp/tr/bench.json 2>/dev/null
python - "$2" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(__import__("glob").glob("/tmp/tr/**/*kernel_trace.csv", recursive=True)[0])) if sys.argv[1] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(d), "launches")
for i in range(0, len(d), 10):
    print(i, " ".join(f"{x:6.1f}" for x in d[i:i + 10]))
PY
tail -c 600 /tmp/tr/bench.json
