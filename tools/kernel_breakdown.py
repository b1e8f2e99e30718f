"""Per-step kernel time table from a rocprofv3 --stats kernel_stats CSV. usage: kernel_breakdown.py <csv> [steps_in_run] [min_us]
default steps = 85: bench.py --steps 20 --warmup 5 --repeats 3 (5 + 3*20 graph-replayed + 20 kernel-by-kernel)"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 85
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
tot, launches = 0.0, 0.0
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "").split("(")[0][:80]
    per = float(r["TotalDurationNs"]) / steps / 1e3
    tot += per
    launches += int(r["Calls"]) / steps
    if per > min_us:
        print(f"{per:8.1f} us/step  calls/step {int(r['Calls']) / steps:5.2f}  avg {float(r['AverageNs']) / 1e3:7.1f}  {n}")
print(f"{tot:8.1f} us/step total, {launches:.1f} launches/step")
