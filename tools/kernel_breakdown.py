"""Per-step kernel time table from a rocprofv3 --stats kernel_stats CSV. usage: kernel_breakdown.py <csv> [steps_in_run]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 110  # bench.py default: 10 warm-up + 50 graph + 50 kernel-by-kernel
tot = 0.0
for r in rows:
    n = r["Name"].split("(long")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:84]
    per = float(r["TotalDurationNs"]) / steps / 1e3
    tot += per
    if per > 3:
        print(f"{per:8.1f} us/step  calls/step {int(r['Calls']) / steps:5.1f}  avg {float(r['AverageNs']) / 1e3:7.1f}  {n}")
print(f"{tot:8.1f} us/step total")
