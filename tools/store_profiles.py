"""Copies the artefacts tools/collect_profiles.sh produced (merged back under gpurun_out/) into profiles/:
   traffic.json (via summarize_pmc), slim PMC CSVs, kernel stats, the two bench lines.  usage: store_profiles.py <tag>"""
import csv
import json
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
G, P = ROOT / "gpurun_out", ROOT / "profiles"
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
subprocess.run([sys.executable, str(ROOT / "tools" / "summarize_pmc.py"), str(G / "pmc_fetch"), str(G / "pmc_write"), "c2", tag],
               check=True, stdout=subprocess.DEVNULL)
(P / f"{tag[:3]}_pmc").mkdir(exist_ok=True)
for c, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    with open(G / d / "b_counter_collection.csv") as f, open(P / f"{tag[:3]}_pmc" / f"{c}_bench_c2_no_graph.csv", "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size", "Workgroup_Size", "LDS_Block_Size",
                    "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count"])
        for r in csv.DictReader(f):
            w.writerow([r["Dispatch_Id"], r["Kernel_Name"].split("(long")[0][:110], r["Counter_Name"], r["Counter_Value"], r["Grid_Size"],
                        r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"]])
shutil.copy(G / "prof_stats" / "b_kernel_stats.csv", P / f"{tag[:3]}_kernel_stats_bench_c2.csv")
shutil.copy(G / "bench_c2_under_rocprof.json", P / f"{tag[:3]}_bench_c2_under_rocprof.json")
line = json.loads((G / "bench_c2.json").read_text().strip().splitlines()[-1])
t = json.loads((P / "traffic.json").read_text())["c2"]  # the run itself read the previous round's traffic.json
line["roofline"]["traffic"], line["roofline_gather"]["traffic"] = t["qkv_gemm"], t["gather"]
(P / f"{tag[:3]}_bench_c2_1gpu.json").write_text(json.dumps(line) + "\n")
print(line["value"], line["ms_per_step"], line["roofline"]["frac"], line["roofline"]["avg_launch_us"], line["roofline_gather"]["frac"])
