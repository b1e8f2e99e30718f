"""Copies what tools/collect_round.sh produced (merged back under gpurun_out/<tag>/) into profiles/ as <tag>_*:
bench lines of every config, per-config rocprofv3 kernel statistics, the c2 HBM-traffic counters (-> profiles/traffic.json
via summarize_pmc) and MFMA-pipe counters, the per-launch trace of the roofline GEMM.  usage: store_round.py <tag>"""
import csv
import json
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
G, P = ROOT / "gpurun_out" / tag, ROOT / "profiles"


def last_json(path):
    lines = [l for l in path.read_text().splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def find(d, suffix):
    hits = sorted(d.rglob(f"*{suffix}"))
    return hits[0] if hits else None


# ---- HBM traffic of the gather and the Q|K|V GEMM (c2), corrected as MI355X_MICROARCH.md prescribes
fd, wd = G / "pmc_FETCH_SIZE", G / "pmc_WRITE_SIZE"
for d in (fd, wd):  # summarize_pmc expects <dir>/b_counter_collection.csv
    src = find(d, "counter_collection.csv")
    if src and src != d / "b_counter_collection.csv":
        shutil.copy(src, d / "b_counter_collection.csv")
subprocess.run([sys.executable, str(ROOT / "tools" / "summarize_pmc.py"), str(fd), str(wd), "c2", tag], check=True, stdout=subprocess.DEVNULL)
(P / f"{tag}_pmc").mkdir(exist_ok=True)
for c, d in (("FETCH_SIZE", fd), ("WRITE_SIZE", wd)):
    with open(d / "b_counter_collection.csv") as f, open(P / f"{tag}_pmc" / f"{c}_bench_c2_no_graph.csv", "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                    "Accum_VGPR_Count", "SGPR_Count"])
        for r in csv.DictReader(f):
            w.writerow([r["Dispatch_Id"], r["Kernel_Name"].split("(long")[0][:110], r["Counter_Name"], r["Counter_Value"], r["Grid_Size"],
                        r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"]])
# ---- MFMA pipe utilisation
for d in (G / "pmc_mfma_a", G / "pmc_mfma_b"):
    src = find(d, "counter_collection.csv")
    if src and src != d / "b_counter_collection.csv":
        shutil.copy(src, d / "b_counter_collection.csv")
subprocess.run([sys.executable, str(ROOT / "tools" / "summarize_mfma_pmc.py"), str(G / "pmc_mfma_a"), str(G / "pmc_mfma_b"),
                str(P / f"{tag}_pmc" / "mfma_utilisation_bench_c2.json")], check=True)
# ---- probe-only summaries (the figures behind roofline / roofline_gather of every config, recomputable from profiles/ alone)
if (G / "probe").exists():
    (P / f"{tag}_probe").mkdir(exist_ok=True)
    for f in sorted((G / "probe").iterdir()):
        shutil.copy(f, P / f"{tag}_probe" / f.name)
for name in ("bench_default", "bench_c2_force_dist_1rank_rccl"):
    src = G / f"{name}.json"
    line = last_json(src) if src.exists() else None
    if line is not None:
        (P / f"{tag}_{name}.json").write_text(json.dumps(line) + "\n")
# ---- kernel statistics + bench lines
traffic = json.loads((P / "traffic.json").read_text())["c2"]
for c in ("c1", "c2", "c3", "c4", "c5", "c5h50"):
    st = find(G / f"stats_{c}", "kernel_stats.csv") if (G / f"stats_{c}").exists() else None
    if st:
        shutil.copy(st, P / f"{tag}_kernel_stats_bench_{c}.csv")
    for kind in ("", "_under_rocprof"):
        src = G / f"bench_{c}{kind}.json"
        line = last_json(src) if src.exists() else None
        if line is None:
            continue
        (P / f"{tag}_bench_{c}{kind}_1gpu.json").write_text(json.dumps(line) + "\n")
for kind in ("", "_under_rocprof"):  # the opt-in split precision
    src = G / f"bench_c2_split{kind}.json"
    line = last_json(src) if src.exists() else None
    if line:
        (P / f"{tag}_bench_c2_split_precision{kind}_1gpu.json").write_text(json.dumps(line) + "\n")
st = find(G / "stats_c2_split", "kernel_stats.csv") if (G / "stats_c2_split").exists() else None
if st:
    shutil.copy(st, P / f"{tag}_kernel_stats_bench_c2_split_precision.csv")
for src in sorted(G.glob("bench_*_zipf*.json")) + sorted(G.glob("bench_*_uniform_atomic.json")):  # SURVEY 8(d) Z lines and the accumulation A/B
    line = last_json(src)
    if line:
        (P / f"{tag}_{src.stem}_1gpu.json").write_text(json.dumps(line) + "\n")
for c in ("c2", "c4", "c5"):
    src = G / f"bench_{c}_2ranks_gloo.json"
    line = last_json(src) if src.exists() else None
    if line:
        (P / f"{tag}_bench_{c}_2ranks_gloo_dry_run.json").write_text(json.dumps(line) + "\n")
if (G / "gemm_launch_trace.txt").exists():
    body = (G / "gemm_launch_trace.txt").read_text()
    head = (P / f"{tag}_gemm_launch_trace.txt").read_text().split("\n\n")[0] if (P / f"{tag}_gemm_launch_trace.txt").exists() else ""
    (P / f"{tag}_gemm_launch_trace_final.txt").write_text(head + "\n\n(final build of the round, warm replays in place)\n" + body)
for c in ("c1", "c2", "c3", "c4", "c5", "c5h50"):
    p = P / f"{tag}_bench_{c}_1gpu.json"
    if p.exists():
        x = json.loads(p.read_text())
        print(c, round(x["value"]), f"{x['ms_per_step']:.4f} ms", {k: round(v.get("frac", 0), 3) for k, v in x.items() if k.startswith("roofline")})
