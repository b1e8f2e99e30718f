"""MFMA-pipe utilisation per kernel from two rocprofv3 --pmc passes of `bench.py --no-graph`.

usage: summarize_mfma_pmc.py <dir_pass_a> <dir_pass_b> <out.json>
  pass a: --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
  pass b: --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT
mfma_pipe_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x active cycles).  rocprofv3 reports GRBM_GUI_ACTIVE summed
over the 8 XCDs, so active cycles = GRBM_GUI_ACTIVE / 8 and the fraction is busy / (128 x GRBM_GUI_ACTIVE)."""
import collections
import csv
import json
import sys
from pathlib import Path


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(Path(d) / "b_counter_collection.csv")):
        acc[r["Kernel_Name"].split("(long")[0][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    a, b, out = load(sys.argv[1]), load(sys.argv[2]), {}
    for k in a:
        if "MFMA" not in "".join(a[k].keys()) or sum(a[k].get("SQ_INSTS_MFMA", [0])) == 0:
            continue
        m = {c: sum(v) / len(v) for c, v in a[k].items()}
        w = {c: sum(v) / len(v) for c, v in b.get(k, {}).items()}
        rec = {"launches": len(a[k]["SQ_INSTS_MFMA"]), **m}
        if m.get("GRBM_GUI_ACTIVE"):
            rec["mfma_pipe_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * m["GRBM_GUI_ACTIVE"])
        if w.get("SQ_WAVE_CYCLES"):
            rec["SQ_WAIT_ANY/SQ_WAVE_CYCLES"] = w.get("SQ_WAIT_ANY", 0.0) / w["SQ_WAVE_CYCLES"]
            rec["SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES"] = w.get("SQ_WAIT_INST_ANY", 0.0) / w["SQ_WAVE_CYCLES"]
            rec["SQ_LDS_BANK_CONFLICT"] = w.get("SQ_LDS_BANK_CONFLICT", 0.0)
        out[k] = rec
    Path(sys.argv[3]).write_text(json.dumps(out, indent=1))
    for k, v in out.items():
        print(f"{v.get('mfma_pipe_busy_frac', float('nan')):.3f}  {k}")


if __name__ == "__main__":
    main()
