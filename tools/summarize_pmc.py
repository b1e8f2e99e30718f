"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --no-graph` into profiles/traffic.json.

usage: summarize_pmc.py <fetch_dir> <write_dir> <config> <round-tag>
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB; FETCH_SIZE is doubled
as MI355X_MICROARCH.md section HBM prescribes for gfx950 (it tallies 128-B requests at 64 B for 16-B/lane
loads).  Calibration on this repo's gather (known byte count 196.7 MB): corrected traffic 197.3 MB."""
import collections
import csv
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
# the Q|K|V projection is the SITE = 1 instantiation of the GEMM template (SITE template argument), whatever its tile
KERNELS = {"gather": (r"gather_rows_vec4_kernel", None), "qkv_gemm": (r"gemm_f32_kernel<\d+, \d+, \d+, false, false, true, 0\b", None)}


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    fdir, wdir, cfg, tag = sys.argv[1:5]
    f = per_kernel(Path(fdir) / "b_counter_collection.csv", "FETCH_SIZE")
    w = per_kernel(Path(wdir) / "b_counter_collection.csv", "WRITE_SIZE")
    out, detail = {}, {}
    for key, (needle, _) in KERNELS.items():
        fk = [v for k, vs in f.items() if re.search(needle, k) for v in vs]
        wk = [v for k, vs in w.items() if re.search(needle, k) for v in vs]
        # the user encoder launches the same instantiation on a 640-row problem: keep the news-encoder launches only
        fk = [v for v in fk if v >= 0.5 * max(fk)]
        wk = [v for v in wk if v >= 0.5 * max(wk)]
        fetch, write = sum(fk) / len(fk), sum(wk) / len(wk)
        out[key] = (2 * fetch + write) * 1024
        detail[key] = {"FETCH_SIZE_KiB_avg": fetch, "WRITE_SIZE_KiB_avg": write, "launches": len(fk),
                       "hbm_bytes_per_launch": out[key]}
    tf = ROOT / "profiles" / "traffic.json"
    blob = json.loads(tf.read_text()) if tf.exists() else {}
    blob[cfg] = out
    blob.setdefault("_detail", {})[f"{tag}_{cfg}"] = detail
    tf.write_text(json.dumps(blob, indent=1))
    print(json.dumps(detail, indent=1))


if __name__ == "__main__":
    main()
