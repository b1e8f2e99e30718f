"""Prints the headline fields of bench.py JSON lines: show_bench.py <dir> [names...]"""
import json
import sys
from pathlib import Path

d = Path(sys.argv[1])
names = sys.argv[2:] or sorted(p.stem for p in d.glob("bench_*.json"))
for f in names:
    try:
        t = (d / f"{f}.json").read_text().strip().splitlines()
        x = json.loads([l for l in t if l.startswith("{")][-1])
        r, g = x.get("roofline", {}), x.get("roofline_gather", {})
        print(f"{f:18s} {x['value']:9.0f} impr/s  {x['ms_per_step']:.4f} ms [{x.get('ms_per_step_min', 0):.4f}..{x.get('ms_per_step_max', 0):.4f}] n={x['n_gpus']} "
              f"{x.get('backend') or ''} | roofline {r.get('frac', 0):.3f} ({r.get('avg_launch_us', 0):.1f} us) gather {g.get('frac', 0) or 0:.3f}")
    except Exception as e:
        err = (d / f"{f}.err")
        print(f"{f}: ERR {e}\n{err.read_text()[-1200:] if err.exists() else ''}")
