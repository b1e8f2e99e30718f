"""bench.py prints ONE JSON line with the driver's fields, the roofline of the dominant kernel and (at N = 1) the CPU baseline."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["c2", "c3"])
def test_bench_line_carries_the_contract_fields(cfg):
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", cfg, "--steps", "3", "--warmup", "1", "--repeats", "2",
                          "--cpu-steps", "3"], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["metric"] == "training impressions/sec" and d["unit"] == "impressions/s" and d["n_gpus"] == 1
    assert d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - d["config"]["per_gpu_batch"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    if cfg == "c2":  # the driver's configuration: the CPU port of the same step is timed beside it
        c = d["cpu_baseline"]
        assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and c["sample"] and c["timed_steps"] == 3
        # every figure is measured in the run or says it is static: kernel names and traffic come from rocprofv3 passes over
        # `bench.py --kernel-probe` when rocprofv3 can run, from profiles/traffic.json (labelled) otherwise
        for key in ("roofline", "roofline_gather"):
            rr = d[key]
            assert rr["kernel_name_source"].startswith(("observed in this run", "static:")) and rr["traffic_source"].startswith(("measured in this run", "static:"))
            if rr["traffic_source"].startswith("measured"):
                assert rr["kernel_name_source"].startswith("observed") and "(" in rr["kernel"]  # a demangled signature from the trace
                alg = rr.get("algorithmic_bytes_per_launch")
                if alg:  # the gather's corrected counter traffic equals its known byte count (calibration of the x2 rule)
                    assert abs(rr["traffic"] - alg) < 0.02 * alg, (rr["traffic"], alg)
        fl = d["fit_loop"]  # the host loop + loader keep up with the device step (the timed region itself replays staged batches)
        assert fl["steps"] == 100 and fl["value"] > 0.8 * d["value"] and abs(fl["frac_of_value"] - fl["value"] / d["value"]) < 1e-9
        sp = d["split_precision"]  # the opt-in bf16x6 projections, timed in the same run, with their accuracy evidence beside them
        assert sp["value"] > 0 and sp["dtype"].startswith("f32 (bf16x6") and abs(sp["speedup_vs_exact"] - d["ms_per_step"] / sp["ms_per_step"]) < 1e-9
        e = sp["projection_max_abs_err_vs_fp64"]
        assert e["split_bf16x6"] < 4 * e["exact_fp32"] + 1e-6 and e["split_bf16x6"] < 1e-5 * max(1.0, sp["projection_ref_max_abs"])
        rg = d["roofline_gather"]
        assert rg["achievable_gbs"] >= max(rg["calibration"]["torch_copy_gbs"], rg["calibration"]["float4_row_copy_gbs"]) - 1e-6
        assert abs(rg["frac_of_achievable"] - rg["achieved"] / rg["achievable_gbs"]) < 1e-9 and "table_residency" in rg
        assert len(d["ms_per_step_repeats"]) == d["repeats"] and d["config"]["id_distribution"]["ids"] == "uniform"
        assert set(d["cpu_baseline"]["thread_sweep_ms_per_step"]) and d["cpu_baseline"]["cores"] >= 1
        st = d["roofline_step"]
        assert st["bound"] == "mfma" and abs(st["frac"] - st["achieved"] / st["peak"]) < 1e-9 and 0 < st["frac"] < d["roofline"]["frac"]
        assert abs(st["achieved"] * 1e12 - st["algorithmic_flops_per_step"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * st["achieved"] * 1e12
