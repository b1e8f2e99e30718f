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
    if cfg == "c3":  # (round 5) the NRMSDocVec line carries the same evidence as the headline's: whole-step roofline, counter traffic, CPU leg
        st, c = d["roofline_step"], d["cpu_baseline"]
        assert st["bound"] == "mfma" and abs(st["frac"] - st["achieved"] / st["peak"]) < 1e-9 and 0 < st["frac"] < r["frac"]
        assert abs(st["achieved"] * 1e12 - st["flops_per_step"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * st["achieved"] * 1e12
        assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "CpuDocVecTrainer" in c["sample"]
        assert isinstance(st["launches_per_step"], int) and 8 <= st["launches_per_step"] <= 40 and "counted" in st["launches_note"]
        assert ("tn_finale" in r["kernel"] or "tn_group" in r["kernel"]) and r["algorithmic_flops_per_launch"] > 1e9
        for key in ("roofline", "roofline_gather"):
            assert d[key]["traffic_source"].startswith(("measured in this run", "not measured"))
            if d[key]["traffic_source"].startswith("measured"):
                assert d[key]["traffic"] > 0
        assert d["oracle_pin"].startswith(("unpinned", "pinned")) and isinstance(d["env"], dict)
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
        assert isinstance(st["launches_per_step"], int) and 10 <= st["launches_per_step"] <= 60
        # (round 6, verdict r5 item 1) the driver's ONE command -- no --config -- times every other BASELINE config as a leg of the line
        legs = d["legs"]
        assert set(legs) == {"c1", "c3", "c4", "c5"}
        want = {"c1": ("configs[0]", 32, "32000x300"), "c3": ("configs[2]", 32, "NRMSDocVec"), "c4": ("configs[3]", 32, "history_size=50"),
                "c5": ("configs[4]", 64, "row-sharded")}
        for name, (idx, pb, word) in want.items():
            leg = legs[name]
            assert "error" not in leg and "skipped" not in leg, leg
            assert leg["value"] > 0 and leg["n_gpus"] == 1 and leg["dtype"] == "f32" and leg["launch"] == "hipGraph replay"
            assert idx in leg["config"]["workload"] and word in leg["config"]["workload"] and leg["config"]["per_gpu_batch"] == pb
            assert abs(leg["value"] - pb / (leg["ms_per_step"] * 1e-3)) < 1e-6 * leg["value"] and len(leg["ms_per_step_repeats"]) == 2
            lr, ls = leg["roofline"], leg["roofline_step"]
            assert lr["bound"] == "mfma" and 0 < lr["frac"] < 1 and lr["avg_launch_us"] > 0 and abs(lr["frac"] - lr["achieved"] / lr["peak"]) < 1e-9
            assert ls["bound"] == "mfma" and 0 < ls["frac"] < lr["frac"] and isinstance(ls["launches_per_step"], int)
            assert leg["roofline_gather"]["bound"] == "hbm" and leg["roofline_gather"]["avg_launch_us"] > 0
            if lr["traffic_source"].startswith("measured"):  # rocprofv3 could run: the kernel name is the one the profiler saw
                assert "(" in lr["kernel"] and lr["traffic"] > 0
        assert legs["c3"]["cpu_baseline"]["kind"] == "port" and legs["c3"]["cpu_baseline"]["value"] > 0 and "tn_finale" in legs["c3"]["roofline"]["kernel"]
        assert "cpu_baseline" not in legs["c1"] and legs["c5"]["config"]["global_batch"] == 64


def _bench(argv, timeout, env=None):
    import os
    import time

    t0 = time.time()
    out = subprocess.run([sys.executable, str(ROOT / "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, cwd=str(ROOT),
                         env=dict(os.environ, **(env or {})))
    return out, time.time() - t0


@pytest.mark.gpu
def test_multi_rank_line_carries_the_c4_and_c5_legs_and_the_communication_librarys_view():
    """Verdict r4 item 1: the driver's ONE command (`bench.py --gpus N`, no --config) must yield all the multi-GPU evidence.  Two ranks
    sharing this box's GPU over gloo (RCCL refuses two ranks on one device): a functional dry run of exactly that command."""
    out, _ = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "2"], 1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and "configs[1]" in d["config"]["workload"] and d["value"] > 0
    v = d["rccl_view"]
    assert v["world_size"] == 2 and v["rank_ids_allgathered"] == [0, 1] and [r["rank"] for r in v["ranks"]] == [0, 1]
    assert v["backend"] in ("gloo", "nccl") and (v["backend"] == "gloo" or v["distinct_devices"] == 2)
    assert d["guard"].startswith("active") and d["oracle_pin"].startswith(("unpinned", "pinned")) and isinstance(d["env"], dict)
    assert "comm_exposed_us" in d and d["allreduce_bytes_per_step"] > 0
    legs = d["legs"]
    assert set(legs) == {"c4", "c5"}
    for name, idx in (("c4", 3), ("c5", 4)):
        leg = legs[name]
        assert "error" not in leg, leg
        assert leg["value"] > 0 and leg["n_gpus"] == 2 and f"configs[{idx}]" in leg["config"]["workload"]
        assert abs(leg["value"] - 2 * leg["config"]["per_gpu_batch"] / (leg["ms_per_step"] * 1e-3)) < 1e-6 * leg["value"]
        assert "comm_exposed_us" in leg and leg["allreduce_bytes_per_step"] > 0
    assert legs["c4"]["config"]["per_gpu_batch"] == 32 and "history_size=50" in legs["c4"]["config"]["workload"]
    assert legs["c5"]["config"]["per_gpu_batch"] == 64 and legs["c5"]["exchange"]["world"] == 2 and "row-sharded" in legs["c5"]["config"]["workload"]


@pytest.mark.gpu
def test_the_multi_rank_branch_runs_end_to_end_on_a_one_rank_rccl_group():
    """Verdict r5 item 5: every line of bench.py's `nccl` branch -- torch.distributed.run, init_process_group("nccl", device_id=...),
    rccl_view, SegmentTrace, HangWatchdog, the engine's multi-rank launch form (graph | all-reduce | graph), run_legs with its fresh
    rendezvous, the closing collective flag check -- executes HERE, on a group of one rank (RCCL refuses two ranks on one device),
    before the 8-GPU node runs it; and costs the step almost nothing (the collectives are identities)."""
    plain, _ = _bench(["--steps", "20", "--warmup", "5", "--repeats", "3", "--legs", "", "--no-cpu-baseline", "--no-fit-loop", "--no-split-leg", "--no-probe"], 900)
    assert plain.returncode == 0, plain.stderr[-3000:]
    ref = json.loads([l for l in plain.stdout.splitlines() if l.startswith("{")][0])
    out, _ = _bench(["--gpus", "1", "--force-dist", "--steps", "20", "--warmup", "5", "--repeats", "3"], 1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    v = d["rccl_view"]
    assert d["backend"] == "nccl" and v["backend"] == "nccl" and v["rccl_version"] and not str(v["rccl_version"]).startswith("unavailable")
    assert v["world_size"] == 1 and v["rank_ids_allgathered"] == [0] and v["distinct_devices"] == 1
    assert d["guard"].startswith("active") and "force-dist" in d["dist_note"] and d["n_gpus"] == 1 and d["ranks"] == 1
    assert "comm_exposed_us" in d and d["allreduce_bytes_per_step"] > 0 and d["comm"]["overlap"] in (True, False)
    assert set(d["legs"]) == {"c4", "c5"}
    for name in ("c4", "c5"):
        assert "error" not in d["legs"][name], d["legs"][name]
        assert d["legs"][name]["value"] > 0 and d["legs"][name]["backend"] == "nccl"
    assert d["legs"]["c5"]["exchange"]["world"] == 1
    # one graph boundary with an (identity) all-reduce and the separate optimizer launch of the multi-rank form: measured +2.1 % (1.288 against 1.262 ms)
    assert abs(d["value"] - ref["value"]) < 0.05 * ref["value"], (d["value"], ref["value"])


@pytest.mark.gpu
def test_a_rank_that_skips_its_collectives_ends_the_benchmark_non_zero_and_names_the_segment():
    """The hang watchdog: rank 1 leaves rank 0 alone in the gradient all-reduce.  Without it the run would sit in the collective
    until the driver's own limit; with it the job exits non-zero within EBN_COLLECTIVE_TIMEOUT_S and says where."""
    out, dt = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "2", "--legs", "", "--fault-skip-collectives-on-rank", "1"], 900,
                     env={"EBN_COLLECTIVE_TIMEOUT_S": "20"})
    assert out.returncode != 0, out.stdout[-2000:]
    assert dt < 400, dt
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]  # no line of record from a broken run
    assert "HUNG" in out.stderr and "segment" in out.stderr and "phase [c2: warm-up + timed region" in out.stderr, out.stderr[-3000:]
