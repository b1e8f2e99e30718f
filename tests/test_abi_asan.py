"""AddressSanitizer pass over the HOST side of the C-ABI layer ("ASAN host build of the C-ABI layer").

`make -C ebnerd-benchmark_amd/csrc asan` builds libebnerd_hip_asan.so (host code instrumented, device code untouched);
tests/abi_fuzz_driver.py then calls every entry point of include/ebnerd_hip.h ~300 times with NULL / negative / random
arguments (fake device pointers, real host structs, sizes from 0 to 2^62) in a subprocess that has the ASAN runtime preloaded
and NO visible GPU -- so the host code runs (argument checks, tile / split-K planners, workspace queries, launch geometry)
and no kernel can.  What it found when first run (round 4): an integer division by zero in the split-precision planner for
K > 2^35, and size queries that wrapped negative for extents past 2^31 -- now bounded by EBN_DIM_MAX / saturating."""
import ctypes
import glob
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "ebnerd-benchmark_amd" / "csrc"
ASAN_LIB = CSRC / "libebnerd_hip_asan.so"


def _asan_runtime():
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return hits[-1] if hits else None


@pytest.fixture(scope="module")
def asan_lib():
    rt = _asan_runtime()
    if rt is None or not Path("/opt/rocm/bin/hipcc").exists():
        pytest.skip("no ROCm clang ASAN runtime on this box")
    r = subprocess.run(["make", "-C", str(CSRC), "asan", "-j", str(min(8, os.cpu_count() or 1))], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    assert ASAN_LIB.exists()
    return rt


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_host_side_of_every_entry_point_is_clean_under_asan(asan_lib, seed):
    env = dict(os.environ, LD_PRELOAD=asan_lib, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", HIP_VISIBLE_DEVICES="-1",
               ROCR_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "abi_fuzz_driver.py"), str(ASAN_LIB), str(seed)], capture_output=True, text=True,
                       timeout=900, env=env)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0 and "ABI_FUZZ_OK" in r.stdout, tail
    assert "AddressSanitizer" not in r.stderr, tail
    n_calls = int(r.stdout.split("calls=")[1].split()[0])
    assert n_calls > 15000


def test_size_queries_bound_their_extents():
    """The regressions of the first ASAN pass, on the product library: extents past EBN_DIM_MAX (2^31 - 1) answer 0, products
    that cannot fit an int64 saturate instead of wrapping."""
    from ebrec import _hip

    L = _hip.lib()
    big = 1 << 40
    assert L.ebn_gemm_planes_workspace_floats(big, big, big) == 0  # (used to divide by zero: K / 16 slabs truncated to int 0)
    assert L.ebn_planes_bytes(big, 16) == 0 and L.ebn_gemm_workspace_floats(big, 8, 8) == 0
    assert L.ebn_shard_plan_workspace_ints(big, 8) == 0
    m = (1 << 31) - 1
    assert L.ebn_gemm_split_workspace_bytes(m, m, m) == (1 << 63) - 1  # saturated, not negative
    assert L.ebn_gemm_prec_workspace_bytes(m, m, m, 1) == (1 << 63) - 1
    # and the sizes the engine really asks for are unchanged
    assert L.ebn_planes_bytes(24000, 1024) == 3 * 24064 * 1024 * 2
    assert L.ebn_gemm_workspace_floats(1024, 1200, 24000) > 0
    assert L.ebn_attpool_partials_len(24000, 200) > 0 and L.ebn_user_head_partials_len(32, 200) == 32 * 2 * 200
